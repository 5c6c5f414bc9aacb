"""Do the reference's own callers run UNCHANGED on top of this build's modules?  (test infrastructure; build container only:
it executes files under /root/reference, which never travels to the GPU box.)

Two legs, both on the CPU with tiny synthetic data:

 1. /root/reference/pl_utils/pl_mod.py (`SegmModelPL`, :6-119) and /root/reference/pl_utils/pl_train.py (`train_model`, :6-45)
    are loaded as FILES from the reference tree and driven with the build's `models.unetvgg.UNetVgg`, `loss.dice.dice_loss_sq`,
    `loss.focal.focal_loss_with_logits`, `loss.dice.dice_coefficient`, on `pytorch_lightning` = segk_compat.lightning (the real
    package is not installed here; the reference targets its 0.7/0.8 API).
 2. /root/reference/train_midlslice.py is executed as `__main__` (runpy) with the build's package directory first on sys.path,
    so that its `data_read.read_3dinput`, `models.resnt` and `train_helper.common` imports (:4-6) resolve to the build.  The
    script is not edited: its hard-coded home directory (:16-17) is re-rooted and its 2000 epochs (:24) are capped through the
    environment knobs of the build's `train_general`; the synthetic stand-in of its private data set is sized by SEGK_SYNTH_3D.

Writes tests/golden/callers_pin.json (what ran, final metrics, files written) and adds `callers/*` entries to
tests/golden/oracle_pin.json.

    PYTHONDONTWRITEBYTECODE=1 python oracle/run_reference_callers.py
"""
import importlib.util
import json
import os
import runpy
import sys
import tempfile
from unittest import mock

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "image-segmentation_amd")
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, PKG)

import torch  # noqa: E402

torch.set_num_threads(8)


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def lightning_leg(tmp):
    from segk_compat import lightning
    stand_in = lightning.install()
    import pytorch_lightning as pl
    ref_mod = _load("reference_pl_mod", os.path.join(REF, "pl_utils", "pl_mod.py"))
    ref_train = _load("reference_pl_train", os.path.join(REF, "pl_utils", "pl_train.py"))
    assert ref_mod.__file__.startswith(REF) and ref_train.__file__.startswith(REF)
    from data_read.synthetic import SyntheticSlices
    from loss.dice import dice_coefficient, dice_loss_sq
    from loss.focal import focal_loss_with_logits
    from models.unetvgg import UNetVgg
    import models.unetvgg
    assert models.unetvgg.__file__.startswith(PKG)
    torch.manual_seed(20)
    data = SyntheticSlices(num_img=10, side=64, n_classes=4, in_channels=3)
    train, val, test = torch.utils.data.random_split(data, [6, 2, 2])
    out = os.path.join(tmp, "lightning")
    os.makedirs(out)
    open(os.path.join(out, "stale.txt"), "w").write("x")            # train_model must clear the directory (pl_train.py:19-31)
    model = UNetVgg(out_classes=4)
    module = ref_mod.SegmModelPL(model, train, val, test, 3, 2, loss_criterion=[dice_loss_sq, focal_loss_with_logits],
                                 val_acc_func=dice_coefficient, lr=1e-3, experiment_name="pin")
    before = {k: v.clone() for k, v in model.state_dict().items()}
    trainer = ref_train.train_model(module, out, device=torch.device("cpu"), num_epochs=2)
    metrics = {k: float(v) for k, v in trainer.callback_metrics.items()}
    moved = sum(int(not torch.equal(v, before[k])) for k, v in model.state_dict().items())
    files = sorted(os.listdir(out))
    ckpt = [f for f in files if f.endswith(".ckpt")]
    state = torch.load(os.path.join(out, ckpt[0]), map_location="cpu", weights_only=True) if ckpt else {}
    ok = ("stale.txt" not in files and len(ckpt) == 1 and set(state.get("state_dict", {})) == {"model." + k for k in before}
          and all(k in metrics for k in ("loss", "val_loss", "val_acc")) and all(v == v for v in metrics.values()) and moved > 50)
    return {"ran": bool(ok), "reference_files": ["pl_utils/pl_mod.py", "pl_utils/pl_train.py"],
            "pytorch_lightning": "segk_compat.lightning stand-in" if getattr(pl, "__segk_stand_in__", False) else pl.__version__,
            "stand_in_installed_now": bool(stand_in), "epochs": 2, "metrics": metrics, "files_written": files,
            "state_dict_entries_changed": moved}


def midslice_leg(tmp):
    old_root = "/home/students/thampi/PycharmProjects"
    env = {"SEGK_MAX_EPOCHS": "1", "SEGK_PATH_REMAP": f"{old_root}={tmp}", "SEGK_SYNTH_3D": "8,16,64"}
    real_makedirs = os.makedirs
    made = []

    def makedirs(path, *a, **k):                                     # the script creates its output directory itself (:21-22)
        path = str(path)
        if path.startswith(old_root):
            path = tmp + path[len(old_root):]
        made.append(path)
        return real_makedirs(path, *a, **k)

    seen = {}
    import train_helper.common as TH
    assert TH.__file__.startswith(PKG)
    real_train_general = TH.train_general

    def spy(*a, **k):
        seen["kwargs"] = {kk: (type(v).__name__ if not isinstance(v, (int, float, str, bool, type(None))) else v) for kk, v in k.items()}
        seen["model"] = real_train_general(*a, **k)
        return seen["model"]

    with mock.patch.dict(os.environ, env), mock.patch("os.makedirs", makedirs), mock.patch.object(TH, "train_general", spy):
        runpy.run_path(os.path.join(REF, "train_midlslice.py"), run_name="__main__")
    model = seen["model"]
    import models.resnt
    assert models.resnt.__file__.startswith(PKG)
    metrics = {k: v for k, v in model.segk_metrics.items()}
    out_dir = os.path.join(tmp, "MA_Praise", "outputs", "train_midlsl")
    files = sorted(os.listdir(out_dir))
    ok = (type(model).__name__ == "DetectMid" and len(metrics["train_loss"]) == 1 and len(metrics["val_loss"]) == 1
          and all(v == v for v in metrics["train_loss"] + metrics["val_loss"]) and any(f.startswith("checkpoint_epoch1_train_midlsl") for f in files)
          and seen["kwargs"].get("batch_size") == 12 and seen["kwargs"].get("num_epochs") == 2000)
    return {"ran": bool(ok), "reference_files": ["train_midlslice.py"], "executed_as": "__main__ via runpy, unedited",
            "environment_knobs": env | {"SEGK_PATH_REMAP": f"{old_root}=<temporary directory>"},
            "train_general_called_with": seen["kwargs"], "metrics": {k: v for k, v in metrics.items() if k != "checkpoint"},
            "files_written": files, "directories_created_by_the_script": [d.replace(tmp, "<tmp>") for d in made]}


def main():
    with tempfile.TemporaryDirectory() as tmp:
        legs = {"pl_train.py + pl_mod.py": lightning_leg(tmp), "train_midlslice.py": midslice_leg(tmp)}
    doc = {"torch": torch.__version__, "all_ran": all(leg["ran"] for leg in legs.values()), "legs": legs}
    with open(os.path.join(GOLD, "callers_pin.json"), "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    pf = os.path.join(GOLD, "oracle_pin.json")
    pin = json.load(open(pf))
    for name, leg in legs.items():
        pin["checks"]["callers/" + name] = leg["ran"]
    with open(pf, "w") as f:
        json.dump(pin, f, indent=1, sort_keys=True)
    print(json.dumps(doc, indent=1, sort_keys=True))
    if not doc["all_ran"]:
        raise SystemExit("a reference caller did not run on the build's modules")


if __name__ == "__main__":
    main()
