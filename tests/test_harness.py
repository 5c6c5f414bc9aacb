"""The loop either side of the hot path: `train_general` / `naive_train` with the reference's semantics
(new Adam per epoch, validation every `valid_after` epochs, checkpoint dict) on synthetic slices."""
import os

import numpy as np
import pytest
import torch


def _run(device, tmp_path, side, epochs, fused=None):
    from data_read.synthetic import SyntheticSlices
    from loss.dice import dice_coefficient, dice_loss_sq
    from loss.focal import focal_loss_with_logits
    from models.unetvgg import UNetVgg
    from oracle.ref_models import hash_init_
    from train_helper.common import train_general
    torch.manual_seed(20)
    data = SyntheticSlices(num_img=20, side=side, n_classes=4)
    model = hash_init_(UNetVgg(out_classes=4))
    out = train_general(data, str(tmp_path), expt_name="t", val_set_perc=0.2, test_set_perc=0.1, device=torch.device(device),
                        num_epochs=epochs, is_naive_train=True, valid_after=2, net_model=model, batch_size=4, test_batch_size=2,
                        loss_criterion=[dice_loss_sq, focal_loss_with_logits], val_acc_func=dice_coefficient, lr=1e-3)
    return out, out.segk_metrics


def test_train_general_cpu(tmp_path):
    model, m = _run("cpu", tmp_path, side=64, epochs=3)
    assert len(m["train_loss"]) == 3 and len(m["val_loss"]) == 2 and len(m["val_acc"]) == 2    # epochs 0 and 2 validate
    assert m["train_loss"][-1] < m["train_loss"][0]
    assert 0.0 <= m["test_acc"] <= 1.0
    names = [n for n in os.listdir(tmp_path) if "epoch" in n]
    assert len(names) == 1 and names[0].startswith("checkpoint_epoch") and names[0].endswith("_t.pth.tar")
    state = torch.load(os.path.join(tmp_path, names[0]), weights_only=True)
    assert set(state) == {"epoch", "arch", "state_dict"} and state["arch"] == "t"
    from models.unetvgg import UNetVgg
    UNetVgg(out_classes=4).load_state_dict(state["state_dict"], strict=True)


def test_lightning_branch_cpu(tmp_path):
    """is_naive_train=False: SegmModelPL + train_model/test_model on the lightning stand-in (or the real package)."""
    from data_read.synthetic import SyntheticSlices
    from loss.dice import dice_coefficient, dice_loss_sq
    from models.unetvgg import UNetVgg
    from oracle.ref_models import hash_init_
    from train_helper.common import train_general
    torch.manual_seed(20)
    mod = train_general(SyntheticSlices(num_img=16, side=64), str(tmp_path), expt_name="pl", val_set_perc=0.25, test_set_perc=0.125,
                        device=torch.device("cpu"), num_epochs=2, is_naive_train=False, net_model=hash_init_(UNetVgg(out_classes=4)),
                        batch_size=4, test_batch_size=2, loss_criterion=[dice_loss_sq], val_acc_func=dice_coefficient, lr=1e-3)
    m = mod.segk_metrics
    assert {"loss", "val_loss", "val_acc", "test_loss", "test_acc"} <= set(m)
    names = [n for n in os.listdir(tmp_path) if "epoch" in n]
    assert len(names) == 1
    state = torch.load(os.path.join(tmp_path, names[0]), weights_only=True)
    assert all(k.startswith("model.") for k in state["state_dict"])      # keys of the wrapping module, as in the reference


def test_mid_slice_stand_in():
    from data_read.read_3dinput import Meniscus3DMidSlice
    ds = Meniscus3DMidSlice("/nonexistent", num_img=3, in_channels=1, depth=16, side=32)
    vol, mid = ds[1]
    assert vol.shape == (1, 16, 32, 32) and mid.dtype == torch.int64 and 0 <= int(mid) < 80 and len(ds) == 3


@pytest.mark.gpu
def test_train_general_gpu(tmp_path):
    from segk import _lib
    before = _lib.launches
    model, m = _run("cuda", tmp_path, side=64, epochs=3)
    assert _lib.launches - before > 1000, "the HIP library did not run the training loop"
    assert m["train_loss"][-1] < m["train_loss"][0]
    assert all(torch.isfinite(p).all() for p in model.parameters())
    names = [n for n in os.listdir(tmp_path) if "epoch" in n]
    assert len(names) == 1
    state = torch.load(os.path.join(tmp_path, names[0]), weights_only=True)
    # the checkpoint written from the flat GPU parameters loads into a fresh CPU model (reference: save_img.py:122-143)
    from models.unetvgg import UNetVgg
    cpu_model = UNetVgg(out_classes=4)
    cpu_model.load_state_dict(state["state_dict"], strict=True)
    x = torch.randn(1, 3, 64, 64)
    with torch.no_grad():
        a = cpu_model.eval()(x)
        model.load_state_dict(state["state_dict"])
        b = model.eval()(x.cuda()).cpu()
    assert float((a - b).norm() / a.norm()) < 1e-3


def _export_roundtrip(device, tmp_path):
    from torch.utils.data import DataLoader
    from data_read.synthetic import SyntheticSlices
    from models.unetvgg import UNetVgg
    from result_utils.save_img import save_pred_one, test_pred
    from train_helper.naive_train import save_checkpoint
    torch.manual_seed(3)
    model = UNetVgg(out_classes=4).to(device)
    save_checkpoint({"epoch": 1, "arch": "UNetVgg", "state_dict": model.state_dict()}, True, str(tmp_path))
    loader = DataLoader(SyntheticSlices(4, side=64, n_classes=4), batch_size=2)
    fresh = UNetVgg(out_classes=4)
    x, y, pred = test_pred(fresh, str(tmp_path), loader, "t")
    assert pred.shape == (2, 4, 64, 64) and not fresh.training
    mask, dice = save_pred_one(x, y, pred, str(tmp_path), fig_name="out1")
    stored = np.load(os.path.join(tmp_path, "out1_mask.npy"))
    assert stored.dtype == np.uint8 and np.array_equal(stored, pred.argmax(1).numpy()) and 0.0 <= dice <= 1.0
    return x, pred


def test_mask_export_cpu(tmp_path):
    _export_roundtrip("cpu", tmp_path)


@pytest.mark.gpu
def test_mask_export_gpu(tmp_path):
    """test_pred on the GPU: the exported arg-max masks equal those of the same checkpoint evaluated on the CPU path
    wherever the top-2 logit margin exceeds the fp32 tolerance."""
    from models.unetvgg import UNetVgg
    x, pred = _export_roundtrip("cuda", tmp_path)
    name = [n for n in os.listdir(tmp_path) if "epoch" in n][0]
    cpu = UNetVgg(out_classes=4)
    cpu.load_state_dict(torch.load(os.path.join(tmp_path, name), weights_only=True)["state_dict"])
    with torch.no_grad():
        ref = cpu.eval()(x)
    top2 = ref.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 1e-3
    assert torch.equal(pred.argmax(1)[safe], ref.argmax(1)[safe]) and float(safe.float().mean()) > 0.9


@pytest.mark.gpu
def test_mid_slice_training_gpu(tmp_path):
    """The call of train_midlslice.py:25-33 (DetectMid, CrossEntropyLoss, naive loop, sgm_train=False) on stand-in
    volumes: the HIP path trains, writes the reference's checkpoint layout, and the checkpoint loads into a CPU model."""
    from torch import nn
    from data_read.read_3dinput import Meniscus3DMidSlice
    from models.resnt import DetectMid
    from segk import _lib
    from train_helper.common import train_general
    torch.manual_seed(20)
    dataset = Meniscus3DMidSlice("/nonexistent", num_img=10, in_channels=1, depth=32, side=128)
    model = DetectMid(in_classes=1)
    before = _lib.launches
    out = train_general(dataset, str(tmp_path), expt_name="train_midlsl", val_set_perc=0.2, device=torch.device("cuda"), num_epochs=3,
                        is_naive_train=True, valid_after=1, net_model=model, batch_size=4, test_set_perc=0.0,
                        loss_criterion=[nn.CrossEntropyLoss()], val_acc_func=nn.CrossEntropyLoss(), sgm_train=False, test_batch_size=1,
                        lr=1e-4)
    m = out.segk_metrics
    assert _lib.launches - before > 5000, "the HIP library did not run the training loop"
    assert all(np.isfinite(v) for v in m["train_loss"]) and m["train_loss"][-1] < m["train_loss"][0]
    names = [n for n in os.listdir(tmp_path) if "epoch" in n]
    assert len(names) == 1
    state = torch.load(os.path.join(tmp_path, names[0]), weights_only=True)
    cpu_model = DetectMid(in_classes=1)
    cpu_model.load_state_dict(state["state_dict"], strict=True)
    x, _ = dataset[0]
    with torch.no_grad():
        a = cpu_model.eval()(x[None])
        b = out.eval()(x[None].cuda()).cpu()
    assert float((a - b).norm() / a.norm()) < 1e-3


def test_pl_test_module_path_matches_the_reference():
    """detect_region/common.py:13 of the reference does `from pl_utils.pl_test import test_model`."""
    from pl_utils.pl_test import test_model
    from pl_utils.pl_train import test_model as same
    assert test_model is same and test_model(None, "/nonexistent", None) is None


def test_reference_callers_pin_is_present_and_green():
    """oracle/run_reference_callers.py executed the reference's OWN pl_utils/pl_train.py + pl_utils/pl_mod.py and train_midlslice.py
    (unedited, from the reference tree) on this build's modules in the build container; its record must say that both ran."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = json.load(open(os.path.join(root, "tests", "golden", "callers_pin.json")))
    assert doc["all_ran"] is True and set(doc["legs"]) == {"pl_train.py + pl_mod.py", "train_midlslice.py"}
    pl_leg, mid = doc["legs"]["pl_train.py + pl_mod.py"], doc["legs"]["train_midlslice.py"]
    assert pl_leg["ran"] and {"loss", "val_loss", "val_acc"} <= set(pl_leg["metrics"]) and any(f.endswith(".ckpt") for f in pl_leg["files_written"])
    assert mid["ran"] and mid["executed_as"].startswith("__main__") and mid["train_general_called_with"]["net_model"] == "DetectMid"
    assert mid["train_general_called_with"]["num_epochs"] == 2000 and mid["files_written"] == ["checkpoint_epoch1_train_midlsl.pth.tar"]
    pins = json.load(open(os.path.join(root, "tests", "golden", "oracle_pin.json")))["checks"]
    assert pins["callers/train_midlslice.py"] is True and pins["callers/pl_train.py + pl_mod.py"] is True


def test_train_general_environment_knobs(tmp_path, monkeypatch):
    """SEGK_MAX_EPOCHS caps the epoch count and SEGK_PATH_REMAP re-roots the output directory (the knobs that let the reference's
    entry scripts, with their 2000 epochs and hard-coded home directories, run unedited)."""
    import torch
    from data_read.read_3dinput import Meniscus3DMidSlice
    from data_read.synthetic import SyntheticSlices
    from models.unetvgg import UNetSimple
    from train_helper.common import train_general
    monkeypatch.setenv("SEGK_MAX_EPOCHS", "1")
    monkeypatch.setenv("SEGK_PATH_REMAP", f"/nonexistent/home={tmp_path}")
    torch.manual_seed(0)
    net = train_general(SyntheticSlices(num_img=8, side=32, n_classes=3, in_channels=1), "/nonexistent/home/run", num_epochs=50,
                        is_naive_train=True, net_model=UNetSimple(in_classes=1, out_classes=3, channelscale=8), batch_size=4,
                        val_set_perc=0.25, test_set_perc=0.0, device="cpu")
    assert len(net.segk_metrics["train_loss"]) == 1 and os.path.isdir(tmp_path / "run")
    monkeypatch.setenv("SEGK_SYNTH_3D", "5,8,32")
    d = Meniscus3DMidSlice("/anywhere", num_img=None, in_channels=1)
    assert len(d) == 5 and tuple(d[0][0].shape) == (1, 8, 32, 32)
