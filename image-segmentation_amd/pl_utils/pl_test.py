"""`test_model`: reload the best checkpoint and run the Lightning test loop (/root/reference/pl_utils/pl_test.py:3-26;
imported as `from pl_utils.pl_test import test_model` by /root/reference/detect_region/common.py:13)."""
import os

import torch


def test_model(model, checkpoint_path, trainer):
    """Reload the saved best checkpoint (first file with "epoch" in its name) and run the test loop."""
    if trainer is None:
        return None
    for name in sorted(os.listdir(checkpoint_path)):
        path = os.path.join(checkpoint_path, name)
        if os.path.isfile(path) and "epoch" in name:
            checkpoint = torch.load(path, map_location="cpu", weights_only=True)
            model.load_state_dict(checkpoint['state_dict'])
            break
    return trainer.test(model)
