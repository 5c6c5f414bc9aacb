"""Test pass and mask export of a trained model, with the reference's call signatures
(/root/reference/result_utils/save_img.py: naive_test :17-86, test_pred :88-194, save_pred_one :197-265).

What is kept: the checkpoint lookup ("epoch" in the file name, `checkpoint['state_dict']`), eval mode, the
metric/loss average over the test loader, the returned (image, target, prediction) batch, and the arg-max masks.
What is not: test-time augmentation, CRF post-processing and the Hausdorff distance (third-party post-processing
outside the network's arithmetic); figures are written with matplotlib when it is importable, and the masks are
always written as .npy next to them.
"""
import os
import random

import numpy as np
import torch

from loss.dice import dice_coefficient


def _device_of(model):
    return next(model.parameters()).device


def naive_test(testloader, model, loss_func=None, postpr="no", save_incorr=False):
    """Mean of `loss_func(outputs, target)` over the loader in eval mode (default: arg-max Dice)."""
    if postpr not in ("no", "tta", None):
        raise NotImplementedError("CRF / test-time-augmentation post-processing is not part of this build")
    if loss_func is None:
        loss_func = dice_coefficient
    total, count = 0.0, 0
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    model = model.to(device).eval()
    with torch.no_grad():
        for images, segm in testloader:
            outputs = model(images.to(device)).detach()
            total += float(loss_func(outputs, segm.to(device)))
            count += 1
    name = "Dice coefficient" if loss_func is dice_coefficient else "Loss"
    mean = total / max(count, 1)
    print(f"{name} of the network on test images:{mean:.4f}")
    return mean


def test_pred(model, saved_model_dir, test_loader, experiment_name="", trainer=None, print_test_acc=True, loss_func=None,
              sgm_train=True):
    """Load the checkpoint found in `saved_model_dir`, run the test pass and return one (image, target, prediction) batch
    (segmentation) or the list of (target, predicted class) pairs (classification / regression heads)."""
    path = None
    for filename in sorted(os.listdir(saved_model_dir)):
        file_path = os.path.join(saved_model_dir, filename)
        if os.path.isfile(file_path) and "epoch" in file_path:
            path = file_path
            break
    if path is None:
        raise FileNotFoundError(f"no checkpoint with 'epoch' in its name under {saved_model_dir}")
    checkpoint = torch.load(path, map_location="cpu", weights_only=True)
    model.load_state_dict(checkpoint["state_dict"])
    model.eval()
    if trainer is not None:
        trainer.test(model)
    elif print_test_acc:
        naive_test(test_loader, model, loss_func)
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    model = model.to(device)
    if sgm_train:
        n_batches = int(len(test_loader.dataset) / test_loader.batch_size) if len(test_loader.dataset) else 0
        stop = random.randint(0, n_batches)
        x = y = None
        for i, batch in zip(range(stop + 1), test_loader):
            x, y = batch
        with torch.no_grad():
            pred = model(x.to(device)).detach()
        return x, y, pred.cpu()
    pairs = []
    with torch.no_grad():
        for x, y in test_loader:
            pred = model(x.to(device)).detach()
            if pred.dim() > 1 and pred.shape[-1] != 1:
                pred = torch.argmax(pred, dim=1)
            pairs.append((y.clone().cpu().squeeze().numpy(), pred.to(torch.int).cpu().squeeze().numpy()))
            print(f"actual and predicted mid slices (m, l) are {pairs[-1][0]}, {pairs[-1][1]}")
    return pairs


def save_pred_one(x, y, pred_sgm, saved_res_dir, bat_one=False, fig_name="res", nosft=False, channelcnt=None):
    """Write the arg-max mask of the batch (`<fig_name>_mask.npy`, uint8) and, with matplotlib, the reference's
    image / target / prediction figure of one random batch item.  Returns (mask array, Dice coefficient)."""
    os.makedirs(saved_res_dir, exist_ok=True)
    dice = float(dice_coefficient(pred_sgm, y, nosft=nosft, channelcnt=channelcnt))
    mask = pred_sgm.argmax(1) if pred_sgm.dim() == 4 and pred_sgm.shape[1] > 1 else pred_sgm.squeeze(1)
    mask = mask.to(torch.uint8).cpu().numpy()
    np.save(os.path.join(saved_res_dir, f"{fig_name}_mask.npy"), mask)
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except Exception:           # figures are optional; the masks above are the product
        return mask, dice
    target = y.argmax(1) if y.dim() == 4 and y.shape[1] > 1 else y.reshape(mask.shape)
    k = random.randint(0, x.shape[0] - 1)
    fig, ax = plt.subplots(1, 3)
    fig.set_size_inches(18, 10)
    for a, title, img in zip(ax, ("Image", "Actual Segmentation Mask", f"Predicted Mask, dice score:{dice:.4f}"),
                             (x[k, 0].cpu().numpy(), target[k].cpu().numpy(), mask[k])):
        a.set_title(title)
        a.axis("off")
        a.imshow(img)
    fig.savefig(os.path.join(saved_res_dir, f"{fig_name}.png"), bbox_inches="tight")
    plt.close(fig)
    return mask, dice
