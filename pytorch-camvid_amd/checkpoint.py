"""Checkpoint files compatible with the reference's.

  save_checkpoint   <- torch.save(net.state_dict(), '<folder>/<run>/{epoch}-{best|regular}.pth')   train.py:232-240
  load_checkpoint   <- net.load_state_dict(torch.load(weight_path))                               train.py:88-93
  latest_checkpoint <- utils.get_weight_path: newest '*-best.pth' / '*-regular.pth' by mtime     utils.py:95-145
  checkpoint_epoch  <- int(re.search('([0-9]+)-(best|regular).pth', path).group(1))              train.py:111-113
  resume            <- the -resume branch of train.py:88-93 + the scheduler fast-forward :111-114
  save_policy       <- the end-of-epoch rule train.py:232-240 (best after half the epochs, else every SAVE_EPOCH)

A file written here is a plain `state_dict` (161 keys UNet / 182 SegNet, float32, `num_batches_tracked` int64) whose
conv weights are dense OIHW — the byte layout `torch.save` produces for the reference's own modules — so either side
loads the other's files with `load_state_dict(torch.load(path))`.  Loading copies INTO the module's parameters, whose
physical layout ([Cout][3][3][Cin], what the kernels read) therefore survives.  Optimizer state is not part of the
file, exactly as in the reference.
"""
import glob
import os
import re
import warnings

import torch

SAVE_EPOCH = 10                      # reference conf/settings.py:17
_NAME = re.compile(r"([0-9]+)-(best|regular)\.pth$")


def reference_state_dict(net):
    """state_dict with every tensor dense in its logical shape (conv weights OIHW-contiguous), detached clones."""
    out = type(net.state_dict())()
    for k, v in net.state_dict().items():
        out[k] = v.detach().clone(memory_format=torch.contiguous_format)
    return out


def save_checkpoint(net, folder, epoch, kind):
    """Write `<folder>/{epoch}-{kind}.pth` (kind 'best' | 'regular'; train.py:236,240) and return the path.  `folder`
    is the run's checkpoint directory (the reference uses checkpoints/<TIME_NOW>/)."""
    if kind not in ("best", "regular"):
        raise ValueError("kind must be 'best' or 'regular'")
    os.makedirs(folder, exist_ok=True)
    path = os.path.join(folder, f"{int(epoch)}-{kind}.pth")
    torch.save(reference_state_dict(net), path)
    return path


def load_checkpoint(net, path, map_location=None, strict=True):
    """net.load_state_dict(torch.load(path)) (train.py:92).  Works before or after net.cuda(); returns the key report."""
    try:
        sd = torch.load(path, map_location=map_location, weights_only=True)
    except TypeError:                                   # very old torch: no weights_only
        sd = torch.load(path, map_location=map_location)
    return net.load_state_dict(sd, strict=strict)


def latest_checkpoint(checkpoint_folder):
    """utils.get_weight_path: among `<checkpoint_folder>/*/*.pth`, the most recently modified file named
    '<int>-best.pth' or '<int>-regular.pth'; '' when there is none."""
    checkpoint_folder = os.path.abspath(checkpoint_folder)
    files = [f for f in glob.glob(os.path.join(checkpoint_folder, "*", "*.pth")) if _NAME.search(os.path.basename(f))]
    if not files:
        return ""
    return max(files, key=os.path.getmtime)


def checkpoint_epoch(path):
    """Epoch encoded in a checkpoint file name (train.py:111-113)."""
    m = _NAME.search(os.path.basename(path))
    if m is None:
        raise ValueError(f"{path!r} is not named '<epoch>-best.pth' or '<epoch>-regular.pth'")
    return int(m.group(1))


def resume(net, checkpoint_folder, scheduler=None, steps_per_epoch=None, map_location=None):
    """The `-resume` path of train.py:88-93,111-114: load the newest checkpoint under `checkpoint_folder` into `net`
    and, when a scheduler is given, move it to step trained_epochs * steps_per_epoch the way the reference does
    (`train_scheduler.step(n)`).  Returns (trained_epochs, path); (0, '') when nothing is found (the reference would
    fail on the empty path: `torch.load('')`)."""
    path = latest_checkpoint(checkpoint_folder)
    if not path:
        return 0, ""
    load_checkpoint(net, path, map_location=map_location)
    trained = checkpoint_epoch(path)
    if scheduler is not None:
        if steps_per_epoch is None:
            raise ValueError("resume(): steps_per_epoch is needed to fast-forward the scheduler")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")             # the epoch argument of step() is deprecated upstream; same call as the reference
            scheduler.step(trained * steps_per_epoch)
    return trained, path


def save_policy(net, folder, epoch, miou, best_iou, total_epochs, save_epoch=SAVE_EPOCH):
    """End-of-epoch rule of train.py:232-240: a new best mIoU after half the epochs writes '{epoch}-best.pth' (and
    nothing else that epoch); otherwise every `save_epoch`-th epoch writes '{epoch}-regular.pth'.
    Returns (best_iou, written path or None).  (The reference re-zeroes best_iou every epoch, train.py:176 — a bug
    noted in SURVEY.md §0.5; the caller keeps best_iou across epochs here.)"""
    if best_iou < miou and epoch > total_epochs // 2:
        return miou, save_checkpoint(net, folder, epoch, "best")
    if not epoch % save_epoch:
        return best_iou, save_checkpoint(net, folder, epoch, "regular")
    return best_iou, None
