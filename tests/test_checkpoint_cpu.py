"""Checkpoint tooling that needs no GPU: file naming, newest-file selection and epoch parsing of
reference utils.py:95-145 / train.py:111-114, the save rule of train.py:232-240, and the byte layout of a written file."""
import os
import time

import pytest
import torch

import pytorch_camvid_amd as A
from oracle import torch_ref as R


def test_saved_file_is_a_reference_layout_state_dict(tmp_path):
    torch.manual_seed(3)
    net = A.UNet(3, 12)
    path = A.save_checkpoint(net, str(tmp_path / "run1"), 7, "regular")
    assert path.endswith(os.path.join("run1", "7-regular.pth"))
    sd = torch.load(path)
    assert len(sd) == 161
    w = sd["down1.0.conv.0.weight"]
    assert w.shape == (64, 3, 3, 3) and w.is_contiguous()                      # dense OIHW, as the reference writes it
    assert sd["down1.0.conv.1.num_batches_tracked"].dtype == torch.int64
    # the stock-torch rebuild of the reference (same keys as the reference modules) loads the file as is ...
    torch.manual_seed(4)
    ref = R.build("unet", 3, 12)
    ref.load_state_dict(torch.load(path))
    for (k, a), (_, b) in zip(net.state_dict().items(), ref.state_dict().items()):
        assert torch.equal(a, b), k
    # ... and a file written by it loads here, with the kernels' physical weight layout kept
    p2 = str(tmp_path / "run1" / "9-best.pth")
    torch.save(ref.state_dict(), p2)
    torch.manual_seed(5)
    net2 = A.UNet(3, 12)
    A.load_checkpoint(net2, p2)
    assert net2.down2[0].conv[0].weight.is_contiguous(memory_format=torch.channels_last)
    for (k, a), (_, b) in zip(net2.state_dict().items(), ref.state_dict().items()):
        assert torch.equal(a, b), k
    with pytest.raises(ValueError):
        A.save_checkpoint(net, str(tmp_path), 1, "latest")


def test_latest_checkpoint_and_epoch(tmp_path):
    ck = tmp_path / "checkpoints"
    assert A.latest_checkpoint(str(ck)) == ""
    for run, name, age in (("2024-01-01", "10-regular.pth", 300), ("2024-01-01", "75-best.pth", 200),
                           ("2024-02-02", "20-regular.pth", 100), ("2024-02-02", "notes.pth", 0),
                           ("2024-02-02", "x-best.pth", 0)):
        os.makedirs(ck / run, exist_ok=True)
        f = ck / run / name
        f.write_bytes(b"0")
        t = time.time() - age
        os.utime(f, (t, t))
    got = A.latest_checkpoint(str(ck))
    assert got == os.path.abspath(str(ck / "2024-02-02" / "20-regular.pth"))    # newest by mtime among N-best / N-regular
    assert A.checkpoint_epoch(got) == 20 and A.checkpoint_epoch("/a/b/75-best.pth") == 75
    t = time.time() + 50
    os.utime(ck / "2024-01-01" / "75-best.pth", (t, t))
    assert A.latest_checkpoint(str(ck)).endswith("75-best.pth")
    with pytest.raises(ValueError):
        A.checkpoint_epoch("/a/b/model.pth")


def test_save_policy_and_resume(tmp_path):
    torch.manual_seed(0)
    net = A.SegNet(3, 12)
    folder = str(tmp_path / "checkpoints" / "run")
    best, written = 0.0, []
    mious = {1: 0.1, 5: 0.3, 6: 0.2, 7: 0.35, 10: 0.3, 12: 0.5}
    for epoch in range(1, 13):
        best, path = A.save_policy(net, folder, epoch, mious.get(epoch, 0.0), best, total_epochs=12, save_epoch=5)
        if path:
            written.append(os.path.basename(path))
    # epochs <= 6 never write "best"; epoch 5 and 10 are regular saves; 7 and 12 are new bests after half the epochs
    assert written == ["5-regular.pth", "7-best.pth", "10-regular.pth", "12-best.pth"] and best == 0.5
    net2 = A.SegNet(3, 12)
    opt = torch.optim.AdamW(net2.parameters(), lr=5e-4)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=5e-4, steps_per_epoch=30, epochs=20)
    ref_sched = torch.optim.lr_scheduler.OneCycleLR(torch.optim.AdamW(net.parameters(), lr=5e-4), max_lr=5e-4, steps_per_epoch=30, epochs=20)
    trained, path = A.resume(net2, str(tmp_path / "checkpoints"), sched, steps_per_epoch=30)
    assert trained == 12 and path.endswith("12-best.pth") and len(net2.state_dict()) == 182
    for (k, a), (_, b) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert torch.equal(a, b), k
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref_sched.step(12 * 30)                                                  # the reference's own call, train.py:114
    assert sched.get_last_lr() == ref_sched.get_last_lr() and sched.last_epoch == 360
    assert A.resume(net2, str(tmp_path / "nothing")) == (0, "")
