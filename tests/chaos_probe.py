"""Not a test (not collected): measures how often whole-network parity at the small golden geometries is decided by a
discrete event (max-pool arg-max / ReLU-mask flip at a 6-12-sample BatchNorm bottleneck) rather than by kernel error.
Runs SegNet/UNet on the GPU in three conv modes against the stock-torch fp32 CPU rebuild (oracle/torch_ref.py) for several
data seeds and prints the max logits deviation per (seed, mode).   python tests/chaos_probe.py [kind n h w]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_camvid_amd as A  # noqa: E402
from pytorch_camvid_amd.modules import runner_of  # noqa: E402
from oracle import torch_ref as R  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "segnet"
n, h, w = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (2, 64, 96)
for seed in range(8):
    torch.manual_seed(0)
    ref = R.build(kind, 3, 12).train()
    x, t = R.synthetic_batch(n, h, w, 100 + seed)
    with torch.no_grad():
        want = ref(x).numpy()
    row = f"seed {seed}:"
    for mode, (wino, wino4) in (("direct", (False, False)), ("F(2,3)", (True, False)), ("F(4,3)", (True, "always"))):
        torch.manual_seed(0)
        net = A.get_model(kind, 3, 12).cuda().train()
        runner_of(net).wino, runner_of(net).wino4 = wino, wino4
        with torch.no_grad():
            got = net(x.cuda()).cpu().numpy()
        e = np.abs(got - want)
        row += f"  {mode} max {e.max():.2e} frac>5e-4 {(e > 5e-4).mean():.3f}"
    print(row)
