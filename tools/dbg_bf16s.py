"""Diagnostic: two-block stage in bf16 mode vs oracle/bf16_emul.py, every tensor's relative L2 error."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_camvid_amd as A
from oracle import bf16_emul as E, torch_ref as R
from tests.test_gpu_bf16 import _stage_pair

dev = torch.device("cuda:0")
for (ci, c1, c2, n, h, w) in [(3, 64, 64, 2, 16, 40), (64, 128, 12, 2, 9, 33), (128, 64, 64, 1, 24, 70), (64, 64, 64, 4, 64, 64)]:
    ref, mine = _stage_pair(ci, c1, c2, seed=ci + c2)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, ci, h, w, generator=g)
    r = torch.randn(n, c2, h, w, generator=g)
    want = E._stage(ref, E._r(x), last=True)
    (want * r).sum().backward()
    # plain fp32 graph for scale
    ref32, _ = _stage_pair(ci, c1, c2, seed=ci + c2)
    w32 = ref32(x); (w32 * r).sum().backward()
    mine = A.set_conv_precision(mine.to(dev).train(), "bf16")
    out = mine(x.to(dev))
    (out * r.to(dev)).sum().backward()
    print(f"--- {ci}->{c1}->{c2} {n}x{h}x{w}: out rel {float((out.detach().cpu()-want.detach()).norm()/want.detach().norm()):.2e}  (emu vs fp32 {float((want.detach()-w32.detach()).norm()/w32.detach().norm()):.2e})")
    for (k, a), (_, b), (_, c) in zip(ref.named_parameters(), mine.named_parameters(), ref32.named_parameters()):
        ga, gb, gc = a.grad, b.grad.cpu(), c.grad
        print(f"   {k:18s} hip-vs-emu {float((ga-gb).norm()/ga.norm()):.2e}   emu-vs-fp32 {float((ga-gc).norm()/gc.norm()):.2e}   |g| {float(ga.norm()):.2e}")
