"""Synthetic data of the BASELINE.json configs[0] training-parity protocol, shared by the fixture
generator (tests/golden/make_golden.py, which runs the imported reference on it) and the GPU test
(tests/test_gpu_protocol.py).  Pure torch, CPU generators: bit-identical batches on both sides."""
import torch

PROTO = dict(steps=300, batch=2, h=360, w=480, lr=5e-4, val_batches=4, noise=0.5, cell=8)


def proto_batch(i, val=False):
    """Synthetic learnable segmentation batch i: 12-class blobs on an 8x8-pixel grid, the pixel colour is the class colour
    plus gaussian noise.  Deterministic in (i, val) only."""
    P = PROTO
    pal = torch.randn(12, 3, generator=torch.Generator().manual_seed(99))
    g = torch.Generator().manual_seed((500000 if val else 100000) + i)
    coarse = torch.randint(0, 12, (P["batch"], P["h"] // P["cell"], P["w"] // P["cell"]), generator=g)
    masks = coarse.repeat_interleave(P["cell"], 1).repeat_interleave(P["cell"], 2).contiguous()
    images = pal[masks].permute(0, 3, 1, 2).contiguous() + P["noise"] * torch.randn(P["batch"], 3, P["h"], P["w"], generator=g)
    return images, masks
