#!/bin/bash
# Ablation timing of the fp16 split-operand form of the fused F(4,3) kernel (on the GPU box): builds csrc/wino4f.hip with -DCVK_WINO4F_ABLATE into
# gpurun_out/libcvk_abl.so (with the product's other objects) and times the ablation variants.  Bits: 1 no pixel loads, 2 no transform / split /
# LDS stores, 4 no filter DMA, 8 no epilogue, 16 no MFMAs.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT/pytorch-camvid_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCVK_WINO4F_ABLATE -c wino4f.hip -o /tmp/wino4f_abl.o
OBJS=$(ls *.o | grep -v "exp.o" | grep -v "^wino4f.o" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/wino4f_abl.o -o "$GRAFT_REPO_ROOT/gpurun_out/libcvk_abl.so"
cd "$GRAFT_REPO_ROOT"
python3 - <<'PY'
import ctypes, os, torch
ROOT = os.environ["GRAFT_REPO_ROOT"]
abl = ctypes.CDLL(os.path.join(ROOT, "gpurun_out", "libcvk_abl.so"))
for n, at in (("cvk_conv3x3_wino4h_ablate", [ctypes.c_void_p] * 6 + [ctypes.c_int] * 7 + [ctypes.c_void_p]),
              ("cvk_wino4h_weight_transform", [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]),
              ("cvk_absmax_f32", [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p])):
    getattr(abl, n).restype = ctypes.c_int; getattr(abl, n).argtypes = at
abl.cvk_wino4f_weight_floats.restype = ctypes.c_size_t; abl.cvk_wino4f_weight_floats.argtypes = [ctypes.c_int, ctypes.c_int]
s = torch.cuda.current_stream().cuda_stream
N = 8
for (ci, co, H, W) in ((64, 64, 360, 480), (128, 64, 360, 480), (128, 128, 180, 240)):
    x = torch.randn(N, H, W, ci, device="cuda").clamp_min(0); w = (torch.rand(co, 3, 3, ci, device="cuda") * 2 - 1) * 0.05; b = torch.zeros(co, device="cuda")
    y = torch.empty(N, H, W, co, device="cuda")
    Uh = torch.empty(abl.cvk_wino4f_weight_floats(co, ci), device="cuda")
    amw = torch.zeros(256, device="cuda", dtype=torch.int32); amx = torch.zeros(256, device="cuda", dtype=torch.int32)
    assert abl.cvk_absmax_f32(w.data_ptr(), w.numel() // ci, ci, ci, amw.data_ptr(), s) == 0
    assert abl.cvk_absmax_f32(x.data_ptr(), x.numel() // ci, ci, ci, amx.data_ptr(), s) == 0
    assert abl.cvk_wino4h_weight_transform(w.data_ptr(), Uh.data_ptr(), amw.data_ptr(), co, ci, 0, s) == 0
    variants = ((0, "full"), (1, "-loads"), (2, "-xform/split"), (4, "-dma"), (8, "-epi"), (9, "-loads-epi"), (11, "-loads-xform-epi"), (15, "mfma+lds only"),
                (16, "-mfma"), (31, "nothing"), (32, "conversions->shifts"), (33, "-loads, conversions->shifts"), (64, "-LDS stores"), (65, "-loads-LDS stores"))
    best = {a: 1e9 for a, _ in variants}
    for rnd in range(3):
        for a, name in variants:
            def run():
                assert abl.cvk_conv3x3_wino4h_ablate(x.data_ptr(), Uh.data_ptr(), b.data_ptr(), y.data_ptr(), amx.data_ptr(), amw.data_ptr(), N, H, W, ci, co, co, a, s) == 0
            run(); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): run()
            e1.record(); torch.cuda.synchronize()
            best[a] = min(best[a], e0.elapsed_time(e1) / 5 * 1e3)
    print(f"{ci}->{co} {H}x{W}: " + "  ".join(f"{name} {best[a]:.0f}" for a, name in variants), flush=True)
    # the exact-fp32 kernel under the same switches, for comparison
    abl.cvk_conv3x3_wino4f_ablate.restype = ctypes.c_int
    abl.cvk_conv3x3_wino4f_ablate.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 7 + [ctypes.c_void_p]
    abl.cvk_wino4f_weight_transform.restype = ctypes.c_int
    abl.cvk_wino4f_weight_transform.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    Uf = torch.empty(abl.cvk_wino4f_weight_floats(co, ci), device="cuda")
    assert abl.cvk_wino4f_weight_transform(w.data_ptr(), Uf.data_ptr(), co, ci, 0, s) == 0
    v32 = ((0, "full"), (1, "-loads"), (2, "-xform"), (8, "-epi"), (9, "-loads-epi"), (15, "mfma+lds only"), (16, "-mfma"))
    b32 = {a: 1e9 for a, _ in v32}
    for rnd in range(3):
        for a, name in v32:
            def run32():
                assert abl.cvk_conv3x3_wino4f_ablate(x.data_ptr(), Uf.data_ptr(), b.data_ptr(), y.data_ptr(), N, H, W, ci, co, co, a, s) == 0
            run32(); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): run32()
            e1.record(); torch.cuda.synchronize()
            b32[a] = min(b32[a], e0.elapsed_time(e1) / 5 * 1e3)
    print(f"   exact-fp32 kernel: " + "  ".join(f"{name} {b32[a]:.0f}" for a, name in v32), flush=True)
PY
