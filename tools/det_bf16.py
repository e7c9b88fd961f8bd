"""Determinism probe: every bf16-storage conv kernel three times on the same operands at the configs[3] sizes; outputs,
statistics and weight gradients must be bitwise identical (a race in a multi-buffered DMA loop shows up here)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_camvid_amd import _lib
from pytorch_camvid_amd._lib import check

lib = _lib.load(); dev = "cuda"; BF = torch.bfloat16
s = torch.cuda.current_stream().cuda_stream
CASES = [(4, 720, 960, 64, 64), (4, 720, 960, 32, 64), (4, 720, 960, 64, 12), (4, 720, 960, 128, 64), (4, 360, 480, 64, 128),
         (4, 360, 480, 128, 128), (4, 180, 240, 256, 256), (4, 90, 120, 512, 512), (4, 45, 60, 1024, 1024), (1, 64, 96, 64, 128), (2, 96, 128, 128, 128)]
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 3
bad = 0
for (N, H, W, Ci, Co) in CASES:
    g = torch.Generator(device=dev).manual_seed(N + H + Ci + Co)
    x = torch.randn(N, H, W, Ci, device=dev, generator=g).to(BF)
    wd = (torch.randn(Co, 3, 3, Ci, device=dev, generator=g) * (2.0 / (9 * Ci)) ** 0.5)
    b = torch.randn(Co, device=dev, generator=g) * 0.1
    rows = lib.cvk_bf16s_rows_pad(Co)
    wp = torch.zeros(rows * 9 * Ci, device=dev, dtype=BF)
    check(lib.cvk_pack_weight_fwd_bf16(wd.data_ptr(), wp.data_ptr(), Co, Ci, Ci, s))
    P = lib.cvk_bf16s_stat_partials_c(N, H, W, Ci, Co)
    ld_dy = max(32, Co)
    dy = torch.zeros(N, H, W, ld_dy, device=dev, dtype=BF); dy[..., :Co] = torch.randn(N, H, W, Co, device=dev, generator=g).to(BF)
    wdp = torch.zeros(lib.cvk_bf16s_rows_pad(Ci) * 9 * ld_dy, device=dev, dtype=BF)
    check(lib.cvk_pack_weight_dgrad_bf16(wd.data_ptr(), wdp.data_ptr(), Co, Ci, ld_dy, s))
    wsb = lib.cvk_conv3x3_wgrad_bf16s_workspace_bytes(N, H, W, Ci, Co)
    ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    # fp32 reference of the same bf16 operands through the library's direct fp32 kernel
    xf = x.float().contiguous(); wf = wd.to(BF).float().contiguous(); yref = torch.empty(N, H, W, Co if Co % 4 == 0 else (Co + 3) // 4 * 4, device=dev)
    check(lib.cvk_conv3x3_fwd(xf.data_ptr(), wf.data_ptr(), b.data_ptr(), yref.data_ptr(), None, N, H, W, Ci, Co, yref.shape[-1], s))
    yref = yref[..., :Co]
    tol = 2.0 ** -7 * yref.abs() + 1e-2
    ref = None
    for rep in range(REP):
        y = torch.full((N, H, W, Co), float("nan"), device=dev, dtype=BF)
        st = torch.full((2 * P * Co + P,), float("nan"), device=dev)
        check(lib.cvk_conv3x3_bf16s(x.data_ptr(), wp.data_ptr(), b.data_ptr(), y.data_ptr(), st.data_ptr(), st.data_ptr() + 8 * P * Co, N, H, W, Ci, Co, Co, s))
        y2 = torch.full((N, H, W, Co), float("nan"), device=dev, dtype=BF)
        check(lib.cvk_conv3x3_bf16s(x.data_ptr(), wp.data_ptr(), b.data_ptr(), y2.data_ptr(), None, None, N, H, W, Ci, Co, Co, s))
        dx = torch.full((N, H, W, Ci), float("nan"), device=dev, dtype=BF)
        check(lib.cvk_conv3x3_bf16s(dy.data_ptr(), wdp.data_ptr(), None, dx.data_ptr(), None, None, N, H, W, ld_dy, Ci, Ci, s))
        dw = torch.full((Co, 3, 3, Ci), float("nan"), device=dev)
        check(lib.cvk_conv3x3_wgrad_bf16s(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), N, H, W, Ci, Ci, Co, ld_dy, ws.data_ptr(), wsb, s))
        torch.cuda.synchronize()
        cur = {"fwd+stats y": y.view(torch.int16), "stats": st.view(torch.int32), "fwd y": y2.view(torch.int16), "dgrad": dx.view(torch.int16), "wgrad": dw.view(torch.int32)}
        e1 = int(((y.float() - yref).abs() > tol).sum().item()); e2 = int(((y2.float() - yref).abs() > tol).sum().item())
        if e1 or e2:
            bad += 1
            print(f"   WRONG vs fp32 reference in run {rep}: stats variant {e1} elements, plain variant {e2} elements")
        if ref is None:
            ref = {k: v.clone() for k, v in cur.items()}
            same = torch.equal(ref["fwd+stats y"], ref["fwd y"])
            print(f"{N}x{H}x{W} {Ci}->{Co}: finite {all(torch.isfinite(t).all().item() for t in (y.float(), st, dx.float(), dw))}, stats/no-stats kernels agree {same}")
        else:
            for k in cur:
                nd = int((cur[k] != ref[k]).sum().item())
                if nd:
                    bad += 1
                    idx = (cur[k] != ref[k]).nonzero()[:4].tolist()
                    print(f"   NONDETERMINISTIC {k}: {nd} elements differ in run {rep}, first at {idx}")
print("nondeterministic results:", bad)
