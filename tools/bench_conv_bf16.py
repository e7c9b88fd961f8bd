"""Per-layer timing of the bf16-storage convolution kernels on the UNet layer set (forward+stats, data-grad, weight-grad).
    python tools/bench_conv_bf16.py [N H W]        default 4 720 960 (BASELINE.json configs[3])"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("CVK_EXP"):          # CVK_EXP=1: the experiments build (CVK_* kernel knobs)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _exp  # noqa: F401
import torch
from pytorch_camvid_amd import _lib
from pytorch_camvid_amd._lib import check

N, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (4, 720, 960)
lib = _lib.load()
dev = torch.device("cuda:0")
BF = torch.bfloat16
s = torch.cuda.current_stream().cuda_stream
# (name, Cin, Cout, scale divisor)
LAYERS = [("down1.0", 3, 64, 1), ("down1.1", 64, 64, 1), ("down2.0", 64, 128, 2), ("down2.1", 128, 128, 2), ("down3.0", 128, 256, 4),
          ("down3.1", 256, 256, 4), ("down4.0", 256, 512, 8), ("down4.1", 512, 512, 8), ("down5.0", 512, 1024, 16), ("down5.1", 1024, 1024, 16),
          ("ups1", 1024, 512, 8), ("up1.0", 1024, 512, 8), ("up1.1", 512, 512, 8), ("ups2", 512, 256, 4), ("up2.0", 512, 256, 4),
          ("up2.1", 256, 256, 4), ("ups3", 256, 128, 2), ("up3.0", 256, 128, 2), ("up3.1", 128, 128, 2), ("ups4", 128, 64, 1),
          ("up4.0", 128, 64, 1), ("up4.1", 64, 64, 1), ("output", 64, 12, 1)]


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}
for name, ci, co, d in LAYERS:
    h, w = H // d, W // d
    M = N * h * w
    ldx = max(32, ci)
    x = torch.randn(N, h, w, ldx, device=dev).to(BF)
    wt = torch.randn(co, 3, 3, ci, device=dev) * 0.05
    b = torch.zeros(co, device=dev)
    wp = torch.empty(lib.cvk_bf16s_rows_pad(co) * 9 * ldx, device=dev, dtype=BF)
    check(lib.cvk_pack_weight_fwd_bf16(wt.data_ptr(), wp.data_ptr(), co, ci, ldx, s))
    y = torch.empty(M * co, device=dev, dtype=BF)
    P = lib.cvk_bf16s_stat_partials_c(N, h, w, ldx, co)
    st = torch.empty(2 * P * co + P, device=dev)
    flops = 18.0 * M * ci * co
    t = timeit(lambda: check(lib.cvk_conv3x3_bf16s(x.data_ptr(), wp.data_ptr(), b.data_ptr(), y.data_ptr(), st.data_ptr(), st.data_ptr() + 8 * P * co, N, h, w, ldx, co, co, s)))
    KN = {0: "s", 1: "q", 2: "h", 3: "hP", 4: "st", 5: "st2"}
    tl = N * ((h + 15) // 16) * ((w + 31) // 32)
    kf = lib.cvk_conv3x3_bf16s_kernel(N, h, w, ldx, co, 1)
    nt = tl * ((co + 127) // 128 if kf == 1 else (co + 63) // 64)
    row = f"{name:8s} {ci:5d}->{co:5d} @{h:4d}x{w:4d}  fwd[{KN[kf]:3s} {nt / 256.0:5.2f}r] {t*1e6:8.1f}us {flops/t/1e12:7.1f}TF"
    tot["fwd"][0] += flops; tot["fwd"][1] += t
    ld_dy = max(32, co)
    dy = torch.randn(N, h, w, ld_dy, device=dev).to(BF)
    if name != "down1.0":
        wd = torch.empty(lib.cvk_bf16s_rows_pad(ci) * 9 * ld_dy, device=dev, dtype=BF)
        check(lib.cvk_pack_weight_dgrad_bf16(wt.data_ptr(), wd.data_ptr(), co, ci, ld_dy, s))
        dx = torch.empty(M * ldx, device=dev, dtype=BF)
        t = timeit(lambda: check(lib.cvk_conv3x3_bf16s(dy.data_ptr(), wd.data_ptr(), None, dx.data_ptr(), None, None, N, h, w, ld_dy, ci, ldx, s)))
        kd = lib.cvk_conv3x3_bf16s_kernel(N, h, w, ld_dy, ci, 0)
        ntd = tl * ((ci + 127) // 128 if kd == 1 else (ci + 63) // 64)
        row += f"  dgrad[{KN[kd]:3s} {ntd / 256.0:5.2f}r] {t*1e6:8.1f}us {flops/t/1e12:7.1f}TF"
        tot["dgrad"][0] += flops; tot["dgrad"][1] += t
    else:
        row += " " * 45
    dw = torch.empty(co * 9 * ci, device=dev)
    wsb = lib.cvk_conv3x3_wgrad_bf16s_workspace_bytes(N, h, w, ci, co)
    ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    t = timeit(lambda: check(lib.cvk_conv3x3_wgrad_bf16s(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), N, h, w, ci, ldx, co, ld_dy, ws.data_ptr(), wsb, s)))
    row += f"  wgrad {t*1e6:8.1f}us {flops/t/1e12:7.1f}TF"
    tot["wgrad"][0] += flops; tot["wgrad"][1] += t
    print(row, flush=True)
    del x, y, dy, ws
for k, (f, t) in tot.items():
    print(f"total {k:6s} {t*1e3:7.2f} ms  {f/t/1e12:7.1f} TF/s")
