#!/usr/bin/env python3
"""bench.py — forward+backward images/s of UNet(3,12) on synthetic 3x360x480 batches (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]            (N=1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W                   (N>1: one rank per GPU, RCCL)

One "step" = zero_grad(set_to_none) -> net(x) -> CrossEntropy -> backward (-> gradient all-reduce complete for N>1)
on a per-GPU batch of 8 (SURVEY.md §8d timed region; the optimizer step is excluded and reported separately).
Inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.

Extra objects on the line:
  roofline      — the dominant kernel (largest share of step time).  `achieved` / `frac` are the FLOPs the matrix pipe
                  EXECUTES per second over the dense MFMA peak (a hardware fraction, <= 1): algorithmic FLOPs per launch x
                  the kernel's executed share (Winograd F(4,3) runs 9 of 18) / average launch duration, measured live with
                  HIP events on the launch stream during extra instrumented steps after the timed region; the algorithmic
                  rate is kept beside it (`algorithmic_tflops`).  peak = 157.3 TFLOP/s (fp32 MFMA) or 2500 (bf16 MFMA).
  cpu_baseline  — the same loop on the host cores with the stock-torch rebuild of the reference network
                  (oracle/torch_ref.py; the reference itself is stock torch.nn and its source cannot travel to the GPU
                  box), batch 2, rank 0 at N=1 only.
  extra_configs — (N=1, headline configuration only) short legs of the other single-GPU configurations of BASELINE.json in the
                  same process, after the headline measurement: configs[3] (UNet 4x3x720x960, bf16 storage + bf16 MFMA) and
                  configs[4] (SegNet 8x3x360x480 fp32): images/s, ms/step, dominant kernel, its executed fraction of peak.
  dp            — (N>1) what took part and how much of the exchange was exposed: device UUID / PCI bus id of every rank,
                  distinct GPUs, gradient buckets, the compute stream's wait for the all-reduces per step (HIP events around
                  GradSync.finish), per-rank step time min/max, the RCCL environment knobs in effect.
"""
import argparse
import glob
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 MFMA peak (spec), bf16-storage mode only
PEAK_HBM_BPS = 8.0e12             # MI355X_MICROARCH.md: HBM3E spec (6.29e12 measured copy)
PER_GPU_BATCH = 8
H, W = 360, 480


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="unet", choices=["unet", "segnet"])
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH)
    ap.add_argument("--height", type=int, default=H)
    ap.add_argument("--width", type=int, default=W)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"],
                    help="bf16 = bf16-storage mode (BASELINE.json configs[3]: --precision bf16 --height 720 --width 960 --batch 4); the headline is fp32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the configs[3] / configs[4] legs after the headline")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured HIP graph (N=1; reported under `graph_replay`, never `value`)")
    ap.add_argument("--with-optimizer", action="store_true", help="also time AdamW steps (reported separately)")
    ap.add_argument("--dp-overhead", action="store_true",
                    help="(kept for compatibility: since round 5 the leg runs by default at N=1 on the headline configuration) the DP step on a "
                         "world-size-1 RCCL group (real all-reduce launches, CVK_DP_RESERVE_CUS CUs left free), eager and as one captured "
                         "graph; reported under `dp_overhead`, never `value`")
    ap.add_argument("--no-dp-overhead", action="store_true", help="skip the `dp_overhead` leg")
    ap.add_argument("--dp-overhead-child", action="store_true", help=argparse.SUPPRESS)     # internal: the leg's own process (see dp_overhead_in_child)
    ap.add_argument("--w2d-split", type=int, nargs="?", const=3, default=0, choices=[0, 2, 3],
                    help="OPT-IN path (never the default, named in `dtype`): the 2-D Winograd GEMMs of the channel-heavy layers on the 16-bit matrix "
                         "pipe with split fp32 operands — 3 (default of the flag): three bf16 terms, six cross-products; 2: two fp16 terms scaled by "
                         "an exact power of two, three cross-products (csrc/split3.hip, csrc/split_fmt.h, DESIGN.md 5b round 5)")
    ap.add_argument("--with-input-pipeline", action="store_true",
                    help="also time the steps fed from HOST uint8 frames through DevicePrefetcher (pinned staging, 1-byte upload one "
                         "batch ahead, device-side normalisation): the PCIe-inclusive rate, reported separately, never `value`")
    return ap.parse_args()


def config_label(model, batch, h, w, precision, world):
    """Which BASELINE.json configuration a workload is."""
    if model == "segnet" and (batch, h, w, precision) == (8, 360, 480, "fp32"):
        tag = "configs[4]"
    elif model == "unet" and (batch, h, w, precision) == (4, 720, 960, "bf16"):
        tag = "configs[3]"
    elif model == "unet" and (batch, h, w, precision) == (8, 360, 480, "fp32"):
        tag = "configs[1]"
    else:
        tag = "not a BASELINE.json configuration"
    if world > 1:
        tag += " x N ranks, RCCL grad all-reduce (configs[2])" if tag == "configs[1]" else " x N ranks"
    name = "UNET" if model == "unet" else "SegNet"
    return f"{name}(3,12) train fwd+bwd+CE, per-GPU batch {batch} x 3x{h}x{w} {precision} (BASELINE.json {tag})"


def cpu_baseline(model, h, w):
    """Stock-torch rebuild of the reference on the host cores: batch 2, 1 warm-up + 3 timed fwd+bwd steps."""
    from oracle import torch_ref as R
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    threads = max(1, min(cores, 64))
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    net = R.build(model, 3, 12).train()
    x, t = R.synthetic_batch(2, h, w, 1234)
    R.fwd_bwd_step(net, x, t)
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        R.fwd_bwd_step(net, x, t)
    dt = (time.perf_counter() - t0) / n
    return {"value": round(2 / dt, 4), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"stock-torch.nn rebuild of the reference {model} (oracle/torch_ref.py), batch 2 x 3x{h}x{w}, "
                      f"1 warm-up + {n} timed fwd+bwd steps, {dt:.2f} s/step"}


def self_launch(n):
    """Run this script under torch.distributed.run with n ranks on 127.0.0.1 and pass its output through.
    Called before any HIP call of this process (a child process, never an exec)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def executed_share(name):
    # share of the algorithmic (direct-convolution) FLOPs a kernel really executes on the matrix pipe: F(4,3) Winograd 9 of
    # 18, F(2,3) 12 of 18, direct kernels all of them (2-D F(4x4,3x3): 4.5 of 18 times the tile padding — the engine passes
    # the executed count per call)
    return 0.5 if "wino4" in name else ((2.0 / 3.0) if "wino" in name else 1.0)


SPLIT_DTYPES = {
    3: ("f32 tensors; the 2-D Winograd GEMMs of the 13 channel-heavy layers as 3-term bf16 split operands, six exact cross-products per fp32 product on "
        "v_mfma_f32_16x16x32_bf16 with f32 accumulation (OPT-IN path runner.w2d_split = 3, never the default)"),
    2: ("f32 tensors; the 2-D Winograd GEMMs of the 13 channel-heavy layers as 2-term fp16 split operands scaled by an exact power of two per transform "
        "index, three exact cross-products per fp32 product on v_mfma_f32_16x16x32_f16 with f32 accumulation (OPT-IN path runner.w2d_split = 2, "
        "never the default)"),
}


def peak_of(name):
    return PEAK_BF16_MFMA_TFLOPS if ("bf16" in name or "thinb" in name or "split3" in name or "split2h" in name or "wino4h" in name) else PEAK_F32_MFMA_TFLOPS


def kernel_profile(step, dev, nprof=3):
    """Three extra instrumented steps: HIP events around every kernel launch of the path (engine._timed), on the launch stream."""
    from pytorch_camvid_amd import engine
    engine.PROF = []
    for _ in range(nprof):
        step()
    torch.cuda.synchronize(dev)
    agg, mem = {}, {}
    for name, work, e0, e1, unit, executed, nbytes in engine.PROF:
        d = (agg if unit == "flop" else mem).setdefault(name, [0, 0.0, 0.0, 0.0, 0.0])
        d[0] += 1; d[1] += work; d[2] += e0.elapsed_time(e1) * 1e-3
        d[3] += executed if executed is not None else work * executed_share(name)
        d[4] += nbytes if nbytes is not None else 0.0
    engine.PROF = None
    kernels = {k: {"launches_per_step": v[0] // nprof, "avg_us": round(v[2] / v[0] * 1e6, 1),
                   "tflops": round(v[1] / v[2] / 1e12, 2), "ms_per_step": round(v[2] / nprof * 1e3, 3),
                   "executed_frac_of_peak": round(v[3] / v[2] / 1e12 / peak_of(k), 4)} for k, v in agg.items()}
    for k, v in agg.items():
        if v[4] > 0:        # the thin stem / head kernels sit on the HBM side of the ridge (AI 12.9 / 45): both fractions
            kernels[k].update({"algorithmic_GBps": round(v[4] / v[2] / 1e9, 1), "frac_of_8TBps": round(v[4] / v[2] / PEAK_HBM_BPS, 3),
                               "flop_per_byte": round(v[1] / v[4], 1),
                               "bound": "hbm" if v[1] / v[4] < peak_of(k) * 1e12 / PEAK_HBM_BPS else "mfma"})
    hbm_kernels = {k: {"launches_per_step": v[0] // nprof, "ms_per_step": round(v[2] / nprof * 1e3, 3),
                       "algorithmic_GBps": round(v[1] / v[2] / 1e9, 1), "frac_of_8TBps": round(v[1] / v[2] / PEAK_HBM_BPS, 3)}
                   for k, v in mem.items()}
    dom = max(agg.items(), key=lambda kv: kv[1][2])
    cnt, fl, sec, exe = dom[1][:4]
    alg = fl / sec / 1e12                       # algorithmic TFLOP/s (SURVEY.md §8d numerator)
    peak = peak_of(dom[0])
    ach = exe / sec / 1e12                      # FLOPs the MFMA pipe really executes per second: a hardware fraction <= 1
    allf = sum(v[1] for v in agg.values()); alls = sum(v[2] for v in agg.values())
    # time-weighted mean of the per-kernel hardware fractions, each against ITS OWN pipe's peak (a leg can mix fp32-MFMA and 16-bit
    # MFMA kernels: the split-operand legs once printed 2.73 here by pricing bf16-pipe FLOPs against the fp32 peak)
    allfrac = sum(v[3] / 1e12 / peak_of(k) for k, v in agg.items()) / alls
    traffic = None
    pats = ["r*_pmc_hbm_traffic_bf16.json"] if "bf16" in dom[0] else ["r*_pmc_hbm_traffic.json", "r*_pmc_hbm_traffic_fp32.json"]
    tp = sorted((f for pat in pats for f in glob.glob(os.path.join(ROOT, "profiles", pat))), key=os.path.basename)[-1:]
    tp = tp[0] if tp else ""
    if os.path.exists(tp):
        for k, v in json.load(open(tp)).items():
            kk, dd = k.replace(" ", ""), dom[0].replace(" ", "")
            if kk == dd or (dd.endswith(">") and kk.startswith(dd[:-1] + ",")) or (not dd.endswith(">") and kk.startswith(dd + "<")):
                traffic = {"bytes_per_launch": (v["read_MB_per_launch"] + v["write_MB_per_launch"]) * 1e6,
                           "source": "profiles/" + os.path.basename(tp) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
                                     "FETCH doubled per MI355X_MICROARCH.md; recorded run, not this run)"}
    roof = {"bound": "mfma", "kernel": dom[0], "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(ach / peak, 4), "traffic": traffic,
            "algorithmic_tflops": round(alg, 2), "algorithmic_speedup_vs_peak": round(alg / peak, 4),
            "executed_share_of_algorithmic_flops": round(exe / fl, 4),
            "flops_per_launch": fl / cnt, "avg_launch_us": round(sec / cnt * 1e6, 1), "launches_per_step": cnt // nprof,
            "all_conv_kernels": {"algorithmic_tflops": round(allf / alls / 1e12, 2),
                                 "executed_frac_of_peak": round(allfrac, 4),
                                 "ms_per_step": round(alls / nprof * 1e3, 2)},
            "hbm_bound_kernels_ms_per_step": round(sum(v[2] for v in mem.values()) / nprof * 1e3, 2)}
    return roof, kernels, hbm_kernels


def logits_accuracy(net, x):
    """MEASURED in this run (untimed): the headline network's train-mode logits against the dense reference fixture
    tests/golden/unet_s0_8x360x480_dense.npz (every 8th pixel of the imported reference's forward pass on the same seeds: 259,200
    points).  The weights are still the seed-0 initialisation (the bench never steps an optimizer) and x is the bench batch, which is
    exactly what the fixture was generated from.  None when the fixture is not present."""
    path = os.path.join(ROOT, "tests", "golden", "unet_s0_8x360x480_dense.npz")
    if not os.path.exists(path) or tuple(x.shape) != (8, 3, 360, 480):
        return None
    import numpy as np
    ref = np.load(path)["logits_dense"]
    # the fixture is a TRAIN-mode forward pass, which updates the BatchNorm running statistics: put them back, later legs reuse the net
    saved = {k: v.clone() for k, v in net.named_buffers()}
    with torch.no_grad():
        got = net(x)[:, :, ::8, ::8].float().cpu().numpy()
        for k, v in net.named_buffers():
            v.copy_(saved[k])
    dv = np.abs(got - ref)
    tol = None
    try:
        tol = json.load(open(os.path.join(ROOT, "tests", "golden", "drift.json")))["logits_tolerance"]["unet_s0_8x360x480"]["slice_abs"]
    except (OSError, KeyError, ValueError):
        pass
    return {"max_abs_dev": float(f"{dv.max():.3e}"), "share_beyond_3e-4": float(f"{(dv > 3e-4).mean():.3e}"),
            "relative_l2": float(f"{np.linalg.norm(got - ref) / np.linalg.norm(ref):.3e}"), "points": int(ref.size),
            "tolerance_max_abs": tol, "source": "measured in this run against tests/golden/unet_s0_8x360x480_dense.npz (reference-generated)"}


def run_leg(A, dev, model, batch, h, w, precision, steps, warmup, profile, world=1, rank=0, rehearsal=False, want_model=False, split3=False):
    """Build the network, time `steps` steps after `warmup`, optionally profile the kernels.  Returns a dict."""
    from pytorch_camvid_amd import ddp
    torch.manual_seed(0)                                    # identical init on every rank (also broadcast by DataParallel)
    net = A.get_model(model, 3, 12).to(dev).train()
    A.set_conv_precision(net, precision)
    from pytorch_camvid_amd.modules import runner_of as _ro
    _ro(net).w2d_split = int(split3)                        # 0 | 3 | 2: the leg's label decides, not the environment
    wrapped = ddp.DataParallel(net) if world > 1 else net
    lossf = A.CrossEntropyLoss()
    g = torch.Generator().manual_seed(1234 + rank)          # per-rank shard of the global batch
    x = torch.randn(batch, 3, h, w, generator=g).to(dev)
    t = torch.randint(0, 12, (batch, h, w), generator=g).to(dev)
    params = list(net.parameters())

    def step():
        for p in params:
            p.grad = None
        # a training step sees new weights every time (optimizer.step, outside the timed region): the derived weight tensors
        # (Winograd-domain filters, data-grad packs, bf16 copies) are REBUILT inside every timed step, never served from the
        # executor's cache — the cache only helps eval and repeated passes over unchanged weights
        A.mark_weights_dirty(net)
        loss = lossf(wrapped(x), t)
        loss.backward()
        return loss

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(warmup):
        step()
    sync = wrapped.sync if world > 1 else None
    if sync is not None:
        sync.wait_events = []                               # HIP events around every GradSync.finish of the timed steps
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    fence()
    dt_local = time.perf_counter() - t0
    dt = dt_local
    out = {}
    if world > 1:
        tt = torch.tensor([dt_local], device="cpu" if rehearsal else dev, dtype=torch.float64)
        gathered = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(gathered, tt)
        per_rank = [float(v.item()) / steps * 1e3 for v in gathered]
        dt = max(per_rank) * steps / 1e3                    # MAX over ranks
        ev = sync.wait_events
        sync.wait_events = None
        exposed = [a.elapsed_time(b) for a, b in ev] if ev else []
        out["dp_local"] = {"per_rank_ms_per_step": [round(v, 3) for v in per_rank],
                           "allreduce_exposed_ms": round(sum(exposed) / max(len(exposed), 1), 3),
                           "allreduce_exposed_ms_max": round(max(exposed), 3) if exposed else None,
                           "buckets": [{"floats": hi - lo, "MB": round((hi - lo) * 4 / 2 ** 20, 1)} for lo, hi in sync.launched]}
    out.update({"value": world * batch * steps / dt, "ms": dt / steps * 1e3, "loss": float(loss.item()), "step": step, "fence": fence,
                "net": net, "wrapped": wrapped, "lossf": lossf, "params": params, "x": x, "t": t})
    if profile:
        out["roof"], out["kernels"], out["hbm_kernels"] = kernel_profile(step, dev)
    if not want_model:
        for k in ("step", "fence", "net", "wrapped", "lossf", "params", "x", "t"):
            out.pop(k)
        del net, wrapped, x, t, params
        torch.cuda.empty_cache()
    return out


def dp_overhead_leg(A, dev, net, lossf, leg, a, plain_step):
    """VERDICT r3 #5a: everything about data parallel that ONE GPU can show.  A world-size-1 RCCL group; the headline network under
    ddp.DataParallel(always_issue=True): every gradient bucket goes through a real RCCL all-reduce (a copy kernel on RCCL's stream at
    world 1), the executor's persistent kernels leave CVK_DP_RESERVE_CUS CUs free, GradSync.finish makes the compute stream wait.
    Reported: the plain step and the DP step timed back to back in this process, the exposed wait, and the same DP step replayed from
    ONE captured graph (collectives inside) with its host enqueue time."""
    from pytorch_camvid_amd import ddp
    from pytorch_camvid_amd.graph import GraphedStep
    from pytorch_camvid_amd.modules import runner_of
    import socket
    if not dist.is_initialized():
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(port))
        ddp.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    n = max(10, min(a.steps, 40))

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / n * 1e3
        th = 0.0                                             # host time to enqueue ONE step into an idle queue
        for _ in range(5):
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            fn()
            th += time.perf_counter() - t1
        torch.cuda.synchronize(dev)
        return ms, th / 5 * 1e3

    plain_ms, _ = timed(plain_step)
    wrapped = ddp.DataParallel(net, always_issue=True)
    x, t, params = leg["x"], leg["t"], leg["params"]

    def dp_step():
        for p in params:
            p.grad = None
        A.mark_weights_dirty(net)
        lossf(wrapped(x), t).backward()

    wrapped.sync.wait_events = []
    dp_ms, dp_host = timed(dp_step)
    ev = wrapped.sync.wait_events
    wrapped.sync.wait_events = None
    exposed = [e0.elapsed_time(e1) for e0, e1 in ev[3:]] if len(ev) > 3 else []
    buckets = [{"floats": hi - lo, "MB": round((hi - lo) * 4 / 2 ** 20, 1)} for lo, hi in wrapped.sync.launched]
    gs = GraphedStep(net, lossf, x, t, allow_grad_sync=True)
    g_ms, g_host = timed(gs.replay)
    out = {"plain_ms_per_step": round(plain_ms, 3), "dp_world1_ms_per_step": round(dp_ms, 3),
           "overhead_pct": round((dp_ms / plain_ms - 1.0) * 100.0, 2),
           "allreduce_exposed_ms": round(sum(exposed) / max(len(exposed), 1), 3), "buckets": buckets,
           "persistent_workgroups": runner_of(net).persistent_wgs(), "host_enqueue_ms_per_step": round(dp_host, 3),
           "graphed_dp_ms_per_step": round(g_ms, 3), "graphed_dp_host_enqueue_ms_per_step": round(g_host, 4),
           "rccl_env": {k: v for k, v in ddp.rccl_env().items() if v is not None}, "steps": n,
           "what": "one process (a child of bench.py since round 6), same box: plain step vs the step under ddp.DataParallel on a world-size-1 RCCL group with always_issue=True "
                   "(bucketed all-reduces really issued, CU reservation active), eager and replayed from one captured HIP graph; says what "
                   "data parallel costs before any xGMI link is involved — not a scaling measurement"}
    del gs
    runner_of(net).grad_sync = None
    return out


def dp_overhead_in_child(a):
    """Run the dp_overhead leg in a CHILD process and return its object.  The leg creates an RCCL communicator and captures a HIP graph with collectives
    inside: ProcessGroupNCCL's watchdog thread can then abort the process ("operation not permitted on an event last recorded in a capturing stream",
    seen once in six GPU suite runs in round 6; an intermittent SIGSEGV at interpreter exit in the same configuration).  That is torch's thread, not this
    path — and it must not be able to take the headline measurement with it: the parent never creates a process group at N = 1."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--dp-overhead-child", "--steps", str(a.steps), "--warmup", str(min(a.warmup, 3)),
           "--model", a.model, "--batch", str(a.batch), "--height", str(a.height), "--width", str(a.width), "--precision", a.precision]
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    last = None
    for attempt in (1, 2):              # one retry: the abort is a race inside torch
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        except subprocess.TimeoutExpired:
            last = {"error": "the dp_overhead child process timed out"}
            continue
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode == 0 and lines:
            out = json.loads(lines[-1])
            if attempt > 1:
                out["attempts"] = attempt
            return out
        last = {"error": f"dp_overhead child exited with {r.returncode}: {r.stderr.strip().splitlines()[-1][:200] if r.stderr.strip() else ''}"}
    return last


def dp_overhead_child_main(a):
    """The child of dp_overhead_in_child: the headline network, the leg, ONE JSON object on stdout."""
    real_stdout = _claim_stdout()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import pytorch_camvid_amd as A
    leg = run_leg(A, dev, a.model, a.batch, a.height, a.width, a.precision, 3, a.warmup, False, want_model=True)
    out = dp_overhead_leg(A, dev, leg["net"], leg["lossf"], leg, a, leg["step"])
    real_stdout.write(json.dumps(out) + "\n")
    real_stdout.flush()
    import atexit
    atexit._run_exitfuncs()
    sys.stderr.flush()
    os._exit(0)


def dp_identity(dev, world, rehearsal):
    """Which device each rank really runs on (all-gathered): proves N distinct GPUs took part."""
    pr = torch.cuda.get_device_properties(dev)
    uuid = str(getattr(pr, "uuid", ""))
    bus = ""
    try:
        bus = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
    except AttributeError:
        pass
    me = {"rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device_index": dev.index,
          "name": pr.name, "uuid": uuid, "pci_bus_id": bus, "pid": os.getpid()}
    ranks = [None] * world
    dist.all_gather_object(ranks, me)
    ids = {(r["uuid"], r["pci_bus_id"]) for r in ranks}
    return {"ranks_seen": ranks, "distinct_gpus": len(ids), "rehearsal": rehearsal,
            "backend": dist.get_backend(),
            "rccl_env": {k: v for k, v in __import__("pytorch_camvid_amd").ddp.rccl_env().items() if v is not None},
            "note": "the conv grids assume an undisturbed chip (whole rounds of 256 CUs, csrc/wino2d.hip staggered start): RCCL's "
                    "all-reduce kernels occupy NCCL_MAX_NCHANNELS workgroups while backward runs; lower it (e.g. 8-16) if "
                    "allreduce_exposed_ms is small but ms_per_step grows with N, and A/B the 2-D GEMM's staggered start with the experiments build (CVK_W2D_NO_STAGGER=1).  The persistent "
                    "fused F(4,3) kernel and the persistent bf16 kernels (one workgroup per CU) run on CUs - CVK_DP_RESERVE_CUS workgroups under data "
                    "parallel; ddp.init_process_group sets NCCL_MAX_NCHANNELS (default 8) and CVK_DP_RESERVE_CUS together."}


def _claim_stdout():
    """The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a version banner when its first
    communicator is created), so file descriptor 1 is pointed at stderr for the whole run and the line goes out through a saved copy."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    return os.fdopen(saved, "w")


def main():
    import faulthandler
    faulthandler.enable()               # a native crash prints the Python stack on stderr (round 6: an intermittent SIGSEGV in interpreter teardown)
    a = parse()
    if a.dp_overhead_child:
        return dp_overhead_child_main(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start one rank per GPU ourselves (torch.distributed.run as a
        # CHILD process, before anything here touches the GPU) and relay rank 0's JSON line.
        raise SystemExit(self_launch(a.gpus))
    real_stdout = _claim_stdout()       # after the self-launch decision: the child ranks claim their own
    if world != a.gpus:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {a.gpus}: pass --gpus equal to the number of ranks")
    # CVK_REHEARSAL=1: ranks share the visible GPUs and talk over gloo — a plumbing check of the N>1 path on a 1-GPU
    # box (its number is meaningless and is labelled as such); the real thing is one rank per GPU over RCCL.
    rehearsal = os.environ.get("CVK_REHEARSAL") == "1"
    if rehearsal:
        local %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import pytorch_camvid_amd as A

    if world > 1:
        # the RCCL channel count and the CUs the executor leaves to the collectives are set together by ddp.init_process_group
        from pytorch_camvid_amd import ddp as _ddp
        if rehearsal:
            _ddp.init_process_group("gloo")
        else:
            _ddp.init_process_group("nccl", device_id=dev)     # RCCL

    headline = (a.model, a.batch, a.height, a.width, a.precision) == ("unet", PER_GPU_BATCH, H, W, "fp32") and not a.w2d_split
    leg = run_leg(A, dev, a.model, a.batch, a.height, a.width, a.precision, a.steps, a.warmup, not a.no_kernel_profile,
                  world=world, rank=rank, rehearsal=rehearsal, want_model=True, split3=a.w2d_split)
    step, params, net, wrapped, lossf = leg["step"], leg["params"], leg["net"], leg["wrapped"], leg["lossf"]

    # The optional extra loops run on EVERY rank: at N>1 each backward issues the bucketed all-reduces, which all ranks must post.
    opt_ms = None
    if a.with_optimizer:
        opt = torch.optim.AdamW(params, lr=5e-4, weight_decay=0)
        step(); opt.step(); torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(5):
            opt.step()
        torch.cuda.synchronize(dev)
        opt_ms = (time.perf_counter() - t1) / 5 * 1e3

    pipe = None
    if a.with_input_pipeline:
        import numpy as np
        from pytorch_camvid_amd.functional import DevicePrefetcher
        rng = np.random.default_rng(rank)
        nb = 6
        host = [(rng.integers(0, 256, (a.batch, a.height, a.width, 3), dtype=np.uint8),
                 rng.integers(0, 12, (a.batch, a.height, a.width)).astype(np.int64)) for _ in range(nb)]
        nsteps = max(8, min(a.steps, 40))

        def feed():
            for i in range(nsteps + 2):
                yield host[i % nb]

        it = DevicePrefetcher(feed(), device=dev)
        k = 0
        for xb, tb in it:
            if k == 2:
                leg["fence"](); t1 = time.perf_counter()
            for p in params:
                p.grad = None
            A.mark_weights_dirty(net)
            lossf(wrapped(xb), tb).backward()
            k += 1
        leg["fence"]()
        dtp = (time.perf_counter() - t1) / (k - 2)
        pipe = {"images_per_s": round(world * a.batch / dtp, 3), "ms_per_step": round(dtp * 1e3, 3), "steps": k - 2,
                "host_bytes_per_step": int(a.batch * a.height * a.width * (3 + 8)),
                "what": "same step fed from host uint8 BGR frames + int64 masks: pinned staging, upload on a side stream one batch "
                        "ahead, normalisation to float NHWC on the device (functional.DevicePrefetcher); rank 0's clock"}

    graph = None
    if a.graph and world == 1:
        from pytorch_camvid_amd.graph import GraphedStep
        gs = GraphedStep(net, lossf, leg["x"], leg["t"])
        for _ in range(3):
            gs.replay()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(a.steps):
            gs.replay()
        th = time.perf_counter() - t1                       # host time to enqueue a.steps replays
        torch.cuda.synchronize(dev)
        tg = time.perf_counter() - t1
        graph = {"images_per_s": round(a.batch * a.steps / tg, 3), "ms_per_step": round(tg / a.steps * 1e3, 3),
                 "host_enqueue_ms_per_step": round(th / a.steps * 1e3, 4), "loss": round(float(gs.loss.item()), 6),
                 "what": "zero_grad + forward + CE + backward replayed from ONE captured HIP graph (pytorch_camvid_amd.graph.GraphedStep)"}
        del gs

    w2d_tile = None
    if a.precision == "fp32":
        # the accuracy trade of the 2-D Winograd tile is part of the configuration (tests/test_gpu_nets.py dense fixtures)
        from pytorch_camvid_amd.modules import runner_of as _runner_of
        cfg = _runner_of(net).w2tile_cfg
        tf, td = (cfg, cfg) if cfg in (4, 6) else ((4, 6) if a.model == "segnet" else (6, 6))
        w2d_tile = {"tiles": f"channel-heavy layers: 2-D Winograd F({tf}x{tf},3x3) forward / weight-grad, F({td}x{td},3x3) data-grad "
                             "(CVK_W2D_TILE=4 selects the finer-rounding F(4x4,3x3) everywhere: about -8 % images/s)",
                    "logits_vs_reference": logits_accuracy(net, leg["x"]) if (headline and rank == 0) else None}
    dp_over, dp_fail = None, False
    if world == 1 and (a.dp_overhead or headline) and not a.no_dp_overhead and not rehearsal:
        torch.cuda.synchronize(dev)
        dp_over = dp_overhead_in_child(a)          # its own process: RCCL + graph capture cannot take the headline down
        if "error" in dp_over:
            print(f"bench.py: the dp_overhead leg failed: {dp_over['error']}", file=sys.stderr)
            if a.dp_overhead:           # asked for explicitly: the failure is the result
                dp_fail = True

    dp = None
    if world > 1:
        dp = dp_identity(dev, world, rehearsal)
        dp.update(leg["dp_local"])
        if not rehearsal and dp["allreduce_exposed_ms"] > 0.03 * leg["ms"]:
            # ddp.DEFAULT_RCCL_CHANNELS (8) was chosen at world size 1, where no byte crosses xGMI: say so when it shows
            dp["warning"] = (f"the compute stream waited {dp['allreduce_exposed_ms']} ms per step for the gradient all-reduce "
                             f"({100 * dp['allreduce_exposed_ms'] / leg['ms']:.1f} % of the step): raise NCCL_MAX_NCHANNELS / CVK_DP_RESERVE_CUS "
                             "(ddp.init_process_group(rccl_channels=16)) or lower CVK_DDP_BUCKET_MB")
            print("bench.py: " + dp["warning"], file=sys.stderr)

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a.model, a.height, a.width)

    # free the headline network before the extra legs
    for k in ("step", "fence", "net", "wrapped", "lossf", "params", "x", "t"):
        leg.pop(k, None)
    del step, params, net, wrapped, lossf
    torch.cuda.empty_cache()

    extra = None
    if world == 1 and headline and not a.no_extra_configs:
        extra = []
        es, ew = max(3, min(a.steps, 30)), max(2, min(a.warmup, 5))       # (round 6: 30 / 5 instead of 10 / 3 — the short legs read 0.5-0.9 % below a run of their own)
        for (m, b, hh, ww, prec, sp3) in (("unet", 4, 720, 960, "bf16", 0), ("segnet", 8, 360, 480, "fp32", 0),
                                          ("unet", 8, 360, 480, "fp32", 3), ("unet", 8, 360, 480, "fp32", 2)):
            e = run_leg(A, dev, m, b, hh, ww, prec, es, ew, True, split3=sp3)
            r = e["roof"]
            extra.append({"workload": config_label(m, b, hh, ww, prec, 1) + (" — OPT-IN split-operand GEMMs (runner.w2d_split = %d)" % sp3 if sp3 else ""),
                          "dtype": SPLIT_DTYPES[sp3] if sp3 else ("bf16" if prec == "bf16" else "f32"),
                          "images_per_s": round(e["value"], 3), "ms_per_step": round(e["ms"], 3), "steps": es, "warmup": ew,
                          "loss": round(e["loss"], 6), "dominant_kernel": r["kernel"], "executed_frac_of_peak": r["frac"],
                          "peak_tflops": r["peak"], "dominant_kernel_ms_per_step": round(r["avg_launch_us"] * r["launches_per_step"] / 1e3, 3),
                          "all_conv_kernels": r["all_conv_kernels"], "hbm_bound_kernels_ms_per_step": r["hbm_bound_kernels_ms_per_step"],
                          "top_kernels_ms_per_step": dict(sorted([(k, v["ms_per_step"]) for k, v in list(e["kernels"].items()) + list(e["hbm_kernels"].items())],
                                                                 key=lambda kv: -kv[1])[:8])})

    if rank == 0:
        line = {
            "metric": "images/sec fwd+bwd UNet 3x360x480 bs=8" if headline
                      else f"images/sec fwd+bwd {a.model} 3x{a.height}x{a.width} bs={a.batch} ({a.precision})",
            "value": round(leg["value"], 3), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(leg["ms"], 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": (SPLIT_DTYPES[a.w2d_split] if a.w2d_split else
                      {"fp32": "f32", "bf16": "bf16 (bf16 activations/gradients in HBM, bf16 MFMA with f32 accumulate, f32 statistics/parameters)"}[a.precision]),
            "data": "synthetic" + (" (REHEARSAL: ranks share a GPU over gloo — not a measurement)" if rehearsal else ""),
            "config": {"workload": config_label(a.model, a.batch, a.height, a.width, a.precision, world),
                       "global_batch": world * a.batch, "parallelism": f"dp{world}", "timed_region": "zero_grad+forward+CE+backward"
                                   + ("+allreduce" if world > 1 else ""), "loss": round(leg["loss"], 6),
                       "derived_weights": "rebuilt in every timed step (executor cache invalidated per step, as after optimizer.step)",
                       **({"w2d_tile": w2d_tile} if w2d_tile else {})},
            "roofline": leg.get("roof"), "cpu_baseline": cpu,
        }
        if leg.get("kernels"):
            line["conv_kernels"] = leg["kernels"]
        if leg.get("hbm_kernels"):
            line["hbm_kernels"] = leg["hbm_kernels"]
        if extra is not None:
            line["extra_configs"] = extra
        if dp is not None:
            line["dp"] = dp
        if opt_ms is not None:
            line["adamw_ms"] = round(opt_ms, 3)
        if pipe is not None:
            line["with_input_pipeline"] = pipe
        if graph is not None:
            line["graph_replay"] = graph
        if dp_over is not None:
            line["dp_overhead"] = dp_over
        real_stdout.write(json.dumps(line) + "\n")
        real_stdout.flush()
    # The line is out.  Leave WITHOUT the interpreter's teardown: about one run in seven of the default configuration (HIP graphs with captured RCCL
    # collectives, a world-size-1 communicator, ~40 GB of cached device memory) died with SIGSEGV while Python, HIP and RCCL unwound at exit —
    # after the measurement, but a driver sees rc 139.  Ranks of a real N > 1 job still leave their group in order first.
    sys.stderr.flush()
    if dist.is_initialized() and world > 1:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:          # the measurement is printed; a failing teardown must not turn it into an error
            print(f"bench.py: destroy_process_group failed: {e}", file=sys.stderr)
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        # under rocprofv3 the tool library writes its trace files from exit handlers of the native runtime: leave the ordinary way
        if dist.is_initialized() and world == 1:
            dist.destroy_process_group()
        raise SystemExit(3 if dp_fail else 0)
    import atexit
    atexit._run_exitfuncs()             # exit callbacks still run; skipped: object destruction in interpreter finalisation and the native libraries' static destructors
    sys.stderr.flush()
    os._exit(3 if dp_fail else 0)


if __name__ == "__main__":
    main()
