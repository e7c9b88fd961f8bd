#!/usr/bin/env python3
"""bench.py — forward+backward images/s of UNet(3,12) on synthetic 3x360x480 batches (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]            (N=1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W                   (N>1: one rank per GPU, RCCL)

One "step" = zero_grad(set_to_none) -> net(x) -> CrossEntropy -> backward (-> gradient all-reduce complete for N>1)
on a per-GPU batch of 8 (SURVEY.md §8d timed region; the optimizer step is excluded and reported separately).
Inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.

Extra objects on the line:
  roofline     — the dominant kernel (MFMA implicit-GEMM conv, largest share of step time).  `achieved` / `frac` are
                 the FLOPs the matrix pipe EXECUTES per second over the dense MFMA peak (a hardware fraction, <= 1):
                 algorithmic FLOPs per launch x the kernel's executed share (Winograd F(4,3) runs 9 of 18) / average
                 launch duration, measured live with HIP events on the launch stream during extra instrumented steps
                 after the timed region; the algorithmic rate is kept beside it (`algorithmic_tflops`).
                 peak = 157.3 TFLOP/s (fp32 MFMA) or 2500 TFLOP/s (bf16 MFMA, --precision bf16).
  cpu_baseline — the same loop on the host cores with the stock-torch rebuild of the reference network
                 (oracle/torch_ref.py; the reference itself is stock torch.nn and its source cannot travel to the GPU
                 box), batch 2, rank 0 at N=1 only.
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
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 MFMA peak (spec), opt-in --precision bf16 only
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
    ap.add_argument("--with-optimizer", action="store_true", help="also time AdamW steps (reported separately)")
    ap.add_argument("--with-input-pipeline", action="store_true",
                    help="also time the steps fed from HOST uint8 frames through DevicePrefetcher (pinned staging, 1-byte upload one "
                         "batch ahead, device-side normalisation): the PCIe-inclusive rate, reported separately, never `value`")
    return ap.parse_args()


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


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start one rank per GPU ourselves (torch.distributed.run as a
        # CHILD process, before anything here touches the GPU) and relay rank 0's JSON line.
        raise SystemExit(self_launch(a.gpus))
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
    from pytorch_camvid_amd import ddp
    from pytorch_camvid_amd.modules import runner_of

    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)     # RCCL

    torch.manual_seed(0)                                    # identical init on every rank (also broadcast below)
    net = A.get_model(a.model, 3, 12).to(dev).train()
    A.set_conv_precision(net, a.precision)
    model = ddp.DataParallel(net) if world > 1 else net
    lossf = A.CrossEntropyLoss()
    g = torch.Generator().manual_seed(1234 + rank)          # per-rank shard of the global batch
    x = torch.randn(a.batch, 3, a.height, a.width, generator=g).to(dev)
    t = torch.randint(0, 12, (a.batch, a.height, a.width), generator=g).to(dev)
    params = list(net.parameters())

    def step():
        for p in params:
            p.grad = None
        loss = lossf(model(x), t)
        loss.backward()
        return loss

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device="cpu" if rehearsal else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms = dt / a.steps * 1e3
    value = world * a.batch * a.steps / dt

    opt_ms = None
    if a.with_optimizer and rank == 0:
        opt = torch.optim.AdamW(params, lr=5e-4, weight_decay=0)
        step(); opt.step(); torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(5):
            opt.step()
        torch.cuda.synchronize(dev)
        opt_ms = (time.perf_counter() - t1) / 5 * 1e3

    pipe = None
    if a.with_input_pipeline and rank == 0:
        import numpy as np
        from pytorch_camvid_amd.functional import DevicePrefetcher
        rng = np.random.default_rng(0)
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
                torch.cuda.synchronize(dev); t1 = time.perf_counter()
            for p in params:
                p.grad = None
            lossf(model(xb), tb).backward()
            k += 1
        torch.cuda.synchronize(dev)
        dtp = (time.perf_counter() - t1) / (k - 2)
        pipe = {"images_per_s": round(a.batch / dtp, 3), "ms_per_step": round(dtp * 1e3, 3), "steps": k - 2,
                "host_bytes_per_step": int(a.batch * a.height * a.width * (3 + 8)),
                "what": "same step fed from host uint8 BGR frames + int64 masks: pinned staging, upload on a side stream one batch "
                        "ahead, normalisation to float NHWC on the device (functional.DevicePrefetcher)"}

    roof = None
    kernels = None
    hbm_kernels = None
    if not a.no_kernel_profile:
        from pytorch_camvid_amd import engine
        engine.PROF = []
        for _ in range(3):
            step()
        torch.cuda.synchronize(dev)
        agg, mem = {}, {}
        def executed_share(name):
            # share of the algorithmic (direct-convolution) FLOPs a kernel really executes on the matrix pipe:
            # F(4,3) Winograd 9 of 18, F(2,3) 12 of 18, direct kernels all of them (2-D F(4x4,3x3): 4.5 of 18 times the
            # tile padding — the engine passes the executed count per call)
            return 0.5 if "wino4" in name else ((2.0 / 3.0) if "wino" in name else 1.0)

        for name, work, e0, e1, unit, executed in engine.PROF:
            d = (agg if unit == "flop" else mem).setdefault(name, [0, 0.0, 0.0, 0.0])
            d[0] += 1; d[1] += work; d[2] += e0.elapsed_time(e1) * 1e-3
            d[3] += executed if executed is not None else work * executed_share(name)
        engine.PROF = None
        kernels = {k: {"launches_per_step": v[0] // 3, "avg_us": round(v[2] / v[0] * 1e6, 1),
                       "tflops": round(v[1] / v[2] / 1e12, 2), "ms_per_step": round(v[2] / 3 * 1e3, 3)} for k, v in agg.items()}
        hbm_kernels = {k: {"launches_per_step": v[0] // 3, "ms_per_step": round(v[2] / 3 * 1e3, 3),
                           "algorithmic_GBps": round(v[1] / v[2] / 1e9, 1), "frac_of_8TBps": round(v[1] / v[2] / PEAK_HBM_BPS, 3)}
                       for k, v in mem.items()}
        def peak_of(name):
            return PEAK_BF16_MFMA_TFLOPS if ("bf16" in name and "split" not in name) else PEAK_F32_MFMA_TFLOPS

        for k, v in agg.items():
            kernels[k]["executed_frac_of_peak"] = round(v[3] / v[2] / 1e12 / peak_of(k), 4)
        dom = max(agg.items(), key=lambda kv: kv[1][2])
        cnt, fl, sec, exe = dom[1]
        alg = fl / sec / 1e12                       # algorithmic TFLOP/s (SURVEY.md §8d numerator)
        executed = exe / fl
        peak = peak_of(dom[0])
        ach = exe / sec / 1e12                      # FLOPs the MFMA pipe really executes per second: a hardware fraction <= 1
        allf = sum(v[1] for v in agg.values()); alls = sum(v[2] for v in agg.values())
        alle = sum(v[3] for v in agg.values())
        traffic = None
        pat = "r*_pmc_hbm_traffic_bf16.json" if "bf16" in dom[0] else "r*_pmc_hbm_traffic.json"
        tp = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))[-1:]
        tp = tp[0] if tp else ""
        if os.path.exists(tp):
            for k, v in json.load(open(tp)).items():
                if k.replace(" ", "") == dom[0].replace(" ", ""):
                    traffic = {"bytes_per_launch": (v["read_MB_per_launch"] + v["write_MB_per_launch"]) * 1e6,
                               "source": "profiles/" + os.path.basename(tp) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
                                         "FETCH doubled per MI355X_MICROARCH.md; recorded run, not this run)"}
        roof = {"bound": "mfma", "kernel": dom[0], "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": traffic,
                "algorithmic_tflops": round(alg, 2), "algorithmic_speedup_vs_peak": round(alg / peak, 4),
                "executed_share_of_algorithmic_flops": round(executed, 4),
                "flops_per_launch": fl / cnt, "avg_launch_us": round(sec / cnt * 1e6, 1), "launches_per_step": cnt // 3,
                "all_conv_kernels": {"algorithmic_tflops": round(allf / alls / 1e12, 2),
                                     "executed_frac_of_peak": round(alle / alls / 1e12 / peak, 4),
                                     "ms_per_step": round(alls / 3 * 1e3, 2)}}

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a.model, a.height, a.width)

    if rank == 0:
        line = {
            "metric": "images/sec fwd+bwd UNet 3x360x480 bs=8" if a.model == "unet" and (a.height, a.width, a.batch) == (H, W, 8) and a.precision == "fp32"
                      else f"images/sec fwd+bwd {a.model} 3x{a.height}x{a.width} bs={a.batch} ({a.precision})",
            "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16": "bf16 (bf16 activations/gradients in HBM, bf16 MFMA with f32 accumulate, f32 statistics/parameters)"}[a.precision], "data": "synthetic" + (" (REHEARSAL: ranks share a GPU over gloo — not a measurement)" if rehearsal else ""),
            "config": {"workload": f"{a.model.upper() if a.model=='unet' else 'SegNet'}(3,12) train fwd+bwd+CE, per-GPU batch {a.batch} x 3x{a.height}x{a.width} {a.precision} "
                                   f"(BASELINE.json {'configs[3]' if a.precision == 'bf16' else 'configs[1]'}{' x N ranks, RCCL grad all-reduce (configs[2])' if world > 1 else ''})",
                       "global_batch": world * a.batch, "parallelism": f"dp{world}", "timed_region": "zero_grad+forward+CE+backward"
                                   + ("+allreduce" if world > 1 else ""), "loss": round(float(loss.item()), 6)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if kernels:
            line["conv_kernels"] = kernels
        if hbm_kernels:
            line["hbm_kernels"] = hbm_kernels
        if opt_ms is not None:
            line["adamw_ms"] = round(opt_ms, 3)
        if pipe is not None:
            line["with_input_pipeline"] = pipe
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
