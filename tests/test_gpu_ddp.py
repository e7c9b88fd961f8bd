"""The N>1 path through the real executor: two ranks on one GPU, engine + ddp.DataParallel, gradients
exchanged over gloo.  Checked against the equivalence statement of SURVEY.md §8e: N ranks x batch b == ONE process that
runs the N shards as separate BatchNorm groups with the same weights and averages the gradients."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


SHAPE = (2, 48, 64)


@pytest.fixture(scope="module")
def two_ranks():
    """ONE spawn of two ranks on this GPU (gloo): a backward pass whose gradients are saved, then two optimizer steps with torch's AdamW
    and two with the fused flat AdamW (tests/ddp_gpu_worker.py)."""
    from tests.ddp_gpu_worker import run
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(run, args=(2, free_port(), d, SHAPE, 8.0, 2), nprocs=2, join=True)
        return torch.load(os.path.join(d, "rank0.pt")), torch.load(os.path.join(d, "rank1.pt"))


def test_two_ranks_real_engine_equal_grouped_bn_single_process(two_ranks):
    import pytorch_camvid_amd as A
    from oracle import torch_ref as R
    shape = SHAPE
    r0, r1 = two_ranks
    # rank 0's parameters were broadcast: both ranks started from the same weights and ended with the same gradients
    for a, b in zip(r0["w0"], r1["w0"]):
        assert torch.equal(a, b)
    for a, b in zip(r0["grads"], r1["grads"]):
        assert torch.equal(a, b)
    assert len(r0["launched"]) >= 4 and r0["launched"][0][0] == 0
    assert r0["loss"] != r1["loss"]                                  # the shards differ
    # single process, same weights, the two shards as separate BN groups, mean of the gradients
    dev = torch.device("cuda:0")
    torch.manual_seed(100)
    net = A.UNet(3, 12).to(dev).train()
    for p, w in zip(net.parameters(), r0["w0"]):
        assert torch.equal(p.detach().cpu(), w)                      # seed 100 == rank 0's init
    lossf = A.CrossEntropyLoss()
    shard_grads, losses = [], []
    for rank in range(2):
        g = torch.Generator().manual_seed(1234 + rank)
        x = torch.randn(shape[0], 3, shape[1], shape[2], generator=g).to(dev)
        t = torch.randint(0, 12, shape, generator=g).to(dev)
        for p in net.parameters():
            p.grad = None
        l = lossf(net(x), t); l.backward()
        losses.append(l.item())
        shard_grads.append([p.grad.detach().cpu().clone() for p in net.parameters()])
    assert losses[0] == r0["loss"] and losses[1] == r1["loss"]       # per-rank BatchNorm statistics, bitwise deterministic kernels
    for i, (a, b) in enumerate(zip(*shard_grads)):
        want = (a + b) / 2
        assert torch.allclose(r0["grads"][i], want, rtol=1e-6, atol=1e-12), i
    # and against the oracle (stock torch on CPU) for the same grouped formulation
    torch.manual_seed(100)
    ref = R.build("unet", 3, 12).train()
    acc = None
    for rank in range(2):
        x, t = R.synthetic_batch(shape[0], shape[1], shape[2], 1234 + rank)
        R.fwd_bwd_step(ref, x, t)
        gs = [p.grad.clone() for p in ref.parameters()]
        acc = gs if acc is None else [u + v for u, v in zip(acc, gs)]
    rel = []
    for (k, p), got, want in zip(ref.named_parameters(), r0["grads"], acc):
        if k.endswith("conv.0.bias"):
            continue
        want = want / 2
        rel.append(float((got - want).norm() / want.norm()))
    rel = np.array(rel)
    assert np.median(rel) < 2e-2 and rel.max() < 0.3, (np.median(rel), rel.max())   # element-wise, tiny geometry (6-sample BatchNorm at the
    # bottleneck): the reference's own fp32-vs-fp64 runs differ by percents there (tests/test_oracle_golden.py); the exact check is the HIP one above


@pytest.mark.parametrize("which", ["train_adamw", "train_flat"])
def test_two_optimizer_steps_real_engine_no_rank_drift(two_ranks, which):
    """The example's training loop through the real executor under ddp.DataParallel, two ranks on this GPU over gloo, two optimizer steps
    (torch.optim.AdamW as train.py:100, then the fused flat AdamW): different initial weights (broadcast), different shards (different
    losses), bitwise identical parameters afterwards — rank drift is the failure 8 GPUs would show first."""
    r0, r1 = (r[which] for r in two_ranks)
    assert r0["losses"] != r1["losses"] and len(r0["losses"]) == 2
    for i, (a, b) in enumerate(zip(r0["params"], r1["params"])):
        assert torch.equal(a, b), i
    assert any(not torch.equal(a, b) for a, b in zip(r0["bn"], r1["bn"]))      # BatchNorm buffers stay per rank


def test_bench_rehearsal_self_launch(tmp_path):
    """`python bench.py --gpus 2` with no launcher starts its own ranks (child torch.distributed.run) and prints ONE JSON
    line with n_gpus 2; CVK_REHEARSAL=1 lets the two ranks share this box's single GPU over gloo (not a measurement)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CVK_REHEARSAL="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    # --with-input-pipeline / --with-optimizer: the extra loops run on EVERY rank (their backward passes issue all-reduces that
    # all ranks must post: with rank 0 alone the N > 1 bench hung — ADVICE r2)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--height", "96", "--width", "128", "--no-cpu-baseline", "--no-kernel-profile", "--with-input-pipeline", "--with-optimizer"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 4 and rec["value"] > 0 and "REHEARSAL" in rec["data"]
    # the N > 1 record says what took part and how much of the exchange was exposed
    dp = rec["dp"]
    assert len(dp["ranks_seen"]) == 2 and {r["rank"] for r in dp["ranks_seen"]} == {0, 1} and dp["rehearsal"] is True
    assert dp["distinct_gpus"] == 1                                     # rehearsal: both ranks on this box's one GPU, and the line says so
    assert len(dp["per_rank_ms_per_step"]) == 2 and all(v > 0 for v in dp["per_rank_ms_per_step"])
    assert dp["allreduce_exposed_ms"] >= 0 and len(dp["buckets"]) >= 1 and sum(b["floats"] for b in dp["buckets"]) >= 34533924
    assert rec["ms_per_step"] >= max(dp["per_rank_ms_per_step"]) - 1e-6  # MAX over ranks
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "rehearsal_n2.json"), "w") as f:
        f.write(lines[0] + "\n")


def _world1_worker(rank, port, out_path):
    """One process, a world-size-1 RCCL group (runs in a child so that the group dies with it)."""
    import json
    import pytorch_camvid_amd as A
    from pytorch_camvid_amd import ddp
    from pytorch_camvid_amd.modules import runner_of
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    env = ddp.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    torch.manual_seed(3)
    net = A.UNet(3, 12).to(dev).train()
    ref = A.UNet(3, 12).to(dev).train()
    ref.load_state_dict(net.state_dict())
    lossf = A.CrossEntropyLoss()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 3, 48, 64, generator=g).to(dev); t = torch.randint(0, 12, (2, 48, 64), generator=g).to(dev)
    wrapped = ddp.DataParallel(net, always_issue=True, bucket_mb=8.0)
    res = {"env": env, "reserve": runner_of(net).persistent_wgs()}
    lossf(ref(x), t).backward()
    lossf(wrapped(x), t).backward()
    res["buckets"] = len(wrapped.sync.launched)
    res["eager_equal"] = all(torch.equal(p.grad, q.grad) for p, q in zip(net.parameters(), ref.parameters()))
    # the same data-parallel step captured WITH its all-reduces
    st0 = {k: v.clone() for k, v in ref.state_dict().items()}
    gs = A.GraphedStep(net, lossf, x, t, allow_grad_sync=True)
    net.load_state_dict(st0); ref.load_state_dict(st0)
    ok = True
    for it in range(2):
        gi = torch.Generator().manual_seed(20 + it)
        xi = torch.randn(2, 3, 48, 64, generator=gi).to(dev); ti = torch.randint(0, 12, (2, 48, 64), generator=gi).to(dev)
        la = gs.replay(xi, ti)
        for p in ref.parameters():
            p.grad = None
        lb = lossf(ref(xi), ti); lb.backward()
        ok = ok and la.item() == lb.item() and all(torch.equal(p.grad, q.grad) for p, q in zip(net.parameters(), ref.parameters()))
    res["graph_equal"] = bool(ok)
    # bf16 mode under the same wrapper: the weight-gradient slab reductions and conv-bias sums of a bucket are deferred and must be
    # flushed BEFORE the bucket's all-reduce is issued (Runner.grads_ready) — bitwise the plain step again
    A.set_conv_precision(net, "bf16"); A.set_conv_precision(ref, "bf16")
    for p in list(net.parameters()) + list(ref.parameters()):
        p.grad = None
    lossf(ref(x), t).backward()
    lossf(wrapped(x), t).backward()
    res["bf16_equal"] = all(torch.equal(p.grad, q.grad) for p, q in zip(net.parameters(), ref.parameters()))
    torch.cuda.synchronize()
    with open(out_path, "w") as f:
        json.dump(res, f)
    torch.distributed.destroy_process_group()


def test_world1_rccl_group_eager_and_captured_step():
    """what one GPU can show of the data-parallel path over REAL RCCL.  A world-size-1 'nccl' group created through
    ddp.init_process_group (NCCL_MAX_NCHANNELS and CVK_DP_RESERVE_CUS set together); ddp.DataParallel(always_issue=True) issues every
    gradient bucket as an RCCL all-reduce; the gradients equal the plain step bit for bit (AVG over one rank), the persistent kernels
    run under the CU reservation, and the step captured as ONE graph with its collectives (GraphedStep(allow_grad_sync=True)) replays
    to the same loss and gradients."""
    import json
    from torch.multiprocessing.spawn import ProcessExitedException
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "w1.json")
        try:
            mp.spawn(_world1_worker, args=(free_port(), out), nprocs=1, join=True)
        except ProcessExitedException as e:
            # ProcessGroupNCCL's watchdog THREAD can abort a process that captures collectives into a graph ("operation not permitted on an event
            # last recorded in a capturing stream": once in six suite runs in round 6; graph.py quiesces it before the capture since).  That race is
            # inside torch, not in the path under test: one more attempt, and only for a signal death of the worker.
            if e.signal_name not in ("SIGABRT", "SIGSEGV"):
                raise
            mp.spawn(_world1_worker, args=(free_port(), out), nprocs=1, join=True)
        res = json.load(open(out))
    assert res["env"]["NCCL_MAX_NCHANNELS"] is not None and res["env"]["CVK_DP_RESERVE_CUS"] == res["env"]["NCCL_MAX_NCHANNELS"]
    assert res["reserve"] > 0 and res["buckets"] >= 4
    assert res["eager_equal"] and res["graph_equal"] and res["bf16_equal"], res


def test_bench_dp_overhead_line():
    """`bench.py --dp-overhead` (N=1): the plain step, the step under DataParallel on a world-1 RCCL group, the exposed all-reduce
    wait and the graph-captured DP step ride on the line under `dp_overhead`."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "2", "--batch", "2", "--height", "96",
                        "--width", "128", "--no-cpu-baseline", "--no-kernel-profile", "--no-extra-configs", "--dp-overhead"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    o = rec["dp_overhead"]
    for k in ("plain_ms_per_step", "dp_world1_ms_per_step", "overhead_pct", "allreduce_exposed_ms", "graphed_dp_ms_per_step",
              "graphed_dp_host_enqueue_ms_per_step", "persistent_workgroups", "buckets"):
        assert k in o, k
    assert o["plain_ms_per_step"] > 0 and o["dp_world1_ms_per_step"] > 0 and o["persistent_workgroups"] > 0 and len(o["buckets"]) >= 1
    assert o["graphed_dp_host_enqueue_ms_per_step"] < o["host_enqueue_ms_per_step"]
