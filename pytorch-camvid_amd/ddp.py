"""Data-parallel training: one process per GPU, bucketed gradient all-reduce (RCCL over xGMI) overlapped with backward.

The reference's only parallelism is per-core data parallelism with a hidden gradient all-reduce
(legacy/train_tpu.py:115 `xm.optimizer_step`, :214 `dp.DataParallel`).  MI355X-native equivalent:
the executor (engine.Runner) writes every parameter gradient into ONE flat fp32 buffer laid out in reverse execution
order, so the buffer fills front-to-back while backward runs.  Buckets are contiguous slices of that buffer cut at
layer boundaries; the moment the last layer of a bucket has enqueued its weight-grad kernels the slice is handed to
`torch.distributed.all_reduce(async_op=True)` — backend "nccl" is RCCL on ROCm, which orders the collective after
the kernels already enqueued on the compute stream and runs it on its own stream, i.e. under the remaining backward.
No gradient copies (the parameters' .grad are views of the flat buffer), no per-tensor collectives.
BatchNorm statistics stay per rank (the DP semantics of the reference's TPU path and of stock DDP); parameters and
BN buffers are broadcast from rank 0 once at wrap time.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of the full 138 MB gradient set is
per-link bound at ~1.6 ms, so a handful of ~25-50 MB buckets keeps each collective bandwidth-bound rather than
latency-bound while still starting early.
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn


# RCCL's all-reduce kernels run beside backward and hold one CU per channel; the exchange needs ~7 GB/s (138 MB per 34 ms
# step), so a few channels suffice: 8 since round 5 (tools/dp_sweep.sh at world size 1: 2 / 4 / 8 / 16 channels cost the step the same
# within 0.1 % — the exchange is never exposed — so the default is the value that leaves RCCL a comfortable 8 rings for the day the
# links are real and reserves half the CUs of round 4's 16).  The executor leaves CVK_DP_RESERVE_CUS CUs (default 8) to them: its persistent
# one-workgroup-per-CU kernels launch on CUs - CVK_DP_RESERVE_CUS workgroups under data parallel (engine.Runner.persistent_wgs).
# The two numbers belong together, so they are set together — here, not in a benchmark script.
# UNVERIFIED BEYOND WORLD SIZE 1: no N>1 RCCL run has ever been possible on this pool (SCALE_r01-r05 skipped); at world size 1 the
# all-reduce moves nothing over xGMI.  The arithmetic says 8 rings are ample (138 MB per 34 ms step = 4 GB/s against ~150 GB/s per link),
# but if bench.py's `dp.allreduce_exposed_ms` at N>1 is not ~0, raise NCCL_MAX_NCHANNELS (the caller's environment wins) — bench.py prints
# a `dp.warning` when the exposed wait exceeds 3 % of the step.
DEFAULT_RCCL_CHANNELS = 8


def init_process_group(backend="nccl", rccl_channels=None, **kwargs):
    """torch.distributed.init_process_group with the RCCL settings this path is tuned for (VERDICT r3 #5b: users of
    ddp.DataParallel get what bench.py measures).  Before the group is created (RCCL reads its environment at communicator creation):
      * NCCL_MAX_NCHANNELS (unless the caller's environment already sets it) = rccl_channels (default 8);
      * CVK_DP_RESERVE_CUS follows it (unless set): the CUs the executor's persistent kernels leave free for the collectives;
      * HSA_ENABLE_IPC_MODE_LEGACY=0 (the host driver of this pool only supports dmabuf IPC).
    Rendezvous arguments (init_method, rank, world_size, device_id, ...) pass through unchanged.  Returns rccl_env()."""
    ch = DEFAULT_RCCL_CHANNELS if rccl_channels is None else int(rccl_channels)
    if backend == "nccl":
        os.environ.setdefault("NCCL_MAX_NCHANNELS", str(ch))
        os.environ.setdefault("CVK_DP_RESERVE_CUS", os.environ["NCCL_MAX_NCHANNELS"])
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not dist.is_initialized():
        dist.init_process_group(backend, **kwargs)
    return rccl_env()


def rccl_env():
    """The knobs in effect (for logs and bench lines)."""
    return {k: os.environ.get(k) for k in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS", "NCCL_NCHANNELS_PER_PEER", "RCCL_MSCCL_ENABLE",
                                           "NCCL_ALGO", "NCCL_PROTO", "HSA_ENABLE_IPC_MODE_LEGACY", "CVK_DDP_BUCKET_MB",
                                           "CVK_DP_RESERVE_CUS")}


def make_buckets(layer_ranges, bucket_floats):
    """layer_ranges: [(begin, end)] float offsets of each layer's gradients in completion order (ascending offsets).
    Returns [(begin, end, n_layers)] contiguous buckets cut at layer boundaries, each >= bucket_floats except the last."""
    buckets = []
    b0, n = None, 0
    for (lo, hi) in layer_ranges:
        if b0 is None:
            b0 = lo
        n += 1
        if hi - b0 >= bucket_floats:
            buckets.append((b0, hi, n))
            b0, n = None, 0
    if b0 is not None:
        buckets.append((b0, layer_ranges[-1][1], n))
    return buckets


class _SyncCall:
    """State of ONE backward call (engine.Runner.backward): begin() creates it, the executor calls layer_done(slot)
    for each conv block in reverse execution order and finish() at the end.  Nothing is kept on the GradSync between
    calls, so two backward passes through one network inside a single autograd graph do not share bucket cursors."""

    def __init__(self, owner, st):
        self.owner = owner
        offs, params = st.goffs, st.params
        nslots = len(params) // 4
        self.ranges = []
        for slot in range(nslots - 1, -1, -1):                       # completion order == ascending flat offsets
            lo = offs[4 * slot]
            hi = offs[4 * slot + 3] + (params[4 * slot + 3].numel() + 3) // 4 * 4
            self.ranges.append((lo, hi))
        self.buckets = make_buckets(self.ranges, owner.bucket_floats)
        self.next_bucket = 0
        self.layers_done = 0
        self.work = []
        self.launched = []

    def closes_bucket(self):
        """Will the NEXT layer_done() hand a bucket to the all-reduce?  (The executor finalises its queued gradient pieces first.)"""
        if self.next_bucket >= len(self.buckets) or self.layers_done >= len(self.ranges):
            return False
        return self.buckets[self.next_bucket][1] <= self.ranges[self.layers_done][1]

    def in_flight(self):
        """Has an asynchronous collective been issued in this backward pass (it may still be running beside the next kernels)?"""
        return len(self.work) > 0

    def layer_done(self, st, slot):
        self.layers_done += 1
        done_upto = self.ranges[self.layers_done - 1][1]
        while self.next_bucket < len(self.buckets) and self.buckets[self.next_bucket][1] <= done_upto:
            lo, hi, _ = self.buckets[self.next_bucket]
            self.owner._issue(self, st.gflat[lo:hi])
            self.launched.append((lo, hi))
            self.next_bucket += 1

    def finish(self, st):
        assert self.next_bucket == len(self.buckets), "a gradient bucket was never completed"
        ev = None
        if self.owner.wait_events is not None and st.gflat.is_cuda:
            # diagnostics (bench.py): HIP events on the compute stream around its wait for the collectives = the part of
            # the exchange that backward did not hide.  Recorded only, read after the timed region (no host sync here).
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w, t in self.work:
            w.wait()                       # nccl: makes the current stream wait for the collective; no host sync
            if t is not None:
                t.div_(self.owner.world)
        self.work = []
        if ev is not None:
            ev[1].record()
            self.owner.wait_events.append(ev)
        self.owner.launched = self.launched


class GradSync:
    """Bucketed gradient all-reduce hooked into engine.Runner.backward (begin() -> a per-call _SyncCall)."""

    def __init__(self, process_group=None, bucket_mb=None, always_issue=False):
        if bucket_mb is None:       # CVK_DDP_BUCKET_MB: bucket size knob (xGMI rings are per-link bound: few large buckets)
            bucket_mb = float(os.environ.get("CVK_DDP_BUCKET_MB", "32"))
        self.pg = process_group
        self.always_issue = always_issue     # issue the collectives even for world_size 1 (plumbing tests)
        self.bucket_floats = int(bucket_mb * (1 << 20) / 4)
        self.world = dist.get_world_size(process_group)
        self._native_avg = dist.get_backend(process_group) == "nccl"
        self.launched = []       # (begin, end) of the buckets issued during the last finished backward (introspection/tests)
        self.wait_events = None  # a list while bench.py collects (start, end) HIP events around finish()'s stream waits

    def begin(self, st, plan=None):
        return _SyncCall(self, st)

    def _issue(self, call, t):
        if self.world == 1 and not self.always_issue:
            return
        if self._native_avg:
            call.work.append((dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.pg, async_op=True), None))
        elif t.is_cuda:
            # gloo rehearsal with the real engine on a GPU (tests, CVK_REHEARSAL): gloo moves host memory, so the bucket
            # is staged through the CPU synchronously — a plumbing path, never a measurement
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.pg)
            t.copy_(h.div_(self.world))
        else:  # gloo on CPU tensors has no AVG
            call.work.append((dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), t))


class DataParallel(nn.Module):
    """Wraps a pytorch_camvid_amd network for process-per-GPU data parallel training.

    >>> ddp.init_process_group("nccl")             # RCCL, with the channel / CU-reservation settings of this path
    >>> net = DataParallel(UNet(3, 12).cuda())
    Each rank feeds its own minibatch shard; after loss.backward() every rank holds the gradient mean."""

    def __init__(self, module, process_group=None, bucket_mb=None, broadcast=True, always_issue=False):
        super().__init__()
        from .modules import runner_of
        self.module = module
        self.sync = GradSync(process_group, bucket_mb, always_issue)
        if broadcast and self.sync.world > 1:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    # conv weights are stored channels_last: broadcast their dense [Cout][3][3][Cin] view
                    v = t.permute(0, 2, 3, 1) if t.dim() == 4 and not t.is_contiguous() else t
                    v = v if v.is_contiguous() else t.data
                    if self.sync._native_avg or not v.is_cuda:
                        dist.broadcast(v, src=0, group=process_group)
                    else:                                   # gloo + GPU tensors (rehearsal): through host memory
                        h = v.cpu()
                        dist.broadcast(h, src=0, group=process_group)
                        v.copy_(h)
        runner_of(module).grad_sync = self.sync
        runner_of(module).wepoch += 1            # the broadcast rewrote the parameters in place

    def forward(self, x):
        return self.module(x)
