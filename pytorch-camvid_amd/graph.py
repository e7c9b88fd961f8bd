"""The training step as ONE captured HIP graph.

A step of the path is ~500 kernel launches enqueued from Python through ctypes (11-15 ms of host time per step: hidden while
the GPU needs 38 ms for the fp32 headline, the next wall for the bf16 configuration and a jitter source for 8 single-threaded
ranks).  The executor's plan is static per input geometry, every launch goes to the current HIP stream and nothing in the step
synchronises with the host, so `zero_grad -> net(x) -> loss -> backward` (reference train.py:124-131) can be captured once
with torch.cuda.CUDAGraph and replayed with a single launch: activations, saved tensors, the flat gradient buffer and the
workspace come from the graph's private pool, i.e. their addresses are the same in every replay.

    step = GraphedStep(net, lossf, x, t)       # x, t: example batch of the geometry (copied into static buffers)
    for xb, tb in loader:
        loss = step.replay(xb, tb)             # p.grad of every parameter now holds this batch's gradients
        optimizer.step()

BatchNorm running statistics and num_batches_tracked are updated by the kernels on the device, exactly as in eager mode.
Data parallel (round 4): RCCL collectives are stream-capturable, so a step of a ddp.DataParallel network is captured WITH its
bucketed all-reduces (`GraphedStep(dp_net.module, ..., allow_grad_sync=True)`): every rank replays the same graph, the
collectives run on RCCL's stream inside it, and 8 single-threaded ranks no longer sit on 11-15 ms of Python enqueue per step.
gloo groups (CPU rehearsals) cannot be captured.
"""
import torch


class GraphedStep:
    def __init__(self, net, lossf, x, t, warmup=2, allow_grad_sync=False):
        from .modules import runner_of
        R = runner_of(net)
        if R.grad_sync is not None:
            if not allow_grad_sync:
                raise RuntimeError("GraphedStep: the network synchronises gradients (ddp.DataParallel); pass allow_grad_sync=True to "
                                   "capture the bucketed all-reduces with the step (RCCL only)")
            if not getattr(R.grad_sync, "_native_avg", False):
                raise RuntimeError("GraphedStep: only RCCL ('nccl') collectives can be captured; gloo groups run in eager mode")
        if not (x.is_cuda and t.is_cuda):
            raise RuntimeError("GraphedStep needs the example batch on the GPU")
        self.net, self.lossf = net, lossf
        self._runner = R
        # what the captured graph bakes in: replay() refuses to run once any of it has changed
        self._sig = self._signature()
        self.x, self.t = x.clone(), t.clone()
        self.params = [p for p in net.parameters() if p.requires_grad]
        cur = torch.cuda.current_stream(x.device)
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):               # warm-up off the capture: plans, workspace, allocator pools
            for _ in range(max(1, warmup)):
                self._zero()
                lossf(net(self.x), self.t).backward()
        cur.wait_stream(side)
        torch.cuda.synchronize(x.device)
        self._zero()
        self.graph = torch.cuda.CUDAGraph()
        # With a process group alive, ProcessGroupNCCL's watchdog thread polls the events of earlier collectives (hipEventQuery) at any
        # moment; under the default "global" capture mode such a call from ANOTHER thread fails with "operation not permitted when
        # stream is capturing" and the watchdog takes the process down (seen once in three full test runs).  "thread_local" restricts
        # the check to the capturing thread — what torch prescribes for capturing collectives.
        import torch.distributed as dist
        mode = "thread_local" if (R.grad_sync is not None or (dist.is_available() and dist.is_initialized())) else "global"
        if mode == "thread_local":
            # ... and the watchdog's list should be EMPTY when the capture begins: round 6 saw it die once in six suite runs with "operation not
            # permitted on an event last recorded in a capturing stream" (hipErrorCapturedEvent from WorkNCCL::isCompleted).  Every eager
            # collective has finished (synchronize above); three watchdog periods let the thread retire them before the first captured one exists.
            import time
            time.sleep(0.35)
        with torch.cuda.graph(self.graph, capture_error_mode=mode):
            self.loss = lossf(net(self.x), self.t)
            self.loss.backward()
        # every tensor a replay writes must stay alive as long as the graph: the gradients (views of the flat buffer the
        # capture allocated) and the runner's workspace (a later, larger eager call would otherwise release it)
        self.grads = [p.grad for p in self.params]
        self._keep = R._ws

    def _zero(self):
        for p in self.params:
            p.grad = None

    def _signature(self):
        R = self._runner
        return (id(R.grad_sync), bool(self.net.training), bool(R.bf16), R.wino, R.wino4, R.wino4f, R.wgradp, R.wino2d, R.w2tile_cfg,
                R.thin, R.persistent_wgs(), getattr(R, "w2d_split", False), getattr(R, "vplanes", None), getattr(R, "e4p", None),
                getattr(R, "pool_bnred", None), getattr(R, "bnred_fuse", None))

    def replay(self, x=None, t=None):
        """Copy a new batch into the static input buffers (optional) and replay the step.  Returns the (static) loss tensor."""
        if self._signature() != self._sig:
            raise RuntimeError("GraphedStep.replay: the network changed since the capture (train/eval mode, conv precision, a kernel "
                               "knob, or it was wrapped in / unwrapped from ddp.DataParallel): the captured graph would silently run the "
                               "old configuration — build a new GraphedStep")
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        if t is not None:
            self.t.copy_(t, non_blocking=True)
        for p, g in zip(self.params, self.grads):   # zero_grad(set_to_none=True) between steps keeps working
            if p.grad is not g:
                p.grad = g
        self.graph.replay()
        return self.loss
