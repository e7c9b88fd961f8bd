"""Workers for tests/test_ddp_cpu.py: one rank of a world_size-N gloo job on CPU (run via torch.multiprocessing)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class FakeState:
    pass


def block_params(net):
    """[w, b, gamma, beta] per conv block in execution order — the executor's flat parameter list."""
    out = []
    for m in net.modules():
        if hasattr(m, "conv") and isinstance(getattr(m, "conv"), torch.nn.Sequential) and isinstance(m.conv[0], torch.nn.Conv2d):
            out += [m.conv[0].weight, m.conv[0].bias, m.conv[1].weight, m.conv[1].bias]
    return out


def run(rank, world, port, out_dir, bucket_mb):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pytorch_camvid_amd import ddp, engine
    from oracle import torch_ref as R
    torch.set_num_threads(2)
    torch.manual_seed(0)
    net = R.build("unet", 3, 12).train()
    params = block_params(net)
    assert len(params) == 92
    x, t = R.synthetic_batch(1, 32, 32, 1234 + rank)        # this rank's shard of the global batch
    R.fwd_bwd_step(net, x, t)
    st = FakeState()
    st.params = params
    st.goffs, total = engine.layout_grads(params)
    st.gflat = torch.zeros(total)
    sync = ddp.GradSync(bucket_mb=bucket_mb)
    call = sync.begin(st)
    nslots = len(params) // 4
    for slot in range(nslots - 1, -1, -1):                   # backward order: last layer first
        for j in range(4):
            p = params[4 * slot + j]
            o = st.goffs[4 * slot + j]
            st.gflat[o:o + p.numel()] = p.grad.reshape(-1)
        call.layer_done(st, slot)
    call.finish(st)
    torch.save({"flat": st.gflat, "offs": st.goffs, "launched": sync.launched, "total": total,
                "local": [p.grad.clone() for p in params]}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def run_train(rank, world, port, out_dir, bucket_mb, steps):
    """The loop of examples/train_synthetic.py (reference train.py:124-134: zero_grad -> net(x) -> CE -> backward -> AdamW step -> OneCycleLR
    step) with the gradient exchange of ddp.GradSync between backward and the optimizer, `steps` times, every rank on its own shard and with its
    OWN initial weights until the rank-0 broadcast.  What 8 GPUs would show first if the exchange were wrong is drift between the ranks'
    parameters, so the parameters after the last step are what the test compares."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pytorch_camvid_amd import ddp, engine
    from oracle import torch_ref as R
    torch.set_num_threads(2)
    torch.manual_seed(100 + rank)                            # different init per rank: the broadcast must make them equal
    net = R.build("unet", 3, 12).train()
    with torch.no_grad():                                    # what ddp.DataParallel.__init__ does for the product network
        for t in list(net.parameters()) + list(net.buffers()):
            dist.broadcast(t, src=0)
    params = block_params(net)
    opt = torch.optim.AdamW(net.parameters(), lr=5e-4, weight_decay=0.0)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=5e-4, steps_per_epoch=steps, epochs=1)
    st = FakeState()
    st.params = params
    st.goffs, total = engine.layout_grads(params)
    sync = ddp.GradSync(bucket_mb=bucket_mb)
    nslots = len(params) // 4
    losses = []
    for k in range(steps):
        x, t = R.synthetic_batch(1, 32, 32, 1234 + 17 * k + rank)
        losses.append(float(R.fwd_bwd_step(net, x, t)))      # zero_grad + forward + CE + backward
        st.gflat = torch.zeros(total)                        # a fresh flat buffer per backward call, as engine.Runner.backward takes one
        call = sync.begin(st)
        for slot in range(nslots - 1, -1, -1):
            for j in range(4):
                p = params[4 * slot + j]
                o = st.goffs[4 * slot + j]
                st.gflat[o:o + p.numel()] = p.grad.reshape(-1)
                p.grad = st.gflat[o:o + p.numel()].view(p.shape)     # .grad is a view of the flat buffer (engine.Runner.backward)
            call.layer_done(st, slot)
        call.finish(st)
        opt.step()
        sched.step()
    torch.save({"params": [p.detach().clone() for p in net.parameters()], "losses": losses,
                "bn": [b.detach().clone() for b in net.buffers()]}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()
