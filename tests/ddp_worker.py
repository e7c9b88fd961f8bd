"""Worker for tests/test_ddp_cpu.py: one rank of a world_size-2 gloo job on CPU (run via torch.multiprocessing)."""
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
