"""One rank of the world_size-2 rehearsal of the REAL data-parallel path (engine executor + ddp.DataParallel) on one
GPU: ranks share cuda:0 and exchange gradients over gloo (tests/test_gpu_ddp.py)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(rank, world, port, out_dir, shape, bucket_mb, train_steps=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pytorch_camvid_amd as A
    from pytorch_camvid_amd import ddp
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    torch.manual_seed(100 + rank)                            # DIFFERENT init per rank: the broadcast must make them equal
    net = A.UNet(3, 12).to(dev).train()
    dp = ddp.DataParallel(net, bucket_mb=bucket_mb)
    n, h, w = shape
    g = torch.Generator().manual_seed(1234 + rank)           # this rank's shard of the global batch
    x = torch.randn(n, 3, h, w, generator=g).to(dev)
    t = torch.randint(0, 12, (n, h, w), generator=g).to(dev)
    w0 = [p.detach().clone().cpu() for p in net.parameters()]
    loss = A.CrossEntropyLoss()(dp(x), t)
    loss.backward()
    torch.cuda.synchronize()
    out = {"w0": w0, "grads": [p.grad.detach().cpu().clone() for p in net.parameters()], "loss": loss.item(),
           "launched": dp.sync.launched, "bn": [b.detach().cpu().clone() for b in net.buffers()]}
    if train_steps:         # the same ranks go on training (one spawn for both tests: a fresh box pays ~1 min of imports per spawn)
        out["train_adamw"] = _train(A, ddp, net, dp, shape, train_steps, False, rank, dev)
        out["train_flat"] = _train(A, ddp, net, dp, shape, train_steps, True, rank, dev)
    torch.save(out, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _train(A, ddp, net, dp, shape, steps, flat, rank, dev):
    """examples/train_synthetic.py's loop through the REAL executor: ddp.DataParallel + AdamW (torch's or the fused flat one) + OneCycleLR,
    `steps` optimizer steps on this rank's shards; the parameters afterwards must be identical on every rank."""
    opt = A.FlatAdamW(net, lr=5e-4, weight_decay=0.0) if flat else torch.optim.AdamW(net.parameters(), lr=5e-4, weight_decay=0.0)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=5e-4, steps_per_epoch=steps, epochs=1)
    lossf = A.CrossEntropyLoss()
    n, h, w = shape
    losses = []
    for k in range(steps):
        g = torch.Generator().manual_seed(1234 + 17 * k + rank)
        x = torch.randn(n, 3, h, w, generator=g).to(dev)
        t = torch.randint(0, 12, (n, h, w), generator=g).to(dev)
        opt.zero_grad()
        loss = lossf(dp(x), t)
        loss.backward()
        opt.step()
        sched.step()
        losses.append(loss.item())
    torch.cuda.synchronize()
    return {"params": [p.detach().cpu().clone() for p in net.parameters()], "losses": losses, "bn": [b.detach().cpu().clone() for b in net.buffers()]}
