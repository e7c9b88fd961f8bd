#!/usr/bin/env python3
"""The training loop of the reference (train.py:100-134 + validation :169-206) on synthetic CamVid-shaped data,
driving the MI355X-native network.  Real CamVid needs the reference's cv2/torchvision data pipeline, which is out
of scope here; everything from the tensors onward is the product path.

  python examples/train_synthetic.py --net unet --epochs 2 --iters 20 -b 8
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_synthetic.py   # data parallel
"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_camvid_amd as cvk  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-net", "--net", default="unet")
    ap.add_argument("-b", type=int, default=8)                 # reference default 10 (train.py:22)
    ap.add_argument("-lr", type=float, default=5e-4)           # train.py:23
    ap.add_argument("--epochs", type=int, default=2)           # reference 120 (train.py:24)
    ap.add_argument("--iters", type=int, default=20, help="synthetic batches per epoch")
    ap.add_argument("-wd", type=float, default=0.0)            # train.py:25
    ap.add_argument("--flat-adamw", action="store_true", help="one fused optimizer kernel (cvk.FlatAdamW)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--split-operands", type=int, default=0, choices=[0, 2, 3],
                    help="opt-in for fp32: matrix products on the 16-bit matrix pipe with split fp32 operands (cvk.set_split_operands; 2 = fp16 x 2)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        cvk.ddp.init_process_group("nccl", device_id=dev)      # RCCL with the channel count / CU reservation this path is tuned for

    torch.manual_seed(0)
    net = cvk.get_model(a.net, 3, 12).to(dev)                              # utils.get_model (utils.py:147-160)
    cvk.set_conv_precision(net, a.precision)
    cvk.set_split_operands(net, a.split_operands)
    model = cvk.ddp.DataParallel(net) if world > 1 else net
    opt = cvk.FlatAdamW(net, lr=a.lr, weight_decay=a.wd) if a.flat_adamw else \
        torch.optim.AdamW(net.parameters(), lr=a.lr, weight_decay=a.wd)     # train.py:100
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=a.lr, steps_per_epoch=a.iters, epochs=a.epochs)  # :103-104
    loss_fn = cvk.CrossEntropyLoss()                                        # train.py:105

    g = torch.Generator().manual_seed(1234 + rank)
    # a fixed synthetic "dataset": uint8 HWC frames like cv2 delivers + 12-class masks; smooth blobs so it is learnable
    base = torch.rand(a.iters, a.b, 45, 60, generator=g)
    masks = torch.nn.functional.interpolate((base * 12).floor().clamp(0, 11), size=(360, 480), mode="nearest").long()
    for epoch in range(1, a.epochs + 1):
        net.train()
        t0 = time.time()
        for it in range(a.iters):
            m = masks[it].to(dev)
            frames = ((m.unsqueeze(-1) * torch.tensor([20, 15, 10], device=dev)) % 256 +
                      torch.randint(0, 30, (a.b, 360, 480, 3), device=dev)).clamp(0, 255).to(torch.uint8)
            images = cvk.preprocess_uint8(frames)                           # transforms.ToTensor + Normalize on device
            opt.zero_grad()                                                 # train.py:124
            preds = model(images)                                           # :128
            loss = loss_fn(preds, m)                                        # :130
            loss.backward()                                                 # :131
            opt.step(); sched.step()                                        # :133-134
        torch.cuda.synchronize()
        dt = time.time() - t0
        if rank == 0:
            print(f"epoch {epoch}: loss {loss.item():.4f}  lr {sched.get_last_lr()[0]:.6f}  "
                  f"{world * a.b * a.iters / dt:.1f} img/s (incl. data synthesis + optimizer)")
        # validation (train.py:169-206) on two of the batches
        batches = []
        for it in range(2):
            m = masks[it].to(dev)
            frames = ((m.unsqueeze(-1) * torch.tensor([20, 15, 10], device=dev)) % 256).clamp(0, 255).to(torch.uint8)
            batches.append((cvk.preprocess_uint8(frames), m))
        acc, iou, miou = cvk.evaluate(net, batches, num_classes=12, ignore_index=11)
        if rank == 0:
            print(f"          val acc {acc:.4f}  mIoU(11 classes) {miou:.4f}")
    if rank == 0:
        torch.save(net.state_dict(), "/tmp/cvk_synthetic.pth")             # train.py:232-240; loads into the reference too
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
