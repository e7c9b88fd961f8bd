#!/usr/bin/env python3
"""Run the configs[0] protocol (tests/golden/protocol_data.py) through the engine in a given precision and print / save the curve.
    python tools/run_protocol.py [fp32|bf16] [out.npz]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import pytorch_camvid_amd as A          # noqa: E402
from protocol_data import PROTO as P, proto_batch       # noqa: E402


def run(precision):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = A.UNet(3, 12).to(dev).train()
    if precision != "fp32":
        A.set_conv_precision(net, precision)
    opt = torch.optim.AdamW(net.parameters(), lr=P["lr"], weight_decay=0)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=P["lr"], steps_per_epoch=P["steps"], epochs=1)
    lossf = A.CrossEntropyLoss()
    losses = []
    for it in range(P["steps"]):
        x, m = proto_batch(it)
        opt.zero_grad()
        loss = lossf(net(x.to(dev)), m.to(dev))
        loss.backward()
        opt.step(); sched.step()
        losses.append(loss.detach())
    losses = torch.stack(losses).cpu().numpy()
    val = [tuple(t.to(dev) for t in proto_batch(i, val=True)) for i in range(P["val_batches"])]
    rep = A.evaluate_report(net, val, num_classes=12, ignore_index=11)
    return losses, rep


if __name__ == "__main__":
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    G = os.path.join(ROOT, "tests", "golden")
    r0 = dict(np.load(os.path.join(G, "protocol_unet_2x360x480_run0.npz")))
    losses, rep = run(prec)
    d = np.abs(losses - r0["losses"])
    print(prec, "max |loss - ref|", d.max(), "at", int(d.argmax()), "final", losses[-1], "ref", r0["losses"][-1])
    print("every 25th:", [(i, round(float(losses[i]), 4), round(float(r0["losses"][i]), 4)) for i in range(0, 300, 25)])
    print("last-20 mean", losses[-20:].mean(), "ref", r0["losses"][-20:].mean())
    print("mIoU", rep["miou"], "ref", float(r0["miou"]), "val loss", rep["loss"], "ref", float(np.mean(r0["val_loss"])))
    if len(sys.argv) > 2:
        np.savez_compressed(sys.argv[2], losses=losses, miou=rep["miou"], val_loss=rep["loss"], iou=rep["iou"].numpy())
