"""Pins the oracle (oracle/np_ops.py, oracle/torch_ref.py) to golden vectors produced by importing the
reference (tests/golden/make_golden.py).  CPU only."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from oracle import np_ops as O
from oracle import torch_ref as R

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return dict(np.load(os.path.join(G, name), allow_pickle=False))


def close(a, b, rtol=1e-4, atol=1e-5):
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=atol)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "basicconv_*.npz"))))
def test_basicconv_block(path):
    d = dict(np.load(path))
    p = {k[2:]: v for k, v in d.items() if k.startswith("p.")}
    out, cache = O.basic_conv_fwd(d["x"], p, "", train=True)
    close(out, d["y_train"], 2e-4, 2e-5)
    dx, grads = O.basic_conv_bwd(d["r"], cache, p)
    close(dx, d["dx"], 1e-3, 2e-4)
    for k in ("conv.0.weight", "conv.1.weight", "conv.1.bias"):
        close(grads[k], d["g." + k], 1e-3, 3e-4)
    # conv bias grad is mathematically 0 under train-mode BN: absolute tolerance only (SURVEY §7 hard part 3)
    assert np.abs(grads["conv.0.bias"]).max() < 1e-6 and np.abs(d["g.conv.0.bias"]).max() < 1e-3
    m = d["x"].shape[0] * d["x"].shape[2] * d["x"].shape[3]
    rm, rv = O.bn_running_update(p["conv.1.running_mean"], p["conv.1.running_var"], cache["mean"], cache["var"], m)
    close(rm, d["after.conv.1.running_mean"], 1e-4, 1e-6)
    close(rv, d["after.conv.1.running_var"], 1e-4, 1e-6)
    p2 = dict(p); p2["conv.1.running_mean"] = d["after.conv.1.running_mean"]; p2["conv.1.running_var"] = d["after.conv.1.running_var"]
    oe, _ = O.basic_conv_fwd(d["x"], p2, "", train=False)
    close(oe, d["y_eval"], 2e-4, 2e-5)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "upsample2d_*.npz"))))
def test_upsample_block(path):
    d = dict(np.load(path))
    p = {k[2:]: v for k, v in d.items() if k.startswith("p.")}
    up = O.bilinear_up2_fwd(d["x"])
    close(up, d["up_only"], 1e-5, 1e-6)
    close(O.bilinear_up2_bwd(d["r_up"]), d["dx_up_only"], 1e-5, 1e-5)
    out, cache = O.basic_conv_fwd(up, p, "conv.", train=True)
    close(out, d["y_train"], 2e-4, 2e-5)
    du, grads = O.basic_conv_bwd(d["r"], cache, p)
    close(O.bilinear_up2_bwd(du), d["dx"], 1e-3, 2e-4)
    close(grads["conv.conv.0.weight"], d["g.conv.conv.0.weight"], 1e-3, 3e-4)


def test_pool_cat_ce_unpool():
    d = load("ops_pool_cat_ce.npz")
    for t in "abc":
        y, idx = O.maxpool2x2_fwd(d[f"pool_{t}_x"])
        close(y, d[f"pool_{t}_y"], 0, 0)
        close(O.maxpool2x2_bwd(d[f"pool_{t}_r"], idx, d[f"pool_{t}_x"].shape), d[f"pool_{t}_dx"], 0, 0)
        c = O.pad_cat_fwd(d[f"cat_{t}_up"], d[f"cat_{t}_skip"])
        close(c, d[f"cat_{t}_out"], 0, 0)
        du, ds = O.pad_cat_bwd(d[f"cat_{t}_r"], d[f"cat_{t}_up"].shape)
        close(du, d[f"cat_{t}_dup"], 0, 0); close(ds, d[f"cat_{t}_dskip"], 0, 0)
        loss, sm = O.cross_entropy_fwd(d[f"ce_{t}_logits"], d[f"ce_{t}_target"])
        close(loss, d[f"ce_{t}_loss"], 1e-6, 1e-7)
        close(O.cross_entropy_bwd(sm, d[f"ce_{t}_target"]), d[f"ce_{t}_dlogits"], 1e-5, 1e-8)
    for t in "ab":
        x = d[f"unpool_{t}_x"]
        y, idx = O.maxpool2x2_fwd(x)
        close(y, d[f"unpool_{t}_y"], 0, 0)
        assert np.array_equal(idx, d[f"unpool_{t}_idx"])           # first-max tie rule, flat H*W indices
        z = O.maxunpool2x2_fwd(y, idx, x.shape)
        close(z, d[f"unpool_{t}_z"], 0, 0)
        close(O.maxpool2x2_bwd(O.maxunpool2x2_bwd(d[f"unpool_{t}_r"], idx), idx, x.shape), d[f"unpool_{t}_dx"], 0, 0)


def test_miou_histograms():
    d = load("miou_intersect_union.npz")
    for t in "ab":
        ti = np.zeros(12); tu = np.zeros(12)
        for i in range(d[f"{t}_pred"].shape[0]):
            a, b, _, _ = O.intersect_and_union(d[f"{t}_pred"][i].astype(np.int64), d[f"{t}_label"][i].astype(np.int64), 12, 11)
            ti += a; tu += b
        assert np.array_equal(ti, d[f"{t}_inter"]) and np.array_equal(tu, d[f"{t}_union"])


NETS = ["unet_s0_2x48x64", "unet_s1_1x45x60", "unet_s2_2x36x52", "segnet_s0_2x64x96", "segnet_s3_2x45x60"]


@pytest.mark.parametrize("tag", NETS)
def test_torch_rebuild_bit_identical_init_and_forward(tag):
    """Weights by recipe: same seed + same construction order => the rebuild reproduces the reference logits."""
    d = load(tag + ".npz")
    meta = json.loads(str(d["meta"]))
    torch.manual_seed(meta["seed"])
    net = R.build(meta["kind"], 3, 12)
    assert [k for k, _ in net.named_parameters()] == list(d["param_names"])
    n, _, h, w = meta["shape"]
    x, t = R.synthetic_batch(n, h, w, meta["data_seed"])
    net.train()
    out = net(x)
    loss = torch.nn.functional.cross_entropy(out, t)
    loss.backward()
    close(out.detach().numpy(), d["logits"], 1e-5, 1e-6)
    close(loss.item(), d["loss"], 1e-6, 0)
    gl2 = np.array([float(p.grad.double().norm()) for p in net.parameters()])
    big = d["grad_l2"] > 1e-6
    close(gl2[big], d["grad_l2"][big], 1e-3, 0)
    pl2 = np.array([float(p.detach().double().norm()) for p in net.parameters()])
    close(pl2, d["param_l2"], 1e-7, 0)                                 # init is bit-identical


def test_numpy_unet_matches_golden_small():
    """The numpy restatement of the whole UNet (fwd + bwd) against the reference golden at 1x3x45x60 (odd sizes)."""
    d = load("unet_s1_1x45x60.npz")
    meta = json.loads(str(d["meta"]))
    torch.manual_seed(meta["seed"])
    net = R.build("unet", 3, 12)
    p = {k: v.detach().numpy() for k, v in net.state_dict().items()}
    x, t = R.synthetic_batch(1, 45, 60, meta["data_seed"])
    logits, caches = O.unet_forward(x.numpy(), p, train=True)
    close(logits, d["logits"], 2e-3, 2e-4)
    loss, sm = O.cross_entropy_fwd(logits, t.numpy())
    close(loss, d["loss"], 1e-5, 0)
    _, grads = O.unet_backward(O.cross_entropy_bwd(sm, t.numpy()), caches, p)
    names = list(d["param_names"])
    # Whole-net gradients are only comparable loosely: the reference's own fp32 run deviates from an fp64 run
    # of the same torch graph by up to ~18 % on single tensors at this size (6-sample BN at the bottleneck and
    # ReLU-mask flips; measured with oracle/torch_ref in float vs double).  Tight gradient parity is pinned
    # per operator above; here: every tensor within 25 % L2, and 80 % of tensors within 1 %.
    rel = []
    for i, k in enumerate(names):
        if k.endswith("conv.0.bias"):
            assert np.abs(grads[k]).max() < 1e-6
            continue
        gl2 = np.sqrt((grads[k] ** 2).sum())
        rel.append(abs(gl2 - d["grad_l2"][i]) / d["grad_l2"][i])
        err = np.abs(grads[k].ravel()[:64] - d["gs." + k]).max()
        assert err <= 0.25 * d["grad_absmax"][i] + 1e-8, (k, err)
    rel = np.array(rel)
    assert rel.max() < 0.25 and np.mean(rel < 1e-2) >= 0.8, rel


def test_fullsize_golden_present():
    d = load("unet_s0_2x360x480.npz")
    assert abs(float(d["traj_losses"][0]) - 2.677795) < 1e-5      # SURVEY §8a row T probe value
