"""Round-2 GPU tests: checkpoint round trip through the HIP network, torch.jit.trace survivability (reference
train.py:97), two passes through one network inside one autograd graph, optimizer state, prefetcher slot reuse."""
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def test_checkpoint_reference_layout_load_eval_roundtrip(tmp_path):
    """A reference-layout .pth (dense OIHW weights, int64 num_batches_tracked, non-trivial running statistics) written
    from the stock-torch rebuild -> load into the HIP net -> the eval-mode logits of the file's weights -> save again
    -> bit-equal file contents (reference train.py:88-93,232-240)."""
    import pytorch_camvid_amd as A
    from oracle import torch_ref as R
    for kind, shape in (("unet", (2, 45, 60)), ("segnet", (2, 64, 96))):
        torch.manual_seed(11)
        ref = R.build(kind, 3, 12).train()
        x, t = R.synthetic_batch(shape[0], shape[1], shape[2], 21)
        with torch.no_grad():
            for _ in range(2):
                ref(x)                                    # two training-mode passes: running stats leave their init values
        ref.eval()
        with torch.no_grad():
            want = ref(x)
        src = str(tmp_path / "checkpoints" / "ref" / "30-best.pth")
        os.makedirs(os.path.dirname(src))
        torch.save(ref.state_dict(), src)
        torch.manual_seed(12)
        net = A.get_model(kind, 3, 12).to(dev())
        trained, path = A.resume(net, str(tmp_path / "checkpoints"))
        assert trained == 30 and path == os.path.abspath(src)
        net.eval()
        with torch.no_grad():
            got = net(x.to(dev()))
        np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-3, atol=3e-4)
        assert (A.argmax_channels(got).cpu() == want.argmax(1)).float().mean() > 0.995
        out = A.save_checkpoint(net, str(tmp_path / "checkpoints" / "mine"), 31, "regular")
        a = torch.load(src); b = torch.load(out, map_location="cpu")
        assert list(a.keys()) == list(b.keys())
        for k in a:
            assert a[k].dtype == b[k].dtype and b[k].is_contiguous() and torch.equal(a[k], b[k]), k
        import shutil
        shutil.rmtree(tmp_path / "checkpoints")


def test_jit_trace_survives_like_add_graph():
    """reference train.py:97 -> utils.py:10-13: `writer.add_graph(net, tensor)` = torch.jit.trace(net, tensor,
    strict=False) under no_grad on a train-mode net with a (1,3,480,360) tensor.  tensorboard is not installed here, so
    the trace call itself is what is exercised, with its default re-run check."""
    import pytorch_camvid_amd as A
    torch.manual_seed(0)
    net = A.UNet(3, 12).to(dev())                        # train mode, as at train.py:97
    tensor = torch.randn(1, 3, 96, 64).to(next(net.parameters()).device)
    nbt0 = int(net.down1[0].conv[1].num_batches_tracked)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        traced = torch.jit.trace(net, tensor, strict=False)
        out = traced(tensor)
    assert tuple(out.shape) == (1, 12, 96, 64) and torch.isfinite(out).all()
    assert int(net.down1[0].conv[1].num_batches_tracked) > nbt0          # the traced passes update BN statistics like the reference's
    # the network still trains normally afterwards
    x = torch.randn(2, 3, 48, 64, device=dev()); t = torch.randint(0, 12, (2, 48, 64), device=dev())
    A.CrossEntropyLoss()(net(x), t).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


def test_two_passes_through_one_network_in_one_graph():
    """loss = L(net(x1)) + L(net(x2)) must give g1 + g2, and gradients the caller keeps across
    zero_grad(set_to_none=True) must not be overwritten by the next backward."""
    import pytorch_camvid_amd as A
    from oracle import torch_ref as R
    torch.manual_seed(0)
    net = A.UNet(3, 12).to(dev()).train()
    torch.manual_seed(0)
    ref = R.build("unet", 3, 12).train()
    x1, t1 = R.synthetic_batch(2, 48, 64, 31)
    x2, t2 = R.synthetic_batch(2, 48, 64, 32)
    lossf = A.CrossEntropyLoss()
    (lossf(net(x1.to(dev())), t1.to(dev())) + lossf(net(x2.to(dev())), t2.to(dev()))).backward()
    both = [p.grad.clone() for p in net.parameters()]
    singles = []
    for x, t in ((x1, t1), (x2, t2)):
        for p in net.parameters():
            p.grad = None
        lossf(net(x.to(dev())), t.to(dev())).backward()
        singles.append([p.grad for p in net.parameters()])          # kept across the next zero_grad + backward on purpose
    for i, (g, a, b) in enumerate(zip(both, *singles)):
        assert torch.allclose(g, a + b, rtol=1e-5, atol=1e-10), i
    assert not torch.equal(singles[0][0], singles[1][0])             # the kept gradients of pass 1 were not overwritten by pass 2
    (torch.nn.functional.cross_entropy(ref(x1), t1) + torch.nn.functional.cross_entropy(ref(x2), t2)).backward()
    gw = dict(net.named_parameters())["output.conv.0.weight"]
    k = list(dict(net.named_parameters())).index("output.conv.0.weight")
    rw = dict(ref.named_parameters())["output.conv.0.weight"].grad
    assert float((both[k].cpu() - rw).norm() / rw.norm()) < 2e-3
    ag = torch.autograd.grad(lossf(net(x1.to(dev())), t1.to(dev())), list(net.parameters()))
    lossf(net(x2.to(dev())), t2.to(dev())).backward()
    assert torch.equal(ag[0], singles[0][0])                         # autograd.grad results survive a later backward


def test_flat_adamw_state_dict_and_rehome_guard():
    import pytorch_camvid_amd as A
    torch.manual_seed(0)
    net = A.UNet(3, 12).to(dev()).train()
    opt = A.FlatAdamW(net, lr=1e-3, weight_decay=0.0)
    x = torch.randn(2, 3, 32, 32, device=dev()); t = torch.randint(0, 12, (2, 32, 32), device=dev())
    lossf = A.CrossEntropyLoss()
    for _ in range(2):
        opt.zero_grad(); lossf(net(x), t).backward(); opt.step()
    sd = opt.state_dict()
    assert sd["flat_adamw"]["step"] == 2 and float(sd["flat_adamw"]["exp_avg_sq"].sum()) > 0
    net_sd = {k: v.clone() for k, v in net.state_dict().items()}
    opt.zero_grad(); lossf(net(x), t).backward(); opt.step()
    after3 = [p.detach().clone() for p in net.parameters()]
    # resume: fresh net + optimizer, load both states, take the same third step -> identical parameters
    torch.manual_seed(1)
    net2 = A.UNet(3, 12).to(dev()).train()
    net2.load_state_dict(net_sd)
    opt2 = A.FlatAdamW(net2, lr=1e-3, weight_decay=0.0)
    opt2.load_state_dict(sd)
    opt2.zero_grad(); lossf(net2(x), t).backward(); opt2.step()
    for a, b in zip(after3, net2.parameters()):
        assert torch.equal(a, b)
    net2.float().cpu()
    net2.to(dev())                                                    # re-homes the parameters away from the flat buffer
    lossf(net2(x), t).backward()
    with pytest.raises(RuntimeError, match="flat buffer"):
        opt2.step()


def test_unet_fullsize_batch2_all_three_losses():
    """All three reference losses of the 2x3x360x480 golden: AdamW + OneCycleLR(steps_per_epoch=300,
    epochs=1), tolerance per step from tests/golden/drift.json (fp32-vs-fp64 drift of the reference graph)."""
    import json
    import pytorch_camvid_amd as A
    G = os.path.join(os.path.dirname(__file__), "golden")
    d = dict(np.load(os.path.join(G, "unet_s0_2x360x480.npz")))
    tol = json.load(open(os.path.join(G, "drift.json")))["trajectory_tolerance"]["unet_s0_2x360x480"]
    torch.manual_seed(0)
    net = A.UNet(3, 12).to(dev()).train()
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(2, 3, 360, 480, generator=g).to(dev()); t = torch.randint(0, 12, (2, 360, 480), generator=g).to(dev())
    opt = torch.optim.AdamW(net.parameters(), lr=5e-4, weight_decay=0)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=5e-4, steps_per_epoch=300, epochs=1)
    lossf = A.CrossEntropyLoss()
    for i in range(3):
        opt.zero_grad()
        l = lossf(net(x), t); l.backward()
        opt.step(); sched.step()
        assert abs(l.item() - float(d["traj_losses"][i])) < tol[i], (i, l.item(), float(d["traj_losses"][i]), tol[i])


def test_bench_json_contract_single_gpu():
    """`python bench.py` prints ONE JSON line with the driver's fields; the roofline fraction is a hardware fraction (executed
    MFMA FLOPs of the dominant kernel over the dense peak, <= 1) with the algorithmic rate beside it, and the memory-bound
    kernels (incl. the weight transforms and the 2-D Winograd transform passes) are listed with their share of 8 TB/s."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "images/s" and d["dtype"] == "f32"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and "workload" in d["config"]
    assert abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 157.3 and rf["kernel"]
    assert 0.3 < rf["frac"] <= 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["algorithmic_tflops"] >= rf["achieved"] and 0 < rf["executed_share_of_algorithmic_flops"] <= 1.0
    assert all(0 < v["executed_frac_of_peak"] <= 1.0 for v in d["conv_kernels"].values())
    hk = d["hbm_kernels"]
    for name in ("k_bn_relu_apply", "k_w2d_input", "k_w2d_output", "k_weight_transform_batch", "k_ce_fwd", "k_ce_bwd", "k_bilinear_fwd"):
        assert name in hk and 0 < hk[name]["frac_of_8TBps"] <= 1.0, name
    assert "k_conv3x3_wino4f" in d["conv_kernels"]          # the fused F(4,3) kernel carries the 64/128-channel levels
    assert "configs[1]" in d["config"]["workload"]
    # the other single-GPU configurations of BASELINE.json ride on the same line
    ex = d["extra_configs"]
    assert len(ex) == 4 and "configs[3]" in ex[0]["workload"] and "configs[4]" in ex[1]["workload"]
    assert ex[0]["dtype"] == "bf16" and ex[0]["peak_tflops"] == 2500.0 and ex[1]["dtype"] == "f32" and ex[1]["peak_tflops"] == 157.3
    # the opt-in split-operand mode rides along as a labelled extra leg of the headline workload: never the headline itself
    assert "configs[1]" in ex[2]["workload"] and "OPT-IN" in ex[2]["workload"] and "split" in ex[2]["dtype"]
    assert "k_gemm_split3" in ex[2]["top_kernels_ms_per_step"] and "split" not in rf["kernel"] and not any("split" in k for k in d["conv_kernels"])
    assert "configs[1]" in ex[3]["workload"] and "OPT-IN" in ex[3]["workload"] and "fp16" in ex[3]["dtype"]
    assert any("split2h" in k for k in ex[3]["top_kernels_ms_per_step"])
    for e in ex:
        for k in ("workload", "dtype", "images_per_s", "ms_per_step", "dominant_kernel", "executed_frac_of_peak", "loss"):
            assert k in e, k
        assert e["images_per_s"] > 0 and 0 < e["executed_frac_of_peak"] <= 1.0 and 2.0 < e["loss"] < 3.0
        # every printed fraction is a hardware fraction: each kernel against its own pipe's peak (the split legs mix two pipes)
        assert 0 < e["all_conv_kernels"]["executed_frac_of_peak"] <= 1.0, e["all_conv_kernels"]
    assert 0 < rf["all_conv_kernels"]["executed_frac_of_peak"] <= 1.0
