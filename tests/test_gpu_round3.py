"""Round-3 host-side machinery on the GPU: the derived-weight cache with explicit invalidation and the
training step replayed from ONE captured HIP graph (item 6).  Both must be invisible in the numbers: every comparison
below is bitwise against the plain eager path."""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def batch(n, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, 3, h, w, generator=g).to(dev()), torch.randint(0, 12, (n, h, w), generator=g).to(dev())


def _step(net, lossf, x, t):
    for p in net.parameters():
        p.grad = None
    loss = lossf(net(x), t)
    loss.backward()
    return loss


def test_weight_cache_hits_and_every_invalidation_path():
    """reference train.py:124-134: weights change only in optimizer.step(); between steps (and in eval, train.py:169-206)
    the Winograd-domain filters / data-grad packs are reused.  Every way the weights can change must rebuild them."""
    import pytorch_camvid_amd as A
    from pytorch_camvid_amd.modules import runner_of
    torch.manual_seed(0)
    net = A.UNet(3, 12).to(dev()).train()
    ref = A.UNet(3, 12).to(dev()).train()
    ref.load_state_dict(net.state_dict())
    runner_of(ref).wcache = False                        # the uncached executor: ground truth for every comparison
    R = runner_of(net)
    lossf = A.CrossEntropyLoss()
    x, t = batch(2, 48, 64, 3)

    def same():
        la, lb = _step(net, lossf, x, t), _step(ref, lossf, x, t)
        assert la.item() == lb.item()
        for (k, p), q in zip(net.named_parameters(), ref.parameters()):
            assert torch.equal(p.grad, q.grad), k

    same()
    built = R.wcache_builds
    assert built > 0
    net.eval(); ref.eval()
    with torch.no_grad():
        assert torch.equal(net(x), ref(x))
        built = R.wcache_builds
        assert torch.equal(net(x), ref(x))
    assert R.wcache_builds == built, "a second eval-mode forward over unchanged weights rebuilt derived tensors"
    net.train(); ref.train()
    # 0. FAIL-SAFE: a training pass rebuilds its derived tensors, so even a write nobody can see is picked up there
    with torch.no_grad():
        for p, q in zip(net.parameters(), ref.parameters()):
            if p.dim() == 4:
                p.data.mul_(0.8); q.data.mul_(0.8)
    same()
    assert R.wcache_builds > built
    built = R.wcache_builds
    # 1. a torch optimizer step (global post-step hook)
    oa, ob = torch.optim.SGD(net.parameters(), lr=0.1), torch.optim.SGD(ref.parameters(), lr=0.1)
    oa.step(); ob.step()
    same()
    assert R.wcache_builds > built
    built = R.wcache_builds
    # 2. EVAL passes share entries across calls: there an in-place write nobody can see needs mark_weights_dirty (documented contract)
    net.eval(); ref.eval()
    with torch.no_grad():
        assert torch.equal(net(x), ref(x))
        for p, q in zip(net.parameters(), ref.parameters()):
            if p.dim() == 4:
                p.data.mul_(1.25); q.data.mul_(1.25)
        A.mark_weights_dirty(net)
        assert torch.equal(net(x), ref(x))
    net.train(); ref.train()
    same()
    built = R.wcache_builds
    # 3. load_state_dict
    sd = {k: (v * 0.5 if v.dim() == 4 else v.clone()) for k, v in net.state_dict().items()}
    net.load_state_dict(sd); ref.load_state_dict(sd)
    same()
    assert R.wcache_builds > built
    built = R.wcache_builds
    # 4. autograd-visible in-place op (version counter)
    with torch.no_grad():
        for p, q in zip(net.parameters(), ref.parameters()):
            if p.dim() == 4:
                p.mul_(0.9); q.mul_(0.9)
    same()
    assert R.wcache_builds > built
    built = R.wcache_builds
    # 5. the raw-pointer optimizer
    fa, fb = A.FlatAdamW(net, lr=1e-2), A.FlatAdamW(ref, lr=1e-2)
    same()
    fa.step(); fb.step()
    built = R.wcache_builds
    same()
    assert R.wcache_builds > built


def test_graphed_step_is_bitwise_the_eager_step_and_cheap_to_enqueue():
    """zero_grad -> net(x) -> CE -> backward (reference train.py:124-131) replayed from one HIP graph: loss, every gradient
    and the BatchNorm running statistics equal the eager step bit for bit over several optimizer steps (the replay recomputes
    the derived weights from the live parameters), and a replay costs the host far less than the ~500 eager launches."""
    import pytorch_camvid_amd as A
    torch.manual_seed(0)
    net = A.UNet(3, 12).to(dev()).train()
    ref = A.UNet(3, 12).to(dev()).train()
    ref.load_state_dict(net.state_dict())
    lossf = A.CrossEntropyLoss()
    x0, t0 = batch(2, 48, 64, 5)
    gs = A.GraphedStep(net, lossf, x0, t0)               # capture (its warm-up passes advance the BN statistics: resync below)
    net.load_state_dict(ref.state_dict())
    oa, ob = torch.optim.SGD(net.parameters(), lr=0.05), torch.optim.SGD(ref.parameters(), lr=0.05)
    for it in range(3):
        x, t = batch(2, 48, 64, 10 + it)
        la = gs.replay(x, t)
        lb = _step(ref, lossf, x, t)
        assert la.item() == lb.item(), it
        for (k, p), q in zip(net.named_parameters(), ref.parameters()):
            assert torch.equal(p.grad, q.grad), (it, k)
        oa.step(); ob.step()
    for (k, a), b in zip(net.state_dict().items(), ref.state_dict().values()):
        assert torch.equal(a, b), k
    # host cost of enqueueing: one graph launch vs the eager launch sequence (tiny geometry: the GPU work vanishes)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(20):
        gs.replay()
    tg = (time.perf_counter() - t1) / 20
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        _step(ref, lossf, x0, t0)
    te = (time.perf_counter() - t1) / 5
    torch.cuda.synchronize()
    assert tg < 3e-3, f"graph replay costs the host {tg * 1e3:.2f} ms per step"
    assert tg < 0.5 * te, (tg, te)


def test_graphed_step_refuses_data_parallel():
    import pytorch_camvid_amd as A
    from pytorch_camvid_amd.modules import runner_of
    net = A.UNet(3, 12).to(dev()).train()
    runner_of(net).grad_sync = object()
    x, t = batch(1, 32, 48, 1)
    with pytest.raises(RuntimeError, match="allow_grad_sync"):
        A.GraphedStep(net, A.CrossEntropyLoss(), x, t)
    with pytest.raises(RuntimeError, match="RCCL"):                       # a synchroniser that is not an RCCL group cannot be captured
        A.GraphedStep(net, A.CrossEntropyLoss(), x, t, allow_grad_sync=True)
    runner_of(net).grad_sync = None


def test_graphed_step_replay_refuses_a_changed_network():
    """a captured graph bakes in train mode, the conv precision, the kernel knobs and the gradient synchroniser; replay()
    after any of them changed would silently run the OLD configuration (e.g. skip every all-reduce) — it raises instead."""
    import pytorch_camvid_amd as A
    from pytorch_camvid_amd.modules import runner_of
    net = A.UNet(3, 12).to(dev()).train()
    x, t = batch(1, 32, 48, 1)
    gs = A.GraphedStep(net, A.CrossEntropyLoss(), x, t)
    gs.replay()
    net.eval()
    with pytest.raises(RuntimeError, match="changed since the capture"):
        gs.replay()
    net.train()
    gs.replay()
    runner_of(net).grad_sync = object()                                   # e.g. wrapped in ddp.DataParallel afterwards
    with pytest.raises(RuntimeError, match="changed since the capture"):
        gs.replay()
    runner_of(net).grad_sync = None
    A.set_conv_precision(net, "bf16")
    with pytest.raises(RuntimeError, match="changed since the capture"):
        gs.replay()
    A.set_conv_precision(net, "fp32")
    gs.replay()


@pytest.mark.parametrize("Co,Ci", [(64, 32), (128, 256), (96, 160), (40, 72)])
def test_w2d_data_grad_filter_straight_from_the_forward_weights(Co, Ci):
    """cvk_w2d_weight_transform_dgrad(w) == cvk_w2d_weight_transform(cvk_pack_weight_dgrad(w)) bit for bit: the data-grad
    of nn.Conv2d (backward of train.py:131) uses the 180-degree rotated, channel-exchanged filter; the packed copy is gone."""
    from pytorch_camvid_amd import _lib
    lib, check = _lib.load(), _lib.check
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(Co + Ci)
    w = torch.randn(Co, 3, 3, Ci, generator=g).to(dev())                 # channels_last physical layout of an OIHW parameter
    packed = torch.empty(Ci, 9, Co, device=dev())
    check(lib.cvk_pack_weight_dgrad(w.data_ptr(), packed.data_ptr(), Co, Ci, Ci, Co, s), "pack")
    want = torch.empty(36, Ci, Co, device=dev())
    check(lib.cvk_w2d_weight_transform(packed.data_ptr(), want.data_ptr(), Ci, Co, s), "transform")
    got = torch.full((36, Ci, Co), float("nan"), device=dev())
    check(lib.cvk_w2d_weight_transform_dgrad(w.data_ptr(), got.data_ptr(), Co, Ci, s), "direct")
    torch.cuda.synchronize()
    assert torch.equal(got, want)


def test_batched_filter_transforms_equal_the_single_launches():
    """cvk_wino4f_weight_transform_batch / cvk_w2d_weight_transform_batch (round 4: the Winograd-domain filters of a whole step in two
    launches) against the per-layer entry points: the same device functions, so bitwise — forward and data-grad filters, both 2-D tiles."""
    import ctypes
    from pytorch_camvid_amd import _lib
    lib, check = _lib.load(), _lib.check
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(11)
    shapes = [(64, 64), (128, 64), (96, 160), (256, 128)]
    ws = [torch.randn(co, 3, 3, ci, generator=g).to(dev()) for co, ci in shapes]
    # fused F(4,3)
    jobs, want, got = [], [], []
    for w, (co, ci) in zip(ws, shapes):
        for dgrad in (0, 1):
            cn, ck = (co, ci) if not dgrad else (ci, co)
            n = lib.cvk_wino4f_weight_floats(cn, ck)
            a = torch.full((n,), float("nan"), device=dev()); b = torch.full((n,), float("nan"), device=dev())
            check(lib.cvk_wino4f_weight_transform(w.data_ptr(), a.data_ptr(), cn, ck, dgrad, s), "single")
            jobs.append(_lib.WtJob(w.data_ptr(), b.data_ptr(), cn, ck, 0, dgrad)); want.append(a); got.append(b)
    arr = (_lib.WtJob * len(jobs))(*jobs)
    check(lib.cvk_wino4f_weight_transform_batch(ctypes.addressof(arr), len(jobs), s), "batch")
    torch.cuda.synchronize()
    for a, b in zip(want, got):
        assert torch.equal(a, b)
    # 2-D F(4x4) / F(6x6)
    jobs, want, got = [], [], []
    for w, (co, ci) in zip(ws, shapes):
        for tile, nx in ((4, 36), (6, 64)):
            for dgrad in (0, 1):
                a = torch.full((nx * co * ci,), float("nan"), device=dev()); b = torch.full((nx * co * ci,), float("nan"), device=dev())
                fn = getattr(lib, ("cvk_w2d_" if tile == 4 else "cvk_w6_") + ("weight_transform_dgrad" if dgrad else "weight_transform"))
                check(fn(w.data_ptr(), a.data_ptr(), co, ci, s), "single")
                jobs.append(_lib.WtJob(w.data_ptr(), b.data_ptr(), co, ci, tile, dgrad)); want.append(a); got.append(b)
    arr = (_lib.WtJob * len(jobs))(*jobs)
    check(lib.cvk_w2d_weight_transform_batch(ctypes.addressof(arr), len(jobs), s), "batch")
    torch.cuda.synchronize()
    for a, b in zip(want, got):
        assert torch.equal(a, b)


def test_recorded_filter_builds_are_replayed_in_two_launches():
    """The second training pass over a plan rebuilds its Winograd-domain filters through Runner.prebuild_fp32 (recorded by the first pass):
    same losses and gradients as the uncached executor, and the per-layer transform entry points are no longer called."""
    import pytorch_camvid_amd as A
    from pytorch_camvid_amd import engine
    from pytorch_camvid_amd.modules import runner_of
    torch.manual_seed(0)
    net = A.UNet(3, 12).to(dev()).train()
    ref = A.UNet(3, 12).to(dev()).train()
    ref.load_state_dict(net.state_dict())
    runner_of(ref).wcache = False
    runner_of(net).w2d_split = runner_of(ref).w2d_split = 0        # the default path's filter batching (the opt-in split forms build theirs per layer)
    lossf = A.CrossEntropyLoss()
    x, t = batch(2, 96, 128, 3)
    for it in range(3):
        la, lb = _step(net, lossf, x, t), _step(ref, lossf, x, t)
        assert la.item() == lb.item(), it
        for (k, p), q in zip(net.named_parameters(), ref.parameters()):
            assert torch.equal(p.grad, q.grad), (it, k)
    R = runner_of(net)
    assert len(R._wjobs) > 0
    engine.PROF = []
    try:
        _step(net, lossf, x, t)
        names = [r[0] for r in engine.PROF]
    finally:
        engine.PROF = None
    assert "k_weight_transform_batch" in names
    assert not any(n in ("k_wino4f_weight", "k_w2d_weight", "k_w2d_weight_dgrad") for n in names), sorted(set(names))
