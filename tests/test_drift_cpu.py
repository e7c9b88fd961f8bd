"""The tolerances of the GPU parity tests are derived, not chosen: tests/golden/drift.json holds the measured
reproducibility of the reference graph (fp32 vs fp64, fp32 vs 1e-6 input noise) and the measured cost of bf16 storage
(oracle/bf16_emul.py vs fp32).  This CPU test re-measures the cheapest case and checks (a) the committed measurements
are of the magnitude a fresh measurement gives, (b) every committed tolerance follows from them by the stated rule."""
import importlib.util
import json
import os

import numpy as np
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


def _md():
    spec = importlib.util.spec_from_file_location("make_drift", os.path.join(G, "make_drift.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_tolerances_follow_the_rule():
    md = _md()
    d = json.load(open(os.path.join(G, "drift.json")))
    assert d["rule"]["safety"] == md.SAFETY
    for k, v in d["trajectory"].items():
        assert np.allclose(d["trajectory_tolerance"][k], md.tolerance_from(v)), k
        assert d["trajectory_tolerance"][k][0] == md.FLOOR[0]               # step 1 is a pure forward: no drift to speak of
        assert v["drift_fp64"][0] < 1e-6 and v["drift_noise"][0] < 2e-6
    for k, v in d["bf16_cost"].items():
        assert d["bf16_tolerance"][k] == md.bf16_tolerance(v), k
    for k, v in d["bf16_emul_noise"].items():
        assert d["bf16_emul_tolerance"][k] == md.bf16_emul_tolerance(v), k
    for k, v in d["logits"].items():
        assert d["logits_tolerance"][k] == md.logits_tolerance(v), k
        assert 1e-5 < v["fp64_slice_max_abs"] < 1e-3 and 1e-5 < v["noise_slice_max_abs"] < 1e-3     # rounding shows at 1e-4 in the logits
    assert "unet_s0_2x360x480" in d["logits_tolerance"]
    # every fixture the GPU tests read is there
    for k in ("unet_s0_2x48x64", "segnet_s0_2x64x96", "unet_s0_2x360x480"):
        assert k in d["trajectory_tolerance"], k
    for k in ("unet_2x96x128", "unet_4x720x960"):
        assert k in d["bf16_tolerance"] and k in d["bf16_emul_tolerance"], k
        assert os.path.exists(os.path.join(G, f"bf16emu_unet_s0_{k.split('_')[1]}.npz"))


def test_trajectory_drift_remeasured():
    """UNet 2x48x64, 4 AdamW steps: fp32 vs fp64 and one noise seed, measured here and now."""
    md = _md()
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    d = json.load(open(os.path.join(G, "drift.json")))["trajectory"]["unet_s0_2x48x64"]
    gold = dict(np.load(os.path.join(G, "unet_s0_2x48x64.npz")))
    a = md.trajectory("unet", 0, (2, 48, 64), 1234, 4, total_steps=40)
    assert np.abs(a - gold["traj_losses"]).max() < 5e-3 and abs(a[0] - gold["traj_losses"][0]) < 2e-6   # the oracle runs the reference's curve
    b = md.trajectory("unet", 0, (2, 48, 64), 1234, 4, total_steps=40, dtype=torch.float64)
    fresh = np.abs(a - b)
    committed = np.maximum(np.array(d["drift_fp64"]), np.array(d["drift_noise"]))
    # the drift is chaotic (one more sample of the same distribution): the committed bound, times the safety factor,
    # must cover a fresh measurement, and the first step (no optimizer step yet) must be exact to fp32 rounding
    tol = np.array(md.tolerance_from(d))
    assert (fresh <= tol).all(), (fresh, tol)
    assert fresh[0] < 1e-6 and committed[-1] > 1e-5


def test_bf16_emulation_rounding_points():
    """oracle/bf16_emul.py rounds exactly the tensors it says: values are bf16-representable where stored, the logits
    and parameter gradients are not rounded, and switching the rounding off reproduces the fp32 graph."""
    from oracle import torch_ref as R, bf16_emul as E
    torch.manual_seed(0)
    net = R.build("unet", 3, 12).train()
    x, t = R.synthetic_batch(1, 32, 48, 3)
    out = E.unet_forward(net, x)
    assert not torch.equal(out, E._r(out))                        # fp32 logits
    blk = net.down1[0]
    a = E.basic_conv(blk, x)
    assert torch.equal(a, E._r(a))                                # a stored activation is bf16-representable
    loss = torch.nn.functional.cross_entropy(out, t); loss.backward()
    g = net.down3[0].conv[0].weight.grad
    assert not torch.equal(g, E._r(g))                            # parameter gradients stay fp32
    saved = E._r
    try:
        E._r = lambda v: v
        torch.manual_seed(0)
        n2 = R.build("unet", 3, 12).train()
        o2 = E.unet_forward(n2, x)
        torch.manual_seed(0)
        n3 = R.build("unet", 3, 12).train()
        assert torch.allclose(o2, n3(x), rtol=1e-5, atol=1e-6)
    finally:
        E._r = saved


def test_bf16_emulation_segnet_and_input_gradient():
    """The round-4 additions to oracle/bf16_emul.py: segnet_forward is RefSegNet.forward when the rounding is switched off, its logits
    stay fp32, and the input's gradient flows through the rounded import (rounded to bf16 like the device's stored dX)."""
    from oracle import torch_ref as R, bf16_emul as E
    x, t = R.synthetic_batch(1, 32, 64, 3)
    torch.manual_seed(0)
    net = R.build("segnet", 3, 12).train()
    xg = x.clone().requires_grad_(True)
    out = E.segnet_forward(net, xg)
    assert out.shape == (1, 12, 32, 64) and not torch.equal(out, E._r(out))
    torch.nn.functional.cross_entropy(out, t).backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all() and torch.equal(xg.grad, E._r(xg.grad))
    saved = E._r
    try:
        E._r = lambda v: v
        torch.manual_seed(0)
        n2 = R.build("segnet", 3, 12).train()
        o2 = E.segnet_forward(n2, x)
        torch.manual_seed(0)
        n3 = R.build("segnet", 3, 12).train()
        assert torch.allclose(o2, n3(x), rtol=1e-5, atol=1e-6)
    finally:
        E._r = saved


def test_winograd2d_rounding():
    """Derives the tolerance of the 2-D Winograd F(4x4,3x3) path (csrc/wino2d.hip): a numpy fp32 restatement of exactly its
    transforms (points 0, +-1, +-2, inf; V = B^T d B, U = G g G^T, y = A^T M A) against an fp64 direct convolution at the
    deep-layer channel counts, next to a plain fp32 direct convolution.  The GPU tests use 3 x the error measured here."""
    rng = np.random.default_rng(0)
    BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64)
    G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=np.float64)
    AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)

    def direct(x, w):
        H, W = x.shape[0] - 2, x.shape[1] - 2
        y = np.zeros((H, W, w.shape[0]), dtype=x.dtype)
        for dy in range(3):
            for dx in range(3):
                y += x[dy:dy + H, dx:dx + W, :] @ w[:, dy, dx, :].T
        return y

    def w2d(x, w):
        f = np.float32
        H, W = x.shape[0] - 2, x.shape[1] - 2
        U = np.einsum('ij,ojkc,lk->iloc', G.astype(f), w.astype(f), G.astype(f)).astype(f)
        y = np.zeros((H, W, w.shape[0]), dtype=f)
        bt, at = BT.astype(f), AT.astype(f)
        for ty in range(0, H, 4):
            for tx in range(0, W, 4):
                d = x[ty:ty + 6, tx:tx + 6, :].astype(f)
                V = np.einsum('ikc,lk->ilc', np.einsum('ij,jkc->ikc', bt, d), bt)
                M = np.einsum('ilc,iloc->ilo', V, U)
                y[ty:ty + 4, tx:tx + 4, :] = np.einsum('jlo,kl->jko', np.einsum('ji,ilo->jlo', at, M), at)
        return y

    for ci, co in ((1024, 512), (512, 512)):
        H, W = 8, 12
        x = np.maximum(rng.standard_normal((H + 2, W + 2, ci)), 0)
        x[0] = x[-1] = 0; x[:, 0] = x[:, -1] = 0
        b = 1 / np.sqrt(9 * ci)
        w = rng.uniform(-b, b, (co, 3, 3, ci))
        ref = direct(x, w)
        n = np.linalg.norm(ref)
        e_direct = np.linalg.norm(direct(x.astype(np.float32), w.astype(np.float32)) - ref) / n
        e_w2d = np.linalg.norm(w2d(x, w) - ref) / n
        assert e_direct < 5e-7 and 1e-6 < e_w2d < 3e-6, (ci, co, e_direct, e_w2d)      # GPU tolerance: 9e-6 = 3 x 3e-6


def test_logits_tolerance_is_frozen():
    """the full-size logits tolerance is derived from the reference graph's own drift (make_drift.py logits) — and may
    not grow with the kernels.  6.8e-4 is the ceiling: a fixture regenerated with a larger tolerance fails here."""
    import json, os
    d = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "drift.json")))
    for tag, v in d["logits_tolerance"].items():
        assert 0 < v["slice_abs"] <= 6.8e-4, (tag, v)


def test_w6_point_sets_and_exact_transforms():
    """can F(6x6,3x3) be made as accurate as F(4x4,3x3) by other interpolation points or more careful transforms?
    A numpy model of the 2-D path (Cook-Toom matrices for a point set, fp32 storage of U and V, fp32 GEMM accumulation) against an
    fp64 direct convolution at 256 input channels: (1) the standard points {0, +-1, +-2, +-1/2, inf} are within 10 % of the best
    symmetric set of simple rationals; (2) with all three transforms evaluated in fp64 and rounded ONCE the error only drops from
    3.0e-6 to 2.7e-6 — it is the fp32 rounding of the transform-domain operands and products themselves, so no transform
    arithmetic recovers the factor of two to F(4x4) (1.5e-6).  The accuracy trade of the 6x6 tile is inherent."""
    from fractions import Fraction as Fr

    def matrices(pts, m):
        n = m + 2
        a = [float(p) for p in pts]
        AT = np.zeros((m, n)); G = np.zeros((n, 3))
        for j in range(n - 1):
            Nj = np.prod([a[j] - a[k] for k in range(n - 1) if k != j])
            for i in range(m): AT[i, j] = a[j] ** i
            for k in range(3): G[j, k] = a[j] ** k / Nj
        AT[m - 1, n - 1] = 1.0; G[n - 1, 2] = 1.0
        BT = np.zeros((n, n))
        for l in range(n):
            rows = [AT[i, :] * G[:, k] for i in range(m) for k in range(3)]
            rhs = [1.0 if l == i + k else 0.0 for i in range(m) for k in range(3)]
            BT[:, l] = np.linalg.lstsq(np.array(rows), np.array(rhs), rcond=None)[0]
        return AT, G, BT

    def err(pts, m, exact=False, ci=256, co=32):
        AT, G, BT = matrices(pts, m)
        rng = np.random.default_rng(0)
        H = W = 12
        x = np.maximum(rng.standard_normal((H + 2, W + 2, ci)), 0)
        b = 1 / np.sqrt(9 * ci)
        w = rng.uniform(-b, b, (co, 3, 3, ci))
        ref = np.zeros((H, W, co))
        for dy in range(3):
            for dx in range(3):
                ref += x[dy:dy + H, dx:dx + W, :] @ w[:, dy, dx, :].T
        f = np.float32
        T = np.float64 if exact else f
        U = np.einsum('ij,ojkc,lk->iloc', G.astype(T), w.astype(f).astype(T), G.astype(T)).astype(f)
        y = np.zeros((H, W, co))
        for ty in range(0, H, m):
            for tx in range(0, W, m):
                d = x[ty:ty + m + 2, tx:tx + m + 2, :].astype(f).astype(T)
                V = np.einsum('ikc,lk->ilc', np.einsum('ij,jkc->ikc', BT.astype(T), d), BT.astype(T)).astype(f)
                M = np.einsum('ilc,iloc->ilo', V, U)                                      # fp32 accumulation
                y[ty:ty + m, tx:tx + m, :] = np.einsum('jlo,kl->jko', np.einsum('ji,ilo->jlo', AT.astype(T), M.astype(T)), AT.astype(T))
        return np.linalg.norm(y - ref) / np.linalg.norm(ref)

    std6 = [0, 1, -1, 2, -2, Fr(1, 2), Fr(-1, 2)]
    e6, e6x = err(std6, 6), err(std6, 6, exact=True)
    e4 = err([0, 1, -1, 2, -2], 4)
    others = [err([0, 1, -1, b, -b, c, -c], 6) for b, c in ((Fr(1, 2), Fr(7, 4)), (2, Fr(4, 7)), (Fr(7, 4), Fr(4, 7)), (Fr(1, 2), Fr(5, 3)),
                                                            (2, Fr(3, 5)), (Fr(3, 4), Fr(4, 3)), (Fr(1, 2), 3), (Fr(3, 2), Fr(2, 3)))]
    print("F(6x6) standard", e6, "exact transforms", e6x, "F(4x4)", e4, "best alternative", min(others))
    assert min(others) > 0.9 * e6, (e6, others)                  # no alternative point set is meaningfully better
    assert e6x > 0.8 * e6 and e6 > 1.7 * e4, (e6, e6x, e4)      # exact transforms do not close the gap to F(4x4)


def test_gradient_tolerances_follow_the_rule_and_cover_the_fixture():
    """drift.json grads_tolerance is SAFETY x max(fp32-vs-fp64, fp32-vs-noise) per tensor with the stated floors, and names exactly the
    tensors of the dense gradient fixture (tests/golden/unet_s0_8x360x480_grads.npz: six full conv weight gradients, strided samples of the
    others, all BatchNorm gradients; no conv biases)."""
    import json, os
    import numpy as np
    G = os.path.join(os.path.dirname(__file__), "golden")
    sys_path_mod = __import__("sys")
    sys_path_mod.path.insert(0, G)
    import make_drift as MD
    dj = json.load(open(os.path.join(G, "drift.json")))
    g, t = dj["grads"]["unet_s0_8x360x480"], dj["grads_tolerance"]["unet_s0_8x360x480"]
    assert MD.grads_tolerance(g) == t
    fx = np.load(os.path.join(G, "unet_s0_8x360x480_grads.npz"))
    meta = json.loads(str(fx["meta"]))
    keys = sorted(k[2:] for k in fx.files if k.startswith("g."))
    assert keys == sorted(t) and len(keys) == 69 and not any(k.endswith("conv.0.bias") for k in keys)
    from oracle import torch_ref as R
    net = R.build("unet", 3, 12)
    shapes = {k: p.shape for k, p in net.named_parameters()}
    for k in keys:
        n = int(np.prod(shapes[k]))
        want = n if (len(shapes[k]) != 4 or k in meta["dense"]) else (n + meta["stride"] - 1) // meta["stride"]
        assert fx["g." + k].size == want, k
    assert len(meta["dense"]) == 6 and sum(fx["g." + k].size for k in meta["dense"]) == 36864 * 2 + 147456 + 73728 * 2 + 6912
    # the reference's own element-wise gradient drift at this workload is percents for the deep layers (ReLU-mask / arg-max flips) and
    # 1e-3 for the last block: the tolerances inherit that shape
    assert 0.02 < t["down3.1.conv.0.weight"]["rel_l2"] < 0.1 and t["output.conv.0.weight"]["rel_l2"] < 0.01
