#!/usr/bin/env python3
"""VERDICT r4 #8 (study only — the product's fp32 path is exact-fp32 MFMA and stays so): error model of a SPLIT-OPERAND scheme that
would run fp32 convolutions on the bf16 matrix pipe (16x the fp32 MFMA rate on gfx950).

Each fp32 operand is written as a sum of bf16 terms, x = x1 + x2 (+ x3), x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)
(8 + 8 + 8 mantissa bits: three terms hold an fp32 value exactly up to its last bit or two).  A product x*w becomes the largest
cross-products — 3 of them for two terms (x1w1, x1w2, x2w1: ~2^-16 relative), 6 for three terms (+ x1w3, x3w1, x2w2: ~2^-24) — each
EXACT in fp32 (8 x 8 bit mantissas) and accumulated in the fp32 accumulators of v_mfma_f32_16x16x32_bf16.  Matrix-pipe cost in
units of the exact-fp32 MFMA time: terms / 16  ->  3/16 = 0.19 (5.3x ceiling), 6/16 = 0.375 (2.7x ceiling).

The model (numpy; fp32 storage and fp32 accumulation where the device would have them) measures the relative L2 error against an
fp64 direct convolution of one 256-input-channel layer (the shape of tests/test_drift_cpu.py::test_w6_point_sets_and_exact_transforms)
for: direct convolution, 2-D Winograd F(4x4,3x3) and F(6x6,3x3) (the split applied to the TRANSFORM-DOMAIN operands U and V, which
the device keeps as fp32 planes in HBM today), each with exact-fp32 products, the 3-product and the 6-product split.

    python tools/study/split_bf16_model.py            -> table on stdout (recorded in DESIGN.md, round 5)
"""
import sys
from fractions import Fraction as Fr

import numpy as np

f32 = np.float32


def bf16(a):
    """round-to-nearest-even to bfloat16, returned as float32"""
    u = np.ascontiguousarray(a, dtype=f32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(f32).reshape(np.shape(a))


def split(a, terms):
    out, rest = [], np.asarray(a, dtype=f32)
    for _ in range(terms):
        t = bf16(rest)
        out.append(t)
        rest = (rest - t).astype(f32)          # exact in fp32 (Sterbenz-like: t is rest rounded to 8 bits)
    return out


PRODUCTS = {2: [(0, 0), (0, 1), (1, 0)], 3: [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]}


def contract(V, U, spec, terms):
    """sum_c V[..c] * U[..c] with fp32 accumulation; terms 0 = plain fp32 products, 2 / 3 = split operands, largest cross products"""
    if terms == 0:
        return np.einsum(spec, V.astype(f32), U.astype(f32))
    Vs, Us = split(V, terms), split(U, terms)
    acc = None
    for i, j in sorted(PRODUCTS[terms], key=lambda p: -(p[0] + p[1])):     # smallest terms first into the fp32 accumulator
        p = np.einsum(spec, Vs[i], Us[j])                                    # bf16 x bf16 products are exact in fp32; fp32 accumulation
        acc = p if acc is None else (acc + p).astype(f32)
    return acc


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


def layer(ci=256, co=32, H=12, W=12, seed=0):
    rng = np.random.default_rng(seed)
    x = np.maximum(rng.standard_normal((H + 2, W + 2, ci)), 0).astype(f32)      # a post-ReLU activation
    b = 1 / np.sqrt(9 * ci)
    w = rng.uniform(-b, b, (co, 3, 3, ci)).astype(f32)
    ref = np.zeros((H, W, co))
    for dy in range(3):
        for dx in range(3):
            ref += x[dy:dy + H, dx:dx + W, :].astype(np.float64) @ w[:, dy, dx, :].astype(np.float64).T
    return x, w, ref


def err_direct(terms, **kw):
    x, w, ref = layer(**kw)
    H, W = ref.shape[:2]
    y = np.zeros(ref.shape, dtype=f32)
    for dy in range(3):
        for dx in range(3):
            y = (y + contract(x[dy:dy + H, dx:dx + W, :], w[:, dy, dx, :], 'hwc,oc->hwo', terms)).astype(f32)
    return np.linalg.norm(y - ref) / np.linalg.norm(ref)


def err_wino2d(m, terms, **kw):
    pts = [0, 1, -1, 2, -2] if m == 4 else [0, 1, -1, 2, -2, Fr(1, 2), Fr(-1, 2)]
    AT, G, BT = (a.astype(f32) for a in matrices(pts, m))
    x, w, ref = layer(**kw)
    H, W = ref.shape[:2]
    U = np.einsum('ij,ojkc,lk->iloc', G, w, G).astype(f32)                     # fp32 transform, fp32 storage (the device's U planes)
    y = np.zeros(ref.shape)
    for ty in range(0, H, m):
        for tx in range(0, W, m):
            d = x[ty:ty + m + 2, tx:tx + m + 2, :]
            V = np.einsum('ikc,lk->ilc', np.einsum('ij,jkc->ikc', BT, d), BT).astype(f32)
            M = contract(V, U, 'ilc,iloc->ilo', terms)
            y[ty:ty + m, tx:tx + m, :] = np.einsum('jlo,kl->jko', np.einsum('ji,ilo->jlo', AT, M.astype(f32)), AT)
    return np.linalg.norm(y - ref) / np.linalg.norm(ref)


def table(ci=256):
    rows = []
    for name, fn in (("direct 3x3", lambda t: err_direct(t, ci=ci)), ("2-D F(4x4,3x3)", lambda t: err_wino2d(4, t, ci=ci)),
                     ("2-D F(6x6,3x3)", lambda t: err_wino2d(6, t, ci=ci))):
        rows.append((name, fn(0), fn(2), fn(3)))
    return rows


if __name__ == "__main__":
    ci = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    print(f"relative L2 error vs fp64, {ci} input channels  |  exact-fp32 products  |  2 bf16 terms, 3 products  |  3 bf16 terms, 6 products")
    for name, e0, e2, e3 in table(ci):
        print(f"{name:16s} {e0:10.2e} {e2:10.2e} {e3:10.2e}")
    print("matrix-pipe time per product vs one fp32 MFMA:  1  |  3/16 = 0.19  |  6/16 = 0.375")
