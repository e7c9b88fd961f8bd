"""Static-plan executor for the conv/BN/ReLU/pool/upsample/concat hot path.

A network's `forward` (reference models/unet.py:94-156, models/segnet.py:82-119) is recorded ONCE per input shape as
a list of ops over NHWC fp32 activation buffers, then replayed: forward launches the libcvk kernels in order on the
current HIP stream, backward replays the list in reverse with explicit gradient buffers.  The whole network is ONE
autograd node (`_PlanFunction`), so PyTorch autograd sees `(input, *parameters) -> logits` and nothing in between:
no per-layer Python autograd overhead, no tensor version counters on the in-place concat writes, and the executor
knows exactly when each parameter gradient is final (the hook data-parallel bucketing uses, see ddp.py).

Buffers are torch tensors (caching allocator); concat is zero-copy: the upsample branch and the encoder stage write
into the two channel halves of one pre-sized buffer through strided views (cvk_view).
"""
import ctypes

import torch

from . import _lib
from ._lib import View, ViewH, check

_F32 = torch.float32


def pad4(c):
    return (c + 3) // 4 * 4


_BF16 = torch.bfloat16


class ActBuf:
    """Logical NHWC activation buffer [N,H,W,C] stored with pixel stride ld: fp32 with ld = pad4(C), or (bf16-storage
    plans) bf16 with ld = C rounded up to 8, at least 32 — the K-slice of the bf16 convolution kernels."""
    __slots__ = ("id", "N", "H", "W", "C", "ld", "name", "dtype")

    def __init__(self, id_, N, H, W, C, name="", dtype=_F32):
        self.id, self.N, self.H, self.W, self.C, self.name, self.dtype = id_, N, H, W, C, name, dtype
        self.ld = pad4(C) if dtype == _F32 else max(32, (C + 7) // 8 * 8)

    @property
    def esize(self):
        return 4 if self.dtype == _F32 else 2

    @property
    def M(self):
        return self.N * self.H * self.W

    def full_view(self):
        return BufView(self, 0, self.C, 0, 0, self.H, self.W)

    def __repr__(self):
        return f"ActBuf({self.name}#{self.id} {self.N}x{self.H}x{self.W}x{self.C}/{self.ld})"


class BufView:
    """Channel slice [c0, c0+C) and spatial window [y0,y0+H) x [x0,x0+W) of an ActBuf."""
    __slots__ = ("buf", "c0", "C", "y0", "x0", "H", "W")

    def __init__(self, buf, c0, C, y0, x0, H, W):
        assert 0 <= c0 and c0 + C <= buf.ld and 0 <= y0 and y0 + H <= buf.H and 0 <= x0 and x0 + W <= buf.W
        self.buf, self.c0, self.C, self.y0, self.x0, self.H, self.W = buf, c0, C, y0, x0, H, W

    def cview(self, tensor):
        b = self.buf
        ptr = tensor.data_ptr() + 4 * ((self.y0 * b.W + self.x0) * b.ld + self.c0)
        return View(ptr, b.H * b.W * b.ld, b.W * b.ld, b.ld)

    def hview(self, tensor):
        """cvk_viewh (strides in elements of the buffer's dtype) — the bf16-storage kernels' view type."""
        b = self.buf
        ptr = tensor.data_ptr() + b.esize * ((self.y0 * b.W + self.x0) * b.ld + self.c0)
        return ViewH(ptr, b.H * b.W * b.ld, b.W * b.ld, b.ld)

    @property
    def is_full(self):
        b = self.buf
        return self.c0 == 0 and self.C == b.C and self.y0 == 0 and self.x0 == 0 and self.H == b.H and self.W == b.W


class RunState:
    """Per-forward-call tensors: activations, saved-for-backward, gradient buffers."""

    def __init__(self, params, training, need_grad):
        self.params = params
        self.training = training
        self.need_grad = need_grad
        self.act = {}      # buf.id -> tensor [N,H,W,ld]
        self.saved = {}    # op index -> tuple
        self.grad = {}     # buf.id -> grad tensor
        self.gflat = None
        self.stream = None
        self.sync = None   # optional gradient synchroniser (ddp.GradSync)
        self.pooled_by_block = {}   # op index of a conv block -> True when its BN-apply pass also wrote the max pool behind it
        self.colsums = []           # queued column-sum finalisations (Runner.defer_colsum)
        self.wreduces = []          # queued weight-gradient slab reductions of the bf16 path (Runner.defer_wreduce)
        self.pass_token = 0         # derived-weight cache token of this pass (Runner.forward)
        self.bnred = {}             # op index of a conv block -> (partials, count): its BN-backward sums, left by the consumer's data-grad
        self.amax = {}              # buf.id -> amax block (device int32 words): the largest magnitude written into the buffer, left by the passes
                                    # that write it (opt-in fp16 split-operand layers scale by it: Runner.plan_amax)
        self.amax_spare = []        # zeroed amax blocks for the backward pass (|dy| maxima)


def _empty(n, dev, dtype=_F32):
    return torch.empty(n, device=dev, dtype=dtype)


import os

WINO_DEFAULT = os.environ.get("CVK_WINO", "1") != "0"   # 1-D Winograd F(2,3) for eligible layers (Cin % 64 == 0, > 32 columns)


# F(4,3) instead of F(2,3) for the forward / data-grad GEMMs of the layers where it pays (wino4_pays); Runner.wino4 may
# also be set to "always" (tests: every eligible layer, whatever its size)
WINO4_DEFAULT = {"0": False, "1": True, "always": "always"}[os.environ.get("CVK_WINO4", "1")]


def wino_ok(R, k_ch, n_cols):
    return R.wino and k_ch % 64 == 0


def wino4_pays(N, H, W, k_ch, n_cols):
    """F(4,3) or F(2,3) for this layer?  Both kernels do the same work per workgroup (3*k_ch/32 K steps of a 128-row
    tile); F(4,3) needs 6 workgroups per 4 columns, F(2,3) 8, and splits its K loop when the grid is small
    (csrc/wino4.hip plan_wino4), so it wins (measured, tools/bench_conv.py wino wino4) whenever there is more than one
    wave of work.  F(2,3) — the more accurate of the two — keeps the <= 32-column head and layers whose whole F(2,3)
    grid is at most one workgroup per CU anyway (the small golden geometries)."""
    if n_cols <= 32:
        return False
    tn = -(-n_cols // 128) if n_cols > 64 else 1
    return -(-(N * H * ((W + 1) // 2)) // 128) * tn * 4 > 256


# 2-D F(4x4,3x3) (csrc/wino2d.hip) for the channel-heavy layers: 2.25 multiplies per output and input channel instead of
# the 4.5 of the 1-D kernels, paid for with three HBM passes over transform-domain planes.  "0": off, "1": where it pays,
# "always": every eligible layer (tests).
WINO2D_DEFAULT = {"0": False, "1": True, "always": "always"}[os.environ.get("CVK_WINO2D", "1")]


# Output tile of the 2-D path: 6 = F(6x6,3x3) (64 GEMMs, 1.78 multiplies per output, planes 1.78x the activation), 4 = F(4x4,3x3)
# (36 GEMMs, 2.25 / 2.25x).  Both are the same kernels of csrc/wino2d.hip; F(6x6) rounds about twice as coarsely (5e-6 relative L2
# at 256 input channels, tests/test_gpu_w6.py).  "auto" (default): 6, except for networks with MaxUnpool2d (SegNet): their five
# pool/unpool pairs are discontinuous in the arg-max, rounding differences flip single arg-maxes and the logits behind them, and
# the full-size SegNet parity test (distance to the reference <= 2x the reference's own distance to its 1e-6-perturbed twin) holds
# with F(4x4) (18 % of the sampled logits move by > 1e-3, the twin: 11 %) but not with F(6x6) (26 %).
W2TILE_DEFAULT = {"auto": None, "4": 4, "6": 6}[os.environ.get("CVK_W2D_TILE", "auto")]


def w2fn(lib, tile, name):
    """Entry point `name` of the 2-D Winograd family for output tile `tile`: cvk_w2d_<name> (4) or cvk_w6_<name> (6)."""
    return getattr(lib, ("cvk_w6_" if tile == 6 else "cvk_w2d_") + name)


def w2ws(lib, tile, N, H, W, k_ch, cout):
    return (lib.cvk_conv3x3_w6_workspace_bytes if tile == 6 else lib.cvk_conv3x3_w2d_workspace_bytes)(N, H, W, k_ch, cout)


def layer_tile(R, N, H, W, dgrad=False):
    """Output tile of the 2-D path for a layer geometry (dgrad: of a data-grad launch — networks with MaxUnpool2d keep 4x4 tiles in the
    forward pass, where coarser rounding flips pool arg-maxes, but their data-grads run after the indices are fixed: 6x6).  The batched GEMM works on 128-row tiles of the tile index: at the 22x30
    bottleneck (batch 8) 6x6 tiles give 160 rows = two row tiles of which 37 % are padding (138 us, as long as F(4x4)'s three full
    row tiles) and 1.78x the filter-transform bytes (50 vs 28 us at 1024 x 1024 channels) — such layers keep 4x4 tiles."""
    if (R.w2tile_dgrad if dgrad else R.w2tile) != 6:
        return 4
    t6 = N * ((H + 5) // 6) * ((W + 5) // 6)
    t4 = N * ((H + 3) // 4) * ((W + 3) // 4)
    u6 = t6 / (128.0 * ((t6 + 127) // 128))
    u4 = t4 / (128.0 * ((t4 + 127) // 128))
    return 4 if (u6 < 0.7 and u4 > u6 + 0.2) else 6


def wino2d_ok(k_ch, cout, ldy):
    return k_ch % 32 == 0 and cout % 4 == 0 and cout >= 64 and ldy % 4 == 0


def wino2d_pays(N, H, W, k_ch, cout, tile=4):
    """Measured at the UNet batch-8 shapes (tools/bench_conv.py wino4 w2d [--rev]; tile 6: tools/bench_w6.py).  F(4x4,3x3): the 36
    batched GEMMs + transforms beat F(4,3) + its output pass by 13-31 % once Cin*Cout >= 256*256 (256->256 @ 90x120 ... 1024->512 @
    45x60), by 10-13 % for 128<->256 channels at 180x240, and lose below that (the transform passes cost more than the saved
    multiplies).  F(6x6,3x3) is 15-25 % cheaper than F(4x4) on every layer with >= 256 tiles (21 % fewer multiplies and plane
    bytes) and would also take 128->128 @180x240 (435 vs 490 us forward) and 128<->256 @90x120 (170 vs 252 / 302 us) from the fused
    1-D kernel — measured: no gain on the whole step (231.4 vs 232.9 img/s on two boxes) while the logits deviation from the
    reference grows again (headline workload, sampled logits: max 6.9e-4 and 1.2 % beyond 3e-4, against 4.6e-4 / 0.3 % with the layer
    set below; F(4x4): 2.5e-4 / none; the reference's own fp32-vs-1e-6-noise drift: 1.7e-4) — so both tile sizes take the SAME layers."""
    T = N * ((H + 3) // 4) * ((W + 3) // 4)
    return T >= 256 and (k_ch * cout >= 65536 or (k_ch * cout >= 32768 and T >= 16384))


def wgrad2d_pays(N, H, W, k_ch, cout):
    """Weight-grad through the 2-D transform: wins from 256 x 256 channels up (0.60-0.90 of the transposed F(4,3) time at the
    UNet batch-8 shapes), loses below (the dy / x transform passes dominate)."""
    T = N * ((H + 3) // 4) * ((W + 3) // 4)
    return T >= 256 and k_ch * cout >= 65536


# fused F(4,3) (csrc/wino4f.hip) instead of the per-index F(4,3) kernels + output pass: "0" off, "1" on
WINO4F_DEFAULT = os.environ.get("CVK_WINO4F", "1") != "0"


def wino4f_ok(k_ch, cout):
    return k_ch % 32 == 0 and cout % 4 == 0 and cout >= 32


# transposed F(4,3) weight-grad through transform-domain planes (csrc/wgradp.hip): "0" off, "1" where it pays, "always" (tests)
WGRADP_DEFAULT = {"0": False, "1": True, "always": "always"}[os.environ.get("CVK_WGRADP", "1")]


def wgradp_ok(cin_ld, cout, ldy):
    return cin_ld % 64 == 0 and cout % 64 == 0 and ldy == cout


def wgradp_pays(N, H, W, cin_ld, cout):
    return cin_ld == 64 and N * H * ((W + 3) // 4) >= 4096


# Round 6: the fused F(4,3) forward launch also writes the weight-grad's V planes (cvk_conv3x3_wino4f_vplanes, slice-major) — the plane GEMM then
# takes every F(4,3) weight-grad, not only the 64-input-channel ones, and the x -> V pass is gone.  "0" off, "1" on.
VPLANES_DEFAULT = os.environ.get("CVK_VPLANES", "1") != "0"


def vplanes_pays(N, H, W, cin_ld, cout):
    """tools/bench_vplanes.py at the headline shapes (64/128 -> 64/128/256 channels, 360x480 ... 90x120, batch 8): the forward launch costs
    +9 ... +47 us, the weight-grad gains 34 ... 370 us on every layer; bounded by the kernel's 4 GiB plane addressing."""
    rows = N * (H + 2) * (((W + 3) // 4 + 7) // 8 * 8)
    return cin_ld % 64 == 0 and cout <= 512 and N * H * ((W + 3) // 4) >= 4096 and 24 * cin_ld * (rows + 64) < 2 ** 32 - 4096


def amax_blocks(lib, n, dev):
    """n zeroed "amax blocks" (csrc/cvk_common.h: the largest magnitude of a tensor in device memory, a few slots one cache line apart): a list of
    int32 views, one fill for all of them."""
    nw = lib.cvk_amax_block_words()
    t = torch.zeros(n * nw, device=dev, dtype=torch.int32)
    return [t[i * nw:(i + 1) * nw] for i in range(n)]


def split_fmt(R):
    """runner.w2d_split as a split-plane format (csrc/split_fmt.h): 0 = off (exact-fp32 MFMA, the default), 3 = three bf16 terms (True means
    this one), 2 = two scaled fp16 terms."""
    v = getattr(R, "w2d_split", 0)
    return 3 if v is True else (int(v) if v in (2, 3) else 0)


def wino_conv(R, lib, s, x, w, bias, y, sp, N, H, W, k_ch, cout, ldy, flops, what="", dgrad_of=None, keep_v=None, wsrc=None, ck=None,
              bnred=None, v_pre=None, split=False, x_amax=None, h2=False, want_planes=False):
    """y[N,H,W,ldy] = conv3x3(x[N,H,W,k_ch], w[cout][3][3][k_ch]) (+bias, +BN statistics partials at sp) through the
    Winograd kernels: weight transform -> (input transform ->) GEMMs M_xi -> output transform.  2-D F(4x4,3x3) for the
    channel-heavy layers (R.wino2d, wino2d_pays), else 1-D F(4,3) when R.wino4, else F(2,3).
    Data-grad: `w` is a callable returning the rotated/transposed pack (built only if a kernel needs it) and
    dgrad_of = (forward weights [Cout_f][3][3][Cin_f], Cout_f, Cin_f) lets F(4,3) transform straight from them.
    Returns None, or (P, counts pointer) when the statistics partials at sp carry explicit pixel counts (2-D path:
    P = cvk_w2d_stat_partials partials of [sum | M2] followed by the counts -> cvk_bn_finalize_counts).  keep_v: a list that
    receives the transformed input V when the 2-D path runs (the layer's weight-grad reuses it).  wsrc / ck: the parameter the
    filter derives from and the layer's cache key — the transformed filter is then kept across calls (Runner.derived).
    bnred (data-grad only): (yP, scale, shift, mean, rstd pointers, out list) of the block that produced this conv's input; when
    the fused F(4,3) kernel runs and ldy == cout it also leaves that block's BatchNorm-backward sums and appends
    (partials tensor, partial count) to the list.  v_pre (tile, tensor): the transformed input V of the 2-D path, already computed
    with that tile (cvk_w6_dy_transform_both: the backward pass transforms dy once for the data-grad and the weight-grad)."""
    M = N * H * W

    def cached(kind, build, job=None):
        return R.derived((ck, kind), wsrc, build, job) if (ck is not None and wsrc is not None) else build()

    def straight(t):                        # is tensor t the parameter itself (no packed / padded copy in between)?
        return wsrc is not None and t is not None and not callable(t) and t.data_ptr() == wsrc.data_ptr()
    if wino2d_ok(k_ch, cout, ldy) and (R.wino2d == "always" or (R.wino2d and wino2d_pays(N, H, W, k_ch, cout, R.w2tile))):
        tile = layer_tile(R, N, H, W, dgrad=dgrad_of is not None)
        NX = 64 if tile == 6 else 36
        if split:
            # OPT-IN path (runner.w2d_split = 3 | 2): the GEMM stage on the 16-bit matrix pipe with split fp32 operands
            # (csrc/split_fmt.h: three bf16 terms / six cross-products, or two fp16 terms / three cross-products scaled by an exact power of
            # two from the source tensors' largest magnitudes) — the transforms write split planes, cvk_w2d_gemm_split multiplies them, the
            # plain output pass finishes
            fmt = int(split)
            pdt = _BF16 if fmt == 3 else torch.float16
            tag = "split3" if fmt == 3 else "split2h"
            T = w2fn(lib, tile, "tiles")(N, H, W)
            Tp = lib.cvk_split3_rows_pad(T, 256)
            Cp = lib.cvk_split3_rows_pad(cout, 128)
            wraw = dgrad_of[0] if dgrad_of is not None else (w() if callable(w) else w)
            am_w = None
            if fmt == 2:
                def build_amax_w():
                    a = amax_blocks(lib, 1, x.device)[0]
                    _timed(R, "k_absmax", 4.0 * wraw.numel(), lambda: check(
                        lib.cvk_absmax_f32(wraw.data_ptr(), wraw.numel() // 4, 4, 4, a.data_ptr(), s), "cvk_absmax_f32(w)"), "byte")
                    return a
                am_w = R.derived(((ck[0] if ck is not None else None, "a"), "amaxw"), wsrc, build_amax_w) if (ck is not None and wsrc is not None) \
                    else build_amax_w()
            def build_u3():
                u3 = torch.empty(NX * (k_ch // 32) * fmt * Cp * 32, device=x.device, dtype=pdt)
                amp = am_w.data_ptr() if am_w is not None else None
                if dgrad_of is not None:
                    _timed(R, "k_w2d_weight_dgrad+" + tag, 4.0 * 9 * cout * k_ch + 2.0 * fmt * NX * cout * k_ch, lambda: check(
                        lib.cvk_w2d_weight_transform_split(fmt, tile, wraw.data_ptr(), u3.data_ptr(), amp, dgrad_of[1], dgrad_of[2], 1, s),
                        "cvk_w2d_weight_transform_split(dgrad)"), "byte")
                else:
                    _timed(R, "k_w2d_weight+" + tag, 4.0 * 9 * cout * k_ch + 2.0 * fmt * NX * cout * k_ch, lambda: check(
                        lib.cvk_w2d_weight_transform_split(fmt, tile, wraw.data_ptr(), u3.data_ptr(), amp, cout, k_ch, 0, s),
                        "cvk_w2d_weight_transform_split"), "byte")
                return u3
            U3 = cached("w2ds%d_%d" % (fmt, tile), build_u3)
            v3fl = NX * (k_ch // 32) * fmt * Tp * 32
            if v_pre is not None:
                V3, am_x = v_pre[1], getattr(v_pre[1], "cvk_amax", None)
            else:
                V3 = torch.empty(v3fl, device=x.device, dtype=pdt)
                am_x = x_amax           # left by the passes that wrote x (Runner.plan_amax), else measured here
                if fmt == 2 and am_x is None:
                    am_x = amax_blocks(lib, 1, x.device)[0]
                    _timed(R, "k_absmax", 4.0 * M * k_ch, lambda: check(
                        lib.cvk_absmax_f32(x.data_ptr(), M, k_ch, k_ch, am_x.data_ptr(), s), "cvk_absmax_f32(x)"), "byte")
                V3.cvk_amax = am_x          # the planes travel with the word they were scaled by (weight-grad GEMM, data-grad GEMM)
                if keep_v is not None:
                    keep_v.append(V3)
                _timed(R, "k_w2d_input<%s>" % tag, (4.0 * M + 2.0 * fmt * NX * T) * k_ch, lambda: check(
                    lib.cvk_w2d_input_transform_split(fmt, tile, x.data_ptr(), V3.data_ptr(), am_x.data_ptr() if am_x is not None else None,
                                                      N, H, W, k_ch, s), "cvk_w2d_input_transform_split" + what), "byte")
            ws = R.workspace(4 * NX * T * cout + 1024, x.device)
            _timed(R, "k_gemm_" + tag, flops, lambda: check(
                lib.cvk_w2d_gemm_split(fmt, tile, V3.data_ptr(), U3.data_ptr(), ws.data_ptr(), am_x.data_ptr() if am_x is not None else None,
                                       am_w.data_ptr() if am_w is not None else None, NX, T, Tp, k_ch, cout, Cp, s), "cvk_w2d_gemm_split" + what),
                executed=2.0 * (6 if fmt == 3 else 3) * NX * Tp * k_ch * Cp)     # six bf16 / three fp16 MFMA products per fp32 product
            P2 = w2fn(lib, tile, "stat_partials")(N, H, W)
            cnt = sp + 4 * 2 * P2 * cout if sp is not None else None
            _timed(R, "k_w2d_output", 4.0 * (NX * T + M) * cout, lambda: check(
                lib.cvk_w2d_output_plain(tile, ws.data_ptr(), bias, y.data_ptr(), sp, cnt, N, H, W, cout, ldy, s), "cvk_w2d_output_plain" + what), "byte")
            return (P2, cnt) if sp is not None else None
        def build_u2():
            u = _empty(NX * cout * k_ch, x.device)
            if dgrad_of is not None and dgrad_of[1] == k_ch and dgrad_of[2] == cout:       # no channel padding: straight from the forward weights
                _timed(R, "k_w2d_weight_dgrad", 4.0 * (9 + NX) * cout * k_ch, lambda: check(
                    w2fn(lib, tile, "weight_transform_dgrad")(dgrad_of[0].data_ptr(), u.data_ptr(), dgrad_of[1], dgrad_of[2], s),
                    "cvk_w2d_weight_transform_dgrad"), "byte")
                return u
            wt = w() if callable(w) else w
            _timed(R, "k_w2d_weight", 4.0 * (9 + NX) * cout * k_ch, lambda: check(
                w2fn(lib, tile, "weight_transform")(wt.data_ptr(), u.data_ptr(), cout, k_ch, s), "cvk_w2d_weight_transform"), "byte")
            return u
        job = None
        if dgrad_of is not None and dgrad_of[1] == k_ch and dgrad_of[2] == cout and straight(dgrad_of[0]):
            job = ("w2d", NX * cout * k_ch, dgrad_of[1], dgrad_of[2], tile, 1)
        elif dgrad_of is None and straight(w):
            job = ("w2d", NX * cout * k_ch, cout, k_ch, tile, 0)
        U = cached("w2d%d" % tile, build_u2, job)
        T = w2fn(lib, tile, "tiles")(N, H, W)
        vfl = NX * lib.cvk_w2d_tpad(T) * k_ch + 128          # V planes + 512 bytes of slack
        if v_pre is not None and v_pre[0] == tile:
            ws = R.workspace(w2ws(lib, tile, N, H, W, k_ch, cout) - 4 * vfl, x.device)
            V, Mo = v_pre[1].data_ptr(), ws.data_ptr()
        elif keep_v is not None:    # the weight-grad of this layer reuses V: its own tensor instead of the shared workspace
            Vt = _empty(vfl, x.device)
            keep_v.append(Vt)
            ws = R.workspace(w2ws(lib, tile, N, H, W, k_ch, cout) - 4 * vfl, x.device)
            V, Mo = Vt.data_ptr(), ws.data_ptr()
        else:
            ws = R.workspace(w2ws(lib, tile, N, H, W, k_ch, cout), x.device)
            V, Mo = ws.data_ptr(), ws.data_ptr() + 4 * vfl
        P2 = w2fn(lib, tile, "stat_partials")(N, H, W)
        cnt = sp + 4 * 2 * P2 * cout if sp is not None else None
        if not (v_pre is not None and v_pre[0] == tile):
            _timed(R, "k_w2d_input", 4.0 * (M + NX * T) * k_ch, lambda: check(
                w2fn(lib, tile, "input_transform")(x.data_ptr(), V, N, H, W, k_ch, s), "cvk_w2d_input_transform" + what), "byte")
        _timed(R, "k_w2d_gemm<128, 32, 2, 2>", flops, lambda: check(w2fn(lib, tile, "gemm")(V, U.data_ptr(), Mo, T, k_ch, cout, s), "cvk_w2d_gemm" + what),
               executed=2.0 * NX * T * k_ch * cout)   # NX GEMMs of T x k_ch x cout really run on the matrix pipe
        _timed(R, "k_w2d_output", 4.0 * (NX * T + M) * cout, lambda: check(
            w2fn(lib, tile, "output")(Mo, bias, y.data_ptr(), sp, cnt, N, H, W, k_ch, cout, ldy, s), "cvk_w2d_output" + what), "byte")
        return (P2, cnt) if sp is not None else None
    use4 = R.wino4 == "always" or (R.wino4 and wino4_pays(N, H, W, k_ch, ldy))
    if use4 and R.wino4f and wino4f_ok(k_ch, cout) and (dgrad_of is None or (dgrad_of[1] == k_ch and dgrad_of[2] == cout)):
        # fused F(4,3) (csrc/wino4f.hip): all six transform indices in one workgroup, output transform + bias + statistics in
        # registers — no product planes, no output pass
        if h2:
            # OPT-IN fp16 split-operand form (runner.w2d_split = 2): the same kernel with two scaled fp16 terms per operand, 18 fp16 MFMAs per K
            # step instead of 48 fp32 ones; needs the largest magnitudes of x (left by the pass that wrote it, else measured here) and of w
            wraw = dgrad_of[0] if dgrad_of is not None else (w() if callable(w) else w)

            def build_amax_w():
                a = amax_blocks(lib, 1, x.device)[0]
                _timed(R, "k_absmax", 4.0 * wraw.numel(), lambda: check(
                    lib.cvk_absmax_f32(wraw.data_ptr(), wraw.numel() // 4, 4, 4, a.data_ptr(), s), "cvk_absmax_f32(w)"), "byte")
                return a
            am_w = R.derived(((ck[0], "a"), "amaxw"), wsrc, build_amax_w) if (ck is not None and wsrc is not None) else build_amax_w()

            def build_uh():
                uh = _empty(lib.cvk_wino4f_weight_floats(cout, k_ch), x.device)
                _timed(R, "k_wino4h_weight", 4.0 * (9 + 18) * cout * k_ch, lambda: check(
                    lib.cvk_wino4h_weight_transform(wraw.data_ptr(), uh.data_ptr(), am_w.data_ptr(), cout, k_ch, 1 if dgrad_of is not None else 0, s),
                    "cvk_wino4h_weight_transform"), "byte")
                return uh
            Uh = cached("w4h", build_uh)
            am_x = x_amax
            if am_x is None:
                am_x = amax_blocks(lib, 1, x.device)[0]
                _timed(R, "k_absmax", 4.0 * M * k_ch, lambda: check(lib.cvk_absmax_f32(x.data_ptr(), M, k_ch, k_ch, am_x.data_ptr(), s), "cvk_absmax_f32(x)"), "byte")
            Pf = lib.cvk_wino4f_stat_partials(N, H, W)
            cnt = sp + 4 * 2 * Pf * cout if sp is not None else None
            if bnred is not None and sp is None and bias is None and ldy == cout and R.bnred_fuse:
                bpart = _empty(2 * Pf * cout, x.device)
                _timed(R, "k_conv3x3_wino4h<bnred>", flops, lambda: check(
                    lib.cvk_conv3x3_wino4h_bnred(x.data_ptr(), Uh.data_ptr(), y.data_ptr(), am_x.data_ptr(), am_w.data_ptr(), N, H, W, k_ch, cout, ldy,
                                                 *bnred[:5], bpart.data_ptr(), R.launch_wgs(), s), "cvk_conv3x3_wino4h_bnred"), executed=1.5 * flops)
                bnred[5].append((bpart, Pf))
                return None
            _timed(R, "k_conv3x3_wino4h", flops, lambda: check(
                lib.cvk_conv3x3_wino4h(x.data_ptr(), Uh.data_ptr(), bias, y.data_ptr(), sp, cnt, am_x.data_ptr(), am_w.data_ptr(), N, H, W, k_ch, cout,
                                       ldy, R.launch_wgs(), s), "cvk_conv3x3_wino4h" + what), executed=1.5 * flops)      # 3 fp16 products per fp32 product
            return (Pf, cnt) if sp is not None else None
        def build_uf():
            uf = _empty(lib.cvk_wino4f_weight_floats(cout, k_ch), x.device)
            if dgrad_of is not None:
                _timed(R, "k_wino4f_weight", 4.0 * (9 + 18) * cout * k_ch, lambda: check(
                    lib.cvk_wino4f_weight_transform(dgrad_of[0].data_ptr(), uf.data_ptr(), cout, k_ch, 1, s), "cvk_wino4f_weight_transform(dgrad)"), "byte")
            else:
                wt = w() if callable(w) else w
                _timed(R, "k_wino4f_weight", 4.0 * (9 + 18) * cout * k_ch, lambda: check(
                    lib.cvk_wino4f_weight_transform(wt.data_ptr(), uf.data_ptr(), cout, k_ch, 0, s), "cvk_wino4f_weight_transform"), "byte")
            return uf
        job = None
        if dgrad_of is not None and straight(dgrad_of[0]):
            job = ("w4f", lib.cvk_wino4f_weight_floats(cout, k_ch), cout, k_ch, 0, 1)
        elif dgrad_of is None and straight(w):
            job = ("w4f", lib.cvk_wino4f_weight_floats(cout, k_ch), cout, k_ch, 0, 0)
        Uf = cached("w4f", build_uf, job)
        Pf = lib.cvk_wino4f_stat_partials(N, H, W)
        cnt = sp + 4 * 2 * Pf * cout if sp is not None else None
        if bnred is not None and sp is None and bias is None and ldy == cout and R.bnred_fuse:
            bpart = _empty(2 * Pf * cout, x.device)
            _timed(R, "k_conv3x3_wino4f<bnred>", flops, lambda: check(
                lib.cvk_conv3x3_wino4f_bnred(x.data_ptr(), Uf.data_ptr(), y.data_ptr(), N, H, W, k_ch, cout, ldy, *bnred[:5],
                                             bpart.data_ptr(), R.launch_wgs(), s), "cvk_conv3x3_wino4f_bnred"), executed=0.5 * flops)
            bnred[5].append((bpart, Pf))
            return None
        if want_planes and keep_v is not None and dgrad_of is None:
            # forward launch of a layer whose weight-grad runs the plane GEMM (csrc/wgradp.hip): the kernel's staging path leaves V = B^T d
            # behind as six slice-major planes, kept for the backward pass like the 2-D path's V
            rows6 = lib.cvk_wgradp_plane_rows(N, H, W)
            V6 = _empty(6 * rows6 * k_ch, x.device)
            V6.cvk_sm = True
            check(lib.cvk_wgradp_zero_pads_sm(V6.data_ptr(), N, H, W, k_ch, s), "cvk_wgradp_zero_pads_sm")
            _timed(R, "k_conv3x3_wino4f<vplanes>", flops, lambda: check(
                lib.cvk_conv3x3_wino4f_vplanes(x.data_ptr(), Uf.data_ptr(), bias, y.data_ptr(), sp, cnt, V6.data_ptr(), N, H, W, k_ch, cout, ldy,
                                               R.launch_wgs(), s), "cvk_conv3x3_wino4f_vplanes" + what), executed=0.5 * flops)
            keep_v.append(V6)
            return (Pf, cnt) if sp is not None else None
        _timed(R, "k_conv3x3_wino4f", flops, lambda: check(
            lib.cvk_conv3x3_wino4f(x.data_ptr(), Uf.data_ptr(), bias, y.data_ptr(), sp, cnt, N, H, W, k_ch, cout, ldy, R.launch_wgs(), s),
            "cvk_conv3x3_wino4f" + what), executed=0.5 * flops)
        return (Pf, cnt) if sp is not None else None
    if use4:
        def build_u4():
            u = _empty(6 * cout * 3 * k_ch, x.device)
            if dgrad_of is not None and dgrad_of[1] == k_ch and dgrad_of[2] == cout:       # no channel padding on either side
                _timed(R, "k_wino4_weight_dgrad", 4.0 * (9 + 18) * cout * k_ch, lambda: check(
                    lib.cvk_wino4_weight_transform_dgrad(dgrad_of[0].data_ptr(), u.data_ptr(), dgrad_of[1], dgrad_of[2], s),
                    "cvk_wino4_weight_transform_dgrad"), "byte")
            else:
                wt = w() if callable(w) else w
                _timed(R, "k_wino4_weight", 4.0 * (9 + 18) * cout * k_ch, lambda: check(
                    lib.cvk_wino4_weight_transform(wt.data_ptr(), u.data_ptr(), cout, k_ch, s), "cvk_wino4_weight_transform"), "byte")
            return u
        U = cached("w4", build_u4)
        ws = R.workspace(lib.cvk_conv3x3_wino4_workspace_bytes(N, H, W, k_ch, ldy), x.device)
        ksplit = lib.cvk_conv3x3_wino4_ksplit(N, H, W, k_ch, ldy)
        _timed(R, conv_kernel_name("wino4", ldy), flops, lambda: check(
            lib.cvk_conv3x3_wino4_gemm(x.data_ptr(), U.data_ptr(), ws.data_ptr(), N, H, W, k_ch, cout, ldy, s), "cvk_conv3x3_wino4_gemm" + what))
        _timed(R, "k_wino4_output", (4.0 + 6.0 * ksplit) * M * ldy, lambda: check(
            lib.cvk_wino4_output(ws.data_ptr(), bias, y.data_ptr(), sp, N, H, W, cout, ldy, ksplit, s), "cvk_wino4_output"), "byte")
    else:
        def build_u():
            wt = w() if callable(w) else w
            u = _empty(4 * cout * 3 * k_ch, x.device)
            _timed(R, "k_wino_weight", 4.0 * (9 + 12) * cout * k_ch, lambda: check(
                lib.cvk_wino_weight_transform(wt.data_ptr(), u.data_ptr(), cout, k_ch, s), "cvk_wino_weight_transform"), "byte")
            return u
        U = cached("w", build_u)
        ws = R.workspace(lib.cvk_conv3x3_wino_workspace_bytes(N, H, W, ldy), x.device)
        _timed(R, conv_kernel_name("wino", ldy), flops, lambda: check(
            lib.cvk_conv3x3_wino_gemm(x.data_ptr(), U.data_ptr(), ws.data_ptr(), N, H, W, k_ch, cout, ldy, s), "cvk_conv3x3_wino_gemm" + what))
        _timed(R, "k_wino_output", 12.0 * M * ldy, lambda: check(
            lib.cvk_wino_output(ws.data_ptr(), bias, y.data_ptr(), sp, N, H, W, cout, ldy, s), "cvk_wino_output"), "byte")


def conv_kernel_name(kind, n_cols, k_ch=32):
    """Mirror of the tile dispatch in csrc/conv3x3.hip / wino.hip: the kernel-trace name."""
    if kind == "wino4":
        return "k_conv3x3_wino4<128, 128, 2, 2>" if n_cols > 64 else ("k_conv3x3_wino4<128, 64, 2, 2>" if n_cols > 32 else "k_conv3x3_wino4<128, 32, 4, 1>")
    if kind == "wino":
        return "k_conv3x3_wino<128, 128, 2, 2>" if n_cols > 64 else ("k_conv3x3_wino<128, 64, 2, 2>" if n_cols > 32 else "k_conv3x3_wino<128, 32, 4, 1>")
    if kind == "wgrad":
        if n_cols <= 16 and k_ch == 64:
            return "k_wgrad_smallco"
        t = "128, 128, 2, 2" if n_cols > 64 else ("64, 128, 2, 2" if n_cols > 32 else "32, 256, 1, 4")
        if 32 < n_cols <= 64 and k_ch * 9 <= 64:
            t = "64, 64, 2, 2"
        return f"k_conv3x3_wgrad<{t}>"
    t = "128, 128, 2, 2" if n_cols > 64 else ("128, 64, 2, 2" if n_cols > 32 else "256, 32, 4, 1")
    return f"k_conv3x3_igemm<{t}, {'true' if kind == 'fwd' else 'false'}, {'true' if k_ch % 32 == 0 else 'false'}>"


def bf16_kernel_name(lib, N, H, W, cin_ld, cout, stats):
    """The kernel cvk_conv3x3_bf16s(_wg) dispatches for this geometry, as a kernel trace names it (the library answers:
    cvk_conv3x3_bf16s_kernel is a query of the same dispatch code; csrc/conv_bf16s.hip, csrc/conv_bf16p.hip)."""
    k = lib.cvk_conv3x3_bf16s_kernel(N, H, W, cin_ld, cout, 1 if stats else 0)
    st = "true" if stats else "false"
    if k == 1:
        return f"k_conv_bf16q<{st}>"
    if k in (2, 3):
        return f"k_conv_bf16h<{st}, 0, {'true' if k == 3 else 'false'}>"
    if k in (4, 5):
        return f"k_conv_bf16s_strip<{1 if cin_ld == 32 else 2}, {st}>" + (" x2" if k == 5 else "")
    return f"k_conv_bf16s<{128 if cout > 64 else 64}, {st}>"


# ---- derived-weight cache ---------------------------------------------------------------------------------------------
# Winograd-domain filters, data-grad packs and bf16 copies are functions of the conv weights only; a step used to rebuild
# all of them (~55 launches, 0.5 ms, also in eval mode).  They are now kept per layer and rebuilt when the weights may have
# changed.  A raw-pointer optimizer or a `p.data` write does not touch the parameter's version counter, so validity is
# explicit: an entry is valid while (global optimizer epoch, runner epoch, storage pointer, tensor version) are unchanged.
#   * any torch.optim optimizer step      -> global epoch (torch.optim.optimizer.register_optimizer_step_post_hook)
#   * FlatAdamW.step, load_state_dict, ddp.DataParallel's broadcast, mark_weights_dirty(net) -> runner epoch
#   * .to() / .cuda() / in-place autograd-visible ops -> storage pointer / version counter
# FAIL-SAFE in training: a TRAINING pass (forward with a backward to follow) never trusts entries of earlier passes — its
# signature carries a per-pass token, so the tensors are built once per pass (one batched launch in bf16 mode) and shared by that
# pass's forward and backward only.  Writes nobody can see (`p.data.copy_()` of an EMA swap, manual re-initialisation, a foreign
# kernel, a freed-and-reallocated parameter at the same address) are therefore always picked up in training, as stock torch
# picks them up.  Only eval / no-grad passes reuse entries across calls; THERE an invisible in-place write needs
# `mark_weights_dirty(net)`.
# Never used while a stream capture is running: a replayed graph must recompute the derived tensors from the live weights.
WEIGHT_EPOCH = [0]
WCACHE_DEFAULT = os.environ.get("CVK_WEIGHT_CACHE", "1") != "0"


def _bump_epoch(*_a, **_k):
    WEIGHT_EPOCH[0] += 1


try:
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_opt_hook
    _reg_opt_hook(_bump_epoch)
except ImportError:                      # very old torch: no global hook, the cache stays off
    WCACHE_DEFAULT = False


def mark_weights_dirty(module):
    """Tell the executor that the parameters of `module` were changed behind its back (p.data writes, custom kernels):
    every derived weight tensor (Winograd-domain filters, data-grad packs, bf16 copies) is rebuilt on next use."""
    from .modules import runner_of
    runner_of(module).wepoch += 1


PROF = None     # bench.py sets this to a list: every _timed call then appends (kernel name, work, start event, end event, unit, executed)


def _timed(R, name, work, fn, unit="flop", executed=None, nbytes=None):
    """Run fn(); when a profile list is attached (bench.py sets engine.PROF), bracket it with HIP events on the launch
    stream.  `work` is the ALGORITHMIC work of the call: FLOPs for the conv kernels, HBM bytes (each input read once,
    each output written once, at the storage dtype) for the memory-bound kernels.  `executed`: FLOPs the kernel really
    issues to the matrix pipe when that is not a fixed share of `work` (bench.py knows the fixed shares by kernel name).
    `nbytes`: for conv kernels that sit near the HBM side of the ridge (the 3-channel stem, the 12-class head: SURVEY §8d) their
    algorithmic bytes as well — bench.py then reports BOTH roofline fractions for them."""
    if PROF is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    PROF.append((name, work, e0, e1, unit, executed, nbytes))
    return r


# ===================================================================================================== ops
class Op:
    idx = -1

    def fwd(self, R, st):
        raise NotImplementedError

    def bwd(self, R, st):
        pass


class ConvBnRelu(Op):
    """ReLU(BN(conv3x3(x)+b)) — reference BasicConv2d (models/unet.py:5-17) / BasicConv (models/segnet.py:5-17)."""

    def __init__(self, src, dst, pslot, holder, cin, cout, src_needs_grad):
        self.src, self.dst, self.pslot, self.holder = src, dst, pslot, holder
        self.cin, self.cout, self.src_needs_grad = cin, cout, src_needs_grad
        self.pool_dst = None        # the ActBuf of a MaxPool2d(2,2) of this block's output, written by the BN-apply pass
        self.pool_op = None
        assert src.C == cin and dst.C == cout and dst.H == src.H and dst.W == src.W

    def _weight_fwd(self, R, st, w):
        """[Cout][9][ld_in]: the parameter itself when it is channels_last and needs no channel padding."""
        ldx = self.src.ld
        if ldx == self.cin and w.is_contiguous(memory_format=torch.channels_last):
            return w
        wc = w if w.is_contiguous(memory_format=torch.channels_last) else w.contiguous(memory_format=torch.channels_last)
        def build():
            out = _empty(self.cout * 9 * ldx, w.device)
            check(R.lib.cvk_pack_weight_fwd(wc.data_ptr(), out.data_ptr(), self.cout, self.cin, ldx, st.stream), "cvk_pack_weight_fwd")
            return out
        return R.derived(((self.pslot, "f"), "pack"), w, build)

    def _wgrad2d(self, R):
        """Does this layer's weight-grad run through the transposed 2-D F(4x4,3x3) (csrc/wino2d.hip)?  Channel-heavy layers:
        25-40 % faster than the transposed F(4,3) from 256 x 256 channels up (tools/bench_conv.py ww2d)."""
        src, C = self.src, self.cout
        if not (R.wino and src.ld % 4 == 0 and C % 4 == 0 and pad4(C) == C and src.ld >= 32 and C >= 64 and R.wino2d):
            return False
        if R.wino2d == "always" or wgrad2d_pays(src.N, src.H, src.W, src.ld, C):
            return True
        # 128 <-> 256 channels at 180x240: the x transform alone makes it a tie with the transposed F(4,3), but the forward
        # pass of these layers already runs the 2-D path and leaves V behind (0.78 of the F(4,3) time without that pass; 6x6 tiles: 0.46)
        return (wino_ok(R, src.ld, pad4(C)) and wino2d_ok(src.ld, C, pad4(C)) and wino2d_pays(src.N, src.H, src.W, src.ld, C, R.w2tile)
                and src.ld * C >= 32768)

    def _split3(self, R):
        """Does this layer run the OPT-IN split-operand GEMMs (runner.w2d_split; csrc/split3.hip)?  Only layers whose forward, data-grad
        and weight-grad ALL take the 2-D path with one tile size and whose channel counts the split weight-grad GEMM serves — the three
        GEMMs of such a layer share their split planes (V from the forward transform, V' and E from one pass over dy)."""
        if not split_fmt(R) or not self.src_needs_grad:
            return False
        src, C = self.src, self.cout
        N, H, W, ldy = src.N, src.H, src.W, pad4(self.cout)
        if not (self._wgrad2d(R) and R.w2both and ldy == C and src.ld == self.cin and src.ld % 32 == 0 and C % 32 == 0):
            return False
        if not (wino_ok(R, src.ld, ldy) and wino2d_ok(src.ld, C, ldy) and (R.wino2d == "always" or wino2d_pays(N, H, W, src.ld, C, R.w2tile))):
            return False
        if not (wino_ok(R, ldy, src.ld) and wino2d_ok(ldy, src.ld, src.ld) and (R.wino2d == "always" or wino2d_pays(N, H, W, ldy, src.ld, R.w2tile))):
            return False
        if layer_tile(R, N, H, W) != layer_tile(R, N, H, W, dgrad=True):
            return False
        return (C % 256 == 0 and src.ld % 128 == 0) or (src.ld % 256 == 0 and C % 128 == 0)

    def _wgrad4(self, R):
        """Does this layer's weight-grad run through the transposed F(4,3) (when the 2-D path does not take it)?"""
        return bool(R.wino and self.src.ld >= 32 and self.cout > 32 and (R.wino4 == "always" or (R.wino4 and self.src.ld >= 64)))

    def _want_planes(self, R, st):
        """Should the fused forward launch (if that is the kernel the layer runs) leave the weight-grad's V planes behind?"""
        src, C = self.src, self.cout
        return bool(st.need_grad and R.vplanes and R.wgradp and split_fmt(R) == 0 and self._wgrad4(R) and not self._wgrad2d(R)
                    and wgradp_ok(src.ld, C, pad4(C)) and vplanes_pays(src.N, src.H, src.W, src.ld, C))

    def _conv(self, R, st, X, wk, b, y, stats, kind, keep_v=None):
        """y = conv3x3(X, wk) + b (+ BN statistics partials): Winograd kernels when eligible, else direct."""
        lib, s, src = R.lib, st.stream, self.src
        N, H, W, M, C, ldy = src.N, src.H, src.W, src.M, self.cout, pad4(self.cout)
        sp = stats.data_ptr() if stats is not None else None
        if (R.thin and lib.cvk_thin_fwd_supported(src.ld, C, ldy) and (sp is None or src.ld <= 8 or src.ld == 64)
                and H * W * max(src.ld, ldy) * 4 < 2 ** 31):       # one image per buffer resource
            # the stem (3 -> 64) and the classifier head (64 -> 12): csrc/thin.hip, the thin side is one side of a 16x16x4 MFMA
            Pt = lib.cvk_thin_stat_partials(N, H, W, src.ld)
            cnt = sp + 4 * 2 * Pt * C if sp is not None else None
            head = src.ld == 64
            _timed(R, "k_thin_co_fwd" if head else "k_thin_ci_fwd", 18.0 * M * C * self.cin, lambda: check(
                lib.cvk_conv3x3_thin_fwd(X.data_ptr(), wk.data_ptr(), b.data_ptr(), y.data_ptr(), sp, cnt, N, H, W, src.ld, C, ldy, s),
                "cvk_conv3x3_thin_fwd"), executed=18.0 * M * (16 * self.cin if head else C * src.ld), nbytes=4.0 * M * (src.ld + ldy))
            return (Pt, cnt) if sp is not None else None
        if wino_ok(R, src.ld, ldy):
            return wino_conv(R, lib, s, X, wk, b.data_ptr(), y, sp, N, H, W, src.ld, C, ldy, 18.0 * M * C * self.cin, keep_v=keep_v,
                             wsrc=st.params[4 * self.pslot], ck=(self.pslot, "f"), split=split_fmt(R) if (st.need_grad and st.training and self._split3(R)) else 0, x_amax=st.amax.get(src.id),
                             h2=split_fmt(R) == 2 and st.need_grad and st.training, want_planes=keep_v is not None and self._want_planes(R, st))
        else:
            _timed(R, conv_kernel_name("fwd", ldy, src.ld), 18.0 * M * C * self.cin, lambda: check(
                lib.cvk_conv3x3_fwd(X.data_ptr(), wk.data_ptr(), b.data_ptr(), y.data_ptr(), sp, N, H, W, src.ld, C, ldy, s),
                "cvk_conv3x3_fwd"))

    def fwd(self, R, st):
        if st.plan.bf16:
            return self._fwd_bf16(R, st)
        lib, s = R.lib, st.stream
        src, dst = self.src, self.dst
        X = st.act[src.id]
        w, b, gamma, beta = st.params[4 * self.pslot:4 * self.pslot + 4]
        dev = X.device
        N, H, W = src.N, src.H, src.W
        M, C, ldy = src.M, self.cout, pad4(self.cout)
        wk = self._weight_fwd(R, st, w)
        y = _empty(M * ldy, dev)
        bnp = _empty(4 * ldy, dev)                      # mean | rstd | scale | shift
        pm, pr, psc, psh = (bnp.data_ptr() + 4 * ldy * i for i in range(4))
        conv, bn = self.holder.conv_bn()
        keep_v = [] if (st.need_grad and (self._wgrad2d(R) or self._want_planes(R, st))) else None      # transformed input, reused by the weight-grad
        if st.training:
            P = (M + _lib.CVK_STAT_ROWS - 1) // _lib.CVK_STAT_ROWS
            Pm = max(P, lib.cvk_w2d_stat_partials(N, H, W), lib.cvk_w6_stat_partials(N, H, W), lib.cvk_thin_stat_partials(N, H, W, src.ld))   # room for any partial layout (+ counts)
            stats = _empty(2 * Pm * C + Pm, dev)
            if M <= 1:
                raise ValueError(f"Expected more than 1 value per channel when training, got input size {[N, C, H, W]}")
            counted = self._conv(R, st, X, wk, b, y, stats, "fwd", keep_v=keep_v)
            wsb = lib.cvk_bn_finalize_workspace_bytes(Pm, C)
            ws = R.workspace(wsb, dev)
            track = bn.track_running_stats and bn.running_mean is not None
            mom = 0.1 if bn.momentum is None else float(bn.momentum)
            run = (bn.running_mean.data_ptr() if track else None, bn.running_var.data_ptr() if track else None,
                   bn.num_batches_tracked.data_ptr() if track else None)
            if counted is None:
                check(lib.cvk_bn_finalize(stats.data_ptr(), P, M, C, gamma.data_ptr(), beta.data_ptr(), pm, pr, psc, psh, *run,
                                          mom, float(bn.eps), ws.data_ptr(), wsb, s), "cvk_bn_finalize")
            else:       # partials with explicit pixel counts (2-D Winograd path: P2 <= P partials in the same buffer)
                check(lib.cvk_bn_finalize_counts(stats.data_ptr(), counted[1], counted[0], M, C, gamma.data_ptr(), beta.data_ptr(),
                                                 pm, pr, psc, psh, *run, mom, float(bn.eps), ws.data_ptr(), wsb, s),
                      "cvk_bn_finalize_counts")
        else:
            self._conv(R, st, X, wk, b, y, None, "fwd", keep_v=keep_v)
            check(lib.cvk_bn_eval_params(gamma.data_ptr(), beta.data_ptr(), bn.running_mean.data_ptr(),
                                         bn.running_var.data_ptr(), pm, pr, psc, psh, C, float(bn.eps), s), "cvk_bn_eval_params")
        out = R.alloc_act(st, dst.buf, dev)
        pooled = False
        if self.pool_dst is not None and C % 4 == 0:
            # the 2x2 max pool behind this block is written by the same pass (csrc/bn.hip k_bn_relu_apply_pool)
            pool = R.alloc_act(st, self.pool_dst, dev)
            code = None
            if self.pool_op.keep_code:
                code = torch.empty(self.pool_dst.M * self.pool_dst.ld, device=dev, dtype=torch.uint8)
            aw, ap = st.amax.get(dst.buf.id), st.amax.get(self.pool_dst.id)
            if aw is not None or ap is not None:
                rc = _timed(R, "k_bn_relu_apply<pool>", (8.0 + 1.0 + (0.25 if code is not None else 0.0)) * M * C, lambda: lib.cvk_bn_relu_apply_pool_amax(
                    y.data_ptr(), ldy, psc, psh, dst.cview(out), pool.data_ptr(), code.data_ptr() if code is not None else None,
                    N, H, W, C, aw.data_ptr() if aw is not None else None, ap.data_ptr() if ap is not None else None, s), "byte")
            else:
                rc = _timed(R, "k_bn_relu_apply<pool>", (8.0 + 1.0 + (0.25 if code is not None else 0.0)) * M * C, lambda: lib.cvk_bn_relu_apply_pool(
                    y.data_ptr(), ldy, psc, psh, dst.cview(out), pool.data_ptr(), code.data_ptr() if code is not None else None,
                    N, H, W, C, s), "byte")
            pooled = rc == 0
            if not pooled and ap is not None:
                st.amax.pop(self.pool_dst.id)          # the separate pool pass writes that buffer: its reader measures it itself
            if pooled and code is not None:
                st.saved[self.pool_op.idx] = code
        st.pooled_by_block[self.idx] = pooled
        if not pooled:
            aw = st.amax.get(dst.buf.id)
            if aw is not None:
                _timed(R, "k_bn_relu_apply", 8.0 * M * C, lambda: check(
                    lib.cvk_bn_relu_apply_amax(y.data_ptr(), ldy, psc, psh, dst.cview(out), N, H, W, C, aw.data_ptr(), s), "cvk_bn_relu_apply_amax"), "byte")
            else:
                _timed(R, "k_bn_relu_apply", 8.0 * M * C, lambda: check(
                    lib.cvk_bn_relu_apply(y.data_ptr(), ldy, psc, psh, dst.cview(out), N, H, W, C, s), "cvk_bn_relu_apply"), "byte")
        if st.need_grad:
            st.saved[self.idx] = (y, bnp, keep_v[0] if keep_v else None)

    def bwd(self, R, st):
        if st.plan.bf16:
            return self._bwd_bf16(R, st)
        lib, s = R.lib, st.stream
        src, dst = self.src, self.dst
        y, bnp, Vkept = st.saved.pop(self.idx)
        X = st.act[src.id]
        dev = X.device
        N, H, W = src.N, src.H, src.W
        M, C, ldy = src.M, self.cout, pad4(self.cout)
        pm, pr, psc, psh = (bnp.data_ptr() + 4 * ldy * i for i in range(4))
        w = st.params[4 * self.pslot]
        gw, gb, gg, gbe = R.grad_ptrs(st, self.pslot)
        dO = dst.cview(st.grad[dst.buf.id])
        PB = lib.cvk_bn_bwd_blocks(M)
        part = _empty(2 * PB * C, dev)
        pre = st.bnred.pop(self.idx, None)
        if pre is not None:     # the data-grad that wrote dO summed it already (csrc/wino4f.hip BNR epilogue)
            check(lib.cvk_colsum_finalize(pre[0].data_ptr(), pre[1], C, gbe, gg, s), "cvk_colsum_finalize")
        else:
            _timed(R, "k_bn_bwd<reduce>", 8.0 * M * C, lambda: check(
                lib.cvk_bn_bwd_reduce(dO, y.data_ptr(), ldy, psc, psh, pm, pr, part.data_ptr(), N, H, W, C, s), "cvk_bn_bwd_reduce"), "byte")
            check(lib.cvk_colsum_finalize(part.data_ptr(), PB, C, gbe, gg, s), "cvk_colsum_finalize")   # dbeta, dgamma
        del pre
        # round 6: with V planes from the forward launch the plane GEMM reads E0 / E5 (columns of dy) from dy itself — dy then carries a zeroed slack
        # behind its last row (cvk_wgradp_gemm_sm_dy) and the BatchNorm-backward pass writes four E planes instead of six
        planes_kept = Vkept is not None and getattr(Vkept, "cvk_sm", False) and Vkept.numel() == 6 * lib.cvk_wgradp_plane_rows(N, H, W) * src.ld
        e4p = bool(planes_kept and ldy == C and W % 4 == 0 and R.wgradp and R.e4p and self._wgrad4(R) and not self._wgrad2d(R) and wgradp_ok(src.ld, C, ldy))
        if e4p:
            slack = lib.cvk_wgradp_dy_slack(W) * ldy
            dyb = _empty(M * ldy + slack, dev)
            dyb[M * ldy:].zero_()
            dy = dyb[:M * ldy]
        else:
            dy = torch.zeros(M * ldy, device=dev, dtype=_F32) if ldy != C else _empty(M * ldy, dev)
        # layers whose weight-grad runs through the transposed F(4,3) get its transformed dy planes E1..E4 from this pass
        wgrad4 = self._wgrad4(R)
        # channel-heavy layers: transposed 2-D F(4x4,3x3), 36 GEMMs over the tile index (25-40 % faster than the transposed
        # F(4,3) from 256x256 channels up: tools/bench_conv.py ww2d); it transforms dy itself, so no E planes are needed
        wgrad2d = self._wgrad2d(R)
        wgrad4 = wgrad4 and not wgrad2d
        # 64-input-channel layers: both transforms outside the GEMM (csrc/wgradp.hip) — the E planes come from this pass, the V
        # planes cost one pass over x; pays while that pass is cheap (measured: 64 -> 64 @360x480 0.74x, 64 -> 128 @180x240 0.7x
        # the time of the transposed F(4,3) kernel; 128 input channels: the V pass eats the gain)
        # round 6: when the fused forward launch left V behind (slice-major planes), every such layer takes the plane GEMM and no pass over x runs
        have_planes = planes_kept
        wgradp = wgrad4 and R.wgradp and wgradp_ok(src.ld, C, ldy) and (R.wgradp == "always" or have_planes or wgradp_pays(N, H, W, src.ld, C))
        E = None
        E6 = None
        am_dy_fused = None
        want_amax = split_fmt(R) == 2 and st.training and bool(st.amax_spare) and self.src_needs_grad
        if wgradp:
            e4p = e4p and have_planes
            rows6 = lib.cvk_wgradp_plane_rows(N, H, W)
            E6 = _empty((4 if e4p else 6) * rows6 * C, dev)
            check((lib.cvk_wgradp_zero_pads4 if e4p else lib.cvk_wgradp_zero_pads)(E6.data_ptr(), N, H, W, C, s), "cvk_wgradp_zero_pads")
            PBe = lib.cvk_bn_bwd_e_blocks(N, H, W)
            blk = st.amax_spare[-1] if want_amax else None      # the opt-in fp16 data-grad scales by the largest |dy|: left by this pass
            ebytes = (12.0 * M + (16.0 if e4p else 24.0) * N * H * ((W + 3) // 4)) * C
            if blk is not None:
                rc = _timed(R, "k_bn_bwd<dx+E4p>" if e4p else "k_bn_bwd<dx+E6>", ebytes, lambda: lib.cvk_bn_bwd_dx_e_amax(
                    2 if e4p else 1, dO, y.data_ptr(), ldy, psc, psh, pm, pr, gg, gbe, dy.data_ptr(), ldy, E6.data_ptr(), part.data_ptr(),
                    N, H, W, C, 1 if st.training else 0, blk.data_ptr(), s), "byte")
            else:
                rc = _timed(R, "k_bn_bwd<dx+E4p>" if e4p else "k_bn_bwd<dx+E6>", ebytes, lambda: (lib.cvk_bn_bwd_dx_e4p if e4p else lib.cvk_bn_bwd_dx_e6)(
                    dO, y.data_ptr(), ldy, psc, psh, pm, pr, gg, gbe, dy.data_ptr(), ldy, E6.data_ptr(), part.data_ptr(),
                    N, H, W, C, 1 if st.training else 0, s), "byte")
            if rc == 0:
                R.defer_colsum(st, part, PBe, C, gb)           # conv bias grad: finalised with the others, in one launch
                if blk is not None:
                    am_dy_fused = st.amax_spare.pop()
            else:
                E6 = None           # layout not vectorisable (strided view): plain pass below, the weight-grad transforms dy itself
        if E6 is None and wgrad4 and not wgradp and ldy == C and C % 4 == 0:
            E = _empty(4 * N * H * ((W + 3) // 4) * ldy, dev)
            PBe = lib.cvk_bn_bwd_e_blocks(N, H, W)
            blk = st.amax_spare[-1] if want_amax else None
            if blk is not None:
                rc = _timed(R, "k_bn_bwd<dx+E>", (12.0 * M + 16.0 * N * H * ((W + 3) // 4)) * C, lambda: lib.cvk_bn_bwd_dx_e_amax(
                    0, dO, y.data_ptr(), ldy, psc, psh, pm, pr, gg, gbe, dy.data_ptr(), ldy, E.data_ptr(), part.data_ptr(),
                    N, H, W, C, 1 if st.training else 0, blk.data_ptr(), s), "byte")
            else:
                rc = _timed(R, "k_bn_bwd<dx+E>", (12.0 * M + 16.0 * N * H * ((W + 3) // 4)) * C, lambda: lib.cvk_bn_bwd_dx_e(
                    dO, y.data_ptr(), ldy, psc, psh, pm, pr, gg, gbe, dy.data_ptr(), ldy, E.data_ptr(), part.data_ptr(),
                    N, H, W, C, 1 if st.training else 0, s), "byte")
            if rc == 0:
                R.defer_colsum(st, part, PBe, C, gb)           # conv bias grad: finalised with the others, in one launch
                if blk is not None:
                    am_dy_fused = st.amax_spare.pop()
            else:
                E = None            # layout not vectorisable (strided view): plain pass below, wgrad transforms dy itself
        if E is None and E6 is None:
            if want_amax:
                am_dy_fused = st.amax_spare.pop()       # a zeroed word: the pass that writes dy leaves its largest magnitude there
                _timed(R, "k_bn_bwd<dx>", 12.0 * M * C, lambda: check(
                    lib.cvk_bn_bwd_dx_amax(dO, y.data_ptr(), ldy, psc, psh, pm, pr, gg, gbe, dy.data_ptr(), ldy, part.data_ptr(),
                                           N, H, W, C, 1 if st.training else 0, am_dy_fused.data_ptr(), s), "cvk_bn_bwd_dx_amax"), "byte")
            else:
                _timed(R, "k_bn_bwd<dx>", 12.0 * M * C, lambda: check(
                    lib.cvk_bn_bwd_dx(dO, y.data_ptr(), ldy, psc, psh, pm, pr, gg, gbe, dy.data_ptr(), ldy, part.data_ptr(),
                                      N, H, W, C, 1 if st.training else 0, s), "cvk_bn_bwd_dx"), "byte")
            R.defer_colsum(st, part, PB, C, gb)             # conv bias grad: finalised with the others, in one launch
        del y
        # weight-grad AND data-grad on the 2-D path with the same tile: dy is transformed for both in ONE launch (csrc/wino2d.hip
        # k_w2d_dy_both): E for the weight-grad, V' for the data-grad; dy crosses the fabric once
        both2 = None
        fmt = split_fmt(R) if (wgrad2d and st.training and self._split3(R)) else 0
        pdt = {3: _BF16, 2: torch.float16}.get(fmt)
        split3 = fmt if (fmt and Vkept is not None and Vkept.dtype == pdt) else 0
        if split3:
            tile = layer_tile(R, N, H, W)
            NX = 64 if tile == 6 else 36
            T = w2fn(lib, tile, "tiles")(N, H, W)
            Tp = lib.cvk_split3_rows_pad(T, 256)
            tag = "split3" if fmt == 3 else "split2h"
            Eb = torch.empty(NX * (C // 32) * fmt * Tp * 32, device=dev, dtype=pdt)
            Vb = torch.empty(NX * (C // 32) * fmt * Tp * 32, device=dev, dtype=pdt)
            am_dy = am_dy_fused
            if fmt == 2 and am_dy is None:
                am_dy = amax_blocks(lib, 1, dev)[0]
                _timed(R, "k_absmax", 4.0 * M * C, lambda: check(lib.cvk_absmax_f32(dy.data_ptr(), M, C, ldy, am_dy.data_ptr(), s), "cvk_absmax_f32(dy)"), "byte")
            Eb.cvk_amax = Vb.cvk_amax = am_dy
            _timed(R, "k_w2d_dy<both,%s>" % tag, (4.0 * M + 4.0 * fmt * NX * T) * C, lambda: check(
                lib.cvk_w2d_dy_transform_both_split(fmt, tile, dy.data_ptr(), ldy, Vb.data_ptr(), Eb.data_ptr(), 1,
                                                    am_dy.data_ptr() if am_dy is not None else None, N, H, W, C, s),
                "cvk_w2d_dy_transform_both_split"), "byte")
            both2 = (tile, Eb, Vb)
        elif Vkept is not None and Vkept.dtype in (_BF16, torch.float16):
            Vkept = None            # the forward pass ran a split path but this backward pass does not run the same (a knob changed in between)
        if (not split3 and wgrad2d and self.src_needs_grad and R.w2both and ldy == C and wino_ok(R, ldy, src.ld) and wino2d_ok(ldy, src.ld, src.ld)
                and (R.wino2d == "always" or (R.wino2d and wino2d_pays(N, H, W, ldy, src.ld, R.w2tile)))
                and layer_tile(R, N, H, W) == layer_tile(R, N, H, W, dgrad=True)):
            tile = layer_tile(R, N, H, W)
            NX = 64 if tile == 6 else 36
            T = w2fn(lib, tile, "tiles")(N, H, W)
            Tp = lib.cvk_w2d_tpad(T)
            Eb, Vb = _empty(NX * Tp * C + 128, dev), _empty(NX * Tp * C + 128, dev)
            _timed(R, "k_w2d_dy<both>", 4.0 * (M + 2 * NX * T) * C, lambda: check(
                w2fn(lib, tile, "dy_transform_both")(dy.data_ptr(), ldy, Vb.data_ptr(), Eb.data_ptr(), N, H, W, C, s), "cvk_w2d_dy_transform_both"), "byte")
            both2 = (tile, Eb, Vb)
        if self.src_needs_grad:
            if src.id in st.grad:
                raise NotImplementedError("conv data-grad must be the first writer of its input's gradient buffer")
            wc = w if w.is_contiguous(memory_format=torch.channels_last) else w.contiguous(memory_format=torch.channels_last)

            def packed():       # [Cin_pad][9][Cout_pad] rotated + transposed filter for the data-grad-as-forward kernels
                wd_ = _empty(src.ld * 9 * ldy, dev)
                _timed(R, "k_pack_weight_dgrad", 4.0 * 9 * (C * self.cin + src.ld * ldy), lambda: check(
                    lib.cvk_pack_weight_dgrad(wc.data_ptr(), wd_.data_ptr(), C, self.cin, src.ld, ldy, s), "cvk_pack_weight_dgrad"), "byte")
                return wd_
            dX = _empty(M * src.ld, dev).view(N, H, W, src.ld)
            if wino_ok(R, ldy, src.ld):
                # dX is the whole gradient of the producing block's activation when this conv is its only reader: the fused
                # kernel then sums it for that block's BatchNorm backward on the way out (training-mode statistics only)
                prod = st.plan.sole_producer(src) if st.training else None
                bnred = None
                if prod is not None and prod.idx in st.saved and pad4(prod.cout) == prod.cout == src.ld:
                    py, pbnp = st.saved[prod.idx][0], st.saved[prod.idx][1]
                    bnred = (py.data_ptr(), pbnp.data_ptr() + 8 * src.ld, pbnp.data_ptr() + 12 * src.ld, pbnp.data_ptr(),
                             pbnp.data_ptr() + 4 * src.ld, [])
                wino_conv(R, lib, s, dy, packed, None, dX, None, N, H, W, ldy, src.ld, src.ld, 18.0 * M * C * self.cin, "(dgrad)",
                          dgrad_of=(wc, C, self.cin), wsrc=w, ck=(self.pslot, "d"), bnred=bnred,
                          v_pre=(both2[0], both2[2]) if both2 is not None else None, split=split3, x_amax=am_dy_fused,
                          h2=split_fmt(R) == 2 and st.training)
                if bnred is not None and bnred[5]:
                    st.bnred[prod.idx] = bnred[5][0]
            elif R.thin and lib.cvk_thin_fwd_supported(ldy, src.ld, src.ld) and H * W * max(src.ld, ldy) * 4 < 2 ** 31:      # the head's data-grad: 12 -> 64 (csrc/thin.hip)
                wd = R.derived(((self.pslot, "d"), "pack"), w, packed)
                _timed(R, "k_thin_ci_fwd(dgrad)", 18.0 * M * C * self.cin, lambda: check(
                    lib.cvk_conv3x3_thin_fwd(dy.data_ptr(), wd.data_ptr(), None, dX.data_ptr(), None, None, N, H, W, ldy, src.ld, src.ld, s),
                    "cvk_conv3x3_thin_fwd(dgrad)"), executed=18.0 * M * ldy * src.ld, nbytes=4.0 * M * (src.ld + ldy))
            else:
                wd = R.derived(((self.pslot, "d"), "pack"), w, packed)
                _timed(R, conv_kernel_name("dgrad", src.ld, ldy), 18.0 * M * C * self.cin, lambda: check(
                    lib.cvk_conv3x3_fwd(dy.data_ptr(), wd.data_ptr(), None, dX.data_ptr(), None, N, H, W, ldy, src.ld, src.ld, s),
                    "cvk_conv3x3_fwd(dgrad)"))
            st.grad[src.id] = dX
        if wgrad2d and split3:
            tile, Eb3 = both2[0], both2[1]
            NX = 64 if tile == 6 else 36
            T = w2fn(lib, tile, "tiles")(N, H, W)
            Tp = lib.cvk_split3_rows_pad(T, 256)
            f = lib.cvk_w2d_gemm_tn_split3_ksplit(NX, Tp, src.ld, C)
            ws = R.workspace(4 * f * NX * C * src.ld, dev)
            am_e, am_v = getattr(Eb3, "cvk_amax", None), getattr(Vkept, "cvk_amax", None)
            _timed(R, "k_gemm_tn_" + ("split3" if split3 == 3 else "split2h"), 18.0 * M * C * self.cin, lambda: check(
                lib.cvk_w2d_gemm_tn_split(split3, tile, Eb3.data_ptr(), Vkept.data_ptr(), ws.data_ptr(), am_e.data_ptr() if am_e is not None else None,
                                          am_v.data_ptr() if am_v is not None else None, NX, Tp, src.ld, C, s), "cvk_w2d_gemm_tn_split"),
                executed=2.0 * (6 if split3 == 3 else 3) * NX * Tp * src.ld * C)
            _timed(R, "k_w2d_wgrad_out", 4.0 * (NX * f + 9) * C * self.cin, lambda: check(
                lib.cvk_w2d_wgrad_output_f(tile, ws.data_ptr(), gw, self.cin, src.ld, C, f, s), "cvk_w2d_wgrad_output_f"), "byte")
            del Vkept
        elif wgrad2d:
            tile = layer_tile(R, N, H, W)
            NX = 64 if tile == 6 else 36
            T = w2fn(lib, tile, "tiles")(N, H, W)
            Tp = lib.cvk_w2d_tpad(T)
            vfl, efl = NX * Tp * src.ld + 128, NX * Tp * C + 128
            f = w2fn(lib, tile, "wgrad_ksplit")(T, src.ld, C)
            if Vkept is not None and Vkept.numel() != vfl:
                Vkept = None            # forward ran with another tile size (the knob changed in between)
            if Vkept is None:           # forward ran another kernel (e.g. the mode changed in between): transform x now
                Vkept = _empty(vfl, dev)
                _timed(R, "k_w2d_input", 4.0 * (M + NX * T) * src.ld, lambda: check(
                    w2fn(lib, tile, "input_transform")(X.data_ptr(), Vkept.data_ptr(), N, H, W, src.ld, s), "cvk_w2d_input_transform(wgrad)"), "byte")
            if both2 is not None and both2[0] == tile:      # E came with the data-grad's V'
                ws = R.workspace(4 * (f * NX * C * src.ld), dev)
                Ep, Pp = both2[1].data_ptr(), ws.data_ptr()
            else:
                ws = R.workspace(4 * (efl + f * NX * C * src.ld), dev)
                Ep, Pp = ws.data_ptr(), ws.data_ptr() + 4 * efl
                _timed(R, "k_w2d_dy", 4.0 * (M + NX * T) * C, lambda: check(
                    w2fn(lib, tile, "dy_transform")(dy.data_ptr(), ldy, Ep, N, H, W, C, s), "cvk_w2d_dy_transform"), "byte")
            _timed(R, "k_w2d_gemm_tn", 18.0 * M * C * self.cin, lambda: check(
                w2fn(lib, tile, "gemm_tn")(Ep, Vkept.data_ptr(), Pp, T, src.ld, C, s), "cvk_w2d_gemm_tn"), executed=2.0 * NX * Tp * src.ld * C)
            _timed(R, "k_w2d_wgrad_out", 4.0 * (NX * f + 9) * C * self.cin, lambda: check(
                w2fn(lib, tile, "wgrad_output")(Pp, gw, T, self.cin, src.ld, C, s), "cvk_w2d_wgrad_output"), "byte")
            del Vkept
        elif wgradp:
            rows6 = lib.cvk_wgradp_plane_rows(N, H, W)
            wsb = lib.cvk_wgradp_gemm_workspace_bytes(N, H, W, src.ld, C)
            nE = 0 if E6 is not None else 6 * rows6 * C
            nV = 0 if have_planes else 6 * rows6 * src.ld
            ws = R.workspace(4 * (nV + nE) + wsb, dev)
            V6p = Vkept.data_ptr() if have_planes else ws.data_ptr()
            E6p = E6.data_ptr() if E6 is not None else ws.data_ptr() + 4 * nV
            slabp = ws.data_ptr() + 4 * (nV + nE)
            if not have_planes:
                _timed(R, "k_wgradp_planes", 4.0 * (M + 6.0 * rows6) * src.ld, lambda: check(
                    lib.cvk_wgradp_planes(X.data_ptr(), src.ld, V6p, N, H, W, src.ld, 0, s), "cvk_wgradp_planes(x)"), "byte")
            if E6 is None:
                _timed(R, "k_wgradp_planes", 4.0 * (M + 6.0 * rows6) * C, lambda: check(
                    lib.cvk_wgradp_planes(dy.data_ptr(), ldy, E6p, N, H, W, C, 1, s), "cvk_wgradp_planes(dy)"), "byte")
            if e4p and E6 is not None:
                _timed(R, "k_wgradp_gemm", 18.0 * M * C * self.cin, lambda: check(
                    lib.cvk_wgradp_gemm_sm_dy(E6p, dy.data_ptr(), V6p, gw, N, H, W, self.cin, src.ld, C, slabp, wsb, s), "cvk_wgradp_gemm_sm_dy"),
                    executed=9.0 * M * C * self.cin)
            else:
                gemm = lib.cvk_wgradp_gemm_sm if have_planes else lib.cvk_wgradp_gemm
                _timed(R, "k_wgradp_gemm", 18.0 * M * C * self.cin, lambda: check(
                    gemm(E6p, V6p, gw, N, H, W, self.cin, src.ld, C, slabp, wsb, s), "cvk_wgradp_gemm"), executed=9.0 * M * C * self.cin)
            del Vkept
        elif wgrad4:
            # transposed F(4,3): fastest weight-grad on every layer with >= 64 input channels (tools/bench_conv.py wgrad wwino wwino4)
            wsb = lib.cvk_conv3x3_wgrad_wino4_workspace_bytes(N, H, W, src.ld, C, ldy)
            ws = R.workspace(wsb, dev)
            _timed(R, f"k_wgrad_wino4<{'128' if C > 64 else '64'}, 128, 2, 2>", 18.0 * M * C * self.cin, lambda: check(
                lib.cvk_conv3x3_wgrad_wino4(X.data_ptr(), dy.data_ptr(), E.data_ptr() if E is not None else None, gw, N, H, W,
                                            self.cin, src.ld, C, ldy, ws.data_ptr(), wsb, s),
                "cvk_conv3x3_wgrad_wino4"))
        elif R.thin and lib.cvk_thin_wgrad_supported(self.cin, src.ld, C, ldy) and H * W * max(src.ld, ldy) * 4 < 2 ** 31:
            wsb = lib.cvk_conv3x3_thin_wgrad_workspace_bytes(N, H, W, src.ld, C)
            ws = R.workspace(wsb, dev)
            head = src.ld == 64
            _timed(R, "k_thin_co_wgrad" if head else "k_thin_ci_wgrad", 18.0 * M * C * self.cin, lambda: check(
                lib.cvk_conv3x3_thin_wgrad(X.data_ptr(), dy.data_ptr(), gw, N, H, W, self.cin, src.ld, C, ldy, ws.data_ptr(), wsb, s),
                "cvk_conv3x3_thin_wgrad"), executed=18.0 * M * (16 * self.cin if head else C * 16 / 3.0), nbytes=4.0 * M * (src.ld + ldy))
        elif R.wino and src.ld >= 32 and C > 32 and (src.ld > 64 or C > 64):   # 64->64 layers: the direct kernel is faster
            wsb = lib.cvk_conv3x3_wgrad_wino_workspace_bytes(N, H, W, src.ld, C)
            ws = R.workspace(wsb, dev)
            _timed(R, f"k_wgrad_wino<{'128' if C > 64 else '64'}, 128, 2, 2>", 18.0 * M * C * self.cin, lambda: check(
                lib.cvk_conv3x3_wgrad_wino(X.data_ptr(), dy.data_ptr(), gw, N, H, W, self.cin, src.ld, C, ldy, ws.data_ptr(), wsb, s),
                "cvk_conv3x3_wgrad_wino"))
        else:
            wsb = lib.cvk_conv3x3_wgrad_workspace_bytes(N, H, W, src.ld, C)
            ws = R.workspace(wsb, dev)
            _timed(R, conv_kernel_name("wgrad", C, src.ld), 18.0 * M * C * self.cin, lambda: check(
                lib.cvk_conv3x3_wgrad(X.data_ptr(), dy.data_ptr(), gw, N, H, W, self.cin, src.ld, C, ldy, ws.data_ptr(), wsb, s),
                "cvk_conv3x3_wgrad"))
        R.grads_ready(st, self.pslot)

    # ---- bf16-storage mode (BASELINE.json configs[3]; csrc/conv_bf16s.hip, csrc/elem_bf16.hip) ------------------------
    def _thin_bf16_mode(self, R, lib, dgrad, ld_dy=0):
        """csrc/thin_bf16.hip mode of this layer's forward (dgrad=False) / data-grad launch, 0 = the general kernels (also with CVK_THIN=0)."""
        if not R.thin or self.src.H * self.src.W * max(self.src.ld, self.cout, ld_dy, 64) * 2 >= 2 ** 31:
            return 0
        if dgrad:
            return lib.cvk_thin_bf16_mode(self.cin, self.cout, ld_dy, self.src.ld, 1) if self.src.ld == 64 and self.cin == 64 else 0
        return lib.cvk_thin_bf16_mode(self.cin, self.cout, self.src.ld, self.cout, 0)

    def _fwd_bf16(self, R, st):
        """bf16 NHWC activations in HBM: conv (bf16 MFMA, fp32 accumulate, fp32 statistics from the accumulators) writes
        the pre-BN tensor y as bf16; ONE elementwise pass applies BN + ReLU, writes the bf16 activation through the
        (concat) view and, for encoder stages, the 2x2 max-pooled tensor as well."""
        lib, s = R.lib, st.stream
        src, dst = self.src, self.dst
        X = st.act[src.id]
        w, b, gamma, beta = st.params[4 * self.pslot:4 * self.pslot + 4]
        dev = X.device
        N, H, W, M, C = src.N, src.H, src.W, src.M, self.cout
        if C % 4:
            raise NotImplementedError("bf16 mode needs output channel counts that are multiples of 4")
        wc = w if w.is_contiguous(memory_format=torch.channels_last) else w.contiguous(memory_format=torch.channels_last)
        def build_wb():
            t = torch.empty(lib.cvk_bf16s_rows_pad(C) * 9 * src.ld, device=dev, dtype=_BF16)
            check(lib.cvk_pack_weight_fwd_bf16(wc.data_ptr(), t.data_ptr(), C, self.cin, src.ld, s), "cvk_pack_weight_fwd_bf16")
            return t
        # the stem (3 -> 64) and the head (64 -> 12) run the register-only thin kernels (csrc/thin_bf16.hip, round 5)
        tmode = self._thin_bf16_mode(R, lib, False)

        def build_thin():
            t = torch.empty(lib.cvk_thin_bf16_pack_elems(tmode), device=dev, dtype=_BF16)
            check(lib.cvk_pack_weight_thin_bf16(wc.data_ptr(), t.data_ptr(), C, self.cin, tmode, s), "cvk_pack_weight_thin_bf16")
            return t
        wb = R.derived(((self.pslot, "f"), "thinb"), w, build_thin) if tmode else R.derived(((self.pslot, "f"), "bf16"), w, build_wb)
        y = torch.empty(M * C, device=dev, dtype=_BF16)
        bnp = _empty(4 * C, dev)
        pm, pr, psc, psh = (bnp.data_ptr() + 4 * C * i for i in range(4))
        conv, bn = self.holder.conv_bn()
        flops = 18.0 * M * C * self.cin
        tname = {1: "k_thinb_head_fwd", 2: "k_thinb_wide<1>"}.get(tmode)
        if st.training:
            if M <= 1:
                raise ValueError(f"Expected more than 1 value per channel when training, got input size {[N, C, H, W]}")
            P = lib.cvk_thin_bf16_stat_partials(N, H, W) if tmode else lib.cvk_bf16s_stat_partials_c(N, H, W, src.ld, C)
            stats = _empty(2 * P * C + P, dev)
            cnt = stats.data_ptr() + 4 * 2 * P * C
            if tmode:
                _timed(R, tname, flops, lambda: check(
                    lib.cvk_conv3x3_thin_bf16(X.data_ptr(), wb.data_ptr(), b.data_ptr(), y.data_ptr(), stats.data_ptr(), cnt, N, H, W, src.ld, C, C,
                                              tmode, s), "cvk_conv3x3_thin_bf16"), nbytes=2.0 * M * (src.ld + C))
            else:
                _timed(R, bf16_kernel_name(lib, N, H, W, src.ld, C, True), flops, lambda: check(
                    lib.cvk_conv3x3_bf16s_wg(X.data_ptr(), wb.data_ptr(), b.data_ptr(), y.data_ptr(), stats.data_ptr(), cnt, N, H, W, src.ld, C, C,
                                             R.launch_wgs(), s), "cvk_conv3x3_bf16s"))
            wsb = lib.cvk_bn_finalize_workspace_bytes(P, C)
            ws = R.workspace(wsb, dev)
            track = bn.track_running_stats and bn.running_mean is not None
            mom = 0.1 if bn.momentum is None else float(bn.momentum)
            check(lib.cvk_bn_finalize_counts(stats.data_ptr(), cnt, P, M, C, gamma.data_ptr(), beta.data_ptr(), pm, pr, psc, psh,
                                             bn.running_mean.data_ptr() if track else None,
                                             bn.running_var.data_ptr() if track else None,
                                             bn.num_batches_tracked.data_ptr() if track else None,
                                             mom, float(bn.eps), ws.data_ptr(), wsb, s), "cvk_bn_finalize_counts")
        else:
            if tmode:
                _timed(R, tname, flops, lambda: check(
                    lib.cvk_conv3x3_thin_bf16(X.data_ptr(), wb.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, N, H, W, src.ld, C, C, tmode, s),
                    "cvk_conv3x3_thin_bf16"), nbytes=2.0 * M * (src.ld + C))
            else:
                _timed(R, bf16_kernel_name(lib, N, H, W, src.ld, C, False), flops, lambda: check(
                    lib.cvk_conv3x3_bf16s(X.data_ptr(), wb.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, N, H, W, src.ld, C, C, s),
                    "cvk_conv3x3_bf16s"))
            check(lib.cvk_bn_eval_params(gamma.data_ptr(), beta.data_ptr(), bn.running_mean.data_ptr(),
                                         bn.running_var.data_ptr(), pm, pr, psc, psh, C, float(bn.eps), s), "cvk_bn_eval_params")
        out = R.alloc_act(st, dst.buf, dev)
        out_f32 = 1 if dst.buf.dtype == _F32 else 0
        pool = None
        if self.pool_dst is not None:
            pool = R.alloc_act(st, self.pool_dst, dev)
        nbytes = (2.0 + (4.0 if out_f32 else 2.0)) * M * C + (0.5 * M * C if pool is not None else 0.0)
        _timed(R, "k_apply_bf16" + ("<pool>" if pool is not None else ""), nbytes, lambda: check(
            lib.cvk_bn_relu_apply_bf16(y.data_ptr(), C, psc, psh, dst.hview(out), out_f32, pool.data_ptr() if pool is not None else None,
                                       N, H, W, C, s), "cvk_bn_relu_apply_bf16"), "byte")
        if st.need_grad:
            st.saved[self.idx] = (y, bnp)

    def _bwd_bf16(self, R, st):
        lib, s = R.lib, st.stream
        src, dst = self.src, self.dst
        y, bnp = st.saved.pop(self.idx)
        X = st.act[src.id]
        dev = X.device
        N, H, W, M, C = src.N, src.H, src.W, src.M, self.cout
        pm, pr, psc, psh = (bnp.data_ptr() + 4 * C * i for i in range(4))
        w = st.params[4 * self.pslot]
        gw, gb, gg, gbe = R.grad_ptrs(st, self.pslot)
        G = st.grad[dst.buf.id]
        dO = dst.hview(G)
        do_f32 = 1 if dst.buf.dtype == _F32 else 0
        esz = 4.0 if do_f32 else 2.0
        PB = lib.cvk_bn_bwd_blocks_bf16(M)
        part = _empty(2 * PB * C, dev)
        pre = st.bnred.pop(self.idx, None)
        if pre is not None:     # the pass that wrote dO last summed it already (csrc/elem_bf16.hip k_pool_bwd_bnred_bf16)
            check(lib.cvk_colsum_finalize(pre[0].data_ptr(), pre[1], C, gbe, gg, s), "cvk_colsum_finalize")
        else:
            _timed(R, "k_bnbwd_bf16<reduce>", (esz + 2.0) * M * C, lambda: check(
                lib.cvk_bn_bwd_reduce_bf16(dO, do_f32, y.data_ptr(), C, psc, psh, pm, pr, part.data_ptr(), N, H, W, C, s),
                "cvk_bn_bwd_reduce_bf16"), "byte")
            check(lib.cvk_colsum_finalize(part.data_ptr(), PB, C, gbe, gg, s), "cvk_colsum_finalize")   # dbeta, dgamma
        del pre
        ld_dy = max(32, C)                              # the data-grad GEMM reads dy in 32-channel K slices
        dy = torch.empty(M * ld_dy, device=dev, dtype=_BF16)
        _timed(R, "k_bnbwd_bf16<dx>", (esz + 4.0) * M * C, lambda: check(
            lib.cvk_bn_bwd_dx_bf16(dO, do_f32, y.data_ptr(), C, psc, psh, pm, pr, gg, gbe, dy.data_ptr(), ld_dy, part.data_ptr(),
                                   N, H, W, C, 1 if st.training else 0, s), "cvk_bn_bwd_dx_bf16"), "byte")
        R.defer_colsum(st, part, PB, C, gb)             # conv bias grad: finalised with the others, in one launch
        del y
        wc = w if w.is_contiguous(memory_format=torch.channels_last) else w.contiguous(memory_format=torch.channels_last)
        flops = 18.0 * M * C * self.cin
        if self.src_needs_grad:
            if src.id in st.grad:
                raise NotImplementedError("conv data-grad must be the first writer of its input's gradient buffer")
            def build_wd():
                t = torch.empty(lib.cvk_bf16s_rows_pad(self.cin) * 9 * ld_dy, device=dev, dtype=_BF16)
                check(lib.cvk_pack_weight_dgrad_bf16(wc.data_ptr(), t.data_ptr(), C, self.cin, ld_dy, s), "cvk_pack_weight_dgrad_bf16")
                return t
            dmode = self._thin_bf16_mode(R, lib, True, ld_dy)

            def build_thin_d():
                t = torch.empty(lib.cvk_thin_bf16_pack_elems(dmode), device=dev, dtype=_BF16)
                check(lib.cvk_pack_weight_thin_bf16(wc.data_ptr(), t.data_ptr(), C, self.cin, dmode, s), "cvk_pack_weight_thin_bf16")
                return t
            dX = torch.empty((N, H, W, src.ld), device=dev, dtype=_BF16)
            if dmode:
                wd = R.derived(((self.pslot, "d"), "thinb"), w, build_thin_d)
                _timed(R, "k_thinb_wide<3>(dgrad)", flops, lambda: check(
                    lib.cvk_conv3x3_thin_bf16(dy.data_ptr(), wd.data_ptr(), None, dX.data_ptr(), None, None, N, H, W, ld_dy, C, src.ld, dmode, s),
                    "cvk_conv3x3_thin_bf16(dgrad)"), nbytes=2.0 * M * (ld_dy + src.ld))
            else:
                wd = R.derived(((self.pslot, "d"), "bf16"), w, build_wd)
                _timed(R, bf16_kernel_name(lib, N, H, W, ld_dy, self.cin, False), flops, lambda: check(
                    lib.cvk_conv3x3_bf16s_wg(dy.data_ptr(), wd.data_ptr(), None, dX.data_ptr(), None, None, N, H, W, ld_dy, self.cin, src.ld,
                                             R.launch_wgs(), s), "cvk_conv3x3_bf16s(dgrad)"))
            st.grad[src.id] = dX
        # partial slabs now, the sum over the slabs with every other layer's in ONE launch (Runner.flush_wreduces): nobody reads a weight
        # gradient before the end of backward (or the all-reduce of its bucket); 23 reductions of ~11 us were 0.26 ms of a 21 ms step
        S = lib.cvk_conv3x3_wgrad_bf16s_splits(N, H, W, self.cin, C)
        n = C * 9 * self.cin
        slab = None if S == 1 else _empty(S * n, dev)           # one slab: it is the gradient itself, written in place
        _timed(R, "k_wgrad_bf16r", flops, lambda: check(
            lib.cvk_conv3x3_wgrad_bf16s_slabs(X.data_ptr(), dy.data_ptr(), gw if slab is None else slab.data_ptr(), N, H, W, self.cin, src.ld, C,
                                              ld_dy, 4 * S * n, s), "cvk_conv3x3_wgrad_bf16s_slabs"))
        if slab is not None:
            R.defer_wreduce(st, slab, S, n, gw)
        R.grads_ready(st, self.pslot)


class MaxPool(Op):
    """nn.MaxPool2d(2,2) (models/unet.py:92) / with indices (models/segnet.py:79)."""

    def __init__(self, src_view, dst_buf, keep_code):
        self.src, self.dst, self.keep_code = src_view, dst_buf, keep_code
        self.fused = False          # produced by the preceding block's BN-apply pass
        self.producer_idx = -1
        if src_view.H < 2 or src_view.W < 2:
            raise RuntimeError(f"max_pool2d: input {src_view.H}x{src_view.W} is too small for a 2x2 window")

    def fwd(self, R, st):
        v, d = self.src, self.dst
        X = st.act[v.buf.id]
        if st.plan.bf16:
            if not self.fused:
                raise NotImplementedError("bf16 mode: a max pool must directly follow a conv block (it is fused into its BN pass)")
            return                                      # the producing block's BN-apply pass already wrote st.act[d.id]
        if self.fused and st.pooled_by_block.get(self.producer_idx, False):
            return                                      # the producing block's BN-apply pass already wrote st.act[d.id] (and the code)
        out = R.alloc_act(st, d, X.device)
        code = None
        if self.keep_code:
            code = torch.empty(d.M * d.ld, device=X.device, dtype=torch.uint8)
            st.saved[self.idx] = code
        _timed(R, "k_maxpool_fwd", 5.0 * v.buf.N * v.H * v.W * v.C, lambda: check(
            R.lib.cvk_maxpool2x2_fwd(v.cview(X), out.data_ptr(), code.data_ptr() if code is not None else None,
                                     v.buf.N, v.H, v.W, v.C, st.stream), "cvk_maxpool2x2_fwd"), "byte")

    def bwd(self, R, st):
        v, d = self.src, self.dst
        if d.id not in st.grad:
            return
        X = st.act[v.buf.id]
        acc = v.buf.id in st.grad
        if not acc:
            if not v.is_full:
                st.grad[v.buf.id] = torch.zeros_like(X)
            else:
                st.grad[v.buf.id] = torch.empty_like(X)
        # round 6: when the pool directly follows a conv block (its BN-apply pass pooled) this pass is the LAST writer of the block's output
        # gradient — the consumers behind it in the plan ran earlier in backward — and touches every element of it: it then also leaves the block's
        # BatchNorm-backward sums (csrc/pointwise.hip k_pool_scatter_bnred, csrc/elem_bf16.hip k_pool_bwd_bnred_bf16) and the block's reduce pass is
        # not launched
        prod = st.plan.ops[self.producer_idx] if (self.fused and self.producer_idx >= 0) else None
        fuse_sums = (prod is not None and st.training and R.bnred_fuse and R.pool_bnred and isinstance(prod, ConvBnRelu) and prod.idx in st.saved
                     and prod.idx not in st.bnred and prod.dst.buf is v.buf
                     and (prod.dst.c0, prod.dst.C, prod.dst.y0, prod.dst.x0, prod.dst.H, prod.dst.W) == (v.c0, v.C, v.y0, v.x0, v.H, v.W)
                     and prod.cout == v.C and st.plan.last_gradient_writer(v, prod.idx) == self.idx)
        if st.plan.bf16:
            PBp = R.lib.cvk_maxpool2x2_bwd_bnred_blocks_bf16(v.buf.N, v.H, v.W, v.C) if fuse_sums else 0
            if PBp > 0:
                py, pbnp = st.saved[prod.idx][0], st.saved[prod.idx][1]
                part = _empty(2 * PBp * v.C, X.device)
                rc = _timed(R, "k_pool_bwd_bf16(+bnred)", (2.0 * 0.25 + 2.0 + (4.0 if acc else 2.0) + 2.0) * v.buf.N * v.H * v.W * v.C,
                            lambda: R.lib.cvk_maxpool2x2_bwd_bnred_bf16(
                                st.grad[d.id].data_ptr(), v.hview(X), v.hview(st.grad[v.buf.id]), 1 if acc else 0, v.buf.N, v.H, v.W, v.C,
                                py.data_ptr(), v.C, pbnp.data_ptr() + 8 * v.C, pbnp.data_ptr() + 12 * v.C, pbnp.data_ptr(), pbnp.data_ptr() + 4 * v.C,
                                part.data_ptr(), st.stream), "byte")
                if rc == 0:
                    st.bnred[prod.idx] = (part, PBp)
                    st.grad.pop(d.id)
                    return
            _timed(R, "k_pool_bwd_bf16", (2.0 * 0.25 + 2.0 + (4.0 if acc else 2.0)) * v.buf.N * v.H * v.W * v.C, lambda: check(
                R.lib.cvk_maxpool2x2_bwd_bf16(st.grad[d.id].data_ptr(), v.hview(X), v.hview(st.grad[v.buf.id]), 1 if acc else 0,
                                              v.buf.N, v.H, v.W, v.C, st.stream), "cvk_maxpool2x2_bwd_bf16"), "byte")
            st.grad.pop(d.id)
            return
        code = st.saved.get(self.idx)
        if fuse_sums and pad4(prod.cout) == prod.cout:
            PBp = R.lib.cvk_maxpool2x2_bwd_bnred_blocks(v.buf.N, v.H, v.W, v.C)
            if PBp > 0:
                py, pbnp = st.saved[prod.idx][0], st.saved[prod.idx][1]
                ldp = pad4(prod.cout)
                part = _empty(2 * PBp * v.C, X.device)
                rc = _timed(R, "k_pool_scatter(bwd+bnred)", (1.0 + (0.25 if code is not None else 4.0) + (8.0 if acc else 4.0) + 4.0) * v.buf.N * v.H * v.W * v.C,
                            lambda: R.lib.cvk_maxpool2x2_bwd_bnred(
                                st.grad[d.id].data_ptr(), v.cview(X), code.data_ptr() if code is not None else None, v.cview(st.grad[v.buf.id]),
                                1 if acc else 0, v.buf.N, v.H, v.W, v.C, py.data_ptr(), ldp, pbnp.data_ptr() + 8 * ldp, pbnp.data_ptr() + 12 * ldp,
                                pbnp.data_ptr(), pbnp.data_ptr() + 4 * ldp, part.data_ptr(), st.stream), "byte")
                if rc == 0:
                    st.bnred[prod.idx] = (part, PBp)
                    st.grad.pop(d.id)
                    return
        # bytes per input element: the pooled gradient (1) + the arg-max source (the activations, 4, or the 1-byte codes, 0.25) + the
        # gradient written (4) or accumulated (8)
        _timed(R, "k_pool_scatter(bwd)", (1.0 + (0.25 if code is not None else 4.0) + (8.0 if acc else 4.0)) * v.buf.N * v.H * v.W * v.C, lambda: check(
            R.lib.cvk_maxpool2x2_bwd(st.grad[d.id].data_ptr(), v.cview(X), code.data_ptr() if code is not None else None,
                                     v.cview(st.grad[v.buf.id]), 1 if acc else 0, v.buf.N, v.H, v.W, v.C, st.stream),
            "cvk_maxpool2x2_bwd"), "byte")
        st.grad.pop(d.id)


class Unpool(Op):
    """nn.MaxUnpool2d(2)(x, idx, output_size) (models/segnet.py:80,104-116)."""

    def __init__(self, src_buf, pool_op, dst_buf):
        self.src, self.pool, self.dst = src_buf, pool_op, dst_buf

    def fwd(self, R, st):
        V = st.act[self.src.id]
        d = self.dst
        out = R.alloc_act(st, d, V.device)
        if st.plan.bf16:
            # bf16 plans keep no index tensor: the arg-max is recomputed from the pooling layer's stored input, and the forward unpool
            # is the pool's backward kernel without accumulation (a scatter of the pooled values; every other pixel 0)
            pv = self.pool.src
            X = st.act[pv.buf.id]
            assert d.ld == d.C and self.src.ld == d.C, "bf16 unpool: dense tensors (C % 8 == 0, C >= 32)"
            _timed(R, "k_unpool_fwd_bf16", (2.0 * 0.25 + 2.0 + 2.0) * d.M * d.ld, lambda: check(
                R.lib.cvk_maxpool2x2_bwd_bf16(V.data_ptr(), pv.hview(X), d.full_view().hview(out), 0, d.N, d.H, d.W, d.C, st.stream),
                "cvk_maxpool2x2_bwd_bf16(unpool)"), "byte")
            return
        code = st.saved[self.pool.idx]
        _timed(R, "k_unpool_fwd", 5.25 * d.M * d.ld, lambda: check(        # reads pooled values + 1-byte codes, writes the full frame
            R.lib.cvk_maxunpool2x2_fwd(V.data_ptr(), code.data_ptr(), out.data_ptr(), d.N, d.H, d.W, d.ld, st.stream),
            "cvk_maxunpool2x2_fwd"), "byte")

    def bwd(self, R, st):
        d = self.dst
        g = st.grad.pop(d.id)
        dv = torch.empty_like(st.act[self.src.id])
        if st.plan.bf16:
            pv = self.pool.src
            X = st.act[pv.buf.id]
            _timed(R, "k_unpool_bwd_bf16", (2.0 + 2.0 + 2.0 * 0.25) * d.M * d.ld, lambda: check(
                R.lib.cvk_maxunpool2x2_bwd_bf16(g.data_ptr(), pv.hview(X), dv.data_ptr(), d.N, d.H, d.W, d.C, st.stream),
                "cvk_maxunpool2x2_bwd_bf16"), "byte")
            assert self.src.id not in st.grad
            st.grad[self.src.id] = dv
            return
        code = st.saved[self.pool.idx]
        _timed(R, "k_unpool_bwd", 5.25 * d.M * d.ld, lambda: check(
            R.lib.cvk_maxunpool2x2_bwd(g.data_ptr(), code.data_ptr(), dv.data_ptr(), d.N, d.H, d.W, d.ld, st.stream),
            "cvk_maxunpool2x2_bwd"), "byte")
        assert self.src.id not in st.grad
        st.grad[self.src.id] = dv


class Upsample(Op):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) (models/unet.py:25)."""

    def __init__(self, src_buf, dst_buf):
        self.src, self.dst = src_buf, dst_buf

    def fwd(self, R, st):
        X = st.act[self.src.id]
        out = R.alloc_act(st, self.dst, X.device)
        b = self.src
        if st.plan.bf16:
            _timed(R, "k_bilinear_fwd_bf16", 10.0 * b.M * b.ld, lambda: check(
                R.lib.cvk_bilinear_up2_fwd_bf16(X.data_ptr(), out.data_ptr(), b.N, b.H, b.W, b.ld, st.stream), "cvk_bilinear_up2_fwd_bf16"), "byte")
            return
        _timed(R, "k_bilinear_fwd", 20.0 * b.M * b.ld, lambda: check(
            R.lib.cvk_bilinear_up2_fwd(X.data_ptr(), out.data_ptr(), b.N, b.H, b.W, b.ld, st.stream), "cvk_bilinear_up2_fwd"), "byte")
        aw = st.amax.get(self.dst.id)
        if aw is not None:          # align_corners interpolation is a convex combination: |out| <= max |in| (an upper bound serves the scale)
            aw.copy_(st.amax[b.id])

    def bwd(self, R, st):
        g = st.grad.pop(self.dst.id)
        b = self.src
        dx = torch.empty_like(st.act[b.id])
        if st.plan.bf16:
            _timed(R, "k_bilinear_bwd_bf16", 10.0 * b.M * b.ld, lambda: check(
                R.lib.cvk_bilinear_up2_bwd_bf16(g.data_ptr(), dx.data_ptr(), b.N, b.H, b.W, b.ld, st.stream), "cvk_bilinear_up2_bwd_bf16"), "byte")
        else:
            _timed(R, "k_bilinear_bwd", 20.0 * b.M * b.ld, lambda: check(
                R.lib.cvk_bilinear_up2_bwd(g.data_ptr(), dx.data_ptr(), b.N, b.H, b.W, b.ld, st.stream), "cvk_bilinear_up2_bwd"), "byte")
        assert b.id not in st.grad
        st.grad[b.id] = dx


class ZeroFrame(Op):
    """The F.pad border of models/unet.py:120-123, written in place into the concat buffer (forward only)."""

    def __init__(self, view):
        self.view = view

    def fwd(self, R, st):
        v = self.view
        b = v.buf
        t = R.alloc_act(st, b, st.device)
        chan = BufView(b, v.c0, v.C, 0, 0, b.H, b.W)
        if st.plan.bf16:
            check(R.lib.cvk_zero_frame_bf16(chan.hview(t), b.N, b.H, b.W, v.C, v.y0, v.x0, v.H, v.W, st.stream), "cvk_zero_frame_bf16")
            return
        check(R.lib.cvk_zero_frame(chan.cview(t), b.N, b.H, b.W, v.C, v.y0, v.x0, v.H, v.W, st.stream), "cvk_zero_frame")


# ===================================================================================================== plan
class Plan:
    """Recorded op list for one (N, H, W) input geometry."""

    def __init__(self, N, cin, H, W, bf16=False):
        self.N, self.cin, self.H, self.W = N, cin, H, W
        self.bf16 = bool(bf16)      # bf16-storage plan: activation buffers are bf16 (the logits buffer stays fp32)
        self.bufs, self.ops, self.holders = [], [], []
        self._producer = {}         # (buffer id, first channel) -> the ConvBnRelu op that writes that view
        self._readers = {}          # buffer id -> number of ops that read the buffer (each one adds to its gradient)
        self.input = self.new_buf(cin, H, W, "input")
        self.output = None
        self.input_needs_grad = False

    def new_buf(self, C, H, W, name="", f32=False):
        b = ActBuf(len(self.bufs), self.N, H, W, C, name, _BF16 if (self.bf16 and not f32) else _F32)
        self.bufs.append(b)
        return b

    def add(self, op):
        op.idx = len(self.ops)
        self.ops.append(op)
        return op

    def _reads(self, buf):
        self._readers[buf.id] = self._readers.get(buf.id, 0) + 1

    def last_gradient_writer(self, view, producer_idx):
        """Index of the op whose backward pass writes LAST into the gradient of `view` (channels [c0, c0 + C) of its buffer): backward runs the plan
        in reverse, so it is the FIRST op behind the producer that reads any of those channels."""
        for op in self.ops[producer_idx + 1:]:
            s = getattr(op, "src", None)
            if s is None:
                continue
            if isinstance(s, BufView):
                if s.buf is view.buf and s.c0 < view.c0 + view.C and view.c0 < s.c0 + s.C:
                    return op.idx
            elif s is view.buf:
                return op.idx
        return -1

    def sole_producer(self, buf):
        """The ConvBnRelu block whose output is exactly `buf` when `buf` has ONE reader — that reader's data-grad is then
        the complete dL/d(activation) of the block, and may carry the block's BatchNorm-backward sums (csrc/wino4f.hip BNR)."""
        op = self._producer.get((buf.id, 0))
        if op is None or not op.dst.is_full or op.dst.buf is not buf or self._readers.get(buf.id, 0) != 1 or self.output is None \
                or self.output.buf is buf:
            return None
        return op

    # ---- builders used by the modules -----------------------------------------------------------------------
    def conv_bn_relu(self, src_buf, holder, dst_view=None):
        cin, cout = holder.in_channels, holder.out_channels
        if src_buf.C != cin:
            raise RuntimeError(f"Given groups=1, weight of size [{cout}, {cin}, 3, 3], expected input[{src_buf.N}, {src_buf.C}, "
                               f"{src_buf.H}, {src_buf.W}] to have {cin} channels, but got {src_buf.C} channels instead")
        if dst_view is None:
            dst_view = self.new_buf(cout, src_buf.H, src_buf.W, holder.tag).full_view()
        pslot = len(self.holders)
        self.holders.append(holder)
        needs = src_buf is not self.input or self.input_needs_grad
        self._reads(src_buf)
        op = self.add(ConvBnRelu(src_buf, dst_view, pslot, holder, cin, cout, needs))
        self._producer[(dst_view.buf.id, dst_view.c0)] = op
        return dst_view

    def maxpool(self, src_view, keep_code=False):
        dst = self.new_buf(src_view.C, src_view.H // 2, src_view.W // 2, "pool")
        self._reads(src_view.buf)
        op = self.add(MaxPool(src_view, dst, keep_code))
        prod = self._producer.get((src_view.buf.id, src_view.c0))
        if prod is not None and prod.dst.C == src_view.C and prod.pool_dst is None \
                and prod.dst.H == src_view.H and prod.dst.W == src_view.W and prod.dst.y0 == src_view.y0 and prod.dst.x0 == src_view.x0:
            prod.pool_dst = dst         # the block's BN-apply pass writes the pooled tensor too (csrc/elem_bf16.hip, csrc/bn.hip)
            prod.pool_op = op
            op.fused = True
            op.producer_idx = prod.idx
        return op

    def unpool(self, src_buf, pool_op):
        v = pool_op.src
        dst = self.new_buf(v.C, v.H, v.W, "unpool")
        self._reads(src_buf)
        self.add(Unpool(src_buf, pool_op, dst))
        return dst

    def upsample(self, src_buf):
        dst = self.new_buf(src_buf.C, 2 * src_buf.H, 2 * src_buf.W, "up")
        self._reads(src_buf)
        self.add(Upsample(src_buf, dst))
        return dst

    def zero_frame(self, view):
        self.add(ZeroFrame(view))

    def seal(self, thin=True):
        """Called once after recording.  bf16-storage plans hand their result (the logits) to the caller and to the loss
        in fp32: the buffer of the output view becomes an fp32 buffer (written by the last block's BN-apply pass).
        Their imported input keeps a pixel pitch of 8 instead of 32 channels when every reader is a stem the thin kernel serves
        (csrc/thin_bf16.hip mode 2: <= 4 real channels -> 64; the weight-grad kernel reads any pitch % 8 == 0): a 3-channel image
        was 64 bytes per pixel — 177 MB written by the import pass and read twice per step at 4 x 720 x 960, 133 MB of it padding."""
        if self.bf16:
            b = self.output.buf
            if not self.output.is_full:
                raise NotImplementedError("bf16 mode: the network output must be a whole buffer")
            b.dtype = _F32
            b.ld = pad4(b.C)
            readers = [op for op in self.ops if getattr(op, "src", None) is self.input]
            if (thin and not self.input_needs_grad and self.input.C <= 4 and readers and self.input is not self.output.buf
                    and all(isinstance(op, ConvBnRelu) and op.cout == 64 for op in readers)
                    and self.input.H * self.input.W * 64 * 2 < 2 ** 31):
                self.input.ld = 8


_SPLIT_ENV_LOGGED = False


def _split_mode_from_env():
    """CVK_W2D_SPLIT (unset | 0 | 1 = 3 | 2 | 3) selects the OPT-IN split-operand arithmetic for every network of the process.  Anything else is
    an error that names the variable; a non-zero value is reported once on stderr, because it moves the fp32 products onto 16-bit MFMAs."""
    global _SPLIT_ENV_LOGGED
    raw = os.environ.get("CVK_W2D_SPLIT")
    if raw is None:
        return 0
    modes = {"0": 0, "1": 3, "3": 3, "2": 2}
    if raw.strip() not in modes:
        raise ValueError(f"CVK_W2D_SPLIT={raw!r}: allowed values are 0 (off, the default), 2 (two fp16 terms) and 3 or 1 (three bf16 terms); "
                         "or leave it unset and call pytorch_camvid_amd.set_split_operands(net, mode)")
    mode = modes[raw.strip()]
    if mode and not _SPLIT_ENV_LOGGED:
        _SPLIT_ENV_LOGGED = True
        import sys
        print(f"pytorch_camvid_amd: CVK_W2D_SPLIT={raw} — fp32 convolution products run as split {'fp16 x 2' if mode == 2 else 'bf16 x 3'} operands "
              "(opt-in, not the exact-fp32 default)", file=sys.stderr)
    return mode


class Runner:
    """Executes a Plan.  One Runner per module instance; plans are cached per input geometry."""

    def __init__(self):
        self.lib = _lib.load()
        self._ws = None
        self._dp_wgs = None
        self._collectives_in_flight = False     # a gradient bucket's all-reduce has been issued and backward is still running
        self.grad_sync = None       # set by ddp.DataParallel
        self.wino = WINO_DEFAULT
        self.wino4 = WINO4_DEFAULT
        self.wino4f = WINO4F_DEFAULT
        self.wgradp = WGRADP_DEFAULT
        self.vplanes = VPLANES_DEFAULT      # fused forward launches leave the weight-grad's V planes behind (round 6)
        self.e4p = os.environ.get("CVK_E4P", "1") != "0"    # ... and the plane GEMM reads E0 / E5 from dy: four E planes instead of six (round 6)
        self.thin = os.environ.get("CVK_THIN", "1") != "0"    # csrc/thin.hip for the stem and the classifier head
        self.w2both = os.environ.get("CVK_W2D_DY_BOTH", "1") != "0"   # one launch transforms dy for the data-grad and the weight-grad
        self.bnred_fuse = os.environ.get("CVK_BNRED_FUSE", "1") != "0"   # BN-backward sums in the fused data-grad's epilogue
        self.pool_bnred = os.environ.get("CVK_POOL_BNRED", "1") != "0"   # ... and in the max-pool backward pass behind a conv block (round 6)
        self.wino2d = WINO2D_DEFAULT
        # OPT-IN split-operand modes (cvk.set_split_operands): the matrix products of the fp32 convolutions on the 16-bit
        # matrix pipe with split fp32 operands (csrc/split_fmt.h).  Not the product default; bench.py names the mode in `dtype` when it is on.
        self.w2d_split = _split_mode_from_env()      # 0 off | 3: bf16 x 3 | 2: fp16 x 2
        self.w2tile_cfg = W2TILE_DEFAULT    # None = auto (see W2TILE_DEFAULT), 4 or 6 = forced
        self.w2tile = 6                     # the tile of the plan being executed (set by forward / backward)
        self.w2tile_dgrad = 6               # ... of its data-grad launches
        self.bf16 = False           # opt-in: bf16-storage mode (modules.set_conv_precision; BASELINE.json configs[3])
        self.wcache = WCACHE_DEFAULT
        self.wepoch = 0             # bumped by mark_weights_dirty / FlatAdamW.step / load_state_dict
        self._wc = {}               # (layer slot, kind) -> (signature, tensor)
        self._wjobs, self._wjobs_cfg = {}, None      # batchable builds recorded by the last pass (prebuild_fp32)
        self._pass_token = 0        # 0: eval / no-grad passes (entries shared across calls); > 0: the training pass being executed
        self._passes = 0
        self.wcache_builds = 0      # derived tensors built since creation (tests / diagnostics)

    def persistent_wgs(self):
        """Workgroup cap of the persistent (one-workgroup-per-CU) kernels: 0 = every CU.  Under data-parallel training
        CVK_DP_RESERVE_CUS (default 8) CUs are left to RCCL's all-reduce kernels, which run beside backward."""
        gs = self.grad_sync
        if gs is None or (getattr(gs, "world", 1) <= 1 and not getattr(gs, "always_issue", False)):
            return 0
        if self._dp_wgs is None:
            cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
            self._dp_wgs = max(8, cus - int(os.environ.get("CVK_DP_RESERVE_CUS", "8")))
        return self._dp_wgs

    def launch_wgs(self):
        """Workgroup cap for a persistent-kernel launch NOW: the data-parallel cap only while a collective can be in flight — from the
        first gradient bucket's all-reduce to the end of backward (round 5).  The forward pass and the part of backward before the
        first bucket run on every CU: a capped grid changes the tile rounds of the persistent walk (768 tiles = 3 rounds of 256
        workgroups but 4 of 240: measured +7-8 % on the whole bf16 step at world 1 for ANY channel count between 2 and 16, so the cost
        was the lost fit, not the CUs), and there is nothing to leave CUs to before the first bucket is issued.  Results are bitwise
        independent of the cap (tests/test_gpu_wino4f.py, tests/test_gpu_bf16.py)."""
        return self.persistent_wgs() if self._collectives_in_flight else 0

    def derived(self, key, src, build, job=None):
        """The derived weight tensor `key` of parameter `src`: cached while the weights are provably unchanged.
        job = (family, floats, rows, cols, tile, dgrad): how a batched launch can build it straight from `src` (prebuild_fp32 then builds
        it with every other recorded tensor of the network at the start of the next pass)."""
        if not self.wcache or key is None or torch.cuda.is_current_stream_capturing():
            self.wcache_builds += 1
            return build()
        if job is not None:
            self._wjobs[key] = (src, job)
        sig = (WEIGHT_EPOCH[0], self.wepoch, src.data_ptr(), src._version, self._pass_token)
        ent = self._wc.get(key)
        if ent is not None and ent[0] == sig:
            return ent[1]
        t = build()
        self.wcache_builds += 1
        self._wc[key] = (sig, t)
        return t

    def prebuild_fp32(self, plan, st, need_grad):
        """fp32 plans: the Winograd-domain filters the previous pass over this plan asked for (fused F(4,3): 16 per UNet step; 2-D forward
        and data-grad filters: 26), rebuilt in TWO launches (cvk_wino4f_weight_transform_batch, cvk_w2d_weight_transform_batch) instead
        of 42 of 5-15 us each.  Which filters a layer needs is decided where the layer runs (wino_conv); that code records a job with
        every cached tensor it builds straight from a parameter, and this pass replays the record."""
        if not self.wcache or torch.cuda.is_current_stream_capturing():
            return
        cfg = (id(plan), self.w2tile, self.w2tile_dgrad, self.wino, self.wino4, self.wino4f, self.wino2d, self.wgradp, self.vplanes, self.e4p, self.thin, self.w2d_split)
        if cfg != self._wjobs_cfg:          # another plan or other kernel knobs: the record starts over with this pass
            self._wjobs, self._wjobs_cfg = {}, cfg
            return
        live = {p.data_ptr() for p in st.params}
        lib, fam_jobs = self.lib, {"w4f": [], "w2d": []}
        for key, (src, (fam, floats, rows, cols, tile, dgrad)) in self._wjobs.items():
            if src.data_ptr() not in live or (dgrad and not need_grad):
                continue
            sig = (WEIGHT_EPOCH[0], self.wepoch, src.data_ptr(), src._version, self._pass_token)
            ent = self._wc.get(key)
            if ent is not None and ent[0] == sig:
                continue
            t = _empty(floats, src.device)
            fam_jobs[fam].append((_lib.WtJob(src.data_ptr(), t.data_ptr(), rows, cols, tile, dgrad), 4.0 * (src.numel() + floats)))
            self._wc[key] = (sig, t)
            self.wcache_builds += 1
        for fam, fn in (("w4f", lib.cvk_wino4f_weight_transform_batch), ("w2d", lib.cvk_w2d_weight_transform_batch)):
            jobs = fam_jobs[fam]
            for i in range(0, len(jobs), _lib.WT_BATCH_MAX):
                chunk = jobs[i:i + _lib.WT_BATCH_MAX]
                arr = (_lib.WtJob * len(chunk))(*[j for j, _ in chunk])
                _timed(self, "k_weight_transform_batch", sum(b for _, b in chunk),           # filter read + transformed filter written
                       lambda: check(fn(ctypes.addressof(arr), len(chunk), st.stream), "cvk_%s_weight_transform_batch" % fam), "byte")

    def prepack_bf16(self, plan, st, need_grad):
        """bf16 plans: every weight pack the step will ask for and the cache does not hold (forward packs; data-grad packs when a
        backward pass follows) in ONE launch (cvk_pack_weights_bf16_batch) instead of one ~9 us launch per layer and direction."""
        if not self.wcache or torch.cuda.is_current_stream_capturing():
            return
        lib, jobs, keep = self.lib, [], []
        for op in plan.ops:
            if not isinstance(op, ConvBnRelu):
                continue
            w = st.params[4 * op.pslot]
            sig = (WEIGHT_EPOCH[0], self.wepoch, w.data_ptr(), w._version, self._pass_token)
            want = []
            if not op._thin_bf16_mode(self, lib, False):        # thin layers pack their own (tiny) filter formats where they run
                want.append((((op.pslot, "f"), "bf16"), 0, op.cout, op.src.ld))
            if need_grad and op.src_needs_grad and not op._thin_bf16_mode(self, lib, True, max(32, op.cout)):
                want.append((((op.pslot, "d"), "bf16"), 1, op.cin, max(32, op.cout)))
            for key, dgrad, rows, kpad in want:
                ent = self._wc.get(key)
                if ent is not None and ent[0] == sig:
                    continue
                wc = w if w.is_contiguous(memory_format=torch.channels_last) else w.contiguous(memory_format=torch.channels_last)
                t = torch.empty(lib.cvk_bf16s_rows_pad(rows) * 9 * kpad, device=w.device, dtype=_BF16)
                jobs.append(_lib.PackJob(wc.data_ptr(), t.data_ptr(), op.cout, op.cin, kpad, dgrad))
                keep.append(wc)
                self._wc[key] = (sig, t)
                self.wcache_builds += 1
        for i in range(0, len(jobs), _lib.PACK_BATCH_MAX):
            chunk = jobs[i:i + _lib.PACK_BATCH_MAX]
            arr = (_lib.PackJob * len(chunk))(*chunk)
            check(lib.cvk_pack_weights_bf16_batch(ctypes.addressof(arr), len(chunk), st.stream), "cvk_pack_weights_bf16_batch")

    def workspace(self, nbytes, dev):
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != dev:
            self._ws = torch.empty(max(nbytes, 1 << 20), device=dev, dtype=torch.uint8)
        return self._ws

    def alloc_act(self, st, buf, dev):
        t = st.act.get(buf.id)
        if t is None:
            shape = (buf.N, buf.H, buf.W, buf.ld)
            t = torch.zeros(shape, device=dev, dtype=buf.dtype) if buf.ld != buf.C else torch.empty(shape, device=dev, dtype=buf.dtype)
            st.act[buf.id] = t
        return t

    # ---- flat gradient buffer: parameter grads are views, laid out in REVERSE execution order --------------------
    def layout_grads(self, plan, params):
        return layout_grads(params)

    def grad_ptrs(self, st, slot):
        base = st.gflat.data_ptr()
        return tuple(base + 4 * st.goffs[4 * slot + j] for j in range(4))

    def defer_colsum(self, st, part, PB, C, out_ptr):
        """Queue a column-sum finalisation whose result nobody reads before the end of backward (or the all-reduce of its
        gradient bucket): flushed in ONE launch by flush_colsums."""
        st.colsums.append((part, PB, C, out_ptr))
        if len(st.colsums) >= _lib.COLSUM_BATCH_MAX:
            self.flush_colsums(st)

    def defer_wreduce(self, st, slab, splits, n, dw_ptr):
        st.wreduces.append((slab, splits, n, dw_ptr))
        if len(st.wreduces) >= _lib.WREDUCE_BATCH_MAX:
            self.flush_wreduces(st)

    def flush_wreduces(self, st):
        if not st.wreduces:
            return
        arr = (_lib.WReduceJob * len(st.wreduces))(*[_lib.WReduceJob(sl.data_ptr(), dw, n, sp, 0) for sl, sp, n, dw in st.wreduces])
        check(self.lib.cvk_wgrad_reduce_bf16s_batch(ctypes.addressof(arr), len(st.wreduces), st.stream), "cvk_wgrad_reduce_bf16s_batch")
        st.wreduces.clear()

    def flush_colsums(self, st):
        if not st.colsums:
            return
        arr = (_lib.ColsumJob * len(st.colsums))(*[_lib.ColsumJob(p.data_ptr(), o, pb, c) for p, pb, c, o in st.colsums])
        check(self.lib.cvk_colsum_finalize_batch(ctypes.addressof(arr), len(st.colsums), st.stream), "cvk_colsum_finalize_batch")
        st.colsums.clear()

    def grads_ready(self, st, slot):
        if st.sync is not None:
            if st.sync.closes_bucket():     # this layer completes a bucket that is handed to the all-reduce now: its queued bias and
                self.flush_colsums(st)      # weight gradient finalisations must have run (round 5: only then — flushing after every
                self.flush_wreduces(st)     # layer turned the two batched launches of a backward pass into 23 + 23 under data parallel)
            st.sync.layer_done(st, slot)
            self._collectives_in_flight = st.sync.in_flight()

    # ---- forward / backward -------------------------------------------------------------------------------------
    def plan_amax(self, plan, st):
        """fp16 split-operand mode (w2d_split = 2): the transforms of a split layer scale by the EXACT largest magnitude of the tensor they
        read (csrc/split_fmt.h).  Where every pass that writes a layer's input can leave that maximum on its way (BN-apply passes: csrc/bn.hip
        *_amax; bilinear upsampling: bounded by its input's maximum; the zero frame), the input's buffer gets a device word here and no extra
        pass runs; any other input is measured by cvk_absmax_f32 in wino_conv.  One zero fill per step for all words."""
        if split_fmt(self) != 2 or plan.bf16 or not (st.training and st.need_grad):
            return
        # every conv block that can run a split form: the 2-D Winograd layers (_split3) and the fused F(4,3) of the 64/128-channel levels
        layers = [op for op in plan.ops if isinstance(op, ConvBnRelu) and (op._split3(self) or (op.src.ld % 32 == 0 and op.cout % 4 == 0))]
        if not layers:
            return

        def writers(buf):
            out = []
            for op in plan.ops:
                if isinstance(op, ConvBnRelu) and (op.dst.buf is buf or (op.pool_dst is buf and op.cout % 4 == 0)):
                    out.append(op)
                elif isinstance(op, (MaxPool, Unpool, Upsample)) and op.dst is buf and not (isinstance(op, MaxPool) and op.fused):
                    out.append(op)
                elif isinstance(op, ZeroFrame) and op.view.buf is buf:
                    out.append(op)
            return out

        want, ok = [], {}

        def fusable(buf):
            if buf.id in ok:
                return ok[buf.id]
            ok[buf.id] = False                      # (cycles cannot occur; guards the recursion anyway)
            ws = writers(buf)
            good = bool(ws) and buf is not plan.input
            for op in ws:
                if isinstance(op, Upsample):
                    good = good and fusable(op.src)
                elif not isinstance(op, (ConvBnRelu, ZeroFrame)):
                    good = False
            ok[buf.id] = good
            if good:
                want.append(buf)
            return good
        for op in layers:
            fusable(op.src)
        blocks = amax_blocks(self.lib, len(want) + len(layers), st.device)
        for i, buf in enumerate(want):
            st.amax[buf.id] = blocks[i]
        st.amax_spare = blocks[len(want):]

    def tile_for(self, plan):
        """(forward / weight-grad tile, data-grad tile) of the 2-D path for a plan."""
        if self.w2tile_cfg in (4, 6):
            return self.w2tile_cfg, self.w2tile_cfg
        return (4, 6) if any(isinstance(op, Unpool) for op in plan.ops) else (6, 6)

    def forward(self, plan, x, params, training, need_grad):
        dev = x.device
        self.w2tile, self.w2tile_dgrad = self.tile_for(plan)
        self._collectives_in_flight = False     # finish() of the previous backward waited for every collective
        st = RunState(params, training, need_grad)
        st.device = dev
        st.plan = plan
        if training and need_grad:          # a training pass: derived weights are rebuilt for it (fail-safe, see the cache notes above)
            self._passes += 1
            st.pass_token = self._passes
        self._pass_token = st.pass_token
        st.stream = torch.cuda.current_stream(dev).cuda_stream
        inb = plan.input
        xp = x.permute(0, 2, 3, 1)
        if plan.bf16:
            t = torch.empty((inb.N, inb.H, inb.W, inb.ld), device=dev, dtype=_BF16)
            sN, sC, sH, sW = x.stride()
            check(self.lib.cvk_import_nchw_bf16(x.data_ptr(), sN, sC, sH, sW, t.data_ptr(), inb.ld, inb.N, inb.C, inb.H, inb.W,
                                                st.stream), "cvk_import_nchw_bf16")
            st.act[inb.id] = t
        elif inb.ld == inb.C and xp.is_contiguous():
            st.act[inb.id] = xp                               # already dense NHWC: zero-copy
        else:
            t = torch.empty((inb.N, inb.H, inb.W, inb.ld), device=dev, dtype=_F32)
            sN, sC, sH, sW = x.stride()
            check(self.lib.cvk_import_nchw(x.data_ptr(), sN, sC, sH, sW, t.data_ptr(), inb.ld, inb.N, inb.C, inb.H, inb.W,
                                           st.stream), "cvk_import_nchw")
            st.act[inb.id] = t
        if plan.bf16:
            self.prepack_bf16(plan, st, need_grad)
        else:
            self.prebuild_fp32(plan, st, need_grad)
            self.plan_amax(plan, st)
        for op in plan.ops:
            op.fwd(self, st)
        ov = plan.output
        out = st.act[ov.buf.id][..., :ov.buf.C].permute(0, 3, 1, 2)    # logical NCHW, channels_last strides
        if not need_grad:
            st.act.clear()
            st.saved.clear()
        return out, st

    def backward(self, plan, st, gout):
        dev = gout.device
        self.w2tile, self.w2tile_dgrad = self.tile_for(plan)
        st.stream = torch.cuda.current_stream(dev).cuda_stream
        self._pass_token = st.pass_token
        params = st.params
        st.goffs, total = self.layout_grads(plan, params)
        # A FRESH flat buffer per backward call (caching allocator: the block freed by `p.grad = None` comes straight
        # back).  The returned .grad tensors are views of it and autograd may hold them for as long as it likes (a
        # second pass through the same network inside one graph, torch.autograd.grad results the caller keeps, manual
        # accumulation across zero_grad(set_to_none=True)): no later backward ever writes into memory handed out here.
        st.gflat = torch.empty(total, device=dev, dtype=_F32)
        st.sync = self.grad_sync.begin(st, plan) if self.grad_sync is not None else None
        self._collectives_in_flight = False
        ob = plan.output.buf
        gp = gout.permute(0, 2, 3, 1)
        if ob.ld == ob.C and gp.is_contiguous():
            st.grad[ob.id] = gp
        else:
            g = torch.zeros((ob.N, ob.H, ob.W, ob.ld), device=dev, dtype=_F32) if ob.ld != ob.C else \
                torch.empty((ob.N, ob.H, ob.W, ob.ld), device=dev, dtype=_F32)
            sN, sC, sH, sW = gout.stride()
            check(self.lib.cvk_import_nchw(gout.data_ptr(), sN, sC, sH, sW, g.data_ptr(), ob.ld, ob.N, ob.C, ob.H, ob.W,
                                           st.stream), "cvk_import_nchw")
            st.grad[ob.id] = g
        for op in reversed(plan.ops):
            op.bwd(self, st)
        self.flush_colsums(st)
        self.flush_wreduces(st)
        dx = None
        if plan.input_needs_grad:
            gi = st.grad[plan.input.id]
            dx = gi[..., :plan.input.C].permute(0, 3, 1, 2)
            if plan.bf16:                   # the stem's data-grad is stored as bf16 like every other dX; x.grad is fp32 like x
                dx = dx.float()
        grads = []
        for i, p in enumerate(params):
            n = p.numel()
            seg = st.gflat[st.goffs[i]:st.goffs[i] + n]
            if p.dim() == 4:   # [Cout][3][3][Cin] storage -> logical OIHW with channels_last strides
                co, ci, kh, kw = p.shape
                seg = seg.view(co, kh, kw, ci).permute(0, 3, 1, 2)
            else:
                seg = seg.view(p.shape)
            grads.append(seg)
        if st.sync is not None:
            st.sync.finish(st)
        self._collectives_in_flight = False
        st.act.clear(); st.saved.clear(); st.grad.clear()
        return dx, grads


def layout_grads(params):
    """Offsets (in floats) of every parameter inside the flat gradient buffer.  `params` is the executor's flat list
    [w, b, gamma, beta] per conv block in execution order; the LAST-executed block comes first in the buffer, so the
    buffer fills front to back during backward and contiguous prefixes can be all-reduced early (ddp.GradSync)."""
    offs = [0] * len(params)
    o = 0
    for slot in range(len(params) // 4 - 1, -1, -1):
        for j in range(4):
            i = 4 * slot + j
            offs[i] = o
            o += (params[i].numel() + 3) // 4 * 4     # keep every view 16-byte aligned
    return offs, o


class _PlanFunction(torch.autograd.Function):
    """The whole recorded network as one autograd node: (x, *parameters) -> logits."""

    @staticmethod
    def forward(ctx, runner, plan, training, x, *params):
        out, st = runner.forward(plan, x, params, training, True)
        ctx.runner, ctx.plan, ctx.st = runner, plan, st
        return out

    @staticmethod
    def backward(ctx, gout):
        st = ctx.st
        if st is None:
            raise RuntimeError("Trying to backward through the plan a second time (activations were already freed)")
        ctx.st = None
        dx, grads = ctx.runner.backward(ctx.plan, st, gout)
        return (None, None, None, dx, *grads)


def run_plan(runner, plan, training, x, params):
    need = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
    if not need:
        out, _ = runner.forward(plan, x, params, training, False)
        return out
    return _PlanFunction.apply(runner, plan, training, x, *params)
