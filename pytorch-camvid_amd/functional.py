"""Loss / evaluation operators of the hot path as autograd-aware callables on HIP tensors.

  CrossEntropyLoss  <- nn.CrossEntropyLoss() of reference train.py:105 (mean over N*H*W, no ignored class)
  argmax_channels   <- preds.argmax(dim=1) of train.py:191
  ConfusionMeter    <- utils.intersect_and_union / mean_iou (utils.py:162-228) accumulated on device, with the
                       np.float crash (utils.py:210) and the per-batch-sum bug of train.py:192-206 not reproduced
                       (SURVEY.md §0.5): IoU = sum(intersection)/sum(union) per class over the whole set.
"""
import torch
import torch.nn as nn

from . import _lib
from ._lib import check


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _as_nhwc(logits):
    """[N,C,H,W] logical -> (tensor whose memory is dense NHWC rows, ld).  Zero-copy for channels_last producers
    (our networks return channels_last strides), one copy otherwise."""
    if logits.dim() != 4:
        raise ValueError("expected logits of shape [N, C, H, W]")
    p = logits.permute(0, 2, 3, 1)
    if p.is_contiguous():
        return p, logits.shape[1]
    st = p.stride()
    N, H, W, C = p.shape
    if st[3] == 1 and st[2] >= C and st[1] == st[2] * W and st[0] == st[1] * H:
        return p, st[2]                                   # padded pixel stride (ld > C)
    return p.contiguous(), C


class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, grad_scale, ignore_index=-100):
        lib = _lib.load()
        if not logits.is_cuda:
            raise RuntimeError("pytorch_camvid_amd.CrossEntropyLoss needs HIP tensors (no CPU fallback)")
        if logits.dtype != torch.float32 or target.dtype != torch.int64:
            raise RuntimeError(f"expected float32 logits and int64 target, got {logits.dtype} / {target.dtype}")
        N, C, H, W = logits.shape
        if tuple(target.shape) != (N, H, W):
            raise ValueError(f"Expected target size {[N, H, W]}, got {list(target.shape)}")
        lg, ld = _as_nhwc(logits)
        tg = target.contiguous()
        M = N * H * W
        part = torch.empty(3 * lib.cvk_ce_blocks(M), device=logits.device, dtype=torch.float32)
        loss3 = torch.empty(3, device=logits.device, dtype=torch.float32)     # mean loss | valid pixels | out-of-range targets
        e0 = e1 = None
        from . import engine
        engine._timed(None, "k_ce_fwd", 4.0 * M * ld + 8.0 * M, lambda: check(
            lib.cvk_softmax_ce_fwd(lg.data_ptr(), ld, tg.data_ptr(), part.data_ptr(), loss3.data_ptr(), M, C, int(ignore_index),
                                   _stream(logits)), "cvk_softmax_ce_fwd"), "byte")
        ctx.save_for_backward(lg, tg, loss3)
        ctx.meta = (N, C, H, W, ld, grad_scale, int(ignore_index))
        _CrossEntropy.last_status = loss3
        return loss3[0].clone()     # not a view: loss3 is saved for backward (its divisor) and published as the status

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        lg, tg, loss3 = ctx.saved_tensors
        N, C, H, W, ld, grad_scale, ignore_index = ctx.meta
        M = N * H * W
        d = torch.empty((N, H, W, C), device=lg.device, dtype=torch.float32)
        g = gout.contiguous()
        from . import engine
        engine._timed(None, "k_ce_bwd", 4.0 * M * ld + 8.0 * M + 4.0 * M * C, lambda: check(
            lib.cvk_softmax_ce_bwd(lg.data_ptr(), ld, tg.data_ptr(), loss3.data_ptr(), g.data_ptr(), float(grad_scale), d.data_ptr(),
                                   C, M, C, ignore_index, _stream(lg)), "cvk_softmax_ce_bwd"), "byte")
        return d.permute(0, 3, 1, 2), None, None, None


class CrossEntropyLoss(nn.Module):
    """Drop-in for the reference's `nn.CrossEntropyLoss()` (train.py:105): reduction='mean', no class weights, no label
    smoothing; targets are int64 class indices in [0, C) or `ignore_index` (default -100 as in torch: such pixels are
    left out of the mean and get no gradient).  Any other out-of-range target makes the loss NaN — torch raises a
    device-side assert there; raising here would need a host sync in every step — and `last_ce_status()` reports the
    count.  `grad_scale` multiplies the backward only (data-parallel training can fold 1/world_size in here)."""

    def __init__(self, grad_scale=1.0, ignore_index=-100):
        super().__init__()
        self.grad_scale = grad_scale
        self.ignore_index = ignore_index

    def forward(self, logits, target):
        return _CrossEntropy.apply(logits, target, self.grad_scale, self.ignore_index)


def cross_entropy(logits, target, ignore_index=-100):
    return _CrossEntropy.apply(logits, target, 1.0, ignore_index)


def last_ce_status():
    """(valid pixels, out-of-range targets) of the most recent cross-entropy forward — one device->host copy; raises
    IndexError like torch's `Target out of bounds` when the second number is not zero."""
    st = getattr(_CrossEntropy, "last_status", None)
    if st is None:
        raise RuntimeError("no cross-entropy forward has run yet")
    _, valid, bad = st.cpu().tolist()
    if bad:
        raise IndexError(f"Target out of bounds: {int(bad)} pixels have a class index outside [0, C) that is not ignore_index")
    return int(valid), int(bad)


def argmax_channels(logits):
    """preds.argmax(dim=1) (train.py:191): int64 [N,H,W]; first maximum wins."""
    lib = _lib.load()
    N, C, H, W = logits.shape
    lg, ld = _as_nhwc(logits.detach())
    out = torch.empty((N, H, W), device=logits.device, dtype=torch.int64)
    check(lib.cvk_argmax_channels(lg.data_ptr(), ld, out.data_ptr(), N * H * W, C, _stream(logits)), "cvk_argmax_channels")
    return out


class ConfusionMeter:
    """Device-side accumulation of per-class intersection / prediction / label pixel counts (utils.py:162-190)."""

    def __init__(self, num_classes=12, ignore_index=11, device="cuda"):
        self.num_classes, self.ignore_index = num_classes, ignore_index
        self.hist = torch.zeros((3, num_classes), device=device, dtype=torch.int64)

    def reset(self):
        self.hist.zero_()

    def update(self, pred, label):
        lib = _lib.load()
        p = pred.contiguous(); l = label.contiguous()
        if p.dtype != torch.int64 or l.dtype != torch.int64 or p.shape != l.shape:
            raise ValueError("pred and label must be int64 tensors of the same shape")
        check(lib.cvk_confusion_accumulate(p.data_ptr(), l.data_ptr(), self.hist.data_ptr(), p.numel(), self.num_classes,
                                           self.ignore_index, _stream(p)), "cvk_confusion_accumulate")

    def compute(self):
        """(overall accuracy, per-class IoU tensor, mIoU over the non-ignored classes) — one device->host copy."""
        h = self.hist.cpu().double()
        inter, pred, lab = h[0], h[1], h[2]
        union = pred + lab - inter
        iou = inter / union
        valid = [c for c in range(self.num_classes) if c != self.ignore_index]
        acc = float(inter.sum() / lab.sum().clamp(min=1))
        return acc, iou, float(torch.nanmean(iou[valid]))

    def precision_recall(self):
        """Mean precision and recall over the non-ignored classes (the two extra numbers reference eval.py:70-79 prints;
        legacy/metrics.py:33-57: diag / column sums and diag / row sums of the confusion matrix).  Pixels whose label is
        ignore_index are left out of every count, as in utils.intersect_and_union (utils.py:170-172)."""
        h = self.hist.cpu().double()
        valid = [c for c in range(self.num_classes) if c != self.ignore_index]
        prec = (h[0] / (h[1] + 1e-15))[valid].mean()
        rec = (h[0] / (h[2] + 1e-15))[valid].mean()
        return float(prec), float(rec)


# reference conf/settings.py:8-9 (BGR order, as cv2 decodes)
CAMVID_MEAN = (0.42019099703461577, 0.41323568513979647, 0.4010048431259079)
CAMVID_STD = (0.30598050258519743, 0.3089986932156864, 0.3054061869915674)


def preprocess_uint8(images_u8, mean=CAMVID_MEAN, std=CAMVID_STD):
    """Device-side ToTensor + Normalize (reference transforms.py:485-538): uint8 [N,H,W,3] on the GPU -> float32
    logical [N,3,H,W] (a channels_last view of an NHWC-4 buffer).  Replaces the per-sample CPU float conversion and the
    pageable `images.cuda()` copy of 4 bytes/value (train.py:126) by a 1 byte/value upload + one kernel."""
    import ctypes
    lib = _lib.load()
    if images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or images_u8.shape[-1] != 3 or not images_u8.is_cuda:
        raise ValueError("expected a uint8 HIP tensor of shape [N, H, W, 3]")
    src = images_u8.contiguous()
    N, H, W, _ = src.shape
    dst = torch.empty((N, H, W, 4), device=src.device, dtype=torch.float32)
    m = (ctypes.c_float * 3)(*mean); sd = (ctypes.c_float * 3)(*std)
    check(lib.cvk_preprocess_u8(src.data_ptr(), dst.data_ptr(), N, H, W, m, sd, _stream(src)), "cvk_preprocess_u8")
    return dst[..., :3].permute(0, 3, 1, 2)


@torch.no_grad()
def evaluate(net, batches, num_classes=12, ignore_index=11):
    """Validation pass of reference train.py:169-206 / eval.py:44-80 without their bugs: eval-mode forward, device-side
    argmax and histogram accumulation over the WHOLE set, one host copy at the end.
    `batches` yields (images [N,3,H,W] float32, masks [N,H,W] int64) on the GPU.  Returns (accuracy, per-class IoU, mIoU)."""
    was_training = net.training
    net.eval()
    meter = None
    for images, masks in batches:
        logits = net(images)
        if meter is None:
            meter = ConfusionMeter(num_classes, ignore_index, logits.device)
        meter.update(argmax_channels(logits), masks)
    net.train(was_training)
    if meter is None:
        raise ValueError("evaluate(): no batches")
    return meter.compute()


def evaluate_report(net, batches, num_classes=12, ignore_index=11, loss_fn=None):
    """The report of reference eval.py:44-80: {"miou", "precision", "recall", "loss" (mean over batches), "accuracy",
    "iou" (per class)} for a set of (images, masks) batches on the GPU; eval-mode forward under no_grad, argmax and
    histograms on the device, one host copy at the end (the reference copies N*H*W int64 per batch, eval.py:60-62)."""
    loss_fn = loss_fn or CrossEntropyLoss()
    was_training = net.training
    net.eval()
    meter, loss_sum, n = None, None, 0
    with torch.no_grad():
        for images, masks in batches:
            logits = net(images)
            if meter is None:
                meter = ConfusionMeter(num_classes, ignore_index, logits.device)
            l = loss_fn(logits, masks).detach()
            loss_sum = l if loss_sum is None else loss_sum + l
            n += 1
            meter.update(argmax_channels(logits), masks)
    net.train(was_training)
    if meter is None:
        raise ValueError("evaluate_report(): no batches")
    acc, iou, miou = meter.compute()
    prec, rec = meter.precision_recall()
    return {"miou": miou, "precision": prec, "recall": rec, "loss": float(loss_sum) / n, "accuracy": acc, "iou": iou}


def predict(net, image_u8, out_size=None, mean=CAMVID_MEAN, std=CAMVID_STD):
    """reference predict.py:35-57 from the decoded image onward: `image_u8` is one uint8 [H, W, 3] frame (BGR, as cv2
    decodes; a CPU or GPU tensor or a numpy array) already at the network's input size; normalisation, eval-mode forward
    and channel argmax run on the device.  Returns the int64 class map [H, W]; with out_size=(h, w) it is resized by
    nearest neighbour the way `cv2.resize(..., INTER_NEAREST)` does (predict.py:55; source index floor(dst * in / out)).
    Image decoding / PIL resizing to IMAGE_SIZE stay on the host side (cv2 / PIL are not part of this package)."""
    dev = next(net.parameters()).device
    img = torch.as_tensor(image_u8)
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[-1] != 3:
        raise ValueError("expected one uint8 image of shape [H, W, 3]")
    x = preprocess_uint8(img.to(dev).unsqueeze(0), mean, std)
    was_training = net.training
    net.eval()
    with torch.no_grad():
        cls = argmax_channels(net(x))[0]
    net.train(was_training)
    if out_size is not None:
        h, w = out_size
        H, W = cls.shape
        yi = torch.clamp((torch.arange(h, device=dev, dtype=torch.float64) * (H / h)).floor().long(), max=H - 1)
        xi = torch.clamp((torch.arange(w, device=dev, dtype=torch.float64) * (W / w)).floor().long(), max=W - 1)
        cls = cls[yi][:, xi]
    return cls


class DevicePrefetcher:
    """the reference converts every frame to float on the CPU (transforms.py:485-538) and uploads 4 bytes per
    value from pageable memory inside the step (`images.cuda()`, train.py:126-127).  This iterator takes (uint8 frames
    [N,H,W,3] BGR, int64 masks [N,H,W]) batches from any host iterable (numpy arrays or CPU tensors), stages them in
    pinned buffers and uploads 1 byte per value on a side stream one batch ahead of the consumer; normalisation to the
    network's float NHWC layout happens on the device (`preprocess_uint8`).  Yields (images float32 [N,3,H,W] view, masks).
    The yielded tensors are safe to use on the current stream (the side stream's work is awaited before they are handed out)."""

    def __init__(self, batches, device="cuda", mean=CAMVID_MEAN, std=CAMVID_STD):
        self.it = iter(batches)
        self.dev = torch.device(device)
        self.mean, self.std = mean, std
        self.stream = torch.cuda.Stream(self.dev)
        self._pin = [None, None]        # two pinned staging slots (frames, masks), reused when shapes repeat
        self._busy = [None, None]       # per slot: event recorded after the H2D copies that READ its pinned buffers
        self._slot = 0
        self._next = None
        self._stage()

    def _pinned(self, slot, which, like):
        cur = self._pin[slot]
        buf = None if cur is None else cur[which]
        if buf is None or buf.shape != like.shape or buf.dtype != like.dtype:
            buf = torch.empty(like.shape, dtype=like.dtype, pin_memory=True)
            if cur is None:
                self._pin[slot] = [None, None]
            self._pin[slot][which] = buf
        return buf

    def _stage(self):
        try:
            frames, masks = next(self.it)
        except StopIteration:
            self._next = None
            return
        frames, masks = torch.as_tensor(frames), torch.as_tensor(masks)
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
            raise ValueError("expected uint8 frames of shape [N, H, W, 3]")
        slot = self._slot
        self._slot ^= 1
        if self._busy[slot] is not None:
            self._busy[slot].synchronize()              # the upload that last read this slot must be done before the host overwrites it
        pf = self._pinned(slot, 0, frames); pf.copy_(frames)
        pm = self._pinned(slot, 1, masks); pm.copy_(masks)
        with torch.cuda.stream(self.stream):
            gf = pf.to(self.dev, non_blocking=True)
            gm = pm.to(self.dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._busy[slot] = ev
        self._next = (gf, gm, ev)

    def __iter__(self):
        return self

    def __next__(self):
        if self._next is None:
            raise StopIteration
        gf, gm, ev = self._next
        cur = torch.cuda.current_stream(self.dev)
        cur.wait_event(ev)
        gf.record_stream(cur); gm.record_stream(cur)
        self._stage()                                   # upload of the following batch overlaps the consumer's step
        return preprocess_uint8(gf, self.mean, self.std), gm
