"""FlatAdamW — torch.optim.AdamW semantics (reference train.py:100,133) as ONE fused kernel over flat buffers.

The executor already writes every gradient into one flat fp32 buffer (engine.layout_grads).  FlatAdamW re-homes the
parameters into a flat buffer with the SAME layout, so a step is a single `cvk_adamw_step` launch over 34.5 M
elements (7 x 138 MB of HBM traffic) instead of ~10 multi-tensor launches over 92 tensors.  It is a
`torch.optim.Optimizer`: `param_groups[0]["lr"]` / `["betas"]` are read every step, so `OneCycleLR` (train.py:103-104)
drives it unchanged.  One parameter group (the reference uses one)."""
import torch

from . import _lib, engine
from ._lib import check


def _block_params(net):
    from .modules import _Block
    out = []
    for m in net.modules():
        if isinstance(m, _Block):
            out.extend(m.block_params())
    return out


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, net, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        params = _block_params(net)                       # execution order == the executor's flat parameter list
        if len(params) != len(list(net.parameters())):
            raise ValueError("FlatAdamW needs a network made only of conv+BN blocks (UNet / SegNet)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._plist = params
        self._offs, total = engine.layout_grads(params)
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdamW: move the network to the GPU first")
        self._flat = torch.zeros(total, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p, o in zip(params, self._offs):
                n = p.numel()
                seg = self._flat[o:o + n]
                if p.dim() == 4:                          # storage [Cout][3][3][Cin] == channels_last OIHW
                    co, ci, kh, kw = p.shape
                    view = seg.view(co, kh, kw, ci).permute(0, 3, 1, 2)
                else:
                    view = seg.view(p.shape)
                view.copy_(p)
                p.data = view                             # the module now owns a view of the flat buffer
        self._m = torch.zeros_like(self._flat)
        self._v = torch.zeros_like(self._flat)
        self._gbuf = None
        self._step = 0

    def _flat_grad(self):
        """The executor's flat gradient buffer when every .grad is the expected view of it, else a gathered copy."""
        p0 = self._plist[0]
        if p0.grad is None:
            raise RuntimeError("FlatAdamW.step(): gradients missing")
        base = p0.grad.data_ptr() - 4 * self._offs[0]
        ok = all(p.grad is not None and p.grad.data_ptr() == base + 4 * o for p, o in zip(self._plist, self._offs))
        if ok:
            st = p0.grad.untyped_storage()
            start = (base - st.data_ptr()) // 4
            if base >= st.data_ptr() and (start + self._flat.numel()) * 4 <= st.nbytes():
                return torch.empty(0, device=p0.device, dtype=torch.float32).set_(st, start, (self._flat.numel(),))
        if self._gbuf is None:
            self._gbuf = torch.zeros_like(self._flat)
        for p, o in zip(self._plist, self._offs):
            g = p.grad.permute(0, 2, 3, 1) if p.dim() == 4 else p.grad
            self._gbuf[o:o + p.numel()].view(g.shape).copy_(g)
        return self._gbuf

    def _check_homes(self):
        """The module must still read its weights from the flat buffer: net.to()/.cuda()/.float() after construction
        re-homes p.data and step() would then update memory nobody reads."""
        base = self._flat.data_ptr()
        for p, o in zip(self._plist, self._offs):
            if p.data_ptr() != base + 4 * o:
                raise RuntimeError("FlatAdamW: a parameter no longer lives in the optimizer's flat buffer (the network was "
                                   "moved or cast after the optimizer was built); construct FlatAdamW after net.to(device)")

    # ---- optimizer state: exp_avg / exp_avg_sq / step travel with state_dict() like torch.optim.AdamW's do ----------
    def state_dict(self):
        sd = super().state_dict()
        sd["flat_adamw"] = {"step": self._step, "exp_avg": self._m.clone(), "exp_avg_sq": self._v.clone(),
                            "offsets": list(self._offs)}
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        extra = state_dict.pop("flat_adamw", None)
        super().load_state_dict(state_dict)
        if extra is not None:
            if list(extra["offsets"]) != list(self._offs) or extra["exp_avg"].numel() != self._m.numel():
                raise ValueError("FlatAdamW.load_state_dict: the saved state belongs to a different network layout")
            self._step = int(extra["step"])
            self._m.copy_(extra["exp_avg"]); self._v.copy_(extra["exp_avg_sq"])

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        self._check_homes()
        self._step += 1
        grad = self._flat_grad()
        lib = _lib.load()
        check(lib.cvk_adamw_step(self._flat.data_ptr(), grad.data_ptr(), self._m.data_ptr(), self._v.data_ptr(), self._flat.numel(),
                                 float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
                                 self._step, torch.cuda.current_stream(self._flat.device).cuda_stream), "cvk_adamw_step")
        engine._bump_epoch()          # the kernel wrote the parameters through raw pointers: derived weight tensors are stale
        return loss
