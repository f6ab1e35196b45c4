"""FusedAdam -- ``torch.optim.Adam``-compatible optimizer over the param groups of
MNESLAM.create_optimizer (mneslam_mp.py:459-469) whose ``step()`` is one HIP launch
(mne_adam_step: single pass over p, g, m, v for every tensor of every group).

Semantics are torch.optim.Adam's (amsgrad=False, maximize=False): L2 weight decay folded into the
gradient, bias correction by the per-parameter step count, dense update of every element (cells
with zero gradient still move while their first moment is non-zero -- required for parity,
SURVEY.md section 7).
"""
import ctypes as C

import torch

from . import _lib


def _dense_like(p, t):
    """``t`` as an fp32 tensor laid out exactly like ``p`` (same strides), copying only if needed."""
    if t.stride() == p.stride() and t.dtype == torch.float32:
        return t
    out = torch.empty_like(p, dtype=torch.float32)            # preserve_format
    out.copy_(t)
    return out


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)

    def _state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = 0
            # preserve_format: same physical layout as p; the moments are fp32 also for half-precision planes
            st["exp_avg"] = torch.zeros_like(p, dtype=torch.float32)
            st["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.float32)
        return st

    def segments(self, zero_grad_buffers=None, advance=True):
        """ctypes segment array for every parameter that has a gradient (advances the step counts unless
        ``advance`` is False: a recorded iteration reads its step from the device clock, ``step`` is then the base)."""
        segs = []
        keep = []
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None and getattr(p, "grad32", None) is not None:
                    p.grad32 = None              # .grad was dropped by someone else's zero_grad: the fp32 sum beside it is stale
                g = p.grad if zero_grad_buffers is None else zero_grad_buffers.get(p)   # explicit map: only those params
                if g is None:
                    continue
                if zero_grad_buffers is None and p.dtype == torch.float16:
                    # half-precision plane: autograd's .grad has the parameter's dtype, and a mean-reduced plane gradient
                    # sits below fp16's range (flushes to zero below ~3e-8; Adam is scale-free, so those would be lost
                    # updates).  The render node leaves the fp32 sum beside it (hip_path.RenderFunction.backward): that is
                    # what the update consumes.  Without it the step would silently run on the flushed fp16 gradient: refused.
                    if getattr(p, "grad32", None) is None:
                        raise RuntimeError("a half-precision plane has .grad but no fp32 gradient sum (grad32): its gradient did not "
                                           "come from this package's render node")
                    g = p.grad32 if p.grad32.device == p.device else p.grad32.to(p.device)
                if p.dtype not in (torch.float32, torch.float16):
                    raise TypeError("FusedAdam supports float32 parameters (and float16 planes: fp32 gradient and moments)")
                if not (p.is_contiguous() or p.is_contiguous(memory_format=torch.channels_last)):
                    raise ValueError("FusedAdam needs dense parameters")
                st = self._state(p)
                if advance:
                    st["step"] += 1
                g = _dense_like(p, g)
                keep.append(g)
                s = _lib.AdamSeg()
                s.p, s.g = p.data_ptr(), g.data_ptr()
                s.m, s.v = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                s.n = p.numel()
                s.lr, s.beta1, s.beta2 = float(group["lr"]), float(b1), float(b2)
                s.eps, s.weight_decay = float(group["eps"]), float(group["weight_decay"])
                s.step = st["step"] if advance else st["step"] + 1
                s.p_f16 = 1 if p.dtype == torch.float16 else 0
                segs.append((s, p))
        return segs, keep

    @torch.no_grad()
    def step(self, closure=None, zero_grad=False, grad_buffers=None, clock=None):
        """``clock`` (mneslam_amd._lib.Clock, graph recording only): the kernel adds the device step offset to the
        parameters' CURRENT step + 1 and the python step counts are left alone (the caller advances them per replay)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        segs, keep = self.segments(grad_buffers, advance=clock is None)
        if not segs:
            return loss
        lib = _lib.load()
        for i in range(0, len(segs), 32):
            chunk = segs[i:i + 32]
            arr = (_lib.AdamSeg * len(chunk))(*[s for s, _ in chunk])
            _lib.check(lib.mne_adam_step(arr, len(chunk), 1 if zero_grad else 0,
                                         C.byref(clock) if clock is not None else None, _lib.stream_for(chunk[0][1])),
                       "mne_adam_step")
        if zero_grad and grad_buffers is None:
            # the kernel zeroed the buffers it was given: for a half-precision plane that is the fp32 side sum (or a temporary
            # fp32 copy of .grad), not autograd's fp16 .grad itself
            for _, p in segs:
                if p.dtype == torch.float16 and p.grad is not None:
                    p.grad.zero_()
        return loss

    def zero_grad(self, set_to_none=True):
        for group in self.param_groups:
            for p in group["params"]:
                if getattr(p, "grad32", None) is not None:
                    p.grad32 = None                      # the fp32 gradient sum of a half-precision plane (see ``segments``)
        return super().zero_grad(set_to_none=set_to_none)
