"""The reference's training utilities on the MI355X path, same names and signatures:

  get_optimizer(model, opt_config)            training/sched_optim.py:5-27   -> HipAdam, a torch.optim.Optimizer whose step is the
                                                                               fused clip + Adam (+ EMA) HIP kernel
  AMPContext(enabled)                          utils/amp_utils.py:5-41
  EMA(model, decay)                            utils/io_ckpt.py:9-54
  save_checkpoint / load_checkpoint            utils/io_ckpt.py:56-118        -> byte-compatible torch.save layout

Checkpoint compatibility: HipAdam's state_dict is torch.optim.Adam's ({'state': {i: {'step', 'exp_avg',
'exp_avg_sq'}}, 'param_groups': [...]}), so a checkpoint written here loads into the reference (torch.optim.Adam) and vice versa.
"""
from __future__ import annotations

import contextlib
from pathlib import Path
from typing import Dict, Optional

import torch

from . import autograd as AG
from ._lib import F32
from .cut import AMPContext as _AMPBase
from .runtime import ADAM_CHUNK


class HipAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (single-tensor formula of torch/optim/adam.py) executed by gan_adam_step: global-norm clipping,
    the Adam update of every tensor and (optionally) an EMA shadow update in three launches regardless of the tensor count.
    Parameters whose `.grad` is None are skipped exactly as torch does (no step increment)."""

    def __init__(self, params, lr=2e-4, betas=(0.5, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False):
        if weight_decay != 0.0 or amsgrad:
            raise NotImplementedError("the reference configs use weight_decay 0 and no amsgrad")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False))
        self._ema: Dict[int, torch.Tensor] = {}      # id(param) -> shadow tensor (attached by EMA)
        self._ema_decay = 0.0
        self._plans = {}
        self.last_grad_norm: Optional[torch.Tensor] = None

    def attach_ema(self, shadow_by_param: Dict[int, torch.Tensor], decay: float):
        self._ema, self._ema_decay, self._plans = dict(shadow_by_param), float(decay), {}

    def _state(self, p):
        st = self.state[p]
        if "exp_avg" not in st:
            st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(p.data), torch.zeros_like(p.data)
            st["step"] = torch.zeros(1, dtype=torch.int32, device=p.device)
            st["grad_buf"] = torch.zeros_like(p.data)
        elif st["step"].dtype != torch.int32 or st["step"].device != p.device:      # loaded from a torch.optim.Adam checkpoint
            st["step"] = st["step"].reshape(-1)[:1].to(device=p.device, dtype=torch.int32).contiguous()
            st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"].to(p.device).contiguous(), st["exp_avg_sq"].to(p.device).contiguous()
        if "grad_buf" not in st:
            st["grad_buf"] = torch.zeros_like(p.data)
        return st

    def _plan(self, gi, group, live):
        """Prebuilt launch for one param group and one pattern of present gradients (ops hold raw pointers)."""
        key = (gi, live, group["lr"], group["betas"], group["eps"], torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else 0)
        pl = self._plans.get(key)
        if pl is None:
            ps = group["params"]
            dev = ps[0].device
            ctx = AG._new_ctx(dev, F32)
            ents, ct, co = [], [], []
            for i, p in enumerate(ps):
                st = self._state(p)
                ents.append({"p": p.data.view(-1), "g": st["grad_buf"].view(-1) if live[i] else None, "m": st["exp_avg"].view(-1),
                             "v": st["exp_avg_sq"].view(-1), "ema": self._ema[id(p)].view(-1) if id(p) in self._ema else None, "step": st["step"]})
                for off in range(0, p.numel(), ADAM_CHUNK):
                    ct.append(i); co.append(off)
            pl = {"ctx": ctx, "norm": torch.zeros(4, dtype=torch.float32, device=dev),
                  "ct": torch.tensor(ct, dtype=torch.int32, device=dev), "co": torch.tensor(co, dtype=torch.int64, device=dev),
                  "ws": torch.zeros(len(ct) + 16, dtype=torch.float32, device=dev), "ops": {}}
            pl["table"] = ctx.ops.make_adam_table(ents)
            pl["n"], pl["nchunks"] = len(ps), len(ct)
            self._plans[key] = pl
        return pl

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm: Optional[float] = None):
        loss = closure() if closure is not None else None
        for gi, group in enumerate(self.param_groups):
            ps = group["params"]
            for p in ps:
                assert p.dtype == torch.float32 and p.is_contiguous()
            live = tuple(p.grad is not None for p in ps)
            if not any(live):
                continue
            pl = self._plan(gi, group, live)
            for p, on in zip(ps, live):
                if on:
                    self.state[p]["grad_buf"].copy_(p.grad)
            mn = float(max_grad_norm) if max_grad_norm is not None else 0.0
            op = pl["ops"].get(mn)
            if op is None:
                b1, b2 = group["betas"]
                op = pl["ops"][mn] = pl["ctx"].ops.adam_step(pl["table"], pl["n"], pl["ct"], pl["co"], pl["nchunks"], group["lr"], b1, b2, group["eps"],
                                                              mn, 1.0, self._ema_decay, pl["norm"], pl["ws"])
            op()
            self.last_grad_norm = pl["norm"]
            for p, on in zip(ps, live):            # EMA.update covers every parameter, also one the kernel skipped for want of a gradient
                if not on and id(p) in self._ema:
                    self._ema[id(p)].mul_(self._ema_decay).add_(p.data, alpha=1.0 - self._ema_decay)
        AG.notify_weights_changed()      # parameters were written through raw pointers: module bridges must repack
        return loss

    # torch.optim.Adam's checkpoint layout: 'step' as a float tensor; the private gradient staging buffer is not state
    def state_dict(self):
        sd = super().state_dict()
        out_state = {}
        for k, st in sd["state"].items():
            out_state[k] = {"step": st["step"].detach().float().reshape(()).cpu(), "exp_avg": st["exp_avg"], "exp_avg_sq": st["exp_avg_sq"]}
        return {"state": out_state, "param_groups": sd["param_groups"]}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plans = {}
        for group in self.param_groups:
            group["betas"] = tuple(group["betas"])
            for p in group["params"]:
                if p in self.state and "exp_avg" in self.state[p]:
                    self._state(p)


_FUSED_PLANS = {}


def fused_adam_launch(params, grads, m, v, ema, steps, lr, b1, b2, eps, max_norm, grad_scale, ema_decay, inv_scale=None, skip_nonfinite=False):
    """Body of torch.ops.mi355x_gan.fused_clip_adam_ema_: clip_grad_norm_(max_norm; 0 = off) on grads * grad_scale, one Adam
    step of every tensor (per-tensor step counters `steps`, int32 [n], incremented on the device) and, if `ema` is non-empty,
    shadow <- decay * shadow + (1 - decay) * p, in three launches regardless of the tensor count.  inv_scale (device float) and
    skip_nonfinite: GradScaler's unscale_ / step.  Returns (total gradient norm, found_inf).
    The prebuilt launch is keyed on the STATE tensors only (params, m, v, ema, steps): gradients are fresh tensors every step under
    zero_grad(set_to_none=True), so they are copied into staging buffers the plan owns (one multi-tensor copy) -- a plan per gradient
    address would grow without bound and pin every old gradient."""
    n = len(params)
    assert n and len(grads) == n and len(m) == n and len(v) == n and len(ema) in (0, n) and steps.dtype == torch.int32 and steps.numel() == n
    key = tuple((t.data_ptr(), t.numel()) for grp in (params, m, v, ema) for t in grp) + (steps.data_ptr(), inv_scale.data_ptr() if inv_scale is not None else 0,
                                                                                         torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else 0)
    pl = _FUSED_PLANS.get(key)
    if pl is None:
        dev = params[0].device
        ctx = AG._new_ctx(dev, F32)
        ents, ct, co, gbuf = [], [], [], []
        for i in range(n):
            for t in (params[i], m[i], v[i]) + ((ema[i],) if ema else ()):
                assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == params[i].numel()
            gbuf.append(torch.zeros(params[i].numel(), dtype=torch.float32, device=dev))
            ents.append({"p": params[i].view(-1), "g": gbuf[i], "m": m[i].view(-1), "v": v[i].view(-1),
                         "ema": ema[i].view(-1) if ema else None, "step": steps[i:i + 1]})
            for off in range(0, params[i].numel(), ADAM_CHUNK):
                ct.append(i); co.append(off)
        if len(_FUSED_PLANS) >= 8:      # a handful of optimisers per process; anything beyond that is a caller churning its state tensors
            _FUSED_PLANS.pop(next(iter(_FUSED_PLANS)))
        pl = _FUSED_PLANS[key] = {"ctx": ctx, "norm": torch.zeros(4, dtype=torch.float32, device=dev), "table": ctx.ops.make_adam_table(ents),
                                  "ct": torch.tensor(ct, dtype=torch.int32, device=dev), "co": torch.tensor(co, dtype=torch.int64, device=dev),
                                  "ws": torch.zeros(len(ct) + 16, dtype=torch.float32, device=dev), "gbuf": gbuf,
                                  "keep": (params, m, v, ema, steps, inv_scale), "ops": {}}
    for g, p in zip(grads, params):
        assert g.dtype == torch.float32 and g.numel() == p.numel()
    torch._foreach_copy_(pl["gbuf"], [g.reshape(-1) for g in grads])
    hk = (lr, b1, b2, eps, max_norm, grad_scale, ema_decay if ema else 0.0, bool(skip_nonfinite))
    op = pl["ops"].get(hk)
    if op is None:
        op = pl["ops"][hk] = pl["ctx"].ops.adam_step(pl["table"], n, pl["ct"], pl["co"], len(pl["ct"]), lr, b1, b2, eps, max_norm, grad_scale,
                                                      ema_decay if ema else 0.0, pl["norm"], pl["ws"], inv_scale=inv_scale, skip_nonfinite=skip_nonfinite)
    op()
    AG.notify_weights_changed()
    return pl["norm"][:1].clone(), pl["norm"][2:3].clone()


def get_optimizer(model, opt_config: dict):
    """sched_optim.py:5-27 (Adam only on this path; the reference's default type is also 'adam')."""
    kind = opt_config.get("type", "adam").lower()
    if kind != "adam":
        raise NotImplementedError(f"optimizer type {kind!r}: the MI355X path implements the reference's configured Adam")
    return HipAdam(model.parameters(), lr=opt_config.get("lr", 2e-4), betas=tuple(opt_config.get("betas", [0.5, 0.999])),
                   weight_decay=opt_config.get("weight_decay", 0.0))


class _UnitScaler:
    """GradScaler stand-in: bf16 has fp32's exponent range, so nothing is scaled (scale == 1, no skipped steps)."""

    def scale(self, x):
        return x

    def get_scale(self):
        return 1.0

    def unscale_(self, optimizer):
        pass

    def step(self, optimizer):
        return optimizer.step()

    def update(self):
        pass

    def state_dict(self):
        return {}

    def load_state_dict(self, sd):
        pass


class AMPContext(_AMPBase):
    """utils/amp_utils.py:5-41.  `enabled` selects bf16 operands inside the HIP modules (set their `compute_dtype`); autocast is a
    no-op because no ATen op is on the path, and the scaler is the identity."""

    def __init__(self, enabled: bool = True):
        super().__init__(enabled)
        self.scaler = _UnitScaler()

    def autocast(self):
        return contextlib.nullcontext()

    def scale_and_step(self, loss, optimizer, scaler=None):
        loss.backward()
        optimizer.step()

    def scale_backward(self, loss):
        loss.backward()

    def step_optimizer(self, optimizer, max_grad_norm=None):
        if not isinstance(optimizer, HipAdam):      # no ATen fallback on this path: get_optimizer() hands out the fused optimiser
            raise TypeError(f"AMPContext.step_optimizer drives HipAdam (training.get_optimizer); got {type(optimizer).__name__}, whose "
                            "clip_grad_norm_ + step would run on ATen")
        optimizer.step(max_grad_norm=max_grad_norm)            # clipping is fused into the optimiser launch


class EMA:
    """utils/io_ckpt.py:9-54.  With `optimizer=` (a HipAdam over the same parameters) the shadow update rides in the optimiser's
    fused launch and update() only counts; stand-alone it is two multi-tensor launches."""

    def __init__(self, model: torch.nn.Module, decay: float = 0.999, optimizer: Optional[HipAdam] = None):
        self.model, self.decay = model, decay
        self.shadow, self.backup = {}, {}
        for name, p in model.named_parameters():
            if p.requires_grad:
                self.shadow[name] = p.data.clone()
        self._fused = optimizer is not None
        if self._fused:
            optimizer.attach_ema({id(p): self.shadow[n] for n, p in model.named_parameters() if p.requires_grad}, decay)

    @torch.no_grad()
    def update(self):
        if self._fused:
            return
        names = [n for n, p in self.model.named_parameters() if p.requires_grad]
        sh = [self.shadow[n] for n in names]
        torch._foreach_mul_(sh, self.decay)
        torch._foreach_add_(sh, [p.data for n, p in self.model.named_parameters() if p.requires_grad], alpha=1.0 - self.decay)

    @torch.no_grad()
    def apply_shadow(self):
        for name, p in self.model.named_parameters():
            if p.requires_grad:
                self.backup[name] = p.data.clone()
                p.data.copy_(self.shadow[name])

    @torch.no_grad()
    def restore(self):
        for name, p in self.model.named_parameters():
            if p.requires_grad:
                p.data.copy_(self.backup[name])
        self.backup = {}

    def state_dict(self):
        return {"decay": self.decay, "shadow": self.shadow}

    def load_state_dict(self, state_dict):
        self.decay = state_dict["decay"]
        for k, v in state_dict["shadow"].items():      # in place: a fused optimiser holds pointers to the shadow tensors
            self.shadow[k].copy_(v)


def save_checkpoint(path: str, step: int, generator, discriminator, opt_G, opt_D, ema_G: Optional[EMA] = None, scaler=None,
                    metrics: Optional[Dict] = None, config: Optional[Dict] = None):
    """utils/io_ckpt.py:56-87, same keys."""
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    ckpt = {"step": step, "generator": generator.state_dict(), "discriminator": discriminator.state_dict(), "opt_G": opt_G.state_dict(),
            "opt_D": opt_D.state_dict(), "metrics": metrics or {}, "config": config or {}}
    if ema_G is not None:
        ckpt["ema_G"] = ema_G.state_dict()
    if scaler is not None:
        ckpt["scaler"] = scaler.state_dict()
    torch.save(ckpt, path)


def load_checkpoint(path: str, generator, discriminator, opt_G=None, opt_D=None, ema_G: Optional[EMA] = None, scaler=None, device="cuda") -> Dict:
    """utils/io_ckpt.py:90-118.  Loaded with weights_only=True: a checkpoint is tensors and plain containers, nothing is unpickled."""
    ckpt = torch.load(path, map_location=device, weights_only=True)
    generator.load_state_dict(ckpt["generator"])
    discriminator.load_state_dict(ckpt["discriminator"])
    if opt_G is not None and "opt_G" in ckpt:
        opt_G.load_state_dict(ckpt["opt_G"])
    if opt_D is not None and "opt_D" in ckpt:
        opt_D.load_state_dict(ckpt["opt_D"])
    if ema_G is not None and "ema_G" in ckpt:
        ema_G.load_state_dict(ckpt["ema_G"])
    if scaler is not None and "scaler" in ckpt:
        scaler.load_state_dict(ckpt["scaler"])
    return ckpt
