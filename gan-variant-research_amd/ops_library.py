"""Op-level PyTorch custom ops: ``torch.ops.mi355x_gan.*`` (SURVEY.md §8b "native surface", BASELINE.json north_star).

The fast path of this package is module- and step-granular (autograd.py, cut.py): activations stay in halo-NHWC between
kernels.  This file is the *compatibility* path underneath it: every ATen op the reference's model files dispatch
(GAN_Variant1/models/generator_resnet_attn.py:24-66,104-163; discriminator_patchgan.py:26-54; Basic_GAN/src/models.py:10-103)
is registered in the PyTorch dispatcher as a functional op with a Tensor-in / Tensor-out schema and an autograd formula,

    mi355x_gan::conv2d_fwd / conv2d_dgrad / conv2d_wgrad
    mi355x_gan::conv_transpose2d_fwd / conv_transpose2d_dgrad / conv_transpose2d_wgrad
    mi355x_gan::instance_norm_fwd / instance_norm_bwd
    mi355x_gan::reflection_pad2d / reflection_pad2d_bwd
    mi355x_gan::fused_clip_adam_ema_

so that a network assembled layer by layer from the `Conv2d` / `ConvTranspose2d` / `InstanceNorm2d` / `ReflectionPad2d` modules
below -- drop-in for their torch.nn namesakes, same parameters and state_dict keys -- runs on the HIP kernels unchanged.
Each call converts its NCHW fp32 operands to halo-NHWC, runs the same C-ABI launches the fused path plans (convplan.ConvLayer,
gan_in_*), and converts back: correct and differentiable, not fast (two layout passes per op).  Tensors are PyTorch-allocated;
launches go to the stream that is current when an op's plan is first built; errors surface as GanError (a RuntimeError).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import autograd as AG
from ._lib import ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, BF16, F32, HALO_NONE, HALO_REFLECT, HALO_REPLICATE, HALO_ZERO
from .convplan import ConvLayer
from .runtime import Program, View, cpad

PAD_ZERO, PAD_REFLECT = 0, 1
_COMPUTE_DTYPE = [F32]
_PLANS = {}
IN_EPS = 1e-5


def set_compute_dtype(dtype: int):
    """F32 (exact fp32 MFMA, the parity mode; default) or BF16 (bf16 operands, fp32 accumulation) for the ops of this library."""
    assert dtype in (F32, BF16)
    _COMPUTE_DTYPE[0] = dtype


def _stream_key() -> int:
    """A plan's launches are bound to the HIP stream that was current when it was built (raw handle in the prebuilt calls) while its
    staging copies run on whatever stream is current at the call: the current stream is therefore part of every plan key, so a caller
    working on another stream (a prefetch stream, `with torch.cuda.stream(s)`) gets a plan of its own instead of an unordered pair."""
    return torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else 0


def _plan(key, make):
    key = key + (_stream_key(),)
    p = _PLANS.get(key)
    if p is None:
        p = _PLANS[key] = make()
    return p


def _dev_key(t: torch.Tensor):
    return (t.device.type, t.device.index)


class _ConvPlan:
    """Staging tensors, views and prebuilt launches of one convolution geometry on one device (ops hold raw pointers)."""

    def __init__(self, device, dtype, x_shape, w_shape, has_bias, stride, pad, pad_mode, act, transposed):
        self.ctx = AG._new_ctx(device, dtype)
        ops, ctx = self.ctx.ops, self.ctx
        B, Cin, H, W = x_shape
        k = w_shape[2]
        self.transposed, self.reflect = transposed, pad_mode == PAD_REFLECT
        if transposed:
            assert (k, stride, pad) == (3, 2, 1) and w_shape[0] == Cin, "ConvTranspose2d is supported as k3 s2 p1 output_padding 1"
            Cout, Ho, Wo = w_shape[1], 2 * H, 2 * W
        else:
            assert w_shape[1] == Cin and stride in (1, 2)
            Cout, Ho, Wo = w_shape[0], (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        if self.reflect:
            assert not transposed and stride == 1 and pad < min(H, W), "reflect padding: stride-1 convolutions"
        f32 = lambda *s: torch.zeros(*s, dtype=torch.float32, device=device)
        self.w, self.b = f32(w_shape), (f32(Cout) if has_bias else None)
        self.gw, self.gb = f32(w_shape), (f32(Cout) if has_bias else None)
        self.layer = ConvLayer(ctx, self.w, self.b, self.gw, self.gb, k, stride, pad, transposed)
        self.x_in, self.y_out = f32(x_shape), f32(B, Cout, Ho, Wo)
        self.dy_in, self.dx_out = f32(B, Cout, Ho, Wo), f32(x_shape)
        xh = 1 if transposed else pad
        self.xv = ctx.view(B, H, W, cpad(Cin), xh)
        self.yv = ctx.view(B, Ho, Wo, cpad(Cout), 0)
        dyh = 1 if (transposed or stride == 2) else k - 1
        self.dyv = ctx.view(B, Ho, Wo, cpad(Cout), dyh)
        mode = HALO_REFLECT if self.reflect else HALO_ZERO
        self.fwd = Program("conv2d_fwd")
        self.fwd.add(ops.nchw_to_view(self.x_in, Cin, self.xv, mode))
        self.fwd.add(self.layer.fwd(self.xv, self.yv, act))
        self.fwd.add(ops.view_to_nchw(self.yv, Cout, self.y_out))
        # input gradient (act must be none: the activation's backward is the caller's, see _conv_backward)
        self.dgrad = Program("conv2d_dgrad")
        self.dgrad.add(ops.nchw_to_view(self.dy_in, Cout, self.dyv, HALO_ZERO))
        if self.reflect:      # gradient on the padded domain, folded back over the reflect halo
            dxp = ctx.view(B, H, W, cpad(Cin), pad)
            dxv = ctx.view(B, H, W, cpad(Cin), 0)
            self.dgrad.add(self.layer.dgrad(self.dyv, dxp, padded_domain=True))
            self.dgrad.add(ops.fold_add(None, dxp, True, dxv))
        else:
            dxv = ctx.view(B, H, W, cpad(Cin), 0)
            self.dgrad.add(self.layer.dgrad(self.dyv, dxv))
        self.dgrad.add(ops.view_to_nchw(dxv, Cin, self.dx_out))
        self.wgrad = Program("conv2d_wgrad")
        self.wgrad.add(ops.nchw_to_view(self.x_in, Cin, self.xv, mode))
        self.wgrad.add(ops.nchw_to_view(self.dy_in, Cout, self.dyv, HALO_ZERO))
        self.wgrad.add(self.layer.wgrad(self.xv, self.dyv, accumulate=False, bias_too=has_bias))
        self.repack = Program("repack")            # built last: refreshes exactly the operand copies the programs above use
        rp = self.layer.repack_ops()
        if rp:
            self.repack.add(ops.pack_weight_batch([o.pack_args for o in rp]))

    def load_weight(self, w, b):
        self.w.copy_(w)
        if self.b is not None:
            self.b.copy_(b)
        self.repack.run()


def _conv_plan(x_shape, w, b, stride, pad, pad_mode, act, transposed) -> _ConvPlan:
    key = ("conv", _dev_key(w), _COMPUTE_DTYPE[0], tuple(x_shape), tuple(w.shape), b is not None, stride, pad, pad_mode, act, transposed)
    return _plan(key, lambda: _ConvPlan(w.device, _COMPUTE_DTYPE[0], tuple(x_shape), tuple(w.shape), b is not None, stride, pad, pad_mode, act, transposed))


def _f(t: torch.Tensor) -> torch.Tensor:
    return t.detach().float().contiguous()


# ------------------------------------------------------------------------------------------------ op definitions
_LIB = torch.library.Library("mi355x_gan", "DEF")
_LIB.define("conv2d_fwd(Tensor x, Tensor w, Tensor? b, int stride, int pad, int pad_mode, int act) -> Tensor")
_LIB.define("conv2d_dgrad(Tensor dy, Tensor w, int[] x_shape, int stride, int pad, int pad_mode) -> Tensor")
_LIB.define("conv2d_wgrad(Tensor x, Tensor dy, int[] w_shape, bool has_bias, int stride, int pad, int pad_mode) -> (Tensor, Tensor)")
_LIB.define("conv_transpose2d_fwd(Tensor x, Tensor w, Tensor? b, int act) -> Tensor")
_LIB.define("conv_transpose2d_dgrad(Tensor dy, Tensor w, int[] x_shape) -> Tensor")
_LIB.define("conv_transpose2d_wgrad(Tensor x, Tensor dy, int[] w_shape, bool has_bias) -> (Tensor, Tensor)")
_LIB.define("instance_norm_fwd(Tensor x, float eps, int act, Tensor? residual) -> (Tensor, Tensor)")
_LIB.define("instance_norm_bwd(Tensor x, Tensor stats, Tensor gy, int act) -> Tensor")
_LIB.define("reflection_pad2d(Tensor x, int pad) -> Tensor")
_LIB.define("reflection_pad2d_bwd(Tensor gy, int pad) -> Tensor")
_LIB.define("replication_pad2d(Tensor x, int pad) -> Tensor")
_LIB.define("replication_pad2d_bwd(Tensor gy, int pad) -> Tensor")
_LIB.define("act_bwd(Tensor y, Tensor gy, int act) -> Tensor")
_LIB.define("fused_clip_adam_ema_(Tensor(a!)[] params, Tensor[] grads, Tensor(b!)[] m, Tensor(c!)[] v, Tensor(d!)[] ema, Tensor(e!) steps, float lr, float b1, "
            "float b2, float eps, float max_norm, float grad_scale, float ema_decay, Tensor? inv_scale=None, bool skip_nonfinite=False) -> (Tensor, Tensor)")
_LIB.define("patchnce_fwd(Tensor src_feat, Tensor tgt_feat, Tensor ids, float temperature) -> (Tensor, Tensor)")
_LIB.define("patchnce_bwd(Tensor saved, Tensor grad_loss) -> Tensor")
_LIB.define("diffaugment_fwd(Tensor x, Tensor params) -> Tensor")
_LIB.define("diffaugment_bwd(Tensor gy, Tensor params) -> Tensor")
_LIB.define("allreduce_bucket_(Tensor(a!) flat, str group) -> Tensor(a!)")


def _conv_fwd(x, w, b, stride, pad, pad_mode, act, transposed=False):
    p = _conv_plan(x.shape, w, b, stride, pad, pad_mode, act, transposed)
    p.load_weight(_f(w), None if b is None else _f(b))
    p.x_in.copy_(x)
    p.fwd.run()
    return p.y_out.clone()


def _conv_dgrad(dy, w, x_shape, stride, pad, pad_mode, transposed=False):
    p = _conv_plan(x_shape, w, None, stride, pad, pad_mode, ACT_NONE, transposed)
    p.load_weight(_f(w), None)
    p.dy_in.copy_(dy)
    p.dgrad.run()
    return p.dx_out.clone()


def _conv_wgrad(x, dy, w_shape, has_bias, stride, pad, pad_mode, transposed=False):
    key = ("convw", _dev_key(x), _COMPUTE_DTYPE[0], tuple(x.shape), tuple(w_shape), has_bias, stride, pad, pad_mode, ACT_NONE, transposed)
    p = _plan(key, lambda: _ConvPlan(x.device, _COMPUTE_DTYPE[0], tuple(x.shape), tuple(w_shape), has_bias, stride, pad, pad_mode, ACT_NONE, transposed))
    p.x_in.copy_(x)
    p.dy_in.copy_(dy)
    p.wgrad.run()
    return p.gw.clone(), (p.gb.clone() if has_bias else torch.zeros(0, device=x.device))


_IMPL = "CompositeExplicitAutograd"       # one implementation for every device: the C ABI below it is what is device-specific
_LIB.impl("conv2d_fwd", lambda x, w, b, stride, pad, pad_mode, act: _conv_fwd(_f(x), w, b, stride, pad, pad_mode, act), _IMPL)
_LIB.impl("conv2d_dgrad", lambda dy, w, x_shape, stride, pad, pad_mode: _conv_dgrad(_f(dy), w, tuple(x_shape), stride, pad, pad_mode), _IMPL)
_LIB.impl("conv2d_wgrad", lambda x, dy, w_shape, has_bias, stride, pad, pad_mode: _conv_wgrad(_f(x), _f(dy), tuple(w_shape), has_bias, stride, pad, pad_mode), _IMPL)
_LIB.impl("conv_transpose2d_fwd", lambda x, w, b, act: _conv_fwd(_f(x), w, b, 2, 1, PAD_ZERO, act, True), _IMPL)
_LIB.impl("conv_transpose2d_dgrad", lambda dy, w, x_shape: _conv_dgrad(_f(dy), w, tuple(x_shape), 2, 1, PAD_ZERO, True), _IMPL)
_LIB.impl("conv_transpose2d_wgrad", lambda x, dy, w_shape, has_bias: _conv_wgrad(_f(x), _f(dy), tuple(w_shape), has_bias, 2, 1, PAD_ZERO, True), _IMPL)


# ---- InstanceNorm2d (non-affine, biased variance, no running stats: generator_resnet_attn.py:56, Basic_GAN/src/models.py:13)
class _NormPlan:
    def __init__(self, device, dtype, shape, act, has_res):
        self.ctx = AG._new_ctx(device, dtype)
        ops, ctx = self.ctx.ops, self.ctx
        B, C, H, W = shape
        f32 = lambda *s: torch.zeros(*s, dtype=torch.float32, device=device)
        self.x_in, self.r_in, self.g_in = f32(shape), (f32(shape) if has_res else None), f32(shape)
        self.y_out, self.dx_out, self.stats = f32(shape), f32(shape), f32(B * cpad(C) * 2)
        xv, yv, rv = ctx.view(B, H, W, cpad(C), 0), ctx.view(B, H, W, cpad(C), 0), (ctx.view(B, H, W, cpad(C), 0) if has_res else None)
        gv, dxv = ctx.view(B, H, W, cpad(C), 0), ctx.view(B, H, W, cpad(C), 0)
        ws = f32(B * 96 * cpad(C) * 2 + B * cpad(C) * 2 + 64)
        self.fwd = Program("instance_norm_fwd")
        self.fwd.add(ops.nchw_to_view(self.x_in, C, xv, HALO_NONE))
        if has_res:
            self.fwd.add(ops.nchw_to_view(self.r_in, C, rv, HALO_NONE))
        self.fwd.add(ops.in_partial(xv, ws))
        self.fwd.add(ops.in_apply_parts(xv, ws, ops.in_partial_count(xv), IN_EPS, self.stats, act, rv, yv, HALO_NONE))
        self.fwd.add(ops.view_to_nchw(yv, C, self.y_out))
        self.bwd = Program("instance_norm_bwd")
        self.bwd.add(ops.nchw_to_view(self.x_in, C, xv, HALO_NONE))
        self.bwd.add(ops.nchw_to_view(self.g_in, C, gv, HALO_NONE))
        self.bwd.add(ops.in_bwd(xv, self.stats, act, gv, False, None, dxv, ws))
        self.bwd.add(ops.view_to_nchw(dxv, C, self.dx_out))


def _norm_plan(x, act, has_res) -> _NormPlan:
    key = ("norm", _dev_key(x), _COMPUTE_DTYPE[0], tuple(x.shape), act, has_res)
    return _plan(key, lambda: _NormPlan(x.device, _COMPUTE_DTYPE[0], tuple(x.shape), act, has_res))


def _in_fwd(x, eps, act, residual):
    assert abs(eps - IN_EPS) < 1e-12, "InstanceNorm2d eps is 1e-5 in the reference (torch default)"
    p = _norm_plan(x, act, residual is not None)
    p.x_in.copy_(x)
    if residual is not None:
        p.r_in.copy_(residual)
    p.fwd.run()
    return p.y_out.clone(), p.stats.clone()


def _in_bwd(x, stats, gy, act):
    p = _norm_plan(x, act, False)
    p.x_in.copy_(x)
    p.g_in.copy_(gy)
    p.stats.copy_(stats)
    p.bwd.run()
    return p.dx_out.clone()


_LIB.impl("instance_norm_fwd", lambda x, eps, act, residual: _in_fwd(_f(x), eps, act, None if residual is None else _f(residual)), _IMPL)
_LIB.impl("instance_norm_bwd", lambda x, stats, gy, act: _in_bwd(_f(x), stats, _f(gy), act), _IMPL)


# ---- ReflectionPad2d and the stand-alone activation backward
class _PadPlan:
    def __init__(self, device, shape, pad, mode=HALO_REFLECT):
        self.ctx = AG._new_ctx(device, F32)
        ops, ctx = self.ctx.ops, self.ctx
        B, C, H, W = shape
        assert mode == HALO_REPLICATE or pad < min(H, W)
        f32 = lambda *s: torch.zeros(*s, dtype=torch.float32, device=device)
        self.x_in, self.y_out = f32(shape), f32(B, C, H + 2 * pad, W + 2 * pad)
        self.g_in, self.dx_out = f32(B, C, H + 2 * pad, W + 2 * pad), f32(shape)
        v = ctx.view(B, H, W, cpad(C), pad)
        whole = View(v.t, B, H + 2 * pad, W + 2 * pad, v.C, 0, v.dtype)       # the padded extent as a plain image (same storage)
        self.fwd = Program("reflection_pad2d")
        self.fwd.add(ops.nchw_to_view(self.x_in, C, v, mode))
        self.fwd.add(ops.view_to_nchw(whole, C, self.y_out))
        dxv = ctx.view(B, H, W, cpad(C), 0)
        self.bwd = Program("pad2d_bwd")
        self.bwd.add(ops.nchw_to_view(self.g_in, C, whole, HALO_NONE))
        # the streaming fold needs H, W >= 2*pad+2 (each pixel receives at most one reflection per axis); small maps take the general one
        self.bwd.add(ops.fold_add(None, v, True, dxv) if mode == HALO_REFLECT and min(H, W) >= 2 * pad + 2 else ops.pad_fold(v, mode, dxv))
        self.bwd.add(ops.view_to_nchw(dxv, C, self.dx_out))


def _pad_plan(shape, device, pad, mode=HALO_REFLECT) -> _PadPlan:
    return _plan(("pad", (device.type, device.index), tuple(shape), pad, mode), lambda: _PadPlan(device, tuple(shape), pad, mode))


def _pad_fwd(x, pad, mode=HALO_REFLECT):
    p = _pad_plan(x.shape, x.device, pad, mode)
    p.x_in.copy_(x)
    p.fwd.run()
    return p.y_out.clone()


def _pad_bwd(gy, pad, mode=HALO_REFLECT):
    B, C, Hp, Wp = gy.shape
    p = _pad_plan((B, C, Hp - 2 * pad, Wp - 2 * pad), gy.device, pad, mode)
    p.g_in.copy_(gy)
    p.bwd.run()
    return p.dx_out.clone()


class _ActPlan:
    def __init__(self, device, shape, act):
        self.ctx = AG._new_ctx(device, F32)
        ops, ctx = self.ctx.ops, self.ctx
        B, C, H, W = shape
        f32 = lambda *s: torch.zeros(*s, dtype=torch.float32, device=device)
        self.y_in, self.g_in, self.dx_out = f32(shape), f32(shape), f32(shape)
        yv, gv, dv = (ctx.view(B, H, W, cpad(C), 0) for _ in range(3))
        self.prog = Program("act_bwd")
        self.prog.add(ops.nchw_to_view(self.y_in, C, yv, HALO_NONE))
        self.prog.add(ops.nchw_to_view(self.g_in, C, gv, HALO_NONE))
        self.prog.add(ops.act_bwd(yv, act, gv, False, None, dv))
        self.prog.add(ops.view_to_nchw(dv, C, self.dx_out))


def _act_bwd(y, gy, act):
    p = _plan(("act", _dev_key(y), tuple(y.shape), act), lambda: _ActPlan(y.device, tuple(y.shape), act))
    p.y_in.copy_(y)
    p.g_in.copy_(gy)
    p.prog.run()
    return p.dx_out.clone()


_LIB.impl("reflection_pad2d", lambda x, pad: _pad_fwd(_f(x), pad), _IMPL)
_LIB.impl("reflection_pad2d_bwd", lambda gy, pad: _pad_bwd(_f(gy), pad), _IMPL)
_LIB.impl("replication_pad2d", lambda x, pad: _pad_fwd(_f(x), pad, HALO_REPLICATE), _IMPL)
_LIB.impl("replication_pad2d_bwd", lambda gy, pad: _pad_bwd(_f(gy), pad, HALO_REPLICATE), _IMPL)
_LIB.impl("act_bwd", lambda y, gy, act: _act_bwd(_f(y), _f(gy), act), _IMPL)


# ---- fused clip_grad_norm_ + Adam + EMA.update (amp_utils.py:29-41, sched_optim.py:5-27, io_ckpt.py:23-29)
def _fused_adam(params, grads, m, v, ema, steps, lr, b1, b2, eps, max_norm, grad_scale, ema_decay, inv_scale=None, skip_nonfinite=False):
    from .training import fused_adam_launch
    return fused_adam_launch(list(params), list(grads), list(m), list(v), list(ema), steps, lr, b1, b2, eps, max_norm, grad_scale, ema_decay,
                             inv_scale, skip_nonfinite)


_LIB.impl("fused_clip_adam_ema_", _fused_adam, _IMPL)


# ---- PatchNCE (patchnce_cut.py:42-110) and DiffAugment (diffaugment.py:94-106): value and input gradient come out of one pass each,
#      so the forward op hands the gradient for a unit upstream gradient to its backward op as `saved`
def _nce_fwd(src, tgt, ids, temperature):
    from . import losses as LS
    P = int(ids.numel())
    p = LS._plan(("nce", tuple(tgt.shape), tgt.device, P, float(temperature)), lambda: LS._NcePlan(tuple(tgt.shape), tgt.device, P, temperature))
    p.src.copy_(src); p.tgt.copy_(tgt); p.ids.copy_(ids)
    p.fwd.run()
    return p.loss.clone().reshape(()), p.g.clone()


def _aug_run(x, prm, backward):
    from . import losses as LS
    p = LS._plan(("aug", tuple(x.shape), x.device), lambda: LS._AugPlan(tuple(x.shape), x.device))
    p.prm.copy_(prm)
    if backward:
        p.gy.copy_(x)
        p.bwd.run()
        return p.gx.clone()
    p.x.copy_(x)
    p.fwd.run()
    return p.y.clone()


def _allreduce_bucket(flat, group):
    """The gradient all-reduce of one flat bucket (SUM over RCCL on a GPU, gloo on the CPU tests): `group` names a process group
    ('' = the default one); without an initialised group the single-process trainers call it as a no-op."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        pg = dist.distributed_c10d._resolve_process_group(group) if group else None
        dist.all_reduce(flat, group=pg)
    return flat


_LIB.impl("patchnce_fwd", lambda src, tgt, ids, temperature: _nce_fwd(_f(src), _f(tgt), ids.to(torch.int32), temperature), _IMPL)
_LIB.impl("patchnce_bwd", lambda saved, g: saved * g, _IMPL)
_LIB.impl("diffaugment_fwd", lambda x, prm: _aug_run(_f(x), prm, False), _IMPL)
_LIB.impl("diffaugment_bwd", lambda gy, prm: _aug_run(_f(gy), prm, True), _IMPL)
_LIB.impl("allreduce_bucket_", _allreduce_bucket, _IMPL)


# ------------------------------------------------------------------------------------------------ autograd formulas
def _act_chain(ctx_act, y, g):
    """dL/d(pre-activation) from dL/dy: the fused epilogue activation's derivative, through its output."""
    return g if ctx_act == ACT_NONE else torch.ops.mi355x_gan.act_bwd(y, g, ctx_act)


def _conv2d_setup(ctx, inputs, output):
    x, w, b, stride, pad, pad_mode, act = inputs
    ctx.save_for_backward(x, w, output if act != ACT_NONE else None)
    ctx.geom = (stride, pad, pad_mode, act, b is not None)


def _conv2d_backward(ctx, g):
    x, w, y = ctx.saved_tensors
    stride, pad, pad_mode, act, has_bias = ctx.geom
    g = _act_chain(act, y, g.contiguous())
    dx = torch.ops.mi355x_gan.conv2d_dgrad(g, w, list(x.shape), stride, pad, pad_mode) if ctx.needs_input_grad[0] else None
    dw, db = torch.ops.mi355x_gan.conv2d_wgrad(x, g, list(w.shape), has_bias, stride, pad, pad_mode) if (ctx.needs_input_grad[1] or has_bias) else (None, None)
    return dx, dw, (db if has_bias else None), None, None, None, None


def _convT_setup(ctx, inputs, output):
    x, w, b, act = inputs
    ctx.save_for_backward(x, w, output if act != ACT_NONE else None)
    ctx.geom = (act, b is not None)


def _convT_backward(ctx, g):
    x, w, y = ctx.saved_tensors
    act, has_bias = ctx.geom
    g = _act_chain(act, y, g.contiguous())
    dx = torch.ops.mi355x_gan.conv_transpose2d_dgrad(g, w, list(x.shape)) if ctx.needs_input_grad[0] else None
    dw, db = torch.ops.mi355x_gan.conv_transpose2d_wgrad(x, g, list(w.shape), has_bias)
    return dx, dw, (db if has_bias else None), None


def _in_setup(ctx, inputs, output):
    x, eps, act, residual = inputs
    ctx.save_for_backward(x, output[1])
    ctx.act, ctx.has_res = act, residual is not None


def _in_backward(ctx, gy, gstats):
    x, stats = ctx.saved_tensors
    dx = torch.ops.mi355x_gan.instance_norm_bwd(x, stats, gy.contiguous(), ctx.act)
    return dx, None, None, (gy if ctx.has_res else None)


torch.library.register_autograd("mi355x_gan::conv2d_fwd", _conv2d_backward, setup_context=_conv2d_setup)
torch.library.register_autograd("mi355x_gan::conv_transpose2d_fwd", _convT_backward, setup_context=_convT_setup)
torch.library.register_autograd("mi355x_gan::instance_norm_fwd", _in_backward, setup_context=_in_setup)
torch.library.register_autograd("mi355x_gan::reflection_pad2d", lambda ctx, g: (torch.ops.mi355x_gan.reflection_pad2d_bwd(g.contiguous(), ctx.pad), None),
                                setup_context=lambda ctx, inputs, output: setattr(ctx, "pad", inputs[1]))


def _nce_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])


def _nce_backward(ctx, g_loss, g_saved):
    # the source features carry no gradient (patchnce_cut.py:138-142 computes them under no_grad), nor do the ids
    return None, torch.ops.mi355x_gan.patchnce_bwd(ctx.saved_tensors[0], g_loss), None, None


def _aug_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[1])


torch.library.register_autograd("mi355x_gan::patchnce_fwd", _nce_backward, setup_context=_nce_setup)
torch.library.register_autograd("mi355x_gan::diffaugment_fwd", lambda ctx, g: (torch.ops.mi355x_gan.diffaugment_bwd(g.contiguous(), ctx.saved_tensors[0]), None),
                                setup_context=_aug_setup)
torch.library.register_autograd("mi355x_gan::replication_pad2d", lambda ctx, g: (torch.ops.mi355x_gan.replication_pad2d_bwd(g.contiguous(), ctx.pad), None),
                                setup_context=lambda ctx, inputs, output: setattr(ctx, "pad", inputs[1]))


# ------------------------------------------------------------------------------------------------ drop-in modules
_ACTS = {None: ACT_NONE, "relu": ACT_RELU, "leaky_relu": ACT_LRELU, "tanh": ACT_TANH}


class Conv2d(nn.Conv2d):
    """torch.nn.Conv2d (same constructor, parameters, init and state_dict) whose forward is mi355x_gan::conv2d_fwd.  Supported: square
    kernels, stride 1 or 2, dilation 1, groups 1, padding_mode 'zeros' or 'reflect' -- the reference's layers."""

    def forward(self, x):
        assert self.dilation == (1, 1) and self.groups == 1 and self.kernel_size[0] == self.kernel_size[1] and self.stride[0] == self.stride[1]
        assert self.padding_mode in ("zeros", "reflect") and self.padding[0] == self.padding[1]
        return torch.ops.mi355x_gan.conv2d_fwd(x, self.weight, self.bias, self.stride[0], self.padding[0],
                                               PAD_REFLECT if self.padding_mode == "reflect" else PAD_ZERO, ACT_NONE)


class ConvTranspose2d(nn.ConvTranspose2d):
    """torch.nn.ConvTranspose2d(k=3, stride=2, padding=1, output_padding=1), the reference's upsampling layer
    (generator_resnet_attn.py:146-149, Basic_GAN/src/models.py:50-51), on mi355x_gan::conv_transpose2d_fwd."""

    def forward(self, x, output_size=None):
        assert output_size is None and self.kernel_size == (3, 3) and self.stride == (2, 2) and self.padding == (1, 1) and self.output_padding == (1, 1)
        return torch.ops.mi355x_gan.conv_transpose2d_fwd(x, self.weight, self.bias, ACT_NONE)


class InstanceNorm2d(nn.InstanceNorm2d):
    """torch.nn.InstanceNorm2d as the reference builds it (affine=False, track_running_stats=False, eps 1e-5): no parameters, no buffers."""

    def forward(self, x):
        assert not self.affine and not self.track_running_stats
        return torch.ops.mi355x_gan.instance_norm_fwd(x, float(self.eps), ACT_NONE, None)[0]


class ReflectionPad2d(nn.Module):
    def __init__(self, padding: int):
        super().__init__()
        self.padding = int(padding)

    def forward(self, x):
        return torch.ops.mi355x_gan.reflection_pad2d(x, self.padding)


class ReplicationPad2d(nn.Module):
    """nn.ReplicationPad2d (generator_resnet_attn.py:26-27,45-46: padding_type='replicate', off in the shipped configs)."""

    def __init__(self, padding: int):
        super().__init__()
        self.padding = int(padding)

    def forward(self, x):
        return torch.ops.mi355x_gan.replication_pad2d(x, self.padding)


class _BatchNormFn(torch.autograd.Function):
    """Training-mode nn.BatchNorm2d on the InstanceNorm kernels: a (B,C,H,W) batch is ONE (1,C,B*H,W) instance.  The affine part and the
    running statistics are elementwise / per-channel torch ops (generator_resnet_attn.py:57-58: norm='batch', off in the shipped configs)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        B, C, H, W = x.shape
        x1 = x.permute(1, 0, 2, 3).reshape(1, C, B * H, W).contiguous()
        assert abs(eps - IN_EPS) < 1e-12
        xh1, stats = torch.ops.mi355x_gan.instance_norm_fwd(x1, eps, ACT_NONE, None)
        xh = xh1.reshape(C, B, H, W).permute(1, 0, 2, 3).contiguous()
        ctx.save_for_backward(x1, stats, xh, weight)
        ctx.shape = (B, C, H, W)
        ctx.mark_non_differentiable(stats)
        y = xh * weight.view(1, C, 1, 1) + bias.view(1, C, 1, 1)
        return y, stats

    @staticmethod
    def backward(ctx, gy, _gstats):
        x1, stats, xh, weight = ctx.saved_tensors
        B, C, H, W = ctx.shape
        gw, gb = (gy * xh).sum((0, 2, 3)), gy.sum((0, 2, 3))
        g1 = (gy * weight.view(1, C, 1, 1)).permute(1, 0, 2, 3).reshape(1, C, B * H, W).contiguous()
        dx1 = torch.ops.mi355x_gan.instance_norm_bwd(x1, stats, g1, ACT_NONE)
        return dx1.reshape(C, B, H, W).permute(1, 0, 2, 3).contiguous(), gw, gb, None


class BatchNorm2d(nn.BatchNorm2d):
    """torch.nn.BatchNorm2d (affine, running statistics) whose training-mode normalisation runs on the InstanceNorm kernels."""

    def forward(self, x):
        assert self.affine and self.track_running_stats
        if not self.training:
            scale = self.weight / torch.sqrt(self.running_var + self.eps)
            return x * scale.view(1, -1, 1, 1) + (self.bias - self.running_mean * scale).view(1, -1, 1, 1)
        y, stats = _BatchNormFn.apply(x, self.weight, self.bias, float(self.eps))
        with torch.no_grad():          # running statistics: momentum update with the UNBIASED batch variance (torch semantics)
            C = x.shape[1]
            n = x.numel() // C
            st = stats[:C * 2].view(C, 2)
            mean, var = st[:, 0], (1.0 / st[:, 1] ** 2 - self.eps)
            m = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked + 1)
            self.running_mean.mul_(1 - m).add_(mean, alpha=m)
            self.running_var.mul_(1 - m).add_(var * (n / max(n - 1, 1)), alpha=m)
            self.num_batches_tracked += 1
        return y
