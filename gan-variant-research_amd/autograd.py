"""Module-granular autograd bridge: the reference-named nn.Modules as ordinary PyTorch modules on the HIP kernels.

`G(x)`, `G.get_feature_layers(x, ids)` and `D(x)` return NCHW fp32 tensors that carry an autograd node; `loss.backward()` runs
the network's prebuilt HIP backward program and hands parameter / input gradients to autograd.  This is the path on which the
reference's own training scripts (training/train_cutpp.py:206-331, Basic_GAN/src/train.py:66-122) drive the kernels unchanged:
any optimiser over `module.parameters()`, `GradScaler`, `clip_grad_norm_`, `.detach()`, several live forward passes per step.
The fused `CutTrainer` / `CycleGANTrainer` remain the fast path (one forward of G shared by three consumers, no layout round trips).

One autograd node per *module call* (not per layer): activations stay in halo-NHWC inside a pass slot that is leased from the
module's pool at forward time and returned when autograd drops the node.  What this bridge does not provide is double backward
through a module (the reference's `r1_regularization` uses `create_graph=True`): use `r1_regularization` below instead.
"""
from __future__ import annotations

import weakref
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import F32, BF16, HALO_NONE, HALO_ZERO
from .nets import DiscriminatorNet, GeneratorNet, generator_keys
from .runtime import Ctx, HipOps, Program, View, cpad

# tests replace this with the CPU emulator's constructor; the product always builds HipOps (no CPU fallback)
_OPS_FACTORY: Callable[[torch.device], object] = lambda device: HipOps(device)


# Kernels that update parameters through raw pointers (HipAdam) do not bump the tensors' autograd version counters; they bump
# this epoch instead, and every bridge refreshes its operand copies when either changed.
_WEIGHT_EPOCH = [0]


def notify_weights_changed():
    _WEIGHT_EPOCH[0] += 1


def _new_ctx(device: torch.device, dtype: int) -> Ctx:
    ops = _OPS_FACTORY(device)
    if getattr(ops, "is_hip", False) and device.type != "cuda":
        raise _lib.GanError("the MI355X path needs tensors on a GPU (there is no CPU fallback)")
    return Ctx(ops, device, dtype)


class _Lease:
    """Returns a pass slot to its pool when the autograd node that holds it is destroyed."""

    def __init__(self, slot):
        self.slot = slot

    def __del__(self):
        if self.slot.node is not None and self.slot.node() is None:   # not re-leased to another node meanwhile
            self.slot.busy, self.slot.node = False, None


class _Slot:
    def __init__(self):
        self.busy = False
        self.node = None                  # weakref to the autograd node (Function ctx) that leased the slot
        self.bwd: Dict[tuple, Program] = {}

    def reclaim(self) -> bool:
        """Free if never leased, if its autograd node is gone, or if autograd has released the node's saved tensors
        (backward ran without retain_graph): the same lifetime PyTorch gives a layer's saved activations."""
        if not self.busy:
            return True
        node = self.node() if self.node is not None else None
        if node is not None:
            try:
                node.saved_tensors
                return False
            except RuntimeError:
                pass
        self.busy, self.node = False, None
        return True


class _Bridge:
    """Engine state of one module: parameter aliases, gradient buffers, the network plan and a pool of pass slots."""

    def __init__(self, module: torch.nn.Module, device: torch.device, dtype: int):
        self.module, self.device, self.dtype = module, device, dtype
        self.ctx = _new_ctx(device, dtype)
        self.names = [k for k, _ in module.named_parameters()]
        self.plist = [p for _, p in module.named_parameters()]
        for p in self.plist:
            assert p.dtype == torch.float32 and p.is_contiguous() and p.device == device, "parameters: contiguous fp32 on the module's device"
        # the engine reads the Parameters' own storage: optimiser updates are seen by the next repack, nothing is copied
        self.params = {k: p.data for k, p in zip(self.names, self.plist)}
        self.grads = {k: torch.zeros_like(p.data) for k, p in zip(self.names, self.plist)}
        self.net = self._make_net()
        self.pool: Dict[tuple, List[_Slot]] = {}
        self._packed_version = None
        self._repack: Optional[Program] = None

    # the operand copies are refreshed when any parameter was modified in place since the last forward
    def _sync_weights(self):
        for k, p in zip(self.names, self.plist):
            if p.data.data_ptr() != self.params[k].data_ptr():
                raise RuntimeError(f"parameter {k} was re-allocated after the first forward (e.g. .to()/.half()): rebuild the module's bridge")
        ver = (_WEIGHT_EPOCH[0],) + tuple(p._version for p in self.plist)
        if ver != self._packed_version:
            if self._repack is None:
                self._repack = self._repack_program()
            self._repack.run()
            self._packed_version = ver

    def _repack_program(self) -> Program:
        return self.net.repack_program()

    def _lease(self, key, make) -> _Slot:
        slots = self.pool.setdefault(key, [])
        for s in slots:
            if s.reclaim():
                s.busy = True
                return s
        s = make()
        s.busy = True
        slots.append(s)
        self._repack = None            # planning a new pass may have allocated operand copies: rebuild the repack program
        self._packed_version = None
        return s

    def param_grads(self, needs: Sequence[bool]):
        return tuple(self.grads[k].clone() if need else None for k, need in zip(self.names, needs))


# ------------------------------------------------------------------------------------------------ generator
class _GenBridge(_Bridge):
    def __init__(self, module, device, dtype, style):
        self.style = style
        super().__init__(module, device, dtype)

    def _make_net(self):
        m = self.module
        in_c, out_c = getattr(m, "in_c", getattr(m, "input_nc", 3)), getattr(m, "out_c", getattr(m, "output_nc", 3))
        self.reflect = getattr(m, "padding_type", "reflect") == "reflect"
        act = _lib.ACT_LRELU if getattr(m, "activation", "relu") == "leaky_relu" else _lib.ACT_RELU
        return GeneratorNet(self.ctx, self.params, self.grads, self.style, m.n_blocks, m.ngf, in_c, out_c, need_input_grad=True,
                            reflect=self.reflect, block_act=act)

    def touched(self, last: Optional[int]) -> List[bool]:
        """Which parameters a pass that stops after numbered activation `last` (None: full) has gradients for."""
        k_init, k_down, k_blk, k_up, k_out = generator_keys(self.style, self.net.n_blocks, reflect=self.reflect)
        nb = self.net.n_blocks
        full = last is None
        last = self.net.n_layers - 1 if full else last
        keys = [k_init] + k_down[:max(0, min(2, last))]
        keys += [k for i, pair in enumerate(k_blk) if 3 + i <= last for k in pair]
        keys += [k for j, k in enumerate(k_up) if 3 + nb + j <= last]
        if full:
            keys.append(k_out)
        pref = tuple(k + "." for k in keys)
        return [n.startswith(pref) for n in self.names]

    def real_channels(self, i: int) -> int:
        """Channel count of numbered activation i (views pad channels to a power of two)."""
        g, nb = self.net.ngf, self.net.n_blocks
        return g if i == 0 else 2 * g if i == 1 else 4 * g if i < 3 + nb else 2 * g if i == 3 + nb else g

    def _make_slot(self, B, H, W, last):
        ctx, ops, net = self.ctx, self.ctx.ops, self.net
        s = _Slot()
        s.gp = net.new_pass(B, H, W, last)
        s.xin = torch.zeros(B, net.in_c, H, W, dtype=torch.float32, device=self.device)
        s.fwd = s.gp.fwd_program(s.xin)
        s.out = None
        if s.gp.full:
            s.out = torch.zeros(B, net.out_c, H, W, dtype=torch.float32, device=self.device)
            s.fwd.add(ops.view_to_nchw(s.gp.img, net.out_c, s.out))
            s.g_out = torch.zeros_like(s.out)
            s.g_img = ctx.view(B, H, W, s.gp.img.C, 0)
        s.feat_out, s.feat_gin, s.feat_gview = {}, {}, {}
        s.last = last
        s.gx = torch.zeros_like(s.xin)
        return s

    def forward(self, x: torch.Tensor, feat_ids: Optional[Tuple[int, ...]], keep: bool):
        B, C, H, W = x.shape
        assert C == self.net.in_c
        last = None if feat_ids is None else max(feat_ids)
        # Forward-only calls of a module with `use_graph = True` replay the pass as ONE hipGraph launch: at small batch the ~110
        # launches of a generator pass cost more host time than GPU time.  Ops bake the stream that is current when they are built,
        # so a graph slot is planned, warmed up and captured on a stream of its own; the replay runs on the caller's stream.
        use_graph = (not keep) and feat_ids is None and getattr(self.module, "use_graph", False) and getattr(self.ctx.ops, "is_hip", False)

        def make_graph_slot():
            gs = torch.cuda.Stream(self.device)
            with torch.cuda.stream(gs):
                slot = self._make_slot(B, H, W, last)
            slot.gstream, slot.graph = gs, None
            return slot
        s = self._lease((B, H, W, last, "graph"), make_graph_slot) if use_graph else self._lease((B, H, W, last), lambda: self._make_slot(B, H, W, last))
        try:
            self._sync_weights()
            s.xin.copy_(x)
            if use_graph:
                if s.graph is None:
                    s.gstream.wait_stream(torch.cuda.current_stream(self.device))
                    with torch.cuda.stream(s.gstream):
                        s.fwd.run()                      # eager once: first-launch work (function attributes) must not happen under capture
                    torch.cuda.synchronize(self.device)
                    s.graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(s.graph, stream=s.gstream):
                        s.fwd.run()
                s.graph.replay()
            else:
                s.fwd.run()
            if feat_ids is None:
                outs = (s.out.clone(),)
            else:
                outs = []
                for i in feat_ids:
                    if i not in s.feat_out:      # staging tensor + prebuilt conversion, once per (slot, layer)
                        a = s.gp.acts[i]
                        t = torch.zeros(a.B, self.real_channels(i), a.H, a.W, dtype=torch.float32, device=self.device)
                        s.feat_out[i] = (t, self.ctx.ops.view_to_nchw(a, t.shape[1], t))
                    t, op = s.feat_out[i]
                    op()
                    outs.append(t.clone())
                outs = tuple(outs)
        except BaseException:
            s.busy = False
            raise
        if not keep:
            s.busy = False
        return s, outs

    def backward(self, s: _Slot, feat_ids, grads_out, need_x: bool):
        ctx, ops, gp = self.ctx, self.ctx.ops, s.gp
        key = (feat_ids, tuple(g is not None for g in grads_out), need_x)
        prog = s.bwd.get(key)
        if prog is None:
            prog = Program("G.autograd.bwd")
            hooks = {}
            g_img = None
            if feat_ids is None:
                prog.add(ops.nchw_to_view(s.g_out, self.net.out_c, s.g_img, HALO_NONE))
                g_img = s.g_img
            else:
                for i, g in zip(feat_ids, grads_out):
                    if g is None:
                        continue
                    a = gp.acts[i]
                    s.feat_gin[i] = torch.zeros(a.B, self.real_channels(i), a.H, a.W, dtype=torch.float32, device=self.device)
                    s.feat_gview[i] = ctx.view(a.B, a.H, a.W, a.C, 0)
                    prog.add(ops.nchw_to_view(s.feat_gin[i], self.real_channels(i), s.feat_gview[i], HALO_NONE))

                    def mk(i=i):
                        return lambda gv: [ops.fold_add(gv, s.feat_gview[i], False, gv)]
                    hooks[i] = mk()
            prog.add(gp.bwd_program(g_img, False, None, hooks=hooks, accumulate=False, need_input_grad=need_x))
            if need_x:
                tmp = ctx.view(gp.B, gp.H, gp.W, gp.x0.C, 0)
                prog.add(ops.fold_add(None, gp.g_input, gp.g_input_fold, tmp))      # reflection-pad gradient of the first layer
                prog.add(ops.view_to_nchw(tmp, self.net.in_c, s.gx))
            s.bwd[key] = prog
            self._repack, self._packed_version = None, None   # planning may have allocated gradient-order operand copies
            self._sync_weights()
        if feat_ids is None:
            s.g_out.copy_(grads_out[0])
        else:
            for i, g in zip(feat_ids, grads_out):
                if g is not None:
                    s.feat_gin[i].copy_(g)
        prog.run()
        return s.gx.clone() if need_x else None


class _GenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bridge: _GenBridge, feat_ids, keep, x, *params):
        slot, outs = bridge.forward(x.detach(), feat_ids, keep)
        ctx.bridge, ctx.feat_ids = bridge, feat_ids
        if keep:
            ctx.slot, ctx.lease = slot, _Lease(slot)
            ctx.save_for_backward(x)              # its release by autograd marks the end of the slot's lease (see _Slot.reclaim)
            slot.node = weakref.ref(ctx)
        return outs

    @staticmethod
    def backward(ctx, *grads_out):
        ctx.saved_tensors                         # raises PyTorch's own error on a second backward without retain_graph
        b = ctx.bridge
        gs = tuple(None if g is None else g.detach().float().contiguous() for g in grads_out)
        if all(g is None for g in gs):
            return (None, None, None, None) + (None,) * len(b.plist)
        gx = b.backward(ctx.slot, ctx.feat_ids, gs, ctx.needs_input_grad[3])
        needs = [n and t for n, t in zip(ctx.needs_input_grad[4:], b.touched(ctx.slot.last))]
        return (None, None, None, gx) + b.param_grads(needs)


# ------------------------------------------------------------------------------------------------ discriminator
class _DiscBridge(_Bridge):
    """One PatchGAN discriminator (prefixes of length 1) or the scales of MultiscaleDiscriminator: scale i sees the input
    average-pooled i times (discriminator_patchgan.py:102-116)."""

    def __init__(self, module, device, dtype, style, prefixes, ndf, n_layers):
        self.style, self.ndf, self.n_layers = style, ndf, n_layers
        self.prefixes = [prefixes] if isinstance(prefixes, str) else list(prefixes)
        super().__init__(module, device, dtype)

    SN_EPS = 1e-12          # torch.nn.utils.spectral_norm default

    def _make_net(self):
        # spectral norm (discriminator_patchgan.py:21-23): the parameters are `weight_orig`; the convolutions run on
        # W_sn = weight_orig / sigma, which is recomputed -- with one power iteration in training mode -- by EVERY forward.  Two
        # live forwards therefore hold different weights, so the normalised weights, their operand copies and the (u, v, sigma)
        # the backward needs belong to the pass slot, not to the bridge (see _slot_weights).
        self.conv_keys = [k[:-len(".weight_orig")] for k in self.names if k.endswith(".weight_orig")]
        self.sn = bool(self.conv_keys)
        if self.sn:
            self.bufs = dict(self.module.named_buffers())
            for k in self.conv_keys:
                self.grads[k + ".weight"] = torch.zeros_like(self.params[k + ".weight_orig"])     # dL/dW_sn, written by the wgrad kernels
                self.params[k + ".weight"] = torch.zeros_like(self.params[k + ".weight_orig"])    # shape template for the bridge-level plan
        self.nets = self._build_nets(self.params)
        return self.nets[0]

    def _build_nets(self, params):
        return [DiscriminatorNet(self.ctx, params, self.grads, self.style, p, self.ndf, self.n_layers) for p in self.prefixes]

    def _repack_program(self) -> Program:
        prog = Program("D.repack")
        for net in self.nets:
            prog.add(net.repack_program())
        return prog

    def _slot_weights(self, s: _Slot):
        """Without spectral norm a slot shares the bridge's networks.  With it, the slot owns W_sn, the operand copies made from
        it and snapshots of u, v, sigma, and two programs (training / eval) that produce them from weight_orig."""
        if not self.sn:
            s.nets = self.nets
            return
        ctx, ops = self.ctx, self.ctx.ops
        params = dict(self.params)
        s.sn_state = {}
        for k in self.conv_keys:
            W = self.params[k + ".weight_orig"]
            h, w = W.shape[0], W.numel() // W.shape[0]
            st = {"W": W, "Wsn": torch.zeros_like(W), "u": ctx.f32(h), "v": ctx.f32(w), "sigma": ctx.f32(1),
                  "ws": ctx.f32(ops.spectral_norm_ws_floats(h, w)), "u_buf": self.bufs[k + ".weight_u"], "v_buf": self.bufs[k + ".weight_v"]}
            params[k + ".weight"] = st["Wsn"]
            s.sn_state[k] = st
        s.nets = self._build_nets(params)
        s.sn_fwd = {}
        for training in (True, False):
            prog = Program("D.spectral_norm")
            for k, st in s.sn_state.items():
                prog.add(ops.spectral_norm_fwd(st["W"], st["u_buf"], st["v_buf"], training, self.SN_EPS, st["sigma"], st["Wsn"], st["ws"]))
                prog.add(lambda st=st: (st["u"].copy_(st["u_buf"]), st["v"].copy_(st["v_buf"])))
            s.sn_fwd[training] = prog
        s.repack = None

    def _sn_bwd_program(self, s: _Slot) -> Program:
        """dL/dweight_orig from dL/dW_sn (u, v, sigma as the slot's forward left them)."""
        prog = Program("D.spectral_norm.bwd")
        for k, st in s.sn_state.items():
            prog.add(self.ctx.ops.spectral_norm_bwd(self.grads[k + ".weight"], st["Wsn"], st["u"], st["v"], st["sigma"],
                                                    self.grads[k + ".weight_orig"], st["ws"]))
        return prog

    def _prepare_weights(self, s: _Slot):
        """Operand copies current for this slot's next launches."""
        if not self.sn:
            self._sync_weights()
            return
        for k, p in zip(self.names, self.plist):
            if p.data.data_ptr() != self.params[k].data_ptr():
                raise RuntimeError(f"parameter {k} was re-allocated after the first forward (e.g. .to()/.half()): rebuild the module's bridge")
        s.sn_fwd[bool(self.module.training)].run()
        self._slot_repack(s)

    def _slot_repack(self, s: _Slot):
        if s.repack is None:
            s.repack = Program("D.repack.slot")
            for net in s.nets:
                s.repack.add(net.repack_program())
        s.repack.run()

    def _passes(self, s: _Slot, B, H, W):
        self._slot_weights(s)
        dps = []
        for net in s.nets:
            dps.append(net.new_pass(B, H, W))
            H, W = (H - 1) // 2 + 1, (W - 1) // 2 + 1       # AvgPool2d(3, 2, 1)
        return dps

    def _make_slot(self, B, H, W):
        ops, net = self.ctx.ops, self.net
        s = _Slot()
        s.dps = self._passes(s, B, H, W)
        s.xin = torch.zeros(B, net.in_c, H, W, dtype=torch.float32, device=self.device)
        s.fwd = Program("D.autograd.fwd")
        s.fwd.add(ops.nchw_to_view(s.xin, net.in_c, s.dps[0].x, HALO_ZERO))
        s.outs, s.g_outs = [], []
        for i, dp in enumerate(s.dps):
            if i > 0:
                s.fwd.add(ops.avgpool_fwd(s.dps[i - 1].x, dp.x))
            s.fwd.add(dp.fwd_program())
            lg = dp.logits
            s.outs.append(torch.zeros(B, 1, lg.H, lg.W, dtype=torch.float32, device=self.device))
            s.fwd.add(ops.view_to_nchw(lg, 1, s.outs[-1]))
            s.g_outs.append(torch.zeros_like(s.outs[-1]))
        s.gx = torch.zeros_like(s.xin)
        return s

    def forward(self, x, keep: bool):
        B, C, H, W = x.shape
        assert C == self.net.in_c
        s = self._lease((B, H, W), lambda: self._make_slot(B, H, W))
        try:
            self._prepare_weights(s)
            s.xin.copy_(x)
            s.fwd.run()
            outs = tuple(o.clone() for o in s.outs)
        except BaseException:
            s.busy = False
            raise
        if not keep:
            s.busy = False
        return s, outs

    def backward(self, s: _Slot, gs: Sequence[torch.Tensor], need_x: bool, need_w: bool):
        ops = self.ctx.ops
        key = (need_x, need_w)
        prog = s.bwd.get(key)
        if prog is None:
            prog = Program("D.autograd.bwd")
            for dp, g_out in zip(s.dps, s.g_outs):
                gl = dp.grad_logits_view()
                prog.add(ops.nchw_to_view(g_out, 1, gl, HALO_ZERO))
                prog.add(dp.bwd_program(gl, wgrad=need_w, accumulate=False, need_input_grad=need_x))
            if need_x:
                for i in range(len(s.dps) - 2, -1, -1):      # dL/dx_i += pool^T dL/dx_{i+1}
                    prog.add(ops.avgpool_bwd(s.dps[i + 1].g_input, s.dps[i].g_input, True))
                prog.add(ops.view_to_nchw(s.dps[0].g_input, self.net.in_c, s.gx))
            if need_w and self.sn:
                prog.add(self._sn_bwd_program(s))
            s.bwd[key] = prog
            if self.sn:                      # planning the backward allocated operand copies: fill them from the slot's W_sn
                s.repack = None
                self._slot_repack(s)
            else:
                self._repack, self._packed_version = None, None
                self._sync_weights()
        for dst, g in zip(s.g_outs, gs):
            dst.copy_(g)
        prog.run()
        return s.gx.clone() if need_x else None


class _DiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bridge: _DiscBridge, keep, x, *params):
        slot, outs = bridge.forward(x.detach(), keep)
        ctx.bridge = bridge
        if keep:
            ctx.slot, ctx.lease = slot, _Lease(slot)
            ctx.save_for_backward(x)
            slot.node = weakref.ref(ctx)
        return outs

    @staticmethod
    def backward(ctx, *gs):
        ctx.saved_tensors
        b = ctx.bridge
        need_w = any(ctx.needs_input_grad[3:])
        gx = b.backward(ctx.slot, [g.detach().float().contiguous() for g in gs], ctx.needs_input_grad[2], need_w)
        return (None, None, gx) + (b.param_grads(ctx.needs_input_grad[3:]) if need_w else (None,) * len(b.plist))


# ------------------------------------------------------------------------------------------------ module-facing helpers
def _bridge_of(module, make) -> _Bridge:
    p = next(module.parameters())
    dtype = getattr(module, "compute_dtype", F32)
    b = getattr(module, "_hip_bridge", None)
    if b is None or b.device != p.device or b.dtype != dtype:
        b = make(p.device, dtype)
        object.__setattr__(module, "_hip_bridge", b)      # not a submodule / buffer: plain attribute
    return b


def generator_forward(module, x: torch.Tensor, style: str, feat_ids: Optional[Sequence[int]] = None):
    """ResNetGenerator.forward / get_feature_layers (generator_resnet_attn.py:165-188, 190-235) with autograd."""
    b = _bridge_of(module, lambda dev, dt: _GenBridge(module, dev, dt, style))
    ids = None if feat_ids is None else tuple(int(i) for i in feat_ids)
    if ids is not None and not ids:
        return []
    keep = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in b.plist))
    outs = _GenFn.apply(b, ids, keep, x.float().contiguous(), *b.plist)
    return outs[0] if ids is None else list(outs)


def discriminator_forward(module, x: torch.Tensor, style: str, prefix, ndf: int, n_layers: int):
    """PatchGANDiscriminator.forward (discriminator_patchgan.py:56-63) / NLayerDiscriminator.forward with autograd; with a list of
    prefixes, MultiscaleDiscriminator.forward (:102-116): the list of per-scale logits from ONE autograd node."""
    b = _bridge_of(module, lambda dev, dt: _DiscBridge(module, dev, dt, style, prefix, ndf, n_layers))
    keep = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in b.plist))
    outs = _DiscFn.apply(b, keep, x.float().contiguous(), *b.plist)
    return outs[0] if isinstance(prefix, str) else list(outs)


# ------------------------------------------------------------------------------------------------ R1 (double backward)
class _R1Fn(torch.autograd.Function):
    """r1 = mean_b sum_chw (d sum D(x) / dx)^2 with its parameter gradients computed in the forward (explicit second-order
    program, DPass.r1_program); backward scales them by the incoming gradient."""

    @staticmethod
    def forward(ctx, bridge: _DiscBridge, x, *params):
        B, C, H, W = x.shape
        s = bridge._lease((B, H, W, "r1"), lambda: _r1_slot(bridge, B, H, W))
        try:
            bridge._prepare_weights(s)        # spectral norm: the R1 forward is a training-mode forward too (one more power iteration)
            s.xin.copy_(x)
            s.prog.run()
            loss = s.loss.clone().reshape(())
            last_biases = [net.convs[-1].grad_b for net in bridge.nets]
            grads = []
            for k in bridge.names:
                g = bridge.grads[k]
                if any(g is lb for lb in last_biases):
                    grads.append(None)                        # the reference's graph never reaches the last bias
                elif k.endswith(".bias"):
                    grads.append(torch.zeros_like(g))         # d/db of an input gradient is zero
                else:
                    grads.append(g.clone())
            ctx.grads = grads
        finally:
            s.busy = False                                     # nothing of the pass is needed after the forward
        return loss

    @staticmethod
    def backward(ctx, g):
        return (None, None) + tuple(None if t is None or not need else t * g for t, need in zip(ctx.grads, ctx.needs_input_grad[2:]))


def _r1_slot(bridge: _DiscBridge, B, H, W) -> _Slot:
    """With K scales D_total(x) = sum_i sum D_i(P^i x) (P = the average pool), so g = sum_i (P^i)^T g_i and
    d r1 / d theta_i = <(2/B) P^i g, d g_i / d theta_i>: every scale runs its own first-order half, the input gradients are
    folded back through the pools into g, and each scale's second-order half is seeded with g pooled down to its resolution."""
    ctx, ops, net = bridge.ctx, bridge.ctx.ops, bridge.net
    s = _Slot()
    s.dps = bridge._passes(s, B, H, W)
    s.xin = torch.zeros(B, net.in_c, H, W, dtype=torch.float32, device=bridge.device)
    s.loss = ctx.f32(1)
    s.scratch = ctx.f32(1)
    s.prog = Program("R1.autograd")
    s.prog.add(ops.nchw_to_view(s.xin, net.in_c, s.dps[0].x, HALO_ZERO))
    for i, dp in enumerate(s.dps):
        if i > 0:
            s.prog.add(ops.avgpool_fwd(s.dps[i - 1].x, dp.x))
        s.prog.add(dp.fwd_program())
        s.prog.add(dp.r1_first(s.scratch))
    for i in range(len(s.dps) - 2, -1, -1):
        s.prog.add(ops.avgpool_bwd(s.dps[i + 1].g_input, s.dps[i].g_input, True))
    us = [ctx.view(B, dp.H, dp.W, dp.x.C, 1) for dp in s.dps]
    s.prog.add(ops.r1_reduce(s.dps[0].g_input, net.in_c, 1.0, s.loss, us[0], ctx.scratch("r1_ws", 1024)))
    for i, dp in enumerate(s.dps):
        if i > 0:
            s.prog.add(ops.avgpool_fwd(us[i - 1], us[i]))
        s.prog.add(dp.r1_second(us[i]))
    if bridge.sn:
        s.prog.add(bridge._sn_bwd_program(s))
    return s


def r1_regularization(discriminator, real_images: torch.Tensor, amp_ctx=None) -> torch.Tensor:
    """Drop-in for training/train_cutpp.py:165-203 (whose `create_graph=True` double backward cannot cross a module-granular
    autograd node): same value, and `.backward()` deposits the same discriminator gradients.  Always fp32, like the reference;
    `amp_ctx` is accepted for signature compatibility (its loss scaling cancels in the reference: :186-196)."""
    d0 = discriminator
    b = getattr(d0, "_hip_bridge_r1", None)
    p = next(d0.parameters())
    if b is None or b.device != p.device:
        b = _DiscBridge(d0, p.device, F32, "cut", [f"discriminators.{i}.model." for i in range(getattr(d0, "num_scales", 1))], d0.ndf, d0.n_layers)
        object.__setattr__(d0, "_hip_bridge_r1", b)
    return _R1Fn.apply(b, real_images.detach().float().contiguous(), *b.plist)
