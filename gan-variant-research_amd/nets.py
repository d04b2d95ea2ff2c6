"""Generator / discriminator execution engines: explicit forward and backward *programs* over halo-NHWC buffers.

The reference builds its networks from torch.nn modules and lets autograd derive the backward pass
(GAN_Variant1/models/generator_resnet_attn.py:74-235, discriminator_patchgan.py:7-128, Basic_GAN/src/models.py:7-107).
Here every network is a fixed, hand-scheduled list of kernel launches: activations are written once in the
layout their consumer wants (reflect / zero halo materialised by the producer), InstanceNorm + ReLU + residual
add + padding are one pass, and gradients of reflection padding are folded by the consumer.
"""
from __future__ import annotations

import os

from typing import Callable, Dict, List, Optional, Sequence

import torch

from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH, BF16, F32, FP8, HALO_REFLECT, HALO_ZERO
from .convplan import ConvLayer
from .runtime import IN_WS_CHUNKS, Ctx, Program, View, cpad

IN_EPS = 1e-5


def generator_keys(style: str, n_blocks: int, n_down: int = 2, reflect: bool = True):
    """state_dict key prefixes in execution order: (init, [down], [(blk_a, blk_b)], [up], out).  With zero padding the
    reference builds no pad modules (generator_resnet_attn.py:24-30,43-46,110-113,157-160), which shifts the Sequential indices."""
    if style == "cut":
        i0, ia, ib = (1, 1, 5) if reflect else (0, 0, 3)
        return (f"initial.{i0}", [f"downsample.{3*i}" for i in range(n_down)],
                [(f"res_blocks.{b}.conv_block.{ia}", f"res_blocks.{b}.conv_block.{ib}") for b in range(n_blocks)],
                [f"upsample.{3*i}" for i in range(n_down)], f"output.{i0}")
    assert style == "basic" and n_down == 2
    base = 10
    ups = base + n_blocks
    return ("net.1", ["net.4", "net.7"], [(f"net.{base+b}.block.1", f"net.{base+b}.block.5") for b in range(n_blocks)],
            [f"net.{ups}", f"net.{ups+3}"], f"net.{ups+7}")


class _Net:
    def __init__(self, ctx: Ctx, params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor]):
        self.ctx, self.params, self.grads = ctx, params, grads
        self.layers: List[ConvLayer] = []
        self._gbufs = {}

    def _conv(self, key, k, s, p, transposed=False, need_dgrad=True) -> ConvLayer:
        w, b = self.params[key + ".weight"], self.params.get(key + ".bias")
        layer = ConvLayer(self.ctx, w, b, self.grads[key + ".weight"], self.grads.get(key + ".bias"), k, s, p, transposed, need_dgrad)
        self.layers.append(layer)
        return layer

    def repack_program(self) -> Program:
        """Refreshes every operand copy of the network after an optimiser step: ONE batched launch."""
        prog = Program("repack")
        ops = [op for layer in self.layers for op in layer.repack_ops()]
        if ops:
            prog.add(self.ctx.ops.pack_weight_batch([op.pack_args for op in ops]))
        return prog

    def gbuf(self, tag: str, B, H, W, C, halo) -> View:
        """Gradient scratch shared by every pass of this network (launches are stream-ordered)."""
        key = (tag, B, H, W, C, halo)
        v = self._gbufs.get(key)
        if v is None:
            v = self.ctx.view(B, H, W, C, halo)
            self._gbufs[key] = v
        return v

    def in_ws(self, B, C) -> torch.Tensor:
        return self.ctx.scratch("in_ws", B * IN_WS_CHUNKS * C * 2 + B * C * 2 + (B * 1024 + 32) * C)


class GeneratorNet(_Net):
    """9-block ResNet generator of both trainers (CUT: biased convs; Basic_GAN: bias-free but the last)."""

    def __init__(self, ctx, params, grads, style="cut", n_blocks=9, ngf=64, in_c=3, out_c=3, need_input_grad=True, reflect=True,
                 block_act=ACT_RELU, fp8=False):
        """reflect: `padding_type` 'reflect' (the configs' value) or 'zero'; block_act: the residual blocks' activation (ReLU in the
        configs, LeakyReLU(0.2) for activation='leaky_relu', generator_resnet_attn.py:60-66).
        fp8: the residual blocks' 3x3 convolutions (generator_resnet_attn.py:33,48 -- 88 % of the generator's FLOPs) read e4m3 copies of
        their operands in the forward pass and in the input gradient (BASELINE.json configs[4]); weight gradients, InstanceNorm, the
        first / last layers and everything stored stay bf16 / fp32.  bf16 mode with reflect padding only."""
        super().__init__(ctx, params, grads)
        self.fp8 = bool(fp8)
        assert not self.fp8 or (ctx.dtype == BF16 and reflect and 4 * ngf >= 128 and (4 * ngf) % 128 == 0), "fp8 blocks: bf16 mode, reflect padding, >= 128 channels"
        self.style, self.n_blocks, self.ngf, self.in_c, self.out_c = style, n_blocks, ngf, in_c, out_c
        self.reflect, self.block_act = reflect, block_act
        self.pad_mode = HALO_REFLECT if reflect else HALO_ZERO
        k_init, k_down, k_blk, k_up, k_out = generator_keys(style, n_blocks, reflect=reflect)
        self.c_init = self._conv(k_init, 7, 1, 3, need_dgrad=need_input_grad)
        self.c_down = [self._conv(k, 3, 2, 1) for k in k_down]
        self.c_blk = [(self._conv(a, 3, 1, 1), self._conv(b, 3, 1, 1)) for a, b in k_blk]
        self.c_up = [self._conv(k, 3, 2, 1, transposed=True) for k in k_up]
        self.c_out = self._conv(k_out, 7, 1, 3)
        self.n_layers = 3 + n_blocks + 2  # numbered activations of get_feature_layers

    def new_pass(self, B, H, W, last_layer: Optional[int] = None) -> "GPass":
        return GPass(self, B, H, W, last_layer)


class GPass:
    """Buffers of one generator forward (+ its backward).  last_layer=None runs to the output image; otherwise the
    pass stops after that numbered activation (get_feature_layers numbering, generator_resnet_attn.py:204-233)."""

    def __init__(self, net: GeneratorNet, B, H, W, last_layer):
        assert H % 4 == 0 and W % 4 == 0
        self.net, self.B, self.H, self.W = net, B, H, W
        ctx, g, nb = net.ctx, net.ngf, net.n_blocks
        self.full = last_layer is None
        self.last = net.n_layers - 1 if self.full else last_layer
        v = ctx.view
        self.x0 = v(B, H, W, cpad(net.in_c), 3)
        # raw conv outputs, statistics and normalised activations; acts[i] = numbered activation i
        self.raw, self.stats, self.acts = [], [], []
        dims = [(H, W, g, 1)]                                   # H0: zero halo 1 (stride-2 conv, pad 1)
        dims += [(H // 2, W // 2, 2 * g, 1), (H // 4, W // 4, 4 * g, 1)]   # H1 (zero), R0 (reflect)
        dims += [(H // 4, W // 4, 4 * g, 1)] * nb               # block outputs (reflect; the last feeds ConvT: zero)
        dims += [(H // 2, W // 2, 2 * g, 1), (H, W, g, 3)]      # H3 (zero, feeds ConvT), H4 (reflect 3)
        for i, (h, w, c, halo) in enumerate(dims):
            if i > self.last:
                break
            self.acts.append(v(B, h, w, c, halo))
        n_raw = 0
        for i in range(self.last + 1):
            if 3 <= i < 3 + nb:
                h, w, c, _ = dims[i]
                self.raw.append((v(B, h, w, c, 0), v(B, h, w, c, 0)))
                self.stats.append((ctx.f32(B * c * 2), ctx.f32(B * c * 2)))
                n_raw += 1
            else:
                h, w, c, _ = dims[i]
                self.raw.append(v(B, h, w, c, 0))
                self.stats.append(ctx.f32(B * c * 2))
        self.mid = [v(B, H // 4, W // 4, 4 * g, 1) for _ in range(min(nb, max(0, self.last - 2)))]  # T_k: reflect halo 1
        if net.fp8:      # e4m3 copies of the residual convolutions' inputs (block input, T_k); gradients: see bwd_program
            self.in8 = [v(B, H // 4, W // 4, 4 * g, 1, dtype=FP8) for _ in self.mid]
            self.mid8 = [v(B, H // 4, W // 4, 4 * g, 1, dtype=FP8) for _ in self.mid]
        self.img = v(B, H, W, cpad(net.out_c), 0) if self.full else None

    def _fp8_grad_bufs(self, B, h, w, c):
        """e4m3 copy of a residual convolution's output gradient (zero halo 2) with its per-image amax / scale: one set per network and
        shape (producer and consumer are neighbours on the main stream)."""
        net = self.net
        key = ("fp8g", B, h, w, c)
        bufs = net._gbufs.get(key)
        if bufs is None:
            bufs = net._gbufs[key] = (net.ctx.view(B, h, w, c, 2, dtype=FP8), net.ctx.f32(B), net.ctx.f32(B, 1.0))
        return bufs

    def halo_mode(self, i: int) -> int:
        nb = self.net.n_blocks
        if i == 2 or (3 <= i < 3 + nb - 1) or i == 4 + nb:
            return self.net.pad_mode
        return HALO_ZERO

    # ------------------------------------------------------------------ forward
    def fwd_program(self, src) -> Program:
        """src: contiguous NCHW fp32 tensor (the reference's input) or a plain C=8 image View."""
        net, ops, nb = self.net, self.net.ctx.ops, self.net.n_blocks
        prog = Program("G.fwd")
        if isinstance(src, View):
            prog.add(ops.view_copy(src, self.x0, net.pad_mode))
        else:
            prog.add(ops.nchw_to_view(src, net.in_c, self.x0, net.pad_mode))

        def norm(i, raw, stats, act, residual=None, out=None, conv=None, out8=None):
            out = self.acts[i] if out is None else out
            halo = self.halo_mode(i) if out is self.acts[i] else net.pad_mode
            ws = net.in_ws(self.B, raw.C)
            fused = not os.environ.get("GAN_NO_FUSED_FINALIZE")
            if conv is not None and conv.stats_parts:     # the convolution's epilogue already wrote per-tile (sum, sum of squares)
                if fused and conv.stats_parts <= 16:          # few tiles: the apply pass adds them up itself (no statistics launch at all)
                    prog.add(ops.in_apply_parts(raw, ws, conv.stats_parts, IN_EPS, stats, act, residual, out, halo, out8))
                    return
                prog.add(ops.in_stats_from_parts(ws, conv.stats_parts, self.B, raw.C, raw.H * raw.W, IN_EPS, stats))
            elif fused:
                prog.add(ops.in_partial(raw, ws))
                prog.add(ops.in_apply_parts(raw, ws, ops.in_partial_count(raw), IN_EPS, stats, act, residual, out, halo, out8))
                return
            else:
                prog.add(ops.in_stats(raw, IN_EPS, stats, ws))
            prog.add(ops.in_apply(raw, stats, act, residual, out, halo))
            if out8 is not None:
                prog.add(ops.quantize_fp8(out, out8))

        prog.add(net.c_init.fwd(self.x0, self.raw[0], stats_ws=net.in_ws(self.B, self.raw[0].C)))
        norm(0, self.raw[0], self.stats[0], ACT_RELU, conv=net.c_init)
        for i in (1, 2):
            if i > self.last:
                return prog
            prog.add(net.c_down[i - 1].fwd(self.acts[i - 1], self.raw[i]))
            norm(i, self.raw[i], self.stats[i], ACT_RELU, out8=self.in8[0] if (net.fp8 and i == 2 and self.last > 2) else None)
        for k in range(nb):
            i = 3 + k
            if i > self.last:
                return prog
            ca, cb = net.c_blk[k]
            ra, rb = self.raw[i]
            sa, sb = self.stats[i]
            if net.fp8:      # the e4m3 operand copies come out of the InstanceNorm passes that produce the bf16 tensors
                prog.add(ca.fwd8(self.in8[k], ra, stats_ws=net.in_ws(self.B, ra.C)))
                norm(i, ra, sa, net.block_act, out=self.mid[k], conv=ca, out8=self.mid8[k])
                prog.add(cb.fwd8(self.mid8[k], rb, stats_ws=net.in_ws(self.B, rb.C)))
                nxt8 = self.in8[k + 1] if (k + 1 < nb and i + 1 <= self.last) else None
                norm(i, rb, sb, ACT_NONE, residual=self.acts[i - 1], conv=cb, out8=nxt8)
                continue
            prog.add(ca.fwd(self.acts[i - 1], ra, stats_ws=net.in_ws(self.B, ra.C)))
            norm(i, ra, sa, net.block_act, out=self.mid[k], conv=ca)
            prog.add(cb.fwd(self.mid[k], rb, stats_ws=net.in_ws(self.B, rb.C)))
            norm(i, rb, sb, ACT_NONE, residual=self.acts[i - 1], conv=cb)
        for j in range(2):
            i = 3 + nb + j
            if i > self.last:
                return prog
            prog.add(net.c_up[j].fwd(self.acts[i - 1], self.raw[i]))
            norm(i, self.raw[i], self.stats[i], ACT_RELU)
        if self.full:
            prog.add(net.c_out.fwd(self.acts[-1], self.img, ACT_TANH))
        return prog

    # ------------------------------------------------------------------ backward
    def bwd_program(self, g_img: Optional[View] = None, g_img_fold: bool = False, g_img2: Optional[View] = None,
                    hooks: Optional[Dict[int, Callable[[View], list]]] = None, accumulate: bool = False,
                    need_input_grad: bool = False, bucket: Optional[tuple] = None) -> Program:
        """Backward of this pass.  g_img (+ g_img2): gradient wrt the output image (folded over a reflect halo if
        g_img_fold).  hooks[i](g_view) returns ops that add extra gradient into activation i's gradient before it
        is consumed (PatchNCE).  Weight gradients are written (accumulate=False) or added into net.grads."""
        net, ctx, ops, nb, B = self.net, self.net.ctx, self.net.ctx.ops, self.net.n_blocks, self.B
        hooks = hooks or {}
        rf = net.reflect      # reflection padding: input gradients are produced on the padded domain and folded by their consumer
        prog = Program("G.bwd")
        H, W, g = self.H, self.W, net.ngf
        acc = accumulate

        def hook(i, gv):
            if i in hooks:
                prog.add(hooks[i](gv))

        # Weight gradients go to a second HIP stream: they are MFMA-bound and nothing on the backward chain waits for them, so
        # they overlap with the HBM-bound part of the chain (InstanceNorm backward, reflection folds: 63 % of its time hidden in a
        # two-kernel probe).  An event after the producer of dy orders the side stream; before the main stream rewrites a dy buffer
        # it waits for the side launches that read it; the program ends with a join.
        side = ops.side()
        readers = {}     # id(dy buffer) -> event recorded on the side stream after its last reader

        def wgrad_side(conv, x, dy, bias_too):
            ev = ops.new_event()
            prog.add(ops.record(ev))
            prog.add(side.wait(ev))
            prog.add(conv.wgrad(x, dy, acc, bias_too=bias_too, ops=side))
            done = ops.new_event()
            prog.add(side.record(done))
            readers[id(dy.t)] = done

        def before_write(buf: View):
            ev = readers.pop(id(buf.t), None)
            if ev is not None:
                prog.add(ops.wait(ev))

        # bias gradients in front of a norm are column sums of dx: the norm backward leaves per-block partials in a buffer of the
        # layer's own and ONE launch at the end of the program sums them for every layer (two tiny launches per layer less on the chain)
        defer = os.environ.get("GAN_BIAS_DEFER", "1") != "0"
        bias_items = []
        if not hasattr(self, "_bias_parts"):
            self._bias_parts = {}

        def inbwd(raw, stats, act, gy, fold, dx, conv=None, amax=None, parts=None):
            """InstanceNorm backward; with `conv`, its bias gradient (column sums of dx) comes out of the same pass; with `amax`, max|dx|
            per image too (the scale of dx's e4m3 copy); with `parts` = (ws, nparts, mode) the two sums were written by the producer of gy
            (an input-gradient epilogue, ConvLayer.dgrad(chain=...)) and only the apply half runs."""
            if parts is not None:
                part = None
                if conv is not None and conv.grad_b is not None:
                    nparts = ops.in_bwd_bias_parts(raw)
                    part = self._bias_parts.get(id(conv))
                    if part is None:
                        part = self._bias_parts[id(conv)] = ctx.f32(nparts * raw.C)
                    bias_items.append((part, nparts, raw.C, conv.grad_b, conv.cout, acc))
                prog.add(ops.in_bwd_parts(raw, stats, act, gy, fold, dx, parts[0], parts[1], parts[2], part))
                return
            if amax is not None:
                part = None
                if conv is not None and conv.grad_b is not None:
                    nparts = ops.in_bwd_bias_parts(raw)
                    part = self._bias_parts.get(id(conv))
                    if part is None:
                        part = self._bias_parts[id(conv)] = ctx.f32(nparts * raw.C)
                    bias_items.append((part, nparts, raw.C, conv.grad_b, conv.cout, acc))
                prog.add(ops.in_bwd_amax(raw, stats, act, gy, fold, dx, net.in_ws(B, raw.C), part, amax))
            elif conv is not None and conv.grad_b is not None and defer:
                nparts = ops.in_bwd_bias_parts(raw)
                part = self._bias_parts.get(id(conv))
                if part is None:
                    part = self._bias_parts[id(conv)] = ctx.f32(nparts * raw.C)
                prog.add(ops.in_bwd_bias_deferred(raw, stats, act, gy, fold, None, dx, net.in_ws(B, raw.C), part))
                bias_items.append((part, nparts, raw.C, conv.grad_b, conv.cout, acc))
            elif conv is not None and conv.grad_b is not None:
                prog.add(ops.in_bwd_bias(raw, stats, act, gy, fold, None, dx, net.in_ws(B, raw.C), conv.grad_b, conv.cout, acc))
            else:
                prog.add(ops.in_bwd(raw, stats, act, gy, fold, None, dx, net.in_ws(B, raw.C)))

        i = self.last
        g_cur: Optional[View] = None   # gradient wrt acts[i]
        g_fold = False
        # Backward chain of the residual blocks (bf16, reflect padding, ReLU blocks): the input gradient of a block's second convolution
        # leaves the two sums of the ReLU'd norm's backward in its epilogue (ConvLayer.dgrad(chain=...)); gan_in_bwd_parts then applies
        # with x and g read once -- one of the five HBM-bound passes per block is gone (DESIGN 3.6 has what else was built and measured).
        use_chain = False
        if self.last >= 3 and ctx.dtype == BF16 and rf and not net.fp8 and net.block_act == ACT_RELU and not os.environ.get("GAN_NO_BWD_CHAIN"):
            r0 = self.raw[3][0]
            use_chain = net.c_blk[0][1].can_chain(net.gbuf("dy_blk_b0", B, r0.H, r0.W, r0.C, 2), net.gbuf("g_blk_p", B, r0.H, r0.W, r0.C, 1))
        self.bwd_chain = use_chain
        if self.full:
            assert g_img is not None
            dyo = net.gbuf("dy_out", B, H, W, self.img.C, 6)
            prog.add(ops.act_bwd(self.img, ACT_TANH, g_img, g_img_fold, g_img2, dyo))
            wgrad_side(net.c_out, self.acts[-1], dyo, True)
            g_cur = net.gbuf("g_h4", B, H, W, g, 3)
            prog.add(net.c_out.dgrad(dyo, g_cur, padded_domain=rf))
            g_fold = rf
        else:
            a = self.acts[i]
            g_cur = net.gbuf(f"g_act{a.H}x{a.C}", B, a.H, a.W, a.C, 0)
            prog.add(ops.zero_(g_cur.t))
        # ---- upsampling layers
        while i >= 3 + nb:
            j = i - (3 + nb)
            hook(i, g_cur)
            a_in = self.acts[i - 1]
            dy = net.gbuf(f"dy_up{j}", B, self.raw[i].H, self.raw[i].W, self.raw[i].C, 1)
            inbwd(self.raw[i], self.stats[i], ACT_RELU, g_cur, g_fold, dy, net.c_up[j])
            wgrad_side(net.c_up[j], a_in, dy, False)
            g_cur = net.gbuf(f"g_act{a_in.H}x{a_in.C}", B, a_in.H, a_in.W, a_in.C, 0)
            prog.add(net.c_up[j].dgrad(dy, g_cur))
            g_fold = False
            i -= 1
        # ---- residual blocks
        while i >= 3:
            k = i - 3
            hook(i, g_cur)
            ca, cb = net.c_blk[k]
            ra, rb = self.raw[i]
            sa, sb = self.stats[i]
            c4, h4, w4 = ra.C, ra.H, ra.W
            dyb = net.gbuf(f"dy_blk_b{k % 2}", B, h4, w4, c4, 2)   # two sets, alternating: the side stream reads them one block late
            before_write(dyb)
            g_mid = net.gbuf("g_blk_p", B, h4, w4, c4, 1)
            a_parts = None
            if net.fp8:
                # input gradients on e4m3 operands: the norm backward leaves max|dY| per image, the copy is scaled by it
                dy8, am8, sc8 = self._fp8_grad_bufs(B, h4, w4, c4)
                inbwd(rb, sb, ACT_NONE, g_cur, False, dyb, cb, amax=am8)
                wgrad_side(cb, self.mid[k], dyb, False)
                prog.add(ops.quantize_fp8(dyb, dy8, am8, sc8))
                prog.add(cb.dgrad8(dy8, g_mid, sc8, padded_domain=rf))
            else:
                inbwd(rb, sb, ACT_NONE, g_cur, False, dyb, cb)
                wgrad_side(cb, self.mid[k], dyb, False)
                if use_chain:      # the epilogue also sums for the ReLU'd norm, whose saved output mid[k] carries the reflect halo
                    pa = ctx.scratch("bwd_parts_a", B * IN_WS_CHUNKS * c4 * 2)
                    prog.add(cb.dgrad(dyb, g_mid, padded_domain=True, chain={"operand": self.mid[k], "ws": pa}))
                    a_parts = (pa, cb.chain_parts, 1)
                else:
                    prog.add(cb.dgrad(dyb, g_mid, padded_domain=rf))
            dya = net.gbuf(f"dy_blk_a{k % 2}", B, h4, w4, c4, 2)
            before_write(dya)
            g_in_p = net.gbuf("g_blk_p", B, h4, w4, c4, 1)
            if net.fp8:
                inbwd(ra, sa, net.block_act, g_mid, rf, dya, ca, amax=am8)
                wgrad_side(ca, self.acts[i - 1], dya, False)
                prog.add(ops.quantize_fp8(dya, dy8, am8, sc8))
                prog.add(ca.dgrad8(dy8, g_in_p, sc8, padded_domain=rf))
            else:
                inbwd(ra, sa, net.block_act, g_mid, rf, dya, ca, parts=a_parts)
                wgrad_side(ca, self.acts[i - 1], dya, False)
                prog.add(ca.dgrad(dya, g_in_p, padded_domain=rf))
            g_next = net.gbuf(f"g_res{k % 2}", B, h4, w4, c4, 0)
            prog.add(ops.fold_add(g_cur, g_in_p, rf, g_next))
            g_cur = g_next
            if bucket is not None and k == bucket[0]:
                # Gradient bucket for data parallelism: every layer from residual block k to the output has its final gradient once
                # the launches queued so far have run -- weight gradients on the side stream, bias sums on this one.  The callback
                # (the trainer's) starts the all-reduce of that slice of the flat gradient behind BOTH, so that it overlaps the rest
                # of this backward; the side stream is only used to carry the two conditions, it does not wait for the collective.
                if bias_items:
                    prog.add(ops.bias_finalize_batch(list(bias_items)))
                    del bias_items[:]
                ev_b = ops.new_event()
                prog.add(ops.record(ev_b))
                prog.add(side.wait(ev_b))
                prog.add(bucket[1])
            i -= 1
        # ---- downsampling layers
        while i >= 1:
            hook(i, g_cur)
            a_in = self.acts[i - 1]
            dy = net.gbuf(f"dy_down{i}", B, self.raw[i].H, self.raw[i].W, self.raw[i].C, 1)
            inbwd(self.raw[i], self.stats[i], ACT_RELU, g_cur, False, dy, net.c_down[i - 1])
            wgrad_side(net.c_down[i - 1], a_in, dy, False)
            g_cur = net.gbuf(f"g_act{a_in.H}x{a_in.C}", B, a_in.H, a_in.W, a_in.C, 0)
            prog.add(net.c_down[i - 1].dgrad(dy, g_cur))
            i -= 1
        hook(0, g_cur)
        dy0 = net.gbuf("dy_init", B, H, W, g, 6 if need_input_grad else 0)
        inbwd(self.raw[0], self.stats[0], ACT_RELU, g_cur, False, dy0, net.c_init)
        wgrad_side(net.c_init, self.x0, dy0, False)
        self.g_input = None
        if need_input_grad:
            self.g_input = net.gbuf("g_x0", B, H, W, self.x0.C, 3)   # reflect: padded domain, the consumer folds (g_input_fold)
            prog.add(net.c_init.dgrad(dy0, self.g_input, padded_domain=rf))
        self.g_input_fold = rf
        if bias_items:
            prog.add(ops.bias_finalize_batch(bias_items))
        join = ops.new_event()       # everything after this program (next pass, all-reduce, optimiser) sees complete gradients
        prog.add(side.record(join))
        prog.add(ops.wait(join))
        return prog


class DiscriminatorNet(_Net):
    """PatchGAN discriminator: CUT (conv+bias+LeakyReLU, no norm) or Basic_GAN (InstanceNorm after convs 2-4)."""

    def __init__(self, ctx, params, grads, style="cut", prefix="discriminators.0.model.", ndf=64, n_layers=3, in_c=3):
        super().__init__(ctx, params, grads)
        self.style, self.in_c = style, in_c
        nconv = n_layers + 2
        if style == "cut":
            keys = [f"{prefix}{2*i}" for i in range(nconv)]
        else:
            keys = ["net.0"] + [f"net.{2+3*i}" for i in range(n_layers)] + [f"net.{2+3*n_layers}"]
        self.chans = [in_c, ndf] + [ndf * min(2**n, 8) for n in range(1, n_layers)] + [ndf * min(2**n_layers, 8), 1]
        self.strides = [2] * n_layers + [1, 1]
        self.convs = [self._conv(k, 4, s, 1) for k, s in zip(keys, self.strides)]
        self.nconv = nconv

    def new_pass(self, B, H, W) -> "DPass":
        return DPass(self, B, H, W)


class DPass:
    def __init__(self, net: DiscriminatorNet, B, H, W):
        self.net, self.B, self.H, self.W = net, B, H, W
        ctx = net.ctx
        self.x = ctx.view(B, H, W, cpad(net.in_c), 1)     # zero halo 1; filled by DiffAugment / layout conversion
        self.acts, self.raw, self.stats = [], [], []
        h, w = H, W
        for li in range(net.nconv):
            s = net.strides[li]
            h, w = (h + 2 - 4) // s + 1, (w + 2 - 4) // s + 1
            c = cpad(net.chans[li + 1])
            last = li == net.nconv - 1
            self.acts.append(ctx.view(B, h, w, c, 0 if last else 1))
            normed = net.style == "basic" and 0 < li < net.nconv - 1
            self.raw.append(ctx.view(B, h, w, c, 0) if normed else None)
            self.stats.append(ctx.f32(B * c * 2) if normed else None)
        self.logits = self.acts[-1]

    def fwd_program(self) -> Program:
        net, ops = self.net, self.net.ctx.ops
        prog = Program("D.fwd")
        xin = self.x
        for li, conv in enumerate(net.convs):
            last = li == net.nconv - 1
            if self.raw[li] is not None:
                prog.add(conv.fwd(xin, self.raw[li]))
                prog.add(ops.in_stats(self.raw[li], IN_EPS, self.stats[li], net.in_ws(self.B, self.raw[li].C)))
                prog.add(ops.in_apply(self.raw[li], self.stats[li], ACT_LRELU, None, self.acts[li], HALO_ZERO))
            else:
                prog.add(conv.fwd(xin, self.acts[li], ACT_NONE if last else ACT_LRELU))
            xin = self.acts[li]
        return prog

    def r1_program(self, scale: float, loss: torch.Tensor, scratch: torch.Tensor) -> Program:
        """R1 penalty (train_cutpp.py:165-203) after this pass's forward: *loss = mean_b sum_chw (d sum D(x) / dx)^2 and the
        weight gradients of scale * r1, as an explicit second-order program (no autograd graph):
        first-order input gradient with the LeakyReLU masks fused in the dgrad epilogues (delta_i kept per layer), then the
        linearised forward u_i = mask_i * (W_i * u_{i-1}) seeded with u_0 = scale * 2 g / B, with dW_i = wgrad(u_{i-1}, delta_i).
        Bias gradients are zero except the last bias, whose gradient is None in the reference (the caller skips it)."""
        ctx, ops = self.net.ctx, self.net.ctx.ops
        pr = Program("R1")
        pr.add(self.r1_first(scratch))
        u = ctx.view(self.B, self.H, self.W, self.x.C, 1)
        pr.add(ops.r1_reduce(self.g_input, self.net.in_c, scale, loss, u, ctx.scratch("r1_ws", 1024)))
        pr.add(self.r1_second(u))
        return pr

    def r1_first(self, scratch: torch.Tensor) -> Program:
        """First-order half of R1: g = d sum D(x) / dx into self.g_input, the per-layer output gradients kept for r1_second."""
        net, ops, B = self.net, self.net.ctx.ops, self.B
        assert net.style == "cut", "R1 is part of the CUT trainer (no norm layers in its discriminator)"
        pr = Program("R1.first")
        lg = self.logits
        ones = net.gbuf("r1_ones", B, lg.H, lg.W, lg.C, 2)
        pr.add(ops.patch_loss(lg, 2, 0.0, -float(B * lg.H * lg.W), scratch, ones))   # d(sum D)/dlogits = 1
        self._r1_deltas: List[View] = []
        pr.add(self.bwd_program(ones, wgrad=False, need_input_grad=True, keep=self._r1_deltas))
        return pr

    def r1_second(self, u: View) -> Program:
        """Second-order half: the weight gradients of <u_0, g(theta)> by the linearised forward from u_0 = `u` (zero halo 1).
        With several scales u_0 is the pooled total input gradient, not this scale's own (autograd.py, _r1_slot)."""
        net, ctx, B = self.net, self.net.ctx, self.B
        pr = Program("R1.second")
        deltas = self._r1_deltas
        for li, conv in enumerate(net.convs):
            delta = deltas[len(deltas) - 1 - li]
            pr.add(conv.wgrad(u, delta, accumulate=False, bias_too=False))
            if li == net.nconv - 1:
                break
            a = self.acts[li]
            nxt = ctx.view(B, a.H, a.W, a.C, 1)
            pr.add(conv.fwd(u, nxt, ACT_NONE, mask=a, use_bias=False))
            u = nxt
        return pr

    def grad_logits_view(self) -> View:
        """Where the loss writes dL/dlogits: zero halo 2 (= k-1-p of the last 4x4 s1 p1 conv's input gradient)."""
        lg = self.logits
        return self.net.gbuf("g_logits", self.B, lg.H, lg.W, lg.C, 2)

    def bwd_program(self, g_logits: View, wgrad: bool = True, accumulate: bool = False, need_input_grad: bool = False,
                    keep: Optional[list] = None, bias_grads: bool = True) -> Program:
        """Backward from dL/dlogits.  keep: if a list, the per-layer output gradients (delta_i) are appended to it and
        live in dedicated buffers (R1's second-order pass needs them)."""
        net, ops, B = self.net, self.net.ctx.ops, self.B
        prog = Program("D.bwd")
        dy = g_logits
        self.g_input = None
        # The discriminator's weight gradients stay on its own stream: they are 2 % of the step, and a fourth compute stream would
        # share a hardware queue with another one as soon as RCCL adds its stream (HIP multiplexes streams onto four queues;
        # measured: the discriminator's stream landed on the main stream's queue and the step lost 1.1 ms)
        side = ops.side() if (wgrad and keep is None and os.environ.get("GAN_D_SIDE")) else None
        for li in range(net.nconv - 1, -1, -1):
            conv = net.convs[li]
            xin = self.acts[li - 1] if li > 0 else self.x
            if keep is not None:
                keep.append(dy)
            if wgrad and side is not None:
                ev = ops.new_event()
                prog.add(ops.record(ev))
                prog.add(side.wait(ev))
                prog.add(conv.wgrad(xin, dy, accumulate, bias_too=bias_grads, ops=side))
            elif wgrad:
                prog.add(conv.wgrad(xin, dy, accumulate, bias_too=bias_grads))
            if li == 0:
                if need_input_grad:
                    self.g_input = net.gbuf("g_dx", B, self.H, self.W, self.x.C, 0)
                    prog.add(conv.dgrad(dy, self.g_input))
                break
            prev = net.convs[li - 1]
            halo = 2 if prev.s == 1 else 1   # what the previous conv's input gradient needs from its dY
            a = self.acts[li - 1]
            tag = f"d_act{li-1}" + ("_keep" if keep is not None else "")
            if self.raw[li - 1] is None:
                nxt = net.gbuf(tag, B, a.H, a.W, a.C, halo)
                prog.add(conv.dgrad(dy, nxt, mask=a))             # LeakyReLU' fused in the epilogue
            else:
                g_a = net.gbuf(f"g_dact{li-1}", B, a.H, a.W, a.C, 0)
                prog.add(conv.dgrad(dy, g_a))
                nxt = net.gbuf(tag, B, a.H, a.W, a.C, halo)
                prog.add(ops.in_bwd(self.raw[li - 1], self.stats[li - 1], ACT_LRELU, g_a, False, None, nxt, net.in_ws(B, a.C)))
            dy = nxt
        if side is not None:
            join = ops.new_event()
            prog.add(side.record(join))
            prog.add(ops.wait(join))
        return prog
