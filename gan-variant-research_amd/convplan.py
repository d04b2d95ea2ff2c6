"""Geometry of one nn.Conv2d / nn.ConvTranspose2d on the generalised tap-convolution kernels.

A ConvLayer owns the packed operand copies of one fp32 master weight (forward order and gradient order) and
turns forward / input-gradient / weight-gradient requests on halo-NHWC views into ConvCall / WgradCall
descriptors: which taps, which halo offsets, which sub-pixel phase.  All shape reasoning of the reference's
convolutions (GAN_Variant1/models/generator_resnet_attn.py:33,48,113,125,146-149,160;
discriminator_patchgan.py:27-51; Basic_GAN/src/models.py:12-103) lives here; the kernels only see descriptors.
"""
from __future__ import annotations

import os

from typing import List, Optional, Tuple

import torch

from ._lib import ACT_NONE, BF16, F32, FP8, GanError
from .runtime import ConvCall, Ctx, View, WgradCall, cpad


def _nw(n_real: int) -> int:
    if n_real <= 16:
        return 16
    if n_real <= 64:
        return 64
    return (n_real + 127) // 128 * 128


class _Pack:
    """One packed operand copy [Nw][ntaps_pad][Cred] of a master weight plus its tap list."""

    def __init__(self, ctx: Ctx, taps: List[Tuple[int, int, int]], n_real: int, c_real: int, swap: bool, i2: int, kk: int):
        self.ctx, self.taps, self.n_real, self.c_real, self.swap, self.i2, self.kk = ctx, taps, n_real, c_real, swap, i2, kk
        self.cred = cpad(c_real)
        bke = 32 if ctx.dtype == F32 else 64
        nt = len(taps)
        while (nt * self.cred) % bke:
            nt += 1
        self.ntaps = nt
        self.nw = _nw(n_real)
        self.khw = ctx.i32([t[2] for t in taps] + [-1] * (nt - len(taps)))
        self._w = None     # row-major copy (generic kernel), allocated on first use
        self._wf = None    # fragment-major copy (range-patch kernel), allocated on first use
        self._wf8 = None   # fragment-major e4m3 copy (range-patch kernel, fp8 operands) and its per-tensor scale max|W| / 448
        self.scale8 = None
        self._tapoff = {}

    @property
    def w(self) -> torch.Tensor:
        if self._w is None:
            self._w = torch.zeros(self.nw * self.ntaps * self.cred, dtype=self.ctx.tdtype, device=self.ctx.device)
        return self._w

    @property
    def wf(self) -> torch.Tensor:
        if self._wf is None:
            self._wf = torch.zeros(self.nw * self.ntaps * self.cred, dtype=self.ctx.tdtype, device=self.ctx.device)
        return self._wf

    @property
    def wf8(self) -> torch.Tensor:
        if self._wf8 is None:
            self._wf8 = torch.zeros(self.nw * self.ntaps * self.cred, dtype=torch.uint8, device=self.ctx.device)
            self.scale8 = torch.ones(1, dtype=torch.float32, device=self.ctx.device)
        return self._wf8

    def finalize8(self, c: ConvCall) -> ConvCall:
        """fp8 operands (c.x is an e4m3 view): the range-patch kernel or nothing -- there is no other fp8 convolution."""
        assert c.x.dtype == FP8
        c.w, c.w_frag, c.w_scale = self.wf8, True, self.scale8
        if not self.ctx.ops.conv_patch_ok(c):
            raise GanError(f"fp8 convolution: B{c.B} {c.Ho}x{c.Wo} Cin{c.Cin} taps{c.ntaps} does not qualify for the range-patch kernel")
        c.tile_rows = self.ctx.ops.conv_patch_tile_rows(c)
        c.tile_cols = self.ctx.ops.conv_patch_tile_cols(c)
        return c

    def pack_ops(self, master: torch.Tensor):
        """Refreshes every operand copy that some planned call uses (plan all calls BEFORE building the repack program)."""
        out = []
        if self._wf8 is not None:
            out.append(self.ctx.ops.pack_weight(master, self._wf8, FP8, self.nw, self.ntaps, self.cred, self.n_real, self.c_real, self.swap,
                                                self.i2, self.kk, self.khw, 1, self.scale8))
        for buf, layout in ((self._w, 0), (self._wf, 1)):
            if buf is not None:
                out.append(self.ctx.ops.pack_weight(master, buf, self.ctx.dtype, self.nw, self.ntaps, self.cred, self.n_real, self.c_real,
                                                    self.swap, self.i2, self.kk, self.khw, layout))
        return out

    def finalize(self, c: ConvCall) -> ConvCall:
        """Picks the kernel for a planned call: range-patch (fragment-major weight copy) if it qualifies, else generic."""
        c.w = None
        if self.ctx.ops.conv_patch_ok(c):
            c.w, c.w_frag = self.wf, True
            c.tile_rows = self.ctx.ops.conv_patch_tile_rows(c)
            c.tile_cols = self.ctx.ops.conv_patch_tile_cols(c)
        else:
            c.w, c.w_frag = self.w, False
            # the 64 -> 3 channel 7x7 layers: the window kernel reads the same row-major weight copy
            rowmajor7 = len(self.taps) == 49 and all(t[:2] == (i // 7, i % 7) for i, t in enumerate(self.taps))
            if rowmajor7 and self.ctx.ops.conv_win7_ok(c, 0, 0):
                c.win7 = (0, 0)
        return c

    def tapoff(self, wp: int) -> torch.Tensor:
        t = self._tapoff.get(wp)
        if t is None:
            t = self.ctx.i32([(dy * wp + dx) * self.cred for dy, dx, _ in self.taps] + [0] * (self.ntaps - len(self.taps)))
            self._tapoff[wp] = t
        return t

    def max_tapoff(self, wp: int) -> int:
        return max((dy * wp + dx) * self.cred for dy, dx, _ in self.taps)


class _PairPack(_Pack):
    """The two x-phases (rx = 0, 1) of one phase row of a transposed convolution / strided input gradient as ONE launch, for layers
    with 64 output channels: output pixels (2a+ry, 2b) and (2a+ry, 2b+1) are neighbours in memory, so the pair is a 128-channel
    "super-pixel" and the launch is a stride-1 convolution with N = 128 over the union of the two phases' taps -- wide enough for the
    range-patch kernel, which the 64-wide phases were not (they ran on the generic kernel at 160-400 TFLOP/s).  Rows 0..63 of the
    packed weight hold phase rx = 0, rows 64..127 phase rx = 1, zero where a phase does not use a tap of the union."""

    def __init__(self, ctx: Ctx, pa: _Pack, pb: _Pack):
        assert pa.phase[0] == pb.phase[0] and pa.dmin[0] == pb.dmin[0] and pa.n_real == pb.n_real == 64
        dmy, dmx = pa.dmin[0], min(pa.dmin[1], pb.dmin[1])
        abs_taps = lambda pk: {(dy + pk.dmin[0], dx + pk.dmin[1]): kidx for dy, dx, kidx in pk.taps}
        ta, tb = abs_taps(pa), abs_taps(pb)
        union = sorted(set(ta) | set(tb))
        super().__init__(ctx, [(dy - dmy, dx - dmx, ta.get((dy, dx), -1)) for dy, dx in union], 128, pa.c_real, pa.swap, pa.i2, pa.kk)
        self.n_half = pa.n_real
        pad = [-1] * (self.ntaps - len(union))
        self.khw_halves = [ctx.i32([ta.get(t, -1) for t in union] + pad), ctx.i32([tb.get(t, -1) for t in union] + pad)]
        self.dmin, self.dmax = (dmy, dmx), (max(pa.dmax[0], pb.dmax[0]), max(pa.dmax[1], pb.dmax[1]))
        self.phase_row = pa.phase[0]

    def pack_ops(self, master: torch.Tensor):
        out, half = [], 64 * self.ntaps * self.cred       # 64 rows = four 16-row fragment groups: the same element count in both layouts
        for buf, layout in ((self._w, 0), (self._wf, 1)):
            if buf is not None:
                for h, khw in enumerate(self.khw_halves):
                    out.append(self.ctx.ops.pack_weight(master, buf[h * half:(h + 1) * half], self.ctx.dtype, 64, self.ntaps, self.cred, self.n_half,
                                                        self.c_real, self.swap, self.i2, self.kk, khw, layout))
        return out


def _phase_taps(k: int, p: int, r: int):
    """Transposed-conv / strided-dgrad taps of output parity r: [(kh, d)] with source index a + d."""
    return [(kh, (r + p - kh) // 2) for kh in range(k) if (r + p - kh) % 2 == 0]


class ConvLayer:
    def __init__(self, ctx: Ctx, weight: torch.Tensor, bias: Optional[torch.Tensor], grad_w: torch.Tensor, grad_b: Optional[torch.Tensor],
                 k: int, stride: int, pad: int, transposed: bool = False, need_dgrad: bool = True, need_wgrad: bool = True):
        self.ctx, self.weight, self.bias, self.grad_w, self.grad_b = ctx, weight, bias, grad_w, grad_b
        self.k, self.s, self.p, self.transposed = k, stride, pad, transposed
        if transposed:
            assert (k, stride, pad) == (3, 2, 1), "ConvTranspose2d is supported as k3 s2 p1 output_padding 1"
            self.cin, self.cout = weight.shape[0], weight.shape[1]
        else:
            assert stride in (1, 2)
            self.cout, self.cin = weight.shape[0], weight.shape[1]
        kk = k * k
        self.kk = kk
        alltaps = [(kh, kw, kh * k + kw) for kh in range(k) for kw in range(k)]
        self.packs: List[_Pack] = []
        if not transposed:
            self.fwd_pack = _Pack(ctx, alltaps, self.cout, self.cin, False, self.cin, kk)
            self.packs.append(self.fwd_pack)
            self.dgrad_packs = None
            if need_dgrad:
                if stride == 1:  # flipped taps over a zero-haloed dY
                    taps = [(a, b, (k - 1 - a) * k + (k - 1 - b)) for a in range(k) for b in range(k)]
                    self.dgrad_packs = [_Pack(ctx, taps, self.cin, self.cout, True, self.cin, kk)]
                else:
                    self.dgrad_packs = self._phase_packs(self.cin, self.cout, self.cin)
                self.packs += self.dgrad_packs
        else:
            self.fwd_packs = self._phase_packs(self.cout, self.cin, self.cout)
            self.packs += self.fwd_packs
            self.dgrad_pack = None
            if need_dgrad:  # strided conv over dY with the weight read as [out=cin][in=cout]
                self.dgrad_pack = _Pack(ctx, alltaps, self.cin, self.cout, False, self.cout, kk)
                self.packs.append(self.dgrad_pack)
        self.need_wgrad = need_wgrad
        # the epilogue reads bias[n] for every stored (padded) channel: keep a zero-padded copy when cout is not a multiple of 8
        self.bias_k = bias
        self._one = ctx.i32([0])
        if bias is not None and cpad(self.cout) != self.cout:
            self.bias_k = torch.zeros(_nw(self.cout), dtype=torch.float32, device=ctx.device)
        self.wg_khw = ctx.i32([t[2] for t in alltaps])
        self._wg_tapoff = {}
        self._pairs = {}
        self.bias_pair = None     # [bias | bias] for paired phases (allocated with the first paired plan)

    def _phase_packs(self, n_real, c_real, i2):
        """Four sub-pixel phases (ry, rx); index [ry*2+rx] -> (_Pack, dmin_y, dmin_x)."""
        out = []
        for ry in range(2):
            ty = _phase_taps(self.k, self.p, ry)
            for rx in range(2):
                tx = _phase_taps(self.k, self.p, rx)
                dmy, dmx = min(d for _, d in ty), min(d for _, d in tx)
                taps = [(dy - dmy, dx - dmx, kh * self.k + kw) for kh, dy in ty for kw, dx in tx]
                pk = _Pack(self.ctx, taps, n_real, c_real, True, i2, self.kk)
                pk.dmin, pk.dmax = (dmy, dmx), (max(d for _, d in ty), max(d for _, d in tx))
                pk.phase = (ry, rx)
                out.append(pk)
        return out

    # ------------------------------------------------------------------ weights
    def repack_ops(self):
        out = [op for pk in self.packs for op in pk.pack_ops(self.weight)]
        if not out:
            # the operand copies are allocated when a call is planned: a repack program built before any fwd / dgrad plan would pack
            # nothing and the launches planned afterwards would multiply by zeros (measured that way, they also run ~20 % too fast:
            # zero operands draw less power and the chip holds a higher clock)
            raise GanError("ConvLayer.repack_ops: no operand copy is planned yet -- plan the forward / input-gradient calls first, then build the repack program")
        if self.bias_k is not self.bias:
            out.append(self.ctx.ops.pack_weight(self.bias, self.bias_k, F32, self.bias_k.numel(), 1, 1, self.cout, 1, False, 1, 1, self._one))
        if self.bias_pair is not None:
            for h in range(2):
                out.append(self.ctx.ops.pack_weight(self.bias, self.bias_pair[64 * h:64 * h + 64], F32, 64, 1, 1, self.cout, 1, False, 1, 1, self._one))
        return out

    # ------------------------------------------------------------------ forward
    def fwd(self, x: View, y: View, act: int = ACT_NONE, mask: Optional[View] = None, use_bias: bool = True,
            stats_ws: Optional[torch.Tensor] = None):
        """Forward launches.  stats_ws: if given and the launch can, its epilogue also writes the InstanceNorm partials of `y`
        there; self.stats_parts then holds their count per image (0: not fused, the caller runs in_stats)."""
        assert x.C == cpad(self.cin) and y.C == cpad(self.cout) and x.B == y.B, (x.C, self.cin, y.C, self.cout)
        ops = self.ctx.ops
        self.stats_parts = 0
        if not self.transposed:
            k, s, p = self.k, self.s, self.p
            ho, wo = (x.H + 2 * p - k) // s + 1, (x.W + 2 * p - k) // s + 1
            assert (ho, wo) == (y.H, y.W) and x.halo >= p, (ho, wo, y.H, y.W, x.halo, p)
            pk = self.fwd_pack
            call = pk.finalize(ConvCall(x.B, ho, wo, pk.cred, pk.ntaps, pk.nw, min(pk.nw, y.C), x, x.halo - p, x.halo - p, s, s,
                                        pk.tapoff(x.Wp), None, self.bias_k if use_bias else None, y, y.halo, y.halo, 1, 1, act, mask,
                                        mask.halo if mask else 0, mask.halo if mask else 0, pk.max_tapoff(x.Wp)))
            if stats_ws is not None and (call.w_frag or call.win7 is not None) and call.Nst == y.C and not os.environ.get("GAN_NO_FUSED_STATS"):
                n = ops.conv_stats_parts(call)
                if 0 < n and x.B * n * y.C * 2 <= stats_ws.numel():
                    call.stats, self.stats_parts = stats_ws, n
            self.last_call = call          # the planned forward call (tests read the tile the planner chose)
            return [ops.conv_igemm(call)]
        assert (y.H, y.W) == (2 * x.H, 2 * x.W)
        return self._phased(self.fwd_packs, x, y, act, self.bias_k if use_bias else None, mask)

    def fwd8(self, x8: View, y: View, act: int = ACT_NONE, stats_ws: Optional[torch.Tensor] = None, in_scale: Optional[torch.Tensor] = None):
        """Forward on e4m3 operands (x8 = gan_quantize_fp8 copy of the input, unit scale unless in_scale; the weight copy is this
        layer's fp8 pack); the result `y` is bf16.  Stride-1 convolutions on the range-patch kernel (the residual 3x3 256->256 layers)."""
        assert not self.transposed and self.s == 1 and x8.dtype == FP8 and y.dtype == BF16 and x8.C == cpad(self.cin) and y.C == cpad(self.cout)
        ops, k, p = self.ctx.ops, self.k, self.p
        assert (y.H, y.W) == (x8.H + 2 * p - k + 1, x8.W + 2 * p - k + 1) and x8.halo >= p
        pk = self.fwd_pack
        call = pk.finalize8(ConvCall(x8.B, y.H, y.W, pk.cred, pk.ntaps, pk.nw, min(pk.nw, y.C), x8, x8.halo - p, x8.halo - p, 1, 1, pk.tapoff(x8.Wp), None,
                                     self.bias_k, y, y.halo, y.halo, 1, 1, act, None, 0, 0, pk.max_tapoff(x8.Wp), in_scale=in_scale))
        self.stats_parts = 0
        if stats_ws is not None and call.Nst == y.C and not os.environ.get("GAN_NO_FUSED_STATS"):
            n = ops.conv_stats_parts(call)
            if 0 < n and x8.B * n * y.C * 2 <= stats_ws.numel():
                call.stats, self.stats_parts = stats_ws, n
        return [ops.conv_igemm(call)]

    def dgrad8(self, dy8: View, dx: View, in_scale: torch.Tensor, padded_domain: bool = False):
        """Input gradient on e4m3 operands: dy8 = per-image-scaled e4m3 copy of dY (zero halo), in_scale its scales; dx is bf16."""
        assert not self.transposed and self.s == 1 and dy8.dtype == FP8 and dx.dtype == BF16
        ops, k, p = self.ctx.ops, self.k, self.p
        pk = self.dgrad_packs[0]
        if padded_domain:
            assert dx.halo == p and dy8.halo >= k - 1
            gh, gw, oy, iy = dx.Hp, dx.Wp, 0, dy8.halo - (k - 1)
        else:
            assert dy8.halo >= k - 1 - p
            gh, gw, oy, iy = dx.H, dx.W, dx.halo, dy8.halo - (k - 1) + p
        call = pk.finalize8(ConvCall(dy8.B, gh, gw, pk.cred, pk.ntaps, pk.nw, min(pk.nw, dx.C), dy8, iy, iy, 1, 1, pk.tapoff(dy8.Wp), None, None, dx,
                                     oy, oy, 1, 1, ACT_NONE, None, 0, 0, pk.max_tapoff(dy8.Wp), in_scale=in_scale))
        call.alg_pixels = dy8.H * dy8.W
        return [ops.conv_igemm(call)]

    def _pair_packs(self, packs):
        """Lazily built _PairPack per phase row (see there); None when the layer does not qualify."""
        key = id(packs)
        if key not in self._pairs:
            ok = (self.ctx.dtype == BF16 and packs[0].n_real == 64 and packs[0].cred % 64 == 0 and not os.environ.get("GAN_NO_PHASE_PAIRS"))
            self._pairs[key] = [_PairPack(self.ctx, packs[2 * ry], packs[2 * ry + 1]) for ry in range(2)] if ok else None
            if ok:
                self.packs += self._pairs[key]
        return self._pairs[key]

    def _phased(self, packs, src: View, dst: View, act, bias, mask):
        """dst[2a+r] = sum_taps src[a + d] * w: used by ConvTranspose2d forward and by the strided conv's input gradient."""
        ops, out = self.ctx.ops, []
        gh, gw = dst.H // 2, dst.W // 2
        pairs = self._pair_packs(packs) if (dst.halo == 0 and mask is None and dst.C == 64) else None
        if pairs is not None:
            # the destination as rows of 128-channel super-pixels (same storage)
            sup = View(dst.t, dst.B, dst.H, gw, 128, 0, dst.dtype)
            if bias is not None and self.bias_pair is None:
                self.bias_pair = torch.zeros(128, dtype=torch.float32, device=self.ctx.device)
            calls = []
            for pk in pairs:
                (dmy, dmx), (dxy, dxx) = pk.dmin, pk.dmax
                assert src.halo + dmy >= 0 and src.halo + dmx >= 0 and gh - 1 + dxy <= src.H - 1 + src.halo and gw - 1 + dxx <= src.W - 1 + src.halo
                calls.append(pk.finalize(ConvCall(src.B, gh, gw, pk.cred, pk.ntaps, 128, 128, src, src.halo + dmy, src.halo + dmx, 1, 1, pk.tapoff(src.Wp), None,
                                                  self.bias_pair if bias is not None else None, sup, pk.phase_row, 0, 2, 1, act, None, 0, 0, pk.max_tapoff(src.Wp))))
            if all(c.w_frag for c in calls):          # both rows on the range-patch kernel, otherwise keep the four single phases
                for c, pk in zip(calls, pairs):           # useful share of the launched FLOPs: taps each half really has / (2 x union)
                    c.flop_scale = sum(int(k) >= 0 for khw in pk.khw_halves for k in khw.tolist()) / (2.0 * len(pk.taps))
                return [ops.conv_igemm(c) for c in calls]
        for pk in packs:
            (ry, rx), (dmy, dmx), (dxy, dxx) = pk.phase, pk.dmin, pk.dmax
            assert src.halo + dmy >= 0 and src.halo + dmx >= 0, "source halo too small (top/left)"
            assert gh - 1 + dxy <= src.H - 1 + src.halo and gw - 1 + dxx <= src.W - 1 + src.halo, "source halo too small (bottom/right)"
            out.append(ops.conv_igemm(pk.finalize(ConvCall(src.B, gh, gw, pk.cred, pk.ntaps, pk.nw, min(pk.nw, dst.C), src, src.halo + dmy, src.halo + dmx,
                                               1, 1, pk.tapoff(src.Wp), None, bias, dst, dst.halo + ry, dst.halo + rx, 2, 2, act, mask,
                                               (mask.halo + ry) if mask else 0, (mask.halo + rx) if mask else 0, pk.max_tapoff(src.Wp)))))
        return out

    # ------------------------------------------------------------------ input gradient
    def dgrad(self, dy: View, dx: View, mask: Optional[View] = None, padded_domain: bool = False, chain: Optional[dict] = None):
        """dx <- dL/d(input).  padded_domain: also produce the gradient on the input's (reflect) halo; the consumer folds it.
        chain (stride-1 layers on the padded domain, bf16) = {"operand": y, "ws": parts}: the launch's epilogue also leaves the two sums of
        the InstanceNorm backward behind a ReLU whose saved output is y (reflect halo p) -- gan_conv_desc.stats_mode 1; self.chain_parts =
        partials per image written to ws.  Raises GanError if the launch cannot carry it (ask can_chain first)."""
        assert dy.C == cpad(self.cout) and dx.C == cpad(self.cin) and dy.B == dx.B
        ops, k, p = self.ctx.ops, self.k, self.p
        self.chain_parts = 0
        assert chain is None or (self.s == 1 and not self.transposed and padded_domain and mask is None)
        if self.transposed:  # regular strided conv over dy
            pk = self.dgrad_pack
            assert dy.halo >= p and (dy.H, dy.W) == (2 * dx.H, 2 * dx.W) and not padded_domain
            return [ops.conv_igemm(pk.finalize(ConvCall(dy.B, dx.H, dx.W, pk.cred, pk.ntaps, pk.nw, min(pk.nw, dx.C), dy, dy.halo - p, dy.halo - p, 2, 2,
                                            pk.tapoff(dy.Wp), None, None, dx, dx.halo, dx.halo, 1, 1, ACT_NONE, mask,
                                            mask.halo if mask else 0, mask.halo if mask else 0, pk.max_tapoff(dy.Wp))))]
        if self.s == 1:
            pk = self.dgrad_packs[0]
            if padded_domain:
                assert dx.halo == p and dy.halo >= k - 1 and mask is None
                gh, gw, oy, iy = dx.Hp, dx.Wp, 0, dy.halo - (k - 1)
            else:
                assert dy.halo >= k - 1 - p
                gh, gw, oy, iy = dx.H, dx.W, dx.halo, dy.halo - (k - 1) + p
            assert gh == dy.H + k - 1 - (0 if padded_domain else 2 * p), (gh, dy.H, k, p)
            call = pk.finalize(ConvCall(dy.B, gh, gw, pk.cred, pk.ntaps, pk.nw, min(pk.nw, dx.C), dy, iy, iy, 1, 1, pk.tapoff(dy.Wp), None,
                                        None, dx, oy, oy, 1, 1, ACT_NONE, mask, mask.halo if mask else 0, mask.halo if mask else 0, pk.max_tapoff(dy.Wp)))
            call.alg_pixels = dy.H * dy.W      # the reference op's M is the forward output (dy) pixel count, not the (padded) input domain
            if chain is not None:
                opd = chain["operand"]
                assert opd.C == dx.C and opd.B == dx.B and (opd.H, opd.W) == (dx.H, dx.W) and opd.halo == p
                call.mask, call.mask_y0, call.mask_x0 = opd, 0, 0
                call.stats, call.stats_mode = chain["ws"], 1
                n = ops.conv_stats_parts(call) if call.w_frag else 0
                if n <= 0 or dy.B * n * dx.C * 2 > chain["ws"].numel():
                    raise GanError(f"input gradient B{dy.B} {gh}x{gw} C{dx.C}: the launch cannot carry the backward chain (ask can_chain first)")
                self.chain_parts = n
            return [ops.conv_igemm(call)]
        assert not padded_domain and (dx.H, dx.W) == (2 * dy.H, 2 * dy.W)
        return self._phased(self.dgrad_packs, dy, dx, ACT_NONE, None, mask)

    def can_chain(self, dy: View, dx: View) -> bool:
        """True if dgrad(dy, dx, padded_domain=True, chain=...) is possible: a stride-1 layer in bf16 whose padded-domain input gradient
        runs on the range-patch kernel (its epilogue carries the sums and the addend)."""
        if self.transposed or self.s != 1 or self.ctx.dtype != BF16 or dx.halo != self.p or dy.halo < self.k - 1:
            return False
        try:      # planning only: dx stands in for the operand (same geometry), nothing is launched
            self.dgrad(dy, dx, padded_domain=True, chain={"operand": dx, "ws": self.ctx.scratch("bwd_parts_a", dx.B * 96 * dx.C * 2)})
        except GanError:
            return False
        return self.chain_parts > 0

    # ------------------------------------------------------------------ weight gradient
    def _nsplit(self, m: int, jtiles: int, ntiles: int, ntiles_skinny: bool = False) -> int:
        want = max(1, (1024 if ntiles_skinny else 512) // max(1, jtiles * ntiles))
        ns = max(1, min(want, m // 512))
        while ns > 1 and (ns - 1) * ((-(-m // ns) + 63) // 64 * 64) >= m:
            ns -= 1
        return ns

    def wgrad(self, x: View, dy: View, accumulate: bool, bias_too: bool = True, ops=None):
        """grad_w (+)= dL/dW from the layer input `x` and the output gradient `dy`.  ops: op layer to build the launches on
        (default: the context's; the backward programs pass the second-stream layer)."""
        ctx, k, p = self.ctx, self.k, self.p
        ops = ctx.ops if ops is None else ops
        epc = 4 if ctx.dtype == F32 else 8
        if not self.transposed:
            g, xo, stride = dy, x, self.s
            assert x.halo >= p
            n_real, c_real, i2 = self.cout, self.cin, self.cin
        else:  # roles swap: rows run over the layer input, the "x operand" is dy
            g, xo, stride = x, dy, 2
            assert dy.halo >= p
            n_real, c_real, i2 = self.cin, self.cout, self.cout
        n, cx = g.C, xo.C
        ktot = self.kk * cx
        m = g.B * g.H * g.W
        jt, ntl = 16 * epc, (16 if n <= 16 else (128 if ctx.dtype == BF16 else 64))
        key = (xo.Wp, cx)
        tapoff = self._wg_tapoff.get(key)
        if tapoff is None:
            tapoff = ctx.i32([(kh * xo.Wp + kw) * cx for kh in range(k) for kw in range(k)])
            self._wg_tapoff[key] = tapoff
        call = WgradCall(g.B, g.H, g.W, cx, self.kk, n, 1, xo, xo.halo - p, xo.halo - p, stride, stride, tapoff, g, g.halo, g.halo, 1, 1, None,
                         max_tapoff=((k - 1) * xo.Wp + (k - 1)) * cx)
        spi = ops.wgrad_patch_splits(call)
        w7 = ops.wgrad_win7_splits(call) if spi == 0 else 0
        if spi > 0:      # range-patch kernel: spi splits per image
            call.nsplit, call.variant = g.B * spi, 1
        elif spi < 0:    # ... or, on many small maps, -spi whole images per split
            assert g.B % -spi == 0
            call.nsplit, call.variant = g.B // -spi, 1
        elif w7 > 0:     # the 64 -> 3 channel 7x7 layer: one slab per persistent block
            call.nsplit, call.variant = w7, 2
        else:
            call.nsplit = self._nsplit(m, -(-ktot // jt), -(-n // ntl), n <= 16)
        ns = call.nsplit
        call.part = part = ctx.scratch("wgrad_part", ns * n * ktot)
        out = [ops.conv_wgrad(call),
               ops.wgrad_reduce(part, ns, n, self.kk, cx, n_real, c_real, False, i2, self.kk, self.wg_khw, self.grad_w, accumulate)]
        if bias_too and self.grad_b is not None:
            out.append(ops.bias_grad(dy, self.cout, self.grad_b, accumulate, ctx.scratch("bias_ws", 256 * max(dy.C, 256))))
        return out
