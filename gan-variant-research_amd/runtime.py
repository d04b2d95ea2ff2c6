"""Device buffers ("halo-NHWC" views), the op layer over the C ABI, and replayable programs.

Every op constructor returns a zero-argument callable with its ctypes arguments prebuilt, so a training step
is a flat list of kernel launches ("program") replayed on one HIP stream without any Python-side shape logic,
host synchronisation or allocation -- the same list can be captured into a hipGraph.
PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import torch

from . import _lib
from ._lib import ACT_NONE, BF16, F32, FP8, HALO_NONE, HALO_REFLECT, HALO_ZERO, GanAdamTensor, GanConvDesc, GanView, GanWgradDesc

IN_WS_CHUNKS = 96
ADAM_CHUNK = 16384


def torch_dtype(code: int):
    return torch.float32 if code == F32 else torch.uint8 if code == FP8 else torch.bfloat16      # FP8: raw e4m3 bytes


class View:
    """A [B][H+2h][W+2h][C] activation buffer with a symmetric spatial halo h."""

    def __init__(self, t: torch.Tensor, B: int, H: int, W: int, C_: int, halo: int, dtype: int):
        self.t, self.B, self.H, self.W, self.C, self.halo, self.dtype = t, B, H, W, C_, halo, dtype
        self.Hp, self.Wp = H + 2 * halo, W + 2 * halo
        assert t.numel() == B * self.Hp * self.Wp * C_ and C_ % (16 if dtype == FP8 else 8) == 0
        self._struct = None

    @property
    def y0(self):
        return self.halo

    @property
    def x0(self):
        return self.halo

    def ptr(self) -> int:
        return self.t.data_ptr()

    def struct(self) -> GanView:
        if self._struct is None:
            self._struct = GanView(self.ptr(), self.B, self.Hp, self.Wp, self.C, self.halo, self.halo, self.H, self.W, self.dtype, 0)
        return self._struct

    def nhwc(self) -> torch.Tensor:
        """Logical interior as a (B,H,W,C) tensor view (tests / debugging)."""
        full = self.t.view(self.B, self.Hp, self.Wp, self.C)
        return full[:, self.halo:self.halo + self.H, self.halo:self.halo + self.W, :]

    def padded(self) -> torch.Tensor:
        return self.t.view(self.B, self.Hp, self.Wp, self.C)

    def batch(self, b0: int, n: int) -> "View":
        """Images b0 .. b0+n-1 as a view of the same storage."""
        assert 0 <= b0 and b0 + n <= self.B
        return View(self.t.view(self.B, -1)[b0:b0 + n].reshape(-1), n, self.H, self.W, self.C, self.halo, self.dtype)


@dataclass
class ConvCall:
    """Python mirror of gan_conv_desc (see include/mi355x_gan.h)."""
    B: int
    Ho: int
    Wo: int
    Cin: int
    ntaps: int
    Nw: int
    Nst: int
    x: View
    in_y0: int
    in_x0: int
    in_sy: int
    in_sx: int
    tapoff: torch.Tensor
    w: Optional[torch.Tensor]
    bias: Optional[torch.Tensor]
    out: View
    out_y0: int
    out_x0: int
    out_sy: int
    out_sx: int
    act: int = ACT_NONE
    mask: Optional[View] = None
    mask_y0: int = 0
    mask_x0: int = 0
    max_tapoff: int = 0
    w_frag: bool = False      # w is the fragment-major copy (range-patch kernel)
    tile_rows: int = 0        # range-patch kernel: pixels per tile, fixed at planning time (HipOps.conv_patch_tile_rows)
    tile_cols: int = 0        # ... and output channels per tile (HipOps.conv_patch_tile_cols, after tile_rows)
    stats: Optional[torch.Tensor] = None   # InstanceNorm partials written by the epilogue (see HipOps.conv_stats_parts)
    win7: Optional[tuple] = None           # (ty0, tx0): run by the 7x7 window kernel, taps row-major from that position (w_layout 2)
    w_scale: Optional[torch.Tensor] = None  # fp8 operands: device float, dequantisation scale of the weight copy
    in_scale: Optional[torch.Tensor] = None  # fp8 operands: device float[B], per-image scale of the input copy (None: 1)
    stats_mode: int = 0                    # 0: forward statistics; 1: backward chain, (sum t [m > 0], sum t m) with m = `mask` at the pixel (gan_conv_desc)
    flop_scale: float = 1.0                # algorithmic / launched FLOPs (paired phases multiply by zero blocks: 0.75); bench.py prices with it
    alg_pixels: Optional[int] = None       # output pixels per image of the REFERENCE op this launch implements, when they differ from Ho*Wo
                                           # (stride-1 input gradients run on the input's -- possibly reflect-padded -- domain); SURVEY §8d

    def alg_flops(self) -> float:
        """Algorithmic FLOPs (2*M*N*K with M the reference op's output pixels), what bench.py and tools/step_ops.py price."""
        px = self.Ho * self.Wo if self.alg_pixels is None else self.alg_pixels
        return 2.0 * self.B * px * self.Nst * self.Cin * self.ntaps * self.flop_scale


@dataclass
class WgradCall:
    """Python mirror of gan_wgrad_desc."""
    B: int
    Ho: int
    Wo: int
    Cx: int
    ntaps: int
    N: int
    nsplit: int
    x: View
    x_y0: int
    x_x0: int
    x_sy: int
    x_sx: int
    tapoff: torch.Tensor
    g: View
    g_y0: int
    g_x0: int
    g_sy: int
    g_sx: int
    part: Optional[torch.Tensor]
    max_tapoff: int = 0
    variant: int = 0          # 1: range-patch kernel (nsplit = B * splits per image); 2: 7x7 window kernel (nsplit = its block count)


Op = Callable[[], None]



class Program:
    """A flat, replayable list of kernel launches."""

    def __init__(self, name: str = ""):
        self.name, self.ops = name, []  # type: str, List[Op]

    def add(self, op):
        if op is None:
            return
        if isinstance(op, (list, tuple)):
            for o in op:
                self.add(o)
        elif isinstance(op, Program):
            self.ops.extend(op.ops)
        else:
            self.ops.append(op)

    def run(self):
        for op in self.ops:
            op()

    def __len__(self):
        return len(self.ops)


class HipOps:
    """Op constructors bound to libmi355x_gan.so and one HIP stream.  `stream` is a raw hipStream_t handle (int)."""

    is_hip = True

    def __init__(self, device: torch.device, stream: Optional[int] = None, torch_stream=None):
        self.lib = _lib.load()
        self.device = device
        self.stream = stream  # None: torch's current stream at op-construction time (the module API builds graph slots that way); see bind()
        self.torch_stream = torch_stream   # the torch.cuda.Stream behind `stream` (needed for event record / wait)
        self._keep = []       # ctypes structs referenced by prebuilt calls
        self._side = None
        self._fork = None

    # ---- a second HIP stream: independent MFMA-bound launches (weight gradients) run there while the HBM-bound chain
    #      (InstanceNorm backward, reflection folds) continues on the main stream; ordering by events recorded in the programs
    def side(self) -> "HipOps":
        if os.environ.get("GAN_SINGLE_STREAM"):     # profiling aid: per-kernel durations without concurrent kernels stretching them
            return self
        if self._side is None:
            ts = torch.cuda.Stream(device=self.device)
            self._side = HipOps(self.device, stream=ts.cuda_stream, torch_stream=ts)
        return self._side

    def fork(self) -> "HipOps":
        """A new op layer on its own (non-blocking) HIP stream of the same device (GAN_SINGLE_STREAM: this one)."""
        if os.environ.get("GAN_SINGLE_STREAM"):
            return self
        if self._fork is None:
            ts = torch.cuda.Stream(device=self.device)
            self._fork = HipOps(self.device, stream=ts.cuda_stream, torch_stream=ts)
        return self._fork

    def bind_queues(self):
        """Creates this op layer's streams (discriminator stream, weight-gradient side stream) and runs one trivial kernel on each.
        HIP binds a stream to one of its few hardware queues at first use; doing that before anything else creates streams (RCCL does
        when the process group is set up) keeps the three compute streams on three different queues."""
        if os.environ.get("GAN_SINGLE_STREAM") or self.device.type != "cuda":
            return
        for h in (self, self.fork(), self.side()):
            with torch.cuda.stream(h._ts()):
                torch.zeros(64, device=self.device).add_(1.0)
        torch.cuda.synchronize(self.device)

    def bind(self):
        """Binds an op layer created without a stream to the stream that is current NOW, for good: launches (raw handle baked into the
        prebuilt calls) and the torch-side events, copies and collectives (_ts) of a trainer then always meet on one stream,
        whatever is current when its programs are built or replayed."""
        if self.stream is None and self.device.type == "cuda":
            self.torch_stream = torch.cuda.current_stream(self.device)
            self.stream = self.torch_stream.cuda_stream
        return self

    def _ts(self):
        return self.torch_stream if self.torch_stream is not None else torch.cuda.current_stream(self.device)

    def new_event(self):
        return torch.cuda.Event()

    def record(self, ev) -> Op:
        """ev marks everything queued so far on this op layer's stream."""
        return lambda: ev.record(self._ts())

    def wait(self, ev) -> Op:
        """Launches queued after this on this op layer's stream start after `ev`."""
        return lambda: self._ts().wait_event(ev)

    def _s(self):
        return C.c_void_p(self.stream if self.stream is not None else torch.cuda.current_stream(self.device).cuda_stream)

    def _call(self, name: str, *args) -> Op:
        fn = getattr(self.lib, name)
        self._keep.append(args)
        lib = self.lib
        if os.environ.get("GAN_DEBUG_SYNC"):   # debugging aid: name every launch and wait for it, so a fault is attributable
            def dbg():
                print(f"[gan] {name}", file=sys.stderr, flush=True)
                rc = fn(*args)
                if rc != 0:
                    raise _lib.GanError(f"{name}: {lib.gan_last_error().decode()}")
                torch.cuda.synchronize()
            return dbg

        def op():
            rc = fn(*args)
            if rc != 0:
                raise _lib.GanError(f"{name}: {lib.gan_last_error().decode()}")
        op.__name__ = name
        return op

    # Prebuilt calls hold RAW device pointers: every tensor / view an op captures is pinned here so that it can never be
    # returned to the caching allocator while a program that uses it is alive.
    def _p(self, t: Optional[torch.Tensor]):
        if t is None:
            return C.c_void_p(0)
        self._keep.append(t)
        return C.c_void_p(t.data_ptr())

    def _v(self, v: Optional[View]):
        if v is None:
            return C.cast(None, _lib.PV)
        self._keep.append(v)
        return C.byref(v.struct())

    # ---- convolution family
    def _conv_desc(self, c: ConvCall) -> GanConvDesc:
        d = GanConvDesc()
        d.dtype, d.B, d.Ho, d.Wo, d.Cin, d.ntaps, d.Nw, d.Nst = c.x.dtype, c.B, c.Ho, c.Wo, c.Cin, c.ntaps, c.Nw, c.Nst
        d.in_, d.in_Hp, d.in_Wp, d.in_y0, d.in_x0, d.in_sy, d.in_sx = c.x.ptr(), c.x.Hp, c.x.Wp, c.in_y0, c.in_x0, c.in_sy, c.in_sx
        d.tapoff, d.w, d.bias = c.tapoff.data_ptr(), (c.w.data_ptr() if c.w is not None else None), (c.bias.data_ptr() if c.bias is not None else None)
        d.out, d.out_Hp, d.out_Wp, d.out_C = c.out.ptr(), c.out.Hp, c.out.Wp, c.out.C
        d.out_y0, d.out_x0, d.out_sy, d.out_sx, d.act = c.out_y0, c.out_x0, c.out_sy, c.out_sx, c.act
        if c.mask is not None:
            assert c.mask.C == c.out.C and c.mask.dtype == c.out.dtype
            d.mask, d.mask_Hp, d.mask_Wp, d.mask_y0, d.mask_x0 = c.mask.ptr(), c.mask.Hp, c.mask.Wp, c.mask_y0, c.mask_x0
        d.stats = c.stats.data_ptr() if c.stats is not None else None
        d.stats_mode = c.stats_mode
        d.max_tapoff = c.max_tapoff
        d.w_layout = 1 if c.w_frag else 0
        d.tile_rows, d.tile_cols = c.tile_rows, c.tile_cols
        d.w_scale = c.w_scale.data_ptr() if c.w_scale is not None else None
        d.in_scale = c.in_scale.data_ptr() if c.in_scale is not None else None
        if c.win7 is not None:
            assert not c.w_frag
            d.w_layout, d.win_ty0, d.win_tx0 = 2, c.win7[0], c.win7[1]
        assert c.out.dtype == (BF16 if c.x.dtype == FP8 else c.x.dtype)       # fp8 operands produce a bf16 result
        return d

    def conv_patch_ok(self, c: ConvCall) -> bool:
        """True if the range-patch kernel takes this call (then `c.w` must be the fragment-major weight copy)."""
        return bool(self.lib.gan_conv_patch_ok(C.byref(self._conv_desc(c))))

    def conv_patch_tile_rows(self, c: ConvCall) -> int:
        """Pixels per tile the range-patch kernel uses for this call (0: not eligible); planned once, carried in the descriptor."""
        return int(self.lib.gan_conv_patch_tile_rows(C.byref(self._conv_desc(c))))

    def conv_patch_tile_cols(self, c: ConvCall) -> int:
        """Output channels per tile (128 | 256) for the call's tile_rows; planned once, carried in the descriptor."""
        return int(self.lib.gan_conv_patch_tile_cols(C.byref(self._conv_desc(c))))

    def conv_patch_variant(self, c: ConvCall) -> dict:
        """The range-patch instantiation this call runs on ({} if it does not qualify): rows, cols, slices, fp8, static9, static_taps."""
        d = self._conv_desc(c)
        v = int(self.lib.gan_conv_patch_variant(C.byref(d))) if c.w_frag else 0
        return {} if v == 0 else {"rows": v & 0xfff, "cols": (v >> 12) & 0xfff, "slices": (v >> 24) & 0xf, "fp8": bool((v >> 28) & 1), "static9": bool((v >> 29) & 1),
                                      "static_taps": 9 if (v >> 29) & 1 else (0, 4, 2, 16)[(v >> 30) & 3]}

    def conv_win7_ok(self, c: ConvCall, ty0: int, tx0: int) -> bool:
        """True if the 7x7 window kernel takes this call with its 49 row-major taps starting at (ty0, tx0)."""
        d = self._conv_desc(c)
        d.win_ty0, d.win_tx0 = ty0, tx0
        return bool(self.lib.gan_conv_win7_ok(C.byref(d)))

    def conv_stats_parts(self, c: ConvCall) -> int:
        """Pixel tiles per image for which this call can write InstanceNorm partials in its epilogue (0: it cannot)."""
        return int(self.lib.gan_conv_stats_parts(C.byref(self._conv_desc(c))))

    def in_stats_from_parts(self, parts, nparts, B, Cc, HW, eps, stats) -> Op:
        return self._call("gan_in_stats_from_parts", self._p(parts), nparts, B, Cc, HW, C.c_float(eps), self._p(stats), self._s())

    def conv_igemm(self, c: ConvCall) -> Op:
        self._keep.append(c)
        op = self._call("gan_conv_igemm", C.byref(self._conv_desc(c)), self._s())
        op.conv = c   # lets bench.py enumerate a program's convolution launches and price them
        return op

    def _wgrad_desc(self, c: WgradCall) -> GanWgradDesc:
        d = GanWgradDesc()
        d.dtype, d.B, d.Ho, d.Wo, d.Cx, d.ntaps, d.N, d.nsplit = c.x.dtype, c.B, c.Ho, c.Wo, c.Cx, c.ntaps, c.N, c.nsplit
        d.x, d.x_Hp, d.x_Wp, d.x_y0, d.x_x0, d.x_sy, d.x_sx = c.x.ptr(), c.x.Hp, c.x.Wp, c.x_y0, c.x_x0, c.x_sy, c.x_sx
        d.tapoff, d.g = c.tapoff.data_ptr(), c.g.ptr()
        d.g_Hp, d.g_Wp, d.g_C, d.g_y0, d.g_x0, d.g_sy, d.g_sx = c.g.Hp, c.g.Wp, c.g.C, c.g_y0, c.g_x0, c.g_sy, c.g_sx
        d.part = c.part.data_ptr() if c.part is not None else None
        d.max_tapoff, d.variant = c.max_tapoff, c.variant
        assert c.g.dtype == c.x.dtype
        return d

    def wgrad_patch_splits(self, c: WgradCall) -> int:
        """Splits per image the range-patch weight-gradient kernel wants for this call (0: not eligible)."""
        return int(self.lib.gan_wgrad_patch_splits(C.byref(self._wgrad_desc(c))))

    def wgrad_win7_splits(self, c: WgradCall) -> int:
        """Slabs the 7x7 window weight-gradient kernel writes for this call (0: not eligible); then variant 2, nsplit = this."""
        return int(self.lib.gan_wgrad_win7_splits(C.byref(self._wgrad_desc(c))))

    def conv_wgrad(self, c: WgradCall) -> Op:
        self._keep.append(c)
        assert c.part.numel() >= c.nsplit * c.N * c.ntaps * c.Cx
        op = self._call("gan_conv_wgrad", C.byref(self._wgrad_desc(c)), self._s())
        op.wgrad = c
        return op

    def wgrad_reduce(self, part, nsplit, N, ntaps, Cx, N_real, C_real, swap, I2, KK, khw, grad, accumulate) -> Op:
        return self._call("gan_wgrad_reduce", self._p(part), nsplit, N, ntaps, Cx, N_real, C_real, int(swap), I2, KK, self._p(khw),
                          self._p(grad), int(accumulate), self._s())

    def pack_weight(self, src, dst, dtype, Nw, ntaps, Cin, N_real, C_real, swap, I2, KK, khw, layout=0, scale=None) -> Op:
        """One operand copy.  dtype FP8 (e4m3, needs `scale`: a device float the batch launch fills with max|W| / 448) exists only in
        the batched form: the returned op then only carries its arguments for pack_weight_batch."""
        if dtype == FP8:
            assert scale is not None and layout == 1

            def op():
                raise _lib.GanError("fp8 operand copies are packed by pack_weight_batch")
        else:
            op = self._call("gan_pack_weight", self._p(src), self._p(dst), dtype, Nw, ntaps, Cin, N_real, C_real, int(swap), I2, KK,
                            self._p(khw), int(layout), self._s())
        op.pack_args = (src, dst, dtype, Nw, ntaps, Cin, N_real, C_real, int(swap), I2, KK, khw, int(layout), scale)
        return op

    def pack_weight_batch(self, packs) -> Op:
        """One launch for many operand copies; `packs` = the pack_args tuples of ops built by pack_weight."""
        arr = (_lib.GanPackDesc * len(packs))()
        first = 0
        any_fp8 = False
        for d, (src, dst, dtype, Nw, ntaps, Cin, N_real, C_real, swap, I2, KK, khw, layout, *rest) in zip(arr, packs):
            assert layout == 0 or (Nw % 16 == 0 and (ntaps * Cin) % 32 == 0)
            scale = rest[0] if rest else None
            self._keep.extend((src, dst, khw, scale))
            d.src, d.dst, d.khw = src.data_ptr(), dst.data_ptr(), khw.data_ptr()
            d.scale = scale.data_ptr() if scale is not None else None
            any_fp8 = any_fp8 or dtype == FP8
            d.dtype, d.Nw, d.ntaps, d.Cin, d.N_real, d.C_real, d.swap, d.I2, d.KK, d.layout = dtype, Nw, ntaps, Cin, N_real, C_real, swap, I2, KK, layout
            d.nblocks = max(1, min(512, (Nw * ntaps * Cin + 1023) // 1024))
            d.first_block, first = first, first + d.nblocks
        table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        pack = self._call("gan_pack_weight_batch", self._p(table), len(packs), first, self._s())
        if not any_fp8:
            return pack
        scales = self._call("gan_weight_scale_batch", self._p(table), len(packs), self._s())   # per-tensor max|W| / 448 first

        def op():
            scales()
            pack()
        op.__name__ = "gan_pack_weight_batch"
        return op

    def bias_grad(self, g: View, N_real, grad, accumulate, ws) -> Op:
        return self._call("gan_bias_grad", self._v(g), N_real, self._p(grad), int(accumulate), self._p(ws), self._s())

    # ---- norm / activations / layout
    def in_stats(self, x: View, eps, stats, ws) -> Op:
        return self._call("gan_in_stats", self._v(x), C.c_float(eps), self._p(stats), self._p(ws), self._s())

    @staticmethod
    def _hbm(op: Op, x: View, ntensors: int) -> Op:
        """Algorithmic HBM bytes of an HBM-bound launch (every operand tensor read or written once), for bench.py."""
        op.hbm_bytes = ntensors * x.B * x.H * x.W * x.C * (4 if x.dtype == F32 else 2)
        return op

    def in_apply(self, x: View, stats, act, residual: Optional[View], y: View, halo_mode) -> Op:
        return self._hbm(self._call("gan_in_apply", self._v(x), self._p(stats), act, self._v(residual), self._v(y), halo_mode, self._s()),
                         x, 2 + (residual is not None))

    def in_partial_count(self, x: View) -> int:
        """Statistics partials per image gan_in_partial writes for x (<= 16: gan_in_apply_parts sums them itself)."""
        n = int(self.lib.gan_in_partial_count(self._v(x)))
        if n < 1:
            raise _lib.GanError(self.lib.gan_last_error().decode())
        return n

    def in_partial(self, x: View, parts) -> Op:
        assert parts.dtype == torch.float32 and parts.numel() >= x.B * self.in_partial_count(x) * x.C * 2
        return self._call("gan_in_partial", self._v(x), self._p(parts), self._s())

    def in_apply_parts(self, x: View, parts, nparts, eps, stats, act, residual: Optional[View], y: View, halo_mode, y8: Optional[View] = None) -> Op:
        """InstanceNorm apply with the statistics summed from `nparts` (<= 16) partial pairs per image inside the pass (no finalize launch).
        y8: an e4m3 view of y's geometry that receives a copy of y in the same pass (fp8 path)."""
        assert 1 <= nparts <= 16 and parts.numel() >= x.B * nparts * x.C * 2 and stats.numel() >= x.B * x.C * 2
        if y8 is not None:
            assert y8.dtype == FP8 and (y8.B, y8.Hp, y8.Wp, y8.C, y8.halo) == (y.B, y.Hp, y.Wp, y.C, y.halo)
            return self._hbm(self._call("gan_in_apply_parts_fp8", self._v(x), self._p(parts), nparts, C.c_float(eps), self._p(stats), act, self._v(residual),
                                        self._v(y), self._v(y8), halo_mode, self._s()), x, 2 + (residual is not None))
        return self._hbm(self._call("gan_in_apply_parts", self._v(x), self._p(parts), nparts, C.c_float(eps), self._p(stats), act, self._v(residual),
                                    self._v(y), halo_mode, self._s()), x, 2 + (residual is not None))

    def in_bwd(self, x: View, stats, act, gy: View, fold, g2: Optional[View], dx: View, ws) -> Op:
        return self._hbm(self._call("gan_in_bwd", self._v(x), self._p(stats), act, self._v(gy), int(fold), self._v(g2), self._v(dx), self._p(ws), self._s()),
                         x, 3 + (g2 is not None))

    def in_bwd_bias(self, x: View, stats, act, gy: View, fold, g2: Optional[View], dx: View, ws, bias_grad, bias_n, accumulate) -> Op:
        return self._hbm(self._call("gan_in_bwd_bias", self._v(x), self._p(stats), act, self._v(gy), int(fold), self._v(g2), self._v(dx), self._p(ws),
                                    self._p(bias_grad), bias_n, int(accumulate), self._s()), x, 3 + (g2 is not None))

    def in_bwd_bias_parts(self, x: View) -> int:
        n = int(self.lib.gan_in_bwd_bias_parts(self._v(x)))
        if n < 0:
            raise _lib.GanError(self.lib.gan_last_error().decode())
        return n

    def in_bwd_bias_deferred(self, x: View, stats, act, gy: View, fold, g2: Optional[View], dx: View, ws, bias_part) -> Op:
        assert bias_part.dtype == torch.float32 and bias_part.numel() >= self.in_bwd_bias_parts(x) * x.C
        return self._hbm(self._call("gan_in_bwd_bias_deferred", self._v(x), self._p(stats), act, self._v(gy), int(fold), self._v(g2), self._v(dx),
                                    self._p(ws), self._p(bias_part), self._s()), x, 3 + (g2 is not None))

    def in_bwd_parts(self, x: View, stats, act, gy: View, fold, dx: View, parts, nparts, parts_mode, bias_part=None) -> Op:
        """The apply half of the InstanceNorm backward alone: the per-(image, channel) sums come as `nparts` partial pairs from the
        input-gradient epilogue that wrote gy (ConvCall.stats_mode 1 | 2 = parts_mode)."""
        assert parts.dtype == torch.float32 and parts.numel() >= x.B * nparts * x.C * 2
        assert bias_part is None or bias_part.numel() >= self.in_bwd_bias_parts(x) * x.C
        return self._hbm(self._call("gan_in_bwd_parts", self._v(x), self._p(stats), act, self._v(gy), int(fold), self._v(dx), self._p(parts), nparts,
                                    parts_mode, self._p(bias_part), self._s()), x, 3)

    def bias_finalize_batch(self, items) -> Op:
        """items: (part, nparts, C, grad, N_real, accumulate) per layer -> one launch."""
        arr = (_lib.GanBiasPartDesc * len(items))()
        first = 0
        for d, (part, nparts, Cc, grad, n_real, acc) in zip(arr, items):
            self._keep.extend((part, grad))
            d.part, d.grad, d.nparts, d.C, d.N_real, d.accumulate = part.data_ptr(), grad.data_ptr(), nparts, Cc, n_real, int(acc)
            d.first_block, first = first, first + (Cc + 31) // 32
        table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        return self._call("gan_bias_finalize_batch", self._p(table), len(items), first, self._s())

    def in_bwd_amax(self, x: View, stats, act, gy: View, fold, dx: View, ws, bias_part, amax) -> Op:
        """InstanceNorm backward that also leaves max|dx| per image in `amax` (float[B]): the scale of dx's e4m3 copy."""
        assert amax.dtype == torch.float32 and amax.numel() >= x.B
        return self._hbm(self._call("gan_in_bwd_amax", self._v(x), self._p(stats), act, self._v(gy), int(fold), self._v(dx), self._p(ws),
                                    self._p(bias_part), self._p(amax), self._s()), x, 3)

    def quantize_fp8(self, src: View, dst: View, amax=None, scale_out=None) -> Op:
        """e4m3 copy of a whole buffer (halo included): unit scale, or per-image scale amax[b] / 448 written to scale_out[b]."""
        assert dst.dtype == FP8 and (src.B, src.Hp, src.Wp, src.C, src.halo) == (dst.B, dst.Hp, dst.Wp, dst.C, dst.halo)
        return self._call("gan_quantize_fp8", self._v(src), self._v(dst), self._p(amax), self._p(scale_out), self._s())

    def fold_add(self, a: Optional[View], b: View, fold, out: View) -> Op:
        return self._call("gan_fold_add", self._v(a), self._v(b), int(fold), self._v(out), self._s())

    def pad_fold(self, g: View, mode, out: View) -> Op:
        """Gradient of a padding layer: out = sum of g (which carries the halo) over the padded positions copied from each pixel."""
        return self._call("gan_pad_fold", self._v(g), mode, self._v(out), self._s())

    def act_bwd(self, y: View, act, g: View, fold, g2: Optional[View], dx: View) -> Op:
        return self._call("gan_act_bwd", self._v(y), act, self._v(g), int(fold), self._v(g2), self._v(dx), self._s())

    def nchw_to_view(self, src: torch.Tensor, Cr, dst: View, halo_mode) -> Op:
        assert src.dtype == torch.float32 and src.is_contiguous()
        return self._call("gan_nchw_to_view", self._p(src), Cr, self._v(dst), halo_mode, self._s())

    def view_to_nchw(self, src: View, Cr, dst: torch.Tensor) -> Op:
        assert dst.dtype == torch.float32 and dst.is_contiguous()
        return self._call("gan_view_to_nchw", self._v(src), Cr, self._p(dst), self._s())

    def view_copy(self, src: View, dst: View, halo_mode) -> Op:
        return self._call("gan_view_copy", self._v(src), self._v(dst), halo_mode, self._s())

    def spectral_norm_ws_floats(self, h, w) -> int:
        return int(_lib.load().gan_spectral_norm_ws_floats(h, w))

    def spectral_norm_fwd(self, W, u, v, power_iter: bool, eps, sigma, Wsn, ws) -> Op:
        h, w = W.shape[0], W.numel() // W.shape[0]
        for t in (W, u, v, sigma, Wsn, ws):
            assert t.dtype == torch.float32 and t.is_contiguous()
        assert u.numel() == h and v.numel() == w and Wsn.numel() == W.numel() and ws.numel() >= self.spectral_norm_ws_floats(h, w)
        return self._call("gan_spectral_norm_fwd", self._p(W), h, w, self._p(u), self._p(v), int(power_iter), float(eps), self._p(sigma),
                          self._p(Wsn), self._p(ws), self._s())

    def spectral_norm_bwd(self, G, Wsn, u, v, sigma, dW, ws) -> Op:
        h, w = G.shape[0], G.numel() // G.shape[0]
        for t in (G, Wsn, u, v, sigma, dW, ws):
            assert t.dtype == torch.float32 and t.is_contiguous()
        assert u.numel() == h and v.numel() == w and dW.numel() == G.numel() and ws.numel() >= self.spectral_norm_ws_floats(h, w)
        return self._call("gan_spectral_norm_bwd", self._p(G), self._p(Wsn), self._p(u), self._p(v), self._p(sigma), h, w, self._p(dW),
                          self._p(ws), self._s())

    def avgpool_fwd(self, x: View, y: View) -> Op:
        return self._call("gan_avgpool_fwd", self._v(x), self._v(y), self._s())

    def avgpool_bwd(self, gy: View, gx: View, accumulate: bool) -> Op:
        return self._call("gan_avgpool_bwd", self._v(gy), self._v(gx), int(accumulate), self._s())

    # ---- augmentation and losses
    def diffaug_fwd(self, x: View, Cr, prm, y: View, ws) -> Op:
        return self._call("gan_diffaug_fwd", self._v(x), Cr, self._p(prm), self._v(y), self._p(ws), self._s())

    def diffaug_bwd(self, gy: View, Cr, prm, gx: View, ws) -> Op:
        return self._call("gan_diffaug_bwd", self._v(gy), Cr, self._p(prm), self._v(gx), self._p(ws), self._s())

    def patch_loss(self, logits: View, mode, target, scale, loss, grad: Optional[View]) -> Op:
        return self._call("gan_patch_loss", self._v(logits), mode, C.c_float(target), C.c_float(scale), self._p(loss), self._v(grad), self._s())

    def l1_loss(self, x: View, Cr, target_nchw, scale, dev_scale, loss, grad: Optional[View], ws) -> Op:
        return self._call("gan_l1_loss", self._v(x), Cr, self._p(target_nchw), C.c_float(scale), self._p(dev_scale), self._p(loss),
                          self._v(grad), self._p(ws), self._s())

    def r1_reduce(self, g: View, Cr, scale, loss, u: Optional[View], ws) -> Op:
        return self._call("gan_r1_reduce", self._v(g), Cr, C.c_float(scale), self._p(loss), self._v(u), self._p(ws), self._s())

    def patchnce_ws_floats(self, B, P, Cc) -> int:
        return int(self.lib.gan_patchnce_ws_floats(B, P, Cc))

    def patchnce_fwd(self, src: View, tgt: View, ids, P, Cc, temperature, weight, loss, ws) -> Op:
        return self._call("gan_patchnce_fwd", self._v(src), self._v(tgt), self._p(ids), P, Cc, C.c_float(temperature), C.c_float(weight),
                          self._p(loss), self._p(ws), self._s())

    def patchnce_bwd(self, tgt: View, ids, P, Cc, temperature, weight, gtgt: View, ws) -> Op:
        return self._call("gan_patchnce_bwd", self._v(tgt), self._p(ids), P, Cc, C.c_float(temperature), C.c_float(weight), self._v(gtgt),
                          self._p(ws), self._s())

    # ---- optimiser
    def adam_step(self, table, ntensors, chunk_tensor, chunk_off, nchunks, lr, b1, b2, eps, max_norm, grad_scale, ema_decay, norm_out, ws,
                  lr_dev=None, inv_scale=None, skip_nonfinite=False) -> Op:
        """lr_dev: device float that holds the learning rate (a scheduler rewrites it; `lr` is then ignored); inv_scale / skip_nonfinite:
        GradScaler semantics (gradients x *inv_scale; a non-finite norm skips the step); norm_out: [3] (norm, clip coefficient, found_inf)."""
        assert norm_out.numel() >= 3
        return self._call("gan_adam_step", self._p(table), ntensors, self._p(chunk_tensor), self._p(chunk_off), nchunks, C.c_float(lr),
                          C.c_float(b1), C.c_float(b2), C.c_float(eps), C.c_float(max_norm), C.c_float(grad_scale), C.c_float(ema_decay),
                          self._p(lr_dev), self._p(inv_scale), int(skip_nonfinite), self._p(norm_out), self._p(ws), self._s())

    def scaler_update(self, scale, inv_scale, tracker, found_inf, growth=2.0, backoff=0.5, interval=2000) -> Op:
        assert tracker.dtype == torch.int32
        return self._call("gan_scaler_update", self._p(scale), self._p(inv_scale), self._p(tracker), self._p(found_inf), C.c_float(growth),
                          C.c_float(backoff), int(interval), self._s())

    def make_adam_table(self, entries: Sequence[dict]) -> torch.Tensor:
        """entries: dicts with tensors p, g (or None), m, v, ema (or None), step (int32 tensor of 1).  -> device uint8 table."""
        arr = (GanAdamTensor * len(entries))()
        self._keep.append(entries)
        for i, e in enumerate(entries):
            arr[i].p, arr[i].m, arr[i].v = e["p"].data_ptr(), e["m"].data_ptr(), e["v"].data_ptr()
            arr[i].g = e["g"].data_ptr() if e.get("g") is not None else None
            arr[i].ema = e["ema"].data_ptr() if e.get("ema") is not None else None
            arr[i].numel, arr[i].step = e["p"].numel(), e["step"].data_ptr()
        raw = bytes(arr)
        return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)

    def fill(self, t: torch.Tensor, value: float) -> Op:
        assert t.dtype == torch.float32
        return self._call("gan_fill_f32", self._p(t), C.c_int64(t.numel()), C.c_float(value), self._s())

    def axpy(self, y: torch.Tensor, x: torch.Tensor, a: float) -> Op:
        return self._call("gan_axpy_f32", self._p(y), self._p(x), C.c_float(a), C.c_int64(y.numel()), self._s())

    def zero_(self, t: torch.Tensor) -> Op:
        """Byte-zero of any buffer (hipMemsetAsync through torch, graph-capturable)."""
        self._keep.append(t)

        def op():
            t.zero_()
        return op


class Ctx:
    """Allocation context: device, operand dtype, op layer and shared scratch workspaces."""

    def __init__(self, ops, device, dtype: int):
        self.ops, self.device, self.dtype = ops, torch.device(device), dtype
        self.tdtype = torch_dtype(dtype)
        self._scratch, self._retired = {}, []

    def view(self, B, H, W, C_, halo=0, dtype: Optional[int] = None) -> View:
        dt = self.dtype if dtype is None else dtype
        t = torch.zeros(B * (H + 2 * halo) * (W + 2 * halo) * C_, dtype=torch_dtype(dt), device=self.device)
        return View(t, B, H, W, C_, halo, dt)

    def f32(self, n, fill=0.0) -> torch.Tensor:
        return torch.full((int(n),), fill, dtype=torch.float32, device=self.device)

    def i32(self, values) -> torch.Tensor:
        return torch.tensor(list(values), dtype=torch.int32, device=self.device)

    def scratch(self, name: str, nfloats: int) -> torch.Tensor:
        """Grow-only shared fp32 workspace (safe to share: all launches are ordered on one stream)."""
        cur = self._scratch.get(name)
        if cur is None or cur.numel() < nfloats:
            if cur is not None:
                self._retired.append(cur)  # ops built earlier hold its raw pointer
            cur = torch.zeros(int(nfloats), dtype=torch.float32, device=self.device)
            self._scratch[name] = cur
        return cur


def cpad(c: int) -> int:
    """Channel padding rule of halo-NHWC: next power of two >= 8."""
    p = 8
    while p < c:
        p *= 2
    return p
