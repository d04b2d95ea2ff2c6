"""TEST INFRASTRUCTURE: a PyTorch-CPU statement of what every libmi355x_gan.so entry point computes.

`EmuOps` has the op-constructor interface of gan_variant_research_amd.runtime.HipOps, so (a) the host logic (tap
tables, halos, phases, program order, optimiser tables) is checked on CPU against the oracle with no GPU, and
(b) the GPU tests compare each HIP kernel with its statement here on the same descriptors.  Never imported by
the product package.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3
HALO_NONE, HALO_ZERO, HALO_REFLECT = 0, 1, 2


def _act(v, act):
    if act == ACT_RELU:
        return torch.relu(v)
    if act == ACT_LRELU:
        return torch.where(v > 0, v, 0.2 * v)
    if act == ACT_TANH:
        return torch.tanh(v)
    return v


def _act_grad_from_out(y, act):
    if act == ACT_RELU:
        return (y > 0).float()
    if act == ACT_LRELU:
        return torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.2))
    if act == ACT_TANH:
        return 1 - y * y
    return torch.ones_like(y)


def _reflect(i, n):
    i = i.abs()
    return torch.where(i >= n, 2 * (n - 1) - i, i)


def _fold(v, fold):
    """(B,H,W,C) float: interior of view v plus, if fold, the reflect-halo contributions (pad = v.halo)."""
    full = v.padded().float()
    p, H, W = v.halo, v.H, v.W
    if not fold:
        return full[:, p:p + H, p:p + W].clone()
    out = torch.zeros(v.B, H, W, v.C)
    ys = _reflect(torch.arange(-p, H + p), H)
    xs = _reflect(torch.arange(-p, W + p), W)
    tmp = torch.zeros(v.B, H, v.Wp, v.C)
    tmp.index_add_(1, ys, full)
    out.index_add_(2, xs, tmp)
    return out


def _store(view, vals, padded_coords=False):
    """writes (B,h,w,C) floats into the interior (or the whole padded extent) in the view's dtype"""
    if padded_coords:
        view.padded().copy_(vals.to(view.t.dtype))
    else:
        view.nhwc().copy_(vals.to(view.t.dtype))


def _e4m3(v):
    """fp32 -> OCP e4m3 (round to nearest even, clamped to +-448) -> fp32: what an e4m3 byte holds."""
    return v.float().clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()


def _vfloat(view):
    """A view's whole padded buffer as floats (fp8 views hold raw e4m3 bytes)."""
    t = view.padded()
    return t.view(torch.float8_e4m3fn).float() if view.dtype == 2 else t.float()


class EmuOps:
    is_hip = False

    def __init__(self):
        self.device = torch.device("cpu")

    # ------------------------------------------------------------------ convolution family
    # one "stream": the emulator executes ops in program order, so events are no-ops
    def side(self):
        return self

    def fork(self):
        return self

    def _ts(self):
        return None

    def new_event(self):
        return None

    def record(self, ev):
        return lambda: None

    def wait(self, ev):
        return lambda: None

    def conv_patch_ok(self, c):
        """Statement of gan_conv_patch_ok (csrc/conv_patch.hip)."""
        fp8 = c.x.dtype == 2
        slots = c.Cin // 2 if fp8 else c.Cin
        if c.x.dtype not in (1, 2) or slots < 64 or slots % 64 or c.Nw % 128 or c.Nst % 8 or c.out.C % 8 or c.max_tapoff <= 0:
            return False
        if fp8 and (c.mask is not None or c.w_scale is None):
            return False
        if c.mask is not None and c.act != ACT_NONE:
            return False
        m_img = c.Ho * c.Wo
        rows = min(256, m_img)
        wraps = (rows - 1) // c.Wo + 1
        jump = max(0, c.x.Wp * c.in_sy - c.Wo * c.in_sx)
        if (rows - 1) * c.in_sx + wraps * jump + c.max_tapoff // c.Cin + 1 > 576:     # the 256-row tile's 9-slice buffers (maps up to 128 wide)
            return False
        if fp8 or c.B * (-(-m_img // 256)) <= 128:
            return True
        rows_used = min(-(-m_img // 256) * 256, -(-m_img // 288) * 288)       # tile utilisation >= 75 % once the CUs are full
        return 4 * m_img >= 3 * rows_used

    def conv_patch_tile_rows(self, c):
        """Statement of gan_conv_patch_tile_rows: any non-zero value for a qualifying call (the emulator has no tiles)."""
        return 256 if self.conv_patch_ok(c) else 0

    def conv_patch_tile_cols(self, c):
        """Statement of gan_conv_patch_tile_cols: any non-zero value for a qualifying call."""
        return 128 if self.conv_patch_ok(c) else 0

    def conv_win7_ok(self, c, ty0, tx0):
        """Mirror of gan_conv_win7_ok (the emulator computes every call the same way; the flag only has to agree with the library)."""
        to3 = c.Cin == 64 and c.ntaps == 49 and c.Nw == 16 and c.Nst == 8 and c.out.C == 8
        from3 = c.Cin == 8 and c.ntaps >= 52 and c.Nw == 64 and c.Nst == 64 and c.out.C == 64
        return (c.x.dtype == 1 and (to3 or from3) and c.in_sy == 1 and c.in_sx == 1
                and c.out_sy == 1 and c.out_sx == 1 and c.mask is None and (c.stats is None or from3) and (c.act == ACT_NONE or (to3 and c.act == ACT_TANH))
                and c.max_tapoff == ((ty0 + 6) * c.x.Wp + tx0 + 6) * c.Cin)

    def conv_stats_parts(self, c):
        """Statement of gan_conv_stats_parts; the emulator reports one part per image."""
        if getattr(c, "win7", None) is not None:            # 7x7 window path: only the 3 -> 64 kernel writes partials
            return 1 if (c.Cin == 8 and c.Nst == 64 and c.act == ACT_NONE) else 0
        if getattr(c, "stats_mode", 0) != 0 and (c.x.dtype != 1 or c.mask is None or c.stats_mode != 1 or c.bias is not None):
            return 0
        return 1 if (self.conv_patch_ok(c) and c.act == ACT_NONE and (c.mask is None or getattr(c, "stats_mode", 0) != 0) and (c.out_sy, c.out_sx) == (1, 1)) else 0

    def in_partial_count(self, x):
        """Statement of gan_in_partial_count; the emulator writes one partial per image."""
        return 1

    def in_partial(self, x, parts):
        def op():
            v = x.nhwc().double()
            parts[:x.B * x.C * 2].view(x.B, 1, x.C, 2).copy_(torch.stack([v.sum((1, 2)), (v * v).sum((1, 2))], -1).unsqueeze(1).float())
        return op

    def in_apply_parts(self, x, parts, nparts, eps, stats, act, residual, y, halo_mode, y8=None):
        """Statement of gan_in_apply_parts(_fp8): gan_in_stats_from_parts, gan_in_apply and (y8) an e4m3 copy of the fp32 result."""
        a = self.in_stats_from_parts(parts, nparts, x.B, x.C, x.H * x.W, eps, stats)
        b = self.in_apply(x, stats, act, residual, y, halo_mode)
        from gan_variant_research_amd.runtime import View
        y32 = View(torch.zeros(y.t.numel()), y.B, y.H, y.W, y.C, y.halo, 0) if y8 is not None else None
        b32 = self.in_apply(x, stats, act, residual, y32, halo_mode) if y8 is not None else None

        def op():
            a()
            b()
            if y8 is not None:        # the kernel converts its fp32 values, not the rounded bf16 ones
                b32()
                y8.padded().copy_(y32.padded().clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8))
        return op

    def in_stats_from_parts(self, parts, nparts, B, Cc, HW, eps, stats):
        def op():
            p = parts[:B * nparts * Cc * 2].view(B, nparts, Cc, 2).double().sum(1)
            mean = p[..., 0] / HW
            var = (p[..., 1] / HW - mean * mean).clamp_min(0)
            stats[:B * Cc * 2].view(B, Cc, 2).copy_(torch.stack([mean, 1.0 / torch.sqrt(var + eps)], -1).float())
        return op

    @staticmethod
    def _unfrag(wf, Nw, K, sub=1):
        """fragment-major [Nw/16][K/32][fg 4][fr 16][8] (x `sub` values per slot: 2 for fp8) -> row-major [Nw][K * sub]"""
        return wf.view(Nw // 16, K // 32, 4, 16, 8 * sub).permute(0, 3, 1, 2, 4).reshape(Nw, K * sub)

    def conv_igemm(self, c):
        def op():
            x = _vfloat(c.x)
            Cin, Wp = c.Cin, c.x.Wp
            assert c.x.C == Cin
            if c.x.dtype == 2:         # e4m3 operands: fragment-major bytes, two channels per 2-byte slot; per-tensor / per-image scales
                assert c.w_frag and self.conv_patch_ok(c)
                wb = c.w.view(torch.float8_e4m3fn).float().view(-1, 2)                       # [slot][2]
                w = self._unfrag(wb.reshape(-1), c.Nw, c.ntaps * Cin // 2, 2).view(c.Nw, c.ntaps, Cin) * float(c.w_scale)
                if c.in_scale is not None:
                    x = x * c.in_scale[:c.B].float().view(c.B, 1, 1, 1)
            elif c.w_frag:
                assert self.conv_patch_ok(c)
                w = self._unfrag(c.w.float(), c.Nw, c.ntaps * Cin).view(c.Nw, c.ntaps, Cin)
            else:
                w = c.w.view(c.Nw, c.ntaps, Cin).float()
            acc = torch.zeros(c.B, c.Ho, c.Wo, c.Nw)
            toff = c.tapoff.tolist()
            bke = 32 if c.x.dtype == 0 else 128 if c.x.dtype == 2 else 64
            assert (c.ntaps * Cin) % bke == 0 and len(toff) == c.ntaps
            ys0 = c.in_y0 + torch.arange(c.Ho) * c.in_sy
            xs0 = c.in_x0 + torch.arange(c.Wo) * c.in_sx
            for t, off in enumerate(toff):
                assert off % Cin == 0
                dy, dx = (off // Cin) // Wp, (off // Cin) % Wp
                if not w[:, t].any():
                    continue
                patch = x[:, ys0 + dy][:, :, xs0 + dx]
                acc += patch @ w[:, t].T
            v = acc[..., :c.Nst]
            if c.bias is not None:
                v = v + c.bias[:c.Nst].float()
            oy = c.out_y0 + torch.arange(c.Ho) * c.out_sy
            ox = c.out_x0 + torch.arange(c.Wo) * c.out_sx
            smode = getattr(c, "stats_mode", 0)
            if getattr(c, "stats", None) is not None:   # fused InstanceNorm partials: one part per image here (any tiling sums to the same)
                assert self.conv_stats_parts(c) == 1 and c.Nst == c.out.C
                if smode == 0:
                    pair = [v.sum((1, 2)), (v * v).sum((1, 2))]
                else:     # backward chain: sums of the rounded gradient against m = the saved ReLU output at the output pixel itself (halo included)
                    assert smode == 1 and c.bias is None and c.act == ACT_NONE
                    m = c.mask.padded().float()[:, c.mask_y0 + torch.arange(c.Ho)][:, :, c.mask_x0 + torch.arange(c.Wo)][..., :c.Nst]
                    vr = v.to(c.out.padded().dtype).float()
                    pair = [(vr * (m > 0)).sum((1, 2)), (vr * m).sum((1, 2))]
                c.stats[:c.B * c.Nst * 2].view(c.B, 1, c.Nst, 2).copy_(torch.stack(pair, -1).unsqueeze(1))
            v = _act(v, c.act)
            if c.mask is not None and smode == 0:
                my = c.mask_y0 + torch.arange(c.Ho) * c.out_sy
                mx = c.mask_x0 + torch.arange(c.Wo) * c.out_sx
                m = c.mask.padded().float()[:, my][:, :, mx][..., :c.Nst]
                v = v * torch.where(m > 0, torch.ones_like(m), torch.full_like(m, 0.2))
            out = c.out.padded()
            out[:, oy[:, None], ox[None, :], :c.Nst] = v.to(out.dtype)
        return op

    def wgrad_patch_splits(self, c):
        """Statement of gan_wgrad_patch_splits (csrc/wgrad_patch.hip)."""
        if c.x.dtype != 1 or c.ntaps != 9 or c.Cx % 64 or c.N % 128 or c.N != c.g.C:
            return 0
        if (c.x_sy, c.x_sx, c.g_sy, c.g_sx) != (1, 1, 1, 1) or c.Ho * c.Wo < 128:
            return 0
        if c.Wo < 16 or c.Wo & (c.Wo - 1) or 128 % c.Wo or c.max_tapoff != (2 * c.x.Wp + 2) * c.Cx:
            return 0
        window = (128 // c.Wo + 2) * ((c.Wo + 2 + 7) // 8 * 8)
        if window > 320 and c.Wo != 128:      # 128-wide maps: the row-ring variant
            return 0
        bps = (c.N // 128) * (c.Cx // 64)
        if c.Ho * c.Wo < 8 * 128 and c.B * bps > 256 and (c.Ho * c.Wo) % 128 == 0 and window <= 320:
            ipb = c.B * bps // 256        # many small images: a split covers -return whole images
            while ipb > 1 and c.B % ipb:
                ipb -= 1
            if ipb > 1:
                return -ipb
        if c.Ho * c.Wo < 8 * 128 and c.B > 64:      # many small images, no even grouping: short splits, B slabs to reduce
            return 0
        spi = (256 + c.B * bps - 1) // (c.B * bps)
        return max(1, min(spi, max(1, c.Ho * c.Wo // 256)))

    def wgrad_win7_splits(self, c):
        """Statement of gan_wgrad_win7_splits (csrc/conv_win7.hip)."""
        to3, from3 = (c.Cx, c.N, c.g.C) == (64, 8, 8), (c.Cx, c.N, c.g.C) == (8, 64, 64)
        if c.x.dtype != 1 or not (to3 or from3) or c.ntaps != 49 or (c.x_sy, c.x_sx, c.g_sy, c.g_sx) != (1, 1, 1, 1):
            return 0
        if c.max_tapoff != (6 * c.x.Wp + 6) * c.Cx:
            return 0
        return min(256, c.B * (-(-c.Ho // 16)) * (-(-c.Wo // 16)))

    def conv_wgrad(self, c):
        def op():
            x = c.x.padded().float()
            g = c.g.padded().float()
            ys = c.x_y0 + torch.arange(c.Ho) * c.x_sy
            xs = c.x_x0 + torch.arange(c.Wo) * c.x_sx
            gy = c.g_y0 + torch.arange(c.Ho) * c.g_sy
            gx = c.g_x0 + torch.arange(c.Wo) * c.g_sx
            gm = g[:, gy][:, :, gx][..., :c.N].reshape(-1, c.N)
            Wp, Cx = c.x.Wp, c.Cx
            assert c.x.C == Cx
            part = torch.zeros(c.N, c.ntaps, Cx)
            for t, off in enumerate(c.tapoff.tolist()):
                dy, dx = (off // Cx) // Wp, (off // Cx) % Wp
                xm = x[:, ys + dy][:, :, xs + dx].reshape(-1, Cx)
                part[:, t] = gm.T @ xm
            buf = c.part.view(-1)
            n = c.N * c.ntaps * Cx
            buf[:c.nsplit * n] = 0
            buf[:n] = part.reshape(-1)   # the split decomposition is a kernel detail; slab 0 carries the sum here
        return op

    def wgrad_reduce(self, part, nsplit, N, ntaps, Cx, N_real, C_real, swap, I2, KK, khw, grad, accumulate):
        def op():
            s = part[:nsplit * N * ntaps * Cx].view(nsplit, N, ntaps, Cx).sum(0)
            g = grad.view(-1)
            new = g.clone() if accumulate else g.clone()
            for t, k in enumerate(khw.tolist()):
                if k < 0:
                    continue
                n_idx = torch.arange(N_real)[:, None]
                c_idx = torch.arange(C_real)[None, :]
                o = ((c_idx * I2 + n_idx) if swap else (n_idx * I2 + c_idx)) * KK + k
                vals = s[:N_real, t, :C_real]
                new[o.reshape(-1)] = (g[o.reshape(-1)] if accumulate else 0) + vals.reshape(-1)
            g.copy_(new)
        return op

    def pack_weight(self, src, dst, dtype, Nw, ntaps, Cin, N_real, C_real, swap, I2, KK, khw, layout=0, scale=None):
        def op():
            if dtype == 2:      # e4m3 copy: per-tensor scale max|W| / 448 (statement of gan_weight_scale_batch + the fp8 pack)
                am = float(src.abs().max())
                scale.fill_(am / 448.0 if am > 0 else 1.0)
            out = torch.zeros(Nw, ntaps, Cin)
            s = src.reshape(-1).float()
            n_idx = torch.arange(N_real)[:, None]
            c_idx = torch.arange(C_real)[None, :]
            for t, k in enumerate(khw.tolist()):
                if k < 0:
                    continue
                o = ((c_idx * I2 + n_idx) if swap else (n_idx * I2 + c_idx)) * KK + k
                out[:N_real, t, :C_real] = s[o]
            if dtype == 2:
                assert layout == 1
                K = ntaps * Cin // 2                                              # 2-byte slots per row
                q = (out / float(scale)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
                q = q.view(Nw // 16, 16, K // 32, 4, 16).permute(0, 2, 3, 1, 4)     # [n16][k32][fg][fr][8 slots x 2 bytes]
                dst.view(-1).copy_(q.reshape(-1))
                return
            if layout == 1:
                K = ntaps * Cin
                out = out.view(Nw // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4)
            dst.view(-1).copy_(out.reshape(-1).to(dst.dtype))
        op.pack_args = (src, dst, dtype, Nw, ntaps, Cin, N_real, C_real, int(swap), I2, KK, khw, int(layout), scale)
        return op

    def pack_weight_batch(self, packs):
        ops = [self.pack_weight(*a) for a in packs]

        def op():
            for o in ops:
                o()
        return op

    def bias_grad(self, g, N_real, grad, accumulate, ws):
        def op():
            s = g.nhwc().float().sum((0, 1, 2))[:N_real]
            grad.copy_(grad + s if accumulate else s)
        return op

    # ------------------------------------------------------------------ norm / activations / layout
    def in_stats(self, x, eps, stats, ws):
        def op():
            v = x.nhwc().float()
            mean = v.mean((1, 2))
            var = v.var((1, 2), unbiased=False)
            stats.view(x.B, x.C, 2).copy_(torch.stack([mean, 1.0 / torch.sqrt(var + eps)], -1))
        return op

    def in_apply(self, x, stats, act, residual, y, halo_mode):
        def op():
            st = stats.view(x.B, 1, 1, x.C, 2)
            v = _act((x.nhwc().float() - st[..., 0]) * st[..., 1], act)
            if residual is not None:
                v = v + residual.nhwc().float()
            if halo_mode == HALO_REFLECT:
                p = y.halo
                ys = _reflect(torch.arange(-p, y.H + p), y.H)
                xs = _reflect(torch.arange(-p, y.W + p), y.W)
                _store(y, v[:, ys][:, :, xs], padded_coords=True)
            else:
                _store(y, v)
        return op

    def in_bwd(self, x, stats, act, gy, fold, g2, dx, ws):
        def op():
            st = stats.view(x.B, 1, 1, x.C, 2)
            xh = (x.nhwc().float() - st[..., 0]) * st[..., 1]
            g = _fold(gy, fold)
            if g2 is not None:
                g = g + g2.nhwc().float()
            if act == ACT_RELU:
                g = g * (xh > 0)
            elif act == ACT_LRELU:
                g = torch.where(xh > 0, g, 0.2 * g)
            m1 = g.mean((1, 2), keepdim=True)
            m2 = (g * xh).mean((1, 2), keepdim=True)
            _store(dx, st[..., 1] * (g - m1 - xh * m2))
        return op

    def in_bwd_bias(self, x, stats, act, gy, fold, g2, dx, ws, bias_grad, bias_n, accumulate):
        inner = self.in_bwd(x, stats, act, gy, fold, g2, dx, ws)

        def op():
            inner()
            s = dx.nhwc().float().sum((0, 1, 2))[:bias_n]
            bias_grad.copy_(bias_grad + s if accumulate else s)
        return op

    def in_bwd_bias_parts(self, x):
        return x.B

    def in_bwd_bias_deferred(self, x, stats, act, gy, fold, g2, dx, ws, bias_part):
        inner = self.in_bwd(x, stats, act, gy, fold, g2, dx, ws)

        def op():
            inner()
            bias_part[:x.B * x.C].view(x.B, x.C).copy_(dx.nhwc().float().sum((1, 2)))
        return op

    def in_bwd_parts(self, x, stats, act, gy, fold, dx, parts, nparts, parts_mode, bias_part=None):
        """Statement of gan_in_bwd_parts: the apply half with the two sums taken from the producer's partials."""
        def op():
            st = stats.view(x.B, 1, 1, x.C, 2)
            xh = (x.nhwc().float() - st[..., 0]) * st[..., 1]
            g = _fold(gy, fold)
            if act == ACT_RELU:
                g = g * (xh > 0)
            elif act == ACT_LRELU:
                g = torch.where(xh > 0, g, 0.2 * g)
            HW = x.H * x.W
            S = parts[:x.B * nparts * x.C * 2].view(x.B, nparts, x.C, 2).double().sum(1)
            m1 = (S[..., 0] / HW).float().view(x.B, 1, 1, x.C)
            if parts_mode == 1:
                m2 = (S[..., 1] / HW).float().view(x.B, 1, 1, x.C)
            else:
                mu, rs = stats.view(x.B, x.C, 2)[..., 0].double(), stats.view(x.B, x.C, 2)[..., 1].double()
                m2 = (rs * (S[..., 1] - mu * S[..., 0]) / HW).float().view(x.B, 1, 1, x.C)
            _store(dx, st[..., 1] * (g - m1 - xh * m2))
            if bias_part is not None:
                bias_part[:x.B * x.C].view(x.B, x.C).copy_(dx.nhwc().float().sum((1, 2)))
        return op

    def bias_finalize_batch(self, items):
        def op():
            for part, nparts, Cc, grad, n_real, acc in items:
                s = part[:nparts * Cc].view(nparts, Cc).sum(0)[:n_real]
                grad.copy_(grad + s if acc else s)
        return op

    def in_bwd_amax(self, x, stats, act, gy, fold, dx, ws, bias_part, amax):
        """Statement of gan_in_bwd_amax: gan_in_bwd_bias_deferred plus max|dx| per image."""
        inner = self.in_bwd_bias_deferred(x, stats, act, gy, fold, None, dx, ws, bias_part) if bias_part is not None else self.in_bwd(x, stats, act, gy, fold, None, dx, ws)

        def op():
            inner()
            amax[:x.B].copy_(dx.nhwc().float().abs().amax((1, 2, 3)))
        return op

    def quantize_fp8(self, src, dst, amax=None, scale_out=None):
        def op():
            v = src.padded().float()
            if amax is not None:
                sc = torch.where(amax[:src.B] > 0, amax[:src.B] / 448.0, torch.ones_like(amax[:src.B]))
                scale_out[:src.B].copy_(sc)
                v = v / sc.view(src.B, 1, 1, 1)
            dst.padded().copy_(v.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8))
        return op

    def pad_fold(self, g, mode, out):
        def op():
            full = g.padded().float()
            p, H, W = g.halo, g.H, g.W
            idx = (lambda n: _reflect(torch.arange(-p, n + p), n)) if mode == HALO_REFLECT else (lambda n: torch.arange(-p, n + p).clamp(0, n - 1))
            tmp = torch.zeros(g.B, H, g.Wp, g.C)
            tmp.index_add_(1, idx(H), full)
            res = torch.zeros(g.B, H, W, g.C)
            res.index_add_(2, idx(W), tmp)
            _store(out, res)
        return op

    def fold_add(self, a, b, fold, out):
        def op():
            v = _fold(b, fold)
            if a is not None:
                v = v + a.nhwc().float()
            _store(out, v)
        return op

    def act_bwd(self, y, act, g, fold, g2, dx):
        def op():
            v = _fold(g, fold)
            if g2 is not None:
                v = v + g2.nhwc().float()
            _store(dx, v * _act_grad_from_out(y.nhwc().float(), act))
        return op

    def nchw_to_view(self, src, Cr, dst, halo_mode):
        def op():
            v = torch.zeros(dst.B, dst.H, dst.W, dst.C)
            v[..., :Cr] = src.permute(0, 2, 3, 1)
            if halo_mode == HALO_REFLECT:
                p = dst.halo
                ys = _reflect(torch.arange(-p, dst.H + p), dst.H)
                xs = _reflect(torch.arange(-p, dst.W + p), dst.W)
                _store(dst, v[:, ys][:, :, xs], padded_coords=True)
            elif halo_mode == 3:       # replicate
                p = dst.halo
                ys, xs = torch.arange(-p, dst.H + p).clamp(0, dst.H - 1), torch.arange(-p, dst.W + p).clamp(0, dst.W - 1)
                _store(dst, v[:, ys][:, :, xs], padded_coords=True)
            else:
                _store(dst, v)
        return op

    def view_to_nchw(self, src, Cr, dst):
        def op():
            dst.copy_(src.nhwc().float()[..., :Cr].permute(0, 3, 1, 2))
        return op

    def view_copy(self, src, dst, halo_mode):
        def op():
            v = src.nhwc().float()
            if halo_mode == HALO_REFLECT:
                p = dst.halo
                ys = _reflect(torch.arange(-p, dst.H + p), dst.H)
                xs = _reflect(torch.arange(-p, dst.W + p), dst.W)
                _store(dst, v[:, ys][:, :, xs], padded_coords=True)
            else:
                _store(dst, v)
        return op

    def spectral_norm_ws_floats(self, h, w):
        return h + w + 272

    def spectral_norm_fwd(self, W, u, v, power_iter, eps, sigma, Wsn, ws):
        def op():
            m = W.reshape(W.shape[0], -1)
            if power_iter:
                v.copy_(F.normalize(torch.mv(m.t(), u), dim=0, eps=eps))
                u.copy_(F.normalize(torch.mv(m, v), dim=0, eps=eps))
            sigma.fill_(float(torch.dot(u, torch.mv(m, v))))
            Wsn.copy_(W / sigma)
        return op

    def spectral_norm_bwd(self, G, Wsn, u, v, sigma, dW, ws):
        def op():
            dot = (G * Wsn).sum()
            dW.copy_((G.reshape(G.shape[0], -1) - dot * torch.outer(u, v)).reshape(G.shape) / sigma)
        return op

    def avgpool_fwd(self, x, y):
        def op():
            v = x.nhwc().float().permute(0, 3, 1, 2)
            _store(y, F.avg_pool2d(v, 3, 2, 1, count_include_pad=False).permute(0, 2, 3, 1))
        return op

    def avgpool_bwd(self, gy, gx, accumulate):
        def op():
            with torch.enable_grad():
                probe = torch.zeros(gx.B, gx.C, gx.H, gx.W, dtype=torch.float32, requires_grad=True)
                out = F.avg_pool2d(probe, 3, 2, 1, count_include_pad=False)
                g, = torch.autograd.grad(out, probe, gy.nhwc().float().permute(0, 3, 1, 2))
            g = g.permute(0, 2, 3, 1)
            _store(gx, gx.nhwc().float() + g if accumulate else g)
        return op

    # ------------------------------------------------------------------ augmentation and losses
    @staticmethod
    def _aug_masks(prm, B, H, W):
        p = prm.view(B, 12)
        hh = torch.arange(H).view(1, H, 1)
        ww = torch.arange(W).view(1, 1, W)
        tx, ty = p[:, 3].long().view(B, 1, 1), p[:, 4].long().view(B, 1, 1)
        sh, sw = hh + tx, ww + ty
        inr = (sh >= 0) & (sh < H) & (sw >= 0) & (sw < W)
        cut = ((hh >= p[:, 5].long().view(B, 1, 1)) & (hh <= p[:, 6].long().view(B, 1, 1))
               & (ww >= p[:, 7].long().view(B, 1, 1)) & (ww <= p[:, 8].long().view(B, 1, 1)))
        valid = inr & ~cut
        return p, sh.clamp(0, H - 1).expand(B, H, W), sw.clamp(0, W - 1).expand(B, H, W), valid

    def diffaug_fwd(self, x, Cr, prm, y, ws):
        def op():
            B, H, W = x.B, x.H, x.W
            p, sh, sw, valid = self._aug_masks(prm, B, H, W)
            v = x.nhwc().float()[..., :Cr]
            br, sat, con = (p[:, i].view(B, 1, 1, 1) for i in range(3))
            mu = v.mean((1, 2, 3), keepdim=True) + br
            t = v + br
            mc = t.mean(3, keepdim=True)
            s = (t - mc) * sat + mc
            col = (s - mu) * con + mu
            bb = torch.arange(B).view(B, 1, 1).expand(B, H, W)
            out = torch.zeros(B, H, W, y.C)
            out[..., :Cr] = col[bb, sh, sw] * valid.unsqueeze(-1)
            _store(y, out)
        return op

    def diffaug_bwd(self, gy, Cr, prm, gx, ws):
        def op():
            B, H, W = gx.B, gx.H, gx.W
            p, sh, sw, valid = self._aug_masks(prm, B, H, W)
            g = gy.nhwc().float()[..., :Cr] * valid.unsqueeze(-1)
            gsum = g.sum((1, 2, 3), keepdim=True)
            # scatter back to source coordinates (a pure shift: each source pixel is read at most once)
            gs = torch.zeros(B, H, W, Cr)
            bb = torch.arange(B).view(B, 1, 1).expand(B, H, W)
            gs.index_put_((bb[valid], sh[valid], sw[valid]), g[valid], accumulate=True)
            br, sat, con = (p[:, i].view(B, 1, 1, 1) for i in range(3))
            gsat = con * gs + (1 - con) * gsum / (Cr * H * W)
            gt = sat * gsat + (1 - sat) * gsat.mean(3, keepdim=True)
            out = torch.zeros(B, H, W, gx.C)
            out[..., :Cr] = gt
            _store(gx, out)
        return op

    def patch_loss(self, logits, mode, target, scale, loss, grad):
        def op():
            v = logits.nhwc().float()[..., 0]
            n = v.numel()
            if mode == 0:
                f, d = torch.relu(1 - v), -(v < 1).float()
            elif mode == 1:
                f, d = torch.relu(1 + v), (v > -1).float()
            elif mode == 2:
                f, d = -v, -torch.ones_like(v)
            elif mode == 3:
                f, d = (v - target) ** 2, 2 * (v - target)
            else:
                f = torch.relu(v) - v * target + torch.log1p(torch.exp(-v.abs()))
                d = torch.sigmoid(v) - target
            loss.fill_(float(f.sum() * scale / n))
            if grad is not None:
                out = torch.zeros(grad.B, grad.H, grad.W, grad.C)
                out[..., 0] = d * scale / n
                _store(grad, out)
        return op

    def l1_loss(self, x, Cr, target_nchw, scale, dev_scale, loss, grad, ws):
        def op():
            d = x.nhwc().float()[..., :Cr] - target_nchw.permute(0, 2, 3, 1)
            n = d.numel()
            loss.fill_(float(d.abs().sum() * scale / n))
            if grad is not None:
                gs = scale / n * (float(dev_scale) if dev_scale is not None else 1.0)
                out = torch.zeros(grad.B, grad.H, grad.W, grad.C)
                out[..., :Cr] = torch.sign(d) * gs
                _store(grad, out)
        return op

    def r1_reduce(self, g, Cr, scale, loss, u, ws):
        def op():
            v = g.nhwc().float()[..., :Cr]
            loss.fill_(float((v * v).sum() / g.B))
            if u is not None:
                out = torch.zeros(u.B, u.H, u.W, u.C)
                out[..., :Cr] = v * (scale * 2.0 / g.B)
                _store(u, out)
        return op

    def patchnce_ws_floats(self, B, P, Cc):
        return 3 * B * P * Cc + 3 * B * P + ((B + 3) // 4) * 4 + 64

    def _nce(self, src, tgt, ids, Cc, temperature):
        W = tgt.W
        ys, xs = ids.long() // W, ids.long() % W
        t = tgt.nhwc().float()[:, ys, xs, :Cc].clone().requires_grad_(True)
        s = src.nhwc().float()[:, ys, xs, :Cc]
        sn = torch.nn.functional.normalize(s, dim=2, eps=1e-6)
        tn = torch.nn.functional.normalize(t, dim=2, eps=1e-6)
        logits = (torch.bmm(tn, sn.transpose(1, 2)) / temperature).clamp(-50, 50)
        B, P = logits.shape[:2]
        per = torch.nn.functional.cross_entropy(logits.reshape(B * P, P), torch.arange(P).repeat(B), reduction="none").reshape(B, P).mean(1)
        per = torch.where(torch.isfinite(per), per, torch.zeros_like(per))
        return per.sum() / B, t, (ys, xs)

    def patchnce_fwd(self, src, tgt, ids, P, Cc, temperature, weight, loss, ws):
        def op():
            l, _, _ = self._nce(src, tgt, ids, Cc, temperature)
            loss.add_(float(l.detach()) * weight)
            ws[0] = 1.0  # marks "forward ran"; the emulated backward recomputes from src/tgt
            self._nce_src = getattr(self, "_nce_src", {})
            self._nce_src[ws.data_ptr()] = src
        return op

    def patchnce_bwd(self, tgt, ids, P, Cc, temperature, weight, gtgt, ws):
        def op():
            src = self._nce_src[ws.data_ptr()]
            # closed form (no autograd here: the op may run below the dispatcher's autograd layer, where nothing records):
            # dL/dlogits = (softmax - onehot) / (B P), zero where the clamp is active; through the division by T and the normalisation of t
            W_ = tgt.W
            ys, xs = ids.long() // W_, ids.long() % W_
            t = tgt.nhwc().float()[:, ys, xs, :Cc]
            sfeat = src.nhwc().float()[:, ys, xs, :Cc]
            sn = torch.nn.functional.normalize(sfeat, dim=2, eps=1e-6)
            nt = t.norm(dim=2, keepdim=True).clamp_min(1e-6)
            tn = t / nt
            raw = torch.bmm(tn, sn.transpose(1, 2)) / temperature
            logits = raw.clamp(-50, 50)
            B_, P_ = logits.shape[:2]
            per = torch.nn.functional.cross_entropy(logits.reshape(B_ * P_, P_), torch.arange(P_).repeat(B_), reduction="none").reshape(B_, P_).mean(1)
            dlog = (torch.softmax(logits, dim=2) - torch.eye(P_).unsqueeze(0)) / (B_ * P_)
            dlog = dlog * ((raw > -50) & (raw < 50)) * torch.isfinite(per).view(B_, 1, 1)
            dtn = torch.bmm(dlog, sn) / temperature
            g = weight * (dtn - tn * (dtn * tn).sum(2, keepdim=True)) / nt
            g = torch.where(t.norm(dim=2, keepdim=True) > 1e-6, g, weight * dtn / 1e-6)
            buf = gtgt.nhwc()
            acc = buf.float()
            for i in range(ids.numel()):
                acc[:, ys[i], xs[i], :Cc] += g[:, i]
            buf.copy_(acc.to(buf.dtype))
        return op

    # ------------------------------------------------------------------ optimiser
    def make_adam_table(self, entries):
        return entries  # the emulator keeps the python dicts

    def adam_step(self, table, ntensors, chunk_tensor, chunk_off, nchunks, lr, b1, b2, eps, max_norm, grad_scale, ema_decay, norm_out, ws,
                  lr_dev=None, inv_scale=None, skip_nonfinite=False):
        def op():
            gs = grad_scale * (float(inv_scale) if inv_scale is not None else 1.0)
            rate = float(lr_dev) if lr_dev is not None else lr
            live = [e for e in table if e.get("g") is not None]
            tot = math.sqrt(sum(float(((e["g"] * gs) ** 2).sum()) for e in live))
            coef = min(1.0, max_norm / (tot + 1e-6)) if max_norm > 0 else 1.0
            found = not math.isfinite(tot)
            norm_out[0], norm_out[1], norm_out[2] = tot, coef, float(found)
            if skip_nonfinite and found:
                return
            for e in live:
                g = e["g"] * (gs * coef)
                t = int(e["step"]) + 1
                e["step"].fill_(t)
                e["m"].lerp_(g, 1 - b1)
                e["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
                bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
                denom = (e["v"].sqrt() / math.sqrt(bc2)).add_(eps)
                e["p"].addcdiv_(e["m"], denom, value=-(rate / bc1))
                if e.get("ema") is not None:
                    e["ema"].copy_((1.0 - ema_decay) * e["p"] + ema_decay * e["ema"])
        return op

    def scaler_update(self, scale, inv_scale, tracker, found_inf, growth=2.0, backoff=0.5, interval=2000):
        def op():
            if float(found_inf) != 0.0:
                scale.mul_(backoff); tracker.zero_()
            elif int(tracker) + 1 >= interval:
                scale.mul_(growth); tracker.zero_()
            else:
                tracker.add_(1)
            inv_scale.copy_(1.0 / scale)
        return op

    def fill(self, t, value):
        return lambda: t.fill_(value)

    def axpy(self, y, x, a):
        return lambda: y.add_(x, alpha=a)

    def zero_(self, t):
        return lambda: t.zero_()
