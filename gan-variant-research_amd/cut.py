"""Host-side mirror of the reference's GAN_Variant1 (CUT) trainer on the MI355X kernels.

Same names, argument meaning and state_dict keys as the reference (paths relative to its root):
ResNetGenerator / MultiscaleDiscriminator (models/generator_resnet_attn.py:74-235, models/discriminator_patchgan.py:75-128),
DiffAugment (training/diffaugment.py:76-106), get_optimizer (training/sched_optim.py:5-27), EMA (utils/io_ckpt.py:9-53),
AMPContext (utils/amp_utils.py:5-41), build_models / train_step (training/train_cutpp.py:88-124, 206-331).

The modules are parameter containers (default PyTorch init on the CPU generator, reference key names); all
arithmetic runs in libmi355x_gan.so through CutTrainer, which owns the buffers and the replayable step programs.
"""
from __future__ import annotations

import math
import os
import random
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from ._lib import ACT_NONE, BF16, F32, HALO_NONE, HALO_ZERO
from .nets import DiscriminatorNet, GeneratorNet
from .runtime import ADAM_CHUNK, Ctx, HipOps, Program, View, cpad

NCE_LAYERS_DEFAULT = (0, 4, 8, 12, 16)


def set_seed(seed: int):
    """utils/seed_dist.py:7-12."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


# ------------------------------------------------------------------------------------------------
# parameter containers with the reference's module tree (so state_dict keys and init order match)
# ------------------------------------------------------------------------------------------------
def _notify_weights_changed():
    from . import autograd as AG
    AG.notify_weights_changed()


def _slot(n):
    return [nn.Identity() for _ in range(n)]


class _ResBlockParams(nn.Module):
    def __init__(self, c, reflect=True):
        super().__init__()
        if reflect:      # [pad, conv, norm, act, pad, conv, norm] (generator_resnet_attn.py:24-52)
            seq = _slot(7)
            seq[1], seq[5] = nn.Conv2d(c, c, 3), nn.Conv2d(c, c, 3)
        else:            # zero padding lives in the convolutions: [conv, norm, act, conv, norm]
            seq = _slot(5)
            seq[0], seq[3] = nn.Conv2d(c, c, 3, padding=1), nn.Conv2d(c, c, 3, padding=1)
        self.conv_block = nn.Sequential(*seq)


class ResNetGenerator(nn.Module):
    """Signature of models/generator_resnet_attn.py:86-103 (the attention/style kwargs are accepted and ignored there too)."""

    def __init__(self, input_nc=3, output_nc=3, ngf=64, n_blocks=9, n_downsampling=2, padding_type="reflect", norm="instance",
                 activation="relu", use_attention=True, attn_layers=(3, 7), use_channel_attn=True, channel_attn_layers=(5,),
                 use_style_dropout=True, alpha_min=0.4, alpha_max=0.9):
        super().__init__()
        self.input_nc, self.output_nc, self.ngf, self.n_blocks = input_nc, output_nc, ngf, n_blocks
        self.padding_type, self.activation, self.norm, self.n_downsampling = padding_type, activation, norm, n_downsampling
        self.compute_dtype = F32
        # The switches the shipped configs leave at their defaults but the reference's constructor offers (generator_resnet_attn.py:24-66,
        # 110-162): replicate padding, norm 'batch' | 'none' (any other string is 'none' there too), no block activation, other depths.
        # The module-granular engine (nets.GeneratorNet) is built for instance norm / reflect | zero / two down-samplings; everything else
        # is assembled LAYER BY LAYER from the op-level modules (ops_library), with the reference's Sequential indices and state_dict keys.
        self._layerwise = not (padding_type in ("reflect", "zero") and activation in ("relu", "leaky_relu") and norm == "instance" and n_downsampling == 2)
        if self._layerwise:
            self._build_layerwise()
            return
        rf = padding_type == "reflect"
        if rf:
            s = _slot(4); s[1] = nn.Conv2d(input_nc, ngf, 7)
        else:            # no pad module in front (generator_resnet_attn.py:110-116): the conv sits at index 0
            s = _slot(3); s[0] = nn.Conv2d(input_nc, ngf, 7, padding=3)
        self.initial = nn.Sequential(*s)
        s = _slot(6); s[0], s[3] = nn.Conv2d(ngf, 2 * ngf, 3, stride=2, padding=1), nn.Conv2d(2 * ngf, 4 * ngf, 3, stride=2, padding=1)
        self.downsample = nn.Sequential(*s)
        self.res_blocks = nn.ModuleList([_ResBlockParams(4 * ngf, rf) for _ in range(n_blocks)])
        s = _slot(6)
        s[0] = nn.ConvTranspose2d(4 * ngf, 2 * ngf, 3, stride=2, padding=1, output_padding=1)
        s[3] = nn.ConvTranspose2d(2 * ngf, ngf, 3, stride=2, padding=1, output_padding=1)
        self.upsample = nn.Sequential(*s)
        if rf:
            s = _slot(3); s[1] = nn.Conv2d(ngf, output_nc, 7)
        else:
            s = _slot(2); s[0] = nn.Conv2d(ngf, output_nc, 7, padding=3)
        self.output = nn.Sequential(*s)
        self.compute_dtype = F32          # BF16: bf16 operands, fp32 accumulation (throughput mode)

    def _build_layerwise(self):
        """The reference's module tree (generator_resnet_attn.py:104-163, ResidualBlock :19-52) from the op-level modules."""
        from . import ops_library as L
        pt, norm, ngf, nd = self.padding_type, self.norm, self.ngf, self.n_downsampling
        outer_norm = (lambda c: L.InstanceNorm2d(c)) if norm == "instance" else (lambda c: nn.Identity())     # :114,126,150: instance or nothing
        blk_norm = (lambda c: L.InstanceNorm2d(c)) if norm == "instance" else (lambda c: L.BatchNorm2d(c)) if norm == "batch" else (lambda c: nn.Identity())
        blk_act = (lambda: nn.ReLU(True)) if self.activation == "relu" else (lambda: nn.LeakyReLU(0.2, True)) if self.activation == "leaky_relu" else (lambda: nn.Identity())
        m = [L.ReflectionPad2d(3)] if pt == "reflect" else []
        m += [L.Conv2d(self.input_nc, ngf, 7, padding=0 if pt == "reflect" else 3), outer_norm(ngf), nn.ReLU(True)]
        self.initial = nn.Sequential(*m)
        m = []
        for i in range(nd):
            m += [L.Conv2d(ngf * 2 ** i, ngf * 2 ** (i + 1), 3, stride=2, padding=1), outer_norm(ngf * 2 ** (i + 1)), nn.ReLU(True)]
        self.downsample = nn.Sequential(*m)
        c = ngf * 2 ** nd

        class _Block(nn.Module):
            def __init__(blk):
                super().__init__()
                pad = (lambda: [L.ReflectionPad2d(1)]) if pt == "reflect" else (lambda: [L.ReplicationPad2d(1)]) if pt == "replicate" else (lambda: [])
                cp = 1 if pt == "zero" else 0          # :36,49 of the reference: only 'zero' pads inside the convolution
                blk.conv_block = nn.Sequential(*(pad() + [L.Conv2d(c, c, 3, padding=cp), blk_norm(c), blk_act()] + pad() + [L.Conv2d(c, c, 3, padding=cp), blk_norm(c)]))

            def forward(blk, x):
                return x + blk.conv_block(x)
        self.res_blocks = nn.ModuleList([_Block() for _ in range(self.n_blocks)])
        m = []
        for i in range(nd):
            cc = ngf * 2 ** (nd - i)
            m += [L.ConvTranspose2d(cc, cc // 2, 3, stride=2, padding=1, output_padding=1), outer_norm(cc // 2), nn.ReLU(True)]
        self.upsample = nn.Sequential(*m)
        m = [L.ReflectionPad2d(3)] if pt == "reflect" else []
        m += [L.Conv2d(ngf, self.output_nc, 7, padding=0 if pt == "reflect" else 3), nn.Tanh()]
        self.output = nn.Sequential(*m)

    def _set_layer_dtype(self):
        from . import ops_library as L
        L.set_compute_dtype(self.compute_dtype)

    def forward(self, x):
        """(B,3,H,W) fp32 in [-1,1] -> (B,3,H,W) fp32 on the HIP kernels, differentiable (autograd.py): one autograd node per call.
        The fused CutTrainer is the fast training path; this is the drop-in nn.Module path."""
        if self._layerwise:            # generator_resnet_attn.py:165-188, op by op
            self._set_layer_dtype()
            x = self.downsample(self.initial(x))
            for blk in self.res_blocks:
                x = blk(x)
            return self.output(self.upsample(x))
        from . import autograd as AG
        return AG.generator_forward(self, x, "cut")

    def get_feature_layers(self, x, layer_ids=None):
        """generator_resnet_attn.py:190-235: numbered activations; ids beyond the last one are silently ignored there too."""
        if self._layerwise:
            self._set_layer_dtype()
            ids = list(NCE_LAYERS_DEFAULT) if layer_ids is None else list(layer_ids)
            feats, idx = [], 0
            x = self.initial(x)
            if idx in ids:
                feats.append(x)
            idx += 1
            for seq in (self.downsample, None, self.upsample):
                if seq is None:
                    for blk in self.res_blocks:
                        x = blk(x)
                        if idx in ids:
                            feats.append(x)
                        idx += 1
                    continue
                for mod in seq:
                    x = mod(x)
                    if isinstance(mod, nn.ReLU):
                        if idx in ids:
                            feats.append(x)
                        idx += 1
            return feats
        from . import autograd as AG
        ids = feature_layers_present(list(NCE_LAYERS_DEFAULT) if layer_ids is None else list(layer_ids), self.n_blocks)
        return AG.generator_forward(self, x, "cut", ids)


class _PatchGANParams(nn.Module):
    def __init__(self, input_nc, ndf, n_layers, use_spectral_norm=False):
        super().__init__()
        chans = [input_nc, ndf] + [ndf * min(2**n, 8) for n in range(1, n_layers)] + [ndf * min(2**n_layers, 8), 1]
        seq = []
        for i in range(len(chans) - 1):
            conv = nn.Conv2d(chans[i], chans[i + 1], 4, stride=2 if i < n_layers else 1, padding=1)
            if use_spectral_norm:
                # registration only (weight_orig / weight_u / weight_v, their initial draws and state_dict handling are torch's,
                # discriminator_patchgan.py:21-23); the module's forward hook never runs here: the power iteration and the
                # normalisation are gan_spectral_norm_fwd / _bwd (autograd.py)
                conv = nn.utils.spectral_norm(conv)
            seq.append(conv)
            if i < len(chans) - 2:
                seq.append(nn.Identity())
        self.model = nn.Sequential(*seq)


class MultiscaleDiscriminator(nn.Module):
    """Signature and defaults of models/discriminator_patchgan.py:81-88.  The baseline config (one scale, no spectral norm) is what the
    fused CutTrainer runs; more scales and spectral norm run through this module API (autograd.py)."""

    def __init__(self, input_nc=3, ndf=64, n_layers=3, num_scales=3, use_spectral_norm=True):
        super().__init__()
        self.input_nc, self.ndf, self.n_layers, self.num_scales, self.use_spectral_norm = input_nc, ndf, n_layers, num_scales, use_spectral_norm
        self.discriminators = nn.ModuleList([_PatchGANParams(input_nc, ndf, n_layers, use_spectral_norm) for _ in range(num_scales)])
        self.compute_dtype = F32

    def forward(self, x):
        """discriminator_patchgan.py:102-116: list of per-scale logits, scale i on the input average-pooled i times
        (AvgPool2d(3, 2, 1, count_include_pad=False), :100); differentiable, one autograd node (autograd.py)."""
        from . import autograd as AG
        return AG.discriminator_forward(self, x, "cut", [f"discriminators.{i}.model." for i in range(self.num_scales)], self.ndf, self.n_layers)


def build_models(config, device):
    """training/train_cutpp.py:88-124: construction on the CPU generator, then .to(device)."""
    g, d = config["model"]["generator"], config["model"]["discriminator"]
    gen = ResNetGenerator(3, 3, g["ngf"], g["n_blocks"], g["n_downsampling"], g["padding_type"], g["norm"], g["activation"]).to(device)
    disc = MultiscaleDiscriminator(3, d["ndf"], d["n_layers"], d["num_scales"], d.get("use_spectral_norm", True)).to(device)
    return gen, disc


class DiffAugment:
    """training/diffaugment.py:76-106.  Holds the policy; the draws are made by CutTrainer (or injected for parity)."""

    def __init__(self, policy: Optional[Sequence[str]] = None):
        self.policy = list(policy) if policy is not None else ["color", "translation", "cutout_light"]

    def sample(self, B, H, W, generator=None) -> Dict[str, torch.Tensor]:
        """Per-sample draws in the reference's order and shapes (diffaugment.py:8,15,22,29-30,47-48), on the CPU generator."""
        d = {}
        for pol in self.policy:
            if pol == "color":
                for k in ("brightness", "saturation", "contrast"):
                    d[k] = torch.rand(B, 1, 1, 1, generator=generator)
            elif pol == "translation":
                sx, sy = int(H * 0.125 + 0.5), int(W * 0.125 + 0.5)
                d["tx"] = torch.randint(-sx, sx + 1, size=[B, 1, 1], generator=generator)
                d["ty"] = torch.randint(-sy, sy + 1, size=[B, 1, 1], generator=generator)
            elif pol in ("cutout", "cutout_light"):
                ratio = 0.5 if pol == "cutout" else 0.2
                ch, cw = int(H * ratio + 0.5), int(W * ratio + 0.5)
                d["cut_h"], d["cut_w"] = torch.tensor(ch), torch.tensor(cw)
                d["cx"] = torch.randint(0, H + (1 - ch % 2), size=[B, 1, 1], generator=generator)
                d["cy"] = torch.randint(0, W + (1 - cw % 2), size=[B, 1, 1], generator=generator)
        return d

    @staticmethod
    def to_params(d: Dict[str, torch.Tensor], B, H, W) -> torch.Tensor:
        """Draws -> the [B][12] fp32 table of gan_diffaug_*: brightness add, saturation and contrast factors, translation,
        inclusive cutout window (rows/cols are clamped exactly as diffaugment.py:55-58 does)."""
        p = torch.zeros(B, 12, dtype=torch.float32)
        p[:, 1] = 1.0
        p[:, 2] = 1.0
        p[:, 5], p[:, 6], p[:, 7], p[:, 8] = 1, 0, 1, 0   # empty window
        if "brightness" in d:
            p[:, 0] = d["brightness"].reshape(B).float() - 0.5
            p[:, 1] = d["saturation"].reshape(B).float() * 2
            p[:, 2] = d["contrast"].reshape(B).float() + 0.5
        if "tx" in d:
            p[:, 3], p[:, 4] = d["tx"].reshape(B).float(), d["ty"].reshape(B).float()
        if "cx" in d:
            ch, cw = int(d["cut_h"]), int(d["cut_w"])
            cx, cy = d["cx"].reshape(B), d["cy"].reshape(B)
            p[:, 5] = (cx - ch // 2).clamp(0, H - 1).float()
            p[:, 6] = (cx + ch - 1 - ch // 2).clamp(0, H - 1).float()
            p[:, 7] = (cy - cw // 2).clamp(0, W - 1).float()
            p[:, 8] = (cy + cw - 1 - cw // 2).clamp(0, W - 1).float()
        return p


class AMPContext:
    """utils/amp_utils.py:5-41.  On the MI355X path "amp" selects bf16 operands with fp32 accumulation; there is no
    GradScaler (bf16 has fp32's exponent range), so scale == 1 and found_inf never skips a step."""

    def __init__(self, enabled: bool = True):
        self.enabled = enabled

    @property
    def dtype(self):
        return BF16 if self.enabled else F32


class FusedAdam:
    """get_optimizer's torch.optim.Adam (sched_optim.py:5-27) over a flat parameter block, fused with clip_grad_norm_
    and EMA.update (amp_utils.py:29-41, io_ckpt.py:23-29) in one multi-tensor launch."""

    def __init__(self, ctx: Ctx, names: List[str], shapes: List[torch.Size], init: Dict[str, torch.Tensor], lr=2e-4, betas=(0.5, 0.999),
                 eps=1e-8, weight_decay=0.0, ema_decay: Optional[float] = None):
        if weight_decay != 0.0:
            raise NotImplementedError("weight_decay != 0 is not used by the reference configs")
        self.ctx, self.names, self.lr, self.betas, self.eps, self.ema_decay = ctx, names, lr, betas, eps, ema_decay
        sizes = [int(np.prod(s)) for s in shapes]
        self.offsets = np.concatenate([[0], np.cumsum([(n + 3) // 4 * 4 for n in sizes])]).astype(np.int64)  # 16-byte aligned slices
        total = int(self.offsets[-1])
        dev = ctx.device
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self.flat_ema = torch.zeros_like(self.flat_p) if ema_decay is not None else None
        self.steps = torch.zeros(len(names), dtype=torch.int32, device=dev)
        self.params, self.grads, self.shadow = {}, {}, {}
        for i, (n, shp, sz) in enumerate(zip(names, shapes, sizes)):
            o = int(self.offsets[i])
            self.params[n] = self.flat_p[o:o + sz].view(shp)
            self.grads[n] = self.flat_g[o:o + sz].view(shp)
            self.params[n].copy_(init[n])
            if self.flat_ema is not None:
                self.shadow[n] = self.flat_ema[o:o + sz].view(shp)
                self.shadow[n].copy_(init[n])
        self.sizes = sizes
        ct, co = [], []
        for i, sz in enumerate(sizes):
            for off in range(0, sz, ADAM_CHUNK):
                ct.append(i)
                co.append(off)
        self.nchunks = len(ct)
        self.chunk_tensor = torch.tensor(ct, dtype=torch.int32, device=dev)
        self.chunk_off = torch.tensor(co, dtype=torch.int64, device=dev)
        self.norm_out = torch.zeros(4, dtype=torch.float32, device=dev)
        self.ws = torch.zeros(self.nchunks + 16, dtype=torch.float32, device=dev)
        self._tables = {}
        # the learning rate lives in one device float that the prebuilt launches read (gan_adam_step lr_dev): a scheduler
        # (Basic_GAN/src/train.py:54-58,125: LambdaLR) rewrites it between epochs without rebuilding any program
        self.base_lr = float(lr)
        self.lr_dev = torch.full((1,), float(lr), dtype=torch.float32, device=dev)

    def set_lr(self, lr: float):
        """What LambdaLR.step() does to param_groups[0]['lr'] (train.py:125): takes effect with the next step."""
        self.lr = float(lr)
        self.lr_dev.fill_(float(lr))

    def table(self, skip: Sequence[str] = ()) -> torch.Tensor:
        key = tuple(sorted(skip))
        if key not in self._tables:
            ents = []
            for i, n in enumerate(self.names):
                o, sz = int(self.offsets[i]), self.sizes[i]
                ents.append({"p": self.flat_p[o:o + sz], "g": None if n in skip else self.flat_g[o:o + sz], "m": self.flat_m[o:o + sz],
                             "v": self.flat_v[o:o + sz], "ema": self.flat_ema[o:o + sz] if self.flat_ema is not None else None,
                             "step": self.steps[i:i + 1]})
            self._tables[key] = self.ctx.ops.make_adam_table(ents)
        return self._tables[key]

    def step_op(self, max_norm: Optional[float], grad_scale: float = 1.0, skip: Sequence[str] = ()):
        return self.ctx.ops.adam_step(self.table(skip), len(self.names), self.chunk_tensor, self.chunk_off, self.nchunks, self.lr,
                                      self.betas[0], self.betas[1], self.eps, max_norm if max_norm is not None else 0.0, grad_scale,
                                      self.ema_decay if self.ema_decay is not None else 0.0, self.norm_out, self.ws, lr_dev=self.lr_dev)


def _adam_state_dict(opt: "FusedAdam", with_initial_lr: bool = False) -> dict:
    """torch.optim.Adam.state_dict() layout (what utils/io_ckpt.py:70-71 stores), parameters numbered in state_dict order."""
    steps = opt.steps.cpu()
    state = {}
    for i, n in enumerate(opt.names):
        o, sz = int(opt.offsets[i]), opt.sizes[i]
        shp = opt.params[n].shape
        state[i] = {"step": steps[i].float().reshape(()), "exp_avg": opt.flat_m[o:o + sz].view(shp).clone(),
                    "exp_avg_sq": opt.flat_v[o:o + sz].view(shp).clone()}
    group = {"lr": opt.lr, "betas": tuple(opt.betas), "eps": opt.eps, "weight_decay": 0.0, "amsgrad": False, "maximize": False, "foreach": None,
             "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(opt.names)))}
    if with_initial_lr:      # torch.optim.lr_scheduler.LambdaLR adds it to every param group it schedules (Basic_GAN/src/train.py:54-58)
        group["initial_lr"] = opt.base_lr
    return {"state": state, "param_groups": [group]}


def _load_adam_state_dict(opt: "FusedAdam", sd: dict):
    opt.flat_m.zero_(); opt.flat_v.zero_(); opt.steps.zero_()
    steps = torch.zeros(len(opt.names), dtype=torch.int32)
    for i, st in sd["state"].items():
        i = int(i)
        n = opt.names[i]
        o, sz = int(opt.offsets[i]), opt.sizes[i]
        opt.flat_m[o:o + sz].copy_(st["exp_avg"].reshape(-1))
        opt.flat_v[o:o + sz].copy_(st["exp_avg_sq"].reshape(-1))
        steps[i] = int(float(st["step"]))
    opt.steps.copy_(steps)
    g = sd["param_groups"][0]
    if (tuple(g["betas"]), g["eps"]) != (tuple(opt.betas), opt.eps):
        raise ValueError("checkpoint optimiser hyper-parameters differ from the trainer's config (they are baked into its programs)")
    if g["lr"] != opt.lr:       # a scheduler had moved it (Basic_GAN's LambdaLR): the device scalar follows the checkpoint
        opt.set_lr(g["lr"])
    if "initial_lr" in g:
        opt.base_lr = float(g["initial_lr"])


def get_optimizer_config(opt_config: dict) -> dict:
    """sched_optim.py:16-18 defaults."""
    return {"lr": opt_config.get("lr", 2e-4), "betas": tuple(opt_config.get("betas", [0.5, 0.999])),
            "weight_decay": opt_config.get("weight_decay", 0.0)}


def identity_weight_at(step: int, config: dict) -> float:
    lw, ws = config["loss_weights"], config.get("warmup_steps", 20000)   # train_cutpp.py:224-228
    if step < ws:
        return lw["identity_warm"] + (lw["identity_final"] - lw["identity_warm"]) * (step / ws)
    return lw["identity_final"]


def feature_layers_present(layer_ids, n_blocks=9, n_down=2) -> List[int]:
    """get_feature_layers silently ignores ids beyond the last numbered activation (generator_resnet_attn.py:204-233)."""
    return [i for i in range(1 + n_down + n_blocks + n_down) if i in layer_ids]


# ------------------------------------------------------------------------------------------------
# the trainer
# ------------------------------------------------------------------------------------------------
LOSS_SLOTS = {"d_real": 0, "d_fake": 1, "r1": 2, "g_adv": 3, "nce": 4, "identity": 5, "scratch": 6, "idw": 7}


class CutTrainer:
    """Owns parameters, optimiser state, activation buffers and the step programs of one rank.

    train_step(step, photos, monets) reproduces training/train_cutpp.py:206-331 with two legal savings (results
    unchanged): G(photos) is computed once per step and shared by the D-step, the G-step and the PatchNCE source
    features (generator weights do not change in between), and the D weight gradients the reference's G-step
    backward deposits (and its next zero_grad discards) are never computed.
    """

    def __init__(self, generator: ResNetGenerator, discriminator: MultiscaleDiscriminator, config: dict, batch_size: int, image_size: int,
                 device="cuda", amp: Optional[bool] = None, ops=None, world_size: int = 1, process_group=None, fp8: Optional[bool] = None):
        """fp8 (default: config['mi355x']['fp8'], else False): the residual blocks' convolutions read e4m3 operand copies in the forward pass
        and in the input gradient (BASELINE.json configs[4]); needs amp (bf16) -- see nets.GeneratorNet."""
        self.config = config
        self.B, self.S = batch_size, image_size
        self.device = torch.device(device)
        amp = config.get("amp", True) if amp is None else amp
        self.amp = AMPContext(amp)
        self.fp8 = bool((config.get("mi355x") or {}).get("fp8", False) if fp8 is None else fp8)
        if self.fp8 and not amp:
            raise ValueError("fp8 convolutions exist in the bf16 (amp) mode only: fp32 is the parity mode")
        self.ops = ops if ops is not None else HipOps(self.device)
        if hasattr(self.ops, "bind"):
            self.ops.bind()                 # one stream for this trainer's launches and its torch-side copies / events, from now on
        if hasattr(self.ops, "bind_queues"):
            self.ops.bind_queues()          # no-op cost; matters when this runs before torch.distributed creates RCCL's streams
        self.ctx = Ctx(self.ops, self.device, self.amp.dtype)
        # The discriminator lives on its own HIP stream: its step (forward / backward / R1 / Adam) and its forward-backward for the
        # generator's adversarial gradient depend on the generator only through the fake image, so they run concurrently with the
        # PatchNCE feature pass and its backward; two events per step order the streams (train_step).
        self.opsD = self.ops.fork()
        self.ctxD = Ctx(self.opsD, self.device, self.amp.dtype) if self.opsD is not self.ops else self.ctx
        self.ctx32 = self.ctxD if self.amp.dtype == F32 else Ctx(self.opsD, self.device, F32)
        self.world_size, self.pg = world_size, process_group
        lw = config["loss_weights"]
        self.policy = config["diffaugment"].get("policy", ["color", "translation", "cutout"]) if config["diffaugment"].get("enable", False) else None
        self.aug = DiffAugment(self.policy) if self.policy is not None else None
        self.generator, self.discriminator = generator, discriminator
        if getattr(generator, "_layerwise", False) or getattr(generator, "padding_type", "reflect") != "reflect" or getattr(generator, "activation", "relu") != "relu":
            raise NotImplementedError("the fused CutTrainer runs the reference configuration (reflect padding, ReLU blocks); "
                                      "module_step.train_step drives the other generator variants on the same kernels")
        if getattr(discriminator, "num_scales", 1) != 1 or getattr(discriminator, "use_spectral_norm", False):
            raise NotImplementedError("the fused CutTrainer runs the reference configuration (one discriminator scale, no spectral norm); "
                                      "module_step.train_step drives the optional discriminator variants on the same kernels")

        # ---- parameters -> flat fp32 blocks (master weights, grads, Adam moments, EMA shadow)
        gsd = {k: v.detach().to(self.device, torch.float32) for k, v in generator.state_dict().items()}
        dsd = {k: v.detach().to(self.device, torch.float32) for k, v in discriminator.state_dict().items()}
        og, od = get_optimizer_config(config["optim"]["G"]), get_optimizer_config(config["optim"]["D"])
        self.opt_G = FusedAdam(self.ctx, list(gsd), [v.shape for v in gsd.values()], gsd, og["lr"], og["betas"], 1e-8, og["weight_decay"],
                               ema_decay=config["ema"]["decay"])
        self.opt_D = FusedAdam(self.ctxD, list(dsd), [v.shape for v in dsd.values()], dsd, od["lr"], od["betas"], 1e-8, od["weight_decay"])
        for mod, opt in ((generator, self.opt_G), (discriminator, self.opt_D)):   # modules now alias the trained block
            for k, p in mod.named_parameters():
                p.data = opt.params[k]

        B, S = self.B, self.S
        nb, ngf = generator.n_blocks, generator.ngf
        self.G = GeneratorNet(self.ctx, self.opt_G.params, self.opt_G.grads, "cut", nb, ngf, need_input_grad=True, fp8=self.fp8)
        self.D = DiscriminatorNet(self.ctxD, self.opt_D.params, self.opt_D.grads, "cut", ndf=discriminator.ndf, n_layers=discriminator.n_layers)
        self.D32 = self.D if self.ctx32 is self.ctxD else DiscriminatorNet(self.ctx32, self.opt_D.params, self.opt_D.grads, "cut",
                                                                          ndf=discriminator.ndf, n_layers=discriminator.n_layers)
        self.nce_layers = feature_layers_present(config["patchnce"]["nce_layers"], nb) if lw["patchnce"] > 0 else []
        self.P = config["patchnce"]["num_patches"]
        # Identity warm-up (identity weight > 0): G(photos) and G(monets) are two full passes through the same weights, and
        # InstanceNorm is per sample -> they run as ONE pass over 2B images (half the launches, one split-K reduction per layer);
        # after the warm-up the generator pass holds the photos only.  `merge_identity_pass: false` keeps three separate passes.
        self.merge_identity = bool(config.get("merge_identity_pass", True))
        self.p2 = self.G.new_pass(B, S, S, last_layer=max(self.nce_layers)) if self.nce_layers else None
        self.d_rf, self.d_fake = self.D.new_pass(2 * B, S, S), self.D.new_pass(B, S, S)   # D-step pass (real | fake), G-step pass
        self.d_r1 = self.D32.new_pass(B, S, S)

        f32 = self.ctx.f32
        self.both = torch.zeros(2 * B, 3, S, S, dtype=torch.float32, device=self.device)   # photos | monets, one staging tensor
        self.photos, self.monets = self.both[:B], self.both[B:]
        self.fake_out = torch.zeros_like(self.photos)
        self.losses = f32(16)
        self.nce_hw = [(S >> (0 if i == 0 else 1 if i == 1 else 2 if i < 3 + nb else 1 if i == 3 + nb else 0)) ** 2 for i in self.nce_layers]
        # every per-step random draw (3 DiffAugment tables, PatchNCE ids) lives in ONE device block filled by ONE non-blocking copy
        # from pinned memory: seven small pageable copies would each stall the host until the stream drains (measured: 7 idle gaps
        # of ~0.1 ms per step and no host lead over the GPU)
        seg = [B * 12] * 3 + [min(self.P, hw) for hw in self.nce_hw]
        offs = [0]
        for n in seg:
            offs.append(offs[-1] + (n + 63) // 64 * 64)
        self._rnd_dev = torch.zeros(offs[-1], dtype=torch.int32, device=self.device)
        self._rnd_host = [torch.zeros(offs[-1], dtype=torch.int32, pin_memory=self.device.type == "cuda") for _ in range(2)]
        self._rnd_events = [None, None]
        self._rnd_turn = 0
        cut = lambda t, i: t[offs[i]:offs[i] + seg[i]]
        self.prm = {k: cut(self._rnd_dev, i).view(torch.float32) for i, k in enumerate(("real", "fake_d", "fake_g"))}
        self.nce_ids = [cut(self._rnd_dev, 3 + i) for i in range(len(self.nce_hw))]
        self._rnd_host_views = [({k: cut(h, i).view(torch.float32) for i, k in enumerate(("real", "fake_d", "fake_g"))},
                                 [cut(h, 3 + i) for i in range(len(self.nce_hw))]) for h in self._rnd_host]
        self._modes = {}
        self._use_mode(self.merge_identity and identity_weight_at(0, config) > 0)

    # ------------------------------------------------------------------ program construction
    def _slot(self, name) -> torch.Tensor:
        i = LOSS_SLOTS[name]
        return self.losses[i:i + 1]

    def _aug_fwd(self, src: View, dst: View, prm) -> list:
        ops = self.opsD                          # DiffAugment feeds the discriminator: its stream
        if self.aug is None:
            return [ops.view_copy(src, dst, HALO_ZERO)]
        return [ops.diffaug_fwd(src, 3, prm, dst, self.ctxD.scratch("aug_ws", self.B + 16))]

    def _use_mode(self, merged: bool):
        """Makes the step programs of one mode current (building them on first use).  merged: the identity pass rides in the
        generator pass (2B images); otherwise G(photos) and G(monets) are separate passes and the latter is optional."""
        if merged not in self._modes:
            self._modes[merged] = self._build_mode(merged)
            self._build_updates()            # repack programs must see every operand copy planned so far
            self._device_sync()              # parameters may have been written on another stream (construction, checkpoint load)
            self.G.repack_program().run()
            self.D.repack_program().run()
            self._device_sync()
        m = self._modes[merged]
        self.mode_merged = merged
        for k, val in m.items():
            setattr(self, k, val)

    def side_streams(self):
        """Every torch stream other than the caller's current one that this trainer's programs launch on."""
        out = []
        for o in (self.ops, self.opsD):
            for h in (o, getattr(o, "_side", None)):
                ts = getattr(h, "torch_stream", None) if h is not None else None
                if ts is not None and ts not in out:
                    out.append(ts)
        return out

    def _device_sync(self):
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def _build_mode(self, merged: bool) -> dict:
        cfg, ops, ctx, B, S = self.config, self.ops, self.ctx, self.B, self.S
        lw = cfg["loss_weights"]
        if merged:
            p1 = self.G.new_pass(2 * B, S, S)
            p3 = None
            fake = p1.img.batch(0, B)
            prog_gfwd = p1.fwd_program(self.both)
        else:
            p1, p3 = self.G.new_pass(B, S, S), self.G.new_pass(B, S, S)
            fake = p1.img
            prog_gfwd = p1.fwd_program(self.photos)

        def src_feat(li):
            return p1.acts[li].batch(0, B) if merged else p1.acts[li]
        # photos as a plain C=8 image view (DiffAugment / R1 input)
        photos_v = ctx.view(B, S, S, 8, 0)
        prog_gfwd.add(ops.nchw_to_view(self.photos, 3, photos_v, HALO_ZERO))

        # ---- D step (train_cutpp.py:231-254): D(aug(photos)) and D(aug(fake.detach())) as ONE pass over 2B images (the
        # discriminator has no cross-sample statistics); the two hinge terms read / write their halves of the logits
        pd = Program("D-step")
        drf = self.d_rf
        pd.add(self._aug_fwd(photos_v, drf.x.batch(0, B), self.prm["real"]))
        pd.add(self._aug_fwd(fake, drf.x.batch(B, B), self.prm["fake_d"]))
        pd.add(drf.fwd_program())
        gl = drf.grad_logits_view()
        pd.add(self.opsD.patch_loss(drf.logits.batch(0, B), 0, 0.0, 0.5, self._slot("d_real"), gl.batch(0, B)))
        pd.add(self.opsD.patch_loss(drf.logits.batch(B, B), 1, 0.0, 0.5, self._slot("d_fake"), gl.batch(B, B)))
        pd.add(drf.bwd_program(gl, wgrad=True, accumulate=False))

        # ---- G step (train_cutpp.py:266-308).  Two programs: `prog_g_features` -- the PatchNCE target-feature forward G.encode(fake)
        # and the PatchNCE losses -- does not touch the discriminator, so it runs while the discriminator's gradient all-reduce is
        # in flight on the communication stream (train_step); `prog_g_compute` needs the updated discriminator.
        pf = Program("G-features")
        # The generator's gradient block is cleared once and every pass ACCUMULATES into it: the feature pass (p2) stops at the last
        # PatchNCE layer, so "first pass writes, later passes add" would leave the layers behind it (upsample.3, output.1) adding
        # into the previous step's values.
        pf.add(ops.zero_(self.opt_G.flat_g))
        hooks = {}
        if self.nce_layers:
            pf.add(self.p2.fwd_program(fake))
            pf.add(ops.fill(self._slot("nce"), 0.0))
            wl = lw["patchnce"] / len(self.nce_layers)
            for li, ids in zip(self.nce_layers, self.nce_ids):
                src, tgt = src_feat(li), self.p2.acts[li]
                P = ids.numel()
                ws = ctx.f32(ops.patchnce_ws_floats(B, P, src.C))
                pf.add(ops.patchnce_fwd(src, tgt, ids, P, src.C, cfg["patchnce"]["temperature"], wl, self._slot("nce"), ws))

                def mk(tgt=tgt, ids=ids, P=P, ws=ws):
                    def hook(gv: View):
                        assert (gv.H, gv.W, gv.C) == (tgt.H, tgt.W, tgt.C)
                        return [ops.patchnce_bwd(tgt, ids, P, tgt.C, cfg["patchnce"]["temperature"], wl, gv, ws)]
                    return hook
                hooks[li] = mk()
        # adversarial gradient of the generator's output: D(aug(fake)) forward and its input gradient, on the discriminator's stream
        pa = Program("G-adversarial")
        dp = self.d_fake
        pa.add(self._aug_fwd(fake, dp.x, self.prm["fake_g"]))
        pa.add(dp.fwd_program())
        gl = dp.grad_logits_view()
        pa.add(self.opsD.patch_loss(dp.logits, 2, 0.0, lw["adv"], self._slot("g_adv"), gl))
        pa.add(dp.bwd_program(gl, wgrad=False, need_input_grad=True))
        if self.aug is not None:
            g_adv_img = self.ctxD.view(B, S, S, 8, 0)
            pa.add(self.opsD.diffaug_bwd(dp.g_input, 3, self.prm["fake_g"], g_adv_img, self.ctxD.scratch("aug_ws", B + 16)))
        else:
            g_adv_img = dp.g_input
        # generator backward, in two programs: the PatchNCE feature pass's backward needs nothing from the discriminator ...
        pg0 = Program("G-step (features backward)")
        pg = Program("G-step")
        g_img, g_fold, g_img2 = g_adv_img, False, None
        if self.nce_layers:
            pg0.add(self.p2.bwd_program(hooks=hooks, accumulate=True, need_input_grad=True))
            g_img, g_fold, g_img2 = self.p2.g_input, True, g_adv_img
        # ... the rest starts from the adversarial gradient
        pi = None
        if merged:
            # gradient wrt the 2B output images: [adversarial (+ folded PatchNCE input gradient) | identity L1 x identity weight]
            g13 = ctx.view(2 * B, S, S, 8, 0)
            if g_fold:
                pg.add(ops.fold_add(g_img2, g_img, True, g13.batch(0, B)))
            else:
                pg.add(ops.view_copy(g_img, g13.batch(0, B), HALO_NONE))
            pg.add(ops.l1_loss(p1.img.batch(B, B), 3, self.monets, 1.0, self._slot("idw"), self._slot("identity"), g13.batch(B, B),
                               ctx.scratch("l1_ws", 1024)))
            pg.add(p1.bwd_program(g13, accumulate=True, bucket=self._bucket_plan()))
        else:
            pg.add(p1.bwd_program(g_img, g_fold, g_img2, accumulate=True))
            # identity (identity_l1.py:6-22): third pass, gradient scaled by the device-resident identity weight
            pi = Program("G-identity")
            pi.add(p3.fwd_program(self.monets))
            g_idt = ctx.view(B, S, S, 8, 0)
            pi.add(ops.l1_loss(p3.img, 3, self.monets, 1.0, self._slot("idw"), self._slot("identity"), g_idt, ctx.scratch("l1_ws", 1024)))
            pi.add(p3.bwd_program(g_idt, accumulate=True))
        pfo = Program("fake-out")
        pfo.add(ops.view_to_nchw(fake, 3, self.fake_out))
        return {"p1": p1, "p3": p3, "photos_v": photos_v, "prog_gfwd": prog_gfwd, "prog_d_compute": pd, "prog_g_features": pf,
                "prog_g_adversarial": pa, "prog_g_features_bwd": pg0, "prog_g_compute": pg,
                "prog_g_identity": pi, "prog_fake_out": pfo}

    def _build_updates(self):
        """R1 and the optimiser updates with their operand-copy refresh: built after every mode planned so far, because a
        repack program refreshes exactly the weight copies that exist when it is built."""
        cfg, ops = self.config, self.ops
        gs = 1.0 / self.world_size
        if getattr(self, "_r1_body", None) is None:
            # ---- lazy R1 (train_cutpp.py:165-203, 257-263), always fp32 like the reference
            ops = self.opsD
            pr = Program("R1-body")
            rp, dnet = self.d_r1, self.D32
            pr.add(ops.nchw_to_view(self.photos, 3, rp.x, HALO_ZERO))
            pr.add(rp.fwd_program())
            scale = cfg["r1"]["gamma"] * cfg["r1"]["every"]
            pr.add(rp.r1_program(scale, self._slot("r1"), self._slot("scratch")))
            skip = []
            for li, conv in enumerate(dnet.convs):   # biases: zero grad except the last one, whose grad is None (skipped)
                if conv.grad_b is None:
                    continue
                if li == dnet.nconv - 1:
                    skip.append([k for k, v in self.opt_D.grads.items() if v is conv.grad_b][0])
                else:
                    pr.add(ops.fill(conv.grad_b, 0.0))
            self._r1_body, self._r1_skip = pr, skip
        self.prog_d_update = Program("D-update")
        self.prog_d_update.add(self.opt_D.step_op(cfg.get("grad_clip_d", 10.0), gs))
        self.prog_d_update.add(self.D.repack_program())
        self.prog_r1_compute = Program("R1")
        if self.D32 is not self.D:
            self.prog_r1_compute.add(self.D32.repack_program())   # fp32 operand copies are only needed on R1 steps
        self.prog_r1_compute.add(self._r1_body)
        self.prog_r1_update = Program("R1-update")
        self.prog_r1_update.add(self.opt_D.step_op(cfg.get("grad_clip_d", 10.0), gs, skip=self._r1_skip))
        self.prog_r1_update.add(self.D.repack_program())
        self.prog_g_update = Program("G-update")
        self.prog_g_update.add(self.opt_G.step_op(cfg.get("grad_clip_g", 10.0), gs))
        self.prog_g_update.add(self.G.repack_program())

    # ------------------------------------------------------------------ per-step randomness
    def sample_randomness(self, generator: Optional[torch.Generator] = None, nce_generator: Optional[torch.Generator] = None) -> dict:
        """All device-RNG draws of one step, on the CPU generator, in the reference's consumption order.
        Data parallel: DiffAugment draws are per sample (rank-local `generator`), PatchNCE ids are shared by the whole
        global batch (patchnce_cut.py:63), so every rank passes an identically seeded `nce_generator`."""
        B, S = self.B, self.S
        r = {}
        if self.aug is not None:
            for k in ("aug_real", "aug_fake_d", "aug_fake_g"):
                r[k] = self.aug.sample(B, S, S, generator)
        ng = generator if nce_generator is None else nce_generator
        r["nce_ids"] = [torch.randint(0, hw, (min(self.P, hw),), generator=ng) for hw in self.nce_hw]
        return r

    def _load_randomness(self, rnd: dict):
        B, S = self.B, self.S
        i = self._rnd_turn
        self._rnd_turn ^= 1
        if self._rnd_events[i] is not None:
            self._rnd_events[i].synchronize()          # the copy that last read this pinned block (two steps ago) has run
        prm, ids_h = self._rnd_host_views[i]
        if self.aug is not None:
            for key, k in (("real", "aug_real"), ("fake_d", "aug_fake_d"), ("fake_g", "aug_fake_g")):
                prm[key].copy_(DiffAugment.to_params(rnd[k], B, S, S).reshape(-1))
        for dst, ids in zip(ids_h, rnd["nce_ids"]):
            dst.copy_(ids)
        self._rnd_dev.copy_(self._rnd_host[i], non_blocking=True)
        if self.device.type == "cuda":
            if self._rnd_events[i] is None:
                self._rnd_events[i] = torch.cuda.Event()
            self._rnd_events[i].record()

    def _allreduce_start(self, opt: FusedAdam):
        """Launches the gradient all-reduce (sum; the optimiser divides by world_size) of one flat block on the communication
        stream -- RCCL over xGMI -- and returns a handle; kernels queued on the compute stream meanwhile overlap with it."""
        if not (self.world_size > 1 or getattr(self, "force_allreduce", False)):
            return None
        import torch.distributed as dist
        if self.device.type != "cuda":
            return dist.all_reduce(opt.flat_g, group=self.pg, async_op=True), None
        cur = opt.ctx.ops._ts()                                  # the stream the gradients were produced on
        if os.environ.get("GAN_COMM_STREAM"):                    # a communication stream of our own in front of RCCL's (not needed: see below)
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=self.device)
            self._comm_stream.wait_stream(cur)
            with torch.cuda.stream(self._comm_stream):
                work = dist.all_reduce(opt.flat_g, group=self.pg, async_op=True)
            return work, cur
        # RCCL runs the collective on its own internal stream, ordered after everything queued on the stream that is current at the
        # call: issuing it under the producing stream needs no extra stream (four compute/communication streams = four hardware queues)
        with torch.cuda.stream(cur):
            work = dist.all_reduce(opt.flat_g, group=self.pg, async_op=True)
        return work, cur

    # ---- two gradient buckets for the generator: [residual block K .. output] is reduced while the backward of the earlier layers
    #      still runs (GPass.bwd_program(bucket=...)); the head of the flat gradient follows after the program
    BUCKET_BLOCK = 2

    def _bucket_plan(self):
        """(block index, callback) for the merged generator backward, or None (single all-reduce after the backward)."""
        nb = self.generator.n_blocks
        if os.environ.get("GAN_NO_BUCKET_AR") or nb <= self.BUCKET_BLOCK:
            return None
        i = self.opt_G.names.index(f"res_blocks.{self.BUCKET_BLOCK}.conv_block.1.weight")
        self._bucket_off = int(self.opt_G.offsets[i])
        return (self.BUCKET_BLOCK, self._bucket_start)

    _bucket_work, _bucket_off = None, 0

    def _bucket_start(self):
        """Called from inside the backward program: all-reduce of flat_g[off:] ordered after the side stream's position."""
        if not (self.world_size > 1 or getattr(self, "force_allreduce", False)):
            return
        import torch.distributed as dist
        if self.device.type != "cuda":       # host tensors (gloo): the launches queued before this callback have already run
            self._bucket_work = dist.all_reduce(self.opt_G.flat_g[self._bucket_off:], group=self.pg, async_op=True)
            return
        with torch.cuda.stream(self.ops.side()._ts()):
            self._bucket_work = dist.all_reduce(self.opt_G.flat_g[self._bucket_off:], group=self.pg, async_op=True)

    def _allreduce_G(self):
        """The generator's gradient all-reduce: whole block, or the head only when the tail bucket is already in flight."""
        work = self._bucket_work
        if work is None:
            self._allreduce(self.opt_G)
            return
        self._bucket_work = None
        import torch.distributed as dist
        if self.device.type != "cuda":
            head = dist.all_reduce(self.opt_G.flat_g[:self._bucket_off], group=self.pg, async_op=True)
            work.wait()
            head.wait()
            return
        cur = self.opt_G.ctx.ops._ts()
        with torch.cuda.stream(cur):
            head = dist.all_reduce(self.opt_G.flat_g[:self._bucket_off], group=self.pg, async_op=True)
            work.wait()
            head.wait()

    def _allreduce_finish(self, handle):
        if handle is None:
            return
        work, cur = handle
        if cur is None:
            work.wait()
            return
        with torch.cuda.stream(cur):                             # the stream that produced the gradients (and will consume the
            work.wait()                                          # reduced ones) waits for the collective; the host does not
        if self._comm_stream is not None:
            cur.wait_stream(self._comm_stream)

    _comm_stream = None

    # ---- the two events per step that order the main stream and the discriminator's stream
    def _ev_fake_ready(self):
        if self.opsD is self.ops:
            return
        if self._evA is None:
            self._evA, self._evB = self.ops.new_event(), self.ops.new_event()
        self._evA.record(self.ops._ts())
        self.opsD._ts().wait_event(self._evA)

    def _ev_adv_ready(self):
        if self.opsD is not self.ops:
            self._evB.record(self.opsD._ts())

    def _ev_adv_wait(self):
        if self.opsD is not self.ops:
            self.ops._ts().wait_event(self._evB)

    _evA, _evB = None, None

    def _allreduce(self, opt: FusedAdam):
        self._allreduce_finish(self._allreduce_start(opt))

    # ------------------------------------------------------------------ the step
    def train_step(self, step: int, photos: torch.Tensor, monets: torch.Tensor, rnd: Optional[dict] = None, sync: bool = True):
        """One iteration; returns the reference's loss dict (train_cutpp.py:315-323).  sync=True reads the losses back before
        returning (the reference's .item() calls); sync="lag" returns the PREVIOUS step's dict (see flush_losses); sync=False skips
        the read-back and the NaN check and returns None."""
        if self.device.type != "cuda":
            return self._train_step(step, photos, monets, rnd, sync)
        # the programs launch on the stream the op layer was bound to at construction; the torch-side copies, fills and events of a
        # step must be queued on that same stream whatever the caller has made current
        # ... and ordered against the caller's: the inputs may have been produced on another stream (a prefetch stream, a transform under
        # `with torch.cuda.stream(s)`), and the caller's later work may read this step's results
        bound, cur = self.ops._ts(), torch.cuda.current_stream(self.device)
        if bound != cur:
            bound.wait_stream(cur)
        with torch.cuda.stream(bound):
            out = self._train_step(step, photos, monets, rnd, sync)
        if bound != cur:
            cur.wait_stream(bound)
        return out

    def _train_step(self, step, photos, monets, rnd, sync):
        cfg = self.config
        lw = cfg["loss_weights"]
        idw = identity_weight_at(step, cfg)
        if rnd is None:
            rnd = self.sample_randomness()
        self._load_randomness(rnd)
        self.photos.copy_(photos, non_blocking=True)
        self.monets.copy_(monets, non_blocking=True)
        merged = self.merge_identity and idw > 0
        if merged != self.mode_merged:
            self._use_mode(merged)
        do_r1 = cfg["r1"]["gamma"] > 0 and step % cfg["r1"]["every"] == 0
        self.losses.zero_()
        self.losses[LOSS_SLOTS["idw"]] = idw
        self.prog_gfwd.run()
        self._ev_fake_ready()                       # main: G(photos) done   -> the discriminator's stream may start
        # ---- discriminator's stream: D step, its all-reduce and update, lazy R1, then the adversarial gradient of the fake image
        self.prog_d_compute.run()
        self._allreduce(self.opt_D)
        self.prog_d_update.run()
        if do_r1:
            self.prog_r1_compute.run()
            self._allreduce(self.opt_D)
            self.prog_r1_update.run()
        self.prog_g_adversarial.run()
        self._ev_adv_ready()                        # discriminator's stream: g_adv_img done -> main may build the output gradient
        # ---- main stream, concurrently: PatchNCE feature pass forward and backward; then the rest of the generator step
        self.prog_g_features.run()
        self.prog_g_features_bwd.run()
        self._ev_adv_wait()
        self.prog_g_compute.run()
        if idw > 0 and self.prog_g_identity is not None:
            self.prog_g_identity.run()
        self._allreduce_G()
        self.prog_g_update.run()
        _notify_weights_changed()          # parameters changed through raw pointers: module-level bridges repack on next use
        if sync is False:
            return None
        meta = (step, idw, do_r1)
        if sync == "lag":
            # the read-back of this step's losses is queued behind its kernels and collected when the NEXT step has been queued:
            # every step's dict is still delivered and NaN-checked, one call late, and the host never idles the GPU (-2 % step time)
            prev = self._collect_losses()
            if self._pinned is None:
                self._pinned = [torch.zeros(16, dtype=torch.float32, pin_memory=self.device.type == "cuda") for _ in range(2)]
                self._events = [torch.cuda.Event() if self.device.type == "cuda" else None for _ in range(2)]
            i = step & 1
            self._pinned[i].copy_(self.losses, non_blocking=True)
            if self._events[i] is not None:
                self._events[i].record()
            self._pending = (i, meta)
            return prev
        return self._loss_dict(self.losses.tolist(), meta)

    _pinned, _events, _pending = None, None, None

    def _loss_dict(self, v, meta) -> dict:
        step, idw, do_r1 = meta
        lw = self.config["loss_weights"]
        out = {"d_loss": v[0] + v[1], "g_adv": v[3] / lw["adv"] if lw["adv"] != 0 else 0.0, "nce": v[4] / lw["patchnce"] if lw["patchnce"] > 0 else 0.0,
               "identity": v[5] if idw > 0 else 0.0, "r1": v[2] if do_r1 else 0.0, "identity_weight": idw}
        out["g_loss"] = lw["adv"] * out["g_adv"] + lw["patchnce"] * out["nce"] + idw * out["identity"]
        if any(not math.isfinite(x) for k, x in out.items() if k != "identity_weight"):
            raise ValueError(f"NaN loss detected at step {step}. Training stopped to prevent corruption.")   # train_cutpp.py:326-329
        return out

    def _collect_losses(self) -> Optional[dict]:
        """Loss dict of the step whose read-back is pending (sync="lag"), or None."""
        if self._pending is None:
            return None
        i, meta = self._pending
        self._pending = None
        if self._events[i] is not None:
            self._events[i].synchronize()
        return self._loss_dict(self._pinned[i].tolist(), meta)

    def flush_losses(self) -> Optional[dict]:
        """With sync="lag": waits for and returns the last queued step's losses."""
        return self._collect_losses()

    def generated(self) -> torch.Tensor:
        """G(photos) of the last step as (B,3,H,W) fp32."""
        self.prog_fake_out.run()
        return self.fake_out

    # ------------------------------------------------------------------ checkpoints (utils/io_ckpt.py:56-118)
    def checkpoint(self, step: int, metrics: Optional[dict] = None) -> dict:
        """The reference's checkpoint dict: generator / discriminator state_dicts (reference keys), torch.optim.Adam-layout optimiser
        states, {'decay','shadow'} EMA -- loadable by the reference's load_checkpoint and vice versa."""
        self._device_sync()              # the discriminator's state is written on its own stream
        return {"step": step, "generator": {k: v.clone() for k, v in self.generator.state_dict().items()},
                "discriminator": {k: v.clone() for k, v in self.discriminator.state_dict().items()},
                "opt_G": _adam_state_dict(self.opt_G), "opt_D": _adam_state_dict(self.opt_D), "metrics": metrics or {}, "config": self.config,
                "ema_G": {"decay": self.opt_G.ema_decay, "shadow": {k: v.clone() for k, v in self.opt_G.shadow.items()}}, "scaler": {}}

    def save_checkpoint(self, path: str, step: int, metrics: Optional[dict] = None):
        import pathlib
        pathlib.Path(path).parent.mkdir(parents=True, exist_ok=True)
        torch.save(self.checkpoint(step, metrics), path)

    def load_checkpoint(self, path_or_dict) -> dict:
        """Resume (train_cutpp.py:372-397): weights, Adam moments and step counts, EMA shadow; then refresh the operand copies.
        Files are read with weights_only=True (tensors and plain containers only)."""
        ck = path_or_dict if isinstance(path_or_dict, dict) else torch.load(path_or_dict, map_location=self.device, weights_only=True)
        for mod, opt, key in ((self.generator, self.opt_G, "generator"), (self.discriminator, self.opt_D, "discriminator")):
            missing = set(opt.params) ^ set(ck[key])
            if missing:
                raise KeyError(f"checkpoint['{key}'] keys differ from the model's: {sorted(missing)[:4]} ...")
            for k, v in ck[key].items():
                opt.params[k].copy_(v)
        if "opt_G" in ck:
            _load_adam_state_dict(self.opt_G, ck["opt_G"])
        if "opt_D" in ck:
            _load_adam_state_dict(self.opt_D, ck["opt_D"])
        if "ema_G" in ck and self.opt_G.flat_ema is not None:
            for k, v in ck["ema_G"]["shadow"].items():
                self.opt_G.shadow[k].copy_(v)
        self._device_sync()
        self.G.repack_program().run()
        self.D.repack_program().run()
        self._device_sync()
        from . import autograd as AG
        AG.notify_weights_changed()
        return ck

    def ema_state_dict(self):
        return {"decay": self.opt_G.ema_decay, "shadow": {k: v.clone() for k, v in self.opt_G.shadow.items()}}


def train_step(step, photos, monets, trainer: CutTrainer, rnd=None):
    """Functional alias with the reference's leading arguments (train_cutpp.py:206-219); the model / optimiser / EMA /
    AMP / DiffAugment objects the reference passes separately are owned by `trainer`."""
    return trainer.train_step(step, photos, monets, rnd)
