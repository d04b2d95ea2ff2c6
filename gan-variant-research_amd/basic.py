"""Host-side mirror of the reference's Basic_GAN (CycleGAN) trainer on the MI355X kernels.

Same names / state_dict keys as Basic_GAN/src/models.py:23-107 (ResnetGenerator, NLayerDiscriminator), Basic_GAN/src/losses.py:5-30
(GANLoss, cycle / identity L1) and the inner loop of Basic_GAN/src/train.py:66-122 (CycleGANTrainer.train_iteration).  The
modules are parameter containers with the reference's module tree; all arithmetic runs in libmi355x_gan.so.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from ._lib import BF16, F32, HALO_ZERO
from .cut import FusedAdam, _adam_state_dict, _load_adam_state_dict
from .nets import DiscriminatorNet, GeneratorNet
from .runtime import Ctx, HipOps, Program, View


def _slots(n):
    return [nn.Identity() for _ in range(n)]


class _ResnetBlockParams(nn.Module):
    def __init__(self, dim):
        super().__init__()
        seq = _slots(7)
        seq[1], seq[5] = nn.Conv2d(dim, dim, 3, bias=False), nn.Conv2d(dim, dim, 3, bias=False)
        self.block = nn.Sequential(*seq)


class ResnetGenerator(nn.Module):
    """Basic_GAN/src/models.py:23-65: bias-free convs except the last; keys net.{1,4,7}, net.{10..}.block.{1,5}, net.{19,22}, net.26."""

    def __init__(self, in_c=3, out_c=3, ngf=64, n_blocks=9):
        super().__init__()
        assert n_blocks in [6, 9], "CycleGAN baseline typically uses 6 or 9 blocks"   # models.py:26
        self.in_c, self.out_c, self.ngf, self.n_blocks = in_c, out_c, ngf, n_blocks
        layers = _slots(4)
        layers[1] = nn.Conv2d(in_c, ngf, 7, bias=False)
        mult = 1
        for _ in range(2):
            layers += [nn.Conv2d(ngf * mult, ngf * mult * 2, 3, stride=2, padding=1, bias=False)] + _slots(2)
            mult *= 2
        layers += [_ResnetBlockParams(ngf * mult) for _ in range(n_blocks)]
        for _ in range(2):
            layers += [nn.ConvTranspose2d(ngf * mult, ngf * mult // 2, 3, stride=2, padding=1, output_padding=1, bias=False)] + _slots(2)
            mult //= 2
        layers += _slots(1) + [nn.Conv2d(ngf, out_c, 7)] + _slots(1)
        self.net = nn.Sequential(*layers)
        self.compute_dtype = F32

    def forward(self, x):
        """Basic_GAN/src/models.py:64-65 on the HIP kernels, differentiable (autograd.py)."""
        from . import autograd as AG
        return AG.generator_forward(self, x, "basic")


class NLayerDiscriminator(nn.Module):
    """Basic_GAN/src/models.py:71-107: keys net.{0,11}.{weight,bias}, net.{2,5,8}.weight -- with `spectral`, the three middle
    convolutions carry weight_orig / weight_u / weight_v instead (models.py:67-69; normalisation by gan_spectral_norm_*, autograd.py)."""

    def __init__(self, in_c=3, ndf=64, n_layers=3, spectral=False):
        super().__init__()
        self.in_c, self.ndf, self.n_layers, self.spectral = in_c, ndf, n_layers, spectral
        seq = [nn.Conv2d(in_c, ndf, 4, stride=2, padding=1)] + _slots(1)
        mult = 1
        for n in range(1, n_layers + 1):
            prev, mult = mult, min(2**n, 8)
            conv = nn.Conv2d(ndf * prev, ndf * mult, 4, stride=2 if n < n_layers else 1, padding=1, bias=False)
            seq += [nn.utils.spectral_norm(conv) if spectral else conv] + _slots(2)
        seq += [nn.Conv2d(ndf * mult, 1, 4, stride=1, padding=1)]
        self.net = nn.Sequential(*seq)
        self.compute_dtype = F32

    def forward(self, x):
        """Basic_GAN/src/models.py:106-107 on the HIP kernels, differentiable (autograd.py)."""
        from . import autograd as AG
        return AG.discriminator_forward(self, x, "basic", "", self.ndf, self.n_layers)


class GANLoss:
    """Basic_GAN/src/losses.py:5-22: 'lsgan' -> MSE against {1,0}; 'bce' -> BCE-with-logits."""

    def __init__(self, gan_mode: str = "lsgan"):
        assert gan_mode in ("lsgan", "bce")
        self.gan_mode = gan_mode

    @property
    def kernel_mode(self) -> int:
        return 3 if self.gan_mode == "lsgan" else 4


def build_models(cfg, device):
    """Basic_GAN/src/train.py:20-25."""
    m = cfg["model"]
    mk_g = lambda: ResnetGenerator(ngf=m["ngf"], n_blocks=m["n_blocks"]).to(device)
    mk_d = lambda: NLayerDiscriminator(ndf=m["ndf"], spectral=m["spectral_norm_d"]).to(device)
    return mk_g(), mk_g(), mk_d(), mk_d()


def lambda_rule(epoch: int, start_decay: int, total_epochs: int) -> float:
    """Basic_GAN/src/train.py:27-31: 1 until `start_decay`, then linear decay to 0 at the final epoch."""
    if epoch < start_decay:
        return 1.0
    return max(0.0, 1.0 - float(epoch - start_decay) / float(max(1, total_epochs - start_decay)))


SLOTS = {"gan_b": 0, "gan_a": 1, "cyc_a": 2, "cyc_b": 3, "idt_a": 4, "idt_b": 5, "da_real": 6, "da_fake": 7, "db_real": 8, "db_fake": 9}


class CycleGANTrainer:
    """One rank's buffers and step programs for the loop body of Basic_GAN/src/train.py:66-122.

    Legal saving (results unchanged): D_A(fake_A) / D_B(fake_B) are forwarded once; the generator step back-propagates through
    them without weight gradients and the discriminator steps reuse the same activations (D's weights do not change in between).
    """

    def __init__(self, G_A2B: ResnetGenerator, G_B2A: ResnetGenerator, D_A: NLayerDiscriminator, D_B: NLayerDiscriminator, cfg: dict,
                 batch_size: int, image_size: int, device="cuda", amp: Optional[bool] = None, ops=None, world_size: int = 1, process_group=None):
        self.cfg, self.B, self.S = cfg, batch_size, image_size
        if getattr(D_A, "spectral", False) or getattr(D_B, "spectral", False):
            raise NotImplementedError("the fused CycleGANTrainer runs Basic_GAN/configs/baseline.yaml (spectral_norm_d: false); "
                                      "spectral-norm discriminators run through the nn.Module API (autograd.py)")
        self.device = torch.device(device)
        amp = cfg["training"].get("amp", True) if amp is None else amp
        self.dtype = BF16 if amp else F32
        self.ops = ops if ops is not None else HipOps(self.device)
        if hasattr(self.ops, "bind"):
            self.ops.bind()
        self.ctx = Ctx(self.ops, self.device, self.dtype)
        self.world_size, self.pg = world_size, process_group
        self.gan = GANLoss(cfg["loss"]["gan"])
        lr_g, lr_d, betas = cfg["optim"]["lr_g"], cfg["optim"]["lr_d"], tuple(cfg["optim"]["betas"])
        f32 = lambda sd: {k: v.detach().to(self.device, torch.float32) for k, v in sd.items()}
        gab, gba = f32(G_A2B.state_dict()), f32(G_B2A.state_dict())
        both = {**{"ab." + k: v for k, v in gab.items()}, **{"ba." + k: v for k, v in gba.items()}}   # optim_G owns both (train.py:45-48)
        self.opt_G = FusedAdam(self.ctx, list(both), [v.shape for v in both.values()], both, lr_g, betas)
        da, db = f32(D_A.state_dict()), f32(D_B.state_dict())
        self.opt_DA = FusedAdam(self.ctx, list(da), [v.shape for v in da.values()], da, lr_d, betas)
        self.opt_DB = FusedAdam(self.ctx, list(db), [v.shape for v in db.values()], db, lr_d, betas)
        for mod, opt, pre in ((G_A2B, self.opt_G, "ab."), (G_B2A, self.opt_G, "ba."), (D_A, self.opt_DA, ""), (D_B, self.opt_DB, "")):
            for k, p in mod.named_parameters():
                p.data = opt.params[pre + k]
        sub = lambda d, pre: {k[len(pre):]: v for k, v in d.items() if k.startswith(pre)}
        nb, ngf = G_A2B.n_blocks, G_A2B.ngf
        self.Gab = GeneratorNet(self.ctx, sub(self.opt_G.params, "ab."), sub(self.opt_G.grads, "ab."), "basic", nb, ngf)
        self.Gba = GeneratorNet(self.ctx, sub(self.opt_G.params, "ba."), sub(self.opt_G.grads, "ba."), "basic", nb, ngf)
        self.DA = DiscriminatorNet(self.ctx, self.opt_DA.params, self.opt_DA.grads, "basic", ndf=D_A.ndf, n_layers=D_A.n_layers)
        self.DB = DiscriminatorNet(self.ctx, self.opt_DB.params, self.opt_DB.grads, "basic", ndf=D_B.ndf, n_layers=D_B.n_layers)
        B, S = self.B, self.S
        self.real_a = torch.zeros(B, 3, S, S, dtype=torch.float32, device=self.device)
        self.real_b = torch.zeros_like(self.real_a)
        self.losses = self.ctx.f32(16)
        self.sched_epoch = 0          # LambdaLR's epoch counter (train.py:54-58): advanced by scheduler_step() at every epoch end
        self._build()
        for net in (self.Gab, self.Gba, self.DA, self.DB):
            net.repack_program().run()

    def _slot(self, name):
        i = SLOTS[name]
        return self.losses[i:i + 1]

    def _build(self):
        ops, ctx, B, S = self.ops, self.ctx, self.B, self.S
        lam_c, lam_i = float(self.cfg["loss"]["lambda_cycle"]), float(self.cfg["loss"]["lambda_identity"])
        mode = self.gan.kernel_mode
        gs = 1.0 / self.world_size
        ws = ctx.scratch("l1_ws", 1024)
        P = {}
        # ---- generator step (train.py:71-96): six generator forwards, two discriminator forwards
        fwd = Program("G-fwd")
        P["ab_a"], P["ba_b"] = self.Gab.new_pass(B, S, S), self.Gba.new_pass(B, S, S)          # fake_B = G_A2B(A), fake_A = G_B2A(B)
        fwd.add(P["ab_a"].fwd_program(self.real_a)); fwd.add(P["ba_b"].fwd_program(self.real_b))
        P["ba_fb"], P["ab_fa"] = self.Gba.new_pass(B, S, S), self.Gab.new_pass(B, S, S)        # rec_A = G_B2A(fake_B), rec_B = G_A2B(fake_A)
        fwd.add(P["ba_fb"].fwd_program(P["ab_a"].img)); fwd.add(P["ab_fa"].fwd_program(P["ba_b"].img))
        P["ab_b"], P["ba_a"] = self.Gab.new_pass(B, S, S), self.Gba.new_pass(B, S, S)          # idt_B = G_A2B(B), idt_A = G_B2A(A)
        fwd.add(P["ab_b"].fwd_program(self.real_b)); fwd.add(P["ba_a"].fwd_program(self.real_a))
        self.db_fake, self.da_fake = self.DB.new_pass(B, S, S), self.DA.new_pass(B, S, S)
        self.db_real, self.da_real = self.DB.new_pass(B, S, S), self.DA.new_pass(B, S, S)
        fwd.add(ops.view_copy(P["ab_a"].img, self.db_fake.x, HALO_ZERO)); fwd.add(self.db_fake.fwd_program())
        fwd.add(ops.view_copy(P["ba_b"].img, self.da_fake.x, HALO_ZERO)); fwd.add(self.da_fake.fwd_program())
        self.P = P

        bwd = Program("G-bwd")
        gv = lambda: ctx.view(B, S, S, 8, 0)
        # adversarial terms: generators want D(fake) = real
        gl_b, gl_a = self.db_fake.grad_logits_view(), self.da_fake.grad_logits_view()
        bwd.add(ops.patch_loss(self.db_fake.logits, mode, 1.0, 1.0, self._slot("gan_b"), gl_b))
        bwd.add(self.db_fake.bwd_program(gl_b, wgrad=False, need_input_grad=True))
        g_adv_b = self.db_fake.g_input      # scratch of D_B: untouched until D_B's own step
        bwd.add(ops.patch_loss(self.da_fake.logits, mode, 1.0, 1.0, self._slot("gan_a"), gl_a))
        bwd.add(self.da_fake.bwd_program(gl_a, wgrad=False, need_input_grad=True))
        g_adv_a = self.da_fake.g_input
        # cycle terms -> gradient wrt the fakes through the second generator
        g_rec_a, g_rec_b = gv(), gv()
        bwd.add(ops.l1_loss(P["ba_fb"].img, 3, self.real_a, lam_c, None, self._slot("cyc_a"), g_rec_a, ws))
        bwd.add(P["ba_fb"].bwd_program(g_rec_a, accumulate=False, need_input_grad=True))
        g_fb_cyc = P["ba_fb"].g_input       # padded-domain gradient (halo 3) in G_B2A's scratch: no later G_B2A pass asks for an input gradient
        bwd.add(ops.l1_loss(P["ab_fa"].img, 3, self.real_b, lam_c, None, self._slot("cyc_b"), g_rec_b, ws))
        bwd.add(P["ab_fa"].bwd_program(g_rec_b, accumulate=False, need_input_grad=True))
        g_fa_cyc = P["ab_fa"].g_input
        # first generators: adversarial + folded cycle gradient
        bwd.add(P["ab_a"].bwd_program(g_fb_cyc, True, g_adv_b, accumulate=True))
        bwd.add(P["ba_b"].bwd_program(g_fa_cyc, True, g_adv_a, accumulate=True))
        # identity terms
        g_idt_b, g_idt_a = gv(), gv()
        bwd.add(ops.l1_loss(P["ab_b"].img, 3, self.real_b, lam_i, None, self._slot("idt_b"), g_idt_b, ws))
        bwd.add(P["ab_b"].bwd_program(g_idt_b, accumulate=True))
        bwd.add(ops.l1_loss(P["ba_a"].img, 3, self.real_a, lam_i, None, self._slot("idt_a"), g_idt_a, ws))
        bwd.add(P["ba_a"].bwd_program(g_idt_a, accumulate=True))
        self.prog_g_fwd, self.prog_g_bwd = fwd, bwd

        # ---- discriminator steps (train.py:99-114): real pass + the fake pass forwarded above
        def d_step(net, real_src, p_real, p_fake, s_real, s_fake):
            prog = Program("D-step")
            prog.add(ops.nchw_to_view(real_src, 3, p_real.x, HALO_ZERO))
            prog.add(p_real.fwd_program())
            gl = p_real.grad_logits_view()
            prog.add(ops.patch_loss(p_real.logits, mode, 1.0, 0.5, self._slot(s_real), gl))
            prog.add(p_real.bwd_program(gl, wgrad=True, accumulate=False))
            gl2 = p_fake.grad_logits_view()
            prog.add(ops.patch_loss(p_fake.logits, mode, 0.0, 0.5, self._slot(s_fake), gl2))
            prog.add(p_fake.bwd_program(gl2, wgrad=True, accumulate=True))
            return prog
        self.prog_da = d_step(self.DA, self.real_a, self.da_real, self.da_fake, "da_real", "da_fake")
        self.prog_db = d_step(self.DB, self.real_b, self.db_real, self.db_fake, "db_real", "db_fake")
        # ---- updates (no clipping, no EMA in this trainer); built last: all operand copies are planned by now
        self.upd_g = Program("G-update"); self.upd_g.add(self.opt_G.step_op(None, gs)); self.upd_g.add(self.Gab.repack_program()); self.upd_g.add(self.Gba.repack_program())
        self.upd_da = Program("DA-update"); self.upd_da.add(self.opt_DA.step_op(None, gs)); self.upd_da.add(self.DA.repack_program())
        self.upd_db = Program("DB-update"); self.upd_db.add(self.opt_DB.step_op(None, gs)); self.upd_db.add(self.DB.repack_program())

    # ---- epoch end (Basic_GAN/src/train.py:124-137): LambdaLR x 3, then the checkpoint dict
    def scheduler_step(self) -> float:
        """sched_G.step(); sched_D_A.step(); sched_D_B.step() (train.py:54-58,125): the scheduler's epoch counter advances and every
        optimiser's learning rate becomes base_lr * lambda_rule(counter) -- one device float each, the step programs stay as built."""
        tr, opt = self.cfg["training"], self.cfg["optim"]
        self.sched_epoch += 1
        lam = lambda_rule(self.sched_epoch, int(opt.get("lr_decay_after", tr.get("epochs", 1))), int(tr.get("epochs", 1)))
        for o in (self.opt_G, self.opt_DA, self.opt_DB):
            o.set_lr(o.base_lr * lam)
        return lam

    def _module_state(self, opt, prefix=""):
        return {k[len(prefix):]: v.detach().clone() for k, v in opt.params.items() if k.startswith(prefix)}

    def save_checkpoint(self, path: str, epoch: int):
        """The dict of train.py:127-137, key for key: epoch, the four state_dicts under the reference's module keys, and the three
        torch.optim.Adam state_dicts (optim_G numbers list(G_A2B.parameters()) + list(G_B2A.parameters()), train.py:45-48; the param
        groups carry the `initial_lr` LambdaLR put there and the decayed `lr`).  torch.optim.Adam.load_state_dict accepts them."""
        ck = {"epoch": int(epoch),
              "G_A2B": self._module_state(self.opt_G, "ab."), "G_B2A": self._module_state(self.opt_G, "ba."),
              "D_A": self._module_state(self.opt_DA), "D_B": self._module_state(self.opt_DB),
              "optim_G": _adam_state_dict(self.opt_G, True), "optim_D_A": _adam_state_dict(self.opt_DA, True),
              "optim_D_B": _adam_state_dict(self.opt_DB, True)}
        torch.save(ck, path)
        return ck

    def load_checkpoint(self, path: str) -> int:
        """Resumes from a checkpoint in the reference's layout (train.py:127-137; the reference has no resume code of its own: the epoch
        counter of the three schedulers is taken from `epoch`, as LambdaLR(last_epoch=epoch) would).  Returns the epoch."""
        ck = torch.load(path, map_location="cpu", weights_only=True)
        for key, opt, pre in (("G_A2B", self.opt_G, "ab."), ("G_B2A", self.opt_G, "ba."), ("D_A", self.opt_DA, ""), ("D_B", self.opt_DB, "")):
            for k, v in ck[key].items():
                opt.params[pre + k].copy_(v)
        for key, opt in (("optim_G", self.opt_G), ("optim_D_A", self.opt_DA), ("optim_D_B", self.opt_DB)):
            _load_adam_state_dict(opt, ck[key])
        self.sched_epoch = int(ck["epoch"])
        for net in (self.Gab, self.Gba, self.DA, self.DB):
            net.repack_program().run()
        from .cut import _notify_weights_changed
        _notify_weights_changed()
        return self.sched_epoch

    def _allreduce(self, opt):
        if self.world_size > 1:
            import torch.distributed as dist
            dist.all_reduce(opt.flat_g, group=self.pg)

    def train_iteration(self, real_a: torch.Tensor, real_b: torch.Tensor, sync: bool = True) -> Optional[Dict[str, float]]:
        """One loop iteration; returns {'loss_G','loss_D_A','loss_D_B'} (the values train.py:118-122 shows in its progress bar)."""
        if self.device.type != "cuda":
            return self._train_iteration(real_a, real_b, sync)
        # the programs launch on the stream the op layer was bound to at construction: the copies of this call go there too, ordered
        # after the caller's current stream (which may have produced the inputs) and before its later work
        bound, cur = self.ops._ts(), torch.cuda.current_stream(self.device)
        if bound != cur:
            bound.wait_stream(cur)
        with torch.cuda.stream(bound):
            out = self._train_iteration(real_a, real_b, sync)
        if bound != cur:
            cur.wait_stream(bound)
        return out

    def _train_iteration(self, real_a, real_b, sync):
        self.real_a.copy_(real_a, non_blocking=True)
        self.real_b.copy_(real_b, non_blocking=True)
        self.prog_g_fwd.run()
        self.prog_g_bwd.run()
        self._allreduce(self.opt_G)
        self.upd_g.run()
        self.prog_da.run(); self._allreduce(self.opt_DA); self.upd_da.run()
        self.prog_db.run(); self._allreduce(self.opt_DB); self.upd_db.run()
        from .cut import _notify_weights_changed
        _notify_weights_changed()
        if not sync:
            return None
        v = self.losses.tolist()
        return {"loss_G": sum(v[0:6]), "loss_D_A": v[6] + v[7], "loss_D_B": v[8] + v[9]}
