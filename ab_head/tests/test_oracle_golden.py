"""Pins the CPU oracle (oracle/) against golden vectors produced by the imported reference
(oracle/make_golden.py).  CPU only."""
import numpy as np
import torch

from oracle import basic_ref, cut_ref

T = torch.from_numpy


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def _models():
    cut_ref.set_seed(42)
    return cut_ref.init_generator(), cut_ref.init_discriminator()


def test_init_and_models(golden):
    g = golden("cut_models.npz")
    gp, dp = _models()
    for name, p in (("G", gp), ("D", dp)):
        keys = [k[len(f"init.{name}."):] for k in g if k.startswith(f"init.{name}.")]
        assert sorted(keys) == sorted(p.keys())
        for k in keys:
            close(p[k].reshape(-1)[:16], g[f"init.{name}.{k}"], 0, 0)
            close(p[k].double().sum().float(), g[f"initsum.{name}.{k}"], 1e-6, 1e-6)
    x = T(g["x64"])
    with torch.no_grad():
        close(cut_ref.generator_forward(gp, x), g["G64"], 1e-4, 1e-5)
        feats = cut_ref.generator_features(gp, x, [0, 4, 8, 12, 16])
        assert len(feats) == int(g["nfeats"]) == 4
        for i, f in enumerate(feats):
            assert list(f.shape) == list(g[f"feat{i}.shape"])
            close(f[:, :8, :4, :4], g[f"feat{i}.slice"], 1e-4, 1e-5)
        close(cut_ref.discriminator_forward(dp, x)[0], g["D64"], 1e-4, 1e-5)
        close(cut_ref.discriminator_forward(dp, cut_ref.generator_forward(gp, x))[0], g["D64_of_G"], 1e-4, 1e-5)


def test_losses(golden):
    g = golden("cut_losses.npz")
    for tag in "abc":
        tgt = T(g[f"nce.{tag}.tgt"]).requires_grad_(True)
        loss = cut_ref.patchnce_layer(T(g[f"nce.{tag}.src"]), tgt, T(g[f"nce.{tag}.ids"]))
        close(loss, g[f"nce.{tag}.loss"], 1e-5)
        close(torch.autograd.grad(loss, tgt)[0], g[f"nce.{tag}.gtgt"], 1e-4, 1e-7)
    r, f = T(g["hinge.real"]).requires_grad_(True), T(g["hinge.fake"]).requires_grad_(True)
    dl = cut_ref.d_hinge([r], [f])
    close(dl, g["hinge.d"])
    gr, gf = torch.autograd.grad(dl, [r, f])
    close(gr, g["hinge.d.greal"]); close(gf, g["hinge.d.gfake"])
    close(cut_ref.g_hinge([f]), g["hinge.g"])
    # DiffAugment: the oracle sampler must reproduce the reference's global-RNG draws
    x = T(g["aug.x"]).requires_grad_(True)
    torch.manual_seed(int(g["aug.seed"]))
    d = cut_ref.sample_diffaugment(3, 32, 32)
    y = cut_ref.diffaugment(x, d)
    close(y, g["aug.y"], 1e-5, 1e-6)
    close(torch.autograd.grad((y * T(g["aug.w"])).sum(), x)[0], g["aug.gx"], 1e-5, 1e-6)


def test_optim(golden):
    g = golden("cut_optim.npz")
    p = {k: T(g[f"p0.{k}"]).clone() for k in "abz"}
    ema = {k: v.clone() for k, v in p.items()}
    opt = cut_ref.AdamState(p)
    for s in range(3):
        opt.step(p, {k: T(g[f"g{s}.{k}"]) for k in "abz"}, 10.0)
        cut_ref.ema_update(ema, p, 0.999)
        for k in "abz":
            close(p[k], g[f"p{s+1}.{k}"], 1e-6, 1e-7)
            close(ema[k], g[f"ema{s+1}.{k}"], 1e-6, 1e-7)


def _run_steps(g, tag, S, use_aug):
    torch.set_num_threads(1)
    gp, dp = _models()
    og, od = cut_ref.AdamState(gp), cut_ref.AdamState(dp)
    ema = {k: v.detach().clone() for k, v in gp.items()}
    cfg = cut_ref.default_config()
    photos, monets = T(g[f"{tag}.photos"]), T(g[f"{tag}.monets"])
    for step in range(2):
        torch.manual_seed(9000 + step)
        rnd = cut_ref.sample_step_randomness(2, S, S, use_aug=use_aug)
        out = cut_ref.train_step(step, photos, monets, gp, dp, og, od, ema, cfg, rnd)
        tol = 2e-5 if step == 0 else 1e-3  # step >=1: reference is not bit-reproducible itself
        for k, v in out.items():
            np.testing.assert_allclose(v, float(g[f"{tag}.step{step}.{k}"]), rtol=tol, atol=1e-6, err_msg=f"{tag} step{step} {k}")
        if step == 0:
            with torch.no_grad():
                close(cut_ref.generator_forward(gp, photos), g[f"{tag}.G_after_step0"], 1e-3, 2e-4)
                close(cut_ref.discriminator_forward(dp, photos)[0], g[f"{tag}.Dreal_after_step0"], 1e-3, 2e-4)
    torch.set_num_threads(8)


def test_train_step_noaug(golden):
    _run_steps(golden("cut_steps.npz"), "noaug64", 64, False)


def test_train_step_aug(golden):
    _run_steps(golden("cut_steps.npz"), "aug32", 32, True)
    _run_steps(golden("cut_steps.npz"), "aug64", 64, True)


def test_basic(golden):
    g = golden("basic.npz")
    torch.set_num_threads(1)
    torch.manual_seed(0)
    gab, gba = basic_ref.init_generator(), basic_ref.init_generator()
    da, db = basic_ref.init_discriminator(), basic_ref.init_discriminator()
    for name, p in (("G_ab", gab), ("G_ba", gba), ("D_a", da), ("D_b", db)):
        keys = [k[len(f"init.{name}."):] for k in g if k.startswith(f"init.{name}.")]
        assert sorted(keys) == sorted(p.keys())
        for k in keys:
            close(p[k].reshape(-1)[:16], g[f"init.{name}.{k}"], 0, 0)
    a, b = T(g["real_a"]), T(g["real_b"])
    with torch.no_grad():
        close(basic_ref.generator_forward(gab, a), g["G_ab(a)"], 1e-4, 1e-5)
        close(basic_ref.discriminator_forward(da, a), g["D_a(a)"], 1e-4, 1e-5)
    for mode in ("lsgan", "bce"):
        close(basic_ref.gan_loss(T(g["gl.pred"]), True, mode), g[f"gl.{mode}.real"])
        close(basic_ref.gan_loss(T(g["gl.pred"]), False, mode), g[f"gl.{mode}.fake"])
    both = {**{"ab." + k: v for k, v in gab.items()}, **{"ba." + k: v for k, v in gba.items()}}
    og, oa, ob = cut_ref.AdamState(both), cut_ref.AdamState(da), cut_ref.AdamState(db)
    for it in range(2):
        out = basic_ref.train_iteration(a, b, gab, gba, da, db, og, oa, ob)
        for k, v in out.items():
            np.testing.assert_allclose(v, float(g[f"it{it}.{k}"]), rtol=2e-5 if it == 0 else 1e-3, err_msg=f"it{it} {k}")
    torch.set_num_threads(8)
