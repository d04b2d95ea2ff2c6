"""The module-granular autograd bridge (gan_variant_research_amd/autograd.py) against the oracle under torch autograd.

CPU: the bridge's host logic runs on the emulator (tests/emulator.py stands in for libmi355x_gan.so); the same cases run on
the HIP kernels in tests/test_gpu_parity.py::test_autograd_bridge_hip."""
import numpy as np
import pytest
import torch

import gan_variant_research_amd as pkg
from gan_variant_research_amd import autograd as AG, basic as BG, cut as C
from oracle import basic_ref, cut_ref
from tests.emulator import EmuOps


def _close(a, b, tol, what):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    scale = max(float(b.abs().max()), 1e-6)
    err = float((a - b).abs().max()) / scale
    assert err < tol, f"{what}: max error {err:.3e} of scale {scale:.3e}"


def bridge_cases(device, tol):
    """Every behaviour of the bridge on one device; `tol` = max error relative to the tensor's max magnitude."""
    torch.manual_seed(0)
    G = C.ResNetGenerator(3, 3, ngf=8, n_blocks=2).to(device)
    D = C.MultiscaleDiscriminator(3, ndf=8, n_layers=3, num_scales=1, use_spectral_norm=False).to(device)
    gen = torch.Generator().manual_seed(1)
    x = (torch.rand(2, 3, 32, 32, generator=gen) * 2 - 1)
    wy = torch.randn(2, 3, 32, 32, generator=gen)

    def oracle_params(m):
        return {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}

    # ---- G: output, input gradient, parameter gradients
    xg = x.clone().to(device).requires_grad_(True)
    y = G(xg)
    (y * wy.to(device)).sum().backward()
    p = oracle_params(G)
    xo = x.clone().requires_grad_(True)
    yo = cut_ref.generator_forward(p, xo, n_blocks=2)
    (yo * wy).sum().backward()
    _close(y, yo, tol, "G(x)")
    _close(xg.grad, xo.grad, tol * 5, "dL/dx through G")
    for k, prm in G.named_parameters():
        if not k.endswith(".bias") or k.startswith("output"):     # biases in front of a non-affine norm: exact zero vs rounding noise
            _close(prm.grad, p[k].grad, tol * 5, f"G grad {k}")
    # ---- features: truncated pass, gradients only for the layers it ran
    G.zero_grad()
    ids = [0, 2, 4, 16]                                         # 16 does not exist: ignored like the reference does
    feats = G.get_feature_layers(x.to(device), ids)
    fo = cut_ref.generator_features(p, x, ids, n_blocks=2)
    assert len(feats) == len(fo) == 3
    ws = [torch.randn(f.shape, generator=gen) for f in fo]
    for k in p:
        p[k].grad = None
    sum((f * w.to(device)).sum() for f, w in zip(feats, ws)).backward()
    sum((f * w).sum() for f, w in zip(fo, ws)).backward()
    for f, g in zip(feats, fo):
        _close(f, g, tol, "feature")
    for k, prm in G.named_parameters():
        if p[k].grad is None:
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, f"{k}: layer was not on the feature path"
        elif not k.endswith(".bias"):
            _close(prm.grad, p[k].grad, tol * 5, f"feature-path grad {k}")
    # ---- D alone and G -> D chained, two live generator passes at once (pool), detach
    G.zero_grad(); D.zero_grad()
    pd = oracle_params(D)
    for k in p:
        p[k].grad = None
    x2 = torch.rand(2, 3, 32, 32, generator=gen) * 2 - 1
    fake, idt = G(x.to(device)), G(x2.to(device))
    loss = D(fake)[0].mean() + 0.5 * D(fake.detach())[0].pow(2).mean() + 0.1 * (idt - x2.to(device)).abs().mean()
    loss.backward()
    fo_, io_ = cut_ref.generator_forward(p, x, n_blocks=2), cut_ref.generator_forward(p, x2, n_blocks=2)
    lo = cut_ref.discriminator_forward(pd, fo_)[0].mean() + 0.5 * cut_ref.discriminator_forward(pd, fo_.detach())[0].pow(2).mean() \
        + 0.1 * (io_ - x2).abs().mean()
    lo.backward()
    _close(loss, lo, tol, "chained loss")
    for k, prm in D.named_parameters():
        _close(prm.grad, pd[k].grad, tol * 5, f"D grad {k}")
    for k, prm in G.named_parameters():
        if not k.endswith(".bias") or k.startswith("output"):
            _close(prm.grad, p[k].grad, tol * 10, f"chained G grad {k}")
    assert sum(len(v) for v in G._hip_bridge.pool.values()) >= 2, "two generator passes were alive together"
    # a slot is leased until autograd frees its node's saved tensors (backward without retain_graph) or the node dies
    assert all(s.reclaim() for v in G._hip_bridge.pool.values() for s in v), "slots return to the pool after backward"
    # ---- R1: value and discriminator gradients of 3.5 * r1, against the oracle's create_graph double backward
    D.zero_grad()
    for k in pd:
        pd[k].grad = None
    r1 = AG.r1_regularization(D, x.to(device))
    (3.5 * r1).backward()
    r1o = cut_ref.r1_penalty(pd, x.clone())
    (3.5 * r1o).backward()
    _close(r1, r1o, tol, "r1")
    for k, prm in D.named_parameters():
        if pd[k].grad is None:
            assert prm.grad is None, f"{k}: the reference leaves this gradient None"
        else:
            _close(prm.grad, pd[k].grad, tol * 10, f"R1 grad {k}")
    # ---- an optimiser step is seen by the next forward; no_grad keeps no slot
    opt = torch.optim.SGD(G.parameters(), lr=0.1)
    opt.step()
    with torch.no_grad():
        y2 = G(x.to(device))
    yo2 = cut_ref.generator_forward({k: v.detach().cpu() for k, v in G.state_dict().items()}, x, n_blocks=2)
    _close(y2, yo2, tol, "forward after optimiser step")
    assert float((y2.cpu() - yo.detach()).abs().max()) > 1e-4, "the step must have changed the output"
    assert all(s.reclaim() for v in G._hip_bridge.pool.values() for s in v)
    n_slots = sum(len(v) for v in G._hip_bridge.pool.values())
    for _ in range(3):                                          # steady state: the pool does not grow
        G(x.to(device).requires_grad_(True)).sum().backward()
    assert sum(len(v) for v in G._hip_bridge.pool.values()) == n_slots
    # ---- Basic_GAN modules
    torch.manual_seed(1)
    GB, DB = BG.ResnetGenerator(3, 3, ngf=8, n_blocks=6).to(device), BG.NLayerDiscriminator(3, ndf=8, n_layers=3).to(device)
    pg, pdb = oracle_params(GB), oracle_params(DB)
    lb = DB(GB(x.to(device))).pow(2).mean()
    lb.backward()
    lbo = basic_ref.discriminator_forward(pdb, basic_ref.generator_forward(pg, x, n_blocks=6)).pow(2).mean()
    lbo.backward()
    _close(lb, lbo, tol, "Basic_GAN chained loss")
    for k, prm in GB.named_parameters():
        _close(prm.grad, pg[k].grad, tol * 10, f"Basic G grad {k}")
    for k, prm in DB.named_parameters():
        _close(prm.grad, pdb[k].grad, tol * 10, f"Basic D grad {k}")


def optional_cases(device, tol, tags=("ms3", "sn2")):
    """Architecture switches that are the reference constructors' defaults but off in its YAML, against vectors the
    reference itself produced (tests/golden/cut_optional.npz, oracle/make_golden.py:gen_optional): outputs per scale, the loss of
    a hinge D-step + G-step mix, its input and parameter gradients, R1 with its second-order parameter gradients, eval mode."""
    import os
    from gan_variant_research_amd import losses as L
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cut_optional.npz"))
    T = lambda k: torch.from_numpy(np.asarray(g[k])).to(device)
    if "bsn" in tags:
        # Basic_GAN: spectral norm on the three middle convolutions, InstanceNorm after them (Basic_GAN/src/models.py:67-101)
        D = BG.NLayerDiscriminator(3, 4, 3, spectral=True).to(device)
        sd = {k[len("bsn.sd."):]: T(k) for k in g.files if k.startswith("bsn.sd.")}
        assert list(D.state_dict()) == list(sd), "Basic_GAN state_dict keys / order"
        D.load_state_dict(sd)
        x, y = T("x")[:, :, :64, :64].contiguous(), T("y")[:, :, :64, :64].contiguous()
        D(x)
        xr = x.clone().requires_grad_(True)
        o = D(xr)
        _close(o, T("bsn.out"), tol, "bsn out")
        loss = L.GANLoss("lsgan")(o, True) + 0.5 * L.GANLoss("lsgan")(D(y), False)
        _close(loss, T("bsn.loss"), tol, "bsn loss")
        names = [k for k, _ in D.named_parameters()]
        grads = torch.autograd.grad(loss, [xr] + [p for _, p in D.named_parameters()])
        _close(grads[0], T("bsn.gx"), tol * 10, "bsn dL/dx")
        for k, gr in zip(names, grads[1:]):
            _close(gr, T(f"bsn.gw.{k}"), tol * 10, f"bsn grad {k}")
        for k, v in D.state_dict().items():
            if k.endswith("_u") or k.endswith("_v"):
                _close(v, T(f"bsn.sd_after.{k}"), tol, f"bsn buffer {k}")
        D.eval()
        with torch.no_grad():
            _close(D(x), T("bsn.eval_out"), tol, "bsn eval out")
    for tag in [t for t in tags if t != "bsn"]:
        ns, sn = (3, False) if tag == "ms3" else (2, True)
        D = C.MultiscaleDiscriminator(3, 4, 3, num_scales=ns, use_spectral_norm=sn).to(device)
        sd = {k[len(tag) + 4:]: T(k) for k in g.files if k.startswith(f"{tag}.sd.")}
        assert list(D.state_dict()) == list(sd), "state_dict keys / order"
        D.load_state_dict(sd)
        x, y = T("x"), T("y")
        if sn:
            D(x)                                        # the golden ran one extra training-mode forward first
        xr = x.clone().requires_grad_(True)
        outs = D(xr)
        assert isinstance(outs, list) and len(outs) == ns
        for i, o in enumerate(outs):
            _close(o, T(f"{tag}.out{i}"), tol, f"{tag} out{i}")
        loss = L.discriminator_hinge_loss(outs, D(y)) + 0.25 * L.generator_hinge_loss(outs)
        _close(loss, T(f"{tag}.loss"), tol, f"{tag} loss")
        names = [k for k, _ in D.named_parameters()]
        grads = torch.autograd.grad(loss, [xr] + [p for _, p in D.named_parameters()])
        _close(grads[0], T(f"{tag}.gx"), tol * 5, f"{tag} dL/dx")
        for k, gr in zip(names, grads[1:]):
            _close(gr, T(f"{tag}.gw.{k}"), tol * 5, f"{tag} grad {k}")
        for k, v in D.state_dict().items():
            if k.endswith("_u") or k.endswith("_v"):
                _close(v, T(f"{tag}.sd_after.{k}"), tol, f"{tag} buffer {k} after the forwards")
        D.zero_grad()
        r1 = AG.r1_regularization(D, x.clone())
        _close(r1, T(f"{tag}.r1"), tol * 2, f"{tag} r1")
        r1.backward()
        for k, p_ in D.named_parameters():
            want = T(f"{tag}.r1.gw.{k}")
            if want.numel() == 0:
                assert p_.grad is None, f"{tag}: {k} must have no R1 gradient"
            else:
                _close(p_.grad, want, tol * 10, f"{tag} R1 grad {k}")
        D.eval()
        with torch.no_grad():
            for i, o in enumerate(D(x)):
                _close(o, T(f"{tag}.eval_out{i}"), tol, f"{tag} eval out{i}")
        D.train()


def generator_variant_cases(device, tol, tags=("gz", "grl")):
    """ResNetGenerator with padding_type='zero' (no pad modules: other state_dict indices, zero halos, no gradient folding) and / or
    activation='leaky_relu' in the residual blocks, against vectors the reference produced (cut_optional.npz: gz = zero + leaky_relu,
    grl = reflect + leaky_relu): initialisation from the same seed, output, feature maps, input and parameter gradients."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cut_optional.npz"))
    T = lambda k: torch.from_numpy(np.asarray(g[k])).to(device)
    for tag in tags:
        pad, act = ("zero", "leaky_relu") if tag == "gz" else ("reflect", "leaky_relu")
        C.set_seed(11)
        G = C.ResNetGenerator(3, 3, ngf=8, n_blocks=2, padding_type=pad, activation=act)
        keys = [k[len(tag) + 6:] for k in g.files if k.startswith(f"{tag}.init.")]
        assert list(G.state_dict()) == keys, (list(G.state_dict())[:4], keys[:4])
        for k, v in G.state_dict().items():
            want = g[f"{tag}.init.{k}"]
            np.testing.assert_allclose(np.concatenate([v.reshape(-1)[:4].double().numpy(), [float(v.double().sum())]]), want, rtol=1e-6, atol=1e-7, err_msg=k)
        G = G.to(device)
        x = T("x")[:, :, :32, :32].contiguous()
        xr = x.clone().requires_grad_(True)
        y = G(xr)
        feats = G.get_feature_layers(xr, [0, 2, 4])
        _close(y, T(f"{tag}.y"), tol, f"{tag} G(x)")
        for i, f in enumerate(feats):
            _close(f, T(f"{tag}.feat{i}"), tol, f"{tag} feature {i}")
        loss = (y * T(f"{tag}.w")).sum() + sum((f * f).mean() for f in feats)
        _close(loss, T(f"{tag}.loss"), tol, f"{tag} loss")
        names = [k for k, _ in G.named_parameters()]
        grads = torch.autograd.grad(loss, [xr] + [p for _, p in G.named_parameters()])
        _close(grads[0], T(f"{tag}.gx"), tol * 5, f"{tag} dL/dx")
        for k, gr in zip(names, grads[1:]):
            if not k.endswith(".bias") or k.startswith("output"):      # biases in front of a non-affine norm: rounding noise on both sides
                _close(gr, T(f"{tag}.gw.{k}"), tol * 5, f"{tag} grad {k}")


def generator_switch_cases(device, tol, tags):
    """The remaining constructor switches (replicate padding, norm 'batch' / 'none', no block activation, 1 and 3 down-samplings):
    cut.ResNetGenerator assembles these layer by layer from the op-level modules (ops_library) with the reference's Sequential indices.
    Against vectors the reference produced (cut_variants.npz, oracle/make_golden.py:gen_variants)."""
    import os
    from oracle.make_golden import VARIANTS
    from gan_variant_research_amd import ops_library as L
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cut_variants.npz"))
    T = lambda k: torch.from_numpy(np.asarray(g[k])).to(device)
    for tag in tags:
        kw = VARIANTS[tag]
        L._PLANS.clear()
        C.set_seed(11)
        G = C.ResNetGenerator(3, 3, **{"ngf": 8, "n_blocks": 2, **kw})
        assert G._layerwise
        keys = [k[len(tag) + 6:] for k in g.files if k.startswith(f"{tag}.init.")]
        assert list(G.state_dict()) == keys, (list(G.state_dict())[:6], keys[:6])
        for k, v in G.state_dict().items():
            want = g[f"{tag}.init.{k}"]
            np.testing.assert_allclose(np.concatenate([v.reshape(-1)[:4].double().numpy(), [float(v.double().sum())]]), want, rtol=1e-6, atol=1e-7, err_msg=k)
        G = G.to(device)
        x = T(f"{tag}.x")
        xr = x.clone().requires_grad_(True)
        y = G(xr)
        feats = G.get_feature_layers(xr, [int(i) for i in g[f"{tag}.ids"]])
        _close(y, T(f"{tag}.y"), tol, f"{tag} G(x)")
        for i, f in enumerate(feats):
            _close(f, T(f"{tag}.feat{i}"), tol, f"{tag} feature {i}")
        loss = (y * T(f"{tag}.w")).sum() + sum((f * f).mean() for f in feats)
        _close(loss, T(f"{tag}.loss"), tol, f"{tag} loss")
        names = [k for k, _ in G.named_parameters()]
        grads = torch.autograd.grad(loss, [xr] + [p for _, p in G.named_parameters()])
        _close(grads[0], T(f"{tag}.gx"), tol * 5, f"{tag} dL/dx")
        mods = dict(G.named_modules())

        def rounding_noise(k):      # a conv bias in front of a mean-removing norm: gradient identically zero, both sides hold rounding noise
            if not k.endswith(".bias") or k.startswith("output") or not isinstance(mods[k.rsplit(".", 1)[0]], (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                return False
            norm = kw.get("norm", "instance")
            return norm == "instance" or (norm == "batch" and "conv_block" in k)
        for k, gr in zip(names, grads[1:]):
            if not rounding_noise(k):
                _close(gr, T(f"{tag}.gw.{k}"), tol * 5, f"{tag} grad {k}")
        if kw.get("norm") == "batch":
            for k, v in G.state_dict().items():
                if "running" in k or "num_batches" in k:
                    _close(v.double(), T(f"{tag}.after.{k}").double(), tol, f"{tag} {k} after two training-mode forwards")
            G.eval()
            with torch.no_grad():
                _close(G(x), T(f"{tag}.eval_y"), tol, f"{tag} eval-mode output")


@pytest.mark.parametrize("tag", ["grep", "gbn", "gnn", "gd1", "gd3"])
def test_generator_switches_on_emulator(monkeypatch, tag):
    monkeypatch.setattr(AG, "_OPS_FACTORY", lambda device: EmuOps())
    generator_switch_cases(torch.device("cpu"), 2e-4, tags=(tag,))


@pytest.mark.parametrize("tag", ["gz", "grl"])
def test_generator_variants_on_emulator(monkeypatch, tag):
    monkeypatch.setattr(AG, "_OPS_FACTORY", lambda device: EmuOps())
    generator_variant_cases(torch.device("cpu"), 2e-4, tags=(tag,))


@pytest.mark.parametrize("tag", ["ms3", "sn2", "bsn"])
def test_optional_discriminators_on_emulator(monkeypatch, tag):
    monkeypatch.setattr(AG, "_OPS_FACTORY", lambda device: EmuOps())
    from gan_variant_research_amd import losses as L
    monkeypatch.setattr(L, "_PLANS", {})
    optional_cases(torch.device("cpu"), 2e-4, tags=(tag,))


def test_autograd_bridge_on_emulator(monkeypatch):
    monkeypatch.setattr(AG, "_OPS_FACTORY", lambda device: EmuOps())
    bridge_cases(torch.device("cpu"), 2e-4)


def loss_cases(device, tol):
    """The reference-named loss / augmentation callables (gan_variant_research_amd/losses.py) against the oracle, values and gradients."""
    from gan_variant_research_amd import losses as L
    gen = torch.Generator().manual_seed(3)

    def pair(shape, scale=1.0):
        t = torch.randn(shape, generator=gen) * scale
        return t.clone().to(device).requires_grad_(True), t.clone().requires_grad_(True)

    # hinge (adv_hinge.py) and GANLoss (Basic_GAN/src/losses.py)
    (rg, ro), (fg, fo) = pair((2, 1, 6, 6)), pair((2, 1, 6, 6))
    lg, lo = L.discriminator_hinge_loss([rg], [fg]) + 2 * L.generator_hinge_loss([fg]), cut_ref.d_hinge([ro], [fo]) + 2 * cut_ref.g_hinge([fo])
    lg.backward(); lo.backward()
    _close(lg, lo, tol, "hinge"); _close(rg.grad, ro.grad, tol, "hinge d/dreal"); _close(fg.grad, fo.grad, tol, "hinge d/dfake")
    for mode in ("lsgan", "bce"):
        (pg_, po_) = pair((2, 1, 5, 5))
        a, b = L.GANLoss(mode)(pg_, True) + L.GANLoss(mode)(pg_ * 0.5, False), basic_ref.gan_loss(po_, True, mode) + basic_ref.gan_loss(po_ * 0.5, False, mode)
        a.backward(); b.backward()
        _close(a, b, tol, f"GANLoss {mode}"); _close(pg_.grad, po_.grad, tol, f"GANLoss {mode} grad")
    # L1
    (xg, xo), (tg, to) = pair((2, 3, 8, 8)), pair((2, 3, 8, 8))
    a, b = L.l1_loss(xg, tg), (xo - to).abs().mean()
    a.backward(); b.backward()
    _close(a, b, tol, "l1"); _close(xg.grad, xo.grad, tol, "l1 d/dx"); _close(tg.grad, to.grad, tol, "l1 d/dtarget")
    # PatchNCE with injected ids (and duplicates), one 64-channel layer (MFMA tiling on the GPU) and one 24-channel layer
    for shape, P in (((2, 64, 8, 8), 32), ((2, 24, 6, 6), 20)):
        (sg, so), (tg, to) = pair(shape), pair(shape)
        ids = torch.randint(0, shape[2] * shape[3], (P,), generator=gen)
        ids[1] = ids[0]
        fn = L.PatchNCELoss(0.07, P, [0])
        a = fn([sg], [tg], [ids.to(device)])
        b = cut_ref.patchnce_layer(so.detach(), to, ids, 0.07)
        (1.7 * a).backward(); (1.7 * b).backward()
        _close(a, b, tol, "patchnce"); _close(tg.grad, to.grad, tol * 5, "patchnce d/dtgt")
        assert sg.grad is None
    # DiffAugment with recorded draws
    aug = L.DiffAugment(["color", "translation", "cutout"])
    (xg, xo) = pair((3, 3, 16, 16))
    w = torch.randn(3, 3, 16, 16, generator=gen)
    yg = aug(xg, generator=torch.Generator().manual_seed(11))
    yo = cut_ref.diffaugment(xo, aug.last_draws)
    (yg * w.to(device)).sum().backward(); (yo * w).sum().backward()
    _close(yg, yo, tol, "diffaugment"); _close(xg.grad, xo.grad, tol, "diffaugment grad")


def test_loss_callables_on_emulator(monkeypatch):
    from gan_variant_research_amd import losses as L
    monkeypatch.setattr(AG, "_OPS_FACTORY", lambda device: EmuOps())
    monkeypatch.setattr(L, "_PLANS", {})
    loss_cases(torch.device("cpu"), 2e-5)


def training_cases(device, tol, tmp_path):
    """HipAdam / AMPContext / EMA / checkpoints (gan_variant_research_amd/training.py) against torch.optim.Adam and the reference's formulas."""
    from gan_variant_research_amd import training as T
    torch.manual_seed(5)
    G = C.ResNetGenerator(3, 3, ngf=8, n_blocks=2).to(device)
    D = C.MultiscaleDiscriminator(3, ndf=8, n_layers=3, num_scales=1, use_spectral_norm=False).to(device)
    import copy
    Gt = copy.deepcopy(G)                                   # twin driven by torch.optim.Adam + clip_grad_norm_ + the reference's EMA formula
    opt = T.get_optimizer(G, {"type": "adam", "lr": 2e-4, "betas": [0.5, 0.999]})
    opt_t = torch.optim.Adam(Gt.parameters(), lr=2e-4, betas=(0.5, 0.999))
    ema = T.EMA(G, 0.9, optimizer=opt)
    shadow_t = {k: v.detach().clone() for k, v in Gt.named_parameters()}
    amp = T.AMPContext(False)
    gen = torch.Generator().manual_seed(9)
    for it in range(3):
        grads = [torch.randn(p.shape, generator=gen).to(device) * (10.0 if it == 0 else 0.01) for p in G.parameters()]
        for p, pt, g in zip(G.parameters(), Gt.parameters(), grads):
            p.grad, pt.grad = g.clone(), g.clone()
        if it == 1:                                         # a tensor without gradient is skipped (no step increment), as torch does
            list(G.parameters())[3].grad = None; list(Gt.parameters())[3].grad = None
        amp.step_optimizer(opt, max_grad_norm=10.0)
        ema.update()
        torch.nn.utils.clip_grad_norm_([p for p in Gt.parameters() if p.grad is not None], 10.0)
        opt_t.step()
        for k, v in Gt.named_parameters():
            shadow_t[k] = (1.0 - 0.9) * v.data + 0.9 * shadow_t[k]
    for (k, p), pt in zip(G.named_parameters(), Gt.parameters()):
        _close(p, pt, tol, f"HipAdam param {k}")
        _close(ema.shadow[k], shadow_t[k], tol, f"fused EMA {k}")
    # the module bridge sees parameters written through raw pointers
    x = (torch.rand(1, 3, 16, 16, generator=gen) * 2 - 1).to(device)
    with torch.no_grad():
        y0 = G(x)
    for p in G.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    with torch.no_grad():
        y1 = G(x)
    _close(y1, cut_ref.generator_forward({k: v.detach().cpu() for k, v in G.state_dict().items()}, x.cpu(), n_blocks=2), 2e-4, "forward after HipAdam step")
    assert float((y1 - y0).abs().max()) > 1e-5
    # checkpoint: reference layout, loads into torch.optim.Adam (and back) with identical continuation
    opt_D = T.get_optimizer(D, {"lr": 1e-4})
    path = str(tmp_path / "ckpt" / "step_3.pt")
    T.save_checkpoint(path, 3, G, D, opt, opt_D, ema_G=ema, scaler=amp.scaler, metrics={"d_loss": 1.0}, config={"a": 1})
    raw = torch.load(path, map_location="cpu", weights_only=True)
    assert set(raw) == {"step", "generator", "discriminator", "opt_G", "opt_D", "metrics", "config", "ema_G", "scaler"}
    assert set(raw["ema_G"]) == {"decay", "shadow"} and set(raw["opt_G"]) == {"state", "param_groups"}
    assert set(raw["opt_G"]["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and raw["opt_G"]["state"][0]["step"].dtype == torch.float32
    G2 = C.ResNetGenerator(3, 3, ngf=8, n_blocks=2).to(device)
    D2 = C.MultiscaleDiscriminator(3, ndf=8, n_layers=3, num_scales=1, use_spectral_norm=False).to(device)
    ref_opt = torch.optim.Adam(G2.parameters(), lr=2e-4, betas=(0.5, 0.999))       # what the reference would construct
    ck = T.load_checkpoint(path, G2, D2, opt_G=ref_opt, device=str(device))
    assert ck["step"] == 3
    opt3 = T.get_optimizer(G2, {"lr": 2e-4})
    opt3.load_state_dict(ref_opt.state_dict())                                     # and back from torch's layout
    g = [torch.randn(p.shape, generator=gen).to(device) * 0.01 for p in G.parameters()]
    for p, p2, gg in zip(G.parameters(), G2.parameters(), g):
        p.grad, p2.grad = gg.clone(), gg.clone()
    opt.step(); opt3.step()
    for (k, p), p2 in zip(G.named_parameters(), G2.parameters()):
        _close(p2, p, tol, f"continuation after checkpoint round trip {k}")


def test_training_utilities_on_emulator(monkeypatch, tmp_path):
    monkeypatch.setattr(AG, "_OPS_FACTORY", lambda device: EmuOps())
    training_cases(torch.device("cpu"), 2e-6, tmp_path)


def module_step_cases(device, tol0, tol1):
    """The reference-shaped step driver on the module API (module_step.train_step) vs the oracle's train_step, steps 0 and 1,
    DiffAugment on with injected draws: the same comparison tests/cases.py::run_cut_steps makes for the fused CutTrainer."""
    from gan_variant_research_amd import losses as L, module_step as MS, training as T
    from tests import cases
    cfg = cases.small_config()
    cfg["diffaugment"]["enable"] = True
    B, S = 2, 32
    C.set_seed(42)
    gen, disc = C.build_models(cfg, "cpu")
    gen, disc = gen.to(device), disc.to(device)
    cut_ref.set_seed(42)
    gp, dp = cut_ref.init_generator(), cut_ref.init_discriminator()
    og, od = cut_ref.AdamState(gp), cut_ref.AdamState(dp)
    ema_ref = {k: v.detach().clone() for k, v in gp.items()}
    opt_G, opt_D = T.get_optimizer(gen, cfg["optim"]["G"]), T.get_optimizer(disc, cfg["optim"]["D"])
    ema = T.EMA(gen, cfg["ema"]["decay"], optimizer=opt_G)
    amp = T.AMPContext(False)
    aug = L.DiffAugment(cfg["diffaugment"]["policy"])
    g = torch.Generator().manual_seed(1234)
    photos = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    monets = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    for step in range(2):
        torch.manual_seed(9000 + step)
        rnd = cut_ref.sample_step_randomness(B, S, S, use_aug=True)
        ref = cut_ref.train_step(step, photos, monets, gp, dp, og, od, ema_ref, cfg, rnd)
        rnd_dev = {k: ([t.to(device) for t in v] if k == "nce_ids" else v) for k, v in rnd.items()}
        got = MS.train_step(step, photos.to(device), monets.to(device), gen, disc, opt_G, opt_D, ema, amp, aug, cfg, device, rnd=rnd_dev)
        assert list(got) == list(ref)
        for k in ref:
            atol = max(tol0 * 0.1 if step == 0 else 2e-4, 1e-3 if k == "g_adv" else 0.0)
            np.testing.assert_allclose(got[k], ref[k], rtol=tol0 if step == 0 else tol1, atol=atol, err_msg=f"step{step} {k}")
    for k, v in gen.state_dict().items():          # parameters to 2*lr (Adam's sign-like first updates on zero gradients)
        np.testing.assert_allclose(v.cpu().numpy(), gp[k].detach().numpy(), rtol=0, atol=1e-3, err_msg=k)


def optional_step_cases(device, tol0, tol1):
    """module_step.train_step with the reference constructors' default discriminator family (two scales here, spectral norm) against
    the reference's own train_step (tests/golden/cut_optional.npz, step_sn2.*): steps 0-1 at 64x64, DiffAugment on; also pins
    build_models' initialisation order (u, v draws) and the number of power iterations a step performs."""
    import os
    from gan_variant_research_amd import losses as L, module_step as MS, training as T
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cut_optional.npz"))
    from tests import cases
    cfg = cases.small_config()            # = configs/train_gan_cutpp.yaml's values (oracle/cut_ref.py:default_config), amp off
    cfg["diffaugment"]["enable"] = True
    cfg["model"]["discriminator"]["num_scales"] = 2
    cfg["model"]["discriminator"]["use_spectral_norm"] = True
    B, S = 2, 64
    C.set_seed(42)
    gen, disc = C.build_models(cfg, "cpu")
    gen, disc = gen.to(device), disc.to(device)
    opt_G, opt_D = T.get_optimizer(gen, cfg["optim"]["G"]), T.get_optimizer(disc, cfg["optim"]["D"])
    ema = T.EMA(gen, cfg["ema"]["decay"], optimizer=opt_G)
    amp = T.AMPContext(False)
    aug = L.DiffAugment(cfg["diffaugment"]["policy"])
    gi = torch.Generator().manual_seed(1234)
    photos = torch.rand(B, 3, S, S, generator=gi) * 2 - 1
    monets = torch.rand(B, 3, S, S, generator=gi) * 2 - 1
    for step in range(2):
        torch.manual_seed(9000 + step)
        rnd = cut_ref.sample_step_randomness(B, S, S, use_aug=True)
        rnd_dev = {k: ([t.to(device) for t in v] if k == "nce_ids" else v) for k, v in rnd.items()}
        got = MS.train_step(step, photos.to(device), monets.to(device), gen, disc, opt_G, opt_D, ema, amp, aug, cfg, device, rnd=rnd_dev)
        for k, v in got.items():
            want = float(g[f"step_sn2.step{step}.{k}"])
            atol = max(tol0 * 0.1 if step == 0 else 2e-4, 1e-3 if k == "g_adv" else 0.0)
            np.testing.assert_allclose(v, want, rtol=tol0 if step == 0 else tol1, atol=atol, err_msg=f"step{step} {k}")
    _close(disc.state_dict()["discriminators.1.model.6.weight_u"], torch.from_numpy(g["step_sn2.u_after"]), 5e-3, "weight_u after two steps")


def test_module_step_with_default_discriminator_family_on_emulator(monkeypatch):
    from gan_variant_research_amd import losses as L
    monkeypatch.setattr(AG, "_OPS_FACTORY", lambda device: EmuOps())
    monkeypatch.setattr(L, "_PLANS", {})
    torch.set_num_threads(4)
    optional_step_cases(torch.device("cpu"), 1e-4, 2e-3)


def test_module_step_on_emulator(monkeypatch):
    from gan_variant_research_amd import losses as L
    monkeypatch.setattr(AG, "_OPS_FACTORY", lambda device: EmuOps())
    monkeypatch.setattr(L, "_PLANS", {})
    torch.set_num_threads(4)
    module_step_cases(torch.device("cpu"), 2e-4, 2e-3)


def test_fused_trainer_matches_module_step_over_three_steps(monkeypatch):
    """Two independent drivers of the same step -- the fused CutTrainer programs and the reference-shaped module-API step -- must agree
    beyond the two steps the golden vectors pin (catches state carried wrongly from one step to the next, e.g. gradient blocks)."""
    from gan_variant_research_amd import losses as L, module_step as MS, training as T
    from tests import cases
    monkeypatch.setattr(AG, "_OPS_FACTORY", lambda device: EmuOps())
    monkeypatch.setattr(L, "_PLANS", {})
    torch.set_num_threads(4)
    cfg = cases.small_config()
    cfg["diffaugment"]["enable"] = True
    B, S = 2, 32
    C.set_seed(42)
    gen, disc = C.build_models(cfg, "cpu")
    tr = C.CutTrainer(gen, disc, cfg, B, S, device="cpu", amp=False, ops=EmuOps())
    C.set_seed(42)
    gen2, disc2 = C.build_models(cfg, "cpu")
    opt_G, opt_D = T.get_optimizer(gen2, cfg["optim"]["G"]), T.get_optimizer(disc2, cfg["optim"]["D"])
    ema, amp, aug = T.EMA(gen2, cfg["ema"]["decay"], optimizer=opt_G), T.AMPContext(False), L.DiffAugment(cfg["diffaugment"]["policy"])
    g = torch.Generator().manual_seed(1234)
    photos, monets = torch.rand(B, 3, S, S, generator=g) * 2 - 1, torch.rand(B, 3, S, S, generator=g) * 2 - 1
    for step in range(3):
        torch.manual_seed(9000 + step)
        rnd = cut_ref.sample_step_randomness(B, S, S, use_aug=True)
        torch.manual_seed(9000 + step)
        rnd_tr = tr.sample_randomness()
        a = tr.train_step(step, photos, monets, rnd_tr)
        b = MS.train_step(step, photos, monets, gen2, disc2, opt_G, opt_D, ema, amp, aug, cfg, torch.device("cpu"), rnd=rnd)
        for k in a:
            np.testing.assert_allclose(a[k], b[k], rtol=2e-3, atol=1e-3 if k == "g_adv" else 2e-4, err_msg=f"step {step} {k}")
    worst = max(float((tr.opt_G.params[k] - v.data).abs().max()) for k, v in gen2.named_parameters() if not k.endswith(".bias"))
    assert worst < 1.3e-3, worst    # three sign-like Adam steps of lr 2e-4: a weight whose gradient is rounding noise can move +lr in one run and -lr in the other


def test_inference_path_on_emulator(monkeypatch, tmp_path):
    """inference.py (generate_folder.py): EMA weights are preferred, uint8 conversion, folder round trip through PIL."""
    from gan_variant_research_amd import inference as I
    monkeypatch.setattr(AG, "_OPS_FACTORY", lambda device: EmuOps())
    torch.manual_seed(3)
    G = C.ResNetGenerator(3, 3, ngf=8, n_blocks=2)
    shadow = {k: v.detach() * 0.5 for k, v in G.state_dict().items()}
    ck = tmp_path / "ckpt_final.pt"
    torch.save({"step": 7, "generator": G.state_dict(), "ema_G": {"decay": 0.999, "shadow": shadow}, "config": {}}, ck)
    G2 = I.load_generator(str(ck), device="cpu", ngf=8, n_blocks=2, bf16=False)
    for k, v in G2.state_dict().items():
        assert torch.equal(v, shadow[k]), k                      # EMA weights win over 'generator'
    assert I.pick_state_dict({"G_ema": {"w": torch.zeros(1)}}) == {"w": torch.zeros(1)}
    x = torch.rand(2, 3, 16, 16) * 2 - 1
    u8 = I.stylize(G2, x)
    want = cut_ref.generator_forward({k: v for k, v in shadow.items()}, x, n_blocks=2)
    want = (want.clamp(-1, 1) * 0.5 + 0.5).mul(255).round()
    assert u8.dtype == torch.uint8 and int((u8.float() - want).abs().max()) <= 1
    from PIL import Image
    src = tmp_path / "photos" / "sub"
    src.mkdir(parents=True)
    for i in range(3):
        Image.fromarray((torch.rand(20, 24, 3) * 255).byte().numpy()).save(src / f"p{i}.png")
    n = I.stylize_folder(G2, str(tmp_path / "photos"), str(tmp_path / "out"), device="cpu", img_size=16, batch=2)
    outs = sorted((tmp_path / "out" / "sub").glob("*.jpg"))
    assert n == 3 and len(outs) == 3 and Image.open(outs[0]).size == (16, 16)
