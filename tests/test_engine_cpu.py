"""Host logic on CPU: the engine (tap tables, halos, sub-pixel phases, program order, optimiser tables) driven through
the emulator (tests/emulator.py) must reproduce the oracle.  No GPU, no HIP kernel launches."""
import numpy as np
import pytest
import torch

from gan_variant_research_amd import BF16, F32
from gan_variant_research_amd import cut as C
from gan_variant_research_amd.runtime import Ctx
from oracle import cut_ref
from tests import cases
from tests.emulator import EmuOps


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("geom", cases.GEOMS[:8])
def test_conv_geometry(geom, dtype):
    cases.run_conv_geometry(Ctx(EmuOps(), "cpu", dtype), geom, dtype)


@pytest.mark.parametrize("geom", [cases.GEOMS[8], cases.GEOMS[9], cases.GEOMS[10], cases.GEOMS[11], cases.GEOMS[13]])
def test_conv_geometry_fragment_major(geom):
    """Full-width bf16 layers take the range-patch kernel: fragment-major weight copies, checked through the emulator."""
    ctx = Ctx(EmuOps(), "cpu", BF16)
    seen = []
    orig = ctx.ops.conv_igemm
    ctx.ops.conv_igemm = lambda c: (seen.append(c.w_frag), orig(c))[1]
    cases.run_conv_geometry(ctx, geom, BF16)
    assert any(seen), "no call qualified for the range-patch kernel"


def test_module_state_dict_keys_and_init():
    cut_ref.set_seed(42)
    gp, dp = cut_ref.init_generator(), cut_ref.init_discriminator()
    C.set_seed(42)
    gen, disc = C.build_models(cases.small_config(), "cpu")
    gs, ds = gen.state_dict(), disc.state_dict()
    assert list(gs) == list(gp) and list(ds) == list(dp)
    for k in gp:
        assert torch.equal(gs[k], gp[k]), k
    for k in dp:
        assert torch.equal(ds[k], dp[k]), k


@pytest.mark.parametrize("use_aug", [True, False])
def test_cut_train_step_matches_oracle(use_aug):
    torch.set_num_threads(4)
    cases.run_cut_steps("cpu", EmuOps(), use_aug)


def test_basic_gan_iterations_match_oracle():
    torch.set_num_threads(4)
    cases.run_basic_iterations("cpu", EmuOps())


def test_cut_trainer_checkpoint_round_trip(tmp_path):
    """CutTrainer.save_checkpoint / load_checkpoint (utils/io_ckpt.py:56-118 layout): a resumed trainer continues bit-identically,
    and the file's optimiser state loads into the reference's torch.optim.Adam."""
    import copy
    from gan_variant_research_amd import cut as C
    cfg = cases.small_config()
    cfg["diffaugment"]["enable"] = True
    B, S = 2, 32

    def make():
        C.set_seed(42)
        gen, disc = C.build_models(cfg, "cpu")
        return C.CutTrainer(gen, disc, cfg, B, S, device="cpu", amp=False, ops=EmuOps())
    g = torch.Generator().manual_seed(5)
    photos, monets = torch.rand(B, 3, S, S, generator=g) * 2 - 1, torch.rand(B, 3, S, S, generator=g) * 2 - 1
    a = make()
    rnds = []
    for step in range(3):
        torch.manual_seed(100 + step)
        rnds.append(a.sample_randomness())
    a.train_step(0, photos, monets, rnds[0])
    a.train_step(1, photos, monets, rnds[1])
    path = str(tmp_path / "run" / "ckpt_2.pt")
    a.save_checkpoint(path, 2, {"d_loss": 0.5})
    raw = torch.load(path, map_location="cpu", weights_only=True)
    assert set(raw) == {"step", "generator", "discriminator", "opt_G", "opt_D", "metrics", "config", "ema_G", "scaler"} and raw["step"] == 2
    assert list(raw["generator"])[:2] == ["initial.1.weight", "initial.1.bias"] and "discriminators.0.model.8.bias" in raw["discriminator"]
    assert float(raw["opt_G"]["state"][0]["step"]) == 2.0 and float(raw["opt_D"]["state"][0]["step"]) == 3.0   # D: two steps + R1 at step 0
    # the reference's optimiser accepts the state
    ref_gen = copy.deepcopy(a.generator)
    ref_opt = torch.optim.Adam(ref_gen.parameters(), lr=2e-4, betas=(0.5, 0.999))
    ref_opt.load_state_dict(raw["opt_G"])
    assert torch.equal(ref_opt.state[list(ref_gen.parameters())[0]]["exp_avg"], raw["opt_G"]["state"][0]["exp_avg"])
    # resume in a fresh trainer and continue: identical to the uninterrupted run
    want = a.train_step(2, photos, monets, rnds[2])
    b = make()
    assert b.load_checkpoint(path)["step"] == 2
    got = b.train_step(2, photos, monets, rnds[2])
    assert got == want
    for k, v in a.opt_G.params.items():
        assert torch.equal(v, b.opt_G.params[k]), k
    for k, v in a.ema_state_dict()["shadow"].items():
        assert torch.equal(v, b.ema_state_dict()["shadow"][k]), k


def test_lagged_loss_readback_delivers_every_step():
    """train_step(sync="lag") returns the previous step's dict; the values equal the synchronous ones."""
    cfg = cases.small_config()
    B, S = 2, 32

    def make():
        C.set_seed(42)
        gen, disc = C.build_models(cfg, "cpu")
        return C.CutTrainer(gen, disc, cfg, B, S, device="cpu", amp=False, ops=EmuOps())
    g = torch.Generator().manual_seed(5)
    photos, monets = torch.rand(B, 3, S, S, generator=g) * 2 - 1, torch.rand(B, 3, S, S, generator=g) * 2 - 1
    a, b = make(), make()
    rnds = []
    for step in range(3):
        torch.manual_seed(100 + step)
        rnds.append(a.sample_randomness())
    want = [a.train_step(s, photos, monets, rnds[s]) for s in range(3)]
    got = [b.train_step(s, photos, monets, rnds[s], sync="lag") for s in range(3)]
    assert got[0] is None and got[1] == want[0] and got[2] == want[1]
    assert b.flush_losses() == want[2] and b.flush_losses() is None


def test_merged_identity_pass_equals_separate_passes_and_mode_switch():
    """G(photos) and G(monets) as one 2B pass (identity warm-up) == three separate passes; when the identity weight reaches zero the
    trainer switches to the photos-only programs (train_cutpp.py:224-228, 293-295)."""
    B, S = 2, 32
    g = torch.Generator().manual_seed(5)
    photos, monets = torch.rand(B, 3, S, S, generator=g) * 2 - 1, torch.rand(B, 3, S, S, generator=g) * 2 - 1

    def make(merge):
        cfg = cases.small_config()
        cfg["diffaugment"]["enable"] = True
        cfg["warmup_steps"] = 2                      # identity weight: 0.1 at step 0, 0.05 at step 1, 0 from step 2 on
        cfg["merge_identity_pass"] = merge
        C.set_seed(42)
        gen, disc = C.build_models(cfg, "cpu")
        return C.CutTrainer(gen, disc, cfg, B, S, device="cpu", amp=False, ops=EmuOps())
    a, b = make(True), make(False)
    assert a.mode_merged and not b.mode_merged and a.p1.B == 2 * B and b.p1.B == B
    for step in range(4):
        torch.manual_seed(100 + step)
        rnd = a.sample_randomness()
        la, lb = a.train_step(step, photos, monets, rnd), b.train_step(step, photos, monets, rnd)
        assert a.mode_merged == (step < 2)
        assert la["identity_weight"] == lb["identity_weight"] and (la["identity"] == 0.0) == (step >= 2)
        for k in la:
            # from step 1 on, parameters whose gradient is rounding noise have moved by +-lr differently: same
            # tolerances as the oracle comparison of tests/cases.py
            np.testing.assert_allclose(la[k], lb[k], rtol=2e-5 if step == 0 else 2e-3, atol=1e-3 if (k == "g_adv" and step) else 2e-5,
                                       err_msg=f"step {step} {k}")
    worst = max(float((a.opt_G.params[k] - v).abs().max()) for k, v in b.opt_G.params.items())
    assert worst < 1.7e-3, worst     # four sign-like Adam steps of lr 2e-4: +lr in one run, -lr in the other, on weights whose gradient is rounding noise


def test_fp8_conv_path_host_logic():
    """fp8 operand copies + descriptor plumbing of the bottleneck convolution on the emulator (quantisation, per-tensor / per-image
    scales, fragment-major e4m3 packing, padded-domain input gradient)."""
    from gan_variant_research_amd import BF16
    from gan_variant_research_amd.runtime import Ctx
    cases.run_conv_fp8(Ctx(EmuOps(), "cpu", BF16))


def test_cut_step_with_fp8_block_convolutions_on_emulator():
    """The fp8 mode of the fused trainer (e4m3 operand copies for the residual convolutions' forward and input gradient, everything
    else bf16) against the fp32 oracle: step-0 losses within 8 % and the generated image within 0.3 max / 0.06 rms on its [-1, 1] scale
    (e4m3 keeps 3 mantissa bits and the image has passed 18 such convolutions; the emulator applies the same roundings)."""
    tr, img, ref = cases.run_cut_steps("cpu", EmuOps(), True, amp=True, S=32, B=2, nsteps=1, tol0=8e-2, ptol=4.5e-4, fp8=True)
    assert tr.fp8 and tr.G.fp8
    assert float((img - ref).abs().max()) < 0.3 and float((img - ref).pow(2).mean().sqrt()) < 0.06


def test_cut_step_bf16_backward_chain_on_emulator(monkeypatch):
    """The backward chain of the residual blocks (generator_resnet_attn.py:56,64,71 under autograd) as the bf16 trainer runs it: block
    gradients kept on the padded domain, the skip path added and the InstanceNorm-backward sums left by the input-gradient epilogues
    (ConvLayer.dgrad(chain=...), gan_in_bwd_parts).  Two steps against the fp32 oracle with the bf16 tolerance; and the summed generator
    gradient of step 0 against the same step with the chain switched off (fold_add + two-pass norm backward): the two orders of
    computation may differ by bf16 rounding only (measured 1.4 % of the gradient's norm; a wrong sum or a missing skip path is > 30 %)."""
    tr, img, ref = cases.run_cut_steps("cpu", EmuOps(), True, amp=True, S=32, B=2, nsteps=2, tol0=4e-2, tol1=6e-2, atol1=2e-2, ptol=4.5e-4)
    assert tr.p2.bwd_chain and all(m["p1"].bwd_chain for m in tr._modes.values() if m is not None)
    assert float((img - ref).abs().max()) < 5e-2
    tr1, _, _ = cases.run_cut_steps("cpu", EmuOps(), True, amp=True, S=32, B=2, nsteps=1, tol0=4e-2, ptol=4.5e-4)
    monkeypatch.setenv("GAN_NO_BWD_CHAIN", "1")
    tr2, _, _ = cases.run_cut_steps("cpu", EmuOps(), True, amp=True, S=32, B=2, nsteps=1, tol0=4e-2, ptol=4.5e-4)
    assert tr1.p2.bwd_chain and not tr2.p2.bwd_chain
    err = float((tr1.opt_G.flat_g - tr2.opt_G.flat_g).norm() / tr2.opt_G.flat_g.norm())
    assert err < 3e-2, err


def test_lambda_rule_and_scheduler_match_the_reference(golden):
    """Basic_GAN/src/train.py:27-31 and torch's LambdaLR on it (reference-generated basic_sched.npz): lambda_rule value for value, and the
    learning-rate sequence CycleGANTrainer.scheduler_step() hands its three fused optimisers."""
    from gan_variant_research_amd import basic as BG
    g = golden("basic_sched.npz")
    for key in [k for k in g if k.startswith("lambda.")]:
        _, start, total = key.split(".")
        want = g[key]
        got = [BG.lambda_rule(e, int(start), int(total)) for e in range(len(want))]
        assert got == [float(w) for w in want], key
    cfg = cases.basic_config()
    cfg["training"].update({"epochs": 6})
    cfg["optim"]["lr_decay_after"] = 3
    torch.manual_seed(0)
    mods = BG.build_models(cfg, "cpu")
    tr = BG.CycleGANTrainer(*mods, cfg, 1, 16, device="cpu", amp=False, ops=EmuOps())
    seq = [tr.opt_G.lr]
    for _ in range(7):
        tr.scheduler_step()
        seq.append(tr.opt_G.lr)
        assert tr.opt_DA.lr == tr.opt_G.lr == tr.opt_DB.lr and float(tr.opt_G.lr_dev) == np.float32(tr.opt_G.lr)
    np.testing.assert_allclose(seq, g["lr_seq.3.6"], rtol=1e-15, atol=0)
    assert tr.opt_G.base_lr == float(g["initial_lr"])


def test_basic_gan_checkpoint_round_trip(tmp_path):
    cases.run_basic_checkpoint_case("cpu", EmuOps, tmp_path)


def test_fused_adam_gradscaler_semantics():
    """gan_adam_step's GradScaler knobs on the emulator (the HIP twin runs in test_gpu_parity): gradients are unscaled by the device-side
    1/scale, an overflow skips the update and the step counters and raises found_inf, gan_scaler_update backs the scale off / grows it."""
    ops = EmuOps()
    p, g, m, v = torch.ones(8), torch.full((8,), 512.0), torch.zeros(8), torch.zeros(8)
    step = torch.zeros(1, dtype=torch.int32)
    table = ops.make_adam_table([{"p": p, "g": g, "m": m, "v": v, "ema": None, "step": step}])
    norm, ws = torch.zeros(4), torch.zeros(32)
    scale, inv, tracker = torch.tensor([1024.0]), torch.tensor([1.0 / 1024.0]), torch.zeros(1, dtype=torch.int32)
    ct, co = torch.zeros(1, dtype=torch.int32), torch.zeros(1, dtype=torch.int64)
    step_op = ops.adam_step(table, 1, ct, co, 1, 1e-3, 0.5, 0.999, 1e-8, 0.0, 1.0, 0.0, norm, ws, inv_scale=inv, skip_nonfinite=True)
    upd = ops.scaler_update(scale, inv, tracker, norm[2:3], 2.0, 0.5, 2)
    step_op(); upd()
    assert abs(float(norm[0]) - 0.5 * 8 ** 0.5) < 1e-6 and float(norm[2]) == 0 and int(step) == 1 and float(scale) == 1024.0 and int(tracker) == 1
    assert torch.allclose(p, torch.full((8,), 1.0 - 1e-3), atol=1e-6)
    g[3] = float("inf")
    before = p.clone()
    step_op(); upd()
    assert float(norm[2]) == 1 and int(step) == 1 and torch.equal(p, before) and float(scale) == 512.0 and int(tracker) == 0
    g.fill_(256.0)
    step_op(); upd(); step_op(); upd()
    assert int(step) == 3 and float(scale) == 1024.0 and abs(float(inv) - 1.0 / 1024.0) < 1e-12
