"""Host logic on CPU: the engine (tap tables, halos, sub-pixel phases, program order, optimiser tables) driven through
the emulator (tests/emulator.py) must reproduce the oracle.  No GPU, no HIP kernel launches."""
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


@pytest.mark.parametrize("geom", [cases.GEOMS[8], cases.GEOMS[10], cases.GEOMS[13]])
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
