"""The C-ABI library loads and exports every symbol include/mi355x_gan.h declares (no compute calls: no GPU here)."""
import os
import re

import gan_variant_research_amd as pkg
from gan_variant_research_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported_and_bound():
    hdr = open(os.path.join(ROOT, "include", "mi355x_gan.h")).read()
    declared = set(re.findall(r"\b(gan_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert lib.gan_version() >= 100


def test_errors_are_reported_not_swallowed():
    lib = _lib.load()
    rc = lib.gan_conv_igemm(None, None)
    assert rc != 0 and b"null" in lib.gan_last_error()


def test_missing_library_fails_loudly(tmp_path):
    import importlib
    import pytest
    saved = _lib._lib
    _lib._lib = None
    try:
        with pytest.raises(_lib.GanError):
            _lib.load(str(tmp_path / "nope.so"))
    finally:
        _lib._lib = saved


def test_argument_validation_messages():
    """Every entry point validates before it launches: bad descriptors come back as a negative code with a message that names the
    problem.  These calls fail in validation, so no GPU is touched."""
    import ctypes as C
    lib = _lib.load()
    d = _lib.GanConvDesc()
    d.dtype, d.B, d.Ho, d.Wo, d.Cin, d.ntaps, d.Nw, d.Nst = _lib.BF16, 1, 4, 4, 24, 9, 16, 8      # Cin not a power of two
    assert lib.gan_conv_igemm(C.byref(d), None) < 0 and b"Cin=24" in lib.gan_last_error()
    d.Cin, d.ntaps = 8, 9                                                                          # 9*8 is not a multiple of 64
    assert lib.gan_conv_igemm(C.byref(d), None) < 0 and b"not a multiple" in lib.gan_last_error()
    assert lib.gan_conv_patch_ok(C.byref(d)) == 0 and lib.gan_conv_patch_ok(None) == 0
    w = _lib.GanWgradDesc()
    assert lib.gan_wgrad_patch_splits(C.byref(w)) == 0
    v = _lib.GanView()                                                                             # null view
    assert lib.gan_in_stats(C.byref(v), C.c_float(1e-5), None, None, None) < 0 and lib.gan_last_error()
    assert lib.gan_patchnce_ws_floats(2, 16, 64) == 3 * 2 * 16 * 64 + 3 * 2 * 16 + 4 + 64
    assert lib.gan_adam_step(None, 0, None, None, 0, C.c_float(1e-3), C.c_float(0.5), C.c_float(0.999), C.c_float(1e-8), C.c_float(0), C.c_float(1),
                             C.c_float(0), None, None, 0, None, None, None) < 0
    assert lib.gan_pack_weight_batch(None, 0, 0, None) < 0 and b"pack_weight_batch" in lib.gan_last_error()
    # operand tensors beyond the kernels' 32-bit byte offsets are refused, not truncated (a batch of 600 images of a 130 x 130 x 256 map)
    big = _lib.GanConvDesc()
    big.dtype, big.B, big.Ho, big.Wo, big.Cin, big.ntaps, big.Nw, big.Nst = _lib.BF16, 600, 128, 128, 256, 9, 256, 256
    big.in_Hp = big.in_Wp = 130
    big.in_sy = big.in_sx = big.out_sy = big.out_sx = 1
    big.out_Hp = big.out_Wp = 128
    big.out_C = 256
    big.in_ = big.w = big.out = big.tapoff = 1 << 20            # aligned, never dereferenced: validation fails first
    assert lib.gan_conv_igemm(C.byref(big), None) < 0 and b"4 GiB" in lib.gan_last_error()
    assert lib.gan_conv_patch_ok(C.byref(big)) == 0


def test_library_shares_the_hip_runtime_torch_uses():
    """Loading the package before anything imported torch must still end with ONE libamdhip64 in the process: with two
    (/opt/rocm's for this library, the wheel's for torch) launches from the library fail with hipErrorNoDevice."""
    import subprocess
    import sys
    code = ("import gan_variant_research_amd as p; lib = p._lib.load(); import torch; "
            "r = p._lib.hip_runtimes_mapped(); print(len(r), r); assert len(r) == 1, r")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr


def test_planner_predicates_equal_their_emulator_statements():
    """gan_conv_patch_ok / gan_conv_stats_parts-free planning and gan_wgrad_patch_splits are pure host functions: the library's answers
    must equal the statements in tests/emulator.py (which is what the CPU suite plans with) on the shapes the trainers meet -- CUT's
    64x64 residual maps, Basic_GAN's 16x16 ones at batch 256 (splits spanning several whole images: negative answer), 128-wide maps."""
    import torch
    from gan_variant_research_amd import BF16
    from gan_variant_research_amd.convplan import ConvLayer
    from gan_variant_research_amd.runtime import Ctx, HipOps
    from tests.emulator import EmuOps
    emu, hip = EmuOps(), HipOps(torch.device("cpu"))          # no launch is built or run on `hip`: descriptors only
    convs, wgrads = [], []
    conv_op, wgrad_op = emu.conv_igemm, emu.conv_wgrad
    emu.conv_igemm = lambda c: (convs.append(c), conv_op(c))[1]
    emu.conv_wgrad = lambda c: (wgrads.append(c), wgrad_op(c))[1]
    ctx = Ctx(emu, torch.device("cpu"), BF16)
    seen_negative = False
    for B, H, Cc in ((2, 64, 256), (16, 64, 256), (64, 16, 256), (256, 16, 256), (48, 16, 256), (8, 128, 128), (16, 32, 256), (3, 20, 64)):
        w = torch.zeros(Cc, Cc, 3, 3)
        layer = ConvLayer(ctx, w, torch.zeros(Cc), torch.zeros_like(w), torch.zeros(Cc), 3, 1, 1)
        x, y = ctx.view(B, H, H, Cc, 1), ctx.view(B, H, H, Cc, 0)
        dy, dx = ctx.view(B, H, H, Cc, 2), ctx.view(B, H, H, Cc, 1)
        del convs[:], wgrads[:]
        layer.fwd(x, y)
        layer.dgrad(dy, dx, padded_domain=True)
        layer.wgrad(x, dy, False, bias_too=False)
        assert len(convs) == 2 and len(wgrads) == 1
        for c in convs:
            assert bool(hip.conv_patch_ok(c)) == bool(emu.conv_patch_ok(c)), (B, H, Cc, c.Ho)
        a, b = hip.wgrad_patch_splits(wgrads[0]), emu.wgrad_patch_splits(wgrads[0])
        assert a == b, (B, H, Cc, a, b)
        seen_negative = seen_negative or a < 0
    assert seen_negative
