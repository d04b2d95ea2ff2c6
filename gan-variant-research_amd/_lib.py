"""ctypes binding of libmi355x_gan.so (the C ABI declared in include/mi355x_gan.h).

The library is the product: there is no CPU or PyTorch fallback.  `load()` raises if the shared object is
missing or does not export every symbol the header declares.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmi355x_gan.so")

F32, BF16, FP8 = 0, 1, 2     # FP8: OCP e4m3 operand copies of the bottleneck convolutions (BASELINE.json configs[4])
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3
HALO_NONE, HALO_ZERO, HALO_REFLECT, HALO_REPLICATE = 0, 1, 2, 3

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class GanView(C.Structure):
    _fields_ = [("ptr", vp), ("B", i32), ("Hp", i32), ("Wp", i32), ("C", i32), ("y0", i32), ("x0", i32), ("H", i32), ("W", i32),
                ("dtype", i32), ("_pad", i32)]


class GanConvDesc(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("Ho", i32), ("Wo", i32), ("Cin", i32), ("ntaps", i32), ("Nw", i32), ("Nst", i32),
                ("in_", vp), ("in_Hp", i32), ("in_Wp", i32), ("in_y0", i32), ("in_x0", i32), ("in_sy", i32), ("in_sx", i32),
                ("tapoff", vp), ("w", vp), ("bias", vp), ("out", vp),
                ("out_Hp", i32), ("out_Wp", i32), ("out_C", i32), ("out_y0", i32), ("out_x0", i32), ("out_sy", i32), ("out_sx", i32),
                ("act", i32), ("mask", vp), ("mask_Hp", i32), ("mask_Wp", i32), ("mask_y0", i32), ("mask_x0", i32), ("stats", vp), ("max_tapoff", i32), ("w_layout", i32),
                ("win_ty0", i32), ("win_tx0", i32), ("tile_rows", i32), ("tile_cols", i32), ("w_scale", vp), ("in_scale", vp),
                ("stats_mode", i32), ("_pad2", i32)]


class GanWgradDesc(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("Ho", i32), ("Wo", i32), ("Cx", i32), ("ntaps", i32), ("N", i32), ("nsplit", i32),
                ("x", vp), ("x_Hp", i32), ("x_Wp", i32), ("x_y0", i32), ("x_x0", i32), ("x_sy", i32), ("x_sx", i32),
                ("tapoff", vp), ("g", vp),
                ("g_Hp", i32), ("g_Wp", i32), ("g_C", i32), ("g_y0", i32), ("g_x0", i32), ("g_sy", i32), ("g_sx", i32),
                ("part", vp), ("max_tapoff", i32), ("variant", i32)]


class GanAdamTensor(C.Structure):
    _fields_ = [("p", vp), ("g", vp), ("m", vp), ("v", vp), ("ema", vp), ("numel", i64), ("step", vp), ("_pad", i64)]


class GanPackDesc(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("khw", vp), ("dtype", i32), ("Nw", i32), ("ntaps", i32), ("Cin", i32), ("N_real", i32), ("C_real", i32),
                ("swap", i32), ("I2", i32), ("KK", i32), ("layout", i32), ("first_block", i32), ("nblocks", i32), ("scale", vp)]


class GanBiasPartDesc(C.Structure):
    _fields_ = [("part", vp), ("grad", vp), ("nparts", i32), ("C", i32), ("N_real", i32), ("accumulate", i32), ("first_block", i32), ("_pad", i32)]


class GanInputJob(C.Structure):
    _fields_ = [("src", vp), ("src_stride", i32), ("crop_y", i32), ("crop_x", i32), ("crop_h", i32), ("crop_w", i32), ("res_h", i32), ("res_w", i32),
                ("win_y", i32), ("win_x", i32), ("flip", i32), ("order", i32 * 4), ("factor", f32 * 4), ("hue_shift", i32),
                ("hb_off", i32), ("hk_off", i32), ("hksize", i32), ("vb_off", i32), ("vk_off", i32), ("vksize", i32)]


PV, PC, PW = C.POINTER(GanView), C.POINTER(GanConvDesc), C.POINTER(GanWgradDesc)

# name -> (restype, argtypes); must list every function of include/mi355x_gan.h
PROTOTYPES = {
    "gan_last_error": (C.c_char_p, []),
    "gan_version": (C.c_int, []),
    "gan_conv_igemm": (C.c_int, [PC, vp]),
    "gan_conv_win7_ok": (C.c_int, [PC]),
    "gan_conv_wgrad": (C.c_int, [PW, vp]),
    "gan_wgrad_patch_splits": (C.c_int, [PW]),
    "gan_wgrad_win7_splits": (C.c_int, [PW]),
    "gan_wgrad_reduce": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp]),
    "gan_conv_patch_ok": (C.c_int, [PC]),
    "gan_conv_patch_tile_rows": (C.c_int, [PC]),
    "gan_conv_patch_tile_cols": (C.c_int, [PC]),
    "gan_conv_patch_variant": (C.c_int, [PC]),
    "gan_conv_stats_parts": (C.c_int, [PC]),
    "gan_in_stats_from_parts": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, f32, vp, vp]),
    "gan_pack_weight": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]),
    "gan_pack_weight_batch": (C.c_int, [vp, C.c_int, C.c_int, vp]),
    "gan_weight_scale_batch": (C.c_int, [vp, C.c_int, vp]),
    "gan_quantize_fp8": (C.c_int, [PV, PV, vp, vp, vp]),
    "gan_in_bwd_amax": (C.c_int, [PV, vp, C.c_int, PV, C.c_int, PV, vp, vp, vp, vp]),
    "gan_bias_grad": (C.c_int, [PV, C.c_int, vp, C.c_int, vp, vp]),
    "gan_in_stats": (C.c_int, [PV, f32, vp, vp, vp]),
    "gan_in_finalize": (C.c_int, [vp, C.c_int, C.c_int, f32, vp]),
    "gan_in_apply": (C.c_int, [PV, vp, C.c_int, PV, PV, C.c_int, vp]),
    "gan_in_partial_count": (C.c_int, [PV]),
    "gan_in_partial": (C.c_int, [PV, vp, vp]),
    "gan_in_apply_parts": (C.c_int, [PV, vp, C.c_int, f32, vp, C.c_int, PV, PV, C.c_int, vp]),
    "gan_in_apply_parts_fp8": (C.c_int, [PV, vp, C.c_int, f32, vp, C.c_int, PV, PV, PV, C.c_int, vp]),
    "gan_in_bwd": (C.c_int, [PV, vp, C.c_int, PV, C.c_int, PV, PV, vp, vp]),
    "gan_in_bwd_bias": (C.c_int, [PV, vp, C.c_int, PV, C.c_int, PV, PV, vp, vp, C.c_int, C.c_int, vp]),
    "gan_fold_add": (C.c_int, [PV, PV, C.c_int, PV, vp]),
    "gan_pad_fold": (C.c_int, [PV, C.c_int, PV, vp]),
    "gan_act_bwd": (C.c_int, [PV, C.c_int, PV, C.c_int, PV, PV, vp]),
    "gan_nchw_to_view": (C.c_int, [vp, C.c_int, PV, C.c_int, vp]),
    "gan_view_to_nchw": (C.c_int, [PV, C.c_int, vp, vp]),
    "gan_view_copy": (C.c_int, [PV, PV, C.c_int, vp]),
    "gan_avgpool_fwd": (C.c_int, [PV, PV, vp]),
    "gan_in_bwd_bias_parts": (C.c_int, [PV]),
    "gan_in_bwd_bias_deferred": (C.c_int, [PV, vp, C.c_int, PV, C.c_int, PV, PV, vp, vp, vp]),
    "gan_bias_finalize_batch": (C.c_int, [vp, C.c_int, C.c_int, vp]),
    "gan_in_bwd_parts": (C.c_int, [PV, vp, C.c_int, PV, C.c_int, PV, vp, C.c_int, C.c_int, vp, vp]),
    "gan_resize_ksize": (C.c_int, [C.c_int, C.c_int]),
    "gan_resize_coeffs": (C.c_int, [C.c_int, C.c_int, vp, vp, C.c_int]),
    "gan_input_pipeline": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, vp, vp, vp]),
    "gan_spectral_norm_ws_floats": (C.c_int64, [C.c_int, C.c_int]),
    "gan_spectral_norm_fwd": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_float, vp, vp, vp, vp]),
    "gan_spectral_norm_bwd": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]),
    "gan_avgpool_bwd": (C.c_int, [PV, PV, C.c_int, vp]),
    "gan_diffaug_fwd": (C.c_int, [PV, C.c_int, vp, PV, vp, vp]),
    "gan_diffaug_bwd": (C.c_int, [PV, C.c_int, vp, PV, vp, vp]),
    "gan_patch_loss": (C.c_int, [PV, C.c_int, f32, f32, vp, PV, vp]),
    "gan_l1_loss": (C.c_int, [PV, C.c_int, vp, f32, vp, vp, PV, vp, vp]),
    "gan_r1_reduce": (C.c_int, [PV, C.c_int, f32, vp, PV, vp, vp]),
    "gan_patchnce_ws_floats": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "gan_patchnce_fwd": (C.c_int, [PV, PV, vp, C.c_int, C.c_int, f32, f32, vp, vp, vp]),
    "gan_patchnce_bwd": (C.c_int, [PV, vp, C.c_int, C.c_int, f32, f32, PV, vp, vp]),
    "gan_adam_step": (C.c_int, [vp, C.c_int, vp, vp, C.c_int, f32, f32, f32, f32, f32, f32, f32, vp, vp, C.c_int, vp, vp, vp]),
    "gan_scaler_update": (C.c_int, [vp, vp, vp, vp, f32, f32, C.c_int, vp]),
    "gan_fill_f32": (C.c_int, [vp, C.c_int64, f32, vp]),
    "gan_axpy_f32": (C.c_int, [vp, vp, f32, C.c_int64, vp]),
}

_lib = None


class GanError(RuntimeError):
    pass


def hip_runtimes_mapped():
    """Distinct libamdhip64 files mapped into this process (Linux)."""
    try:
        with open("/proc/self/maps") as f:
            return sorted({ln.split()[-1] for ln in f if "libamdhip64" in ln})
    except OSError:
        return []


def load(path: str = LIB_PATH):
    """Loads the shared library once; raises GanError if it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise GanError(f"{path} not found: build it with `make -C gan-variant-research_amd/csrc` "
                       "(or __graft_entry__.build()); there is no fallback path")
    # One HIP runtime per process: PyTorch bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Loaded after torch, this
    # library binds to that copy and shares its device, streams and allocations; loaded first it would pull in /opt/rocm's copy,
    # torch would then map its own beside it, and launches from here would fail with hipErrorNoDevice.
    import torch  # noqa: F401
    lib = C.CDLL(path)
    runtimes = hip_runtimes_mapped()
    if len(runtimes) > 1:
        raise GanError(f"two HIP runtimes mapped in this process ({', '.join(runtimes)}): import torch before loading {path}")
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise GanError(f"{path} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        raise GanError(f"{what}: {load().gan_last_error().decode()} (rc={rc})")
