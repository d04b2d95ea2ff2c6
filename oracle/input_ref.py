"""CPU restatement of the reference's image transforms (TEST INFRASTRUCTURE ONLY: never imported by the product path).

What the reference runs per image (GAN_Variant1/dataio/transforms.py:10-49; Basic_GAN/src/data.py:8-26) is a chain of PIL
operations driven by torchvision glue:

    RandomCropResize   img.crop(box) -> img.resize((S, S), BICUBIC)          transforms.py:17-27  (np.random draws)
    RandomHorizontalFlip                                                      transforms.py:34
    ColorJitter(0.05, 0.05, 0.05, 0.02)   PIL ImageEnhance blends / HSV shift, in a random order     transforms.py:35
    ToTensor, Normalize(0.5, 0.5)                                             transforms.py:36-37

The pixel arithmetic lives in Pillow (pinned here: 12.2.0, importable in this image) and is restated below in numpy, integer for
integer and float32/float64 exactly where Pillow's C code uses them.  Every function is pinned BIT-EXACTLY against Pillow itself in
tests/test_input_pipeline.py (resize on random sizes; blends on random factors; RGB<->HSV exhaustively over all 2^24 triples).
torchvision is absent from this image, so its glue -- which Pillow call each transform makes and in which order the random numbers
are drawn -- is restated from torchvision 0.20.1's published source (torchvision/transforms/{transforms,_functional_pil}.py) and is
"parity unpinned" at that boundary.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2          # Pillow src/libImaging/Resample.c


# ------------------------------------------------------------------------------------------------ bicubic resize
def _bicubic(x: float, a: float = -0.5) -> float:
    """Resample.c bicubic_filter (Keys, a = -0.5), support 2."""
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resize_coeffs(in_size: int, in0: float, in1: float, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc: per output index the first source index and tap count
    (bounds[o] = (xmin, n)) and the taps as PRECISION_BITS fixed point."""
    scale = filterscale = (in1 - in0) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            p = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + p * (1 << PRECISION_BITS)) if p < 0 else int(0.5 + p * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis(img: np.ndarray, bounds: np.ndarray, kk: np.ndarray, axis: int) -> np.ndarray:
    a = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.zeros((len(bounds),) + a.shape[1:], np.int64)
    for o, (xmin, n) in enumerate(bounds):
        acc = np.full(a.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for k in range(int(n)):
            acc += a[xmin + k] * int(kk[o, k])
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out.astype(np.uint8), 0, axis)


def resize_bicubic(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """Image.resize((out_w, out_h), BICUBIC) on an HWC uint8 image: horizontal pass, then vertical, uint8 in between; a pass
    whose size does not change is skipped (Resample.c ImagingResample)."""
    H, W = img.shape[:2]
    x = img
    if out_w != W:
        x = _resample_axis(x, *resize_coeffs(W, 0, W, out_w), axis=1)
    if out_h != H:
        x = _resample_axis(x, *resize_coeffs(H, 0, H, out_h), axis=0)
    return x


# ------------------------------------------------------------------------------------------------ ImageEnhance / blend
def to_gray(img: np.ndarray) -> np.ndarray:
    """convert("L"): ITU-R 601-2 luma in 16.16 fixed point (Convert.c rgb2l)."""
    r, g, b = (img[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend(in1: np.ndarray, in2: np.ndarray, alpha: float) -> np.ndarray:
    """Image.blend (Blend.c): float32 `in1 + alpha * (in2 - in1)`, truncated; clipped first when alpha is outside [0, 1]."""
    a = np.float32(alpha)
    t = in1.astype(np.float32) + a * (in2.astype(np.int32) - in1.astype(np.int32)).astype(np.float32)
    if 0.0 <= alpha <= 1.0:
        return t.astype(np.int32).astype(np.uint8)
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t.astype(np.int32))).astype(np.uint8)


def gray_mean(img: np.ndarray) -> int:
    """ImageEnhance.Contrast: int(ImageStat.Stat(image.convert("L")).mean[0] + 0.5)."""
    g = to_gray(img)
    return int(int(g.sum()) / g.size + 0.5)


def adjust_brightness(img, f):      # ImageEnhance.Brightness: blend(black, image, f)
    return blend(np.zeros_like(img), img, f)


def adjust_contrast(img, f):        # ImageEnhance.Contrast: blend(mean grey, image, f)
    return blend(np.full_like(img, gray_mean(img)), img, f)


def adjust_saturation(img, f):      # ImageEnhance.Color: blend(image.convert("L").convert("RGB"), image, f)
    return blend(np.repeat(to_gray(img)[..., None], 3, axis=2), img, f)


# ------------------------------------------------------------------------------------------------ hue (RGB <-> HSV, Convert.c)
def rgb_to_hsv(x: np.ndarray) -> np.ndarray:
    """Convert.c rgb2hsv_row: float32 variables, double constants (so the sums and the scaling by 255 run in double)."""
    shp = x.shape
    x = x.reshape(-1, 3)
    r, g, b = (x[:, i].astype(np.int32) for i in range(3))
    maxc, minc = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    f32, f64 = np.float32, np.float64
    cr = (maxc - minc).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        s = cr / maxc.astype(f32)
        rc, gc, bc = ((maxc - c).astype(f32) / cr for c in (r, g, b))
        h = np.where(r == maxc, (bc - gc).astype(f32),
                     np.where(g == maxc, (2.0 + rc.astype(f64) - bc.astype(f64)).astype(f32), (4.0 + gc.astype(f64) - rc.astype(f64)).astype(f32)))
        h = np.fmod(h.astype(f64) / 6.0 + 1.0, 1.0).astype(f32)
        uh = np.clip((h.astype(f64) * 255.0).astype(np.int64), 0, 255)
        us = np.clip((s.astype(f64) * 255.0).astype(np.int64), 0, 255)
    grey = minc == maxc
    return np.stack([np.where(grey, 0, uh), np.where(grey, 0, us), maxc], -1).astype(np.uint8).reshape(shp)


def hsv_to_rgb(x: np.ndarray) -> np.ndarray:
    """Convert.c hsv2rgb: sector i = floor(h*6/255), remainder f and fs in float32, p/q/t by C round() of double products."""
    shp = x.shape
    x = x.reshape(-1, 3)
    h, s, v = x[:, 0], x[:, 1], x[:, 2]
    f32, f64 = np.float32, np.float64
    hf = h.astype(f32).astype(f64) * 6.0 / 255.0
    i = np.floor(hf).astype(np.int32)
    f = (hf - i.astype(f32).astype(f64)).astype(f32).astype(f64)
    fs = (s.astype(f32).astype(f64) / 255.0).astype(f32).astype(f64)
    vf = v.astype(f32).astype(f64)
    rnd = lambda a: np.clip(np.where(a >= 0, np.floor(a + 0.5), np.ceil(a - 0.5)).astype(np.int64), 0, 255)   # C round(): half away from zero
    p, q, t = rnd(vf * (1.0 - fs)), rnd(vf * (1.0 - fs * f)), rnd(vf * (1.0 - fs * (1.0 - f)))
    vi = v.astype(np.int64)
    m = i % 6
    out = np.stack([np.choose(m, [vi, q, p, p, t, vi]), np.choose(m, [t, vi, vi, q, p, p]), np.choose(m, [p, p, t, vi, vi, q])], -1)
    out = np.where((s == 0)[:, None], np.stack([vi, vi, vi], -1), out)
    return out.astype(np.uint8).reshape(shp)


def hue_shift(hue_factor: float) -> int:
    """torchvision _functional_pil.adjust_hue: the H channel (uint8) gets `uint8(hue_factor * 255)` added with wrap-around."""
    return int(hue_factor * 255) % 256


def adjust_hue(img, hue_factor):
    hsv = rgb_to_hsv(img)
    hsv[..., 0] = (hsv[..., 0].astype(np.int32) + hue_shift(hue_factor)).astype(np.uint8)
    return hsv_to_rgb(hsv)


# ------------------------------------------------------------------------------------------------ the chains
JITTER = (adjust_brightness, adjust_contrast, adjust_saturation, adjust_hue)


def to_tensor_normalize(img: np.ndarray) -> np.ndarray:
    """ToTensor (uint8 HWC -> float32 CHW / 255) then Normalize(0.5, 0.5): float32 sub and div."""
    x = img.transpose(2, 0, 1).astype(np.float32) / np.float32(255)
    return (x - np.float32(0.5)) / np.float32(0.5)


def apply(img: np.ndarray, job: Dict) -> np.ndarray:
    """One image through a job (the dict gan_variant_research_amd.dataio builds): crop -> bicubic resize -> window of the resized
    image -> flip -> jitter ops in `order` -> float32 CHW in [-1, 1]."""
    cy, cx, ch, cw = job["crop"]
    x = img[cy:cy + ch, cx:cx + cw]
    x = resize_bicubic(x, job["resize"][0], job["resize"][1])
    oy, ox, oh, ow = job["window"]
    x = x[oy:oy + oh, ox:ox + ow]
    if job["flip"]:
        x = x[:, ::-1]
    for op in job["order"]:
        if op >= 0:
            x = JITTER[op](np.ascontiguousarray(x), job["factor"][op])
    return to_tensor_normalize(np.ascontiguousarray(x))


def apply_pil(img: np.ndarray, job: Dict) -> np.ndarray:
    """The same job executed by Pillow itself, with the calls torchvision's PIL backend makes (the pin for `apply`)."""
    from PIL import Image, ImageEnhance
    im = Image.fromarray(img, "RGB")
    cy, cx, ch, cw = job["crop"]
    im = im.crop((cx, cy, cx + cw, cy + ch))
    im = im.resize((job["resize"][1], job["resize"][0]), Image.BICUBIC)
    oy, ox, oh, ow = job["window"]
    im = im.crop((ox, oy, ox + ow, oy + oh))
    if job["flip"]:
        im = im.transpose(Image.FLIP_LEFT_RIGHT)
    for op in job["order"]:
        if op == 0:
            im = ImageEnhance.Brightness(im).enhance(job["factor"][0])
        elif op == 1:
            im = ImageEnhance.Contrast(im).enhance(job["factor"][1])
        elif op == 2:
            im = ImageEnhance.Color(im).enhance(job["factor"][2])
        elif op == 3:
            h, s, v = im.convert("HSV").split()
            np_h = np.array(h, dtype=np.uint8)
            np_h = (np_h.astype(np.int32) + hue_shift(job["factor"][3])).astype(np.uint8)
            im = Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB")
    return to_tensor_normalize(np.asarray(im))
