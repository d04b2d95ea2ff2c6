"""Device-side input pipeline: the reference's per-image PIL transforms as one batched HIP launch sequence.

The reference decodes a JPEG on a DataLoader worker and then runs, still on the CPU and per image,

    GAN_Variant1/dataio/transforms.py:30-39   RandomCropResize(S, (0.85, 1.0)) -> RandomHorizontalFlip -> ColorJitter(.05,.05,.05,.02)
                                              -> ToTensor -> Normalize(0.5, 0.5)            (train)
    GAN_Variant1/dataio/transforms.py:42-49   Resize([S, S], BICUBIC) -> ToTensor -> Normalize           (eval)
    Basic_GAN/src/data.py:8-26                Resize(load, BICUBIC) -> RandomCrop(S) -> flip | Resize(S) -> CenterCrop(S), then the same tail

Here the decoder hands over uint8 HWC images (device tensors), the random parameters are drawn on the host in the reference's order
(`*_job` functions below: numpy's global generator for the crop, torch's for flip and jitter, exactly as the reference mixes them) and
`InputPipeline.run` produces the (B,3,S,S) fp32 batch in [-1,1] on the GPU -- bit-identical to what PIL + torchvision would have
produced for the same draws (tests/test_input_pipeline.py checks against Pillow itself; torchvision is absent from this image, its
glue is restated from its published source).  There is no CPU fallback: without the HIP library `run` raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import GanError, GanInputJob

BRIGHTNESS, CONTRAST, SATURATION, HUE = 0, 1, 2, 3


# ------------------------------------------------------------------------------------------------ jobs (host side, per image)
def _job(h, w, crop, resize, window, flip=False, order=(-1, -1, -1, -1), factor=(1.0, 1.0, 1.0, 0.0)) -> Dict:
    return {"size": (int(h), int(w)), "crop": tuple(int(v) for v in crop), "resize": tuple(int(v) for v in resize),
            "window": tuple(int(v) for v in window), "flip": bool(flip), "order": tuple(int(v) for v in order),
            "factor": tuple(float(v) for v in factor)}


def color_jitter_params(brightness=0.05, contrast=0.05, saturation=0.05, hue=0.02) -> Tuple[Tuple[int, ...], Tuple[float, ...]]:
    """torchvision ColorJitter.get_params: a permutation of the four ops, then one uniform draw each for brightness, contrast,
    saturation (in [max(0, 1-x), 1+x]) and hue (in [-hue, hue]), all from torch's global generator."""
    order = tuple(int(i) for i in torch.randperm(4))
    draw = lambda lo, hi: float(torch.empty(1).uniform_(lo, hi))
    b = draw(max(0.0, 1 - brightness), 1 + brightness)
    c = draw(max(0.0, 1 - contrast), 1 + contrast)
    s = draw(max(0.0, 1 - saturation), 1 + saturation)
    h = draw(-hue, hue)
    return order, (b, c, s, h)


def train_job(h: int, w: int, image_size: int = 256, scale=(0.85, 1.0)) -> Dict:
    """get_train_transforms (transforms.py:30-39).  Draw order: np.random.uniform, np.random.randint x2 (RandomCropResize, :19-23),
    torch.rand(1) (flip), torch.randperm(4) + four uniforms (ColorJitter)."""
    sc = np.random.uniform(*scale)
    cs = int(min(w, h) * sc)
    i = np.random.randint(0, h - cs + 1)
    j = np.random.randint(0, w - cs + 1)
    flip = bool(torch.rand(1) < 0.5)
    order, factor = color_jitter_params()
    return _job(h, w, (i, j, cs, cs), (image_size, image_size), (0, 0, image_size, image_size), flip, order, factor)


def eval_job(h: int, w: int, image_size: int = 256) -> Dict:
    """get_eval_transforms (transforms.py:42-49): the whole image resized to S x S."""
    return _job(h, w, (0, 0, h, w), (image_size, image_size), (0, 0, image_size, image_size))


def _resize_smaller_edge(h: int, w: int, size: int) -> Tuple[int, int]:
    """torchvision Resize(int): the smaller edge becomes `size`, the other int(size * long / short)."""
    if w <= h:
        return int(size * h / w), size
    return size, int(size * w / h)


def basic_job(h: int, w: int, load_size: int = 286, crop_size: int = 256, train: bool = True) -> Dict:
    """Basic_GAN/src/data.py:8-26.  Train: Resize(load_size) -> RandomCrop(crop_size) (torch.randint x2, skipped when nothing to crop)
    -> RandomHorizontalFlip (torch.rand(1)).  Eval: Resize(crop_size) -> CenterCrop(crop_size)."""
    if train:
        rh, rw = _resize_smaller_edge(h, w, load_size)
        if rh == crop_size and rw == crop_size:
            i = j = 0
        else:
            i = int(torch.randint(0, rh - crop_size + 1, size=(1,)))
            j = int(torch.randint(0, rw - crop_size + 1, size=(1,)))
        flip = bool(torch.rand(1) < 0.5)
        return _job(h, w, (0, 0, h, w), (rh, rw), (i, j, crop_size, crop_size), flip)
    rh, rw = _resize_smaller_edge(h, w, crop_size)
    i, j = int(round((rh - crop_size) / 2.0)), int(round((rw - crop_size) / 2.0))
    return _job(h, w, (0, 0, h, w), (rh, rw), (i, j, crop_size, crop_size))


# ------------------------------------------------------------------------------------------------ the device pipeline
class InputPipeline:
    """Batched transform on one GPU.  `run(images, jobs)`: images = uint8 (H, W, 3) device tensors (one per job, any sizes), jobs from
    the `*_job` functions (all with an S x S window) -> (B, 3, S, S) fp32 in [-1, 1].  Tap tables are cached per (source, target) size;
    jobs and tables travel in one pinned block and one asynchronous copy per batch."""

    def __init__(self, image_size: int, device, max_batch: int = 64, max_rows: int = 1024):
        self.S, self.device = int(image_size), torch.device(device)
        if self.device.type != "cuda":
            raise GanError("the input pipeline runs on the GPU (there is no CPU fallback)")
        self.lib = _lib.load()
        self.max_batch, self.max_rows = max_batch, max_rows
        S = self.S
        self._tmp = torch.zeros(max_batch * max_rows * S * 4, dtype=torch.uint8, device=self.device)
        self._img = torch.zeros(max_batch * S * S * 4, dtype=torch.uint8, device=self.device)
        self._mean = torch.zeros(max_batch, dtype=torch.int32, device=self.device)
        self._taps: Dict[Tuple[int, int], Tuple[np.ndarray, np.ndarray, int]] = {}
        self._block_bytes = 0
        self._host = self._dev = None

    def taps(self, in_size: int, out_size: int):
        """(bounds [out][2], taps [out][ksize], ksize) of Pillow's bicubic resize in_size -> out_size, from the library."""
        key = (in_size, out_size)
        t = self._taps.get(key)
        if t is None:
            k = self.lib.gan_resize_ksize(in_size, out_size)
            if k < 0:
                raise GanError(self.lib.gan_last_error().decode())
            bounds, kk = np.zeros((out_size, 2), np.int32), np.zeros((out_size, k), np.int32)
            _lib.check(self.lib.gan_resize_coeffs(in_size, out_size, bounds.ctypes.data, kk.ctypes.data, k), "gan_resize_coeffs")
            t = self._taps[key] = (bounds, kk, k)
        return t

    def run(self, images: Sequence[torch.Tensor], jobs: Sequence[Dict], out: Optional[torch.Tensor] = None) -> torch.Tensor:
        B, S = len(jobs), self.S
        if B == 0 or B != len(images) or B > self.max_batch:
            raise GanError(f"input pipeline: {len(images)} images, {B} jobs, max_batch {self.max_batch}")
        # ---- tables block: [jobs (B x 112 bytes) | int32 taps of every distinct (in, out) pair]
        offs, parts, n = {}, [], 0
        structs = (GanInputJob * B)()
        for b, (im, jb) in enumerate(zip(images, jobs)):
            if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3 or im.device != self.device or im.stride(2) != 1 or im.stride(1) != 3:
                raise GanError(f"input pipeline: image {b} must be a uint8 (H, W, 3) tensor on {self.device} with packed pixels")
            if tuple(im.shape[:2]) != jb["size"]:
                raise GanError(f"input pipeline: image {b} is {tuple(im.shape[:2])}, its job was drawn for {jb['size']}")
            if jb["window"][2:] != (S, S):
                raise GanError(f"input pipeline: job {b} has a {jb['window'][2:]} window, the pipeline produces {S}x{S}")
            cy, cx, ch, cw = jb["crop"]
            if ch > self.max_rows:
                raise GanError(f"input pipeline: image {b} needs {ch} source rows, max_rows is {self.max_rows}")
            js = structs[b]
            js.src, js.src_stride = im.data_ptr(), im.stride(0)
            js.crop_y, js.crop_x, js.crop_h, js.crop_w = cy, cx, ch, cw
            js.res_h, js.res_w = jb["resize"]
            js.win_y, js.win_x = jb["window"][:2]
            js.flip = int(jb["flip"])
            for s in range(4):
                js.order[s] = jb["order"][s]
                js.factor[s] = jb["factor"][s]
            js.hue_shift = int(jb["factor"][HUE] * 255) % 256       # torchvision adjust_hue: uint8(hue_factor * 255), wrapping
            for axis, (i_sz, o_sz) in (("h", (cw, js.res_w)), ("v", (ch, js.res_h))):
                key = (i_sz, o_sz)
                if key not in offs:
                    bounds, kk, k = self.taps(i_sz, o_sz)
                    offs[key] = (n, n + bounds.size, k)
                    parts += [bounds.reshape(-1), kk.reshape(-1)]
                    n += bounds.size + kk.size
                bo, ko, k = offs[key]
                if axis == "h":
                    js.hb_off, js.hk_off, js.hksize = bo, ko, k
                else:
                    js.vb_off, js.vk_off, js.vksize = bo, ko, k
        jbytes = C.sizeof(GanInputJob) * B
        total = jbytes + 4 * n
        if total > self._block_bytes:
            self._block_bytes = max(total * 2, 1 << 16)
            self._host = [torch.zeros(self._block_bytes, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
            self._dev = torch.zeros(self._block_bytes, dtype=torch.uint8, device=self.device)
            self._events, self._turn = [None, None], 0
        t = self._turn
        self._turn ^= 1
        if self._events[t] is not None:
            self._events[t].synchronize()
        host = self._host[t]
        hv = host.numpy()
        hv[:jbytes] = np.frombuffer(structs, dtype=np.uint8)
        hv[jbytes:total].view(np.int32)[:] = np.concatenate(parts)
        ts = torch.cuda.current_stream(self.device)          # the copy and the kernels are ordered on torch's current stream
        self._dev[:total].copy_(host[:total], non_blocking=True)
        if self._events[t] is None:
            self._events[t] = torch.cuda.Event()
        self._events[t].record()
        if out is None:
            out = torch.empty(B, 3, S, S, dtype=torch.float32, device=self.device)
        assert out.shape == (B, 3, S, S) and out.dtype == torch.float32 and out.is_contiguous() and out.device == self.device
        rc = self.lib.gan_input_pipeline(self._dev.data_ptr(), hv.ctypes.data, B, self._dev.data_ptr() + jbytes, S, self._tmp.data_ptr(), self.max_rows,
                                         self._img.data_ptr(), self._mean.data_ptr(), out.data_ptr(), ts.cuda_stream)
        _lib.check(rc, "gan_input_pipeline")
        return out


# ------------------------------------------------------------------------------------------------ reference-named constructors
class _Transform:
    """Callable in the shape of the reference's composed transform, but batched: `tf(images)` draws one job per image and returns the
    device batch.  `tf.last_jobs` keeps the draws (tests replay them through Pillow)."""

    def __init__(self, make_job, image_size, device, **kw):
        self.make_job, self.image_size = make_job, image_size
        self.pipe = InputPipeline(image_size, device, **kw)
        self.last_jobs: List[Dict] = []

    def __call__(self, images: Sequence[torch.Tensor]) -> torch.Tensor:
        self.last_jobs = [self.make_job(int(im.shape[0]), int(im.shape[1])) for im in images]
        return self.pipe.run(images, self.last_jobs)


def get_train_transforms(image_size: int = 256, use_gray_world: bool = False, device="cuda", **kw) -> _Transform:
    """transforms.py:30-39 (use_gray_world is accepted and unused there too)."""
    return _Transform(lambda h, w: train_job(h, w, image_size, (0.85, 1.0)), image_size, device, **kw)


def get_eval_transforms(image_size: int = 256, device="cuda", **kw) -> _Transform:
    """transforms.py:42-49."""
    return _Transform(lambda h, w: eval_job(h, w, image_size), image_size, device, **kw)


def basic_image_tf(load_size: int, crop_size: int, train: bool, device="cuda", **kw) -> _Transform:
    """Basic_GAN/src/data.py:8-26 `_image_tf`."""
    return _Transform(lambda h, w: basic_job(h, w, load_size, crop_size, train), crop_size, device, **kw)
