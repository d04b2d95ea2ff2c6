"""Device-side input pipeline.

CPU: the numpy restatement (oracle/input_ref.py) is pinned bit-exactly against Pillow itself -- the library that executes the
reference's transforms (GAN_Variant1/dataio/transforms.py:10-49, Basic_GAN/src/data.py:8-26) -- and against a committed fixture
(tests/golden/input_pipeline.npz, produced by Pillow through oracle/make_golden.py); the library's host-side tap generator is checked
against the restatement; the job samplers are checked for draw order and ranges.  torchvision is absent from this image: which Pillow
call each transform makes is restated from its published source ("parity unpinned" at that boundary).
GPU: `InputPipeline.run` must equal Pillow bit for bit on the final fp32 batch."""
import os

import numpy as np
import pytest
import torch

from gan_variant_research_amd import _lib, dataio
from oracle import input_ref as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "input_pipeline.npz")


def _rand_image(rng, h, w, smooth=False):
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if smooth:      # photo-like: low-frequency content, so that saturation / hue see realistic chroma
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([127 + 120 * np.sin(yy / 17.0 + c) * np.cos(xx / 23.0 - c) for c in range(3)], -1)
        img = np.clip(base + rng.integers(-20, 21, (h, w, 3)), 0, 255).astype(np.uint8)
    return img


def test_resize_restatement_equals_pillow():
    from PIL import Image
    rng = np.random.default_rng(0)
    for (H, W, oh, ow) in [(256, 256, 256, 256), (217, 217, 256, 256), (240, 230, 256, 256), (256, 256, 286, 286), (300, 400, 256, 256),
                           (700, 513, 256, 256), (64, 50, 96, 96), (255, 257, 256, 256), (333, 500, 286, 429), (31, 37, 8, 8)]:
        img = _rand_image(rng, H, W)
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BICUBIC))
        assert np.array_equal(R.resize_bicubic(img, oh, ow), ref), (H, W, oh, ow)


def test_enhance_restatements_equal_pillow():
    from PIL import Image, ImageEnhance
    rng = np.random.default_rng(1)
    for trial in range(12):
        img = _rand_image(rng, 48, 40, smooth=trial % 2 == 0)
        pil = Image.fromarray(img)
        for f in list(rng.uniform(0.95, 1.05, 6)) + [0.0, 0.5, 1.0, 1.05, 0.95, 1.5, 2.0]:
            f = float(f)
            assert np.array_equal(R.adjust_brightness(img, f), np.asarray(ImageEnhance.Brightness(pil).enhance(f))), ("brightness", f)
            assert np.array_equal(R.adjust_contrast(img, f), np.asarray(ImageEnhance.Contrast(pil).enhance(f))), ("contrast", f)
            assert np.array_equal(R.adjust_saturation(img, f), np.asarray(ImageEnhance.Color(pil).enhance(f))), ("saturation", f)
    assert np.array_equal(R.to_gray(img), np.asarray(pil.convert("L")))


def test_hsv_restatements_equal_pillow_exhaustively():
    """All 2^24 RGB triples through convert("HSV"), all 2^24 HSV triples through convert("RGB")."""
    from PIL import Image
    v = np.arange(256, dtype=np.uint8)
    a, b, c = np.meshgrid(v, v, v, indexing="ij")
    cube = np.stack([a, b, c], -1).reshape(4096, 4096, 3)
    assert np.array_equal(R.rgb_to_hsv(cube), np.asarray(Image.fromarray(cube, "RGB").convert("HSV")))
    assert np.array_equal(R.hsv_to_rgb(cube), np.asarray(Image.fromarray(cube, "HSV").convert("RGB")))


def _jobs(rng_seed, sizes, kind):
    np.random.seed(rng_seed)
    torch.manual_seed(rng_seed)
    if kind == "train":
        return [dataio.train_job(h, w, 64) for h, w in sizes]
    if kind == "eval":
        return [dataio.eval_job(h, w, 64) for h, w in sizes]
    return [dataio.basic_job(h, w, 72, 64, train=(kind == "basic_train")) for h, w in sizes]


SIZES = [(64, 64), (80, 100), (131, 97), (256, 256), (70, 300)]


@pytest.mark.parametrize("kind", ["train", "eval", "basic_train", "basic_eval"])
def test_chain_restatement_equals_pillow(kind):
    rng = np.random.default_rng(3)
    for (h, w), job in zip(SIZES, _jobs(11, SIZES, kind)):
        img = _rand_image(rng, h, w, smooth=True)
        got, want = R.apply(img, job), R.apply_pil(img, job)
        assert got.shape == (3, 64, 64) and got.dtype == np.float32
        assert np.array_equal(got, want), (kind, h, w, job)


def test_restatement_equals_committed_fixture():
    g = np.load(GOLD)
    for i in range(int(g["n"])):
        job = {"crop": tuple(g[f"{i}.crop"]), "resize": tuple(g[f"{i}.resize"]), "window": tuple(g[f"{i}.window"]), "flip": bool(g[f"{i}.flip"]),
               "order": tuple(g[f"{i}.order"]), "factor": tuple(float(v) for v in g[f"{i}.factor"])}
        assert np.array_equal(R.apply(g[f"{i}.image"], job), g[f"{i}.out"]), i


def test_library_tap_tables_equal_restatement():
    lib = _lib.load()
    for (i, o) in [(217, 256), (256, 256), (256, 286), (1024, 256), (50, 96), (513, 256), (300, 72), (8, 64), (64, 8)]:
        bounds, kk = R.resize_coeffs(i, 0, i, o)
        k = lib.gan_resize_ksize(i, o)
        assert k == kk.shape[1]
        b2, k2 = np.zeros((o, 2), np.int32), np.zeros((o, k), np.int32)
        assert lib.gan_resize_coeffs(i, o, b2.ctypes.data, k2.ctypes.data, k) == 0
        assert np.array_equal(b2, bounds) and np.array_equal(k2, kk), (i, o)
    assert lib.gan_resize_coeffs(10, 20, b2.ctypes.data, k2.ctypes.data, 3) != 0 and b"ksize" in lib.gan_last_error()


def test_job_samplers_draw_in_the_reference_order():
    """transforms.py:19-23 draws the crop from numpy's global generator; flip and ColorJitter come from torch's (torchvision)."""
    np.random.seed(5); torch.manual_seed(5)
    job = dataio.train_job(200, 300, 256)
    np.random.seed(5); torch.manual_seed(5)
    sc = np.random.uniform(0.85, 1.0); cs = int(200 * sc)
    i = np.random.randint(0, 200 - cs + 1); j = np.random.randint(0, 300 - cs + 1)
    flip = bool(torch.rand(1) < 0.5)
    order = tuple(int(v) for v in torch.randperm(4))
    f = [float(torch.empty(1).uniform_(lo, hi)) for lo, hi in ((0.95, 1.05), (0.95, 1.05), (0.95, 1.05), (-0.02, 0.02))]
    assert job["crop"] == (i, j, cs, cs) and job["flip"] == flip and job["order"] == order and job["factor"] == tuple(f)
    assert job["resize"] == (256, 256) and job["window"] == (0, 0, 256, 256)
    assert 0.85 * 200 - 1 <= cs <= 200 and sorted(order) == [0, 1, 2, 3]
    e = dataio.eval_job(200, 300, 256)
    assert e["crop"] == (0, 0, 200, 300) and e["order"] == (-1,) * 4 and not e["flip"]
    b = dataio.basic_job(200, 300, 286, 256, train=True)
    assert b["resize"] == (286, 429) and b["window"][2:] == (256, 256) and 0 <= b["window"][0] <= 30 and 0 <= b["window"][1] <= 173
    c = dataio.basic_job(200, 300, 286, 256, train=False)
    assert c["resize"] == (256, 384) and c["window"] == (0, 64, 256, 256)


def test_pipeline_needs_the_gpu():
    with pytest.raises(_lib.GanError):
        dataio.InputPipeline(64, "cpu")


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["train", "eval", "basic_train", "basic_eval"])
def test_device_pipeline_equals_pillow(kind):
    rng = np.random.default_rng(7)
    dev = torch.device("cuda:0")
    pipe = dataio.InputPipeline(64, dev, max_batch=8, max_rows=320)
    for rep in range(2):
        jobs = _jobs(21 + rep, SIZES, kind)
        imgs = [_rand_image(rng, h, w, smooth=True) for h, w in SIZES]
        out = pipe.run([torch.from_numpy(im).to(dev) for im in imgs], jobs)
        torch.cuda.synchronize()
        for b, (im, job) in enumerate(zip(imgs, jobs)):
            want = R.apply_pil(im, job)
            assert np.array_equal(out[b].cpu().numpy(), want), (kind, rep, b, job, float(np.abs(out[b].cpu().numpy() - want).max()))


@pytest.mark.gpu
def test_device_pipeline_equals_committed_fixture_and_full_size():
    dev = torch.device("cuda:0")
    g = np.load(GOLD)
    pipe = dataio.InputPipeline(32, dev, max_batch=8, max_rows=128)
    n = int(g["n"])
    jobs = [{"size": tuple(g[f"{i}.image"].shape[:2]), "crop": tuple(int(v) for v in g[f"{i}.crop"]), "resize": tuple(int(v) for v in g[f"{i}.resize"]),
             "window": tuple(int(v) for v in g[f"{i}.window"]), "flip": bool(g[f"{i}.flip"]), "order": tuple(int(v) for v in g[f"{i}.order"]),
             "factor": tuple(float(v) for v in g[f"{i}.factor"])} for i in range(n)]
    out = pipe.run([torch.from_numpy(g[f"{i}.image"]).to(dev) for i in range(n)], jobs)
    for i in range(n):
        assert np.array_equal(out[i].cpu().numpy(), g[f"{i}.out"]), i
    # the metric's size: 16 photos of 256x256 through the train transform, against the restatement
    rng = np.random.default_rng(9)
    tf = dataio.get_train_transforms(256, device=dev, max_batch=16, max_rows=256)
    np.random.seed(1); torch.manual_seed(1)
    imgs = [_rand_image(rng, 256, 256, smooth=True) for _ in range(16)]
    out = tf([torch.from_numpy(im).to(dev) for im in imgs])
    assert out.shape == (16, 3, 256, 256) and float(out.min()) >= -1.0 and float(out.max()) <= 1.0
    for b in (0, 7, 15):
        assert np.array_equal(out[b].cpu().numpy(), R.apply(imgs[b], tf.last_jobs[b])), b
    with pytest.raises(_lib.GanError):
        pipe.run([torch.zeros(40, 40, 3, dtype=torch.uint8, device=dev)], [dataio.eval_job(41, 40, 32)])
