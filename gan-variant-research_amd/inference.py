"""Forward-only generator inference (GAN_Variant1/generate_folder.py:112-252) on the HIP kernels.

`load_generator` takes the reference's checkpoints (or this package's: same layout) and prefers the EMA weights exactly as
generate_folder.py:125-170 does; `stylize` is the tensor-level body of `stylize_folder` (:207-252): G(x) then
clamp -> *0.5 + 0.5 -> *255 -> round -> uint8 (:183-185).  Decoding / resizing / JPEG encoding of files is the input pipeline
(torchvision + PIL in the reference) and stays outside this package; `stylize_folder` is provided for PIL-readable folders.
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Iterable, Optional

import torch

from ._lib import BF16, F32
from .cut import ResNetGenerator

_LEGACY_KEYS = ("ema_state_dict", "G_ema", "G_state_dict", "state_dict")


def pick_state_dict(ckpt: Dict) -> Dict[str, torch.Tensor]:
    """generate_folder.py:125-170: EMA shadow, then 'generator', then legacy names, then a raw / nested state dict."""
    ema = ckpt.get("ema_G")
    if isinstance(ema, dict) and isinstance(ema.get("shadow"), dict):
        return ema["shadow"]
    if isinstance(ckpt.get("generator"), dict):
        return ckpt["generator"]
    for k in _LEGACY_KEYS:
        if isinstance(ckpt.get(k), dict):
            return ckpt[k]
    if ckpt and all(isinstance(v, torch.Tensor) for v in ckpt.values()):
        return ckpt
    for v in ckpt.values():
        if isinstance(v, dict) and v and all(isinstance(x, torch.Tensor) for x in v.values()):
            return v
    raise KeyError(f"no generator state_dict in checkpoint (keys: {list(ckpt)[:10]})")


def load_generator(ckpt_path: str, device: str = "cuda", ngf: int = 64, n_blocks: int = 9, bf16: bool = True, use_graph: bool = False) -> ResNetGenerator:
    """generate_folder.py:189-205.  The file is read with weights_only=True (tensors and plain containers; nothing is unpickled)."""
    ckpt = torch.load(ckpt_path, map_location=device, weights_only=True)
    if not isinstance(ckpt, dict):
        raise ValueError(f"Checkpoint {ckpt_path} is not a dict; got {type(ckpt)}")
    G = ResNetGenerator(3, 3, ngf, n_blocks).to(device)
    missing, unexpected = G.load_state_dict(pick_state_dict(ckpt), strict=False)
    if missing or unexpected:
        print(f"[WARN] generator state_dict: {len(missing)} missing, {len(unexpected)} unexpected keys (e.g. {(list(missing) + list(unexpected))[:4]})")
    G.eval()
    for p in G.parameters():
        p.requires_grad_(False)
    G.compute_dtype = BF16 if bf16 else F32      # the reference runs inference under autocast (:237)
    # use_graph: forward-only passes replay one hipGraph per input shape (autograd._GenBridge.forward).  Off by default: on ROCm 7.2 the
    # replay of the ~110-node graph is slower than the eager launches at small batch (1.70 vs 1.14 ms at B=1, equal at B=16; tools/bench_infer.py)
    G.use_graph = use_graph
    return G


def to_uint8(y: torch.Tensor) -> torch.Tensor:
    """[-1,1] -> uint8, generate_folder.py:183-185 (stays on the device; the caller moves it)."""
    return y.clamp(-1, 1).mul(0.5).add(0.5).mul(255).round().byte()


@torch.inference_mode()
def stylize(G: ResNetGenerator, x: torch.Tensor) -> torch.Tensor:
    """(B,3,H,W) fp32 in [-1,1] on the GPU -> (B,3,H,W) uint8 on the GPU."""
    return to_uint8(G(x))


@torch.inference_mode()
def stylize_folder(G, src_dir: str, out_dir: str, device: str = "cuda", img_size: int = 256, batch: int = 16, limit: Optional[int] = None) -> int:
    """generate_folder.py:207-252 with PIL doing what torchvision's Resize(BILINEAR) / ToTensor / Normalize / ToPILImage do there."""
    import numpy as np
    from PIL import Image
    exts = {".jpg", ".jpeg", ".png", ".bmp", ".webp", ".tif", ".tiff"}
    src_root, out_root = Path(src_dir), Path(out_dir)
    paths = sorted(p for p in src_root.rglob("*") if p.suffix.lower() in exts)
    if limit is not None:
        paths = paths[:limit]
    if not paths:
        raise FileNotFoundError(f"No images found under: {src_dir}")
    out_root.mkdir(parents=True, exist_ok=True)
    for i in range(0, len(paths), batch):
        chunk = paths[i:i + batch]
        arr = np.stack([np.asarray(Image.open(p).convert("RGB").resize((img_size, img_size), Image.BILINEAR), dtype=np.float32) for p in chunk])
        x = torch.from_numpy(arr).permute(0, 3, 1, 2).div(255.0).sub(0.5).div(0.5).contiguous().to(device)
        y = stylize(G, x).cpu().permute(0, 2, 3, 1).numpy()
        for p, img in zip(chunk, y):
            save = (out_root / p.relative_to(src_root)).with_suffix(".jpg")
            save.parent.mkdir(parents=True, exist_ok=True)
            Image.fromarray(img).save(save, format="JPEG", quality=95, subsampling=0, optimize=True)
    return len(paths)
