"""Training driver with the reference's command line (GAN_Variant1/training/train_cutpp.py:39-85, 340-498):

    python -m gan_variant_research_amd.train_cutpp --config GAN_Variant1/configs/train_gan_cutpp.yaml [--resume CKPT] [--set a.b=c ...]

Same flags, same YAML schema (the keys the reference reads; its dead keys are accepted and ignored), same `--set` coercion
(true/false -> bool, then int, then float, else string), same checkpoint layout and file names, same loss CSV / JSON log lines.
What differs is what runs underneath: the step is the fused `cut.CutTrainer` on the HIP kernels instead of torch.nn modules.
Image folders are read with Pillow on the host and transformed on the device (dataio.py, the reference's train transform); when the
configured folders do not exist -- or with `--synthetic` -- uniform noise batches of the right shape stand in (there is no dataset on
the benchmark box).  Build-only keys live under `mi355x:` (amp dtype, synthetic data).
"""
from __future__ import annotations

import argparse
import json
import random
from collections import defaultdict
from pathlib import Path
from typing import Iterator, List, Optional

import numpy as np
import torch
import yaml

from . import cut as C

IMAGE_EXTS = {".jpg", ".jpeg", ".png", ".bmp", ".webp"}


def parse_args(argv=None):
    """train_cutpp.py:39-48."""
    p = argparse.ArgumentParser(description="Train CUT++ GAN (MI355X-native step)")
    p.add_argument("--config", type=str, default="GAN_Variant1/configs/train_gan_cutpp.yaml", help="Path to config file")
    p.add_argument("--resume", type=str, default=None, help="Path to checkpoint to resume from")
    p.add_argument("--set", nargs="+", default=[], help="Override config values (e.g., loss_weights.adv=0.5)")
    p.add_argument("--synthetic", action="store_true", help="uniform-noise batches instead of the image folders (build-only flag)")
    return p.parse_args(argv)


def override_config(config: dict, overrides) -> dict:
    """train_cutpp.py:51-85: `a.b.c=value`; missing intermediate dicts are created; value coercion true/false, int, float, string;
    entries without '=' are skipped."""
    for item in overrides:
        if "=" not in item:
            continue
        path, value = item.split("=", 1)
        keys = path.split(".")
        cur = config
        for k in keys[:-1]:
            if k not in cur:
                cur[k] = {}
            cur = cur[k]
        low = value.lower()
        if low == "true":
            value = True
        elif low == "false":
            value = False
        else:
            for cast in (int, float):
                try:
                    value = cast(value)
                    break
                except ValueError:
                    pass
        cur[keys[-1]] = value
    return config


def _list_images(folder) -> List[Path]:
    root = Path(folder)
    return sorted(p for p in root.rglob("*") if p.suffix.lower() in IMAGE_EXTS) if root.is_dir() else []


def folder_batches(paths: List[Path], batch: int, image_size: int, device, seed: int) -> Iterator[torch.Tensor]:
    """Shuffled, drop_last epochs over an image folder: Pillow decode on the host, the reference's train transform on the device
    (dataio.get_train_transforms: random-crop-resize bicubic, flip, ColorJitter, normalise -- transforms.py:10-39)."""
    from PIL import Image
    from .dataio import get_train_transforms
    tf = get_train_transforms(image_size, device=device)
    rng = random.Random(seed)
    while True:
        order = list(paths)
        rng.shuffle(order)
        for i in range(0, len(order) - batch + 1, batch):
            imgs = [torch.from_numpy(np.asarray(Image.open(p).convert("RGB"))) for p in order[i:i + batch]]
            yield tf(imgs)


def synthetic_batches(batch: int, image_size: int, device, seed: int) -> Iterator[torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    while True:
        yield (torch.rand(batch, 3, image_size, image_size, generator=g) * 2 - 1).to(device)


class LossLog:
    """utils/loss_tracker.py:25-42 (CSV `step,d_loss,g_loss`, flushed per step) and the JSON line of train_cutpp.py:449-459."""

    def __init__(self, log_dir: Path):
        log_dir.mkdir(parents=True, exist_ok=True)
        self.csv = open(log_dir / "losses_history.csv", "a")      # the file name of utils/loss_tracker.py:17
        if self.csv.tell() == 0:
            self.csv.write("step,d_loss,g_loss\n")
        self.txt = log_dir / "train_log.txt"

    def step(self, step, losses):
        self.csv.write(f"{step},{losses['d_loss']},{losses['g_loss']}\n")
        self.csv.flush()

    def summary(self, step, avg):
        with open(self.txt, "a") as f:
            f.write(f"Step {step}: {json.dumps(avg)}\n")

    def close(self):
        self.csv.close()


def main(argv=None, ops=None, device: Optional[str] = None) -> dict:
    """train_cutpp.py:340-498.  `ops` / `device` are test hooks (the CPU suite drives the host logic through the emulator)."""
    args = parse_args(argv)
    with open(args.config) as f:
        config = yaml.safe_load(f)
    config = override_config(config, args.set)
    C.set_seed(config.get("seed", 42))
    device = torch.device(device if device is not None else "cuda")
    print(f"Using device: {device}")
    ckpt_dir, log_dir = Path(config["output"]["checkpoint_dir"]), Path(config["output"]["log_dir"])
    ckpt_dir.mkdir(parents=True, exist_ok=True)
    log = LossLog(log_dir)
    B, S = int(config["batch_size"]), int(config["image_size"])
    build = config.get("mi355x", {}) or {}
    photos_paths, monet_paths = _list_images(config["data"]["photos_dir"]), _list_images(config["data"]["monet_dir"])
    synthetic = bool(args.synthetic or build.get("synthetic", False))
    if not synthetic:      # like the reference, a wrong data path is an error -- never a silent run on noise that still writes ckpt_*.pt
        for name, paths in (("photos_dir", photos_paths), ("monet_dir", monet_paths)):
            if len(paths) < B:
                raise FileNotFoundError(f"data.{name} = {config['data'][name]!r} holds {len(paths)} images (< batch_size {B}); "
                                        "pass --synthetic (or mi355x.synthetic: true) to train on uniform-noise batches instead")
    if synthetic:
        print("[train_cutpp] --synthetic: uniform-noise batches stand in for the data loaders")
        photos_it, monet_it = synthetic_batches(B, S, device, 1234), synthetic_batches(B, S, device, 4321)
        steps_per_epoch = 7038 // B          # the reference's photo count (train_gan_cutpp.yaml: 70 epochs x 7038 // 12 steps)
    else:
        seed = config.get("seed", 42)
        photos_it, monet_it = folder_batches(photos_paths, B, S, device, seed), folder_batches(monet_paths, B, S, device, seed + 1)
        steps_per_epoch = len(photos_paths) // B
        print(f"Photos: {len(photos_paths)}, Monet: {len(monet_paths)}")

    generator, discriminator = C.build_models(config, "cpu")
    trainer = C.CutTrainer(generator, discriminator, config, B, S, device=device, amp=config.get("amp", True), ops=ops)
    start_step = 0
    if args.resume:
        start_step = int(trainer.load_checkpoint(args.resume)["step"])
        print(f"Resumed from step {start_step}")
    max_steps = config.get("max_steps", None)
    if max_steps is None:
        max_steps = config["epochs"] * steps_per_epoch
    print(f"Training for {max_steps} steps")

    acc = defaultdict(list)
    step, losses = start_step, {}
    while step < max_steps:
        losses = trainer.train_step(step, next(photos_it), next(monet_it))
        for k, v in losses.items():
            acc[k].append(v)
        log.step(step, losses)
        if step % config.get("log_every", 100) == 0 and step > 0:
            log.summary(step, {k: float(np.mean(v)) for k, v in acc.items()})
            acc.clear()
        if step % config["metrics"]["save_checkpoint_every"] == 0 and step > 0:
            path = ckpt_dir / f"ckpt_step{step}.pt"
            trainer.save_checkpoint(str(path), step)
            print(f"\nSaved checkpoint to {path}")
        step += 1
    final = ckpt_dir / "ckpt_final.pt"
    trainer.save_checkpoint(str(final), step)
    print(f"\nTraining complete. Final checkpoint: {final}")
    log.close()
    return {"step": step, "losses": losses, "checkpoint": str(final)}


if __name__ == "__main__":
    main()
