"""The command-line driver (gan_variant_research_amd/train_cutpp.py) against the reference's driver contract
(GAN_Variant1/training/train_cutpp.py:39-85, 340-498): flags, YAML schema incl. dead keys, `--set` coercion, checkpoint names and
layout, resume, loss CSV.  CPU: the step runs on the emulator (host logic only)."""
import os

import pytest
import torch
import yaml

from gan_variant_research_amd import train_cutpp as T
from tests.emulator import EmuOps

# the reference's YAML schema: the keys its driver reads plus a sample of the dead ones, values of the shipped config
SCHEMA = """
image_size: 256
batch_size: 12
epochs: 70
max_steps: null
data: {photos_dir: "data/photo_jpg", monet_dir: "data/monet_jpg", photos_tfrec: "data/photo_tfrec"}
output: {checkpoint_dir: "CKPT", log_dir: "LOGS"}
optim:
  G: {lr: 2.0e-4, betas: [0.5, 0.999], weight_decay: 0.0, scheduler: {type: cosine, lr_min: 5.0e-5}}
  D: {lr: 2.0e-4, betas: [0.5, 0.999], weight_decay: 0.0, scheduler: {type: cosine, lr_min: 5.0e-5}}
grad_clip_g: 10.0
grad_clip_d: 10.0
loss_weights: {adv: 1.0, patchnce: 1.0, identity_warm: 0.1, identity_final: 0.0, palette: 0.0, repulsion: 0.0, featmatch: 0.0}
warmup_steps: 20000
model:
  generator: {base: resnet9, n_downsampling: 2, n_blocks: 9, ngf: 64, norm: instance, activation: relu, padding_type: reflect, use_attention: false}
  discriminator: {base: patchgan, num_scales: 1, ndf: 64, n_layers: 3, norm: none, use_spectral_norm: false, receptive_field: 70}
patchnce: {num_patches: 256, temperature: 0.07, nce_layers: [0, 4, 8, 12, 16], nce_includes_all_negatives_from_minibatch: false}
diffaugment: {enable: true, policy: [color, translation, cutout]}
r1: {gamma: 10.0, every: 16}
ema: {decay: 0.999, warmup_steps: 100}
metrics: {compute_fid: false, eval_every: 500, save_checkpoint_every: 2000}
io: {num_workers: 8, pin_memory: true, amp: true}
early_stop: {enable: false}
"""


def test_override_config_coercion_rules():
    cfg = {"a": {"b": 1}, "loss_weights": {"adv": 1.0}}
    out = T.override_config(cfg, ["a.b=2", "loss_weights.adv=0.5", "amp=false", "x.y.z=True", "name=resnet9", "skipped", "k=1e-3", "neg=-4"])
    assert out["a"]["b"] == 2 and isinstance(out["a"]["b"], int)
    assert out["loss_weights"]["adv"] == 0.5 and out["amp"] is False and out["x"]["y"]["z"] is True
    assert out["name"] == "resnet9" and out["k"] == 1e-3 and out["neg"] == -4 and "skipped" not in out


def test_cli_defaults_match_the_reference():
    a = T.parse_args([])
    assert a.config == "GAN_Variant1/configs/train_gan_cutpp.yaml" and a.resume is None and a.set == []
    a = T.parse_args(["--config", "c.yaml", "--resume", "r.pt", "--set", "a=1", "b.c=2"])
    assert (a.config, a.resume, a.set) == ("c.yaml", "r.pt", ["a=1", "b.c=2"])


@pytest.mark.parametrize("source", ["schema", "reference_yaml"])
def test_driver_runs_checkpoints_and_resumes(tmp_path, source):
    if source == "reference_yaml":
        cfg_path = "/root/reference/GAN_Variant1/configs/train_gan_cutpp.yaml"      # the shipped config, unchanged (only where the reference is present)
        if not os.path.exists(cfg_path):
            pytest.skip("reference checkout not present")
    else:
        cfg_path = str(tmp_path / "cfg.yaml")
        with open(cfg_path, "w") as f:
            f.write(SCHEMA)
    torch.set_num_threads(4)
    ck, lg = str(tmp_path / "ck"), str(tmp_path / "lg")
    sets = ["image_size=32", "batch_size=2", "max_steps=2", "amp=false", f"output.checkpoint_dir={ck}", f"output.log_dir={lg}",
            "metrics.save_checkpoint_every=1", "log_every=1"]
    r = T.main(["--config", cfg_path, "--synthetic", "--set"] + sets, ops=EmuOps(), device="cpu")
    assert r["step"] == 2 and set(r["losses"]) == {"d_loss", "g_adv", "nce", "identity", "r1", "identity_weight", "g_loss"}
    assert sorted(os.listdir(ck)) == ["ckpt_final.pt", "ckpt_step1.pt"]
    ckpt = torch.load(os.path.join(ck, "ckpt_final.pt"), weights_only=True)
    assert ckpt["step"] == 2 and {"generator", "discriminator", "opt_G", "opt_D", "ema_G", "config", "metrics", "scaler"} <= set(ckpt)
    assert "initial.1.weight" in ckpt["generator"] and "shadow" in ckpt["ema_G"]
    rows = open(os.path.join(lg, "losses_history.csv")).read().strip().splitlines()
    assert rows[0] == "step,d_loss,g_loss" and [ln.split(",")[0] for ln in rows[1:]] == ["0", "1"]
    assert open(os.path.join(lg, "train_log.txt")).read().startswith("Step 1: {")
    # --resume continues at the stored step with the stored optimiser state
    r2 = T.main(["--config", cfg_path, "--synthetic", "--resume", os.path.join(ck, "ckpt_final.pt"), "--set"] + sets + ["max_steps=3"], ops=EmuOps(), device="cpu")
    assert r2["step"] == 3
    assert yaml.safe_load(open(cfg_path))["batch_size"] == 12       # the file itself is never rewritten
    # a wrong data path is an error (as in the reference), not a silent run on noise: the synthetic batches need the explicit flag
    with pytest.raises(FileNotFoundError, match="photos_dir"):
        T.main(["--config", cfg_path, "--set"] + sets + ["data.photos_dir=/nonexistent/photos"], ops=EmuOps(), device="cpu")
