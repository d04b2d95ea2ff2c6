"""Data parallelism: N ranks on equal shards of the global batch, gradients summed by all-reduce and scaled
by 1/N inside the fused optimiser, must reproduce the single-rank step on the whole batch.  Runs on CPU: gloo, world_size 2,
kernels replaced by the emulator -- this checks the sharding / reduction / shared-randomness logic, not the kernels."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gan_variant_research_amd import cut as C
from tests import cases
from tests.emulator import EmuOps

S, BG = 32, 2


def _inputs():
    g = torch.Generator().manual_seed(1234)
    return torch.rand(BG, 3, S, S, generator=g) * 2 - 1, torch.rand(BG, 3, S, S, generator=g) * 2 - 1


def _global_randomness(tr_like):
    torch.manual_seed(4242)
    return tr_like.sample_randomness()


def _shard(rnd, lo, hi):
    out = {"nce_ids": rnd["nce_ids"]}   # shared by the whole global batch (patchnce_cut.py:63)
    for k in ("aug_real", "aug_fake_d", "aug_fake_g"):
        out[k] = {n: (v[lo:hi] if v.dim() > 0 else v) for n, v in rnd[k].items()}
    return out


def _make(B, world=1, pg=None):
    cfg = cases.small_config()
    C.set_seed(42)
    gen, disc = C.build_models(cfg, "cpu")
    return C.CutTrainer(gen, disc, cfg, B, S, device="cpu", amp=False, ops=EmuOps(), world_size=world, process_group=pg)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    bl = BG // world
    tr = _make(bl, world, dist.group.WORLD)
    photos, monets = _inputs()
    ref = _make(BG)                      # only used to draw the GLOBAL randomness with the same consumption order
    rnd = _shard(_global_randomness(ref), rank * bl, (rank + 1) * bl)
    losses = tr.train_step(0, photos[rank * bl:(rank + 1) * bl], monets[rank * bl:(rank + 1) * bl], rnd)
    if rank == 0:
        out["g"] = {k: v.clone() for k, v in tr.opt_G.params.items()}
        out["d"] = {k: v.clone() for k, v in tr.opt_D.params.items()}
    out[f"loss{rank}"] = losses
    dist.destroy_process_group()


def test_two_ranks_equal_one_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    torch.set_num_threads(4)
    single = _make(BG)
    photos, monets = _inputs()
    ref_losses = single.train_step(0, photos, monets, _global_randomness(single))
    for k in ("d_loss", "g_adv", "nce", "identity", "r1"):   # batch means: the global value is the mean of the rank values
        avg = 0.5 * (out["loss0"][k] + out["loss1"][k])
        np.testing.assert_allclose(avg, ref_losses[k], rtol=2e-4, atol=2e-5, err_msg=k)
    for name, opt in (("g", single.opt_G), ("d", single.opt_D)):
        for k, v in out[name].items():
            # Adam's first update is +-lr * sign(g): allow one sign flip (2 lr) on near-zero gradients, twice for D (R1 step)
            np.testing.assert_allclose(v.numpy(), opt.params[k].numpy(), rtol=0, atol=4.5e-4 * (2 if name == "d" else 1), err_msg=k)
