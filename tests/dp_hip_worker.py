"""One data-parallel rank on a real GPU over RCCL (started by tests/test_gpu_parity.py::test_two_gpu_ranks_equal_one_rank as a
fresh process per GPU): runs CUT step 0 on its shard with the two-bucket generator all-reduce on and saves the summed gradients."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    rank, world, out_path = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), sys.argv[1]
    from gan_variant_research_amd import cut as C
    from gan_variant_research_amd.runtime import HipOps
    from tests import cases, test_dp_gloo as T
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    ops = HipOps(dev)
    ops.bind_queues()                                  # compute streams take their hardware queues before RCCL's do
    dist.init_process_group("nccl", device_id=dev)
    bl = T.BG // world
    cfg = cases.small_config()
    C.set_seed(42)
    gen, disc = C.build_models(cfg, "cpu")
    tr = C.CutTrainer(gen, disc, cfg, bl, T.S, device=dev, amp=False, ops=ops, world_size=world, process_group=dist.group.WORLD)
    assert tr._bucket_off > 0                          # the merged backward carries the tail-bucket callback
    photos, monets = T._inputs()
    ref = T._make(T.BG)                                # CPU twin, only to draw the GLOBAL randomness in the same order
    rnd = T._shard(T._global_randomness(ref), rank * bl, (rank + 1) * bl)
    losses = tr.train_step(0, photos[rank * bl:(rank + 1) * bl].to(dev), monets[rank * bl:(rank + 1) * bl].to(dev), rnd)
    torch.cuda.synchronize()
    torch.save({"losses": losses, "flat_g": tr.opt_G.flat_g.cpu(), "flat_gd": tr.opt_D.flat_g.cpu(), "bucket_off": tr._bucket_off},
               f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
