"""Benchmark of the hot path: images/sec of the full CUT G+D train step (256x256, batch 16 per GPU) on MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one call of CutTrainer.train_step (GAN_Variant1/training/train_cutpp.py:206-331 of the reference): shared
G forward, D step (+ lazy R1 every 16th step), G step with PatchNCE and identity loss, fused clip+Adam+EMA updates,
on synthetic inputs already resident in HBM.  Rank 0 prints ONE JSON line.  Extra objects:
  roofline     -- the dominant kernel (conv_patch_kernel: all of its launches of one step) timed live with HIP events on the
                  launch stream, against the dense bf16 MFMA peak;
  cpu_baseline -- the PyTorch-CPU oracle's train step timed on this host's cores on a bounded sample (B=2);
  power        -- board power and firmware shader clock of this GPU during the timed steps (hwmon), beside the board's power cap:
                  the step's MFMA kernels run power-limited, which bounds `roofline.frac` well below 1.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3
PEAK_BF16_AT_CAP_TFLOPS = 1988.0   # bare MFMA loop on random bf16 at the 1400 W board limit (profiles/r03_mfma_power.txt, the faster of two boxes)
PEAK_FP8_TFLOPS = 5000.0    # dense fp8 MFMA peak (v_mfma_scale_f32_16x16x128_f8f6f4), same guide
# canonical algorithmic conv FLOPs of one step per image at 256^2 (identity on) = 935.9 GFLOP, MINUS what this build does not
# execute: the reference's get_feature_layers also runs the second up-sampling layer, whose output no PatchNCE layer id reaches (ids stop
# at 12); the feature pass here stops behind id 12.  ConvT 128->64 on 128x128: 2.416 GFLOP forward, x3 with both gradients.
GFLOP_PER_IMAGE = 935.9 - 3 * 2.416


def default_config():
    return {
        "loss_weights": {"adv": 1.0, "patchnce": 1.0, "identity_warm": 0.1, "identity_final": 0.0},
        "warmup_steps": 20000, "grad_clip_g": 10.0, "grad_clip_d": 10.0,
        "patchnce": {"nce_layers": [0, 4, 8, 12, 16], "temperature": 0.07, "num_patches": 256},
        "r1": {"gamma": 10.0, "every": 16}, "ema": {"decay": 0.999},
        "diffaugment": {"enable": True, "policy": ["color", "translation", "cutout"]},
        "model": {"generator": {"ngf": 64, "n_blocks": 9, "n_downsampling": 2, "padding_type": "reflect", "norm": "instance", "activation": "relu"},
                  "discriminator": {"ndf": 64, "n_layers": 3, "num_scales": 1, "use_spectral_norm": False}},
        "optim": {"G": {"lr": 2e-4, "betas": [0.5, 0.999], "weight_decay": 0.0}, "D": {"lr": 2e-4, "betas": [0.5, 0.999], "weight_decay": 0.0}},
    }


def dominant_kernel_roofline(trainer, iters=10):
    """Roofline of the dominant kernel of the step, measured live.

    bf16: `conv_patch_kernel` (csrc/conv_patch.hip) -- every convolution / input-gradient launch of one non-R1 step that the
    range-patch path takes (the 3x3 256->256 residual convolutions and their dgrads, ConvT phases, deep D layers).  All of
    them are replayed back to back `iters` times between two HIP events on the launch stream, so that
      ms_per_launch = elapsed / (iters * launches)  is the figure `rocprofv3 --kernel-trace --stats` reports as the kernel's
      average duration for the same step (profiles/), and
      achieved = sum of algorithmic FLOPs (ConvCall.alg_flops: 2*M*N*K with M the REFERENCE op's output pixels -- an input
      gradient launched on the reflect-padded 66x66 domain is priced as 64x64) / elapsed.
    fp32 (--fp32): the same over every `conv_igemm_kernel` launch.  The single 3x3 256->256 forward (77.3 GFLOP at B=16) is
    reported beside it as `res_fwd_*`."""
    bf16 = trainer.amp.enabled
    fp8 = getattr(trainer, "fp8", False)
    progs = [trainer.prog_gfwd, trainer.prog_d_compute, trainer.prog_d_update, trainer.prog_g_features, trainer.prog_g_adversarial, trainer.prog_g_features_bwd, trainer.prog_g_compute, trainer.prog_g_identity, trainer.prog_g_update]
    calls = [o for p in progs if p is not None for o in p.ops if getattr(o, "conv", None) is not None and (o.conv.w_frag or not bf16)]
    if fp8:      # configs[4]: price the e4m3 launches of the kernel (the residual convolutions) against the dense fp8 peak
        calls = [o for o in calls if o.conv.x.dtype == 2]
    flops = sum(o.conv.alg_flops() for o in calls)      # 2*M*N*K with M the reference op's output pixels

    def timed(ops, n):
        """ms per pass over `ops`, between two HIP events on the caller's stream; launches that the programs put on the trainer's
        other streams (discriminator, weight-gradient side streams) are joined into that stream before the closing event."""
        main = torch.cuda.current_stream()
        for o in ops:
            o()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s in trainer.side_streams():
            s.wait_event(e0)
        for _ in range(n):
            for o in ops:
                o()
        for s in trainer.side_streams():
            ev = torch.cuda.Event()
            ev.record(s)
            main.wait_event(ev)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    ms = timed(calls, iters)
    conv = trainer.G.c_blk[0][0]
    x, y = trainer.p1.acts[2], trainer.p1.raw[3][0]
    res_ms = timed(conv.fwd8(trainer.p1.in8[0], y) if fp8 else conv.fwd(x, y), 20)
    res_flops = 2.0 * x.B * y.H * y.W * conv.cout * conv.cin * 9   # x.B = 2 x batch while the identity pass rides in the generator pass
    peak = PEAK_FP8_TFLOPS if fp8 else PEAK_BF16_TFLOPS if bf16 else PEAK_F32_TFLOPS
    ach = flops / (ms * 1e-3) / 1e12
    # algorithmic HBM bytes of the same launches (every operand tensor once): input incl. its halo, output, weight copy
    esz = {0: 4, 1: 2, 2: 1}
    alg_bytes = sum(o.conv.B * o.conv.x.Hp * o.conv.x.Wp * o.conv.x.C * esz[o.conv.x.dtype] + o.conv.B * o.conv.Ho * o.conv.Wo * o.conv.Nst * esz[o.conv.out.dtype]
                    + o.conv.Nw * o.conv.ntaps * o.conv.Cin * esz[o.conv.x.dtype] for o in calls)
    step_flop = GFLOP_PER_IMAGE * 1e9 * trainer.B * (trainer.S / 256.0) ** 2      # the FLOP count scales with the pixel count
    # the HBM-bound side of the path: every InstanceNorm forward / backward launch of the step, replayed the same way;
    # achieved = algorithmic bytes (each operand tensor once) / elapsed
    norm = [o for p in progs if p is not None for o in p.ops if getattr(o, "hbm_bytes", None)]
    norm_ms = timed(norm, iters)
    norm_bytes = float(sum(o.hbm_bytes for o in norm))
    trainer.norm_hbm = {"bound": "hbm", "achieved": round(norm_bytes / (norm_ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(norm_bytes / (norm_ms * 1e-3) / 8e12, 4), "kernel": "in_apply / in_bwd_* (InstanceNorm forward and backward)",
                        "launches_per_step": len(norm), "bytes_per_step": norm_bytes, "ms_per_step": round(norm_ms, 4)}
    return {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            # PMC bytes exist for the configuration the committed passes ran (bf16, 256x256, batch 16): anything else reports null
            "traffic": pmc_traffic("conv_patch") if (bf16 and not fp8 and trainer.S == 256 and trainer.B == 16) else None,
            "traffic_algorithmic": round(alg_bytes / max(len(calls), 1)),
            "kernel": ("conv_patch_kernel<..., FP8> (e4m3 launches)" if fp8 else "conv_patch_kernel (+ conv_patch_bwdchain_kernel: the same body with the "
                       "backward-chain epilogue)") if bf16 else "conv_igemm_kernel<float,...>",
            "launches_per_step": len(calls), "flop_per_step": flops, "ms_per_launch": round(ms / max(len(calls), 1), 5),
            "share_of_step_conv_flop": round(flops / step_flop, 3),
            "res_fwd_tflops": round(res_flops / (res_ms * 1e-3) / 1e12, 2), "res_fwd_ms": round(res_ms, 4),
            # what the 1400 W board limit leaves of `peak` for bf16 MFMAs on non-zero data: a bare loop of v_mfma_f32_16x16x32_bf16 on random
            # register operands, no memory traffic at all, measured on this pool (tools/probe/mfma_power.hip, profiles/r03_mfma_power.txt;
            # two boxes: 1842 and 1988 TFLOP/s at 1.91 / 2.04 GHz) -- a reference measurement, not part of this run
            "peak_at_power_cap": PEAK_BF16_AT_CAP_TFLOPS if (bf16 and not fp8) else None}


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same
    command (profiles/rNN_pmc_traffic.json, written by tools/pmc_traffic.py with the guide's gfx950 correction: FETCH_SIZE x 2);
    averaged over the kernel's instantiations, weighted by launches.  None if no pass has been committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    rows = [v for k, v in json.load(open(files[-1])).items() if k.startswith(kernel_prefix)]
    n = sum(r["launches"] for r in rows)
    return round(sum(r["hbm_bytes_per_launch"] * r["launches"] for r in rows) / n) if n else None


def cpu_baseline(image_size=256, batch=2, steps=8):
    """The oracle (oracle/cut_ref.py, checked against the reference) on the host cores: same step, fp32, B=2."""
    from oracle import cut_ref
    # the GPU box gives one job a share of the host (16 cores per GPU): never oversubscribe it
    ncores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    torch.set_num_threads(ncores)
    cut_ref.set_seed(42)
    gp, dp = cut_ref.init_generator(), cut_ref.init_discriminator()
    og, od = cut_ref.AdamState(gp), cut_ref.AdamState(dp)
    ema = {k: v.detach().clone() for k, v in gp.items()}
    cfg = cut_ref.default_config()
    g = torch.Generator().manual_seed(1234)
    photos = torch.rand(batch, 3, image_size, image_size, generator=g) * 2 - 1
    monets = torch.rand(batch, 3, image_size, image_size, generator=g) * 2 - 1
    times = []
    for step in range(1, steps + 2):   # starts at step 1 (non-R1), first one is warm-up
        rnd = cut_ref.sample_step_randomness(batch, image_size, image_size, generator=g)
        t0 = time.time()
        cut_ref.train_step(step, photos, monets, gp, dp, og, od, ema, cfg, rnd)
        times.append(time.time() - t0)
        print(f"[bench] cpu_baseline step {step}: {times[-1]:.2f} s on {ncores} threads", file=sys.stderr, flush=True)
    dt = sum(times[1:]) / len(times[1:])
    return {"value": round(batch / dt, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} timed steps of the same CUT step at {image_size}x{image_size}, batch {batch}, fp32, after 1 warm-up step"}


def launch_ranks(n: int) -> None:
    """`python bench.py --gpus N` without a launcher: start N fresh processes of this script, one per GPU, with the environment
    torch.distributed.run would give them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), relay rank 0's JSON line
    and fail if any rank fails.  This parent never touches the GPU, and nothing that has initialised the GPU is exec'ed."""
    import socket
    import subprocess
    import threading
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL's peer mappings need it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        time.sleep(0.2)
        failed = next((p for p in procs if p.poll() not in (None, 0)), None)
    failed = failed or next((p for p in procs if p.returncode != 0), None)
    if failed is not None:          # one rank died: the others would wait in a collective, stop exactly the processes started here
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=10)
    for ln in lines:                # rank 0 prints ONE JSON line; anything else it wrote goes to stderr
        (sys.stdout if ln.lstrip().startswith("{") else sys.stderr).write(ln)
    sys.stdout.flush()
    if failed is not None:
        raise SystemExit(f"[bench] rank {procs.index(failed)} exited with code {failed.returncode}")


class BoardPower(threading.Thread):
    """Board power and firmware-reported shader clock of THIS GPU while the timed steps run: hwmon `power1_input` / `freq1_input` read
    from sysfs every 20 ms by a host thread (no GPU call).  The MFMA kernels of the step run at the board's power limit (DESIGN 3.7:
    the convolution alone draws the 1400 W cap and holds ~2.0-2.1 of its 2.4 GHz), so the bench line carries the evidence."""

    def __init__(self, dev):
        super().__init__(daemon=True)
        self.node, self.halt, self.w, self.mhz = None, False, [], []
        try:
            pr = torch.cuda.get_device_properties(dev)
            want = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
            for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
                if os.path.basename(os.path.realpath(os.path.join(h, "..", ".."))).startswith(want) and os.path.exists(h + "/power1_input"):
                    self.node = h
        except Exception:
            self.node = None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except Exception:
            return None

    def run(self):
        while not self.halt and self.node:
            w, f = self._read(self.node + "/power1_input"), self._read(self.node + "/freq1_input")
            if w is not None:
                self.w.append(w * 1e-6)
            if f is not None:
                self.mhz.append(f * 1e-6)
            time.sleep(0.02)

    def report(self):
        if not self.node or not self.w:
            return None
        cap = self._read(self.node + "/power1_cap")
        return {"board_w": round(sum(self.w) / len(self.w), 1), "board_w_max": round(max(self.w), 1), "cap_w": round(cap * 1e-6, 1) if cap else None,
                "sclk_mhz": round(sum(self.mhz) / len(self.mhz), 0) if self.mhz else None, "samples": len(self.w),
                "source": "hwmon power1_input / freq1_input of this GPU, sampled every 20 ms during the timed steps"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--fp32", action="store_true", help="parity mode (exact fp32 MFMA) instead of bf16")
    ap.add_argument("--fp8", action="store_true", help="BASELINE.json configs[4]: e4m3 operand copies for the residual convolutions (forward, dgrad); "
                                                       "use with --size 512 --batch 8")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["cut", "basic"], default="cut",
                    help="cut: BASELINE.json configs[2] (the metric's config); basic: configs[1], Basic_GAN CycleGAN 64x64 batch 256")
    ap.add_argument("--launch-check", action="store_true",
                    help="no GPU work: every rank joins a gloo group on the host, rank 0 prints the rank count (tests the --gpus N launcher)")
    args = ap.parse_args()
    if args.workload == "basic":
        return main_basic(args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)       # plain `python bench.py --gpus N`: this process becomes the launcher (no GPU call before)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks (WORLD_SIZE={world})")
    if args.launch_check:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank)])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "max_rank": int(t.item())}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pg = None
    # the trainer's compute streams are created and used once BEFORE the process group exists (HipOps.bind_queues)
    from gan_variant_research_amd.runtime import HipOps
    ops = HipOps(dev)
    if not os.environ.get("GAN_NO_BIND_QUEUES"):
        ops.bind_queues()
    force_dist = bool(int(os.environ.get("GAN_FORCE_DIST", "0")))   # exercise the RCCL path even with one rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:                                       # GAN_FORCE_DIST without a launcher: a one-rank group
            for k, v in (("MASTER_PORT", "29531"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
                os.environ.setdefault(k, v)
        # RCCL prints its banner (host name, library path) on STDOUT when the communicator is created; the contract is ONE JSON
        # line on stdout, so file descriptor 1 points at stderr until the communicator exists
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI
            pg = dist.group.WORLD
            warm = torch.zeros(1, device=dev)
            dist.all_reduce(warm)                            # creates the communicator
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    from gan_variant_research_amd import cut as C
    cfg = default_config()
    C.set_seed(42)                                       # identical replicas on every rank
    gen, disc = C.build_models(cfg, dev)
    tr = C.CutTrainer(gen, disc, cfg, args.batch, args.size, device=dev, amp=not args.fp32, ops=ops, world_size=world, process_group=pg, fp8=args.fp8)
    if force_dist and not os.environ.get("GAN_FORCE_DIST_NOAR"):
        tr.force_allreduce = True
    g = torch.Generator().manual_seed(1234 + rank)
    photos = (torch.rand(args.batch, 3, args.size, args.size, generator=g) * 2 - 1).to(dev)
    monets = (torch.rand(args.batch, 3, args.size, args.size, generator=g) * 2 - 1).to(dev)
    aug_gen = torch.Generator().manual_seed(99 + rank)   # per-sample DiffAugment draws differ per rank
    nce_gen = torch.Generator().manual_seed(7)           # PatchNCE ids are shared by the whole (global) batch

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    step = 1   # starts at 1: the timed window contains R1 steps at multiples of 16 and the identity warm-up is active
    last = None
    for _ in range(args.warmup):
        last = tr.train_step(step, photos, monets, tr.sample_randomness(aug_gen, nce_gen))
        step += 1
    barrier()
    meter = BoardPower(dev) if (rank == 0 and not os.environ.get("GAN_NO_POWER_SAMPLER")) else None
    if meter is not None:
        meter.start()
    t0 = time.perf_counter()
    # sync="lag": each step's loss dict (and NaN check) is delivered one call later, so the read-back of step k overlaps the
    # queueing of step k+1; flush_losses() inside the timed region collects the last one -- all K dicts are produced in the window
    for _ in range(args.steps):
        r = tr.train_step(step, photos, monets, tr.sample_randomness(aug_gen, nce_gen), sync="lag")
        last = r if r is not None else last
        step += 1
    last = tr.flush_losses() or last
    barrier()
    dt = time.perf_counter() - t0
    if meter is not None:
        meter.halt = True
        meter.join()
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ips = args.batch * world * args.steps / dt
        peak = PEAK_F32_TFLOPS if args.fp32 else PEAK_BF16_TFLOPS
        out = {
            "metric": f"images/sec (G+D train step) {args.size}x{args.size} CUT", "value": round(ips, 3), "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.fp32 else ("fp8-e4m3 (residual convolutions: forward, dgrad) + bf16" if args.fp8 else "bf16"),
            "data": "synthetic",
            "config": {"workload": f"CUT ResNet-9 G + PatchGAN D + PatchNCE + identity + lazy R1 + DiffAugment, {args.size}x{args.size}, "
                                   f"batch {args.batch} per GPU (BASELINE.json configs[{4 if args.fp8 else 2}])", "global_batch": args.batch * world,
                       "parallelism": f"dp{world}"},
            "last_losses": last,
        }
        out["power"] = meter.report() if meter is not None else None
        out["roofline"] = dominant_kernel_roofline(tr)
        # the step's algorithmic FLOPs against the peak(s) they ran on: with --fp8 the e4m3 launches' share is priced at the fp8 peak
        share8 = out["roofline"]["share_of_step_conv_flop"] if args.fp8 else 0.0
        peak_mix = 1.0 / (share8 / PEAK_FP8_TFLOPS + (1.0 - share8) / peak)
        out["step_mfma_frac"] = round(ips / world * GFLOP_PER_IMAGE * (args.size / 256.0) ** 2 / 1e3 / peak_mix, 4)
        out["step_mfma_peak"] = round(peak_mix, 1)
        out["norm_hbm"] = getattr(tr, "norm_hbm", None)      # the HBM-bound kernels of the path, measured the same way
        print(f"[bench] gpu: {ips:.1f} images/s, {dt / args.steps * 1e3:.2f} ms/step; roofline {out['roofline']['achieved']} TFLOP/s", file=sys.stderr, flush=True)
        if not args.no_cpu_baseline and world == 1:     # the CPU baseline is reported at N=1 only
            out["cpu_baseline"] = cpu_baseline(args.size)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()

        dist.destroy_process_group()


def main_basic(args):
    """BASELINE.json configs[1]: Basic_GAN (CycleGAN) inner loop at 64x64, bf16, batch 256 on one GPU (a parity / secondary
    bench line; the headline metric is the CUT line)."""
    from gan_variant_research_amd import basic as BG
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    B = 256 if args.batch == 16 else args.batch
    S = 64 if args.size == 256 else args.size
    cfg = {"training": {"amp": not args.fp32, "seed": 0}, "optim": {"lr_g": 2e-4, "lr_d": 2e-4, "betas": [0.5, 0.999]},
           "loss": {"gan": "lsgan", "lambda_cycle": 10.0, "lambda_identity": 0.5},
           "model": {"ngf": 64, "ndf": 64, "n_blocks": 9, "spectral_norm_d": False}}
    torch.manual_seed(0)
    tr = BG.CycleGANTrainer(*BG.build_models(cfg, dev), cfg, B, S, device=dev, amp=not args.fp32)
    g = torch.Generator().manual_seed(1234)
    a = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).to(dev)
    b = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).to(dev)
    last = None
    for _ in range(args.warmup):
        last = tr.train_iteration(a, b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = tr.train_iteration(a, b)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"metric": "images/sec (G+D train step) Basic_GAN CycleGAN 64x64", "value": round(B * args.steps / dt, 3), "unit": "images/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32" if args.fp32 else "bf16", "data": "synthetic",
           "config": {"workload": f"Basic_GAN CycleGAN (2 ResNet-9 G + 2 PatchGAN D with InstanceNorm), {S}x{S}, batch {B} (BASELINE.json configs[1])",
                      "global_batch": B, "parallelism": "dp1"}, "last_losses": last}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
