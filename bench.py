#!/usr/bin/env python3
"""Headline benchmark: joint VAE+seg train-step throughput (volumes/s) on synthetic 96^3 volumes, batch 2 per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = zero_grad -> Segmentation fwd -> frozen VAE fwd -> Dice losses -> backward -> [RCCL all-reduce] -> SGD(momentum)
(the `joint_train` method of the reference, main_source.py:449-471,660-661), bf16 activations / fp32 accumulate,
inputs resident in HBM.  Rank 0 prints ONE JSON line with the whole-job volumes/s, plus
  roofline      live HIP-event timing of the dominant kernel's launches inside real steps vs its roofline bound
  cpu_baseline  the same step on the host cores with the CPU oracle (plain eager PyTorch fp32), bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

SIDE, BATCH, DIM = 96, 2, 128
# algorithmic work per volume per joint_train step at 96^3: 3 Seg passes + 2 VAE passes
FLOPS_PER_VOLUME = 164.0e9
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--side", type=int, default=SIDE)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--detail", action="store_true", help="print the slowest individual launches (stderr)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to exercise the N>1 path on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="testing aid: every rank uses cuda:0")
    ap.add_argument("--force-dist", action="store_true", help="testing aid: initialise the process group and run the gradient "
                    "all-reduce path even with one rank (exercises RCCL on a 1-GPU box)")
    return ap.parse_args()


def build(side, dtype, rank):
    import joint_model as M
    from oracle import ref_cpu as O      # only for the RNG-free weight fill / synthetic inputs shared with the tests
    seg = M.Segmentation(n_channels=1, n_class=2, norm_type=1)
    vae = M.VAE(n_channels=2, n_class=2, norm_type=1, dim=DIM, spatial=side)
    joint = M.Joint(models=[seg, vae])
    O.deterministic_fill_(joint, seed=0)
    joint = joint.cuda()
    for p in joint.Vae.parameters():
        p.requires_grad = False
    joint.Vae.eval()
    M.set_kernel_dtype(joint, torch.bfloat16 if dtype == "bf16" else torch.float32)
    img = O.synthetic_image(BATCH, side, seed=2 + 10 * rank).cuda()
    lab = O.synthetic_label(BATCH, side, seed=3 + 10 * rank).cuda()
    return joint, img, lab


def usable_cores():
    """CPU threads this process may really use: affinity mask, capped by the cgroup CPU quota when one is set
    (256 runnable threads under a small quota ran the oracle 30x slower than 8 threads did)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return max(1, min(n, 64))        # eager conv3d on CPU stops scaling well before 64 threads


def cpu_baseline(side, steps, budget_s=40.0):
    """The oracle's joint_train step (stock eager PyTorch fp32) on the host cores; 1 warm-up + `steps` timed."""
    from oracle import ref_cpu as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    joint = O.build_joint(side)
    img, lab = O.synthetic_image(BATCH, side, 2), O.synthetic_label(BATCH, side, 3)
    opt = torch.optim.SGD(joint.Seg.parameters(), lr=1e-2, momentum=0.9)
    times, t_start = [], time.perf_counter()
    for i in range(steps + 1):
        t0 = time.perf_counter()
        opt.zero_grad()
        loss, _ = O.joint_train_losses(joint, img, lab)
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget_s and i >= 1:      # bounded sample: stop once the budget is spent
            break
    timed = sorted(times[1:]) if len(times) > 1 else times
    med = timed[len(timed) // 2]
    return {"value": BATCH / med, "unit": "volumes/s", "cores": cores, "kind": "port",
            "sample": "1 warm-up + %d timed joint_train steps at %d^3 B=%d, eager PyTorch fp32 (oracle/ref_cpu.py), "
                      "median, %d threads" % (len(timed), side, BATCH, cores)}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (a.gpus, a.gpus))
    if a.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    use_dist = world > 1 or a.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(a.backend)

    from vae_segmentation_amd import ddp, optim, profiling
    from vae_segmentation_amd import train as T

    joint, img, lab = build(a.side, a.dtype, rank)
    opt = optim.SGD([{"params": joint.Seg.parameters(), "lr": 1e-2}, {"params": joint.Vae.parameters(), "lr": 0.0}],
                    lr=1e-2, momentum=0.9, weight_decay=0.0)
    seg_params = [p for p in joint.Seg.parameters()]
    sync = ddp.FlatGradSync(seg_params) if use_dist else None
    if sync is not None:
        sync.broadcast_parameters(0)

    def loss_fn():
        return T.joint_train_losses(joint, img, lab, lambda_vae=0.1)

    if a.no_graph:
        from vae_segmentation_amd import ops as _ops
        _ops.set_overlap(False)

        def step():
            for p in seg_params:
                p.grad = None
            loss, _ = loss_fn()
            loss.backward()
            if sync is not None:
                opt.step_with(seg_params, sync())
            else:
                opt.step()
            return loss
    else:
        gs = T.GraphedStep(loss_fn, seg_params, opt, warmup=2)
        graph_grads = [p.grad for p in seg_params]

        def step():
            gs.graph.replay()
            if sync is not None:
                opt.step_with(seg_params, sync(graph_grads))
            else:
                opt.step()
            return gs.loss

    for _ in range(a.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())

    roof, cpu = None, None
    if rank == 0:
        # dominant-kernel timing: HIP events around every launch of that kernel inside real (eager) steps
        roof = profiling.dominant_kernel_roofline(lambda: (loss_fn()[0]).backward(), seg_params, a.dtype, steps=2)
        if a.detail:
            tot = sum(r[0] for r in profiling.LAST_LAUNCHES) / 2
            print("timed launches: %.3f ms per step over %d launches" % (tot, len(profiling.LAST_LAUNCHES) // 2), file=sys.stderr)
            for ms, kid, det, nb, fl in profiling.LAST_LAUNCHES[:60]:
                print("%8.1f us  %-42s %-52s %7.1f GB/s %8.2f TF/s" % (ms * 1e3, kid, det, nb / ms / 1e6, fl / ms / 1e9), file=sys.stderr)
    if use_dist:
        dist.barrier()
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a.side, a.cpu_steps)

    if rank == 0:
        vols = world * BATCH * a.steps / dt
        out = {
            "metric": "3D train-step volumes/sec at 96^3 batch=2 (joint VAE+seg)", "value": vols, "unit": "volumes/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if a.dtype == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: %d^3 joint VAE+seg training step (joint_train), batch=%d/GPU, %s activations + fp32 accumulate, "
                                   "SGD momentum 0.9, VAE frozen, HIP-graph replay" % (a.side, BATCH, a.dtype),
                       "global_batch": world * BATCH, "parallelism": "dp%d" % world, "final_loss": final_loss,
                       "step_flops_fraction_of_mfma_peak": FLOPS_PER_VOLUME * BATCH * a.steps / dt / (MFMA_PEAK_TFLOPS["bf16" if a.dtype == "bf16" else "f32"] * 1e12)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
