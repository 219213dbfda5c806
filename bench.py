#!/usr/bin/env python3
"""Headline benchmark: joint VAE+seg train-step throughput (volumes/s) on synthetic 96^3 volumes, batch 2 per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (one rank per GPU), or as
plain `python bench.py --gpus N`, in which case this process spawns that launcher itself — before it touches the GPU — and
relays rank 0's single JSON line.

A step = zero_grad -> Segmentation fwd -> frozen VAE fwd -> Dice losses -> backward -> [RCCL all-reduces, overlapped with the
weight-gradient kernels] -> SGD(momentum)  (the `joint_train` method of the reference, main_source.py:449-471,660-661),
bf16 activations / fp32 accumulate, inputs resident in HBM, HIP-graph replay.  Rank 0 prints ONE JSON line:
  value/ms_per_step  whole-job volumes/s over all ranks (barrier + synchronize on both sides, max over ranks)
  roofline           the WHOLE STEP against its binding roofline: algorithmic bytes per step (1.28 GB per volume,
                     every activation touched once per pass) / ms_per_step against HBM peak — recomputable from this line alone;
                     `families` lists the per-kernel-family figures from live HIP-event timing (or --no-families)
  fp32_parity_mode   the same step with the fp32 kernels (the mode that meets the 1e-3 parity gate), N=1 only
  cpu_baseline       the same step on the host cores with the CPU oracle (plain eager PyTorch fp32), bounded sample, N=1 only.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SIDE, BATCH, DIM = 96, 2, 128
# algorithmic work per volume per joint_train step at 96^3: 3 Seg passes + 2 VAE passes
FLOPS_PER_VOLUME = 164.0e9
BYTES_PER_VOLUME = {"bf16": 1.28e9, "fp16": 1.28e9, "fp32": 2.56e9}
# --config: the workload.  joint96 = BASELINE configs[1], the configuration the metric is quoted on and the DEFAULT (what the driver times);
# da128 = configs[3] (128^3 teacher-student domain_adaptation step, batch 1), joint160 = one GPU's share of configs[4] (160^3 joint_train, batch 2,
# fp16 storage + dynamic loss scaling).  Algorithmic FLOPs / fused-minimum bytes per volume per step: BASELINE.md section 3.
CONFIGS = {
    "joint96": {"side": 96, "batch": 2, "method": "joint_train", "dtype": "bf16", "flops": 164.0e9, "bytes16": 1.28e9,
                "name": "configs[1]: %d^3 joint VAE+seg training step (joint_train), batch=%d/GPU"},
    "da128": {"side": 128, "batch": 1, "method": "domain_adaptation", "dtype": "bf16", "flops": 545.3e9, "bytes16": 4.2e9,
              "name": "configs[3]: %d^3 teacher-student domain-adaptation step (domain_adaptation, loss type 0: student Seg+VAE forward/backward, teacher "
                      "Seg+VAE forward, pseudo-label, three Dice terms), batch=%d/GPU"},
    "joint160": {"side": 160, "batch": 2, "method": "joint_train", "dtype": "fp16", "flops": 759.3e9, "bytes16": 5.9e9,
                 "name": "configs[4], one GPU's share: %d^3 joint VAE+seg training step (joint_train), batch=%d/GPU"},
}
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# fp32: the parity mode runs its 3x3x3 convolutions and their weight gradients (96.7 % of the FLOPs) on the bf16 matrix cores, six exact limb
# products per fp32 product (csrc/igemm_k3x.h): its matrix peak is the bf16 dense peak / 6 = 416.7 TFLOP/s of fp32-equivalent work
# (the exact-f32 MFMA it replaces peaks at 157.3)
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 2500.0 / 6.0}
EXACT_F32_MFMA_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 — stated beside the limb peak wherever the fp32 mode's fraction is quoted


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="joint96", choices=sorted(CONFIGS), help="workload: joint96 = BASELINE configs[1] (default, the metric's "
                    "configuration), da128 = configs[3], joint160 = one GPU's share of configs[4]")
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp16", "fp32"], help="storage type of activations / packed weights (default: the config's)")
    ap.add_argument("--side", type=int, default=None, help="volume side (default: the config's; another value is a plumbing run without a roofline)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32-mode", action="store_true", help="skip the fp32 (parity-mode) timing entry")
    ap.add_argument("--no-families", action="store_true", help="skip the live per-kernel-family timing pass")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--detail", action="store_true", help="print the slowest individual launches (stderr)")
    ap.add_argument("--dump-launches", default=None, help="write every live-timed launch (kernel, detail, us, bytes, flops) to this JSON file")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to exercise the N>1 path on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="testing aid: every rank uses cuda:0")
    ap.add_argument("--force-dist", action="store_true", help="testing aid: initialise the process group and run the gradient "
                    "all-reduce path even with one rank (exercises RCCL on a 1-GPU box)")
    ap.add_argument("--master-port", type=int, default=29533)
    ap.add_argument("--recompute", action="store_true", help="activation recomputation in the Down / Up blocks (joint_model.set_recompute, DESIGN 4.4)")
    ap.add_argument("--no-exchange-forms", action="store_true", help="N > 1: skip the no-exchange / other-exchange-form timing legs")
    ap.add_argument("--other-form", action="store_true", help="N > 1: after the no-exchange leg, also time the OTHER exchange form (two buckets, bucket 0 all-reduced beside "
                    "the remaining weight-gradient nodes of the same graph) for the record.  Opt-in since the end of round 6: the form has never run on more than one "
                    "RCCL rank, and rank 0's line must not depend on a leg that is not the measurement")
    ap.add_argument("--no-other-form", action="store_true", help="(default since round 6; kept for older command lines)")
    ap.add_argument("--no-other-configs", action="store_true", help="N = 1, default workload: skip the configs[3] / configs[4] entries (`other_configs`)")
    ap.add_argument("--legs-budget-s", type=float, default=60.0, help="N > 1: the other exchange form is timed only if everything before it "
                    "(set-up, main timing, the no-exchange leg) took less than this on every rank: rank 0's line must not wait for it")
    a = ap.parse_args()
    cfg = CONFIGS[a.config]
    a.dtype = a.dtype or cfg["dtype"]
    a.side = a.side or cfg["side"]
    a.batch = cfg["batch"]
    return a


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a CHILD process (this process has not
    initialised the GPU and never will) and relay its output; rank 0 of the child job prints the JSON line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(a.master_port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def build(side, dtype, rank, batch=BATCH, teacher=False):
    """-> (joint, img, label[, teacher]): Joint(Segmentation, frozen VAE(spatial=side)) with the RNG-free weight fill, synthetic inputs resident in HBM;
    teacher: a second, fully frozen Joint whose Seg has another fill (main_target.py:397-406: the teacher starts as a copy and drifts by EMA)."""
    import torch
    import joint_model as M
    from vae_segmentation_amd import synthetic as O      # the package's own RNG-free weight fill / synthetic volumes (the oracle is imported by cpu_baseline only)
    kd = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[dtype]

    def make(seg_seed=None, frozen=False):
        seg = M.Segmentation(n_channels=1, n_class=2, norm_type=1)
        vae = M.VAE(n_channels=2, n_class=2, norm_type=1, dim=DIM, spatial=side)
        joint = M.Joint(models=[seg, vae])
        O.deterministic_fill_(joint, seed=0)
        if seg_seed is not None:
            O.deterministic_fill_(joint.Seg, seed=seg_seed)
        joint = joint.cuda()
        for p in (joint.parameters() if frozen else joint.Vae.parameters()):
            p.requires_grad = False
        joint.Vae.eval()
        return M.set_kernel_dtype(joint, kd)

    joint = make()
    img = O.synthetic_image(batch, side, seed=2 + 10 * rank).cuda()
    lab = O.synthetic_label(batch, side, seed=3 + 10 * rank).cuda()
    if teacher:
        return joint, img, lab, make(seg_seed=1, frozen=True)
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


def cpu_baseline(side, steps, budget_s=40.0, batch=BATCH, method="joint_train"):
    """The oracle's step of the same workload (stock eager PyTorch fp32) on the host cores; 1 warm-up + `steps` timed."""
    import torch
    from oracle import ref_cpu as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    joint = O.build_joint(side)
    teacher = None
    if method == "domain_adaptation":
        teacher = O.build_joint(side)
        O.deterministic_fill_(teacher.Seg, seed=1)
        for p in teacher.parameters():
            p.requires_grad = False
    img, lab = O.synthetic_image(batch, side, 2), O.synthetic_label(batch, side, 3)
    opt = torch.optim.SGD(joint.Seg.parameters(), lr=1e-2, momentum=0.9)
    times, t_start = [], time.perf_counter()
    for i in range(steps + 1):
        t0 = time.perf_counter()
        opt.zero_grad()
        if teacher is not None:
            loss, _ = O.domain_adaptation_losses(joint, teacher, img, lab, lambda_vae=1.0, domain_loss_type=0)
        else:
            loss, _ = O.joint_train_losses(joint, img, lab)
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget_s and i >= 1:      # bounded sample: stop once the budget is spent
            break
    timed = sorted(times[1:]) if len(times) > 1 else times
    med = timed[len(timed) // 2]
    return {"value": batch / med, "unit": "volumes/s", "cores": cores, "kind": "port",
            # the spread of the timed steps (the figure wobbles 0.76 - 1.2 volumes/s from box to box and run to run: host cores shared with other tenants)
            "value_min": batch / timed[-1], "value_max": batch / timed[0], "seconds_per_step": [round(t, 3) for t in times[1:]],
            "sample": "1 warm-up + %d timed %s steps at %d^3 B=%d, eager PyTorch fp32 (oracle/ref_cpu.py), "
                      "median, %d threads" % (len(timed), method, side, batch, cores)}


def make_step(a, dtype, rank, use_dist, overlap=None, info=None):
    """-> (step(), loss_fn, seg_params, closer): one train step of configs[1] in the given kernel dtype.
    overlap: None = the default exchange form (ddp.FlatGradSync: one bucket unless VS_DDP_OVERLAP=1), True / False forces the form.
    info (dict, optional): receives what was built ("tail_in_graph", "buckets")."""
    from vae_segmentation_amd import ddp, optim
    from vae_segmentation_amd import train as T
    method = CONFIGS[a.config]["method"]
    teacher = None
    if method == "domain_adaptation":
        joint, img, lab, teacher = build(a.side, dtype, rank, a.batch, teacher=True)
    else:
        joint, img, lab = build(a.side, dtype, rank, a.batch)
    opt = optim.SGD([{"params": joint.Seg.parameters(), "lr": 1e-2}, {"params": joint.Vae.parameters(), "lr": 0.0}],
                    lr=1e-2, momentum=0.9, weight_decay=0.0)
    seg_params = [p for p in joint.Seg.parameters()]
    scaler = optim.LossScaler() if dtype == "fp16" else None          # fp16 storage: dynamic loss scale, on the device
    kw = {} if scaler is None else {"scaler": scaler}
    sync = ddp.FlatGradSync(seg_params, overlap=overlap) if use_dist else None
    if sync is not None:
        sync.broadcast_parameters(0)
    if info is not None:
        info["buckets"] = len(sync.buckets) if sync is not None else 0
        info["tail_in_graph"] = False

    def loss_fn():
        if teacher is not None:          # main_target.py:520-596, loss type 0, lambda_vae 1.0 (scripts/target/domain_msd_dh.bash:12-13); the schedule stays on the device
            return T.domain_adaptation_losses(joint, teacher, img, lab, lambda_vae=1.0, domain_loss_type=0, host_schedule=False)
        return T.joint_train_losses(joint, img, lab, lambda_vae=0.1)

    if a.no_graph:
        def step():
            for p in seg_params:
                p.grad = None
            loss, _ = loss_fn()
            if scaler is not None:
                loss.backward(gradient=scaler.seed)
            else:
                loss.backward()
            if sync is not None:
                sync()
                opt.step_with(*sync.live(), **kw)
            else:
                opt.step(**kw)
            return loss
    else:
        gs = T.GraphedStep(loss_fn, seg_params, opt, grad_sync=sync, warmup=2, scaler=scaler)
        step = gs.step
        if info is not None:
            info["tail_in_graph"] = bool(gs.tail)

    def closer():
        """before the eager family-timing passes: gradients back to ordinary tensors, and the captured step's autograd graph and gradient
        accumulators (created on its warm-up stream) released — eager passes on the default stream would otherwise reuse them and pay
        autograd's cross-stream synchronisation on every parameter"""
        if sync is not None:
            sync.close()
        if not a.no_graph:
            if gs.grad_sync is not None:
                gs.grad_sync.close()              # the flat buffer GraphedStep made for its captured optimiser launch
            gs.loss = gs.aux = gs._accumulators = None
    return step, loss_fn, seg_params, closer


HOST_ISSUE = {"s": 0.0}


def timed_steps(step, steps, warmup, fence):
    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    HOST_ISSUE["s"] = time.perf_counter() - t0      # host time to ISSUE the steps (graph launch + the eager exchange / optimiser launches), before the fence
    fence()
    return time.perf_counter() - t0, loss


def step_roofline(dtype, ms_per_step, families, config="joint96"):
    """Whole-step roofline: the step is HBM-bound by the algorithm (2.56 GB -> 0.32 ms at 8 TB/s against
    328 GF -> 0.13 ms at the bf16 MFMA peak), so `achieved` = algorithmic bytes per step / measured step time."""
    from vae_segmentation_amd import profiling
    cfg = CONFIGS[config]
    nbytes = cfg["bytes16"] * (2.0 if dtype == "fp32" else 1.0) * cfg["batch"]
    flops = cfg["flops"] * cfg["batch"]
    sec = ms_per_step * 1e-3
    t_hbm, t_mfma = nbytes / (HBM_PEAK_GBS * 1e9), flops / (MFMA_PEAK_TFLOPS[dtype] * 1e12)
    out = {"scope": "whole step (one HIP-graph replay + optimiser launches)",
           "algorithmic_bytes_per_step": nbytes, "algorithmic_flops_per_step": flops,
           "mfma": dict({"achieved": flops / sec / 1e12, "peak": MFMA_PEAK_TFLOPS[dtype], "unit": "TFLOP/s", "frac": flops / sec / 1e12 / MFMA_PEAK_TFLOPS[dtype]},
                        **({"peak_note": "bf16 dense peak / 6 (six limb products per fp32 product); the exact-f32 MFMA peak is %.1f TFLOP/s: frac_of_exact_f32_peak"
                                         % EXACT_F32_MFMA_PEAK_TFLOPS, "frac_of_exact_f32_peak": flops / sec / 1e12 / EXACT_F32_MFMA_PEAK_TFLOPS} if dtype == "fp32" else {})),
           "hbm": {"achieved": nbytes / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / sec / 1e9 / HBM_PEAK_GBS}}
    lead = "hbm" if t_hbm >= t_mfma else "mfma"
    out.update({"bound": lead, "achieved": out[lead]["achieved"], "peak": out[lead]["peak"], "unit": out[lead]["unit"], "frac": out[lead]["frac"],
                # `traffic` comes from the newest committed PMC summary (profiles/rNN_hbm_traffic.json: the joint96 bf16 step), not from this run:
                # traffic_stale true means that summary was taken on other kernel sources than the ones that just ran
                "traffic": profiling.measured_step_traffic() if (config == "joint96" and dtype == "bf16") else None,
                "traffic_stale": profiling.traffic_is_stale() if (config == "joint96" and dtype == "bf16") else None})
    if families is not None:
        out["families"] = families
    return out


def other_configs(a, rank, torch):
    """configs[3] (da128) and one GPU's share of configs[4] (joint160, fp16 storage + dynamic loss scaling) timed in the SAME run as the
    headline workload, with the same protocol (HIP-graph replay, inputs resident, synchronise on both sides) on a short leg — 3 warm-up +
    10 timed steps each — and the same whole-step roofline object, so that the driver's record carries a number for every single-GPU
    configuration of BASELINE.json."""
    import copy
    import joint_model
    out = []
    # configs[4] is worded "160^3 fp16 ... WITH activation checkpointing": timed both ways (round 6) — recomputation off is how the step runs by default
    # (3.5 GB of 288 GB: DESIGN.md section 4.4), recomputation on is the configuration as BASELINE.json names it
    for name, recompute in (("da128", False), ("joint160", False), ("joint160", True)):
        b = copy.copy(a)
        cfg = CONFIGS[name]
        b.config, b.side, b.batch, b.dtype = name, cfg["side"], cfg["batch"], cfg["dtype"]
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        info = {}
        joint_model.set_recompute(recompute)
        try:
            step, _, _, closer = make_step(b, b.dtype, rank, False, info=info)
            n, w = 10, 3
            dt, loss = timed_steps(step, n, w, lambda: torch.cuda.synchronize())
        finally:
            joint_model.set_recompute(False)
        ms = 1e3 * dt / n
        out.append({"config": name, "workload": (cfg["name"] % (b.side, b.batch)) + ", %s activations + fp32 accumulate%s, %s, SGD momentum 0.9, VAE frozen, "
                    "HIP-graph replay" % (b.dtype, " + dynamic loss scaling" if b.dtype == "fp16" else "",
                                          "WITH activation recomputation in every Down / Up block (\"checkpointing\")" if recompute else "activations kept (no recomputation)"),
                    "activation_recomputation": recompute,
                    "dtype": {"bf16": "bf16", "fp16": "f16", "fp32": "f32"}[b.dtype], "value": b.batch * n / dt, "unit": "volumes/s",
                    "ms_per_step": ms, "steps": n, "warmup": w, "final_loss": float(loss.item()), "tail_in_graph": info.get("tail_in_graph", False),
                    "peak_device_memory_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2),
                    "roofline": step_roofline(b.dtype, ms, None, name)})
        closer()
        del step, loss
    torch.cuda.empty_cache()
    return out


def main():
    t_process = time.perf_counter()
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and a.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(a))        # nothing in this process has touched the GPU
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # this pool's driver supports dmabuf IPC only: RCCL needs it on every rank, whoever launched us (before the HIP runtime loads)
    # fd 1 carries the JSON line and nothing else: native libraries (RCCL prints a version banner to stdout at init) get fd 2
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if a.share_gpu:
        from vae_segmentation_amd import ops as _ops0
        _ops0.device_is_shared(True)            # several ranks on one card: no in-kernel hand-offs (they need the device to themselves: ops.device_is_shared)
    if a.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    use_dist = world > 1 or a.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(a.master_port))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(a.backend)

    from vae_segmentation_amd import profiling

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # N > 1 self-check: the ranks the launcher promised are the ranks the collective library really connected (sum of ones over the job)
    nranks_seen = 1
    if use_dist:
        one = torch.ones(1, device="cuda" if a.backend == "nccl" else "cpu")
        dist.all_reduce(one)
        nranks_seen = int(round(float(one.item())))
        if nranks_seen != dist.get_world_size() or dist.get_world_size() != world:
            raise SystemExit("process group has %d ranks, the collective summed %d, WORLD_SIZE=%d" % (dist.get_world_size(), nranks_seen, world))

    def max_over_ranks(sec):
        if not use_dist:
            return sec
        t = torch.tensor([sec], dtype=torch.float64, device="cuda" if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if a.recompute:
        import joint_model
        joint_model.set_recompute(True)
    info = {}
    step, loss_fn, seg_params, closer = make_step(a, a.dtype, rank, use_dist, info=info)
    dt, loss = timed_steps(step, a.steps, a.warmup, fence)
    dt = max_over_ranks(dt)
    final_loss = float(loss.item())
    host_issue_ms = 1e3 * HOST_ISSUE["s"] / a.steps
    del loss                                    # the last reference to the captured step's autograd graph (see closer())
    ms_per_step = 1e3 * dt / a.steps

    # ---- the exchange, measured in the same run (every rank takes part: these are collective) ----------------------------------------
    # exposed cost of the gradient exchange = this job's step minus the same step without the exchange (replicas drift apart in that leg: it is
    # timed, not trained); and the other exchange form (two buckets, bucket 0 all-reduced under the remaining weight-gradient kernels), so
    # that ONE run decides which form should be the default at this world size.
    exchange = None
    if use_dist and not a.no_graph and not a.no_exchange_forms:
        n_x = max(10, min(a.steps, 50))
        closer()
        del step, loss_fn, seg_params
        torch.cuda.empty_cache()
        info0, info1 = {}, {}
        step_n, _, _, closer_n = make_step(a, a.dtype, rank, False, info=info0)
        dt_n, _ = timed_steps(step_n, n_x, 3, fence)
        dt_n = max_over_ranks(dt_n)
        closer_n()
        del step_n
        torch.cuda.empty_cache()
        exchange = {"steps": n_x, "no_exchange_ms_per_step": round(1e3 * dt_n / n_x, 4),
                    "exposed_exchange_ms_per_step": round(ms_per_step - 1e3 * dt_n / n_x, 4),
                    "default_form": {"buckets": info.get("buckets"), "tail_in_graph": info.get("tail_in_graph"), "ms_per_step": round(ms_per_step, 4)},
                    "other_form": None}
        if rank == 0:       # the measured line so far, for the record, should the last leg never return (nothing parses stderr)
            print("bench.py: default exchange form %.4f ms per step, without exchange %.4f; timing the other form now"
                  % (ms_per_step, 1e3 * dt_n / n_x), file=sys.stderr, flush=True)
        other = not (os.environ.get("VS_DDP_OVERLAP", "0") == "1")
        # every rank takes the same branch (the legs are collective): the slowest rank's clock decides
        spent = max_over_ranks(time.perf_counter() - t_process)
        exchange["seconds_before_other_form"] = round(spent, 1)
        if not a.other_form or a.no_other_form or spent > a.legs_budget_s:
            if a.other_form and not a.no_other_form:
                exchange["other_form_skipped"] = "%.0f s spent before it (budget %.0f s): the line is printed instead" % (spent, a.legs_budget_s)
            # the step rebuilt here only feeds rank 0's eager family-timing pass (loss_fn, seg_params): without the exchange — no further collective is captured
            step, loss_fn, seg_params, closer = make_step(a, a.dtype, rank, False, info=info1)
        else:
            step, loss_fn, seg_params, closer = make_step(a, a.dtype, rank, True, overlap=other, info=info1)
            dt_o, _ = timed_steps(step, n_x, 3, fence)
            dt_o = max_over_ranks(dt_o)
            exchange["other_form"] = {"buckets": info1.get("buckets"), "overlap": other, "tail_in_graph": info1.get("tail_in_graph"),
                                      "ms_per_step": round(1e3 * dt_o / n_x, 4)}

    families, fp32_mode, cpu, others = None, None, None, None
    if rank == 0 and not a.no_families:
        # per-kernel-family timing: HIP events around every launch inside real (eager) forward+backward passes
        closer()                                 # gradients back to ordinary tensors for the eager profiling pass
        families = profiling.kernel_families(lambda: (loss_fn()[0]).backward(), seg_params, a.dtype, steps=2)
        if a.dump_launches:
            with open(a.dump_launches, "w") as f:
                json.dump([{"us": round(ms * 1e3, 3), "kernel": kid, "detail": det, "bytes": nb, "flops": fl}
                           for ms, kid, det, nb, fl in profiling.LAST_LAUNCHES], f)
        if a.detail:
            tot = sum(r[0] for r in profiling.LAST_LAUNCHES) / 2
            print("timed launches: %.3f ms per step over %d launches" % (tot, len(profiling.LAST_LAUNCHES) // 2), file=sys.stderr)
            for ms, kid, det, nb, fl in profiling.LAST_LAUNCHES[:60]:
                print("%8.1f us  %-42s %-52s %7.1f GB/s %8.2f TF/s" % (ms * 1e3, kid, det, nb / ms / 1e6, fl / ms / 1e9), file=sys.stderr)
    if use_dist:
        dist.barrier()
    if rank == 0 and world == 1 and not a.no_fp32_mode and a.dtype != "fp32":
        del step, loss_fn, seg_params
        torch.cuda.empty_cache()
        step32, _, _, closer32 = make_step(a, "fp32", rank, False)
        n32 = max(5, min(a.steps, 10))
        dt32, _ = timed_steps(step32, n32, 2, lambda: torch.cuda.synchronize())
        fp32_mode = {"ms_per_step": 1e3 * dt32 / n32, "value": a.batch * n32 / dt32, "unit": "volumes/s", "steps": n32,
                     "tflops": CONFIGS[a.config]["flops"] * a.batch * n32 / dt32 / 1e12, "limb_peak_tflops": MFMA_PEAK_TFLOPS["fp32"],
                     "exact_f32_mfma_peak_tflops": EXACT_F32_MFMA_PEAK_TFLOPS,
                     "note": "fp32 storage; 3x3x3 convolutions and their weight gradients on the bf16 matrix cores through three-limb operand splitting (six exact "
                             "limb products per product, fp32 accumulation: csrc/igemm_k3x.h) - the mode that meets the 1e-3 parity gate (tests/test_gpu_model.py); "
                             "round 3 ran it on the exact-f32 MFMA at 9.83 ms"}
        closer32()
        del step32
    if rank == 0 and world == 1 and not use_dist and a.config == "joint96" and a.side == CONFIGS["joint96"]["side"] and not a.no_other_configs \
            and not a.no_graph and not a.recompute:
        peak_main = round(torch.cuda.max_memory_allocated() / 1e9, 2)
        others = other_configs(a, rank, torch)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a.side, a.cpu_steps, batch=a.batch, method=CONFIGS[a.config]["method"])

    from vae_segmentation_amd import ops as _ops
    _ops.chain_fault()                       # a chain / epilogue-apply kernel that gave up a bounded wait (csrc/chain.h) would have produced invalid steps: refuse to report them
    if rank == 0:
        vols = world * a.batch * a.steps / dt
        cfg = CONFIGS[a.config]
        out = {
            "metric": ("3D train-step volumes/sec at 96^3 batch=2 (joint VAE+seg)" if a.config == "joint96" else
                       "3D train-step volumes/sec at %d^3 batch=%d (%s)" % (a.side, a.batch, cfg["method"])), "value": vols, "unit": "volumes/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "fp16": "f16", "fp32": "f32"}[a.dtype], "data": "synthetic",
            "config": {"workload": (cfg["name"] % (a.side, a.batch)) + ", %s activations + fp32 accumulate%s, SGD momentum 0.9, VAE frozen, HIP-graph replay"
                                   % (a.dtype, " + dynamic loss scaling" if a.dtype == "fp16" else ""),
                       "global_batch": world * a.batch, "parallelism": "dp%d" % world, "final_loss": final_loss,
                       "activation_recomputation": bool(a.recompute),
                       "peak_device_memory_GB": peak_main if others is not None else round(torch.cuda.max_memory_allocated() / 1e9, 2),
                       # host time per step to issue the work (one graph launch, then the exchange and the optimiser eagerly); far below ms_per_step = the
                       # step is GPU-bound and capturing those tail launches into the graph as well would not shorten it
                       "host_issue_ms_per_step": round(host_issue_ms, 4),
                       # the whole step — pass, exchange, optimiser, weight re-pack — is ONE replayed HIP graph when true (train.GraphedStep, captured tail)
                       "tail_in_graph": info.get("tail_in_graph", False),
                       "nranks": nranks_seen, "exchange": exchange,
                       "grad_exchange": (("2-bucket RCCL all-reduce, bucket 0 under the full-resolution weight-gradient kernels"
                                          if os.environ.get("VS_DDP_OVERLAP", "0") == "1" else
                                          "one RCCL all-reduce of the flat gradient buffer (written in place by the weight-gradient kernels) after the pass")
                                         if use_dist else "none (1 rank)")},
            "roofline": step_roofline(a.dtype, ms_per_step / 1.0, families, a.config) if a.side == cfg["side"] else None,
            "fp32_parity_mode": fp32_mode, "cpu_baseline": cpu,
            # the other single-GPU configurations of BASELINE.json, timed in this run on short legs (3 warm-up + 10 steps): configs[3], configs[4]'s share
            "other_configs": others,
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
