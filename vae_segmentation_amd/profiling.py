"""Live per-kernel timing for bench.py's roofline leg: HIP events (torch.cuda.Event on the launch stream) around
every launch of each libvaeseg kernel inside real train steps, summed per kernel instantiation."""
import torch

from . import ops

LAST_LAUNCHES = []
HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec); ~6.3 TB/s achievable
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "f32": 2500.0 / 6.0}      # f32: six bf16 limb products per fp32 product (csrc/igemm_k3x.h)


LAST_EVENT_OVERHEAD_US = 0.0


def _spin(us, stream):
    """tools/probe/libvsprobe.so: vs_spin — a measurement aid, deliberately not an entry point of libvaeseg.so"""
    from tools import probe
    probe.check(probe.lib.vs_spin(int(us), stream), "spin")


ops.PROFILE_SPIN[0] = _spin


def event_bracket_overhead_us(trials=24, spin_us=20):
    """What a HIP-event bracket adds to the kernel inside it (launch + the second event's marker packet): brackets of a
    kernel of KNOWN duration — vs_spin idles for exactly spin_us on the 100 MHz device clock — primed like the real ones."""
    stream = torch.cuda.current_stream().cuda_stream
    over = []
    for _ in range(trials):
        _spin(ops.PROFILE_PRIME_US, stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _spin(spin_us, stream)
        e1.record()
        torch.cuda.synchronize()
        over.append(e0.elapsed_time(e1) * 1e3 - spin_us)
    over.sort()
    return max(0.0, over[len(over) // 2])


def collect(fwd_bwd, params, steps=2):
    """-> {kernel id: dict(launches, ms, bytes, flops)} averaged over `steps` eager forward+backward passes.  Every launch
    sits in its own HIP-event bracket on the launch stream; the bracket's own cost (event_bracket_overhead_us, ~5 us, as
    much as a small kernel) is measured in the same pass and subtracted."""
    for p in params:
        p.grad = None
    fwd_bwd()                       # warm (allocator, pack cache)
    torch.cuda.synchronize()
    ops.PROFILE = []
    try:
        for _ in range(steps):
            for p in params:
                p.grad = None
            fwd_bwd()
        torch.cuda.synchronize()
        recs = ops.PROFILE
    finally:
        ops.PROFILE = None
    agg = {}
    global LAST_LAUNCHES, LAST_EVENT_OVERHEAD_US
    LAST_EVENT_OVERHEAD_US = event_bracket_overhead_us()
    dur = lambda e0, e1: max(e0.elapsed_time(e1) - LAST_EVENT_OVERHEAD_US * 1e-3, 1e-4)
    LAST_LAUNCHES = sorted(((dur(e0, e1), kid, det, nb, fl) for kid, nb, fl, det, e0, e1 in recs), reverse=True)
    for kid, nb, fl, det, e0, e1 in recs:
        a = agg.setdefault(kid, {"launches": 0, "ms": 0.0, "bytes": 0.0, "flops": 0.0})
        a["launches"] += 1
        a["ms"] += dur(e0, e1)
        a["bytes"] += nb
        a["flops"] += fl
    for a in agg.values():
        for k in ("launches", "ms", "bytes", "flops"):
            a[k] = a[k] / steps
    return agg


def sources_sha():
    """sha256 (first 16 hex digits) over the kernel sources (csrc/*.hip, *.h, *.inc and include/vaeseg.h): stored with a PMC traffic summary
    (tools/pmc_traffic.py) so that a bench line can tell whether its `traffic` figure was measured on the kernels it ran."""
    import glob
    import hashlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "vae_segmentation_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "vae_segmentation_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(root, "vae_segmentation_amd", "csrc", "*.inc")) + [os.path.join(root, "include", "vaeseg.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def traffic_is_stale(path=None):
    """True when the newest committed PMC summary was taken on different kernel sources than the ones in the tree (or carries no hash)."""
    import json
    try:
        table = json.load(open(path or _traffic_path()))
    except Exception:
        return None
    return table.get("_sources_sha16") != sources_sha()


def _traffic_path():
    import glob
    import os
    cands = sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r*_hbm_traffic.json")))
    return cands[-1] if cands else ""


def measured_step_traffic(path=None):
    """HBM bytes per train step summed over all kernels, from the newest committed PMC summary (profiles/rNN_hbm_traffic.json:
    per-kernel FETCH_SIZE x2 + WRITE_SIZE per launch x launches per step); None when there is none."""
    import json
    try:
        table = json.load(open(path or _traffic_path()))
    except Exception:
        return None
    if "_step_total_bytes" in table:
        return table["_step_total_bytes"]
    try:
        tot = float(sum(rec["hbm_bytes_per_launch"] * rec.get("launches_per_step", 0) for rec in table.values() if isinstance(rec, dict)))
    except Exception:
        return None
    return tot if tot > 0 else None


FAMILIES = (
    ("conv3x3x3, C>=32 levels (k3b<32,..>, k3s)", lambda k: k.startswith("k3b_kernel<32") or k.startswith("k3s_kernel") or k.startswith("chain")),
    ("conv3x3x3, 16-channel layers (k3b<16|8,..>)", lambda k: k.startswith("k3b_kernel<16") or k.startswith("k3b_kernel<8")),
    ("conv3x3x3, 8-channel full-resolution layers (k3t; k3tw = backward-data + the layer's weight gradient)", lambda k: k.startswith("k3t_kernel") or k.startswith("k3tw_kernel")),
    ("conv3x3x3, fp32 kernels (k3_kernel<float>)", lambda k: k.startswith("k3_kernel")),
    ("composed Up heads: transposed conv + 3x3x3 as one operator (k4t, k4g)", lambda k: k.startswith("k4t_kernel") or k.startswith("k4g_kernel")),
    ("stride-2 / transposed convs (g1)", lambda k: k.startswith("g1_kernel")),
    ("weight gradients (grouped)", lambda k: k.startswith("wgrad_multi") or k.startswith("g3")),
    ("InstanceNorm+ReLU backward apply", lambda k: k.startswith("in_relu_bwd")),
)


def kernel_families(fwd_bwd, params, dtype, steps=2):
    """Per-kernel-family summary of live HIP-event timing inside real eager passes: launches, time, algorithmic bytes / FLOPs and the
    fraction of the roofline that bounds the family (HBM for byte-bound families, dense MFMA peak of the dtype otherwise)."""
    agg = collect(fwd_bwd, params, steps)
    peak_tf = MFMA_PEAK_TFLOPS["f32" if dtype == "fp32" else dtype]
    fam = {}
    for kid, a in agg.items():
        name = next((n for n, pred in FAMILIES if pred(kid)), "other timed launches")
        f = fam.setdefault(name, {"launches_per_step": 0.0, "ms_per_step": 0.0, "bytes": 0.0, "flops": 0.0, "kernels": []})
        f["launches_per_step"] += a["launches"]
        f["ms_per_step"] += a["ms"]
        f["bytes"] += a["bytes"]
        f["flops"] += a["flops"]
        f["kernels"].append(kid)
    out = []
    for name, f in sorted(fam.items(), key=lambda kv: -kv[1]["ms_per_step"]):
        sec = f["ms_per_step"] * 1e-3
        hbm_bound = f["bytes"] / (HBM_PEAK_GBS * 1e9) >= f["flops"] / (peak_tf * 1e12)
        ach = f["bytes"] / sec / 1e9 if hbm_bound else f["flops"] / sec / 1e12
        peak = HBM_PEAK_GBS if hbm_bound else peak_tf
        out.append({"family": name, "bound": "hbm" if hbm_bound else "mfma", "achieved": round(ach, 2), "peak": peak,
                    "unit": "GB/s" if hbm_bound else "TFLOP/s", "frac": round(ach / peak, 4),
                    "launches_per_step": round(f["launches_per_step"], 1), "ms_per_step": round(f["ms_per_step"], 4),
                    "avg_launch_us": round(1e3 * f["ms_per_step"] / max(f["launches_per_step"], 1e-9), 2),
                    "algorithmic_bytes_per_step": f["bytes"], "algorithmic_flops_per_step": f["flops"],
                    "kernels": sorted(set(f["kernels"]))})
    return {"method": "HIP events around every launch of real eager passes, bracket overhead %.2f us subtracted" % LAST_EVENT_OVERHEAD_US,
            "timed_ms_per_step": round(sum(f["ms_per_step"] for f in fam.values()), 4), "entries": out}


def _norm_name(name):
    name = name.replace("void ", "").split("(")[0]
    return name.replace(" ", "")


def measured_traffic(kid, path=None):
    """HBM bytes per launch of kernel `kid` from the committed PMC summary (tools/pmc_traffic.py over two rocprofv3 --pmc
    passes, FETCH_SIZE doubled per MI355X_MICROARCH.md); None when no summary covers this kernel."""
    import json
    import os
    path = path or _traffic_path()
    try:
        table = json.load(open(path))
    except Exception:
        return None
    want = _norm_name(kid)
    for name, rec in table.items():
        if _norm_name(name) == want:
            return rec["hbm_bytes_per_launch"]
    return None


def dominant_kernel_roofline(fwd_bwd, params, dtype, steps=2, kernel=None):
    agg = collect(fwd_bwd, params, steps)
    if not agg:
        return None
    kid = kernel if kernel in agg else max(agg, key=lambda k: agg[k]["ms"])
    a = agg[kid]
    sec = a["ms"] * 1e-3
    # fp32-storage kernels (k3_kernel<float,..>, g1_kernel<float,..>, g3_kernel<float,..>) use the exact-f32 MFMA
    mfma_peak = MFMA_PEAK_TFLOPS["f32"] if "float" in kid else MFMA_PEAK_TFLOPS["bf16"]
    t_hbm = a["bytes"] / (HBM_PEAK_GBS * 1e9)
    t_mfma = a["flops"] / (mfma_peak * 1e12)
    total_ms = sum(v["ms"] for v in agg.values())
    if t_hbm >= t_mfma:
        ach, peak, unit, bound = a["bytes"] / sec / 1e9, HBM_PEAK_GBS, "GB/s", "hbm"
    else:
        ach, peak, unit, bound = a["flops"] / sec / 1e12, mfma_peak, "TFLOP/s", "mfma"
    top = sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:6]

    def bound_of(v):
        peak_tf = MFMA_PEAK_TFLOPS["f32"] if "float" in v[0] else MFMA_PEAK_TFLOPS["bf16"]
        s = v[1]["ms"] * 1e-3
        if v[1]["bytes"] / (HBM_PEAK_GBS * 1e9) >= v[1]["flops"] / (peak_tf * 1e12):
            return {"kernel": v[0], "bound": "hbm", "achieved": round(v[1]["bytes"] / s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(v[1]["bytes"] / s / 1e9 / HBM_PEAK_GBS, 4), "launches_per_step": v[1]["launches"],
                    "avg_launch_us": round(1e3 * v[1]["ms"] / v[1]["launches"], 2), "traffic": measured_traffic(v[0])}
        return {"kernel": v[0], "bound": "mfma", "achieved": round(v[1]["flops"] / s / 1e12, 2), "peak": peak_tf, "unit": "TFLOP/s",
                "frac": round(v[1]["flops"] / s / 1e12 / peak_tf, 4), "launches_per_step": v[1]["launches"],
                "avg_launch_us": round(1e3 * v[1]["ms"] / v[1]["launches"], 2), "traffic": measured_traffic(v[0])}
    return {"bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak, "traffic": measured_traffic(kid),
            "kernel": kid, "launches_per_step": a["launches"], "avg_launch_us": 1e3 * a["ms"] / a["launches"],
            "share_of_timed_kernel_time": a["ms"] / total_ms, "event_bracket_overhead_us_subtracted": round(LAST_EVENT_OVERHEAD_US, 2),
            "algorithmic_bytes_per_launch": a["bytes"] / a["launches"], "algorithmic_flops_per_launch": a["flops"] / a["launches"],
            "top_kernels_ms_per_step": {k: round(v["ms"], 4) for k, v in top},
            # the same figures for the other heavy kernels, each against the roofline that bounds it (algorithmic bytes or flops)
            "other_kernels": [bound_of(v) for v in top if v[0] != kid]}
