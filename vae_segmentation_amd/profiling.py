"""Live per-kernel timing for bench.py's roofline leg: HIP events (torch.cuda.Event on the launch stream) around
every launch of each libvaeseg kernel inside real train steps, summed per kernel instantiation."""
import torch

from . import ops

LAST_LAUNCHES = []
HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec); ~6.3 TB/s achievable
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}


def collect(fwd_bwd, params, steps=2):
    """-> {kernel id: dict(launches, ms, bytes, flops)} averaged over `steps` eager forward+backward passes."""
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
    global LAST_LAUNCHES
    LAST_LAUNCHES = sorted(((e0.elapsed_time(e1), kid, det, nb, fl) for kid, nb, fl, det, e0, e1 in recs), reverse=True)
    for kid, nb, fl, det, e0, e1 in recs:
        a = agg.setdefault(kid, {"launches": 0, "ms": 0.0, "bytes": 0.0, "flops": 0.0})
        a["launches"] += 1
        a["ms"] += e0.elapsed_time(e1)
        a["bytes"] += nb
        a["flops"] += fl
    for a in agg.values():
        for k in ("launches", "ms", "bytes", "flops"):
            a[k] = a[k] / steps
    return agg


def _norm_name(name):
    name = name.replace("void ", "").split("(")[0]
    return name.replace(" ", "")


def measured_traffic(kid, path=None):
    """HBM bytes per launch of kernel `kid` from the committed PMC summary (tools/pmc_traffic.py over two rocprofv3 --pmc
    passes, FETCH_SIZE doubled per MI355X_MICROARCH.md); None when no summary covers this kernel."""
    import json
    import os
    path = path or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r01_hbm_traffic.json")
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
    # wgrad runs on the exact-f32 MFMA whatever the storage type; the implicit-GEMM convs on the storage type's MFMA
    mfma_peak = MFMA_PEAK_TFLOPS["f32"] if (kid.startswith("g3_kernel") or "float" in kid) else MFMA_PEAK_TFLOPS["bf16"]
    t_hbm = a["bytes"] / (HBM_PEAK_GBS * 1e9)
    t_mfma = a["flops"] / (mfma_peak * 1e12)
    total_ms = sum(v["ms"] for v in agg.values())
    if t_hbm >= t_mfma:
        ach, peak, unit, bound = a["bytes"] / sec / 1e9, HBM_PEAK_GBS, "GB/s", "hbm"
    else:
        ach, peak, unit, bound = a["flops"] / sec / 1e12, mfma_peak, "TFLOP/s", "mfma"
    top = sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:6]
    return {"bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak, "traffic": measured_traffic(kid),
            "kernel": kid, "launches_per_step": a["launches"], "avg_launch_us": 1e3 * a["ms"] / a["launches"],
            "share_of_timed_kernel_time": a["ms"] / total_ms,
            "algorithmic_bytes_per_launch": a["bytes"] / a["launches"], "algorithmic_flops_per_launch": a["flops"] / a["launches"],
            "top_kernels_ms_per_step": {k: round(v["ms"], 4) for k, v in top}}
