"""Diagnostic: per-phase cycle sums of k3b_kernel in its backward-data forms — plain (gradient already applied) vs fused apply (FA) — on one shape.
needs tools/_dbg/libvaeseg_stamps.so (tools/build_stamps.sh).  usage: python tools/stamps_k3_fa.py N SIDE C M"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vae_segmentation_amd import _lib, ops
dbg = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("VS_DBG_DIR", "_stamps"), os.environ.get("VS_DBG_LIB", "libvaeseg_stamps.so")))
for name, (restype, argtypes) in _lib.parse_header().items():
    fn = getattr(dbg, name); fn.restype = restype; fn.argtypes = argtypes
dbg.vs_debug_read_k3_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
n, s, c, m = [int(v) for v in sys.argv[1:5]]
dt = torch.bfloat16
g = torch.randn(n, s, s, s, c, device="cuda").to(dt)
ax = torch.randn(n, s, s, s, c, device="cuda").to(dt)
mx = torch.randn(n, s, s, s, m, device="cuda").to(dt)
w = torch.randn(c, m, 3, 3, 3, device="cuda") * 0.05
wpb = ops.pack_weight(w, ops.VS_PACK_ROWS_D1_FLIP, c, dt)
axs, mxs = ops.instnorm_stats(ax), ops.instnorm_stats(mx)
asums = ops._new_stats(n, c, g.device)
y = torch.empty_like(mx)
dx = torch.empty_like(g)
names = ["prologue (tables, first loads)", "barrier 1 (prev stage read by all) + tile setup", "vmcnt wait + transform + LDS write", "barrier 2", "next-stage load issue + mask loads", "MFMA phase", "epilogue", "-"]
TICK_NS = 1.0 / 2.2


def report(tag):
    nwg = 2048
    buf = np.zeros(nwg * 8, dtype=np.uint64)
    dbg.vs_debug_read_k3_stamps(buf.ctypes.data, nwg * 8)
    raw = buf.reshape(nwg, 8)
    raw = raw[raw[:, :7].sum(1) > 0]
    t_start = (raw[:, 7] >> np.uint64(32)).astype(np.int64) & 0xffffffff
    raw = raw[t_start > t_start.max() - 10000]            # this launch's workgroups only (the buffer keeps the entries of earlier launches with larger grids)
    t_start = (raw[:, 7] >> np.uint64(32)).astype(np.int64) & 0xffffffff
    t_end = (raw[:, 7] & np.uint64(0xffffffff)).astype(np.int64)
    t0 = t_start.min()
    print("== %s: %d workgroups; starts 5/50/95/100 %% (us): %s  ends: %s" % (tag, len(raw), [round(float(v) * 0.01, 2) for v in np.percentile(t_start - t0, [5, 50, 95, 100])],
                                                                          [round(float(v) * 0.01, 2) for v in np.percentile(t_end - t0, [5, 50, 95, 100])]))
    st = raw.astype(np.int64)
    st[:, 7] = 0
    tot = st.sum(1)
    for i, nm in enumerate(names[:7]):
        print("   %-52s median %8d ticks (%.2f us)" % (nm, np.median(st[:, i]), np.median(st[:, i]) * TICK_NS * 1e-3))
    print("   total median %.2f us, max %.2f us" % (np.median(tot) * TICK_NS * 1e-3, tot.max() * TICK_NS * 1e-3))


def clear():
    pass


for mode in ("plain", "fused", "fused_nodx"):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(3):
        sums = ops._new_stats(n, m, g.device)
        torch.cuda.synchronize()
        ev0.record()
        if mode == "plain":
            rc = dbg.vs_conv_gather_bwd_data(g.data_ptr(), wpb.data_ptr(), y.data_ptr(), mx.data_ptr(), mxs.data_ptr(), sums.data_ptr(), n, s, s, s, c, m, 0, 1, 1e-5, None)
        else:
            rc = dbg.vs_conv_k3_bwd_data_fused_apply(g.data_ptr(), ax.data_ptr(), axs.data_ptr(), asums.data_ptr(), wpb.data_ptr(), y.data_ptr(), mx.data_ptr(), mxs.data_ptr(),
                                                     sums.data_ptr(), dx.data_ptr() if mode == "fused" else None, n, s, s, s, c, m, 1, 1e-5, None)
        ev1.record()
        assert rc == 0, rc
        torch.cuda.synchronize()
    print("%s: launch %.1f us (event bracket, includes ~5 us of bracket)" % (mode, ev0.elapsed_time(ev1) * 1e3))
    report(mode)
