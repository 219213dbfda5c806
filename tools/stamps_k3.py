"""Diagnostic: per-phase cycle sums of k3b_kernel (needs tools/_dbg/libvaeseg_stamps.so from tools/build_stamps.sh)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vae_segmentation_amd import _lib, ops
dbg = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dbg", os.environ.get("VS_DBG_LIB", "libvaeseg_stamps.so")))
for name, (restype, argtypes) in _lib.parse_header().items():
    fn = getattr(dbg, name); fn.restype = restype; fn.argtypes = argtypes
dbg.vs_debug_read_k3_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
n, c, m, s = [int(v) for v in sys.argv[1:5]]
x = torch.randn(n, s, s, s, c, device="cuda").to(torch.bfloat16)
w = torch.randn(m, c, 3, 3, 3, device="cuda") * 0.05
wp = ops.pack_weight(w, 0, c, torch.bfloat16)
xs = ops.instnorm_stats(x)
y = torch.empty(n, s, s, s, m, device="cuda", dtype=torch.bfloat16)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(int(os.environ.get('VS_ITERS', 3))):
    ys = torch.zeros(ops.STAT_SLOTS, n, m, 2, dtype=torch.float64, device="cuda")
    ev0.record()
    rc = dbg.vs_conv_gather_fwd(x.data_ptr(), xs.data_ptr(), wp.data_ptr(), None, y.data_ptr(), ys.data_ptr(), n, s, s, s, c, m, 0, 1, 1e-5, None)
    ev1.record()
    assert rc == 0, rc
    torch.cuda.synchronize()
print("launch %.1f us" % (ev0.elapsed_time(ev1) * 1e3))
TICK_NS = 1.0 / 2.2   # s_memtime ticks at ~2.2 GHz under this load (calibrated against s_memrealtime)
nwg = 2048
buf = np.zeros(nwg * 8, dtype=np.uint64)
dbg.vs_debug_read_k3_stamps(buf.ctypes.data, nwg * 8)
raw = buf.reshape(nwg, 8)
raw = raw[raw[:, :7].sum(1) > 0]
# slot 7: (start << 32) | end of the workgroup on the 100 MHz s_memrealtime clock -> dispatch timeline
t_start = (raw[:, 7] >> np.uint64(32)).astype(np.int64) & 0xffffffff
t_end = (raw[:, 7] & np.uint64(0xffffffff)).astype(np.int64)
t0 = t_start.min()
print("workgroup starts (us after the first): percentiles 5/50/95/100:", [round(float(v) * 0.01, 2) for v in np.percentile(t_start - t0, [5, 50, 95, 100])],
      " ends: 5/50/95/100:", [round(float(v) * 0.01, 2) for v in np.percentile(t_end - t0, [5, 50, 95, 100])])
late = (t_start - t0) > 300
print("workgroups starting > 3 us after the first: %d of %d" % (int(late.sum()), len(raw)))
st = raw.astype(np.int64)
st[:, 7] = 0
names = ["prologue (tables, first loads)", "barrier 1 (prev stage read by all) + tile setup", "vmcnt wait + transform + LDS write", "barrier 2", "next-stage load issue", "MFMA phase", "epilogue", "-"]
tot = st.sum(1)
print("total ticks/WG percentiles 5/25/50/75/95/100:", [int(v) for v in np.percentile(tot, [5, 25, 50, 75, 95, 100])], " (%.1f us max)" % (tot.max() * TICK_NS * 1e-3))
print("workgroups with stamps:", len(st), " median total ticks/WG:", int(np.median(tot)))
for i, nm in enumerate(names[:7]):
    print("%-50s median %8d ticks  %5.1f %%  (%.1f us)" % (nm, np.median(st[:, i]), 100 * np.median(st[:, i]) / np.median(tot), np.median(st[:, i]) * TICK_NS * 1e-3))
