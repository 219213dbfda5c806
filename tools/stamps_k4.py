"""Diagnostic: per-phase cycle sums of k4t_kernel (composed Up forward; needs tools/_dbg/libvaeseg_stamps.so from tools/build_stamps.sh).
usage: python tools/stamps_k4.py N C Co S"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vae_segmentation_amd import _lib, ops
dbg = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dbg", "libvaeseg_stamps.so"))
for name, (restype, argtypes) in _lib.parse_header().items():
    fn = getattr(dbg, name); fn.restype = restype; fn.argtypes = argtypes
dbg.vs_debug_read_k4_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
n, c, co, s = [int(v) for v in sys.argv[1:5]]
dt = torch.bfloat16
x = torch.randn(n, s, s, s, c, device="cuda").to(dt)
xs = ops.instnorm_stats(x)
wt = torch.randn(c, c, 2, 2, 2, device="cuda") * 0.2
bt = torch.randn(c, device="cuda") * 0.1
w3 = torch.randn(co, c, 3, 3, 3, device="cuda") * 0.05
plan = ops.up_plan(wt, bt, w3, dt)
y = torch.empty(n, 2 * s, 2 * s, 2 * s, co, device="cuda", dtype=dt)
for it in range(3):
    ys = torch.zeros(4, n, co, 2, dtype=torch.float64, device="cuda")
    rc = dbg.vs_up_conv_fwd(x.data_ptr(), xs.data_ptr(), plan["img_f"].data_ptr(), plan["taps_f"].data_ptr(), plan["btab"].data_ptr(), y.data_ptr(), ys.data_ptr(), n, s, s, s, c, co, 1, 1e-5, None)
    assert rc == 0, rc
    torch.cuda.synchronize()
nwg = 2048
buf = np.zeros(nwg * 8, dtype=np.uint64)
dbg.vs_debug_read_k4_stamps(buf.ctypes.data, nwg * 8)
raw = buf.reshape(nwg, 8)
raw = raw[raw[:, :7].sum(1) > 0]
t_start = (raw[:, 7] >> np.uint64(32)).astype(np.int64) & 0xffffffff
t_end = (raw[:, 7] & np.uint64(0xffffffff)).astype(np.int64)
t0 = t_start.min()
print("workgroup starts (us after the first) 5/50/95/100:", [round(float(v) * 0.01, 2) for v in np.percentile(t_start - t0, [5, 50, 95, 100])],
      " ends 5/50/95/100:", [round(float(v) * 0.01, 2) for v in np.percentile(t_end - t0, [5, 50, 95, 100])])
st = raw.astype(np.int64); st[:, 7] = 0
names = ["prologue", "barrier 1 (prev stage read)", "vmcnt wait + normalise + LDS write", "barrier 2", "next-stage load issue", "MFMA loop", "epilogue + flush"]
tot = st.sum(1)
print("workgroups:", len(st), " total ticks/WG 5/50/95/100:", [int(v) for v in np.percentile(tot, [5, 50, 95, 100])])
for i, nm in enumerate(names):
    print("%-40s median %8d ticks  %5.1f %%   p95 %8d" % (nm, np.median(st[:, i]), 100 * np.median(st[:, i]) / np.median(tot), np.percentile(st[:, i], 95)))
