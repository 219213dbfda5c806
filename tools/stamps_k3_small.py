"""Diagnostic: per-phase s_memtime stamps of k3_small (needs tools/_dbg/libvaeseg_stamps.so built with -DVS_STAMPS)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vae_segmentation_amd import _lib, ops
dbg = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dbg", "libvaeseg_stamps.so"))
for name, (restype, argtypes) in _lib.parse_header().items():
    fn = getattr(dbg, name); fn.restype = restype; fn.argtypes = argtypes
dbg.vs_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
n, c, m, s = [int(v) for v in sys.argv[1:5]]
x = torch.randn(n, s, s, s, c, device="cuda").to(torch.bfloat16)
w = torch.randn(m, c, 3, 3, 3, device="cuda") * 0.05
wp = ops.pack_weight(w, 0, c, torch.bfloat16)
xs = ops.instnorm_stats(x)
y = torch.empty(n, s, s, s, m, device="cuda", dtype=torch.bfloat16)
for it in range(3):
    ys = torch.zeros(n, m, 2, dtype=torch.float64, device="cuda")
    rc = dbg.vs_conv_gather_fwd(x.data_ptr(), xs.data_ptr(), wp.data_ptr(), None, y.data_ptr(), ys.data_ptr(), n, s, s, s, c, m, 0, 1, 1e-5, None)
    assert rc == 0, rc
    torch.cuda.synchronize()
nwg = n * ((m + 15) // 16)
buf = np.zeros(nwg * 16, dtype=np.uint64)
dbg.vs_debug_read_stamps(buf.ctypes.data, nwg * 16)
st = buf.reshape(nwg, 16)[:, :13].astype(np.int64)
t0 = st[:, 0].min()
names = ["start", "setup done", "staging loads issued", "tables done", "barrier A", "LDS writes done", "A loads issued", "barrier B", "MFMA done", "barrier C", "partials in LDS", "epilogue stores", "end"]
rel = st - st[:, :1]
print("workgroups:", nwg, " start skew (ticks): max %d" % (st[:, 0].max() - t0))
for i, nm in enumerate(names):
    print("%-22s median +%7d ticks   (phase %6d)" % (nm, np.median(rel[:, i]), np.median(rel[:, i] - (rel[:, i - 1] if i else 0))))
print("kernel span (first start -> last end): %d ticks" % (st[:, 12].max() - t0))
