"""Diagnostic: where the time of a chain kernel (csrc/chain.h) goes, layer by layer — 100 MHz stamps of the first XCD slot's workgroups.
Needs tools/_dbg/libvaeseg_chainstamps.so:  VS_STAMPS_DEF=VS_CHAIN_STAMPS VS_STAMPS_OUT=libvaeseg_chainstamps.so bash tools/build_stamps.sh
usage: python tools/chain_stamps.py N CIN COUT SIDE [bwd]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VS_LIBVAESEG"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dbg", "libvaeseg_chainstamps.so")
os.environ["VS_DETERMINISTIC"] = "0"
import numpy as np
import torch
from vae_segmentation_amd import _lib, ops
dbg = ctypes.CDLL(os.environ["VS_LIBVAESEG"])
dbg.vs_debug_read_chain_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
n, cin, cout, side = [int(v) for v in sys.argv[1:5]]
bwd = len(sys.argv) > 5 and sys.argv[5] == "bwd"
dt = torch.bfloat16
x = torch.randn(n, side, side, side, cin, device="cuda").to(dt)
ws = [(torch.randn(co, ci, 3, 3, 3, device="cuda") / (27 * ci) ** 0.5) for ci, co in ((cin, cout), (cout, cout), (cout, cout))]
gy = torch.randn(n, side, side, side, cout, device="cuda").to(dt)
NAMES_F = ["request weights", "wait for the previous layer", "x + statistics loaded, tables built", "stage loop", "partials + epilogue", "-", "-", "-", "drain + arrive"]
for it in range(5):
    xi = x.clone().requires_grad_(True)
    with ops.arena_scope(x.device):
        y, ys = ops.ConvK3Chain.apply(xi, None, ws[0], None, ws[1], None, ws[2], None)
        torch.cuda.synchronize()
        if not bwd:
            buf = np.zeros(64 * 64, dtype=np.uint64)
            dbg.vs_debug_read_chain_stamps(buf.ctypes.data, 64 * 64)
        y.backward(gy)
        torch.cuda.synchronize()
        if bwd:
            buf = np.zeros(64 * 64, dtype=np.uint64)
            dbg.vs_debug_read_chain_stamps(buf.ctypes.data, 64 * 64)
ops.chain_fault()
raw = buf.reshape(64, 64).astype(np.int64)
raw = raw[raw[:, 0] > 0]
t0 = raw[:, 0].min()
print("%s chain %d x %d^3, %d -> %d: %d workgroups of sample 0; us after the first workgroup's start (median over workgroups; min..max)" % ("backward" if bwd else "forward", n, side, cin, cout, len(raw)))
labels = {0: "entry", 1: "weights requested", 2: "wait over", 3: "tables built (x, statistics in)", 4: "stage loop done", 5: "epilogue done", 6: "arrived (before apply)", 7: "wait over (apply)", 8: "apply done", 9: "arrived / layer end"}
prev = None
for l in range(3):
    for k in range(10):
        col = raw[:, l * 16 + k]
        if (col <= 0).all():
            continue
        v = (col[col > 0] - t0) * 0.01
        med = float(np.median(v))
        print("  layer %d  %-34s %7.2f  (%6.2f .. %6.2f)%s" % (l, labels[k], med, v.min(), v.max(), "" if prev is None else "   +%.2f" % (med - prev)))
        prev = med
    ph = raw[:, l * 16 + 10:l * 16 + 16].astype(np.float64)
    if ph[:, 5].max() > 0:
        names = ["barrier (previous stage read by all)", "vmcnt wait + normalise + LDS write", "barrier", "issue of the stage after next", "LDS reads + MFMA", "body total (shader clocks)"]
        body_us = None
        for k in range(6):
            print("      stage-loop phase %-42s median %8.0f clocks" % (names[k], np.median(ph[:, k])))
