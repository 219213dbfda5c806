"""ctypes binding of tools/probe/libvsprobe.so: measurement aids that are NOT part of the product ABI (include/vaeseg.h).

  vs_spin(microseconds, stream)                       one wave idles on the stream (<= 1000 us): keeps the queue busy in front of a HIP-event
                                                      bracket so the bracket does not measure the command processor's wake-up latency
  vs_debug_store_probe(p, n_wg, mode, stream)         what a kernel's stores / loads add to a dependent graph node (tools/launch_floor.py)
  vs_debug_grid_barrier_probe(flags, ticks, n_wg, iters, mode, stream)    cost of device-wide barriers / group protocols (tools/barrier_probe.py)
  vs_debug_down_composed_probe(x, w_packed, y, n, d, h, w, stream)        the Down head (stride-2 conv, then 3x3x3 conv) as one 6x6x6 / stride-2 operator (tools/down_probe.py)
  vs_debug_xcd_group_probe(flags, ticks, xcc, err, payload, slots, n_wg, iters, gsz, pb, mode, stream)   one per-sample layer boundary kept inside a launch (tools/xcd_probe.py)
  vs_debug_xcd_chain_probe(err, payload, slots, n_wg, gsz, pb, it, stream)                               the same round as one launch of a dependent chain
"""
import ctypes
import os

import torch  # noqa: F401  (the HIP runtime in the process must be torch's: vae_segmentation_amd/_lib.py)

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvsprobe.so")
if not os.path.exists(_PATH):
    raise ImportError("%s not found — build it with `make -C vae_segmentation_amd/csrc`" % _PATH)
lib = ctypes.CDLL(_PATH)
_V, _I = ctypes.c_void_p, ctypes.c_int
lib.vs_spin.argtypes, lib.vs_spin.restype = [_I, _V], _I
lib.vs_debug_store_probe.argtypes, lib.vs_debug_store_probe.restype = [_V, _I, _I, _V], _I
lib.vs_debug_grid_barrier_probe.argtypes, lib.vs_debug_grid_barrier_probe.restype = [_V, _V, _I, _I, _I, _V], _I
lib.vs_debug_down_composed_probe.argtypes, lib.vs_debug_down_composed_probe.restype = [_V, _V, _V, _I, _I, _I, _I, _V], _I      # tools/down_probe.py
lib.vs_debug_xcd_group_probe.argtypes, lib.vs_debug_xcd_group_probe.restype = [_V, _V, _V, _V, _V, _V, _I, _I, _I, _I, _I, _V], _I    # tools/xcd_probe.py
lib.vs_debug_xcd_chain_probe.argtypes, lib.vs_debug_xcd_chain_probe.restype = [_V, _V, _V, _I, _I, _I, _I, _V], _I


def check(code, what=""):
    if code != 0:
        raise RuntimeError("libvsprobe %s failed (%d)" % (what, code))
