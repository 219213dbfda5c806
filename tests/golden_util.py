"""Helpers shared by the CPU (oracle vs golden) and GPU (HIP vs golden / oracle) parity tests."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


def sample_idx(n, k=64):
    if n <= k:
        return np.arange(n)
    return (np.arange(k, dtype=np.int64) * (n - 1)) // (k - 1)


def flat64(t):
    return t.detach().double().cpu().reshape(-1).numpy()


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def check_tensor(gold, prefix, t, k=64, rtol=1e-4, what=""):
    """Compare tensor t with the golden summary stored under prefix.* ; returns worst relative error."""
    a = flat64(t)
    errs = {}
    g_l2 = float(gold[prefix + ".l2"])
    scale = max(g_l2 / np.sqrt(a.size), 1e-30)          # rms magnitude of the tensor
    s = a[sample_idx(a.size, k)]
    gs = gold[prefix + ".samples"].astype(np.float64)
    errs["samples"] = float(np.abs(s - gs).max() / max(np.abs(gs).max(), scale))
    errs["l2"] = abs(np.sqrt((a * a).sum()) - g_l2) / max(g_l2, 1e-30)
    errs["sum"] = abs(a.sum() - float(gold[prefix + ".sum"])) / max(float(gold[prefix + ".abssum"]), 1e-30)
    worst = max(errs.values())
    assert worst <= rtol, "%s %s: rel errs %s > %g" % (what, prefix, errs, rtol)
    return worst


def check_grads(gold, prefix, named_grads, k=16, rtol=1e-3, dead_atol=1e-5, what=""):
    """named_grads: iterable of (name, grad tensor or None).  Dead-bias grads (conv biases feeding an
    InstanceNorm) are ~1e-7 noise in the reference: compared with an absolute tolerance."""
    worst = 0.0
    for name, g in named_grads:
        key = "%s.grad.%s" % (prefix, name)
        if key + ".none" in gold:
            assert g is None, "%s: expected no grad for %s" % (what, name)
            continue
        assert g is not None, "%s: missing grad for %s" % (what, name)
        a = flat64(g)
        g_l2 = float(gold[key + ".l2"])
        gs = gold[key + ".samples"].astype(np.float64)
        s = a[sample_idx(a.size, k)]
        my_l2 = float(np.sqrt((a * a).sum()))
        if g_l2 < dead_atol * np.sqrt(a.size) * 10:      # numerically-dead parameter
            assert my_l2 <= max(10 * g_l2, dead_atol * np.sqrt(a.size) * 10), \
                "%s: dead grad %s too large: %g vs ref %g" % (what, name, my_l2, g_l2)
            continue
        e1 = abs(my_l2 - g_l2) / g_l2
        e2 = float(np.abs(s - gs).max() / max(np.abs(gs).max(), g_l2 / np.sqrt(a.size)))
        worst = max(worst, e1, e2)
        assert max(e1, e2) <= rtol, "%s: grad %s rel err l2 %g samples %g > %g" % (what, name, e1, e2, rtol)
    return worst
