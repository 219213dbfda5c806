"""torch.autograd.Function wrappers around the libvaeseg C ABI (include/vaeseg.h).

PyTorch supplies device memory, streams and the autograd graph; every byte of arithmetic on the hot path
is done by the HIP kernels.  Nothing here has a CPU path: tensors must live on the GPU.

Internal activations are channels-last ``(N, D, H, W, C)`` tensors in the kernel dtype (fp32 or bf16).
A *lazy* activation is the pair ``(raw, stats)``: ``raw`` is a conv output before InstanceNorm+ReLU and
``stats`` the fp64 ``(N, C, 2)`` (sum, sumsq) its producer accumulated; consumers normalise on load.
For autograd, ``stats`` is a non-differentiable side output and the gradient attached to ``raw`` is the
total derivative (the InstanceNorm+ReLU backward is applied by the consumer's backward).
"""
import ctypes as _ctypes
import os
import weakref as _weakref

import torch

from . import _lib
from ._lib import (VS_BF16, VS_CONV_K2S2, VS_CONV_K3, VS_CONV_UP, VS_F16, VS_F32, VS_PACK_ROWS_D0, VS_PACK_ROWS_D1_FLIP,
                   VS_PACK_SCATTER_D1, check, lib)

_DT_TORCH = {VS_F32: torch.float32, VS_BF16: torch.bfloat16, VS_F16: torch.float16}
_DT_VS = {v: k for k, v in _DT_TORCH.items()}
KERNEL_DTYPES = tuple(_DT_VS)
# PACK-ONLY dtype (include/vaeseg.h VS_F32X3): the three-bf16-limb image of an fp32 3x3x3 weight that the parity mode's limb kernels read
# (csrc/igemm_k3x.h; vs_conv_k3_f32_limbs).  Tensors stay float32; only pack_weight / pack_weight_cached accept the marker.
VS_F32X3 = 3
F32X3 = "f32x3"
_DT_TORCH[VS_F32X3] = F32X3
_DT_VS[F32X3] = VS_F32X3
_F32_LIMBS = bool(lib.vs_conv_k3_f32_limbs(96, 96, 96, 8))


def k3_pack_dtype(x):
    """the image format of a 3x3x3 weight applied to the channels-last tensor x: the limb image in the fp32 parity mode where the library runs
    that launch on the bf16 matrix cores (vs_conv_k3_f32_limbs: not the volumes up to 6^3 of the deep levels), the storage type itself otherwise"""
    if x.dtype == torch.float32 and lib.vs_conv_k3_f32_limbs(x.shape[1], x.shape[2], x.shape[3], x.shape[4]):
        return F32X3
    return x.dtype


def vs_of(dtype):
    """torch dtype -> VS_* enum of include/vaeseg.h"""
    try:
        return _DT_VS[dtype]
    except KeyError:
        raise TypeError("kernel dtype must be float32, bfloat16 or float16, got %s" % (dtype,)) from None

EPS_IN = 1e-5       # nn.InstanceNorm3d default eps (joint_model.py:11)


def _stream():
    return torch.cuda.current_stream().cuda_stream


# Optional per-launch timing (bench.py roofline leg): when a list is installed here, every C-ABI launch below is
# bracketed by HIP events recorded on the stream the kernel is launched on; entries are
# (kernel id, algorithmic bytes, algorithmic flops, start event, end event).
PROFILE = None


def _esize(t):
    return 4 if t.dtype == torch.float32 else 2


class VsConfig(_ctypes.Structure):
    """vs_config (include/vaeseg.h): the library's tuning switches"""
    _fields_ = [(n, _ctypes.c_int) for n in ("k3_small", "k3_tall", "k3_wgs_per_cu", "k3t_wgs_per_cu", "k3f_min_wgs", "mt_min_wgs", "f32_limbs", "g1_limbs", "k3x_ck",
                                            "k3x_toeplitz", "fuse_wgrad", "epilogue_apply", "chain", "k2s2_stream", "k2s8_wgs_per_cu", "up_wgs_per_cu", "up_rb",
                                            "wgrad_uber", "wgrad_mpack", "wgrad_swap", "wgrad_big", "wgrad_xcd", "k3_short_tiles", "wgrad_bias_fold")] + \
               [(n, _ctypes.c_longlong) for n in ("wgrad_wgs", "wgrad_f32_tiles", "wgrad_group_wgs", "wgrad_big_min_voxels")]


def get_config():
    """-> dict of the active library's tuning switches (vs_get_config)"""
    if lib.vs_config_bytes() != _ctypes.sizeof(VsConfig):
        raise _lib.VaesegError("vs_config layout mismatch: the library's struct has %d bytes, this binding %d" % (lib.vs_config_bytes(), _ctypes.sizeof(VsConfig)))
    c = VsConfig()
    check(lib.vs_get_config(_ctypes.addressof(c)), "get_config")
    return {n: getattr(c, n) for n, _ in VsConfig._fields_ if n != "reserved_"}


def set_config(**kw):
    """Change tuning switches of the library (vs_set_config, the single writer; every loaded build gets the same values).  Call between launches.
    -> the previous values of the switches that were changed (hand them back to set_config to restore)."""
    cur = get_config()
    unknown = [k for k in kw if k not in cur]
    if unknown:
        raise KeyError("not a vs_config field: %s" % ", ".join(unknown))
    old = {k: cur[k] for k in kw}
    cur.update(kw)
    c = VsConfig(**cur)
    for l in lib.loaded():
        check(l.vs_set_config(_ctypes.addressof(c)), "set_config")
    return old


class config:
    """with ops.config(wgrad_mpack=0): ...  — tuning switches changed for a block (tests, A/B runs), restored on the way out"""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = set_config(**self.kw)
        return self

    def __exit__(self, *exc):
        set_config(**self.old)
        return False


def _pick_mt(rows16, tiles):
    """mirror of pick_mt() in csrc/conv_api.hip (kernel instantiation naming only)."""
    min_wgs = get_config()["mt_min_wgs"]
    for mt in (64, 32, 16):
        if rows16 % mt:
            continue
        if tiles * (rows16 // mt) >= min_wgs or mt == 16:
            return mt
    return 16


def _k3_kid(tname, ck, mt, sums=False, geom=None, m=None, lazy=False):
    """kernel instantiation name of a 3x3x3 launch (mirrors g1_dispatch_k3_* / k3b_use_tall in csrc; rocprof prints the same
    string).  geom = (n, d, h, w) of the convolution's grid, m = stored output channels.  The 16-bit kernels carry their storage
    type as the LAST template argument."""
    if tname == "float":
        return "k3_kernel<float,%d,%d,0>" % (ck, mt)
    hs = "true" if (lazy and not sums) else "false"
    if ck == 8 and m == 8:
        return "k3t_kernel<0,%s,8,%s,%s>" % ("true" if sums else "false", hs, tname)
    cfg = get_config()
    if ck == 32 and geom is not None and (geom[1] + 2) * (geom[2] + 2) * (geom[3] + 2) <= 512 and cfg["k3_small"] != 0:
        tv = (geom[1] + 2) * (geom[2] + 2) * (geom[3] + 2)
        return "k3s_kernel<%s,%d,%s,%s>" % ("true" if sums else "false", 128 if tv <= 128 else 512, hs, tname)
    yt = 4
    if geom is not None and ck == 32 and mt == 16 and m is not None and cfg["k3_short_tiles"]:      # csrc/igemm_k3_h16.inc: short tiles for the under-filled launches
        n, d, h, w = geom
        rt = (m + 15) // 16
        zx = n * ((d + 3) // 4) * ((w + 15) // 16)
        if zx * ((h + 3) // 4) * rt <= 128 and zx * ((h + 1) // 2) * rt <= 256:
            yt = 1 if (cfg["k3_short_tiles"] >= 2 and zx * ((h + 1) // 2) * rt <= 128 and zx * h * rt <= 256) else 2
    if geom is not None and ck < 32 and mt == 16 and cfg["k3_tall"] != 0:
        n, d, h, w = geom
        tiles = n * ((d + 3) // 4) * ((h + 3) // 4) * ((w + 15) // 16)
        tall = n * ((d + 3) // 4) * ((h + 7) // 8) * ((w + 15) // 16)
        if cfg["k3_tall"] == 1 or (tall >= 256 and tiles <= 2048):
            yt = 8
    return "k3b_kernel<%d,%d,0,%s,%d,%s,%s>" % (ck, min(mt, 32), "true" if sums else "false", yt, hs, tname)


PROFILE_PRIME_US = 80
PROFILE_SPIN = [None]          # set by profiling.py (the measurement aid lives in tools/probe/libvsprobe.so, not in the product library)


class _timed:
    def __init__(self, kid, nbytes, flops, detail=""):
        self.rec = None if PROFILE is None else [kid, float(nbytes), float(flops), detail]

    def __enter__(self):
        if self.rec is not None:
            if PROFILE_PRIME_US:
                # keep the queue busy while the bracket is enqueued (tools/probe: vs_spin): the two event packets and
                # the kernel then run back to back, as the kernel does inside the replayed graph
                PROFILE_SPIN[0](PROFILE_PRIME_US, _stream())
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.rec.append(e)

    def __exit__(self, *exc):
        if self.rec is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.rec.append(e)
            PROFILE.append(tuple(self.rec))
        return False


def _p(t):
    return None if t is None else t.data_ptr()


def _tname(t):
    """the storage type as it appears in the profiler's kernel names"""
    return {torch.float32: "float", torch.bfloat16: "unsigned short", torch.float16: "_Float16"}[t.dtype]


def vs_dtype(t):
    return vs_of(t.dtype)


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("vae_segmentation_amd runs on the GPU only (libvaeseg HIP kernels); "
                               "got a CPU tensor — there is no CPU fallback")


def cpad(c):
    if c <= 8:
        return 8
    if c <= 16:
        return 16
    return (c + 31) // 32 * 32


# ------------------------------------------------------------------------------------------------
# weight packing (fragment order); frozen weights are packed once and cached
# ------------------------------------------------------------------------------------------------


def pack_weight(w, form, c_pad, dtype, out=None):
    """w: (d0, d1, k, k, k) fp32 parameter -> uint8 buffer holding the MFMA-fragment image (written into `out` when given)."""
    _require_cuda(w)
    w = w.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    d0, d1 = w.shape[0], w.shape[1]
    ntaps = w[0, 0].numel()
    dt = vs_of(dtype)
    if form == VS_PACK_ROWS_D0:
        rows, gemm_taps = d0, ntaps
    elif form == VS_PACK_ROWS_D1_FLIP:
        rows, gemm_taps = d1, ntaps
    else:
        rows, gemm_taps = ntaps * d1, 1
    nbytes = lib.vs_packed_weight_bytes(rows, c_pad, gemm_taps, dt)
    buf = torch.empty(nbytes, dtype=torch.uint8, device=w.device) if out is None else out
    if buf.numel() != nbytes:
        raise ValueError("pack_weight: out buffer holds %d bytes, image needs %d" % (buf.numel(), nbytes))
    check(lib.vs_pack_weight(w.data_ptr(), buf.data_ptr(), d0, d1, ntaps, c_pad, form, dt, _stream()), "pack_weight")
    return buf


def pack_weight_cached(param, form, c_pad, dtype):
    """Packed image of a conv weight.
    Frozen parameters: packed once per (tensor object, version, epoch); the cache lives on the tensor itself, so it dies
    with it and a recycled device address can never alias another weight.
    Trainable parameters: the first use registers (param, form, c_pad, dtype) in a repack plan and packs it; afterwards
    `repack_trainable()` — called by the native optimisers right after their update kernel — refreshes EVERY registered
    image in one multi-tensor launch, so a training step issues no per-layer pack launches.  An image is trusted only if
    it was packed at the parameter's current (version, epoch); otherwise it is re-packed on the spot."""
    dt = vs_of(dtype)
    key = (form, c_pad, dt)
    if not param.is_leaf:
        # a weight computed from a parameter in this very pass (GSConv3d's |w| / sum, the 1x1x1 conv embedded in a 3x3x3 one): a
        # temporary — packed on the spot, never registered anywhere
        return pack_weight(param, form, c_pad, dtype)
    if param.requires_grad:
        plan = getattr(param, "_vs_pack_plan", None)
        if plan is None:
            plan = {}
            param._vs_pack_plan = plan
        ent = plan.get(key)
        stamp = (param._version, param.data_ptr(), _TRAIN_EPOCH[0])
        if ent is None:
            buf = pack_weight(param, form, c_pad, dtype)
            plan[key] = [buf, stamp]
            _REPACK["params"][id(param)] = param
            _REPACK["dirty"] = True
            return buf
        if ent[1] != stamp:
            pack_weight(param, form, c_pad, dtype, out=ent[0])
            ent[1] = stamp
        return ent[0]
    cache = _frozen_cache(param)
    stamp = (param._version, param.data_ptr(), _PACK_EPOCH[0])
    ent = cache.get(key)
    if ent is None:
        cache[key] = [pack_weight(param, form, c_pad, dtype), stamp]
        return cache[key][0]
    if ent[1] != stamp:
        # re-pack INTO the existing buffer: a captured HIP graph holds this address (GraphedStep / TestTimeFinetune replay the
        # frozen VAE / teacher with pointer arguments only), so an invalidated image must be refreshed in place, never re-allocated
        pack_weight(param, form, c_pad, dtype, out=ent[0])
        ent[1] = stamp
    return ent[0]


def _frozen_cache(param):
    """{key: [buffer, stamp]} living on the tensor object itself: it dies with the tensor, so a recycled device address can
    never alias another weight; buffers are never replaced once handed out (see pack_weight_cached)."""
    cache = getattr(param, "_vs_pack_cache", None)
    if cache is None:
        cache = {}
        param._vs_pack_cache = cache
    return cache


def refresh_frozen_packs(module):
    """Re-pack, in place and now, every cached image of `module`'s frozen weights (call after writing them through raw
    pointers — optim.ema_update, load_state_dict on a network a captured graph replays): graph replays that follow read the
    new weights without an eager forward in between."""
    for prm in module.parameters():
        for dt, plan in (getattr(prm, "_vs_up_plan", None) or {}).items():       # composed Up images cached on the 3x3x3 weight (up_plan)
            wt, bt = plan["src"]
            with torch.no_grad():
                _up_compose_into(plan, wt.detach(), None if bt is None else bt.detach(), prm.detach(), dt)
                plan["stamp"] = _up_stamp(wt, bt, prm)
        cache = getattr(prm, "_vs_pack_cache", None)
        if not cache:
            continue
        stamp = (prm._version, prm.data_ptr(), _PACK_EPOCH[0])
        for key, ent in cache.items():
            if key[0] == "lin":
                ent[0].copy_(_linear_layout(prm, key[1], key[2], key[3]))
            else:
                form, c_pad, dt = key
                pack_weight(prm, form, c_pad, _DT_TORCH[dt], out=ent[0])
            ent[1] = stamp


_TRAIN_EPOCH = [0]
_REPACK = {"params": {}, "dirty": True, "descs": None, "blocks": 0, "n": 0, "entries": []}


def repack_trainable():
    """Re-pack every registered trainable weight image with ONE launch (call after the weights changed in place)."""
    import struct
    r = _REPACK
    _recompose_trainable_ups()
    live = [p for p in r["params"].values() if getattr(p, "_vs_pack_plan", None)]
    if not live:
        return
    if r["dirty"]:
        recs, entries, blocks = [], [], 0
        for p in live:
            d0, d1 = p.shape[0], p.shape[1]
            ntaps = p[0, 0].numel()
            for (form, c_pad, dt), ent in p._vs_pack_plan.items():
                total = ent[0].numel() // (4 if dt == VS_F32 else 2)
                recs.append(struct.pack("<QQiiiiiiiiq", p.data_ptr(), ent[0].data_ptr(), d0, d1, ntaps, c_pad, form, dt, blocks, 0, total))
                entries.append((p, ent))
                blocks += (total // (4 if dt == VS_F32 else 8) + 255) // 256        # 256-thread blocks of 16-byte fragments
        raw = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8)
        r["descs"] = raw.to(live[0].device)
        r["blocks"], r["n"], r["entries"], r["dirty"] = blocks, len(recs), entries, False
    check(lib.vs_pack_weight_multi(r["descs"].data_ptr(), r["n"], r["blocks"], _stream()), "pack_weight_multi")
    for p, ent in r["entries"]:
        ent[1] = (p._version, p.data_ptr(), _TRAIN_EPOCH[0])


def repack_table():
    """The descriptor table repack_trainable() launches with right now (a device tensor, or None).  A captured graph that contains the
    re-pack launch holds a reference to it: repack_trainable() REPLACES the table when other images register, it never rewrites one in place."""
    return _REPACK["descs"]


_PACK_EPOCH = [0]


def clear_pack_cache():
    """Invalidate every cached packed weight (call after writing frozen weights through raw pointers, e.g. EMA)."""
    _PACK_EPOCH[0] += 1


def frozen_linear_layout(param, kind, c, v):
    """Frozen fc weights re-laid ONCE for contiguous access (cached on the tensor like the packed conv images; trainable weights keep the
    reference layout and the permuting kernels).  The reference flattens NCDHW (k = c*V + v) while activations are channels-last
    (ph = v*C + c), so read in k order the activation is a 2-byte gather and fc2's backward walks a weight column with a 512-byte stride.
      "cols_cl"  : (J, K) -> the K columns in channels-last order          (fc_mean / fc_std forward, backward w.r.t. x)
      "rows_cl_t": (J, K) -> (K, J) transposed, the J rows in channels-last order   (fc2 backward w.r.t. z)"""
    key = ("lin", kind, c, v)
    cache = _frozen_cache(param)
    stamp = (param._version, param.data_ptr(), _PACK_EPOCH[0])
    ent = cache.get(key)
    if ent is None:
        cache[key] = [_linear_layout(param, kind, c, v), stamp]
        return cache[key][0]
    if ent[1] != stamp:
        ent[0].copy_(_linear_layout(param, kind, c, v))         # in place: see pack_weight_cached
        ent[1] = stamp
    return ent[0]


def _linear_layout(param, kind, c, v):
    w = param.detach()
    with torch.no_grad():
        if kind == "cols_cl":
            return w.view(w.shape[0], c, v).permute(0, 2, 1).contiguous().view(w.shape[0], c * v)
        return w.view(c, v, w.shape[1]).permute(2, 1, 0).contiguous().view(w.shape[1], c * v)


def weights_changed():
    """Called by the native optimisers after they updated parameters through raw pointers (torch's version counters do
    not see that): start a new epoch and refresh all registered trainable images in one launch."""
    _TRAIN_EPOCH[0] += 1
    repack_trainable()


# ------------------------------------------------------------------------------------------------
# raw launch helpers (no autograd)
# ------------------------------------------------------------------------------------------------
# The fp64 (sum, sumsq) buffers must be zero before their producer's atomics; allocating each with torch.zeros costs
# one ~4 us fill launch per layer (~170 per step).  Instead the top-level modules open an arena per forward: ONE zeroed
# fp64 buffer sized from the previous step's use, from which the per-layer buffers are carved as views (the views keep
# the arena's storage alive for backward; a new forward gets a new arena, so nothing is ever re-zeroed under a reader).
_ARENA = {"buf": None, "off": 0, "used": 0, "need": 1 << 17, "depth": 0}


def stats_arena_begin(device):
    """Start a new zeroed arena (call at the top of a forward pass; backward keeps carving from the same one).  Inside an
    arena_scope (Joint.forward: Segmentation then VAE) the inner forwards share the scope's arena: one zero-fill launch per step."""
    drop_stale_wgrads()
    a = _ARENA
    if a["depth"] > 0 and a["buf"] is not None and a["buf"].device == torch.device(device):
        return
    a["need"] = max(a["need"], a["used"])
    n = (a["need"] + (a["need"] >> 2) + 1) // 2 * 2                  # doubles, a multiple of 16 bytes
    a["buf"] = torch.empty(n, dtype=torch.float64, device=device)
    check(lib.vs_zero_fill(a["buf"].data_ptr(), n * 8, _stream()), "zero_fill")
    a["off"] = 0
    a["used"] = 0


class arena_scope:
    """with arena_scope(device): every forward inside shares one statistics arena (one zero fill)."""

    def __init__(self, device):
        self.device = device

    def __enter__(self):
        if _ARENA["depth"] == 0:
            stats_arena_begin(self.device)
        _ARENA["depth"] += 1

    def __exit__(self, *exc):
        _ARENA["depth"] -= 1
        return False


STAT_SLOTS = lib.vs_stat_slots()      # VS_STAT_SLOTS of the loaded library: partial copies of every statistics buffer; consumers add them


def _new_stats(n, c, device, width=2):
    """A zeroed statistics buffer double[STAT_SLOTS][n][c][2] (width 2), carved from the arena; width 1 = plain zeroed scratch [1][n][c][1]."""
    a = _ARENA
    slots = STAT_SLOTS if width == 2 else 1
    cnt = slots * n * c * width
    a["used"] += cnt
    buf = a["buf"]
    if buf is None or buf.device != device or a["off"] + cnt > buf.numel():
        a["fallbacks"] = a.get("fallbacks", 0) + 1
        return torch.zeros(slots, n, c, width, dtype=torch.float64, device=device)
    out = buf[a["off"]:a["off"] + cnt].view(slots, n, c, width)
    a["off"] += cnt
    return out


def set_deterministic(on=True):
    """Switch the package to the deterministic build of the library (libvaeseg_det.so, include/vaeseg.h vs_get_deterministic): every
    per-(n,c) statistic is accumulated with commuting integer atomics, so two runs on the same inputs agree bit for bit.  Parity runs use it
    (tests/conftest.py, env VS_DETERMINISTIC=1); the throughput default is the fp64-atomic build.  A statistics buffer must be produced and
    consumed by one build: switch between passes only (captured HIP graphs keep the kernels of the build they were captured under)."""
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    lib.use_deterministic(bool(on))


def is_deterministic():
    return bool(lib.vs_get_deterministic()) if hasattr(lib, "vs_get_deterministic") else False


def stats_total(stats):
    """(n, c, 2) totals of a statistics buffer (sum over its partial copies / its fixed-point limbs) — for tests / inspection"""
    if lib.vs_stat_interleaved():
        s, n, c, w = stats.shape
        stats = stats.reshape(n, c, s, w).permute(2, 0, 1, 3)
    if is_deterministic():
        limbs = stats.contiguous().view(torch.int64).double()
        return (limbs[0] * 2.0 ** 40 + limbs[1]) + (limbs[2] * 2.0 ** -40 + limbs[3] * 2.0 ** -80)
    return stats.sum(0)


def conv_gather(x, xs, wp, bias, m_out, kind, want_stats, real_channels=None):
    n, d, h, w, c = x.shape
    if kind == VS_CONV_K2S2:
        out_shape = (n, d // 2, h // 2, w // 2, m_out)
    else:
        out_shape = (n, d, h, w, m_out)
    y = torch.empty(out_shape, dtype=x.dtype, device=x.device)
    ys = _new_stats(n, m_out, x.device) if want_stats else None
    kid = nb = fl = None
    if PROFILE is not None:
        ck, taps = min(c, 32), (27 if kind == VS_CONV_K3 else 8)
        if kind == VS_CONV_K3:
            tiles = n * ((d + 3) // 4) * ((h + 3) // 4) * ((w + 15) // 16)
        else:
            tiles = n * ((y.numel() // (n * m_out) + 255) // 256)
        rows16 = (m_out + 15) // 16 * 16
        tname = _tname(x)
        if kind == VS_CONV_K3:
            kid = _k3_kid(tname, ck, _pick_mt(rows16, tiles), geom=(n, d, h, w), m=m_out, lazy=xs is not None)
        else:
            kid = "g1_kernel<%s,%d,%d,%d,0>" % (tname, ck, kind, _pick_mt(rows16, tiles))
        cr = real_channels[0] if real_channels else c
        mr = real_channels[1] if real_channels else m_out
        vox_out = y.numel() // m_out
        nb = (x.numel() // c * cr + vox_out * mr) * _esize(x) + cr * mr * taps * _esize(x)
        fl = 2.0 * vox_out * taps * cr * mr
    with _timed(kid, nb, fl, "x%s->m%d" % (tuple(x.shape), m_out)):
        check(lib.vs_conv_gather_fwd(x.data_ptr(), _p(xs), wp.data_ptr(), _p(bias), y.data_ptr(), _p(ys), n, d, h, w, c,
                                     m_out, kind, vs_dtype(x), EPS_IN, _stream()), "conv_gather_fwd")
    return y, ys


def conv_scatter(x, xs, wp, bias, m_out):
    n, d, h, w, c = x.shape
    y = torch.empty((n, 2 * d, 2 * h, 2 * w, m_out), dtype=x.dtype, device=x.device)
    kid = nb = fl = None
    if PROFILE is not None:
        tiles = n * ((d * h * w + 255) // 256)
        rows16 = (8 * m_out + 15) // 16 * 16
        kid = "g1_kernel<%s,%d,2,%d,2>" % (_tname(x), min(c, 32), _pick_mt(rows16, tiles))
        nb = (x.numel() + y.numel()) * _esize(x) + 8 * c * m_out * _esize(x)
        fl = 2.0 * (x.numel() // c) * 8 * c * m_out
    with _timed(kid, nb, fl, "x%s->m%d" % (tuple(x.shape), m_out)):
        check(lib.vs_conv_scatter_fwd(x.data_ptr(), _p(xs), wp.data_ptr(), _p(bias), y.data_ptr(), n, d, h, w, c, m_out,
                                      vs_dtype(x), EPS_IN, _stream()), "conv_scatter_fwd")
    return y


# An activation that feeds both the next encoder level (a lazy conv) and an additive U-Net skip (joint_model.py:380,382) receives two
# gradients.  The skip's (Materialize.backward, which runs first: the decoder is differentiated before the encoder) is parked here, keyed
# by the raw tensor, and summed in by the apply pass of the conv's backward (vs_instnorm_relu_bwd_apply_add) instead of by an ATen add
# launch of autograd's.  A gradient nobody collected by the end of the pass is an error (it would be lost), checked by an engine callback.
_PENDING = {"grads": {}, "callback": False}


def _park_gradient(x, g):
    _PENDING["grads"][(x.data_ptr(), tuple(x.shape))] = g
    if not _PENDING["callback"]:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_pending_done)
            _PENDING["callback"] = True
        except RuntimeError:
            pass


def _collect_gradient(x):
    return _PENDING["grads"].pop((x.data_ptr(), tuple(x.shape)), None)


def _pending_done():
    _PENDING["callback"] = False
    left = len(_PENDING["grads"])
    _PENDING["grads"].clear()
    if left:
        raise RuntimeError("%d parked skip gradient(s) were never collected by the conv that shares their input: gradients lost" % left)


# Deferred IN-backward apply (the 8-channel full-resolution layers, igemm_k3t.h FA): where a lazy activation x is produced by an 8 -> 8 3x3x3
# conv and consumed by exactly one 3x3x3 conv (the modules mark x: mark_defer_apply), the consumer's backward hands over the UN-applied
# gradient g = dL/da together with (x, stats, sums) through this registry, and the producer's backward-data kernel applies
# rstd * (g*mask - m1 - xhat*m2) while it stages its input — the standalone apply launch (3 tensor passes at 96^3) disappears.
# An entry nobody took by the end of the pass means a consumer treated an un-applied gradient as applied: that is an error, not a fallback.
FUSE_APPLY = os.environ.get("VS_FUSE_APPLY", "1") != "0"
EPILOGUE_APPLY = os.environ.get("VS_EPILOGUE_APPLY", "1") != "0"      # A/B switch of k3b_kernel<..., EA> (the library reads the same variable)
EPILOGUE_APPLY_S2 = os.environ.get("VS_EPILOGUE_APPLY_S2", "1") != "0"      # ... of g1_kernel's epilogue apply alone (the stride-2 / transposed backward-data launches)
# channels of the activations whose producer has a fused-apply kernel: 8 / 16 (k3t, single-chunk k3b: the 96^3 / 48^3 levels).  The 32-channel
# form (k3b<32,...,FA>, the 24^3 / 12^3 levels) measured slower twice (round 4: 2.519 -> 2.566 ms per step, profiles/r04_ab_fused_apply_32ch.json)
# and left the library in round 5; so did 16-channel half stages of the same layers (two waves per SIMD, 19 launches fewer, +30..+43 us per step:
# profiles/r05_ab_fused_apply_half_stages.json).  The library has the last word (vs_conv_k3_fused_apply_supported).
_FA_CHANNELS = (8, 16)
_FA_CHANNELS_F32 = (8, 16)       # 16: k3x_kernel<8, 16, .., FA> (the 16 -> 16 layers of the 48^3 level; the library has the last word)
FUSE_APPLY_F32 = os.environ.get("VS_FUSE_APPLY_F32", "1") != "0"      # A/B switch of the parity mode's fused apply
_LAZY_APPLY = {"grads": {}, "callback": False}


def mark_defer_apply(x, producer):
    """x: the raw output of `producer` (an nn.Conv3d holder run by ConvK3, 3x3x3) about to be consumed, exactly once, by a conv op that honours
    the mark (ConvK3, ConvK3Softmax[CL], ConvK2S2, ConvT2S2).  Marked when the producer's backward-data launch has a fused-apply kernel
    (vs_conv_k3_fused_apply_supported: the single-chunk layers of the full- and half-resolution levels)."""
    # fp32 parity mode: the 8-channel full-resolution layers only (k3xt_kernel<..., FA>, csrc/igemm_k3x.h; round 5)
    chans = _FA_CHANNELS if x.dtype != torch.float32 else (_FA_CHANNELS_F32 if FUSE_APPLY_F32 else ())
    if FUSE_APPLY and x.shape[-1] in chans and tuple(producer.weight.shape[2:]) == (3, 3, 3):
        x._vs_defer_apply = True
    return x


def _fa_supported(gy, x, lazy_input):
    n, d, h, w, c = gy.shape
    return bool(lib.vs_conv_k3_fused_apply_supported(n, d, h, w, c, x.shape[-1], 1 if lazy_input else 0, vs_dtype(gy)))


def _defer_register(g, x, xs, sums):
    # g itself is kept in the entry: the key is its address, and a live reference is what guarantees that address cannot be recycled for another
    # gradient of the same shape before the producing conv's backward takes the entry
    _LAZY_APPLY["grads"][(g.data_ptr(), tuple(g.shape))] = (x, xs, sums, g)
    if not _LAZY_APPLY["callback"]:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_lazy_apply_done)
            _LAZY_APPLY["callback"] = True
        except RuntimeError:
            pass


def _take_lazy(g):
    ent = _LAZY_APPLY["grads"].pop((g.data_ptr(), tuple(g.shape)), None) if _LAZY_APPLY["grads"] else None
    return None if ent is None else ent[:3]


def _lazy_apply_done():
    _LAZY_APPLY["callback"] = False
    left = len(_LAZY_APPLY["grads"])
    _LAZY_APPLY["grads"].clear()
    if left:
        raise RuntimeError("%d un-applied gradient(s) (deferred InstanceNorm-backward apply) were not taken by the producing conv: gradients wrong" % left)


def apply_lazy(g, lazy):
    """the standalone apply of a deferred gradient (a producer that cannot fuse it)"""
    x, xs, sums = lazy
    n, c = x.shape[0], x.shape[-1]
    check(lib.vs_instnorm_relu_bwd_apply(g.data_ptr(), x.data_ptr(), xs.data_ptr(), sums.data_ptr(), g.data_ptr(), n, x.numel() // (n * c), c,
                                         vs_dtype(x), EPS_IN, _stream()), "instnorm_relu_bwd_apply")
    return g


def _ea_sync(n, device):
    """the arrival counters of one epilogue-apply launch: 8 shards of 128 bytes per sample, 128-byte aligned, zeroed with the statistics arena"""
    cnt = 128 * n
    buf = _new_stats(1, cnt + 16, device, width=1).view(-1)
    off = ((-buf.data_ptr()) % 128) // 8
    return buf[off:off + cnt]


def conv_bwd_data_lazy(gy, wpb, x, xs, kind, scatter=False, real_channels=None, defer=False, lazy=None, want_dx=False, fuse_wgrad=None):
    """Gradient w.r.t. the raw tensor x of a lazy activation a = relu(instnorm(x)) that fed a conv:
    g = conv-backward-data(gy) with the InstanceNorm+ReLU-backward sums accumulated in the same kernel's epilogue,
    then the (in-place) apply pass.  kind / scatter select the backward-data form of the forward conv.
    lazy = (act_x, act_stats, act_sums): gy itself is an un-applied gradient (see _LAZY_APPLY) — its apply is fused into this launch's
    staging and the call returns (g, applied gy or None [want_dx]).  defer: leave g un-applied and register it for x's producer.
    fuse_wgrad = (weight, m_real, c_real): the launch also forms the layer's weight gradient (see _wgrad_fusable) and the call returns (g, gw)."""
    n, c = x.shape[0], x.shape[-1]
    sums = _new_stats(n, c, x.device)
    g = torch.empty_like(x)
    gn, gd, gh, gw, gc = gy.shape
    dt = vs_dtype(x)
    if fuse_wgrad is not None:
        weight, m_real, c_real = fuse_wgrad
        ax, axs, asums = lazy if lazy is not None else (None, None, None)
        nslabs = lib.vs_conv_k3_bwd_data_wgrad_slabs(gn, gd, gh, gw)
        slabs = torch.empty(nslabs * 1728, dtype=torch.float32, device=x.device)
        kid = nb = fl = None
        if PROFILE is not None:
            kid = "k3tw_kernel<%s,%s>" % (_tname(x), "true" if lazy is not None else "false")
            nb = ((3 if lazy is not None else 2) * gy.numel() + 2 * g.numel()) * _esize(x)
            fl = 2.0 * 2.0 * (g.numel() // c) * 27 * gc * c
        with _timed(kid, nb, fl, "bwd+wgrad gy%s->m%d" % (tuple(gy.shape), c)):
            check(lib.vs_conv_k3_bwd_data_wgrad(gy.data_ptr(), _p(ax), _p(axs), _p(asums), wpb.data_ptr(), g.data_ptr(), x.data_ptr(), xs.data_ptr(),
                                                sums.data_ptr(), slabs.data_ptr(), gn, gd, gh, gw, gc, c, dt, EPS_IN, _stream()), "conv_k3_bwd_data_wgrad")
        gw_t = _group_submit_slabs(weight, slabs, nslabs, m_real, c_real, (gy, x, xs))
        if defer:
            _defer_register(g, x, xs, sums)
        else:
            _apply_in_place(g, x, xs, sums)
        return g, gw_t
    if lazy is not None:
        ax, axs, asums = lazy
        dx = torch.empty_like(gy) if want_dx else None
        kid = nb = fl = None
        if PROFILE is not None:
            tiles = gn * ((gd + 3) // 4) * ((gh + 3) // 4) * ((gw + 15) // 16)
            kid = _k3_kid(_tname(x), min(gc, 32), _pick_mt((c + 15) // 16 * 16, tiles), sums=True, geom=(gn, gd, gh, gw), m=c) + "+apply"
            nb = (2 * gy.numel() + 2 * g.numel() + (gy.numel() if want_dx else 0)) * _esize(x) + gc * c * 27 * _esize(x)
            fl = 2.0 * (g.numel() // c) * 27 * gc * c
        with _timed(kid, nb, fl, "bwd+apply gy%s->m%d" % (tuple(gy.shape), c)):
            check(lib.vs_conv_k3_bwd_data_fused_apply(gy.data_ptr(), ax.data_ptr(), axs.data_ptr(), asums.data_ptr(), wpb.data_ptr(), g.data_ptr(),
                                                      x.data_ptr(), xs.data_ptr(), sums.data_ptr(), _p(dx), gn, gd, gh, gw, gc, c, dt, EPS_IN,
                                                      _stream()), "conv_k3_bwd_data_fused_apply")
        if defer:
            _defer_register(g, x, xs, sums)
        else:
            _apply_in_place(g, x, xs, sums)
        return g, dx
    if (not scatter and kind == VS_CONV_K3 and not defer and EPILOGUE_APPLY and (x.data_ptr(), tuple(x.shape)) not in _PENDING["grads"]
            and lib.vs_conv_k3_bwd_data_applied_supported(gn, gd, gh, gw, gc, c, dt)):
        # the 24^3 / 12^3 levels: the backward-data launch applies the InstanceNorm+ReLU backward to its own outputs (csrc/igemm_k3b.h EA): no apply launch
        sync = _ea_sync(gn, x.device)
        kid = nb = fl = None
        if PROFILE is not None:
            tiles = gn * ((gd + 3) // 4) * ((gh + 3) // 4) * ((gw + 15) // 16)
            kid = _k3_kid(_tname(x), min(gc, 32), 16, sums=True, geom=(gn, gd, gh, gw), m=c) + "+ea"
            nb = (gy.numel() + 2 * g.numel()) * _esize(x) + gc * c * 27 * _esize(x)
            fl = 2.0 * (g.numel() // c) * 27 * gc * c
        with _timed(kid, nb, fl, "bwd+ea gy%s->m%d" % (tuple(gy.shape), c)):
            check(lib.vs_conv_k3_bwd_data_applied(gy.data_ptr(), wpb.data_ptr(), g.data_ptr(), x.data_ptr(), xs.data_ptr(), sums.data_ptr(), sync.data_ptr(),
                                                  _chain_fault_word(x.device).data_ptr(), gn, gd, gh, gw, gc, c, dt, EPS_IN, _stream()), "conv_k3_bwd_data_applied")
        return g
    if (kind != VS_CONV_K3 and not defer and EPILOGUE_APPLY and EPILOGUE_APPLY_S2 and lib.vs_conv_s2_bwd_data_applied_supported(gn, gd, gh, gw, gc, c, 1 if scatter else 0, dt)):
        # the stride-2 / transposed launches of the <= 48^3 levels: g1_kernel applies the InstanceNorm+ReLU backward to its own outputs (csrc/igemm.h), the
        # skip's parked gradient of the same tensor summed in as the standalone apply would
        sync = _ea_sync(gn, x.device)
        add = _collect_gradient(x)
        kid = nb = fl = None
        if PROFILE is not None:
            tiles = gn * ((gd * gh * gw + 255) // 256) if scatter else n * ((g.numel() // (n * c) + 255) // 256)
            rows = (8 * c + 15) // 16 * 16 if scatter else (c + 15) // 16 * 16
            kid = "g1_kernel<%s,%d,%d,%d,%d>+ea" % (_tname(x), min(gc, 32), 2 if scatter else kind, _pick_mt(rows, tiles), 2 if scatter else 0)
            nb = (gy.numel() + (2 if add is None else 3) * g.numel()) * _esize(x) + 8 * gc * c * _esize(x)
            fl = 2.0 * ((gy.numel() // gc) if scatter else (g.numel() // c)) * 8 * gc * c
        with _timed(kid, nb, fl, "bwd+ea gy%s->m%d" % (tuple(gy.shape), c)):
            check(lib.vs_conv_s2_bwd_data_applied(gy.data_ptr(), wpb.data_ptr(), g.data_ptr(), x.data_ptr(), xs.data_ptr(), sums.data_ptr(), _p(add), sync.data_ptr(),
                                                  _chain_fault_word(x.device).data_ptr(), gn, gd, gh, gw, gc, c, 1 if scatter else 0, dt, EPS_IN, _stream()),
                  "conv_s2_bwd_data_applied")
        return g
    if scatter:
        kid = nb = fl = None
        if PROFILE is not None:
            tiles = gn * ((gd * gh * gw + 255) // 256)
            kid = "g1_kernel<%s,%d,2,%d,2>" % (_tname(x), min(gc, 32),
                                               _pick_mt((8 * c + 15) // 16 * 16, tiles))
            nb = (gy.numel() + 2 * g.numel()) * _esize(x) + 8 * gc * c * _esize(x)
            fl = 2.0 * (gy.numel() // gc) * 8 * gc * c
        with _timed(kid, nb, fl, "bwd gy%s->m%d" % (tuple(gy.shape), c)):
            check(lib.vs_conv_scatter_bwd_data(gy.data_ptr(), wpb.data_ptr(), g.data_ptr(), x.data_ptr(), xs.data_ptr(),
                                               sums.data_ptr(), gn, gd, gh, gw, gc, c, dt, EPS_IN, _stream()), "conv_scatter_bwd_data")
    else:
        kid = nb = fl = None
        if PROFILE is not None:
            taps = 27 if kind == VS_CONV_K3 else 8
            if kind == VS_CONV_K3:
                tiles = gn * ((gd + 3) // 4) * ((gh + 3) // 4) * ((gw + 15) // 16)
            else:
                tiles = n * ((g.numel() // (n * c) + 255) // 256)
            tname = _tname(x)
            if kind == VS_CONV_K3:
                kid = _k3_kid(tname, min(gc, 32), _pick_mt((c + 15) // 16 * 16, tiles), sums=True, geom=(gn, gd, gh, gw), m=c)
            else:
                kid = "g1_kernel<%s,%d,%d,%d,0>" % (tname, min(gc, 32), kind, _pick_mt((c + 15) // 16 * 16, tiles))
            cr = real_channels[0] if real_channels else gc
            mr = real_channels[1] if real_channels else c
            vox_out = g.numel() // c
            nb = (gy.numel() // gc * cr + 2 * vox_out * mr) * _esize(x) + cr * mr * taps * _esize(x)
            fl = 2.0 * vox_out * taps * cr * mr
        with _timed(kid, nb, fl, "bwd gy%s->m%d" % (tuple(gy.shape), c)):
            check(lib.vs_conv_gather_bwd_data(gy.data_ptr(), wpb.data_ptr(), g.data_ptr(), x.data_ptr(), xs.data_ptr(),
                                              sums.data_ptr(), gn, gd, gh, gw, gc, c, kind, dt, EPS_IN, _stream()), "conv_gather_bwd_data")
    if defer:
        _defer_register(g, x, xs, sums)
        return g
    return _apply_in_place(g, x, xs, sums)


def _apply_in_place(g, x, xs, sums):
    n, c = x.shape[0], x.shape[-1]
    voxels = x.numel() // (n * c)
    tname = _tname(x)
    add = _collect_gradient(x)                 # the skip's gradient of the same tensor, if one was parked
    with _timed("in_relu_bwd_apply_kernel<%s>" % tname, (3 if add is None else 4) * x.numel() * _esize(x), 6.0 * x.numel(), str(tuple(x.shape))):
        check(lib.vs_instnorm_relu_bwd_apply_add(g.data_ptr(), x.data_ptr(), xs.data_ptr(), sums.data_ptr(), _p(add), g.data_ptr(), n, voxels,
                                                 c, vs_dtype(x), EPS_IN, _stream()), "instnorm_relu_bwd_apply")
    return g


def conv_wgrad(p, ps, q, qs, m_real, c_real, kind, out_shape, out_ptr=None):
    """dW (fp32, reference layout [m][c][taps]) ; the voxel loop runs over p's grid.  With out_ptr the result goes to that
    preallocated fp32 buffer of out_shape (deferred side-stream launches) and nothing is returned."""
    n, dp, hp, wp_, m_ch = p.shape
    c_ch = q.shape[-1]
    nbytes = lib.vs_conv_wgrad_workspace_bytes(n, dp, hp, wp_, m_ch, c_ch, kind)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=p.device)
    dw = torch.empty(out_shape, dtype=torch.float32, device=p.device) if out_ptr is None else None
    dw_ptr = dw.data_ptr() if out_ptr is None else out_ptr
    kid = nb = fl = None
    if PROFILE is not None:
        taps = 27 if kind == VS_CONV_K3 else 8
        cb, kk = 16 if c_ch >= 16 else 8, 0 if kind == VS_CONV_K3 else 1
        kid = "g3_kernel<float,%d,%d>" % (cb, kk) if p.dtype == torch.float32 else "g3b_kernel<%d,%d>" % (cb, kk)
        nb = (p.numel() // m_ch * m_real + q.numel() // c_ch * c_real) * _esize(p) + m_real * c_real * taps * 4
        fl = 2.0 * (p.numel() // m_ch) * taps * m_real * c_real
    with _timed(kid, nb, fl, "p%s q%s" % (tuple(p.shape), tuple(q.shape))):
        check(lib.vs_conv_wgrad(p.data_ptr(), _p(ps), q.data_ptr(), _p(qs), dw_ptr, ws.data_ptr(), nbytes, n, dp, hp,
                                wp_, m_ch, c_ch, m_real, c_real, kind, vs_dtype(p), EPS_IN, _stream()), "conv_wgrad")
    return dw


def bias_grad(g, c_real, out_ptr=None):
    c_ch = g.shape[-1]
    rows = g.numel() // c_ch
    db = torch.empty(c_real, dtype=torch.float32, device=g.device) if out_ptr is None else None
    check(lib.vs_bias_grad(g.data_ptr(), db.data_ptr() if out_ptr is None else out_ptr, rows, c_ch, c_real, vs_dtype(g), _stream()),
          "bias_grad")
    return db


# ------------------------------------------------------------------------------------------------
# grouped weight gradients (default): one descriptor per layer during backward, a handful of launches at its end
# ------------------------------------------------------------------------------------------------
# A step has 34 weight-gradient kernels, 34 slab reductions and 9 bias gradients (+ their zero fills); most belong to layers
# with a few dozen workgroups and cost a launch round trip each (4.7 us minimum, 0.85 ms per 96^3 step together).  Nothing in
# backward reads them, so they are collected as vs_wgrad_desc records and issued by vs_conv_wgrad_multi when the pass ends
# (autograd-engine callback): layers of one kernel instantiation share one grid, every reduction shares one.
import ctypes as _ct


class WgradDesc(_ct.Structure):
    """vs_wgrad_desc (include/vaeseg.h)"""
    _fields_ = [("p", _ct.c_void_p), ("p_stats", _ct.c_void_p), ("q", _ct.c_void_p), ("q_stats", _ct.c_void_p),
                ("dw", _ct.c_void_p), ("bias_g", _ct.c_void_p), ("db", _ct.c_void_p), ("bias_rows", _ct.c_longlong),
                ("bias_c_ch", _ct.c_int), ("bias_c_real", _ct.c_int),
                ("n", _ct.c_int), ("dp", _ct.c_int), ("hp", _ct.c_int), ("wp", _ct.c_int), ("m_ch", _ct.c_int), ("c_ch", _ct.c_int),
                ("m_real", _ct.c_int), ("c_real", _ct.c_int), ("kind", _ct.c_int), ("reserved_", _ct.c_int)]


_GROUP = {"enabled": os.environ.get("VS_WGRAD_GROUP", "1") != "0", "descs": [], "keep": [], "callback": False, "dtype": None,
          "bytes": 0.0, "flops": 0.0, "split": None, "slots": {}}
# "slots": id(weight) -> (gw, gb) handed to autograd by the FIRST use of that weight in the current backward pass.  A weight used several
# times (the VAE inside Embed runs three times per forward, joint_model.py:469-500) queues one descriptor per use, all with the first use's
# destination: vs_conv_wgrad_multi sums the uses, and the later uses return None to autograd (nothing left to accumulate).


def _issue_wgrads(entries):
    descs = [x[0] for x in entries]
    arr = (WgradDesc * len(descs))(*descs)
    dt = vs_of(_GROUP["dtype"])
    dev = torch.device("cuda", torch.cuda.current_device())
    nbytes = lib.vs_conv_wgrad_multi_workspace_bytes(_ct.addressof(arr), len(descs), dt)
    if nbytes == 0:
        _GROUP["descs"], _GROUP["keep"] = [], []
        raise _lib.VaesegError("vs_conv_wgrad_multi: unsupported layer in the deferred weight-gradient list")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    nb, fl = sum(x[2] for x in entries), sum(x[3] for x in entries)
    with _timed("wgrad_multi(%d layers)" % len(descs), nb, fl):
        check(lib.vs_conv_wgrad_multi(_ct.addressof(arr), len(descs), ws.data_ptr(), nbytes, dt, EPS_IN, _stream()), "conv_wgrad_multi")
    return ws


def set_wgrad_grouping(enabled=True):
    """Defer weight/bias gradients to the end of backward and issue them as grouped launches (default on)."""
    flush_wgrads()
    _GROUP["enabled"] = bool(enabled)


def set_wgrad_split(first=None):
    """Two-phase issue for the data-parallel step (ddp.FlatGradSync): with `first` = a predicate on a parameter, the launches at the end
    of backward cover only the layers whose weight satisfies it; the caller issues the rest with flush_wgrads() after it has started the
    all-reduce of the first bucket, so that the exchange runs under the remaining weight-gradient kernels.  None: one phase."""
    flush_wgrads()
    _GROUP["split"] = first


def drop_stale_wgrads():
    """Forget deferred descriptors left behind by a backward pass that raised before its end-of-pass callback ran (their raw pointers
    may be recycled by now).  Called at the top of every forward (stats_arena_begin)."""
    g = _GROUP
    if g["descs"] and not g["callback"]:
        return                                  # a caller-managed second phase (set_wgrad_split) is pending: not stale
    if g["descs"] or g["callback"]:
        g["descs"], g["keep"], g["callback"], g["bytes"], g["flops"] = [], [], False, 0.0, 0.0
    g["slots"] = {}
    _UP_JOBS.clear()
    # the same for gradients parked / handed over un-applied by a pass that died: their addresses may be recycled by now, and a later
    # backward must never mistake a fresh tensor at such an address for one of them
    for reg in (_PENDING, _LAZY_APPLY):
        reg["grads"].clear()
        reg["callback"] = False


def _group_submit(weight, keep, wgrad_args, bias_args, gw, gb, up_co=0):
    p, ps, q, qs, m_real, c_real, kind = wgrad_args
    g = _GROUP
    if g["descs"] and g["dtype"] != p.dtype:
        flush_wgrads()
    g["dtype"] = p.dtype
    n, dp, hp, wp_, m_ch = p.shape
    gw_ptr = gw if isinstance(gw, int) else gw.data_ptr()
    if kind == VS_CONV_UP:                   # p is the FINE gradient (n, 2dp, 2hp, 2wp, Co): the descriptor's grid is the coarse one, its rows the 8 Co view channels
        dp, hp, wp_, m_ch = dp // 2, hp // 2, wp_ // 2, 8 * up_co
    d = WgradDesc(p.data_ptr(), _p(ps), q.data_ptr(), _p(qs), gw_ptr, None, None, 0, 0, 0,
                  n, dp, hp, wp_, m_ch, q.shape[-1], m_real, c_real, kind, up_co)
    if bias_args is not None:
        bg = bias_args[0]
        gb_ptr = gb if isinstance(gb, int) else gb.data_ptr()
        d.bias_g, d.db, d.bias_rows, d.bias_c_ch, d.bias_c_real = bg.data_ptr(), gb_ptr, bg.numel() // bg.shape[-1], bg.shape[-1], bias_args[1]
        g["keep"].append(bg)
    taps = 8 if kind == VS_CONV_K2S2 else 27
    nb = (p.numel() + q.numel() // q.shape[-1] * c_real) * _esize(p) + m_real * c_real * taps * 4 if kind == VS_CONV_UP else \
        (p.numel() // m_ch * m_real + q.numel() // q.shape[-1] * c_real) * _esize(p) + m_real * c_real * taps * 4
    fl = 2.0 * (q.numel() // q.shape[-1]) * taps * m_real * c_real if kind == VS_CONV_UP else 2.0 * (p.numel() // m_ch) * taps * m_real * c_real
    first = g["split"] is None or bool(g["split"](weight))
    voxels = n * dp * hp * wp_
    g["descs"].append((d, first, nb, fl, voxels))
    g["keep"].extend(t for t in keep if t is not None)
    if not g["callback"]:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_group_backward_done)
            g["callback"] = True
        except RuntimeError:
            flush_wgrads()                      # not inside a backward pass: nothing will call back


def _group_backward_done():
    _USE_EPOCH[0] += 1                          # a backward pass ended: the use counts of the next forward start over (_count_use)
    _GROUP["callback"] = False
    flush_wgrads(first_only=_GROUP["split"] is not None)
    _GROUP["slots"] = {}


def pending_wgrads():
    return len(_GROUP["descs"])


def flush_wgrads(first_only=False):
    """Issue the deferred weight/bias gradients on the current stream (vs_conv_wgrad_multi): all of them, or — under
    set_wgrad_split — only the first phase."""
    g = _GROUP
    if not g["descs"]:
        return
    now = [e for e in g["descs"] if e[1]] if first_only else g["descs"]
    later = [e for e in g["descs"] if not e[1]] if first_only else []
    g["descs"] = later
    if not now:
        return
    _issue_wgrads(now)
    if not later:
        _run_up_jobs()                          # parameter-space chain rule of the composed Up heads: reads the dWeff the launch above reduced
        g["keep"] = []                          # launched on the current stream: the allocator may recycle the inputs now


# Weight gradient fused into the backward-data launch (csrc/igemm_k3tw.h; round 5): the 8 -> 8 3x3x3 layers whose backward-data kernel already holds both
# operands of dW.  Taken when the weight is used exactly ONCE in this pass (counted at forward time; _USE_EPOCH moves on when a backward pass ends) and its
# gradient can be deferred like every grouped one; the slabs the launch writes join the grouped reduction as a VS_WGRAD_SLABS descriptor.
FUSE_WGRAD = os.environ.get("VS_FUSE_WGRAD", "1") != "0"
VS_WGRAD_SLABS = 16
_USE_EPOCH = [0]


def _count_use(weight):
    u = getattr(weight, "_vs_use", None)
    if u is None or u[0] != _USE_EPOCH[0]:
        weight._vs_use = [_USE_EPOCH[0], 1]
    else:
        u[1] += 1


def _wgrad_fusable(gy, x, weight):
    if not (FUSE_WGRAD and _GROUP["enabled"] and gy.dtype != torch.float32 and gy.shape[-1] == 8 and x.shape[-1] == 8):
        return False
    if getattr(weight, "_vs_use", None) != [_USE_EPOCH[0], 1] or _GROUP["slots"].get(id(weight)) is not None:
        return False
    if not (weight.is_leaf and weight.grad is None and not _has_hooks(weight) and not torch.is_grad_enabled()):
        return False                            # the conditions under which _side_grads defers a gradient
    n, d, h, w, _ = gy.shape
    return bool(lib.vs_conv_k3_bwd_data_wgrad_supported(n, d, h, w, 8, 8, vs_dtype(gy)))


def _group_submit_slabs(weight, slabs, nslabs, m_real, c_real, keep, bias=None):
    """register the slabs of a fused launch with the pass's grouped weight gradients; -> the (still unwritten) gradient tensor handed to autograd
    (bias = (parameter, partial sums double [nslabs][m_real]): -> (gw, gb))"""
    g = _GROUP
    if g["descs"] and g["dtype"] != keep[0].dtype:
        flush_wgrads()
    g["dtype"] = keep[0].dtype
    gw = _grad_slot(weight, weight.shape)
    gb = None
    d = WgradDesc(slabs.data_ptr(), None, None, None, gw.data_ptr(), None, None, 0, 0, 0, nslabs, 0, 0, 0, 8, 8, m_real, c_real, VS_WGRAD_SLABS, 0)
    if bias is not None:
        gb = _grad_slot(bias[0], (m_real,))
        d.bias_g, d.db, d.bias_rows, d.bias_c_ch, d.bias_c_real = bias[1].data_ptr(), gb.data_ptr(), nslabs, m_real, m_real
        g["keep"].append(bias[1])
    # third field: this destination belongs to a SLAB descriptor — vs_conv_wgrad_multi does not let a regular descriptor share it (the forward-time use
    # count can read 1 with two live uses when an unrelated backward pass ended between two forwards of the same weight); _side_grads gives a later use its own
    g["slots"][id(weight)] = (gw.data_ptr(), None if gb is None else gb.data_ptr(), True)
    first = g["split"] is None or bool(g["split"](weight))
    g["descs"].append((d, first, float(slabs.numel() * 4), 0.0, 0))
    g["keep"].append(slabs)
    if not g["callback"]:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_group_backward_done)
            g["callback"] = True
        except RuntimeError:
            flush_wgrads()
    return gw if bias is None else (gw, gb)


def _grad_slot(param, shape):
    """Where a parameter's gradient is written: its slice of the data-parallel flat bucket when ddp.FlatGradSync registered one
    (the all-reduce then needs no gather launch; autograd adopts the view as .grad), a fresh tensor otherwise."""
    view = getattr(param, "_vs_grad_view", None)
    if view is not None and param.grad is None:
        return view.detach()                    # a fresh alias: AccumulateGrad adopts a gradient only if nobody else holds the tensor object
    return torch.empty(shape, dtype=torch.float32, device=param.device)


def _has_hooks(t):
    return bool(getattr(t, "_backward_hooks", None)) or bool(getattr(t, "_post_accumulate_grad_hooks", None))


def _side_grads(weight, keep, wgrad_args, bias_args, bias=None):
    """Allocate dW (and db) now; their kernels are deferred — grouped at the end of backward (default) — or run at once when grouping is
    off, the parameter already holds a gradient to accumulate into, carries hooks, or grad mode is on (autograd would then clone the
    still-unwritten tensor); -> (gw, gb)."""
    grouping = _GROUP["enabled"]
    slot = _GROUP["slots"].get(id(weight)) if grouping and weight.is_leaf else None
    if slot is not None and len(slot) > 2 and slot[2]:
        slot = None                             # the first use went into its backward-data launch (slabs): this use gets a destination of its own, autograd adds the two
    if slot is not None:
        # a later use of the same weight in this pass: one more descriptor with the first use's destination, nothing returned to autograd
        _group_submit(weight, keep, wgrad_args, bias_args if slot[1] is not None else None, slot[0], slot[1])
        return None, None
    gw = _grad_slot(weight, weight.shape)
    gb = None
    if bias_args is not None:
        gb = (_grad_slot(bias, (bias_args[1],)) if bias is not None else
              torch.empty(bias_args[1], dtype=torch.float32, device=keep[0].device))
    # deferring hands autograd a still-unwritten tensor: only sound when its consumer is the parameter's AccumulateGrad (a leaf), which keeps it
    # untouched until the pass ends; a non-leaf weight's gradient is read by the next backward node at once
    deferrable = (weight.is_leaf and weight.grad is None and not _has_hooks(weight) and not torch.is_grad_enabled()
                  and (bias is None or (bias.is_leaf and not _has_hooks(bias))))
    if grouping and deferrable:
        _GROUP["slots"][id(weight)] = (gw.data_ptr(), None if gb is None else gb.data_ptr())      # addresses, not tensors: see below
        # the descriptor holds raw pointers only: AccumulateGrad must find gw / gb unshared to adopt them as .grad without a copy
        _group_submit(weight, keep, wgrad_args, bias_args, gw, gb)
        return gw, gb
    conv_wgrad(*wgrad_args, weight.shape, out_ptr=gw.data_ptr())
    if bias_args is not None:
        bias_grad(bias_args[0], bias_args[1], out_ptr=gb.data_ptr())
    return gw, gb


def _dead_bias_grad(bias, n, device):
    """Gradient of a conv bias that feeds InstanceNorm (exactly zero): the parameter's (never written, zero) slot of
    the flat bucket when one is registered, else zeros carved from the statistics arena (no fill launch).  Later uses of the same
    bias in one pass return None (nothing to accumulate)."""
    if bias is not None and _GROUP["enabled"]:
        key = ("dead", id(bias))
        if key in _GROUP["slots"]:
            return None
        _GROUP["slots"][key] = True
        if not _GROUP["callback"]:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(_group_backward_done)
                _GROUP["callback"] = True
            except RuntimeError:
                _GROUP["slots"].pop(key, None)
    view = getattr(bias, "_vs_grad_view", None) if bias is not None else None
    if view is not None and bias.grad is None:
        return view.detach()
    return _new_stats(1, (n + 1) // 2, device, width=1).view(-1).view(torch.float32)[:n]


def in_relu_bwd(g, x, xs, inplace=True):
    """gradient w.r.t. the raw tensor x of a = relu(instnorm(x)), given g = dL/da (same layout)."""
    if xs is None:
        return g
    n, c = x.shape[0], x.shape[-1]
    voxels = x.numel() // (n * c)
    sums = _new_stats(n, c, x.device)
    dt = vs_dtype(x)
    tname = _tname(x)
    with _timed("in_relu_bwd_reduce_kernel<%s>" % tname, 2 * x.numel() * _esize(x), 4.0 * x.numel(), str(tuple(x.shape))):
        check(lib.vs_instnorm_relu_bwd_reduce(g.data_ptr(), x.data_ptr(), xs.data_ptr(), sums.data_ptr(), n, voxels, c, dt,
                                              EPS_IN, _stream()), "instnorm_relu_bwd_reduce")
    gx = g if inplace else torch.empty_like(g)
    with _timed("in_relu_bwd_apply_kernel<%s>" % tname, 3 * x.numel() * _esize(x), 6.0 * x.numel(), str(tuple(x.shape))):
        check(lib.vs_instnorm_relu_bwd_apply(g.data_ptr(), x.data_ptr(), xs.data_ptr(), sums.data_ptr(), gx.data_ptr(), n,
                                             voxels, c, dt, EPS_IN, _stream()), "instnorm_relu_bwd_apply")
    return gx


def _contig(t):
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------------
# autograd Functions
# ------------------------------------------------------------------------------------------------
class ConvK3(torch.autograd.Function):
    """3x3x3 conv (pad 1) on a lazy input; output is lazy (raw + stats) — joint_model.py:40-48,106-108.
    The bias of a conv that feeds InstanceNorm is mathematically dead: it is neither added
    nor differentiated (its gradient is returned as zeros, the reference's is ~1e-7 noise)."""

    @staticmethod
    def forward(ctx, x, xs, weight, bias, live_bias=False):
        _require_cuda(x, weight)
        cout, cin = weight.shape[0], weight.shape[1]
        wp = pack_weight_cached(weight, VS_PACK_ROWS_D0, x.shape[-1], k3_pack_dtype(x))
        # live_bias: the general norm path (NormAct) — under BatchNorm in eval mode the bias is not cancelled by the normalisation
        ctx.live_bias = bool(live_bias) and bias is not None
        b_use = bias if ctx.live_bias else None
        if b_use is not None and cout < cpad(cout):
            b_use = torch.nn.functional.pad(b_use.detach().float(), (0, cpad(cout) - cout))      # the kernels read one bias per padded output row
        y, ys = conv_gather(x, xs, wp, b_use, cpad(cout), VS_CONV_K3, True, real_channels=(cin, cout))
        ctx.save_for_backward(x, xs, weight)
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias                   # a Parameter (long-lived leaf): only its gradient slot is looked up in backward
        ctx.defer = bool(getattr(x, "_vs_defer_apply", False)) and xs is not None     # see _LAZY_APPLY
        if ctx.needs_input_grad[2]:
            _count_use(weight)                # see _wgrad_fusable
        ctx.mark_non_differentiable(ys)
        ctx.set_materialize_grads(False)      # otherwise autograd zero-fills a gradient for the stats output every backward
        return y, ys

    @staticmethod
    def backward(ctx, gy, _gys):
        x, xs, weight = ctx.saved_tensors
        if gy is None:
            return None, None, None, None, None
        gy = _contig(gy)
        cout, cin = weight.shape[0], weight.shape[1]
        gx = gw = gb = None
        lazy = _take_lazy(gy)                   # gy is an un-applied gradient handed over by the consumer of this conv's output
        if lazy is not None and not (ctx.needs_input_grad[0] and _fa_supported(gy, x, xs is not None)):
            gy, lazy = apply_lazy(gy, lazy), None     # no fused-apply kernel for this launch / no input gradient wanted: apply now
        if ctx.needs_input_grad[0]:
            wpb = pack_weight_cached(weight, VS_PACK_ROWS_D1_FLIP, gy.shape[-1], k3_pack_dtype(gy))
            want_gb0 = ctx.has_bias and ctx.needs_input_grad[3] and ctx.live_bias
            if xs is not None and ctx.needs_input_grad[2] and not want_gb0 and _wgrad_fusable(gy, x, weight):
                # the backward-data launch of this layer forms its weight gradient as well (nothing else reads the applied gradient)
                gx, gw = conv_bwd_data_lazy(gy, wpb, x, xs, VS_CONV_K3, real_channels=(cout, cin), defer=ctx.defer, lazy=lazy,
                                            fuse_wgrad=(weight, cout, cin))
                gb = _dead_bias_grad(ctx.bias_ref, ctx.bias_ref.shape[0], x.device) if ctx.has_bias and ctx.needs_input_grad[3] else None
                return gx, None, gw, gb, None
            if lazy is not None and xs is not None:
                gx, gy = conv_bwd_data_lazy(gy, wpb, x, xs, VS_CONV_K3, real_channels=(cout, cin), defer=ctx.defer, lazy=lazy,
                                            want_dx=ctx.needs_input_grad[2])
            elif lazy is not None:                # the conv's own input is a stored tensor (VAE.in_block on the prediction): no sums to fuse
                ax, axs, asums = lazy
                gx = torch.empty_like(x)
                dx = torch.empty_like(gy) if ctx.needs_input_grad[2] else None
                n_, d_, h_, w_, c_ = gy.shape
                check(lib.vs_conv_k3_bwd_data_fused_apply(gy.data_ptr(), ax.data_ptr(), axs.data_ptr(), asums.data_ptr(), wpb.data_ptr(), gx.data_ptr(),
                                                          None, None, None, _p(dx), n_, d_, h_, w_, c_, x.shape[-1], vs_dtype(x), EPS_IN, _stream()),
                      "conv_k3_bwd_data_fused_apply")
                gy = dx
            elif xs is not None:
                gx = conv_bwd_data_lazy(gy, wpb, x, xs, VS_CONV_K3, real_channels=(cout, cin), defer=ctx.defer)
            else:
                gx, _ = conv_gather(gy, None, wpb, None, x.shape[-1], VS_CONV_K3, False, real_channels=(cout, cin))
        want_gb = ctx.has_bias and ctx.needs_input_grad[3]
        if ctx.needs_input_grad[2]:
            gw, gb = _side_grads(weight, (gy, x, xs), (gy, None, x, xs, cout, cin, VS_CONV_K3),
                                 (gy, cout) if want_gb and ctx.live_bias else None, ctx.bias_ref if ctx.live_bias else None)
        elif want_gb and ctx.live_bias:
            gb = bias_grad(gy, cout)
        if want_gb and not ctx.live_bias:
            gb = _dead_bias_grad(ctx.bias_ref, ctx.bias_ref.shape[0], x.device)
        return gx, None, gw, gb, None


# ------------------------------------------------------------------------------------------------
# Chains (csrc/chain.h, round 6): the 3x3x3 convolutions of one DoubleConv at the small volumes of the deep levels as ONE launch each way.
# InstanceNorm3d is per (sample, channel), so consecutive layers of one sample hand their tensors over inside the launch instead of ending it.
# ------------------------------------------------------------------------------------------------
CHAIN = os.environ.get("VS_CHAIN", "1") != "0"


class ChainLayer(_ct.Structure):
    """vs_chain_layer (include/vaeseg.h)"""
    _fields_ = [("x", _ct.c_void_p), ("x_stats", _ct.c_void_p), ("w_packed", _ct.c_void_p), ("y", _ct.c_void_p), ("y_stats", _ct.c_void_p),
                ("mask_x", _ct.c_void_p), ("mask_stats", _ct.c_void_p), ("sums", _ct.c_void_p), ("c_in", _ct.c_int), ("m_out", _ct.c_int),
                ("apply", _ct.c_int), ("reserved_", _ct.c_int)]


_CHAIN_FAULT = {}


def _chain_fault_word(device):
    """the device word a chain kernel raises when one of its bounded waits gave up (never in a correct launch); read back by chain_fault()"""
    key = torch.device(device)
    t = _CHAIN_FAULT.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the chain fault word must exist before a capture starts: run the step eagerly once (GraphedStep warms up)")
        t = _CHAIN_FAULT[key] = torch.zeros(32, dtype=torch.int32, device=key)
    return t


def chain_fault():
    """Host check (synchronises): raises if any chain kernel of this process ever gave up a wait — its results, and everything after them, are invalid."""
    for dev, t in _CHAIN_FAULT.items():
        if int(t[0].item()) != 0:
            raise _lib.VaesegError("a chain kernel on %s gave up a bounded wait (workgroups of one sample not co-resident): results invalid; set VS_CHAIN=0" % (dev,))


def device_is_shared(shared=True):
    """The in-kernel hand-offs (csrc/chain.h: chain kernels, epilogue apply) need every workgroup of a sample resident while its peers wait for it — true for a
    launch that has the device to itself (one process per GPU, the step's one stream: the deployment model), NOT when another process or stream runs kernels of the
    same kind on the same GPU at the same time: two launches that each hold part of the chip and wait for workgroups the other one keeps out never finish their
    waits (the bounded spins give up and raise the fault word: wrong numbers, reported by chain_fault()).  A host that shares a GPU between ranks (the test-suite's
    gloo ranks on one card, bench.py --share-gpu) calls this first: both forms are switched off in the library (vs_set_config) and here."""
    global CHAIN, EPILOGUE_APPLY
    on = 0 if shared else 1
    set_config(chain=on, epilogue_apply=on)
    CHAIN = EPILOGUE_APPLY = not shared


def _chain_sync(n, device):
    cnt = lib.vs_conv_k3_chain_sync_bytes(n) // 8
    buf = _new_stats(1, cnt + 16, device, width=1).view(-1)
    off = ((-buf.data_ptr()) % 128) // 8
    return buf[off:off + cnt]


def chain_ok(x, xs, convs):
    """may these consecutive 3x3x3 convs (nn.Conv3d holders; InstanceNorm + ReLU between them) on the lazy input (x, xs) run as one chain?"""
    if not CHAIN or len(convs) < 2 or len(convs) > 3 or x.dtype not in KERNEL_DTYPES:
        return False
    if getattr(x, "_vs_defer_apply", False):
        return False                            # x's producer wants its gradient un-applied (the 8 / 16-channel layers): not these volumes
    if any(tuple(c.weight.shape[2:]) != (3, 3, 3) for c in convs):
        return False
    n, d, h, w, c = x.shape
    cmax = max([c] + [cpad(cv.weight.shape[0]) for cv in convs])
    return bool(lib.vs_conv_k3_chain_supported(n, d, h, w, cmax, vs_dtype(x)))


class ConvK3Chain(torch.autograd.Function):
    """2 or 3 consecutive [3x3x3 conv -> InstanceNorm3d -> ReLU] triples of a DoubleConv (joint_model.py:35-52) on a lazy input; the output is lazy.
    forward: ONE launch (vs_conv_k3_chain); backward: ONE launch for the backward-data convs and the InstanceNorm+ReLU backward applies between them,
    the weight gradients join the pass's grouped launch as those of ConvK3 do.  The biases are dead, as in ConvK3."""

    @staticmethod
    def forward(ctx, x, xs, *params):
        _require_cuda(x)
        nl = len(params) // 2
        weights, biases = params[0::2], params[1::2]
        n, d, h, w, _ = x.shape
        layers = (ChainLayer * nl)()
        cur, curs, outs, keep = x, xs, [], []
        nb = fl = 0.0
        for l in range(nl):
            wt = weights[l]
            m = cpad(wt.shape[0])
            wp = pack_weight_cached(wt, VS_PACK_ROWS_D0, cur.shape[-1], k3_pack_dtype(cur))
            y = torch.empty((n, d, h, w, m), dtype=x.dtype, device=x.device)
            ys = _new_stats(n, m, x.device)
            layers[l] = ChainLayer(cur.data_ptr(), _p(curs), wp.data_ptr(), y.data_ptr(), ys.data_ptr(), None, None, None, cur.shape[-1], m, 0, 0)
            keep.append(wp)
            nb += (cur.numel() + y.numel() + wt.numel()) * _esize(x)
            fl += 2.0 * n * d * h * w * 27 * wt.shape[0] * wt.shape[1]
            outs.append((y, ys))
            cur, curs = y, ys
            if ctx.needs_input_grad[2 + 2 * l]:
                _count_use(wt)
        sync = _chain_sync(n, x.device)
        with _timed("k3_chain<%s,fwd,%d>" % (_tname(x), nl), nb, fl, "x%s" % (tuple(x.shape),)):
            check(lib.vs_conv_k3_chain(_ct.addressof(layers), nl, 0, None, sync.data_ptr(), _chain_fault_word(x.device).data_ptr(), n, d, h, w,
                                       vs_dtype(x), EPS_IN, _stream()), "conv_k3_chain (forward)")
        ctx.nl = nl
        ctx.bias_refs = biases
        ctx.save_for_backward(x, xs, *[t for pair in outs[:-1] for t in pair], *weights)
        ctx.mark_non_differentiable(outs[-1][1])
        ctx.set_materialize_grads(False)
        return outs[-1]

    @staticmethod
    def backward(ctx, gy, _gys):
        nl = ctx.nl
        saved = ctx.saved_tensors
        none = (None,) * (2 + 2 * nl)
        if gy is None:
            return none
        acts = [(saved[0], saved[1])] + [(saved[2 + 2 * i], saved[3 + 2 * i]) for i in range(nl - 1)]      # the input of layer l
        weights = saved[2 * nl:]
        gy = _contig(gy)
        lazy = _take_lazy(gy)
        if lazy is not None:
            gy = apply_lazy(gy, lazy)
        n, d, h, w, _ = gy.shape
        dev = gy.device
        layers = (ChainLayer * nl)()
        applied = [None] * nl                      # dL/d(raw output of layer l): what layer l's weight gradient reads
        applied[nl - 1] = gy
        g_in, add, keep = gy, None, []
        nb = fl = 0.0
        for k, l in enumerate(range(nl - 1, -1, -1)):
            ax, axs = acts[l]
            wt = weights[l]
            wpb = pack_weight_cached(wt, VS_PACK_ROWS_D1_FLIP, g_in.shape[-1], k3_pack_dtype(g_in))
            g = torch.empty_like(ax)
            sums = _new_stats(n, ax.shape[-1], dev) if axs is not None else None
            if l == 0 and axs is not None:
                add = _collect_gradient(ax)         # a skip's gradient of the block's (lazy) input, parked by Materialize.backward
            layers[k] = ChainLayer(g_in.data_ptr(), None, wpb.data_ptr(), g.data_ptr(), None, ax.data_ptr() if axs is not None else None, _p(axs), _p(sums),
                                   g_in.shape[-1], ax.shape[-1], 1 if axs is not None else 0, 0)
            keep += [wpb, sums]
            nb += (g_in.numel() + (4 if axs is not None else 1) * g.numel() + wt.numel()) * _esize(gy)
            fl += 2.0 * n * d * h * w * 27 * wt.shape[0] * wt.shape[1]
            if l > 0:
                applied[l - 1] = g
            g_in = g
        sync = _chain_sync(n, dev)
        with _timed("k3_chain<%s,bwd,%d>" % (_tname(gy), nl), nb, fl, "gy%s" % (tuple(gy.shape),)):
            check(lib.vs_conv_k3_chain(_ct.addressof(layers), nl, 1, _p(add), sync.data_ptr(), _chain_fault_word(dev).data_ptr(), n, d, h, w,
                                       vs_dtype(gy), EPS_IN, _stream()), "conv_k3_chain (backward)")
        out = [g_in if ctx.needs_input_grad[0] else None, None]
        for l in range(nl):
            wt, bias = weights[l], ctx.bias_refs[l]
            gw = gb = None
            if ctx.needs_input_grad[2 + 2 * l]:
                ax, axs = acts[l]
                gw, _ = _side_grads(wt, (applied[l], ax, axs), (applied[l], None, ax, axs, wt.shape[0], wt.shape[1], VS_CONV_K3), None, None)
            if bias is not None and ctx.needs_input_grad[3 + 2 * l]:
                gb = _dead_bias_grad(bias, bias.shape[0], dev)
            out += [gw, gb]
        return tuple(out)


class ConvK3Softmax(torch.autograd.Function):
    """out_block (3x3x3 conv, live bias) + Softmax(dim=1) -> planar fp32 probabilities (joint_model.py:224-225,265-266 / 366-367,386-388):
    ONE fused launch for two classes (every BASELINE configuration), conv + softmax pass for 1 or 3..8 classes."""

    @staticmethod
    def forward(ctx, x, xs, weight, bias, drop_p=0.0, drop_seed=0):
        _require_cuda(x, weight)
        nc = weight.shape[0]
        if not 1 <= nc <= 8:
            raise NotImplementedError("out_block + softmax: n_class must be 1..8, got %d" % nc)
        n, d, h, w, c = x.shape
        wp = pack_weight_cached(weight, VS_PACK_ROWS_D0, c, k3_pack_dtype(x))
        prob = torch.empty((n, nc, d, h, w), dtype=torch.float32, device=x.device)
        if nc == 2:
            check(lib.vs_conv_k3_softmax2_dropout_fwd(x.data_ptr(), _p(xs), wp.data_ptr(), _p(bias), prob.data_ptr(), n, d, h, w, c,
                                                      vs_dtype(x), EPS_IN, float(drop_p), drop_seed, _stream()), "conv_k3_softmax2_fwd")
        else:       # more structures than one (main_source.py:92-93): the plain conv, then the softmax as its own pass
            bias8 = None if bias is None else torch.nn.functional.pad(bias.detach().float(), (0, 8 - nc))      # the conv kernels read one bias per padded row
            logits, _ = conv_gather(x, xs, wp, bias8, 8, VS_CONV_K3, False, real_channels=(weight.shape[1], nc))
            check(lib.vs_softmax_cl_fwd(logits.data_ptr(), prob.data_ptr(), n, d * h * w, 8, nc, vs_dtype(x), float(drop_p), drop_seed, _stream()),
                  "softmax_cl_fwd")
        ctx.save_for_backward(x, xs, weight, prob)
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias
        ctx.drop = (float(drop_p), drop_seed)
        ctx.defer = bool(getattr(x, "_vs_defer_apply", False)) and xs is not None
        if ctx.needs_input_grad[2]:
            _count_use(weight)                # see _wgrad_fusable
        return prob

    @staticmethod
    def backward(ctx, gprob):
        x, xs, weight, prob = ctx.saved_tensors
        n, d, h, w, c = x.shape
        gprob = _contig(gprob.float())
        nc = weight.shape[0]
        if nc == 2 and x.dtype != torch.float32:
            fused = _out_block_bwd_fused(ctx, x, xs, weight, prob, gprob, None)      # the whole backward as one launch (csrc/igemm_k3tw.h SM)
            if fused is not None:
                return fused
        gl = torch.empty((n, d, h, w, 8), dtype=x.dtype, device=x.device)
        if nc == 2:
            check(lib.vs_softmax2_dropout_bwd(prob.data_ptr(), gprob.data_ptr(), gl.data_ptr(), n, d * h * w, 8, vs_dtype(x),
                                              ctx.drop[0], ctx.drop[1], _stream()), "softmax2_bwd")
        else:
            check(lib.vs_softmax_cl_bwd(prob.data_ptr(), gprob.data_ptr(), gl.data_ptr(), n, d * h * w, 8, nc, vs_dtype(x),
                                        ctx.drop[0], ctx.drop[1], _stream()), "softmax_cl_bwd")
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            wpb = pack_weight_cached(weight, VS_PACK_ROWS_D1_FLIP, 8, k3_pack_dtype(x))
            if xs is not None:
                gx = conv_bwd_data_lazy(gl, wpb, x, xs, VS_CONV_K3, real_channels=(nc, weight.shape[1]), defer=ctx.defer)
            else:
                gx, _ = conv_gather(gl, None, wpb, None, c, VS_CONV_K3, False)
        if ctx.needs_input_grad[2]:
            gw, gb = _side_grads(weight, (gl, x, xs), (gl, None, x, xs, nc, weight.shape[1], VS_CONV_K3),
                                 (gl, nc) if ctx.has_bias and ctx.needs_input_grad[3] else None, ctx.bias_ref)
        elif ctx.has_bias and ctx.needs_input_grad[3]:
            gb = bias_grad(gl, nc)
        return gx, None, gw, gb, None, None


class ConvK3SoftmaxCL(torch.autograd.Function):
    """ConvK3Softmax that ALSO returns the probabilities as the channels-last bf16 tensor the next network reads
    (Joint.forward: Segmentation's prediction is the VAE's input, joint_model.py:447-450) -> (prob, prob_cl).  Saves the
    vs_pack_planar launch in forward and, in backward, vs_unpack_planar and autograd's add of the two gradients of prob:
    the softmax backward takes both parts (vs_softmax2_cl_bwd).  bf16 kernels only."""

    @staticmethod
    def forward(ctx, x, xs, weight, bias, drop_p=0.0, drop_seed=0):
        _require_cuda(x, weight)
        if weight.shape[0] != 2 or x.dtype == torch.float32:
            raise NotImplementedError("fused out_block+softmax with a channels-last copy: n_class == 2, 16-bit storage")
        n, d, h, w, c = x.shape
        wp = pack_weight_cached(weight, VS_PACK_ROWS_D0, c, k3_pack_dtype(x))
        prob = torch.empty((n, 2, d, h, w), dtype=torch.float32, device=x.device)
        prob_cl = torch.empty((n, d, h, w, 8), dtype=x.dtype, device=x.device)
        check(lib.vs_conv_k3_softmax2_cl_fwd(x.data_ptr(), _p(xs), wp.data_ptr(), _p(bias), prob.data_ptr(), prob_cl.data_ptr(), n, d, h, w, c,
                                             vs_dtype(x), EPS_IN, float(drop_p), drop_seed, _stream()), "conv_k3_softmax2_cl_fwd")
        ctx.save_for_backward(x, xs, weight, prob)
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias
        ctx.drop = (float(drop_p), drop_seed)
        ctx.defer = bool(getattr(x, "_vs_defer_apply", False)) and xs is not None
        if ctx.needs_input_grad[2]:
            _count_use(weight)                # see _wgrad_fusable
        ctx.set_materialize_grads(False)
        return prob, prob_cl

    @staticmethod
    def backward(ctx, gprob, gcl):
        x, xs, weight, prob = ctx.saved_tensors
        if gprob is None and gcl is None:
            return None, None, None, None, None, None
        n, d, h, w, c = x.shape
        gprob = None if gprob is None else _contig(gprob.float())
        gcl = None if gcl is None else _contig(gcl)
        fused = _out_block_bwd_fused(ctx, x, xs, weight, prob, gprob, gcl)
        if fused is not None:
            return fused
        gl = torch.empty((n, d, h, w, 8), dtype=x.dtype, device=x.device)
        check(lib.vs_softmax2_cl_bwd(prob.data_ptr(), _p(gprob), _p(gcl), gl.data_ptr(), n, d * h * w, 8, vs_dtype(x), ctx.drop[0], ctx.drop[1],
                                     _stream()), "softmax2_cl_bwd")
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            wpb = pack_weight_cached(weight, VS_PACK_ROWS_D1_FLIP, 8, k3_pack_dtype(x))
            if xs is not None:
                gx = conv_bwd_data_lazy(gl, wpb, x, xs, VS_CONV_K3, real_channels=(2, weight.shape[1]), defer=ctx.defer)
            else:
                gx, _ = conv_gather(gl, None, wpb, None, c, VS_CONV_K3, False)
        if ctx.needs_input_grad[2]:
            gw, gb = _side_grads(weight, (gl, x, xs), (gl, None, x, xs, 2, weight.shape[1], VS_CONV_K3),
                                 (gl, 2) if ctx.has_bias and ctx.needs_input_grad[3] else None, ctx.bias_ref)
        elif ctx.has_bias and ctx.needs_input_grad[3]:
            gb = bias_grad(gl, 2)
        return gx, None, gw, gb, None, None


FUSE_SOFTMAX_BWD = os.environ.get("VS_FUSE_SOFTMAX_BWD", "1") != "0"


def _out_block_bwd_fused(ctx, x, xs, weight, prob, gprob, gcl):
    """out_block's backward as ONE launch (vs_conv_k3_softmax2_bwd_data, csrc/igemm_k3tw.h SM): the softmax backward runs while the backward-data kernel stages
    its tile — the gradient of the logits is never stored — and a trainable layer's weight and bias gradients come out of the same launch (slabs + partials for the
    grouped reduction).  -> the tuple ConvK3SoftmaxCL.backward returns, or None when this launch does not apply (the three-launch path then runs)."""
    n, d, h, w, c = x.shape
    want_w, want_b = ctx.needs_input_grad[2], ctx.has_bias and ctx.needs_input_grad[3]
    if not (FUSE_SOFTMAX_BWD and ctx.needs_input_grad[0] and xs is not None and c == 8 and (gcl is None or gcl.shape[-1] == 8)):
        return None
    if not lib.vs_conv_k3_bwd_data_wgrad_supported(n, d, h, w, 8, 8, vs_dtype(x)):
        return None
    if want_w:
        if not _wgrad_fusable(x, x, weight):    # (the logits' gradient has x's grid, storage type and 8 stored channels)
            return None
        bias = ctx.bias_ref
        if want_b and not (bias.is_leaf and bias.grad is None and not _has_hooks(bias)):
            return None
    elif want_b:
        return None                             # a frozen weight under a trainable bias: not a configuration of the reference
    wpb = pack_weight_cached(weight, VS_PACK_ROWS_D1_FLIP, 8, k3_pack_dtype(x))
    sums = _new_stats(n, c, x.device)
    g = torch.empty_like(x)
    nslabs = lib.vs_conv_k3_bwd_data_wgrad_slabs(n, d, h, w) if want_w else 0
    slabs = torch.empty(nslabs * 1728, dtype=torch.float32, device=x.device) if want_w else None
    bpart = torch.empty(nslabs * 2, dtype=torch.float64, device=x.device) if (want_w and want_b) else None
    kid = nb = fl = None
    if PROFILE is not None:
        kid = "k3tw_kernel<%s,2,%s>" % (_tname(x), "true" if want_w else "false")
        nb = (4 * n * d * h * w * 4) + (0 if gcl is None else gcl.numel() * _esize(x)) + 2 * g.numel() * _esize(x)
        fl = (2.0 if want_w else 1.0) * 2.0 * (g.numel() // c) * 27 * 2 * weight.shape[1]
    with _timed(kid, nb, fl, "out_block bwd%s (%d, %d, %d, %d)" % ("+wgrad" if want_w else "", n, d, h, w)):
        check(lib.vs_conv_k3_softmax2_bwd_data(prob.data_ptr(), _p(gprob), _p(gcl), wpb.data_ptr(), g.data_ptr(), x.data_ptr(), xs.data_ptr(), sums.data_ptr(),
                                               _p(slabs), _p(bpart), n, d, h, w, vs_dtype(x), EPS_IN, ctx.drop[0], ctx.drop[1], _stream()),
              "conv_k3_softmax2_bwd_data")
    gw = gb = None
    if want_w:
        if want_b:
            gw, gb = _group_submit_slabs(weight, slabs, nslabs, 2, weight.shape[1], (x,), bias=(ctx.bias_ref, bpart))
        else:
            gw = _group_submit_slabs(weight, slabs, nslabs, 2, weight.shape[1], (x,))
    if ctx.defer:
        _defer_register(g, x, xs, sums)
    else:
        _apply_in_place(g, x, xs, sums)
    return g, None, gw, gb, None, None


def out_block_softmax(x, xs, weight, bias, drop_p=0.0, drop_seed=0):
    """out_block + softmax -> planar fp32 probabilities; in bf16 mode the tensor carries its channels-last copy as `_vs_cl`, which a
    network that takes it as input (VAE / Joint) uses instead of re-packing it (set VS_SOFTMAX_CL=0 to disable)."""
    if x.dtype != torch.float32 and weight.shape[0] == 2 and os.environ.get("VS_SOFTMAX_CL", "1") != "0":
        prob, prob_cl = ConvK3SoftmaxCL.apply(x, xs, weight, bias, drop_p, drop_seed)
        prob._vs_cl = prob_cl
        return prob
    return ConvK3Softmax.apply(x, xs, weight, bias, drop_p, drop_seed)


def planar_input(x, dtype):
    """channels-last kernel-dtype view of a planar (N, C, D, H, W) input: the producer's own channels-last copy when it left one
    (out_block_softmax), vs_pack_planar otherwise."""
    cl = getattr(x, "_vs_cl", None)
    if cl is not None and cl.dtype == dtype and cl.shape[0] == x.shape[0] and tuple(cl.shape[1:4]) == tuple(x.shape[2:]):
        return cl
    return PackPlanar.apply(x, dtype)


class ConvK2S2(torch.autograd.Function):
    """Conv3d(C, C, 2, stride 2) with live bias on a lazy input; output is final (feeds a conv directly)
    — joint_model.py:130."""

    @staticmethod
    def forward(ctx, x, xs, weight, bias):
        _require_cuda(x, weight)
        wp = pack_weight_cached(weight, VS_PACK_ROWS_D0, x.shape[-1], x.dtype)
        y, _ = conv_gather(x, xs, wp, bias, cpad(weight.shape[0]), VS_CONV_K2S2, False)
        ctx.save_for_backward(x, xs, weight)
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias
        ctx.defer = bool(getattr(x, "_vs_defer_apply", False)) and xs is not None     # see _LAZY_APPLY
        return y

    @staticmethod
    def backward(ctx, gy):
        x, xs, weight = ctx.saved_tensors
        gy = _contig(gy)
        cout, cin = weight.shape[0], weight.shape[1]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            wpb = pack_weight_cached(weight, VS_PACK_SCATTER_D1, gy.shape[-1], gy.dtype)
            if xs is not None:
                gx = conv_bwd_data_lazy(gy, wpb, x, xs, VS_CONV_K2S2, scatter=True, defer=ctx.defer)
            else:
                gx = conv_scatter(gy, None, wpb, None, x.shape[-1])
        if ctx.needs_input_grad[2]:
            gw, gb = _side_grads(weight, (gy, x, xs), (gy, None, x, xs, cout, cin, VS_CONV_K2S2),
                                 (gy, cout) if ctx.has_bias and ctx.needs_input_grad[3] else None, ctx.bias_ref)
        elif ctx.has_bias and ctx.needs_input_grad[3]:
            gb = bias_grad(gy, cout)
        return gx, None, gw, gb


class ConvT2S2(torch.autograd.Function):
    """ConvTranspose3d(C, C, 2, stride 2) with live bias on a lazy input; final output — joint_model.py:118."""

    @staticmethod
    def forward(ctx, x, xs, weight, bias):
        _require_cuda(x, weight)
        wp = pack_weight_cached(weight, VS_PACK_SCATTER_D1, x.shape[-1], x.dtype)
        y = conv_scatter(x, xs, wp, bias, cpad(weight.shape[1]))
        ctx.save_for_backward(x, xs, weight)
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias
        ctx.defer = bool(getattr(x, "_vs_defer_apply", False)) and xs is not None     # see _LAZY_APPLY
        return y

    @staticmethod
    def backward(ctx, gy):
        x, xs, weight = ctx.saved_tensors
        gy = _contig(gy)
        cin, cout = weight.shape[0], weight.shape[1]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            wpb = pack_weight_cached(weight, VS_PACK_ROWS_D0, gy.shape[-1], gy.dtype)
            if xs is not None:
                gx = conv_bwd_data_lazy(gy, wpb, x, xs, VS_CONV_K2S2, defer=ctx.defer)
            else:
                gx, _ = conv_gather(gy, None, wpb, None, x.shape[-1], VS_CONV_K2S2, False)
        if ctx.needs_input_grad[2]:
            gw, gb = _side_grads(weight, (gy, x, xs), (x, xs, gy, None, cin, cout, VS_CONV_K2S2),
                                 (gy, cout) if ctx.has_bias and ctx.needs_input_grad[3] else None, ctx.bias_ref)
        elif ctx.has_bias and ctx.needs_input_grad[3]:
            gb = bias_grad(gy, cout)
        return gx, None, gw, gb


# ------------------------------------------------------------------------------------------------
# composed Up block head: ConvTranspose3d(C, C, 2, stride 2) -> Conv3d(C, Co, 3, padding 1) as one operator (csrc/igemm_k4.h)
# ------------------------------------------------------------------------------------------------
FUSE_UP = os.environ.get("VS_FUSE_UP", "1") != "0"
# composed where it is measured faster than the two-launch pair.  Same-box sweep of this threshold on the 96^3 step (frozen VAE heads only),
# coarse voxels per sample: 48^3 only 2.551 ms; + 24^3 2.531; + 12^3 2.517; + 6^3 2.523; + 3^3 2.550 — on the 6^3 / 3^3 grids the 4x4x16-tile
# kernels of igemm_k4.h are mostly padding and their backward walks 8 Co / 32 short chunk stages.  At 160^3 (coarse 80^3 / 40^3 / 20^3 ...):
# 6.99 -> 6.88 ms once the 40^3 level is in.  Isolated kernels, forward / backward-data: 48^3 x 16: 19.8 / 20.0 us (pair: 27.6 + 19.7 / 34 + 12.4).
FUSE_UP_MIN_VOXELS = int(os.environ.get("VS_FUSE_UP_MIN_VOXELS", "1000"))
# Trainable weights through the composed head (tests/test_gpu_up.py: dW3, dW2, db2 against autograd): the weight-gradient side gains (the largest
# layer of the 16-channel bucket leaves it, the transposed conv's weight gradient and its full-resolution operand disappear) and the per-step
# helpers cost ~80 us whatever the volume (re-composition 32, boundary sums 21, chain rule 28: microseconds of work in launches bound by
# their dependent round trips).  Measured, same box: 96^3 B=2 (2 x 48^3 coarse voxels) 2.69 vs 2.66 ms and 128^3 B=1 (64^3) 3.87 vs 3.84 — a loss;
# 160^3 B=2 (2 x 80^3) 6.84 vs 7.19 ms — a 5 % gain.  Enabled from FUSE_UP_TRAINABLE_MIN_VOXELS coarse voxels (batch included) up;
# VS_FUSE_UP_TRAINABLE=0 / 1 forces it off / on.
_FUT = os.environ.get("VS_FUSE_UP_TRAINABLE", "")
FUSE_UP_TRAINABLE = _FUT != "0"
FUSE_UP_TRAINABLE_MIN_VOXELS = 0 if _FUT == "1" else int(os.environ.get("VS_FUSE_UP_TRAINABLE_MIN_VOXELS", "500000"))     # coarse voxels of the whole batch


def _up_stamp(wt, bt, w3):
    return (wt._version, wt.data_ptr(), w3._version, w3.data_ptr(), None if bt is None else (bt._version, bt.data_ptr()),
            _TRAIN_EPOCH[0] if (w3.requires_grad or wt.requires_grad) else _PACK_EPOCH[0])


# (id(w3), dtype) -> weak references to (wt, bt, w3): the trainable Up heads whose composed images repack_trainable() re-composes after every
# optimiser step.  Weak: a model that has been dropped (TestTimeFinetune copies, test fixtures) leaves the registry with its parameters, the
# plan itself lives on w3 (`_vs_up_plan`).  Entries are NOT pruned by use: a captured graph replays the composed kernels without running up_plan.
_UP_TRAINABLE = {}


def _up_compose_into(plan, wt, bt, w3, dtype):
    """(re)compose in place; the caller sets plan["stamp"] from the PARAMETERS (detached aliases carry their own version counters)"""
    cin, cm, co = wt.shape[0], wt.shape[1], w3.shape[0]
    check(lib.vs_up_compose(wt.data_ptr(), _p(bt), w3.data_ptr(), plan["weff"].data_ptr(), plan["img_f"].data_ptr(), plan["img_b"].data_ptr(),
                            plan["taps_f"].data_ptr(), plan["taps_b"].data_ptr(), plan["btab"].data_ptr(), cin, cm, co, vs_of(dtype), _stream()), "up_compose")


def up_plan(wt, bt, w3, dtype):
    """The composed images of an Up block's first two layers (FROZEN weights: packed once, cached on the 3x3x3 weight tensor at stable
    addresses — a captured HIP graph holds the pointers — and re-composed IN PLACE when any of the three tensors changed:
    refresh_frozen_packs / clear_pack_cache, as for the packed conv weights)."""
    plans = getattr(w3, "_vs_up_plan", None)
    if plans is None:
        plans = {}
        w3._vs_up_plan = plans
    plan = plans.get(dtype)
    if plan is None:
        cin, cm, co = wt.shape[0], wt.shape[1], w3.shape[0]
        sz = (_ct.c_size_t * 6)()
        check(lib.vs_up_compose_sizes(cin, cm, co, _ct.addressof(sz)), "up_compose_sizes")
        dev = w3.device
        buf = lambda nbytes: torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        plan = {"weff": buf(sz[0]), "img_f": buf(sz[1]), "img_b": buf(sz[2]), "taps_f": buf(sz[3]), "taps_b": buf(sz[4]), "btab": buf(sz[5]),
                "stamp": None, "src": (wt, bt)}
        plans[dtype] = plan
    if (w3.requires_grad or wt.requires_grad) and (id(w3), dtype) not in _UP_TRAINABLE:
        _UP_TRAINABLE[(id(w3), dtype)] = (_weakref.ref(wt), None if bt is None else _weakref.ref(bt), _weakref.ref(w3))
    if plan["stamp"] != _up_stamp(wt, bt, w3):
        with torch.no_grad():
            _up_compose_into(plan, wt.detach(), None if bt is None else bt.detach(), w3.detach(), dtype)
            plan["stamp"] = _up_stamp(wt, bt, w3)
    return plan


def _recompose_trainable_ups():
    for key, (rwt, rbt, rw3) in list(_UP_TRAINABLE.items()):
        wt, bt, w3, dtype = rwt(), None if rbt is None else rbt(), rw3(), key[1]
        plan = None if w3 is None else (getattr(w3, "_vs_up_plan", None) or {}).get(dtype)
        if wt is None or plan is None or (rbt is not None and bt is None) or id(w3) != key[0]:
            del _UP_TRAINABLE[key]          # the model is gone
            continue
        with torch.no_grad():
            _up_compose_into(plan, wt.detach(), None if bt is None else bt.detach(), w3.detach(), dtype)
            plan["stamp"] = _up_stamp(wt, bt, w3)


def up_composed_ok(x, tconv, conv3):
    """Can this Up block head run as the composed operator?  16-bit storage, channel pairs the kernels are instantiated for
    (vs_up_supported), large enough a grid to win (FUSE_UP_MIN_VOXELS).  Trainable weights: their gradients come from the composed
    weight-gradient (VS_CONV_UP descriptor in the grouped end-of-pass launches) through the parameter-space chain rule (vs_up_chain) — needs
    the grouped, single-phase weight-gradient mode (the default)."""
    wt, w3 = tconv.weight, conv3.weight
    if not FUSE_UP or x.dtype == torch.float32:
        return False
    trainable = wt.requires_grad or w3.requires_grad or (tconv.bias is not None and tconv.bias.requires_grad)
    if trainable and (not FUSE_UP_TRAINABLE or not _GROUP["enabled"] or _GROUP["split"] is not None
                      or not (wt.is_leaf and w3.is_leaf)):
        return False
    if tuple(wt.shape[2:]) != (2, 2, 2) or tuple(w3.shape[2:]) != (3, 3, 3) or w3.shape[1] != wt.shape[1] or x.shape[-1] != wt.shape[0]:
        return False
    if x.shape[1] * x.shape[2] * x.shape[3] < FUSE_UP_MIN_VOXELS:
        return False
    if trainable and x.shape[0] * x.shape[1] * x.shape[2] * x.shape[3] < FUSE_UP_TRAINABLE_MIN_VOXELS:
        return False
    return bool(lib.vs_up_supported(wt.shape[0], wt.shape[1], w3.shape[0], vs_dtype(x)))


class UpConvK3(torch.autograd.Function):
    """relu(instnorm(x)) [lazy] -> ConvTranspose3d(C, C, 2, 2) + bias -> Conv3d(C, Co, 3, pad 1) as ONE launch on the coarse grid
    (joint_model.py:116-120 + 40); output lazy (raw + statistics) on the fine grid.  No intermediate tensor, 3.4x fewer multiply-adds."""

    @staticmethod
    def forward(ctx, x, xs, wt, bt, w3, b3=None):
        _require_cuda(x, wt, w3)
        n, d, h, w, c = x.shape
        co = w3.shape[0]
        plan = up_plan(wt, bt, w3, x.dtype)
        ctx.b3 = b3                                # the 3x3x3 conv's bias: dead (InstanceNorm follows) — its gradient is returned as zeros
        y = torch.empty((n, 2 * d, 2 * h, 2 * w, co), dtype=x.dtype, device=x.device)
        ys = _new_stats(n, co, x.device)
        kid = nb = fl = None
        if PROFILE is not None:
            kid = "k4t_kernel<%d,%s>" % (min(c, 32), _tname(x))
            nb = (x.numel() + y.numel()) * _esize(x) + 64 * co * c * _esize(x)
            fl = 2.0 * (x.numel() // c) * 64 * co * c
        with _timed(kid, nb, fl, "up x%s->m%d" % (tuple(x.shape), co)):
            check(lib.vs_up_conv_fwd(x.data_ptr(), _p(xs), plan["img_f"].data_ptr(), plan["taps_f"].data_ptr(), plan["btab"].data_ptr(), y.data_ptr(),
                                     ys.data_ptr(), n, d, h, w, c, co, vs_dtype(x), EPS_IN, _stream()), "up_conv_fwd")
        ctx.save_for_backward(x, xs)
        ctx.plan = plan
        ctx.co = co
        ctx.params = (wt, bt, w3)               # Parameters (long-lived leaves): read by the chain rule, their gradient slots looked up in backward
        ctx.defer = bool(getattr(x, "_vs_defer_apply", False)) and xs is not None
        ctx.mark_non_differentiable(ys)
        ctx.set_materialize_grads(False)
        return y, ys

    @staticmethod
    def backward(ctx, gy, _gys):
        x, xs = ctx.saved_tensors
        if gy is None:
            return None, None, None, None, None, None
        gy = _contig(gy)
        lazy = _take_lazy(gy)
        if lazy is not None:
            gy = apply_lazy(gy, lazy)
        gwt = gbt = gw3 = gb3 = None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3] or ctx.needs_input_grad[4]:
            gwt, gbt, gw3 = _up_weight_grads(gy, x, xs, ctx.params, ctx.co, ctx.needs_input_grad)
        if ctx.b3 is not None and ctx.needs_input_grad[5]:
            gb3 = _dead_bias_grad(ctx.b3, ctx.b3.shape[0], x.device)
        if not ctx.needs_input_grad[0]:
            return None, None, gwt, gbt, gw3, gb3
        n, d, h, w, c = x.shape
        gx = torch.empty_like(x)
        sums = _new_stats(n, c, x.device) if xs is not None else None
        plan = ctx.plan
        kid = nb = fl = None
        if PROFILE is not None:
            kid = "k4g_kernel<%s>" % _tname(x)
            nb = (gy.numel() + (2 if xs is not None else 1) * x.numel()) * _esize(x) + 64 * ctx.co * c * _esize(x)
            fl = 2.0 * (x.numel() // c) * 64 * ctx.co * c
        with _timed(kid, nb, fl, "up bwd gy%s->m%d" % (tuple(gy.shape), c)):
            check(lib.vs_up_conv_bwd_data(gy.data_ptr(), plan["img_b"].data_ptr(), plan["taps_b"].data_ptr(), gx.data_ptr(), _p(x if xs is not None else None),
                                          _p(xs), _p(sums), n, d, h, w, ctx.co, c, vs_dtype(x), EPS_IN, _stream()), "up_conv_bwd_data")
        if xs is not None:
            if ctx.defer:
                _defer_register(gx, x, xs, sums)
            else:
                _apply_in_place(gx, x, xs, sums)
        return gx, None, gwt, gbt, gw3, gb3


_UP_JOBS = {}       # id(w3) -> the pending chain-rule job of a trainable composed Up head in the current backward pass


def _up_weight_grads(gy, x, xs, params, co, needs):
    """Weight gradients of the composed head: dWeff27 from a VS_CONV_UP descriptor in the grouped end-of-pass launch, the boundary sums of gy
    now (vs_up_faces), and a chain-rule job that flush_wgrads() runs right after the grouped launch.  -> (g wt, g bt, g w3) for autograd
    (unwritten until then, like every deferred weight gradient); later uses of the same weights in one pass add to the first use's job."""
    wt, bt, w3 = params
    n, d, h, w, c = x.shape
    job = _UP_JOBS.get(id(w3))
    first = job is None
    if first:
        dev = x.device
        gwt = _grad_slot(wt, wt.shape) if needs[2] else None
        gbt = _grad_slot(bt, bt.shape) if (bt is not None and needs[3]) else None
        gw3 = _grad_slot(w3, w3.shape) if needs[4] else None
        # the job holds raw ADDRESSES of the gradients, not the tensors: AccumulateGrad adopts a gradient as .grad (no copy of the still
        # unwritten buffer) only if nobody else holds the tensor object; .grad keeps the storage alive until the chain rule has written it
        job = {"dweff": torch.empty(8 * co * c * 27, dtype=torch.float32, device=dev),
               "faces": _new_stats(1, 27 * co // 2, dev),          # zeroed, statistics format: vs_up_faces accumulates (every use of the weights)
               "gwt": _p(gwt), "gbt": _p(gbt), "gw3": _p(gw3), "params": params, "co": co}
        _UP_JOBS[id(w3)] = job
    check(lib.vs_up_faces(gy.data_ptr(), job["faces"].data_ptr(), n, 2 * d, 2 * h, 2 * w, co, vs_dtype(gy), _stream()), "up_faces")
    _group_submit(w3, (gy, x, xs), (gy, None, x, xs, 8 * co, c, VS_CONV_UP), None, job["dweff"], None, up_co=co)
    deferrable = all(p is None or (p.is_leaf and p.grad is None and not _has_hooks(p)) for p in (wt, bt, w3)) and not torch.is_grad_enabled()
    if not deferrable:
        flush_wgrads()                          # runs the chain rule too
    return (gwt, gbt, gw3) if first else (None, None, None)


def _run_up_jobs():
    for job in _UP_JOBS.values():
        wt, bt, w3 = job["params"]
        check(lib.vs_up_chain(job["dweff"].data_ptr(), job["faces"].data_ptr(), wt.data_ptr(), _p(bt), w3.data_ptr(), job["gwt"], job["gbt"],
                              job["gw3"], wt.shape[0], wt.shape[1], job["co"], _stream()), "up_chain")
    _UP_JOBS.clear()


class Materialize(torch.autograd.Function):
    """(raw, stats) [+ (raw2, stats2)] -> relu(instnorm(raw)) [+ relu(instnorm(raw2))] as a final tensor.
    With two operands this is the additive U-Net skip (joint_model.py:380,382)."""

    @staticmethod
    def forward(ctx, x, xs, x2, x2s, park_second=False):
        _require_cuda(x)
        ctx.park_second = bool(park_second)
        n, c = x.shape[0], x.shape[-1]
        voxels = x.numel() // (n * c)
        out = torch.empty_like(x)
        check(lib.vs_instnorm_relu_fwd(x.data_ptr(), _p(xs), _p(x2), _p(x2s), out.data_ptr(), n, voxels, c, vs_dtype(x),
                                       EPS_IN, _stream()), "instnorm_relu_fwd")
        ctx.save_for_backward(x, xs, x2, x2s)
        return out

    @staticmethod
    def backward(ctx, g):
        x, xs, x2, x2s = ctx.saved_tensors
        g = _contig(g)
        g1 = g2 = None
        if (x2 is not None and xs is not None and x2s is not None and ctx.needs_input_grad[0] and ctx.needs_input_grad[2] and PROFILE is None):
            # both operands lazy (the U-Net skips): reduce and apply of the two in one launch each
            n, c = x.shape[0], x.shape[-1]
            voxels = x.numel() // (n * c)
            s1, s2 = _new_stats(n, c, x.device), _new_stats(n, c, x.device)
            g1, g2 = torch.empty_like(g), torch.empty_like(g)
            check(lib.vs_instnorm_relu_bwd_pair(g.data_ptr(), x.data_ptr(), xs.data_ptr(), s1.data_ptr(), g1.data_ptr(), x2.data_ptr(),
                                                x2s.data_ptr(), s2.data_ptr(), g2.data_ptr(), n, voxels, c, vs_dtype(x), EPS_IN, _stream()),
                  "instnorm_relu_bwd_pair")
            if ctx.park_second and not torch.is_grad_enabled():
                _park_gradient(x2, g2)          # the encoder conv that shares x2 sums it into its own gradient (no autograd add launch)
                return g1, None, None, None, None
            return g1, None, g2, None, None
        if ctx.needs_input_grad[0]:
            g1 = in_relu_bwd(g, x, xs, inplace=False) if xs is not None else g
        if x2 is not None and ctx.needs_input_grad[2]:
            g2 = in_relu_bwd(g, x2, x2s, inplace=False) if x2s is not None else g
        return g1, None, g2, None, None


VS_NORM_INSTANCE, VS_NORM_BATCH, VS_NORM_BATCH_EVAL, VS_NORM_NONE = 0, 1, 2, 3
VS_ACT_RELU, VS_ACT_SOFTPLUS = 0, 1


class NormAct(torch.autograd.Function):
    """(raw conv output, its statistics) -> act(norm(raw) * gamma + beta) as a stored tensor: the block settings no entry point of the
    reference uses — BatchNorm3d (joint_model.py:13, with running statistics), Softplus (joint_model.py:38) — and InstanceNorm with
    either activation.  Unfused on purpose (vaeseg.h: general normalisation); the InstanceNorm + ReLU configuration of every entry point
    never comes here (it stays lazy, fused into the convs).  bn = the nn.BatchNorm3d holder (running buffers, momentum, eps, training)
    or None for InstanceNorm."""

    @staticmethod
    def forward(ctx, x, xs, gamma, beta, bn, act, c_real):
        _require_cuda(x)
        n, c = x.shape[0], x.shape[-1]
        voxels = x.numel() // (n * c)
        if bn is None:
            mode, eps, mom, rm, rv, nbt = VS_NORM_INSTANCE, EPS_IN, 0.0, None, None, None
        elif isinstance(bn, str):             # "none": activation only (the conv -> ReLU pairs of the *_GS blocks, joint_model.py:58-63)
            mode, eps, mom, rm, rv, nbt = VS_NORM_NONE, EPS_IN, 0.0, None, None, None
        else:
            track = bn.track_running_stats and bn.running_mean is not None
            mode = VS_NORM_BATCH if (bn.training or not track) else VS_NORM_BATCH_EVAL
            eps = float(bn.eps)
            mom = 0.0 if bn.momentum is None else float(bn.momentum)
            rm, rv, nbt = (bn.running_mean, bn.running_var, bn.num_batches_tracked) if track else (None, None, None)
            if bn.momentum is None and track and mode == VS_NORM_BATCH:
                raise NotImplementedError("BatchNorm3d(momentum=None) (cumulative average) has no native form; the reference uses momentum=0.1")
        tab = torch.empty(2, n, c, dtype=torch.float32, device=x.device)
        check(lib.vs_norm_tables(_p(xs), n, c, c_real, float(voxels), mode, eps, mom, _p(rm), _p(rv), _p(nbt), tab[0].data_ptr(), tab[1].data_ptr(),
                                 _stream()), "norm_tables")
        y = torch.empty_like(x)
        check(lib.vs_norm_act_fwd(x.data_ptr(), tab[0].data_ptr(), tab[1].data_ptr(), _p(gamma), _p(beta), y.data_ptr(), n, voxels, c, c_real, act,
                                  vs_dtype(x), _stream()), "norm_act_fwd")
        ctx.save_for_backward(x, tab, gamma, beta)
        ctx.mode, ctx.act, ctx.c_real = mode, act, c_real
        return y

    @staticmethod
    def backward(ctx, g):
        x, tab, gamma, beta = ctx.saved_tensors
        g = _contig(g)
        n, c = x.shape[0], x.shape[-1]
        voxels = x.numel() // (n * c)
        dt, st = vs_dtype(x), _stream()
        sums = _new_stats(n, c, x.device)
        check(lib.vs_norm_act_bwd_reduce(g.data_ptr(), x.data_ptr(), tab[0].data_ptr(), tab[1].data_ptr(), _p(gamma), _p(beta), sums.data_ptr(), n,
                                         voxels, c, ctx.c_real, ctx.act, dt, st), "norm_act_bwd_reduce")
        coef = torch.empty(n, c, 3, dtype=torch.float32, device=x.device)
        dgamma = torch.empty_like(gamma) if gamma is not None and ctx.needs_input_grad[2] else None
        dbeta = torch.empty_like(beta) if beta is not None and ctx.needs_input_grad[3] else None
        check(lib.vs_norm_act_bwd_finish(sums.data_ptr(), n, c, ctx.c_real, float(voxels), ctx.mode, tab[1].data_ptr(), _p(gamma), coef.data_ptr(),
                                         _p(dgamma), _p(dbeta), st), "norm_act_bwd_finish")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            check(lib.vs_norm_act_bwd_apply(g.data_ptr(), x.data_ptr(), tab[0].data_ptr(), tab[1].data_ptr(), _p(gamma), _p(beta), coef.data_ptr(),
                                            dx.data_ptr(), n, voxels, c, ctx.c_real, ctx.act, dt, st), "norm_act_bwd_apply")
        return dx, None, dgamma, dbeta, None, None, None


class GSNorm(torch.autograd.Function):
    """GSNorm3d (joint_model.py:17-33) on a channels-last tensor: every channel divided by the sum over its group (+1e-4)"""

    @staticmethod
    def forward(ctx, x, num_group):
        _require_cuda(x)
        c = x.shape[-1]
        y = torch.empty_like(x)
        check(lib.vs_gsnorm_fwd(x.data_ptr(), y.data_ptr(), x.numel() // c, c, num_group, vs_dtype(x), _stream()), "gsnorm_fwd")
        ctx.save_for_backward(x)
        ctx.num_group = num_group
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = _contig(g)
        c = x.shape[-1]
        dx = torch.empty_like(x)
        check(lib.vs_gsnorm_bwd(g.data_ptr(), x.data_ptr(), dx.data_ptr(), x.numel() // c, c, ctx.num_group, vs_dtype(x), _stream()), "gsnorm_bwd")
        return dx, None


class UpsampleTrilinear(torch.autograd.Function):
    """torch.nn.Upsample(scale_factor=s, mode='trilinear') (joint_model.py:69,323-325) on a channels-last tensor"""

    @staticmethod
    def forward(ctx, x, scale):
        _require_cuda(x)
        n, d, h, w, c = x.shape
        y = torch.empty((n, d * scale, h * scale, w * scale, c), dtype=x.dtype, device=x.device)
        check(lib.vs_upsample_trilinear_fwd(x.data_ptr(), y.data_ptr(), n, d, h, w, c, scale, vs_dtype(x), _stream()), "upsample_trilinear_fwd")
        ctx.shape, ctx.scale = (n, d, h, w, c), scale
        return y

    @staticmethod
    def backward(ctx, g):
        g = _contig(g)
        n, d, h, w, c = ctx.shape
        dx = torch.empty(ctx.shape, dtype=g.dtype, device=g.device)
        scratch = torch.empty((n * d * h * w * c + 3) // 4 * 4, dtype=torch.float32, device=g.device)
        check(lib.vs_upsample_trilinear_bwd(g.data_ptr(), dx.data_ptr(), scratch.data_ptr(), n, d, h, w, c, ctx.scale, vs_dtype(g), _stream()),
              "upsample_trilinear_bwd")
        return dx, None


class Softmax2(torch.autograd.Function):
    """nn.Softmax(dim=1) as its own pass: channels-last logits (channels 0..n_class-1) -> planar fp32 (N, n_class, D, H, W); n_class 1..8"""

    @staticmethod
    def forward(ctx, x, n_class=2):
        _require_cuda(x)
        n, d, h, w, c = x.shape
        prob = torch.empty((n, n_class, d, h, w), dtype=torch.float32, device=x.device)
        if n_class == 2:
            check(lib.vs_softmax2_fwd(x.data_ptr(), prob.data_ptr(), n, d * h * w, c, vs_dtype(x), _stream()), "softmax2_fwd")
        else:
            check(lib.vs_softmax_cl_fwd(x.data_ptr(), prob.data_ptr(), n, d * h * w, c, n_class, vs_dtype(x), 0.0, 0, _stream()), "softmax_cl_fwd")
        ctx.save_for_backward(prob)
        ctx.c, ctx.dtype = c, x.dtype
        return prob

    @staticmethod
    def backward(ctx, gprob):
        (prob,) = ctx.saved_tensors
        gprob = _contig(gprob.float())
        n, nc, d, h, w = prob.shape
        g = torch.empty((n, d, h, w, ctx.c), dtype=ctx.dtype, device=prob.device)
        if nc == 2:
            check(lib.vs_softmax2_bwd(prob.data_ptr(), gprob.data_ptr(), g.data_ptr(), n, d * h * w, ctx.c, vs_of(ctx.dtype), _stream()), "softmax2_bwd")
        else:
            check(lib.vs_softmax_cl_bwd(prob.data_ptr(), gprob.data_ptr(), g.data_ptr(), n, d * h * w, ctx.c, nc, vs_of(ctx.dtype), 0.0, 0, _stream()),
                  "softmax_cl_bwd")
        return g, None


_DROPOUT_CALLS = [0]


def next_dropout_seed():
    """A fresh 64-bit seed per dropout site per call, a pure function of torch's seed and a call counter (host side: no
    device sync).  Under HIP-graph replay the captured seeds would repeat, so drawing one during capture raises; TestTimeFinetune
    runs models with dropout > 0 eagerly."""
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        raise RuntimeError("dropout > 0 inside HIP-graph capture: the host-side mask seed would be baked into the graph and every replay "
                           "would reuse the same masks (the reference draws fresh ones per call) — run this model eagerly (graph=False)")
    _DROPOUT_CALLS[0] += 1
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + _DROPOUT_CALLS[0] * 0xD1B54A32D192ED03) % (1 << 64)


def dropout_mask(count, p, seed, device="cuda"):
    """The multipliers (0 or 1/(1-p)) a dropout site with this (p, seed) applies, in the site tensor's memory order (vs_dropout_mask)."""
    mask = torch.empty(count, dtype=torch.float32, device=device)
    check(lib.vs_dropout_mask(mask.data_ptr(), count, float(p), seed, _stream()), "dropout_mask")
    return mask


class Dropout(torch.autograd.Function):
    """F.dropout(x, p, training=True) on a final channels-last activation (joint_model.py:256-264,379-385)."""

    @staticmethod
    def forward(ctx, x, p, seed):
        _require_cuda(x)
        x = _contig(x)
        out = torch.empty_like(x)
        check(lib.vs_dropout(x.data_ptr(), out.data_ptr(), x.numel(), float(p), seed, vs_dtype(x), _stream()), "dropout")
        ctx.cfg = (float(p), seed)
        return out

    @staticmethod
    def backward(ctx, g):
        p, seed = ctx.cfg
        g = _contig(g)
        out = torch.empty_like(g)
        check(lib.vs_dropout(g.data_ptr(), out.data_ptr(), g.numel(), p, seed, vs_dtype(g), _stream()), "dropout")
        return out, None, None


class PackPlanar(torch.autograd.Function):
    """planar fp32 (N, C, D, H, W) -> channels-last (N, D, H, W, 8) in the kernel dtype (zero padded)."""

    @staticmethod
    def forward(ctx, x, dtype):
        _require_cuda(x)
        x = _contig(x.float())
        n, c, d, h, w = x.shape
        out = torch.empty((n, d, h, w, cpad(c)), dtype=dtype, device=x.device)
        check(lib.vs_pack_planar(x.data_ptr(), out.data_ptr(), n, d * h * w, c, cpad(c), vs_dtype(out), _stream()),
              "pack_planar")
        ctx.c = c
        return out

    @staticmethod
    def backward(ctx, g):
        g = _contig(g)
        n, d, h, w, cp = g.shape
        out = torch.empty((n, ctx.c, d, h, w), dtype=torch.float32, device=g.device)
        check(lib.vs_unpack_planar(g.data_ptr(), out.data_ptr(), n, d * h * w, ctx.c, cp, vs_dtype(g), _stream()),
              "unpack_planar")
        return out, None


class UnpackPlanar(torch.autograd.Function):
    """channels-last (N, D, H, W, Cp) -> planar fp32 (N, C, D, H, W), C <= Cp."""

    @staticmethod
    def forward(ctx, x, c):
        _require_cuda(x)
        x = _contig(x)
        n, d, h, w, cp = x.shape
        out = torch.empty((n, c, d, h, w), dtype=torch.float32, device=x.device)
        check(lib.vs_unpack_planar(x.data_ptr(), out.data_ptr(), n, d * h * w, c, cp, vs_dtype(x), _stream()), "unpack_planar")
        ctx.meta = (cp, x.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        cp, dtype = ctx.meta
        g = _contig(g.float())
        n, c, d, h, w = g.shape
        out = torch.empty((n, d, h, w, cp), dtype=dtype, device=g.device)
        check(lib.vs_pack_planar(g.data_ptr(), out.data_ptr(), n, d * h * w, c, cp, vs_dtype(out), _stream()), "pack_planar")
        return out, None


class LinearCL(torch.autograd.Function):
    """y = act(W x + b) where x is a channels-last activation read in the reference's NCDHW flatten order
    (joint_model.py:241-243).  y is fp32 (B, J)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        _require_cuda(x, weight)
        n, d, h, w, c = x.shape
        k_in, j_out = c * d * h * w, weight.shape[0]
        y = torch.empty((n, j_out), dtype=torch.float32, device=x.device)
        ctx.cl = (not weight.requires_grad) and d * h * w > 1       # frozen: contiguous channels-last copy of the weight, no permutation
        wt = frozen_linear_layout(weight, "cols_cl", c, d * h * w) if ctx.cl else weight
        check(lib.vs_linear_fwd(x.data_ptr(), vs_dtype(x), wt.data_ptr(), _p(bias), y.data_ptr(), n, k_in, j_out, 0 if ctx.cl else c,
                                0 if ctx.cl else d * h * w, 1 if relu else 0, _stream()), "linear_fwd")
        ctx.save_for_backward(x, weight, y if relu else None)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, yrelu = ctx.saved_tensors
        gy = _contig(gy.float())
        n, d, h, w, c = x.shape
        k_in, j_out = c * d * h * w, weight.shape[0]
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(weight) if ctx.needs_input_grad[1] else None
        gb = torch.empty(j_out, dtype=torch.float32, device=x.device) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        cl = ctx.cl and gw is None
        wt = frozen_linear_layout(weight, "cols_cl", c, d * h * w) if cl else weight
        check(lib.vs_linear_bwd(x.data_ptr(), vs_dtype(x), wt.data_ptr(), gy.data_ptr(), _p(yrelu), _p(gx), _p(gw),
                                _p(gb), n, k_in, j_out, 0 if cl else c, 0 if cl else d * h * w, _stream()), "linear_bwd")
        return gx, gw, gb, None


class LinearCLPair(torch.autograd.Function):
    """(act1(W1 x + b1), act2(W2 x + b2)) for two layers of equal shape on the same channels-last input — fc_mean and fc_std
    (joint_model.py:241-243) — in one forward launch; the backward runs LinearCL's kernels for the outputs that received a gradient."""

    @staticmethod
    def forward(ctx, x, w1, b1, relu1, w2, b2, relu2):
        _require_cuda(x, w1, w2)
        if w1.shape != w2.shape:
            raise ValueError("LinearCLPair needs two layers of the same shape")
        n, d, h, w, c = x.shape
        k_in, j_out = c * d * h * w, w1.shape[0]
        y1 = torch.empty((n, j_out), dtype=torch.float32, device=x.device)
        y2 = torch.empty((n, j_out), dtype=torch.float32, device=x.device)
        ctx.cl = (not w1.requires_grad) and (not w2.requires_grad) and d * h * w > 1
        wa = frozen_linear_layout(w1, "cols_cl", c, d * h * w) if ctx.cl else w1
        wb = frozen_linear_layout(w2, "cols_cl", c, d * h * w) if ctx.cl else w2
        check(lib.vs_linear_fwd_pair(x.data_ptr(), vs_dtype(x), wa.data_ptr(), _p(b1), y1.data_ptr(), 1 if relu1 else 0, wb.data_ptr(), _p(b2),
                                     y2.data_ptr(), 1 if relu2 else 0, n, k_in, j_out, 0 if ctx.cl else c, 0 if ctx.cl else d * h * w, _stream()),
              "linear_fwd_pair")
        ctx.save_for_backward(x, w1, w2, y1 if relu1 else None, y2 if relu2 else None)
        ctx.has_bias = (b1 is not None, b2 is not None)
        ctx.set_materialize_grads(False)
        return y1, y2

    @staticmethod
    def backward(ctx, g1, g2):
        x, w1, w2, r1, r2 = ctx.saved_tensors
        n, d, h, w, c = x.shape
        k_in, j_out = c * d * h * w, w1.shape[0]
        out = [None] * 7
        gx_total = None
        for gy, wt, yrelu, iw, ib, hb in ((g1, w1, r1, 1, 2, ctx.has_bias[0]), (g2, w2, r2, 4, 5, ctx.has_bias[1])):
            if gy is None:
                continue
            gy = _contig(gy.float())
            gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
            gw = torch.empty_like(wt) if ctx.needs_input_grad[iw] else None
            gb = torch.empty(j_out, dtype=torch.float32, device=x.device) if (hb and ctx.needs_input_grad[ib]) else None
            cl = ctx.cl and gw is None
            wuse = frozen_linear_layout(wt, "cols_cl", c, d * h * w) if cl else wt
            check(lib.vs_linear_bwd(x.data_ptr(), vs_dtype(x), wuse.data_ptr(), gy.data_ptr(), _p(yrelu), _p(gx), _p(gw), _p(gb), n, k_in,
                                    j_out, 0 if cl else c, 0 if cl else d * h * w, _stream()), "linear_bwd")
            out[iw], out[ib] = gw, gb
            if gx is not None:
                gx_total = gx if gx_total is None else gx_total + gx
        out[0] = gx_total
        return tuple(out)


class LinearToCL(torch.autograd.Function):
    """fc2 (joint_model.py:248-253): fp32 latent (B, K) -> channels-last activation (B, s, s, s, C) holding
    view(B, C, s, s, s) of the reference's output."""

    @staticmethod
    def forward(ctx, z, weight, bias, c, side, dtype):
        _require_cuda(z, weight)
        z = _contig(z.float())
        b, k_in = z.shape
        j_out = weight.shape[0]
        y = torch.empty((b, side, side, side, c), dtype=dtype, device=z.device)
        check(lib.vs_linear_fwd_perm_out(z.data_ptr(), weight.data_ptr(), _p(bias), y.data_ptr(), vs_dtype(y), b, k_in,
                                         j_out, c, side ** 3, _stream()), "linear_fwd_perm_out")
        ctx.save_for_backward(z, weight)
        ctx.has_bias = bias is not None
        ctx.geom = (c, side)
        return y

    @staticmethod
    def backward(ctx, gy):
        z, weight = ctx.saved_tensors
        gy = _contig(gy)
        c, side = ctx.geom
        b, k_in = z.shape
        j_out = weight.shape[0]
        gz = torch.empty_like(z) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(weight) if ctx.needs_input_grad[1] else None
        gb = torch.empty(j_out, dtype=torch.float32, device=z.device) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        if gz is not None and gw is None and gb is None and not weight.requires_grad and side > 1:
            # frozen fc2: gz[b][k] = sum_ph W2T[k][ph] * gy[b][ph] with the transposed, channels-last-ordered copy — the plain GEMV kernel,
            # every access contiguous (the permuting kernel walks a weight column with a 512-byte stride)
            wt = frozen_linear_layout(weight, "rows_cl_t", c, side ** 3)
            check(lib.vs_linear_fwd(gy.data_ptr(), vs_dtype(gy), wt.data_ptr(), None, gz.data_ptr(), b, j_out, k_in, 0, 0, 0, _stream()),
                  "linear_fwd (fc2 backward)")
            return gz, None, None, None, None, None
        check(lib.vs_linear_perm_out_bwd(z.data_ptr(), weight.data_ptr(), gy.data_ptr(), vs_dtype(gy), _p(gz), _p(gw),
                                         _p(gb), b, k_in, j_out, c, side ** 3, _stream()), "linear_perm_out_bwd")
        return gz, gw, gb, None, None, None


class Reparam(torch.autograd.Function):
    """z = mean + noise * std * scale (joint_model.py:248)."""

    @staticmethod
    def forward(ctx, mean, std, noise, scale):
        _require_cuda(mean, std, noise)
        z = torch.empty_like(mean)
        check(lib.vs_reparam_fwd(mean.data_ptr(), std.data_ptr(), noise.data_ptr(), float(scale), z.data_ptr(),
                                 mean.numel(), _stream()), "reparam_fwd")
        ctx.save_for_backward(noise)
        ctx.scale = float(scale)
        return z

    @staticmethod
    def backward(ctx, gz):
        (noise,) = ctx.saved_tensors
        gz = _contig(gz)
        gm = torch.empty_like(gz) if ctx.needs_input_grad[0] else None
        gs = torch.empty_like(gz) if ctx.needs_input_grad[1] else None
        check(lib.vs_reparam_bwd(gz.data_ptr(), noise.data_ptr(), ctx.scale, _p(gm), _p(gs), gz.numel(), _stream()),
              "reparam_bwd")
        return gm, gs, None, None


class KL(torch.autograd.Function):
    """utils/evaluation.py:42-45."""

    @staticmethod
    def forward(ctx, mean, std):
        _require_cuda(mean, std)
        mean, std = _contig(mean.float()), _contig(std.float())
        out = torch.empty((), dtype=torch.float32, device=mean.device)
        check(lib.vs_kl_fwd(mean.data_ptr(), std.data_ptr(), out.data_ptr(), mean.shape[0], mean.shape[1], _stream()), "kl_fwd")
        ctx.save_for_backward(mean, std)
        return out

    @staticmethod
    def backward(ctx, g):
        mean, std = ctx.saved_tensors
        g = _contig(g.float())
        gm = torch.empty_like(mean) if ctx.needs_input_grad[0] else None
        gs = torch.empty_like(std) if ctx.needs_input_grad[1] else None
        check(lib.vs_kl_bwd(mean.data_ptr(), std.data_ptr(), g.data_ptr(), _p(gm), _p(gs), mean.shape[0], mean.shape[1],
                            _stream()), "kl_bwd")
        return gm, gs


class Dice(torch.autograd.Function):
    """Soft Dice over channels [bot, top) of two planar fp32 (B, C, D, H, W) tensors.
    return_mean=True -> scalar mean over (b, c); False -> per-sample (B,) means (utils/evaluation.py:68-79)."""

    @staticmethod
    def forward(ctx, s, t, bot, top, eps, return_mean):
        _require_cuda(s, t)
        s, t = _contig(s.float()), _contig(t.float())
        b, c = s.shape[0], s.shape[1]
        voxels = s.numel() // (b * c)
        sums = torch.empty((b, c, 3), dtype=torch.float64, device=s.device)
        per = torch.empty(b, dtype=torch.float32, device=s.device)
        mean = torch.empty((), dtype=torch.float32, device=s.device)
        check(lib.vs_dice_fwd(s.data_ptr(), t.data_ptr(), sums.data_ptr(), per.data_ptr(), mean.data_ptr(), b, c, voxels,
                              bot, top, float(eps), _stream()), "dice_fwd")
        ctx.save_for_backward(s, t, sums)
        ctx.cfg = (bot, top, float(eps), bool(return_mean))
        return mean if return_mean else per

    @staticmethod
    def backward(ctx, g):
        s, t, sums = ctx.saved_tensors
        bot, top, eps, return_mean = ctx.cfg
        g = _contig(g.float())
        b, c = s.shape[0], s.shape[1]
        voxels = s.numel() // (b * c)
        gs = torch.empty_like(s) if ctx.needs_input_grad[0] else None
        gt = torch.empty_like(t) if ctx.needs_input_grad[1] else None
        if gs is not None or gt is not None:
            check(lib.vs_dice_bwd(s.data_ptr(), t.data_ptr(), sums.data_ptr(), g.data_ptr(), 1 if return_mean else 0,
                                  _p(gs), _p(gt), b, c, voxels, bot, top, eps, _stream()), "dice_bwd")
        return gs, gt, None, None, None, None


class LabelTarget:
    """A Dice target given as its LABEL volume (B,1,D,H,W; values 0..n_class-1): dice_loss_sum evaluates the one-hot (main_source.py:449-451) on
    the fly inside the loss kernels, so the one-hot tensor is neither written nor read (vs_dice_loss_multi_labels_*)."""

    def __init__(self, label):
        _require_cuda(label)
        self.label = _contig(label.float())


class DiceLossSum(torch.autograd.Function):
    """final = sum_j w[j] * (1 - avg_dsc(s, t_j))  — the loss line of every train method (main_source.py:469-471,
    main_target.py:588-592) — in TWO forward launches (per-block partials, finish) and ONE backward launch: the source is read once
    for all targets, no atomics, fixed summation order (the unfused spelling costs ~18 launches of 4.7 us for two terms).  -> (final, terms[k]); terms are the individual (1 - Dice) values, not differentiable."""

    @staticmethod
    def forward(ctx, s, bot, top, eps, weights, *targets):
        # weights: one number per target, or (number, True) for a target that is a LABEL volume (LabelTarget: one-hot on the fly)
        is_lab = [isinstance(w, tuple) and bool(w[1]) for w in weights]
        weights = [w[0] if isinstance(w, tuple) else w for w in weights]
        _require_cuda(s, *targets)
        k = len(targets)
        s = _contig(s.float())
        ts = [_contig(t.float()) for t in targets]
        b, c = s.shape[0], s.shape[1]
        voxels = s.numel() // (b * c)
        for t, lab in zip(ts, is_lab):
            if lab and t.numel() != b * voxels:
                raise ValueError("label target: expected %d x %d labels, got %s" % (b, voxels, tuple(t.shape)))
        scratch = torch.empty(lib.vs_dice_loss_multi_scratch_doubles(k, b, c), dtype=torch.float64, device=s.device)
        terms = torch.empty(k, dtype=torch.float32, device=s.device)
        final = torch.empty((), dtype=torch.float32, device=s.device)
        tp = (_ct.c_void_p * k)(*[None if lab else t.data_ptr() for t, lab in zip(ts, is_lab)])
        lp = (_ct.c_void_p * k)(*[t.data_ptr() if lab else None for t, lab in zip(ts, is_lab)])
        wp = (_ct.c_float * k)(*[float(w) for w in weights])
        check(lib.vs_dice_loss_multi_labels_fwd(s.data_ptr(), _ct.addressof(tp), _ct.addressof(lp), _ct.addressof(wp), k, scratch.data_ptr(),
                                                terms.data_ptr(), final.data_ptr(), b, c, voxels, bot, top, float(eps), _stream()), "dice_loss_multi_fwd")
        ctx.save_for_backward(s, scratch, *ts)
        ctx.cfg = (bot, top, float(eps), [float(w) for w in weights], is_lab)
        ctx.mark_non_differentiable(terms)
        ctx.set_materialize_grads(False)      # no zero-fill launch for the terms' gradient
        return final, terms

    @staticmethod
    def backward(ctx, g, _gterms):
        s, scratch = ctx.saved_tensors[:2]
        ts = ctx.saved_tensors[2:]
        bot, top, eps, weights, is_lab = ctx.cfg
        if g is None:
            return (None,) * (5 + len(ts))
        k = len(ts)
        b, c = s.shape[0], s.shape[1]
        voxels = s.numel() // (b * c)
        g = _contig(g.float())
        gs = torch.empty_like(s) if ctx.needs_input_grad[0] else None
        gts = [torch.empty_like(t) if ctx.needs_input_grad[5 + j] and not is_lab[j] else None for j, t in enumerate(ts)]
        if gs is not None or any(x is not None for x in gts):
            tp = (_ct.c_void_p * k)(*[None if lab else t.data_ptr() for t, lab in zip(ts, is_lab)])
            lp = (_ct.c_void_p * k)(*[t.data_ptr() if lab else None for t, lab in zip(ts, is_lab)])
            gp = (_ct.c_void_p * k)(*[_p(x) for x in gts])
            wp = (_ct.c_float * k)(*weights)
            check(lib.vs_dice_loss_multi_labels_bwd(s.data_ptr(), _ct.addressof(tp), _ct.addressof(lp), _ct.addressof(wp), k, scratch.data_ptr(),
                                                    g.data_ptr(), _p(gs), _ct.addressof(gp), b, c, voxels, bot, top, eps, _stream()), "dice_loss_multi_bwd")
        return (gs, None, None, None, None) + tuple(gts)


FUSED_LOSS = [os.environ.get("VS_FUSED_LOSS", "1") != "0"]


def dice_loss_sum(source, targets_and_weights, botindex=0, topindex=2, eps=1e-6):
    """sum_j w_j * (1 - avg_dsc(source, target_j, botindex, topindex)) -> (final, [1 - Dice_j ...]); channel selection as
    utils/evaluation.py:66-70.  VS_FUSED_LOSS=0 spells it with Dice.apply and torch scalar arithmetic, as the reference does."""
    channels = source.shape[1]
    bot, top = (botindex, min(topindex, channels)) if channels > 1 else (0, 1)
    if not FUSED_LOSS[0] or len(targets_and_weights) > 4:
        targets_and_weights = [(onehot(t.label, channels) if isinstance(t, LabelTarget) else t, w) for t, w in targets_and_weights]
        terms = [1 - Dice.apply(source, t, bot, top, eps, True) for t, _ in targets_and_weights]
        final = None
        for term, (_, w) in zip(terms, targets_and_weights):
            part = term if w == 1 else w * term
            final = part if final is None else final + part
        return final, terms
    final, terms = DiceLossSum.apply(source, bot, top, eps, [(w, isinstance(t, LabelTarget)) for t, w in targets_and_weights],
                                     *[t.label if isinstance(t, LabelTarget) else t for t, _ in targets_and_weights])
    return final, list(terms.unbind(0))


class BCE(torch.autograd.Function):
    """nn.BCELoss() (mean) — utils/evaluation.py:29-39."""

    @staticmethod
    def forward(ctx, p, t):
        _require_cuda(p, t)
        p, t = _contig(p.float()), _contig(t.float())
        out = torch.empty((), dtype=torch.float32, device=p.device)
        scratch = torch.empty(1, dtype=torch.float64, device=p.device)
        check(lib.vs_bce_fwd(p.data_ptr(), t.data_ptr(), out.data_ptr(), scratch.data_ptr(), p.numel(), _stream()), "bce_fwd")
        ctx.save_for_backward(p, t)
        return out

    @staticmethod
    def backward(ctx, g):
        p, t = ctx.saved_tensors
        gp = torch.empty_like(p)
        g = _contig(g.float())
        check(lib.vs_bce_bwd(p.data_ptr(), t.data_ptr(), g.data_ptr(), gp.data_ptr(), p.numel(), _stream()), "bce_bwd")
        return gp, None


# ------------------------------------------------------------------------------------------------
# non-differentiable helpers
# ------------------------------------------------------------------------------------------------
def onehot(label, n_class=2):
    """(B,1,D,H,W) labels (any real dtype, values 0..n_class-1) -> planar fp32 one-hot (main_source.py:449-451)."""
    _require_cuda(label)
    lab = _contig(label.float())
    b = lab.shape[0]
    out = torch.empty((b, n_class) + tuple(lab.shape[2:]), dtype=torch.float32, device=lab.device)
    check(lib.vs_onehot(lab.data_ptr(), out.data_ptr(), b, lab.numel() // b, n_class, _stream()), "onehot")
    return out


def hard_onehot(mask):
    """(B,C,D,H,W) scores -> planar fp32 one-hot of the channel argmax (utils/evaluation.py:58-64), any C >= 1, one launch."""
    _require_cuda(mask)
    m = _contig(mask.detach().float())
    b, c = m.shape[0], m.shape[1]
    out = torch.empty_like(m)
    check(lib.vs_hard_onehot(m.data_ptr(), out.data_ptr(), b, c, m.numel() // (b * c), _stream()), "hard_onehot")
    return out


def binarize(a, mode=0, lo=0.2, hi=0.8):
    _require_cuda(a)
    a = _contig(a.detach().float())
    out = torch.empty_like(a)
    check(lib.vs_binarize(a.data_ptr(), out.data_ptr(), a.numel(), mode, float(lo), float(hi), _stream()), "binarize")
    return out


def instnorm_stats(x):
    n, c = x.shape[0], x.shape[-1]
    st = _new_stats(n, c, x.device)
    check(lib.vs_instnorm_stats(x.data_ptr(), st.data_ptr(), n, x.numel() // (n * c), c, vs_dtype(x), _stream()),
          "instnorm_stats")
    return st
