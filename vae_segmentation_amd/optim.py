"""Multi-tensor optimiser steps on libvaeseg kernels: one launch per param group instead of the reference's
per-tensor torch.optim loops (main_source.py:279-294, main_target.py:347-352), and the EMA teacher update
(main_target.py:512-516).  Semantics and state keys follow torch.optim.SGD / torch.optim.Adam so optimizer
state_dicts stay interchangeable with the reference's checkpoints (main_source.py:827-843)."""
import torch

from . import ops
from ._lib import check, lib

_CHUNK = 4096        # must equal MT_CHUNK in csrc/misc.hip


class _Tables:
    """Device-side pointer / size / block tables for a list of tensor tuples; rebuilt only when an address changes."""

    def __init__(self):
        self.key = None
        self.dev = None
        self.n_blocks = 0

    def get(self, lists, device):
        key = tuple(t.data_ptr() for lst in lists for t in lst) + tuple(t.numel() for t in lists[0])
        if key != self.key:
            n = len(lists[0])
            ptrs = [torch.tensor([t.data_ptr() for t in lst], dtype=torch.int64) for lst in lists]
            sizes = torch.tensor([t.numel() for t in lists[0]], dtype=torch.int64)
            bm = []
            for i, t in enumerate(lists[0]):
                for start in range(0, t.numel(), _CHUNK):
                    bm += [i, start]
            bm = torch.tensor(bm, dtype=torch.int32)
            self.dev = [p.to(device) for p in ptrs] + [sizes.to(device), bm.to(device)]
            self.n_blocks = bm.numel() // 2
            self.key = key
        return self.dev, self.n_blocks


def _stream():
    return torch.cuda.current_stream().cuda_stream


class LossScaler:
    """Dynamic loss scaling for fp16 storage (BASELINE configs[4]) with torch.cuda.amp.GradScaler's protocol, entirely on the device:
    the scale, the overflow flag and the growth counter are device scalars, so a captured HIP graph follows them and nothing syncs.

        loss.backward(gradient=scaler.seed)        # every gradient comes out `scale` times too large
        optimizer.step(scaler=scaler)              # checks the gradients, divides by the scale, skips the step on inf / nan, updates the scale
    """

    def __init__(self, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, device="cuda"):
        self.scale = torch.full((1,), float(init_scale), dtype=torch.float32, device=device)
        self.found_inf = torch.zeros(1, dtype=torch.float32, device=device)
        self.tracker = torch.zeros(1, dtype=torch.int32, device=device)
        self.cfg = (float(growth_factor), float(backoff_factor), int(growth_interval))
        self._tabs = {}                  # one table per caller slot (param group): alternating groups must not rebuild each other's table

    @property
    def seed(self):
        """0-dim view of the scale: pass it as the upstream gradient of the (scalar) loss"""
        return self.scale.view(())

    def prepare(self, grads, slot=0):
        """build the device-side table check(grads, slot) needs (a stream capture cannot contain the host-to-device copy)"""
        return self._tabs.setdefault(slot, _Tables()).get([grads], grads[0].device)

    def check(self, grads, slot=0):
        (gp, sizes, bm), nb = self.prepare(grads, slot)
        check(lib.vs_grad_finite_multi(gp.data_ptr(), sizes.data_ptr(), bm.data_ptr(), nb, self.found_inf.data_ptr(), _stream()), "grad_finite_multi")

    def update(self):
        g, b, i = self.cfg
        check(lib.vs_loss_scale_update(self.scale.data_ptr(), self.tracker.data_ptr(), self.found_inf.data_ptr(), g, b, i, _stream()), "loss_scale_update")


class SGD(torch.optim.Optimizer):
    """torch.optim.SGD(lr, momentum, weight_decay) — dampening 0, no nesterov — as one kernel per group."""

    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self._tables = {}
        self._hyper = {}                 # group index -> [device float[3] = (lr, momentum, weight_decay), the host values it holds]

    @torch.no_grad()
    def step_with(self, params, grads, scaler=None, device_hyper=False):
        """step() with explicit gradient tensors (e.g. the views of a DDP flat bucket) instead of p.grad.
        device_hyper: the launch reads lr / momentum / weight decay from device memory (sync_hyper) instead of taking them as kernel
        arguments — the form a captured HIP graph needs to follow a schedule without re-capture."""
        return self.step(_override={id(p): g for p, g in zip(params, grads)}, scaler=scaler, _device_hyper=device_hyper)

    @torch.no_grad()
    def prepare(self, params, grads, scaler=None):
        """Build the device-side pointer tables, the momentum buffers and the device-resident hyperparameters for step_with(params, grads)
        WITHOUT updating anything: a stream capture that includes the optimiser (train.GraphedStep, captured tail) must find them ready —
        building them copies host tables to the device, which a capture cannot contain."""
        return self.step(_override={id(p): g for p, g in zip(params, grads)}, scaler=scaler, _dry=True)

    def hyper(self):
        return [(float(g["lr"]), float(g["momentum"]), float(g["weight_decay"])) for g in self.param_groups]

    @torch.no_grad()
    def sync_hyper(self):
        """Write each group's current (lr, momentum, weight_decay) into its device-resident triple if a scheduler moved them (stream-ordered
        before whatever is launched next: a graph replay that follows reads the new values).  -> number of groups rewritten."""
        moved = 0
        for gi, vals in enumerate(self.hyper()):
            ent = self._hyper.get(gi)
            if ent is not None and ent[1] != vals:
                ent[0].copy_(torch.tensor(vals, dtype=torch.float32), non_blocking=False)
                ent[1] = vals
                moved += 1
        return moved

    def _hyper_dev(self, gi, device):
        vals = self.hyper()[gi]
        ent = self._hyper.get(gi)
        if ent is None or ent[0].device != device:
            ent = self._hyper[gi] = [torch.tensor(vals, dtype=torch.float32).to(device), vals]
        return ent

    def capture_key(self, params):
        """what a captured step_with(params, ...) launch has baked in besides the gradient addresses: the parameter set and the momentum buffers"""
        return tuple((id(p), p.data_ptr(), self.state[p]["momentum_buffer"].data_ptr() if "momentum_buffer" in self.state.get(p, {}) else 0)
                     for p in params)

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0, _override=None, scaler=None, _dry=False, _device_hyper=False):
        ops.flush_wgrads()
        todo = []
        for gi, group in enumerate(self.param_groups):
            if _override is not None:
                ps = [p for p in group["params"] if id(p) in _override]
            else:
                ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("vae_segmentation_amd.optim.SGD needs contiguous fp32 CUDA parameters")
            if _override is not None:
                grads = [_override[id(p)] for p in ps]
            else:
                grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
            bufs = []
            for p in ps:
                st = self.state[p]
                if "momentum_buffer" not in st or st["momentum_buffer"] is None:
                    st["momentum_buffer"] = torch.zeros_like(p)      # buf = 0*m + g on the first step = torch's clone(g)
                bufs.append(st["momentum_buffer"])
            tab = self._tables.setdefault(gi, _Tables())
            (pp, gp, bp, sizes, bm), nb = tab.get([ps, grads, bufs], ps[0].device)
            todo.append((group, grads, pp, gp, bp, sizes, bm, nb, gi))
        if _dry:
            for t in todo:
                self._hyper_dev(t[8], t[2].device)
                if scaler is not None:
                    scaler.prepare(t[1], t[8])
            return None
        if scaler is not None:
            for t in todo:                                   # every group is checked before any group is updated
                scaler.check(t[1], t[8])
        sp, fp = (scaler.scale.data_ptr(), scaler.found_inf.data_ptr()) if scaler is not None else (None, None)
        for group, grads, pp, gp, bp, sizes, bm, nb, gi in todo:
            if _device_hyper:
                hyp = self._hyper_dev(gi, pp.device)[0]
                check(lib.vs_sgd_momentum_dev_multi(pp.data_ptr(), gp.data_ptr(), bp.data_ptr(), sizes.data_ptr(), bm.data_ptr(), nb,
                                                    hyp.data_ptr(), sp, fp, _stream()), "sgd_momentum_dev_multi")
                continue
            check(lib.vs_sgd_momentum_scaled_multi(pp.data_ptr(), gp.data_ptr(), bp.data_ptr(), sizes.data_ptr(), bm.data_ptr(), nb,
                                                   float(group["lr"]) * float(grad_scale), float(group["momentum"]),
                                                   float(group["weight_decay"]), 0, sp, fp, _stream()), "sgd_momentum_multi")
        if scaler is not None:
            scaler.update()
        ops.weights_changed()
        return None


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam(lr, betas, eps, weight_decay) as one kernel per group.
    Deviation under a LossScaler (fp16 only; the reference has no mixed precision): when the device-side overflow flag skips an update the
    host-side ``state['step']`` has still advanced (reading the flag would sync), so after a skipped step the bias corrections run one step
    ahead of torch.cuda.amp.GradScaler's."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = {}

    @torch.no_grad()
    def step_with(self, params, grads, scaler=None):
        return self.step(_override={id(p): g for p, g in zip(params, grads)}, scaler=scaler)

    @torch.no_grad()
    def step(self, closure=None, _override=None, scaler=None):
        ops.flush_wgrads()
        todo = []
        for gi, group in enumerate(self.param_groups):
            if _override is not None:
                ps = [p for p in group["params"] if id(p) in _override]
                grads = [_override[id(p)] for p in ps]
            else:
                ps = [p for p in group["params"] if p.grad is not None]
                grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
            if not ps:
                continue
            m1, m2 = [], []
            for p in ps:
                st = self.state[p]
                if "exp_avg" not in st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] = int(st["step"]) + 1
                m1.append(st["exp_avg"])
                m2.append(st["exp_avg_sq"])
            step = int(self.state[ps[0]]["step"])
            tab = self._tables.setdefault(gi, _Tables())
            (pp, gp, ap, vp, sizes, bm), nb = tab.get([ps, grads, m1, m2], ps[0].device)
            todo.append((group, grads, pp, gp, ap, vp, sizes, bm, nb, step))
        if scaler is not None:
            for i, t in enumerate(todo):
                scaler.check(t[1], i)
        sp, fp = (scaler.scale.data_ptr(), scaler.found_inf.data_ptr()) if scaler is not None else (None, None)
        for group, grads, pp, gp, ap, vp, sizes, bm, nb, step in todo:
            b1, b2 = group["betas"]
            check(lib.vs_adam_scaled_multi(pp.data_ptr(), gp.data_ptr(), ap.data_ptr(), vp.data_ptr(), sizes.data_ptr(), bm.data_ptr(),
                                           nb, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                           float(group["weight_decay"]), step, sp, fp, _stream()), "adam_multi")
        if scaler is not None:
            scaler.update()
        ops.weights_changed()
        return None


_EMA_TABLES = {}


@torch.no_grad()
def ema_update(teacher, student, alpha):
    """teacher <- alpha*teacher + (1-alpha)*student over matching state_dict entries (main_target.py:512-516)."""
    sd_t, sd_s = teacher.state_dict(), student.state_dict()
    ts = [sd_t[k] for k in sd_s if sd_s[k].dtype == torch.float32]
    ss = [sd_s[k] for k in sd_s if sd_s[k].dtype == torch.float32]
    tab = _EMA_TABLES.setdefault((id(teacher), id(student)), _Tables())
    (tp, sp, sizes, bm), nb = tab.get([ts, ss], ts[0].device)
    check(lib.vs_ema_multi(tp.data_ptr(), sp.data_ptr(), sizes.data_ptr(), bm.data_ptr(), nb, float(alpha), _stream()), "ema_multi")
    ops.clear_pack_cache()      # the teacher's cached packed weights are stale now ...
    ops.refresh_frozen_packs(teacher)   # ... and are re-packed in place at once: captured graphs that replay the teacher hold these addresses
