"""Data parallelism for the train step: one process per GPU, replicated parameters, ONE exchange step per train step —
RCCL all-reduces (torch.distributed backend "nccl" is RCCL on ROCm) of the trainable gradients, which live in one flat
fp32 buffer.  This replaces the reference's single-process nn.DataParallel (main_source.py:354, main_target.py:436):
its per-forward parameter broadcast and output gather disappear (replicas stay bit-identical because every rank
applies the same averaged gradient), and its gradient reduce-to-GPU0 becomes the all-reduce.

InstanceNorm is per-sample and the Dice/KL losses are per-sample means, so with equal per-rank batch the mean of
the per-rank losses equals the reference's global-batch loss and the averaged gradient equals its gradient.

Layout and overlap (MI355X: xGMI is point-to-point, the 9.1 MB exchange is latency- not bandwidth-bound, and the step is
2-3 ms, so what matters is that the exchange is not serialised behind the pass):
  * the grouped weight-gradient launches write dW / db STRAIGHT into the parameters' slices of the flat buffer
    (``param._vs_grad_view``; autograd adopts the slice as ``.grad``) — no gather launch, no second copy of the gradients;
  * the buffer is ordered in two buckets.  Bucket 0 holds the large tensors (>= ``split_numel`` elements: the 12^3 .. 3^3
    levels, 98 % of the bytes, whose weight gradients are cheap), bucket 1 the small ones (the 96^3 / 48^3 layers, whose
    weight gradients are the expensive ones, plus all biases).  At the end of backward the bucket-0 gradients are computed
    first (ops.set_wgrad_split), their all-reduce is started asynchronously (the process group's own RCCL stream), and the
    bucket-1 weight-gradient kernels — about half of the step's weight-gradient time — run underneath it; bucket 1 (a few
    hundred KB) follows.  Measured on one MI355X with one rank through RCCL (bench.py --force-dist, profiles/README.md):
    +0.01 ms per step for a single bucket after the pass, +0.06 ms for the two-phase form (the second graph launch).
"""
import os

import torch
import torch.distributed as dist

from ._lib import check, lib
from .optim import _Tables


_CAPTURABLE = {}
_AVG_OK = {}


def _probe_capture(group):
    """This rank's own attempt: capture (NOT replay) a tiny all-reduce of `group` into a HIP graph.  -> (captured?, replay callable).
    The communicator is set up by an eager collective first; nothing collective is replayed here, so a rank whose capture failed has issued
    exactly the same collectives as one whose capture succeeded (the agreement below must pair like with like)."""
    if dist.get_backend(group) != "nccl" or not torch.cuda.is_available():
        return False, None
    try:
        x = torch.ones(256, device="cuda")
        dist.all_reduce(x, group=group)                  # communicator set-up happens eagerly, outside the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                dist.all_reduce(x, group=group)
        torch.cuda.current_stream().wait_stream(side)

        def replay():
            g.replay()
            torch.cuda.synchronize()
            return bool(torch.isfinite(x).all())
        return True, replay
    except Exception as e:                               # noqa: BLE001 — whatever the stack refuses, the eager tail works
        import sys
        print("vae_segmentation_amd.ddp: all-reduce is not capturable on this stack (%s: %s); the exchange stays outside the step's graph"
              % (type(e).__name__, e), file=sys.stderr)
        try:
            torch.cuda.synchronize()
        except Exception:                                # noqa: BLE001
            pass
        return False, None


def _agree(ok, group):
    """MIN over the ranks of a yes/no answer (an eager collective every rank issues, whatever its own answer): one 'no' decides for all."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return bool(ok)
    dev = "cuda" if (dist.get_backend(group) == "nccl" and torch.cuda.is_available()) else "cpu"
    flag = torch.tensor([1.0 if ok else 0.0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item() > 0.5)


def collective_capturable(group=None):
    """Can an all-reduce of this process group be captured into a HIP graph?  Answered once per group: every rank tries to CAPTURE a tiny one,
    the ranks agree on the outcome (MIN), and only if every rank captured do they all REPLAY it — a collective again, so the same number of
    collectives is issued on every rank on every path — and agree once more on the result.  train.GraphedStep keeps the tail of the step
    eager when the answer is no."""
    if not dist.is_initialized():
        return True
    key = id(group)
    if key in _CAPTURABLE:
        return _CAPTURABLE[key]
    ok, replay = _probe_capture(group)
    ok = _agree(ok, group)                               # before anybody replays: a rank that could not capture has nothing to replay
    if ok:
        try:
            ok = bool(replay())
        except Exception:                                # noqa: BLE001
            ok = False
        ok = _agree(ok, group)
    if dist.get_backend(group) == "nccl" and torch.cuda.is_available():
        _let_watchdog_reap()
    _CAPTURABLE[key] = ok
    return ok


def average_supported(group=None):
    """Does the collective library reduce with ReduceOp.AVG?  Asked ONCE per group with an eager collective (every rank asks, so the ranks
    stay paired) and remembered: FlatGradSync.start() never has to discover it inside a stream capture, where a fallback would void the graph."""
    key = id(group)
    if key not in _AVG_OK:
        ok = False
        if dist.is_initialized() and dist.get_backend(group) == "nccl" and torch.cuda.is_available():
            try:
                x = torch.ones(4, device="cuda")
                dist.all_reduce(x, op=dist.ReduceOp.AVG, group=group)
                torch.cuda.synchronize()
                ok = bool((x == 1).all())
            except (RuntimeError, ValueError):           # a build without ncclAvg: sum, then scale
                ok = False
            ok = _agree(ok, group)
            _let_watchdog_reap()
        _AVG_OK[key] = ok
    return _AVG_OK[key]


_WATCHDOG_PERIOD_S = 0.1          # torch's ProcessGroupNCCL watchdog: one event query per collective in flight every 100 ms
_LAST_EAGER = [0.0]               # time.monotonic() of the last eager collective this package knows of


def note_eager_collective():
    """call after issuing an eager (un-captured) collective: quiesce_before_capture() measures its wait from here"""
    import time
    _LAST_EAGER[0] = time.monotonic()


def quiesce_before_capture(group=None):
    """torch's RCCL process group checks the completion of every eager collective from a watchdog thread (an event query every 100 ms).  A stream capture
    that starts before the watchdog has seen the last eager collective complete makes that query an "operation not permitted when stream is capturing"
    and takes the process down — seen when this module's probes ran right in front of GraphedStep's capture, and possible wherever an eager collective
    (the entry points' per-epoch barrier, broadcast_parameters, an eager step) precedes a (re-)capture.  So EVERY capture of a step goes
    through here first (train.GraphedStep._capture): the device is drained — every eager collective has completed — and the host then waits until at
    least three watchdog periods have passed since the last eager collective this package issued (two periods when it knows of none: one issued by
    other code just before the call), so the watchdog has polled the completed work and holds no event it would query during the capture.
    A no-op without an initialised RCCL group.  Captures are rare (once per GraphedStep, again when the live parameter set changes)."""
    import time
    if not (dist.is_available() and dist.is_initialized() and torch.cuda.is_available()):
        return
    try:
        if dist.get_backend(group) != "nccl":
            return
    except (RuntimeError, ValueError):
        return
    torch.cuda.synchronize()
    since = time.monotonic() - _LAST_EAGER[0]
    known_recent = _LAST_EAGER[0] != 0.0 and since <= 3 * _WATCHDOG_PERIOD_S
    time.sleep(3 * _WATCHDOG_PERIOD_S - since if known_recent else 2 * _WATCHDOG_PERIOD_S)


def _let_watchdog_reap():
    """after one of this module's once-per-group probe collectives"""
    note_eager_collective()
    quiesce_before_capture()


class FlatGradSync:
    """Flat gradient buffer + bucketed all-reduce.  Use:

        sync = FlatGradSync(params)                  # registers the gradient slots (overlap=True: and two-phase weight gradients)
        ... forward, backward ...
        sync()                                       # all-reduce (overlap=True: start(0) -> remaining weight gradients -> start(1) -> wait)
        optimizer.step_with(*sync.live())            # NOT (sync.params, views): a parameter without a gradient in this pass must not be updated

    train.GraphedStep drives the same phases around two captured graphs.  ``direct=False`` keeps the classic form: gradients
    wherever autograd put them, one gather launch (vs_copy_scale_multi), one all-reduce."""

    def __init__(self, params, process_group=None, direct=True, split_numel=8192, overlap=None, exchange=True):
        """exchange=False: no collective at all — the flat buffer only pins every gradient to a fixed address (train.GraphedStep's captured
        optimiser launch on one rank)."""
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.exchange = bool(exchange)
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        dev = self.params[0].device
        self.direct = bool(direct) and dev.type == "cuda"
        # overlap=False (default): ONE bucket, all-reduced on the launching stream right after the pass — measured cheapest at this step
        # size (one rank through RCCL on one MI355X: +0.01 ms per step, against +0.06 .. +0.2 ms for the two-phase overlapped forms,
        # whose second graph launch and cross-stream edges cost more than the ~9 MB collective they hide; profiles/README.md).
        # overlap=True / VS_DDP_OVERLAP=1: the two-bucket form described above.
        if overlap is None:
            overlap = os.environ.get("VS_DDP_OVERLAP", "0") == "1"
        self.overlap = bool(overlap)
        if not self.overlap:
            split_numel = 0
        big = [p for p in self.params if p.numel() >= split_numel] if self.direct else list(self.params)
        small = [p for p in self.params if p.numel() < split_numel] if self.direct else []
        order = big + small
        total = sum(p.numel() for p in order)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        slot, off = {}, 0
        for p in order:
            slot[id(p)] = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        n0 = sum(p.numel() for p in big)
        self.buckets = [self.flat[:n0], self.flat[n0:]] if small else [self.flat]
        self.views = [slot[id(p)] for p in self.params]                 # aligned with self.params
        self._first_ids = {id(p) for p in big}
        self._second_ids = {id(p) for p in small}
        self._tab, self._tab0 = _Tables(), _Tables()
        self._async = self.overlap
        self._works = []
        self._no_grad = set()           # ids of parameters whose .grad was None in the last pass (see _stragglers / live)
        self._avg = None                # ReduceOp.AVG available?  resolve_avg() asks once, eagerly
        if self.direct:
            from . import ops
            for p, v in zip(self.params, self.views):
                p._vs_grad_view = v
            if len(self.buckets) > 1:
                ops.set_wgrad_split(lambda w: id(w) in self._first_ids)

    def close(self):
        """Unregister the gradient slots (parameters go back to ordinary .grad tensors, weight gradients to one phase)."""
        if self.direct:
            from . import ops
            for p in self.params:
                if getattr(p, "_vs_grad_view", None) is not None:
                    p._vs_grad_view = None
            ops.set_wgrad_split(None)

    def broadcast_parameters(self, src=0):
        if self.world > 1:
            for p in self.params:
                dist.broadcast(p.data, src, group=self.group)
            note_eager_collective()

    # -- phases ---------------------------------------------------------------------------------------------------
    def live(self):
        """(params, views) of the parameters that take part in this step: requires_grad may be switched off for some of them after
        construction (embed_train freezes its Encoder on even epochs, main_source.py:550-554) — their slots are still exchanged (zeroed by
        _stragglers) but must never reach the optimiser.  The same holds for a parameter that received NO gradient in this pass
        (`.grad is None`): the single-GPU path and the reference's optimisers skip it, so it is skipped here too — no momentum drift or
        weight decay on a gradient that does not exist, and the same result at every world size."""
        keep = [i for i, p in enumerate(self.params) if p.requires_grad and id(p) not in self._no_grad]
        return [self.params[i] for i in keep], [self.views[i] for i in keep]

    def _stragglers(self, grads, only=None):
        """Gradients that did not land in their slot (accumulation into an existing .grad, hooks, direct=False, every non-conv parameter:
        Linear weights, BatchNorm affine): gathered by one multi-tensor copy.  In the steady state of the direct mode with conv-only
        networks this list is empty and nothing is launched.  A parameter that received NO gradient in this pass is remembered in
        self._no_grad and its slot is ZEROED: the slot is exchanged like the rest, so it must hold a well-defined contribution — if the ranks
        disagree on who got a gradient (data-dependent branches), the rank that has one applies g / world and not (stale average + g) / world
        (ADVICE r04); live() keeps the slot from THIS rank's optimiser.
        only: a set of parameter ids restricting the pass to one bucket."""
        src, dst = [], []
        for p, g, v in zip(self.params, grads, self.views):
            if only is not None and id(p) not in only:
                continue
            if g is None:
                if id(p) not in self._no_grad or (self.exchange and self.world > 1):     # on one rank a zeroed slot stays zero until a gradient lands in it
                    v.zero_()
                self._no_grad.add(id(p))
                continue
            self._no_grad.discard(id(p))
            if g.data_ptr() != v.data_ptr():
                src.append(g)
                dst.append(v)
        return src, dst

    def verify_live_agreement(self):
        """Every rank must skip the SAME parameters (those that received no gradient in the last pass): live() filters by this rank's own set, and a captured
        tail freezes it — ranks that disagreed would apply g / world on one side and nothing on the other, and the replicas would drift apart silently
        (ADVICE r05).  An eager collective (MIN and MAX of a hash of the set): called once per capture by train.GraphedStep (after its warm-up pass) and
        available to eager loops; raises on every rank when the ranks disagree.  A no-op on one rank."""
        if not self.exchange or self.world == 1 or not dist.is_initialized():
            return
        h = 0
        for i, p in enumerate(self.params):
            if id(p) in self._no_grad or not p.requires_grad:
                h = (h * 1000003 + i + 1) % 2147483647
        dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
        t = torch.tensor([h, -h], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        note_eager_collective()
        lo, hi = int(t[0].item()), -int(t[1].item())
        if lo != hi:
            raise RuntimeError("data-parallel ranks disagree on which parameters received a gradient in this pass (rank-dependent control flow in the "
                               "loss?): the update would differ per rank and the replicas diverge — make the pass identical on every rank")

    def gather(self, grads=None, tab=None, only=None):
        grads = [p.grad for p in self.params] if grads is None else grads
        src, dst = self._stragglers(grads, only)
        if not src:
            return
        if self.flat.is_cuda:
            (sp, dp, sizes, bm), nb = (tab or self._tab).get([src, dst], self.flat.device)
            check(lib.vs_copy_scale_multi(sp.data_ptr(), dp.data_ptr(), sizes.data_ptr(), bm.data_ptr(), nb, 1.0,
                                          torch.cuda.current_stream().cuda_stream), "copy_scale_multi")
        else:   # gloo / CPU tensors: host-side staging for the multi-process CPU tests of the exchange logic
            for v, g in zip(dst, src):
                v.copy_(g)

    def start(self, i):
        """Start the all-reduce of bucket i, ordered after everything issued so far on the current stream, WITHOUT blocking that stream:
        torch's RCCL process group runs collectives on its own stream (async_op=True), so kernels launched next — the bucket-1
        weight gradients — run underneath it.  (A communication stream of our own was measured and dropped: its extra cross-stream
        edges cost 0.14 ms per step on top of the collective's.)"""
        if not self.exchange or (self.world == 1 and not dist.is_initialized()):
            return
        b = self.buckets[i]
        if not b.is_cuda:                                            # CPU tensors (gloo tests)
            dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group)
            b.mul_(1.0 / self.world)
            return
        if not torch.cuda.is_current_stream_capturing():
            note_eager_collective()                                 # an eager step's exchange: a (re-)capture that follows must wait for the watchdog (quiesce_before_capture)
        if self.resolve_avg():
            self._works.append((dist.all_reduce(b, op=dist.ReduceOp.AVG, group=self.group, async_op=self._async), None))   # RCCL averages in the reduction
            return
        self._works.append((dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group, async_op=self._async), b if self.world > 1 else None))

    def resolve_avg(self):
        if self._avg is None:
            self._avg = average_supported(self.group)
        return self._avg

    def wait(self):
        """Make the current stream wait for the started collectives (and apply the 1/world scale where the collective only sums)."""
        for work, b in self._works:
            if work is not None:
                work.wait()
            if b is not None:
                check(lib.vs_scale_copy(b.data_ptr(), b.data_ptr(), b.numel(), 1.0 / self.world, torch.cuda.current_stream().cuda_stream), "scale_copy")
        self._works = []

    def __call__(self, grads=None):
        """Whole exchange after a backward pass.  grads: tensors aligned with self.params (default: p.grad).  Returns the
        averaged-gradient views (aligned with self.params)."""
        from . import ops
        grads = [p.grad for p in self.params] if grads is None else grads
        if len(self.buckets) == 1:
            if self.flat.is_cuda:
                ops.flush_wgrads()
            self.gather(grads)
            self.start(0)
        else:
            self.gather(grads, self._tab0, only=self._first_ids)    # bucket-0 stragglers (fc weights ...) BEFORE its all-reduce starts
            self.start(0)
            ops.flush_wgrads()                                       # bucket-1 weight gradients run under bucket 0's all-reduce
            self.gather(grads, only=self._second_ids)
            self.start(1)
        self.wait()
        return self.views
