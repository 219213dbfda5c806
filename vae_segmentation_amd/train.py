"""Train-step bodies of the reference's main_source.py / main_target.py on the native modules, plus a
HIP-graph replayed step.  Each *_losses function mirrors the loss arithmetic of one ``--method``:

  joint_train_losses        main_source.py:449-471     lambda_vae*(1-Dice(pred,recon)) + (1-Dice(pred,gt)), eps 1e-4
  seg_train_losses          main_source.py:421-441
  vae_train_losses          main_source.py:389-413     (1-Dice(recon,gt)) + 2e-5*KL, z = mean + noise*std*0.35
  domain_adaptation_losses  main_target.py:520-596     student/teacher, domain_loss_type 0 / 8 / 9, eps 1e-6
  finetune_losses / TestTimeFinetune   main_target.py:809-953   per-case test-time training + hard-Dice validation
  embed_train_losses / refine_vae_losses / sep_joint_train_losses    main_source.py:546-659
  discriminator_train_loss / domain_adaptation_dis_losses            main_target.py:491-501, 696-732
"""
import os

import torch

from . import ops
from .evaluation import EPS_EVALUATION, EPS_MAIN_SOURCE, KLloss, avg_dsc, binarize, confident_binarize


# The ground truth of these two methods feeds the loss only: its one-hot (main_source.py:449-451) is evaluated inside the loss kernels from the
# label volume (ops.LabelTarget) instead of being written as a tensor first — one launch and 21 MB per step at 96^3.  VS_LAZY_ONEHOT=0 materialises
# batch["gt"] as the reference does (the other methods, whose networks or losses read batch["gt"], always do).
LAZY_ONEHOT = os.environ.get("VS_LAZY_ONEHOT", "1") != "0"


def _gt(batch, label, n_class):
    if LAZY_ONEHOT and label.is_cuda:
        batch["label"] = label
        return ops.LabelTarget(label)
    batch["gt"] = ops.onehot(label, n_class)
    return batch["gt"]


def joint_train_losses(joint, img, label, lambda_vae=0.1, eps=EPS_MAIN_SOURCE, n_class=2):
    batch = {"img": img}
    gt = _gt(batch, label, n_class)
    batch = joint(batch, "img", "pred", "recon")
    # final = lambda_vae * (1 - avg_dsc(pred, recon)) + (1 - avg_dsc(pred, gt))        (main_source.py:469-471), one launch each way
    final, (recon_loss, dsc_loss) = ops.dice_loss_sum(batch["pred"], [(batch["recon"], lambda_vae), (gt, 1.0)],
                                                      botindex=1, topindex=n_class, eps=eps)
    return final, {"recon_loss": recon_loss, "dice_loss": dsc_loss, "batch": batch}


def seg_train_losses(seg, img, label, eps=EPS_MAIN_SOURCE, n_class=2):
    batch = {"img": img}
    gt = _gt(batch, label, n_class)
    batch = seg(batch, "img", "pred")
    final, (dsc_loss,) = ops.dice_loss_sum(batch["pred"], [(gt, 1.0)], botindex=1, topindex=n_class, eps=eps)
    return final, {"dice_loss": dsc_loss, "batch": batch}


def vae_train_losses(vae, label, scale=0.35, noise=None, eps=EPS_MAIN_SOURCE, n_class=2):
    gt = ops.onehot(label, n_class)
    recon, mean, std = vae(gt, if_random=True, scale=scale, noise=noise)
    batch = {"gt": gt, "recon": recon, "mean": mean, "std": std}
    kl = KLloss(batch)
    dsc_loss = 1 - avg_dsc(batch, "recon", "gt", botindex=1, topindex=n_class, eps=eps)
    final = dsc_loss + 0.00002 * kl
    return final, {"dice_loss": dsc_loss, "kl_loss": kl, "batch": batch}


def lambda_schedule(recon_loss, lambda_vae):
    """main_target.py:551-554 — reads the loss on the host, as the reference's `if recon_loss < 0.15` does."""
    r = float(recon_loss)
    if r < 0.15:
        return lambda_vae * 0.6
    if r < 0.225:
        return lambda_vae * 1.2
    if r < 0.3:
        return lambda_vae * 2.0
    return lambda_vae * 3.0


def domain_adaptation_losses(student, teacher, img, label, lambda_vae=1.0, domain_loss_type=0, kl=False,
                             use_confident_binarize=False, eps=EPS_EVALUATION, n_class=2, only_pseudo=False, epoch=0,
                             turn_epoch=-1, lambda_vae_warmup=0, host_schedule=True):
    """main_target.py:520-596.  The branch order is the reference's: only_pseudo (:548-549), domain_loss_type 8 / 15 / 16 (:550-560),
    9 (:561-566), 11 - 14 (:571-582), --turn_epoch alternation (:583-587), then the default with its --lambda_vae_warmup ramp (:588-592).
    Type 10 reads a tensor of the validation loop that does not exist at that point of the reference (`val_batch`, :568) and raises there too.
    host_schedule=False evaluates the `if recon_loss < 0.15 ...` ladder of types 8 / 9 on the device (lambda_schedule_device) instead of
    reading the loss on the host as the reference does: same value, no sync — the form a captured HIP graph needs."""
    batch = {"img": img, "gt": ops.onehot(label, n_class)}
    batch = student(batch, "img", "pred", "recon", dropout=True)
    with torch.no_grad():
        batch = teacher(batch, "img", "fake", "_unused")          # also (re)sets batch["mean"/"std"] (joint_model.py:451)
    batch["fake"] = confident_binarize(batch["fake"]) if use_confident_binarize else binarize(batch["fake"])
    recon_loss = 1 - avg_dsc(batch, "pred", "recon", botindex=1, topindex=n_class, eps=eps)
    klloss = KLloss(batch)
    dsc_loss = 1 - avg_dsc(batch, "pred", "gt", botindex=1, topindex=n_class, eps=eps)
    fake_loss = 1 - avg_dsc(batch, "pred", "fake", botindex=1, topindex=n_class, eps=eps)
    if only_pseudo:
        final = fake_loss
    elif domain_loss_type in (8, 15, 16, 9) and not host_schedule:
        final = finetune_loss(recon_loss, fake_loss, klloss, lambda_vae, 9 if domain_loss_type == 9 else 8, kl, False)
    elif domain_loss_type in (8, 15, 16):
        cur = lambda_schedule(recon_loss.detach(), lambda_vae)
        if cur > 1:
            final = recon_loss + (klloss if kl else 0) + 1 / cur * fake_loss
        else:
            final = cur * (recon_loss + (klloss if kl else 0)) + fake_loss
    elif domain_loss_type == 9:
        cur = lambda_schedule(recon_loss.detach(), lambda_vae)
        final = (cur * recon_loss + fake_loss) / (1 + cur)
    elif domain_loss_type == 11:
        final = lambda_vae * recon_loss + fake_loss + recon_loss * fake_loss
    elif domain_loss_type == 12:
        final = lambda_vae * recon_loss + fake_loss - recon_loss * fake_loss
    elif domain_loss_type == 13:
        recon_loss = torch.clamp(recon_loss - 0.15, min=0)        # `recon_loss -= 0.15; recon_loss[recon_loss < 0] = 0`
        final = lambda_vae * recon_loss
    elif domain_loss_type == 14:
        recon_loss = torch.clamp(recon_loss - 0.1, min=0)
        final = lambda_vae * recon_loss + fake_loss
    elif domain_loss_type == 10:
        raise NotImplementedError("domain_loss_type 10 reads `val_batch`, undefined in the reference's training loop (main_target.py:568)")
    elif turn_epoch != -1:              # every other domain_loss_type (1-7, ...) falls through to the default branches, as in the reference (:583-592)
        final = lambda_vae * recon_loss if (epoch // turn_epoch) % 2 == 0 else lambda_vae * recon_loss + fake_loss
    elif epoch >= lambda_vae_warmup:
        final = lambda_vae * recon_loss + fake_loss
        if kl:
            final = final + 0.00002 * lambda_vae * klloss
    else:
        final = lambda_vae * epoch / lambda_vae_warmup * recon_loss + fake_loss
    return final, {"recon_loss": recon_loss, "kl_loss": klloss, "dice_loss": dsc_loss, "dice_loss_fake": fake_loss,
                   "batch": batch}


def domain_adaptation_pseudo_losses(student, teacher, img, label, lambda_vae=1.0, domain_loss_type=0, use_confident_binarize=False,
                                    eps=EPS_EVALUATION, n_class=2, host_schedule=True):
    """main_target.py:615-661 — the domain-adaptation step of a run started with --pseudo_list (a second, pseudo-labelled loader exists): the same two
    forwards and three Dice terms as domain_adaptation_losses, but its own, shorter ladder of final losses: domain_loss_type 8 (:636-647), lambda_vae >= 1000
    (recon * lambda / 10000, :648-649), else lambda * recon + fake (:650-651); no KL term, no turn_epoch / warm-up branches."""
    batch = {"img": img, "gt": ops.onehot(label, n_class)}
    batch = student(batch, "img", "pred", "recon", dropout=True)
    with torch.no_grad():
        batch = teacher(batch, "img", "fake", "_unused")
    batch["fake"] = confident_binarize(batch["fake"]) if use_confident_binarize else binarize(batch["fake"])
    recon_loss = 1 - avg_dsc(batch, "pred", "recon", botindex=1, topindex=n_class, eps=eps)
    dsc_loss = 1 - avg_dsc(batch, "pred", "gt", botindex=1, topindex=n_class, eps=eps)
    fake_loss = 1 - avg_dsc(batch, "pred", "fake", botindex=1, topindex=n_class, eps=eps)
    if domain_loss_type == 8 and not host_schedule:
        final = finetune_loss(recon_loss, fake_loss, None, lambda_vae, 8, False, False)
    elif domain_loss_type == 8:
        cur = lambda_schedule(recon_loss.detach(), lambda_vae)
        final = recon_loss + 1 / cur * fake_loss if cur > 1 else cur * recon_loss + fake_loss
    elif lambda_vae >= 1000:
        final = recon_loss * lambda_vae / 10000
    else:
        final = lambda_vae * recon_loss + fake_loss
    return final, {"recon_loss": recon_loss, "dice_loss_fake": fake_loss, "dice_loss": dsc_loss, "batch": batch}


@torch.no_grad()
def pseudo_batch_losses(student, img, label, eps=EPS_EVALUATION, n_class=2):
    """main_target.py:667-687 — the batch of the pseudo-labelled loader that follows every such step: one student forward (dropout on, as in the reference) and
    the two Dice terms against its own reconstruction and the pseudo label.  The reference computes `final_loss = dsc_loss` here and then only LOGS it — there
    is no backward / optimizer.step() behind it (:681-687) — so this is a forward-only evaluation (no autograd graph is recorded)."""
    batch = {"img": img, "gt": ops.onehot(label, n_class)}
    batch = student(batch, "img", "pred", "recon", dropout=True)
    recon_loss = 1 - avg_dsc(batch, "pred", "recon", botindex=1, topindex=n_class, eps=eps)
    dsc_loss = 1 - avg_dsc(batch, "pred", "gt", botindex=1, topindex=n_class, eps=eps)
    return {"recon_loss_pseudo": recon_loss, "dice_loss_pseudo": dsc_loss, "final_loss_pseudo": dsc_loss}


# ----------------------------------------------------------------------------------------------------
# remaining train methods of the reference: loss bodies on the native modules
# ----------------------------------------------------------------------------------------------------
def embed_train_losses(embed, img, label, eps=EPS_MAIN_SOURCE, n_class=2, noise=None):
    """main_source.py:546-590 (`embed_train`): Embed(Encoder, VAE, Fusion) in test_mode; the caller toggles the Encoder's requires_grad by
    epoch parity (:550-554).  final = (dsc1 + dsc2 + inpaint) / 3 + mse / 10 + 2e-5 * KL + recon."""
    gt = ops.onehot(label, n_class)
    batch = embed({"img": img, "venous_pancreas_only": gt}, "img", "pred", test_mode=True, noise=noise)
    batch["gt"] = gt
    d = lambda key: 1 - avg_dsc(batch, key, "gt", botindex=1, topindex=n_class, eps=eps)
    dsc1, dsc2, recon, inpaint = d("pred"), d("init_seg"), d("gt_recon"), d("seg_recon")
    kl = KLloss(batch, mean_key="latent_code_gt", std_key="latent_code_std")
    mse = torch.mean((batch["latent_code"] - batch["latent_code_gt"]) ** 2)                    # nn.MSELoss() on two (B, dim) codes
    final = (dsc1 + dsc2 + inpaint) / 3 + mse / 10 + 0.00002 * kl + recon
    return final, {"dice_loss1": dsc1, "dice_loss2": dsc2, "mse_loss": mse, "kl_loss": kl, "recon_loss": recon, "inpaint_loss": inpaint,
                   "batch": batch}


def refine_vae_losses(embed, img, label, eps=EPS_MAIN_SOURCE, n_class=2, noise=None):
    """main_source.py:591-628 (`refine_vae`): Encoder frozen by the caller (:596-597); final = inpaint + 2e-5 * KL + recon."""
    gt = ops.onehot(label, n_class)
    batch = embed({"img": img, "venous_pancreas_only": gt}, "img", "pred", test_mode=True, noise=noise)
    batch["gt"] = gt
    d = lambda key: 1 - avg_dsc(batch, key, "gt", botindex=1, topindex=n_class, eps=eps)
    recon, inpaint, init = d("gt_recon"), d("seg_recon"), d("init_seg")
    kl = KLloss(batch, mean_key="latent_code_gt", std_key="latent_code_std")
    final = inpaint + 0.00002 * kl + recon
    return final, {"recon_loss": recon, "inpaint_loss": inpaint, "kl_loss": kl, "init_loss": init, "batch": batch}


def sep_joint_train_losses(joint, teacher, img, label, eps=EPS_MAIN_SOURCE, n_class=2):
    """main_source.py:629-659 (`sep_joint_train`): per-sample Dice scores (return_mean=False) of the student against its own VAE
    reconstruction, of the teacher against its reconstruction, and of the student against the teacher's prediction, the latter weighted by
    the teacher's squared reconstruction score:  final = 0.1 * (1 - mean(recon)) + 1 - mean(dsc * recon_tea^2).
    (The reference leaves autograd on for the teacher pass; its parameters are frozen and its inputs carry no gradient, so no_grad here
    changes nothing but memory.)"""
    batch = {"img": img, "gt": ops.onehot(label, n_class)}
    batch = joint(batch, "img", "pred", "recon")
    with torch.no_grad():
        tb = teacher({"img": img}, "img", "pred_tea", "recon_tea")
    batch["pred_tea"], batch["recon_tea"] = tb["pred_tea"], tb["recon_tea"]
    kw = dict(botindex=1, topindex=n_class, return_mean=False, eps=eps)
    recon = avg_dsc(batch, "pred", "recon", **kw)
    recon_tea = avg_dsc(batch, "pred_tea", "recon_tea", **kw)
    dsc = avg_dsc(batch, "pred", "pred_tea", **kw)
    final = 0.1 * (1 - torch.mean(recon)) + 1 - torch.mean(dsc * recon_tea ** 2)
    return final, {"recon_loss": 1 - torch.mean(recon), "dice_loss": 1 - torch.mean(dsc), "batch": batch}


def discriminator_train_loss(dis, mask, score):
    """main_target.py:491-501 (`discriminator_train`): the Encoder regresses a quality score of a (B,1,D,H,W) mask; mean squared error."""
    out = dis(mask)
    final = torch.mean((score.to(out) - out) ** 2)
    return final, {"final_loss": final, "score_out": out}


def domain_adaptation_dis_losses(student, teacher_seg, img, label, lambda_vae=1.0, epoch=1, lambda_vae_warmup=0,
                                 use_confident_binarize=False, eps=EPS_EVALUATION, n_class=2):
    """main_target.py:696-732 (`domain_adaptation_dis`): student = Joint2(Seg, Dis) with dropout flag on, teacher = a frozen Segmentation
    giving the pseudo-label; final = lambda * (1 - mean(score)) + (1 - Dice(pred, pseudo)), lambda ramped over lambda_vae_warmup epochs."""
    batch = {"img": img, "gt": ops.onehot(label, n_class)}
    batch = student(batch, "img", "pred", "score", dropout=True)
    with torch.no_grad():
        batch = teacher_seg(batch, "img", "fake")
    batch["fake"] = confident_binarize(batch["fake"]) if use_confident_binarize else binarize(batch["fake"])
    dsc_loss = 1 - avg_dsc(batch, "pred", "gt", botindex=1, topindex=n_class, eps=eps)
    fake_loss = 1 - avg_dsc(batch, "pred", "fake", botindex=1, topindex=n_class, eps=eps)
    dis_loss = 1 - batch["score"].mean()
    lam = lambda_vae if epoch >= lambda_vae_warmup else lambda_vae * epoch / lambda_vae_warmup
    final = lam * dis_loss + fake_loss
    return final, {"discriminator_loss": dis_loss, "dice_loss_fake": fake_loss, "dice_loss": dsc_loss, "batch": batch}


def capture_safe_accumulators(params):
    """The AccumulateGrad node of every parameter, guaranteed to have been created under the CURRENT stream (call this under the side
    stream the warm-up runs on) or by an earlier call of this function; the caller keeps the returned list alive through the capture.

    Why: autograd creates a parameter's AccumulateGrad node lazily, records the stream that was current at that moment, and re-uses the
    node for as long as any autograd graph references it.  A node born in an eager pass on the default stream — kept alive by a loss or
    an output somebody still holds — makes the captured backward hop to the legacy default stream, which cannot join a stream capture:
    hipStreamEndCapture then crashes the process (seen with a test that kept `loss, aux` of an eager step).  A node nobody else owns is
    simply re-created here; one that survives without a reference of ours is reported."""
    accs, stale = [], []
    for p in params:
        if not p.requires_grad:
            continue
        with torch.enable_grad():
            acc = p.view_as(p).grad_fn.next_functions[0][0]
            if not acc.metadata.get("vs_capture_safe"):
                acc.metadata["vs_probe"] = True
                del acc
                acc = p.view_as(p).grad_fn.next_functions[0][0]          # the same node only if an older autograd graph owns it
                if acc.metadata.get("vs_probe"):
                    stale.append(p)
                else:
                    acc.metadata["vs_capture_safe"] = True
        accs.append(acc)
    if stale:
        raise RuntimeError("GraphedStep: %d parameter(s) are still referenced by the autograd graph of an earlier eager pass (a loss or output "
                           "that is still alive); their gradient accumulators belong to the default stream and a stream capture cannot "
                           "include it.  Drop those tensors (del loss, aux) — or build the GraphedStep first — and try again." % len(stale))
    return accs


class GraphedStep:
    """zero_grad -> forward -> losses -> backward captured once into a HIP graph and replayed per step; the
    optimiser (one multi-tensor kernel) and, under data parallelism, the gradient all-reduces run eagerly around the replays.

    ``loss_fn()`` must read its inputs from tensors that stay at fixed addresses (copy new data into them).
    ``grad_sync``: a ddp.FlatGradSync.  With its two buckets the pass is captured as TWO graphs — [forward, backward, bucket-0
    weight gradients] and [bucket-1 weight gradients] — and a step is: replay 1, start the all-reduce of bucket 0 on the
    communication stream, replay 2 underneath it, all-reduce bucket 1, wait, optimiser on the averaged views.
    ``scaler``: an optim.LossScaler (fp16 storage) — backward is seeded with its device-resident scale, the optimiser unscales / skips.
    ``capture_tail`` (default: on, VS_GRAPH_TAIL=0 switches it off): the TAIL of the step — the gradient all-reduce (RCCL collectives capture into
    a HIP graph on this stack: tools/rccl_capture_probe.py), [the LossScaler's finite check,] the SGD launch[, the loss-scale update] and the
    re-pack of the trainable weight images — is part of the same graph, so a step is ONE graph launch and nothing else.  Conditions (otherwise
    the tail stays eager, `self.tail` says which): one bucket (not the two-phase overlapped form), optim.SGD (Adam's bias corrections are
    host-side per-step state), every gradient written straight into its flat slot (no stragglers to gather), the process group is RCCL or absent.
    The learning rate, momentum and weight decay are read by the captured launch from DEVICE memory (optim.SGD.sync_hyper rewrites the three
    floats when a scheduler moved them: main_source.py:674-677 — no re-capture); what IS baked in — the set of live parameters and the
    addresses of parameters and momentum buffers — is compared on every step() and a change (requires_grad toggled, load_state_dict) re-captures
    (`self.recaptures`).  A LossScaler's scale, overflow flag and growth tracker are device scalars, so the scaled step captures as it is.
    Models with dropout > 0 cannot be captured (ops.next_dropout_seed raises during capture): run them eagerly."""

    def __init__(self, loss_fn, params, optimizer, grad_sync=None, warmup=2, scaler=None, capture_tail=None):
        from . import ddp as _ddp
        from . import optim as _optim
        self.loss_fn, self.params, self.optimizer, self.grad_sync = loss_fn, list(params), optimizer, grad_sync
        self._one = None
        self.scaler = scaler                     # optim.LossScaler for fp16 storage: its scale is a device scalar the captured backward reads
        self.graph = self.graph2 = None
        self.loss = None
        self.aux = None
        self.recaptures = 0
        self.tail_fallback = False               # True: the tail was meant to be captured and the capture failed (see _capture_guarded)
        two_phase = self._two_phase = grad_sync is not None and len(grad_sync.buckets) > 1
        if capture_tail is None:
            capture_tail = os.environ.get("VS_GRAPH_TAIL", "1") != "0"
        import torch.distributed as dist
        # grad_sync None = the caller wants no exchange (one rank, or a timing leg): nothing collective is captured then
        rccl_or_none = (grad_sync is None or (not dist.is_initialized()) or not getattr(grad_sync, "exchange", True)
                        or (dist.get_backend(grad_sync.group) == "nccl" and _ddp.collective_capturable(grad_sync.group)))
        # round 6: the two-bucket overlapped form captures its tail too — ONE graph whose bucket-0 all-reduce node (the process group's stream, forked inside
        # the capture) runs beside the bucket-1 weight-gradient nodes: no second graph launch, no eager cross-stream edges
        self.tail = (bool(capture_tail) and warmup >= 1 and isinstance(optimizer, _optim.SGD)
                     and rccl_or_none and all(p.is_cuda for p in self.params))
        if two_phase and os.environ.get("VS_GRAPH_TAIL_OVERLAP", "1") == "0":
            self.tail = False                    # A/B: the round-5 form (two graphs, eager exchange between them)
        if grad_sync is not None and getattr(grad_sync, "exchange", True) and dist.is_initialized():
            grad_sync.resolve_avg()              # one eager probe collective per group, HERE — well before the capture (ddp._let_watchdog_reap says why)
        self._own_sync = False
        if self.tail and grad_sync is None:
            # one rank, no exchange: the flat buffer still serves — it gives every gradient a FIXED address, which the captured optimiser launch needs
            self.grad_sync = grad_sync = _ddp.FlatGradSync(self.params, overlap=False, exchange=False)
            self._own_sync = True
        self._warm(warmup)
        self._capture_guarded()

    def _capture_guarded(self):
        """_capture(); when the captured tail holds a collective (RCCL all-reduce inside the graph), a capture that fails on ANY rank sends EVERY rank to the
        eager tail instead of ending the job: the ranks agree on the outcome with one eager collective (nothing collective was executed by the capture
        itself, on the rank that failed or on the ones that did not, so they stay paired), the pass is warmed once more and captured WITHOUT the tail —
        step() then runs exchange, optimiser and re-pack eagerly behind the replay (the round-3 form).  ddp.collective_capturable() probes a tiny
        all-reduce beforehand; this covers what the probe cannot (the real bucket, more than one rank, the stack's state at that moment)."""
        import torch.distributed as dist
        from . import ddp as _ddp
        s = self.grad_sync
        if not (self.tail and s is not None and getattr(s, "exchange", True) and dist.is_initialized()):
            return self._capture()
        err = None
        try:
            self._capture()
        except Exception as e:                   # noqa: BLE001 — whatever the stack refused inside the capture
            err = e
            if torch.cuda.is_current_stream_capturing():
                raise                            # the failed capture could not even be ended: nothing to fall back to in this process
            try:
                torch.cuda.synchronize()
            except Exception:                    # noqa: BLE001
                pass
        ok = _ddp._agree(err is None, s.group)
        _ddp.note_eager_collective()
        if ok:
            return
        import sys
        print("vae_segmentation_amd.train: capturing the step's tail (all-reduce + optimiser) failed on %s (%s); every rank keeps the tail eager"
              % ("this rank" if err is not None else "another rank", "%s: %s" % (type(err).__name__, err) if err is not None else "-"), file=sys.stderr, flush=True)
        self.graph = self.graph2 = None
        self.tail = False
        self.tail_fallback = True
        s._works = []
        self._warm(1)
        self._capture()

    def _warm(self, passes):
        """eager passes on a side stream before a capture: every kernel's lazily built state (packed images, plans, arena size) exists, and
        the pass tells which parameters receive a gradient — _prepare_tail builds the tail's device tables from that, eagerly"""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.loss = self.aux = None
            self._accumulators = capture_safe_accumulators(self.params)      # kept: the capture must find these, not default-stream ones
            for _ in range(passes):
                self._eager_fwd_bwd()
            if self.tail:
                self._prepare_tail()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

    def _tail_key(self):
        """what the captured tail has baked in: which parameters take part, where they and their momentum buffers live"""
        live, _ = self.grad_sync.live()
        return tuple(p.requires_grad for p in self.params) + self.optimizer.capture_key(live)

    def _prepare_tail(self):
        """after a pass (warm-up, or the last replay before a re-capture): can the tail be captured (every live gradient sits in its flat
        slot), and are the tables it needs built?  Everything that copies a host table to the device happens HERE, eagerly — optimiser
        pointer tables and momentum buffers, the device-resident hyperparameters, the scaler's table, the re-pack descriptor table."""
        s = self.grad_sync
        src, _ = s._stragglers([p.grad for p in self.params])
        if src or not s.direct:
            self.tail = False                    # a gather launch with tables built at capture time would be needed: the tail stays eager
            if self._own_sync:
                s.close()
                self.grad_sync, self._own_sync = None, False
            return
        s.resolve_avg()                          # does the collective library average?  asked once, eagerly: a capture cannot fall back
        s.verify_live_agreement()                # the set of parameters the captured tail will skip is frozen now: it must be the same on every rank
        self.optimizer.prepare(*s.live(), scaler=self.scaler)
        ops.repack_trainable()                   # builds the descriptor table of the multi-tensor re-pack (same images, same weights: idempotent)

    def _capture(self):
        two_phase = self._two_phase
        from . import ddp as _ddp
        _ddp.quiesce_before_capture(getattr(self.grad_sync, "group", None))      # an eager collective just before a capture takes the process down (ddp.py)
        self.graph = self.graph2 = None
        self.graph = torch.cuda.CUDAGraph()
        for p in self.params:
            p.grad = None
        # the warm-up's autograd graph and its accumulators (warm-up stream) go; new ones are made under the CAPTURE stream, the stream every
        # node of the captured backward runs on: no cross-stream edge per parameter inside the graph (autograd syncs an AccumulateGrad node's
        # stream with its producer's through events, and warns about the mismatch).  Same step time either way (2.727 vs 2.727 ms, same box).
        self.loss = self.aux = self._accumulators = None
        with torch.cuda.graph(self.graph):
            self._accumulators = capture_safe_accumulators(self.params)
            self._eager_fwd_bwd(rest=not two_phase)
            if self.tail:
                s = self.grad_sync
                s._stragglers([p.grad for p in self.params])      # bookkeeping only (which parameters got no gradient): _prepare_tail saw no stragglers
                try:
                    s.start(0)
                    if os.environ.get("VS_TEST_FAIL_TAIL_CAPTURE") == "1" and getattr(s, "exchange", True):
                        raise RuntimeError("injected failure inside the capture (tests/test_gpu_ddp.py)")
                    if two_phase:
                        ops.flush_wgrads()       # the bucket-1 layers' weight gradients (the 96^3 / 48^3 levels): graph nodes BESIDE bucket 0's all-reduce node
                        s.start(1)
                    s.wait()
                except Exception:
                    # leave the capture JOINED before the error travels on: a collective started on the process group's stream that never joins back makes
                    # capture_end fail ("unjoined work") with the capture still active — nothing could be captured in this process afterwards (_capture_guarded)
                    try:
                        s.wait()
                    except Exception:            # noqa: BLE001
                        pass
                    raise
                kw = {} if self.scaler is None else {"scaler": self.scaler}
                self.optimizer.step_with(*s.live(), device_hyper=True, **kw)
        if self.tail:
            self._tail_captured = self._tail_key()
            # the captured re-pack launch reads THIS descriptor table by address: a later registration of other trainable images (another
            # model's GraphedStep) replaces ops' table, and this graph's must outlive that (it used to be freed under the graph)
            self._repack_table = ops.repack_table()
        if two_phase and not self.tail:
            self.graph2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph2, pool=self.graph.pool()):
                ops.flush_wgrads()
        self.grads = [p.grad for p in self.params]

    def _eager_fwd_bwd(self, rest=True):
        for p in self.params:
            p.grad = None
        self.loss, self.aux = self.loss_fn()
        if self.scaler is not None:
            self.loss.backward(gradient=self.scaler.seed)
        else:
            if self._one is None or self._one.device != self.loss.device:
                self._one = torch.ones((), dtype=self.loss.dtype, device=self.loss.device)
            self.loss.backward(gradient=self._one)     # a constant seed: autograd's own ones_like would be a fill launch in every replay
        if rest:
            ops.flush_wgrads()   # the second weight-gradient phase (set_wgrad_split), when the caller did not keep it for a graph of its own

    def step(self):
        if self.tail:
            if self._tail_key() != self._tail_captured:      # requires_grad toggled, momentum buffers replaced (load_state_dict): baked into the capture
                self.recaptures += 1
                self.graph = None                            # the old capture's pool goes first
                self._warm(1)                                # one eager pass with the new set: who gets a gradient now, tables rebuilt outside the capture
                self._capture_guarded()
        if self.tail:                                        # (a re-capture may have found stragglers and left the tail eager)
            self.optimizer.sync_hyper()                      # a scheduler moved lr / momentum / weight decay: three floats in device memory, no re-capture
            self.graph.replay()
            return self.loss
        self.graph.replay()
        s = self.grad_sync
        kw = {} if self.scaler is None else {"scaler": self.scaler}
        if s is None:
            self.optimizer.step(**kw)
            return self.loss
        if self.graph2 is not None:
            # bucket-0 stragglers (gradients autograd did not write into their flat slot: Linear weights, BatchNorm affine, ...) must be in
            # the bucket BEFORE its all-reduce starts; copying them afterwards would both miss the average and race with the collective
            s.gather(self.grads, s._tab0, only=s._first_ids)
            s.start(0)
            self.graph2.replay()
            s.gather(self.grads, only=s._second_ids)
            s.start(1)
        else:
            s.gather(self.grads)
            s.start(0)
        s.wait()
        self.optimizer.step_with(*s.live(), **kw)
        return self.loss


# ----------------------------------------------------------------------------------------------------
# test-time training (main_target.py:809-953)
# ----------------------------------------------------------------------------------------------------
def lambda_schedule_device(recon_loss, lambda_vae):
    """lambda_schedule (main_target.py:838-841) evaluated on the device: the reference branches on the host value of the loss
    (`if recon_loss < 0.15`), which would force a sync per iteration and cannot be captured into a HIP graph; torch.where on
    the detached loss selects the same constant."""
    r = recon_loss.detach()
    lam = float(lambda_vae)
    c = lambda v: torch.full_like(r, v)
    return torch.where(r < 0.15, c(lam * 0.6), torch.where(r < 0.225, c(lam * 1.2), torch.where(r < 0.3, c(lam * 2.0), c(lam * 3.0))))


def finetune_loss(recon_loss, fake_loss, klloss, lambda_vae=1.0, domain_loss_type=0, kl=False, only_pseudo=False):
    """Loss of one test-time-training iteration, main_target.py:835-884 — the branches the shipped scripts reach: only_pseudo
    (:835-836), domain_loss_type 8 (:837-847) and 9 (:848-853), and the default lambda_vae*recon + fake (:881-882)."""
    if only_pseudo:
        return fake_loss
    k = klloss if kl else 0
    if domain_loss_type == 8:
        cur = lambda_schedule_device(recon_loss, lambda_vae)
        return torch.where(cur > 1, recon_loss + k + fake_loss / cur, cur * (recon_loss + k) + fake_loss)
    if domain_loss_type == 9:
        cur = lambda_schedule_device(recon_loss, lambda_vae)
        return (cur * recon_loss + fake_loss) / (1 + cur)
    if domain_loss_type == 0:
        return lambda_vae * recon_loss + fake_loss
    raise NotImplementedError("finetune loss: only_pseudo and domain_loss_type 0, 8, 9 (the variants the reference's scripts use)")


def finetune_losses(student, teacher, img, label, lambda_vae=1.0, domain_loss_type=0, kl=False, only_pseudo=False,
                    use_confident_binarize=False, n_class=2):
    """Forward + losses of one test-time-training iteration (main_target.py:814-884)."""
    batch = {"img": img, "gt": ops.onehot(label, n_class)}
    batch = student(batch, "img", "pred", "recon", dropout=True)
    with torch.no_grad():                       # the reference leaves autograd on here; its teacher is frozen, so nothing differs
        batch = teacher(batch, "img", "fake", "_unused")
    klloss = KLloss(batch)
    batch["fake"] = confident_binarize(batch["fake"]) if use_confident_binarize else binarize(batch["fake"])
    recon_loss = 1 - avg_dsc(batch, "pred", "recon", botindex=1, topindex=n_class, eps=EPS_EVALUATION)
    dsc_loss = 1 - avg_dsc(batch, "pred", "gt", botindex=1, topindex=n_class, eps=EPS_EVALUATION)
    fake_loss = 1 - avg_dsc(batch, "pred", "fake", botindex=1, topindex=n_class, eps=EPS_EVALUATION)
    final = finetune_loss(recon_loss, fake_loss, klloss, lambda_vae, domain_loss_type, kl, only_pseudo)
    return final, {"recon_loss": recon_loss, "dice_loss_fake": fake_loss, "dice_loss": dsc_loss, "final_loss": final}


class TestTimeFinetune:
    """Per-case test-time training + validation of the reference (main_target.py:809-953).

    ``model`` is the adapted network, ``model_ft`` its per-case copy, ``teacher`` the frozen pseudo-label network (all Joint).
    run(img, label): model_ft <- model (:811); ``steps`` iterations of finetune_losses + SGD(lr, weight_decay, momentum 0)
    (:886-891); then the batch-1 forward of both networks and their hard Dice against the label (:902-953).
    The iteration is B=1 and launch-latency bound, so it is captured once into a HIP graph (inputs are copied into fixed
    buffers) and replayed; nothing in the loop syncs with the host — the per-iteration loss scalars stay on the device and are
    returned as tensors.  The frozen VAE of model_ft is synchronised with model's once, at construction (the reference
    re-copies identical values on every case)."""

    __test__ = False                           # not a pytest class

    def __init__(self, model, model_ft, teacher, spatial, steps=1, lr=1e-2, weight_decay=0.0, lambda_vae=1.0, domain_loss_type=0,
                 kl=False, only_pseudo=False, use_confident_binarize=False, n_class=2, graph=True, device="cuda"):
        from . import optim
        self.model, self.model_ft, self.teacher, self.steps, self.n_class = model, model_ft, teacher, int(steps), n_class
        with torch.no_grad():
            model_ft.load_state_dict(model.state_dict())
        for p in model_ft.Vae.parameters():
            p.requires_grad = False
        model_ft.Vae.eval()
        ops.clear_pack_cache()
        if graph and (getattr(model_ft, "seg_dropout", 0.0) or getattr(model_ft, "vae_decoder_dropout", 0.0)):
            graph = False       # dropout > 0: a captured graph would replay the same masks every iteration; the reference draws fresh ones
        self.graph = bool(graph)
        self.params = [p for p in model_ft.Seg.parameters() if p.requires_grad]
        self.src = [p for p in model.Seg.parameters()][:len(self.params)]
        self.img = torch.zeros(1, 1, spatial, spatial, spatial, device=device)
        self.label = torch.zeros(1, 1, spatial, spatial, spatial, device=device)
        self.opt = optim.SGD(self.params, lr=lr, momentum=0.0, weight_decay=weight_decay)
        kw = dict(lambda_vae=lambda_vae, domain_loss_type=domain_loss_type, kl=kl, only_pseudo=only_pseudo,
                  use_confident_binarize=use_confident_binarize, n_class=n_class)
        self.loss_fn = lambda: finetune_losses(self.model_ft, self.teacher, self.img, self.label, **kw)
        self.stepper = GraphedStep(self.loss_fn, self.params, self.opt) if graph else None
        if graph:                               # the capture's warm-up iterations moved model_ft: start every case from model
            self._reset()

    @torch.no_grad()
    def _reset(self):
        for d, s in zip(self.params, self.src):
            d.copy_(s)
        ops.weights_changed()

    def run(self, img, label):
        """-> (per-iteration list of loss dicts [device scalars], score_noft, score, pred)"""
        self.img.copy_(img)
        self.label.copy_(label)
        self._reset()
        log = []
        for _ in range(self.steps):
            if self.stepper is not None:
                self.stepper.step()
                aux = self.stepper.aux
            else:
                for p in self.params:
                    p.grad = None
                loss, aux = self.loss_fn()
                loss.backward()
                self.opt.step()
            log.append({k: v.detach().clone() for k, v in aux.items()})
        with torch.no_grad():
            batch = {"img": self.img, "gt": ops.onehot(self.label, self.n_class)}
            batch = self.model(batch, "img", "pred_noft", "recon_noft")
            batch = self.model_ft(batch, "img", "pred", "recon")
            score_noft = avg_dsc(batch, "pred_noft", "gt", binary=True, botindex=1, topindex=self.n_class)
            score = avg_dsc(batch, "pred", "gt", binary=True, botindex=1, topindex=self.n_class)
        return log, score_noft, score, batch["pred"]
