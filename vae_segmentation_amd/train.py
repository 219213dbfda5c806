"""Train-step bodies of the reference's main_source.py / main_target.py on the native modules, plus a
HIP-graph replayed step.  Each *_losses function mirrors the loss arithmetic of one ``--method``:

  joint_train_losses        main_source.py:449-471     lambda_vae*(1-Dice(pred,recon)) + (1-Dice(pred,gt)), eps 1e-4
  seg_train_losses          main_source.py:421-441
  vae_train_losses          main_source.py:389-413     (1-Dice(recon,gt)) + 2e-5*KL, z = mean + noise*std*0.35
  domain_adaptation_losses  main_target.py:520-596     student/teacher, domain_loss_type 0 / 8 / 9, eps 1e-6
"""
import torch

from . import ops
from .evaluation import EPS_EVALUATION, EPS_MAIN_SOURCE, KLloss, avg_dsc, binarize, confident_binarize


def joint_train_losses(joint, img, label, lambda_vae=0.1, eps=EPS_MAIN_SOURCE, n_class=2):
    batch = {"img": img, "gt": ops.onehot(label, n_class)}
    batch = joint(batch, "img", "pred", "recon")
    recon_loss = 1 - avg_dsc(batch, "pred", "recon", botindex=1, topindex=n_class, eps=eps)
    dsc_loss = 1 - avg_dsc(batch, "pred", "gt", botindex=1, topindex=n_class, eps=eps)
    final = lambda_vae * recon_loss + dsc_loss
    return final, {"recon_loss": recon_loss, "dice_loss": dsc_loss, "batch": batch}


def seg_train_losses(seg, img, label, eps=EPS_MAIN_SOURCE, n_class=2):
    batch = {"img": img, "gt": ops.onehot(label, n_class)}
    batch = seg(batch, "img", "pred")
    dsc_loss = 1 - avg_dsc(batch, "pred", "gt", botindex=1, topindex=n_class, eps=eps)
    return dsc_loss, {"dice_loss": dsc_loss, "batch": batch}


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
                             use_confident_binarize=False, eps=EPS_EVALUATION, n_class=2):
    batch = {"img": img, "gt": ops.onehot(label, n_class)}
    batch = student(batch, "img", "pred", "recon", dropout=True)
    with torch.no_grad():
        batch = teacher(batch, "img", "fake", "_unused")          # also (re)sets batch["mean"/"std"] (joint_model.py:451)
    batch["fake"] = confident_binarize(batch["fake"]) if use_confident_binarize else binarize(batch["fake"])
    recon_loss = 1 - avg_dsc(batch, "pred", "recon", botindex=1, topindex=n_class, eps=eps)
    klloss = KLloss(batch)
    dsc_loss = 1 - avg_dsc(batch, "pred", "gt", botindex=1, topindex=n_class, eps=eps)
    fake_loss = 1 - avg_dsc(batch, "pred", "fake", botindex=1, topindex=n_class, eps=eps)
    if domain_loss_type == 8:
        cur = lambda_schedule(recon_loss.detach(), lambda_vae)
        if cur > 1:
            final = recon_loss + (klloss if kl else 0) + 1 / cur * fake_loss
        else:
            final = cur * (recon_loss + (klloss if kl else 0)) + fake_loss
    elif domain_loss_type == 9:
        cur = lambda_schedule(recon_loss.detach(), lambda_vae)
        final = (cur * recon_loss + fake_loss) / (1 + cur)
    elif domain_loss_type == 0:
        final = lambda_vae * recon_loss + fake_loss
        if kl:
            final = final + 0.00002 * lambda_vae * klloss
    else:
        raise NotImplementedError("domain_loss_type %r" % (domain_loss_type,))
    return final, {"recon_loss": recon_loss, "kl_loss": klloss, "dice_loss": dsc_loss, "dice_loss_fake": fake_loss,
                   "batch": batch}


class GraphedStep:
    """zero_grad -> forward -> losses -> backward captured once into a HIP graph and replayed per step; the
    optimiser (one multi-tensor kernel) and, under DDP, the gradient all-reduce run eagerly after each replay.

    ``loss_fn()`` must read its inputs from tensors that stay at fixed addresses (copy new data into them)."""

    def __init__(self, loss_fn, params, optimizer, grad_sync=None, warmup=2, overlap=True):
        ops.set_overlap(overlap)
        self.loss_fn, self.params, self.optimizer, self.grad_sync = loss_fn, list(params), optimizer, grad_sync
        self.graph = None
        self.loss = None
        self.aux = None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._eager_fwd_bwd()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        for p in self.params:
            p.grad = None
        with torch.cuda.graph(self.graph):
            self._eager_fwd_bwd()

    def _eager_fwd_bwd(self):
        for p in self.params:
            p.grad = None
        self.loss, self.aux = self.loss_fn()
        self.loss.backward()
        ops.join_side()          # weight-gradient kernels run on a side stream (a parallel branch of the captured graph)

    def step(self):
        self.graph.replay()
        if self.grad_sync is not None:
            self.grad_sync()
        self.optimizer.step()
        return self.loss
