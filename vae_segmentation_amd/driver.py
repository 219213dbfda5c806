"""Shared machinery of the main_source.py / main_target.py entry points.

The reference's scripts (main_source.py 853 lines, main_target.py 1063 lines) are argparse + a NumPy/SimpleITK data
pipeline + the train / validate / checkpoint loop.  The data pipeline, TensorBoard writer and plotting are out of scope
 and their dependencies are absent, so the entry points here keep the reference's flag names and
loop semantics (methods, loss bodies, frozen sub-nets, optimiser groups, epoch arithmetic, checkpoint dict layout, score
JSON) and feed the step from a deterministic synthetic dataset (`--synthetic`, the default) or, with `--real_data`, from merge.npy cases
run through the device data pipeline (data_gpu.py).

One process per GPU: under torchrun (WORLD_SIZE > 1) every rank builds the same replica, takes its own shard of the
synthetic volumes and averages gradients with one RCCL all-reduce per step (vae_segmentation_amd.ddp) — the replacement
for the reference's nn.DataParallel wrap (main_source.py:354, main_target.py:436-438).
"""
import json
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import ddp, ops, optim
from . import train as T
from .evaluation import EPS_EVALUATION, EPS_MAIN_SOURCE, avg_dsc
from .modules import Embed, Encoder, Fusion, Joint, Joint2, Segmentation, VAE, set_kernel_dtype

LABEL_KEY, IMG_KEY = "venous_pancreas", "venous"          # main_source.py:355-356


# ----------------------------------------------------------------------------------------------------
# synthetic data (stands in for BaseDataset + the transform stack of main_source.py:189-243)
# ----------------------------------------------------------------------------------------------------
def _hash_uniform(n, stream, seed):
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = idx * np.uint64(0x9E3779B97F4A7C15) + np.uint64((seed * 0x9E3779B97F4A7C15 + stream * 0xD1B54A32D192ED03 + 12345) % (1 << 64))
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return ((x >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32)


class SyntheticVolumes(torch.utils.data.Dataset):
    """Image ~ clip(N(0,1) + blob contrast, -1, 1) as after Clip / CenterIntensities (main_source.py:211-212); label = a
    jittered ellipsoid (1 = organ).  Item i is a pure function of (seed, i): every rank / run sees the same volumes."""

    def __init__(self, count, side, seed=0):
        self.count, self.side, self.seed = count, side, seed

    def __len__(self):
        return self.count

    def __getitem__(self, i):
        s = self.side
        u = _hash_uniform(8, 17, self.seed * 100003 + i)
        ax = (np.arange(s, dtype=np.float32) + 0.5) / s - 0.5
        z, y, x = np.meshgrid(ax, ax, ax, indexing="ij")
        c = (u[0:3] - 0.5) * 0.2
        r = 0.18 + 0.14 * u[3:6]
        d2 = ((z - c[0]) / r[0]) ** 2 + ((y - c[1]) / r[1]) ** 2 + ((x - c[2]) / r[2]) ** 2
        label = (d2 < 1.0).astype(np.float32)
        n = s ** 3
        u1 = np.maximum(_hash_uniform(n, 1, self.seed * 7919 + i), 1e-7).astype(np.float64)
        u2 = _hash_uniform(n, 2, self.seed * 7919 + i).astype(np.float64)
        noise = (np.sqrt(-2 * np.log(u1)) * np.cos(2 * np.pi * u2)).astype(np.float32).reshape(s, s, s)
        img = np.clip(0.6 * noise + 0.8 * label - 0.2, -1, 1)
        return {IMG_KEY: torch.from_numpy(img[None]), LABEL_KEY: torch.from_numpy(label[None])}


# ----------------------------------------------------------------------------------------------------
# real data: merge.npy cases from a json list, transformed on the device (main_source.py:125-131,186-243 with data_gpu.py's pipeline)
# ----------------------------------------------------------------------------------------------------
def filedict_from_json(json_path, key, epoch=1):
    """main_source.py:123-131: the case names under `key`, repeated `epoch` times"""
    import json
    with open(json_path, "r") as f:
        names = json.load(f).get(key, [])
    return list(names) * epoch


def mask_index_of(args):
    """main_source.py:92-95"""
    if str(args.pan_index) != "10":
        return [[0, 0]] + [[int(f), i + 1] for i, f in enumerate(str(args.pan_index).split(","))]
    return [[0, 0], [[1, 2], 1]]


class DeviceCaseLoader:
    """What BaseDataset + the transform stack + DataLoader deliver in the reference (main_source.py:186-243), with the transforms on the
    device: np.load of each case's merge array is the only host work; relabel / CropResize / MySpatialTransform / Clip / CenterIntensities
    run as vs_data_* kernels (data_gpu.py).  Yields {IMG_KEY, LABEL_KEY: (B, 1, S, S, S) CUDA tensors}."""

    def __init__(self, names, root, args, batch_size, train, shuffle, rank=0, world=1, seed=0):
        from . import data_gpu
        names = list(names)
        if train and world > 1:
            names = names[:len(names) // world * world]          # every rank the same number of steps (the all-reduce is collective)
        self.names, self.root, self.bs, self.shuffle = names[rank::world], root, batch_size, shuffle
        self.patch, self.mask_index, self.epoch, self.seed = (args.size,) * 3, mask_index_of(args), 0, seed + rank
        self.drop_last = train
        self.shift = int(getattr(args, "shift", 0)) if train else 0           # main_target.py:204 (training crops only; validation: CropResize default)
        self.transform = None
        if train and not getattr(args, "no_aug", False):                       # main_source.py:195-205
            self.transform = data_gpu.MySpatialTransform(
                self.patch, [d // 2 - 5 for d in self.patch], random_crop=True, scale=(0.85, 1.15), do_elastic_deform=False, alpha=(0, 500),
                do_rotation=True, sigma=(10, 30.), angle_x=(-0.2, 0.2), angle_y=(-0.2, 0.2), angle_z=(-0.2, 0.2), border_mode_data="constant",
                border_cval_data=-1024, data_key=IMG_KEY, p_el_per_sample=0, label_key=LABEL_KEY, p_scale_per_sample=1, p_rot_per_sample=1,
                rng=np.random.RandomState(seed + 1000 * rank))
        self._dg = data_gpu

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return len(self.names) // self.bs if self.drop_last else (len(self.names) + self.bs - 1) // self.bs

    def __iter__(self):
        order = np.arange(len(self.names))
        if self.shuffle:
            np.random.RandomState(self.seed + self.epoch).shuffle(order)
        for b in range(len(self)):
            imgs, labs = [], []
            for i in order[b * self.bs:(b + 1) * self.bs]:
                merge = torch.from_numpy(np.load(os.path.join(self.root, self.names[i])).astype(np.float32)).cuda(non_blocking=True)
                img, lab = self._dg.train_sample(merge, self.patch, self.mask_index, self.transform, field=IMG_KEY, shift=self.shift)
                imgs.append(img); labs.append(lab)
            yield {IMG_KEY: torch.cat(imgs), LABEL_KEY: torch.cat(labs)}


def make_loaders(args, rank, world):
    if getattr(args, "real_data", False):
        json_path = os.path.join("lists", args.data_path)
        train = DeviceCaseLoader(filedict_from_json(json_path, args.train_list, args.eval_epoch), args.data_root, args, args.batch_size, True,
                                 shuffle=args.method != "domain_adaptation", rank=rank, world=world)       # main_source.py:232-236
        val = DeviceCaseLoader(filedict_from_json(json_path, args.val_list), args.val_data_root, args, 1, False, False)
        return train, val, train                      # the loader doubles as the 'sampler': set_epoch reshuffles
    train = SyntheticVolumes(args.synthetic_train, args.size, seed=1)
    val = SyntheticVolumes(args.synthetic_val, args.size, seed=2)
    sampler = None
    if world > 1:
        sampler = torch.utils.data.distributed.DistributedSampler(train, num_replicas=world, rank=rank, shuffle=True, drop_last=True)
    tl = torch.utils.data.DataLoader(train, batch_size=args.batch_size, shuffle=sampler is None, sampler=sampler,
                                     num_workers=0, pin_memory=True, drop_last=True)        # drop_last: main_source.py:237
    vl = torch.utils.data.DataLoader(val, batch_size=1, shuffle=False, num_workers=0, pin_memory=True)
    return tl, vl, sampler


def make_pseudo_loader(args, rank, world):
    """main_target.py:228-258,299-307: the second, pseudo-labelled training loader of a --pseudo_list run — its own case list, data root and structure indices
    (--pseudo_data_root, --pseudo_pan_index), the training transform stack, shuffled, drop_last."""
    if getattr(args, "real_data", False):
        import copy
        pargs = copy.copy(args)
        pargs.pan_index = args.pseudo_pan_index                # main_target.py:233: NumpyLoader_Multi_merge(mask_index=pseudo_mask_index)
        return DeviceCaseLoader(filedict_from_json(os.path.join("lists", args.data_path), args.pseudo_list, args.eval_epoch), args.pseudo_data_root, pargs,
                                args.batch_size, True, shuffle=True, rank=rank, world=world, seed=7)
    ds = SyntheticVolumes(args.synthetic_train, args.size, seed=3)
    return torch.utils.data.DataLoader(ds, batch_size=args.batch_size, shuffle=True, num_workers=0, pin_memory=True, drop_last=True)


# ----------------------------------------------------------------------------------------------------
# model / optimiser construction (main_source.py:245-294, 300-346; main_target.py:314-352, 395-433)
# ----------------------------------------------------------------------------------------------------
def n_class_of(args):
    return 1 + len(str(args.pan_index).split(",")) if str(args.pan_index) != "10" else 2


def build_joint(args):
    nc = n_class_of(args)
    seg = Segmentation(n_channels=1, n_class=nc, norm_type=1)
    vae = VAE(n_channels=nc, n_class=nc, norm_type=1, dim=128, spatial=args.size)
    return Joint(models=[seg, vae], vae_forward_scale=getattr(args, "vae_forward_scale", 0.0),                 # main_target.py:324
                 vae_decoder_dropout=getattr(args, "vae_decoder_dropout", 0.0), seg_dropout=getattr(args, "seg_dropout", 0.0))


def freeze(module):
    for p in module.parameters():
        p.requires_grad = False
    module.eval()
    return module


def load_prefix(module, prefix, name="best_model.ckpt"):
    """--load_prefix* : '3dmodel/<prefix>/<name>' holding {'epoch','model_state_dict','optimizer_state_dict'}."""
    path = os.path.join("3dmodel", prefix, name)
    sd = torch.load(path, map_location="cpu")["model_state_dict"]
    module.load_state_dict(sd)
    ops.clear_pack_cache()
    print("loaded %s" % path)


def make_optimizer(args, groups):
    if getattr(args, "adam", False):
        return optim.Adam(groups, lr=args.lr_seg, betas=(0.9, 0.999), weight_decay=0.0)
    return optim.SGD(groups, lr=args.lr_seg, momentum=0.9, weight_decay=0.0)


def save_checkpoint(prefix, epoch_label, model, optimizer, best):
    path = os.path.join("3dmodel", prefix)
    os.makedirs(path, exist_ok=True)
    blob = {"epoch": epoch_label, "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict()}
    torch.save(blob, os.path.join(path, "model_epoch%d.ckpt" % epoch_label))
    if best:
        torch.save(blob, os.path.join(path, "best_model.ckpt"))


# ----------------------------------------------------------------------------------------------------
# validation (main_source.py:688-822): batch 1, hard Dice of the argmax prediction against the label
# ----------------------------------------------------------------------------------------------------
@torch.no_grad()
def validate(method, model, loader, nc):
    """main_source.py:688-822 / main_target.py:754-805: batch-1 forwards and hard Dice per case.  Forward only (no autograd graph is recorded: nothing is kept
    for a backward pass — the per-(n, c) statistics the conv epilogues accumulate are the FORWARD's own, the next layer normalises with them)."""
    scores = {}
    for i, batch in enumerate(loader):
        gt = ops.onehot(batch[LABEL_KEY].cuda(non_blocking=True), nc)
        if method == "vae_train":
            pred, _, _ = model(gt, if_random=False)
        elif method == "discriminator_train":                              # no segmentation to score: report 1 - squared error of the score
            tgt = batch[LABEL_KEY].float().mean((1, 2, 3, 4)).view(-1, 1).cuda() * 4
            scores[i] = float(1 - ((model(batch[LABEL_KEY].cuda().float()) - tgt) ** 2).mean().item())
            continue
        elif method in ("embed_train", "refine_vae"):                      # main_source.py:735-745: Embed in test mode
            pred = model({IMG_KEY: batch[IMG_KEY].cuda(non_blocking=True), "venous_pancreas_only": gt}, IMG_KEY, "pred", test_mode=True)["pred"]
        else:
            seg = model.Seg if hasattr(model, "Seg") else model
            pred = seg({IMG_KEY: batch[IMG_KEY].cuda(non_blocking=True)}, IMG_KEY, "pred")["pred"]
        scores[i] = avg_dsc({"p": pred, "g": gt}, "p", "g", binary=True, botindex=1, topindex=nc).item()
    return scores


def validate_finetune(runner, loader):
    """main_target.py:807-953 with --val_finetune k: per case, test-time training of a copy of the model, then hard Dice of the
    finetuned and of the untouched network.  -> (scores, scores_noft)"""
    scores, scores_noft = {}, {}
    for i, batch in enumerate(loader):
        _, noft, ft, _ = runner.run(batch[IMG_KEY].cuda(non_blocking=True), batch[LABEL_KEY].cuda(non_blocking=True))
        scores[i], scores_noft[i] = ft.item(), noft.item()
    return scores, scores_noft


# ----------------------------------------------------------------------------------------------------
# the loop
# ----------------------------------------------------------------------------------------------------
def run(args, side="source"):
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise SystemExit("the native kernels need a GPU (there is no CPU path)")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if args.save_epoch % args.eval_epoch != 0:                            # main_source.py asserts this; an assert would vanish under python -O
        raise SystemExit("--save_epoch must be a multiple of --eval_epoch")
    nc = n_class_of(args)
    method = args.method
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    eps = EPS_MAIN_SOURCE if side == "source" else EPS_EVALUATION          # the two scripts use different epsilons

    teacher = None
    if method == "vae_train":
        model = VAE(n_channels=nc, n_class=nc, norm_type=1, dim=128, soft=bool(args.softrelu), spatial=args.size)
        trainable = model
    elif method == "seg_train":
        model = Segmentation(n_channels=1, n_class=nc, norm_type=1)
        trainable = model
    elif method in ("joint_train", "domain_adaptation", "sep_joint_train"):
        model = build_joint(args)
        trainable = model.Seg
        if method in ("domain_adaptation", "sep_joint_train"):
            teacher = build_joint(args)                                     # main_target.py:323-328 / main_source.py:270-273 (tea_model)
    elif method in ("embed_train", "refine_vae"):                          # main_source.py:259-264
        model = Embed(models=[Encoder(n_channels=1, dim=128, norm_type=1, spatial=args.size),
                              VAE(n_channels=nc, n_class=nc, norm_type=1, dim=128, spatial=args.size),
                              Fusion(n_channels_img=1, n_channels_mask=nc, n_class=nc, norm_type=1)])
        trainable = model
    elif method == "discriminator_train":                                  # main_target.py:318-319
        model = Encoder(n_channels=1, dim=1, norm_type=1, spatial=args.size)
        trainable = model
    elif method == "domain_adaptation_dis":                                # main_target.py:337-342
        model = Joint2(models=[Segmentation(n_channels=1, n_class=nc, norm_type=1), Encoder(n_channels=1, dim=1, norm_type=1, spatial=args.size)],
                       seg_dropout=getattr(args, "seg_dropout", 0.0))
        trainable = model.Seg
        teacher = Segmentation(n_channels=1, n_class=nc, norm_type=1)
    else:
        raise ValueError("Try a valid method.")                           # main_source.py:275 / main_target.py:343
    model = model.cuda()
    from_scratch = bool(getattr(args, "from_scratch", False)) and method == "domain_adaptation"
    if from_scratch:
        teacher = teacher.cuda()
    if args.load_prefix:                                                    # main_target.py:355-366: --from_scratch loads the TEACHER's Seg, the student keeps its initialisation
        target = teacher if from_scratch else model
        load_prefix(target.Seg if hasattr(target, "Seg") else target, args.load_prefix, args.checkpoint_name)
    if args.load_prefix_vae and hasattr(model, "Vae"):
        if from_scratch:
            load_prefix(teacher.Vae, args.load_prefix_vae)                  # :370-371
        load_prefix(model.Vae, args.load_prefix_vae)
    if getattr(args, "load_prefix_encoder", None):                          # :384-390
        if method == "discriminator_train":
            load_prefix(model, args.load_prefix_encoder)
        elif hasattr(model, "Dis"):
            load_prefix(model.Dis, args.load_prefix_encoder)
    if args.load_prefix_joint and hasattr(model, "Seg"):
        load_prefix(model, args.load_prefix_joint, args.checkpoint_name)
    if hasattr(model, "Vae") and method != "refine_vae":
        freeze(model.Vae)                                                   # main_source.py:343-346
    if getattr(args, "fix_layer", False) and method in ("joint_train", "domain_adaptation"):     # main_target.py:400-406: only the last decoder block and the head train
        for prm in model.Seg.parameters():
            prm.requires_grad = False
        for prm in list(model.Seg.up5.parameters()) + list(model.Seg.out_block.parameters()):
            prm.requires_grad = True
    if method == "refine_vae":                                              # main_source.py:347-353: VAE encoder half frozen, decoder trained
        for name, prm in model.Vae.named_parameters():
            prm.requires_grad = name.split(".")[0] not in ("in_block", "down1", "down2", "down3", "down4", "down5", "fc_mean", "fc_std")
        for prm in model.Encoder.parameters():                              # :596-597
            prm.requires_grad = False
    if method == "domain_adaptation_dis":
        freeze(model.Dis)                                                   # main_target.py:407-411
    if teacher is not None and method in ("sep_joint_train", "domain_adaptation_dis"):
        teacher = teacher.cuda()
        teacher.load_state_dict(model.state_dict() if method == "sep_joint_train" else model.Seg.state_dict())   # main_source.py:330-340 / main_target.py:366
        freeze(teacher)
        set_kernel_dtype(teacher, dtype)
    elif teacher is not None:
        teacher = teacher.cuda()
        if getattr(args, "only_pseudo", False):
            # main_target.py:421-425: the loaded network becomes the frozen pseudo-label teacher, a freshly initialised one is trained
            model, teacher = teacher, model
            freeze(model.Vae)
            trainable = model.Seg
        elif not args.test_only and from_scratch:
            pass                                                            # main_target.py:427: the teacher keeps what --load_prefix* gave it
        elif not args.test_only:
            teacher.load_state_dict(model.state_dict())                     # main_target.py:426-427
        else:
            teacher.load_state_dict(model.state_dict())                     # main_target.py:379-380
        freeze(teacher)
        set_kernel_dtype(teacher, dtype)
    set_kernel_dtype(model, dtype)
    from . import modules as _modules
    _modules.set_recompute(bool(getattr(args, "recompute", False)))

    if hasattr(model, "Seg") and hasattr(model, "Vae"):
        groups = [{"params": list(model.Seg.parameters()), "lr": args.lr_seg, "model": "Seg"},
                  {"params": list(model.Vae.parameters()), "lr": args.lr_vae, "model": "Vae"}]
    else:
        groups = [{"params": list(model.parameters()), "lr": args.lr_seg}]
    optimizer = make_optimizer(args, groups)
    params = [p for p in trainable.parameters() if p.requires_grad]
    sync = ddp.FlatGradSync(params) if world > 1 else None
    if sync is not None:
        sync.broadcast_parameters(0)
        # the trainable list is not the whole replica: frozen or freshly initialised layers (--fix_layer, --from_scratch, a VAE without a checkpoint) and
        # the teacher must agree across ranks too, or the replicas differ from the first forward on
        seen = {id(p) for p in params}
        for mod in (model, teacher):
            if mod is None:
                continue
            for t in list(mod.parameters()) + list(mod.buffers()):
                if id(t) not in seen and t.is_cuda:
                    seen.add(id(t))
                    dist.broadcast(t.data, 0)
        ddp.note_eager_collective()
        ops.weights_changed()
    # fp16 storage: Dice gradients are O(1e-6), below fp16's normal range — dynamic loss scaling on the device (optim.LossScaler)
    scaler = optim.LossScaler() if dtype == torch.float16 else None
    skw = {} if scaler is None else {"scaler": scaler}

    runner = None
    if method == "domain_adaptation" and getattr(args, "val_finetune", 0) and rank == 0:
        model_ft = build_joint(args).cuda()
        set_kernel_dtype(model_ft, dtype)
        runner = T.TestTimeFinetune(model, model_ft, teacher, args.size, steps=args.val_finetune, lr=args.lr_finetune,
                                    lambda_vae=args.lambda_vae, domain_loss_type=getattr(args, "domain_loss_type", 0),
                                    kl=getattr(args, "kl", False), only_pseudo=getattr(args, "only_pseudo", False),
                                    use_confident_binarize=getattr(args, "use_confident_binarize", False), n_class=nc)

    train_loader, val_loader, sampler = make_loaders(args, rank, world)
    pseudo_loader = make_pseudo_loader(args, rank, world) if (method == "domain_adaptation" and getattr(args, "pseudo_list", None) is not None) else None
    pseudo_itr, pseudo_cycles = None, 0
    pimg_buf = plab_buf = None
    lambda_vae = args.lambda_vae
    turn_epoch, warmup_epochs = getattr(args, "turn_epoch", -1), getattr(args, "lambda_vae_warmup", 0)
    has_dropout = bool(getattr(args, "seg_dropout", 0.0) or getattr(args, "vae_decoder_dropout", 0.0))
    use_graph = not getattr(args, "no_graph", False) and not has_dropout      # dropout draws fresh masks per call: eager (train.GraphedStep)
    bs, side_ = args.batch_size, args.size
    img_buf = torch.zeros(bs, 1, side_, side_, side_, device="cuda")         # fixed-address inputs of the captured step
    lab_buf = torch.zeros(bs, 1, side_, side_, side_, device="cuda")
    score_buf = torch.zeros(bs, 1, device="cuda")                           # discriminator_train's regression target (venous_score)
    cur = {"epoch": 0}
    # the Embed methods draw VAE noise per call on the device and embed_train toggles requires_grad by epoch: launched eagerly
    if method in ("embed_train", "refine_vae", "vae_train"):
        use_graph = False

    def loss_fn():
        if method == "vae_train":
            return T.vae_train_losses(model, lab_buf, scale=0.35, eps=eps, n_class=nc)
        if method == "seg_train":
            return T.seg_train_losses(model, img_buf, lab_buf, eps=eps, n_class=nc)
        if method == "joint_train":
            return T.joint_train_losses(model, img_buf, lab_buf, lambda_vae=lambda_vae, eps=eps, n_class=nc)
        if method == "sep_joint_train":
            return T.sep_joint_train_losses(model, teacher, img_buf, lab_buf, eps=eps, n_class=nc)
        if method == "embed_train":
            return T.embed_train_losses(model, img_buf, lab_buf, eps=eps, n_class=nc)
        if method == "refine_vae":
            return T.refine_vae_losses(model, img_buf, lab_buf, eps=eps, n_class=nc)
        if method == "discriminator_train":
            return T.discriminator_train_loss(model, lab_buf, score_buf)
        if method == "domain_adaptation_dis":
            return T.domain_adaptation_dis_losses(model, teacher, img_buf, lab_buf, lambda_vae=lambda_vae, epoch=cur["epoch"],
                                                  lambda_vae_warmup=warmup_epochs,
                                                  use_confident_binarize=getattr(args, "use_confident_binarize", False), eps=eps, n_class=nc)
        if pseudo_loader is not None:                                       # main_target.py:615-661: a --pseudo_list run has its own loss ladder
            return T.domain_adaptation_pseudo_losses(model, teacher, img_buf, lab_buf, lambda_vae=lambda_vae,
                                                     domain_loss_type=getattr(args, "domain_loss_type", 0),
                                                     use_confident_binarize=getattr(args, "use_confident_binarize", False), eps=eps, n_class=nc,
                                                     host_schedule=not use_graph)

        def da():
            return T.domain_adaptation_losses(model, teacher, img_buf, lab_buf, lambda_vae=lambda_vae,
                                              domain_loss_type=getattr(args, "domain_loss_type", 0), kl=getattr(args, "kl", False),
                                              use_confident_binarize=getattr(args, "use_confident_binarize", False), eps=eps, n_class=nc,
                                              only_pseudo=getattr(args, "only_pseudo", False), epoch=cur["epoch"], turn_epoch=turn_epoch,
                                              lambda_vae_warmup=warmup_epochs, host_schedule=not use_graph)
        n_mc = max(1, int(getattr(args, "vae_mont_number", 1)))
        if n_mc == 1:
            return da()
        # main_target.py:530-603 (--vae_mont_number N): N forward passes of student and teacher per step (they differ by their dropout draws),
        # the four loss terms averaged, one backward through all of them
        final, aux = da()
        tot = {k: v for k, v in aux.items() if k != "batch" and k != "kl_loss"}
        for _ in range(n_mc - 1):
            f, a = da()
            final = final + f
            tot = {k: tot[k] + a[k] for k in tot}
            aux = a
        out = {k: v / n_mc for k, v in tot.items()}
        out["kl_loss"], out["batch"] = aux["kl_loss"], aux["batch"]         # the reference logs the LAST pass's KL (:605)
        return final / n_mc, out

    def loss_key(epoch):
        """what of the loss expression depends on the epoch (main_target.py:583-592): a captured step is rebuilt when it changes"""
        if method == "domain_adaptation_dis":                               # main_target.py:720-723: lambda ramps with the epoch during the warm-up
            return min(epoch, warmup_epochs)
        if method != "domain_adaptation" or getattr(args, "domain_loss_type", 0) != 0 or getattr(args, "only_pseudo", False):
            return 0
        if turn_epoch != -1:
            return (epoch // turn_epoch) % 2
        return min(epoch, warmup_epochs)

    stepper, stepper_key = None, None
    best, n_outer = 0.0, max(1, args.max_epoch // args.eval_epoch)
    for epoch in range(n_outer):
        cur["epoch"] = epoch
        if sampler is not None:
            sampler.set_epoch(epoch)
        skip_da = method in ("domain_adaptation", "domain_adaptation_dis") and epoch == 0 and not getattr(args, "train_first_epoch", False)    # main_target.py:506,697: `if epoch == 0: continue`
        if method == "embed_train":                                         # main_source.py:550-554: the image encoder trains on odd epochs only
            for prm in model.Encoder.parameters():
                prm.requires_grad = epoch % 2 == 1
            params = [p for p in trainable.parameters() if p.requires_grad]
        if not args.test_only and not skip_da:
            model.train()
            if hasattr(model, "Vae"):
                model.Vae.eval()
            if use_graph and (stepper is None or stepper_key != loss_key(epoch)):
                stepper = None                                        # release the old capture's memory pool first
                stepper = T.GraphedStep(loss_fn, params, optimizer, grad_sync=sync, warmup=1, scaler=scaler)
                stepper_key = loss_key(epoch)
                if rank == 0:
                    print("train step captured as a HIP graph (%s)" % ("2 graphs + bucketed all-reduce" if stepper.graph2 is not None else "1 graph"))
            t0, seen = time.time(), 0
            n_iter = len(train_loader)
            for idx, batch in enumerate(train_loader):
                img_buf.copy_(batch[IMG_KEY], non_blocking=True)
                lab_buf.copy_(batch[LABEL_KEY], non_blocking=True)
                if method == "discriminator_train":          # synthetic stand-in for the venous_score field: a function of the mask
                    score_buf.copy_(batch[LABEL_KEY].float().mean((1, 2, 3, 4)).view(-1, 1) * 4, non_blocking=True)
                teacher_next = None
                if pseudo_loader is not None and getattr(args, "pseudo_save_epoch", 0) and epoch % args.pseudo_save_epoch == 0:
                    # main_target.py:633-635 (--pseudo_list runs): the pseudo-label network is re-loaded from the student AFTER this iteration's two forwards and
                    # BEFORE its optimizer step, i.e. the next iteration's pseudo-labels come from the weights this one started with: keep them, load after the step
                    with torch.no_grad():
                        teacher_next = {k: v.clone() for k, v in model.state_dict().items()}
                    if getattr(args, "tag", False):                                  # :635 — ahead of this iteration's loss
                        lambda_vae /= 10
                        stepper_key = None
                elif pseudo_loader is None and method == "domain_adaptation" and getattr(args, "pseudo_save_epoch", 0):
                    # EMA teacher (main_target.py:508-518): every `pseudo_save_epoch` epochs, at the first iteration of an epoch slice or
                    # every iteration; never in epoch 0
                    every = max(1, args.pseudo_save_epoch // args.eval_epoch)
                    if epoch != 0 and epoch % every == 0 and (getattr(args, "update_every_iteration", False) or idx % max(1, n_iter // args.eval_epoch) == 0):
                        optim.ema_update(teacher.Seg, model.Seg, args.alpha)
                if method == "domain_adaptation_dis" and getattr(args, "pseudo_save_epoch", 0) and epoch % args.pseudo_save_epoch == 0:
                    # main_target.py:710-712: the pseudo-label network is re-loaded from the student (in place: a captured graph replays it)
                    with torch.no_grad():
                        teacher.load_state_dict(model.Seg.state_dict())
                    ops.clear_pack_cache()
                    ops.refresh_frozen_packs(teacher)
                    if getattr(args, "tag", False):
                        lambda_vae /= 10
                        stepper_key = None                                   # lambda is baked into the captured loss expression
                if use_graph and stepper is not None and stepper_key is None:
                    stepper = None
                    stepper = T.GraphedStep(loss_fn, params, optimizer, grad_sync=sync, warmup=1, scaler=scaler)
                    stepper_key = loss_key(epoch)
                if stepper is not None:
                    stepper.step()
                    aux = stepper.aux
                else:
                    # optimizer.zero_grad() over ALL parameters, as the reference does every iteration (main_source.py:660): a parameter that is
                    # frozen for this epoch (embed_train's Encoder on even epochs) must not keep the gradient of its last trained step
                    for grp in optimizer.param_groups:
                        for p in grp["params"]:
                            p.grad = None
                    loss, aux = loss_fn()
                    if scaler is not None:
                        loss.backward(gradient=scaler.seed)
                    else:
                        loss.backward()
                    if sync is not None:
                        sync()
                        optimizer.step_with(*sync.live(), **skw)
                    else:
                        optimizer.step(**skw)
                if teacher_next is not None:
                    with torch.no_grad():
                        teacher.load_state_dict(teacher_next)                        # in place: a captured graph replays the same storage
                    ops.clear_pack_cache()
                    ops.refresh_frozen_packs(teacher)
                if pseudo_loader is not None:
                    # main_target.py:663-687: the next batch of the pseudo-labelled loader (cycled), student forward, two Dice terms — logged, never stepped on
                    pb = next(pseudo_itr, None) if pseudo_itr is not None else None
                    if pb is None:
                        pseudo_cycles += 1
                        if hasattr(pseudo_loader, "set_epoch"):
                            pseudo_loader.set_epoch(pseudo_cycles)                  # a DataLoader(shuffle=True) draws a new order per pass (main_target.py:299-307)
                        pseudo_itr = iter(pseudo_loader)
                        pb = next(pseudo_itr)
                    if pimg_buf is None:
                        pimg_buf, plab_buf = torch.zeros_like(img_buf), torch.zeros_like(lab_buf)
                    pimg_buf.copy_(pb[IMG_KEY], non_blocking=True)
                    plab_buf.copy_(pb[LABEL_KEY], non_blocking=True)
                    aux = dict(aux)
                    aux.update(T.pseudo_batch_losses(model, pimg_buf, plab_buf, eps=eps, n_class=nc))
                seen += bs
                if rank == 0 and idx % args.display_freq == 0:              # logging syncs the host every display_freq steps only
                    parts = ", ".join("%s %.4f" % (k, v.item()) for k, v in aux.items() if k != "batch")
                    print("[%3d, %3d] loss: %s" % ((epoch + 1) * args.eval_epoch, idx + 1, parts))
                if args.max_iters and idx + 1 >= args.max_iters:
                    break
            torch.cuda.synchronize()
            if rank == 0:
                print("epoch %d: %.2f volumes/s per rank (%s)" % (epoch + 1, seen / max(time.time() - t0, 1e-9),
                                                                   "graph replay" if stepper is not None else "eager"))
        if rank == 0:
            model.eval()
            if runner is not None and (epoch != 0 or args.test_only):             # main_target.py:811
                scores, scores_noft = validate_finetune(runner, val_loader)
                print("epoch %d validation result without finetuning: %f" % (epoch + 1, float(np.mean(list(scores_noft.values())))))
            else:
                scores = validate(method, model, val_loader, nc)
            mean = float(np.mean(list(scores.values()))) if scores else 0.0
            os.makedirs(os.path.join("tensorboard", args.prefix), exist_ok=True)
            with open(os.path.join("tensorboard", args.prefix, "score_%d.json" % epoch), "w") as f:
                json.dump(scores, f)
            print("epoch %d validation result: %f, best result %f." % (epoch + 1, mean, best))
            if not args.test_only and (epoch + 1) % max(1, args.save_epoch // args.eval_epoch) == 0:
                save_checkpoint(args.prefix, (epoch + 1) * args.eval_epoch, model, optimizer, mean > best)
            best = max(best, mean)
        if world > 1:
            dist.barrier()
            ddp.note_eager_collective()         # the next epoch may rebuild / re-capture its GraphedStep: ddp.quiesce_before_capture waits on this
        if args.test_only:
            break
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        print("Finished Training")
    return best


def check_target_flags(a):
    """main_target.py:145-170: the reference's own consistency asserts, plus what this entry point does with the flags it does not act on."""
    import sys
    def need(cond, what):                                                    # the reference asserts; `python -O` strips asserts, an entry point must not
        if not cond:
            raise SystemExit("main_target.py: inconsistent flags — " + what)
    if a.vae_mont_number != 1:
        need(a.vae_forward_scale != 0.0, "--vae_mont_number != 1 needs a non-zero --vae_forward_scale")                                 # main_target.py:145
    if a.from_scratch:
        need(a.method == "domain_adaptation" and not a.test_only, "--from_scratch belongs to --method domain_adaptation, not to --test_only")   # :157-159
    if a.kl:
        need(a.method == "domain_adaptation" and a.domain_loss_type in (0, 8), "--kl belongs to domain_adaptation with --domain_loss_type 0 or 8")   # :162-164
    if a.update_every_iteration:
        need(a.pseudo_save_epoch == 1, "--update_every_iteration needs --pseudo_save_epoch 1")                                           # :168
    if a.generate_bounding_boxes:
        need(a.method == "domain_adaptation", "--generate_bounding_boxes belongs to --method domain_adaptation")                        # :170
    if a.pseudo_list is not None:
        need(a.method == "domain_adaptation", "--pseudo_list is read by --method domain_adaptation only (main_target.py:615; domain_adaptation_dis runs "
                                              "without it, :689)")
    ignored = [n for n in ("save_more_reference", "save_eval_result", "analysis_figure_name", "generate_bounding_boxes", "resume") if getattr(a, n)]
    if ignored:
        print("main_target.py: accepted and ignored (they drive figure / volume dumps the native entry point does not write): %s"
              % ", ".join("--" + n for n in ignored), file=sys.stderr)


def add_native_flags(parser):
    g = parser.add_argument_group("native (MI355X) additions")
    g.add_argument("--synthetic", action="store_true", default=True, help="synthetic volumes (default: no dataset ships with this repository)")
    g.add_argument("--real_data", action="store_true", help="train / validate on the merge.npy cases of lists/<data_path>[train_list | val_list] under "
                   "data_root / val_data_root, transformed on the device (data_gpu.py) — the reference's loader stack (main_source.py:186-243)")
    g.add_argument("--size", type=int, default=128, help="cubic patch side (reference: patch_size 128, main_source.py:117)")
    g.add_argument("--dtype", default="fp32", choices=["fp32", "bf16", "fp16"], help="kernel storage dtype")
    g.add_argument("--synthetic_train", type=int, default=16)
    g.add_argument("--synthetic_val", type=int, default=2)
    g.add_argument("--max_iters", type=int, default=0, help="stop each epoch after this many steps (0 = whole loader)")
    g.add_argument("--display_freq", type=int, default=10)
    g.add_argument("--no_graph", action="store_true", help="eager launches instead of the HIP-graph replayed step")
    g.add_argument("--recompute", action="store_true", help="activation recomputation (checkpointing) of the Down / Up blocks: interior activations are "
                   "re-computed in backward instead of kept (BASELINE configs[4]); trades bandwidth for memory")
    g.add_argument("--train_first_epoch", action="store_true", help="domain_adaptation: also train in epoch 0 (the reference only "
                   "validates there, main_target.py:506)")
