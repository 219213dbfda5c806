#!/usr/bin/env python3
"""Source-domain trainer — same flags as the reference's main_source.py (argparse block main_source.py:25-57), native step.

Methods (all native): vae_train, seg_train, joint_train (the ones the reference's scripts/source/*.bash use), sep_joint_train, embed_train,
refine_vae (main_source.py:546-659).
Data: --synthetic volumes (see vae_segmentation_amd/driver.py).  Multi-GPU: `python -m torch.distributed.run
--nproc-per-node N main_source.py ...` (one process per GPU, RCCL gradient all-reduce) instead of -G/nn.DataParallel."""
import argparse

from vae_segmentation_amd import driver


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("prefix", help="prefix")
    p.add_argument("-P", "--target_phase", default="arterial")
    p.add_argument("-G", "--GPU", default="0,1,2,3", help="kept for CLI compatibility; ranks come from torchrun")
    p.add_argument("-b", "--batch_size", type=int, default=4, help="per-process batch here (the reference splits it over -G GPUs)")
    p.add_argument("-E", "--max_epoch", type=int, default=1600)
    p.add_argument("--save_epoch", type=int, default=50)
    p.add_argument("--eval_epoch", type=int, default=50)
    p.add_argument("--turn_epoch", type=int, default=-1)
    p.add_argument("-S", "--softrelu", type=int, default=0)
    p.add_argument("-M", "--method", default="vae_train")
    p.add_argument("-R", "--data_root", default="../nih_data/numpy_data/")
    p.add_argument("-V", "--val_data_root", default="../nih_data/numpy_data/")
    p.add_argument("-l", "--data_path", default="Multi_all.json")
    p.add_argument("-t", "--train_list", default="NIH_train")
    p.add_argument("-v", "--val_list", default="NIH_val")
    p.add_argument("--load_prefix", default=None)
    p.add_argument("--checkpoint_name", default="best_model.ckpt")
    p.add_argument("--load_prefix_vae", default=None)
    p.add_argument("--load_prefix_joint", default=None)
    p.add_argument("--pan_index", default="1")
    p.add_argument("--lambda_vae", type=float, default=0.1)
    p.add_argument("--lambda_vae_warmup", type=int, default=0)
    p.add_argument("--lr_seg", type=float, default=1e-2)
    p.add_argument("--lr_vae", type=float, default=0)
    p.add_argument("--test_only", action="store_true")
    p.add_argument("--resume", action="store_true", help="parsed and unused, as in the reference")
    p.add_argument("--save_more_reference", action="store_true")
    p.add_argument("--save_eval_result", action="store_true")
    p.add_argument("--no_aug", action="store_true")
    p.add_argument("--adam", action="store_true")
    p.add_argument("--mode", type=int, default=0)
    driver.add_native_flags(p)
    return p.parse_args(argv)


if __name__ == "__main__":
    driver.run(parse(), side="source")
