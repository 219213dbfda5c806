#!/usr/bin/env python3
"""Target-domain trainer — the flags of the reference's main_target.py that its launch scripts (scripts/target/*.bash) use,
native step.  Methods (all native): vae_train, domain_adaptation (student/teacher Joint nets, binarised or confident pseudo-labels,
domain_loss_type 0 / 8 / 9 / 11-16, --only_pseudo, --turn_epoch, --lambda_vae_warmup, optional KL term, optional EMA teacher, test-time
training with --val_finetune), discriminator_train, domain_adaptation_dis.  Uses the
utils/evaluation.py epsilon (1e-6), as main_target.py does (it imports avg_dsc from there, main_target.py:23)."""
import argparse

from vae_segmentation_amd import driver


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("prefix", help="prefix")
    p.add_argument("-G", "--GPU", default="0,1,2,3", help="kept for CLI compatibility; ranks come from torchrun")
    p.add_argument("-b", "--batch_size", type=int, default=4)
    p.add_argument("-E", "--max_epoch", type=int, default=1600)
    p.add_argument("--save_epoch", type=int, default=50)
    p.add_argument("--eval_epoch", type=int, default=50)
    p.add_argument("--turn_epoch", type=int, default=-1)
    p.add_argument("-S", "--softrelu", type=int, default=0)
    p.add_argument("-M", "--method", default="domain_adaptation")
    p.add_argument("-R", "--data_root", default="../nih_data/numpy_data/")
    p.add_argument("-V", "--val_data_root", default="../nih_data/numpy_data/")
    p.add_argument("-l", "--data_path", default="Multi_all.json")
    p.add_argument("-t", "--train_list", default="MSD_train")
    p.add_argument("-v", "--val_list", default="MSD_val")
    p.add_argument("--load_prefix", default=None)
    p.add_argument("--checkpoint_name", default="best_model.ckpt")
    p.add_argument("--load_prefix_vae", default=None)
    p.add_argument("--load_prefix_joint", default=None)
    p.add_argument("--pan_index", default="1")
    p.add_argument("--lambda_vae", type=float, default=1.0)
    p.add_argument("--lambda_vae_warmup", type=int, default=0)
    p.add_argument("--lr_seg", type=float, default=1e-2)
    p.add_argument("--lr_vae", type=float, default=0)
    p.add_argument("--domain_loss_type", type=int, default=0)
    p.add_argument("--kl", action="store_true")
    p.add_argument("--use_confident_binarize", action="store_true")
    p.add_argument("--pseudo_save_epoch", type=int, default=0)
    p.add_argument("--update_every_iteration", action="store_true")
    p.add_argument("--alpha", type=float, default=0.995)
    p.add_argument("--seg_dropout", type=float, default=0.0)
    p.add_argument("--vae_decoder_dropout", type=float, default=0.0)
    p.add_argument("--val_finetune", type=int, default=0, help="test-time training iterations per validation case (main_target.py:72)")
    p.add_argument("--lr_finetune", type=float, default=1e-2)
    p.add_argument("--only_pseudo", action="store_true")
    p.add_argument("--test_only", action="store_true")
    p.add_argument("--adam", action="store_true")
    driver.add_native_flags(p)
    return p.parse_args(argv)


if __name__ == "__main__":
    driver.run(parse(), side="target")
