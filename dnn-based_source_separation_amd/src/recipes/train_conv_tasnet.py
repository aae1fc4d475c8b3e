"""
Command-line trainer for Conv-TasNet on wsj0-mix style data, single- or multi-GPU:

    python -m recipes.train_conv_tasnet --train_wav_root ... --train_list_path ... [options]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m recipes.train_conv_tasnet ...

Option names and defaults follow the reference's egs/wsj0-mix/conv-tasnet/local/train.py:21-66 and train.sh:28-59
(paper-best model, Adam 1e-3, clip 5, PIT over SI-SDR, 4-s segments at 8 kHz).
"""
import argparse
import os

import torch
import torch.distributed as dist

from criterion.pit import PIT1d
from criterion.sdr import NegSISDR
from models.conv_tasnet import ConvTasNet

from .trainer import Trainer
from .wsj0mix import EvalDataLoader, TrainDataLoader, WaveEvalDataset, WaveTrainDataset, shard_for_rank


def _flag(v):
    return str(v).lower() in ("1", "true", "yes")


def build_parser():
    ap = argparse.ArgumentParser(description="Training of Conv-TasNet (fused MI355X path)")
    ap.add_argument("--train_wav_root", required=True)
    ap.add_argument("--valid_wav_root", required=True)
    ap.add_argument("--train_list_path", required=True)
    ap.add_argument("--valid_list_path", required=True)
    ap.add_argument("--sample_rate", "-sr", type=int, default=8000)
    ap.add_argument("--duration", type=float, default=4.0)
    ap.add_argument("--valid_duration", type=float, default=10.0)
    ap.add_argument("--enc_basis", default="trainable")
    ap.add_argument("--dec_basis", default="trainable")
    ap.add_argument("--enc_nonlinear", default=None)
    ap.add_argument("--n_basis", "-N", type=int, default=512)
    ap.add_argument("--kernel_size", "-L", type=int, default=16)
    ap.add_argument("--stride", type=int, default=None)
    ap.add_argument("--sep_bottleneck_channels", "-B", type=int, default=128)
    ap.add_argument("--sep_hidden_channels", "-H", type=int, default=512)
    ap.add_argument("--sep_skip_channels", "-Sc", type=int, default=128)
    ap.add_argument("--sep_kernel_size", "-P", type=int, default=3)
    ap.add_argument("--sep_num_blocks", "-R", type=int, default=3)
    ap.add_argument("--sep_num_layers", "-X", type=int, default=8)
    ap.add_argument("--dilated", type=_flag, default=True)
    ap.add_argument("--separable", type=_flag, default=True)
    ap.add_argument("--causal", type=_flag, default=False)
    ap.add_argument("--sep_nonlinear", default="prelu")
    ap.add_argument("--sep_norm", type=_flag, default=True)
    ap.add_argument("--mask_nonlinear", default="sigmoid")
    ap.add_argument("--n_sources", type=int, default=2)
    ap.add_argument("--criterion", default="sisdr", choices=["sisdr"])
    ap.add_argument("--optimizer", default="adam", choices=["adam"])
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--weight_decay", type=float, default=0.0)
    ap.add_argument("--max_norm", type=float, default=5.0)
    ap.add_argument("--batch_size", type=int, default=4, help="utterance segments per GPU")
    ap.add_argument("--epochs", type=int, default=100)
    ap.add_argument("--model_dir", default="./tmp/model")
    ap.add_argument("--loss_dir", default="./tmp/loss")
    ap.add_argument("--sample_dir", default="./tmp/sample")
    ap.add_argument("--continue_from", default=None)
    ap.add_argument("--overwrite", type=_flag, default=False)
    ap.add_argument("--num_workers", type=int, default=2)
    ap.add_argument("--seed", type=int, default=111)
    return ap


def main(argv=None):
    args = build_parser().parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo", rank=rank, world_size=world)
    torch.manual_seed(args.seed)
    if not torch.cuda.is_available():
        raise RuntimeError("the fused Conv-TasNet path needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    samples = int(args.sample_rate * args.duration)
    train_set = WaveTrainDataset(args.train_wav_root, args.train_list_path, samples=samples, overlap=samples // 2, n_sources=args.n_sources)
    valid_set = WaveEvalDataset(args.valid_wav_root, args.valid_list_path, max_samples=int(args.sample_rate * args.valid_duration), n_sources=args.n_sources)
    if rank == 0:
        print("Training dataset includes {} samples.".format(len(train_set)))
        print("Valid dataset includes {} samples.".format(len(valid_set)))
    def train_loader(epoch):
        shard = shard_for_rank(train_set, rank, world, seed=args.seed, epoch=epoch) if world > 1 else train_set
        return TrainDataLoader(shard, batch_size=args.batch_size, shuffle=True, drop_last=True, num_workers=args.num_workers)

    loader = {"train": train_loader(0), "valid": EvalDataLoader(valid_set, batch_size=1, shuffle=False)}
    args.reshard = train_loader if world > 1 else None      # every rank trains on a different subset each epoch

    stride = args.kernel_size // 2 if args.stride is None else args.stride
    model = ConvTasNet(args.n_basis, args.kernel_size, stride=stride, enc_basis=args.enc_basis, dec_basis=args.dec_basis,
                       enc_nonlinear=args.enc_nonlinear, sep_hidden_channels=args.sep_hidden_channels,
                       sep_bottleneck_channels=args.sep_bottleneck_channels, sep_skip_channels=args.sep_skip_channels,
                       sep_kernel_size=args.sep_kernel_size, sep_num_blocks=args.sep_num_blocks, sep_num_layers=args.sep_num_layers,
                       dilated=args.dilated, separable=args.separable, causal=args.causal, sep_nonlinear=args.sep_nonlinear,
                       sep_norm=args.sep_norm, mask_nonlinear=args.mask_nonlinear, n_sources=args.n_sources).to(dev)
    if rank == 0:
        print(model)
        print("# Parameters: {}".format(model.num_parameters), flush=True)
    trainer = Trainer(model, loader, PIT1d(NegSISDR(), n_sources=args.n_sources), args)
    trainer.run()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
