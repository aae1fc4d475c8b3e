"""
Trainer / Tester for the fused Conv-TasNet path.

Behaviour follows the reference's TrainerBase / TesterBase (egs/wsj0-mix/common/src/driver.py:20-131, 132-226, 228-370):
epoch = train pass + validation pass; `best.pth` on a new best validation loss, `last.pth` every epoch; after the
validation loss fails to improve on the previous epoch 3 times in a row the learning rate is halved each further time,
after 10 the run stops; `--continue_from` resumes, an existing `best.pth` is refused unless `overwrite`.

What is different: one process per GPU (torchrun) instead of nn.DataParallel, the step is
`sepkernels.train.FusedTrainStep` (forward + PIT + backward + one RCCL all-reduce + fused clip/Adam on flat buffers),
batches arrive through a pinned, asynchronous prefetcher, and only rank 0 validates / writes files.

Checkpoint format = the reference's (driver.py:208-226): `model.get_config()` keys + `state_dict`, `optim_dict`
(torch.optim.Adam layout, so either side can resume the other's file), `best_loss`, `no_improvement`, `train_loss`,
`valid_loss`, `epoch`.
"""
import os
import time

import torch

from utils.checkpoint import load_checkpoint
import torch.distributed as dist

from sepkernels.train import FusedTrainStep

from .audio_io import write_wav
from .wsj0mix import DevicePrefetcher

BITS_PER_SAMPLE_WSJ0 = 16


def _is_dist():
    return dist.is_available() and dist.is_initialized()


def _rank():
    return dist.get_rank() if _is_dist() else 0


class Trainer:
    def __init__(self, model, loader, pit_criterion, args):
        """loader = {'train': ..., 'valid': ...}; args: namespace with model_dir, loss_dir, sample_dir, epochs, lr,
        max_norm, continue_from, overwrite, sample_rate (and optionally weight_decay, use_cuda)."""
        self.model, self.pit_criterion = model, pit_criterion
        self.train_loader, self.valid_loader = loader["train"], loader["valid"]
        self.device = next(model.parameters()).device
        self.sample_rate = args.sample_rate
        self.max_norm = args.max_norm
        self.epochs = args.epochs
        self.model_dir, self.loss_dir, self.sample_dir = args.model_dir, args.loss_dir, args.sample_dir
        self.is_main = _rank() == 0
        if self.is_main:
            for d in (self.model_dir, self.loss_dir, self.sample_dir):
                os.makedirs(d, exist_ok=True)
        # the overwrite guard of the reference (driver.py:70-76) is decided on rank 0 and SHARED before any collective of the
        # step is issued: a rank-0-only exception would leave the other ranks hanging in their first all-reduce
        best = os.path.join(self.model_dir, "best.pth")
        refuse = bool(self.is_main and not getattr(args, "continue_from", None) and os.path.exists(best) and not getattr(args, "overwrite", False))
        if _is_dist() and dist.get_world_size() > 1:
            flag = torch.tensor([int(refuse)], device=self.device if dist.get_backend() == "nccl" else "cpu")
            dist.broadcast(flag, src=0)
            refuse = bool(flag.item())
        if refuse:
            raise ValueError("{} already exists. If you continue to run, set --overwrite to be True.".format(best))
        # auto_record: the first batch of the training shape records the step's launch list, every later batch of that shape is ONE C-ABI call
        # (sep_run_sequence) -- at the recipes' batch sizes (2 - 4 utterances) the eager step is bound by its ~360 Python launches (7.1 against
        # 9.6 ms per step at 4).  Other shapes (a short last batch), other criteria and CPU tensors step eagerly.  SEPK_SEQUENCE=0 turns it off.
        self.step = FusedTrainStep(model, pit_criterion, lr=args.lr, weight_decay=getattr(args, "weight_decay", 0.0),
                                   max_norm=args.max_norm or 0.0, auto_record=os.environ.get("SEPK_SEQUENCE", "1") != "0")
        self.reshard = getattr(args, "reshard", None)       # callable(epoch) -> this rank's train loader for that epoch, or None
        self.train_loss = torch.empty(self.epochs)
        self.valid_loss = torch.empty(self.epochs)
        if getattr(args, "continue_from", None):
            ck = load_checkpoint(args.continue_from, getattr(args, "trust_pickle", None))
            self.start_epoch = ck["epoch"]
            self.train_loss[:self.start_epoch] = ck["train_loss"][:self.start_epoch]
            self.valid_loss[:self.start_epoch] = ck["valid_loss"][:self.start_epoch]
            self.best_loss = ck["best_loss"]
            self.prev_loss = float(self.valid_loss[self.start_epoch - 1])
            self.no_improvement = ck["no_improvement"]
            model.load_state_dict(ck["state_dict"])
            self.step.load_optim_state_dict(ck["optim_dict"])
        else:
            self.start_epoch = 0
            self.best_loss = float("infinity")
            self.prev_loss = float("infinity")
            self.no_improvement = 0

    # ---- epoch loops -----------------------------------------------------------------------------------
    def run(self):
        for epoch in range(self.start_epoch, self.epochs):
            t0 = time.time()
            train_loss = self.run_one_epoch_train(epoch)
            valid_loss = self.run_one_epoch_eval(epoch)
            if self.is_main:
                print("[Epoch {}/{}] loss (train): {:.5f}, loss (valid): {:.5f}, {:.3f} [sec]".format(
                    epoch + 1, self.epochs, train_loss, valid_loss, time.time() - t0), flush=True)
            self.train_loss[epoch] = train_loss
            self.valid_loss[epoch] = valid_loss
            stop = False
            if valid_loss < self.best_loss:
                self.best_loss = valid_loss
                self.no_improvement = 0
                self.save_model(epoch, os.path.join(self.model_dir, "best.pth"))
            elif valid_loss >= self.prev_loss:
                self.no_improvement += 1
                if self.no_improvement >= 10:
                    stop = True
                elif self.no_improvement >= 3:
                    if self.is_main:
                        print("Learning rate: {} -> {}".format(self.step.lr, 0.5 * self.step.lr))
                    self.step.lr *= 0.5
            else:
                self.no_improvement = 0
            if stop:
                if self.is_main:
                    print("Stop training")
                break
            self.prev_loss = valid_loss
            self.save_model(epoch, os.path.join(self.model_dir, "last.pth"))
            if self.is_main:
                torch.save({"train_loss": self.train_loss[:epoch + 1].clone(), "valid_loss": self.valid_loss[:epoch + 1].clone()},
                           os.path.join(self.loss_dir, "loss.pth"))

    def run_one_epoch_train(self, epoch):
        self.model.train()
        if self.reshard is not None:                        # a fresh shard of the training set per epoch (same permutation on every rank)
            self.train_loader = self.reshard(epoch)
        total = torch.zeros((), device=self.device, dtype=torch.float64)   # no .item() per step: the step stays asynchronous
        n = 0
        for idx, (mixture, sources) in enumerate(DevicePrefetcher(self.train_loader, self.device)):
            loss = self.step(mixture, sources)
            total += loss.double()
            n += 1
            if (idx + 1) % 100 == 0 and self.is_main:
                print("[Epoch {}/{}] iter {}/{} loss: {:.5f}".format(epoch + 1, self.epochs, idx + 1, len(self.train_loader), loss.item()), flush=True)
        if _is_dist():
            dist.all_reduce(total)
            total /= dist.get_world_size()
        return total.item() / max(n, 1)

    def run_one_epoch_eval(self, epoch):
        """Rank 0 validates every utterance (B = 1, variable length); the scalar is broadcast so all ranks take the same
        scheduling decisions."""
        self.model.eval()
        valid = torch.zeros((), device=self.device, dtype=torch.float64)
        if self.is_main:
            n_valid = len(self.valid_loader.dataset)
            with torch.no_grad():
                for idx, (mixture, sources, ids) in enumerate(DevicePrefetcher(self.valid_loader, self.device)):
                    output = self.model(mixture)
                    loss, _ = self.pit_criterion(output, sources, batch_mean=False)
                    valid += loss.sum(dim=0).double()
                    if idx < 5:
                        self._save_samples(epoch, ids[0], mixture[0], output[0])
            valid /= max(n_valid, 1)
        if _is_dist():
            dist.broadcast(valid, src=0)
        return valid.item()

    def _save_samples(self, epoch, ID, mixture, estimates):
        save_dir = os.path.join(self.sample_dir, ID)
        os.makedirs(save_dir, exist_ok=True)
        mix = mixture.reshape(-1, mixture.shape[-1]).cpu()
        write_wav(os.path.join(save_dir, "mixture.wav"), mix / mix.abs().max().clamp_min(1e-12), self.sample_rate, BITS_PER_SAMPLE_WSJ0)
        for k, est in enumerate(estimates.cpu()):
            est = est.reshape(-1, est.shape[-1])
            write_wav(os.path.join(save_dir, "epoch{}-{}.wav".format(epoch + 1, k + 1)), est / est.abs().max().clamp_min(1e-12),
                      self.sample_rate, BITS_PER_SAMPLE_WSJ0)

    # ---- checkpoint --------------------------------------------------------------------------------------
    def save_model(self, epoch, model_path="./tmp.pth"):
        if not self.is_main:
            return
        ck = self.model.get_config()
        ck["state_dict"] = {k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()}
        ck["optim_dict"] = self.step.optim_state_dict()
        ck["best_loss"] = self.best_loss
        ck["no_improvement"] = self.no_improvement
        ck["train_loss"] = self.train_loss
        ck["valid_loss"] = self.valid_loss
        ck["epoch"] = epoch + 1
        torch.save(ck, model_path)


class Tester:
    __test__ = False        # (not a pytest class)
    """Variable-length inference over a test list: PIT loss, loss improvement over the unprocessed mixture, SI-SDR
    improvement; writes up to 10 example utterances.  (The reference additionally calls mir_eval / PESQ, which are
    not part of this path: driver.py:277-370.)"""

    def __init__(self, model, loader, pit_criterion, args):
        self.model, self.loader, self.pit_criterion = model, loader, pit_criterion
        self.sample_rate, self.n_sources = args.sample_rate, args.n_sources
        self.out_dir = os.path.abspath(args.out_dir) if getattr(args, "out_dir", None) else None
        if self.out_dir:
            os.makedirs(self.out_dir, exist_ok=True)
        self.device = next(model.parameters()).device
        if getattr(args, "model_path", None):
            ck = load_checkpoint(args.model_path, getattr(args, "trust_pickle", None))
            model.load_state_dict(ck["state_dict"])

    def run(self):
        from criterion.sdr import sisdr
        self.model.eval()
        n = len(self.loader.dataset)
        tot_loss = tot_imp = tot_sisdri = 0.0
        print("ID, Loss, Loss improvement, SI-SDR improvement", flush=True)
        with torch.no_grad():
            for idx, (mixture, sources, ids) in enumerate(DevicePrefetcher(self.loader, self.device)):
                output = self.model(mixture)
                rep = mixture.expand(-1, self.n_sources, -1).contiguous()
                loss_mix, _ = self.pit_criterion(rep, sources, batch_mean=False)
                loss, perm = self.pit_criterion(output, sources, batch_mean=False)
                est = output[0][perm[0]]                                  # estimates reordered to the targets' order
                sisdri = (sisdr(est, sources[0]) - sisdr(rep[0], sources[0])).mean().item()
                l, lm = loss.sum().item(), loss_mix.sum().item()
                print("{}, {:.3f}, {:.3f}, {:.3f}".format(ids[0], l, lm - l, sisdri), flush=True)
                tot_loss += l
                tot_imp += lm - l
                tot_sisdri += sisdri
                if idx < 10 and self.out_dir:
                    mix = mixture[0].cpu()
                    write_wav(os.path.join(self.out_dir, "{}.wav".format(ids[0])), mix / mix.abs().max().clamp_min(1e-12),
                              self.sample_rate, BITS_PER_SAMPLE_WSJ0)
                    for k in range(self.n_sources):
                        e = est[k:k + 1].cpu()
                        write_wav(os.path.join(self.out_dir, "{}_{}-estimated.wav".format(ids[0], k + 1)),
                                  e / e.abs().max().clamp_min(1e-12), self.sample_rate, BITS_PER_SAMPLE_WSJ0)
        res = {"loss": tot_loss / n, "loss_improvement": tot_imp / n, "sisdr_improvement": tot_sisdri / n}
        print("Loss: {loss:.3f}, loss improvement: {loss_improvement:.3f}, SI-SDR improvement: {sisdr_improvement:.3f}".format(**res))
        return res
