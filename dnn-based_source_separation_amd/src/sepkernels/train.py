"""
Data-parallel train step for the fused Conv-TasNet path: one process per GPU, utterances sharded across ranks,
every rank runs forward + SI-SDR/PIT + backward locally, then ONE all-reduce (RCCL over xGMI; `nccl` backend on
ROCm) of the flat fp32 gradient buffer, then global-norm clip + Adam fused in two kernels on the flat buffers.

Replaces the single-process nn.DataParallel step of reference egs/wsj0-mix/common/src/driver.py:141-157
(scatter / per-forward parameter broadcast / gather / loss+clip+Adam on GPU 0) -- SURVEY.md section 8(e).
Equal per-rank batches make mean-of-rank-means the global batch mean, so the averaged gradient equals the
reference's single-process result.  The collective is 19.9 MB per step (4,984,881 floats): latency-, not
bandwidth-bound on 7 x 153 GB/s xGMI links; it is issued in three asynchronous pieces (one per TCN block, last block
first) so that most of it hides under the rest of backward (SEPK_DDP_BUCKETS=0: one call after backward).

One call per pass: a step is ~360 kernel launches of fixed shapes, and driven from Python each costs ~30 us of interpreter time -- below
~10 utterances per GPU (the recipes train with 2 - 4) the eager step is bound by that, not by the GPU.  `record(mixture, sources)` runs ONE
step through the ordinary entry points while the binding records every launch (sepkernels.Sequence: entry point + arguments, forward,
criterion, backward, clip, Adam -- all of it library entry points, no torch kernel in between); later calls with the same shapes copy
their batch into the recorded input buffers and hand the list to sep_run_sequence, a C loop over the same entry points (include/sepkernels.h,
ABI 23).  The two scalars that change from step to step (Adam's step count, the learning rate) live in device memory for that
(sep_adam_step_dev).  No hipGraph: replays of the captured full-size step were measured wrong in half the runs on this stack
(profiles/r07_round5_experiments.md, r07m) and that path (FusedTrainStep.capture, GraphedStep) is gone.  With ranks > 1 the gradient buckets
cut the recorded list into segments: a replayed step runs a segment, issues that bucket's asynchronous all-reduce, runs the next segment ...,
waits for the exchange, then runs clip + Adam (`_seq_marks`).
"""
import os

import torch
import torch.distributed as dist

import sepkernels


class FusedTrainStep:
    def __init__(self, model, criterion, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=5.0,
                 process_group=None, distributed=None, uneven_batches=False, time_collectives=False, exercise_collectives=False, auto_record=None):
        self.model, self.criterion = model, criterion
        # auto_record: the first call on a GPU batch records the step (record()), later calls of that shape replay it; other shapes, other
        # criteria and ranks > 1 step eagerly.  Default: the SEPK_SEQUENCE switch (off unless SEPK_SEQUENCE=1).
        self.auto_record = (os.environ.get("SEPK_SEQUENCE", "0") == "1") if auto_record is None else bool(auto_record)
        self.lr, self.betas, self.eps, self.weight_decay, self.max_norm = lr, betas, eps, weight_decay, max_norm
        self.group = process_group
        self.distributed = dist.is_available() and dist.is_initialized() if distributed is None else distributed
        self.world = dist.get_world_size(process_group) if self.distributed else 1
        flat = model.flat_parameters()
        if flat is None:
            raise RuntimeError("FusedTrainStep needs the model's parameters co-located in one flat buffer")
        self.flat = flat
        # exercise_collectives: take the exchange path (broadcast, bucketed asynchronous all-reduces from inside backward, their waits and
        # event brackets) ALSO in a process group of one rank, where every collective is the identity: the RCCL call sequence of the N-GPU
        # step then runs on the single GPU a test box has (tests/test_gpu_model.py) instead of for the first time on the 8-GPU node
        self.comm = self.distributed and (self.world > 1 or bool(exercise_collectives))
        if self.comm:
            dist.broadcast(self.flat, src=0, group=process_group)      # replaces DataParallel's per-forward replicate
        self.gflat = torch.zeros_like(flat)
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)
        self.sqnorm = torch.zeros(1, device=flat.device, dtype=torch.float64)
        self.step_count = 0
        self.bucketed = os.environ.get("SEPK_DDP_BUCKETS", "1") != "0"
        # uneven_batches: ranks may hold DIFFERENT numbers of utterances in a step (a last batch that does not divide; the reference's
        # nn.DataParallel scatters such a batch unevenly and still takes the mean over all of it).  Each rank then back-propagates the SUM
        # over its utterances, the utterance counts are summed across ranks by a 1-element all-reduce hidden under the forward pass, and
        # the reduced gradient is divided by the global count.  Off by default: the recipes shard equally and drop the tail.
        self.uneven = bool(uneven_batches)
        # time_collectives: HIP events around the waits on the gradient all-reduces -> last_comm (bench.py's `ranks` block)
        self.time_collectives = bool(time_collectives)
        self.last_comm = None
        self.last_bucket_bytes = []
        # recorded-sequence state (see record())
        self._seq = None
        self._static = None
        self._step_dev = None
        self._lr_dev = None

    def _trainable_mask(self):
        """None when every parameter trains (the usual case: nothing extra runs), else a 0 / 1 vector over the flat buffer, rebuilt when the
        set of frozen parameters changes.  Frozen parameters (requires_grad = False) get a zero gradient before the norm is taken, and their
        values / moments are put back after the fused Adam (which would otherwise apply weight decay to them); alignment gaps hold zeros."""
        key = tuple(p.requires_grad for p in self._params())
        if all(key):
            return None
        if getattr(self, "_mask_key", None) != key or self._mask.device != self.flat.device:
            mask = torch.zeros_like(self.flat)
            for (off, n, _), p in zip(self._spans(), self._params()):
                if p.requires_grad:
                    mask[off:off + n] = 1.0
            self._mask, self._mask_key = mask, key
        return self._mask

    def exposed_comm_ms(self):
        """milliseconds the compute stream waited for the last step's gradient exchange (time_collectives=True; synchronises)"""
        if self.last_comm is None:
            return None
        self.last_comm[1].synchronize()
        return self.last_comm[0].elapsed_time(self.last_comm[1])

    def zero_grad(self):
        for p in self.model.parameters():
            p.grad = None

    def _rebind(self):
        """model.to() / .float() / load through _apply re-home the parameters into a NEW flat buffer: follow it (moments are kept
        when only the storage moved, and refused when the size changed) instead of running Adam on a buffer nobody reads."""
        flat = self.model.flat_parameters()
        if flat is self.flat:
            return
        if flat is None or flat.numel() != self.flat.numel():
            raise RuntimeError("the model's parameters are no longer co-located in a flat buffer of the size this step was built for")
        self.flat = flat
        self.gflat, self.m, self.v = (t.to(flat.device) for t in (self.gflat, self.m, self.v))
        self.sqnorm = self.sqnorm.to(flat.device)
        self._seq = None            # a recorded step writes through the OLD buffers' addresses: it has to be recorded again

    # ---- the whole step as one recorded launch sequence ---------------------------------------------------------------------
    def recordable(self):
        """None when record() can take this step, else the reason it cannot"""
        from criterion.pit import PIT
        from criterion.sdr import SISDR, NegSISDR
        if self.comm and self.uneven:
            return "uneven_batches weights the loss by a count that is all-reduced under the forward pass: eager only"
        if not getattr(self.model, "fused", False):
            return "the recorded step is offered for the fused kernel sequence only; this model runs the {} path".format(
                "derived-basis" if getattr(self.model, "fused_derived", False) else "staged" if getattr(self.model, "staged", False) else "composed")
        from criterion.pit import SinkPIT
        c = self.criterion
        pit_ok = isinstance(c, PIT) and type(c.criterion) in (SISDR, NegSISDR) and c.criterion.reduction in ("mean", "sum")
        sink_ok = type(c) is SinkPIT and type(c.criterion) in (SISDR, NegSISDR) and c.iteration >= 0 and c.coldness != 0
        if not (pit_ok or sink_ok):
            return ("the recorded step implements PIT and SinkPIT over SI-SDR / NegSI-SDR (criterion.pit.PIT1d(NegSISDR()), "
                    "criterion.pit.SinkPIT(NegSISDR(), ...)); other criteria run eagerly")
        if self._trainable_mask() is not None:
            return "frozen parameters (requires_grad = False) are handled by the eager step"
        if os.environ.get("SEPK_SIDE_STREAM", "0") == "1":
            return "SEPK_SIDE_STREAM=1 puts the weight gradients on a second stream: eager only"
        if getattr(self.model, "in_channels", 1) != 1:
            return "multi-channel input: the recorded criterion scores (B, n_sources, T) estimates"
        return None

    def record(self, mixture, sources):
        """Run ONE training step on this batch while recording every launch, and keep the list: later calls with batches of the same shape
        replay it through sep_run_sequence (one C-ABI call per step).  Returns the loss of the recorded step (it IS a training step).
        The step is the eager one's arithmetic, launch for launch; only the glue differs: clearing of accumulators by sep_memset, the
        weight bound by sep_absmax, the tail of PIT (batch mean, gradient weights of the chosen permutation) by sep_pit_finish instead of
        torch kernels.  reference: egs/wsj0-mix/common/src/driver.py:141-157."""
        from . import net as _net
        from criterion.sdr import NegSISDR
        why = self.recordable()
        if why is not None:
            raise RuntimeError("FusedTrainStep.record: " + why)
        self._rebind()
        K = sepkernels.backend()
        from criterion.pit import SinkPIT
        model, crit = self.model, self.criterion.criterion
        sink = type(self.criterion) is SinkPIT
        dev = self.flat.device
        B, n_src, T = sources.shape
        if tuple(mixture.shape) != (B, 1, T) or n_src != model.n_sources:
            raise ValueError("record: mixture {} / sources {} do not fit a {}-source model".format(tuple(mixture.shape), tuple(sources.shape), model.n_sources))
        f32 = dict(device=dev, dtype=torch.float32)
        self._static = (mixture.detach().to(**f32).contiguous().clone(), sources.detach().to(**f32).contiguous().clone())
        mix, src = self._static
        self._lr_dev = torch.tensor([self.lr], **f32)
        self._lr_host = self.lr
        self._step_dev = torch.tensor([self.step_count], device=dev, dtype=torch.int32)
        named = model._named_tensors()
        P = {k: v.detach() for k, v in named}
        offs, total = model._offsets, self.gflat.numel()
        G = {k: self.gflat[offs[k]:offs[k] + v.numel()].view(v.shape) for k, v in named}
        cfg = model.get_config()
        if not sink:
            patterns = self.criterion.patterns
            Pn = patterns.size(0)
            perms32 = patterns.to(device=dev, dtype=torch.int32).contiguous()
            scale = 1.0 / (B * n_src) if crit.reduction == "mean" else 1.0 / B
        sign = -1.0 if isinstance(crit, NegSISDR) else 1.0
        self.zero_grad()
        seq = sepkernels.Sequence()
        out = {}
        # ranks > 1: the gradient buckets of the eager step (one per TCN block, last block first, then block 0 + the head: _FusedConvTasNetFn.backward)
        # cut the list into segments -- `marks` holds (ops recorded when the bucket became final, first float, one past the last float); a
        # replay runs a segment, issues that bucket's asynchronous all-reduce, runs the next segment ...
        marks, works = [], []
        self.last_bucket_bytes = []
        R = cfg["sep_num_blocks"]
        starts = {r: offs["separator.tdcn.net.{}.net.0.bottleneck_conv1d.weight".format(r)] for r in range(R)}

        def bucket(lo, hi):
            marks.append((len(seq), lo, hi))
            works.append(self._bucket_allreduce(lo, hi))

        def on_ready(r):
            bucket(starts[r], total if r == R - 1 else starts[r + 1])
        bucketed = self.comm and self.bucketed
        with torch.no_grad(), sepkernels.recording(seq):
            amax = None
            if sepkernels.gemm_arith() == sepkernels.ARITH_F16X3 and hasattr(K, "absmax"):
                amax = torch.empty(1, **f32)
                K.absmax(self.flat, amax, self.flat.numel())          # (alignment gaps of the flat buffer hold zeros)
            prev = sepkernels.set_weights_amax(amax)
            try:
                est, _, sv = _net._forward(cfg, P, mix, False, True)
                est3 = est.view(B, n_src, T)
                # PIT over the SI-SDR pair matrix (criterion/pit.py::_fused_pit, criterion/sdr.py::_SISDRPairsFn), launch for launch
                dots = K.zeros(B, n_src, n_src, device=dev, dtype=torch.float64)
                tt = K.zeros(B, n_src, device=dev, dtype=torch.float64)
                xx = K.zeros(B, n_src, device=dev, dtype=torch.float64)
                val = torch.empty(B, n_src, n_src, **f32)
                K.sisdr_dots(est3, src, dots, tt, xx, B, n_src, T, True)
                K.sisdr_from_dots(dots, tt, xx, val, B, n_src, True, crit.eps)
                loss = torch.empty(1, **f32)
                gw = torch.empty(B, n_src, n_src, **f32)
                if not sink:
                    best_val = torch.empty(B, **f32)
                    best_idx = torch.empty(B, device=dev, dtype=torch.int64)
                    K.pit_search(val, perms32, Pn, n_src, B, True, crit.reduction == "mean", best_val, best_idx)      # max SI-SDR = min NegSI-SDR
                    pattern = torch.empty(B, n_src, device=dev, dtype=torch.int64)
                    K.pit_finish(best_val, best_idx, perms32, Pn, n_src, B, sign, scale, loss, gw, pattern)
                else:
                    # Sinkhorn PIT (criterion/pit.py::sinkpit): costs C = -SI-SDR for both criterion classes, per-item losses from the log-domain
                    # iterations, their batch mean (sign flipped for a maximised criterion), and back: dL/dC through every iteration, dL/d SI-SDR = -dL/dC
                    iters, cold = int(self.criterion.iteration), float(self.criterion.coldness)
                    C = torch.empty(B, n_src, n_src, **f32)
                    K.axpby(val, -1.0, None, 0.0, C, B * n_src * n_src)
                    zwork = torch.empty(B, 2 * iters + 1, n_src, n_src, device=dev, dtype=torch.float64)
                    item_loss, Pm = torch.empty(B, **f32), torch.empty(B, n_src, n_src, **f32)
                    K.sinkhorn_fwd(C, zwork, item_loss, Pm, B, n_src, cold, iters)
                    lsign = -1.0 if bool(crit.maximize) else 1.0
                    K.pit_finish(item_loss, None, None, 0, n_src, B, lsign, 0.0, loss, None, None)
                    dloss = torch.full((B,), lsign / B, **f32)                   # written once, before the recording's first launch reads it
                    dC = torch.empty(B, n_src, n_src, **f32)
                    K.sinkhorn_bwd(C, zwork, dloss, dC, B, n_src, cold, iters)
                    K.axpby(dC, -1.0, None, 0.0, gw, B * n_src * n_src)
                    pattern = Pm                                                 # (the soft assignment; last_pattern takes its argmax on demand)
                d_est = torch.empty_like(est3)
                K.sisdr_bwd(est3, src, dots, tt, xx, gw, d_est, B, n_src, T, True, crit.eps)
                _net._backward(cfg, P, sv, d_est.view(B, n_src, 1, T), G, on_ready if bucketed else None, False)
                if bucketed:
                    bucket(0, starts[1] if R > 1 else total)
                elif self.comm:
                    bucket(0, total)
            finally:
                sepkernels.set_weights_amax(prev)
            self._wait_buckets(works)
            n = self.gflat.numel()
            K.memset(self.sqnorm, 0)
            if self.max_norm and self.max_norm > 0:
                K.sqnorm(self.gflat, self.sqnorm, n)
            K.adam_step_dev(self.flat, self.gflat, self.m, self.v, self.sqnorm, n, self._lr_dev, self._step_dev, self.betas[0], self.betas[1],
                            self.eps, self.weight_decay, float(self.max_norm or 0.0), 1.0 / self.world)
            out["loss"], out["pattern"] = loss, pattern
        self.step_count += 1
        self._seq = seq
        self._seq_marks = marks
        self.last_buckets = len(marks)
        self._seq_loss, self._seq_pattern = out["loss"].view(()), out["pattern"]
        self._seq_key = (tuple(mixture.shape), tuple(sources.shape), tuple(self.betas), self.eps, self.weight_decay, self.max_norm,
                         sepkernels.gemm_arith(), id(self.criterion))
        return self._seq_loss

    @property
    def last_pattern(self):
        """(B, n_sources) int64 permutation of the last recorded / replayed step (PIT: the chosen one; SinkPIT: argmax of the soft assignment)"""
        p = getattr(self, "_seq_pattern", None)
        if p is None:
            return None
        return p if p.dtype == torch.int64 else torch.argmax(p, dim=2)

    def _seq_valid(self, mixture, sources):
        return self._seq is not None and self._seq_key == (tuple(mixture.shape), tuple(sources.shape), tuple(self.betas), self.eps, self.weight_decay,
                                                           self.max_norm, sepkernels.gemm_arith(), id(self.criterion))

    def _replay(self, mixture, sources):
        if self.model.flat_parameters() is not self.flat:      # model.to() / .float() since the recording: the list is stale
            self._seq = None
            return self._eager(mixture, sources)
        self._static[0].copy_(mixture)
        self._static[1].copy_(sources)
        if self._lr_host != self.lr:
            self._lr_dev.fill_(self.lr)
            self._lr_host = self.lr
        if not self._seq_marks:
            self._seq.run()
        else:
            pos, works = 0, []
            self.last_bucket_bytes = []
            for upto, lo, hi in self._seq_marks:
                self._seq.run(pos, upto)
                works.append(self._bucket_allreduce(lo, hi))
                pos = upto
            self._wait_buckets(works)
            self._seq.run(pos, len(self._seq))
        self.step_count += 1
        return self._seq_loss

    def _bucket_allreduce(self, lo, hi):
        self.last_bucket_bytes.append(4 * (hi - lo))
        return dist.all_reduce(self.gflat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _wait_buckets(self, works):
        """the compute stream waits for the gradient exchange (HIP events around the waits when time_collectives is on)"""
        if not works:
            return
        timed = self.time_collectives and self.gflat.is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for w in works:
            w.wait()
        if timed:
            e1.record()
            self.last_comm = (e0, e1)

    def __call__(self, mixture, sources):
        if self._seq is not None:
            if self._seq_valid(mixture, sources):
                return self._replay(mixture, sources)
        elif (self.auto_record and mixture.is_cuda and mixture.dim() == 3 and sources.dim() == 3 and mixture.shape[1] == 1
              and sources.shape[0] == mixture.shape[0] and sources.shape[2] == mixture.shape[2] and self.recordable() is None):
            return self.record(mixture, sources)
        return self._eager(mixture, sources)

    def _eager(self, mixture, sources):
        K = sepkernels.backend()
        model = self.model
        self._rebind()
        self.zero_grad()
        model._grad_sink = self.gflat                 # backward writes every gradient straight into the flat buffer
        model._sink_placed = False
        works = []
        self.last_bucket_bytes = []
        count_work = None
        if self.comm and self.uneven:
            count = torch.tensor([float(mixture.shape[0])], device=self.gflat.device, dtype=torch.float32)
            count_work = (dist.all_reduce(count, op=dist.ReduceOp.SUM, group=self.group, async_op=True), count)
        if self.comm and self.bucketed:
            # one asynchronous RCCL all-reduce per TCN block, issued as soon as the block's gradients are final, so the
            # exchange of the late layers travels under the differentiation of the early ones (3 buckets at paper-best)
            def bucket_ready(lo, hi):
                self.last_bucket_bytes.append(4 * (hi - lo))
                works.append(dist.all_reduce(self.gflat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            model._grad_bucket_hook = bucket_ready
        try:
            est = model(mixture)
            loss, _ = self.criterion(est, sources)
            (loss * float(mixture.shape[0]) if count_work is not None else loss).backward()
        finally:
            model._grad_sink = None
            model._grad_bucket_hook = None
        if not getattr(model, "fused", True):
            # the module-by-module / staged / derived-basis paths of ConvTasNet leave their gradients in .grad like any torch module: put
            # them where the flat step reads them (a parameter without a gradient -- a fixed Fourier basis, a frozen tensor -- contributes
            # zeros).  The derived-basis path has written the separator's gradients in place already (_sink_placed): only the rest moves.
            placed = bool(getattr(model, "_sink_placed", False))
            model._sink_placed = False
            if not placed:
                self.gflat.zero_()
            dst, src = [], []
            for (off, n, _), (name, q) in zip(self._spans(), [(k, v) for k, v in model.named_parameters() if v.is_floating_point()]):
                if placed and name.startswith("separator."):
                    continue
                if q.grad is not None:
                    dst.append(self.gflat[off:off + n])
                    src.append(q.grad.reshape(-1))
                elif placed:
                    self.gflat[off:off + n].zero_()
            if dst:
                torch._foreach_copy_(dst, src)
        self.last_buckets = len(works)
        grad_scale = 1.0 / self.world
        if self.comm:
            timed = self.time_collectives and self.gflat.is_cuda
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if works:
                for w in works:
                    w.wait()
            else:
                self.last_bucket_bytes = [4 * self.gflat.numel()]
                dist.all_reduce(self.gflat, op=dist.ReduceOp.SUM, group=self.group)
            if timed:
                e1.record()
                self.last_comm = (e0, e1)         # elapsed = what the compute stream had to wait for the exchange (nothing else lies between)
            if count_work is not None:
                count_work[0].wait()
                grad_scale = 1.0 / float(count_work[1].item())
        n = self.gflat.numel()
        mask = self._trainable_mask()
        frozen = None
        if mask is not None:
            self.gflat.mul_(mask)         # frozen parameters (requires_grad = False) take no part in the norm and are not moved by Adam
            frozen = self.flat * (1.0 - mask)
        self.sqnorm.zero_()
        if self.max_norm and self.max_norm > 0:
            K.sqnorm(self.gflat, self.sqnorm, n)
        self.step_count += 1
        if self._step_dev is not None:
            self._step_dev.fill_(self.step_count)          # an eager step between replays (other shapes) keeps the device count in step
        K.adam_step(self.flat, self.gflat, self.m, self.v, self.sqnorm, n, self.lr, self.betas[0], self.betas[1], self.eps,
                    self.weight_decay, float(self.max_norm or 0.0), grad_scale, self.step_count)
        if frozen is not None:
            self.flat.mul_(mask).add_(frozen)
            self.m.mul_(mask)
            self.v.mul_(mask)
        return loss.detach()

    # ---- optimizer state in torch.optim.Adam's state_dict layout (checkpoint interchange with the reference's
    #      driver.py:208-226 / 51-68: `optim_dict`) ---------------------------------------------------------
    def _params(self):
        """the parameters that live in the flat buffer: every floating-point one (a Fourier basis carries an integer `time_seq`)"""
        return [p for p in self.model.parameters() if p.is_floating_point()]

    def _spans(self):
        base = self.flat.data_ptr()
        spans = []
        for p in self._params():
            off = (p.data_ptr() - base) // self.flat.element_size()
            if off < 0 or off + p.numel() > self.flat.numel():
                raise RuntimeError("parameter is not a view of the flat buffer")
            spans.append((off, p.numel(), tuple(p.shape)))
        return spans

    def _state_indices(self):
        """position of every flat-buffer parameter in model.parameters() -- torch.optim.Adam(model.parameters()) numbers ALL of them, also
        the integer `time_seq` a Fourier basis carries, so the state of a checkpoint the reference wrote (or will read) is keyed by these"""
        return [i for i, p in enumerate(self.model.parameters()) if p.is_floating_point()]

    def optim_state_dict(self):
        state = {}
        for i, (off, n, shape) in zip(self._state_indices(), self._spans()):
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": self.m[off:off + n].detach().reshape(shape).cpu().clone(),
                        "exp_avg_sq": self.v[off:off + n].detach().reshape(shape).cpu().clone()}
        n_all = sum(1 for _ in self.model.parameters())
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "params": list(range(n_all))}
        return {"state": state, "param_groups": [group]}

    def load_optim_state_dict(self, sd):
        spans = self._spans()
        groups = sd.get("param_groups", [])
        if groups:
            g = groups[0]
            self.lr = g.get("lr", self.lr)
            self.betas = tuple(g.get("betas", self.betas))
            self.eps = g.get("eps", self.eps)
            self.weight_decay = g.get("weight_decay", self.weight_decay)
        state = sd.get("state", {})
        steps = []
        for i, (off, n, _) in zip(self._state_indices(), spans):
            st = state.get(i, state.get(str(i)))
            if st is None:
                continue
            if st["exp_avg"].numel() != n:
                raise ValueError("optimizer state {} has {} elements, the parameter at that position {}".format(i, st["exp_avg"].numel(), n))
            self.m[off:off + n].copy_(st["exp_avg"].reshape(-1).to(self.m.device, self.m.dtype))
            self.v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1).to(self.v.device, self.v.dtype))
            steps.append(int(float(st["step"])))
        if steps:
            if len(set(steps)) != 1:
                raise ValueError("per-parameter Adam step counts differ; the fused step keeps one")
            self.step_count = steps[0]
        # a recorded step carries betas / eps / weight_decay as launch constants (its key: a change makes __call__ step eagerly until
        # record() is called again) and the step count in device memory: refresh the count
        if self._step_dev is not None:
            self._step_dev.fill_(self.step_count)
