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

hipGraph: a step is ~400 kernel launches with fixed shapes, and the gaps between them were 10-17 % of the profiled wall time
(profiles/r02c_kernel_stats.md).  `capture(mixture, sources)` records ONE step -- forward, criterion, backward on both streams,
clip, Adam -- into a graph over static input buffers and the caching allocator's private pool; later calls with the same shapes
copy their batch in and replay it.  The two scalars that change from step to step (Adam's step count, the learning rate) live in
device memory for that (sep_adam_step_dev).  Single-process only: the RCCL all-reduce stays eager.
"""
import os

import torch
import torch.distributed as dist

import sepkernels


class FusedTrainStep:
    def __init__(self, model, criterion, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=5.0,
                 process_group=None, distributed=None, uneven_batches=False, time_collectives=False, exercise_collectives=False):
        self.model, self.criterion = model, criterion
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
        # graph state (see capture())
        self._graph = None
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
        self._graph = None          # a captured step writes through the OLD buffers' addresses: it has to be recorded again

    # ---- hipGraph of the whole step -----------------------------------------------------------------------------------
    def capture(self, mixture, sources, warmup=3):
        """Record one step on inputs of this shape.  `warmup` eager steps run first on a side stream (allocator and lazy
        initialisation settle there, as torch.cuda.graphs asks for); they ARE training steps.  Returns the loss of the captured step
        (which is executed too).
        EXPERIMENTAL on this stack (ROCm 7.0 / torch 2.10): the replayed step equals the eager one on the small configurations of the
        tests, but replays of the paper-best step at 16 utterances ended with wrong losses (inf, 49.98 for 0.0812) in about half of 20 runs
        (profiles/r07_round5_experiments.md, r07m) -- the runtime's pre-built graph packets: with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 every replay is
        right to the last bit and no faster than eager launches (r07o).  The eager step is the product's step, and bench.py's default."""
        if self.comm:
            raise RuntimeError("graph capture of the train step is single-process only (the gradient all-reduce stays eager)")
        if not getattr(self.model, "fused", True):
            # measured (profiles/r07_round5_experiments.md, r07k): the staged causal sequence records, but its replay ends in a GPU memory
            # access fault -- the layer-by-layer autograd Functions own workspaces the capture does not pin.  Refused instead of offered.
            raise RuntimeError("graph capture is offered for the fused kernel sequence only; this model runs the {} path".format(
                "derived-basis" if getattr(self.model, "fused_derived", False) else "staged" if getattr(self.model, "staged", False) else "composed"))
        dev = self.flat.device
        self._static = (torch.empty_like(mixture), torch.empty_like(sources))
        self._static[0].copy_(mixture)
        self._static[1].copy_(sources)
        self._lr_dev = torch.tensor([self.lr], device=dev, dtype=torch.float32)
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._eager(*self._static)
        torch.cuda.current_stream(dev).wait_stream(s)
        self._step_dev = torch.tensor([self.step_count], device=dev, dtype=torch.int32)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._static_loss = self._eager(*self._static, graph=True)
        self._graph = g
        self._graph_shapes = (tuple(mixture.shape), tuple(sources.shape))
        self._graph_hyper = (tuple(self.betas), self.eps, self.weight_decay)
        # capture only records: run the step it recorded once, so that the caller sees warmup + 1 steps done
        return self._replay()

    def _replay(self):
        if self.model.flat_parameters() is not self.flat:      # model.to() / .float() since the capture: the graph is stale
            self._graph = None
            return self._eager(*self._static)
        self._lr_dev.fill_(self.lr)
        self._graph.replay()
        self.step_count += 1
        return self._static_loss

    def __call__(self, mixture, sources):
        if self._graph is not None and (tuple(mixture.shape), tuple(sources.shape)) == self._graph_shapes:
            self._static[0].copy_(mixture)
            self._static[1].copy_(sources)
            return self._replay()
        return self._eager(mixture, sources)

    def _eager(self, mixture, sources, graph=False):
        K = sepkernels.backend()
        model = self.model
        if not graph:
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
        if graph:
            K.adam_step_dev(self.flat, self.gflat, self.m, self.v, self.sqnorm, n, self._lr_dev, self._step_dev, self.betas[0], self.betas[1],
                            self.eps, self.weight_decay, float(self.max_norm or 0.0), grad_scale)
        else:
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
        # a captured step carries betas / eps / weight_decay as launch constants and the step count in device memory: refresh the count,
        # and drop the graph if any of the constants moved (capture() records it again)
        if self._step_dev is not None:
            self._step_dev.fill_(self.step_count)
        if self._graph is not None and getattr(self, "_graph_hyper", None) != (tuple(self.betas), self.eps, self.weight_decay):
            self._graph = None



class GraphedStep:
    """forward + criterion + backward [+ clip] + optimizer step of ANY separator of this tree, recorded once into a hipGraph and replayed.

    For separators whose parameters are ordinary tensors (DPRNN-TasNet, DPTNet ...; ConvTasNet has FusedTrainStep.capture): shapes are
    fixed and nothing in these steps reads back to the host, so the whole step can be one graph launch.  `optimizer` must keep its state
    on the device and step without a host sync (torch.optim.Adam(..., capturable=True)).  Measured on MI355X (profiles/r04d_dual.txt):
    replay equals the eager step to 1e-4 (tests/test_gpu_model.py) and takes the SAME time at the recipes' sizes (DPRNN-TasNet 36.9 ms
    either way: the step is bound by its kernels) -- it pays where the launches are the bottleneck (small batches, short utterances).
    Models with dropout diverged under replay on this stack (GALRNet, SepFormer: loss inf): use it for dropout-free configurations.

        step = GraphedStep(model, criterion, optimizer, max_norm=5.0)
        loss = step(mixture, sources)          # first call: three eager steps on a side stream (allocator warm-up), capture, then replay

    The warm-up and the recording run real optimizer steps; `restore=True` (default) puts parameters, gradients-free, and the Adam
    moments / step counts back IN PLACE afterwards (the graph holds their addresses), so that the first replay is the first step.
    Batches of another shape need another GraphedStep.  reference: egs/wsj0-mix/common/src/driver.py:132-164 (the eager step)."""

    def __init__(self, model, criterion, optimizer, max_norm=None, warmup=3, restore=True):
        for g in optimizer.param_groups:
            if not g.get("capturable", False):
                raise ValueError("GraphedStep needs an optimizer that steps on the device: torch.optim.Adam(..., capturable=True)")
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.max_norm, self.warmup, self.restore = max_norm, int(warmup), bool(restore)
        self._graph = self._static = self._loss = None

    def _eager(self, mixture, sources):
        self.optimizer.zero_grad(set_to_none=False)
        out = self.criterion(self.model(mixture), sources)
        loss = out[0] if isinstance(out, (tuple, list)) else out
        loss.backward()
        if self.max_norm is not None:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.max_norm)
        self.optimizer.step()
        return loss.detach()

    def capture(self, mixture, sources):
        # Recorded dropout diverges on this stack (GALRNet / SepFormer at their recipe sizes: loss inf after the first replays,
        # profiles/r04d_dual.txt -- the philox offset of the captured native_dropout does not advance as the eager one does): refuse
        # rather than train on it.
        live = [n for n, m in self.model.named_modules() if isinstance(m, torch.nn.modules.dropout._DropoutNd) and m.training and m.p > 0]
        if live:
            raise RuntimeError("GraphedStep: active dropout ({}{}) is not supported under hipGraph replay on this stack -- use the eager "
                               "step, model.eval(), or dropout 0".format(", ".join(live[:3]), " ..." if len(live) > 3 else ""))
        params = [p for p in self.model.parameters()]
        saved = [p.detach().clone() for p in params] if self.restore else None
        # optimizer state that exists ALREADY (continue_from / load_state_dict / earlier eager steps) is put back after the recording;
        # only state born during the warm-up starts from zero
        saved_state = {}
        if self.restore:
            for p, st in self.optimizer.state.items():
                saved_state[p] = {k: v.detach().clone() for k, v in st.items() if torch.is_tensor(v)}
        self._static = (mixture.clone(), sources.clone())
        for p in params:                                   # gradients must exist (and keep their addresses) before the recording
            if p.grad is None and p.requires_grad:
                p.grad = torch.zeros_like(p)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                self._eager(*self._static)
        torch.cuda.current_stream().wait_stream(side)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._loss = self._eager(*self._static)
        if self.restore:
            with torch.no_grad():
                for p, s in zip(params, saved):
                    p.copy_(s)
                for p, st in self.optimizer.state.items():
                    old = saved_state.get(p, {})
                    for k, v in st.items():
                        if torch.is_tensor(v):
                            if k in old:
                                v.copy_(old[k])            # in place: the graph holds the addresses
                            else:
                                v.zero_()                  # Adam-family state born in the warm-up starts at zero
        return self

    def __call__(self, mixture, sources):
        if self._graph is None:
            self.capture(mixture, sources)
        elif mixture.shape != self._static[0].shape or sources.shape != self._static[1].shape:
            raise ValueError("GraphedStep was recorded for batches of shape {} / {}".format(tuple(self._static[0].shape), tuple(self._static[1].shape)))
        self._static[0].copy_(mixture)
        self._static[1].copy_(sources)
        self._graph.replay()
        return self._loss
