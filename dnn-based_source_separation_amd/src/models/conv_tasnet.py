"""
Conv-TasNet on MI355X.

Drop-in for reference src/models/conv_tasnet.py:16-378: same constructor keywords, `forward` /
`extract_latent` / `get_config` / `get_package` / `build_model` / `build_from_pretrained` / `num_parameters`,
same module tree and state_dict keys, same exception types for bad configurations.
The arithmetic of the whole network is the fused kernel sequence of sepkernels/net.py behind one
torch.autograd.Function; parameters live as views of one flat fp32 buffer laid out so that the output- and
skip-head weights of every TCN layer are adjacent ([Wo;Ws] is then a single GEMM operand) and so that the
data-parallel gradient all-reduce is a single RCCL call on one buffer.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from utils.filterbank import choose_filterbank
from utils.tasnet import choose_layer_norm
from models.tdcn import TimeDilatedConvNet
from utils.checkpoint import load_checkpoint
from sepkernels import net as _net

SAMPLE_RATE_MUSDB18 = 44100
SAMPLE_RATE_LIBRISPEECH = 16000
EPS = 1e-12


class _FusedConvTasNetFn(torch.autograd.Function):
    """forward(mixture, cfg, names, want_latent, *params) -> est[, latent]; backward runs the hand-written
    backward kernel sequence and returns one gradient per parameter (views of a single flat buffer)."""

    @staticmethod
    def forward(ctx, mixture, cfg, names, want_latent, grad_sink, *params):
        ctx.set_materialize_grads(False)
        ctx.grad_sink, ctx.bucket_hook = (grad_sink if isinstance(grad_sink, tuple) else (grad_sink, None))
        # want_latent may arrive as (want_latent, grad mode of the caller): inside Function.forward the grad mode is always off and
        # needs_input_grad ignores torch.no_grad(), so the caller says whether anything will be differentiated
        want_latent, grad_on = want_latent if isinstance(want_latent, tuple) else (want_latent, True)
        need_bwd = grad_on and (ctx.needs_input_grad[0] or any(ctx.needs_input_grad[5:]))
        P = dict(zip(names, params))
        est, latent, sv = _net.forward(cfg, P, mixture, want_latent=want_latent, save=need_bwd)
        ctx.cfg, ctx.names, ctx.sv = cfg, names, sv
        ctx.save_for_backward(*params)
        if want_latent:
            ctx.mark_non_differentiable(latent)
            return est, latent
        return est

    @staticmethod
    def backward(ctx, d_est, *unused):
        params = ctx.saved_tensors
        names = ctx.names
        if d_est is None:
            return (None, None, None, None, None) + tuple(None for _ in params)
        # one flat gradient buffer with the same packing as ConvTasNet._flatten_parameters
        offs, total = _layout([(n, p.numel()) for n, p in zip(names, params)])
        placed = ctx.grad_sink if isinstance(ctx.grad_sink, dict) else None      # derived-basis models: name -> where that gradient goes
        flat = None if placed is not None else ctx.grad_sink
        if flat is None or flat.numel() != total:
            flat = torch.empty(total, device=d_est.device, dtype=d_est.dtype)
        G = {n: flat[offs[n]:offs[n] + p.numel()].view(p.shape) for n, p in zip(names, params)}
        if placed is not None:
            G.update(placed)
        hook = ctx.bucket_hook if flat is ctx.grad_sink else None
        on_ready = None
        if hook is not None:
            # gradient buckets for the data-parallel step: [block R-1 .. end of buffer], then one TCN block at a time;
            # what is left (block 0 and the head) is the caller's last bucket after backward returns
            R = ctx.cfg["sep_num_blocks"]
            starts = {r: offs["separator.tdcn.net.{}.net.0.bottleneck_conv1d.weight".format(r)] for r in range(R)}

            def on_ready(r):
                hook(starts[r], total if r == R - 1 else starts[r + 1])
        d_mix = _net.backward(ctx.cfg, dict(zip(names, params)), ctx.sv, d_est, G, on_ready=on_ready, want_dmix=ctx.needs_input_grad[0])
        if hook is not None:
            hook(0, starts[1] if ctx.cfg["sep_num_blocks"] > 1 else total)
        ctx.sv = None
        grads = tuple(None if (placed is not None and n in placed) else G[n] for n in names)      # placed: already where the caller wants them
        G = None
        return (d_mix, None, None, None, None) + grads


def _layout(named_sizes):
    """Offsets (in floats) of every parameter inside the flat buffer.  Every tensor starts 16-byte aligned; for each
    TCN layer the pair (output_pointwise.weight, skip_pointwise.weight) and the pair of their biases are adjacent."""
    sizes = dict(named_sizes)
    order, seen = [], set()
    for n, _ in named_sizes:
        if n in seen:
            continue
        if n.endswith("output_pointwise_conv1d.weight"):
            base = n[:-len("output_pointwise_conv1d.weight")]
            group = [base + "output_pointwise_conv1d.weight", base + "skip_pointwise_conv1d.weight",
                     base + "output_pointwise_conv1d.bias", base + "skip_pointwise_conv1d.bias"]
            group = [g for g in group if g in sizes]
        else:
            group = [n]
        for g in group:
            if g not in seen:
                order.append(g)
                seen.add(g)
    offs, cur = {}, 0
    for n in order:
        cur = (cur + 3) // 4 * 4
        offs[n] = cur
        cur += sizes[n]
    return offs, (cur + 3) // 4 * 4


class ConvTasNet(nn.Module):
    pretrained_model_ids = {
        "wsj0-mix": {8000: {2: {"enc_relu": "1yy-o7TyS1EcBWZ41rskMAVavtuEi4fMe"}, 3: {"enc_relu": "1-4Abl7LnEtwqMnAFQOcNLUOaDbgp3NoG"}},
                     16000: {2: "", 3: ""}},
        "wham/enhance-single": {8000: "1-6oiSK_CEE5Vl4OCy8TinA0cKsFFfGUg", 16000: ""},
        "wham/enhance-both": {8000: "1-GISUVcWjMeP3GLvojz9b0svw6gkmd2G", 16000: ""},
        "wham/separate-noisy": {8000: "1-0ckoPjaIiTJwv9Qotz6fkY2xeC77xdi", 16000: ""},
        "musdb18": {SAMPLE_RATE_MUSDB18: {"4sec_L20": "1A6dIofHZJQCUkyq-vxZ6KbPmEHLcf4WK", "8sec_L20": "1C4uv2z0w1s4rudIMaErLyEccNprJQWSZ",
                                          "8sec_L64": "1paXNGgH8m0kiJTQnn1WH-jEIurCKXwtw"}},
        "librispeech": {SAMPLE_RATE_LIBRISPEECH: {2: "1NI6Q_WZHiTKkgkNTEcZE1yHskHgYUHpy"}},
    }

    def __init__(self, n_basis, kernel_size, stride=None, enc_basis=None, dec_basis=None,
                 sep_hidden_channels=256, sep_bottleneck_channels=128, sep_skip_channels=128, sep_kernel_size=3,
                 sep_num_blocks=3, sep_num_layers=8, dilated=True, separable=True, sep_nonlinear="prelu", sep_norm=True,
                 mask_nonlinear="sigmoid", causal=True, n_sources=2, eps=EPS, **kwargs):
        super().__init__()
        if stride is None:
            stride = kernel_size // 2
        assert kernel_size % stride == 0, "kernel_size is expected divisible by stride"

        self.in_channels = kwargs.get("in_channels", 1)
        self.n_basis = n_basis
        self.kernel_size, self.stride = kernel_size, stride
        self.enc_basis, self.dec_basis = enc_basis, dec_basis
        if enc_basis == "trainable" and not dec_basis == "pinv":
            self.enc_nonlinear = kwargs["enc_nonlinear"]
        else:
            self.enc_nonlinear = None
        if enc_basis in ["Fourier", "trainableFourier", "trainableFourierTrainablePhase"] or \
                dec_basis in ["Fourier", "trainableFourier", "trainableFourierTrainablePhase"]:
            self.window_fn = kwargs["window_fn"]
            self.enc_onesided, self.enc_return_complex = kwargs["enc_onesided"], kwargs["enc_return_complex"]
        else:
            self.window_fn = None
            self.enc_onesided, self.enc_return_complex = None, None

        self.sep_hidden_channels, self.sep_bottleneck_channels, self.sep_skip_channels = \
            sep_hidden_channels, sep_bottleneck_channels, sep_skip_channels
        self.sep_kernel_size = sep_kernel_size
        self.sep_num_blocks, self.sep_num_layers = sep_num_blocks, sep_num_layers
        self.dilated, self.separable, self.causal = dilated, separable, causal
        self.sep_nonlinear, self.sep_norm = sep_nonlinear, sep_norm
        self.mask_nonlinear = mask_nonlinear
        self.n_sources = n_sources
        self.eps = eps

        # same construction order as the reference -> identical RNG consumption -> identical default weights per seed
        encoder, decoder = choose_filterbank(n_basis, kernel_size=kernel_size, stride=stride, enc_basis=enc_basis,
                                             dec_basis=dec_basis, **kwargs)
        self.encoder = encoder
        self.separator = Separator(n_basis, bottleneck_channels=sep_bottleneck_channels, hidden_channels=sep_hidden_channels,
                                   skip_channels=sep_skip_channels, kernel_size=sep_kernel_size, num_blocks=sep_num_blocks,
                                   num_layers=sep_num_layers, dilated=dilated, separable=separable, causal=causal,
                                   nonlinear=sep_nonlinear, norm=sep_norm, mask_nonlinear=mask_nonlinear, n_sources=n_sources, eps=eps)
        self.decoder = decoder

        self._flat = None
        self._names = None
        self._grad_sink = None      # optional flat buffer the backward pass writes into (sepkernels.train.FusedTrainStep)
        # Which way forward() runs is decided HERE, not at the first call: the fused HIP kernel sequence for the configuration
        # family it implements, else the module-by-module composition (unsupported option combinations
        # fall back to the PyTorch composition) -- `fused_reason` says why.
        try:
            _net.check_supported(self.get_config())
            self.fused, self.fused_reason = True, None
        except NotImplementedError as e:
            self.fused, self.fused_reason = False, str(e)
        # ... and between the two: the CAUSAL family (the reference constructor's default: cLN everywhere, all padding on the left,
        # tdcn.py:98,125-127,169) runs layer by layer on this library's kernels -- `staged`: every 1x1 convolution on the MFMA GEMM,
        # PReLU + cLN in sep_cln_*, the dilated depthwise taps in sep_depthwise_*, encoder / mask / decoder as in the fused path -- on
        # (B, C, ldt) workspace rows throughout.  Not one fused sequence (each norm is a pass of its own), no torch convolution either.
        self.staged, self.staged_reason = self._staged_supported()
        # ... and the LINEAR filterbanks other than the learned pair -- Fourier bases (fixed / trainable frequencies / trainable phase) with a
        # real-valued two-sided latent, the pseudo-inverse decoder: the reference recipe's settings for its published "Fourier / Fourier" and
        # "trainable / pseudo-inverse" rows (egs/wsj0-mix/conv-tasnet/train.sh:21-26, README rows 3-4) -- are a convolution with SOME analysis
        # matrix and a transposed convolution with SOME synthesis matrix: the fused kernel sequence runs them unchanged on bases that torch
        # forms from the filterbank's parameters each pass (a few hundred kilobytes), and autograd carries the kernels' basis gradients back
        # into frequency / phase / window / encoder weight.  `fused_derived`.
        self.fused_derived = (not self.fused) and self._derived_supported()
        self._flatten_parameters()

    # ------------------------------------------------------------------ parameter storage
    def _flatten_parameters(self):
        """Re-home every parameter as a view of one flat buffer (see _layout).  Idempotent; called after
        construction and after every .to()/.cuda()/.float() (nn.Module._apply)."""
        named = [(n, p) for n, p in self.named_parameters() if p.is_floating_point()]      # (a Fourier basis keeps an integer `time_seq`: not part of the buffer)
        if not named:
            return
        dev, dt = named[0][1].device, named[0][1].dtype
        if any(p.device != dev or p.dtype != dt for _, p in named):
            self._flat = None
            return
        offs, total = _layout([(n, p.numel()) for n, p in named])
        flat = torch.zeros(total, device=dev, dtype=dt)
        with torch.no_grad():
            for n, p in named:
                v = flat[offs[n]:offs[n] + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
        self._flat = flat
        self._names = [n for n, _ in named]
        self._offsets = offs

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._flatten_parameters()
        return out

    def flat_parameters(self):
        """The flat fp32 buffer all parameters are views of (None if they are not co-located)."""
        if self._flat is None:
            return None
        for n, p in self.named_parameters():   # cheap sanity: views still intact?
            if p.is_floating_point() and p.data_ptr() != self._flat.data_ptr() + self._flat.element_size() * self._offsets[n]:
                self._flatten_parameters()
                break
        return self._flat

    # ------------------------------------------------------------------ forward
    def forward(self, input):
        output, _ = self._run(input, want_latent=False)
        return output

    def extract_latent(self, input):
        """
        Args:
            input (batch_size, 1, T) or (batch_size, 1, n_mics, T)
        Returns:
            output (batch_size, n_sources, T) or (batch_size, n_sources, n_mics, T)
            latent (batch_size, n_sources, n_basis, T'), where T' = (T-K)//S+1
        """
        return self._run(input, want_latent=True)

    def _run(self, input, want_latent):
        n_dims = input.dim()
        if n_dims == 3:
            batch_size, C_in, T = input.size()
            assert C_in == 1, "input.size() is expected (?, 1, ?), but given {}".format(input.size())
            mixture = input
        elif n_dims == 4:
            batch_size, C_in, n_mics, T = input.size()
            assert C_in == 1, "input.size() is expected (?, 1, ?, ?), but given {}".format(input.size())
            mixture = input.view(batch_size, n_mics, T)
        else:
            raise ValueError("Not support {} dimension input".format(n_dims))
        cfg = self.get_config()
        if mixture.size(1) != self.in_channels:
            raise ValueError("input has {} channels, the model was built with in_channels={}".format(mixture.size(1), self.in_channels))
        if self.fused_derived and ((mixture.is_cuda and mixture.dtype == torch.float32) or _net.backend().name != "hip"):
            est, latent = self._run_fused_derived(mixture.contiguous(), want_latent)
            if n_dims == 3:
                est = est.view(batch_size, self.n_sources, T)
            return est, latent
        if not self.fused and self.staged and ((mixture.is_cuda and mixture.dtype == torch.float32) or _net.backend().name != "hip"):
            est, latent = self._run_staged(mixture.contiguous(), want_latent)
            if n_dims == 3:
                est = est.view(batch_size, self.n_sources, T)
            return est, latent
        if not self.fused:
            est, latent = self._run_composed(mixture.contiguous(), want_latent)
            if n_dims == 3:
                est = est.view(batch_size, self.n_sources, T)
            return est, latent
        if not mixture.is_cuda and _net.backend().name == "hip":
            # A model and an input the CALLER left on the CPU (`local/test.py --use_cuda 0`, demo.py; reference
            # egs/wsj0-mix/conv-tasnet/local/test.py:25,41-43): the module-by-module composition on ATen, i.e. the reference's own
            # arithmetic -- not this library, never chosen for a tensor on the GPU and never because the HIP library is missing (a
            # model on 'cuda' without it fails in sepkernels.load()).  SEPK_STRICT_DEVICE=1 turns the case back into the loud error.
            if os.environ.get("SEPK_STRICT_DEVICE", "0") == "1":
                raise RuntimeError("ConvTasNet (MI355X build) runs on the GPU only: move the model and the input to 'cuda' "
                                   "(SEPK_STRICT_DEVICE=1 forbids the ATen composition for CPU tensors).")
            est, latent = self._run_composed(mixture.contiguous(), want_latent)
            if n_dims == 3:
                est = est.view(batch_size, self.n_sources, T)
            return est, latent
        mixture = mixture.contiguous()
        if mixture.dtype != torch.float32 and _net.backend().name == "hip":
            mixture = mixture.float()
        named = self._named_tensors()
        names = tuple(n for n, _ in named)
        params = tuple(p for _, p in named)
        sink = getattr(self, "_grad_sink", None)
        hook = getattr(self, "_grad_bucket_hook", None)
        out = _FusedConvTasNetFn.apply(mixture, cfg, names, (want_latent, torch.is_grad_enabled()), (sink, hook) if hook is not None else sink, *params)
        if want_latent:
            est, latent = out
            F = _net.Geometry(T, self.kernel_size, self.stride).F
            latent = latent[..., :F]
        else:
            est, latent = out, None
        if n_dims == 3:
            est = est.view(batch_size, self.n_sources, T)
        return est, latent

    def _named_tensors(self):
        """(name, tensor) of every parameter, also inside an nn.DataParallel replica: torch's replicate() empties `_parameters`
        of the replica modules and hands them the broadcast copies as plain attributes / `_former_parameters` (they carry the
        autograd edge back to the master parameters), so the walk goes through those.  The copies are separate tensors, not
        views of one flat buffer: net.py then runs the two heads of a layer as two products."""
        named = list(self.named_parameters())
        if named:
            return named
        out = []
        for mname, mod in self.named_modules():
            for k, v in getattr(mod, "_former_parameters", {}).items():
                if v is not None:
                    out.append(((mname + "." if mname else "") + k, v))
        if not out:
            raise RuntimeError("ConvTasNet has no parameters to run with (a module replica without `_former_parameters`?); for "
                               "multi-GPU training use one process per GPU with sepkernels.train.FusedTrainStep")
        return out

    _FOURIER_BASES = ("Fourier", "trainableFourier", "trainableFourierTrainablePhase")

    def _derived_cfg(self):
        return dict(self.get_config(), enc_basis="trainable", dec_basis="trainable", enc_nonlinear=None)

    def _derived_supported(self):
        enc, dec = self.enc_basis, self.dec_basis
        fourier = self._FOURIER_BASES
        if enc == "trainable" and dec == "trainable":
            return False                                         # the fused family proper (or outside it for another reason)
        if enc not in ("trainable",) + fourier or dec not in ("trainable", "pinv") + fourier:
            return False                                         # gated encoder: not linear
        if (enc in fourier or dec in fourier) and (self.enc_onesided or self.enc_return_complex):
            return False                                         # complex latent (magnitude masked, phase kept) / one-sided (N + 2 rows: not a multiple of 16)
        if enc == "trainable" and self.enc_nonlinear:
            return False
        try:
            _net.check_supported(self._derived_cfg())
        except NotImplementedError:
            return False
        return True

    def _derived_bases(self):
        """(E, D): analysis basis (N, in_channels, L) and synthesis basis (N, in_channels, L) of this model's filterbank, differentiable in
        its parameters.  Row order = the reference's channel order: Fourier [real parts of all bins | imaginary parts], reference
        src/models/filterbank.py:60-113 (encoder), :160-203 (decoder), :253-323 (pseudo-inverse)."""
        from models.filterbank import FourierEncoder, FourierDecoder, PinvDecoder, _all_bins
        enc, dec = self.encoder, self.decoder
        if isinstance(enc, FourierEncoder):
            E = enc.get_basis().unsqueeze(1)                     # [window cos ; -window sin] over all bins
        else:
            E = enc.conv1d.weight

        def synthesis(mod, window):                              # est = convT(xr, br) - convT(xi, bi): one transposed convolution with [br ; -bi]
            ang = mod._angles()
            n = mod.n_basis
            br, bi = _all_bins(torch.cos(ang), n, 0, 1.0), _all_bins(torch.sin(ang), n, 0, -1.0)
            return torch.cat([window * br / n, -(window * bi / n)], 0).unsqueeze(1)
        if isinstance(dec, FourierDecoder):
            D = synthesis(dec, dec.optimal_window)
        elif isinstance(dec, PinvDecoder):
            D = synthesis(enc, enc.window) if isinstance(enc, FourierEncoder) else dec.get_basis()
        else:
            D = dec.conv_transpose1d.weight
        return E.contiguous(), D.contiguous()

    def _run_fused_derived(self, mixture, want_latent):
        cfg = self._derived_cfg()
        E, D = self._derived_bases()
        sep = [(n, p) for n, p in self._named_tensors() if n.startswith("separator.")]
        names = ("encoder.conv1d.weight",) + tuple(n for n, _ in sep) + ("decoder.conv_transpose1d.weight",)
        # inside FusedTrainStep the separator's gradients go straight to their places in the step's flat buffer (no per-tensor .grad copies);
        # only the two bases' gradients travel through autograd into the filterbank's few parameters
        sink, placed = getattr(self, "_grad_sink", None), None
        if sink is not None and self._flat is not None and sink.numel() == self._flat.numel():
            placed = {n: sink[self._offsets[n]:self._offsets[n] + p.numel()].view(p.shape) for n, p in sep}
            self._sink_placed = True
        out = _FusedConvTasNetFn.apply(mixture.float() if _net.backend().name == "hip" else mixture, cfg, names, (want_latent, torch.is_grad_enabled()), placed, E,
                                       *[p for _, p in sep], D)
        if want_latent:
            est, latent = out
            return est, latent[..., :_net.Geometry(mixture.shape[-1], self.kernel_size, self.stride).F]
        return out, None

    def _staged_supported(self):
        cfg = self.get_config()
        problems = []
        if self.fused:
            return False, "the fused sequence takes this configuration"
        if not cfg.get("causal"):
            problems.append("causal=False outside the fused family")
        if cfg.get("enc_basis") != "trainable" or cfg.get("dec_basis") != "trainable":
            problems.append("enc_basis/dec_basis must be 'trainable'")
        if cfg.get("enc_nonlinear") not in (None, "", "relu"):
            problems.append("enc_nonlinear must be None or 'relu'")
        if not cfg.get("separable", True) or not cfg.get("dilated", True):
            problems.append("separable=True and dilated=True are required")
        if cfg.get("sep_nonlinear") != "prelu" or not cfg.get("sep_norm", True):
            problems.append("sep_nonlinear='prelu' and sep_norm=True are required")
        for k in ("n_basis", "sep_hidden_channels", "sep_bottleneck_channels", "sep_skip_channels"):
            if cfg[k] % 16:
                problems.append("{} must be a multiple of 16".format(k))
        if cfg["kernel_size"] % cfg["stride"]:
            problems.append("kernel_size must be divisible by stride")
        return (not problems), ("; ".join(problems) or None)

    def _run_staged(self, mixture, want_latent):
        """The causal Conv-TasNet of the reference (conv_tasnet.py:121-171 with tdcn.py:107-147, 177-196 and norm.py:58-101) as a sequence of
        this library's kernels on (B, C, ldt) rows, autograd through sepkernels.functional:
            encoder (+ReLU)                  EncodeFn                         sep_encoder_fwd
            cLN, 1x1 bottleneck              PaddedCLNFn, PaddedPointwiseFn   sep_cln_*, sep_pw_gemm
            per layer  1x1 -> PReLU+cLN -> depthwise (left padding (P-1) d) -> PReLU+cLN -> [1x1 output + residual ; 1x1 skip, summed]
                                             PaddedPointwiseFn, PaddedCLNFn, PaddedDepthwiseFn, PaddedHeadsFn
            PReLU + 1x1 mask, sigmoid | softmax, mask * w -> decoder -> crop     PaddedPointwiseFn, torch elementwise, MaskDecodeFn"""
        from sepkernels.functional import EncodeFn, PaddedPointwiseFn, PaddedCLNFn, PaddedDepthwiseFn, PaddedHeadsFn, MaskDecodeFn
        B, Cin, T = mixture.shape
        sep = self.separator
        geo = _net.Geometry(T, self.kernel_size, self.stride)
        F_ = geo.F
        wa = _net._weights_amax(dict(self.named_parameters()))      # ONE operand bound for the ~100 products of the pass (None off the f16x3 arithmetic)
        w = EncodeFn.apply(mixture, self.encoder.conv1d.weight, self.stride, self.enc_nonlinear == "relu")
        x = PaddedCLNFn.apply(w, F_, None, sep.norm1d.gamma, sep.norm1d.beta, sep.norm1d.eps)
        x = PaddedPointwiseFn.apply(x, F_, sep.bottleneck_conv1d.weight, sep.bottleneck_conv1d.bias, None, wa)
        total = None
        for block in sep.tdcn.net:
            for layer in block.net:
                dw = layer.separable_conv1d
                d, P = layer.dilation, layer.kernel_size
                a = PaddedPointwiseFn.apply(x, F_, layer.bottleneck_conv1d.weight, layer.bottleneck_conv1d.bias, None, wa)
                v1 = PaddedCLNFn.apply(a, F_, layer.nonlinear1d.weight, layer.norm1d.gamma, layer.norm1d.beta, layer.norm1d.eps)
                z = PaddedDepthwiseFn.apply(v1, F_, dw.depthwise_conv1d.weight, dw.depthwise_conv1d.bias, d, (P - 1) * d)
                v2 = PaddedCLNFn.apply(z, F_, dw.nonlinear1d.weight, dw.norm1d.gamma, dw.norm1d.beta, dw.norm1d.eps)
                out = dw.output_pointwise_conv1d if dw.dual_head else None
                x, total = PaddedHeadsFn.apply(v2, F_, out.weight if out is not None else None, out.bias if out is not None else None,
                                               dw.skip_pointwise_conv1d.weight, dw.skip_pointwise_conv1d.bias, x, total, wa)
        m = PaddedPointwiseFn.apply(total, F_, sep.mask_conv1d.weight, sep.mask_conv1d.bias, sep.prelu.weight, wa)
        m = torch.sigmoid(m) if self.mask_nonlinear == "sigmoid" else torch.softmax(m, dim=1)
        out = MaskDecodeFn.apply(w, m, self.decoder.conv_transpose1d.weight, self.stride, T, want_latent)
        if want_latent:
            est, latent = out
            return est, latent[..., :F_]
        return out, None

    def _run_composed(self, mixture, want_latent):
        """The reference's own sequence (conv_tasnet.py:121-171) on this repository's modules, for configurations outside
        the fused family: pad -> encoder -> separator -> mask * w -> decoder (transposed convolution = overlap-add) -> crop.
        Torch convolutions on the device of the input; gLN through sep_gln_*, cLN as its prefix-sum composition."""
        B, _, T = mixture.shape
        L, S, n_src, N = self.kernel_size, self.stride, self.n_sources, self.n_basis
        padding = (S - (T - L) % S) % S
        left = padding // 2
        x = F.pad(mixture, (left, padding - left))
        w = self.encoder(x)
        if torch.is_complex(w):                                    # Fourier basis with complex output: mask the magnitude, keep the phase
            mag, phase = torch.abs(w), torch.angle(w)
            latent = mag.unsqueeze(1) * self.separator(mag) * torch.exp(1j * phase.unsqueeze(1))
        else:
            latent = w.unsqueeze(1) * self.separator(w)            # (B, n_src, N, F)
        y = self.decoder(latent.reshape(B * n_src, N, -1))
        y = y.view(B, n_src, self.in_channels, -1)[..., left:left + T]
        return y, (latent if want_latent else None)

    # ------------------------------------------------------------------ config / checkpoints
    def get_config(self):
        return {
            "in_channels": self.in_channels, "n_basis": self.n_basis, "kernel_size": self.kernel_size, "stride": self.stride,
            "enc_basis": self.enc_basis, "dec_basis": self.dec_basis, "enc_nonlinear": self.enc_nonlinear,
            "window_fn": self.window_fn, "enc_onesided": self.enc_onesided, "enc_return_complex": self.enc_return_complex,
            "sep_hidden_channels": self.sep_hidden_channels, "sep_bottleneck_channels": self.sep_bottleneck_channels,
            "sep_skip_channels": self.sep_skip_channels, "sep_kernel_size": self.sep_kernel_size,
            "sep_num_blocks": self.sep_num_blocks, "sep_num_layers": self.sep_num_layers,
            "dilated": self.dilated, "separable": self.separable, "causal": self.causal,
            "sep_nonlinear": self.sep_nonlinear, "sep_norm": self.sep_norm, "mask_nonlinear": self.mask_nonlinear,
            "n_sources": self.n_sources, "eps": self.eps,
        }

    def get_package(self):
        return self.get_config()

    @classmethod
    def build_model(cls, model_path, load_state_dict=False, trust_pickle=None):
        """trust_pickle: see utils.checkpoint.load_checkpoint (None = the SEPK_TRUST_CHECKPOINTS switch)"""
        config = load_checkpoint(model_path, trust_pickle)
        return cls._from_config(config, load_state_dict)

    @classmethod
    def _from_config(cls, config, load_state_dict=False):
        model = cls(
            config.get("n_bases") or config["n_basis"], in_channels=config.get("in_channels") or 1,
            kernel_size=config["kernel_size"], stride=config["stride"],
            enc_basis=config.get("enc_bases") or config["enc_basis"], dec_basis=config.get("dec_bases") or config["dec_basis"],
            enc_nonlinear=config["enc_nonlinear"], window_fn=config["window_fn"],
            enc_onesided=config.get("enc_onesided") or None, enc_return_complex=config.get("enc_return_complex") or None,
            sep_hidden_channels=config["sep_hidden_channels"], sep_bottleneck_channels=config["sep_bottleneck_channels"],
            sep_skip_channels=config["sep_skip_channels"], sep_kernel_size=config["sep_kernel_size"],
            sep_num_blocks=config["sep_num_blocks"], sep_num_layers=config["sep_num_layers"],
            dilated=config["dilated"], separable=config["separable"], causal=config["causal"],
            sep_nonlinear=config["sep_nonlinear"], sep_norm=config["sep_norm"], mask_nonlinear=config["mask_nonlinear"],
            n_sources=config["n_sources"], eps=config["eps"])
        if load_state_dict:
            model.load_state_dict(config["state_dict"])
        return model

    @classmethod
    def build_from_pretrained(cls, root="./pretrained", quiet=False, load_state_dict=True, **kwargs):
        """Same task table and directory convention as the reference (conv_tasnet.py:238-310).  The download itself
        needs the reference's gdown helper (utils.utils); an already-downloaded checkpoint is loaded directly."""
        task = kwargs.get("task")
        trust_pickle = kwargs.get("trust_pickle")          # utils.checkpoint.load_checkpoint; downloads are NOT trusted by default
        if task not in cls.pretrained_model_ids:
            raise KeyError("Invalid task ({}) is specified.".format(task))
        ids = cls.pretrained_model_ids[task]
        extra = {}
        model_choice = kwargs.get("model_choice") or "best"
        if task in ["wsj0-mix", "wsj0"]:
            sample_rate, n_sources = kwargs.get("sample_rate") or 8000, kwargs.get("n_sources") or 2
            config = kwargs.get("config") or "enc_relu"
            model_id = ids[sample_rate][n_sources][config]
            download_dir = os.path.join(root, cls.__name__, task, "sr{}/{}speakers/{}".format(sample_rate, n_sources, config))
            extra["n_sources"] = n_sources
        elif task == "musdb18":
            sample_rate = kwargs.get("sample_rate") or SAMPLE_RATE_MUSDB18
            config = kwargs.get("config") or "4sec_L20"
            model_id = ids[sample_rate][config]
            download_dir = os.path.join(root, cls.__name__, task, "sr{}".format(sample_rate), config)
        elif task in ["wham/separate-noisy", "wham/enhance-single", "wham/enhance-both"]:
            sample_rate = kwargs.get("sample_rate") or 8000
            model_id = ids[sample_rate]
            download_dir = os.path.join(root, cls.__name__, task, "sr{}".format(sample_rate))
        elif task == "librispeech":
            sample_rate, n_sources = kwargs.get("sample_rate") or SAMPLE_RATE_LIBRISPEECH, kwargs.get("n_sources") or 2
            model_id = ids[sample_rate][n_sources]
            download_dir = os.path.join(root, cls.__name__, task, "sr{}/{}speakers".format(sample_rate, n_sources))
            extra["n_sources"] = n_sources
        else:
            raise NotImplementedError("Not support task={}.".format(task))
        extra["sample_rate"] = sample_rate
        model_path = os.path.join(download_dir, "model", "{}.pth".format(model_choice))
        if not os.path.exists(model_path):
            from utils.utils import download_pretrained_model_from_google_drive   # reference helper (gdown), reused as-is
            download_pretrained_model_from_google_drive(model_id, download_dir, quiet=quiet)
        config = load_checkpoint(model_path, trust_pickle)
        model = cls._from_config(config, load_state_dict=load_state_dict)
        if task == "musdb18":
            extra.update({"sources": config["sources"], "n_sources": len(config["sources"])})
        for key, value in extra.items():
            setattr(model, key, value)
        return model

    @property
    def num_parameters(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)


class Separator(nn.Module):
    """norm -> 1x1 bottleneck -> TCN -> PReLU -> 1x1 mask conv -> sigmoid | softmax (reference conv_tasnet.py:322-378).
    Inside a fused ConvTasNet this is a parameter container; `forward` is the stand-alone / fallback composition."""

    def __init__(self, num_features, bottleneck_channels=128, hidden_channels=256, skip_channels=128, kernel_size=3,
                 num_blocks=3, num_layers=8, dilated=True, separable=True, causal=True, nonlinear="prelu", norm=True,
                 mask_nonlinear="sigmoid", n_sources=2, eps=EPS):
        super().__init__()
        self.num_features, self.n_sources = num_features, n_sources
        norm_name = "cLN" if causal else "gLN"
        self.norm1d = choose_layer_norm(norm_name, num_features, causal=causal, eps=eps)
        self.bottleneck_conv1d = nn.Conv1d(num_features, bottleneck_channels, kernel_size=1, stride=1)
        self.tdcn = TimeDilatedConvNet(bottleneck_channels, hidden_channels=hidden_channels, skip_channels=skip_channels,
                                       kernel_size=kernel_size, num_blocks=num_blocks, num_layers=num_layers, dilated=dilated,
                                       separable=separable, causal=causal, nonlinear=nonlinear, norm=norm)
        self.prelu = nn.PReLU()
        self.mask_conv1d = nn.Conv1d(skip_channels, n_sources * num_features, kernel_size=1, stride=1)
        if mask_nonlinear not in ("sigmoid", "softmax"):
            raise ValueError("Cannot support {}".format(mask_nonlinear))
        self.mask_nonlinear = nn.Sigmoid() if mask_nonlinear == "sigmoid" else nn.Softmax(dim=1)

    def forward(self, input):
        """input (batch_size, num_features, T') -> mask (batch_size, n_sources, num_features, T')"""
        x = self.tdcn(self.bottleneck_conv1d(self.norm1d(input)))
        x = self.mask_nonlinear(self.mask_conv1d(self.prelu(x)))
        return x.view(x.size(0), self.n_sources, self.num_features, x.size(-1))


from sepkernels.shadowed import fall_through as _fall_through      # names of the reference's same-named module this tree does not define

__getattr__ = _fall_through(__name__, __file__)
