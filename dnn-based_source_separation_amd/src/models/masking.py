"""
What DPTNet, GALRNet and SepFormer share on MI355X (reference src/models/dptnet.py:15-330, galrnet.py:13-165,
sepformer.py:16-280 repeat it three times): the shell -- filterbank choice, input padding, `w * mask`, decoder, crop,
config / checkpoint plumbing -- and the gated end of their separators -- PReLU, 1x1 `map`, gated tanh unit, optional
1x1 output convolution, mask activation.

Two executions of the same arithmetic, chosen per call:

* kernel path (learned real bases, channel counts in multiples of 16, fp32 tensors the backend takes): analysis basis,
  every 1x1 convolution, chunking / overlap-add and `w * mask` + synthesis basis run on libsepkernels through the
  autograd Functions of sepkernels/functional.py, on (B, C, ldt) rows that keep the workspace stride from the encoder to
  the decoder (no repacking in between); the two gate convolutions of the GTU are ONE product over the stacked weights;
* composition (everything else: Fourier / pseudo-inverse bases, odd channel counts, float64): the module-by-module
  torch composition, SURVEY.md section 8b's fallback.

The dual-path cores in between (attention, feed-forward, recurrences on (B, C, S, K)) are the same modules either way.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from sepkernels import net as _net
from sepkernels.functional import EncodeFn, MaskDecodeFn, PaddedPointwiseFn, segment_geometry, takes as _takes
from utils.filterbank import choose_filterbank
from utils.model import choose_nonlinear
from utils.checkpoint import load_checkpoint
from models.filterbank import Decoder, Encoder
from models.gtu import GTU1d       # noqa: F401  (the separators build theirs through this module)

EPS = 1e-12

FOURIER_BASES = ("Fourier", "trainableFourier", "trainableFourierTrainablePhase")


def make_mask_nonlinear(name):
    if name in ("relu", "sigmoid"):
        return choose_nonlinear(name)
    if name == "softmax":
        return choose_nonlinear(name, dim=1)
    raise ValueError("Cannot support {}".format(name))



class GatedMaskSeparator(nn.Module):
    """The part of a separator after overlap-add: core (B, C, n_frames) -> mask (B, n_sources, num_features, n_frames).
    Sub-classes create `prelu`, `map`, `gtu`, `mask_nonlinear` (and `bottleneck_conv1d_out` for SepFormer) under the
    reference's names and call `_mask` / `_mask_padded`."""

    def _chunk_padding(self, n_frames):
        pad_left, pad_right, _ = segment_geometry(n_frames, self.chunk_size, self.hop_size)
        return pad_left, pad_right

    def _mask(self, core, batch_size, n_frames):
        x = self.map(self.prelu(core)).view(batch_size * self.n_sources, self.num_features, n_frames)
        x = self.gtu(x)
        out = getattr(self, "bottleneck_conv1d_out", None)
        if out is not None:
            x = out(x)
        return self.mask_nonlinear(x).view(batch_size, self.n_sources, self.num_features, n_frames)

    def _mask_padded(self, core, n_frames):
        """core (B, C, ldt) -> activated mask (B, n_sources*num_features, ldt)"""
        B, _, ldt = core.shape
        N = self.num_features
        x = PaddedPointwiseFn.apply(core, n_frames, self.map.weight, self.map.bias, self.prelu.weight)
        Wg, bg = self.gtu.stacked()
        ab = PaddedPointwiseFn.apply(x.view(B * self.n_sources, N, ldt), n_frames, Wg, bg, None)
        x = torch.tanh(ab[:, :N]) * torch.sigmoid(ab[:, N:])
        out = getattr(self, "bottleneck_conv1d_out", None)
        if out is not None:
            x = PaddedPointwiseFn.apply(x, n_frames, out.weight, out.bias, None)
        return self.mask_nonlinear(x).view(B, self.n_sources * N, ldt)       # softmax: dim 1 of (B*n_sources, N, .) as in the reference

    def padded_problems(self):
        """why `mask_padded` cannot run (empty: it can)"""
        problems = []
        widths = {"num_features": self.num_features, "n_sources*num_features": self.n_sources * self.num_features}
        if getattr(self, "bottleneck_channels", None) is not None:
            widths["bottleneck_channels"] = self.bottleneck_channels
        for k, v in widths.items():
            if v % 16:
                problems.append("{} must be a multiple of 16".format(k))
        if self.prelu.weight.numel() != 1:
            problems.append("one PReLU slope expected")
        return problems


class MaskingTasNet(nn.Module):
    """encoder -> separator (mask) -> decoder.  Sub-classes list their separator-side constructor arguments in
    SEP_KEYS (stored as attributes of the same name, written to the config in that order) and build `self.separator`."""
    SEP_KEYS = ()
    pretrained_model_ids = {}
    CONFIG_HAS_IN_CHANNELS = False
    MULTICHANNEL_INPUT = False

    def _init_filterbank(self, n_basis, kernel_size, stride, enc_basis, dec_basis, kwargs):
        if stride is None:
            stride = kernel_size // 2
        assert kernel_size % stride == 0, "kernel_size is expected divisible by stride"
        self.in_channels = kwargs.get("in_channels", 1)
        self.n_basis = n_basis
        self.kernel_size, self.stride = kernel_size, stride
        self.enc_basis, self.dec_basis = enc_basis, dec_basis
        self.enc_nonlinear = kwargs["enc_nonlinear"] if (enc_basis == "trainable" and not dec_basis == "pinv") else None
        if enc_basis in FOURIER_BASES or dec_basis in FOURIER_BASES:
            self.window_fn = kwargs["window_fn"]
            self.enc_onesided, self.enc_return_complex = kwargs["enc_onesided"], kwargs["enc_return_complex"]
        else:
            self.window_fn, self.enc_onesided, self.enc_return_complex = None, None, None
        self.encoder, self.decoder = choose_filterbank(n_basis, kernel_size=kernel_size, stride=stride, enc_basis=enc_basis,
                                                       dec_basis=dec_basis, **kwargs)
        return stride

    # ------------------------------------------------------------------------------------------------ execution
    def kernel_path_problems(self):
        problems = []
        if type(self.encoder) is not Encoder or type(self.decoder) is not Decoder:
            problems.append("enc_basis/dec_basis must be 'trainable'")
        elif self.encoder.nonlinear and not isinstance(getattr(self.encoder, "nonlinear1d", None), nn.ReLU):
            problems.append("enc_nonlinear must be None or 'relu'")
        return problems + self.separator.padded_problems()

    def forward(self, input):
        output, _ = self.extract_latent(input)
        return output

    def extract_latent(self, input):
        """input (batch_size, 1, T) [or (batch_size, 1, n_mics, T)] -> output (batch_size, n_sources, T) [(.., n_mics, T)],
        latent (batch_size, n_sources, n_basis, T')"""
        n_dims = input.dim()
        if n_dims == 3:
            batch_size, C_in, T = input.size()
            assert C_in == 1, "input.size() is expected (?, 1, ?), but given {}".format(input.size())
            mixture = input
        elif n_dims == 4 and self.MULTICHANNEL_INPUT:
            batch_size, C_in, n_mics, T = input.size()
            assert C_in == 1, "input.size() is expected (?, 1, ?, ?), but given {}".format(input.size())
            mixture = input.view(batch_size, n_mics, T)
        else:
            raise ValueError("Not support {} dimension input".format(n_dims))
        # the kernel path's EncodeFn / HeadFn do not differentiate with respect to the mixture itself: such calls take the composition
        wants_input_grad = torch.is_grad_enabled() and mixture.requires_grad
        if _takes(mixture) and not wants_input_grad and not self.kernel_path_problems():
            est, latent = self._run_kernels(mixture.contiguous())
        else:
            est, latent = self._run_composed(mixture)
        if n_dims == 3:
            est = est.view(batch_size, self.n_sources, T)
        return est, latent

    def _enter(self, mixture):
        """-> (w (B, N, ldt), what the separator's `mask_padded` takes)"""
        w = EncodeFn.apply(mixture, self.encoder.conv1d.weight, self.stride, self.encoder.nonlinear)
        return w, w

    def _run_kernels(self, mixture):
        T = mixture.shape[-1]
        geo = _net.Geometry(T, self.kernel_size, self.stride)
        w, entry = self._enter(mixture)
        mask = self.separator.mask_padded(entry, geo.F)
        est, latent = MaskDecodeFn.apply(w, mask, self.decoder.conv_transpose1d.weight, self.stride, T, True)
        return est, latent[..., :geo.F]

    def _run_composed(self, mixture):
        batch_size, n_mics, T = mixture.shape
        padding = (self.stride - (T - self.kernel_size) % self.stride) % self.stride
        pad_left = padding // 2
        w = self.encoder(F.pad(mixture, (pad_left, padding - pad_left)))
        if torch.is_complex(w):
            amplitude, phase = torch.abs(w), torch.angle(w)
            w_hat = amplitude.unsqueeze(1) * self.separator(amplitude) * torch.exp(1j * phase.unsqueeze(1))
        else:
            w_hat = w.unsqueeze(1) * self.separator(w)
        x_hat = self.decoder(w_hat.reshape(batch_size * self.n_sources, self.n_basis, -1))
        x_hat = x_hat.view(batch_size, self.n_sources, n_mics, -1)
        return x_hat[..., pad_left:pad_left + T], w_hat

    # ------------------------------------------------------------------------------------------------ configuration
    def get_config(self):
        config = {"in_channels": self.in_channels} if self.CONFIG_HAS_IN_CHANNELS else {}
        for k in ("n_basis", "kernel_size", "stride", "enc_basis", "dec_basis", "enc_nonlinear", "window_fn", "enc_onesided",
                  "enc_return_complex") + tuple(self.SEP_KEYS) + ("mask_nonlinear", "causal", "n_sources", "eps"):
            config[k] = getattr(self, k)
        return config

    def get_package(self):
        return self.get_config()

    @classmethod
    def build_model(cls, model_path, load_state_dict=False, trust_pickle=None):
        config = load_checkpoint(model_path, trust_pickle)
        legacy = {"n_basis": "n_bases", "enc_basis": "enc_bases", "dec_basis": "dec_bases"}       # keys of older checkpoints
        args = {}
        for k in ("n_basis", "kernel_size", "stride", "enc_basis", "dec_basis", "enc_nonlinear", "window_fn") + tuple(cls.SEP_KEYS) + \
                ("mask_nonlinear", "causal", "n_sources", "eps"):
            args[k] = (config.get(legacy[k]) or config[k]) if k in legacy else config[k]
        args["enc_onesided"] = config.get("enc_onesided") or None
        args["enc_return_complex"] = config.get("enc_return_complex") or None
        if cls.CONFIG_HAS_IN_CHANNELS:
            args["in_channels"] = config.get("in_channels") or 1
        model = cls(args.pop("n_basis"), args.pop("kernel_size"), **args)
        if load_state_dict:
            model.load_state_dict(config["state_dict"])
        return model

    @classmethod
    def build_from_pretrained(cls, root="./pretrained", quiet=False, load_state_dict=True, **kwargs):
        """Task table (`pretrained_model_ids`) and directory convention of the reference (dptnet.py:219-262,
        sepformer.py:232-272); the download needs the reference's gdown helper, a checkpoint already on disk is loaded directly."""
        task = kwargs.get("task")
        if task not in cls.pretrained_model_ids:
            raise KeyError("Invalid task ({}) is specified.".format(task))
        defaults = {"wsj0-mix": 8000, "wsj0": 8000, "librispeech": 16000}
        if task not in defaults:
            raise NotImplementedError("Not support task={}.".format(task))
        sample_rate, n_sources = kwargs.get("sample_rate") or defaults[task], kwargs.get("n_sources") or 2
        model_choice = kwargs.get("model_choice") or "best"
        model_id = cls.pretrained_model_ids[task][sample_rate][n_sources]
        download_dir = os.path.join(root, cls.__name__, task, "sr{}/{}speakers".format(sample_rate, n_sources))
        model_path = os.path.join(download_dir, "model", "{}.pth".format(model_choice))
        if not os.path.exists(model_path):
            from utils.utils import download_pretrained_model_from_google_drive   # reference helper (gdown), reused as-is
            download_pretrained_model_from_google_drive(model_id, download_dir, quiet=quiet)
        model = cls.build_model(model_path, load_state_dict=load_state_dict)
        model.n_sources, model.sample_rate = n_sources, sample_rate
        return model

    @property
    def num_parameters(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)
