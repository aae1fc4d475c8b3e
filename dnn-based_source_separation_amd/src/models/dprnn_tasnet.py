"""
DPRNN-TasNet on MI355X (BASELINE.json configs[3]).  API, module tree and state_dict keys of reference
src/models/dprnn_tasnet.py:15-351.  Encoder + first gLN + bottleneck ("head") and PReLU + mask conv + sigmoid +
mask*w + decoder/overlap-add ("tail") are the Conv-TasNet kernels of libsepkernels; chunking is sep_segment /
sep_overlap_add; the dual-path recurrences are the interim torch.nn.LSTM composition of models/dprnn.py.
"""
import torch
import torch.nn as nn

from utils.filterbank import choose_filterbank
from utils.tasnet import choose_layer_norm
from models.dprnn import DPRNN
from models.transform import Segment1d, OverlapAdd1d
from sepkernels import net as _net
from sepkernels.functional import HeadFn, TailFn, SegmentFn, OverlapAddFn, HEAD_KEYS, TAIL_KEYS

EPS = 1e-12


class DPRNNTasNet(nn.Module):
    def __init__(self, n_basis, kernel_size, stride=None, enc_basis=None, dec_basis=None, sep_hidden_channels=128,
                 sep_bottleneck_channels=64, sep_chunk_size=100, sep_hop_size=50, sep_num_blocks=6, sep_norm=True,
                 mask_nonlinear="sigmoid", causal=True, rnn_type="lstm", n_sources=2, eps=EPS, **kwargs):
        super().__init__()
        if stride is None:
            stride = kernel_size // 2
        assert kernel_size % stride == 0, "kernel_size is expected divisible by stride"
        self.in_channels = kwargs.get("in_channels", 1)
        self.n_basis = n_basis
        self.kernel_size, self.stride = kernel_size, stride
        self.enc_basis, self.dec_basis = enc_basis, dec_basis
        self.enc_nonlinear = kwargs["enc_nonlinear"] if (enc_basis == "trainable" and not dec_basis == "pinv") else None
        self.window_fn, self.enc_onesided, self.enc_return_complex = None, None, None
        self.sep_hidden_channels, self.sep_bottleneck_channels = sep_hidden_channels, sep_bottleneck_channels
        self.sep_chunk_size, self.sep_hop_size = sep_chunk_size, sep_hop_size
        self.sep_num_blocks = sep_num_blocks
        self.causal = causal
        self.sep_norm = sep_norm
        self.mask_nonlinear = mask_nonlinear
        self.rnn_type = rnn_type
        self.n_sources = n_sources
        self.eps = eps
        encoder, decoder = choose_filterbank(n_basis, kernel_size=kernel_size, stride=stride, enc_basis=enc_basis, dec_basis=dec_basis, **kwargs)
        self.encoder = encoder
        self.separator = Separator(n_basis, bottleneck_channels=sep_bottleneck_channels, hidden_channels=sep_hidden_channels,
                                   chunk_size=sep_chunk_size, hop_size=sep_hop_size, num_blocks=sep_num_blocks, norm=sep_norm,
                                   mask_nonlinear=mask_nonlinear, causal=causal, rnn_type=rnn_type, n_sources=n_sources, eps=eps)
        self.decoder = decoder

    def forward(self, input):
        output, _ = self._run(input, False)
        return output

    def extract_latent(self, input):
        """input (B, 1, T) -> output (B, n_sources, T), latent (B, n_sources, n_basis, T')"""
        return self._run(input, True)

    def _kernel_cfg(self):
        if self.causal:
            raise NotImplementedError("causal DPRNN-TasNet (cLN) is not implemented on the MI355X path")
        if self.mask_nonlinear != "sigmoid":
            raise NotImplementedError("mask_nonlinear must be 'sigmoid'")
        if self.enc_nonlinear not in (None, "", "relu"):
            raise NotImplementedError("enc_nonlinear must be None or 'relu'")
        if self.n_basis % 16 or self.sep_bottleneck_channels % 16 or (self.n_sources * self.n_basis) % 16:
            raise NotImplementedError("n_basis and sep_bottleneck_channels must be multiples of 16")
        return {"n_basis": self.n_basis, "kernel_size": self.kernel_size, "stride": self.stride,
                "sep_bottleneck_channels": self.sep_bottleneck_channels, "n_sources": self.n_sources,
                "enc_nonlinear": self.enc_nonlinear, "eps": self.eps}

    def _run(self, input, want_latent):
        n_dim = input.dim()
        if n_dim == 3:
            batch_size, C_in, T = input.size()
            assert C_in == 1, "input.size() is expected (?, 1, ?), but given {}".format(input.size())
            mixture = input
        elif n_dim == 4:
            batch_size, C_in, n_mics, T = input.size()
            assert C_in == 1, "input.size() is expected (?, 1, ?, ?), but given {}".format(input.size())
            mixture = input.view(batch_size, n_mics, T)
        else:
            raise ValueError("Not support {} dimension input".format(n_dim))
        cfg = self._kernel_cfg()
        if not mixture.is_cuda and _net.backend().name == "hip":
            raise RuntimeError("DPRNNTasNet (MI355X build) runs on the GPU only; there is no CPU fallback.")
        mixture = mixture.contiguous()
        if mixture.dtype != torch.float32 and _net.backend().name == "hip":
            mixture = mixture.float()
        P = dict(self.named_parameters())
        w, x0 = HeadFn.apply(mixture, cfg, *[P[k] for k in HEAD_KEYS])
        geo = _net.Geometry(T, self.kernel_size, self.stride)
        core = self.separator.dual_path(x0, geo.F, geo.ldt)
        out = TailFn.apply(w, core, cfg, geo, tuple(mixture.shape), want_latent, *[P[k] for k in TAIL_KEYS])
        est, latent = (out if want_latent else (out, None))
        if latent is not None:
            latent = latent[..., :geo.F]
        if n_dim == 3:
            est = est.view(batch_size, self.n_sources, T)
        return est, latent

    def get_config(self):
        return {"in_channels": self.in_channels, "n_basis": self.n_basis, "kernel_size": self.kernel_size, "stride": self.stride,
                "enc_basis": self.enc_basis, "dec_basis": self.dec_basis, "enc_nonlinear": self.enc_nonlinear,
                "window_fn": self.window_fn, "enc_onesided": self.enc_onesided, "enc_return_complex": self.enc_return_complex,
                "sep_hidden_channels": self.sep_hidden_channels, "sep_bottleneck_channels": self.sep_bottleneck_channels,
                "sep_chunk_size": self.sep_chunk_size, "sep_hop_size": self.sep_hop_size, "sep_num_blocks": self.sep_num_blocks,
                "causal": self.causal, "sep_norm": self.sep_norm, "mask_nonlinear": self.mask_nonlinear,
                "rnn_type": self.rnn_type, "n_sources": self.n_sources, "eps": self.eps}

    def get_package(self):
        return self.get_config()

    @classmethod
    def build_model(cls, model_path, load_state_dict=False):
        config = torch.load(model_path, map_location=lambda storage, loc: storage, weights_only=False)
        model = cls(config.get("n_bases") or config["n_basis"], in_channels=config.get("in_channels") or 1,
                    kernel_size=config["kernel_size"], stride=config["stride"],
                    enc_basis=config.get("enc_bases") or config["enc_basis"], dec_basis=config.get("dec_bases") or config["dec_basis"],
                    enc_nonlinear=config["enc_nonlinear"], window_fn=config["window_fn"],
                    enc_onesided=config.get("enc_onesided") or None, enc_return_complex=config.get("enc_return_complex") or None,
                    sep_hidden_channels=config["sep_hidden_channels"], sep_bottleneck_channels=config["sep_bottleneck_channels"],
                    sep_chunk_size=config["sep_chunk_size"], sep_hop_size=config["sep_hop_size"], sep_num_blocks=config["sep_num_blocks"],
                    sep_norm=config["sep_norm"], mask_nonlinear=config["mask_nonlinear"], causal=config["causal"],
                    rnn_type=config.get("rnn_type", "lstm"), n_sources=config["n_sources"], eps=config["eps"])
        if load_state_dict:
            model.load_state_dict(config["state_dict"])
        return model

    @property
    def num_parameters(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)


class Separator(nn.Module):
    def __init__(self, num_features, bottleneck_channels=64, hidden_channels=128, chunk_size=100, hop_size=50, num_blocks=6,
                 norm=True, mask_nonlinear="sigmoid", causal=True, rnn_type="lstm", n_sources=2, eps=EPS):
        super().__init__()
        self.num_features, self.n_sources = num_features, n_sources
        self.chunk_size, self.hop_size = chunk_size, hop_size
        self.norm = norm
        norm_name = "cLN" if causal else "gLN"
        self.norm1d = choose_layer_norm(norm_name, num_features, causal=causal, eps=eps)
        self.bottleneck_conv1d = nn.Conv1d(num_features, bottleneck_channels, kernel_size=1, stride=1)
        self.segment1d = Segment1d(chunk_size, hop_size)
        self.dprnn = DPRNN(bottleneck_channels, hidden_channels, num_blocks=num_blocks, causal=causal, norm=norm, rnn_type=rnn_type, eps=eps)
        self.overlap_add1d = OverlapAdd1d(chunk_size, hop_size)
        self.prelu = nn.PReLU()
        self.mask_conv1d = nn.Conv1d(bottleneck_channels, n_sources * num_features, kernel_size=1, stride=1)
        if mask_nonlinear not in ("sigmoid", "softmax"):
            raise ValueError("Cannot support {}".format(mask_nonlinear))
        self.mask_nonlinear = nn.Sigmoid() if mask_nonlinear == "sigmoid" else nn.Softmax(dim=1)

    def dual_path(self, x0, n_frames, ldt):
        """bottleneck output (B, Bn, ldt) -> pad + segment -> DPRNN -> overlap-add + crop (B, Bn, ldt)
        (reference dprnn_tasnet.py:335-345 between bottleneck_conv1d and prelu)."""
        seg = SegmentFn.apply(x0, n_frames, self.chunk_size, self.hop_size)
        y = self.dprnn(seg)
        return OverlapAddFn.apply(y, n_frames, ldt, self.hop_size)

    def forward(self, input):
        raise NotImplementedError("the DPRNN-TasNet separator runs inside DPRNNTasNet (head / dual_path / tail)")
