"""
DPRNN-TasNet on MI355X (BASELINE.json configs[3]).  API, module tree and state_dict keys of reference
src/models/dprnn_tasnet.py:15-351.  Encoder + first gLN + bottleneck ("head") and PReLU + mask conv + sigmoid | softmax +
mask*w + decoder/overlap-add ("tail") are the Conv-TasNet kernels of libsepkernels; chunking is sep_segment /
sep_overlap_add; the dual-path recurrences are models/dprnn.py (LSTM sweep kernels).  Configurations outside that family
(causal: cLN in front of the bottleneck; Fourier / pinv bases; widths off the multiples of 16; float64) run as the
module-by-module composition of the shared shell (models/masking.py), with cLN, gLN, chunking and the recurrences still on
their kernels.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from utils.tasnet import choose_layer_norm
from models.dprnn import DPRNN
from models.masking import MaskingTasNet
from models.transform import Segment1d, OverlapAdd1d
from utils.checkpoint import load_checkpoint
from sepkernels import net as _net
from sepkernels.functional import HeadFn, TailFn, SegmentFn, OverlapAddFn, HEAD_KEYS, TAIL_KEYS, segment_geometry

EPS = 1e-12
SAMPLE_RATE_LIBRISPEECH = 16000


class DPRNNTasNet(MaskingTasNet):
    pretrained_model_ids = {"wsj0-mix": {8000: {2: "1-2DOUDi2NImL7akQzTXLpDqJsJL4HyiY", 3: "1-5EhjEBiArjFat4gXyNkKyUjAkTvkgU0"}, 16000: {2: "", 3: ""}},
                            "librispeech": {SAMPLE_RATE_LIBRISPEECH: {2: "1hTmxhI8JQlNnWVjwWUBGYlC7O_-ykK4H"}}}
    SEP_KEYS = ("sep_hidden_channels", "sep_bottleneck_channels", "sep_chunk_size", "sep_hop_size", "sep_num_blocks", "sep_norm", "rnn_type")
    CONFIG_HAS_IN_CHANNELS = True
    MULTICHANNEL_INPUT = True

    def __init__(self, n_basis, kernel_size, stride=None, enc_basis=None, dec_basis=None, sep_hidden_channels=128,
                 sep_bottleneck_channels=64, sep_chunk_size=100, sep_hop_size=50, sep_num_blocks=6, sep_norm=True,
                 mask_nonlinear="sigmoid", causal=True, rnn_type="lstm", n_sources=2, eps=EPS, **kwargs):
        super().__init__()
        self.sep_hidden_channels, self.sep_bottleneck_channels = sep_hidden_channels, sep_bottleneck_channels
        self.sep_chunk_size, self.sep_hop_size = sep_chunk_size, sep_hop_size
        self.sep_num_blocks = sep_num_blocks
        self.causal = causal
        self.sep_norm = sep_norm
        self.mask_nonlinear = mask_nonlinear
        self.rnn_type = rnn_type
        self.n_sources = n_sources
        self.eps = eps
        stride = self._init_filterbank(n_basis, kernel_size, stride, enc_basis, dec_basis, kwargs)
        encoder, decoder = self.encoder, self.decoder           # registration order of the reference: encoder, separator, decoder
        del self.encoder, self.decoder
        self.encoder = encoder
        self.separator = Separator(n_basis, bottleneck_channels=sep_bottleneck_channels, hidden_channels=sep_hidden_channels,
                                   chunk_size=sep_chunk_size, hop_size=sep_hop_size, num_blocks=sep_num_blocks, norm=sep_norm,
                                   mask_nonlinear=mask_nonlinear, causal=causal, rnn_type=rnn_type, n_sources=n_sources, eps=eps)
        self.decoder = decoder

    def kernel_path_problems(self):
        problems = super().kernel_path_problems()
        if self.causal:
            problems.append("causal=True puts a cLN in front of the bottleneck (the fused head normalises globally)")
        return problems

    def _run_kernels(self, mixture):
        T = mixture.shape[-1]
        cfg = {"n_basis": self.n_basis, "kernel_size": self.kernel_size, "stride": self.stride,
               "sep_bottleneck_channels": self.sep_bottleneck_channels, "n_sources": self.n_sources,
               "enc_nonlinear": "relu" if self.encoder.nonlinear else None, "mask_nonlinear": self.mask_nonlinear, "eps": self.eps}
        P = dict(self.named_parameters())
        w, x0 = HeadFn.apply(mixture, cfg, *[P[k] for k in HEAD_KEYS])
        geo = _net.Geometry(T, self.kernel_size, self.stride)
        core = self.separator.dual_path(x0, geo.F, geo.ldt)
        est, latent = TailFn.apply(w, core, cfg, geo, tuple(mixture.shape), True, *[P[k] for k in TAIL_KEYS])
        return est, latent[..., :geo.F]

    def get_config(self):
        config = super().get_config()
        return {k: config[k] for k in ("in_channels", "n_basis", "kernel_size", "stride", "enc_basis", "dec_basis", "enc_nonlinear", "window_fn",
                                       "enc_onesided", "enc_return_complex", "sep_hidden_channels", "sep_bottleneck_channels", "sep_chunk_size",
                                       "sep_hop_size", "sep_num_blocks", "causal", "sep_norm", "mask_nonlinear", "rnn_type", "n_sources", "eps")}

    @classmethod
    def build_model(cls, model_path, load_state_dict=False, trust_pickle=None):
        config = load_checkpoint(model_path, trust_pickle)
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
        """the separator module by module (reference dprnn_tasnet.py:323-349): input (batch_size, num_features, n_frames) ->
        mask (batch_size, n_sources, num_features, n_frames)"""
        batch_size, _, n_frames = input.size()
        pad_left, pad_right, _ = segment_geometry(n_frames, self.chunk_size, self.hop_size)
        x = F.pad(self.bottleneck_conv1d(self.norm1d(input)), (pad_left, pad_right))
        x = F.pad(self.overlap_add1d(self.dprnn(self.segment1d(x))), (-pad_left, -pad_right))
        x = self.mask_nonlinear(self.mask_conv1d(self.prelu(x)))
        return x.view(batch_size, self.n_sources, self.num_features, n_frames)

    def padded_problems(self):
        """why the head / dual_path / tail kernel sequence cannot run (empty: it can)"""
        widths = {"num_features": self.num_features, "bottleneck_channels": self.bottleneck_conv1d.out_channels,
                  "n_sources*num_features": self.n_sources * self.num_features}
        return ["{} must be a multiple of 16".format(k) for k, v in widths.items() if v % 16]


from sepkernels.shadowed import fall_through as _fall_through      # names of the reference's same-named module this tree does not define

__getattr__ = _fall_through(__name__, __file__)
