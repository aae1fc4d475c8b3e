"""
GALRNet on MI355X: constructor, module tree, state_dict keys and config of reference src/models/galrnet.py:13-268.
No bottleneck: the encoder output itself is chunked.  Shell and gated mask end: models/masking.py; blocks: models/galr.py.
"""
import torch.nn as nn
import torch.nn.functional as F

from sepkernels.functional import OverlapAddFn, SegmentFn
from utils.tasnet import choose_layer_norm
from models.galr import GALR
from models.gtu import GTU1d
from models.masking import EPS, GatedMaskSeparator, MaskingTasNet, make_mask_nonlinear
from models.transform import OverlapAdd1d, Segment1d


class GALRNet(MaskingTasNet):
    SEP_KEYS = ("sep_hidden_channels", "sep_chunk_size", "sep_hop_size", "sep_down_chunk_size", "sep_num_blocks", "sep_num_heads",
                "sep_norm", "sep_dropout", "low_dimension")

    def __init__(self, n_basis, kernel_size, stride=None, enc_basis=None, dec_basis=None, sep_hidden_channels=128, sep_chunk_size=100,
                 sep_hop_size=50, sep_down_chunk_size=None, sep_num_blocks=6, sep_num_heads=8, sep_norm=True, sep_dropout=0.1,
                 mask_nonlinear="relu", causal=True, n_sources=2, low_dimension=True, eps=EPS, **kwargs):
        super().__init__()
        self.sep_hidden_channels = sep_hidden_channels
        self.sep_chunk_size, self.sep_hop_size, self.sep_down_chunk_size = sep_chunk_size, sep_hop_size, sep_down_chunk_size
        self.sep_num_blocks, self.sep_num_heads = sep_num_blocks, sep_num_heads
        self.sep_norm, self.sep_dropout, self.low_dimension = sep_norm, sep_dropout, low_dimension
        self.causal, self.mask_nonlinear = causal, mask_nonlinear
        self.n_sources, self.eps = n_sources, eps
        self._init_filterbank(n_basis, kernel_size, stride, enc_basis, dec_basis, kwargs)
        encoder, decoder = self.encoder, self.decoder           # registration order of the reference: encoder, separator, decoder
        del self.encoder, self.decoder
        self.encoder = encoder
        self.separator = Separator(n_basis, hidden_channels=sep_hidden_channels, chunk_size=sep_chunk_size, hop_size=sep_hop_size,
                                   down_chunk_size=sep_down_chunk_size, num_blocks=sep_num_blocks, num_heads=sep_num_heads, norm=sep_norm,
                                   dropout=sep_dropout, mask_nonlinear=mask_nonlinear, low_dimension=low_dimension, causal=causal,
                                   n_sources=n_sources, eps=eps)
        self.decoder = decoder


class Separator(GatedMaskSeparator):
    """chunks -> norm -> GALR blocks -> overlap-add -> gated mask end (reference galrnet.py:166-246)"""

    def __init__(self, num_features, hidden_channels=128, chunk_size=100, hop_size=50, down_chunk_size=None, num_blocks=6, num_heads=4,
                 norm=True, dropout=0.1, mask_nonlinear="relu", low_dimension=True, causal=True, n_sources=2, eps=EPS):
        super().__init__()
        self.num_features, self.n_sources = num_features, n_sources
        self.chunk_size, self.hop_size = chunk_size, hop_size
        self.segment1d = Segment1d(chunk_size, hop_size)
        self.norm2d = choose_layer_norm("cLN" if causal else "gLN", num_features, causal=causal, eps=eps)
        if low_dimension:
            if down_chunk_size is None:
                raise ValueError("Specify down_chunk_size")
            extra = dict(chunk_size=chunk_size, down_chunk_size=down_chunk_size)
        else:
            extra = {}
        self.galr = GALR(num_features, hidden_channels, num_blocks=num_blocks, num_heads=num_heads, norm=norm, dropout=dropout,
                         low_dimension=low_dimension, causal=causal, eps=eps, **extra)
        self.overlap_add1d = OverlapAdd1d(chunk_size, hop_size)
        self.prelu = nn.PReLU()
        self.map = nn.Conv1d(num_features, n_sources * num_features, kernel_size=1, stride=1)
        self.gtu = GTU1d(num_features, num_features, kernel_size=1, stride=1)
        self.mask_nonlinear = make_mask_nonlinear(mask_nonlinear)

    def forward(self, input):
        """input (batch_size, num_features, n_frames) -> (batch_size, n_sources, num_features, n_frames)"""
        batch_size, _, n_frames = input.size()
        pad_left, pad_right = self._chunk_padding(n_frames)
        x = self.galr(self.norm2d(self.segment1d(F.pad(input, (pad_left, pad_right)))))
        x = F.pad(self.overlap_add1d(x), (-pad_left, -pad_right))
        return self._mask(x, batch_size, n_frames)

    def mask_padded(self, w, n_frames):
        x = SegmentFn.apply(w, n_frames, self.chunk_size, self.hop_size)
        x = self.galr(self.norm2d(x))
        return self._mask_padded(OverlapAddFn.apply(x, n_frames, w.shape[2], self.hop_size), n_frames)


from sepkernels.shadowed import fall_through as _fall_through      # names of the reference's same-named module this tree does not define

__getattr__ = _fall_through(__name__, __file__)
