"""
DPTNet (dual-path transformer network) on MI355X: constructor, module tree, state_dict keys and config of reference
src/models/dptnet.py:15-568.  Shell, 1x1 convolutions, chunking and the gated mask end: models/masking.py (libsepkernels);
recurrence of the feed-forward sub-block: the LSTM sweep kernels; layer norms: the gLN / cLN kernels; attention:
`nn.MultiheadAttention` (library SDPA / GEMMs -- which is what attention is).
"""
import torch.nn as nn
import torch.nn.functional as F

import torch

from sepkernels.functional import (attention_core, ChunkToTokensFn, OverlapAddFn, PaddedPointwiseFn, SegmentFn, TokenGLNFn, TokensToChunkFn, dense_apply, linear_apply,
                                   lstm_apply, token_gln_ok)
from utils.model import choose_nonlinear, choose_rnn
from utils.tasnet import choose_layer_norm
from models.gtu import GTU1d
from models.masking import EPS, GatedMaskSeparator, MaskingTasNet, make_mask_nonlinear
from models.transform import OverlapAdd1d, Segment1d


class DPTNet(MaskingTasNet):
    pretrained_model_ids = {"wsj0-mix": {8000: {2: "1QJnJEK8aed7_ED07jD7buyGb37giEDUx", 3: "1Rfb_vS8r2_Oqpg_zAV9y4WMzv106yrSP"},
                                         16000: {2: "", 3: ""}}}
    SEP_KEYS = ("sep_hidden_channels", "sep_bottleneck_channels", "sep_chunk_size", "sep_hop_size", "sep_num_blocks", "sep_num_heads",
                "sep_norm", "sep_nonlinear", "sep_dropout")

    def __init__(self, n_basis, kernel_size, stride=None, enc_basis=None, dec_basis=None, sep_bottleneck_channels=64,
                 sep_hidden_channels=256, sep_chunk_size=100, sep_hop_size=None, sep_num_blocks=6, sep_num_heads=4, sep_norm=True,
                 sep_nonlinear="relu", sep_dropout=0, mask_nonlinear="relu", causal=False, n_sources=2, eps=EPS, **kwargs):
        super().__init__()
        if sep_hop_size is None:
            sep_hop_size = sep_chunk_size // 2
        if stride is None:
            stride = kernel_size // 2
        assert kernel_size % stride == 0, "kernel_size is expected divisible by stride"
        assert n_basis % sep_num_heads == 0, "n_basis must be divisible by sep_num_heads"
        self.sep_bottleneck_channels, self.sep_hidden_channels = sep_bottleneck_channels, sep_hidden_channels
        self.sep_chunk_size, self.sep_hop_size = sep_chunk_size, sep_hop_size
        self.sep_num_blocks, self.sep_num_heads = sep_num_blocks, sep_num_heads
        self.sep_norm, self.sep_nonlinear, self.sep_dropout = sep_norm, sep_nonlinear, sep_dropout
        self.causal, self.mask_nonlinear = causal, mask_nonlinear
        self.n_sources, self.eps = n_sources, eps
        self._init_filterbank(n_basis, kernel_size, stride, enc_basis, dec_basis, kwargs)
        encoder, decoder = self.encoder, self.decoder           # registration order of the reference: encoder, separator, decoder
        del self.encoder, self.decoder
        self.encoder = encoder
        self.separator = Separator(n_basis, bottleneck_channels=sep_bottleneck_channels, hidden_channels=sep_hidden_channels,
                                   chunk_size=sep_chunk_size, hop_size=sep_hop_size, num_blocks=sep_num_blocks, num_heads=sep_num_heads,
                                   norm=sep_norm, nonlinear=sep_nonlinear, dropout=sep_dropout, mask_nonlinear=mask_nonlinear,
                                   causal=causal, n_sources=n_sources, eps=eps)
        self.decoder = decoder


class Separator(GatedMaskSeparator):
    """1x1 bottleneck -> chunks -> norm -> dual-path transformer -> overlap-add -> gated mask end (reference dptnet.py:278-348)"""

    def __init__(self, num_features, bottleneck_channels=32, hidden_channels=128, chunk_size=100, hop_size=None, num_blocks=6,
                 num_heads=4, norm=True, nonlinear="relu", dropout=0, mask_nonlinear="relu", causal=True, n_sources=2, eps=EPS):
        super().__init__()
        if hop_size is None:
            hop_size = chunk_size // 2
        self.num_features, self.n_sources = num_features, n_sources
        self.bottleneck_channels = bottleneck_channels
        self.chunk_size, self.hop_size = chunk_size, hop_size
        self.bottleneck_conv1d = nn.Conv1d(num_features, bottleneck_channels, kernel_size=1, stride=1)
        self.segment1d = Segment1d(chunk_size, hop_size)
        self.norm2d = choose_layer_norm("cLN" if causal else "gLN", bottleneck_channels, causal=causal, eps=eps)
        self.dptransformer = DualPathTransformer(bottleneck_channels, hidden_channels, num_blocks=num_blocks, num_heads=num_heads,
                                                 norm=norm, nonlinear=nonlinear, dropout=dropout, causal=causal, eps=eps)
        self.overlap_add1d = OverlapAdd1d(chunk_size, hop_size)
        self.prelu = nn.PReLU()
        self.map = nn.Conv1d(bottleneck_channels, n_sources * num_features, kernel_size=1, stride=1)
        self.gtu = GTU1d(num_features, num_features, kernel_size=1, stride=1)
        self.mask_nonlinear = make_mask_nonlinear(mask_nonlinear)

    def forward(self, input):
        """input (batch_size, num_features, n_frames) -> (batch_size, n_sources, num_features, n_frames)"""
        batch_size, _, n_frames = input.size()
        pad_left, pad_right = self._chunk_padding(n_frames)
        x = F.pad(self.bottleneck_conv1d(input), (pad_left, pad_right))
        x = self.dptransformer(self.norm2d(self.segment1d(x)))
        x = F.pad(self.overlap_add1d(x), (-pad_left, -pad_right))
        return self._mask(x, batch_size, n_frames)

    def mask_padded(self, w, n_frames):
        x = PaddedPointwiseFn.apply(w, n_frames, self.bottleneck_conv1d.weight, self.bottleneck_conv1d.bias, None)
        x = SegmentFn.apply(x, n_frames, self.chunk_size, self.hop_size)
        x = self.dptransformer(self.norm2d(x))
        return self._mask_padded(OverlapAddFn.apply(x, n_frames, w.shape[2], self.hop_size), n_frames)


class DualPathTransformer(nn.Module):
    def __init__(self, num_features, hidden_channels, num_blocks=6, num_heads=4, norm=True, nonlinear="relu", dropout=0, causal=False, eps=EPS):
        super().__init__()
        self.net = nn.Sequential(*[DualPathTransformerBlock(num_features, hidden_channels, num_heads=num_heads, norm=norm,
                                                            nonlinear=nonlinear, dropout=dropout, causal=causal, eps=eps)
                                   for _ in range(num_blocks)])

    def forward(self, input):
        """(batch_size, num_features, S, chunk_size) -> same shape"""
        if self._tokens_ok(input):
            return self._forward_tokens(input)
        return self.net(input)

    def _tokens_ok(self, input):
        """every norm of the stack a gLN (a causal model's inter-chunk path carries cLN: cumulative along the sequence, another kernel), a
        feature count sep_gln_tokens_* takes, tensors the backend takes"""
        B, C, S, K = input.shape
        if max(S, K) > 65535 or B * ((C + 31) // 32) > 65535 or B * max(S, K) > 65535:      # grid limits of sep_chunk_to_tokens / sep_gln_tokens_* / sep_attn_*
            return False
        probe = input.new_empty(1, 1, input.shape[1])
        for block in self.net:
            for tr in (block.intra_chunk_block.transformer, block.inter_chunk_block.transformer):
                for sub in (tr.multihead_attn_block, tr.subnet):
                    if sub.norm and not token_gln_ok(probe, sub.norm1d):
                        return False
        return token_gln_ok(probe, _ANY_GLN)

    def _forward_tokens(self, input):
        """The same stack with the features innermost from end to end: (B, C, S, K) -> tokens (B*S, K, C) once (sep_chunk_to_tokens), every
        sub-block on token-major rows (attention projections and the Linear on csrc/linear.hip, the LSTM sweeps, gLN on sep_gln_tokens_*),
        one transposing copy between the intra- and the inter-chunk path, back once (sep_tokens_to_chunk).  The module-by-module form
        above makes ~10 strided copies per transformer and direction (244 per step at the recipe's size)."""
        B, C, S, K = input.shape
        x = ChunkToTokensFn.apply(input, False)                                  # (B*S, K, C): a sequence per (b, s)
        for block in self.net:
            x = block.intra_chunk_block.transformer.forward_tokens(x)
            x = x.view(B, S, K, C).transpose(1, 2).contiguous().view(B * K, S, C)    # a sequence per (b, k)
            x = block.inter_chunk_block.transformer.forward_tokens(x)
            x = x.view(B, K, S, C).transpose(1, 2).contiguous().view(B * S, K, C)
        return TokensToChunkFn.apply(x, (B, C, S, K), False)


class _AnyGLN:
    pass


_AnyGLN.__name__ = "GlobalLayerNorm"
_ANY_GLN = _AnyGLN()


class DualPathTransformerBlock(nn.Module):
    def __init__(self, num_features, hidden_channels, num_heads=4, norm=True, nonlinear="relu", dropout=0, causal=False, eps=EPS):
        super().__init__()
        self.intra_chunk_block = IntraChunkTransformer(num_features, hidden_channels, num_heads=num_heads, norm=norm,
                                                       nonlinear=nonlinear, dropout=dropout, eps=eps)
        self.inter_chunk_block = InterChunkTransformer(num_features, hidden_channels, num_heads=num_heads, norm=norm,
                                                       nonlinear=nonlinear, dropout=dropout, causal=causal, eps=eps)

    def forward(self, input):
        return self.inter_chunk_block(self.intra_chunk_block(input))


class _ChunkPathTransformer(nn.Module):
    """One path of a block: the `improved transformer` along one of the two chunk axes, every other index a batch entry.
    Unlike the DPRNN paths there is no residual connection around the path (the sub-blocks have their own)."""
    SEQ_AXIS = None         # 3: along the chunk (intra), 2: across chunks (inter)

    def __init__(self, num_features, hidden_channels, num_heads, norm, nonlinear, dropout, causal, eps):
        super().__init__()
        self.num_features = num_features
        self.transformer = ImprovedTransformer(num_features, hidden_channels, num_heads=num_heads, norm=norm, nonlinear=nonlinear,
                                               dropout=dropout, causal=causal, eps=eps)

    def forward(self, input):
        """(batch_size, num_features, S, chunk_size) -> same shape"""
        B, C, S, K = input.size()
        if self.SEQ_AXIS == 3:
            x = input.permute(3, 0, 2, 1).reshape(K, B * S, C)
            return self.transformer(x).view(K, B, S, C).permute(1, 3, 2, 0)
        x = input.permute(2, 0, 3, 1).reshape(S, B * K, C)
        return self.transformer(x).view(S, B, K, C).permute(1, 3, 0, 2)


class IntraChunkTransformer(_ChunkPathTransformer):
    SEQ_AXIS = 3

    def __init__(self, num_features, hidden_channels, num_heads=4, norm=True, nonlinear="relu", dropout=0, eps=EPS):
        super().__init__(num_features, hidden_channels, num_heads, norm, nonlinear, dropout, False, eps)


class InterChunkTransformer(_ChunkPathTransformer):
    SEQ_AXIS = 2

    def __init__(self, num_features, hidden_channels, num_heads=4, causal=False, norm=True, nonlinear="relu", dropout=0, eps=EPS):
        super().__init__(num_features, hidden_channels, num_heads, norm, nonlinear, dropout, causal, eps)


class ImprovedTransformer(nn.Module):
    """self-attention sub-block, then the recurrent `feed-forward` sub-block; (T, batch_size, num_features) in and out"""

    def __init__(self, num_features, hidden_channels, num_heads=4, norm=True, nonlinear="relu", dropout=0, causal=False, eps=EPS):
        super().__init__()
        self.multihead_attn_block = MultiheadAttentionBlock(num_features, num_heads, norm=norm, dropout=dropout, causal=causal, eps=eps)
        self.subnet = FeedForwardBlock(num_features, hidden_channels, norm=norm, nonlinear=nonlinear, causal=causal, eps=eps)

    def forward(self, input):
        return self.subnet(self.multihead_attn_block(input))

    def forward_tokens(self, x):
        """x (nseq, L, C), features contiguous -> the same shape"""
        return self.subnet.forward_tokens(self.multihead_attn_block.forward_tokens(x))


def _norm_time_first(norm1d, x):
    """layer norm of the TasNet family, defined on (batch_size, C, T), applied to (T, batch_size, C)"""
    return norm1d(x.permute(1, 2, 0)).permute(2, 0, 1).contiguous()


class MultiheadAttentionBlock(nn.Module):
    def __init__(self, embed_dim, num_heads, norm=True, dropout=0, causal=False, eps=EPS):
        super().__init__()
        self.dropout = dropout != 0
        self.norm = norm
        self.multihead_attn = nn.MultiheadAttention(embed_dim, num_heads)
        if self.dropout:
            self.dropout1d = nn.Dropout(p=dropout)
        if self.norm:
            self.norm1d = choose_layer_norm("cLN" if causal else "gLN", embed_dim, causal=causal, eps=eps)

    def forward(self, input):
        """(T, batch_size, embed_dim) -> same shape: attention + input [-> dropout] [-> norm] (no mask, also when causal:
        reference dptnet.py:505-527)"""
        x = self.multihead_attn(input, input, input, need_weights=False)[0] + input
        if self.dropout:
            x = self.dropout1d(x)
        return _norm_time_first(self.norm1d, x) if self.norm else x

    def forward_tokens(self, x):
        """(nseq, L, embed_dim) -> same shape.  nn.MultiheadAttention's arithmetic (torch/nn/functional.py multi_head_attention_forward: packed
        input projection in the order q, k, v; head h = features h*d .. (h+1)*d - 1; scores scaled by 1/sqrt(d); softmax; output projection; no
        mask, no attention dropout) on batch-first rows: the projections on csrc/linear.hip, the core on csrc/attn.hip (sep_attn_*)."""
        mha = self.multihead_attn
        N, L, C = x.shape
        h = mha.num_heads
        qkv = dense_apply(x, mha.in_proj_weight, mha.in_proj_bias).view(N, L, 3, h, C // h)
        o = attention_core(qkv)                                                   # (N, L, C): csrc/attn.hip, or SDPA for long sequences / wide heads
        y = dense_apply(o, mha.out_proj.weight, mha.out_proj.bias) + x
        if self.dropout:
            y = self.dropout1d(y)
        return TokenGLNFn.apply(y, self.norm1d.norm.weight, self.norm1d.norm.bias, self.norm1d.eps) if self.norm else y


class FeedForwardBlock(nn.Module):
    def __init__(self, num_features, hidden_channels, norm=True, nonlinear="relu", causal=False, eps=EPS):
        super().__init__()
        self.norm = norm
        self.rnn = choose_rnn("lstm", input_size=num_features, hidden_size=hidden_channels, batch_first=False, bidirectional=not causal)
        self.nonlinear1d = choose_nonlinear(nonlinear)
        self.fc = nn.Linear((1 if causal else 2) * hidden_channels, num_features)
        if self.norm:
            self.norm1d = choose_layer_norm("cLN" if causal else "gLN", num_features, causal=causal, eps=eps)

    def forward(self, input):
        """(T, batch_size, num_features) -> same shape: LSTM -> activation -> Linear, + input [-> norm]"""
        h = lstm_apply(input.transpose(0, 1), self.rnn).transpose(0, 1)          # the sweep kernels run sequence-major per batch row
        x = linear_apply(self.nonlinear1d(h), self.fc) + input
        return _norm_time_first(self.norm1d, x) if self.norm else x

    def forward_tokens(self, x):
        """(nseq, L, num_features) -> same shape, batch-first: no transposes around the recurrence"""
        y = linear_apply(self.nonlinear1d(lstm_apply(x, self.rnn)), self.fc) + x
        return TokenGLNFn.apply(y, self.norm1d.norm.weight, self.norm1d.norm.bias, self.norm1d.eps) if self.norm else y


from sepkernels.shadowed import fall_through as _fall_through      # names of the reference's same-named module this tree does not define

__getattr__ = _fall_through(__name__, __file__)
