"""
SepFormer on MI355X: constructor, module tree, state_dict keys and config of reference src/models/sepformer.py:16-560.
Front: encoder -> gLN -> 1x1 bottleneck, i.e. the fused head of the TasNet kernels (HeadFn) when not causal.  Shell and
gated mask end (with the extra 1x1 output convolution): models/masking.py.  Core: stacks of `nn.TransformerEncoder` layers
along and across chunks (library attention / GEMMs), each stack closed by a gLN / cLN (kernels) and wrapped in a residual.
"""
import torch.nn as nn
import torch.nn.functional as F

from sepkernels.functional import HeadFn, HEAD_KEYS, OverlapAddFn, PaddedPointwiseFn, SegmentFn
from utils.tasnet import choose_layer_norm
from models.gtu import GTU1d
from models.masking import EPS, GatedMaskSeparator, MaskingTasNet, make_mask_nonlinear
from models.transform import OverlapAdd1d, Segment1d
from models.transformer import PositionalEncoding


class SepFormer(MaskingTasNet):
    pretrained_model_ids = {"wsj0-mix": {8000: {2: "1-9pOv2B612IykvpA6kaGZSg4AUQPnoCg", 3: "1-Rz31CGWVVzYVHXgIdp7Tuc0__K2SCPs"}}}
    SEP_KEYS = ("sep_bottleneck_channels", "sep_chunk_size", "sep_hop_size", "sep_num_blocks", "sep_num_layers_intra",
                "sep_num_layers_inter", "sep_num_heads_intra", "sep_num_heads_inter", "sep_d_ff_intra", "sep_d_ff_inter", "sep_norm",
                "sep_nonlinear", "sep_dropout")
    CONFIG_HAS_IN_CHANNELS = True
    MULTICHANNEL_INPUT = True

    def __init__(self, n_basis, kernel_size, stride=None, enc_basis=None, dec_basis=None, sep_bottleneck_channels=None,
                 sep_chunk_size=250, sep_hop_size=125, sep_num_blocks=2, sep_num_layers_intra=8, sep_num_layers_inter=8,
                 sep_num_heads_intra=8, sep_num_heads_inter=8, sep_d_ff_intra=1024, sep_d_ff_inter=1024, sep_norm=True,
                 sep_nonlinear="relu", sep_dropout=1e-1, mask_nonlinear="relu", causal=True, n_sources=2, eps=EPS, **kwargs):
        super().__init__()
        if sep_bottleneck_channels is None:
            sep_bottleneck_channels = n_basis
        self.sep_bottleneck_channels = sep_bottleneck_channels
        self.sep_chunk_size, self.sep_hop_size = sep_chunk_size, sep_hop_size
        self.sep_num_blocks = sep_num_blocks
        self.sep_num_layers_intra, self.sep_num_layers_inter = sep_num_layers_intra, sep_num_layers_inter
        self.sep_num_heads_intra, self.sep_num_heads_inter = sep_num_heads_intra, sep_num_heads_inter
        self.sep_d_ff_intra, self.sep_d_ff_inter = sep_d_ff_intra, sep_d_ff_inter
        self.sep_norm, self.sep_nonlinear, self.sep_dropout = sep_norm, sep_nonlinear, sep_dropout
        self.causal, self.mask_nonlinear = causal, mask_nonlinear
        self.n_sources, self.eps = n_sources, eps
        self._init_filterbank(n_basis, kernel_size, stride, enc_basis, dec_basis, kwargs)
        encoder, decoder = self.encoder, self.decoder           # registration order of the reference: encoder, separator, decoder
        del self.encoder, self.decoder
        self.encoder = encoder
        self.separator = Separator(n_basis, sep_bottleneck_channels, chunk_size=sep_chunk_size, hop_size=sep_hop_size,
                                   num_blocks=sep_num_blocks, num_layers_intra=sep_num_layers_intra, num_layers_inter=sep_num_layers_inter,
                                   num_heads_intra=sep_num_heads_intra, num_heads_inter=sep_num_heads_inter, d_ff_intra=sep_d_ff_intra,
                                   d_ff_inter=sep_d_ff_inter, norm=sep_norm, nonlinear=sep_nonlinear, dropout=sep_dropout,
                                   mask_nonlinear=mask_nonlinear, causal=causal, n_sources=n_sources, eps=eps)
        self.decoder = decoder

    def _enter(self, mixture):
        """not causal: analysis basis, gLN and the 1x1 bottleneck as the fused head of the TasNet kernels -> (w, bottleneck output)"""
        if self.causal:
            return super()._enter(mixture)
        cfg = {"n_basis": self.n_basis, "kernel_size": self.kernel_size, "stride": self.stride, "n_sources": self.n_sources,
               "sep_bottleneck_channels": self.sep_bottleneck_channels, "enc_nonlinear": "relu" if self.encoder.nonlinear else None,
               "eps": self.eps}
        sep = self.separator
        params = {"encoder.conv1d.weight": self.encoder.conv1d.weight, "separator.norm1d.norm.weight": sep.norm1d.norm.weight,
                  "separator.norm1d.norm.bias": sep.norm1d.norm.bias, "separator.bottleneck_conv1d.weight": sep.bottleneck_conv1d_in.weight,
                  "separator.bottleneck_conv1d.bias": sep.bottleneck_conv1d_in.bias}
        w, x0 = HeadFn.apply(mixture, cfg, *[params[k] for k in HEAD_KEYS])
        return w, (w, x0)


class Separator(GatedMaskSeparator):
    """norm -> 1x1 bottleneck -> chunks -> SepFormer blocks -> overlap-add -> gated mask end with a 1x1 output convolution
    (reference sepformer.py:281-361)"""

    def __init__(self, num_features, bottleneck_channels, chunk_size=250, hop_size=125, num_blocks=2, num_layers_intra=8,
                 num_layers_inter=8, num_heads_intra=8, num_heads_inter=8, d_ff_intra=1024, d_ff_inter=1024, norm=True, nonlinear="relu",
                 dropout=1e-1, mask_nonlinear="relu", causal=False, n_sources=2, eps=EPS):
        super().__init__()
        self.num_features, self.n_sources = num_features, n_sources
        self.bottleneck_channels = bottleneck_channels
        self.chunk_size, self.hop_size = chunk_size, hop_size
        self.norm = norm
        self.norm1d = choose_layer_norm("cLN" if causal else "gLN", num_features, causal=causal, eps=eps)
        self.bottleneck_conv1d_in = nn.Conv1d(num_features, bottleneck_channels, kernel_size=1, stride=1)
        self.segment1d = Segment1d(chunk_size, hop_size)
        self.dptransformer = SepFormerBackbone(num_blocks=num_blocks, num_layers_intra=num_layers_intra, num_layers_inter=num_layers_inter,
                                               num_heads_intra=num_heads_intra, num_heads_inter=num_heads_inter, d_intra=bottleneck_channels,
                                               d_inter=bottleneck_channels, d_ff_intra=d_ff_intra, d_ff_inter=d_ff_inter, norm=norm,
                                               dropout=dropout, nonlinear=nonlinear, causal=causal, eps=eps)
        self.overlap_add1d = OverlapAdd1d(chunk_size, hop_size)
        self.prelu = nn.PReLU()
        self.map = nn.Conv1d(bottleneck_channels, n_sources * num_features, kernel_size=1, stride=1)
        self.gtu = GTU1d(num_features, num_features, kernel_size=1, stride=1)
        self.bottleneck_conv1d_out = nn.Conv1d(num_features, num_features, kernel_size=1, stride=1)
        self.mask_nonlinear = make_mask_nonlinear(mask_nonlinear)

    def forward(self, input):
        """input (batch_size, num_features, n_frames) -> (batch_size, n_sources, num_features, n_frames)"""
        batch_size, _, n_frames = input.size()
        pad_left, pad_right = self._chunk_padding(n_frames)
        x = F.pad(self.bottleneck_conv1d_in(self.norm1d(input)), (pad_left, pad_right))
        x = self.dptransformer(self.segment1d(x))
        x = F.pad(self.overlap_add1d(x), (-pad_left, -pad_right))
        return self._mask(x, batch_size, n_frames)

    def mask_padded(self, entry, n_frames):
        if isinstance(entry, tuple):                   # (w, bottleneck output) of the fused head
            ldt, x = entry[0].shape[2], entry[1]
        else:                                          # causal: cLN is a kernel of its own, on the valid frames
            ldt = entry.shape[2]
            x = F.pad(self.norm1d(entry[..., :n_frames]), (0, ldt - n_frames))
            x = PaddedPointwiseFn.apply(x, n_frames, self.bottleneck_conv1d_in.weight, self.bottleneck_conv1d_in.bias, None)
        x = self.dptransformer(SegmentFn.apply(x, n_frames, self.chunk_size, self.hop_size))
        return self._mask_padded(OverlapAddFn.apply(x, n_frames, ldt, self.hop_size), n_frames)


class SepFormerBackbone(nn.Module):
    def __init__(self, num_blocks=2, num_layers_intra=8, num_layers_inter=8, num_heads_intra=8, num_heads_inter=8, d_intra=256,
                 d_inter=256, d_ff_intra=1024, d_ff_inter=1024, norm=True, dropout=1e-1, nonlinear="relu", causal=False, eps=EPS):
        super().__init__()
        self.net = nn.Sequential(*[SepFormerBlock(num_layers_intra=num_layers_intra, num_layers_inter=num_layers_inter,
                                                  num_heads_intra=num_heads_intra, num_heads_inter=num_heads_inter, d_intra=d_intra,
                                                  d_inter=d_inter, d_ff_intra=d_ff_intra, d_ff_inter=d_ff_inter, norm=norm, dropout=dropout,
                                                  nonlinear=nonlinear, causal=causal, eps=eps) for _ in range(num_blocks)])

    def forward(self, input):
        """(batch_size, num_features, S, chunk_size) -> same shape"""
        return self.net(input)


class SepFormerBlock(nn.Module):
    def __init__(self, num_layers_intra=8, num_layers_inter=8, num_heads_intra=8, num_heads_inter=8, d_intra=256, d_inter=256,
                 d_ff_intra=1024, d_ff_inter=1024, norm=True, dropout=1e-1, nonlinear="relu", causal=False, eps=EPS):
        super().__init__()
        self.intra_transformer = IntraTransformer(d_intra, num_layers=num_layers_intra, num_heads=num_heads_intra, d_ff=d_ff_intra,
                                                  norm=norm, dropout=dropout, nonlinear=nonlinear, eps=eps)
        self.inter_transformer = InterTransformer(d_inter, num_layers=num_layers_inter, num_heads=num_heads_inter, d_ff=d_ff_inter,
                                                  norm=norm, dropout=dropout, nonlinear=nonlinear, causal=causal, eps=eps)

    def forward(self, input):
        return self.inter_transformer(self.intra_transformer(input))


class _ChunkPathEncoder(nn.Module):
    """A transformer-encoder stack along one of the two chunk axes with a residual connection around it.
    Its input is x + (x + code): the reference adds the OUTPUT of its positional-encoding module -- which already contains
    x -- back onto x (sepformer.py:470-472, 512-514), and a drop-in keeps that."""
    SEQ_AXIS = None         # 3: along the chunk (intra), 2: across chunks (inter)

    def __init__(self, num_features, num_layers, num_heads, d_ff, norm, nonlinear, dropout, norm_first, default_norm, eps):
        super().__init__()
        self.num_features = num_features
        if isinstance(norm, int):                      # True / False (bool is an int) -> gLN, or cLN for a causal inter path
            final = LayerNormWrapper(default_norm, num_features, causal=False, batch_first=False, eps=eps) if norm else None
        else:                                          # a norm name
            final = LayerNormWrapper(norm, num_features, causal=False, batch_first=False, eps=eps)
        self.positional_encoding = PositionalEncoding(num_features, batch_first=False)
        layer = nn.TransformerEncoderLayer(num_features, num_heads, d_ff, dropout=dropout, activation=nonlinear, layer_norm_eps=eps,
                                           batch_first=False, norm_first=norm_first)
        self.transformer = nn.TransformerEncoder(layer, num_layers=num_layers, norm=final, enable_nested_tensor=False)

    def _tokens_ok(self, input):
        """the token-major route: the kernels' tensors, the layout pair's grid limits, no final norm or a gLN that sep_gln_tokens_* takes
        (the causal inter path's cLN stays on the strided route)"""
        from sepkernels.functional import takes, token_gln_ok
        B, C, S, K = input.size()
        if not takes(input) or max(S, K) > 65535 or B * ((C + 31) // 32) > 65535:
            return False
        final = self.transformer.norm
        return final is None or token_gln_ok(input.new_empty(1, 1, C), final.norm1d)

    @staticmethod
    def _layer_tokens(layer, x):
        """one nn.TransformerEncoderLayer (torch/nn/modules/transformer.py: post-norm, or pre-norm with norm_first) on batch-first rows
        (N, L, C): the four dense layers on csrc/linear.hip -- hipBLASLt's picks for these fp32 shapes (32 K tokens x 256 x 1024) run at a
        third of the fp32 MFMA rate, profiles/r05zc_sepformer_kernel_stats.md --, the core on csrc/attn.hip (sep_attn_*) with the layer's
        attention dropout."""
        from sepkernels.functional import attention_core, dense_apply, relu_dropout, residual_layer_norm
        sa = layer.self_attn
        N, L, C = x.shape
        h = sa.num_heads

        def attend(u):
            qkv = dense_apply(u, sa.in_proj_weight, sa.in_proj_bias).view(N, L, 3, h, C // h)
            o = attention_core(qkv, sa.dropout if layer.training else 0.0)
            return dense_apply(o, sa.out_proj.weight, sa.out_proj.bias)

        def feed(u):
            if _ff_on_conv_kernels(layer, u):
                return _feed_forward_channel_major(layer, u)
            f = dense_apply(u, layer.linear1.weight, layer.linear1.bias)
            f = relu_dropout(f, rate(layer.dropout)) if layer.activation is F.relu else layer.dropout(layer.activation(f))
            return dense_apply(f, layer.linear2.weight, layer.linear2.bias)

        def rate(drop):
            return drop.p if layer.training else 0.0

        if layer.norm_first:
            x = x + layer.dropout1(attend(residual_layer_norm(x, None, layer.norm1)))
            return x + layer.dropout2(feed(residual_layer_norm(x, None, layer.norm2)))
        # post-norm (the reference's layers): dropout of the branch, residual sum and layer norm in one pass each way (sep_rownorm_*)
        x = residual_layer_norm(x, attend(x), layer.norm1, rate(layer.dropout1))
        return residual_layer_norm(x, feed(x), layer.norm2, rate(layer.dropout2))

    def _forward_tokens(self, input):
        """forward with the features innermost: one tiled transpose in (sep_chunk_to_tokens), the stack on (sequences, steps, C) rows, the
        final gLN on sep_gln_tokens_*, one tiled transpose out"""
        from sepkernels.functional import ChunkToTokensFn, TokensToChunkFn, TokenGLNFn
        B, C, S, K = input.size()
        inter = self.SEQ_AXIS == 2
        t = ChunkToTokensFn.apply(input, inter)                                   # (B*S, K, C) | (B*K, S, C)
        pe = self.positional_encoding
        x = t + pe.dropout(t + pe.positional_encoding[:t.size(1), 0])
        for layer in self.transformer.layers:
            x = self._layer_tokens(layer, x)
        final = self.transformer.norm
        if final is not None:
            gn = final.norm1d
            x = TokenGLNFn.apply(x, gn.norm.weight, gn.norm.bias, gn.eps)
        return TokensToChunkFn.apply(x, (B, C, S, K), inter) + input

    def forward(self, input):
        """(batch_size, num_features, S, chunk_size) -> same shape"""
        if self._tokens_ok(input):
            return self._forward_tokens(input)
        B, C, S, K = input.size()
        if self.SEQ_AXIS == 3:
            x = input.permute(3, 0, 2, 1).reshape(K, B * S, C)
            y = self.transformer(x + self.positional_encoding(x)).view(K, B, S, C).permute(1, 3, 2, 0)
        else:
            x = input.permute(2, 0, 3, 1).reshape(S, B * K, C)
            y = self.transformer(x + self.positional_encoding(x)).view(S, B, K, C).permute(1, 3, 0, 2)
        return y + input


def _ff_on_conv_kernels(layer, u):
    """the feed-forward pair on the 1x1-convolution kernels?  Widths in multiples of 128 (their tiles), the two-part fp16 arithmetic, the
    device (the CPU stand-in of the tests takes the route too, so that the fixtures exercise it)"""
    import sepkernels
    from sepkernels.functional import takes
    C, Fd = layer.linear1.in_features, layer.linear1.out_features
    on_device = sepkernels.backend().name == "hip"
    return (takes(u) and C % 128 == 0 and Fd % 128 == 0 and (not on_device or sepkernels.gemm_arith() == sepkernels.ARITH_F16X3)
            and layer.linear1.bias is not None and layer.linear2.bias is not None)


def _feed_forward_channel_major(layer, u):
    """linear2(dropout(activation(linear1(u)))) of an nn.TransformerEncoderLayer on tokens u (N, L, C), computed CHANNEL-MAJOR on the 1x1-
    convolution kernels (sep_pw_gemm / sep_pw_wgrad through PaddedPointwiseFn): all N L tokens are the frames of one (C, frames) sample.
    At SepFormer's sizes (33 K tokens, 256 <-> 1024) the two products are compute-bound; the dense kernel multiplies on the fp32 MFMA
    (210 + 241 us forward, 444 + 353 backward), its two-part fp16 form is load-bound at the 128 x 128 tile
    (profiles/r05zy_linear16_experiment.md), the convolution kernels' 256-row tiles are not -- at the price of two tiled transposes (34 MB
    each) around the pair.  ReLU without dropout rides the second product's prologue (PReLU with slope 0)."""
    from sepkernels.functional import ChunkToTokensFn, TokensToChunkFn, PaddedPointwiseFn, relu_dropout
    from sepkernels import net as _net
    N, L, C = u.shape
    ntok = N * L
    ldt = -(-ntok // 128) * 128
    t = F.pad(u.reshape(ntok, C), (0, 0, 0, ldt - ntok))                                          # rows of zeros up to the kernels' frame stride
    x = TokensToChunkFn.apply(t.view(1, ldt, C), (1, C, 1, ldt), False).view(1, C, ldt)
    w1, w2 = layer.linear1.weight, layer.linear2.weight
    wa = _net._weights_amax({"w1": w1, "w2": w2})
    h = PaddedPointwiseFn.apply(x, ntok, w1, layer.linear1.bias, None, wa)
    drop = layer.training and layer.dropout.p > 0
    if layer.activation is F.relu and not drop:
        y = PaddedPointwiseFn.apply(h, ntok, w2, layer.linear2.bias, u.new_zeros(1), wa)
    elif layer.activation is F.relu:
        y = PaddedPointwiseFn.apply(relu_dropout(h, layer.dropout.p), ntok, w2, layer.linear2.bias, None, wa)     # one pass each way: sep_relu_drop_*
    else:
        y = PaddedPointwiseFn.apply(layer.dropout(layer.activation(h)), ntok, w2, layer.linear2.bias, None, wa)
    return ChunkToTokensFn.apply(y.view(1, C, 1, ldt), False).view(ldt, C)[:ntok].view(N, L, C)


class IntraTransformer(_ChunkPathEncoder):
    SEQ_AXIS = 3

    def __init__(self, num_features, num_layers=8, num_heads=8, d_ff=1024, norm=True, nonlinear="relu", dropout=1e-1, norm_first=False, eps=EPS):
        super().__init__(num_features, num_layers, num_heads, d_ff, norm, nonlinear, dropout, norm_first, "gLN", eps)


class InterTransformer(_ChunkPathEncoder):
    SEQ_AXIS = 2

    def __init__(self, num_features, num_layers=8, num_heads=8, d_ff=1024, norm=True, nonlinear="relu", dropout=1e-1, causal=False,
                 norm_first=False, eps=EPS):
        super().__init__(num_features, num_layers, num_heads, d_ff, norm, nonlinear, dropout, norm_first, "cLN" if causal else "gLN", eps)


class LayerNormWrapper(nn.Module):
    """a (batch_size, C, T) layer norm of the TasNet family on (T, batch_size, C) -- or (batch_size, T, C) if batch_first"""

    def __init__(self, norm_name, num_features, causal=False, batch_first=False, eps=EPS):
        super().__init__()
        self.batch_first = batch_first
        extra = {"n_dims": 1} if norm_name in ("BN", "batch", "batch_norm") else {}
        self.norm1d = choose_layer_norm(norm_name, num_features, causal=causal, eps=eps, **extra)

    def forward(self, input):
        if self.batch_first:
            return self.norm1d(input.permute(0, 2, 1).contiguous()).permute(0, 2, 1).contiguous()
        return self.norm1d(input.permute(1, 2, 0).contiguous()).permute(2, 0, 1).contiguous()


from sepkernels.shadowed import fall_through as _fall_through      # names of the reference's same-named module this tree does not define

__getattr__ = _fall_through(__name__, __file__)
