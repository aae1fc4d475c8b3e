"""
GALR blocks (globally attentive, locally recurrent): module tree / parameter names of reference src/models/galr.py:9-226.
Locally: the intra-chunk bi-LSTM path of DPRNN (models/dprnn.py, LSTM sweep kernels + gLN kernels).  Globally: self-attention
across chunks for every position of a chunk -- optionally of a chunk squeezed from K to Q positions by a Linear map --
between a layer norm along the channels and a gLN / cLN, with sinusoidal position codes added in front.
"""
import torch
import torch.nn as nn

from utils.tasnet import choose_layer_norm
from models.dprnn import IntraChunkRNN as LocallyRecurrentBlock

EPS = 1e-12


class GALR(nn.Module):
    def __init__(self, num_features, hidden_channels, num_blocks=6, num_heads=8, norm=True, dropout=1e-1, low_dimension=True,
                 causal=False, eps=EPS, **kwargs):
        super().__init__()
        self.net = nn.Sequential(*[GALRBlock(num_features, hidden_channels, num_heads=num_heads, norm=norm, dropout=dropout,
                                             low_dimension=low_dimension, causal=causal, eps=eps, **kwargs) for _ in range(num_blocks)])

    def forward(self, input):
        """(batch_size, num_features, S, chunk_size) -> same shape"""
        return self.net(input)


class GALRBlock(nn.Module):
    def __init__(self, num_features, hidden_channels, num_heads=8, causal=False, norm=True, dropout=1e-1, low_dimension=True, eps=EPS, **kwargs):
        super().__init__()
        self.intra_chunk_block = LocallyRecurrentBlock(num_features, hidden_channels=hidden_channels, norm=norm, eps=eps)
        if low_dimension:
            self.inter_chunk_block = LowDimensionGloballyAttentiveBlock(num_features, chunk_size=kwargs["chunk_size"],
                                                                        down_chunk_size=kwargs["down_chunk_size"], num_heads=num_heads,
                                                                        causal=causal, norm=norm, dropout=dropout, eps=eps)
        else:
            self.inter_chunk_block = GloballyAttentiveBlock(num_features, num_heads=num_heads, causal=causal, norm=norm, dropout=dropout, eps=eps)

    def forward(self, input):
        return self.inter_chunk_block(self.intra_chunk_block(input))


_POSITION_CODES = {}       # (S, Q, C, device, dtype) -> code, ONE bounded LRU for all blocks, see GloballyAttentiveBlockBase._position_code
_POSITION_CODES_MAX = 16


class GloballyAttentiveBlockBase(nn.Module):
    def positional_encoding(self, length, dimension, base=10000):
        """(length, dimension): [sin(p / base^(i/dimension)) for i < dimension/2 | the cosines]  (halves, not interleaved)"""
        assert dimension % 2 == 0, "dimension is expected even number but given odd number."
        position = torch.arange(length).unsqueeze(dim=1)
        index = (torch.arange(dimension // 2) / dimension).unsqueeze(dim=0)
        angle = position / base ** index
        return torch.cat([torch.sin(angle), torch.cos(angle)], dim=1)

    def _position_code(self, S, Q, C, like, tokens=False):
        """(C, S, Q) code on the device / in the dtype of `like`, built once per shape: forming it on the host and copying it
        over in every forward would stall the launch queue once per block"""
        # The code depends on the shape only, not on the block: one module-level LRU (not an attribute: deepcopy / state_dict must not
        # carry device tensors along; not keyed by id(self): nn.DataParallel replicas are fresh objects every forward and destroyed
        # models would leave their entries behind).
        cache = _POSITION_CODES
        key = (S, Q, C, like.device, like.dtype, bool(tokens))           # tokens: the same code laid out (Q, S, C) for token-major rows
        code = cache.pop(key, None)
        if code is None:
            if len(cache) >= _POSITION_CODES_MAX:            # a few shapes stay resident (training length, validation utterances): least recently used goes
                cache.pop(next(iter(cache)))
            code = self.positional_encoding(length=S * Q, dimension=C).t().reshape(C, S, Q).to(device=like.device, dtype=like.dtype)
            if tokens:
                code = code.permute(2, 1, 0).contiguous()
        cache[key] = code                                    # (re-)inserted last = most recently used
        return code

    def _attend_tokens(self, x):
        """_attend with the features innermost from end to end: (B, C, S, Q) -> one sequence per (b, q) as token-major rows (B*Q, S, C)
        (sep_chunk_to_tokens), the channel norm on the rows as they are, the attention batch-first with its projections on csrc/linear.hip and its core on csrc/attn.hip,
        gLN over a sample's (Q, S, C) block on sep_gln_tokens_*, back (sep_tokens_to_chunk): two tiled transposes instead of six strided
        copies per block and direction."""
        from sepkernels.functional import attention_core, ChunkToTokensFn, TokensToChunkFn, TokenGLNFn, dense_apply, residual_layer_norm
        B, C, S, Q = x.size()
        t = ChunkToTokensFn.apply(x, True)                                       # (B*Q, S, C)
        if self.norm:
            t = residual_layer_norm(t, None, self.norm2d_in.norm)                 # the channel norm on sep_rownorm_*
        seq = (t.view(B, Q, S, C) + self._position_code(S, Q, C, x, tokens=True)).view(B * Q, S, C)
        mha = self.multihead_attn
        h = mha.num_heads
        qkv = dense_apply(seq, mha.in_proj_weight, mha.in_proj_bias).view(B * Q, S, 3, h, C // h)
        y = dense_apply(attention_core(qkv), mha.out_proj.weight, mha.out_proj.bias)
        if self.dropout:
            y = self.dropout1d(y)
        y = y + seq
        if self.norm:
            gn = self.norm2d_out
            y = TokenGLNFn.apply(y.view(B, Q * S, C), gn.norm.weight, gn.norm.bias, gn.eps).view(B * Q, S, C)
        return TokensToChunkFn.apply(y, (B, C, S, Q), True)

    def _tokens_ok(self, x):
        """the token-major route: the kernels' tensors, the layout pair's grid limits, and -- with norms -- gLN over features that
        sep_gln_tokens_* takes (the causal cLN stays on the strided route)"""
        from sepkernels.functional import takes, token_gln_ok
        B, C, S, Q = x.size()
        if not takes(x) or S > 65535 or B * ((C + 31) // 32) > 65535:
            return False
        return not self.norm or token_gln_ok(x.new_empty(1, 1, C), self.norm2d_out)

    def _attend(self, x):
        """x (batch_size, num_features, S, Q): [channel norm ->] + position code -> attention over S -> [dropout] + its input
        -> [gLN / cLN].  The position code runs over the FLATTENED (S, Q) index, as the reference's does."""
        if self._tokens_ok(x):
            return self._attend_tokens(x)
        B, C, S, Q = x.size()
        if self.norm:
            x = self.norm2d_in(x)
        code = self._position_code(S, Q, C, x)
        seq = (x + code).permute(2, 0, 3, 1).reshape(S, B * Q, C)
        y = self.multihead_attn(seq, seq, seq, need_weights=False)[0]
        if self.dropout:
            y = self.dropout1d(y)
        y = (y + seq).view(S, B, Q, C).permute(1, 3, 0, 2).contiguous()
        return self.norm2d_out(y) if self.norm else y

    def _make_core(self, num_features, num_heads, causal, norm, dropout, eps):
        self.norm = norm
        if norm:
            self.norm2d_in = LayerNormAlongChannel(num_features, eps=eps)
        self.multihead_attn = nn.MultiheadAttention(num_features, num_heads)
        self.dropout = dropout is not None
        if self.dropout:
            self.dropout1d = nn.Dropout(p=dropout)
        if norm:
            self.norm2d_out = choose_layer_norm("cLN" if causal else "gLN", num_features, causal=causal, eps=eps)


class GloballyAttentiveBlock(GloballyAttentiveBlockBase):
    def __init__(self, num_features, num_heads=8, causal=False, norm=True, dropout=1e-1, eps=EPS):
        super().__init__()
        self._make_core(num_features, num_heads, causal, norm, dropout, eps)

    def forward(self, input):
        """(batch_size, num_features, S, K) -> same shape"""
        return self._attend(input) + input


class LowDimensionGloballyAttentiveBlock(GloballyAttentiveBlockBase):
    def __init__(self, num_features, chunk_size=100, down_chunk_size=32, num_heads=8, causal=False, norm=True, dropout=1e-1, eps=EPS):
        super().__init__()
        self.down_chunk_size = down_chunk_size
        self.fc_map = nn.Linear(chunk_size, down_chunk_size)          # registration order of the reference: fc_map, core, fc_inv
        self._make_core(num_features, num_heads, causal, norm, dropout, eps)
        self.fc_inv = nn.Linear(down_chunk_size, chunk_size)

    @staticmethod
    def _along_chunk(x, fc):
        """nn.Linear along the chunk axis (100 -> 32 -> 100 at the recipe's sizes).  The library's pick for 20 736 rows x 100 x 32 in fp32 runs
        82 us (profiles/r05zj_galrnet_kernel_stats.md: 1 ms of the step for 0.13 GFLOP); on csrc/linear.hip with both widths padded with
        zeros to its multiples of 64 the product and the two small copies take a third of that."""
        from sepkernels.functional import DenseFn, takes
        K_in, N_out = fc.in_features, fc.out_features
        if not takes(x) or (K_in % 64 == 0 and N_out % 64 == 0) or max(K_in, N_out) > 256:
            return fc(x)
        Kp, Np = -(-K_in // 64) * 64, -(-N_out // 64) * 64
        F = torch.nn.functional
        y = DenseFn.apply(F.pad(x, (0, Kp - K_in)), F.pad(fc.weight, (0, Kp - K_in, 0, Np - N_out)), F.pad(fc.bias, (0, Np - N_out)))
        return y[..., :N_out]

    def forward(self, input):
        """(batch_size, num_features, S, K) -> same shape; attention runs on K squeezed to Q = down_chunk_size positions"""
        return self._along_chunk(self._attend(self._along_chunk(input, self.fc_map)), self.fc_inv) + input


class LayerNormAlongChannel(nn.Module):
    """nn.LayerNorm over the channel axis of (batch_size, num_features, *)"""

    def __init__(self, num_features, eps=EPS):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.norm = nn.LayerNorm(num_features, eps=eps)

    def forward(self, input):
        return self.norm(input.movedim(1, -1)).movedim(-1, 1).contiguous()

    def __repr__(self):
        return "{}({}, eps={})".format(self.__class__.__name__, self.num_features, self.eps)


from sepkernels.shadowed import fall_through as _fall_through      # names of the reference's same-named module this tree does not define

__getattr__ = _fall_through(__name__, __file__)
