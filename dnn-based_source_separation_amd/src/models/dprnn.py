"""
Dual-path RNN blocks: module tree / parameter names of reference src/models/dprnn.py:9-148.

The recurrences run on libsepkernels (`sep_lstm_fwd/bwd`: one persistent workgroup per 16 sequences, W_hh held in
registers as MFMA fragments, one barrier per step); `nn.LSTM` is only the parameter container, so checkpoints keep the
reference's keys (`rnn.weight_ih_l0`, `rnn.weight_hh_l0_reverse`, ...).  Hidden sizes the sweep kernels do not cover run on
torch's own LSTM.  The input projection, the Linear after the RNN
and the weight-gradient products are plain library GEMMs; the global layer norm after every path is the libsepkernels
gLN; layout changes are torch views/permutes.
"""
import torch.nn as nn

from sepkernels.functional import ChunkToTokensFn, TokensToChunkFn, linear_apply, lstm_apply, takes

from modules.norm import CumulativeLayerNorm1d
from utils.model import choose_rnn
from utils.tasnet import choose_layer_norm

EPS = 1e-12


class DPRNN(nn.Module):
    def __init__(self, num_features, hidden_channels, num_blocks=6, norm=True, causal=False, rnn_type="lstm", eps=EPS):
        super().__init__()
        self.net = nn.Sequential(*[DPRNNBlock(num_features, hidden_channels, norm=norm, causal=causal, rnn_type=rnn_type, eps=eps)
                                   for _ in range(num_blocks)])

    def forward(self, input):
        """(B, F, S, K) -> (B, F, S, K)"""
        return self.net(input)


class DPRNNBlock(nn.Module):
    def __init__(self, num_features, hidden_channels, causal, norm=True, rnn_type="lstm", eps=EPS):
        super().__init__()
        self.intra_chunk_block = IntraChunkRNN(num_features, hidden_channels, norm=norm, rnn_type=rnn_type, eps=eps)
        self.inter_chunk_block = InterChunkRNN(num_features, hidden_channels, norm=norm, causal=causal, rnn_type=rnn_type, eps=eps)

    def forward(self, input):
        return self.inter_chunk_block(self.intra_chunk_block(input))


class _PathRNN(nn.Module):
    """Shared body of the intra- and inter-chunk paths: sequence axis -> bi-LSTM -> Linear -> gLN -> + input."""

    def __init__(self, num_features, hidden_channels, bidirectional, norm_name, norm, rnn_type, causal, eps):
        super().__init__()
        if rnn_type != "lstm":
            raise NotImplementedError("Not support {}.".format(rnn_type))            # as the reference (dprnn.py:59-62, 110-113)
        self.num_features, self.hidden_channels = num_features, hidden_channels
        self.norm = norm
        self.rnn = choose_rnn(rnn_type, input_size=num_features, hidden_size=hidden_channels, batch_first=True, bidirectional=bidirectional)
        self.fc = nn.Linear((2 if bidirectional else 1) * hidden_channels, num_features)
        if norm:
            self.norm1d = choose_layer_norm(norm_name, num_features, causal=causal, eps=eps)

    def _run(self, input, seq_axis):
        """input (B, F, S, K); seq_axis 3 -> recur over K for every (b, s); 2 -> over S for every (b, k)."""
        B, F, S, K = input.shape
        inter = seq_axis == 2
        # gLN's statistics run over all of (F, S*K) and its gain / bias are per feature, so the frame ORDER behind the Linear does not
        # matter to it: the rows go straight back to (B, F, S, K) and the inter-chunk path needs no second permutation.  cLN (the causal
        # inter-chunk path) accumulates along the sequence axis and keeps the reference's order below.
        order_free = not self.norm or not isinstance(self.norm1d, CumulativeLayerNorm1d)
        if order_free and S <= 65535 and B * ((F + 31) // 32) <= 65535 and takes(input):     # grid limits of sep_chunk_to_tokens / sep_tokens_to_chunk
            x = ChunkToTokensFn.apply(input, inter)                  # tiled transposes (csrc/linear.hip) instead of strided copies
            x = lstm_apply(x, self.rnn)          # the sweep kernels for 16 / 32 / 64 / 128 units, torch's LSTM otherwise
            x = linear_apply(x, self.fc)                             # csrc/linear.hip for feature counts % 64 == 0
            x = TokensToChunkFn.apply(x, (B, F, S, K), inter)
            if self.norm:
                x = self.norm1d(x.view(B, F, S * K)).view(B, F, S, K)
            return x + input
        if seq_axis == 3:
            x = input.permute(0, 2, 3, 1).reshape(B * S, K, F)
        else:
            x = input.permute(0, 3, 2, 1).reshape(B * K, S, F)
        x = lstm_apply(x, self.rnn)
        x = linear_apply(x, self.fc)                             # (B*S, K, F) or (B*K, S, F)
        x = x.reshape(B, S * K, F).permute(0, 2, 1).contiguous()  # (B, F, S*K) [or (B, F, K*S)]
        if self.norm:
            x = self.norm1d(x)                                   # statistics over all of (F, S*K): order-free
        if seq_axis == 3:
            x = x.view(B, F, S, K)
        else:
            x = x.view(B, F, K, S).permute(0, 1, 3, 2)
        return x + input


class IntraChunkRNN(_PathRNN):
    def __init__(self, num_features, hidden_channels, norm=True, rnn_type="lstm", eps=EPS):
        super().__init__(num_features, hidden_channels, True, "gLN", norm, rnn_type, False, eps)

    def forward(self, input):
        return self._run(input, 3)


class InterChunkRNN(_PathRNN):
    def __init__(self, num_features, hidden_channels, causal, norm=True, rnn_type="lstm", eps=EPS):
        if rnn_type == "lstm":
            # the reference builds this path's LSTM twice (dprnn.py:108-121: once unconditionally, then again by `causal`) and keeps the
            # second: the first one's draws from the global generator are part of "same seed -> same initial weights"
            choose_rnn(rnn_type, input_size=num_features, hidden_size=hidden_channels, batch_first=True, bidirectional=True)
        super().__init__(num_features, hidden_channels, not causal, "cLN" if causal else "gLN", norm, rnn_type, causal, eps)

    def forward(self, input):
        return self._run(input, 2)


from sepkernels.shadowed import fall_through as _fall_through      # names of the reference's same-named module this tree does not define

__getattr__ = _fall_through(__name__, __file__)
