"""Sinusoidal position code of the SepFormer transformers (reference src/models/transformer.py:7-44): a buffer of
(max_len, num_features) values, sine and cosine of one frequency in ADJACENT features; `forward` returns input + code."""
import torch
import torch.nn as nn


class PositionalEncoding(nn.Module):
    def __init__(self, num_features, dropout=0, max_len=5000, base=10000, batch_first=False):
        super().__init__()
        self.batch_first = batch_first
        exponent = torch.arange(0, num_features, 2) / num_features
        angle = torch.arange(max_len).unsqueeze(dim=1) / base ** exponent.unsqueeze(dim=0)          # (max_len, num_features/2)
        table = torch.stack([torch.sin(angle), torch.cos(angle)], dim=-1).view(max_len, num_features)
        self.register_buffer("positional_encoding", table if batch_first else table.view(max_len, 1, num_features))
        self.dropout = nn.Dropout(p=dropout)

    def forward(self, input):
        """(T, batch_size, num_features), or (batch_size, T, num_features) if batch_first -> input + code, same shape"""
        if self.batch_first:
            return self.dropout(input + self.positional_encoding[:, :input.size(1)])
        return self.dropout(input + self.positional_encoding[:input.size(0)])


from sepkernels.shadowed import fall_through as _fall_through      # names of the reference's same-named module this tree does not define

__getattr__ = _fall_through(__name__, __file__)
