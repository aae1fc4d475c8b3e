"""choose_layer_norm with the reference's names and error behaviour (reference src/utils/tasnet.py:14-32)."""
import torch.nn as nn

from modules.norm import GlobalLayerNorm, CumulativeLayerNorm1d

EPS = 1e-12


def choose_layer_norm(name, num_features, causal=False, eps=EPS, **kwargs):
    if name == "cLN":
        return CumulativeLayerNorm1d(num_features, eps=eps)
    if name == "gLN":
        if causal:
            raise ValueError("Global Layer Normalization is NOT causal.")
        return GlobalLayerNorm(num_features, eps=eps)
    if name in ["BN", "batch", "batch_norm"]:
        n_dims = kwargs.get("n_dims") or 1
        if n_dims == 1:
            return nn.BatchNorm1d(num_features, eps=eps)
        if n_dims == 2:
            return nn.BatchNorm2d(num_features, eps=eps)
        raise NotImplementedError("n_dims is expected 1 or 2, but give {}.".format(n_dims))
    raise NotImplementedError("Not support {} layer normalization.".format(name))
