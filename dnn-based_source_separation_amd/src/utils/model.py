"""Factories kept API-compatible with reference src/utils/model.py:3-31 (thin; no arithmetic lives here)."""
import torch.nn as nn


def choose_nonlinear(name, **kwargs):
    table = {"relu": nn.ReLU, "sigmoid": nn.Sigmoid, "tanh": nn.Tanh, "leaky-relu": nn.LeakyReLU, "gelu": nn.GELU}
    if name == "softmax":
        assert "dim" in kwargs, "dim is expected for softmax."
        return nn.Softmax(**kwargs)
    if name not in table:
        raise NotImplementedError("Invalid nonlinear function is specified. Choose 'relu' instead of {}.".format(name))
    return table[name]()


def choose_rnn(name, **kwargs):
    table = {"rnn": nn.RNN, "lstm": nn.LSTM, "gru": nn.GRU}
    if name not in table:
        raise NotImplementedError("Invalid RNN is specified. Choose 'rnn', 'lstm', or 'gru' instead of {}.".format(name))
    return table[name](**kwargs)


from sepkernels.shadowed import fall_through as _fall_through      # names of the reference's same-named module this tree does not define

__getattr__ = _fall_through(__name__, __file__)
