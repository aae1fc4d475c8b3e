"""choose_filterbank (reference src/utils/filterbank.py:5-46) for the learned (trainable) basis pair that the
MI355X path covers.  Fourier / pseudo-inverse / gated bases are outside SURVEY.md section 8 and fail loudly."""
from models.filterbank import Encoder, Decoder

EPS = 1e-12


def choose_filterbank(hidden_channels, kernel_size, stride=None, enc_basis="trainable", dec_basis="trainable", **kwargs):
    in_channels = kwargs.get("in_channels") or 1
    if enc_basis != "trainable":
        raise NotImplementedError("Not support {} for encoder (MI355X path implements the trainable basis only)".format(enc_basis))
    if dec_basis != "trainable":
        raise NotImplementedError("Not support {} for decoder (MI355X path implements the trainable basis only)".format(dec_basis))
    encoder = Encoder(in_channels, hidden_channels, kernel_size, stride=stride, nonlinear=kwargs.get("enc_nonlinear"))
    decoder = Decoder(hidden_channels, in_channels, kernel_size, stride=stride)
    return encoder, decoder
