"""choose_filterbank (reference src/utils/filterbank.py:5-66): the encoder / decoder pair for a basis name.  The learned
(`trainable`) pair is what the fused MI355X path runs; Fourier, pseudo-inverse and gated bases are torch compositions used by
ConvTasNet's fallback path (models/filterbank.py)."""
from models.filterbank import Decoder, Encoder, FourierDecoder, FourierEncoder, GatedEncoder, PinvDecoder

EPS = 1e-12

_FOURIER = {"Fourier": (False, False), "trainableFourier": (True, False), "trainableFourierTrainablePhase": (True, True)}   # name -> (trainable, trainable_phase)


def assert_monoral(in_channels):
    assert in_channels == 1, "`in_channels` is expected 1, but given {}.".format(in_channels)


def compute_valid_basis(hidden_channels, onesided=True, return_complex=True):
    """FFT size whose encoder output has `hidden_channels` feature rows"""
    if return_complex:
        if not onesided:
            return hidden_channels
        assert hidden_channels % 2 == 1, "`hidden_channels` is expected odd."
        return 2 * (hidden_channels - 1)
    assert hidden_channels % 2 == 0, "`hidden_channels` is expected even."
    return 2 * (hidden_channels // 2 - 1) if onesided else hidden_channels // 2


def _fourier_args(hidden_channels, kwargs):
    onesided, return_complex = bool(kwargs["enc_onesided"]), bool(kwargs["enc_return_complex"])
    return compute_valid_basis(hidden_channels, onesided=onesided, return_complex=return_complex), onesided, return_complex


def choose_filterbank(hidden_channels, kernel_size, stride=None, enc_basis="trainable", dec_basis="trainable", **kwargs):
    in_channels = kwargs.get("in_channels") or 1
    if enc_basis == "trainable":
        encoder = Encoder(in_channels, hidden_channels, kernel_size, stride=stride, nonlinear=None if dec_basis == "pinv" else kwargs["enc_nonlinear"])
    elif enc_basis in _FOURIER:
        assert_monoral(in_channels)
        n_basis, onesided, return_complex = _fourier_args(hidden_channels, kwargs)
        encoder = FourierEncoder(n_basis, kernel_size, stride=stride, window_fn=kwargs["window_fn"], trainable=_FOURIER[enc_basis][0],
                                 trainable_phase=_FOURIER[enc_basis][1], onesided=onesided, return_complex=return_complex)
    elif enc_basis == "trainableGated":
        encoder = GatedEncoder(in_channels, hidden_channels, kernel_size=kernel_size, stride=stride, eps=kwargs.get("eps") or EPS)
    else:
        raise NotImplementedError("Not support {} for encoder".format(enc_basis))

    if dec_basis == "trainable":
        decoder = Decoder(hidden_channels, in_channels, kernel_size, stride=stride)
    elif dec_basis in _FOURIER:
        assert_monoral(in_channels)
        n_basis, onesided, _ = _fourier_args(hidden_channels, kwargs)
        decoder = FourierDecoder(n_basis, kernel_size, stride=stride, window_fn=kwargs["window_fn"], trainable=_FOURIER[dec_basis][0],
                                 trainable_phase=_FOURIER[dec_basis][1], onesided=onesided)
    elif dec_basis == "pinv":
        if enc_basis not in ("trainable", "trainableFourier", "trainableFourierTrainablePhase"):
            raise NotImplementedError("Not support {} for decoder".format(dec_basis))
        assert_monoral(in_channels)
        decoder = PinvDecoder(encoder)
    else:
        raise NotImplementedError("Not support {} for decoder".format(dec_basis))
    return encoder, decoder


from sepkernels.shadowed import fall_through as _fall_through      # names of the reference's same-named module this tree does not define

__getattr__ = _fall_through(__name__, __file__)
