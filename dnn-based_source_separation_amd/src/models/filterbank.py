"""
Filterbanks of the TasNet family.

Learned encoder / decoder of Conv-TasNet: API and state_dict keys of reference src/models/filterbank.py:205-251
(`Encoder.conv1d.weight (N, in_channels, L)`, `Decoder.conv_transpose1d.weight (N, out_channels, L)`, no bias,
default nn.Conv1d / nn.ConvTranspose1d initialisation).  Inside a fused ConvTasNet both run as part of the kernel sequence
(sep_encoder_fwd / sep_decoder_fwd); a stand-alone call without gradients runs those kernels, one that needs gradients the
torch convolution (the SURVEY.md 8b composition).

Fourier (fixed / trainable frequencies / trainable phase), pseudo-inverse and gated bases (reference :12-203, 253-346) are
outside the fused family: written here from their algebra as compositions of torch operations, with the reference's
parameter names (`frequency`, `time_seq`, `window` / `optimal_window`, `phase`, `conv1d_U/V`), for the fallback path of
ConvTasNet.  Analysis basis of bin f: window[n] * exp(-i (omega_f n + phi_f)); synthesis basis: optimal_window[n] *
exp(+i (omega_f n + phi_f)) / n_basis over ALL n_basis bins (a one-sided input is mirrored to its conjugate half first).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

import sepkernels


def _round_up(a, b):
    return (a + b - 1) // b * b


class Encoder(nn.Module):
    def __init__(self, in_channels, n_basis, kernel_size=16, stride=8, nonlinear=None):
        super().__init__()
        self.kernel_size, self.stride = kernel_size, stride
        self.conv1d = nn.Conv1d(in_channels, n_basis, kernel_size=kernel_size, stride=stride, bias=False)   # parameter holder
        if nonlinear is not None and nonlinear != "":
            if nonlinear != "relu":
                raise NotImplementedError("Not support {}".format(nonlinear))
            self.nonlinear1d = nn.ReLU()
            self.nonlinear = True
        else:
            self.nonlinear = False

    def forward(self, input):
        """input (B, in_channels, T) -> (B, n_basis, (T-L)//S+1)"""
        if (torch.is_grad_enabled() and (input.requires_grad or self.conv1d.weight.requires_grad)) or (not input.is_cuda and sepkernels.backend().name == "hip"):
            w = F.conv1d(input, self.conv1d.weight, stride=self.stride)          # differentiable composition
            return torch.relu(w) if self.nonlinear else w
        K = sepkernels.backend()
        x = input.contiguous()
        B, Cin, T = x.shape
        N, L, S = self.conv1d.out_channels, self.kernel_size, self.stride
        nF = (T - L) // S + 1
        ldt = _round_up(nF, 128)
        w = torch.empty(B, N, ldt, device=x.device, dtype=x.dtype)
        stats = torch.zeros(B, sepkernels.STATS_SLOTS, 2, device=x.device, dtype=torch.float64)
        K.encoder_fwd(x, self.conv1d.weight, w, stats, B, Cin, T, N, L, S, nF, ldt, 0, self.nonlinear)
        out = torch.empty(B, N, nF, device=x.device, dtype=x.dtype)
        K.repack(w, ldt, out, nF, B * N, nF)
        return out

    def get_basis(self):
        return self.conv1d.weight


class Decoder(nn.Module):
    def __init__(self, n_basis, out_channels, kernel_size=16, stride=8):
        super().__init__()
        self.kernel_size, self.stride = kernel_size, stride
        self.conv_transpose1d = nn.ConvTranspose1d(n_basis, out_channels, kernel_size=kernel_size, stride=stride, bias=False)

    def forward(self, input):
        """input (B', n_basis, F) -> (B', out_channels, (F-1)*S+L)"""
        if (torch.is_grad_enabled() and (input.requires_grad or self.conv_transpose1d.weight.requires_grad)) or (not input.is_cuda and sepkernels.backend().name == "hip"):
            return F.conv_transpose1d(input, self.conv_transpose1d.weight, stride=self.stride)
        K = sepkernels.backend()
        x = input.contiguous()
        Bp, N, nF = x.shape
        L, S = self.kernel_size, self.stride
        Cout = self.conv_transpose1d.out_channels
        ldt = _round_up(nF, 128)
        wp = torch.empty(Bp, N, ldt, device=x.device, dtype=x.dtype)
        K.repack(x, nF, wp, ldt, Bp * N, nF)
        ones = torch.ones(Bp, N, ldt, device=x.device, dtype=x.dtype)   # mask == 1: plain synthesis + overlap-add
        Tout = (nF - 1) * S + L
        out = torch.empty(Bp, 1, Cout, Tout, device=x.device, dtype=x.dtype)
        K.decoder_fwd(wp, ones, self.conv_transpose1d.weight, out, None, Bp, 1, N, Cout, L, S, nF, ldt, Tout, 0)
        return out.view(Bp, Cout, Tout)

    def get_basis(self):
        return self.conv_transpose1d.weight


# ----------------------------------------------------------------------------------------------------------------------
# Fourier / pseudo-inverse / gated bases (fallback path only)
# ----------------------------------------------------------------------------------------------------------------------
def _window(length, window_fn):
    """reference utils/audio.py:9-24 (periodic windows)"""
    table = {"hann": torch.hann_window, "hamming": torch.hamming_window, "blackman": torch.blackman_window}
    if window_fn not in table:
        raise ValueError("Not support {} window.".format(window_fn))
    return table[window_fn](length, periodic=True)


def _optimal_window(window, hop):
    """synthesis window that makes overlap-add of analysis-windowed frames the identity (reference utils/audio.py:26-43):
    w[n] / sum_k w[(n - k*hop) mod L]^2"""
    L = len(window)
    shifted = torch.stack([torch.roll(window, hop * k) for k in range(L // hop)], 0)
    return window / (shifted ** 2).sum(0)


def _all_bins(x, n_basis, dim, conj_sign):
    """bins 0 .. n_basis/2 -> all n_basis bins: append bins n_basis/2-1 .. 1 in that order, times conj_sign (-1 for an imaginary part)"""
    mirror = torch.flip(x.narrow(dim, 1, n_basis // 2 - 1), dims=(dim,))
    return torch.cat([x, conj_sign * mirror], dim)


class _FourierBase(nn.Module):
    def _angles(self):
        ang = self.frequency.unsqueeze(1) * self.time_seq.unsqueeze(0)                    # (n_basis/2+1, L)
        if self.trainable_phase:
            ang = ang + self.phase.unsqueeze(1)
        return ang

    def extra_repr(self):
        s = "{}, kernel_size={}, stride={}, trainable={}, onesided={}".format(self.n_basis, self.kernel_size, self.stride, self.trainable, self.onesided)
        return s + (", trainable_phase=True" if self.trainable_phase else "")


class FourierEncoder(_FourierBase):
    def __init__(self, n_basis, kernel_size, stride=None, window_fn="hann", trainable=False, trainable_phase=False, onesided=True, return_complex=True):
        super().__init__()
        self.n_basis, self.kernel_size, self.stride = n_basis, kernel_size, stride
        self.trainable, self.trainable_phase, self.onesided, self.return_complex = trainable, trainable_phase, onesided, return_complex
        self.frequency = nn.Parameter(2 * math.pi * torch.arange(n_basis // 2 + 1) / n_basis, requires_grad=trainable)
        self.time_seq = nn.Parameter(torch.arange(kernel_size), requires_grad=False)
        self.window = nn.Parameter(_window(kernel_size, window_fn))
        if trainable_phase:
            self.phase = nn.Parameter(torch.zeros(n_basis // 2 + 1))

    def _basis(self):
        ang = self._angles()
        re, im = torch.cos(ang), -torch.sin(ang)                                          # exp(-i ang)
        if not self.onesided:
            re, im = _all_bins(re, self.n_basis, 0, 1.0), _all_bins(im, self.n_basis, 0, -1.0)
        return self.window * re, self.window * im

    def forward(self, input):
        """(B, 1, T) -> complex (B, bins, frames), or real (B, 2*bins, frames) = [real parts; imaginary parts]"""
        re, im = self._basis()
        yr = F.conv1d(input, re.unsqueeze(1), stride=self.stride)
        yi = F.conv1d(input, im.unsqueeze(1), stride=self.stride)
        return torch.complex(yr, yi) if self.return_complex else torch.cat([yr, yi], 1)

    def get_basis(self):
        return torch.cat(self._basis(), 0)


def _fourier_synthesis(mod, input, window, stride):
    """sum over ALL bins of Re{ X_f * window[n] exp(+i ang_f[n]) } / n_basis, overlap-added"""
    n_basis = mod.n_basis
    if torch.is_complex(input):
        xr, xi = input.real, input.imag
    else:
        xr, xi = torch.chunk(input, 2, dim=1)
    ang = mod._angles()
    br, bi = _all_bins(torch.cos(ang), n_basis, 0, 1.0), _all_bins(torch.sin(ang), n_basis, 0, -1.0)
    br, bi = window * br / n_basis, window * bi / n_basis
    if xr.size(1) == n_basis // 2 + 1:                                                    # one-sided input: rebuild the conjugate half
        xr, xi = _all_bins(xr, n_basis, 1, 1.0), _all_bins(xi, n_basis, 1, -1.0)
    return F.conv_transpose1d(xr, br.unsqueeze(1), stride=stride) - F.conv_transpose1d(xi, bi.unsqueeze(1), stride=stride)


class FourierDecoder(_FourierBase):
    def __init__(self, n_basis, kernel_size, stride=None, window_fn="hann", trainable=False, trainable_phase=False, onesided=True):
        super().__init__()
        self.n_basis, self.kernel_size, self.stride = n_basis, kernel_size, stride
        self.trainable, self.trainable_phase, self.onesided = trainable, trainable_phase, onesided
        self.frequency = nn.Parameter(2 * math.pi * torch.arange(n_basis // 2 + 1) / n_basis, requires_grad=trainable)
        self.time_seq = nn.Parameter(torch.arange(kernel_size), requires_grad=False)
        self.optimal_window = nn.Parameter(_optimal_window(_window(kernel_size, window_fn), stride))
        if trainable_phase:
            self.phase = nn.Parameter(torch.zeros(n_basis // 2 + 1))

    def forward(self, input):
        return _fourier_synthesis(self, input, self.optimal_window, self.stride)

    def get_basis(self):
        ang = self._angles()
        re, im = torch.cos(ang), torch.sin(ang)
        if not self.onesided:
            re, im = _all_bins(re, self.n_basis, 0, 1.0), _all_bins(im, self.n_basis, 0, -1.0)
        return torch.cat([self.optimal_window * re / self.n_basis, self.optimal_window * im / self.n_basis], 0)


class PinvDecoder(nn.Module):
    """Synthesis with the pseudo-inverse of the encoder's basis (divided by the frame overlap), reference :253-323.  Shares the
    encoder's parameters: it has none of its own."""

    def __init__(self, encoder):
        super().__init__()
        self.encoder = encoder
        self.kernel_size, self.stride = encoder.kernel_size, encoder.stride
        if isinstance(encoder, Encoder):
            if encoder.nonlinear:
                raise ValueError("Not support pseudo inverse of 'Conv1d + nonlinear'.")
            self.weight = encoder.conv1d.weight
            if self.weight.size(0) < self.weight.size(2):
                raise ValueError("Cannot compute the left inverse of encoder's weight. In encoder, `out_channels` must be equal to or greater than `kernel_size`.")
        elif isinstance(encoder, FourierEncoder):
            if encoder.onesided or encoder.return_complex:
                raise ValueError("Both encoder.onesided and encoder.return_complex are expected to be False.")
        else:
            raise TypeError("Invalid encoder is given.")

    def get_basis(self):
        w = self.weight.permute(1, 0, 2)                                                  # (in_channels, N, L)
        return torch.pinverse(w).permute(2, 0, 1).contiguous() / (self.kernel_size // self.stride)

    def forward(self, input):
        if isinstance(self.encoder, Encoder):
            return F.conv_transpose1d(input, self.get_basis(), stride=self.stride)
        return _fourier_synthesis(self.encoder, input, self.encoder.window, self.stride)


class GatedEncoder(nn.Module):
    """relu(U x) * sigmoid(V x) on frames of the input normalised by its L2 norm over time (reference :325-346)"""

    def __init__(self, in_channels, n_basis, kernel_size=16, stride=8, eps=1e-12):
        super().__init__()
        self.kernel_size, self.stride, self.eps = kernel_size, stride, eps
        self.conv1d_U = nn.Conv1d(in_channels, n_basis, kernel_size=kernel_size, stride=stride, bias=False)
        self.conv1d_V = nn.Conv1d(in_channels, n_basis, kernel_size=kernel_size, stride=stride, bias=False)
        self.relu, self.sigmoid = nn.ReLU(), nn.Sigmoid()

    def forward(self, input):
        x = input / (torch.linalg.norm(input, dim=2, keepdim=True) + self.eps)
        if self._on_kernels(x):
            # both analysis convolutions on sep_encoder_fwd (the ReLU of U inside the kernel), their basis gradients on sep_unfold + sep_pw_wgrad
            # (sepkernels.functional.EncodeFn); the gate itself is two elementwise torch kernels.  Frames come back in rows of the workspace
            # stride: the valid ones are a view.
            from sepkernels.functional import EncodeFn
            n_frames = (x.shape[-1] - self.kernel_size) // self.stride + 1
            u = EncodeFn.apply(x, self.conv1d_U.weight, self.stride, True)[..., :n_frames]
            v = EncodeFn.apply(x, self.conv1d_V.weight, self.stride, False)[..., :n_frames]
            return u * torch.sigmoid(v)
        return self.relu(self.conv1d_U(x)) * self.sigmoid(self.conv1d_V(x))

    def _on_kernels(self, x):
        """the kernels take the TasNet geometry they were written for: the input already padded to whole frames (the TasNet forwards pad
        before the encoder, conv_tasnet.py:145-149), a kernel that is a multiple of its stride, no gradient with respect to the input"""
        from sepkernels.functional import takes
        return (takes(x) and not x.requires_grad and x.dim() == 3 and self.kernel_size % self.stride == 0
                and x.shape[-1] >= self.kernel_size and (x.shape[-1] - self.kernel_size) % self.stride == 0)


from sepkernels.shadowed import fall_through as _fall_through      # names of the reference's same-named module this tree does not define

__getattr__ = _fall_through(__name__, __file__)
