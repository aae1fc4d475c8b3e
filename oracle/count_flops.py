"""Test / measurement infrastructure (NOT product code): algorithmic FLOP per utterance of the dual-path separators at the sizes of the
reference's own recipes, counted on the UNMODIFIED reference modules (/root/reference/src, CPU) with torch.utils.flop_counter: every
matrix product / convolution / attention product ATen executes in ONE forward pass of one 4-s utterance, x 3 for forward + backward
(the convention of SURVEY.md section 8d).  Elementwise work, normalisations and softmax are not counted (as in section 8d).

    python oracle/count_flops.py            -> prints {config: GFLOP per utterance, fwd + bwd}; bench.py's --config lines quote these
"""
import json
import sys
import types

import torch
from torch.utils.flop_counter import FlopCounterMode

REF = "/root/reference/src"
T = 32000


def main():
    sys.modules.setdefault("torchaudio", types.ModuleType("torchaudio"))
    sys.path.insert(0, REF)
    from models.dprnn_tasnet import DPRNNTasNet
    from models.dptnet import DPTNet
    from models.galrnet import GALRNet
    from models.sepformer import SepFormer
    from models.conv_tasnet import ConvTasNet
    tr = dict(enc_basis="trainable", dec_basis="trainable")
    paper = dict(n_basis=512, kernel_size=16, stride=8, enc_nonlinear=None, sep_hidden_channels=512, sep_bottleneck_channels=128, sep_skip_channels=128,
                 sep_kernel_size=3, sep_num_blocks=3, sep_num_layers=8, dilated=True, separable=True, sep_nonlinear="prelu", sep_norm=True,
                 mask_nonlinear="sigmoid", n_sources=2, **tr)
    work = {
        "convtasnet2": (ConvTasNet, dict(paper, causal=False)),
        "causal": (ConvTasNet, dict(paper, causal=True)),
        "dprnn": (DPRNNTasNet, dict(n_basis=64, kernel_size=2, stride=1, enc_nonlinear=None, sep_hidden_channels=128, sep_bottleneck_channels=64,
                                    sep_chunk_size=250, sep_hop_size=125, sep_num_blocks=6, sep_norm=True, mask_nonlinear="sigmoid", causal=False,
                                    rnn_type="lstm", n_sources=2, **tr)),
        "dptnet": (DPTNet, dict(n_basis=64, kernel_size=2, stride=1, enc_nonlinear=None, sep_bottleneck_channels=64, sep_hidden_channels=128,
                                sep_chunk_size=250, sep_hop_size=125, sep_num_blocks=6, sep_num_heads=4, sep_norm=True, sep_nonlinear="relu",
                                sep_dropout=0, mask_nonlinear="relu", causal=False, n_sources=2, **tr)),
        "galrnet": (GALRNet, dict(n_basis=64, kernel_size=16, stride=8, enc_nonlinear=None, sep_hidden_channels=128, sep_chunk_size=100,
                                  sep_hop_size=50, sep_down_chunk_size=32, sep_num_blocks=6, sep_num_heads=8, sep_norm=True, sep_dropout=1e-1,
                                  mask_nonlinear="relu", causal=False, n_sources=2, low_dimension=True, **tr)),
        "sepformer": (SepFormer, dict(n_basis=256, kernel_size=16, stride=8, enc_nonlinear="relu", sep_bottleneck_channels=256, sep_chunk_size=250,
                                      sep_hop_size=125, sep_num_blocks=2, sep_num_layers_intra=8, sep_num_layers_inter=8, sep_num_heads_intra=8,
                                      sep_num_heads_inter=8, sep_d_ff_intra=1024, sep_d_ff_inter=1024, sep_norm=True, sep_nonlinear="relu",
                                      sep_dropout=1e-1, mask_nonlinear="relu", causal=False, n_sources=2, **tr)),
    }
    only = sys.argv[1:] or list(work)
    out = {}
    for name in only:
        cls, cfg = work[name]
        torch.manual_seed(0)
        model = cls(**cfg).eval()
        x = 0.1 * torch.randn(1, 1, T)
        lstm = [0]

        def lstm_hook(mod, inp, _out):      # aten.lstm is ONE fused op for the counter: its products are added by hand -- per layer and direction
            xx = inp[0]                     # 4H x (I + H) multiply-adds per sequence element (gate pre-activations from the input and the state)
            steps = xx.shape[0] * xx.shape[1]
            for layer in range(mod.num_layers):
                i = mod.input_size if layer == 0 else mod.hidden_size * (2 if mod.bidirectional else 1)
                lstm[0] += 2 * steps * (2 if mod.bidirectional else 1) * 4 * mod.hidden_size * (i + mod.hidden_size)
        hooks = [m.register_forward_hook(lstm_hook) for m in model.modules() if isinstance(m, torch.nn.LSTM)]
        with torch.no_grad(), FlopCounterMode(display=False) as fc:
            model(x)
        for h in hooks:
            h.remove()
        fwd = fc.get_total_flops() + lstm[0]
        out[name] = {"gflop_fwd": fwd / 1e9, "gflop_fwd_bwd": 3 * fwd / 1e9, "of_it_lstm_fwd": lstm[0] / 1e9, "parameters": sum(p.numel() for p in model.parameters())}
        print(name, json.dumps(out[name]), flush=True)
    return out


if __name__ == "__main__":
    main()
