"""Development tool: random small Conv-TasNet CONFIGURATIONS -- widths in any multiple of 16, 1 - 3 speakers, window / hop pairs, blocks x
layers, ReLU or linear encoder, sigmoid or softmax mask, ragged lengths -- through the product's fused path with the host simulation of
the kernel sources behind the C ABI (tools/hostsim.py), against the fp64 oracle (oracle/fast_port.py): forward, permutation, every
gradient.  The first run of this found the sep_pack_weights refusal for widths like 48 / 80 (tests/test_modules_cpu.py::
test_widths_in_odd_multiples_of_16).

    python tools/hostsim_model_fuzz.py [seed] [count]
"""
import os
import random
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "dnn-based_source_separation_amd", "src"), ROOT]
import hostsim, sepkernels
from oracle import fast_port as FP
from models.conv_tasnet import ConvTasNet
from criterion.sdr import NegSISDR
from criterion.pit import PIT1d
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 10
so = os.environ.get("HOSTSIM_LIB") or hostsim.build(tempfile.mkdtemp())      # HOSTSIM_LIB: a prebuilt (e.g. sanitized) library
failed = 0
R = random.Random(seed)
with hostsim.HostSimBackend(so) as K:
    class Named:
        name = "hostsim"
        def __getattr__(self, n): return getattr(K, n)
    old = sepkernels._set_backend_for_tests(Named())
    try:
        for it in range(count):
            S = R.choice([2, 3, 4, 8, 10]); L = S * R.choice([1, 2, 4])
            cin = R.choice([1, 1, 2])                    # stereo: the music recipes
            cfg = dict(n_basis=16 * R.randint(1, 5), kernel_size=L, stride=S, enc_basis="trainable", dec_basis="trainable", enc_nonlinear=R.choice([None, "relu"]),
                       sep_hidden_channels=16 * R.randint(1, 6), sep_bottleneck_channels=R.choice([16, 32, 48, 64, 128, 128]), sep_skip_channels=16 * R.randint(1, 4), sep_kernel_size=3,
                       sep_num_blocks=R.randint(1, 2), sep_num_layers=R.randint(1, 4), dilated=True, separable=True, causal=False, sep_nonlinear="prelu", sep_norm=True,
                       mask_nonlinear=R.choice(["sigmoid", "softmax"]), n_sources=R.randint(1, 3), in_channels=cin)
            if (cfg["n_sources"] * cfg["n_basis"]) % 16: continue
            B, T = R.randint(1, 3), R.randint(L + 5, 1500)
            torch.manual_seed(seed * 100 + it)
            model = ConvTasNet(**cfg)
            assert model.fused, model.fused_reason
            with torch.no_grad():
                for k, p in model.named_parameters():
                    if "norm" in k or k.endswith("nonlinear1d.weight") or k.endswith("prelu.weight"):
                        p.add_(0.1 * torch.randn_like(p))
            if cin == 1:
                src = 0.1 * torch.randn(B, cfg["n_sources"], T); mix = src.sum(1, keepdim=True)
            else:
                src = 0.1 * torch.randn(B, cfg["n_sources"], cin, T); mix = src.sum(1, keepdim=True)
            t0 = time.time()
            est = model(mix)
            p64 = {k: v.detach().double() for k, v in model.state_dict().items()}
            if cin == 1:
                loss, pat = PIT1d(NegSISDR(), n_sources=cfg["n_sources"])(est, src)
                loss.backward()
                ref_out, ref_loss, ref_pat, ref_grads = FP.train_step(p64, cfg, mix, src, dtype=torch.float64)
            else:                                        # stereo: a fixed linear functional of the estimate instead of the PIT loss (the oracle's is mono)
                Wr = torch.randn(est.shape)
                (est * Wr).sum().backward()
                pp = {k: v.clone().requires_grad_(True) for k, v in p64.items()}
                ref_out, _ = FP.conv_tasnet(mix.view(B, cin, T).double(), pp, cfg)
                ref_out = ref_out.view(est.shape)
                (ref_out * Wr.double()).sum().backward()
                ref_grads = {k: v.grad for k, v in pp.items() if v.grad is not None}
                ref_out = ref_out.detach()
                pat = ref_pat = torch.zeros(1)
            e_out = ((est.double() - ref_out).abs().max() / ref_out.abs().max()).item()
            num = den = 0.0
            for k, p in model.named_parameters():
                r = ref_grads[k].double(); num = max(num, (p.grad.double() - r).abs().max().item()); den = max(den, r.abs().max().item())
            floor = 0.0                                  # how ill-conditioned is this draw?  the fp32 oracle against the fp64 one
            if cin == 1:
                _, _, _, g32 = FP.train_step(p64, cfg, mix, src, dtype=torch.float32)
                floor = max((g32[k].double() - ref_grads[k]).abs().max().item() for k in ref_grads) / den
            ok = e_out < 1e-4 and num / den < max(1e-3, 30 * floor) and torch.equal(pat, ref_pat)
            failed += not ok
            if not ok:                                   # which tensors
                for k, pth in model.named_parameters():
                    r = ref_grads[k].double()
                    e = (pth.grad.double() - r).abs().max().item()
                    if e > 1e-4 * den:
                        print("     %-70s err %.2e of its %.2e (largest gradient %.2e)" % (k, e, r.abs().max().item(), den), flush=True)
            print("%s in_channels=%d N=%d B=%d H=%d Sc=%d X=%d R=%d L=%d S=%d n_src=%d %s %s batch=%d T=%d  fwd %.1e grad %.1e (fp32 oracle %.1e)  %.0f s" % ("ok  " if ok else "FAIL", cin, cfg["n_basis"], cfg["sep_bottleneck_channels"], cfg["sep_hidden_channels"], cfg["sep_skip_channels"], cfg["sep_num_layers"], cfg["sep_num_blocks"], L, S, cfg["n_sources"], cfg["enc_nonlinear"], cfg["mask_nonlinear"], B, T, e_out, num / den, floor, time.time() - t0), flush=True)
    finally:
        sepkernels._set_backend_for_tests(old)
print("{} configurations failed".format(failed))
raise SystemExit(1 if failed else 0)
