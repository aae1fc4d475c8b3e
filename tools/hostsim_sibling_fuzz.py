"""Development tool: random small configurations of the dual-path separators (DPRNN-TasNet, DPTNet, GALRNet, SepFormer) through this
tree's modules with the host simulation of the kernel sources behind the C ABI (tools/hostsim.py), against the UNMODIFIED reference
modules (imported from /root/reference/src under private names, fp64, same state_dict): forward and every gradient.
Needs the reference tree; not part of the test suite.

    python tools/hostsim_sibling_fuzz.py [seed] [count]
"""
import os
import random
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "dnn-based_source_separation_amd", "src"), ROOT]
import hostsim                      # noqa: E402
import sepkernels                   # noqa: E402
from criterion.sdr import NegSISDR  # noqa: E402
from criterion.pit import PIT1d     # noqa: E402


def ref_run(kind, cfg, state, mixture, sources):
    """forward + PIT loss + gradients of the reference class `kind` in a separate process (its package names collide with this tree's)"""
    import pickle
    import subprocess

    blob = pickle.dumps((kind, cfg, {k: v.numpy() for k, v in state.items()}, mixture.numpy(), sources.numpy()))
    code = r'''
import sys, types, pickle, importlib
sys.modules.setdefault("torchaudio", types.ModuleType("torchaudio"))
sys.path.insert(0, "/root/reference/src")
import torch
kind, cfg, state, mixture, sources = pickle.loads(sys.stdin.buffer.read())
mod = {"DPRNNTasNet": "models.dprnn_tasnet", "DPTNet": "models.dptnet", "GALRNet": "models.galrnet", "SepFormer": "models.sepformer"}[kind]
cls = getattr(importlib.import_module(mod), kind)
from criterion.sdr import NegSISDR
from criterion.pit import PIT1d
model = cls(**cfg)
model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
model.double()
est = model(torch.from_numpy(mixture).double())
loss, pat = PIT1d(NegSISDR(), n_sources=cfg["n_sources"])(est, torch.from_numpy(sources).double())
loss.backward()
out = (est.detach().numpy(), loss.item(), pat.numpy(), {k: p.grad.numpy() for k, p in model.named_parameters()})
sys.stdout.buffer.write(pickle.dumps(out))
'''
    r = subprocess.run([sys.executable, "-c", code], input=blob, capture_output=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr.decode()[-2000:])
    est, loss, pat, grads = pickle.loads(r.stdout)
    return torch.from_numpy(est), loss, torch.from_numpy(pat), {k: torch.from_numpy(v) for k, v in grads.items()}


def random_config(R):
    from models.dprnn_tasnet import DPRNNTasNet
    from models.dptnet import DPTNet
    from models.galrnet import GALRNet
    from models.sepformer import SepFormer
    S = R.choice([1, 2, 4])
    L = S * R.choice([1, 2])
    n_src = R.randint(1, 3)
    N = 16 * R.randint(1, 4)
    base = dict(n_basis=N, kernel_size=L, stride=S, enc_basis="trainable", dec_basis="trainable", enc_nonlinear=R.choice([None, "relu"]), n_sources=n_src,
                causal=R.random() < 0.3)
    hop = R.randint(2, 9)
    chunk = hop * 2 if R.random() < 0.7 else hop * R.randint(1, 3)
    kind = R.choice(["DPRNNTasNet", "DPTNet", "GALRNet", "SepFormer"])
    if kind == "DPRNNTasNet":
        cfg = dict(base, sep_hidden_channels=R.choice([16, 32, 24]), sep_bottleneck_channels=16 * R.randint(1, 3), sep_chunk_size=chunk, sep_hop_size=hop,
                   sep_num_blocks=R.randint(1, 2), sep_norm=True, mask_nonlinear=R.choice(["sigmoid", "softmax"]), rnn_type="lstm")
        return DPRNNTasNet, kind, cfg
    heads = R.choice([1, 2, 4])
    if kind == "DPTNet":
        cfg = dict(base, sep_bottleneck_channels=16 * R.randint(1, 3), sep_hidden_channels=R.choice([16, 32, 20]), sep_chunk_size=chunk, sep_hop_size=hop,
                   sep_num_blocks=R.randint(1, 2), sep_num_heads=heads, sep_norm=True, sep_nonlinear="relu", sep_dropout=0,
                   mask_nonlinear=R.choice(["relu", "sigmoid", "softmax"]))
        if N % heads or cfg["sep_bottleneck_channels"] % heads:
            return None
        return DPTNet, kind, cfg
    if kind == "GALRNet":
        low = R.random() < 0.5
        cfg = dict(base, sep_hidden_channels=R.choice([16, 32]), sep_chunk_size=chunk, sep_hop_size=hop, sep_down_chunk_size=R.randint(2, 6) if low else None,
                   sep_num_blocks=R.randint(1, 2), sep_num_heads=heads, sep_norm=True, sep_dropout=0.0, mask_nonlinear=R.choice(["relu", "sigmoid"]), low_dimension=low)
        if N % heads:
            return None
        return GALRNet, kind, cfg
    cfg = dict(base, sep_bottleneck_channels=16 * R.randint(1, 3), sep_chunk_size=chunk, sep_hop_size=hop, sep_num_blocks=1, sep_num_layers_intra=R.randint(1, 2),
               sep_num_layers_inter=R.randint(1, 2), sep_num_heads_intra=heads, sep_num_heads_inter=heads, sep_d_ff_intra=R.choice([16, 24]), sep_d_ff_inter=R.choice([16, 24]),
               sep_norm=True, sep_nonlinear="relu", sep_dropout=0.0, mask_nonlinear=R.choice(["relu", "sigmoid"]))
    if cfg["sep_bottleneck_channels"] % heads:
        return None
    return SepFormer, kind, cfg


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    R = random.Random(seed)
    so = os.environ.get("HOSTSIM_LIB") or hostsim.build(tempfile.mkdtemp())      # HOSTSIM_LIB: a prebuilt (e.g. sanitized) library
    failed = done = 0
    with hostsim.HostSimBackend(so) as K:
        class Named:
            name = "hostsim"

            def __getattr__(self, n):
                return getattr(K, n)
        old = sepkernels._set_backend_for_tests(Named())
        try:
            while done < count:
                pick = random_config(R)
                if pick is None:
                    continue
                cls, kind, cfg = pick
                torch.manual_seed(seed * 1000 + done)
                model = cls(**cfg)
                with torch.no_grad():
                    for k, p in model.named_parameters():
                        if "norm" in k or k.endswith(("gamma", "beta", "bias", "prelu.weight")):
                            p.add_(0.05 * torch.randn_like(p))
                B, T = R.randint(1, 2), R.randint(60, 500)
                sources = 0.1 * torch.randn(B, cfg["n_sources"], T)
                mixture = sources.sum(1, keepdim=True)
                state = {k: v.detach().clone() for k, v in model.state_dict().items()}
                t0 = time.time()
                path = "kernels" if not model.kernel_path_problems() else "composed"
                est = model(mixture)
                loss, pat = PIT1d(NegSISDR(), n_sources=cfg["n_sources"])(est, sources)
                loss.backward()
                ref_est, ref_loss, ref_pat, ref_grads = ref_run(kind, cfg, state, mixture, sources)
                e_out = ((est.double() - ref_est).abs().max() / ref_est.abs().max()).item()
                num = den = 0.0
                for k, p in model.named_parameters():
                    r = ref_grads[k].double()
                    num, den = max(num, (p.grad.double() - r).abs().max().item()), max(den, r.abs().max().item())
                ok = e_out < 2e-4 and num / den < 2e-3 and torch.equal(pat, ref_pat)
                failed += not ok
                done += 1
                brief = {k: v for k, v in cfg.items() if k not in ("enc_basis", "dec_basis", "sep_norm", "sep_nonlinear", "rnn_type")}
                print("{} {:11s} {:8s} B={} T={} fwd {:.1e} grad {:.1e} {:.0f} s  {}".format("ok  " if ok else "FAIL", kind, path, B, T, e_out, num / den, time.time() - t0, brief), flush=True)
        finally:
            sepkernels._set_backend_for_tests(old)
    print("{} configurations failed".format(failed))
    return 1 if failed else 0


if __name__ == "__main__":
    raise SystemExit(main())
