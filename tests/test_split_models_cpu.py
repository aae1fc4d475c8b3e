"""CPU: the error models of the two split arithmetics of sep_pw_gemm (tools/split_accuracy.py: part products formed exactly,
fp32 accumulation) -- the claims DESIGN.md section 4 makes about them, kept under test: both stay at the fp32 level
relative to |A||B| per output on operands of very different dynamic range, and the fp16 split NEEDS its per-column scale."""
import importlib.util
import os

import torch

spec = importlib.util.spec_from_file_location("split_accuracy", os.path.join(os.path.dirname(__file__), "..", "tools", "split_accuracy.py"))
SA = importlib.util.module_from_spec(spec)
spec.loader.exec_module(SA)


def test_split_arithmetics_stay_at_fp32_level():
    torch.manual_seed(0)
    A, cs = SA.cases(K=256, M=64, N=128)
    for name, B in cs:
        e = SA.errors(A, B)
        assert e["bf16x6"] <= 2 * e["fp32"] + 1e-7, (name, e)
        assert e["fp16x3 col-scale"] <= 3 * e["fp32"] + 1e-7, (name, e)
    e = SA.errors(A, dict(cs)["columns spread over e^+-6"])
    assert e["fp16x3 one scale"] > 1e3 * e["fp16x3 col-scale"], e       # one scale per tensor is not enough


def test_three_way_bf16_split_is_exact():
    x = torch.randn(10000) * torch.exp(8 * torch.randn(10000))
    h = SA.trunc_bf16(x)
    r = x - h
    m = SA.trunc_bf16(r)
    lo = SA.trunc_bf16(r - m)
    assert torch.equal(h + m + lo, x) and torch.equal(lo, r - m)
