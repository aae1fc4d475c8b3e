"""Where the HOST time of a dual-path training step goes (cProfile over a few eager steps on the GPU box):
    python tools/host_profile.py galrnet [steps]"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import bench_legs as BL                                                     # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "galrnet"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    from criterion.sdr import NegSISDR
    from criterion.pit import PIT1d
    cls, cfg, B, adam, _, _ = BL._dual_path_workloads()[name]
    dev = torch.device("cuda", 0)
    torch.manual_seed(111)
    model = cls(**cfg).to(dev)
    crit = PIT1d(NegSISDR(), n_sources=2)
    opt = torch.optim.Adam(model.parameters(), **adam)
    src = (0.1 * torch.randn(B, 2, BL.T_SAMPLES, generator=torch.Generator().manual_seed(111))).to(dev)
    mix = src.sum(1, keepdim=True).contiguous()

    def step():
        opt.zero_grad(set_to_none=True)
        loss, _ = crit(model(mix), src)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t_host = time.perf_counter() - t0                                       # launches only: the host runs ahead of the device
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("{}: host {:.2f} ms / step to issue, {:.2f} ms / step with the device drained".format(name, 1e3 * t_host / steps, 1e3 * t_all / steps))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
    st.sort_stats("cumulative").print_stats("dnn-based_source_separation_amd|torch/nn/functional|torch/autograd/function", 45)


if __name__ == "__main__":
    main()
