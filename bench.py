"""
bench.py -- separated audio frames/sec (fwd + SI-SDR/PIT + bwd [+ all-reduce] + clip + Adam) of Conv-TasNet paper-best
(N512 L16 B128 H512 Sc128 P3 X8 R3, 2 speakers) on synthetic 4 s @ 8 kHz mixtures, 16 utterances per GPU.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One JSON line on rank 0.  A "step" is one pass of the hot path over one batch resident in HBM.  `roofline` is for the
dominant kernel (the fp32-MFMA pointwise GEMM `pw_gemm_direct_kernel`): its launches are bracketed with HIP events on the
launch stream during the timed steps; achieved = algorithmic FLOPs of those launches / their summed duration.
`cpu_baseline` is the oracle's functional port (oracle/fast_port.py, same ATen CPU kernels as the reference) timed
on this box's host cores on a bounded sample (N=1 runs only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "dnn-based_source_separation_amd", "src")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PAPER = dict(n_basis=512, kernel_size=16, stride=8, enc_basis="trainable", dec_basis="trainable", enc_nonlinear=None,
             sep_hidden_channels=512, sep_bottleneck_channels=128, sep_skip_channels=128, sep_kernel_size=3,
             sep_num_blocks=3, sep_num_layers=8, dilated=True, separable=True, causal=False, sep_nonlinear="prelu",
             sep_norm=True, mask_nonlinear="sigmoid", n_sources=2)
T_SAMPLES = 32000            # 4 s @ 8 kHz
PER_GPU_BATCH = 16
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
HBM_PEAK_TBS = 8.0


class TimedBackend:
    """Wraps the kernel facade: brackets every launch of the two MFMA kernels with HIP events (recorded on the
    current stream = the launch stream) and tallies their algorithmic FLOPs."""

    def __init__(self, inner):
        self._inner = inner
        self.enabled = False
        self.records = {"pw_gemm": [], "pw_wgrad": []}
        self.variants = {}
        self.name = inner.name

    def __getattr__(self, item):
        return getattr(self._inner, item)

    def _timed(self, key, flops, fn, kw):
        if not self.enabled:
            return fn(**kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(**kw)
        e1.record()
        self.records[key].append((e0, e1, flops))

    def pw_gemm(self, **kw):
        if self.enabled:
            # algorithmic HBM bytes of this launch: every operand row once (fp32), weights ignored
            M, K = kw["M"], kw["K"]
            msp = kw.get("m_split", 0)
            rows = K + M                                                     # X in, Y out
            rows += (M - msp if kw.get("accumulate") else 0)                 # accumulated part read back
            rows += ((msp or M) if kw.get("epi_res") is not None else 0)     # residual
            rows += (M if kw.get("epi_aux") is not None else 0)              # PReLU-bwd / row-sum operand
            rows += (2 * K if kw.get("pro_store") is not None else 0)        # gLN-bwd: pre-activation in, d(pre-activation) out
            key = "T{}P{}S{}".format(int(bool(kw.get("trans_a"))), int(kw.get("pro_mode", 0)), int(bool(kw.get("k_split"))))
            c, a = self.variants.get(key, (0, 0.0))
            self.variants[key] = (c + 1, a + 4.0 * rows * kw["B"] * kw["T"])
        self._timed("pw_gemm", 2.0 * kw["M"] * kw["K"] * kw["B"] * kw["T"], self._inner.pw_gemm, kw)

    def pw_wgrad(self, **kw):
        self._timed("pw_wgrad", 2.0 * kw["M"] * kw["N"] * kw["B"] * kw["T"], self._inner.pw_wgrad, kw)

    def summary(self, key):
        ms = sum(a.elapsed_time(b) for a, b, _ in self.records[key])
        fl = sum(f for _, _, f in self.records[key])
        return len(self.records[key]), ms, fl


# ---- workload constants; restated here so that the timed path imports nothing from oracle/ -------
def num_frames(T, L, S):
    """Encoder frames of a T-sample utterance with ConvTasNet's input padding (reference conv_tasnet.py:145-149)."""
    padding = (S - (T - L) % S) % S
    return (T + padding - L) // S + 1


def flops_per_frame(cfg):
    """Forward FLOP per frame: 2 x the MAC/frame formula of SURVEY.md section 8(d)."""
    N, L = cfg["n_basis"], cfg["kernel_size"]
    Bn, H, Sc = cfg["sep_bottleneck_channels"], cfg["sep_hidden_channels"], cfg["sep_skip_channels"]
    P, X, R, ns = cfg["sep_kernel_size"], cfg["sep_num_layers"], cfg["sep_num_blocks"], cfg["n_sources"]
    mac = N * L + N * Bn + (R * X - 1) * (2 * Bn * H + H * Sc + H * P) + (Bn * H + H * Sc + H * P) + Sc * ns * N + ns * N * L
    return 2 * mac


def bytes_per_frame(cfg):
    """Forward algorithmic HBM bytes per frame (fp32), SURVEY.md section 8(d)."""
    N = cfg["n_basis"]
    Bn, H, Sc = cfg["sep_bottleneck_channels"], cfg["sep_hidden_channels"], cfg["sep_skip_channels"]
    X, R, ns, S = cfg["sep_num_layers"], cfg["sep_num_blocks"], cfg["n_sources"], cfg["stride"]
    return 4 * (R * X * (2 * Bn + 4 * H + 2 * Sc) + (2 * N + Bn + 2 * ns * N + ns * S))


def pmc_traffic(variant_tally):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (separate FETCH_SIZE and
    WRITE_SIZE runs, gfx950 x2 correction on FETCH_SIZE; tools/pmc_passes.sh + tools/pmc_traffic.py).  PMC counters cannot
    be read from inside the timed process, so this is the value measured on the profiled run of this same command."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        pmc = {k: v["bytes_per_launch"] for k, v in t["gemm_variants"].items() if v["bytes_per_launch"] > 1e6}   # rocprofv3 reports ~0 for one instantiation: left out on both sides
        n = tr = al = 0.0
        for key, (cnt, abytes) in variant_tally.items():      # launch-weighted over the template variants both sides saw
            if key in pmc:
                n += cnt
                tr += cnt * pmc[key]
                al += abytes
        if n == 0:
            return {"traffic": None}
        return {"traffic": tr / n, "traffic_unit": "bytes/launch", "algorithmic_bytes_per_launch": al / n,
                "traffic_over_algorithmic": tr / al, "traffic_source": t["source"],
                "traffic_variants_covered": "{:.0f} of {:.0f} launches".format(n, sum(c for c, _ in variant_tally.values()))}
    except (OSError, KeyError, ValueError):
        return {"traffic": None}


def cpu_baseline(sample_steps=2):
    """Reference-equivalent CPU path (oracle/fast_port.py) on the host cores, bounded sample: B=2 utterances/step."""
    from oracle import fast_port as FP       # the ONLY oracle import of this file: the cpu_baseline leg
    from models.conv_tasnet import ConvTasNet
    torch.manual_seed(111)
    model = ConvTasNet(**PAPER)
    p = {k: v.detach().clone() for k, v in model.state_dict().items()}
    B = 2
    g = torch.Generator().manual_seed(111)
    sources = 0.1 * torch.randn(B, 2, T_SAMPLES, generator=g)
    mixture = sources.sum(1, keepdim=True)
    all_cores = torch.get_num_threads()
    best = None
    for cores in sorted({all_cores, min(all_cores, 32)}, reverse=True):   # oneDNN often peaks below the full core count
        torch.set_num_threads(cores)
        for _ in range(2):                  # oneDNN primitive caches / allocator warm-up
            FP.train_step(p, PAPER, mixture, sources, dtype=torch.float32)
        t0 = time.perf_counter()
        for _ in range(sample_steps):
            FP.train_step(p, PAPER, mixture, sources, dtype=torch.float32)
        dt_c = (time.perf_counter() - t0) / sample_steps
        if best is None or dt_c < best[0]:
            best = (dt_c, cores)
    torch.set_num_threads(all_cores)
    dt, cores = best
    frames = B * num_frames(T_SAMPLES, 16, 8)
    return {"value": frames / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "{} timed fwd+PIT+bwd steps (after 2 warm-up) of B={} paper-best utterances, fp32, torch CPU "
                      "(oracle/fast_port.py: same ATen conv/GroupNorm kernels as the reference modules), {:.2f} s/step".format(
                          sample_steps, B, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="utterances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-f32-pass", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # nccl = RCCL on ROCm (one rank per GPU).  SEPK_BENCH_BACKEND=gloo + SEPK_BENCH_ONE_GPU=1 exist only to exercise the
        # multi-rank code path on a single-GPU box (all ranks on device 0, all-reduce through the host).
        dist.init_process_group(os.environ.get("SEPK_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus {} but WORLD_SIZE {}".format(args.gpus, world), file=sys.stderr)
    if os.environ.get("SEPK_BENCH_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import sepkernels
    from sepkernels.train import FusedTrainStep
    from models.conv_tasnet import ConvTasNet
    from criterion.sdr import NegSISDR
    from criterion.pit import PIT1d

    sepkernels.load()          # fail loudly if the HIP library is missing
    timed = TimedBackend(sepkernels.backend())
    sepkernels._backend = timed

    torch.manual_seed(111)
    model = ConvTasNet(**PAPER).to(dev)
    crit = PIT1d(NegSISDR(), n_sources=2)
    step = FusedTrainStep(model, crit, lr=1e-3, max_norm=5.0)   # recipe defaults: adam 1e-3, clip 5 (train.sh:50-57)
    g = torch.Generator().manual_seed(111 + rank)
    sources = (0.1 * torch.randn(args.batch, 2, T_SAMPLES, generator=g)).to(dev)
    mixture = sources.sum(1, keepdim=True).contiguous()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step(mixture, sources)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(mixture, sources)
    sync()
    elapsed = time.perf_counter() - t0
    # Second pass of the SAME K steps with every MFMA-kernel launch bracketed by HIP events (roofline.achieved).  Kept out
    # of the headline region: the ~300 event pairs per step cost 1.05 ms/step (3.5 %) of dispatch bubbles (measured A/B).
    elapsed_instr = None
    if not args.no_kernel_timing:
        # kernel-alone durations: the weight gradients go back onto the main stream for this pass (on their side stream
        # they overlap the input-gradient chain, and an event pair around a launch would time the overlap, not the kernel)
        side_prev = os.environ.get("SEPK_SIDE_STREAM")
        os.environ["SEPK_SIDE_STREAM"] = "0"
        step(mixture, sources)
        sync()
        timed.enabled = True
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(mixture, sources)
        sync()
        elapsed_instr = time.perf_counter() - t1
        timed.enabled = False
        if side_prev is None:
            del os.environ["SEPK_SIDE_STREAM"]
        else:
            os.environ["SEPK_SIDE_STREAM"] = side_prev
    # Third pass (N = 1 only): the same K steps with sep_pw_gemm on the fp32 MFMA instruction instead of the default exact
    # bf16 three-way split, reported beside the headline so that both arithmetics are on record from the same process.
    arith_name = sepkernels.gemm_arith_name()
    elapsed_f32 = None
    if world == 1 and arith_name != "f32" and not args.no_f32_pass:
        sepkernels.set_gemm_arith("f32")
        step(mixture, sources)
        sync()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            step(mixture, sources)
        sync()
        elapsed_f32 = time.perf_counter() - t2
        sepkernels.set_gemm_arith(arith_name)
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()

    F = num_frames(T_SAMPLES, PAPER["kernel_size"], PAPER["stride"])
    frames_per_step = world * args.batch * F
    value = frames_per_step * args.steps / elapsed
    fl_frame, by_frame = 3 * flops_per_frame(PAPER), 3 * bytes_per_frame(PAPER)

    if rank == 0:
        out = {
            "metric": "separated audio frames/sec (fwd+bwd), Conv-TasNet 2-spk 4s@8kHz", "value": value, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Conv-TasNet paper-best (N=512,L=16,B=128,H=512,Sc=128,P=3,X=8,R=3) 2-spk, 4 s @ 8 kHz "
                                   "synthetic mixtures, {} utterances/GPU, fwd + PIT(NegSI-SDR) + bwd + clip(5) + Adam".format(args.batch),
                       "global_batch": world * args.batch, "frames_per_utterance": F, "parallelism": "dp{}".format(world),
                       "utt_per_s": value / F, "samples_per_s": value / F * T_SAMPLES, "final_loss": float(loss),
                       "gemm_arith": arith_name + {
                           "bf16x6": " (fp32 operands split exactly into 3 bf16 parts, 6 of 9 part products on the bf16 MFMA, fp32 "
                                     "accumulation; error vs fp64 at the fp32-MFMA path's level)",
                           "f16x3": " (1x1-conv GEMMs: fp32 operands scaled by exact powers of two -- one for the weights, one per "
                                    "frame column -- and split into 2 fp16 parts, 3 of 4 part products on the fp16 MFMA, fp32 "
                                    "accumulation; weight gradients: exact 3-part bf16 split; error vs fp64 at the fp32-MFMA path's level)",
                           "f32": " (v_mfma_f32_32x32x2_f32)"}[arith_name]},
            "step_roofline": {"mfma_frac": value / world * fl_frame / (FP32_MFMA_PEAK_TFLOPS * 1e12),
                              "hbm_frac": value / world * by_frame / (HBM_PEAK_TBS * 1e12),
                              "algorithmic_flop_per_frame": fl_frame, "algorithmic_bytes_per_frame": by_frame},
        }
        if elapsed_f32 is not None:
            out["fp32_mfma_pass"] = {"value": frames_per_step * args.steps / elapsed_f32, "unit": "frames/s",
                                     "ms_per_step": 1e3 * elapsed_f32 / args.steps,
                                     "what": "same process, same K steps, SEP_ARITH_F32 for every sep_pw_gemm / sep_pw_wgrad"}
        if not args.no_kernel_timing:
            n, ms, fl = timed.summary("pw_gemm")
            nw, msw, flw = timed.summary("pw_wgrad")
            ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            out["roofline"] = {"bound": "mfma", "kernel": "pw_gemm_direct_kernel", "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": ach / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
                               "launches_per_step": n / args.steps, "avg_launch_ms": ms / max(n, 1),
                               "flop_per_launch_avg": fl / max(n, 1), "share_of_step": ms / (1e3 * elapsed_instr),
                               "measured": "HIP events around every launch, second pass of the same {} steps with the weight "
                                           "gradients on the main stream, i.e. no kernel overlap ({:.2f} ms/step with the events "
                                           "in)".format(args.steps, 1e3 * elapsed_instr / args.steps),
                               "arith": arith_name,
                               "peak_is": "dense fp32 MFMA (v_mfma_f32_32x32x2_f32); achieved = algorithmic fp32 flop / time"}
            out["roofline"].update(pmc_traffic(timed.variants))
            achw = flw / (msw * 1e-3) / 1e12 if msw > 0 else 0.0
            out["roofline_wgrad"] = {"bound": "mfma", "kernel": "pw_wgrad_split_kernel" if arith_name != "f32" else "pw_wgrad_direct_kernel", "achieved": achw, "peak": FP32_MFMA_PEAK_TFLOPS,
                                     "unit": "TFLOP/s", "frac": achw / FP32_MFMA_PEAK_TFLOPS, "launches_per_step": nw / args.steps,
                                     "avg_launch_ms": msw / max(nw, 1), "share_of_step": msw / (1e3 * elapsed_instr)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
