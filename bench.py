"""
bench.py -- separated audio frames/sec (fwd + SI-SDR/PIT + bwd [+ all-reduce] + clip + Adam) of Conv-TasNet paper-best
(N512 L16 B128 H512 Sc128 P3 X8 R3, 2 speakers) on synthetic 4 s @ 8 kHz mixtures, 16 utterances per GPU.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus N ...            (WORLD_SIZE unset: re-executes itself under torch.distributed.run with N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE compact JSON line (< 3 KB) as the LAST line of stdout; everything else (per-kernel table, traffic table, the legs'
descriptions) goes to profiles/bench_detail.json (and gpurun_out/bench_detail.json when that directory exists), named in the line.

A "step" is one pass of the hot path over one batch resident in HBM.  `roofline` is for ONE kernel instance: the launch class with the
largest share of the step's kernel time (HIP events around every launch on the launch stream, in a second pass of the same K steps);
achieved = algorithmic bytes of those launches / their summed duration against 8 TB/s.  `cpu_baseline` is the reference's CPU path
(the unmodified reference where /root/reference exists, else oracle/fast_port.py) on this box's host cores, bounded sample, N = 1 only.
The legs outside the timed region live in tools/bench_legs.py.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "dnn-based_source_separation_amd", "src"), os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench_legs as legs  # noqa: E402
from bench_legs import (PAPER, T_SAMPLES, PER_GPU_BATCH, HBM_PEAK_TBS, MFMA_PER_PRODUCT, FP32_MFMA_PEAK_TFLOPS, num_frames,  # noqa: E402
                        flops_per_frame, bytes_per_frame)

# the launcher / rendezvous / timing plumbing on a box without a GPU (tests/test_distributed_cpu.py): tiny Conv-TasNet on the tests'
# CPU emulator of the kernels, flagged in the line; never a measurement
DRY_CFG = dict(PAPER, n_basis=64, sep_hidden_channels=128, sep_bottleneck_channels=64, sep_skip_channels=64, sep_num_blocks=3, sep_num_layers=1)      # three TCN blocks = three gradient buckets, like the paper-best model
DRY_T = 4000


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n):
    """`python bench.py --gpus N` with no launcher around it (the driver's command shape): start N ranks, one per GPU, under
    torch.distributed.run on 127.0.0.1 and hand their output through; the JSON line is rank 0's."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def write_detail(detail):
    """the full record beside the compact line (one file per rank count); returns the repository-relative path that was written (or None)"""
    written = None
    name = "bench_detail.json" if detail["n_gpus"] == 1 else "bench_detail_n{}.json".format(detail["n_gpus"])
    for rel in (os.path.join("profiles", name), os.path.join("gpurun_out", name)):
        path = os.path.join(ROOT, rel)
        if rel.startswith("gpurun_out") and not os.path.isdir(os.path.dirname(path)):
            continue
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                json.dump(detail, f, indent=1)
            written = written or rel
        except OSError as e:
            print("bench.py: could not write {}: {}".format(rel, e), file=sys.stderr)
    return written


def _r(x, n=4):
    return None if x is None else float("{:.{}g}".format(x, n + 2)) if isinstance(x, float) else x


def compact_line(detail):
    """The line the driver parses: the contract's keys + roofline (ONE kernel instance) + step_roofline + fp32_mfma_pass + cpu_baseline +
    hipified_baseline, short strings only."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    out = {k: detail[k] for k in keep}
    c = detail["config"]
    out["config"] = {k: c[k] for k in ("workload", "global_batch", "per_gpu_batch", "parallelism", "rccl_ranks", "launch", "final_loss") if k in c}
    if detail.get("ranks") and detail["n_gpus"] > 1:
        out["config"]["ddp_buckets"] = detail["ranks"].get("ddp_buckets")
    r = detail.get("roofline")
    if r:
        out["roofline"] = {k: _r(r.get(k)) for k in ("kernel", "launch_class", "family", "bound", "achieved", "peak", "unit", "frac", "frac_8d", "family_frac", "family_share", "traffic",
                                                      "traffic_over_algorithmic", "avg_launch_us", "launches_per_step", "share_of_kernel_time",
                                                      "algorithmic_bytes_per_launch")}
    rf = detail.get("roofline_family")
    if rf:
        out["roofline_family"] = {k: {"frac": _r(v["frac"]), "share": _r(v["share"])} for k, v in rf.items()}
    s = detail.get("step_roofline")
    if s:
        out["step_roofline"] = {k: _r(s.get(k)) for k in ("hbm_frac", "matrix_pipe_frac", "traffic_GB_per_step", "traffic_over_algorithmic")}
    f = detail.get("fp32_mfma_pass")
    if f:
        out["fp32_mfma_pass"] = {"value": _r(f["value"]), "ms_per_step": _r(f["ms_per_step"]), "frac": _r(f.get("frac"))}
    cb = detail.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {"value": _r(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": cb["sample_short"]}
    hb = detail.get("hipified_baseline")
    if hb:
        out["hipified_baseline"] = {"value": _r(hb.get("value")), "ms_per_step": _r(hb.get("ms_per_step"))}
    if detail.get("slowest_kernels"):
        out["slowest_kernels"] = detail["slowest_kernels"]
    for k in ("dry_run", "detail"):
        if detail.get(k):
            out[k] = detail[k]
    for k in ("value", "ms_per_step"):
        out[k] = _r(out[k], 6)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU (default: 16 for the Conv-TasNet configs, the recipe's batch size for the dual-path ones)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-f32-pass", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc passes that measure roofline.traffic (N = 1 only)")
    ap.add_argument("--no-stock", action="store_true", help="skip the hipified_baseline leg (stock torch.nn modules on the same device)")
    ap.add_argument("--eager", "--no-graph", dest="eager", action="store_true",
                    help="launch every kernel from Python instead of replaying the recorded launch sequence (N = 1; same as SEPK_SEQUENCE=0)")
    ap.add_argument("--config", default="convtasnet2", choices=["convtasnet2", "sinkpit4", "causal", "dprnn", "dptnet", "galrnet", "sepformer"],
                    help="convtasnet2 (default) = BASELINE.json configs[1]/[2], the headline; sinkpit4 = configs[4] (paper-best Conv-TasNet, "
                         "4 speakers, SinkPIT(NegSI-SDR, coldness 1, 200 iterations)); dprnn = configs[3] (DPRNN-TasNet N64 L2 F64 H128 K250 P125 B6, batch 2); dptnet / galrnet / sepformer = the reference recipes' "
                         "own sizes of those separators (SURVEY.md section 8 row f4)")
    ap.add_argument("--basis", default="trainable", choices=["trainable", "fourier", "pinv"],
                    help="convtasnet2 only: the filterbank -- trainable / trainable (the headline), Fourier / Fourier with the recipe's two-sided real-valued "
                         "latent, or trainable / pseudo-inverse (the reference README's other two published rows): linear filterbanks run the fused "
                         "kernel sequence on bases derived from their parameters")
    ap.add_argument("--sink-iters", type=int, default=200, help="sinkpit4: Sinkhorn iterations (the tutorial recipe uses 200; the paper's ablation 10)")
    args = ap.parse_args()
    if args.config in ("causal", "dprnn", "dptnet", "galrnet", "sepformer"):
        return legs.bench_dual_path(args)
    if args.batch is None:
        args.batch = PER_GPU_BATCH

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus {} but WORLD_SIZE {}: the launcher and the flag disagree".format(args.gpus, world))
    # nccl = RCCL on ROCm (one rank per GPU).  SEPK_BENCH_BACKEND=gloo exists to exercise the multi-rank code path where there is no
    # second GPU: with SEPK_BENCH_ONE_GPU=1 all ranks share device 0; on a box with no GPU at all it is the dry run described at DRY_CFG.
    backend_name = os.environ.get("SEPK_BENCH_BACKEND", "nccl") if world > 1 else None
    dry = (not torch.cuda.is_available()) and os.environ.get("SEPK_BENCH_BACKEND") == "gloo"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend_name, rank=rank, world_size=world)
    if os.environ.get("SEPK_BENCH_ONE_GPU") == "1":
        local_rank = 0

    import sepkernels
    from sepkernels.train import FusedTrainStep
    from models.conv_tasnet import ConvTasNet
    from criterion.sdr import NegSISDR
    from criterion.pit import PIT1d

    if dry:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from emulator import EmuBackend
        sepkernels._set_backend_for_tests(EmuBackend())
        dev = torch.device("cpu")
        torch.set_num_threads(2)
        args.no_kernel_timing = args.no_f32_pass = args.no_pmc = args.no_stock = args.no_cpu_baseline = True
        t_samples, cfg_model = DRY_T, dict(DRY_CFG)
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py: no GPU visible -- the bench measures the HIP path and has no CPU fallback "
                             "(SEPK_BENCH_BACKEND=gloo on a GPU-less box runs the launcher dry run only)")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit("bench.py: rank {} has no GPU ({} visible): one rank per GPU".format(local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        sepkernels.load()          # fail loudly if the HIP library is missing
        t_samples, cfg_model = T_SAMPLES, dict(PAPER)
    timed = legs.TimedBackend(sepkernels.backend())
    sepkernels._backend = timed

    torch.manual_seed(111)
    n_src = 2
    if args.config == "sinkpit4":                      # egs/tutorials/sinkpit_conv-tasnet/train.sh:7,38,42-43
        from criterion.pit import SinkPIT
        n_src = 4
        cfg_model.update(n_sources=4, mask_nonlinear="softmax")
        crit = SinkPIT(NegSISDR(), n_sources=4, coldness=1.0, iteration=args.sink_iters)
    else:
        crit = PIT1d(NegSISDR(), n_sources=2)
    if args.basis == "fourier":
        cfg_model.update(enc_basis="Fourier", dec_basis="Fourier", window_fn="hann", enc_onesided=0, enc_return_complex=0)
    elif args.basis == "pinv":
        cfg_model.update(dec_basis="pinv")
    model = ConvTasNet(**cfg_model).to(dev)
    assert args.basis == "trainable" or model.fused_derived
    step = FusedTrainStep(model, crit, lr=1e-3, max_norm=5.0, time_collectives=world > 1 and not dry)   # recipe defaults: adam 1e-3, clip 5 (train.sh:50-57)
    g = torch.Generator().manual_seed(111 + rank)
    sources = (0.1 * torch.randn(args.batch, n_src, t_samples, generator=g)).to(dev)
    mixture = sources.sum(1, keepdim=True).contiguous()

    def sync():
        if world > 1:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    def timed_steps(n):
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            out = step(mixture, sources)
        sync()
        return time.perf_counter() - t0, out

    def instrumented_pass(n):
        """The same n steps with every kernel launch bracketed by HIP events.  Kept out of the headline region: the ~300
        event pairs per step cost ~1 ms/step of dispatch bubbles (measured A/B).  The weight gradients go back onto the main
        stream for this pass: on their side stream they overlap the input-gradient chain, and an event pair around a launch
        would time the overlap, not the kernel."""
        side_prev = os.environ.get("SEPK_SIDE_STREAM")
        os.environ["SEPK_SIDE_STREAM"] = "0"
        seq, step._seq = step._seq, None                  # the event brackets live in the Python launch wrappers: eager for this pass
        auto, step.auto_record = step.auto_record, False
        step(mixture, sources)
        timed.reset()
        timed.enabled = True
        el, _ = timed_steps(n)
        timed.enabled = False
        step._seq, step.auto_record = seq, auto
        if side_prev is None:
            del os.environ["SEPK_SIDE_STREAM"]
        else:
            os.environ["SEPK_SIDE_STREAM"] = side_prev
        return el

    # N = 1: the step is recorded ONCE (sepkernels.Sequence: every launch of forward + PIT + backward + clip + Adam through the library's own
    # entry points, no hipGraph) and replayed by one sep_run_sequence call per step -- the launches leave a C loop instead of ~360 Python
    # wrappers.  Same kernels, same arguments, same order as the eager step (losses agree to the last bit: tests/test_gpu_model.py).
    # SEPK_SEQUENCE=0 or --eager: the eager step.  N > 1: the same list in segments, one per gradient bucket, the bucket's asynchronous
    # all-reduce issued between them (FusedTrainStep._replay).
    use_seq = (not dry and not args.eager and os.environ.get("SEPK_SEQUENCE", "1") != "0" and step.recordable() is None)
    seq_note = None
    done = 0
    if use_seq:
        try:
            loss = step.record(mixture, sources)             # a training step: the first of the W warm-up steps
            done = 1
        except Exception as e:                               # noqa: BLE001 -- reported in the line, the measurement goes on eagerly
            seq_note = "eager (recording failed: {}: {})".format(type(e).__name__, str(e)[:120])
            print("bench.py: " + seq_note, file=sys.stderr)
            use_seq = False
            step._seq = None
    for _ in range(max(0, args.warmup - done)):
        loss = step(mixture, sources)
    elapsed, loss = timed_steps(args.steps)              # THE timed region: K steps, nothing else in it
    my_elapsed = elapsed
    final_loss = float(loss)                             # NOW: a replayed step returns the recorded step's loss buffer, which a later recording (fp32 pass) replaces

    arith_name = sepkernels.gemm_arith_name()
    by_kernel = roof_g = roof_w = None
    if not args.no_kernel_timing:
        el_i = instrumented_pass(args.steps)
        by_kernel = legs.roofline_by_kernel(timed, args.steps, arith_name)
        roof_g = legs.kernel_roofline(timed, "pw_gemm", arith_name, args.steps, el_i, "sep_pw_gemm launches of the step (all template instances)")
        roof_w = legs.kernel_roofline(timed, "pw_wgrad", legs.wgrad_arith(arith_name), args.steps, el_i, "sep_pw_wgrad launches of the step (all template instances)")
    # N = 1 only: the same K steps with sep_pw_gemm / sep_pw_wgrad on the fp32 MFMA instruction (v_mfma_f32_32x32x2_f32), i.e. the
    # reference's own arithmetic, reported beside the headline (peak 157.3 TFLOP/s)
    F = num_frames(t_samples, PAPER["kernel_size"], PAPER["stride"])
    fl_frame, by_frame = 3 * flops_per_frame(cfg_model), 3 * bytes_per_frame(cfg_model)
    f32_pass = None
    if world == 1 and arith_name != "f32" and not args.no_f32_pass:
        sepkernels.set_gemm_arith("f32")
        step._seq = None
        if use_seq:
            step.record(mixture, sources)                  # the recorded launches carry the arithmetic: record the step again
        else:
            step(mixture, sources)
        el_f32, _ = timed_steps(args.steps)
        v32 = args.batch * F * args.steps / el_f32
        f32_pass = {"value": v32, "unit": "frames/s", "ms_per_step": 1e3 * el_f32 / args.steps, "dtype": "f32",
                    "frac": v32 * fl_frame / (FP32_MFMA_PEAK_TFLOPS * 1e12), "frac_of": "step FLOP x frames/s / 157.3 TFLOP/s (dense fp32 MFMA)",
                    "what": "same process, same K steps, SEP_ARITH_F32 (v_mfma_f32_32x32x2_f32) for every sep_pw_gemm / sep_pw_wgrad"}
        sepkernels.set_gemm_arith(arith_name)
        step._seq = None

    rank_ms = [1e3 * my_elapsed / args.steps]
    comm_ms = [None]
    if world > 1:
        # exposed time of the gradient exchange in the LAST timed step on every rank: HIP events on the compute stream around its waits on
        # the (asynchronous, bucketed) all-reduces -- everything else of the exchange ran under the backward pass
        mine = None if dry else step.exposed_comm_ms()
        allc = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allc, torch.tensor([-1.0 if mine is None else mine], device=dev, dtype=torch.float64))
        comm_ms = [c.item() for c in allc]
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
        allt = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allt, torch.tensor([my_elapsed], device=dev, dtype=torch.float64))
        rank_ms = [1e3 * t.item() / args.steps for t in allt]

    frames_per_step = world * args.batch * F
    value = frames_per_step * args.steps / elapsed

    if rank == 0:
        per, pipe = MFMA_PER_PRODUCT[arith_name]
        paper_name = "Conv-TasNet paper-best (N=512,L=16,B=128,H=512,Sc=128,P=3,X=8,R=3)"
        if args.config == "convtasnet2":
            workload = "{} 2-spk, 4 s @ 8 kHz synthetic mixtures, {} utterances/GPU, fwd + PIT(NegSI-SDR) + bwd + clip(5) + Adam".format(paper_name, args.batch)
            if args.basis != "trainable":
                workload += " -- filterbank: " + {"fourier": "Fourier / Fourier (two-sided, real-valued latent)", "pinv": "trainable / pseudo-inverse"}[args.basis]
        else:
            workload = ("{} 4-spk, softmax mask, 4 s @ 8 kHz synthetic mixtures, {} utterances/GPU, fwd + SinkPIT(NegSI-SDR, coldness 1, "
                        "{} iterations) + bwd + clip(5) + Adam").format(paper_name, args.batch, args.sink_iters)
        detail = {
            "metric": "separated audio frames/sec (fwd+bwd), Conv-TasNet 2-spk 4s@8kHz" if args.config == "convtasnet2" else
                      "separated audio frames/sec (fwd+bwd), Conv-TasNet 4-spk 4s@8kHz with Sinkhorn-PIT (BASELINE configs[4])",
            "value": value, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f16x3": "f32 (f16x3: fp32 operands/accumulators, products from a 2-part fp16 split on the fp16 MFMA; fp32-MFMA pass beside it)",
                      "bf16x6": "f32 (bf16x6: exact 3-part bf16 split on the bf16 MFMA)", "f32": "f32"}[arith_name],
            "data": "synthetic",
            "config": {"workload": workload, "global_batch": world * args.batch, "per_gpu_batch": args.batch, "frames_per_utterance": F,
                       "parallelism": "dp{}".format(world), "rccl_ranks": world if backend_name == "nccl" else 0,
                       "utt_per_s": value / F, "samples_per_s": value / F * t_samples, "final_loss": final_loss,
                       "launch": "recorded sequence, one sep_run_sequence call per step" if use_seq else (seq_note or "eager (one Python call per launch)"),
                       "gemm_arith": arith_name},
            "step_roofline": {"hbm_frac": value / world * by_frame / (HBM_PEAK_TBS * 1e12),
                              "matrix_pipe_frac": value / world * fl_frame / (pipe / per * 1e12),
                              "matrix_pipe_peak_tflops_equiv": pipe / per,
                              "algorithmic_flop_per_frame": fl_frame, "algorithmic_bytes_per_frame": by_frame,
                              "what": "whole step against both roofs: algorithmic bytes (SURVEY.md 8d) x frames/s / 8 TB/s, and algorithmic fp32 "
                                      "FLOP x frames/s / (dense MFMA peak of the issued instruction / MFMAs per fp32 product); the binding one is HBM"},
            "ranks": {"backend": backend_name, "rccl_ranks": world if backend_name == "nccl" else 0, "ms_per_step_per_rank": rank_ms,
                      "ddp_buckets": getattr(step, "last_buckets", None), "bucket_bytes": getattr(step, "last_bucket_bytes", None),
                      "exposed_allreduce_ms_per_rank": comm_ms,
                      "expected": "19.94 MB of fp32 gradients per step in 3 buckets (one per TCN block, last block first); on 8 MI355X a ring "
                                  "all-reduce moves 2 x 7/8 x 19.94 MB = 34.9 MB per rank over xGMI links of ~153 GB/s per direction: ~0.23 ms if fully "
                                  "exposed, < 1.5 % of the step.  Weak scaling, 16 utterances per rank."},
        }
        if dry:
            detail["dry_run"] = ("NOT a measurement: no GPU on this box; tiny Conv-TasNet on the tests' CPU emulator, T = {} -- exercises the launcher, "
                                 "rendezvous, barrier and max-over-ranks timing only").format(t_samples)
            detail["config"]["workload"] = "dry run (launcher test): tiny Conv-TasNet (N=64,B=64,H=128,Sc=64,X=1,R=3) on the CPU emulator, {} utterances/rank".format(args.batch)
        if f32_pass is not None:
            detail["fp32_mfma_pass"] = f32_pass
        per_kernel_traffic = None
        if world == 1 and args.config == "convtasnet2" and not args.no_pmc and by_kernel is not None:
            torch.cuda.empty_cache()                      # the counter passes run this command again in child processes (~10 GB each of the 288)
            per_kernel_traffic, err = legs.measure_pmc_traffic(args.batch)
            if per_kernel_traffic:
                live, table, step_total = legs.traffic_block(per_kernel_traffic)
                detail["hbm_traffic"] = {"step_total_GB": step_total / 1e9, "over_algorithmic": step_total / (args.batch * F * by_frame), "per_kernel": table,
                                         "groups": {k: {"bytes_per_launch": v[0], "launches_per_step": v[1]} for k, v in live.items()},
                                         "source_hash": legs.kernel_source_hash(),
                                         "what": "HBM bytes of one step (forward + loss + backward + clip + Adam, one stream), rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, "
                                                 "separate passes of this command run by bench.py after the timed region; kernels above 1 MB per launch listed"}
                detail["step_roofline"]["traffic_GB_per_step"] = step_total / 1e9
                detail["step_roofline"]["traffic_over_algorithmic"] = step_total / (args.batch * F * by_frame)
            else:
                detail["hbm_traffic"] = {"error": err}
                print("bench.py: traffic passes failed: {}".format(err), file=sys.stderr)
        if by_kernel is not None:
            legs.attach_kernel_instances(by_kernel, per_kernel_traffic)
            detail["roofline_by_kernel"] = by_kernel
            detail["roofline_groups"] = {"gemm": roof_g, "wgrad": roof_w}
            dom = legs.dominant_kernel_roofline(by_kernel, arith_name)
            if dom is not None:
                detail["roofline"] = dom
            detail["roofline_family"] = legs.roofline_family(by_kernel)
            detail["slowest_kernels"] = legs.furthest_from_roof(by_kernel)
        if world == 1 and args.config == "convtasnet2" and not args.no_stock:
            detail["inference"] = legs.inference_leg(model, dev)
            detail["hipified_baseline"] = legs.hipified_baseline(mixture, sources)
        if world == 1 and not args.no_cpu_baseline:
            detail["cpu_baseline"] = legs.cpu_baseline()
        if not dry and (not (args.no_kernel_timing and args.no_pmc and args.no_stock and args.no_cpu_baseline and args.no_f32_pass) or world > 1):
            detail["detail"] = write_detail(detail)    # (the counter passes' child runs and other stripped runs do not overwrite the record)
        print(json.dumps(compact_line(detail)), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
