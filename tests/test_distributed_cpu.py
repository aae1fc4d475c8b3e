"""CPU, world_size 2, gloo: the data-parallel train step (sepkernels/train.py).  Utterances are sharded across ranks,
each rank runs forward + PIT + backward locally, the flat gradient buffer is all-reduced in three asynchronous buckets (one
per TCN block, issued from inside backward), then clip + Adam.
Checked: (1) ranks end with bit-identical parameters; (2) they equal a single-process step on the concatenated
batch (mean of equal-sized rank means == global mean), i.e. the reference's nn.DataParallel result.
Kernels are the CPU emulator here (no GPU in this container); the collective logic is what is under test."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = dict(n_basis=64, kernel_size=16, stride=8, enc_basis="trainable", dec_basis="trainable", enc_nonlinear="relu",
           sep_hidden_channels=128, sep_bottleneck_channels=64, sep_skip_channels=64, sep_kernel_size=3, sep_num_blocks=3,
           sep_num_layers=2, dilated=True, separable=True, causal=False, sep_nonlinear="prelu", sep_norm=True,
           mask_nonlinear="sigmoid", n_sources=2)


def _setup_paths():
    for p in (ROOT, os.path.join(ROOT, "dnn-based_source_separation_amd", "src"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _data():
    g = torch.Generator().manual_seed(7)
    sources = 0.1 * torch.randn(4, 2, 2000, generator=g)
    return sources.sum(1, keepdim=True), sources


def _run_steps(mixture, sources, nsteps, distributed):
    import sepkernels
    from emulator import EmuBackend
    from sepkernels.train import FusedTrainStep
    from models.conv_tasnet import ConvTasNet
    from criterion.sdr import NegSISDR
    from criterion.pit import PIT1d
    sepkernels._set_backend_for_tests(EmuBackend())
    torch.manual_seed(111)
    model = ConvTasNet(**CFG)
    step = FusedTrainStep(model, PIT1d(NegSISDR(), n_sources=2), lr=1e-3, max_norm=5.0, distributed=distributed)
    losses = [step(mixture, sources).item() for _ in range(nsteps)]
    return model.flat_parameters().clone(), losses + [float(step.last_buckets)]


def _worker(rank, world, port, out_dir):
    _setup_paths()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    mixture, sources = _data()
    per = mixture.shape[0] // world
    sl = slice(rank * per, (rank + 1) * per)                       # utterance sharding
    flat, losses = _run_steps(mixture[sl], sources[sl], 2, True)
    assert losses.pop() == (3.0 if os.environ.get("SEPK_DDP_BUCKETS", "1") != "0" else 0.0)   # block 2 + tail, block 1, block 0 + head
    torch.save({"flat": flat, "losses": losses}, os.path.join(out_dir, "rank{}.pt".format(rank)))
    dist.destroy_process_group()


def test_two_rank_step_matches_single_process(tmp_path):
    _setup_paths()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    assert torch.equal(r0["flat"], r1["flat"])                     # replicas stay in lock-step
    import sepkernels
    old = sepkernels.backend()
    try:
        mixture, sources = _data()
        flat, losses = _run_steps(mixture, sources, 2, False)      # single process, global batch
        assert losses.pop() == 0.0
    finally:
        sepkernels._set_backend_for_tests(old)
    assert (r0["flat"] - flat).abs().max() <= 2e-5 * flat.abs().max()
    # the global loss is the mean of the two rank losses
    for k in range(2):
        assert abs(0.5 * (r0["losses"][k] + r1["losses"][k]) - losses[k]) <= 1e-4 * abs(losses[k])
    assert losses[1] < losses[0]                                   # and the optimiser actually descends


# ---------------------------------------------------------------------------------------------------------------------------
# The recorded launch sequence (ABI 23) under data parallelism: the gradient buckets cut the recorded list into segments, a replayed step
# runs segment / all-reduce / segment ... / waits / clip + Adam.  Kernels: the kernel SOURCES on the host simulation (tools/hostsim.py) --
# the emulator has no C ABI to record.  Two ranks, two steps each way (eager; recorded + one replay): same losses, same parameters.
def _recorded_worker(rank, world, port, out_dir, so):
    _setup_paths()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import hostsim
    import sepkernels
    from sepkernels.train import FusedTrainStep
    from models.conv_tasnet import ConvTasNet
    from criterion.sdr import NegSISDR
    from criterion.pit import PIT1d
    g = torch.Generator().manual_seed(7)
    batches = [0.1 * torch.randn(2, 2, 803, generator=g) for _ in range(2)]           # (both ranks draw the same stream; each takes its utterance)
    out = {}
    with hostsim.HostSimBackend(so) as K:
        class Named:
            name = "hostsim"

            def __getattr__(self, attr):
                return getattr(K, attr)
        sepkernels._set_backend_for_tests(Named())
        for recorded in (False, True):
            torch.manual_seed(111)
            model = ConvTasNet(**dict(CFG, sep_num_layers=1))
            step = FusedTrainStep(model, PIT1d(NegSISDR(), n_sources=2), lr=1e-3, max_norm=5.0, distributed=True)
            losses = []
            for i, src in enumerate(batches):
                src = src[rank:rank + 1]
                mix = src.sum(1, keepdim=True).contiguous()
                if recorded and i == 0:
                    assert step.recordable() is None
                    losses.append(float(step.record(mix, src)))
                    assert [m[1:] for m in step._seq_marks][-1][0] == 0 and len(step._seq_marks) == 3
                    assert all(a[0] < b[0] for a, b in zip(step._seq_marks, step._seq_marks[1:])) and step._seq_marks[-1][0] < len(step._seq)
                else:
                    losses.append(float(step(mix, src)))
                assert step.last_buckets == 3 and sum(step.last_bucket_bytes) == 4 * step.gflat.numel()
            assert (step._seq is not None) == recorded
            out[recorded] = (losses, model.flat_parameters().detach().clone())
    torch.save(out, os.path.join(out_dir, "rec_rank{}.pt".format(rank)))
    dist.destroy_process_group()


def test_two_rank_recorded_step_matches_the_eager_two_rank_step(tmp_path):
    import pytest
    _setup_paths()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import hostsim
    if hostsim.compiler() is None:
        pytest.skip("needs clang++ for the host simulation of the kernel sources")
    so = hostsim.build(str(tmp_path))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_recorded_worker, args=(2, port, str(tmp_path), so), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rec_rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rec_rank1.pt"))
    for r in (r0, r1):
        (le, pe), (lr, pr) = r[False], r[True]
        assert le == lr and le[0] != le[1], (le, lr)                # the recorded step is the eager step, launch for launch
        assert (pe - pr).abs().max() <= 2e-7 * pe.abs().max()
    assert torch.equal(r0[True][1], r1[True][1])                   # replicas stay in lock-step
    assert r0[True][0] != r1[True][0]                              # (on different utterances)


# ---------------------------------------------------------------------------------------------------------------------------
# The dual-path separators shard the same way (one process per GPU, utterances split across ranks); their parameters are ordinary
# tensors, so the gradient exchange is torch's DistributedDataParallel (RCCL on the GPUs, gloo here) around the module whose
# 1x1 convolutions / chunking / norms / recurrences are custom autograd Functions on the C ABI.
DPT_CFG = dict(n_basis=32, kernel_size=4, stride=2, enc_basis="trainable", dec_basis="trainable", enc_nonlinear="relu",
               sep_bottleneck_channels=32, sep_hidden_channels=16, sep_chunk_size=12, sep_num_blocks=1, sep_num_heads=4, sep_dropout=0,
               mask_nonlinear="relu", causal=False, n_sources=2)


def _dual_path_steps(mixture, sources, nsteps, ddp):
    import sepkernels
    from emulator import EmuBackend
    from models.dptnet import DPTNet
    from criterion.sdr import NegSISDR
    from criterion.pit import PIT1d
    sepkernels._set_backend_for_tests(EmuBackend())
    torch.manual_seed(111)
    model = DPTNet(**DPT_CFG).double()          # fp64: Adam turns rounding noise on near-zero gradients into O(lr) steps in fp32
    mixture, sources = mixture.double(), sources.double()
    assert not model.kernel_path_problems()
    net = torch.nn.parallel.DistributedDataParallel(model) if ddp else model
    crit = PIT1d(NegSISDR(), n_sources=2)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    losses = []
    for _ in range(nsteps):
        opt.zero_grad()
        loss, _ = crit(net(mixture), sources)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
        opt.step()
        losses.append(loss.item())
    return torch.cat([p.detach().reshape(-1) for p in model.parameters()]), losses


def _dual_path_worker(rank, world, port, out_dir):
    _setup_paths()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    mixture, sources = _data()
    per = mixture.shape[0] // world
    sl = slice(rank * per, (rank + 1) * per)
    flat, losses = _dual_path_steps(mixture[sl], sources[sl], 2, True)
    torch.save({"flat": flat, "losses": losses}, os.path.join(out_dir, "dp_rank{}.pt".format(rank)))
    dist.destroy_process_group()


def test_dual_path_separator_under_distributed_data_parallel(tmp_path):
    _setup_paths()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dual_path_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "dp_rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "dp_rank1.pt"))
    assert torch.equal(r0["flat"], r1["flat"])                     # replicas stay in lock-step
    import sepkernels
    old = sepkernels.backend()
    try:
        mixture, sources = _data()
        flat, losses = _dual_path_steps(mixture, sources, 2, False)      # single process, global batch
    finally:
        sepkernels._set_backend_for_tests(old)
    assert torch.allclose(flat, r0["flat"], rtol=0, atol=1e-9), (flat - r0["flat"]).abs().max()
    for k in range(2):                                              # global mean loss == mean of the two rank means
        assert abs(0.5 * (r0["losses"][k] + r1["losses"][k]) - losses[k]) < 1e-9


# ---------------------------------------------------------------------------------------------------------------------------
# Uneven shards: the last batch of an epoch need not divide by the world size (the reference's nn.DataParallel scatters it unevenly
# and still averages over all of it).  FusedTrainStep(uneven_batches=True) weighs every rank by its utterance count.
def _uneven_worker(rank, world, port, out_dir):
    _setup_paths()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import sepkernels
    from emulator import EmuBackend
    from sepkernels.train import FusedTrainStep
    from models.conv_tasnet import ConvTasNet
    from criterion.sdr import NegSISDR
    from criterion.pit import PIT1d
    sepkernels._set_backend_for_tests(EmuBackend())
    mixture, sources = _data()
    sl = slice(0, 3) if rank == 0 else slice(3, 4)                 # 3 utterances here, 1 there
    torch.manual_seed(111)
    model = ConvTasNet(**CFG)
    step = FusedTrainStep(model, PIT1d(NegSISDR(), n_sources=2), lr=1e-3, max_norm=5.0, distributed=True, uneven_batches=True)
    for _ in range(2):
        step(mixture[sl], sources[sl])
    torch.save({"flat": model.flat_parameters().clone(), "bucket_bytes": list(step.last_bucket_bytes)}, os.path.join(out_dir, "un_rank{}.pt".format(rank)))
    dist.destroy_process_group()


def test_uneven_shards_equal_the_global_batch(tmp_path):
    _setup_paths()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_uneven_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "un_rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "un_rank1.pt"))
    assert torch.equal(r0["flat"], r1["flat"])
    assert len(r0["bucket_bytes"]) == 3 and sum(r0["bucket_bytes"]) == 4 * r0["flat"].numel()      # three buckets, every gradient once
    import sepkernels
    old = sepkernels.backend()
    try:
        mixture, sources = _data()
        flat, _ = _run_steps(mixture, sources, 2, False)           # single process, all four utterances: the true global mean
    finally:
        sepkernels._set_backend_for_tests(old)
    assert (r0["flat"] - flat).abs().max() <= 2e-5 * flat.abs().max()


# A rank that raises in the middle of a step must bring the job DOWN, not leave its peers waiting in an all-reduce: the failing process
# exits non-zero without touching another collective, and the launcher (torch.multiprocessing.spawn here, torchrun in the recipes) ends
# the others.  What is checked: the job fails within seconds instead of hanging until the collective's time-out.
def _failing_worker(rank, world, port):
    _setup_paths()
    import datetime
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))
    torch.set_num_threads(2)
    import sepkernels
    from emulator import EmuBackend
    from sepkernels.train import FusedTrainStep
    from models.conv_tasnet import ConvTasNet
    from criterion.sdr import NegSISDR
    from criterion.pit import PIT1d
    sepkernels._set_backend_for_tests(EmuBackend())
    mixture, sources = _data()
    torch.manual_seed(111)
    model = ConvTasNet(**CFG)
    step = FusedTrainStep(model, PIT1d(NegSISDR(), n_sources=2), lr=1e-3, max_norm=5.0, distributed=True)
    sl = slice(2 * rank, 2 * rank + 2)
    step(mixture[sl], sources[sl])
    if rank == 1:
        raise RuntimeError("rank 1 fails in its second step (e.g. a corrupt utterance)")
    step(mixture[sl], sources[sl])                                  # rank 0 is inside the gradient exchange when its peer dies


def test_a_failing_rank_ends_the_job_instead_of_hanging_it():
    import time
    import pytest
    _setup_paths()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    t0 = time.time()
    with pytest.raises(Exception) as err:
        mp.spawn(_failing_worker, args=(2, port), nprocs=2, join=True)
    assert "rank 1 fails" in str(err.value)
    assert time.time() - t0 < 300, "the surviving rank was left waiting in its all-reduce"


def test_bench_gpus_flag_starts_the_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the driver's command shape) must itself start two ranks: on this GPU-less box
    with SEPK_BENCH_BACKEND=gloo it is the launcher dry run (tests' emulator, flagged `dry_run`), and the LAST stdout line is one compact
    JSON object with n_gpus == 2 that fits the driver's 8 KB stdout window with room to spare."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["SEPK_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = r.stdout.strip().splitlines()[-1]
    assert len(line) < 3072
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 4
    assert out["dry_run"] and out["value"] > 0 and out["ms_per_step"] > 0
    for k in ("metric", "unit", "steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in out


def test_bench_eight_ranks_dry_run_has_the_driver_run_shape():
    """`python bench.py --gpus 8 --steps K --warmup W` exactly as the driver's scaling run issues it (default batch: 16 utterances per
    rank), on this GPU-less box as the gloo dry run: eight ranks rendezvous on 127.0.0.1, every rank steps its own shard, the gradient
    exchange goes out in three buckets (one per TCN block), the timing is the max over ranks, and stdout carries ONE JSON line -- rank 0's
    -- with the whole-job figures (global batch 128).  So that the first real `--gpus 8` cannot fail on plumbing; no curve is claimed."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SEPK_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    json_lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(json_lines) == 1 and r.stdout.strip().splitlines()[-1] == json_lines[0]
    out = json.loads(json_lines[0])
    assert out["n_gpus"] == 8 and out["config"]["parallelism"] == "dp8" and out["config"]["global_batch"] == 128 and out["config"]["per_gpu_batch"] == 16
    assert out["config"]["ddp_buckets"] == 3 and out["scaling"] == "weak" and out["dry_run"]
    assert out["steps"] == 2 and out["warmup"] == 1 and out["value"] > 0
    assert abs(out["value"] - 128 * 499 / (out["ms_per_step"] * 1e-3)) <= 1e-3 * out["value"]      # whole-job frames (T = 4000 -> 499 frames) / the max-over-ranks step time


def test_bench_without_gpu_fails_loudly():
    """no GPU, no dry-run switch: the bench refuses instead of measuring a fallback"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SEPK_BENCH_BACKEND")}
    if torch.cuda.is_available():
        return
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0 and "no GPU visible" in r.stderr
