"""
bench.py -- separated audio frames/sec (fwd + SI-SDR/PIT + bwd [+ all-reduce] + clip + Adam) of Conv-TasNet paper-best
(N512 L16 B128 H512 Sc128 P3 X8 R3, 2 speakers) on synthetic 4 s @ 8 kHz mixtures, 16 utterances per GPU.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One JSON line on rank 0.  A "step" is one pass of the hot path over one batch resident in HBM.  `roofline` is for the
dominant kernel group, the 1x1-convolution GEMM `sep_pw_gemm` (pw_gemm_pc_kernel / pw_gemm_coop_kernel: fp32 products from a
two-part fp16 split on v_mfma_f32_32x32x16_f16): its launches are bracketed with HIP events on the launch stream in a second
pass of the same K steps; the roof is min(matrix pipe / MFMAs per product, HBM x FLOP per byte) -- HBM for these shapes --
and achieved = algorithmic bytes of those launches / their summed duration.  `cpu_baseline` is the oracle's functional port
(oracle/fast_port.py, same ATen CPU kernels as the reference; pinned to the live reference by tests/test_oracle_vs_reference_cpu.py)
timed on this box's host cores on a bounded sample (N=1 runs only).
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
F16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA (32x32x16)
HBM_PEAK_TBS = 8.0
# matrix-pipe instructions per fp32 product and the pipe they run on, per arithmetic of the contraction
MFMA_PER_PRODUCT = {"f16x3": (3, F16_MFMA_PEAK_TFLOPS), "bf16x6": (6, F16_MFMA_PEAK_TFLOPS), "f32": (1, FP32_MFMA_PEAK_TFLOPS)}


class TimedBackend:
    """Wraps the kernel facade: in the instrumented pass EVERY launch is bracketed with HIP events (recorded on the current stream =
    the launch stream) and booked under its launch class with its algorithmic work:
      flop        algorithmic fp32 FLOP (the two MFMA kernels)
      bytes_seq   algorithmic HBM bytes of THIS kernel sequence: every tensor the launch has to read or write, once, fp32, valid
                  frames only (weights and per-row vectors not counted)
      bytes_8d    the same launch under SURVEY.md section 8d's convention: forward = 2Bn + 4H + 2Sc rows per layer (conv1: Bn + H, depthwise:
                  2H, heads: H + Bn + 2Sc) and head / tail 2N + Bn + 2 n_src N + n_src S; backward = 2 x forward, booked as input-gradient kernel
                  = its forward counterpart, weight-gradient kernel = its forward counterpart, depthwise backward = 2 x depthwise forward
    """

    def __init__(self, inner):
        self._inner = inner
        self.enabled = False
        self.records = []
        self.name = inner.name

    def __getattr__(self, item):
        fn = getattr(self._inner, item)
        if not callable(fn) or item.startswith("_"):
            return fn

        def call(*a, **kw):
            if not self.enabled:
                return fn(*a, **kw)
            cls, flop, bseq, b8d = self._classify(item, a, kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            self.records.append((e0, e1, cls, flop, bseq, b8d))
            return out
        return call

    @staticmethod
    def _classify(name, a, kw):
        """(launch class, flop, bytes_seq, bytes_8d) -- None where a notion does not apply (latency-bound helpers)"""
        if name == "pw_gemm":
            B, M, K, T = kw["B"], kw["M"], kw["K"], kw["T"]
            col = 4.0 * B * T
            msp = kw.get("m_split", 0)
            rows = K + M
            rows += (M - msp if kw.get("accumulate") else 0)
            rows += ((msp or M) if kw.get("epi_res") is not None else 0)
            rows += (M if kw.get("epi_aux") is not None else 0)
            rows += (2 * K if kw.get("pro_store") is not None else 0)
            pro, ef, tr = kw.get("pro_mode", 0), kw.get("epi_flags", 0), kw.get("trans_a", 0)
            if not tr:
                cls = {0: "conv1", 1: "mask", 2: "bottleneck", 3: "heads"}.get(pro, "gemm")
                r8 = K + M + (M - msp if kw.get("accumulate") else 0)          # 8d: in + out (+ the skip sum read back)
            else:
                cls = "conv1^T" if pro == 4 else "mask^T" if ef & 8 else "heads^T" if kw.get("k_split") or (ef == 0 and M > K) else \
                    "bottleneck^T" if ef & 16 else "gemm^T"
                r8 = K + M                                                       # 8d: the forward counterpart's in + out
                r8 += (K - kw["k_split"]) if kw.get("k_split") else 0            # heads: + the skip sum read back
            return "gemm " + cls, 2.0 * M * K * B * T, rows * col, r8 * col
        if name == "pw_wgrad":
            B, M, N, T = kw["B"], kw["M"], kw["N"], kw["T"]
            col = 4.0 * B * T
            cls = "decoder basis" if kw.get("g_mul") else "heads" if kw.get("x_mode", 0) in (1, 3) and N >= M else \
                "mask" if kw.get("x_mode", 0) == 1 else "bottleneck" if kw.get("x_mode", 0) == 2 else "conv1" if M > N and N >= 64 else "basis / other"
            rows = M + N + (M if kw.get("g_mul") else 0) // max(1, kw.get("g_div", 1))
            return "wgrad " + cls, 2.0 * M * N * B * T, rows * col, (M + N) * col
        if name == "dwconv_fwd":
            B, C, T = a[10], a[11], a[12]
            return "depthwise fwd", None, 2.0 * C * 4 * B * T, 2.0 * C * 4 * B * T
        if name == "dwconv_bwd":
            B, C, T = a[18], a[19], a[20]
            # z is formed again from `a` (sep_dwconv_bwd with bd): dv2 and a in, dv1 out; SURVEY 8d counts the z read as well
            return "depthwise bwd", None, 3.0 * C * 4 * B * T, 4.0 * C * 4 * B * T
        if name == "encoder_fwd":
            B, Tin, N, F = a[4], a[6], a[7], a[10]
            return "encoder fwd", None, 4.0 * B * (N * F + Tin), 4.0 * B * N * F
        if name == "decoder_fwd":
            B, ns, N, F, Tout = a[5], a[6], a[7], a[11], a[13]
            return "decoder fwd", None, 4.0 * B * ((ns + 1) * N * F + ns * Tout), 4.0 * B * (ns * N * F + ns * Tout)
        if name == "decoder_bwd":
            B, ns, N, F, Tout = a[6], a[7], a[8], a[12], a[14]
            return "decoder bwd", None, 4.0 * B * ((2 * ns + 2) * N * F + ns * Tout), 2 * 4.0 * B * (ns * N * F + ns * Tout)
        if name == "head_bwd":
            B, C, T = a[6], a[7], a[8]
            return "head bwd", None, 4.0 * 4 * B * C * T, 4.0 * B * C * T
        if name == "reduce_slabs":
            return "reduce_slabs", None, 4.0 * sum(sg[3] * (sg[4] + 1) for sg in a[0]), None
        if name == "gln_bwd_from_wgrad":
            B, M, N, sps = a[14], a[15], a[16], a[17]
            return "gln sums from wgrad", None, 4.0 * B * M * N * (sps + 1), None
        if name in ("gln_bwd_finalize", "f64_to_f32", "pack_weights", "unfold", "sqnorm", "adam_step", "adam_step_dev", "softmax_ch_fwd", "softmax_ch_bwd"):
            return name, None, None, None
        return name, None, None, None

    def reset(self):
        self.records = []

    def by_class(self):
        out = {}
        for e0, e1, cls, flop, bseq, b8d in self.records:
            r = out.setdefault(cls, {"n": 0, "ms": 0.0, "flop": 0.0, "bytes_seq": 0.0, "bytes_8d": 0.0, "has_bytes": bseq is not None, "has_8d": b8d is not None})
            r["n"] += 1
            r["ms"] += e0.elapsed_time(e1)
            r["flop"] += flop or 0.0
            r["bytes_seq"] += bseq or 0.0
            r["bytes_8d"] += b8d or 0.0
        return out

    def summary(self, key):
        """(launches, ms, flop, bytes_seq) of a group: key = "pw_gemm" | "pw_wgrad" """
        pre = "gemm " if key == "pw_gemm" else "wgrad "
        rs = [r for c, r in self.by_class().items() if c.startswith(pre)]
        return sum(r["n"] for r in rs), sum(r["ms"] for r in rs), sum(r["flop"] for r in rs), sum(r["bytes_seq"] for r in rs)


def roofline_by_kernel(timed, steps, arith_name):
    """One entry per launch class of the step (instrumented pass: HIP events around every launch, all on one stream): launches per step,
    average duration, algorithmic bytes per launch under both conventions (TimedBackend), and the fraction of the roof that bounds the
    class -- min(HBM at 8 TB/s, matrix pipe of the arithmetic the class issues: roof_of()); the weight gradients run the arithmetic
    wgrad_arith() names."""
    out = {}
    tot_ms = sum(r["ms"] for r in timed.by_class().values())
    for cls, r in sorted(timed.by_class().items(), key=lambda kv: -kv[1]["ms"]):
        n, ms = r["n"], r["ms"]
        e = {"launches_per_step": n / steps, "avg_us": 1e3 * ms / n, "ms_per_step": ms / steps, "share_of_kernel_time": ms / tot_ms}
        if r["has_bytes"] and ms > 0:
            e["algorithmic_MB_per_launch"] = r["bytes_seq"] / n / 1e6
            e["GBps"] = r["bytes_seq"] / (ms * 1e-3) / 1e9
            e["hbm_frac"] = e["GBps"] / (HBM_PEAK_TBS * 1e3)
        if r["has_8d"] and ms > 0:
            e["survey_8d_MB_per_launch"] = r["bytes_8d"] / n / 1e6
            e["hbm_frac_8d"] = r["bytes_8d"] / (ms * 1e-3) / 1e9 / (HBM_PEAK_TBS * 1e3)
        if r["flop"] > 0 and ms > 0:
            ar = wgrad_arith(arith_name) if cls.startswith("wgrad") else arith_name
            per, pipe = MFMA_PER_PRODUCT[ar]
            e["tflops_equiv"] = r["flop"] / (ms * 1e-3) / 1e12
            e["matrix_pipe_frac"] = e["tflops_equiv"] / (pipe / per)
            e["bound"] = roof_of(ar, r["flop"], r["bytes_seq"])[0] if r["has_bytes"] else "mfma"
        elif r["has_bytes"]:
            e["bound"] = "hbm"
        else:
            e["bound"] = "latency"
        out[cls] = e
    return out


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


def kernel_source_hash():
    """sha256 over the kernel sources and the ABI header: what a traffic table is valid for"""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "dnn-based_source_separation_amd", "csrc")
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(csrc, fn), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "sepkernels.h"), "rb").read())
    h.update(open(os.path.join(ROOT, "dnn-based_source_separation_amd", "src", "sepkernels", "net.py"), "rb").read())
    return h.hexdigest()[:16]


def measure_pmc_traffic(batch, timeout_s=150):
    """HBM bytes per launch of every kernel of the step, measured NOW: this command's own step (2 + 1 steps, one stream) under
    `rocprofv3 --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE` (separate passes, as MI355X_MICROARCH.md prescribes; counters only, no
    tracing), outside the timed region.  FETCH_SIZE x 2: on gfx950 it tallies the 128-byte requests at 64 bytes -- re-checked on this
    library's access patterns with known byte counts (tools/fetch_calib.hip, profiles/r03b_fetch_calib.txt: contiguous, 64-byte and
    128-byte row segments, global_load and LDS-DMA all report exactly half; WRITE_SIZE reports exactly the bytes written).
    Returns {kernel name: (launches, read bytes, written bytes per launch)} or None when rocprofv3 is not available / fails."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    res = {}
    env = dict(os.environ, SEPK_SIDE_STREAM="0", TMPDIR="/tmp")
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="sepk_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", ctr, "-d", d, "--", sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--batch", str(batch),
                   "--no-cpu-baseline", "--no-f32-pass", "--no-kernel-timing", "--no-pmc", "--no-stock"]
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
            dbs = glob.glob(d + "/**/*.db", recursive=True)
            if not dbs:
                return None
            con = sqlite3.connect(dbs[0])
            for name, n, avg in con.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? group by kernel_name", (ctr,)):
                short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                e = res.setdefault(short, [0, 0.0, 0.0])
                e[0] = max(e[0], n)
                e[1 if ctr == "FETCH_SIZE" else 2] = avg * 1024.0 * (2.0 if ctr == "FETCH_SIZE" else 1.0)
            con.close()
        except (subprocess.SubprocessError, OSError, sqlite3.Error):
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return res


def traffic_block(per_kernel, steps_in_capture=3.0):
    """(per-group traffic for the roofline objects, per-kernel table, step total) from measure_pmc_traffic's result"""
    import re
    groups = {"gemm": [0, 0.0], "wgrad": [0, 0.0]}
    table, total = {}, 0.0
    for name, (n, rd, wr) in sorted(per_kernel.items()):
        total += n * (rd + wr)
        if rd + wr >= 1e6:
            table[name] = {"launches_per_step": n / steps_in_capture, "read_MB": rd / 1e6, "write_MB": wr / 1e6}
        g = "gemm" if re.match(r"pw_gemm_", name) else "wgrad" if re.match(r"pw_wgrad", name) else None
        if g and rd + wr >= 1e6:          # rocprofv3 returns zeros for one kernel of a capture now and then: left out
            groups[g][0] += n
            groups[g][1] += n * (rd + wr)
    return ({g: (v[1] / max(v[0], 1), v[0] / steps_in_capture) for g, v in groups.items()}, table, total / steps_in_capture)


def pmc_traffic(group, live=None):
    """roofline.traffic of a kernel group ("gemm" / "wgrad"): HBM bytes per launch, launch-weighted over the group's launches in a step.
    `live` = this run's own measurement (measure_pmc_traffic); else the committed table profiles/hbm_traffic.json -- used ONLY if it was
    measured on these very kernel sources (its `source_hash` stamp must equal kernel_source_hash(); round 2 once reported a stale copy)."""
    if live is not None:
        per_launch, launches = live[group]
        return {"traffic": per_launch, "traffic_unit": "bytes/launch", "traffic_launches_per_step": launches, "traffic_live": True,
                "traffic_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of this command, run by bench.py itself after the timed region (FETCH_SIZE x2 on gfx950)"}
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if t.get("source_hash") != kernel_source_hash():
            return {"traffic": None, "traffic_note": "profiles/hbm_traffic.json was measured on other kernel sources (stamp {} != {}): not reported".format(
                t.get("source_hash"), kernel_source_hash())}
        g = t["groups"][group]
        return {"traffic": g["bytes_per_launch"], "traffic_unit": "bytes/launch", "traffic_launches_per_step": g["launches_per_step"],
                "traffic_live": False, "traffic_source": t["source"]}
    except (OSError, KeyError, ValueError):
        return {"traffic": None}


def wgrad_arith(arith_name):
    """arithmetic of the weight-gradient products: the forward arithmetic, except that SEPK_WGRAD_F16=0 keeps the exact three-way bf16
    split (wgrad_pc.hip) under f16x3"""
    if arith_name == "f16x3" and os.environ.get("SEPK_WGRAD_F16", "1") == "0":
        return "bf16x6"
    return arith_name


def roof_of(arith, flop, nbytes):
    """Physical roof of a launch mix with `flop` algorithmic fp32 FLOP over `nbytes` algorithmic HBM bytes in arithmetic `arith`:
    min(matrix-pipe peak / MFMAs per product, HBM peak x FLOP per byte), as (bound, roof in TFLOP/s-equivalent, both terms)."""
    per, pipe = MFMA_PER_PRODUCT[arith]
    mfma_roof = pipe / per
    hbm_roof = HBM_PEAK_TBS * flop / nbytes
    return ("hbm" if hbm_roof <= mfma_roof else "mfma"), min(mfma_roof, hbm_roof), mfma_roof, hbm_roof


def kernel_roofline(timed, key, arith, steps, elapsed_instr, names):
    n, ms, fl, by = timed.summary(key)
    if n == 0 or ms <= 0:
        return None
    bound, roof_tf, mfma_roof, hbm_roof = roof_of(arith, fl, by)
    tf = fl / (ms * 1e-3) / 1e12
    gbs = by / (ms * 1e-3) / 1e9
    out = {"kernel": names, "arith": arith, "bound": bound,
           "achieved": gbs if bound == "hbm" else tf, "peak": HBM_PEAK_TBS * 1e3 if bound == "hbm" else mfma_roof,
           "unit": "GB/s" if bound == "hbm" else "TFLOP/s", "frac": tf / roof_tf,
           "roofs_tflops_equiv": {"matrix_pipe": mfma_roof, "hbm": hbm_roof, "what": "matrix pipe = dense MFMA peak of the instruction "
                                  "the arithmetic issues / MFMAs per fp32 product; hbm = 8 TB/s x algorithmic FLOP per algorithmic byte"},
           "achieved_tflops_equiv": tf, "achieved_GBps_algorithmic": gbs,
           "launches_per_step": n / steps, "avg_launch_ms": ms / n, "flop_per_launch_avg": fl / n,
           "algorithmic_bytes_per_launch": by / n, "share_of_step": ms / (1e3 * elapsed_instr),
           "measured": "HIP events around every launch, separate pass of the same {} steps with the weight gradients on the main "
                       "stream, i.e. no kernel overlap ({:.2f} ms/step with the events in)".format(steps, 1e3 * elapsed_instr / steps)}
    return out


REFERENCE_SRC = "/root/reference/src"

# Runs in a child process with the reference's src/ as the ONLY package root (its flat package names -- models, criterion, utils ... -- are
# the ones this repository's drop-in uses too): the unmodified reference classes, timed exactly like the port below.
_REFERENCE_TIMER = r"""
import json, sys, time, torch
sys.path.insert(0, sys.argv[1])
from models.conv_tasnet import ConvTasNet
from criterion.sdr import NegSISDR
from criterion.pit import PIT1d
cfg, T, timed_steps, do16 = json.loads(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
torch.manual_seed(111)
model = ConvTasNet(**cfg)
crit = PIT1d(NegSISDR(), n_sources=2)
g = torch.Generator().manual_seed(111)
def step(mixture, sources):
    for q in model.parameters():
        q.grad = None
    loss, _ = crit(model(mixture), sources)
    loss.backward()
def run(B, cores, n):
    sources = 0.1 * torch.randn(B, 2, T, generator=g)
    mixture = sources.sum(1, keepdim=True)
    torch.set_num_threads(cores)
    step(mixture, sources)
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); step(mixture, sources); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]
all_cores = torch.get_num_threads()
best = min((run(2, c, 2), c) for c in sorted({all_cores, min(all_cores, 32)}, reverse=True))
cores = best[1]
out = {"cores": cores, "dt2": run(2, cores, timed_steps)}
if do16:
    out["dt16"] = run(16, cores, 1)
print("REFJSON" + json.dumps(out))
"""


def _time_reference(timed_steps, do16):
    """{cores, dt2[, dt16]} of the unmodified reference in a child process, or None where /root/reference is absent (the GPU boxes)"""
    import subprocess
    if not os.path.isdir(REFERENCE_SRC):
        return None
    try:
        r = subprocess.run([sys.executable, "-c", _REFERENCE_TIMER, REFERENCE_SRC, json.dumps(PAPER), str(T_SAMPLES), str(timed_steps), str(int(do16))],
                           capture_output=True, text=True, timeout=900, env={k: v for k, v in os.environ.items() if k != "PYTHONPATH"})
        line = [q for q in r.stdout.splitlines() if q.startswith("REFJSON")]
        return json.loads(line[-1][7:]) if line else None
    except (subprocess.SubprocessError, OSError, ValueError):
        return None


def cpu_baseline(timed_steps=5):
    """The reference's CPU path on the host cores, bounded sample: `timed_steps` fwd+PIT+bwd steps of B=2 paper-best utterances (median),
    plus one step at the benchmark's own B=16 when the host has the memory.  kind "reference": the unmodified reference classes
    (/root/reference/src exists: the build container); kind "port": oracle/fast_port.py, the same path on torch.nn.functional (same ATen
    CPU kernels), whose equality with the live reference at paper-best is tests/test_oracle_vs_reference_cpu.py."""
    F = num_frames(T_SAMPLES, 16, 8)
    try:
        free_gb = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE") / 2 ** 30
    except (ValueError, OSError):
        free_gb = 0.0
    do16 = free_gb > 48                   # one B=16 step keeps ~14 GB of activations for autograd
    ref = _time_reference(timed_steps, do16)
    if ref is not None:
        kind, cores, dt, dt16 = "reference", ref["cores"], ref["dt2"], ref.get("dt16")
        what = "the unmodified reference in a child process (/root/reference/src: models.conv_tasnet.ConvTasNet, criterion.pit.PIT1d(criterion.sdr.NegSISDR()))"
    else:
        from oracle import fast_port as FP       # the ONLY oracle import of this file: the cpu_baseline leg
        from models.conv_tasnet import ConvTasNet
        torch.manual_seed(111)
        model = ConvTasNet(**PAPER)
        p = {k: v.detach().clone() for k, v in model.state_dict().items()}
        g = torch.Generator().manual_seed(111)

        def run(B, cores, n):
            sources = 0.1 * torch.randn(B, 2, T_SAMPLES, generator=g)
            mixture = sources.sum(1, keepdim=True)
            torch.set_num_threads(cores)
            FP.train_step(p, PAPER, mixture, sources, dtype=torch.float32)      # oneDNN primitive caches / allocator warm-up
            ts = []
            for _ in range(n):
                t0 = time.perf_counter()
                FP.train_step(p, PAPER, mixture, sources, dtype=torch.float32)
                ts.append(time.perf_counter() - t0)
            return sorted(ts)[len(ts) // 2]

        all_cores = torch.get_num_threads()
        best = min((run(2, c, 2), c) for c in sorted({all_cores, min(all_cores, 32)}, reverse=True))   # oneDNN often peaks below the full core count
        cores = best[1]
        dt = run(2, cores, timed_steps)
        dt16 = run(16, cores, 1) if do16 else None
        torch.set_num_threads(all_cores)
        kind = "port"
        what = ("oracle/fast_port.py (same ATen conv / GroupNorm kernels as the reference modules; equality with the live reference is tested in "
                "tests/test_oracle_vs_reference_cpu.py; /root/reference is not present on this box)")
    out = {"value": 2 * F / dt, "unit": "frames/s", "cores": cores, "kind": kind,
           "sample": "median of {} timed fwd+PIT+bwd steps (after warm-up) of B=2 paper-best utterances, fp32, torch CPU: {}, {:.2f} s/step".format(timed_steps, what, dt)}
    if dt16 is not None:
        out["batch16"] = {"value": 16 * F / dt16, "unit": "frames/s", "s_per_step": dt16, "sample": "one timed step (after one warm-up) at the benchmark's B=16"}
    return out


def inference_leg(model, dev, reps=20):
    """the validation / test regime of the reference's driver (egs/wsj0-mix/common/src/driver.py:166-206,
    277-370) -- ONE utterance of its natural length through the model under torch.no_grad() (no activations kept, no backward packs) --
    as separated frames per second at 4 s and 10 s @ 8 kHz, plus the training batch size for comparison."""
    out = {}
    g = torch.Generator().manual_seed(7)
    for label, B, T in (("1x4s", 1, 32000), ("1x10s", 1, 80000), ("16x4s", 16, 32000)):
        x = (0.1 * torch.randn(B, 1, T, generator=g)).to(dev)
        with torch.no_grad():
            for _ in range(3):
                model(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                model(x)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        F = num_frames(T, PAPER["kernel_size"], PAPER["stride"])
        out[label] = {"frames_per_s": B * F / dt, "ms_per_forward": 1e3 * dt, "real_time_factor": B * T / 8000.0 / dt}
    out["what"] = "forward only under torch.no_grad(), paper-best Conv-TasNet, {} timed passes after 3 warm-up; real_time_factor = seconds of audio separated per second".format(reps)
    return out


def hipified_baseline(mixture, sources, steps=5):
    """SURVEY.md section 8d's "hipified baseline": the same training step on stock PyTorch-ROCm ops (nn.Conv1d / nn.GroupNorm / nn.PReLU /
    nn.ConvTranspose1d modules, autograd, torch.optim.Adam -> MIOpen / rocBLAS / ATen kernels; tools/stock_torch_convtasnet.py), same
    batch, same device, timed after the headline region.  What the device gives without this library's kernels."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import stock_torch_convtasnet as stock
    try:
        dt, nparam = stock.time_train_step(PAPER, mixture, sources, steps=steps, warmup=2)
    except RuntimeError as e:       # e.g. out of memory on a small device
        return {"value": None, "error": str(e)[:200]}
    finally:
        torch.cuda.empty_cache()
    B = mixture.shape[0]
    F = num_frames(T_SAMPLES, PAPER["kernel_size"], PAPER["stride"])
    return {"value": B * F / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt, "parameters": nparam,
            "what": "stock torch.nn modules + autograd + torch.optim.Adam on the same device and batch (MIOpen / rocBLAS / ATen), fp32, {} timed steps after 2 warm-up".format(steps)}


def _dual_path_workloads():
    """--config name -> (class, constructor arguments, recipe batch size, Adam arguments, GFLOP/utterance or None, what runs where)"""
    from models.dprnn_tasnet import DPRNNTasNet
    from models.dptnet import DPTNet
    from models.galrnet import GALRNet
    from models.sepformer import SepFormer
    tr = dict(enc_basis="trainable", dec_basis="trainable")
    return {
        # BASELINE.json configs[3]: egs/wsj0-mix/dprnn-tasnet/train.sh:28-37
        "dprnn": (DPRNNTasNet, dict(n_basis=64, kernel_size=2, stride=1, enc_nonlinear=None, sep_hidden_channels=128, sep_bottleneck_channels=64,
                                    sep_chunk_size=250, sep_hop_size=125, sep_num_blocks=6, sep_norm=True, mask_nonlinear="sigmoid", causal=False,
                                    rnn_type="lstm", n_sources=2, **tr), 2, dict(lr=1e-3), 980.07,
                  "DPRNN-TasNet N=64 L=2 F=64 H=128 K=250 P=125 B=6 (BASELINE configs[3])"),
        # SURVEY.md section 8 row f4, the reference recipes' own sizes: egs/wsj0-mix/{dptnet,galrnet,sepformer}/train.sh
        "dptnet": (DPTNet, dict(n_basis=64, kernel_size=2, stride=1, enc_nonlinear=None, sep_bottleneck_channels=64, sep_hidden_channels=128,
                                sep_chunk_size=250, sep_hop_size=125, sep_num_blocks=6, sep_num_heads=4, sep_norm=True, sep_nonlinear="relu",
                                sep_dropout=0, mask_nonlinear="relu", causal=False, n_sources=2, **tr), 1, dict(lr=1e-3), None,
                   "DPTNet N=64 L=2 F=64 d_ff=128 K=250 P=125 B=6 h=4 (egs/wsj0-mix/dptnet/train.sh:28-44)"),
        "galrnet": (GALRNet, dict(n_basis=64, kernel_size=16, stride=8, enc_nonlinear=None, sep_hidden_channels=128, sep_chunk_size=100,
                                  sep_hop_size=50, sep_down_chunk_size=32, sep_num_blocks=6, sep_num_heads=8, sep_norm=True, sep_dropout=1e-1,
                                  mask_nonlinear="relu", causal=False, n_sources=2, low_dimension=True, **tr), 4, dict(lr=1e-3, weight_decay=1e-6), None,
                    "GALRNet D=64 M=16 H=128 K=100 P=50 Q=32 N=6 J=8 (egs/wsj0-mix/galrnet/train.sh:28-42)"),
        "sepformer": (SepFormer, dict(n_basis=256, kernel_size=16, stride=8, enc_nonlinear="relu", sep_bottleneck_channels=256, sep_chunk_size=250,
                                      sep_hop_size=125, sep_num_blocks=2, sep_num_layers_intra=8, sep_num_layers_inter=8, sep_num_heads_intra=8,
                                      sep_num_heads_inter=8, sep_d_ff_intra=1024, sep_d_ff_inter=1024, sep_norm=True, sep_nonlinear="relu",
                                      sep_dropout=1e-1, mask_nonlinear="relu", causal=False, n_sources=2, **tr), 4, dict(lr=15e-5), None,
                      "SepFormer F=256 L=16 B=256 C=250 P=125 N=2 K=8+8 h=8 d_ff=1024 (egs/wsj0-mix/sepformer/train.sh:27-47)"),
    }


def bench_dual_path(args):
    """The dual-path separators at the sizes of the reference's own recipes, 2 speakers, 4 s @ 8 kHz, the recipe's batch size, one
    GPU: forward + PIT(NegSI-SDR) + backward + clip(5) + Adam (torch.optim.Adam: these models' parameters are ordinary tensors).
    A frame is one encoder frame.  Analysis / synthesis bases, every 1x1 convolution of the separator's two ends, chunking /
    overlap-add, gLN, the LSTM time recurrences, the LSTMs' input projections and the Linear layers behind them (with their input / weight
    gradients: csrc/linear.hip, fp32 on the matrix pipe) are this library's kernels; attention and the transformer feed-forward layers
    are library calls (torch -> hipBLASLt / SDPA), as DESIGN.md states."""
    import sepkernels
    from criterion.sdr import NegSISDR
    from criterion.pit import PIT1d
    sepkernels.load()
    dev = torch.device("cuda", 0)
    cls, cfg, recipe_batch, adam, gflop, label = _dual_path_workloads()[args.config]
    torch.manual_seed(111)
    model = cls(**cfg).to(dev)
    problems = model.kernel_path_problems() if hasattr(model, "kernel_path_problems") else []
    assert not problems, problems
    crit = PIT1d(NegSISDR(), n_sources=2)
    opt = torch.optim.Adam(model.parameters(), **adam)
    B = recipe_batch if args.batch == PER_GPU_BATCH else args.batch
    src = (0.1 * torch.randn(B, 2, T_SAMPLES, generator=torch.Generator().manual_seed(111))).to(dev)
    mix = src.sum(1, keepdim=True).contiguous()

    # (Recording the whole step into a hipGraph was measured in round 3 -- DPRNN-TasNet 57.9 vs 56.5 ms eager, DPTNet 58.4 vs 56.3, GALRNet 16.8 vs
    # 15.9 and a NaN loss from the dropout generator under capture: these steps are no longer launch-bound, and the option is gone.)
    def step():
        opt.zero_grad(set_to_none=True)
        loss, _ = crit(model(mix), src)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
        opt.step()
        return loss.detach()
    launch = "eager (one launch per kernel)"
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    L, S = cfg["kernel_size"], cfg["stride"]
    F = (T_SAMPLES + (S - (T_SAMPLES - L) % S) % S - L) // S + 1
    # Optional second leg (SEPK_GRAPH=1): the same step recorded into a hipGraph and replayed (sepkernels.train.GraphedStep).  Measured
    # (profiles/r04d_dual.txt): replay = eager within 1 % for DPRNN-TasNet and DPTNet (36.9 vs 36.9, 45.2 vs 43.3 ms) -- these steps are
    # bound by their kernels, not by the Python launches -- and the models with dropout (GALRNet, SepFormer) diverge under replay on this
    # stack (loss inf), so the leg is off by default and flags itself invalid there.
    graph_leg = None
    if os.environ.get("SEPK_GRAPH", "0") == "1" and not args.no_graph:
        try:
            from sepkernels.train import GraphedStep
            gopt = torch.optim.Adam(model.parameters(), capturable=True, **adam)
            gstep = GraphedStep(model, crit, gopt, max_norm=5.0)
            gstep.capture(mix, src)
            for _ in range(args.warmup):
                gstep(mix, src)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                gl = gstep(mix, src)
            torch.cuda.synchronize()
            gel = time.perf_counter() - t0
            gl = float(gl)
            graph_leg = {"ms_per_step": 1e3 * gel / args.steps, "value": B * F * args.steps / gel, "unit": "frames/s", "final_loss": gl,
                         "valid": gl == gl and abs(gl) != float("inf"),
                         "what": "the same step (fresh Adam state, parameters where the eager leg left them) as ONE hipGraph launch per step"}
        except Exception as e:                                   # noqa: BLE001 -- a leg that cannot be recorded is reported, not fatal
            graph_leg = {"error": "{}: {}".format(type(e).__name__, e)}
    config = {"workload": "{}, 2 spk, 4 s @ 8 kHz synthetic mixtures, batch {} (recipe default), fwd + PIT(NegSI-SDR) + bwd + clip(5) + Adam".format(label, B),
              "global_batch": B, "frames_per_utterance": F, "parallelism": "dp1", "utt_per_s": B * args.steps / el, "final_loss": float(loss),
              "parameters": model.num_parameters, "launch": launch}
    note = "no roofline: the step is a sequence of library GEMM / attention calls between this library's kernels, none of which dominates"
    roofline = None
    if gflop is not None:
        config["algorithmic_gflop_per_utterance_fwd_bwd"] = gflop
        tf = gflop * 1e9 * B * args.steps / el / 1e12
        # the step's arithmetic is fp32 throughout (the LSTM recurrences on v_mfma_f32_16x16x4 / 4x4x1, projections on rocBLAS fp32): matrix-pipe roof
        roofline = {"kernel": "whole step (rocprofv3, profiles/r04e_dprnn_kernel_stats.md: sep_lstm_fwd / sep_lstm_bwd sweeps 40 % of the kernel time, the dense "
                              "layers of csrc/linear.hip 43 %)", "bound": "mfma", "achieved": tf, "peak": FP32_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": tf / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
                    "what": "algorithmic fp32 FLOP of forward + backward (SURVEY.md 8d: {:.0f} GFLOP per utterance) / step time against the dense fp32 MFMA "
                            "peak; the recurrences are latency-bound chains (one workgroup per 4 or 16 sequences, a barrier per time step), the "
                            "dense layers run at 50 - 70 TFLOP/s".format(gflop)}
        note = "{:.0f} GFLOP per utterance (SURVEY.md 8d): {:.1f} TFLOP/s achieved".format(gflop, tf)
    print(json.dumps({
        "metric": "separated audio frames/sec (fwd+bwd), {} 2-spk 4s@8kHz".format("DPRNN-TasNet" if args.config == "dprnn" else cls.__name__) +
                  (" (BASELINE configs[3])" if args.config == "dprnn" else ""), "value": B * F * args.steps / el, "unit": "frames/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": config, "roofline": roofline, "roofline_note": note,
        "graph_replay": graph_leg}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="utterances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-f32-pass", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc passes that measure roofline.traffic (N = 1 only)")
    ap.add_argument("--no-stock", action="store_true", help="skip the hipified_baseline leg (stock torch.nn modules on the same device)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying the captured step (N = 1)")
    ap.add_argument("--config", default="convtasnet2", choices=["convtasnet2", "sinkpit4", "dprnn", "dptnet", "galrnet", "sepformer"],
                    help="convtasnet2 (default) = BASELINE.json configs[1]/[2], the headline; sinkpit4 = configs[4] (paper-best Conv-TasNet, "
                         "4 speakers, SinkPIT(NegSI-SDR, coldness 1, 200 iterations)); dprnn = configs[3] (DPRNN-TasNet N64 L2 F64 H128 K250 P125 B6, batch 2); dptnet / galrnet / sepformer = the reference recipes' "
                         "own sizes of those separators (SURVEY.md section 8 row f4)")
    args = ap.parse_args()
    if args.config in ("dprnn", "dptnet", "galrnet", "sepformer"):
        return bench_dual_path(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend_name = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # nccl = RCCL on ROCm (one rank per GPU).  SEPK_BENCH_BACKEND=gloo + SEPK_BENCH_ONE_GPU=1 exist only to exercise the
        # multi-rank code path on a single-GPU box (all ranks on device 0, all-reduce through the host).
        backend_name = os.environ.get("SEPK_BENCH_BACKEND", "nccl")
        dist.init_process_group(backend_name, rank=rank, world_size=world)
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
    cfg_model = dict(PAPER)
    n_src = 2
    if args.config == "sinkpit4":                      # egs/tutorials/sinkpit_conv-tasnet/train.sh:7,38,42-43
        from criterion.pit import SinkPIT
        n_src = 4
        cfg_model.update(n_sources=4, mask_nonlinear="softmax")
        crit = SinkPIT(NegSISDR(), n_sources=4, coldness=1.0, iteration=200)
    else:
        crit = PIT1d(NegSISDR(), n_sources=2)
    model = ConvTasNet(**cfg_model).to(dev)
    step = FusedTrainStep(model, crit, lr=1e-3, max_norm=5.0, time_collectives=world > 1)   # recipe defaults: adam 1e-3, clip 5 (train.sh:50-57)
    g = torch.Generator().manual_seed(111 + rank)
    sources = (0.1 * torch.randn(args.batch, n_src, T_SAMPLES, generator=g)).to(dev)
    mixture = sources.sum(1, keepdim=True).contiguous()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_steps(n):
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            out = step(mixture, sources)
        sync()
        return time.perf_counter() - t0, out

    def instrumented_pass(n):
        """The same n steps with every MFMA-kernel launch bracketed by HIP events.  Kept out of the headline region: the ~300
        event pairs per step cost ~1 ms/step of dispatch bubbles (measured A/B).  The weight gradients go back onto the main
        stream for this pass: on their side stream they overlap the input-gradient chain, and an event pair around a launch
        would time the overlap, not the kernel."""
        side_prev = os.environ.get("SEPK_SIDE_STREAM")
        os.environ["SEPK_SIDE_STREAM"] = "0"
        graph, step._graph = step._graph, None            # the event brackets live in the Python launch wrappers: eager for this pass
        step(mixture, sources)
        timed.reset()
        timed.enabled = True
        el, _ = timed_steps(n)
        timed.enabled = False
        step._graph = graph
        if side_prev is None:
            del os.environ["SEPK_SIDE_STREAM"]
        else:
            os.environ["SEPK_SIDE_STREAM"] = side_prev
        return el

    # SEPK_GRAPH=1 (N = 1): the step (forward, PIT, backward on both streams, clip, Adam: ~400 launches) is captured once into a
    # hipGraph and replayed; capture() itself runs 3 eager steps + the captured one, which count as warm-up.
    # Measured (profiles/r02d_graph_vs_eager.md): replay 19.5 ms vs 19.6 eager on ONE stream, but 20.1 vs 18.6 with the weight
    # gradients on their side stream (the two-branch graph loses the overlap the eager streams have) -- so the capture is opt-in.
    use_graph = world == 1 and not args.no_graph and os.environ.get("SEPK_GRAPH", "0") == "1"
    done = 0
    if use_graph:
        loss = step.capture(mixture, sources)
        done = 4
    for _ in range(max(0, args.warmup - done)):
        loss = step(mixture, sources)
    elapsed, loss = timed_steps(args.steps)              # THE timed region: K steps, nothing else in it
    my_elapsed = elapsed

    arith_name = sepkernels.gemm_arith_name()
    roof = roof_w = by_kernel = None
    if not args.no_kernel_timing:
        el_i = instrumented_pass(args.steps)
        by_kernel = roofline_by_kernel(timed, args.steps, arith_name)
        roof = kernel_roofline(timed, "pw_gemm", arith_name, args.steps, el_i,
                               "sep_pw_gemm: pw_gemm_pc_kernel (K >= 512 or M >= 1024) / pw_gemm_coop_kernel")
        roof_w = kernel_roofline(timed, "pw_wgrad", wgrad_arith(arith_name), args.steps, el_i,
                                 {"f16x3": "sep_pw_wgrad: pw_wgrad_pc16_kernel (256 x 128 tiles; other shapes pw_wgrad_pc_kernel, bf16x6)",
                                  "bf16x6": "sep_pw_wgrad: pw_wgrad_pc_kernel", "f32": "sep_pw_wgrad: pw_wgrad_direct_kernel"}[wgrad_arith(arith_name)])
    # N = 1 only: the same K steps with sep_pw_gemm / sep_pw_wgrad on the fp32 MFMA instruction (v_mfma_f32_32x32x2_f32), i.e. the
    # reference's own arithmetic, reported beside the headline with its own roofline (peak 157.3 TFLOP/s)
    f32_pass = None
    if world == 1 and arith_name != "f32" and not args.no_f32_pass:
        sepkernels.set_gemm_arith("f32")
        step._graph = None
        if use_graph:
            step.capture(mixture, sources)                 # the captured launches carry the arithmetic: record the step again
        else:
            step(mixture, sources)
        el_f32, _ = timed_steps(args.steps)
        f32_pass = {"value": world * args.batch * num_frames(T_SAMPLES, PAPER["kernel_size"], PAPER["stride"]) * args.steps / el_f32,
                    "unit": "frames/s", "ms_per_step": 1e3 * el_f32 / args.steps, "dtype": "f32",
                    "what": "same process, same K steps, SEP_ARITH_F32 (v_mfma_f32_32x32x2_f32) for every sep_pw_gemm / sep_pw_wgrad"}
        if not args.no_kernel_timing:
            el_fi = instrumented_pass(args.steps)
            f32_pass["roofline"] = kernel_roofline(timed, "pw_gemm", "f32", args.steps, el_fi, "sep_pw_gemm: pw_gemm_direct_kernel<..., AR = 0>")
        sepkernels.set_gemm_arith(arith_name)
        step._graph = None

    rank_ms = [1e3 * my_elapsed / args.steps]
    comm_ms = [None]
    if world > 1:
        # exposed time of the gradient exchange in the LAST timed step on every rank: HIP events on the compute stream around its waits on
        # the (asynchronous, bucketed) all-reduces -- everything else of the exchange ran under the backward pass
        mine = step.exposed_comm_ms()
        allc = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allc, torch.tensor([-1.0 if mine is None else mine], device=dev, dtype=torch.float64))
        comm_ms = [c.item() for c in allc]
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
        allt = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allt, torch.tensor([my_elapsed], device=dev, dtype=torch.float64))
        rank_ms = [1e3 * t.item() / args.steps for t in allt]

    F = num_frames(T_SAMPLES, PAPER["kernel_size"], PAPER["stride"])
    frames_per_step = world * args.batch * F
    value = frames_per_step * args.steps / elapsed
    fl_frame, by_frame = 3 * flops_per_frame(cfg_model), 3 * bytes_per_frame(cfg_model)

    if rank == 0:
        per, pipe = MFMA_PER_PRODUCT[arith_name]
        out = {
            "metric": "separated audio frames/sec (fwd+bwd), Conv-TasNet 2-spk 4s@8kHz" if args.config == "convtasnet2" else
                      "separated audio frames/sec (fwd+bwd), Conv-TasNet 4-spk 4s@8kHz with Sinkhorn-PIT (BASELINE configs[4])",
            "value": value, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f16x3": "f32 (f16x3 split emulation: fp32 operands and accumulators, products from two fp16 parts per operand on the fp16 MFMA)",
                      "bf16x6": "f32 (bf16x6 split emulation: exact three-way bf16 split on the bf16 MFMA)", "f32": "f32"}[arith_name],
            "data": "synthetic",
            "config": {"workload": ("Conv-TasNet paper-best (N=512,L=16,B=128,H=512,Sc=128,P=3,X=8,R=3) 2-spk, 4 s @ 8 kHz "
                                    "synthetic mixtures, {} utterances/GPU, fwd + PIT(NegSI-SDR) + bwd + clip(5) + Adam" if args.config == "convtasnet2" else
                                    "Conv-TasNet paper-best, 4 speakers, softmax mask, 4 s @ 8 kHz synthetic mixtures, {} utterances/GPU, fwd + "
                                    "SinkPIT(NegSI-SDR, coldness 1, 200 iterations) + bwd + clip(5) + Adam").format(args.batch),
                       "global_batch": world * args.batch, "frames_per_utterance": F, "parallelism": "dp{}".format(world),
                       "utt_per_s": value / F, "samples_per_s": value / F * T_SAMPLES, "final_loss": float(loss),
                       "launch": "hipGraph replay of the captured step" if use_graph else "eager (one launch per kernel)",
                       "gemm_arith": arith_name + {
                           "bf16x6": " (fp32 operands split exactly into 3 bf16 parts, 6 of 9 part products on the bf16 MFMA, fp32 "
                                     "accumulation; error vs fp64 at the fp32-MFMA path's level)",
                           "f16x3": " (1x1-conv GEMMs: fp32 operands scaled by exact powers of two -- one per weight row, one per "
                                    "frame column -- and split into 2 fp16 parts, 3 of 4 part products on the fp16 MFMA, fp32 "
                                    "accumulation; weight gradients: exact 3-part bf16 split; error vs fp64 at the fp32-MFMA path's level)",
                           "f32": " (v_mfma_f32_32x32x2_f32)"}[arith_name]},
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
                                  "exposed, < 1.5 % of the step; the last bucket (block 0 + head, ~6.6 MB, ~0.1 ms) is the only part that cannot hide "
                                  "under backward.  Weak scaling, 16 utterances per rank."},
        }
        if f32_pass is not None:
            out["fp32_mfma_pass"] = f32_pass
        live = None
        if world == 1 and args.config == "convtasnet2" and not args.no_pmc and (roof is not None or roof_w is not None):
            torch.cuda.empty_cache()                      # the counter passes run this command again in child processes (~10 GB each of the 288)
            per_kernel = measure_pmc_traffic(args.batch)
            if per_kernel:
                live, table, step_total = traffic_block(per_kernel)
                out["hbm_traffic"] = {"step_total_GB": step_total / 1e9, "over_algorithmic": step_total / (args.batch * F * by_frame), "per_kernel": table,
                                      "source_hash": kernel_source_hash(),
                                      "what": "HBM bytes of one step (forward + loss + backward + clip + Adam, one stream), rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, "
                                              "separate passes of this command run by bench.py after the timed region; kernels above 1 MB per launch listed"}
        if roof is not None:
            roof.update(pmc_traffic("gemm", live))
            if roof.get("traffic"):
                roof["traffic_over_algorithmic"] = roof["traffic"] / roof["algorithmic_bytes_per_launch"]
            out["roofline"] = roof
        if roof_w is not None:
            roof_w.update(pmc_traffic("wgrad", live))
            if roof_w.get("traffic"):
                roof_w["traffic_over_algorithmic"] = roof_w["traffic"] / roof_w["algorithmic_bytes_per_launch"]
            out["roofline_wgrad"] = roof_w
        if by_kernel is not None:
            out["roofline_by_kernel"] = by_kernel
        if world == 1 and args.config == "convtasnet2" and not args.no_stock:
            out["inference"] = inference_leg(model, dev)
            out["hipified_baseline"] = hipified_baseline(mixture, sources)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
