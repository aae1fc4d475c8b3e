"""
bench_legs.py -- the legs of bench.py that are NOT the timed region: per-launch HIP-event instrumentation and the per-kernel roofline
table, the rocprofv3 --pmc traffic passes, the CPU baseline (reference / oracle port), the stock-torch ("hipified") baseline, the
inference leg and the dual-path separators' bench.  bench.py keeps the launcher, the timed region and the compact JSON line.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_PY = os.path.join(ROOT, "bench.py")

import torch  # noqa: E402

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
                conv1t = pro == 4 or (pro == 0 and M < K and not (ef & 24) and not kw.get("k_split"))     # GLN_BWD prologue, or the plain product on da (direct)
                cls = ("conv1^T" if ef & 2 else "conv1^T (layer 0: no residual gradient)") if conv1t else "mask^T" if ef & 8 else "heads^T" if kw.get("k_split") or (ef == 0 and M > K) else \
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
        if name == "pw_wgrad_batch":
            rows = [TimedBackend._classify("pw_wgrad", (), c) for c in a[0]]
            return rows[0][0], sum(r[1] for r in rows), sum(r[2] for r in rows), sum(r[3] for r in rows)
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


# launch class -> the ONE template instance that serves it at the paper-best shapes (names as rocprofv3 prints them, for the traffic
# counters and the committed kernel-trace summaries); a class whose instance is not listed gets no counter traffic
KERNEL_OF_CLASS = {
    "gemm conv1^T": "pw_gemm_pc_kernel<2, 2, 4, false, 2, 3, 2>",
    "gemm heads^T": "pw_gemm_coop_kernel<4, 0, true, 0, 2>",
    "gemm heads": "pw_gemm_pc_kernel<4, 1, 3, false, 2, 4, 4>",
    "gemm conv1": "pw_gemm_coop_kernel<2, 0, false, 1, 2>",
    "wgrad heads": "pw_wgrad_pc16_kernel<4, 1, 1, true>",
    "wgrad conv1": "pw_wgrad_pc16_batch_kernel<4, 1, 0>",
    "depthwise fwd": "dwconv_fwd_direct_kernel<0, 2>",
    "depthwise bwd": "dwconv_bwd_row_kernel<0, 4, true>",
    "decoder fwd": "decoder_fwd16_kernel<2>",
    "decoder bwd": "decoder_bwd_kernel<16, 2>",
    "head bwd": "head_bwd_kernel",
    "encoder fwd": "encoder_fwd_l16s8_kernel",
    "gln sums from wgrad": "gln_bwd_from_wgrad_kernel",
    "reduce_slabs": "reduce_slabs_kernel",
}


def attach_kernel_instances(by_kernel, per_kernel_traffic):
    """names the template instance behind each launch class and, where the counter passes saw that instance, its HBM bytes per launch"""
    for cls, e in by_kernel.items():
        name = KERNEL_OF_CLASS.get(cls)
        if name is None:
            continue
        e["kernel_instance"] = name
        t = (per_kernel_traffic or {}).get(name)
        if t and t[1] + t[2] >= 1e6 and e.get("algorithmic_MB_per_launch"):
            e["traffic_MB_per_launch"] = (t[1] + t[2]) / 1e6
            e["traffic_over_algorithmic"] = (t[1] + t[2]) / (e["algorithmic_MB_per_launch"] * 1e6)


def family_of(cls):
    """launch class -> kernel family: the 1x1-convolution products (forward + input gradient), their weight gradients (with the slab
    reductions they need), everything else that streams tensors once (depthwise, encoder / decoder, head backward)"""
    if cls.startswith("gemm "):
        return "gemm"
    if cls.startswith("wgrad ") or cls in ("reduce_slabs", "gln sums from wgrad"):
        return "wgrad"
    return "streaming"


def roofline_family(by_kernel):
    """{family: {frac, share, ms_per_step, launches_per_step}}: time-weighted fraction of the HBM roof (sum of the launches' algorithmic bytes /
    sum of their durations / 8 TB/s; launch classes without a byte count -- latency-bound tails -- add time only) and share of the step's
    kernel time, per family.  The per-class table (`roofline_by_kernel`) decides nothing by a 0.02 % tie this way (round-4 verdict item 3)."""
    fam = {}
    for cls, e in by_kernel.items():
        f = fam.setdefault(family_of(cls), {"ms": 0.0, "bytes": 0.0, "share": 0.0, "n": 0.0})
        f["ms"] += e["ms_per_step"]
        f["share"] += e["share_of_kernel_time"]
        f["n"] += e["launches_per_step"]
        if "algorithmic_MB_per_launch" in e:
            f["bytes"] += e["algorithmic_MB_per_launch"] * 1e6 * e["launches_per_step"]
    return {k: {"frac": v["bytes"] / (v["ms"] * 1e-3) / (HBM_PEAK_TBS * 1e12) if v["ms"] > 0 else None, "share": v["share"],
                "ms_per_step": v["ms"], "launches_per_step": v["n"]} for k, v in fam.items()}


def dominant_kernel_roofline(by_kernel, arith_name):
    """`roofline` of the bench line: ONE kernel template instance -- the largest launch class of the FAMILY with the largest share of the step's
    kernel time (the families' own fractions travel beside it as `roofline_family`)"""
    fams = roofline_family(by_kernel)
    top = max(fams, key=lambda k: fams[k]["share"]) if fams else None
    cands = [(e["ms_per_step"], cls) for cls, e in by_kernel.items() if "hbm_frac" in e and family_of(cls) == top]
    if not cands:
        cands = [(e["ms_per_step"], cls) for cls, e in by_kernel.items() if "hbm_frac" in e]
    if not cands:
        return None
    cls = max(cands)[1]
    e = by_kernel[cls]
    hbm = e["bound"] == "hbm"
    out = {"kernel": e.get("kernel_instance", cls), "launch_class": cls, "family": family_of(cls), "bound": e["bound"],
           "achieved": e["GBps"] if hbm else e["tflops_equiv"], "peak": HBM_PEAK_TBS * 1e3 if hbm else MFMA_PER_PRODUCT[arith_name][1] / MFMA_PER_PRODUCT[arith_name][0],
           "unit": "GB/s" if hbm else "TFLOP/s", "frac": e["hbm_frac"] if hbm else e["matrix_pipe_frac"],
           "traffic": e["traffic_MB_per_launch"] * 1e6 if "traffic_MB_per_launch" in e else None, "traffic_unit": "bytes/launch",
           "traffic_over_algorithmic": e.get("traffic_over_algorithmic"),
           "avg_launch_us": e["avg_us"], "launches_per_step": e["launches_per_step"], "share_of_kernel_time": e["share_of_kernel_time"],
           "algorithmic_bytes_per_launch": e["algorithmic_MB_per_launch"] * 1e6,
           # the same launch on SURVEY.md 8d's byte count (no `da` store-back, no re-read the fused minimum does not contain): the strictest of the figures
           "frac_8d": e.get("hbm_frac_8d"), "bytes_8d_per_launch": e["survey_8d_MB_per_launch"] * 1e6 if "survey_8d_MB_per_launch" in e else None,
           "measured": "HIP events around every launch of this class on the launch stream (second pass of the same steps, one stream); achieved = "
                       "algorithmic bytes of the launch (every operand tensor once, fp32, valid frames) / average duration; traffic = rocprofv3 --pmc "
                       "FETCH_SIZE x2 + WRITE_SIZE of the same instance"}
    if "matrix_pipe_frac" in e:
        out["matrix_pipe_frac"] = e["matrix_pipe_frac"]
    if top is not None and family_of(cls) == top:
        out["family_frac"], out["family_share"] = fams[top]["frac"], fams[top]["share"]      # the family's time-weighted figure beside its largest instance
    return out


def furthest_from_roof(by_kernel, n=6, min_share=0.03):
    """[class, avg us, fraction of the bounding roof] of the launch classes above `min_share` of the kernel time, lowest fraction first"""
    rows = []
    for cls, e in by_kernel.items():
        if e["share_of_kernel_time"] < min_share or "hbm_frac" not in e:
            continue
        frac = max(e["hbm_frac"], e.get("matrix_pipe_frac", 0.0))
        rows.append([cls, round(e["avg_us"], 1), round(frac, 3)])
    return sorted(rows, key=lambda r: r[2])[:n]


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
    Returns ({kernel name: (launches, read bytes, written bytes per launch)}, None) or (None, why) -- a failed pass is REPORTED in the
    bench detail (`hbm_traffic.error`), never silently dropped."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    res = {}
    env = dict(os.environ, SEPK_SIDE_STREAM="0", SEPK_SEQUENCE="0", TMPDIR="/tmp")      # (eager launches: one dispatch record per launch for the counters)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="sepk_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", ctr, "-d", d, "--", sys.executable, BENCH_PY, "--steps", "2", "--warmup", "1", "--batch", str(batch),
                   "--no-cpu-baseline", "--no-f32-pass", "--no-kernel-timing", "--no-pmc", "--no-stock"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout_s)
            if r.returncode != 0:
                return None, "{} pass: rocprofv3 rc {}: {}".format(ctr, r.returncode, r.stderr.decode(errors="replace")[-300:])
            dbs = glob.glob(d + "/**/*.db", recursive=True)
            if not dbs:
                return None, "{} pass: rocprofv3 wrote no database".format(ctr)
            con = sqlite3.connect(dbs[0])
            for name, n, avg in con.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? group by kernel_name", (ctr,)):
                short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                e = res.setdefault(short, [0, 0.0, 0.0])
                e[0] = max(e[0], n)
                e[1 if ctr == "FETCH_SIZE" else 2] = avg * 1024.0 * (2.0 if ctr == "FETCH_SIZE" else 1.0)
            con.close()
        except subprocess.TimeoutExpired:
            return None, "{} pass: timed out after {} s".format(ctr, timeout_s)
        except (subprocess.SubprocessError, OSError, sqlite3.Error) as e:
            return None, "{} pass: {}: {}".format(ctr, type(e).__name__, e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return res, None


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
           "sample_short": "median of {} fwd+PIT+bwd steps, B=2 paper-best, fp32 torch CPU, {:.2f} s/step".format(timed_steps, dt),
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
    """--config name -> (class, constructor arguments, recipe batch size, Adam arguments, GFLOP per utterance forward + backward, what runs
    where).  The GFLOP figures are counted on the unmodified reference modules by oracle/count_flops.py (every matrix product / convolution /
    attention product of one forward pass of a 4-s utterance, LSTM gate products included, x 3; DPRNN-TasNet's 980.07 is SURVEY.md 8d's)."""
    from models.dprnn_tasnet import DPRNNTasNet
    from models.dptnet import DPTNet
    from models.galrnet import GALRNet
    from models.sepformer import SepFormer
    from models.conv_tasnet import ConvTasNet
    tr = dict(enc_basis="trainable", dec_basis="trainable")
    return {
        # the reference constructor's default family (causal=True: cLN, all padding on the left) at the paper-best sizes: runs layer by layer
        # on this library's kernels (models/conv_tasnet.py::_run_staged), batch 16 like the headline
        "causal": (ConvTasNet, dict(PAPER, causal=True), 16, dict(lr=1e-3), 117.84,
                   "Conv-TasNet paper-best sizes, CAUSAL (cLN, left padding: reference tdcn.py:98,125-127), staged kernel path"),
        # BASELINE.json configs[3]: egs/wsj0-mix/dprnn-tasnet/train.sh:28-37
        "dprnn": (DPRNNTasNet, dict(n_basis=64, kernel_size=2, stride=1, enc_nonlinear=None, sep_hidden_channels=128, sep_bottleneck_channels=64,
                                    sep_chunk_size=250, sep_hop_size=125, sep_num_blocks=6, sep_norm=True, mask_nonlinear="sigmoid", causal=False,
                                    rnn_type="lstm", n_sources=2, **tr), 2, dict(lr=1e-3), 980.07,
                  "DPRNN-TasNet N=64 L=2 F=64 H=128 K=250 P=125 B=6 (BASELINE configs[3])"),
        # SURVEY.md section 8 row f4, the reference recipes' own sizes: egs/wsj0-mix/{dptnet,galrnet,sepformer}/train.sh
        "dptnet": (DPTNet, dict(n_basis=64, kernel_size=2, stride=1, enc_nonlinear=None, sep_bottleneck_channels=64, sep_hidden_channels=128,
                                sep_chunk_size=250, sep_hop_size=125, sep_num_blocks=6, sep_num_heads=4, sep_norm=True, sep_nonlinear="relu",
                                sep_dropout=0, mask_nonlinear="relu", causal=False, n_sources=2, **tr), 1, dict(lr=1e-3), 1206.76,
                   "DPTNet N=64 L=2 F=64 d_ff=128 K=250 P=125 B=6 h=4 (egs/wsj0-mix/dptnet/train.sh:28-44)"),
        "galrnet": (GALRNet, dict(n_basis=64, kernel_size=16, stride=8, enc_nonlinear=None, sep_hidden_channels=128, sep_chunk_size=100,
                                  sep_hop_size=50, sep_down_chunk_size=32, sep_num_blocks=6, sep_num_heads=8, sep_norm=True, sep_dropout=1e-1,
                                  mask_nonlinear="relu", causal=False, n_sources=2, low_dimension=True, **tr), 4, dict(lr=1e-3, weight_decay=1e-6), 64.81,
                    "GALRNet D=64 M=16 H=128 K=100 P=50 Q=32 N=6 J=8 (egs/wsj0-mix/galrnet/train.sh:28-42)"),
        "sepformer": (SepFormer, dict(n_basis=256, kernel_size=16, stride=8, enc_nonlinear="relu", sep_bottleneck_channels=256, sep_chunk_size=250,
                                      sep_hop_size=125, sep_num_blocks=2, sep_num_layers_intra=8, sep_num_layers_inter=8, sep_num_heads_intra=8,
                                      sep_num_heads_inter=8, sep_d_ff_intra=1024, sep_d_ff_inter=1024, sep_norm=True, sep_nonlinear="relu",
                                      sep_dropout=1e-1, mask_nonlinear="relu", causal=False, n_sources=2, **tr), 4, dict(lr=15e-5), 1184.66,
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
    if args.config == "causal":
        assert model.staged, model.staged_reason
    crit = PIT1d(NegSISDR(), n_sources=2)
    opt = torch.optim.Adam(model.parameters(), **adam)
    B = recipe_batch if args.batch is None else args.batch
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
    if args.config == "causal" and os.environ.get("SEPK_TORCH_ADAM", "0") != "1":
        # Conv-TasNet keeps its parameters in one flat buffer: clip + Adam as the headline's step takes them (sepkernels.train.FusedTrainStep:
        # gradients gathered into the flat buffer, one norm kernel, one fused Adam) instead of torch's foreach passes
        from sepkernels.train import FusedTrainStep
        fused = FusedTrainStep(model, crit, lr=adam.get("lr", 1e-3), max_norm=5.0)

        def step():      # noqa: F811
            return fused(mix, src)
        launch = "eager (one launch per kernel), fused clip + Adam"
        # (replaying this step as a hipGraph was tried in round 5: the replay of the staged sequence faults -- Memory access fault by GPU node,
        #  profiles/r07_round5_experiments.md -- so the causal line keeps eager launches)
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
    config = {"workload": "{}, 2 spk, 4 s @ 8 kHz synthetic mixtures, batch {}{}, fwd + PIT(NegSI-SDR) + bwd + clip(5) + Adam".format(
                  label, B, " (recipe default)" if B == recipe_batch else " (recipe default: {})".format(recipe_batch)),
              "global_batch": B, "frames_per_utterance": F, "parallelism": "dp1", "utt_per_s": B * args.steps / el, "final_loss": float(loss),
              "parameters": model.num_parameters, "launch": launch}
    config["algorithmic_gflop_per_utterance_fwd_bwd"] = gflop
    tf = gflop * 1e9 * B * args.steps / el / 1e12
    if args.config == "causal":
        # same tensors and products as the headline (cLN instead of gLN): whole step against the HBM roof with SURVEY.md 8d's bytes per frame, the
        # matrix-pipe fraction of its f16x3 products beside it
        by_frame = 3 * bytes_per_frame(PAPER)
        per, pipe = MFMA_PER_PRODUCT["f16x3"]
        gbs = B * F * args.steps / el * by_frame / 1e9
        roofline = {"kernel": "whole step (staged sequence: sep_pw_gemm, sep_cln_*, sep_depthwise_*; per-kernel: profiles/r05zm_causal_kernel_stats.md)",
                    "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_TBS * 1e3, "unit": "GB/s", "frac": gbs / (HBM_PEAK_TBS * 1e3), "traffic": None,
                    "frac_is": "of_fused_minimum (the staged sequence itself moves more bytes than SURVEY.md 8d's count: not comparable with a per-kernel frac)",
                    "matrix_pipe_frac": tf / (pipe / per),
                    "what": "algorithmic bytes of fwd + bwd ({} B per frame, SURVEY.md 8d: the fused non-causal sequence's minimum) x frames/s vs 8 TB/s".format(by_frame)}
        note = "{:.2f} GFLOP per utterance, {} B per frame (SURVEY.md 8d)".format(gflop, by_frame)
    else:
        # these steps' arithmetic is fp32 throughout (LSTM recurrences on v_mfma_f32_16x16x4 / 4x4x1, dense layers of csrc/linear.hip and the
        # attention core on the fp32 MFMA): matrix-pipe roof
        roofline = {"kernel": "whole step (LSTM sweeps, dense layers of csrc/linear.hip, attention core of csrc/attn.hip; per-kernel: profiles/*_{}_kernel_stats.md)".format(args.config),
                    "bound": "mfma", "achieved": tf, "peak": FP32_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": tf / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
                    "what": "algorithmic fp32 FLOP of fwd + bwd ({:.2f} GFLOP per utterance: oracle/count_flops.py on the reference modules) / step time vs the dense fp32 MFMA peak".format(gflop)}
        note = "{:.2f} GFLOP per utterance (oracle/count_flops.py): {:.1f} TFLOP/s achieved".format(gflop, tf)
    print(json.dumps({
        "metric": "separated audio frames/sec (fwd+bwd), {} 2-spk 4s@8kHz".format("DPRNN-TasNet" if args.config == "dprnn" else "causal Conv-TasNet" if args.config == "causal" else cls.__name__) +
                  (" (BASELINE configs[3])" if args.config == "dprnn" else ""), "value": B * F * args.steps / el, "unit": "frames/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": config, "roofline": roofline, "roofline_note": note,
        "peak_memory_GB": torch.cuda.max_memory_allocated() / 2 ** 30}), flush=True)
