"""Dump the per-kernel statistics of a rocprofv3 (rocpd sqlite) capture as CSV + markdown for profiles/.
    python tools/export_profile.py gpurun_out/prof_xxx/bench_results.db profiles/r01_xxx [steps]"""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
con = sqlite3.connect(db)
cur = con.cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(out + "_kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "CallsPerStep", "MsPerStep"])
    for name, calls, tot, avg, pct in rows:
        w.writerow([name, calls, int(tot * 1000), int(avg * 1000), "%.4f" % pct, "%.2f" % (calls / steps), "%.4f" % (tot / 1e3 / steps)])
tot_ms = sum(r[2] for r in rows) / 1e3 / steps
with open(out + "_kernel_stats.md", "w") as f:
    f.write("| kernel | calls/step | avg us | ms/step | % |\n|---|---:|---:|---:|---:|\n")
    for name, calls, tot, avg, pct in rows[:25]:
        short = name.replace("(anonymous namespace)::", "").split("(")[0][:70]
        f.write("| `{}` | {:.1f} | {:.1f} | {:.3f} | {:.1f} |\n".format(short, calls / steps, avg, tot / 1e3 / steps, pct))
    f.write("\nGPU kernel time per step: {:.2f} ms ({} steps in the capture)\n".format(tot_ms, int(steps)))
    for label, pref in (("sep_pw_gemm group (pw_gemm_pc_kernel + pw_gemm_coop_kernel + pw_gemm_direct_kernel + pw_gemm_kernel)", "pw_gemm"), ("sep_pw_wgrad group (pw_wgrad_*)", "pw_wgrad")):
        sel = [r for r in rows if pref in r[0].replace("(anonymous namespace)::", "")]
        calls = sum(r[1] for r in sel)
        if calls:
            f.write("{}: {:.1f} launches/step, average duration {:.1f} us, {:.3f} ms/step  (what bench.py's roofline.avg_launch_ms must agree with)\n".format(
                label, calls / steps, sum(r[2] for r in sel) / calls, sum(r[2] for r in sel) / 1e3 / steps))
try:
    q = ("select kernel_name, grid_size_x, counter_name, count(*), avg(value) from counters_collection "
         "group by kernel_name, grid_size_x, counter_name order by kernel_name, grid_size_x, counter_name")
    prow = list(cur.execute(q))
    if prow:
        with open(out + "_pmc.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Kernel", "GridSizeX", "Counter", "Dispatches", "AvgValue"])
            for r in prow:
                w.writerow([r[0], r[1], r[2], r[3], "%.1f" % r[4]])
except sqlite3.Error:
    pass
# idle time between kernels (launch gaps): union of the kernel intervals of the steady-state second half of the capture
try:
    cols = [c[1] for c in cur.execute("pragma table_info(kernels)")]
    sc = "start" if "start" in cols else [c for c in cols if "start" in c][0]
    ec = "end" if "end" in cols else [c for c in cols if "end" in c][0]
    iv = sorted(cur.execute("select {}, {} from kernels".format(sc, ec)))
    iv = iv[len(iv) // 2:]
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for a, b in iv[1:]:
        if a > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = a, b
        else:
            cur_e = max(cur_e, b)
    busy += cur_e - cur_s
    span = iv[-1][1] - iv[0][0]
    with open(out + "_kernel_stats.md", "a") as f:
        f.write("GPU busy (union of kernel intervals) over the second half of the capture: {:.1f} % of {:.1f} ms "
                "wall -> {:.1f} % idle between kernels\n".format(100.0 * busy / span, span / 1e6, 100.0 * (1 - busy / span)))
    print("busy fraction %.3f" % (busy / span))
except Exception as e:  # noqa: BLE001
    print("gap analysis skipped:", e)
print("wrote", out + "_kernel_stats.{csv,md}", "total ms/step %.2f" % tot_ms)
