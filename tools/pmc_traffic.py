"""HBM traffic per launch of every kernel of the step, from the two PMC passes of tools/pmc_passes.sh
(gpurun_out/pmc_FETCH_SIZE.txt, gpurun_out/pmc_WRITE_SIZE.txt; rocprofv3 reports both in KiB).
    python tools/pmc_traffic.py gpurun_out profiles/r01e
writes <prefix>_hbm_traffic.md (per kernel) and profiles/hbm_traffic.json (what bench.py reports as roofline.traffic).
FETCH_SIZE is doubled: on gfx950 it counts the 128-byte requests of wide coalesced reads at 64 bytes
(MI355X_MICROARCH.md, HBM section); checked here on dwconv_fwd/bwd, whose reads are exactly 1 and 3 H-tensors."""
import json
import os
import re
import sys

src, prefix = sys.argv[1], sys.argv[2]


def parse(fn):
    out, key = {}, None
    for line in open(fn):
        if line.startswith("=="):
            m = re.match(r"== (.*) grid=(\d+) dispatches=(\d+)", line.strip())
            key = (m.group(1), int(m.group(2)), int(m.group(3)))
        else:
            c, v = line.split()
            out.setdefault(key, {})[c] = float(v)
    return out


F = parse(os.path.join(src, "pmc_FETCH_SIZE.txt"))
W = parse(os.path.join(src, "pmc_WRITE_SIZE.txt"))
rows, variants = [], {}
groups = {"gemm": {"launches": 0, "bytes": 0.0, "uncounted": 0}, "wgrad": {"launches": 0, "bytes": 0.0, "uncounted": 0}}
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0      # steps in the capture (tools/pmc_passes.sh: 2 + 1 warm-up)
for k in sorted(F):
    name, grid, n = k
    fetch = 2.0 * F[k].get("FETCH_SIZE", 0.0) * 1024.0
    write = W.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0
    rows.append((name, grid, n, fetch, write))
    grp = "gemm" if re.match(r"pw_gemm_(pc|coop|direct)_kernel|pw_gemm_kernel", name) else "wgrad" if re.match(r"pw_wgrad", name) else None
    if grp and fetch + write < 1e6:
        groups[grp]["uncounted"] += n        # rocprofv3 returns zeros for every counter of one kernel per capture (it runs normally in the trace): left out
    elif grp:
        groups[grp]["launches"] += n
        groups[grp]["bytes"] += n * (fetch + write)
    m = re.match(r"pw_gemm_direct_kernel<(true|false), (\d), (true|false)(?:, -?\d+){0,2}>", name)
    if m:
        key = "T{}P{}S{}".format(int(m.group(1) == "true"), m.group(2), int(m.group(3) == "true"))
        v = variants.setdefault(key, {"launches": 0, "bytes": 0.0})
        v["launches"] += n
        v["bytes"] += n * (fetch + write)
with open(prefix + "_hbm_traffic.md", "w") as f:
    f.write("| kernel | grid (threads) | launches | HBM read MB/launch (FETCH_SIZE x2) | HBM write MB/launch | total MB/launch |\n|---|---:|---:|---:|---:|---:|\n")
    for name, grid, n, fetch, write in rows:
        f.write("| `{}` | {} | {} | {:.1f} | {:.1f} | {:.1f} |\n".format(name, grid, n, fetch / 1e6, write / 1e6, (fetch + write) / 1e6))
    tot = sum(n * (fe + wr) for _, _, n, fe, wr in rows)
    f.write("\nAll kernels of the capture: {:.2f} GB.\n".format(tot / 1e9))
out = {"source": os.path.basename(prefix) + "_hbm_traffic.md (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, FETCH_SIZE x2 on gfx950)",
       "groups": {g: {"launches_per_step": v["launches"] / steps, "bytes_per_launch": v["bytes"] / max(v["launches"], 1),
                      "launches_per_step_without_counters": v["uncounted"] / steps} for g, v in groups.items()},
       "step_total_bytes": sum(n * (fe + wr) for _, _, n, fe, wr in rows) / steps,
       "gemm_variants": {k: {"launches": v["launches"], "bytes_per_launch": v["bytes"] / max(v["launches"], 1)} for k, v in variants.items()}}
with open(os.path.join(os.path.dirname(prefix), "hbm_traffic.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print(json.dumps(out["groups"], indent=1), "step total GB:", out["step_total_bytes"] / 1e9)
