"""Per-kernel summary (calls, total, average, share) of a rocprofv3 run from its SQLite database (`-o name` -> name_results.db):
the same table `--stats` prints, for runs whose CSV post-processing did not finish inside the box's time limit.

    python tools/rocpd_summary.py gpurun_out/prof_galr/galr_results.db profiles/r02k_galrnet_kernel_stats.md [steps] [name-filter]
With a name filter: one line per (kernel, grid) of the kernels whose name contains it -- launches of one template at different shapes apart.
"""
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) from kernels "
                     "group by name order by 3 desc").fetchall()
    total, first, last, n = c.execute("select sum(end - start), min(start), max(end), count(*) from kernels").fetchone()
    with open(out, "w") as f:
        f.write("rocprofv3 --kernel-trace, {} dispatches over {} steps (warm-up included): kernel time {:.2f} ms = {:.2f} ms/step, "
                "first-to-last dispatch {:.1f} ms.\n\n".format(n, steps, total / 1e6, total / 1e6 / steps, (last - first) / 1e6))
        f.write("| kernel | calls | total ms | avg us | min us | max us | share |\n|---|---:|---:|---:|---:|---:|---:|\n")
        for name, calls, tot, avg, lo, hi in rows:
            short = name if len(name) <= 110 else name[:107] + "..."
            f.write("| `{}` | {} | {:.3f} | {:.1f} | {:.1f} | {:.1f} | {:.1f} % |\n".format(short.replace("|", "\\|"), calls, tot / 1e6, avg / 1e3,
                                                                                     lo / 1e3, hi / 1e3, 100.0 * tot / total))
        if len(sys.argv) > 4:
            cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
            gcols = [k for k in cols if "grid" in k.lower()]
            f.write("\nlaunches of kernels matching `{}` by grid ({}):\n\n| kernel | grid | calls | avg us | min us | max us |\n|---|---|---:|---:|---:|---:|\n".format(
                sys.argv[4], ", ".join(gcols)))
            q = "select name, {g}, count(*), avg(end - start), min(end - start), max(end - start) from kernels where name like ? group by name, {g} order by 1, 4 desc".format(
                g=", ".join(gcols))
            for r in c.execute(q, ("%" + sys.argv[4] + "%",)):
                name, grid, (calls, avg, lo, hi) = r[0], r[1:1 + len(gcols)], r[1 + len(gcols):]
                f.write("| `{}` | {} | {} | {:.1f} | {:.1f} | {:.1f} |\n".format(name[:60], " x ".join(map(str, grid)), calls, avg / 1e3, lo / 1e3, hi / 1e3))
    print("wrote", out, "kernel ms/step", total / 1e6 / steps)


if __name__ == "__main__":
    main()
