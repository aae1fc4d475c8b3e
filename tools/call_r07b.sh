# r07b: kernel traces of the headline step at 16 and 32 utterances, one stream (kernel-alone durations): which launches do not scale with
# the batch (the step is 2.6 ms + 0.87 ms per utterance by r07a's two points)?  + the idle share of the step's timeline.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R/dnn-based_source_separation_amd/src
export SEPK_SIDE_STREAM=0
Q="--no-cpu-baseline --no-f32-pass --no-kernel-timing --no-pmc --no-stock --steps 6 --warmup 2"
for b in 16 32; do
  timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_b$b -o bench -- python $R/bench.py $Q --batch $b > /tmp/prof_b$b.log 2>&1
  echo "rc=$?"; grep '^{' /tmp/prof_b$b.log | tail -1 | cut -c1-200
  db=$(find /tmp/prof_b$b -name '*.db' | head -1)
  python $R/tools/rocpd_summary.py $db $R/gpurun_out/r07b_b${b}_kernels.md 8
  python - $db <<'P'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
iv = sorted(c.execute("select start, end from kernels"))
iv = iv[len(iv) // 2:]
busy, s, e = 0, iv[0][0], iv[0][1]
for a, b in iv[1:]:
    if a > e:
        busy += e - s; s, e = a, b
    else:
        e = max(e, b)
busy += e - s
span = iv[-1][1] - iv[0][0]
gaps = sorted((iv[i + 1][0] - iv[i][1]) for i in range(len(iv) - 1))
print("second half of the capture: span %.2f ms, busy %.2f ms (%.1f %%), %d launches, median gap %.2f us, p90 %.2f us, sum of positive gaps %.2f ms" % (
    span / 1e6, busy / 1e6, 100.0 * busy / span, len(iv), gaps[len(gaps) // 2] / 1e3, gaps[int(0.9 * len(gaps))] / 1e3, sum(g for g in gaps if g > 0) / 1e6))
P
done
