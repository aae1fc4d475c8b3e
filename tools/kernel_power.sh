# Socket power and shader clock while ONE kernel of the headline step runs in a loop (rocm-smi polled beside tools/gemm_bench.py / stream_bench.py):
#   bash tools/kernel_power.sh          (GPU box)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
poll() {  # $1 = pid: three samples while it lives, 5 s after its start (import + set-up)
  sleep 7
  for i in 1 2 3; do if kill -0 $1 2>/dev/null; then rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Current Socket" | sed 's/GPU\[0\]\t\t: //; s/clock level: 1: //; s/Current Socket Graphics Package Power (W)/W/' | tr '\n' ' '; echo; sleep 0.8; fi; done
}
for c in F2 F3 G3 G2 W2 W3; do
  ( timeout 60 python tools/gemm_bench.py --only $c --packed --reps 70000 2>/dev/null | grep -v "^$" ) &
  P=$!; poll $P; wait $P
done
( REPS=1 timeout 90 python - <<'P'
import os, sys, time, torch
sys.argv = ["x"]
sys.path.insert(0, "tools")
t0 = time.time()
import stream_bench as SB        # runs its table once (import), then loop the two kernels at dilation 8
print("loop")
sys.stdout.flush()
t0 = time.time()
K, d = SB.K, 8
while time.time() - t0 < 5:
    for _ in range(500): K.dwconv_fwd(SB.a, SB.st1, SB.g1, SB.b1, SB.a1, SB.wd, SB.bd, SB.a2, SB.z, SB.st2, SB.B, SB.C, SB.T, SB.ldt, d, 1e-12)
    torch.cuda.synchronize()
print("dw fwd loop done"); sys.stdout.flush()
t0 = time.time()
while time.time() - t0 < 5:
    for _ in range(500): K.dwconv_bwd(SB.dv2, SB.z, SB.a, SB.st1, SB.g1, SB.b1, SB.a1, SB.st2, SB.g2, SB.a2, SB.bsum2, SB.wd, SB.bd, SB.dv1, SB.rp, SB.bacc1, None, None, SB.B, SB.C, SB.T, SB.ldt, d, 1e-12)
    torch.cuda.synchronize()
print("dw bwd loop done")
P
) > /tmp/dw.log 2>&1 &
P=$!
sleep 9
for i in $(seq 1 14); do if kill -0 $P 2>/dev/null; then echo -n "dw t=$i "; rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Current Socket" | sed 's/GPU\[0\]\t\t: //; s/clock level: 1: //; s/Current Socket Graphics Package Power (W)/W/' | tr '\n' ' '; echo; sleep 0.8; fi; done
wait $P; grep -E "loop|done|per 8" /tmp/dw.log
