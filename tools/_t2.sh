cd $GRAFT_REPO_ROOT
echo "=== old (unpacked f16x3)"; python tools/gemm_bench.py --reps 20 --only F,G,P 2>/dev/null
for ns in 2 3; do for mi in 0 1; do
echo "=== packed NS=$ns MIforce=$mi"; SEPK_COOP_NS=$ns SEPK_COOP_MI=$mi python tools/gemm_bench.py --reps 20 --packed --only F,G,P 2>/dev/null
done; done
for a in 1 2 3 4 5; do
echo "=== ABL $a (NS=3)"; SEPK_COOP_NS=3 SEPKERNELS_LIB=$PWD/tools/_abl$a.so python tools/gemm_bench.py --reps 20 --packed --only F2,F3,G3,G2,P0 2>/dev/null
done
