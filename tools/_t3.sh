cd $GRAFT_REPO_ROOT
echo "=== PC prio+saddr"; SEPK_GEMM_KERNEL=pc python tools/gemm_bench.py --reps 20 --packed --only F1,F3,G4,G2,P0 2>/dev/null
echo "=== PC noprio+saddr"; SEPKERNELS_LIB=$PWD/tools/_noprio.so SEPK_GEMM_KERNEL=pc python tools/gemm_bench.py --reps 20 --packed --only F1,F3,G4,G2,P0 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm or pack" 2>&1 | tail -2
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-pass 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step %.2f  gemm avg us %.1f  share %.3f  wgrad avg us %.1f loss %.5f'%(d['ms_per_step'], 1e3*r['avg_launch_ms'], r['share_of_step'], 1e3*d['roofline_wgrad']['avg_launch_ms'], d['config']['final_loss']))"
