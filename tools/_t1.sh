cd $GRAFT_REPO_ROOT
for ns in 2 3; do
echo "=== NS=$ns"
SEPK_COOP_NS=$ns timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "packed_weights_model or pack_weights_repro" 2>&1 | tail -8
done
echo "=== model tests + smoke (NS=2)"
timeout 900 python -m pytest tests/test_gpu_model.py -q -x 2>&1 | tail -8
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2
for cfg in "2 0" "3 0" "2 1" "3 1"; do set -- $cfg
echo "=== bench NS=$1 MI-forced=$2"
SEPK_COOP_NS=$1 SEPK_COOP_MI=$2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-pass 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step %.2f  gemm avg us %.1f  share %.3f  wgrad avg us %.1f'%(d['ms_per_step'], 1e3*r['avg_launch_ms'], r['share_of_step'], 1e3*d['roofline_wgrad']['avg_launch_ms']))"
done
echo "=== old path"; SEPK_COOP=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-pass 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step %.2f  gemm avg us %.1f'%(d['ms_per_step'], 1e3*r['avg_launch_ms']))"
