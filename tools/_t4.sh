cd $GRAFT_REPO_ROOT
for cfg in "auto 512" "auto 256" "pc 0"; do set -- $cfg
echo "=== bench kernel=$1 mink=$2"
SEPK_GEMM_KERNEL=$1 SEPK_PC_MINK=$2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-pass 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step %.2f  gemm avg us %.1f  share %.3f  wgrad avg us %.1f loss %.5f'%(d['ms_per_step'], 1e3*r['avg_launch_ms'], r['share_of_step'], 1e3*d['roofline_wgrad']['avg_launch_ms'], d['config']['final_loss']))"
done
echo "=== old path"; SEPK_COOP=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-pass 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step %.2f  gemm avg us %.1f loss %.5f'%(d['ms_per_step'], 1e3*r['avg_launch_ms'], d['config']['final_loss']))"
timeout 900 python -m pytest tests/test_gpu_model.py -q -x 2>&1 | tail -3
