# r07ze: dense kernel tile sizes against wave quantisation (516 workgroups of 128 x 128 on 512 slots = two rounds)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
for cfg in "128 128" "128 64" "64 128" "64 64"; do set -- $cfg; echo "== tile $1 x $2"; SEPK_LIN_TI=$1 SEPK_LIN_TJ=$2 timeout 300 python tools/linear_bench.py; done | tee gpurun_out/r07ze_linear_tiles.txt
