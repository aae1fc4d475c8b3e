# rocprofv3 kernel trace of the bench step; run on the GPU box through gpurun:  bash tools/profile_step.sh <tag> [steps]
tag=${1:-prof}; steps=${2:-8}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export SEPK_SIDE_STREAM=${SEPK_SIDE_STREAM:-0}   # kernel-alone durations (the default overlaps the weight gradients on a second stream)
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o bench -- python $R/bench.py --steps $steps --warmup 2 --no-cpu-baseline --no-f32-pass --no-kernel-timing --no-pmc --no-stock > /tmp/prof_$tag.log 2>&1
echo "rc=$?"; grep '^{' /tmp/prof_$tag.log | tail -1 | cut -c1-400
db=$(find /tmp/prof_$tag -name '*.db' | head -1)
python $R/tools/export_profile.py $db $R/gpurun_out/$tag $((steps + 2))
