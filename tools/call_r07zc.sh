cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
timeout 300 python tools/host_profile.py galrnet 5 2>&1 | cut -c1-190 > gpurun_out/r07zc_host_galrnet.txt
sed -n '/Ordered by: cumulative/,$p' gpurun_out/r07zc_host_galrnet.txt | head -70
head -3 gpurun_out/r07zc_host_galrnet.txt
