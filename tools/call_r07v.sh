cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
timeout 300 python tools/host_profile.py galrnet 5 2>&1 | cut -c1-200 | tee gpurun_out/r07v_host_galrnet.txt | head -60
timeout 300 python tools/host_profile.py dptnet 5 2>&1 | cut -c1-200 | tee gpurun_out/r07v_host_dptnet.txt | head -12
