cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
timeout 300 python tools/attn_bench.py --sdpa dptnet-inter dptnet-257 dptnet-320 sepformer-intra | tee gpurun_out/r07x_attention_257.txt
