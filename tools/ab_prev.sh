#!/bin/bash
# Development tool: an A/B partner for one gpurun call.  Exports <commit> into _ab_prev/ (git-ignored, travels to the GPU box) and builds its
# library there, so that `python _ab_prev/bench.py ...` and `python bench.py ...` can be timed back to back on ONE box (boxes of the pool
# differ by +-4 %).   bash tools/ab_prev.sh <commit>
set -e
cd "$(dirname "$0")/.."
rm -rf _ab_prev && mkdir _ab_prev
git archive "$1" | tar -x -C _ab_prev --exclude='profiles' --exclude='tests/golden' --exclude='gpurun_out'
(cd _ab_prev && python __graft_entry__.py > /dev/null 2>&1 && ls -la dnn-based_source_separation_amd/libsepkernels.so && rm -rf dnn-based_source_separation_amd/csrc/_obj)
