# r05zb: does producer VALU overlap the consumers' MFMAs?  Register-fed kernel, no global loads: (A) split only, (B) MFMAs only, (C) neither
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
for v in _rf_A _rf_B _rf_C; do echo "== libsepkernels$v"; SEPKERNELS_LIB=$PWD/dnn-based_source_separation_amd/libsepkernels$v.so timeout 300 python tools/gemm_bench.py --only W2,W3 --reps 20 2>&1 | grep "^W" | cut -c1-110; done
