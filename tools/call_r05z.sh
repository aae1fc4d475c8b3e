# r05z: where does the register-fed weight gradient spend its time?  Probe builds (garbage results): no split arithmetic, no MFMAs, no global loads
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
for v in "" _rf_NOSPLIT _rf_NOMFMA _rf_NOLOAD; do echo "== libsepkernels$v"; SEPKERNELS_LIB=$PWD/dnn-based_source_separation_amd/libsepkernels$v.so timeout 300 python tools/gemm_bench.py --only W2,W3 --reps 20 2>&1 | grep "^W" | cut -c1-110; done
