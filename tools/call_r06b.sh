# r06b: ceiling of moving the operand split out of pw_gemm_pc_kernel's producers: probe build (no scaling, no split: garbage results) against the tree
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
for v in "" _nosplit ""; do echo "== libsepkernels$v"; SEPKERNELS_LIB=$PWD/dnn-based_source_separation_amd/libsepkernels$v.so timeout 120 python tools/gemm_bench.py --packed --only F1,F3,F4,G2,G4 --reps 20 2>&1 | grep "^[FG]" | cut -c1-100; done
