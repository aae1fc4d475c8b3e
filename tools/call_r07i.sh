# r07i: SQ counters of the cooperative kernel on conv1 (F2) and heads^T (G3p) standalone: where do the wave cycles of the write-heavy short-contraction shapes go?
# + the GPU tier's summary line (r07h lost it under RCCL's banner)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" \
            "SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pass -d /tmp/pmci_$i -- python $R/tools/gemm_bench.py --packed --only F2,G3p --reps 3 > /tmp/pmci_$i.log 2>&1
  echo "=== pass $i rc=$?"; tail -2 /tmp/pmci_$i.log | cut -c1-160
  python $R/tools/pmc_summary.py /tmp/pmci_$i "%pw_gemm%" > $R/gpurun_out/r07i_pmc_$i.txt 2>&1
  cat $R/gpurun_out/r07i_pmc_$i.txt | cut -c1-200
done
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/r07i_gputests.txt; cat gpurun_out/r07i_gputests.txt
