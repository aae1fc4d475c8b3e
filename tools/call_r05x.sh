# r05x: SQ counters of the three heads-family kernels (W3 heads wgrad, G3 headsT, F3 heads) on their step shapes: where do the wave cycles go?
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" \
            "SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pass -d /tmp/pmcx_$i -- python $R/tools/gemm_bench.py --only W3,G3,F3 --reps 3 > /tmp/pmcx_$i.log 2>&1
  echo "=== pass $i rc=$?"; tail -3 /tmp/pmcx_$i.log | cut -c1-160
  python $R/tools/pmc_summary.py /tmp/pmcx_$i "%pw_%" > $R/gpurun_out/r05x_pmc_$i.txt 2>&1
  cat $R/gpurun_out/r05x_pmc_$i.txt | cut -c1-170
done
