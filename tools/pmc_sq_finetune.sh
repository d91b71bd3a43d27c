cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_sq
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o pmc --output-format csv -- python3 $R/tools/bench_finetune.py --batch 256 --steps 1 > $OUT/g$i.log 2>&1
done
cd $R && python3 tools/pmc_sq.py "gemm_tn_bf16|gconv_wgrad_band_kernel|bn_bwd"
