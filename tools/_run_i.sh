mkdir -p gpurun_out/r03i; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
run() { tag=$1; shift; rm -rf /tmp/pmc_$tag; timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_$tag -o p -- python $R/tools/pmc_one_conv.py "$@" > /dev/null 2>&1; python $R/tools/pmc_summary.py $(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1) > $R/gpurun_out/r03i/pmc_$tag.txt 2>&1
  rm -rf /tmp/pmc2_$tag; timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL --output-format csv -d /tmp/pmc2_$tag -o p -- python $R/tools/pmc_one_conv.py "$@" > /dev/null 2>&1; python $R/tools/pmc_summary.py $(find /tmp/pmc2_$tag -name "*counter_collection.csv" | head -1) >> $R/gpurun_out/r03i/pmc_$tag.txt 2>&1; }
run conv640 conv 8 32 640 640
run conv1280 conv 8 16 1280 1280
run gemm_l1ff gemm 8192 640 2560
run geglu_l0 gemm 32768 2560 320 1
cat $R/gpurun_out/r03i/pmc_conv640.txt $R/gpurun_out/r03i/pmc_gemm_l1ff.txt $R/gpurun_out/r03i/pmc_geglu_l0.txt
