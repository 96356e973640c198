cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_one
rm -rf $O; mkdir -p $O
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM"; do
  n=$(echo $set | cut -d" " -f1)
  rocprofv3 --pmc $set -d $O/$n -o run --output-format csv -- python3 $R/tools/op_bench.py --ops X13/conv,X05/conv --phases fwd,wgrad --reps 3 > $O/$n.log 2>&1
  python3 $R/tools/profile_summary.py pmc $O/$n $O/$n.csv
  rm -rf $O/$n
done
grep -h "halo\|fwd3\|kernel,disp" $O/*.csv | cut -c1-250
