# PMC passes (one counter set per run, no trace domains besides the kernel dispatch records) over any command: tools/prof_pmc_cmd.sh <out-name> python3 <script> [args...]
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
NAME=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$NAME
rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -o p -- "$@" > $O/p$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
python3 tools/pmc_table.py $O/pmc_counters_per_launch.txt $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 $O/p6 > /dev/null
rm -rf $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 $O/p6
cat $O/pmc_counters_per_launch.txt
