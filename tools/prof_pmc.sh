set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5pmc}
rm -rf $O; mkdir -p $O
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-secondary --streams 1"
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -o p -- $CMD > $O/p$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
python3 tools/pmc_table.py $O/pmc_counters_per_launch.txt $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 $O/p6 > /dev/null
rm -rf $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 $O/p6
cat $O/pmc_counters_per_launch.txt
