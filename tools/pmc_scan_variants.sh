# rocprofv3 counter passes over the scan variants; raw CSVs are summarised on the box and removed (they exceed the pull limit)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmcv
rm -rf $O && mkdir -p $O
N=${1:-1000000}
pass() {
  name=$1; shift
  timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -- python tools/pmc_scan_variants.py $N > $O/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $O/$name.log; return 1; }
  python tools/pmc_summarise.py $O/$name >> $O/summary.txt
  rm -rf $O/$name
}
pass p0 GRBM_GUI_ACTIVE GRBM_COUNT &&
pass p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU &&
pass p2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY &&
pass p3 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
cat $O/summary.txt
