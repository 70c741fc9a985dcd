# rocprofv3 counter passes over coral_cigar_scan at config 3 (tools/pmc_target.py); raw CSVs are summarised on the box and removed
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmcs
rm -rf $O && mkdir -p $O
N=${1:-2000000}
pass() {
  name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -- python3 tools/pmc_target.py $N > $O/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $O/$name.log; return 1; }
  python tools/pmc_summarise.py $O/$name >> $O/summary.txt
  rm -rf $O/$name
}
pass p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU &&
pass p2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_WAIT_ANY GRBM_GUI_ACTIVE
cat $O/summary.txt
