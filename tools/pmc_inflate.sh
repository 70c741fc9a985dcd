# rocprofv3 counter passes over k_bgzf_inflate alone (tools/bench_inflate.py); raw CSVs are summarised on the box and removed
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmci
rm -rf $O && mkdir -p $O
N=${1:-30000}
pass() {
  name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -- python3 tools/bench_inflate.py $N 1 2 > $O/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $O/$name.log; return 1; }
  python tools/pmc_summarise.py $O/$name >> $O/summary.txt
  rm -rf $O/$name
}
pass p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU &&
pass p2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_WAIT_ANY GRBM_GUI_ACTIVE &&
pass p3 SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES_LT_64
cat $O/summary.txt
