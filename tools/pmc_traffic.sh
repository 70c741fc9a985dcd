# HBM traffic of the libcoral_hip kernels: two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only), summarised on the box.
#   bash tools/pmc_traffic.sh [config [n_reads]]      (defaults: cfg3 2000000)  ->  gpurun_out/pmct/pmc_traffic_<config>.{json,md}
CFG=${1:-cfg3}
N=${2:-2000000}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmct
rm -rf $O/fetch $O/write && mkdir -p $O
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 tools/pmc_target.py $N $CFG > $O/fetch_$CFG.log 2>&1 || { tail -5 $O/fetch_$CFG.log; exit 1; }
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 tools/pmc_target.py $N $CFG > $O/write_$CFG.log 2>&1 || { tail -5 $O/write_$CFG.log; exit 1; }
REC=$(grep -o "records [0-9]*" $O/fetch_$CFG.log | head -1 | cut -d' ' -f2)
ALG=$(grep -o "alg bytes [0-9]*" $O/fetch_$CFG.log | head -1 | cut -d' ' -f3)
LIB=$(grep "^library " $O/fetch_$CFG.log | head -1 | cut -d' ' -f2-)
python tools/pmc_traffic.py $O/fetch $O/write $N $REC $ALG $O/pmc_traffic_$CFG.json $O/pmc_traffic_$CFG.md $CFG "$LIB"
rm -rf $O/fetch $O/write
