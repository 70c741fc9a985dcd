# HBM traffic of the libcoral_hip kernels on config 3 (2M reads): two rocprofv3 counter passes, summarised on the box.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmct
rm -rf $O && mkdir -p $O
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 tools/pmc_target.py 2000000 > $O/fetch.log 2>&1 || { tail -5 $O/fetch.log; exit 1; }
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 tools/pmc_target.py 2000000 > $O/write.log 2>&1 || { tail -5 $O/write.log; exit 1; }
REC=$(grep -o "records [0-9]*" $O/fetch.log | head -1 | cut -d' ' -f2)
ALG=$(grep -o "alg bytes [0-9]*" $O/fetch.log | head -1 | cut -d' ' -f3)
python tools/pmc_traffic.py $O/fetch $O/write 2000000 $REC $ALG $O/pmc_traffic.json $O/pmc_traffic.md
rm -rf $O/fetch $O/write
