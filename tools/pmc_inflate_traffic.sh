# HBM traffic of k_bgzf_inflate (tools/bench_inflate.py: 17,565 blocks, 0.28 GB in, 1.14 GB out): FETCH_SIZE / WRITE_SIZE in separate passes
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmcit
rm -rf $O && mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -- python3 tools/bench_inflate.py 30000 1 2 > $O/$c.log 2>&1 || { echo "pass $c failed"; tail -5 $O/$c.log; exit 1; }
  python tools/pmc_summarise.py $O/$c >> $O/summary.txt
  rm -rf $O/$c
done
grep -A1 "^kernel" $O/summary.txt | grep -v "^--" | grep "kernel\|inflate"
