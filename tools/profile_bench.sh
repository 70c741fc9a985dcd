# rocprofv3 --kernel-trace --stats of the default bench command (fewer steps, no CPU baseline), summarised into gpurun_out/r02_rocprofv3_kernel_stats_bench_cfg3.{md,csv}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/rp
rm -rf $O && mkdir -p $O
CMD="rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --steps 5 --warmup 2 --cpu-sample 0"
timeout -k 10 500 $CMD > gpurun_out/bench_profiled.log 2>&1 || { tail -5 gpurun_out/bench_profiled.log; exit 1; }
python tools/kernel_stats.py $O gpurun_out/r02_rocprofv3_kernel_stats_bench_cfg3.md gpurun_out/r02_rocprofv3_kernel_stats_bench_cfg3.csv "$CMD" "config 3: 2,000,000 reads x 20 kb, 2,163,774 records, 3.99e9 CIGAR ops = 16.04 GB algorithmic bytes per scan launch; 7 graph builds; the BAM legs on 200,000 reads (1.87 GB BGZF): two decodes + one end-to-end load" > /dev/null
grep '^{"metric"' gpurun_out/bench_profiled.log | tail -1 > gpurun_out/bench_profiled_line.json
rm -rf $O
cat gpurun_out/r02_rocprofv3_kernel_stats_bench_cfg3.md
