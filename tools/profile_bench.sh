# rocprofv3 --kernel-trace --stats of the bench command (fewer steps, no CPU baseline, BAM legs on 200,000 reads), summarised into
# gpurun_out/r03_rocprofv3_kernel_stats_bench_<config>.{md,csv}      bash tools/profile_bench.sh [config]
CFG=${1:-cfg3}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/rp
rm -rf $O && mkdir -p $O
CMD="rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --config $CFG --steps 5 --warmup 2 --cpu-sample 0 --bam-reads 200000"
timeout -k 10 500 $CMD > gpurun_out/bench_profiled_$CFG.log 2>&1 || { tail -5 gpurun_out/bench_profiled_$CFG.log; exit 1; }
grep '^{"metric"' gpurun_out/bench_profiled_$CFG.log | tail -1 > gpurun_out/bench_profiled_line_$CFG.json
python tools/kernel_stats.py $O gpurun_out/r03_rocprofv3_kernel_stats_bench_$CFG.md gpurun_out/r03_rocprofv3_kernel_stats_bench_$CFG.csv "$CMD" "$(python3 -c "import json;d=json.load(open('gpurun_out/bench_profiled_line_$CFG.json'));print(d['config']['workload'],'|',d['config']['records'],'records,',d['config']['cigar_ops'],'CIGAR ops,',d['roofline']['algorithmic_bytes_per_launch'],'algorithmic bytes per scan launch | library',d['config']['library'],'| 7 graph builds + the BAM legs on 200,000 reads')")" "$(python3 -c "import json;print(json.load(open('gpurun_out/bench_profiled_line_$CFG.json'))['roofline']['algorithmic_bytes_per_launch'])")" > /dev/null
rm -rf $O
cat gpurun_out/r03_rocprofv3_kernel_stats_bench_$CFG.md
