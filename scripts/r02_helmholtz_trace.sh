R=$PWD; out=$R/gpurun_out/r02h; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
ARGS="--problem helmholtz --grid 64 --mu 8 --steps 20 --warmup 3"
timeout 400 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python $R/bench.py $ARGS > $out/trace.log 2>&1
db=$(find $out/trace -name "*.db" | head -1)
python $R/scripts/prof_summary.py "$db" rocprofv3 --kernel-trace --stats -- python bench.py $ARGS > $out/kernel_stats.csv
grep '^{"metric"' $out/trace.log | tail -1 > $out/trace_bench_line.json
rm -rf $out/trace
head -25 $out/kernel_stats.csv
