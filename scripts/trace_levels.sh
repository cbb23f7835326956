#!/bin/bash
# developer aid: per-level durations and read bandwidth of the SpTRSV launches (rocprofv3 kernel trace of a short bench run)
# usage: scripts/trace_levels.sh NAME [ENV=VALUE ...]
cd "$(dirname "$0")/.." || exit 1
name=$1; shift
out=gpurun_out/trace_$name
rm -rf "$out" && mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
env "$@" HPDDM_HIP_LEVEL_STATS=$OLDPWD/$out/levels.txt rocprofv3 --kernel-trace -d $OLDPWD/$out -o t -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gmres --no-two-level $BENCH_EXTRA > $OLDPWD/$out/bench.log 2>&1
cd $OLDPWD
db=$(find "$out" -name "*.db" | head -1)
n=$(grep '^{"metric"' "$out/bench.log" | tail -1 | python -c 'import sys, json; print(int(json.loads(sys.stdin.readline())["config"]["launches_per_sptrsv"]) - 2)')
echo "== $name $* (launches $n)"
python scripts/prof_levels.py "$db" "$n" "$out/levels.txt" | tee "$out/levels_table.txt"
if [ -s "$out/levels_table.txt" ] && grep -q "^sum" "$out/levels_table.txt"; then rm -f "$db"; fi
