#!/bin/bash
# Round-4: rocprofv3 kernel stats of the default bench command (and the 64^3 / 256^3 workloads), summaries under gpurun_out/r04k.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04k
mkdir -p $O && cd /tmp && export TMPDIR=/tmp
cd $R
for wl in direct128 direct64 direct256; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -o p -- python3 bench.py --workload $wl --steps 10 --warmup 3 --no-extra --no-cpu-baseline > $O/bench_${wl}_under_rocprof.json 2> $O/bench_${wl}_under_rocprof.err || exit 15
  python3 scripts/prof_summary.py $(find $O/prof_$wl -name "*kernel_stats.csv") 16 40 > $O/r04_bench_${wl}_rocprofv3_kernel_stats.txt
  rm -rf $O/prof_$wl
done
head -45 $O/r04_bench_direct128_rocprofv3_kernel_stats.txt
