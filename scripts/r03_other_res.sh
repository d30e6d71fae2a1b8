#!/bin/bash
# rocprofv3 kernel stats of the 64^3 / 256^3 bench commands on the closing build (the same commands as scripts/r03_measure.sh)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03m; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $R
for wl in direct64 direct256; do
  rm -rf $O/prof_$wl
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -o p -- python3 bench.py --workload $wl --steps 10 --warmup 3 --no-extra --no-cpu-baseline > $O/bench_${wl}_under_rocprof.json 2> $O/bench_${wl}_under_rocprof.err || exit 15
  python3 scripts/prof_summary.py $(find $O/prof_$wl -name "*kernel_stats.csv") 16 30 > $O/r03_bench_${wl}_rocprofv3_kernel_stats.txt
  rm -f $(find $O/prof_$wl -name "*kernel_trace.csv")
  head -6 $O/r03_bench_${wl}_rocprofv3_kernel_stats.txt | cut -c1-150
done
