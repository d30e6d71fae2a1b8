#!/bin/bash
# rocprofv3 kernel stats of the default bench command (direct128), eager (the graph replay hides kernels from the trace)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03p; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof128 -o p -- python3 bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline > $O/bench_direct128_under_rocprof.json 2> $O/bench_direct128_under_rocprof.err || exit 15
python3 scripts/prof_summary.py $(find $O/prof128 -name "*kernel_stats.csv") 16 40 > $O/r03_bench_direct128_rocprofv3_kernel_stats.txt
rm -f $(find $O/prof128 -name "*kernel_trace.csv")
cat $O/r03_bench_direct128_rocprofv3_kernel_stats.txt
