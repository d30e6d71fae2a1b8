#!/bin/bash
# Round-3 measurement pass on the GPU box (one gpurun call): HBM-kernel table, rocprofv3 kernel stats of the 64^3 / 256^3 bench
# commands, the default bench line.  Outputs under gpurun_out/; the summaries judged are copied to profiles/ afterwards.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03m
mkdir -p $O && cd /tmp && export TMPDIR=/tmp
cd $R
HVC_HBM_CASES=$O/hbm_cases.json rocprofv3 --kernel-trace --stats --output-format csv -d $O/hbm_t -o t -- python3 scripts/hbm_kernels.py > $O/hbm_events.log 2>&1 || exit 11
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/hbm_f -o f -- python3 scripts/hbm_kernels.py > $O/hbm_f.log 2>&1 || exit 12
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/hbm_w -o w -- python3 scripts/hbm_kernels.py > $O/hbm_w.log 2>&1 || exit 13
python3 scripts/hbm_summary.py $O/hbm_cases.json $(find $O/hbm_t -name "*kernel_trace.csv") $(find $O/hbm_f -name "*counter_collection.csv") $(find $O/hbm_w -name "*counter_collection.csv") > $O/r03_hbm_kernels.txt 2>&1 || exit 14
cat $O/r03_hbm_kernels.txt
for wl in direct64 direct256; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -o p -- python3 bench.py --workload $wl --steps 10 --warmup 3 --no-extra --no-cpu-baseline > $O/bench_${wl}_under_rocprof.json 2> $O/bench_${wl}_under_rocprof.err || exit 15
  python3 scripts/prof_summary.py $(find $O/prof_$wl -name "*kernel_stats.csv") 16 30 > $O/r03_bench_${wl}_rocprofv3_kernel_stats.txt
  rm -f $(find $O/prof_$wl -name "*kernel_trace.csv")
done
rm -rf $O/hbm_t $O/hbm_f $O/hbm_w
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 16
tail -c 3000 $O/bench_default.json
