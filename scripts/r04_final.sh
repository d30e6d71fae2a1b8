#!/bin/bash
# Round-4 closing measurement pass on the GPU box (one gpurun call).  Outputs under gpurun_out/r04f; the summaries judged are copied to
# profiles/ afterwards (scripts/r04_final_collect.py).  Every rocprofv3 command has the program directly after `--`; counters are
# collected in passes of their own (no trace domains beside --pmc).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04f
mkdir -p $O && cd /tmp && export TMPDIR=/tmp
cd $R
echo "[1] kernel stats of the bench command at the three resolutions"; date
for wl in direct128 direct64 direct256; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -o p -- python3 bench.py --workload $wl --steps 10 --warmup 3 --no-extra --no-cpu-baseline > $O/bench_${wl}_under_rocprof.json 2> $O/bench_${wl}_under_rocprof.err || exit 11
  python3 scripts/prof_summary.py $(find $O/prof_$wl -name "*kernel_stats.csv") 16 40 > $O/r04_bench_${wl}_rocprofv3_kernel_stats.txt
  rm -rf $O/prof_$wl
done
echo "[2] FETCH_SIZE / WRITE_SIZE of the bench kernels, three resolutions"; date
for wl in direct128 direct64 direct256; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf_$wl -o f -- python3 bench.py --workload $wl --graph off --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-extra > $O/pf_$wl.log 2>&1 || exit 12
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw_$wl -o w -- python3 bench.py --workload $wl --graph off --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-extra > $O/pw_$wl.log 2>&1 || exit 13
  python3 scripts/pmc_summary.py $(find $O/pf_$wl -name "*counter_collection.csv") $(find $O/pw_$wl -name "*counter_collection.csv") > $O/r04_pmc_fetch_write_bench_$wl.txt || exit 14
  rm -rf $O/pf_$wl $O/pw_$wl
done
echo "[3] SQ counters of the attention kernels"; date
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA --output-format csv -d $O/pa1 -o a -- python3 scripts/attn_only.py 128 0.1 > $O/pa1.log 2>&1 || exit 15
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM --output-format csv -d $O/pa2 -o a -- python3 scripts/attn_only.py 128 0.1 > $O/pa2.log 2>&1 || exit 16
python3 scripts/pmc_summary.py $(find $O/pa1 -name "*counter_collection.csv") $(find $O/pa2 -name "*counter_collection.csv") > $O/r04_pmc_attention_selfattn_N32768_p0.1.txt || exit 17
rm -rf $O/pa1 $O/pa2
echo "[4] HBM-bound kernels: durations, FETCH / WRITE"; date
HVC_HBM_CASES=$O/hbm_cases.json rocprofv3 --kernel-trace --stats --output-format csv -d $O/hbm_t -o t -- python3 scripts/hbm_kernels.py > $O/hbm_events.log 2>&1 || exit 18
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/hbm_f -o f -- python3 scripts/hbm_kernels.py > $O/hbm_f.log 2>&1 || exit 19
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/hbm_w -o w -- python3 scripts/hbm_kernels.py > $O/hbm_w.log 2>&1 || exit 20
python3 scripts/hbm_summary.py $O/hbm_cases.json $(find $O/hbm_t -name "*kernel_trace.csv") $(find $O/hbm_f -name "*counter_collection.csv") $(find $O/hbm_w -name "*counter_collection.csv") > $O/r04_hbm_kernels.txt 2>&1 || exit 21
rm -rf $O/hbm_t $O/hbm_f $O/hbm_w
echo "[5] GEMM tables, attention A/B"; date
python3 scripts/gemm_vs_blas.py > $O/r04_gemm_vs_hipblaslt.txt 2> $O/gemm_vs.err || exit 22
python3 scripts/gemm_shapes.py direct128 > $O/r04_gemm_shapes_direct128.txt 2> $O/gemm_shapes.err || exit 23
python3 scripts/attn_pipe_ab.py fwd > $O/r04_attention_pipelined_forward_ab.txt 2> $O/attn_ab.err || exit 24
python3 scripts/attn_shapes.py > $O/r04_attention_shapes.txt 2> $O/attn_shapes.err || exit 25
echo "[6] cascade stage 3 (256^3) step under rocprofv3"; date
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o p -- python3 scripts/cascade_fullsize.py 3 1 4 > $O/cascade3_steps.log 2> $O/cascade3.err || exit 26
python3 scripts/prof_summary.py $(find $O/prof_c3 -name "*kernel_stats.csv") 4 40 > $O/r04_cascade_stage3_256_rocprofv3_kernel_stats.txt
rm -rf $O/prof_c3
tail -3 $O/cascade3_steps.log
python3 scripts/cascade_fullsize.py 3 1 6 2>&1 | grep -v amdgpu.ids > $O/cascade3_plain_steps.log || exit 30
python3 scripts/cascade_fullsize.py 2 2 6 2>&1 | grep -v amdgpu.ids > $O/cascade2_plain_steps.log || exit 31
python3 scripts/glue_layers.py 256 3 2>&1 | grep -v amdgpu.ids > $O/glue_after.log || exit 32
echo "[7] default bench line (with cpu_baseline and the other resolutions), and the same with --ddp (RCCL, world size 1)"; date
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 27
python3 bench.py --ddp --no-cpu-baseline --no-extra > $O/bench_ddp_rccl_world1.json 2> $O/bench_ddp.err || exit 28
python3 bench.py --workload direct64 --ddp --no-cpu-baseline --no-extra > $O/bench_direct64_ddp_graph.json 2> $O/bench_ddp64.err || exit 29
tail -c 1200 $O/bench_default.json
echo; date
