#!/bin/bash
# Round-3 closing measurement pass on the GPU box (one gpurun call).  Outputs under gpurun_out/r03f; the summaries judged are copied to
# profiles/ afterwards (scripts/r03_final_collect.py).  Every rocprofv3 command has the program directly after `--`; counters are
# collected in passes of their own (no trace domains beside --pmc).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03f
mkdir -p $O && cd /tmp && export TMPDIR=/tmp
cd $R
echo "[1] kernel stats of the default bench command"; date
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof128 -o p -- python3 bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline > $O/bench_direct128_under_rocprof.json 2> $O/bench_direct128_under_rocprof.err || exit 11
python3 scripts/prof_summary.py $(find $O/prof128 -name "*kernel_stats.csv") 16 40 > $O/r03_bench_direct128_rocprofv3_kernel_stats.txt
rm -f $(find $O/prof128 -name "*kernel_trace.csv")
echo "[2] FETCH_SIZE / WRITE_SIZE of the bench kernels"; date
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-extra > $O/pf.log 2>&1 || exit 12
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-extra > $O/pw.log 2>&1 || exit 13
python3 scripts/pmc_summary.py $(find $O/pf -name "*counter_collection.csv") $(find $O/pw -name "*counter_collection.csv") > $O/r03_pmc_fetch_write_bench_direct128.txt || exit 14
echo "[3] SQ counters of the attention kernels"; date
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA --output-format csv -d $O/pa1 -o a -- python3 scripts/attn_only.py 128 0.1 > $O/pa1.log 2>&1 || exit 15
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM --output-format csv -d $O/pa2 -o a -- python3 scripts/attn_only.py 128 0.1 > $O/pa2.log 2>&1 || exit 16
python3 scripts/pmc_summary.py $(find $O/pa1 -name "*counter_collection.csv") $(find $O/pa2 -name "*counter_collection.csv") > $O/r03_pmc_attention_selfattn_N32768_p0.1.txt || exit 17
echo "[4] GEMM tables"; date
python3 scripts/gemm_vs_blas.py > $O/r03_gemm_vs_hipblaslt.txt 2> $O/gemm_vs.err || exit 18
python3 scripts/gemm_shapes.py direct128 > $O/r03_gemm_shapes_direct128.txt 2> $O/gemm_shapes.err || exit 19
echo "[5] attention A/B against the round-2 kernels is in profiles/r03_attention_*; fp8 table"; date
python3 scripts/attn_fp8_bench.py > $O/r03_attention_fp8_x16_vs_bf16.txt 2> $O/fp8.err || exit 20
echo "[6] cascade stage 3 (256^3) step under rocprofv3"; date
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o p -- python3 scripts/cascade_fullsize.py 3 1 4 > $O/cascade3_steps.log 2> $O/cascade3.err || exit 21
python3 scripts/prof_summary.py $(find $O/prof_c3 -name "*kernel_stats.csv") 4 40 > $O/r03_cascade_stage3_256_rocprofv3_kernel_stats.txt
rm -f $(find $O/prof_c3 -name "*kernel_trace.csv")
tail -3 $O/cascade3_steps.log
echo "[7] default bench line (with cpu_baseline and the other resolutions)"; date
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 22
tail -c 1500 $O/bench_default.json
echo; date
