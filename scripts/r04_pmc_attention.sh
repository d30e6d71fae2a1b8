#!/bin/bash
# Round-4: rocprofv3 --pmc passes over the attention kernels at the 128^3 self-attention shape (dropout 0.1), counters in passes of their own.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04p
mkdir -p $O && cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA --output-format csv -d $O/pa1 -o a -- python3 scripts/attn_only.py 128 0.1 > $O/pa1.log 2>&1 || exit 15
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM --output-format csv -d $O/pa2 -o a -- python3 scripts/attn_only.py 128 0.1 > $O/pa2.log 2>&1 || exit 16
python3 scripts/pmc_summary.py $(find $O/pa1 -name "*counter_collection.csv") $(find $O/pa2 -name "*counter_collection.csv") > $O/r04_pmc_attention_selfattn_N32768_p0.1.txt || exit 17
rm -rf $O/pa1 $O/pa2
python3 bench.py --no-cpu-baseline --no-extra > $O/bench_quick.json 2> $O/bench_quick.err || exit 18
tail -c 1500 $O/bench_quick.json
