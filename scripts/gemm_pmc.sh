#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/gemm_pmc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $R
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- python3 scripts/gemm_pmc.py > $O/f.log 2>&1 || exit 11
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- python3 scripts/gemm_pmc.py > $O/w.log 2>&1 || exit 12
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/h -o h -- python3 scripts/gemm_pmc.py > $O/h.log 2>&1 || exit 13
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/r -o r -- python3 scripts/gemm_pmc.py > $O/r.log 2>&1 || exit 14
for d in f w h r; do f=$(find $O/$d -name "*counter_collection.csv"); python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Kernel_Name"]
    if "gemm_kernel" in n or "Cijk" in n or "gemm" in n.lower():
        print(r["Counter_Name"], n[:60], r["Counter_Value"], r.get("Grid_Size"), r.get("LDS_Block_Size"))
PY
done
