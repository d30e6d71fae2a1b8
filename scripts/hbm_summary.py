"""profiles/r03_hbm_kernels.txt from one scripts/hbm_kernels.py case list + rocprofv3 outputs of the same command:
kernel trace (durations) and two --pmc passes (FETCH_SIZE, WRITE_SIZE).  The script brackets every case with hvc cast launches
(warm-up | timed); this cuts each CSV at those marks and adds up, per case, every kernel launched inside the timed window.
usage: hbm_summary.py cases.json kernel_trace.csv fetch_counter_collection.csv write_counter_collection.csv"""
import csv
import json
import re
import sys

cases = json.load(open(sys.argv[1]))


def windows(path, value_of):
    """[(kernel name, value)] per window between consecutive sentinel (cast) launches, in dispatch order."""
    rows = []
    for r in csv.DictReader(open(path)):
        v = value_of(r)
        if v is not None:
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], v))
    rows.sort()
    out, cur = [], None
    for _, name, v in rows:
        if "cast_kernel" in name:
            if cur is not None:
                out.append(cur)
            cur = []
        elif cur is not None and "at::native" not in name and "rocclr" not in name:
            cur.append((name, v))
    return out


_dm = {}


def short(n):
    if n not in _dm:
        d = n
        if n.startswith("_Z"):
            try:
                import subprocess
                d = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True, check=True).stdout.strip()
            except Exception:       # noqa: BLE001 - cosmetic
                pass
        d = re.sub(r"^void |hvc::\(anonymous namespace\)::|\(anonymous namespace\)::", "", d)
        _dm[n] = re.sub(r"[<(].*", "", d)
    return _dm[n]


dur = windows(sys.argv[2], lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
fetch = windows(sys.argv[3], lambda r: float(r["Counter_Value"]) if r["Counter_Name"] == "FETCH_SIZE" else None)
write = windows(sys.argv[4], lambda r: float(r["Counter_Value"]) if r["Counter_Name"] == "WRITE_SIZE" else None)
assert len(dur) >= 3 * len(cases) - 1, (len(dur), len(cases))
print("# HBM-bound kernels on working sets >= 268 MB per launch, a ring of 3 distinct operand sets (scripts/hbm_kernels.py), MI355X.")
print("# us: rocprofv3 --kernel-trace, sum over the kernels of one call, mean over the timed calls; alg MB: every input read once, every")
print("# output written once; HBM MB: --pmc FETCH_SIZE (x2: gfx950 tallies 128-byte requests as 64, MI355X_MICROARCH.md) + WRITE_SIZE,")
print("# KiB -> bytes, separate passes, per call; peak 8 TB/s spec (6.3 TB/s measured achievable for a float4 copy).")
print(f"{'case':46s} {'us':>8s} {'alg MB':>8s} {'GB/s':>7s} {'of 8TB/s':>8s} {'of 6.3':>7s} {'HBM MB':>8s} {'HBM/alg':>7s}  kernels (launches per call)")
for i, c in enumerate(cases):
    w = 3 * i + 1                                  # [warm-up window, timed window, rest] per case
    n = c["launches"]
    us = sum(v for _, v in dur[w]) / 1e3 / n
    names = {}
    for k, _ in dur[w]:
        names[short(k)] = names.get(short(k), 0) + 1
    traffic = (2 * sum(v for _, v in fetch[w]) + sum(v for _, v in write[w])) * 1024 / n if w < len(fetch) and w < len(write) else float("nan")
    gbps = c["bytes"] / us / 1e3
    ks = ", ".join(f"{k} x{v / n:g}" for k, v in names.items())
    print(f"{c['name']:46s} {us:8.1f} {c['bytes'] / 1e6:8.1f} {gbps:7.0f} {gbps / 8000:8.2f} {gbps / 6300:7.2f} {traffic / 1e6:8.1f} {traffic / c['bytes']:7.2f}  {ks}")
