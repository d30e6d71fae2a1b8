"""Instruction mix of the MFMA-carrying basic blocks of one kernel in a hipcc -S listing."""
import re, sys
from collections import Counter
path, needle = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and needle in l and ":" in l )
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks, cur, lab = [], [], "entry"
for l in lines[start + 1:end]:
    t = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        blocks.append((lab, cur)); cur, lab = [], m.group(1); continue
    if not t or t.startswith((";", ".", "//")): continue
    cur.append(t.split()[0])
blocks.append((lab, cur))
for lab, ins in blocks:
    n_mfma = sum(1 for x in ins if x.startswith("v_mfma"))
    if n_mfma < 1: continue
    c = Counter()
    for x in ins:
        if x.startswith("v_mfma"): c["MFMA"] += 1
        elif x.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")): c["VALU trans"] += 1
        elif x.startswith("v_"): c["VALU " + re.sub(r"_e(32|64)$", "", x)] += 1
        elif x.startswith("ds_"): c["LDS " + x] += 1
        elif x.startswith("s_"): c["SALU/" + ("waitcnt" if "waitcnt" in x else "barrier" if "barrier" in x else "other")] += 1
        elif x.startswith(("global_", "buffer_", "flat_", "scratch_")): c["VMEM " + x] += 1
        else: c["other " + x] += 1
    valu = sum(v for k, v in c.items() if k.startswith("VALU"))
    print(f"block {lab}: {len(ins)} instrs, MFMA {n_mfma}, VALU {valu} ({valu / n_mfma:.1f} per MFMA)")
    for k, v in sorted(c.items(), key=lambda kv: -kv[1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
        print(f"      {k:40s} {v}")
