"""Dev aid: one character per instruction of a kernel's loops in a hipcc -S listing (M mfma, E transcendental, v other VALU, R/W LDS read/write,
G/S global load/store, w s_waitcnt, B barrier, n s_nop, s other scalar, > branch, | label) - shows how the compiler interleaved the phases."""
import re, sys
path, needle = sys.argv[1], sys.argv[2]
start_at = int(sys.argv[3]) if len(sys.argv) > 3 else 0
L = open(path).read().split("\n")
a = next(i for i, l in enumerate(L) if l.startswith("_ZN") and needle in l and l.rstrip().endswith(":") is False and ":" in l)
b = next(i for i in range(a, len(L)) if L[i].startswith(".Lfunc_end"))
seq = []
for l in L[a:b]:
    t = l.strip()
    if not t or t.startswith(";"): continue
    op = t.split()[0]
    if op.startswith(".LBB"):
        seq.append("\n|" + op.rstrip(":") + (" LOOP " if "Loop Header" in l else " ")); continue
    if op.startswith("."): continue
    c = ("M" if op.startswith("v_mfma") else "E" if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")) else "v" if op.startswith("v_") else
         "R" if op.startswith("ds_read") else "W" if op.startswith("ds_write") else "G" if op.startswith(("global_load", "buffer_load")) else
         "S" if op.startswith(("global_store", "buffer_store")) else "w" if op.startswith("s_waitcnt") else "B" if op.startswith("s_barrier") else
         "n" if op.startswith("s_nop") else ">" if op.startswith(("s_cbranch", "s_branch")) else "s" if op.startswith("s_") else "?")
    seq.append(c)
out = "".join(seq)
blocks = [b for b in out.split("\n") if b.count("M") >= 4]
for b in blocks[start_at:start_at + 6]: print(b[:900])
