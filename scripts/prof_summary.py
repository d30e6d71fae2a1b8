"""Summarise a rocprofv3 --kernel-trace --stats CSV (kernel_stats.csv) as a short table."""
import csv, sys, re
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"_ZN3hvc12_GLOBAL__N_1\d+", "hvc::", n)
    return n[:96]
f = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# {f}: total kernel time {tot/1e6:.2f} ms over {steps:g} steps (incl. warmup) -> {tot/1e6/steps:.2f} ms/step")
print(f"{'kernel':96s} {'calls':>7s} {'total_ms':>9s} {'avg_us':>9s} {'%':>6s}")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{short(r['Name']):96s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['AverageNs'])/1e3:9.1f} {float(r['Percentage']):6.2f}")
