"""Summarise a rocprofv3 --kernel-trace --stats run (kernel_stats.csv or the rocpd results .db) as a short table."""
import csv, sys, re, sqlite3, subprocess


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return out.splitlines()
    except Exception:
        return names


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n[:104]


f = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
if f.endswith(".db"):
    c = sqlite3.connect(f)
    rows = [dict(Name=r[0], Calls=r[1], total_us=r[2], avg_us=r[3], pct=r[4])
            for r in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc")]
else:
    rows = [dict(Name=r["Name"], Calls=int(r["Calls"]), total_us=float(r["TotalDurationNs"]) / 1e3, avg_us=float(r["AverageNs"]) / 1e3,
                 pct=float(r["Percentage"])) for r in csv.DictReader(open(f))]
for r, n in zip(rows, demangle([r["Name"] for r in rows])):
    r["Name"] = n
tot = sum(r["total_us"] for r in rows)
print(f"# {f}: total kernel time {tot/1e3:.2f} ms over {steps:g} steps (incl. warmup and the untimed per-kernel pass) -> {tot/1e3/steps:.2f} ms/step")
print(f"{'kernel':104s} {'calls':>7s} {'total_ms':>9s} {'avg_us':>9s} {'%':>6s}")
for r in rows[:top]:
    print(f"{short(r['Name']):104s} {r['Calls']:>7d} {r['total_us']/1e3:9.2f} {r['avg_us']:9.1f} {r['pct']:6.2f}")
