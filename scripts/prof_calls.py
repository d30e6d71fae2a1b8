"""List the individual dispatches of kernels matching a substring from a rocprofv3 rocpd results .db (time order, grid, duration)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "gemm_kernel"
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch_")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol_")][0]
rows = cur.execute(f"select s.kernel_name, d.grid_size_x, d.workgroup_size_x, d.end - d.start, d.start from {kd} d join {ks} s "
                   f"on d.kernel_id = s.id where s.kernel_name like ? order by d.start", (f"%{pat}%",)).fetchall()
if rows:
    t0 = rows[0][4]
    for n, g, wg, dur, st in rows:
        if dur / 1e3 >= min_us:
            print(f"{(st - t0) / 1e6:10.1f} ms  wgs {g // max(wg, 1):7d}  {dur / 1e3:10.1f} us  {n[:110]}")
