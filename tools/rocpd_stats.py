"""Per-kernel statistics from a rocprofv3 `*_results.db` (rocpd sqlite), for runs whose CSV summary was not written: name, calls, total / average µs, share."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(db.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot / 1e3:.0f} us over {steps:g} steps = {tot / 1e3 / steps:.0f} us per step")
print("name,calls,calls_per_step,total_us,avg_us,min_us,max_us,percent")
for n, c, t, a, lo, hi in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 60]:
    print(f"\"{n[:110]}\",{c},{c / steps:.1f},{t / 1e3:.0f},{a / 1e3:.2f},{lo / 1e3:.2f},{hi / 1e3:.2f},{100.0 * t / tot:.2f}")
