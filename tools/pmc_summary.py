"""Per-kernel average of a PMC counter from a rocprofv3 rocpd database: python tools/pmc_summary.py <db> [name-filter]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
filt = sys.argv[2] if len(sys.argv) > 2 else "rdx"
cols = [d[0] for d in cur.execute("select * from pmc_events limit 1").description]
print(cols)
q = """select p.name, p.counter_name, count(*), avg(p.counter_value), min(p.counter_value), max(p.counter_value)
       from pmc_events p
       where p.name like ? group by p.name, p.counter_name order by 4 desc"""
try:
    rows = cur.execute(q, (f"%{filt}%",)).fetchall()
except Exception as e:
    print("query failed:", e)
    rows = []
for r in rows[:200]:
    print(f"{r[1]:12s} n={r[2]:6d} avg={r[3]:14.1f} min={r[4]:14.1f} max={r[5]:14.1f}  {r[0][:110]}")
