"""Summarise a rocprofv3 rocpd sqlite DB (kernel trace) as a per-kernel table: python tools/prof_summary.py <db> [out.md]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, max(vgpr_count), max(lds_size), max(grid_x)/max(workgroup_x) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
lines = ["| % | calls | total ms | avg us | min us | max us | vgpr | lds | max WGs | kernel |", "|---|---|---|---|---|---|---|---|---|---|"]
for r in rows[:45]:
    lines.append(f"| {r[2]/tot*100:.1f} | {r[1]} | {r[2]/1e3:.2f} | {r[3]:.2f} | {r[4]:.2f} | {r[5]:.2f} | {r[6]} | {r[7]} | {r[8]} | `{r[0][:110]}` |")
txt = f"total kernel time {tot/1e3:.1f} ms\n\n" + "\n".join(lines) + "\n"
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt)
