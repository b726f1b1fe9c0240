"""Summarise a rocprofv3 rocpd sqlite DB (kernel trace) as a per-kernel table of librdx's kernels:
python tools/prof_summary.py <db> [out.md] [iterations]   (per-iteration ms column when `iterations` is given; load-time and
torch kernels -- weight generation, packing -- are listed in one line only)"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
iters = float(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, max(vgpr_count), max(lds_size), max(grid_x)/max(workgroup_x) from kernels group by name order by 3 desc").fetchall()
load = ("pack_weight", "from_f32")
mine = [r for r in rows if "rdx" in r[0] and not any(l in r[0] for l in load)]
other = sum(r[2] for r in rows if r not in mine)
tot = sum(r[2] for r in mine)
hdr = "| % | calls | total ms |" + (" ms/iter |" if iters else "") + " avg us | min us | max us | vgpr | lds | max WGs | kernel |"
lines = [hdr, "|---" * (11 if iters else 10) + "|"]
for r in mine[:40]:
    it = f" {r[2]/1e3/iters:.3f} |" if iters else ""
    lines.append(f"| {r[2]/tot*100:.1f} | {r[1]} | {r[2]/1e3:.2f} |{it} {r[3]:.2f} | {r[4]:.2f} | {r[5]:.2f} | {r[6]} | {r[7]} | {r[8]} | `{r[0][:120]}` |")
txt = (f"librdx kernel time {tot/1e3:.2f} ms" + (f" = {tot/1e3/iters:.3f} ms per iteration over {iters:g} iterations" if iters else "")
       + f" (weight generation / packing / torch kernels at load: {other/1e3:.1f} ms, not listed)\n\n" + "\n".join(lines) + "\n")
print(txt)
if len(sys.argv) > 2 and sys.argv[2] != "-":
    open(sys.argv[2], "w").write(txt)
