#!/bin/bash
# encoder-only rocprofv3 pass: per-kernel table + per-dispatch listing of one encode at batch $1 (default 32)
set -u
export PYTHONPATH=$PWD TMPDIR=/tmp
ROOT=$PWD; B=${1:-32}; IT=${2:-5}
OUT=$PWD/gpurun_out/prof; mkdir -p $OUT
name=enc_b$B
rm -rf /tmp/rp_$name
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/rp_$name -o $name --output-format rocpd -- python $ROOT/tools/enc_only.py $B $IT > /tmp/$name.log 2>&1)
tail -3 /tmp/$name.log | head -1
db=$(find /tmp/rp_$name -name "*.db" | head -1)
python $ROOT/tools/prof_summary.py $db $OUT/$name.md $((IT+1)) > /dev/null
python - "$db" "$OUT/${name}_dispatches.txt" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end, grid_x, workgroup_x, lds_size from kernels where name like '%rdx%' order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "img_prep" in r[0]]
last = rows[idx[-1]:]
t0 = last[0][1]
import re
def short(n):
    m = re.search(r"rdx(?:::|\d+)([a-z_0-9]+?)_k", n)
    return (m.group(1) if m else n[:30]) + ("<" + n.split("<", 1)[1][:40] if "<" in n else n[-40:])
with open(sys.argv[2], "w") as f:
    for r in last:
        f.write(f"{(r[1]-t0)/1e3:8.1f} {(r[2]-r[1])/1e3:8.2f} us wgs {r[3]//max(r[4],1):6d} lds {r[5]:6d} {r[0][:90]}\n")
    f.write(f"span {(last[-1][2]-t0)/1e3:.1f} us, kernel sum {sum(r[2]-r[1] for r in last)/1e3:.1f} us, {len(last)} launches\n")
print(open(sys.argv[2]).read().splitlines()[-1])
PY
