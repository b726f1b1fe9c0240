"""Per-decode-step duration of a kernel from a rocprofv3 rocpd database of one bench step (how a decode kernel's time grows with the context):
python tools/attn_ctx_profile.py <db> <kernel-substring> [layers]  -- dispatches in start order, averaged over groups of `layers` (one decode step)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); pat = sys.argv[2]; L = int(sys.argv[3]) if len(sys.argv) > 3 else 32
rows = db.execute("select start, end - start from kernels where name like ? order by start", (f"%{pat}%",)).fetchall()
d = [r[1] / 1e3 for r in rows]
steps = [sum(d[i:i + L]) / L for i in range(0, len(d) - L + 1, L)]
print(f"{len(d)} dispatches, {len(steps)} steps of {L}")
for i in range(0, len(steps), 16):
    print(f"step {i:4d}: " + " ".join(f"{x:6.2f}" for x in steps[i:i + 16]))
