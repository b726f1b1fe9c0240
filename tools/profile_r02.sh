#!/bin/bash
# rocprofv3 kernel-trace baselines of the three phases (run on the GPU box from the repo root): writes gpurun_out/prof/*.md
set -u
export PYTHONPATH=$PWD TMPDIR=/tmp
export OUT=$PWD/gpurun_out/prof; mkdir -p $OUT
run() {  # name, iterations (0 = none), command...
  local name=$1; shift
  local iters=$1; shift
  rm -rf /tmp/rp_$name
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/rp_$name -o $name --output-format rocpd -- "$@" > $OUT/$name.log 2>&1)
  local db=$(find /tmp/rp_$name -name "*.db" | head -1)
  if [ -n "$db" ]; then python $PWD/tools/prof_summary.py $db $OUT/$name.md $iters > /dev/null; else echo "no db for $name" >> $OUT/$name.log; fi
}
run enc_b1 21 python $PWD/tools/enc_only.py 1 20
run enc_b32 6 python $PWD/tools/enc_only.py 32 5
run prefill_b1 11 python $PWD/tools/prefill_only.py 1 160 10
run prefill_b32 4 python $PWD/tools/prefill_only.py 32 160 3
run bench_b1 0 python $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-b32
# per-dispatch listing of ONE batched encode (kernel order = layer order) for the per-layer breakdown
python - <<'PY'
import sqlite3, os
out = os.environ.get("OUT", "gpurun_out/prof")
for name in ("enc_b32", "enc_b1"):
    import glob
    dbs = glob.glob(f"/tmp/rp_{name}/**/*.db", recursive=True)
    if not dbs: continue
    db = dbs[0]
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, start, end, grid_x, workgroup_x, lds_size from kernels where name like '%rdx%' order by start").fetchall()
    # last iteration = everything after the last img_prep launch
    idx = [i for i, r in enumerate(rows) if "img_prep" in r[0]]
    if not idx: continue
    last = rows[idx[-1]:]
    with open(f"{out}/{name}_dispatches.txt", "w") as f:
        t0 = last[0][1]
        for r in last:
            f.write(f"{(r[1]-t0)/1e3:9.1f} us  dur {(r[2]-r[1])/1e3:8.2f} us  wgs {r[3]//max(r[4],1):6d}  lds {r[5]:6d}  {r[0][:100]}\n")
        f.write(f"span {(last[-1][2]-t0)/1e3:.1f} us, kernel sum {sum(r[2]-r[1] for r in last)/1e3:.1f} us, {len(last)} launches\n")
PY
# PMC: HBM traffic of the default decode kernels (separate passes: FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2)
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  (cd /tmp && rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmc_$ctr -o pmc --output-format rocpd -- python $PWD/bench.py --steps 1 --warmup 0 --new-tokens 8 --no-cpu-baseline --no-b32 --no-graph > $OUT/pmc_$ctr.log 2>&1)
  db=$(find /tmp/pmc_$ctr -name "*.db" | head -1)
  if [ -n "$db" ]; then python $PWD/tools/pmc_summary.py $db > $OUT/pmc_$ctr.txt 2>&1; else tail -20 $OUT/pmc_$ctr.log > $OUT/pmc_$ctr.txt; fi
  tail -c 2000 $OUT/pmc_$ctr.log > $OUT/pmc_$ctr.log.tail; rm -f $OUT/pmc_$ctr.log
done
for f in $OUT/*.log; do tail -c 3000 $f > $f.tail; rm -f $f; done
