#!/bin/bash
# Round-6 profiling passes (run on the GPU box from the repo root); text summaries only -> gpurun_out/prof/.
#   bash tools/profile_r06.sh dec N      kernel table of a 256-token decode at batch N
#   bash tools/profile_r06.sh decfp8 N   the same with fp8 weights (round 6: o_proj / down_proj of decode steps W8A16)
#   bash tools/profile_r06.sh bench      kernel table of the default bench command
set -u
ROOT=$PWD
export PYTHONPATH=$ROOT TMPDIR=/tmp OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
trace() {  # name, iterations (0 = none), command...
  local name=$1 iters=$2; shift 2
  rm -rf /tmp/rp_$name
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/rp_$name -o $name --output-format rocpd -- "$@" > /tmp/rp_$name.log 2>&1)
  local db=$(find /tmp/rp_$name -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then python $ROOT/tools/prof_summary.py $db $OUT/$name.md $iters > /dev/null; else tail -5 /tmp/rp_$name.log > $OUT/$name.md; fi
  grep -E "^\{|ms/img|prefill B|^decode B" /tmp/rp_$name.log | tail -1 | cut -c1-600 >> $OUT/$name.md
}
what=${1:-dec}
if [ $what = dec ]; then
  trace dec_b$2 0 python $ROOT/tools/decode_only.py $2 256 1
fi
if [ $what = decfp8 ]; then
  trace dec_fp8_b$2 0 python $ROOT/tools/decode_only.py $2 256 1 1
fi
if [ $what = bench ]; then
  trace bench_default 0 python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline
fi
du -sh $OUT
