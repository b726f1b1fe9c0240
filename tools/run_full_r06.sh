#!/bin/bash
# Round 6: the whole-suite run on the GPU box: the -m gpu suite (timed), smoke(), the default bench line (+ bench_detail.json); `prof` adds the rocprofv3 kernel
# table of the default bench command and the (short) FETCH_SIZE / WRITE_SIZE passes (tools/profile_r05.sh pmc -> tools/pmc_to_json.py r06).
set -u
ROOT=$PWD
export PYTHONPATH=$ROOT TMPDIR=/tmp
OUT=$ROOT/gpurun_out/r06full
mkdir -p $OUT
( time python -m pytest tests/ -x -q -m gpu --durations=30 ) > $OUT/pytest_gpu.log 2>&1; echo "rc $?" >> $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?" >> $OUT/smoke.log
RDX_BENCH_DETAIL=$OUT/bench_detail.json python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?" >> $OUT/bench.err
if [ "${1:-}" = prof ]; then
  timeout 900 bash tools/profile_r05.sh pmc > /dev/null 2>&1
  timeout 600 bash tools/profile_r06.sh bench > /dev/null 2>&1
fi
