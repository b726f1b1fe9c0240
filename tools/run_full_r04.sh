#!/bin/bash
set -u
ROOT=$PWD
export PYTHONPATH=$ROOT TMPDIR=/tmp
OUT=$ROOT/gpurun_out/r04full
mkdir -p $OUT
( time python -m pytest tests/ -x -q -m gpu --durations=15 ) > $OUT/pytest_gpu.log 2>&1; echo "rc $?" >> $OUT/pytest_gpu.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?" >> $OUT/bench.err
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?" >> $OUT/smoke.log
bash tools/profile_r04.sh all > $OUT/profile.log 2>&1
