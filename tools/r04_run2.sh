#!/bin/bash
set -u
ROOT=$PWD
export PYTHONPATH=$ROOT TMPDIR=/tmp
OUT=$ROOT/gpurun_out/r04b
mkdir -p $OUT
timeout 900 python tools/pconv_check.py 32 auto 4x4 2x4 8x4 4x2 2x2 > $OUT/pconv_b32.log 2>&1; echo "rc $?" >> $OUT/pconv_b32.log
timeout 600 python tools/pconv_check.py 1 auto 1x4 2x2 1x2 > $OUT/pconv_b1.log 2>&1; echo "rc $?" >> $OUT/pconv_b1.log
python -m pytest tests/test_gpu_entrypoints.py -x -q -s > $OUT/entry.log 2>&1; echo "entry rc $?" >> $OUT/entry.log
python bench.py --no-b32 --no-fp8 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?" >> $OUT/bench.err
