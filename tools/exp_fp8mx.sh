#!/bin/bash
# fp8 batched prefill A/B harness (round 4, profiles/r04_fp8_prefill_mx.md): for every value in MODES of the environment variable VAR -- a switch compiled into
# gemm8.hip for the duration of an experiment; none is left in the tree -- the fp8 GEMM unit tests (TESTS=1), the 32 x 160 prefill wall time and the
# rocprofv3 kernel table. Without VAR it measures the build as it is. Output -> gpurun_out/fp8mx/ (fresh per call).
ROOT=$(pwd); export PYTHONPATH=$ROOT; OUT=$ROOT/gpurun_out/fp8mx; mkdir -p $OUT
MODES=${MODES:-"0"}; VAR=${VAR:-RDX_EXPERIMENT}
export TMPDIR=/tmp
: > $OUT/test.log; : > $OUT/prefill.log
for mx in $MODES; do
  export $VAR=$mx
  [ -n "${TESTS:-}" ] && python -m pytest tests/test_gpu_gemm.py -q -x -k "fp8_x_fp8" 2>&1 | tail -3 >> $OUT/test.log
  python tools/prefill_only.py 32 160 5 fp8 2>&1 | tail -1 >> $OUT/prefill.log
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof$mx -o p --output-format rocpd -- python $ROOT/tools/prefill_only.py 32 160 3 fp8 > /tmp/prof$mx.log 2>&1)
  db=$(find /tmp/prof$mx -name "*.db" | head -1)
  python tools/prof_summary.py $db - 2>/dev/null | head -10 > $OUT/kern_mx$mx.log
done
