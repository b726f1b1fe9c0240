#!/bin/bash
# fp8 batched prefill: A/B of gemm8.hip variants selected by an environment variable (VAR, default RDX_GEMM8_W4) over MODES; TESTS=1 runs the fp8 GEMM unit
# tests per mode. Output -> gpurun_out/fp8mx/ (fresh per call).
ROOT=$(pwd); export PYTHONPATH=$ROOT; OUT=$ROOT/gpurun_out/fp8mx; mkdir -p $OUT
MODES=${MODES:-"0 1"}; VAR=${VAR:-RDX_GEMM8_W4}
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
