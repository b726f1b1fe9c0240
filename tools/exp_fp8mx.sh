#!/bin/bash
# fp8 batched prefill on the 32x32x64 block-scaled MFMA (gemm8_256x_k) against the 16x16x32 fp8 kernel: RDX_GEMM8_MX = 0 (old), 4 / 5 (ring depth)
ROOT=$(pwd); export PYTHONPATH=$ROOT; OUT=$ROOT/gpurun_out/fp8mx; mkdir -p $OUT
MODES=${MODES:-"0 4 5"}
[ -n "$TESTS" ] && python -m pytest tests/test_gpu_gemm.py -q -x -k "fp8_x_fp8" 2>&1 | tail -5 > $OUT/test.log
for mx in $MODES; do
  RDX_GEMM8_MX=$mx python tools/prefill_only.py 32 160 5 fp8 2>&1 | tail -1 >> $OUT/prefill.log
done
export TMPDIR=/tmp
for mx in $MODES; do
  (cd /tmp && RDX_GEMM8_MX=$mx rocprofv3 --kernel-trace --stats -d /tmp/prof$mx -o p --output-format rocpd -- python $ROOT/tools/prefill_only.py 32 160 3 fp8 > /tmp/prof$mx.log 2>&1)
  db=$(find /tmp/prof$mx -name "*.db" | head -1)
  python tools/prof_summary.py $db - 2>/dev/null | head -24 > $OUT/kern_mx$mx.log
done
