#!/bin/bash
# bf16 batched prefill / encoder after a change of gemm_dma256_k: GEMM unit tests, prefill and encode wall time, per-kernel table
ROOT=$(pwd); export PYTHONPATH=$ROOT; OUT=$ROOT/gpurun_out/bf16_256; mkdir -p $OUT
python -m pytest tests/test_gpu_gemm.py -q -x -k "not fp8" 2>&1 | tail -3 > $OUT/test.log
python tools/prefill_only.py 32 160 5 2>&1 | tail -1 >> $OUT/time.log
python tools/enc_only.py 32 10 2>&1 | tail -1 >> $OUT/time.log
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/profb -o p --output-format rocpd -- python $ROOT/tools/prefill_only.py 32 160 3 > /tmp/profb.log 2>&1)
python tools/prof_summary.py $(find /tmp/profb -name "*.db" | head -1) - 2>/dev/null | head -12 > $OUT/kern.log
