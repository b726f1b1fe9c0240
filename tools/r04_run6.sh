#!/bin/bash
set -u
ROOT=$PWD
export PYTHONPATH=$ROOT TMPDIR=/tmp
OUT=$ROOT/gpurun_out/r04f
mkdir -p $OUT
python -m pytest tests/test_gpu_api.py -x -q > $OUT/t_api.log 2>&1; echo "rc $?" >> $OUT/t_api.log
python -m pytest tests/test_gpu_parity.py -x -q -k "encode or two_image" > $OUT/t_parity_enc.log 2>&1; echo "rc $?" >> $OUT/t_parity_enc.log
python tools/enc_only.py 1 30 >> $OUT/enc_only.log 2>&1
RDX_ENC_GRAPH=0 python tools/enc_only.py 1 30 >> $OUT/enc_only.log 2>&1
for b in 2 8 32; do python tools/enc_only.py $b 5 >> $OUT/enc_only.log 2>&1; done
python tools/attn_time.py 32 > $OUT/attn_time_b32.log 2>&1
