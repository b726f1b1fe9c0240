#!/bin/bash
set -u
ROOT=$PWD
export PYTHONPATH=$ROOT TMPDIR=/tmp
OUT=$ROOT/gpurun_out/r04d
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -x -q -k "encode or two_image or flash" > $OUT/t_parity_enc.log 2>&1; echo "rc $?" >> $OUT/t_parity_enc.log
python -m pytest tests/test_gpu_api.py tests/test_gpu_fullsize.py -x -q -s -k "not full_depth" > $OUT/t_full.log 2>&1; echo "rc $?" >> $OUT/t_full.log
for b in 1 2 8 32 256; do python tools/enc_only.py $b 5 >> $OUT/enc_only.log 2>&1; done
for b in 1 32; do RDX_PCONV=0 python tools/enc_only.py $b 5 >> $OUT/enc_only_old.log 2>&1; done
for b in 1 32; do RDX_PCONV_KSPLIT=0 python tools/enc_only.py $b 5 >> $OUT/enc_only_noks.log 2>&1; done
timeout 900 python tools/pconv_check.py 32 auto 4x4 4x4k4 4x2k4 4x2k8 2x4k8 > $OUT/pconv_b32.log 2>&1; echo "rc $?" >> $OUT/pconv_b32.log
timeout 600 python tools/pconv_check.py 1 auto 1x2 1x2k4 1x2k8 2x2k8 > $OUT/pconv_b1.log 2>&1; echo "rc $?" >> $OUT/pconv_b1.log
trace() {  # name, iterations, command...
  local name=$1 iters=$2; shift 2
  rm -rf /tmp/rp_$name
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/rp_$name -o $name --output-format rocpd -- "$@" > /tmp/rp_$name.log 2>&1)
  local db=$(find /tmp/rp_$name -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then python $ROOT/tools/prof_summary.py $db $OUT/$name.md $iters > /dev/null; else tail -5 /tmp/rp_$name.log > $OUT/$name.md; fi
  grep -E "ms/img" /tmp/rp_$name.log | tail -1 >> $OUT/$name.md
}
trace enc_b1 21 python $ROOT/tools/enc_only.py 1 20
trace enc_b32 6 python $ROOT/tools/enc_only.py 32 5
