#!/bin/bash
# PMC passes over the batch-32 encode (pconv_k family): HBM traffic and MFMA busy per kernel. One counter group per pass (--kernel-trace only).
set -u
ROOT=$PWD
export PYTHONPATH=$ROOT TMPDIR=/tmp OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rm -rf /tmp/pmc_enc_$tag
  (cd /tmp && rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_enc_$tag -o pmc --output-format rocpd -- python $ROOT/tools/enc_only.py 32 3 > /tmp/pmc_enc_$tag.log 2>&1)
  db=$(find /tmp/pmc_enc_$tag -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then python $ROOT/tools/pmc_summary.py $db rdx > $OUT/pmc_enc_$tag.txt 2>&1; else tail -8 /tmp/pmc_enc_$tag.log > $OUT/pmc_enc_$tag.txt; fi
done
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "batch32" --durations=3 > $OUT/t_b32_fulldepth.log 2>&1; echo "rc $?" >> $OUT/t_b32_fulldepth.log
