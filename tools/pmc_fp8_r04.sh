#!/bin/bash
# PMC passes over the fp8 batched prefill (gemm8_256x_k): matrix-pipe busy, LDS activity / conflicts, wave wait states, fabric traffic. One group per pass, --kernel-trace only.
set -u
ROOT=$PWD
export PYTHONPATH=$ROOT TMPDIR=/tmp OUT=$ROOT/gpurun_out/pmc8
mkdir -p $OUT
GROUPS_=${GROUPS_:-"SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CYCLES,GRBM_GUI_ACTIVE,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_INSTS_LDS SQ_WAVE_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_WAIT_INST_LDS,SQ_ACTIVE_INST_ANY,SQ_ACTIVE_INST_LDS FETCH_SIZE WRITE_SIZE"}
i=0
for grp in $GROUPS_; do
  i=$((i+1)); tag=$(echo $grp | cut -d, -f1)${TAG:-}
  rm -rf /tmp/pmc8_$i
  (cd /tmp && timeout 240 rocprofv3 --pmc $(echo $grp | tr ',' ' ') --kernel-trace -d /tmp/pmc8_$i -o pmc --output-format rocpd -- python $ROOT/tools/prefill_only.py 32 160 2 fp8 > /tmp/pmc8_$i.log 2>&1)
  db=$(find /tmp/pmc8_$i -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then python $ROOT/tools/pmc_summary.py $db gemm8_256 > $OUT/pmc8_$tag.txt 2>&1; else tail -8 /tmp/pmc8_$i.log > $OUT/pmc8_$tag.txt; fi
done
