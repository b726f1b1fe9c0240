#!/bin/bash
# round-4 run 1: new tests + bench line + decode-attention PMC diagnosis (batch 32)
set -u
ROOT=$PWD
export PYTHONPATH=$ROOT TMPDIR=/tmp
OUT=$ROOT/gpurun_out/r04a
mkdir -p $OUT
python -m pytest tests/test_gpu_entrypoints.py -x -q -s > $OUT/entry.log 2>&1; echo "entry rc $?" >> $OUT/entry.log
python -m pytest "tests/test_gpu_fullsize.py::test_full_depth_32_layer_decoder_matches_oracle" -x -q -s > $OUT/fulldepth.log 2>&1; echo "fd rc $?" >> $OUT/fulldepth.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?" >> $OUT/bench.err
# PMC: decode attention at batch 32 (eager, 8 tokens)
pmc() {  # name, counters...
  local name=$1; shift
  rm -rf /tmp/pmc_$name
  (cd /tmp && rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_$name -o pmc --output-format rocpd -- python $ROOT/bench.py --batch 32 --steps 1 --warmup 0 --new-tokens 8 --no-cpu-baseline --no-graph > /tmp/pmc_$name.log 2>&1)
  local db=$(find /tmp/pmc_$name -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then python $ROOT/tools/pmc_summary.py $db decode_attention > $OUT/pmc_$name.txt 2>&1; else tail -8 /tmp/pmc_$name.log > $OUT/pmc_$name.txt; fi
}
pmc sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_WAVES
pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pmc grbm GRBM_GUI_ACTIVE
pmc tcp TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
ls -la $OUT
