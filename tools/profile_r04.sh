#!/bin/bash
# Round-4 profiling pass (run on the GPU box from the repo root): rocprofv3 kernel-trace summaries of the encoder (batch 1 / 32), the prefill and
# the bench, the per-shape table of the packed convolution family against the row-major kernels, the PMC passes of the batch-32 decode attention.
# Text summaries only -> gpurun_out/prof/.
set -u
ROOT=$PWD
export PYTHONPATH=$ROOT TMPDIR=/tmp OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
trace() {  # name, iterations (0 = none), command...
  local name=$1 iters=$2; shift 2
  rm -rf /tmp/rp_$name
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/rp_$name -o $name --output-format rocpd -- "$@" > /tmp/rp_$name.log 2>&1)
  local db=$(find /tmp/rp_$name -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then python $ROOT/tools/prof_summary.py $db $OUT/$name.md $iters > /dev/null; else tail -5 /tmp/rp_$name.log > $OUT/$name.md; fi
  grep -E "^\{|ms/img|prefill B" /tmp/rp_$name.log | tail -1 | cut -c1-600 >> $OUT/$name.md
}
pmc() {  # name, counters...
  local name=$1; shift
  rm -rf /tmp/pmc_$name
  (cd /tmp && rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_$name -o pmc --output-format rocpd -- python $ROOT/bench.py --batch 32 --steps 1 --warmup 0 --new-tokens 8 --no-cpu-baseline --no-graph > /tmp/pmc_$name.log 2>&1)
  local db=$(find /tmp/pmc_$name -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then python $ROOT/tools/pmc_summary.py $db decode_attention > $OUT/pmc_$name.txt 2>&1; else tail -8 /tmp/pmc_$name.log > $OUT/pmc_$name.txt; fi
}
what=${1:-all}
if [ $what = all ] || [ $what = trace ]; then
  trace enc_b1 21 python $ROOT/tools/enc_only.py 1 20
  trace enc_b32 6 python $ROOT/tools/enc_only.py 32 5
  trace prefill_b1 11 python $ROOT/tools/prefill_only.py 1 160 10
  trace bench_default 0 python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline
fi
if [ $what = all ] || [ $what = shapes ]; then
  python $ROOT/tools/pconv_check.py 32 auto > $OUT/pconv_shapes_b32.txt 2>&1
  python $ROOT/tools/pconv_check.py 1 auto > $OUT/pconv_shapes_b1.txt 2>&1
  for b in 1 2 8 32 256; do python $ROOT/tools/enc_only.py $b 10 >> $OUT/enc_only.txt 2>&1; done
  for b in 1 32; do RDX_PCONV=0 python $ROOT/tools/enc_only.py $b 10 >> $OUT/enc_only_rowmajor.txt 2>&1; done
fi
if [ $what = pmc ]; then
  pmc sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_WAVES
  pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
  pmc grbm GRBM_GUI_ACTIVE
  pmc tcp TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
fi
du -sh $OUT
