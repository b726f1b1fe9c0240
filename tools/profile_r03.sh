#!/bin/bash
# Round-3 profiling pass (run on the GPU box from the repo root): rocprofv3 kernel-trace summaries of the three phases and of the
# bench, PMC HBM-traffic passes of the decode kernels, and the per-shape encoder table. Text summaries only -> gpurun_out/prof/.
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
pmc() {  # name, counter, command...
  local name=$1 ctr=$2; shift 2
  rm -rf /tmp/pmc_${name}_$ctr
  (cd /tmp && rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmc_${name}_$ctr -o pmc --output-format rocpd -- "$@" > /tmp/pmc_${name}_$ctr.log 2>&1)
  local db=$(find /tmp/pmc_${name}_$ctr -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then python $ROOT/tools/pmc_summary.py $db > $OUT/pmc_${name}_$ctr.txt 2>&1; else tail -8 /tmp/pmc_${name}_$ctr.log > $OUT/pmc_${name}_$ctr.txt; fi
}
what=${1:-all}
if [ $what = all ] || [ $what = trace ]; then
  trace enc_b1 21 python $ROOT/tools/enc_only.py 1 20
  trace enc_b32 6 python $ROOT/tools/enc_only.py 32 5
  trace prefill_b1 11 python $ROOT/tools/prefill_only.py 1 160 10
  trace prefill_b32 4 python $ROOT/tools/prefill_only.py 32 160 3
  trace prefill_b32_fp8 4 python $ROOT/tools/prefill_only.py 32 160 3 fp8
  trace bench_b1 0 python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-b32 --no-fp8
  trace bench_b32 0 python $ROOT/bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline
  trace bench_b32_fp8 0 python $ROOT/bench.py --batch 32 --fp8 --steps 2 --warmup 1 --no-cpu-baseline
fi
if [ $what = all ] || [ $what = pmc ]; then
  for ctr in FETCH_SIZE WRITE_SIZE; do
    pmc b1 $ctr python $ROOT/bench.py --steps 1 --warmup 0 --new-tokens 8 --no-cpu-baseline --no-b32 --no-fp8 --no-graph
    pmc b32 $ctr python $ROOT/bench.py --batch 32 --steps 1 --warmup 0 --new-tokens 8 --no-cpu-baseline --no-graph
    pmc b32fp8 $ctr python $ROOT/bench.py --batch 32 --fp8 --steps 1 --warmup 0 --new-tokens 8 --no-cpu-baseline --no-graph
  done
fi
if [ $what = all ] || [ $what = shapes ]; then
  python $ROOT/tools/enc_kernels.py 32 > $OUT/enc_shapes_b32.txt 2>&1
  python $ROOT/tools/enc_kernels.py 1 > $OUT/enc_shapes_b1.txt 2>&1
fi
du -sh $OUT
if [ $what = prefill ]; then
  trace prefill_b32 4 python $ROOT/tools/prefill_only.py 32 160 3
  trace prefill_b32_fp8 4 python $ROOT/tools/prefill_only.py 32 160 3 fp8
  trace prefill_b1 11 python $ROOT/tools/prefill_only.py 1 160 10
fi
if [ $what = prefill_pmc ]; then
  for ctr in FETCH_SIZE TCC_HIT_sum TCC_MISS_sum; do
    pmc prefill $ctr python $ROOT/tools/prefill_only.py 1 160 2
  done
fi
if [ $what = mfma ]; then
  rm -rf /tmp/pmc_mfma
  (cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_mfma -o pmc --output-format rocpd -- python $ROOT/tools/gemm_bench.py > /tmp/pmc_mfma.log 2>&1)
  db=$(find /tmp/pmc_mfma -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then python $ROOT/tools/pmc_summary.py $db gemm_dma > $OUT/pmc_mfma.txt 2>&1; else tail -8 /tmp/pmc_mfma.log > $OUT/pmc_mfma.txt; fi
  tail -6 /tmp/pmc_mfma.log >> $OUT/pmc_mfma.txt
fi
