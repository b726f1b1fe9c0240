#!/bin/bash
# Round-5 profiling passes (run on the GPU box from the repo root); text summaries only -> gpurun_out/prof/.
#   bash tools/profile_r05.sh decode     kernel tables of a 256-token decode at batch 16 (xs16 on / off) and batch 32
#   bash tools/profile_r05.sh bench      kernel table of the default bench command
#   bash tools/profile_r05.sh final      kernel tables of the final tree: batch 12 / 16 decode, fp8 x fp8 row blocks at 64 / 128 rows
#   bash tools/profile_r05.sh pmc        FETCH_SIZE / WRITE_SIZE passes of bench.py at batch 1 / 32 / 32 fp8 (one counter per pass; tools/pmc_to_json.py r05)
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
  grep -E "^\{|ms/img|prefill B|^decode B" /tmp/rp_$name.log | tail -1 | cut -c1-600 >> $OUT/$name.md
}
pmc1() {  # run tag, counter, bench args...
  local tag=$1 ctr=$2; shift 2
  rm -rf /tmp/pmc_${tag}_$ctr
  (cd /tmp && rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmc_${tag}_$ctr -o pmc --output-format rocpd -- python $ROOT/bench.py "$@" > /tmp/pmc_${tag}_$ctr.log 2>&1)
  local db=$(find /tmp/pmc_${tag}_$ctr -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then python $ROOT/tools/pmc_summary.py $db > $OUT/pmc_${tag}_$ctr.txt 2>&1; else tail -8 /tmp/pmc_${tag}_$ctr.log > $OUT/pmc_${tag}_$ctr.txt; fi
}
what=${1:-decode}
if [ $what = decode ]; then
  trace dec_b16_xs16 0 python $ROOT/tools/decode_only.py 16 256 1
  trace dec_b16_old 0 python $ROOT/tools/decode_only.py 16 256 0
  trace dec_b32 0 python $ROOT/tools/decode_only.py 32 256 1
fi
if [ $what = bench ]; then
  trace bench_default 0 python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline
fi
if [ $what = pmc ]; then
  python -c "from radialog_amd import build; print(build.source_hash())" > $OUT/pmc_tree.txt      # the kernel sources these passes ran on (tools/pmc_to_json.py stamps it)
  for ctr in FETCH_SIZE WRITE_SIZE; do
    # short runs: a PMC pass serialises every dispatch (a 256-token decode under counters did not finish in 30 minutes). The batch-32 passes use a
    # 280-token prompt + 8 tokens so that the decode attention is profiled at the MEAN context of the benchmark's decode (160 .. 415 -> 288)
    pmc1 b1 $ctr --steps 1 --warmup 0 --new-tokens 8 --no-cpu-baseline --no-b32 --no-fp8 --no-f16 --no-enc256 --no-b64 --no-graph
    pmc1 b32 $ctr --batch 32 --steps 1 --warmup 0 --prompt-len 280 --new-tokens 8 --no-cpu-baseline --no-graph
    pmc1 b32fp8 $ctr --batch 32 --fp8 --steps 1 --warmup 0 --prompt-len 280 --new-tokens 8 --no-cpu-baseline --no-graph
  done
fi

if [ $what = prefill ]; then
  trace prefill_b1 11 python $ROOT/tools/prefill_only.py 1 160 10
  RDX_PBLK=0 trace prefill_b1_wstat 11 python $ROOT/tools/prefill_only.py 1 160 10
fi
du -sh $OUT
if [ $what = prefill64 ]; then
  trace prefill_b1_t64 11 python $ROOT/tools/prefill_only.py 1 64 10
fi
if [ $what = dec64 ]; then
  trace dec_b64 0 python $ROOT/tools/decode_only.py 64 256 1
fi
if [ $what = final ]; then      # the final round-5 tree: batch 12 / 16 on xs16 with the 8-wave attention, fp8 x fp8 row blocks at 64 and 128 rows
  trace dec_b12_final 0 python $ROOT/tools/decode_only.py 12 256 1
  trace dec_b16_final 0 python $ROOT/tools/decode_only.py 16 256 1
  trace dec_fp8_b64 0 python $ROOT/tools/decode_only.py 64 256 1 1
  trace dec_fp8_b128 0 python $ROOT/tools/decode_only.py 128 256 1 1
fi
if [ $what = pmc_extra ]; then   # FETCH_SIZE of the round-5 additions: batch 12 (xs16 + the 8-wave attention), fp8 x fp8 row blocks at 64 rows (mean context 288)
  pmc1 b12 FETCH_SIZE --batch 12 --steps 1 --warmup 0 --prompt-len 280 --new-tokens 8 --no-cpu-baseline --no-graph
  pmc1 b64fp8 FETCH_SIZE --batch 64 --fp8 --steps 1 --warmup 0 --prompt-len 280 --new-tokens 8 --no-cpu-baseline --no-graph
fi
