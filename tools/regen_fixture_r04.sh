#!/bin/bash
set -u
ROOT=$PWD
export PYTHONPATH=$ROOT TMPDIR=/tmp
OUT=$ROOT/gpurun_out/r04fix
mkdir -p $OUT
python tools/make_bench_fixture.py > $OUT/fixture.json 2> $OUT/fixture.err
cp tests/golden/bench_tokens.json $OUT/bench_tokens.json
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?" >> $OUT/bench.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_torchrun_np1.json 2> $OUT/bench_torchrun_np1.err; echo "rc $?" >> $OUT/bench_torchrun_np1.err
