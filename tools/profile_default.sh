export PYTHONPATH=$PWD TMPDIR=/tmp
ROOT=$PWD
mkdir -p gpurun_out/prof
rm -rf /tmp/rp_default
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/rp_default -o bench_default --output-format rocpd -- python $ROOT/bench.py > /tmp/rp_default.log 2>&1)
db=$(find /tmp/rp_default -name "*.db" | head -1)
python tools/prof_summary.py $db gpurun_out/prof/bench_default.md 0 > /dev/null
grep -E "^\{" /tmp/rp_default.log | tail -1 | cut -c1-700 >> gpurun_out/prof/bench_default.md
head -8 gpurun_out/prof/bench_default.md | cut -c1-200
