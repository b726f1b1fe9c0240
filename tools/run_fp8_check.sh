mkdir -p gpurun_out/fp8chk
python -m pytest tests -m gpu -q -x -k "fp8" 2>&1 | tail -4 > gpurun_out/fp8chk/tests.log
python bench.py > gpurun_out/fp8chk/bench.json 2> gpurun_out/fp8chk/bench.err; echo "rc $?" >> gpurun_out/fp8chk/bench.err
