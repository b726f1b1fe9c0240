export PYTHONPATH=$PWD
bash tools/profile_r03.sh shapes > /dev/null 2>&1
bash tools/profile_r03.sh mfma > /dev/null 2>&1
for cfg in "1 160" "2 160" "4 160" "8 160" "1 600" "2 600"; do
  set -- $cfg
  for fm in 512 64; do
    echo -n "FLASH_MIN=$fm  "; RDX_FLASH_MIN=$fm python tools/prefill_only.py $1 $2 5 2>&1 | tail -1
  done
done > gpurun_out/flash_min.txt 2>&1
cat gpurun_out/flash_min.txt
