#!/bin/bash
# PMC pass over the batched prefill GEMMs (bf16 gemm_dma256_k, fp8 gemm8_256_k): matrix-pipe busy cycles -> profiles/r04_pmc_prefill_mfma.md. --kernel-trace only.
export PYTHONPATH=$PWD TMPDIR=/tmp; mkdir -p gpurun_out/mf
for mode in bf16 fp8; do
  extra=""; [ $mode = fp8 ] && extra="fp8"
  rm -rf /tmp/pm_$mode
  (cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pm_$mode -o pmc --output-format rocpd -- python $GRAFT_REPO_ROOT/tools/prefill_only.py 32 160 2 $extra > /tmp/pm_$mode.log 2>&1)
  python tools/pmc_summary.py $(find /tmp/pm_$mode -name "*.db" | head -1) gemm > gpurun_out/mf/pmc_$mode.txt 2>&1
done
