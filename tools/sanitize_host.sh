#!/bin/bash
# Host side of librdx (the C ABI: argument checks, workspace management, beam-search bookkeeping, graph capture, RCCL binding) under
# UndefinedBehaviorSanitizer ("sanitizer build of the shim"). Device code is unchanged (the sanitizers do not apply to gfx950
# without xnack+). AddressSanitizer was tried first: ROCm's ASan runtime interposes hsa_amd_memory_pool_allocate and aborts in the HIP
# runtime's first device allocation on a non-xnack box ("out of memory: allocator is trying to allocate 0x400000 bytes"), so it cannot
# run here. Usage:
#   bash tools/sanitize_host.sh build          # here or on the GPU box: radialog_amd/librdx_ubsan.so (in-tree, git-ignored, travels with gpurun)
#   bash tools/sanitize_host.sh run [pytest args]   # on a GPU box: pytest through the sanitized library; reports -> gpurun_out/ubsan_report.*
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/radialog_amd/csrc
OUT=$ROOT/radialog_amd/librdx_ubsan.so
if [ "${1:-build}" = build ]; then
  mkdir -p /tmp/rdx_asan
  SRCS=$(python - <<PY
import sys; sys.path.insert(0, "$ROOT")
from radialog_amd import build
print(" ".join(build.SOURCES + build.HOOK_SOURCES))     # one library: the hooks linked in (_lib.load_hooks takes them from RDX_LIB_PATH when it exports them)
PY
)
  HASH=$(python -c "import sys; sys.path.insert(0, '$ROOT'); from radialog_amd import build; print(build.source_hash())")
  pids=""
  for s in $SRCS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -DRDX_BUILD_HASH="\"$HASH\"" -O3 -g -std=c++17 -fPIC -fsanitize=undefined,bounds,integer-divide-by-zero -fno-sanitize=vptr -fno-omit-frame-pointer \
      -Wno-unused-value -Wno-option-ignored -c $CSRC/$s -o /tmp/rdx_asan/${s%.hip}.o & pids="$pids $!"
  done
  rc=0; for p in $pids; do wait $p || rc=1; done
  [ $rc = 0 ] || { echo "compile failed"; exit 1; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=undefined -shared-libsan -o $OUT /tmp/rdx_asan/*.o || exit 1
  ls -la $OUT
else
  shift
  mkdir -p $ROOT/gpurun_out
  cd $ROOT
    rm -f $ROOT/gpurun_out/ubsan_report*
  RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.ubsan_standalone-x86_64.so)       # the library is dlopen'ed by python: preload the runtime
  LD_PRELOAD=$RT UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0:log_path=$ROOT/gpurun_out/ubsan_report RDX_LIB_PATH=$OUT \
    python -m pytest "$@" > $ROOT/gpurun_out/ubsan.log 2>&1
  tail -5 $ROOT/gpurun_out/ubsan.log
  echo "UBSan reports: $(cat $ROOT/gpurun_out/ubsan_report* 2>/dev/null | grep -c 'runtime error')"
  cat $ROOT/gpurun_out/ubsan_report* 2>/dev/null | grep 'runtime error' | sort | uniq -c | sort -rn | head -20
fi
