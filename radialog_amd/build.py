"""Build librdx.so (HIP C++, gfx950) in-tree with hipcc. No torch extension machinery: the library is a plain
C-ABI shared object loaded with ctypes (radialog_amd/_lib.py). The kernel-test / trace / microbenchmark hooks (include/rdx_hooks.h,
csrc/api_debug.hip) go into a second library, librdx_hooks.so, linked against librdx.so: the product library does not carry them."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "librdx.so")
OUT_HOOKS = os.path.join(HERE, "librdx_hooks.so")
SOURCES = ["gemm.hip", "xstat32.hip", "xs16.hip", "gemm_dma.hip", "gemm8.hip", "attn.hip", "flash.hip", "chain.hip", "elem.hip", "beam.hip", "conv1x1.hip", "wsgemm.hip", "stem.hip", "wstat.hip", "pconv.hip",
           "api.hip", "api_dispatch.hip", "api_encode.hip", "api_llama.hip", "api_comm.hip", "api_inspect.hip"]
HOOK_SOURCES = ["api_debug.hip"]
HEADERS = ["rdx_common.h", "rdx_kernels.h", "rdx_ctx.h", "skinny_body.h", "attn_body.h", "handoff.h", os.path.join("..", "..", "include", "rdx.h"),
           os.path.join("..", "..", "include", "rdx_hooks.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def source_hash() -> str:
    """16 hex digits over every kernel / ABI source: the tree a profile was taken on (profiles/rNN_pmc.json carries it; bench.py flags a
    replayed PMC figure whose tree differs from the one that is running)."""
    import hashlib
    h = hashlib.sha256()
    # kernels, launchers and the ABI implementation; the two public headers under include/ (declarations and prose) are not part of it
    for s in sorted(SOURCES + HOOK_SOURCES) + sorted(h for h in HEADERS if not h.startswith("..")):
        with open(os.path.join(CSRC, s), "rb") as f:
            h.update(s.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def _stale():
    if not os.path.exists(OUT) or not os.path.exists(OUT_HOOKS):
        return True
    t = min(os.path.getmtime(OUT), os.path.getmtime(OUT_HOOKS))
    deps = [os.path.join(CSRC, s) for s in SOURCES + HOOK_SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, hook_objs = [], []
    procs = []
    for s in SOURCES + HOOK_SOURCES:
        o = os.path.join(CSRC, s.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
        (hook_objs if s in HOOK_SOURCES else objs).append(o)
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    # the hooks: their own library, resolved against librdx.so (found next to it at run time)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT_HOOKS] + hook_objs + ["-L" + HERE, "-lrdx", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
