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
           "api.hip", "api_dispatch.hip", "api_encode.hip", "api_llama.hip", "api_comm.hip", "api_inspect.hip", "api_transform.hip"]
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


HASH_FILE = os.path.join(CSRC, ".build_hash")      # the source hash the in-tree objects / libraries were built from (git-ignored, travels with gpurun)


def built_hash():
    try:
        with open(HASH_FILE) as f:
            return f.read().strip()
    except OSError:
        return None


def _stale():
    """The libraries are stale when they are missing or when the hash compiled into them (rdx_build_hash(), mirrored in csrc/.build_hash)
    is not the hash of the sources in the tree -- content, not mtimes (round 6: a checkout or a copy can leave an old .so newer than the
    sources it no longer matches)."""
    if not os.path.exists(OUT) or not os.path.exists(OUT_HOOKS):
        return True
    return built_hash() != source_hash()


def _obj_stale(src, obj, hdr_mtime):
    return (not os.path.exists(obj)) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_mtime)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    h = source_hash()
    objs, hook_objs = [], []
    procs = []
    hdr_mtime = max(os.path.getmtime(os.path.join(CSRC, x)) for x in HEADERS + [os.path.abspath(__file__)] if os.path.exists(os.path.join(CSRC, x)))
    for s in SOURCES + HOOK_SOURCES:
        src, o = os.path.join(CSRC, s), os.path.join(CSRC, s.replace(".hip", ".o"))
        (hook_objs if s in HOOK_SOURCES else objs).append(o)
        # api.hip carries the hash (rdx_build_hash): recompiled on every rebuild; the other units only when they or a header changed
        if not force and s != "api.hip" and not _obj_stale(src, o, hdr_mtime):
            continue
        cmd = [hipcc] + FLAGS + ([f'-DRDX_BUILD_HASH="{h}"'] if s == "api.hip" else []) + ["-c", src, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
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
    with open(HASH_FILE, "w") as f:
        f.write(h + "\n")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
