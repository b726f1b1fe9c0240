"""ctypes binding of librdx.so (the C ABI declared in include/rdx.h).

There is NO CPU fallback: if the HIP library is missing or cannot be loaded, importing callers get an
`RdxLibraryError` -- the product path never routes around the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- must be imported BEFORE librdx.so is loaded so both share torch's HIP runtime

HERE = os.path.dirname(os.path.abspath(__file__))
# RDX_LIB_PATH: a differently-built librdx of the same sources (tools/sanitize_host.sh: the host side under ASan / UBSan); never a fallback
LIB_PATH = os.environ.get("RDX_LIB_PATH") or os.path.join(HERE, "librdx.so")
# kernel-test / trace / microbenchmark hooks (include/rdx_hooks.h): a separate library, loaded only under RDX_DEBUG_HOOKS=1 (tests, tools)
HOOKS_PATH = os.environ.get("RDX_HOOKS_PATH") or os.path.join(HERE, "librdx_hooks.so")

RDX_DTYPE_F16, RDX_DTYPE_BF16 = 0, 1
RDX_W_GEMM, RDX_W_TENSOR, RDX_W_F32, RDX_W_GEMM_FP8 = 0, 1, 2, 3
RDX_SRC_F32, RDX_SRC_F16, RDX_SRC_BF16 = 0, 1, 2


class RdxLibraryError(RuntimeError):
    pass


class RdxError(RuntimeError):
    pass


class RdxConfig(C.Structure):
    _fields_ = [
        ("dtype", C.c_int),
        ("vocab", C.c_int), ("hidden", C.c_int), ("inter", C.c_int), ("layers", C.c_int), ("heads", C.c_int),
        ("max_pos", C.c_int),
        ("rms_eps", C.c_float),
        ("lora_r", C.c_int), ("lora_scale", C.c_float),
        ("qformer_dim", C.c_int),
        ("q_hidden", C.c_int), ("q_layers", C.c_int), ("q_heads", C.c_int), ("q_inter", C.c_int),
        ("q_enc_width", C.c_int), ("q_nquery", C.c_int), ("q_cross_freq", C.c_int),
        ("q_ln_eps", C.c_float),
        ("v_img", C.c_int), ("v_stem", C.c_int), ("v_planes", C.c_int * 4), ("v_blocks", C.c_int * 4),
        ("v_b2v", C.c_int), ("v_proj", C.c_int),
        ("v_ln_eps", C.c_float),
        ("max_batch", C.c_int), ("max_len", C.c_int),
        ("enable_vision", C.c_int), ("enable_llama", C.c_int),
        ("enable_cls", C.c_int), ("cls_hidden", C.c_int), ("cls_classes", C.c_int), ("cls_pool", C.c_int),
    ]


# name -> (restype, argtypes): every symbol include/rdx.h declares
_P = C.c_void_p
SYMBOLS = {
    "rdx_create": (C.c_int, [C.POINTER(_P), C.c_int, C.POINTER(RdxConfig)]),
    "rdx_destroy": (None, [_P]),
    "rdx_last_error": (C.c_char_p, [_P]),
    "rdx_build_hash": (C.c_char_p, []),
    "rdx_sync": (C.c_int, [_P]),
    "rdx_stream": (_P, [_P]),
    "rdx_set_weight": (C.c_int, [_P, C.c_char_p, _P, C.c_int64, C.c_int64, C.c_int]),
    "rdx_set_weight_typed": (C.c_int, [_P, C.c_char_p, _P, C.c_int, C.c_int64, C.c_int64, C.c_int]),
    "rdx_finalize_weights": (C.c_int, [_P]),
    "rdx_transform_image": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "rdx_encode_image": (C.c_int, [_P, _P, C.c_int, _P, _P]),
    "rdx_encode_image2": (C.c_int, [_P, _P, _P, C.c_int, _P, _P]),
    "rdx_classify_findings": (C.c_int, [_P, _P, C.c_int, _P]),
    "rdx_generate": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, _P,
                               C.POINTER(C.c_int), C.c_int]),
    "rdx_prefill": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rdx_decode_step": (C.c_int, [_P, _P]),
    "rdx_decode_step_ids": (C.c_int, [_P, _P, _P]),
    "rdx_prefill_append": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rdx_generate_append": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P,
                                      C.POINTER(C.c_int), C.c_int]),
    "rdx_beam_search": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _P, _P, _P, _P,
                                  C.POINTER(C.c_int)]),
    "rdx_comm_unique_id": (C.c_int, [_P]),
    "rdx_comm_init": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "rdx_allgather_tokens": (C.c_int, [_P, _P, _P, C.c_int, C.c_int]),
    "rdx_comm_world": (C.c_int, [_P]),
    "rdx_kv_read": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "rdx_hidden_read": (C.c_int, [_P, _P]),
    "rdx_time": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "rdx_set_option": (C.c_int, [_P, C.c_char_p, C.c_int]),
}

# include/rdx_hooks.h: every symbol of librdx_hooks.so
HOOK_SYMBOLS = {
    "rdx_attn_trace": (C.c_int, [_P, C.c_int, _P]),
    "rdx_quant_test": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_float, _P, _P]),
    "rdx_gemv_trace": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int]),
    "rdx_gemm_test": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_float, C.c_int]),
    "rdx_kernel_bench": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), _P, C.c_int]),
    "rdx_conv_test": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "rdx_l2_bench": (C.c_int, [_P, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "rdx_logits_test": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int]),
}

_lib = None
_hooks = None


def hooks_enabled() -> bool:
    return os.environ.get("RDX_DEBUG_HOOKS", "0") not in ("", "0")


def _missing_hook(name):
    def raiser(*_a, **_k):
        raise RdxLibraryError(f"{name} is a kernel-test hook of librdx_hooks.so (include/rdx_hooks.h), not part of the product library: "
                              "set RDX_DEBUG_HOOKS=1 before radialog_amd is loaded (tests/conftest.py and the tools/ scripts do)")
    return raiser


def load_hooks(lib):
    """Bind the hooks of include/rdx_hooks.h onto `lib` (the ctypes handle the engine calls through). Under RDX_DEBUG_HOOKS=1 they come from
    librdx_hooks.so (or from an RDX_LIB_PATH build that links them in: tools/sanitize_host.sh); otherwise every hook raises."""
    global _hooks
    if not hooks_enabled():
        for name in HOOK_SYMBOLS:
            setattr(lib, name, _missing_hook(name))
        return
    src = lib
    if not hasattr(lib, next(iter(HOOK_SYMBOLS))):
        if os.environ.get("RDX_LIB_PATH") and os.path.realpath(LIB_PATH) != os.path.realpath(os.path.join(HERE, "librdx.so")):
            # librdx_hooks.so NEEDs "librdx.so" and finds the in-tree one through $ORIGIN: next to an RDX_LIB_PATH build that does not link the
            # hooks in, that would be a SECOND copy of the library with its own contexts
            raise RdxLibraryError(f"RDX_DEBUG_HOOKS is set, but RDX_LIB_PATH={LIB_PATH} does not link the hooks in and {HOOKS_PATH} would load a "
                                  "second librdx.so next to it: build the alternative library with api_debug.hip (tools/sanitize_host.sh does)")
        if not os.path.exists(HOOKS_PATH):
            raise RdxLibraryError(f"RDX_DEBUG_HOOKS is set but {HOOKS_PATH} is missing: build it with `python -m radialog_amd.build`")
        try:
            src = C.CDLL(HOOKS_PATH, mode=C.RTLD_GLOBAL)
        except OSError as e:
            raise RdxLibraryError(f"cannot load {HOOKS_PATH}: {e}") from e
        _hooks = src
    for name, (res, args) in HOOK_SYMBOLS.items():
        try:
            fn = getattr(src, name)
        except AttributeError as e:
            raise RdxLibraryError(f"{HOOKS_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
        if src is not lib:
            setattr(lib, name, fn)


def load():
    """Load librdx.so and bind every C-ABI symbol; raises RdxLibraryError if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RdxLibraryError(
            f"{LIB_PATH} not found: build it with `python -m radialog_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the RaDialog hot path.")
    try:
        # RTLD_GLOBAL only when the hooks will be loaded: librdx_hooks.so resolves the context / launcher symbols against this handle. The
        # product configuration keeps the library's C++ helpers (fail, gargs, run_gemm, ...) out of the process's global namespace.
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL if hooks_enabled() else C.RTLD_LOCAL)
    except OSError as e:
        raise RdxLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RdxLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    check_build_hash(lib)
    load_hooks(lib)
    _lib = lib
    return lib


def build_hash() -> str:
    """The source hash compiled into the loaded library (rdx_build_hash)."""
    return load().rdx_build_hash().decode()


def check_build_hash(lib):
    """Binary <-> source correspondence (round 6): the .so files are git-ignored and travel to the GPU box as built, so the library says which
    sources it was compiled from and the loader compares that with the sources lying next to it. A mismatch is an error (rebuild:
    `python -m radialog_amd.build`); RDX_ALLOW_STALE_LIB=1 turns it into a warning for bisecting with an old binary."""
    from . import build as _b
    try:
        want = _b.source_hash()
    except OSError:
        return                      # a deployment without csrc/ next to the library: nothing to compare with
    got = lib.rdx_build_hash().decode()
    if got != want:
        msg = (f"{LIB_PATH} was built from sources with hash {got}, the tree's radialog_amd/csrc hashes to {want}: the library is stale -- "
               "rebuild it with `python -m radialog_amd.build`")
        if os.environ.get("RDX_ALLOW_STALE_LIB", "0") not in ("", "0"):
            import warnings
            warnings.warn(msg)
        else:
            raise RdxLibraryError(msg)


def check(ctx, rc: int, what: str):
    if rc != 0:
        msg = load().rdx_last_error(ctx)
        raise RdxError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")
