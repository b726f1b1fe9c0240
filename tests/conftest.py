import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
# the kernel-test hooks (include/rdx_hooks.h, librdx_hooks.so) are not part of the product library: the tests ask for them explicitly, before
# radialog_amd._lib is imported
os.environ.setdefault("RDX_DEBUG_HOOKS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes of CPU oracle time (full-depth Vicuna-7B legs); still part of -m gpu")
    # The CPU oracle is most of the GPU suite's wall clock, and torch's default intra-op pool on a many-core host (128 threads on the
    # 256-cpu GPU boxes) is its slowest setting: an 11008 x 4096 fp16 linear over 160 rows takes 97 ms with 128 threads, 12 ms with 32;
    # over 5120 rows 2.5 s against 1.0-1.3 s; bf16 over 5120 rows 267 ms against 80 ms (tools/oracle_threads.py, measured on the box).
    # Thread count does not enter the oracle's arithmetic contract (the golden vectors were produced on 8 cores).
    import torch
    if (os.cpu_count() or 1) >= 64:
        torch.set_num_threads(32)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")


def pytest_report_header(config):
    """Which binary is under test (round 6): the source hash compiled into librdx.so and the tree's -- _lib.load() refuses a mismatch."""
    try:
        from radialog_amd import _lib, build
        return f"librdx: build hash {_lib.build_hash()} (tree {build.source_hash()}), {_lib.LIB_PATH}"
    except Exception as e:          # no library yet: the tests that need it fail loudly themselves
        return f"librdx: not loaded ({type(e).__name__}: {e})"
