import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (oracle/liborc.so), built on demand with gcc."""
    from oracle import oracle as o
    o.build()
    return o


# Test hooks of the engine and the front-end that a developer's shell may still carry from an experiment: a test that
# wants one sets it itself (monkeypatch), every other test must see the product's own behaviour — the launch-count
# asserts of test_gpu_parity.py are about THAT (a suite once failed `820 < 28` under a stray SHK_FLUSH_GROUP_PAGES).
_HOOKS = ("SHK_FLUSH_GROUP_PAGES", "SHK_DEFER_BUDGET", "SHK_LEVEL1_LOG", "SHK_TWO_LEVEL_MIN_PAGES", "SHK_NO_FRESH", "SHK_DEFER",
          "SHK_XL", "SHK_REC32", "SHK_SCATTER32_LDS", "SHK_ALL_LANES", "SHK_PART_G", "SHK_DIST_DENSE", "SHK_DIST_CAP",
          "SHK_DIST_MAX_MESSAGE", "SHK_FASTQ_WINDOW_KB", "SHK_FASTQ_THREADS", "SHK_FASTQ_COPY_THREADS", "SHK_NO_AVX2",
          "SHK_RUN_ASCII", "SHK_GROUP_ROUND_KB", "SHK_TRACE", "SHK_FASTQ_DEBUG", "SHK_PGZ_MIN_KB", "SHK_PGZ_CHUNK_KB",
          "SHK_PGZ_THREADS", "SHK_WIDE_WINDOW", "SHK_BENCH_EXTRAS", "SHK_HOST_PACK", "SHK_ACC_MAX_MRECORDS", "SHK_SLICE_KB", "SHK_NO_MEM_CACHE",
          "SHK_FUSED_HIST", "SHK_INSERT_PAGED", "SHK_INSERT_PAGED_MIN", "SHK_XL64", "SHK_XCHG_LATE_SETTLE", "SHK_DIST_ONE_CALL_SCATTER", "SHK_TEST_GROW_NOMEM", "SHK_WINDOW_KEEP_GIB", "SHK_WINDOW_TABLES", "SHK_SCATTER64", "SHK_S64_INTERLEAVE", "SHK_CTL_OUT_KERNEL", "SHK_EXP_IGNORE_FINALIZE")


@pytest.fixture(autouse=True)
def _no_stray_hooks(monkeypatch):
    if os.environ.get("SHK_KEEP_HOOKS"):   # (a deliberate forced-hook sweep of the whole suite)
        return
    for name in _HOOKS:
        monkeypatch.delenv(name, raising=False)
