"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and
exports every symbol include/shk.h declares.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "shk.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(shk_[a-z_0-9]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from sharkmer_amd.engine import lib_path
    return lib_path()


def test_header_declares_what_python_binds(built):
    from sharkmer_amd.engine import ABI_SYMBOLS
    assert sorted(ABI_SYMBOLS) == _header_functions()


def test_library_exports_every_declared_symbol(built):
    L = ctypes.CDLL(built)
    for name in _header_functions():
        assert hasattr(L, name), f"{name} declared in include/shk.h but not exported"
    L.shk_abi_version.restype = ctypes.c_int
    assert L.shk_abi_version() == 2


def test_code_object_is_gfx950(built):
    blob = open(built, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_90"):
        assert other not in blob


def test_no_cpu_fallback_without_gpu(built):
    """The product path must fail loudly when no GPU is present."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import sharkmer_amd as sa
    with pytest.raises(sa.ShkError) as e:
        sa.KmerEngine(21, 1)
    assert e.value.code == -9
    assert "no CPU fallback" in str(e.value)


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under sharkmer_amd/ may reference it."""
    pkg = os.path.join(ROOT, "sharkmer_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".c")):
                txt = open(os.path.join(dp, fn), errors="replace").read()
                assert "liborc" not in txt and "shk_oracle" not in txt and "import oracle" not in txt \
                    and "from oracle" not in txt, f"{fn} references the oracle"


def test_synth_is_deterministic_and_well_formed():
    import numpy as np
    import sharkmer_amd as sa
    spec = sa.SynthSpec(genome_len=5000, sub_per_64k=500, n_per_64k=100)
    b1, o1 = sa.synth_reads(spec, 10, 50)
    b2, _ = sa.synth_reads(spec, 0, 60)
    assert np.array_equal(b1, b2[10 * 150:])
    assert set(np.unique(b1)) <= set(b"ACGTN")
    assert o1[-1] == len(b1) == 50 * 150
    clean, _ = sa.synth_reads(sa.SynthSpec(genome_len=5000), 0, 60)
    assert b"N" not in clean.tobytes()
    frac = (clean != b2).mean()
    assert 0.001 < frac < 0.03
