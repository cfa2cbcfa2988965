"""CPU-side checks of the C ABI: the in-tree HIP library loads, exports every symbol that
include/itersegx.h declares, refuses to run without a GPU, and its chunk table equals the
host mirror.  No compute calls (there is no GPU here)."""
import os
import re

import numpy as np
import pytest

from iterseg_amd import _lib, chunking

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    with open(os.path.join(ROOT, "include", "itersegx.h")) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(isx_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        from iterseg_amd import _build
        _build.build()
    return _lib.load()


def test_header_symbols_are_exported_and_bound(lib):
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in itersegx.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared          # the ctypes table covers the whole header
    assert lib.isx_version() == 100


def test_chunk_table_matches_host_mirror(lib):
    for shape, chunk, margin in [((512, 512, 512), (10, 256, 256), (1, 64, 64)),
                                 ((11, 257, 300), (10, 256, 256), (1, 64, 64)),
                                 ((9, 40, 56), (4, 32, 32), (1, 8, 8)), ((4, 32, 32), (4, 32, 32), (1, 8, 8))]:
        assert np.array_equal(_lib.make_chunks_table(shape, chunk, margin),
                              chunking.chunk_table(shape, chunk, margin))
    with pytest.raises(ValueError):
        _lib.make_chunks_table((8, 256, 256), (10, 256, 256), (1, 64, 64))


def test_no_cpu_fallback_without_device(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.IsxError) as e:
        _lib.Context(0)
    assert e.value.code == _lib.ISX_ERR_NO_DEVICE


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under iterseg_amd/ may reference it."""
    pkg = os.path.join(ROOT, "iterseg_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                with open(os.path.join(dirpath, fn)) as f:
                    txt = f.read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), fn
                assert "ws_oracle" not in txt and "oracle/" not in txt, fn
