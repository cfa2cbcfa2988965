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


def test_no_packed_fp32_instruction_crosses_halves_of_its_own_destination(tmp_path):
    """Static scan of every kernel the library ships (hipcc -S with the build's own flags, CPU only).

    Round 4: a `v_pk_fma_f32 v[10:11], v[6:7], s[24:25], v[10:11] op_sel:[0,1,1]` -- the accumulator pair overwritten by an
    instruction that routes the HIGH half of that same pair to both results -- lost the low result's addend on whole 16-lane
    rows whenever a second process time-sliced the GPU, and never otherwise (LOG.md).  The cause was not established beyond
    that, so the shape itself is kept out of the product: no packed fp32 instruction whose destination pair overlaps a VGPR
    source with crossed halves (op_sel 1 or op_sel_hi 0 on that source), and no packed fp32 instruction at all in conv55.hip,
    the kernel it happened in (compiled with -fno-slp-vectorize)."""
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    from iterseg_amd import _build
    hipcc = _build._hipcc()
    srcs = [s for s in _build.SOURCES if os.path.exists(os.path.join(_build.CSRC, s))]

    def asm(s):
        out = str(tmp_path / s.replace(".hip", ".s"))
        flags = [f for f in _build.FLAGS if f not in ("-fPIC", "-Wall")] + _build.EXTRA.get(s, [])
        r = subprocess.run([hipcc] + flags + ["-S", "--cuda-device-only", "-o", out, os.path.join(_build.CSRC, s)],
                           capture_output=True, text=True, cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr[-2000:]
        return s, out

    with ThreadPoolExecutor(max_workers=6) as ex:
        files = list(ex.map(asm, srcs))
    packed = {}
    bad = []
    for s, path in files:
        n = 0
        with open(path) as f:
            for line in f:
                t = line.strip().split(";")[0]
                if not t.startswith(("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32")):
                    continue
                n += 1
                lo = re.search(r"op_sel:\[([\d,]+)\]", t)
                hi = re.search(r"op_sel_hi:\[([\d,]+)\]", t)
                lo = [int(x) for x in lo.group(1).split(",")] if lo else [0, 0, 0]
                hi = [int(x) for x in hi.group(1).split(",")] if hi else [1, 1, 1]
                ops = [o.strip() for o in t.split(None, 1)[1].split("op_sel")[0].split(",") if o.strip()]
                regs = [re.match(r"v\[(\d+):(\d+)\]", o) for o in ops]
                if not regs or not regs[0]:
                    continue
                dst = set(range(int(regs[0].group(1)), int(regs[0].group(2)) + 1))
                for i, m in enumerate(regs[1:4]):
                    crossed = (i < len(lo) and lo[i] == 1) or (i < len(hi) and hi[i] == 0)
                    if m and crossed and dst & set(range(int(m.group(1)), int(m.group(2)) + 1)):
                        bad.append((s, t))
        packed[s] = n
    assert packed.get("conv55.hip", 0) == 0, packed
    assert not bad, bad[:10]
