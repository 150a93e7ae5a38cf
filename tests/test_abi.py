"""The C-ABI library loads and exports every symbol include/morna_hip.h declares.

No GPU needed: nothing here launches a kernel.
"""
import os
import re

import pytest

from conftest import ROOT


def _declared():
    with open(os.path.join(ROOT, "include", "morna_hip.h")) as fh:
        src = fh.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(morna_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from morna_amd import _lib, build
    build.build()                      # hipcc cross-compiles gfx950 without a GPU
    L = _lib.lib()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "libmorna_hip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "ctypes binding lacks %s" % n
    assert sorted(_lib.SIGNATURES) == names


def test_host_hash_mirror_matches_golden():
    import json
    import ctypes as C
    from morna_amd import _lib
    with open(os.path.join(ROOT, "tests", "golden", "murmur3_vectors.json")) as fh:
        g = json.load(fh)
    L = _lib.lib()
    for k, h in g["vectors"]:
        b = k.encode("ascii")
        buf = C.create_string_buffer(b, len(b) + 1)
        assert L.morna_hash32(C.cast(buf, C.c_void_p), len(b)) == h, k


def test_no_gpu_fails_loudly():
    """Without a device the product must raise, never fall back to the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from morna_amd.annoy import AnnoyIndex
    with pytest.raises(RuntimeError):
        AnnoyIndex(16)


def test_product_does_not_import_oracle():
    """oracle/ is test infrastructure: nothing under morna_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "morna_amd")):
        if os.path.basename(dirpath) == "build":
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                with open(os.path.join(dirpath, f)) as fh:
                    src = fh.read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "oracle/_build" not in src and "libmorna_oracle" not in src, f
