"""The C-ABI library loads and exports every symbol include/sparse_amd.h declares
(no compute calls: this runs without a GPU)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sparse_amd.h")


def declared_symbols():
    txt = subprocess.check_output(["gcc", "-E", "-P", HEADER]).decode()
    names = set(re.findall(r"\b((?:futhark|sparse_amd)_\w+)\s*\(", txt))
    assert len(names) > 150, len(names)
    return sorted(names)


def test_header_compiles_as_c():
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", HEADER])


def test_library_exports_every_declared_symbol():
    from sparse_amd.compressed import library_path
    path = library_path()
    assert os.path.exists(path), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(path)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_reference_entry_names_present():
    # the 12 entries of compressed_test.fut under their generated names
    want = ["test_csr_eye", "test_csr_sparse", "test_csr_smvm", "test_csr_nnz", "test_csr_coo", "test_csc_eye",
            "test_csc_sparse", "test_csc_nnz", "test_csc_coo", "test_csr_transpose", "test_csc_transpose",
            "test_smsmm"]
    syms = declared_symbols()
    for w in want:
        assert "futhark_entry_" + w in syms


def test_product_never_touches_the_oracle():
    # the oracle is test infrastructure: nothing under sparse_amd/ may import, link or load it
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sparse_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"liboracle|from\s+oracle|import\s+oracle|#include.*oracle|oracle\.oracle", txt):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_missing_library_fails_loudly(monkeypatch):
    import sparse_amd.compressed as c
    monkeypatch.setattr(c, "_lib", None)
    monkeypatch.setattr(c, "_LIBPATH", "/nonexistent/libsparse_amd.so")
    with pytest.raises(c.LibraryMissing):
        c.load_library()
