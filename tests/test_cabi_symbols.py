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


def test_generated_api_surface_of_the_multicore_reference_library():
    # what a harness for `futhark multicore --library` calls besides the entries: the worker-pool
    # knob and the serialisers of the opaque sr / sc types
    syms = declared_symbols()
    assert "futhark_context_config_set_num_threads" in syms
    for kind in ("csr", "csc"):
        for t in ("f64", "f32", "i32"):
            assert "futhark_store_opaque_%s_%s" % (kind, t) in syms
            assert "futhark_restore_opaque_%s_%s" % (kind, t) in syms
    from sparse_amd.compressed import library_path
    lib = ctypes.CDLL(library_path())
    lib.futhark_context_config_new.restype = ctypes.c_void_p
    cfg = ctypes.c_void_p(lib.futhark_context_config_new())
    lib.futhark_context_config_set_num_threads(cfg, 16)      # host-only call: accepted without a GPU
    lib.futhark_context_config_free(cfg)


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
