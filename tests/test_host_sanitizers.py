"""The library's host-side parsers under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY section 5,
"Race detection / sanitizers"): sparse_amd/csrc/mtx.hpp (parallel Matrix Market parser over mmap) and
sparse_amd/csrc/store_format.hpp (what futhark_restore_opaque_* checks on caller bytes), compiled for the CPU
by `make -C sparse_amd/csrc asan` into sparse_amd/lib/hostcheck_asan (hostcheck.cpp; -fno-sanitize-recover, so any
finding kills the process with a report on stderr).  No GPU involved: GPU sanitizers are not available on
this pool, and these two are the places where bytes a caller controls are parsed on the host.

Every case must end in exit status 0 with either "ok ..." or a clean "error: ..." line and nothing from a
sanitizer on stderr."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "sparse_amd", "lib", "hostcheck_asan")


@pytest.fixture(scope="module")
def exe():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "sparse_amd", "csrc"), "asan"])
    assert os.path.exists(EXE)
    return EXE


def run(exe, *args):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe] + [str(a) for a in args], capture_output=True, timeout=120, env=env)
    err = p.stderr.decode(errors="replace")
    assert "Sanitizer" not in err and "runtime error" not in err, err
    assert p.returncode == 0, (p.returncode, err)
    return p.stdout.decode().strip()


GOOD = (b"%%MatrixMarket matrix coordinate real general\n% c\n3 4 5\n1 1 2.5\n3 4 -1e-3\n2 2 +7\n3 1 1E2\n1 1 0.5\n")

MALFORMED = {
    "empty": b"",
    "banner_only": b"%%MatrixMarket matrix coordinate real general",
    "no_newline_at_all": b"%%MatrixMarket matrix coordinate real general 3 3 1 1 1 1",
    "bad_banner": b"%%NotMatrixMarket x y z w\n1 1 1\n1 1 1\n",
    "long_banner_words": b"%%MatrixMarket " + b"a" * 500 + b" " + b"b" * 500 + b" c d\n1 1 0\n",
    "array_layout": b"%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n",
    "complex": b"%%MatrixMarket matrix coordinate complex general\n1 1 1\n1 1 1 0\n",
    "missing_size": b"%%MatrixMarket matrix coordinate real general\n% only comments\n",
    "size_two_fields": b"%%MatrixMarket matrix coordinate real general\n3 3\n1 1 1\n",
    "negative_size": b"%%MatrixMarket matrix coordinate real general\n-3 3 1\n1 1 1\n",
    "huge_digits": b"%%MatrixMarket matrix coordinate real general\n3 3 1\n" + b"9" * 40 + b" 1 1.0\n",
    "huge_size_digits": b"%%MatrixMarket matrix coordinate real general\n" + b"9" * 30 + b" 3 1\n1 1 1\n",
    "i64_min_index": b"%%MatrixMarket matrix coordinate real general\n3 3 1\n-9223372036854775808 1 1.0\n",
    "index_zero": b"%%MatrixMarket matrix coordinate real general\n3 3 1\n0 1 1.0\n",
    "index_past_n": b"%%MatrixMarket matrix coordinate real general\n3 3 1\n4 1 1.0\n",
    "missing_value": b"%%MatrixMarket matrix coordinate real general\n3 3 1\n1 1\n",
    "junk_value": b"%%MatrixMarket matrix coordinate real general\n3 3 1\n1 1 abc\n",
    "trailing_junk": b"%%MatrixMarket matrix coordinate real general\n3 3 1\n1 1 1.0 extra\n",
    "count_mismatch": b"%%MatrixMarket matrix coordinate real general\n3 3 5\n1 1 1.0\n",
    "symmetric_non_square": b"%%MatrixMarket matrix coordinate real symmetric\n3 4 1\n1 1 1.0\n",
    "nul_bytes": b"%%MatrixMarket matrix coordinate real general\n3 3 1\n1 \x00 1.0\n",
    "no_final_newline": b"%%MatrixMarket matrix coordinate real general\n3 3 1\n1 1 1.0",
    "value_at_eof_partial": b"%%MatrixMarket matrix coordinate real general\n3 3 1\n1 1 1e",
    "only_sign": b"%%MatrixMarket matrix coordinate real general\n3 3 1\n- + -\n",
}


def test_mtx_well_formed(exe, tmp_path):
    p = tmp_path / "good.mtx"
    p.write_bytes(GOOD)
    out = run(exe, "mtx", p, 1)
    assert out.split()[:4] == ["ok", "3", "4", "5"]
    q = tmp_path / "back.mtx"
    assert run(exe, "mtxwrite", p, q).startswith("ok 3 4 5")
    assert run(exe, "mtx", q, 2).split()[:4] == ["ok", "3", "4", "5"]


@pytest.mark.parametrize("name", sorted(MALFORMED))
def test_mtx_malformed_is_refused_cleanly(exe, tmp_path, name):
    p = tmp_path / (name + ".mtx")
    p.write_bytes(MALFORMED[name])
    for threads in (1, 4):
        out = run(exe, "mtx", p, threads)
        if name in ("no_final_newline",):
            assert out.startswith("ok 3 3 1")
        else:
            assert out.startswith("error:"), (name, out)


def test_mtx_parallel_chunks_and_a_bad_line_deep_inside(exe, tmp_path):
    """A body large enough to be cut into one chunk per thread (cuts fall at line ends), once clean and once
    with a malformed line in the middle and with the file cut off inside a line."""
    k = 120000
    rng = np.random.default_rng(5)
    r, c = rng.integers(1, 5001, k), rng.integers(1, 7001, k)
    lines = ["%d %d %.17g" % (a, b, v) for a, b, v in zip(r, c, rng.random(k))]
    head = "%%%%MatrixMarket matrix coordinate real general\n5000 7000 %d\n" % k
    p = tmp_path / "big.mtx"
    p.write_text(head + "\n".join(lines) + "\n")
    outs = {run(exe, "mtx", p, t) for t in (1, 3, 8)}
    assert len(outs) == 1 and outs.pop().startswith("ok 5000 7000 %d " % k)
    bad = list(lines)
    bad[k // 2] = "17 x 1.0"
    p.write_text(head + "\n".join(bad) + "\n")
    assert run(exe, "mtx", p, 8).startswith("error: malformed entry line")
    whole = (head + "\n".join(lines) + "\n").encode()
    p.write_bytes(whole[:len(whole) // 2 + 3])
    assert run(exe, "mtx", p, 8).startswith("error:")


def stored(tag=b"f64", n=2, m=3, nnz=2, vb=8, magic=b"SAMDCSR1", cut=None, pad=0):
    b = magic + tag.ljust(8, b"\0") + struct.pack("<qqq", n, m, nnz)
    b += struct.pack("<%dq" % max(n, 0), *range(1, max(n, 0) + 1)) if n < 1000 else b""
    b += struct.pack("<%dq" % max(nnz, 0), *range(max(nnz, 0))) if nnz < 1000 else b""
    b += b"\x01" * (vb * max(nnz, 0)) if nnz < 1000 else b""
    b += b"\0" * pad
    return b if cut is None else b[:cut]


def test_restore_header_checks(exe, tmp_path):
    cases = {
        "good": (stored(), "ok 2 3 2"),
        "good_empty": (stored(n=0, m=0, nnz=0), "ok 0 0 0"),
        "empty_file": (b"", "error: restore: null buffer"),
        "short": (stored(cut=20), "error: restore: buffer shorter than the header"),
        "magic": (stored(magic=b"SAMDCSR2"), "error: restore: not a stored"),
        "tag": (stored(tag=b"f32"), "error: restore: stored element type is f32, wanted f64"),
        "tag_unterminated_binary": (stored(tag=b"\xff\xfe\x80abcd!"), "error: restore: stored element type is ???abcd, wanted f64"),
        "negative": (stored(n=-1), "error: restore: negative size"),
        "overflowing_sizes": (stored(n=2**62, nnz=2**62), "error: restore: size in header out of range"),
        "huge_but_in_range": (stored(n=2**40, nnz=5), "error: restore: buffer holds"),
        "truncated_payload": (stored(cut=40 + 16 + 16 + 15), "error: restore: buffer holds 87 bytes, header describes 88"),
    }
    for name, (blob, want) in cases.items():
        p = tmp_path / (name + ".bin")
        p.write_bytes(blob)
        out = run(exe, "restore", p, "f64", 8)
        assert out.startswith(want), (name, out)
