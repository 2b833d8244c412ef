"""CPU oracle for the diku-dk/sparse `mk_compressed` hot path.

TEST INFRASTRUCTURE ONLY.  Importable from tests/, `__graft_entry__.smoke()`
and bench.py's `cpu_baseline` leg; never from `sparse_amd/`.

`oracle.oracle` is a ctypes binding over `liboracle.so`, the plain-C
restatement in `compressed_oracle.c` (which cites the reference lines).
"""
