"""sparse_amd -- MI355X-native implementation of the diku-dk/sparse
`mk_compressed` hot path (CSR/CSC assembly, SpMV, SpMM, CSR<->CSC) behind a
C-ABI shaped like `futhark c --library` output.

`sparse_amd.compressed` is the host-side mirror of the reference's module
surface (compressed.fut:24-54) over that C ABI, `sparse_amd.mono` the same for
`mk_mono` (mono.fut:24-57); `sparse_amd.synth` generates the
seeded synthetic inputs.  The HIP library is loaded lazily and its absence is a
hard error: there is no CPU fallback.
"""
from .compressed import Context, mk_compressed, LibraryMissing  # noqa: F401
from .mono import mk_mono  # noqa: F401
