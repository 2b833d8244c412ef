/* compressed_oracle.c -- CPU restatement of diku-dk/sparse `mk_compressed`.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check
 * in __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load it, and
 * only as the checker.  Nothing under sparse_amd/ links, imports or calls it;
 * the product path fails loudly when the HIP library is missing.
 *
 * What it restates: /root/reference/lib/github.com/diku-dk/sparse/compressed.fut
 * (module mk_compressed, lines 61-332) for element types f64, f32 and i32.
 * Each function in compressed_oracle.inc cites the lines it follows.
 * mono_oracle.inc restates mono.fut (module mk_mono, lines 61-209) the same way;
 * it is pinned by the 30 known-answer vectors of mono_test.fut
 * (tests/golden/mono_kat.json).
 *
 * Why a restatement: the reference is Futhark source; no `futhark` compiler is
 * present in this image and the arithmetic-bearing helpers are un-vendored
 * packages pinned in /root/reference/futhark.pkg:4-5
 *   github.com/diku-dk/sorts     0.6.0  (merge_sort)
 *   github.com/diku-dk/segmented 0.4.4  (segmented_reduce, expand,
 *                                        expand_outer_reduce, replicated_iota)
 * so the reference path is unbuildable here.  Their published semantics are
 * restated: merge_sort = sorted permutation under the comparator (restated as
 * a stable sort); segmented_reduce = one fold per flag-delimited segment;
 * expand_outer_reduce = per-row fold, empty row -> neutral element;
 * replicated_iota reps = index i repeated reps[i] times.
 *
 * Parity pin: every known-answer vector of the reference's own test file
 * (compressed_test.fut:13-178, 36 cases, all i32) is transcribed to
 * tests/golden/compressed_kat.json and checked by tests/test_oracle.py.
 * Floating point results, SpMM and anything larger than 5x5 are NOT pinned by
 * the reference ("parity unpinned" for those); tests/test_oracle.py cross-checks
 * them against scipy.sparse, which is an independent implementation and is
 * labelled as such.
 */
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define T double
#define SFX f64
#include "compressed_oracle.inc"
#include "mono_oracle.inc"
#undef T
#undef SFX

#define T float
#define SFX f32
#include "compressed_oracle.inc"
#include "mono_oracle.inc"
#undef T
#undef SFX

#define T int32_t
#define SFX i32
#include "compressed_oracle.inc"
#include "mono_oracle.inc"
#undef T
#undef SFX

/* splitmix64 on (seed, index): the counter-based generator every synthetic
 * input in tests/ and bench.py derives from (sparse_amd/synth.py holds the
 * numpy twin; tests assert they agree). */
uint64_t oracle_splitmix64(uint64_t seed, uint64_t index) {
  uint64_t z = seed + (index + 1) * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

int oracle_has_openmp(void) {
#ifdef _OPENMP
  return 1;
#else
  return 0;
#endif
}
