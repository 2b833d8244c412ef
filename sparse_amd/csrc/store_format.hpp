// store_format.hpp -- byte layout of a stored opaque sr / sc value (futhark_store_opaque_* /
// futhark_restore_opaque_*, include/sparse_amd.h) and the checks restore runs on caller bytes before anything is
// copied.  Host code only (no HIP): the same file is compiled for the CPU with -fsanitize=address,undefined by
// `make asan` (hostcheck.cpp) and driven with malformed buffers by tests/test_host_sanitizers.py.
//
// Layout: "SAMDCSR1", 8-byte element tag ("f64", "f32", "i32", zero padded), int64 n, m, nnz of the STORED
// record (for a csc handle: of the transpose, compressed.fut:226), then row_off[n] i64 (inclusive, :94),
// col_idx[nnz] i64, vals[nnz].
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>

namespace samd {

struct StoreHeader {
  char magic[8];
  char tag[8];
  int64_t n, m, nnz;
};

constexpr int64_t kStoreMaxCount = (int64_t)1 << 46;  // rows / entries: far beyond any device, small enough that no size below overflows

// Bytes the whole value occupies for element size `vb`; 0 if the header's sizes are out of range.
inline size_t store_total_bytes(const StoreHeader &h, size_t vb) {
  if (h.n < 0 || h.m < 0 || h.nnz < 0 || h.n > kStoreMaxCount || h.nnz > kStoreMaxCount) return 0;
  return sizeof(StoreHeader) + sizeof(int64_t) * ((size_t)h.n + (size_t)h.nnz) + vb * (size_t)h.nnz;
}

// Reads and checks the header at p.  The generated restore API passes no length, so `avail` is (size_t)-1 there;
// callers that know how many bytes they hold (tests, file loaders) pass it and truncated buffers are refused.
// Returns 0 and fills *h, or non-zero with a message in msg.
inline int store_parse_header(const void *p, size_t avail, const char *want_tag, size_t vb, StoreHeader *h, char *msg,
                              size_t msglen) {
  if (!p) { snprintf(msg, msglen, "restore: null buffer"); return 1; }
  if (avail < sizeof(StoreHeader)) { snprintf(msg, msglen, "restore: buffer shorter than the header"); return 1; }
  memcpy(h, p, sizeof *h);
  if (memcmp(h->magic, "SAMDCSR1", 8) != 0) { snprintf(msg, msglen, "restore: not a stored sparse_amd matrix"); return 1; }
  h->tag[sizeof h->tag - 1] = 0;
  if (strcmp(h->tag, want_tag) != 0) {
    char shown[sizeof h->tag];
    for (size_t i = 0; i < sizeof shown; i++) {  // the tag is caller bytes: show printable ones only
      const unsigned char c = (unsigned char)h->tag[i];
      shown[i] = c == 0 ? 0 : ((c >= 32 && c < 127) ? (char)c : '?');
    }
    snprintf(msg, msglen, "restore: stored element type is %s, wanted %s", shown, want_tag);
    return 1;
  }
  if (h->n < 0 || h->m < 0 || h->nnz < 0) { snprintf(msg, msglen, "restore: negative size in header"); return 1; }
  const size_t total = store_total_bytes(*h, vb);
  if (total == 0) { snprintf(msg, msglen, "restore: size in header out of range"); return 1; }
  if (avail != (size_t)-1 && avail < total) {
    snprintf(msg, msglen, "restore: buffer holds %zu bytes, header describes %zu", avail, total);
    return 1;
  }
  return 0;
}

}  // namespace samd
