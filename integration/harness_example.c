/* harness_example.c -- a caller written ONLY against the generated-library convention.
 *
 * It runs the reference's own SpMV known-answer test (compressed_test.fut:49-55) through
 * futhark_entry_test_csr_smvm, then the same matrix through the typed f64 entries, then a
 * store -> restore round trip of the opaque handle.  The header comes in by macro, so this one
 * unchanged file builds against libsparse_amd.so or against the C library `futhark multicore
 * --library` generates from compressed_test.fut + integration/sparse_entry.fut:
 *
 *   gcc -std=c99 -Iinclude integration/harness_example.c -Lsparse_amd/lib -lsparse_amd \
 *       -Wl,-rpath,$PWD/sparse_amd/lib -o integration/harness_example
 *   gcc -std=c99 -I. -DHARNESS_HEADER='"sparse_entry.h"' integration/harness_example.c \
 *       ./libsparse_ref.so -o harness_ref
 */
#include <stdio.h>
#include <stdlib.h>

#ifndef HARNESS_HEADER
#define HARNESS_HEADER "sparse_amd.h"
#endif
#include HARNESS_HEADER

static void die(struct futhark_context *ctx, const char *what) {
  char *e = futhark_context_get_error(ctx);
  fprintf(stderr, "%s failed: %s\n", what, e ? e : "(no message)");
  free(e);
  exit(1);
}

int main(void) {
  struct futhark_context_config *cfg = futhark_context_config_new();
  futhark_context_config_set_num_threads(cfg, 16);  /* the multicore library's worker pool; a no-op on the GPU */
  struct futhark_context *ctx = futhark_context_new(cfg);
  char *e0 = futhark_context_get_error(ctx);
  if (e0) { fprintf(stderr, "context: %s\n", e0); return 2; }

  const int64_t xs[] = {0, 0, 0, 1, 1, 2, 2, 2, 3, 4, 4};
  const int64_t ys[] = {0, 1, 3, 1, 2, 1, 2, 3, 3, 3, 4};
  const int32_t vs[] = {1, 2, 11, 3, 4, 5, 6, 7, 8, 9, 10};
  const int32_t v[] = {3, 1, 2, 6, 5};
  const int32_t want[] = {71, 11, 59, 48, 104};

  struct futhark_i64_1d *axs = futhark_new_i64_1d(ctx, xs, 11), *ays = futhark_new_i64_1d(ctx, ys, 11);
  struct futhark_i32_1d *avs = futhark_new_i32_1d(ctx, vs, 11), *av = futhark_new_i32_1d(ctx, v, 5);
  struct futhark_i32_1d *out = NULL;
  if (futhark_entry_test_csr_smvm(ctx, &out, 5, 5, axs, ays, avs, av) != 0) die(ctx, "test_csr_smvm");
  if (futhark_context_sync(ctx) != 0) die(ctx, "sync");
  int32_t got[5];
  if (futhark_values_i32_1d(ctx, out, got) != 0) die(ctx, "values");
  for (int i = 0; i < 5; i++)
    if (got[i] != want[i]) { fprintf(stderr, "KAT mismatch at %d: %d != %d\n", i, got[i], want[i]); return 1; }

  /* same matrix as f64 through an opaque handle */
  double dv[11], dx[5], dy[5];
  for (int i = 0; i < 11; i++) dv[i] = vs[i];
  for (int i = 0; i < 5; i++) dx[i] = v[i];
  struct futhark_f64_1d *fvs = futhark_new_f64_1d(ctx, dv, 11), *fx = futhark_new_f64_1d(ctx, dx, 5), *fy = NULL;
  struct futhark_opaque_csr_f64 *A = NULL;
  if (futhark_entry_csr_f64_sparse(ctx, &A, 5, 5, axs, ays, fvs) != 0) die(ctx, "csr_f64_sparse");
  if (futhark_entry_csr_f64_smvm(ctx, &fy, A, fx) != 0) die(ctx, "csr_f64_smvm");
  if (futhark_values_f64_1d(ctx, fy, dy) != 0) die(ctx, "values f64");
  for (int i = 0; i < 5; i++)
    if (dy[i] != (double)want[i]) { fprintf(stderr, "f64 mismatch at %d\n", i); return 1; }

  /* the opaque handle through its generated serialisers, and the product again on the restored one */
  void *bytes = NULL;
  size_t nbytes = 0;
  if (futhark_store_opaque_csr_f64(ctx, A, &bytes, &nbytes) != 0) die(ctx, "store_opaque_csr_f64");
  struct futhark_opaque_csr_f64 *A2 = futhark_restore_opaque_csr_f64(ctx, bytes);
  if (!A2) die(ctx, "restore_opaque_csr_f64");
  free(bytes);
  struct futhark_f64_1d *fy2 = NULL;
  if (futhark_entry_csr_f64_smvm(ctx, &fy2, A2, fx) != 0) die(ctx, "csr_f64_smvm (restored)");
  if (futhark_values_f64_1d(ctx, fy2, dy) != 0) die(ctx, "values f64 (restored)");
  for (int i = 0; i < 5; i++)
    if (dy[i] != (double)want[i]) { fprintf(stderr, "restored f64 mismatch at %d\n", i); return 1; }
  futhark_free_f64_1d(ctx, fy2);
  futhark_free_opaque_csr_f64(ctx, A2);

  futhark_free_f64_1d(ctx, fy); futhark_free_f64_1d(ctx, fx); futhark_free_f64_1d(ctx, fvs);
  futhark_free_opaque_csr_f64(ctx, A);
  futhark_free_i32_1d(ctx, out); futhark_free_i32_1d(ctx, av); futhark_free_i32_1d(ctx, avs);
  futhark_free_i64_1d(ctx, ays); futhark_free_i64_1d(ctx, axs);
  futhark_context_free(ctx);
  futhark_context_config_free(cfg);
  printf("harness ok: [71,11,59,48,104]\n");
  return 0;
}
