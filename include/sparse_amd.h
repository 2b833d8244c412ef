/* sparse_amd.h -- C ABI of libsparse_amd.so, the MI355X-native implementation of
 * the diku-dk/sparse `mk_compressed` hot path.
 *
 * The reference is a Futhark library; the only thing a C caller can bind is the
 * API that `futhark c --library` (or `futhark hip --library`) generates from
 * `entry` definitions.  This header follows that generated-API convention
 * symbol for symbol (context / arrays / opaques / entries; outputs first as
 * pointer-to-pointer, scalars by value, 0 = success, error text through
 * futhark_context_get_error), so a harness written against the reference's
 * generated library links against this one unchanged.
 *
 *   S/ = lib/github.com/diku-dk/sparse/ in the reference tree.
 *
 * Three groups of entry points:
 *   (1) the reference's OWN entries -- S/compressed_test.fut, module
 *       `mk_compressed i32` -- under their generated names
 *       futhark_entry_test_*;
 *   (2) entries for f64 / f32 / i32 with opaque matrix handles.  The reference
 *       exports no such entries (its module is only reachable from Futhark
 *       source); integration/sparse_entry.fut is the entry file a maintainer adds
 *       to the reference to obtain exactly these symbols;
 *   (3) sparse_amd_* extensions with no counterpart in the generated API
 *       (stream hand-over, zero-copy wrapping, structure observation).
 *
 * Conventions that differ from textbook CSR (and are kept at this boundary):
 *   - row_off is the INCLUSIVE prefix sum of length n (S/compressed.fut:94,159);
 *   - all indices are int64_t (S/compressed.fut:66,94-95);
 *   - a CSC matrix of shape n x m IS the CSR matrix of its m x n transpose
 *     (S/compressed.fut:226); transpose moves no data (S/compressed.fut:185,220).
 *
 * All arrays and opaques are owned by the context and live in device memory.
 * futhark_new_* copies from the host; futhark_values_* copies to the host and
 * returns after the copy has completed.  Entries are asynchronous on the
 * context's HIP stream unless they must return a host scalar.
 */
#ifndef SPARSE_AMD_H
#define SPARSE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ context */
struct futhark_context_config;
struct futhark_context;

struct futhark_context_config *futhark_context_config_new(void);
void futhark_context_config_free(struct futhark_context_config *cfg);
void futhark_context_config_set_debugging(struct futhark_context_config *cfg, int flag);
void futhark_context_config_set_profiling(struct futhark_context_config *cfg, int flag);
void futhark_context_config_set_logging(struct futhark_context_config *cfg, int flag);
/* Device selection as in the generated GPU back ends: "#k" selects HIP device k. */
void futhark_context_config_set_device(struct futhark_context_config *cfg, const char *s);

struct futhark_context *futhark_context_new(struct futhark_context_config *cfg);
void futhark_context_free(struct futhark_context *ctx);
int futhark_context_sync(struct futhark_context *ctx);
/* Heap string owned by the caller (free()), or NULL if no error is pending. */
char *futhark_context_get_error(struct futhark_context *ctx);
int futhark_context_clear_caches(struct futhark_context *ctx);
char *futhark_context_report(struct futhark_context *ctx);

/* ------------------------------------------------------------------- arrays */
#define SPARSE_AMD_DECL_ARRAY_1D(NAME, CT)                                                        \
  struct futhark_##NAME##_1d;                                                                     \
  struct futhark_##NAME##_1d *futhark_new_##NAME##_1d(struct futhark_context *ctx,                \
                                                      const CT *data, int64_t dim0);              \
  /* data is a device pointer (hipDeviceptr_t); contents are copied */                            \
  struct futhark_##NAME##_1d *futhark_new_raw_##NAME##_1d(struct futhark_context *ctx,            \
                                                          const void *data, int64_t dim0);        \
  int futhark_free_##NAME##_1d(struct futhark_context *ctx, struct futhark_##NAME##_1d *arr);     \
  int futhark_values_##NAME##_1d(struct futhark_context *ctx, struct futhark_##NAME##_1d *arr,    \
                                 CT *data);                                                       \
  void *futhark_values_raw_##NAME##_1d(struct futhark_context *ctx,                               \
                                       struct futhark_##NAME##_1d *arr);                          \
  const int64_t *futhark_shape_##NAME##_1d(struct futhark_context *ctx,                           \
                                           struct futhark_##NAME##_1d *arr);                      \
  /* extension: alias caller-owned device memory without copying; free releases the handle only */\
  struct futhark_##NAME##_1d *sparse_amd_wrap_##NAME##_1d(struct futhark_context *ctx,            \
                                                          void *data, int64_t dim0);

#define SPARSE_AMD_DECL_ARRAY_2D(NAME, CT)                                                        \
  struct futhark_##NAME##_2d; /* row-major [dim0][dim1] */                                        \
  struct futhark_##NAME##_2d *futhark_new_##NAME##_2d(struct futhark_context *ctx,                \
                                                      const CT *data, int64_t dim0, int64_t dim1);\
  struct futhark_##NAME##_2d *futhark_new_raw_##NAME##_2d(                                        \
      struct futhark_context *ctx, const void *data, int64_t dim0, int64_t dim1);                 \
  int futhark_free_##NAME##_2d(struct futhark_context *ctx, struct futhark_##NAME##_2d *arr);     \
  int futhark_values_##NAME##_2d(struct futhark_context *ctx, struct futhark_##NAME##_2d *arr,    \
                                 CT *data);                                                       \
  void *futhark_values_raw_##NAME##_2d(struct futhark_context *ctx,                               \
                                       struct futhark_##NAME##_2d *arr);                          \
  const int64_t *futhark_shape_##NAME##_2d(struct futhark_context *ctx,                           \
                                           struct futhark_##NAME##_2d *arr);                      \
  struct futhark_##NAME##_2d *sparse_amd_wrap_##NAME##_2d(struct futhark_context *ctx,            \
                                                          void *data, int64_t dim0, int64_t dim1);

SPARSE_AMD_DECL_ARRAY_1D(i64, int64_t)
SPARSE_AMD_DECL_ARRAY_1D(i32, int32_t)
SPARSE_AMD_DECL_ARRAY_1D(f32, float)
SPARSE_AMD_DECL_ARRAY_1D(f64, double)
SPARSE_AMD_DECL_ARRAY_2D(i32, int32_t)
SPARSE_AMD_DECL_ARRAY_2D(f32, float)
SPARSE_AMD_DECL_ARRAY_2D(f64, double)

/* -------------------------------------------- (2) typed matrix entry points
 * TT in {f64, f32, i32}  <->  `module c = mk_compressed TT` (S/compressed.fut:61).
 * futhark_opaque_csr_TT is c.sr[n][m]; futhark_opaque_csc_TT is c.sc[n][m], whose
 * storage is the CSR record of the m x n transpose (S/compressed.fut:226).
 */
#define SPARSE_AMD_DECL_MATRIX(TT, CT)                                                            \
  struct futhark_opaque_csr_##TT;                                                                 \
  struct futhark_opaque_csc_##TT;                                                                 \
  int futhark_free_opaque_csr_##TT(struct futhark_context *ctx, struct futhark_opaque_csr_##TT *o);\
  int futhark_free_opaque_csc_##TT(struct futhark_context *ctx, struct futhark_opaque_csc_##TT *o);\
                                                                                                  \
  /* sr.sparse n m coo -- S/compressed.fut:154-160 (norm_coo :72-87).  Non-zero return on an   */ \
  /* out-of-range index (the assert at :156).                                                   */\
  int futhark_entry_csr_##TT##_sparse(struct futhark_context *ctx,                                \
                                      struct futhark_opaque_csr_##TT **out0, const int64_t n,     \
                                      const int64_t m, const struct futhark_i64_1d *rs,           \
                                      const struct futhark_i64_1d *cs,                            \
                                      const struct futhark_##TT##_1d *vs);                        \
  /* sc.sparse n m coo = sr.sparse m n (swap coo) -- S/compressed.fut:211-212 */                  \
  int futhark_entry_csc_##TT##_sparse(struct futhark_context *ctx,                                \
                                      struct futhark_opaque_csc_##TT **out0, const int64_t n,     \
                                      const int64_t m, const struct futhark_i64_1d *rs,           \
                                      const struct futhark_i64_1d *cs,                            \
                                      const struct futhark_##TT##_1d *vs);                        \
  /* sr.zero :98-103 / sc.zero :193-194 */                                                        \
  int futhark_entry_csr_##TT##_zero(struct futhark_context *ctx,                                  \
                                    struct futhark_opaque_csr_##TT **out0, const int64_t n,       \
                                    const int64_t m);                                             \
  int futhark_entry_csc_##TT##_zero(struct futhark_context *ctx,                                  \
                                    struct futhark_opaque_csc_##TT **out0, const int64_t n,       \
                                    const int64_t m);                                             \
  /* sr.eye :105-113 / sc.eye :199-200 */                                                         \
  int futhark_entry_csr_##TT##_eye(struct futhark_context *ctx,                                   \
                                   struct futhark_opaque_csr_##TT **out0, const int64_t n,        \
                                   const int64_t m);                                              \
  int futhark_entry_csc_##TT##_eye(struct futhark_context *ctx,                                   \
                                   struct futhark_opaque_csc_##TT **out0, const int64_t n,        \
                                   const int64_t m);                                              \
  /* sr.diag :115-120 / sc.diag :202-203 */                                                       \
  int futhark_entry_csr_##TT##_diag(struct futhark_context *ctx,                                  \
                                    struct futhark_opaque_csr_##TT **out0,                        \
                                    const struct futhark_##TT##_1d *v);                           \
  int futhark_entry_csc_##TT##_diag(struct futhark_context *ctx,                                  \
                                    struct futhark_opaque_csc_##TT **out0,                        \
                                    const struct futhark_##TT##_1d *v);                           \
  /* sr.dense :122-132 / sc.dense :214-215 (both return the matrix in its own n x m shape) */     \
  int futhark_entry_csr_##TT##_dense(struct futhark_context *ctx, struct futhark_##TT##_2d **out0,\
                                     const struct futhark_opaque_csr_##TT *a);                    \
  int futhark_entry_csc_##TT##_dense(struct futhark_context *ctx, struct futhark_##TT##_2d **out0,\
                                     const struct futhark_opaque_csc_##TT *a);                    \
  /* sr.scale :148-152 / sc.scale :196-197 */                                                     \
  int futhark_entry_csr_##TT##_scale(struct futhark_context *ctx,                                 \
                                     struct futhark_opaque_csr_##TT **out0, const CT v,           \
                                     const struct futhark_opaque_csr_##TT *a);                    \
  int futhark_entry_csc_##TT##_scale(struct futhark_context *ctx,                                 \
                                     struct futhark_opaque_csc_##TT **out0, const CT v,           \
                                     const struct futhark_opaque_csc_##TT *a);                    \
  /* sr.nnz :162-164 / sc.nnz :205-206 (counts stored values that are not zero) */                \
  int futhark_entry_csr_##TT##_nnz(struct futhark_context *ctx, int64_t *out0,                    \
                                   const struct futhark_opaque_csr_##TT *a);                      \
  int futhark_entry_csc_##TT##_nnz(struct futhark_context *ctx, int64_t *out0,                    \
                                   const struct futhark_opaque_csc_##TT *a);                      \
  /* sr.coo :166-177 / sc.coo :208-209 (unzip3: rows, cols, values in stored order) */            \
  int futhark_entry_csr_##TT##_coo(struct futhark_context *ctx, struct futhark_i64_1d **out0,     \
                                   struct futhark_i64_1d **out1, struct futhark_##TT##_1d **out2, \
                                   const struct futhark_opaque_csr_##TT *a);                      \
  int futhark_entry_csc_##TT##_coo(struct futhark_context *ctx, struct futhark_i64_1d **out0,     \
                                   struct futhark_i64_1d **out1, struct futhark_##TT##_1d **out2, \
                                   const struct futhark_opaque_csc_##TT *a);                      \
  /* sr.+ :179-180, sr.- :182-183 / sc.+ :217, sc.- :218 */                                       \
  int futhark_entry_csr_##TT##_add(struct futhark_context *ctx,                                   \
                                   struct futhark_opaque_csr_##TT **out0,                         \
                                   const struct futhark_opaque_csr_##TT *a,                       \
                                   const struct futhark_opaque_csr_##TT *b);                      \
  int futhark_entry_csr_##TT##_sub(struct futhark_context *ctx,                                   \
                                   struct futhark_opaque_csr_##TT **out0,                         \
                                   const struct futhark_opaque_csr_##TT *a,                       \
                                   const struct futhark_opaque_csr_##TT *b);                      \
  int futhark_entry_csc_##TT##_add(struct futhark_context *ctx,                                   \
                                   struct futhark_opaque_csc_##TT **out0,                         \
                                   const struct futhark_opaque_csc_##TT *a,                       \
                                   const struct futhark_opaque_csc_##TT *b);                      \
  int futhark_entry_csc_##TT##_sub(struct futhark_context *ctx,                                   \
                                   struct futhark_opaque_csc_##TT **out0,                         \
                                   const struct futhark_opaque_csc_##TT *a,                       \
                                   const struct futhark_opaque_csc_##TT *b);                      \
  /* sr.transpose :185-186 / sc.transpose :220-221: re-labels, moves no data (the new handle   */ \
  /* shares storage with its argument; free both).                                              */\
  int futhark_entry_csr_##TT##_transpose(struct futhark_context *ctx,                             \
                                         struct futhark_opaque_csc_##TT **out0,                   \
                                         const struct futhark_opaque_csr_##TT *a);                \
  int futhark_entry_csc_##TT##_transpose(struct futhark_context *ctx,                             \
                                         struct futhark_opaque_csr_##TT **out0,                   \
                                         const struct futhark_opaque_csc_##TT *a);                \
  /* sr.smvm :134-146:  y[n] = A[n][m] x[m] */                                                    \
  int futhark_entry_csr_##TT##_smvm(struct futhark_context *ctx, struct futhark_##TT##_1d **out0, \
                                    const struct futhark_opaque_csr_##TT *a,                      \
                                    const struct futhark_##TT##_1d *x);                           \
  /* sc.vsmm :223-224:  out[m] = a[n] B[n][m] = sr.smvm (transpose B) a */                        \
  int futhark_entry_csc_##TT##_vsmm(struct futhark_context *ctx, struct futhark_##TT##_1d **out0, \
                                    const struct futhark_##TT##_1d *a,                            \
                                    const struct futhark_opaque_csc_##TT *b);                     \
  /* sparse x dense panel.  Not a member of `compressed`; defined as the column-wise smvm      */ \
  /* C[:,j] = sr.smvm A B[:,j] (:134-146; precedent mono.sr.dmsmm, S/mono.fut:161-162).        */ \
  int futhark_entry_csr_##TT##_smm(struct futhark_context *ctx, struct futhark_##TT##_2d **out0,  \
                                   const struct futhark_opaque_csr_##TT *a,                       \
                                   const struct futhark_##TT##_2d *b);                            \
  /* physical CSR(A) -> CSC(A) = sc.sparse n m (sr.coo A)  (:166-177 then :211-212), and back   */\
  /* CSC(A) -> CSR(A) = sr.sparse n m (sc.coo A)  (:208-209 then :154-160).                     */\
  int futhark_entry_csr_##TT##_to_csc(struct futhark_context *ctx,                                \
                                      struct futhark_opaque_csc_##TT **out0,                      \
                                      const struct futhark_opaque_csr_##TT *a);                   \
  int futhark_entry_csc_##TT##_to_csr(struct futhark_context *ctx,                                \
                                      struct futhark_opaque_csr_##TT **out0,                      \
                                      const struct futhark_opaque_csc_##TT *a);                   \
  /* smsmm :268-331: sr[n][m] -> sc[m][k] -> sr[n][k] */                                          \
  int futhark_entry_smsmm_##TT(struct futhark_context *ctx, struct futhark_opaque_csr_##TT **out0,\
                               const struct futhark_opaque_csr_##TT *a,                           \
                               const struct futhark_opaque_csc_##TT *b);                          \
                                                                                                  \
  /* ---- (3) extensions: no counterpart in the generated API ------------------------------- */  \
  /* Stored shape and stored entry count of the underlying CSR record (for a csc handle: of   */  \
  /* the transposed record).                                                                   */ \
  int sparse_amd_csr_##TT##_dims(struct futhark_context *ctx,                                     \
                                 const struct futhark_opaque_csr_##TT *a, int64_t *n, int64_t *m, \
                                 int64_t *stored);                                                \
  int sparse_amd_csc_##TT##_dims(struct futhark_context *ctx,                                     \
                                 const struct futhark_opaque_csc_##TT *a, int64_t *n, int64_t *m, \
                                 int64_t *stored);                                                \
  /* The record fields in the reference's convention (S/compressed.fut:93-96): row_off[n]     */  \
  /* inclusive, col_idx[nnz] as i64, vals[nnz].  The module ascription hides them in the      */  \
  /* reference; parity tests need them to compare index arrays bit for bit.                   */  \
  int sparse_amd_csr_##TT##_fields(struct futhark_context *ctx, struct futhark_i64_1d **row_off,  \
                                   struct futhark_i64_1d **col_idx,                               \
                                   struct futhark_##TT##_1d **vals,                               \
                                   const struct futhark_opaque_csr_##TT *a);                      \
  int sparse_amd_csc_##TT##_fields(struct futhark_context *ctx, struct futhark_i64_1d **row_off,  \
                                   struct futhark_i64_1d **col_idx,                               \
                                   struct futhark_##TT##_1d **vals,                               \
                                   const struct futhark_opaque_csc_##TT *a);                      \
  /* Adopt an already-normalised record (row_off[n] inclusive, col_idx strictly increasing    */  \
  /* within each row): how a row-sharded rank builds its slice without re-sorting.  Checked:  */  \
  /* lengths, monotone row_off, 0 <= col < m.                                                  */ \
  int sparse_amd_csr_##TT##_from_fields(struct futhark_context *ctx,                              \
                                        struct futhark_opaque_csr_##TT **out0, const int64_t n,   \
                                        const int64_t m, const struct futhark_i64_1d *row_off,    \
                                        const struct futhark_i64_1d *col_idx,                     \
                                        const struct futhark_##TT##_1d *vals);               \
  /* sr.smvm (:134-146) into caller-owned storage, for loops that must not create an array per */ \
  /* product (the row-sharded multi-GPU step).  rows == NULL: y[i] = (A x)[i], y has A's n      */ \
  /* elements.  rows != NULL (n distinct indices into y): y[rows[i]] += (A x)[i] -- how a rank  */ \
  /* adds the part of its rows that multiplies x entries received from other ranks (small       */ \
  /* records: one launch, a row summed sub-wavefront-wise, i.e. in another order than smvm).    */ \
  int sparse_amd_csr_##TT##_smvm_into(struct futhark_context *ctx, struct futhark_##TT##_1d *y,   \
                                      const struct futhark_opaque_csr_##TT *a,                    \
                                      const struct futhark_##TT##_1d *x,                          \
                                      const struct futhark_i64_1d *rows);

SPARSE_AMD_DECL_MATRIX(f64, double)
SPARSE_AMD_DECL_MATRIX(f32, float)
SPARSE_AMD_DECL_MATRIX(i32, int32_t)

/* ----------------------- (1) the reference's own entries (mk_compressed i32)
 * Names and argument order are what `futhark c --library compressed_test.fut`
 * generates for the entry definitions cited on each line.
 */
/* S/compressed_test.fut:30-31 */
int futhark_entry_test_csr_eye(struct futhark_context *ctx, struct futhark_i32_2d **out0,
                               const int64_t in0, const int64_t in1);
/* S/compressed_test.fut:44-45 */
int futhark_entry_test_csr_sparse(struct futhark_context *ctx, struct futhark_i32_2d **out0,
                                  const int64_t in0, const int64_t in1,
                                  const struct futhark_i64_1d *in2, const struct futhark_i64_1d *in3,
                                  const struct futhark_i32_1d *in4);
/* S/compressed_test.fut:57-59 (in5 is consumed: `*[m]csr.t`) */
int futhark_entry_test_csr_smvm(struct futhark_context *ctx, struct futhark_i32_1d **out0,
                                const int64_t in0, const int64_t in1,
                                const struct futhark_i64_1d *in2, const struct futhark_i64_1d *in3,
                                const struct futhark_i32_1d *in4, const struct futhark_i32_1d *in5);
/* S/compressed_test.fut:68-69 */
int futhark_entry_test_csr_nnz(struct futhark_context *ctx, int64_t *out0, const int64_t in0,
                               const int64_t in1, const struct futhark_i64_1d *in2,
                               const struct futhark_i64_1d *in3, const struct futhark_i32_1d *in4);
/* S/compressed_test.fut:76-78 */
int futhark_entry_test_csr_coo(struct futhark_context *ctx, struct futhark_i64_1d **out0,
                               struct futhark_i64_1d **out1, struct futhark_i32_1d **out2,
                               const int64_t in0, const int64_t in1,
                               const struct futhark_i64_1d *in2, const struct futhark_i64_1d *in3,
                               const struct futhark_i32_1d *in4);
/* S/compressed_test.fut:104-105 */
int futhark_entry_test_csc_eye(struct futhark_context *ctx, struct futhark_i32_2d **out0,
                               const int64_t in0, const int64_t in1);
/* S/compressed_test.fut:118-119 */
int futhark_entry_test_csc_sparse(struct futhark_context *ctx, struct futhark_i32_2d **out0,
                                  const int64_t in0, const int64_t in1,
                                  const struct futhark_i64_1d *in2, const struct futhark_i64_1d *in3,
                                  const struct futhark_i32_1d *in4);
/* S/compressed_test.fut:128-129 */
int futhark_entry_test_csc_nnz(struct futhark_context *ctx, int64_t *out0, const int64_t in0,
                               const int64_t in1, const struct futhark_i64_1d *in2,
                               const struct futhark_i64_1d *in3, const struct futhark_i32_1d *in4);
/* S/compressed_test.fut:136-138 */
int futhark_entry_test_csc_coo(struct futhark_context *ctx, struct futhark_i64_1d **out0,
                               struct futhark_i64_1d **out1, struct futhark_i32_1d **out2,
                               const int64_t in0, const int64_t in1,
                               const struct futhark_i64_1d *in2, const struct futhark_i64_1d *in3,
                               const struct futhark_i32_1d *in4);
/* S/compressed_test.fut:145-147 */
int futhark_entry_test_csr_transpose(struct futhark_context *ctx, struct futhark_i32_2d **out0,
                                     const int64_t in0, const int64_t in1,
                                     const struct futhark_i64_1d *in2,
                                     const struct futhark_i64_1d *in3,
                                     const struct futhark_i32_1d *in4);
/* S/compressed_test.fut:154-156 */
int futhark_entry_test_csc_transpose(struct futhark_context *ctx, struct futhark_i32_2d **out0,
                                     const int64_t in0, const int64_t in1,
                                     const struct futhark_i64_1d *in2,
                                     const struct futhark_i64_1d *in3,
                                     const struct futhark_i32_1d *in4);
/* S/compressed_test.fut:172-178 */
int futhark_entry_test_smsmm(struct futhark_context *ctx, struct futhark_i32_2d **out0,
                             const int64_t in0, const int64_t in1, const int64_t in2,
                             const struct futhark_i64_1d *in3, const struct futhark_i64_1d *in4,
                             const struct futhark_i32_1d *in5, const struct futhark_i64_1d *in6,
                             const struct futhark_i64_1d *in7, const struct futhark_i32_1d *in8);

/* ---------------------------------------------------- (4) mono sparse matrices
 * TT in {f64, f32, i32}  <->  `module mo = mk_mono TT` (S/mono.fut:61): at most one stored
 * element per row (msr) or per column (msc).  futhark_opaque_msr_TT is mo.sr[n][m]
 * (record col_idx[n], vals[n], S/mono.fut:95); futhark_opaque_msc_TT is mo.sc[n][m], whose
 * storage is the msr record of the m x n transpose (S/mono.fut:203).  SURVEY.md 8(f) row 3.
 */
#define SPARSE_AMD_DECL_MONO(TT, CT)                                                              \
  struct futhark_opaque_msr_##TT;                                                                 \
  struct futhark_opaque_msc_##TT;                                                                 \
  int futhark_free_opaque_msr_##TT(struct futhark_context *ctx, struct futhark_opaque_msr_##TT *o);\
  int futhark_free_opaque_msc_##TT(struct futhark_context *ctx, struct futhark_opaque_msc_##TT *o);\
  /* sparse (S/mono.fut:119-129; sc: :188-189).  Non-zero return for an index out of range  */    \
  /* and for two elements in one row (one column for msc) -- the reference's two asserts.    */   \
  int futhark_entry_msr_##TT##_sparse(struct futhark_context *ctx,                                \
                                      struct futhark_opaque_msr_##TT **out0, const int64_t n,     \
                                      const int64_t m, const struct futhark_i64_1d *rs,           \
                                      const struct futhark_i64_1d *cs,                            \
                                      const struct futhark_##TT##_1d *vs);                        \
  int futhark_entry_msc_##TT##_sparse(struct futhark_context *ctx,                                \
                                      struct futhark_opaque_msc_##TT **out0, const int64_t n,     \
                                      const int64_t m, const struct futhark_i64_1d *rs,           \
                                      const struct futhark_i64_1d *cs,                            \
                                      const struct futhark_##TT##_1d *vs);                        \
  /* zero :97-100 / :170-171, eye :102-105 / :176-177, diag :107-110 / :179-180 */                \
  int futhark_entry_msr_##TT##_zero(struct futhark_context *ctx,                                  \
                                    struct futhark_opaque_msr_##TT **out0, const int64_t n,       \
                                    const int64_t m);                                             \
  int futhark_entry_msc_##TT##_zero(struct futhark_context *ctx,                                  \
                                    struct futhark_opaque_msc_##TT **out0, const int64_t n,       \
                                    const int64_t m);                                             \
  int futhark_entry_msr_##TT##_eye(struct futhark_context *ctx,                                   \
                                   struct futhark_opaque_msr_##TT **out0, const int64_t n,        \
                                   const int64_t m);                                              \
  int futhark_entry_msc_##TT##_eye(struct futhark_context *ctx,                                   \
                                   struct futhark_opaque_msc_##TT **out0, const int64_t n,        \
                                   const int64_t m);                                              \
  int futhark_entry_msr_##TT##_diag(struct futhark_context *ctx,                                  \
                                    struct futhark_opaque_msr_##TT **out0,                        \
                                    const struct futhark_##TT##_1d *v);                           \
  int futhark_entry_msc_##TT##_diag(struct futhark_context *ctx,                                  \
                                    struct futhark_opaque_msc_##TT **out0,                        \
                                    const struct futhark_##TT##_1d *v);                           \
  /* dense :112-114 / :191-192 */                                                                 \
  int futhark_entry_msr_##TT##_dense(struct futhark_context *ctx, struct futhark_##TT##_2d **out0,\
                                     const struct futhark_opaque_msr_##TT *a);                    \
  int futhark_entry_msc_##TT##_dense(struct futhark_context *ctx, struct futhark_##TT##_2d **out0,\
                                     const struct futhark_opaque_msc_##TT *a);                    \
  /* scale :116-117 / :173-174 */                                                                 \
  int futhark_entry_msr_##TT##_scale(struct futhark_context *ctx,                                 \
                                     struct futhark_opaque_msr_##TT **out0, const CT v,           \
                                     const struct futhark_opaque_msr_##TT *a);                    \
  int futhark_entry_msc_##TT##_scale(struct futhark_context *ctx,                                 \
                                     struct futhark_opaque_msc_##TT **out0, const CT v,           \
                                     const struct futhark_opaque_msc_##TT *a);                    \
  /* nnz :131-133 / :182-183 (stored values that are not zero) */                                 \
  int futhark_entry_msr_##TT##_nnz(struct futhark_context *ctx, int64_t *out0,                    \
                                   const struct futhark_opaque_msr_##TT *a);                      \
  int futhark_entry_msc_##TT##_nnz(struct futhark_context *ctx, int64_t *out0,                    \
                                   const struct futhark_opaque_msc_##TT *a);                      \
  /* coo :135-137 / :185-186 (zero values are filtered out, unlike compressed) */                 \
  int futhark_entry_msr_##TT##_coo(struct futhark_context *ctx, struct futhark_i64_1d **out0,     \
                                   struct futhark_i64_1d **out1, struct futhark_##TT##_1d **out2, \
                                   const struct futhark_opaque_msr_##TT *a);                      \
  int futhark_entry_msc_##TT##_coo(struct futhark_context *ctx, struct futhark_i64_1d **out0,     \
                                   struct futhark_i64_1d **out1, struct futhark_##TT##_1d **out2, \
                                   const struct futhark_opaque_msc_##TT *a);                      \
  /* + and - :139-149 / :194-195: non-zero return when the operands' col_idx differ */            \
  int futhark_entry_msr_##TT##_add(struct futhark_context *ctx,                                   \
                                   struct futhark_opaque_msr_##TT **out0,                         \
                                   const struct futhark_opaque_msr_##TT *a,                       \
                                   const struct futhark_opaque_msr_##TT *b);                      \
  int futhark_entry_msr_##TT##_sub(struct futhark_context *ctx,                                   \
                                   struct futhark_opaque_msr_##TT **out0,                         \
                                   const struct futhark_opaque_msr_##TT *a,                       \
                                   const struct futhark_opaque_msr_##TT *b);                      \
  int futhark_entry_msc_##TT##_add(struct futhark_context *ctx,                                   \
                                   struct futhark_opaque_msc_##TT **out0,                         \
                                   const struct futhark_opaque_msc_##TT *a,                       \
                                   const struct futhark_opaque_msc_##TT *b);                      \
  int futhark_entry_msc_##TT##_sub(struct futhark_context *ctx,                                   \
                                   struct futhark_opaque_msc_##TT **out0,                         \
                                   const struct futhark_opaque_msc_##TT *a,                       \
                                   const struct futhark_opaque_msc_##TT *b);                      \
  /* transpose :151-152 / :197-198: no data moves, the handle is re-labelled */                   \
  int futhark_entry_msr_##TT##_transpose(struct futhark_context *ctx,                             \
                                         struct futhark_opaque_msc_##TT **out0,                   \
                                         const struct futhark_opaque_msr_##TT *a);                \
  int futhark_entry_msc_##TT##_transpose(struct futhark_context *ctx,                             \
                                         struct futhark_opaque_msr_##TT **out0,                   \
                                         const struct futhark_opaque_msc_##TT *a);                \
  /* smvm :154-155 */                                                                             \
  int futhark_entry_msr_##TT##_smvm(struct futhark_context *ctx, struct futhark_##TT##_1d **out0, \
                                    const struct futhark_opaque_msr_##TT *a,                      \
                                    const struct futhark_##TT##_1d *v);                           \
  /* vsmm :157-159 (msr: scatter-add, summed per column in ascending row order) / :200-201 */     \
  int futhark_entry_msr_##TT##_vsmm(struct futhark_context *ctx, struct futhark_##TT##_1d **out0, \
                                    const struct futhark_##TT##_1d *v,                            \
                                    const struct futhark_opaque_msr_##TT *a);                     \
  int futhark_entry_msc_##TT##_vsmm(struct futhark_context *ctx, struct futhark_##TT##_1d **out0, \
                                    const struct futhark_##TT##_1d *a,                            \
                                    const struct futhark_opaque_msc_##TT *b);                     \
  /* dmsmm :161-162: dense [n][k] times msr[k][m] -> dense [n][m] */                              \
  int futhark_entry_msr_##TT##_dmsmm(struct futhark_context *ctx, struct futhark_##TT##_2d **out0,\
                                     const struct futhark_##TT##_2d *d,                           \
                                     const struct futhark_opaque_msr_##TT *s);                    \
  /* extension (no counterpart in the generated API): the record fields col_idx[n], vals[n] */    \
  /* (for an msc handle: of the transposed record), for bit-for-bit parity tests.           */    \
  int sparse_amd_msr_##TT##_fields(struct futhark_context *ctx, struct futhark_i64_1d **col_idx,  \
                                   struct futhark_##TT##_1d **vals,                               \
                                   const struct futhark_opaque_msr_##TT *a);                      \
  int sparse_amd_msc_##TT##_fields(struct futhark_context *ctx, struct futhark_i64_1d **col_idx,  \
                                   struct futhark_##TT##_1d **vals,                               \
                                   const struct futhark_opaque_msc_##TT *a);

SPARSE_AMD_DECL_MONO(f64, double)
SPARSE_AMD_DECL_MONO(f32, float)
SPARSE_AMD_DECL_MONO(i32, int32_t)

/* The reference's own entries for mono (mk_mono i32, S/mono_test.fut:5). */
#define SPARSE_AMD_COO_ARGS                                                                       \
  const int64_t in0, const int64_t in1, const struct futhark_i64_1d *in2,                         \
      const struct futhark_i64_1d *in3, const struct futhark_i32_1d *in4
/* S/mono_test.fut:30-31 */
int futhark_entry_test_msr_eye(struct futhark_context *ctx, struct futhark_i32_2d **out0,
                               const int64_t in0, const int64_t in1);
/* S/mono_test.fut:42-43 */
int futhark_entry_test_msr_sparse(struct futhark_context *ctx, struct futhark_i32_2d **out0,
                                  SPARSE_AMD_COO_ARGS);
/* S/mono_test.fut:55-57 (in5 is consumed) */
int futhark_entry_test_msr_smvm(struct futhark_context *ctx, struct futhark_i32_1d **out0,
                                SPARSE_AMD_COO_ARGS, const struct futhark_i32_1d *in5);
/* S/mono_test.fut:66-67 */
int futhark_entry_test_msr_nnz(struct futhark_context *ctx, int64_t *out0, SPARSE_AMD_COO_ARGS);
/* S/mono_test.fut:74-76 */
int futhark_entry_test_msr_coo(struct futhark_context *ctx, struct futhark_i64_1d **out0,
                               struct futhark_i64_1d **out1, struct futhark_i32_1d **out2,
                               SPARSE_AMD_COO_ARGS);
/* S/mono_test.fut:87-89: (n, m, v[n], xs, ys, vs) */
int futhark_entry_test_msr_vsmm(struct futhark_context *ctx, struct futhark_i32_1d **out0,
                                const int64_t in0, const int64_t in1,
                                const struct futhark_i32_1d *in2, const struct futhark_i64_1d *in3,
                                const struct futhark_i64_1d *in4, const struct futhark_i32_1d *in5);
/* S/mono_test.fut:114-115 */
int futhark_entry_test_msc_eye(struct futhark_context *ctx, struct futhark_i32_2d **out0,
                               const int64_t in0, const int64_t in1);
/* S/mono_test.fut:126-127 */
int futhark_entry_test_msc_sparse(struct futhark_context *ctx, struct futhark_i32_2d **out0,
                                  SPARSE_AMD_COO_ARGS);
/* S/mono_test.fut:136-137 */
int futhark_entry_test_msc_nnz(struct futhark_context *ctx, int64_t *out0, SPARSE_AMD_COO_ARGS);
/* S/mono_test.fut:144-146 */
int futhark_entry_test_msc_coo(struct futhark_context *ctx, struct futhark_i64_1d **out0,
                               struct futhark_i64_1d **out1, struct futhark_i32_1d **out2,
                               SPARSE_AMD_COO_ARGS);
/* S/mono_test.fut:153-155 */
int futhark_entry_test_msr_transpose(struct futhark_context *ctx, struct futhark_i32_2d **out0,
                                     SPARSE_AMD_COO_ARGS);
/* S/mono_test.fut:162-164 */
int futhark_entry_test_msc_transpose(struct futhark_context *ctx, struct futhark_i32_2d **out0,
                                     SPARSE_AMD_COO_ARGS);

/* ------------------------------------------- (5) Matrix Market loader (host side)
 * No counterpart in the reference, which has no file format: its inputs only arrive as entry-point
 * arrays.  SURVEY.md 8(f) row 4: the step in front of `sparse`.  Reads a `coordinate` file
 * (real | integer | pattern; general | symmetric | skew-symmetric) into 0-based i64 triples in file
 * order, symmetric files expanded; `threads` > 1 parses the body in parallel.  The three arrays are
 * malloc'ed: release each with sparse_amd_mtx_free.  On failure returns non-zero and, if `error` is
 * not NULL, a malloc'ed message the caller frees.  No GPU is touched. */
int sparse_amd_mtx_read(const char *path, int threads, int64_t *n, int64_t *m, int64_t *count,
                        int64_t **rows, int64_t **cols, double **vals, char **error);
/* Writes `coordinate real general` with 17 significant digits (every f64 round-trips). */
int sparse_amd_mtx_write(const char *path, int64_t n, int64_t m, int64_t count, const int64_t *rows,
                         const int64_t *cols, const double *vals, char **error);
void sparse_amd_mtx_free(void *p);

/* ------------------------------------------------- (3) context extensions
 * No counterpart in the generated API. */
/* Run all subsequent work on the caller's HIP stream (hipStream_t); the context
 * first drains its current stream.  NULL is the HIP null stream (what torch calls
 * its default stream), not "no stream". */
int sparse_amd_context_set_stream(struct futhark_context *ctx, void *hip_stream);
void *sparse_amd_context_get_stream(struct futhark_context *ctx);
/* Tuning / test overrides.  Keys: "spmv_algo" (0 auto, 1 nnz-split stream,
 * 2 wavefront-per-row, 3 sub-wavefront-per-row), "spmv_lanes" (sub-wavefront
 * width, power of two <= 64), "wide_col" (1: keep col_idx as i64 on device),
 * "spmm_algo" (0 streaming row panels with banded long-row tiles on the matrix cores,
 * 2 plain wavefront-per-row, 3 streaming row panels without the matrix cores), "sort_algo"
 * (assembly's radix sort: 0 one global histogram + look-back, 1 histogram pass per digit),
 * "spmv_reorder" (hot-column renumbering for gather-bound SpMV: 0 decided per record, 1 whenever the
 * record is large enough, -1 never), "spmv_cblk" (column-blocked copy for wide records with flat column
 * popularity: 0 decided per record, W > 0 always, with W columns per block, -1 never).  "spmv_pb" (two-pass SpMV with the
 * x and y slices in LDS: 0 large skewed records of 4-byte values, 1 also flat column popularity, 2 always,
 * -1 never), "spmv_col16" (16-bit tile-relative columns for banded / stencil records: 0 decided per record,
 * -1 never).  Read-only: "spmv_last_path", "spmv_last_col_bytes" (what the last product ran).  The per-record
 * choices are made at a record's first product.
 * Returns non-zero for an unknown key. */
int sparse_amd_context_set_option(struct futhark_context *ctx, const char *key, int64_t value);
int64_t sparse_amd_context_get_option(struct futhark_context *ctx, const char *key);
const char *sparse_amd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SPARSE_AMD_H */
