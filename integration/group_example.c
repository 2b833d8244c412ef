/* group_example.c -- a C caller that row-shards one matrix over the GPUs of a node (include/sparse_amd.h
 * section 6): no Python, no torch.  usage: group_example [members [transport]]   (default: 1 member)
 * With one GPU, `group_example 2 peer` rehearses two members on device 0.
 *
 *   gcc -std=c99 -Iinclude integration/group_example.c -Lsparse_amd/lib -lsparse_amd \
 *       -Wl,-rpath,$PWD/sparse_amd/lib -o integration/group_example
 */
#include <stdio.h>
#include <stdlib.h>

#include "sparse_amd.h"

static int fail(struct sparse_amd_group *g, const char *what) {
  char *e = g ? sparse_amd_group_get_error(g) : NULL;
  fprintf(stderr, "%s failed: %s\n", what, e ? e : "(no message)");
  free(e);
  return 1;
}

int main(int argc, char **argv) {
  const int members = argc > 1 ? atoi(argv[1]) : 1;
  const char *transport = argc > 2 ? argv[2] : NULL;
  int devices[64];
  for (int k = 0; k < members && k < 64; k++) devices[k] = (transport && transport[0] == 'p') ? 0 : k;
  struct sparse_amd_group *g = NULL;
  if (sparse_amd_group_new(&g, members, devices, transport) != 0) return fail(g, "group_new");

  /* 1-D Laplacian of n points as unsorted triples with one duplicate pair: tridiagonal (-1, 2, -1) */
  const int64_t n = 100000;
  const int64_t cap = 3 * n + 1;
  int64_t *rs = malloc(sizeof(int64_t) * cap), *cs = malloc(sizeof(int64_t) * cap), k = 0;
  double *vs = malloc(sizeof(double) * cap), *x = malloc(sizeof(double) * n), *y = malloc(sizeof(double) * n);
  for (int64_t i = n - 1; i >= 0; i--) {            /* rows in descending order: assembly has to sort */
    if (i + 1 < n) { rs[k] = i; cs[k] = i + 1; vs[k++] = -1.0; }
    rs[k] = i; cs[k] = i; vs[k++] = i == 7 ? 0.5 : 2.0;
    if (i > 0) { rs[k] = i; cs[k] = i - 1; vs[k++] = -1.0; }
  }
  rs[k] = 7; cs[k] = 7; vs[k++] = 1.5;               /* duplicate of (7,7): 0.5 + 1.5 */
  for (int64_t i = 0; i < n; i++) x[i] = (double)(i % 5);

  struct sparse_amd_dcsr_f64 *A = NULL;
  struct sparse_amd_dvec_f64 *xv = NULL, *yv = NULL;
  if (sparse_amd_dcsr_f64_sparse(g, &A, n, n, rs, cs, vs, k) != 0) return fail(g, "sparse");
  if (sparse_amd_dvec_f64_new(g, &xv, A, 0, x) != 0 || sparse_amd_dvec_f64_new(g, &yv, A, 1, NULL) != 0) return fail(g, "dvec");
  if (sparse_amd_dcsr_f64_smvm(g, yv, A, xv) != 0) return fail(g, "smvm");
  if (sparse_amd_dvec_f64_values(g, yv, y) != 0) return fail(g, "values");
  for (int64_t i = 0; i < n; i++) {
    const double want = 2.0 * x[i] - (i > 0 ? x[i - 1] : 0.0) - (i + 1 < n ? x[i + 1] : 0.0);
    if (y[i] != want) { fprintf(stderr, "mismatch at %lld: %g != %g\n", (long long)i, y[i], want); return 1; }
  }
  int64_t rb[65];
  sparse_amd_dcsr_f64_bounds(g, A, rb, NULL);
  printf("group ok: %d member(s), transport %s, row bounds", members, sparse_amd_group_transport(g));
  for (int q = 0; q <= members; q++) printf(" %lld", (long long)rb[q]);
  printf("\n");
  sparse_amd_dvec_f64_free(g, xv); sparse_amd_dvec_f64_free(g, yv); sparse_amd_dcsr_f64_free(g, A);
  sparse_amd_group_free(g);
  free(rs); free(cs); free(vs); free(x); free(y);
  return 0;
}
