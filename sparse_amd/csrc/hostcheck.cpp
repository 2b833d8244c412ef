// hostcheck.cpp -- the library's host-side parsers compiled for the CPU with -fsanitize=address,undefined
// (`make asan`; no GPU, no HIP): mtx.hpp (parallel Matrix Market parser over mmap) and store_format.hpp (what
// futhark_restore_opaque_* checks on caller bytes).  tests/test_host_sanitizers.py feeds it well-formed and
// malformed inputs; a finding of either sanitizer aborts the process (-fno-sanitize-recover), a clean refusal
// prints "error: ..." and exits 0.
//
//   hostcheck mtx <file> <threads>          parse; prints "ok n m count sum(rows) sum(cols) sum(vals)"
//   hostcheck mtxwrite <in> <out>           parse <in>, write it back as <out>
//   hostcheck restore <file> <tag> <vb>     header checks on the file's bytes (exact-size heap buffer, so an
//                                           over-read is caught), then every payload byte is read
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mtx.hpp"
#include "store_format.hpp"

using namespace samd;

static int do_mtx(const char *path, int threads, const char *out) {
  MtxResult res;
  std::string err;
  if (mtx_read(path, threads, &res, &err) != 0) {
    printf("error: %s\n", err.c_str());
    return 0;
  }
  long long sr = 0, sc = 0;
  double sv = 0;
  for (int64_t i = 0; i < res.count; i++) { sr += res.rows[i]; sc += res.cols[i]; sv += res.vals[i]; }
  printf("ok %lld %lld %lld %lld %lld %.17g\n", (long long)res.n, (long long)res.m, (long long)res.count, sr, sc, sv);
  int rc = 0;
  if (out && mtx_write(out, res.n, res.m, res.count, res.rows, res.cols, res.vals, &err) != 0) {
    printf("error: %s\n", err.c_str());
    rc = 0;
  }
  free(res.rows); free(res.cols); free(res.vals);
  return rc;
}

static int do_restore(const char *path, const char *tag, size_t vb) {
  FILE *f = fopen(path, "rb");
  if (!f) { printf("error: cannot open %s\n", path); return 0; }
  fseek(f, 0, SEEK_END);
  const long len = ftell(f);
  fseek(f, 0, SEEK_SET);
  unsigned char *buf = (unsigned char *)malloc(len > 0 ? (size_t)len : 1);  // exact size: ASan sees any over-read
  if (len > 0 && fread(buf, 1, (size_t)len, f) != (size_t)len) { fclose(f); free(buf); printf("error: short read\n"); return 0; }
  fclose(f);
  StoreHeader h;
  char msg[160];
  if (store_parse_header(len > 0 ? buf : nullptr, (size_t)(len > 0 ? len : 0), tag, vb, &h, msg, sizeof msg) != 0) {
    printf("error: %s\n", msg);
    free(buf);
    return 0;
  }
  // what restore copies to the device, read here byte for byte
  const unsigned char *q = buf + sizeof h;
  unsigned long long sum = 0;
  const size_t payload = store_total_bytes(h, vb) - sizeof h;
  for (size_t i = 0; i < payload; i++) sum += q[i];
  printf("ok %lld %lld %lld %llu\n", (long long)h.n, (long long)h.m, (long long)h.nnz, sum);
  free(buf);
  return 0;
}

int main(int argc, char **argv) {
  if (argc >= 4 && !strcmp(argv[1], "mtx")) return do_mtx(argv[2], atoi(argv[3]), nullptr);
  if (argc >= 4 && !strcmp(argv[1], "mtxwrite")) return do_mtx(argv[2], 1, argv[3]);
  if (argc >= 5 && !strcmp(argv[1], "restore")) return do_restore(argv[2], argv[3], (size_t)atoi(argv[4]));
  fprintf(stderr, "usage: hostcheck mtx <file> <threads> | mtxwrite <in> <out> | restore <file> <tag> <vb>\n");
  return 2;
}
