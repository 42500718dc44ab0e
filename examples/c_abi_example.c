/* The C ABI used directly from C99: build two pyramids from raw sensor frames (uint8 BGR + uint16 depth), align them,
 * print the transformation.  Input files as written by tests/test_cpp_adaptor.py (raw bytes, no header).
 *   cc -std=c99 -Iinclude examples/c_abi_example.c -Ldvo_slam_amd -ldvo_amd -Wl,-rpath,$PWD/dvo_slam_amd */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dvo_amd.h"

static void *read_file(const char *path, size_t bytes) {
  void *p = malloc(bytes);
  FILE *f = fopen(path, "rb");
  if (!p || !f || fread(p, 1, bytes, f) != bytes) {
    fprintf(stderr, "cannot read %s\n", path);
    exit(2);
  }
  fclose(f);
  return p;
}

#define CHECK(call)                                                                                         \
  do {                                                                                                      \
    int rc_ = (call);                                                                                       \
    if (rc_ != DVO_AMD_OK) {                                                                                \
      fprintf(stderr, "%s: %s [%s]\n", #call, dvo_amd_status_string(rc_), dvo_amd_last_error());             \
      return 1;                                                                                             \
    }                                                                                                       \
  } while (0)

int main(int argc, char **argv) {
  if (argc < 11) {
    fprintf(stderr, "usage: %s w h fx fy ox oy ref_bgr ref_depth cur_bgr cur_depth\n", argv[0]);
    return 2;
  }
  const int w = atoi(argv[1]), h = atoi(argv[2]);
  const float fx = (float)atof(argv[3]), fy = (float)atof(argv[4]), ox = (float)atof(argv[5]), oy = (float)atof(argv[6]);
  const size_t n = (size_t)w * (size_t)h;
  unsigned char *rb = read_file(argv[7], 3 * n), *cb = read_file(argv[9], 3 * n);
  unsigned short *rz = read_file(argv[8], 2 * n), *cz = read_file(argv[10], 2 * n);

  dvo_amd_config cfg;
  dvo_amd_default_config(&cfg);
  cfg.last_level = 0;
  dvo_amd_context *ctx = NULL;
  dvo_amd_pyramid *ref = NULL, *cur = NULL;
  CHECK(dvo_amd_context_create(0, &cfg, &ctx));
  CHECK(dvo_amd_pyramid_create_raw(0, rb, 3, 3 * w, rz, w, 1.0f / 5000.0f, 0, w, h, fx, fy, ox, oy, cfg.first_level + 1, 0.0, &ref));
  CHECK(dvo_amd_pyramid_create_raw(0, cb, 3, 3 * w, cz, w, 1.0f / 5000.0f, 0, w, h, fx, fy, ox, oy, cfg.first_level + 1, 0.0, &cur));

  dvo_amd_result res;
  memset(&res, 0, sizeof(res));
  CHECK(dvo_amd_match(ctx, ref, cur, NULL, &res));
  printf("isnan %d loglik %.9g levels %d ticks %d\n", res.is_nan, res.loglik, res.n_levels, res.n_ticks);
  for (int r = 0; r < 4; ++r)
    printf("%.17g %.17g %.17g %.17g\n", res.transformation[r], res.transformation[4 + r], res.transformation[8 + r],
           res.transformation[12 + r]);
  char line[256];
  if (dvo_amd_format_trajectory_line(1305031102.175304, res.transformation, line, (int)sizeof(line)) > 0) fputs(line, stdout);

  dvo_amd_pyramid_release(ref);
  dvo_amd_pyramid_release(cur);
  dvo_amd_context_destroy(ctx);
  free(rb), free(cb), free(rz), free(cz);
  return 0;
}
