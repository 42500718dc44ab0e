// What is _mm_rcp_ps on this host, as a function?  (the reference forms 1/z of the projection and the
// t-distribution weights with rcpps, dense_tracking_impl.cpp:192,700 -- a 12-bit approximation whose bits differ between CPU
// vendors.)  Prints: the smallest k such that rcpps(1.m) depends only on the top k mantissa bits; whether rcpps(x 2^e) is
// rcpps(x) 2^-e exactly over the normal range; the special cases.  gcc -O2 -msse3 rcpps_probe.c -o rcpps_probe
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <xmmintrin.h>
static float rcp(float x) { return _mm_cvtss_f32(_mm_rcp_ss(_mm_set_ss(x))); }
static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float fromb(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
int main(void) {
  int k;
  for (k = 1; k <= 23; ++k) {
    int ok = 1;
    const uint32_t span = 1u << (23 - k);
    for (uint32_t top = 0; top < (1u << k) && ok; ++top) {
      const uint32_t r0 = bits(rcp(fromb(0x3f800000u | (top << (23 - k)))));
      for (uint32_t lo = 1; lo < span; ++lo)
        if (bits(rcp(fromb(0x3f800000u | (top << (23 - k)) | lo))) != r0) { ok = 0; break; }
    }
    if (ok) break;
  }
  printf("rcpps(1.m) depends on the top %d mantissa bits only\n", k);
  // distinct output mantissa bits
  uint32_t or_bits = 0;
  for (uint32_t m = 0; m < (1u << 23); m += 1u << (23 - (k < 23 ? k : 23))) or_bits |= bits(rcp(fromb(0x3f800000u | m)));
  printf("output bits ever set (mantissa part): 0x%06x\n", or_bits & 0x7fffffu);
  // exponent scaling
  long bad = 0, tested = 0;
  for (int e = 1; e <= 254; ++e)
    for (uint32_t m = 0; m < (1u << 23); m += 40961u) {
      const float x = fromb(((uint32_t)e << 23) | m), x1 = fromb(0x3f800000u | m);
      const uint32_t r1 = bits(rcp(x1));  // in (0.5, 1]: exponent field 126 or 127
      const int re = (int)(r1 >> 23) - (e - 127);
      const uint32_t expect = re >= 1 && re <= 254 ? ((uint32_t)re << 23) | (r1 & 0x7fffffu) : 0xffffffffu;
      const uint32_t got = bits(rcp(x));
      ++tested;
      if (expect != 0xffffffffu && got != expect) ++bad;
      if (expect == 0xffffffffu && e > 200 && m == 0) printf("  result would be denormal: rcp(2^%d) = 0x%08x\n", e - 127, got);
    }
  printf("exponent scaling exact for %ld of %ld normal-result samples\n", tested - bad, tested);
  const uint32_t edge[] = {0x00000000u, 0x80000000u, 0x00000001u, 0x007fffffu, 0x00800000u, 0x7e800000u, 0x7e800001u, 0x7effffffu,
                           0x7f000000u, 0x7f7fffffu, 0x7f800000u, 0xff800000u, 0x7fc00000u, 0x7f800001u, 0xffc12345u};
  for (unsigned i = 0; i < sizeof(edge) / 4; ++i) printf("  rcp(0x%08x) = 0x%08x\n", edge[i], bits(rcp(fromb(edge[i]))));
  for (uint32_t m = 0; m < 8; ++m) printf("  table[%u] = 0x%08x\n", m, bits(rcp(fromb(0x3f800000u | (m << (23 - (k < 23 ? k : 23)))))));
  return 0;
}
