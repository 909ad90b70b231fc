/* Verification behind RDEN = 2 of dazimsurftomo_amd/csrc/disp.hip: for a small even divisor den,
 *   q = x*r;  q' = fma(fma(-den, q, x), r, q),  r = RN(1/den)
 * equals the correctly rounded x/den for every float x with 2.5e-31 < |x| < 2.5e30 (sign symmetric).
 *   gcc -O2 -ffp-contract=off -o check_fastdiv tools/check_fastdiv.c -lm
 *   ./check_fastdiv <den> [stride]       stride 1 = exhaustive (1.7e9 values, ~10 s per divisor); prints the mismatch count */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int main(int argc, char **argv) {
  if (argc < 2) return 2;
  const float den = (float)atof(argv[1]);
  const uint32_t stride = argc > 2 ? (uint32_t)atoi(argv[2]) : 1;
  const float r = 1.0f / den;
  uint64_t bad = 0, n = 0;
  for (uint64_t b = 0x0D000000u; b < 0x72000000u; b += stride) {
    const uint32_t bb = (uint32_t)b;
    float x;
    memcpy(&x, &bb, 4);
    const float q = x * r;
    const float q2 = fmaf(fmaf(-den, q, x), r, q);
    n++;
    if (q2 != x / den) bad++;
  }
  printf("den %g: %llu mismatches of %llu\n", den, (unsigned long long)bad, (unsigned long long)n);
  return bad != 0;
}
