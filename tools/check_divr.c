/* Verification of divr() in dazimsurftomo_amd/csrc/rays.hip: for an fp32 divisor d and rd = RN(1/d) in double precision,
 *   (float)((double)x * rd) == x / d     for EVERY float x (all 2^32 bit patterns; NaN results compared as NaN).
 * DESIGN.md section 4 has the argument (the product is within 2^-52 of the quotient, which lies >= 2^-49 from every rounding
 * boundary of the 24-bit format unless it is representable); this program checks it by brute force.
 *   gcc -O2 -ffp-contract=off -o check_divr tools/check_divr.c -lm
 *   ./check_divr <d> [stride]      stride 1 = exhaustive (~15 s per divisor); prints the mismatch count */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int main(int argc, char **argv) {
  if (argc < 2) return 2;
  const float d = (float)atof(argv[1]);
  const uint64_t stride = argc > 2 ? (uint64_t)atoll(argv[2]) : 1;
  const double rd = 1.0 / (double)d;
  uint64_t bad = 0, n = 0;
  for (uint64_t b = 0; b <= 0xffffffffull; b += stride) {
    const uint32_t bb = (uint32_t)b;
    float x;
    memcpy(&x, &bb, 4);
    const float q = (float)((double)x * rd), r = x / d;
    n++;
    if (!(q == r || (isnan(q) && isnan(r)))) bad++;
    else if (q == 0.0f && signbit(q) != signbit(r)) bad++;
  }
  printf("d %.9g: %llu mismatches of %llu\n", d, (unsigned long long)bad, (unsigned long long)n);
  return bad != 0;
}
