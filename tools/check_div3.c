// exhaustive: x / 3.0f == fma(fma(-3, q, x), third, q) with q = x * third, for every finite float (div3_exact, csrc/fmm.hip); gcc -O2 -ffp-contract=off -mfma tools/check_div3.c -lm  (36 s; the one difference is -0 -> +0, never a traveltime)
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
int main(){
  const float third = 1.0f/3.0f;
  unsigned long long bad=0, n=0;
  for (uint64_t u=0; u<=0xffffffffull; u++){
    uint32_t b=(uint32_t)u; float x; memcpy(&x,&b,4);
    if (!isfinite(x)) continue;
    float q = x*third; float r = fmaf(-3.0f,q,x); float q2 = fmaf(r,third,q);
    float ref = x/3.0f;
    uint32_t a1,a2; memcpy(&a1,&q2,4); memcpy(&a2,&ref,4);
    n++;
    if (a1!=a2){ if (bad<10) printf("x=%a got %a want %a\n",x,q2,ref); bad++; }
  }
  printf("tested %llu bad %llu\n",n,bad); return 0;
}
