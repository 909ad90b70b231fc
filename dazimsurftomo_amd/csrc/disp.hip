// disp.hip -- K1 of SURVEY.md: Rayleigh fundamental-mode phase velocities (surfdisp96,
// inv/surfdisp96.f:52-1062) and their finite-difference depth kernels (depthkernel,
// inv/CalSurfG.f90:1-139) for every model column, on gfx950.
//
// Work item = (column, variant): variant 0 is the column itself, the other 6*nz are its +-0.5 %
// perturbations of one knot's Vs, Vp or rho.  One lane per work item; lanes of a wavefront run the
// reference's bracket + Neville root search as independent state machines that meet at the only
// expensive step, the Dunkin compound-matrix secular function (dltar4), which all lanes evaluate
// in lockstep (uniform trip count = number of layers).  Nothing per-lane is indexed dynamically:
// the layer stack is rebuilt on the fly from the column's knots (LDS, broadcast reads) with the
// lane's single perturbed knot patched in, and the Earth-flattening factors, which depend on the
// layer thicknesses only, come from a host table (same libm log/powf the CPU reference calls).
// fp64 VALU + transcendental bound; no MFMA, negligible HBM traffic.
#include <cmath>

#include "dazim_internal.h"

namespace {

constexpr int NL = 200;       // inv/surfdisp96.f:57
constexpr int NP = 60;        // inv/surfdisp96.f:59
constexpr int NZMAX = 64;     // knots per column we accept
constexpr int DT = 256;       // threads per workgroup
constexpr int TW = 64;        // work items per task: one wavefront's (see disp_kernel)
constexpr int TEAM = 16;      // lanes of a work item in the wide bracket search (DispArgs::team)
constexpr int NEVN = 11;      // Neville points kept (x(1..11), inv/surfdisp96.f:569,655)

struct Layer {   // per refined layer, geometry only
  double tmp;    // (ar+ar)/(r0+r1), sphere() :528
  float d;       // flattened thickness, :524
  float rfac;    // btp**(-2.275), :541
  float fm, den; // (2j-1), 2*nsublay of refineGrid2LayerMdl (inv/CalSurfG.f90:2352)
  int iv;        // interval (1-based knot index i); 0 for the half-space
  float rden;    // 1/den rounded to nearest (RDEN kernels)
};

struct DispArgs {
  int ncol, nz, kmax, nvar, mmax;
  int team;            // lanes per work item: 1, or TEAM when an item's bracket search visits TEAM grid points at a time (see disp_kernel)
  int var0, nvarp;     // the variants this launch works on: var0 .. var0 + nvarp - 1 of every column (all of them: 0, nvar; with
                       // option disp.async the column's own curve, 0 / 1, and its perturbed copies, 1 / nvar - 1, are two launches)
  const float *vel;  // [nz][ncol]
  const Layer *lay;  // [mmax]
  const double *t;   // [kmax]
  float *cg;         // [ncol*nvar][kmax]
  // task queue (see disp_kernel): groups of DT work items x chunks of `pchunk` consecutive periods
  int ngroup, nchunk, pchunk;
  unsigned *counter;   // next task
  int *ready;          // [nchunk][ngroup]: chunk c of group g is finished
  double *st_c;        // [ncol*nvar] root of the last finished period (c1 at :297), carried to the next chunk
  double *st_d;        // [ncol*nvar] del1st (the SAVEd first secular value of getsol, :409,424)
  int *st_f;           // [ncol*nvar] 1: the search failed, the remaining periods are 0 (:342-348)
  // first-period fast-forward (see disp_bracket_kernel): per column, the number of dc steps from the start value of the
  // first period's bracket search to the lower end of the bracket, and that lower end
  int ffwd;            // 0 off; 1 the column's own model jumps (exact: every skipped point was evaluated); 2 its perturbed copies too (gated)
  int exp3;            // 1: the three exponentials of a layer by three exp() calls like the reference (option disp.exp3)
  unsigned *ff_stat;   // [4] statistics of mode 2: columns whose copies may not jump (gate), columns with a dip limit, copies that
                       // jumped, copies that found another sign at the arrival point and went back to the start
  int *ff_m;           // [ncol]  0: no information
  double *ff_c;        // [ncol]
  int *ff_v;           // [ncol]  steps the perturbed copies may jump at most (see disp_bracket_kernel: a dip of |del| before the bracket)
  // central differences formed by the launch itself (the copies' launch of disp.async: var0 = 1, an even number of variants per
  // column, so the -/+ pair of a perturbation sits in neighbouring lanes of one wavefront); null: disp_finalize forms them
  double *svs, *svp, *srho;   // [nz][kmax][ncol]
  // ... and pvRc by the launch of the columns' own curves (var0 = 0, one variant per column); null: disp_finalize
  double *pv;                 // [kmax][ncol]
  int *nfail;                 // periods without a root (counted where pv is written)
};

__device__ __forceinline__ double sgn(double x) { return copysign(1.0, x); }

// 1/b and a/b for normal-range operands: hardware reciprocal estimate + Newton steps in explicit FMAs (8-11 instructions instead
// of the ~30 of the IEEE sequence with its scaling and fix-up).  Accurate to an ulp, not correctly rounded: used only inside the
// secular function, whose values feed a root search with tolerance 1e-6 (see the note above dltar4).
__device__ __forceinline__ double frcp(double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double fdiv(double a, double b) {
  const double r = frcp(b);
  const double q = a * r;
  return __builtin_fma(__builtin_fma(-b, q, a), r, q);
}

// sqrt for the normal-range, non-negative operands of the secular function: reciprocal-square-root estimate + two coupled
// Newton steps (9 instructions instead of the ~25 of the correctly rounded sequence with its range scaling); accurate to an
// ulp like frcp/fdiv above.
// rs receives 1/sqrt(x) (the coupled iteration carries it along; one more step brings it to an ulp), which replaces the
// divisions by the square root that follow.  x == 0 gives 0 and an unusable rs: the callers never divide in that case.
__device__ __forceinline__ double fsqrt(double x, double &rs) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  r = __builtin_fma(-g, g, x);
  g = __builtin_fma(r, h, g);
  r = __builtin_fma(-h, g, 0.5);
  h = __builtin_fma(h, r, h);
  rs = h + h;
  return x > 0.0 ? g : 0.0;
}

struct Knots {           // one lane's view of its column
  const float *vs, *vp, *rho;  // LDS, [nz] each
  int pi, pq;                  // perturbed knot (1-based) and quantity (0 vs, 1 vp, 2 rho); pi=0: none
  float pv;                    // perturbed value
  __device__ __forceinline__ float get(int q, int i) const {
    const float *b = q == 0 ? vs : (q == 1 ? vp : rho);
    const float v = b[i - 1];
    return (i == pi && q == pq) ? pv : v;
  }
};

// flattened layer m (1-based): Vp a, Vs b, density rho, thickness d as surfdisp96 holds them after
// refineGrid2LayerMdl + sphere(0,0) + sphere(2,1)
// The three fp32 divisions x/den per layer (den = 2*nsublay, a small even integer) without dividing, bit-identical:
// RDEN = 1: every den is a power of two (the usual sublayers = 3 gives 8), so x/den == x*(1/den) exactly;
// RDEN = 2: q = x*r, q' = fma(fma(-den, q, x), r, q) with r = RN(1/den) equals RN(x/den) for every float x with
//           1e-30 <= |x| <= 1e30 or x = 0 -- checked exhaustively over all 1.7e9 such x for each den in {6, 10, 12, 14, 18, 20,
//           22, 24, 26, 28, 30, 36} (host: fastdiv_ok) -- and `fast` (wavefront-uniform) says that this lane's and its
//           neighbours' knot differences are in that range; otherwise the division is done.
// RDEN = 0: divide.
template <int RDEN>
__device__ __forceinline__ void layer_model(const Knots &K, const Layer *lay, int m, int nz, bool fast, float &a,
                                            float &b, float &rho, float &d) {
  const Layer L = lay[m - 1];
  float rvp, rvs, rrho;
  if (L.iv > 0) {
    const int i = L.iv;
    const float p0 = K.get(1, i), p1 = K.get(1, i + 1);
    const float s0 = K.get(0, i), s1 = K.get(0, i + 1);
    const float r0 = K.get(2, i), r1 = K.get(2, i + 1);
    if (RDEN == 1) {
      rvp = p0 + L.fm * (p1 - p0) * L.rden;
      rvs = s0 + L.fm * (s1 - s0) * L.rden;
      rrho = r0 + L.fm * (r1 - r0) * L.rden;
    } else if (RDEN == 2 && fast) {
      const float tp = L.fm * (p1 - p0), ts = L.fm * (s1 - s0), tr = L.fm * (r1 - r0);
      float qp = tp * L.rden, qs = ts * L.rden, qr = tr * L.rden;
      qp = __builtin_fmaf(__builtin_fmaf(-L.den, qp, tp), L.rden, qp);
      qs = __builtin_fmaf(__builtin_fmaf(-L.den, qs, ts), L.rden, qs);
      qr = __builtin_fmaf(__builtin_fmaf(-L.den, qr, tr), L.rden, qr);
      rvp = p0 + qp;
      rvs = s0 + qs;
      rrho = r0 + qr;
    } else {
      rvp = p0 + L.fm * (p1 - p0) / L.den;
      rvs = s0 + L.fm * (s1 - s0) / L.den;
      rrho = r0 + L.fm * (r1 - r0) / L.den;
    }
  } else {
    rvp = K.get(1, nz);
    rvs = K.get(0, nz);
    rrho = K.get(2, nz);
  }
  a = (float)((double)rvp * L.tmp);
  b = (float)((double)rvs * L.tmp);
  rho = rrho * L.rfac;
  d = L.d;
}

// inv/surfdisp96.f:767-865 with var (:868-985), dnka (:1018-1062) and normc (:989-1014) inlined;
// llw = 1 (no water layer).  normc's log() is dead in the reference and dropped.
// fp64 divisions are ~30 instructions each on CDNA, so three groups of them are replaced by cheaper forms that
// differ from the reference's arithmetic only by fp64 rounding noise (the root is searched to 1e-6 relative and
// rounded to fp32): the normalisation of the compound vector multiplies by the reciprocal of its largest entry
// instead of dividing five times, 1/rho and 1/rho^2 are formed once per layer, and fb/omega uses the reciprocal of
// omega hoisted out of the layer loop; the remaining divisions of the layer loop use frcp/fdiv above.
template <int RDEN>
__device__ double dltar4(const Knots &K, const Layer *lay, int mmax, int nz, bool fast, double wvno, double omga, bool exp3 = false) {
#pragma clang fp contract(fast)   // FMA contraction inside the secular function only (the file is built with -ffp-contract=off)
  double e0, e1, e2, e3, e4;
  double omega = omga;
  if (omega < 1.0e-4) omega = 1.0e-4;
  const double wvno2 = wvno * wvno;
  const double romega = 1.0 / omega;
  float fa, fb, frho, fd;
  layer_model<RDEN>(K, lay, mmax, nz, fast, fa, fb, frho, fd);
  {
    const double xka = omega / (double)fa, xkb = omega / (double)fb;
    double wvnop = wvno + xka, wvnom = fabs(wvno - xka);
    const double ra = sqrt(wvnop * wvnom);
    wvnop = wvno + xkb;
    wvnom = fabs(wvno - xkb);
    const double rb = sqrt(wvnop * wvnom);
    const double t = (double)fb / omega;
    const double gammk = 2.0 * t * t;
    const double gam = gammk * wvno2;
    const double gamm1 = gam - 1.0;
    const double rho1 = (double)frho;
    e0 = rho1 * rho1 * (gamm1 * gamm1 - gam * gammk * ra * rb);
    e1 = -rho1 * ra;
    e2 = rho1 * (gamm1 - gammk * ra * rb);
    e3 = rho1 * rb;
    e4 = wvno2 - ra * rb;
  }
  for (int m = mmax - 1; m >= 1; m--) {
    layer_model<RDEN>(K, lay, m, nz, fast, fa, fb, frho, fd);
    const double xka = fdiv(omega, (double)fa), xkb = fdiv(omega, (double)fb);
    const double t = (double)fb * romega;
    const double gammk = 2.0 * t * t;
    const double gam = gammk * wvno2;
    double wvnop = wvno + xka, wvnom = fabs(wvno - xka);
    double rra;
    const double ra = fsqrt(wvnop * wvnom, rra);
    wvnop = wvno + xkb;
    wvnom = fabs(wvno - xkb);
    double rrb;
    const double rb = fsqrt(wvnop * wvnom, rrb);
    const double dpth = (double)fd, rho1 = (double)frho;
    const double rrho1 = frcp(rho1), rrho2 = rrho1 * rrho1;
    const double p = ra * dpth, q = rb * dpth;
    double w, x, y, z, cosp, cosq, sinp, sinq, fac, pex = 0.0, sex = 0.0;
    // The three exponentials of a layer, exp(-2p), exp(-2q) and exp(-(pex+sex)) (:893,:927,:951), come from two: ep = exp(-pex),
    // eq = exp(-sex) (1 for an oscillatory part), then ep*ep, eq*eq and ep*eq -- fp64 rounding noise inside the secular
    // function, like the reciprocals above.
    if (wvno > xka) pex = p;
    if (wvno > xkb) sex = q;
    const double ep = pex > 0.0 ? exp(-pex) : 1.0, eq = sex > 0.0 ? exp(-sex) : 1.0;
    if (wvno < xka) {
      sincos(p, &sinp, &cosp);   // one argument reduction for both
      w = sinp * rra;
      x = -ra * sinp;
    } else if (wvno == xka) {
      cosp = 1.0;
      w = dpth;
      x = 0.0;
    } else {
      fac = 0.0;
      if (p < 16) fac = exp3 ? exp(-2.0 * p) : ep * ep;
      cosp = (1.0 + fac) * 0.5;
      sinp = (1.0 - fac) * 0.5;
      w = sinp * rra;
      x = ra * sinp;
    }
    if (wvno < xkb) {
      sincos(q, &sinq, &cosq);
      y = sinq * rrb;
      z = -rb * sinq;
    } else if (wvno == xkb) {
      cosq = 1.0;
      y = dpth;
      z = 0.0;
    } else {
      fac = 0.0;
      if (q < 16) fac = exp3 ? exp(-2.0 * q) : eq * eq;
      cosq = (1.0 + fac) * 0.5;
      sinq = (1.0 - fac) * 0.5;
      y = sinq * rrb;
      z = rb * sinq;
    }
    const double exa = pex + sex;
    double a0 = 0.0;
    if (exa < 60.0) a0 = exp3 ? exp(-exa) : ep * eq;
    const double cpcq = cosp * cosq, cpy = cosp * y, cpz = cosp * z, cqw = cosq * w, cqx = cosq * x;
    const double xy = x * y, xz = x * z, wy = w * y, wz = w * z;
    const double gamm1 = gam - 1.0;
    const double twgm1 = gam + gamm1, gmgmk = gam * gammk, gmgm1 = gam * gamm1, gm1sq = gamm1 * gamm1;
    const double rho2 = rho1 * rho1, a0pq = a0 - cpcq;
    const double ca11 = cpcq - 2.0 * gmgm1 * a0pq - gmgmk * xz - wvno2 * gm1sq * wy;
    const double ca12 = (wvno2 * cpy - cqx) * rrho1;
    const double ca13 = -(twgm1 * a0pq + gammk * xz + wvno2 * gamm1 * wy) * rrho1;
    const double ca14 = (cpz - wvno2 * cqw) * rrho1;
    const double ca15 = -(2.0 * wvno2 * a0pq + xz + wvno2 * wvno2 * wy) * rrho2;
    const double ca21 = (gmgmk * cpz - gm1sq * cqw) * rho1;
    const double ca22 = cpcq;
    const double ca23 = gammk * cpz - gamm1 * cqw;
    const double ca24 = -wz;
    const double ca25 = ca14;
    const double ca41 = (gm1sq * cpy - gmgmk * cqx) * rho1;
    const double ca42 = -xy;
    const double ca43 = gamm1 * cpy - gammk * cqx;
    const double ca44 = ca22;
    const double ca45 = ca12;
    const double ca51 = -(2.0 * gmgmk * gm1sq * a0pq + gmgmk * gmgmk * xz + gm1sq * gm1sq * wy) * rho2;
    const double ca52 = ca41;
    const double ca53 = -(gammk * gamm1 * twgm1 * a0pq + gam * gammk * gammk * xz + gamm1 * gm1sq * wy) * rho1;
    const double ca54 = ca21;
    const double ca55 = ca11;
    const double tt = -2.0 * wvno2;
    const double ca31 = tt * ca53, ca32 = tt * ca43, ca33 = a0 + 2.0 * (cpcq - ca11), ca34 = tt * ca23,
                 ca35 = tt * ca13;
    // ee(i) = sum_j e(j)*ca(j,i), accumulated from 0 in the order j = 1..5 (:832-838)
    double n0 = 0.0 + e0 * ca11;  n0 = n0 + e1 * ca21;  n0 = n0 + e2 * ca31;  n0 = n0 + e3 * ca41;  n0 = n0 + e4 * ca51;
    double n1 = 0.0 + e0 * ca12;  n1 = n1 + e1 * ca22;  n1 = n1 + e2 * ca32;  n1 = n1 + e3 * ca42;  n1 = n1 + e4 * ca52;
    double n2 = 0.0 + e0 * ca13;  n2 = n2 + e1 * ca23;  n2 = n2 + e2 * ca33;  n2 = n2 + e3 * ca43;  n2 = n2 + e4 * ca53;
    double n3 = 0.0 + e0 * ca14;  n3 = n3 + e1 * ca24;  n3 = n3 + e2 * ca34;  n3 = n3 + e3 * ca44;  n3 = n3 + e4 * ca54;
    double n4 = 0.0 + e0 * ca15;  n4 = n4 + e1 * ca25;  n4 = n4 + e2 * ca35;  n4 = n4 + e3 * ca45;  n4 = n4 + e4 * ca55;
    // normc (:989-1014): scale by the reciprocal of the largest |entry| (one division instead of five; the scale must stay
    // a continuous function of c -- the root finder interpolates the returned values)
    double t1 = fmax(fmax(fmax(fabs(n0), fabs(n1)), fmax(fabs(n2), fabs(n3))), fabs(n4));
    if (t1 < 1.e-40) t1 = 1.0;
    const double r1 = frcp(t1);
    e0 = n0 * r1;
    e1 = n1 * r1;
    e2 = n2 * r1;
    e3 = n3 * r1;
    e4 = n4 * r1;
  }
  return e0;
}

// inv/surfdisp96.f:361-382, all fp32
__device__ float gtsolh(float a, float b) {
  float c = 0.95f * b;
  for (int i = 0; i < 5; i++) {
    const float gamma = b / a;
    const float kappa = c / b;
    const float k2 = kappa * kappa;
    const float gk2 = (gamma * kappa) * (gamma * kappa);
    const float fac1 = sqrtf(1.0f - gk2);
    const float fac2 = sqrtf(1.0f - k2);
    const float fr = (2.0f - k2) * (2.0f - k2) - 4.0f * fac1 * fac2;
    float frp = -4.0f * (2.0f - k2) * kappa + 4.0f * fac2 * gamma * gamma * kappa / fac1 + 4.0f * fac1 * kappa / fac2;
    frp = frp / b;
    c = c - fr / frp;
  }
  return c;
}

__device__ __forceinline__ void brocher(float vs, float &vp, float &rho) {  // inv/CalSurfG.f90:49-53
  const float v2 = vs * vs, v3 = v2 * vs, v4 = v3 * vs;
  const float p = 0.9409f + 2.0947f * vs - 0.8206f * v2 + 0.2683f * v3 - 0.0251f * v4;
  const float p2 = p * p, p3 = p2 * p, p4 = p3 * p, p5 = p4 * p;
  vp = p;
  rho = 1.6612f * p - 0.4721f * p2 + 0.0671f * p3 - 0.0043f * p4 + 0.000106f * p5;
}

enum { P_G1, P_G2, P_N0, P_NA, P_NB, P_GV, P_GW, P_DONE };

// start-up of surfdisp96 (:134-216): extremal velocities of the layer stack and the start value of the first period's search
template <int RDEN>
__device__ __forceinline__ void startup(const Knots &K, const Layer *lay, int mmax, int nz, bool fast, float &betmx, float &cc1) {
  float betmn = 1.e20f, a_mn = 1.0f, b_mn = 1.0f;
  betmx = -1.e20f;
  int jsol = 1;
  for (int m = 1; m <= mmax; m++) {
    float fa, fb, fr, fd;
    layer_model<RDEN>(K, lay, m, nz, fast, fa, fb, fr, fd);
    if (fb > 0.01f && fb < betmn) {
      betmn = fb;
      a_mn = fa;
      b_mn = fb;
      jsol = 1;
    } else if (fb <= 0.01f && fa < betmn) {
      betmn = fa;
      a_mn = fa;
      b_mn = fb;
      jsol = 0;
    }
    if (fb > betmx) betmx = fb;
  }
  cc1 = jsol == 0 ? betmn : gtsolh(a_mn, b_mn);
  cc1 = .95f * cc1;
  cc1 = .90f * cc1;
}

// The bracket search of the FIRST period starts far below the root (0.855 x the Rayleigh velocity of the slowest layer) and walks
// up in steps of dc = 0.005 km/s: ~100 secular evaluations against ~9 for each later period, 40 % of all the work of an item with
// 16 periods -- and a strictly sequential chain only in its stopping rule: the grid points c_j = c_{j-1} + dc do not depend on
// the function values.  This kernel evaluates them 64 at a time for the column's own model (one wavefront per column, lane j =
// point j), exactly as the reference would one after the other, and records where the first sign change is: m = j* - 1 steps
// from the start to the lower end of the bracket, and that lower end.  disp_kernel then starts an item's first period by
// evaluating the start point (del1st), jumping m steps ahead by the same sequence of additions, evaluating there, and going on
// as the reference does from that point:
//   * for the column's own model nothing is assumed -- every skipped point was evaluated here, with the same function;
//   * the 6*nz perturbed copies (one knot changed by +-0.5 %) jump to 0.02 km/s (four steps) below the column's bracket and
//     check that the sign there is still the start point's.  A perturbed root lies within ~0.003 km/s of the column's (at most
//     0.015 if one knot carried all the sensitivity), so the jump stays below it; if the sign has changed, the item goes back to
//     the start point and searches step by step.  What the sign check cannot see is an even number of sign changes inside the
//     skipped interval: two roots of the perturbed model where the column's model has none.  That happens where the column's
//     secular function touches zero without crossing it (a double root about to be born: 4 of 1 200 columns whose knots are
//     drawn at random, none in layered models with any smoothness), and it shows in the samples this kernel has anyway: the
//     normalised |del| sits on a plateau and falls monotonically into the bracket, except in those columns, where it dips and
//     comes back.  So the kernel also records where |del| first decreases and whether it ever increases again before the
//     bracket; if it does, the perturbed copies jump only to eight steps before that first decrease and walk through the dip
//     step by step like the reference (ff_v).  What remains after that are features of the perturbed copy's function narrower
//     than the grid step that the column's own samples miss: 2 of 28 800 columns whose ten knots are drawn independently from
//     2.6 .. 4.7 km/s, both with a knot more than 36 % slower than a shallower one (stacks of trapped-wave channels); none in
//     57 600 columns of a gradient with independent knot perturbations of up to +-30 %.  The perturbed copies of a column
//     therefore jump only if no knot is more than FF_MAXDROP = 25 % slower than the fastest knot above it; rougher columns
//     are searched step by step.  tests/test_disp_gpu.py compares the jump with the step-by-step search bit for bit on
//     ordinary models, on graded random columns (jump active) and on thousands of such rough columns (gate and dip guard);
//     option disp.ffwd = 0 turns the jump off.
constexpr int FF_BLOCKS = 8;          // 8 x 64 grid points = 2.56 km/s above the start value
constexpr double FF_MARGIN = 0.02;    // km/s below the column's bracket for the perturbed copies
constexpr float FF_MAXDROP = 0.25f;   // largest relative decrease of Vs below a shallower knot for which the perturbed copies jump
template <int RDEN>
__global__ __launch_bounds__(64) void disp_bracket_kernel(DispArgs A) {
  __shared__ Layer s_lay[NL];
  __shared__ float s_knot[3 * NZMAX];
  const int lane = threadIdx.x, col = blockIdx.x;
  const int nz = A.nz, mmax = A.mmax;
  for (int i = lane; i < mmax; i += 64) s_lay[i] = A.lay[i];
  for (int k = lane; k < nz; k += 64) {
    const float vs = A.vel[(size_t)k * A.ncol + col];
    float vp, rho;
    brocher(vs, vp, rho);
    s_knot[k] = vs;
    s_knot[nz + k] = vp;
    s_knot[2 * nz + k] = rho;
  }
  __syncthreads();
  Knots K;
  K.vs = s_knot;
  K.vp = s_knot + nz;
  K.rho = s_knot + 2 * nz;
  K.pi = 0;
  K.pq = -1;
  K.pv = 0.0f;
  bool fast = false;
  if (RDEN == 2) {
    bool ok = true;
    for (int i = 1; i < nz; i++)
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const float ad = fabsf(K.get(q, i + 1) - K.get(q, i));
        ok = ok && (ad == 0.0f || (ad >= 1.0e-30f && ad <= 1.0e28f));
      }
    fast = __all(ok);
  }
  float betmx, cc1;
  startup<RDEN>(K, s_lay, mmax, nz, fast, betmx, cc1);
  const double dc = fabs((double)0.005f), TWOPI = 2.0 * 3.141592653589793;
  const double omega = TWOPI / A.t[0];
  const double climit = (double)betmx + dc;      // a visited point at or above it ends the reference's search (:470)
  double cblk = (double)cc1;                     // grid point 64 * blk
  unsigned long long prev_last = 0;
  int found = 0;
  double cfound = 0.0, clast_prev = 0.0;           // clast_prev: grid point 64 * blk - 1
  double alast_prev = 0.0;                         // |del| at grid point 64 * blk - 1
  int jfirst = -1;                                 // first grid point at which |del| is smaller than at the point before
  bool wiggle = false;                             // ... and it grew again somewhere between there and the bracket
  for (int blk = 0; blk < FF_BLOCKS; blk++) {
    double c = cblk;
    for (int i = 0; i < lane; i++) c = c + dc;   // the reference's c2 = c1 + dc, one step after the other
    const double del = dltar4<RDEN>(K, s_lay, mmax, nz, fast, omega / c, omega, A.exp3 != 0);
    const unsigned long long neg = __ballot(sgn(del) < 0.0);
    const unsigned long long before = (neg << 1) | (blk > 0 ? prev_last : (neg & 1ull));
    const unsigned long long chg = neg ^ before;                     // bit j: the sign changes between points j-1 and j
    const unsigned long long over = __ballot(c >= climit);           // points the reference would not go beyond
    const int jl = over ? __builtin_ctzll(over) : 64;
    const double clast = __shfl(c, 63);
    {   // monotonicity of |del| up to the bracket (points of this block below the sign change, or all 64)
      const double a = fabs(del);
      double ap = __shfl_up(a, 1);
      if (lane == 0) ap = blk > 0 ? alast_prev : a;
      const int jend = (chg != 0 && __builtin_ctzll(chg) <= jl) ? __builtin_ctzll(chg) : 64;   // points < jend are below the bracket
      const unsigned long long below = jend >= 64 ? ~0ull : ((1ull << jend) - 1ull);
      const unsigned long long dec = __ballot(a < ap * (1.0 - 1.0e-6)) & below;
      const unsigned long long inc = __ballot(a > ap * (1.0 + 1.0e-6)) & below;
      unsigned long long after = ~0ull;             // points after the first decrease
      if (jfirst < 0 && dec != 0) {
        const int jd = __builtin_ctzll(dec);
        jfirst = 64 * blk + jd;
        after = jd >= 63 ? 0ull : ~((2ull << jd) - 1ull);
      } else if (jfirst < 0) {
        after = 0ull;
      }
      if ((inc & after) != 0) wiggle = true;
      alast_prev = __shfl(a, 63);
    }
    if (chg != 0) {
      const int j = __builtin_ctzll(chg);
      const double cl = __shfl(c, j > 0 ? j - 1 : 0);
      if (j <= jl) {
        found = 64 * blk + j;
        cfound = j > 0 ? cl : clast_prev;                          // lower end of the bracket = point j-1
      }
      break;
    }
    if (over != 0) break;
    prev_last = neg >> 63;
    clast_prev = clast;
    cblk = clast + dc;
  }
  if (lane == 0) {
    A.ff_m[col] = found > 0 ? found - 1 : 0;
    A.ff_c[col] = cfound;
    float vtop = s_knot[0], drop = 0.0f;             // roughness gate: how much slower than the fastest knot above is any knot?
    for (int k = 1; k < nz; k++) {
      const float v = s_knot[k];
      drop = fmaxf(drop, (vtop - v) / vtop);
      vtop = fmaxf(vtop, v);
    }
    int lim = wiggle ? (jfirst - 8 > 0 ? jfirst - 8 : 0) : 0x3fffffff;   // (no dip: no extra limit)
    if (drop > FF_MAXDROP) lim = 0;
    A.ff_v[col] = lim;
    if (A.ffwd == 2) {
      if (drop > FF_MAXDROP) atomicAdd(&A.ff_stat[0], 1u);
      else if (wiggle) atomicAdd(&A.ff_stat[1], 1u);
    }
  }
}

// Scheduling.  A work item's periods chain (the root of period k seeds the search of period k+1, :262-266), so an item is a
// long serial job: ~15 secular evaluations x layers x periods, and a launch whose workgroups do not fill a whole number of
// rounds pays one item's latency for the stragglers (S-256: 832 workgroups on 768 slots: 56 ms for the first round + 16 ms for
// the 64 left over).  Persistent workgroups pull tasks = (group of DT items, chunk of `pchunk` consecutive periods) from an
// atomic counter, chunk-major; the state an item carries from one chunk to the next is two doubles (the last root and del1st)
// and a fail flag in HBM, and task (g, c) waits for (g, c-1), which was handed out a whole generation earlier (no deadlock:
// tasks only wait for earlier tasks).  Measured on S-256 (tools/disp_only.py): chunks of 2 / 4 / 8 periods 67.1 / 69.0 / 72.6 ms
// against 69.0 ms unchunked -- the lanes of a wavefront run their periods independently and only meet at the end of a task,
// so every chunk boundary adds the wait for the slowest lane (+15 % at one round's worth of work) and eats what the shorter
// tail saves.  The default is therefore one chunk (pchunk = kmax); option "disp.pchunk" selects shorter ones (bit-identical
// results: tests/test_disp_gpu.py).
#ifdef DZ_DISP_STAT
__device__ unsigned long long g_disp_stat[4];
#endif
template <int RDEN, bool TEAMS = false>
__global__ __launch_bounds__(DT, 3) void disp_kernel(DispArgs A) {
  __shared__ Layer s_lay[NL];
  __shared__ double s_t[NP];
  extern __shared__ __attribute__((aligned(16))) float s_knots[];  // [DT / 64][cpb][3][nz]
  __shared__ double s_x[NEVN][DT], s_y[NEVN][DT];
  const int tid = threadIdx.x, lane = tid & 63;
  const int nz = A.nz, kmax = A.kmax, mmax = A.mmax, nvar = A.nvar, nvarp = A.nvarp;
  const long nwork = (long)A.ncol * nvarp;
  for (int i = tid; i < mmax; i += DT) s_lay[i] = A.lay[i];
  for (int i = tid; i < kmax; i += DT) s_t[i] = A.t[i];
  __syncthreads();
  // Tasks are taken per WAVEFRONT (round 3, late): groups of TW = 64 items.  The four wavefronts of a workgroup share the layer
  // table and the periods, nothing else, and with 256-item tasks each waited for the slowest of the four at every task
  // boundary -- 8.3 % of the lane-evaluations a workgroup offered went unused, 3.4 % of those a wavefront offers
  // (tools/disp_stat.sh).  No barrier inside the loop: a wavefront's LDS accesses execute in order.
  // Teams (the column curves of a launch that leaves most of the chip idle: small batches with disp.async).  An item's chain is
  // ~33 secular evaluations per period one after the other, and two thirds of them are the bracket search walking up from below
  // the last root in steps of dc -- a sequence of points that does not depend on the function values (c2 = c1 +- dc, turned round
  // at clow, :437-452), only its END does.  With team = TEAM an item is held by TEAM lanes with identical state; when the search
  // advances, lane t evaluates the t-th point of that sequence, the first sign change (or bound violation) among them is taken
  // exactly as the step-by-step search would have met it, the rest is discarded, and the Neville iteration runs redundantly on
  // all lanes.  ~12 instead of ~33 evaluation latencies per period; bit-identical curves (tests/test_disp_gpu.py).
  const int team = TEAMS ? TEAM : 1, tw = TW / team;       // lanes per item, items per task (A.team says which instantiation runs)
  const int tl = lane & (team - 1), tsh = lane & ~(team - 1);
  const int cpb = (tw + nvarp - 1) / nvarp + 1;  // columns the work items of one group may span
  float *s_knot = s_knots + (size_t)(tid >> 6) * cpb * 3 * nz;
  const unsigned ntask = (unsigned)A.ngroup * (unsigned)A.nchunk;
  for (;;) {
    __builtin_amdgcn_wave_barrier();   // (the previous task's LDS reads are over)
    unsigned task = 0;
    if (lane == 0) {
      task = atomicAdd(A.counter, 1u);
      if (task < ntask && task >= (unsigned)A.ngroup) {   // chunk c > 0: its predecessor (same group, chunk c-1) must have published its state
        const int *flag = A.ready + (task - (unsigned)A.ngroup);
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(20);
      }
    }
    task = (unsigned)__shfl((int)task, 0);
    if (task >= ntask) break;
    __threadfence();   // acquire side for the lanes that did not spin: the state written by another workgroup is visible
    const int chunk = (int)(task / (unsigned)A.ngroup), grp = (int)(task - (unsigned)chunk * (unsigned)A.ngroup);
    const int kbeg = chunk * A.pchunk, kend = min(kbeg + A.pchunk, kmax);
    const long w0 = (long)grp * tw;
    const int col0 = (int)(w0 / nvarp);
    for (int i = lane; i < cpb * nz; i += TW) {
      const int c = i / nz, k = i - c * nz, col = col0 + c;
      if (col < A.ncol) {
        const float vs = A.vel[(size_t)k * A.ncol + col];
        float vp, rho;
        brocher(vs, vp, rho);
        s_knot[(c * 3 + 0) * nz + k] = vs;
        s_knot[(c * 3 + 1) * nz + k] = vp;
        s_knot[(c * 3 + 2) * nz + k] = rho;
      }
    }
    __builtin_amdgcn_wave_barrier();
    const long w = w0 + (team > 1 ? lane / team : lane);
    const bool active = w < nwork;
    const int col = active ? (int)(w / nvarp) : col0;
    const int var = active ? A.var0 + (int)(w - (long)col * nvarp) : 0;
    Knots K;
    K.vs = s_knot + ((col - col0) * 3 + 0) * nz;
    K.vp = s_knot + ((col - col0) * 3 + 1) * nz;
    K.rho = s_knot + ((col - col0) * 3 + 2) * nz;
    K.pi = 0;
    K.pq = -1;
    K.pv = 0.0f;
    if (var > 0) {  // depthkernel's perturbation order: knot i, then (vs,vp,rho), then (-,+)  (:76-124)
      const int v1 = var - 1;
      K.pi = v1 / 6 + 1;
      const int r = v1 - (K.pi - 1) * 6;
      K.pq = r >> 1;
      const float b0 = (K.pq == 0 ? K.vs : (K.pq == 1 ? K.vp : K.rho))[K.pi - 1];
      const float dln = 0.01f;
      K.pv = (r & 1) ? b0 + 0.5f * dln * b0 : b0 - 0.5f * dln * b0;
    }
    bool fast = false;
    if (RDEN == 2) {   // every knot difference the interpolation will see is 0 or inside the range the shortcut was verified on
      bool ok = true;
      for (int i = 1; i < nz; i++)
#pragma unroll
        for (int q = 0; q < 3; q++) {
          const float ad = fabsf(K.get(q, i + 1) - K.get(q, i));
          ok = ok && (ad == 0.0f || (ad >= 1.0e-30f && ad <= 1.0e28f));
        }
      fast = __all(ok);
    }
    // ---- start-up of surfdisp96 (:134-216): extremal velocities, half-space start value (recomputed per chunk: one pass
    // over the layers, the cost of a fraction of one secular evaluation) ----
    float betmx, cc1;
    startup<RDEN>(K, s_lay, mmax, nz, fast, betmx, cc1);
    const float ddc = 0.005f, sone = 1.5f;
    const double onea = (double)sone, TWOPI = 2.0 * 3.141592653589793;
    const double cc = (double)cc1, dc = fabs((double)ddc), cm = cc;
    const size_t wi = active ? (size_t)col * nvar + var : 0;   // the item's place among all (column, variant) pairs
    float *cg = A.cg + wi * kmax;

    // ---- per-lane root-search state (getsol :384-476, nevill :551-668) ----
    int k = kbeg, phase = active ? P_G1 : P_DONE, ifirst = 1, idir = 1, nev = 1, nctrl = 1, mm = 1;
    double c1 = cc, c2 = cc, del1 = 0, del2 = 0, del1st = 0, clow = cc, c3 = cc, del3 = 0, cprev = cc;
    if (chunk > 0 && active) {   // resume where the previous chunk stopped: the "next period" step of :262-266
      if (A.st_f[wi]) {
        phase = P_DONE;          // the search failed earlier: the remaining periods are already 0
      } else {
        cprev = A.st_c[wi];
        del1st = A.st_d[wi];
        ifirst = 0;
        c1 = cprev - onea * dc;
        clow = cm;
      }
    }
    double omega = TWOPI / s_t[kbeg];
    double ceval = c1;
    bool failed = false;
    double wc1 = 0.0, wc2 = 0.0;   // team mode: this lane's step of the wide bracket search (its c1 and c2)
    int wid = 1;                   // ... and the direction after it

#ifdef DZ_DISP_STAT
    unsigned long long st_iter = 0, st_act = 0;
#endif
    while (__any(phase != P_DONE)) {
#ifdef DZ_DISP_STAT
      st_iter++;
      if (phase != P_DONE) st_act++;
#endif
      const double del = dltar4<RDEN>(K, s_lay, mmax, nz, fast, omega / ceval, omega, A.exp3 != 0);
      if (phase == P_DONE) continue;
      bool advance_bracket = false, nev_top = false, nev_body = false, finish = false, fail = false;
      switch (phase) {
        case P_G1:
          del1 = del;
          if (ifirst == 1) del1st = del1;
          idir = (ifirst == 1) ? 1 : (sgn(del1st) * sgn(del1) >= 0.0 ? 1 : -1);
          advance_bracket = true;
          if (A.ffwd && ifirst == 1 && chunk == 0) {   // first period: jump ahead to the bracket found by disp_bracket_kernel
            int m = A.ff_m[col];
            if (var > 0 && A.ffwd < 2) m = 0;          // (default: only the column's own model, for which the jump is exact)
            if (m > 0 && var > 0) {
              const double steps = floor((A.ff_c[col] - FF_MARGIN - c1) / dc);
              m = steps > 0.0 ? (steps < (double)m + 8.0 ? (int)steps : m + 8) : 0;
              const int lim = A.ff_v[col] - 4;      // (a dip of the column's |del| before its bracket: stay in front of it)
              if (m > lim) m = lim > 0 ? lim : 0;
            }
            if (m >= 2) {
              s_x[0][tid] = c1;                        // (the Neville table is idle during the bracket search)
              for (int i = 0; i < m; i++) c1 = c1 + dc;   // the same additions the step-by-step search would have made
              ceval = c1;
              phase = P_GV;
              advance_bracket = false;
              if (var > 0) atomicAdd(&A.ff_stat[2], 1u);
            }
          }
          break;
        case P_GV:   // arrival point of the jump: same sign as the start point -> go on from here, else back to the start
          if (sgn(del) == sgn(del1))
            del1 = del;
          else {
            c1 = s_x[0][tid];
            if (var > 0) atomicAdd(&A.ff_stat[3], 1u);
          }
          advance_bracket = true;
          break;
        case P_G2:
          del2 = del;
          if (sgn(del1) != sgn(del2)) {  // root bracketed -> nevill: first half()
            c3 = 0.5 * (c1 + c2);
            ceval = c3;
            phase = P_N0;
            nev = 1;
            nctrl = 1;
            mm = 1;
          } else {
            c1 = c2;
            del1 = del2;
            if (c1 < cm || c1 >= ((double)betmx + dc))
              fail = true;
            else
              advance_bracket = true;
          }
          break;
        case P_GW: if (TEAMS) {   // TEAM points of the bracket search at once (see above): the first event among them, in the search's order
          const unsigned neg = (unsigned)(__ballot(sgn(del) < 0.0) >> tsh) & ((1u << TEAM) - 1u);
          const unsigned prevneg = ((neg << 1) | (sgn(del1) < 0.0 ? 1u : 0u)) & ((1u << TEAM) - 1u);
          const unsigned chg = neg ^ prevneg;                                     // bit j: sgn(del1) != sgn(del2) at step j
          const unsigned bad = (unsigned)(__ballot(wc2 < cm || wc2 >= ((double)betmx + dc)) >> tsh) & ((1u << TEAM) - 1u);
          const int jchg = chg ? __builtin_ctz(chg) : TEAM, jbad = bad ? __builtin_ctz(bad) : TEAM;
          if (jchg < TEAM && jchg <= jbad) {      // step jchg brackets the root (its bound test is never reached) -> nevill
            const double dprev = __shfl(del, tsh + (jchg > 0 ? jchg - 1 : 0));
            if (jchg > 0) del1 = dprev;
            c1 = __shfl(wc1, tsh + jchg);
            c2 = __shfl(wc2, tsh + jchg);
            del2 = __shfl(del, tsh + jchg);
            idir = __shfl(wid, tsh + jchg);
            c3 = 0.5 * (c1 + c2);
            ceval = c3;
            phase = P_N0;
            nev = 1;
            nctrl = 1;
            mm = 1;
          } else if (jbad < TEAM) {               // step jbad moved c1 out of bounds before any sign change
            c1 = __shfl(wc2, tsh + jbad);
            fail = true;
          } else {                                // no event: go on from the last point
            c1 = __shfl(wc2, tsh + TEAM - 1);
            del1 = __shfl(del, tsh + TEAM - 1);
            idir = __shfl(wid, tsh + TEAM - 1);
            advance_bracket = true;
          }
        }
          break;
        case P_N0:
        case P_NB:
          del3 = del;
          nev_top = true;
          break;
        case P_NA:
          del3 = del;
          nev_body = true;
          break;
      }
      if (TEAMS && advance_bracket) {   // lane tl of the team: the (tl+1)-th point the step-by-step search would visit from here
        double a = c1;
        int id = idir;
        for (int i = 0; i <= tl; i++) {
          for (;;) {
            wc2 = (id > 0) ? a + dc : a - dc;
            if (wc2 <= clow) {
              id = 1;
              a = clow;
              continue;
            }
            break;
          }
          wc1 = a;          // the c1 this step works with (clow if the search was turned round in it)
          a = wc2;          // c1 = c2 for the next step
        }
        wid = id;
        ceval = wc2;
        phase = P_GW;
      } else if (!TEAMS && advance_bracket) {
        for (;;) {
          c2 = (idir > 0) ? c1 + dc : c1 - dc;
          if (c2 <= clow) {
            idir = 1;
            c1 = clow;
            continue;
          }
          break;
        }
        ceval = c2;
        phase = P_G2;
      }
      if (nev_top) {
        nctrl++;
        if (nctrl >= 100)
          finish = true;
        else if (c3 < fmin(c1, c2) || c3 > fmax(c1, c2)) {
          nev = 0;
          c3 = 0.5 * (c1 + c2);
          ceval = c3;
          phase = P_NA;
        } else
          nev_body = true;
      }
      if (nev_body) {
        const double s13 = del1 - del3, s32 = del3 - del2;
        if (sgn(del3) * sgn(del1) < 0.0) {
          c2 = c3;
          del2 = del3;
        } else {
          c1 = c3;
          del1 = del3;
        }
        if (fabs(c1 - c2) <= 1.e-6 * c1)
          finish = true;
        else {
          if (sgn(s13) != sgn(s32)) nev = 0;
          const double ss1 = fabs(del1), s1 = (double)0.01f * ss1;
          const double ss2 = fabs(del2), s2 = (double)0.01f * ss2;
          if (s1 > ss2 || s2 > ss1 || nev == 0) {
            c3 = 0.5 * (c1 + c2);
            nev = 1;
            mm = 1;
          } else {
            if (nev == 2) {
              s_x[mm][tid] = c3;
              s_y[mm][tid] = del3;
            } else {
              s_x[0][tid] = c1;
              s_y[0][tid] = del1;
              s_x[1][tid] = c2;
              s_y[1][tid] = del2;
              mm = 1;
            }
            bool bad = false;
            const double ym = s_y[mm][tid];
            for (int kk = 1; kk <= mm; kk++) {
              const int j = mm - kk + 1;
              const double yj = s_y[j - 1][tid];
              const double denom = ym - yj;
              if (fabs(denom) < 1.0e-10 * fabs(ym)) {
                bad = true;
                break;
              }
              s_x[j - 1][tid] = (-yj * s_x[j][tid] + ym * s_x[j - 1][tid]) / denom;
            }
            if (!bad) {
              c3 = s_x[0][tid];
              nev = 2;
              mm = mm + 1;
              if (mm > 10) mm = 10;
            } else {
              c3 = 0.5 * (c1 + c2);
              nev = 1;
              mm = 1;
            }
          }
          ceval = c3;
          phase = P_NB;
        }
      }
      if (finish) {
        c1 = c3;
        if (c1 > (double)betmx)
          fail = true;
        else {
          cg[k] = (float)c1;  // cg(k) = sngl(c(k)), :292-297
          cprev = c1;
          k++;
          if (k == kend)
            phase = P_DONE;   // chunk finished (k == kmax: the item is finished)
          else {  // next period, :262-266
            ifirst = 0;
            c1 = cprev - onea * dc;
            clow = cm;
            omega = TWOPI / s_t[k];
            ceval = c1;
            phase = P_G1;
          }
        }
      }
      if (fail) {  // :1750-1770
        for (int i = k; i < kmax; i++) cg[i] = 0.0f;
        failed = true;
        phase = P_DONE;
      }
    }
#ifdef DZ_DISP_STAT
    {   // lane-evaluations done / offered by the wavefront (tasks are per wavefront: a workgroup offers what its wavefronts do)
      atomicAdd(&g_disp_stat[0], st_act);
      if (lane == 0) { atomicAdd(&g_disp_stat[1], st_iter * 64ull); atomicAdd(&g_disp_stat[2], st_iter * 64ull); }
    }
#endif
    // ---- the central differences of depthkernel (inv/CalSurfG.f90:90-133) as this launch's epilogue: lanes 2j / 2j + 1 of the
    // wavefront hold the - / + copy of one perturbation (knot K.pi, quantity K.pq) with all their periods finished -- this task's
    // by this wavefront, earlier chunks' visible since the acquire at the top of the task.  Same arithmetic as disp_finalize:
    // ((double)cg2 - (double)cg1) / (double)(0.01f * base), cg the fp32-rounded phase velocities.
    if (A.pv && kend == kmax && active && var == 0 && tl == 0) {   // pvRc = the column's own curve (inv/CalSurfG.f90:60)
      int nzero = 0;
      for (int kk = 0; kk < kmax; kk++) {
        const float c0 = cg[kk];
        A.pv[(size_t)kk * A.ncol + col] = (double)c0;
        nzero += c0 == 0.0f ? 1 : 0;
      }
      if (nzero) atomicAdd(A.nfail, nzero);
    }
    if (A.svs && kend == kmax && team == 1) {
      double *const out = K.pq == 0 ? A.svs : (K.pq == 1 ? A.svp : A.srho);
      const float base = active && var > 0 ? (K.pq == 0 ? K.vs : (K.pq == 1 ? K.vp : K.rho))[K.pi - 1] : 1.0f;
      const double den = (double)(0.01f * base);
      for (int kk = 0; kk < kmax; kk++) {
        const float mine = active ? cg[kk] : 0.0f;
        const float other = __shfl_xor(mine, 1);
        if (active && !(lane & 1)) out[((size_t)(K.pi - 1) * kmax + kk) * A.ncol + col] = ((double)other - (double)mine) / den;
      }
    }
    if (active && kend < kmax && !(chunk > 0 && A.st_f[wi])) {   // hand the chain to the next chunk
      A.st_c[wi] = cprev;
      A.st_d[wi] = del1st;
      A.st_f[wi] = failed ? 1 : 0;
    }
    __threadfence();   // release: the state above before the flag below
    __builtin_amdgcn_wave_barrier();
    if (lane == 0 && kend < kmax) __hip_atomic_store(A.ready + task, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// pvRc and the central differences of depthkernel (inv/CalSurfG.f90:60,90-133)
__global__ void disp_finalize(int ncol, int nz, int kmax, int nvar, const float *__restrict__ vel,
                              const float *__restrict__ cg, double *__restrict__ pv, double *__restrict__ svs,
                              double *__restrict__ svp, double *__restrict__ srho, int *nfail) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (long)ncol * kmax) return;
  const int k = (int)(tid / ncol), col = (int)(tid - (long)k * ncol);
  const float *c = cg + (size_t)col * nvar * kmax;
  if (pv) {
    const float c0 = c[k];
    pv[(size_t)k * ncol + col] = (double)c0;
    if (c0 == 0.0f) atomicAdd(nfail, 1);
  }
  if (!svs) return;
  const float dln = 0.01f;
  for (int i = 0; i < nz; i++) {
    const float vs = vel[(size_t)i * ncol + col];
    float vp, rho;
    brocher(vs, vp, rho);
    const float base[3] = {vs, vp, rho};
    double *out[3] = {svs, svp, srho};
    for (int q = 0; q < 3; q++) {
      const float cg1 = c[(size_t)(1 + i * 6 + q * 2) * kmax + k], cg2 = c[(size_t)(1 + i * 6 + q * 2 + 1) * kmax + k];
      out[q][((size_t)i * kmax + k) * ncol + col] = ((double)cg2 - (double)cg1) / (double)(dln * base[q]);
    }
  }
}

}  // namespace

// = depthkernel (inv/CalSurfG.f90:1); kernels==0 -> only pvRc (CalRayleighPhase behaviour,
// fwd/FwdTraveltimeCPS.f90:4)
extern "C" int dazim_dispersion_kernels(dazim_ctx *ctx, int nx, int ny, int nz, const float *vel_u,
                                        const float *depz, float minthk0, int kmax, const double *periods,
                                        double *pv_u, double *svs_u, double *svp_u, double *srho_u, int *n_failed) {
  if (!ctx || !vel_u || !depz || !periods || !pv_u) return dz_fail(ctx, DAZIM_E_BAD_ARG, "null argument");
  if (nz < 2 || nz > NZMAX || kmax < 1 || kmax > NP || nx < 1 || ny < 1) return dz_fail(ctx, DAZIM_E_BAD_ARG, "bad nz/kmax");
  DZ_HIP(hipSetDevice(ctx->device));
  const bool kernels = svs_u && svp_u && srho_u;
  const int ncol = nx * ny, nvar = kernels ? 1 + 6 * nz : 1;
  // ---- geometry-only layer table: refineGrid2LayerMdl (inv/CalSurfG.f90:2339-2365) + sphere (:510-545) ----
  std::vector<Layer> lay;
  for (int i = 1; i <= nz - 1; i++) {
    const float thk = depz[i] - depz[i - 1];
    const float minthk = thk / minthk0;
    const int nsub = (int)((thk + 1.0e-4f) / minthk) + 1;
    const float newthk = thk / (float)nsub;
    for (int j = 1; j <= nsub; j++) {
      Layer L;
      L.d = newthk;
      L.iv = i;
      L.fm = (float)(2 * j - 1);
      L.den = (float)(2 * nsub);
      lay.push_back(L);
    }
  }
  {
    Layer L;
    L.d = 0.0f;
    L.iv = 0;
    L.fm = 0;
    L.den = 1;
    lay.push_back(L);
  }
  // division-free layer interpolation (see layer_model): 1 = every 2*nsub a power of two, 2 = every 2*nsub a power of two or one
  // of the divisors the reciprocal + correction form was verified on exhaustively, 0 = divide
  int rden = 1;
  for (Layer &L : lay) {
    const int d = (int)L.den;
    L.rden = 1.0f / L.den;
    const bool pow2 = (d & (d - 1)) == 0 && d <= (1 << 20);
    bool fastdiv_ok = false;
    for (int v : {6, 10, 12, 14, 18, 20, 22, 24, 26, 28, 30, 36}) fastdiv_ok = fastdiv_ok || d == v;
    if (!pow2) rden = (fastdiv_ok && rden != 0) ? 2 : 0;
  }
  if (ctx->opts.count("disp.rden") && !ctx->opts["disp.rden"]) rden = 0;   // test knob: keep the divisions
  const int mmax = (int)lay.size();
  if (mmax > NL) return dz_fail(ctx, DAZIM_E_BAD_ARG, "refined model has %d layers > NL=%d", mmax, NL);
  {
    const double ar = 6370.0;
    double dr = 0.0, r0 = ar;
    lay[mmax - 1].d = 1.0f;
    for (int i = 0; i < mmax; i++) {
      dr = dr + (double)lay[i].d;
      const double r1 = ar - dr;
      const double z0 = ar * log(ar / r0), z1 = ar * log(ar / r1);
      lay[i].d = (float)(z1 - z0);
      const double tmp = (ar + ar) / (r0 + r1);
      lay[i].tmp = tmp;
      lay[i].rfac = powf((float)tmp, -2.275f);
      r0 = r1;
    }
    lay[mmax - 1].d = 0.0f;
  }
  DzBuf<float> vel;
  DzBuf<double> pv, svs, svp, srho;
  int rc;
  if ((rc = dz_fmm_finish(ctx))) return rc; // (an asynchronous eikonal call nobody has collected)
  if ((rc = dz_join_aux(ctx))) return rc;   // (an earlier call's perturbed copies may still be running on the auxiliary stream)
  const size_t nk = (size_t)nz * kmax * ncol;
  if ((rc = vel.init(ctx, vel_u, (size_t)nz * ncol, true, false))) return rc;
  if ((rc = pv.init(ctx, pv_u, (size_t)kmax * ncol, false, true))) return rc;
  if (kernels) {
    if ((rc = svs.init(ctx, svs_u, nk, false, true))) return rc;
    if ((rc = svp.init(ctx, svp_u, nk, false, true))) return rc;
    if ((rc = srho.init(ctx, srho_u, nk, false, true))) return rc;
  }
  void *p;
  DispArgs A;
  A.svs = A.svp = A.srho = A.pv = nullptr;
  A.nfail = nullptr;
  A.ncol = ncol;
  A.nz = nz;
  A.kmax = kmax;
  A.nvar = nvar;
  A.var0 = 0;
  A.nvarp = nvar;
  A.team = 1;
  A.mmax = mmax;
  A.vel = vel.dev;
  if ((rc = dz_scratch(ctx, "disp.lay", sizeof(Layer) * NL, &p))) return rc;
  A.lay = (Layer *)p;
  DZ_HIP(hipMemcpyAsync(p, lay.data(), sizeof(Layer) * mmax, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = dz_scratch(ctx, "disp.t", sizeof(double) * NP, &p))) return rc;
  A.t = (double *)p;
  DZ_HIP(hipMemcpyAsync(p, periods, sizeof(double) * kmax, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = dz_scratch(ctx, "disp.cg", sizeof(float) * (size_t)ncol * nvar * kmax, &p))) return rc;
  A.cg = (float *)p;
  if ((rc = dz_scratch(ctx, "disp.nfail", 16, &p))) return rc;
  int *d_nfail = (int *)p;
  DZ_HIP(hipMemsetAsync(d_nfail, 0, 4, ctx->stream));
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  auto knot_lds = [&](int nvarp, int items = TW) { return (size_t)(DT / TW) * ((items + nvarp - 1) / nvarp + 1) * 3 * nz * sizeof(float); };
  // Option disp.async (device-resident vel and sen_* -- pvRc may be a host array, it is complete and copied when the call
  // returns --, depth kernels wanted, one period chunk): the column's own curves -- all the
  // eikonal solve needs -- are one launch on the context's stream, the 6*nz perturbed copies (72/73 of the work, wanted only by
  // the G rows) another one on the auxiliary stream, and the call returns when the first is done.  dazim_rays_build_G*, the next
  // dazim_dispersion_kernels, dazim_sync and dazim_free join the auxiliary stream; anybody else who reads sen_* (or overwrites
  // vel) before one of these calls dazim_sync first.  What it buys: the dispersion kernel's last, partly filled round of
  // workgroups (S-256: 64 of 832) and the eikonal kernel share the chip (tools/exp_overlap.py: 293 against 306 ms), and on
  // small batches (S-128) the two kernels, neither of which fills it, run side by side.
  bool async = kernels && ctx->opts.count("disp.async") && ctx->opts["disp.async"] && !vel.staged && !svs.staged &&
               !svp.staged && !srho.staged && !(ctx->opts.count("disp.pchunk") && ctx->opts["disp.pchunk"] > 0 && ctx->opts["disp.pchunk"] < kmax);
  // the column curves of an asynchronous call in teams (disp_kernel: TEAM grid points of the bracket search at a time) when the
  // extra wavefronts do not add a round of workgroups to the copies beside them: everything fits one round (S-128), or the copies
  // need a second, partly filled round anyway (S-256: 832 + 183 workgroups on 768 slots; the curves are what the eikonal kernel
  // waits for, 23.5 -> 12 ms, step 370 -> 366 ms).  test4_Yunnan (0.89 of a round, 1.01 with teams: 62 -> 25 ms of column curves,
  // but the copies beside them 139 -> 163 ms) stays without.  Option disp.team = 1 / 2 forces them on / off.
  const double wg_copies = (double)(((long)ncol * (nvar - 1) + DT - 1) / DT), wg_teams = (double)((ncol * TEAM + DT - 1) / DT);
  const double wg_round = (double)((long)ctx->num_cu * 3);
  bool teams = wg_copies + wg_teams <= 0.98 * wg_round || (wg_copies > 1.02 * wg_round && wg_copies + wg_teams <= 1.9 * wg_round);
  if (ctx->opts.count("disp.team") && ctx->opts["disp.team"] == 1) teams = true;
  if (ctx->opts.count("disp.team") && ctx->opts["disp.team"] == 2) teams = false;
  const size_t dyn_lds = knot_lds(nvar), dyn_lds_base = teams ? knot_lds(1, TW / TEAM) : knot_lds(1);
  if (dyn_lds_base + 56 * 1024 > 160 * 1024) async = false;   // (very many knots: the 64 columns of a base task would not fit the LDS)
  const size_t dyn_max = async && dyn_lds_base > dyn_lds ? dyn_lds_base : dyn_lds;
  DZ_HIP(hipFuncSetAttribute((const void *)disp_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_max));
  DZ_HIP(hipFuncSetAttribute((const void *)disp_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_max));
  DZ_HIP(hipFuncSetAttribute((const void *)disp_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_max));
  DZ_HIP(hipFuncSetAttribute((const void *)disp_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_max));
  DZ_HIP(hipFuncSetAttribute((const void *)disp_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_max));
  DZ_HIP(hipFuncSetAttribute((const void *)disp_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_max));
  if (async && (rc = dz_aux_init(ctx))) return rc;
  {
    const long nwork = (long)ncol * nvar;
    // task queue: groups of DT items x chunks of pchunk periods (see disp_kernel); persistent workgroups, as many as are resident
    A.ngroup = (int)((nwork + TW - 1) / TW);
    A.pchunk = kmax;
    if (ctx->opts.count("disp.pchunk") && ctx->opts["disp.pchunk"] > 0) A.pchunk = ctx->opts["disp.pchunk"];
    if (A.pchunk > kmax) A.pchunk = kmax;
    A.nchunk = (kmax + A.pchunk - 1) / A.pchunk;
    const size_t ntask = (size_t)A.ngroup * A.nchunk;
    if ((rc = dz_scratch(ctx, "disp.ready", ntask * 4 + 64, &p))) return rc;
    A.ready = (int *)p;
    A.counter = (unsigned *)((char *)p + ntask * 4 + 16 - (ntask * 4) % 16);
    DZ_HIP(hipMemsetAsync(p, 0, ntask * 4 + 64, ctx->stream));
    if ((rc = dz_scratch(ctx, "disp.st_c", (size_t)nwork * 8, &p))) return rc;
    A.st_c = (double *)p;
    if ((rc = dz_scratch(ctx, "disp.st_d", (size_t)nwork * 8, &p))) return rc;
    A.st_d = (double *)p;
    if ((rc = dz_scratch(ctx, "disp.st_f", (size_t)nwork * 4, &p))) return rc;
    A.st_f = (int *)p;
    const void *kf = rden == 1 ? (const void *)disp_kernel<1> : (rden == 2 ? (const void *)disp_kernel<2> : (const void *)disp_kernel<0>);
    int occ = 3;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kf, DT, dyn_lds) != hipSuccess || occ < 1) occ = 1;
    // option disp.occ: at most that many workgroups per CU (room for a kernel of another stream on the same CUs)
    if (ctx->opts.count("disp.occ") && ctx->opts["disp.occ"] >= 1 && ctx->opts["disp.occ"] < occ) occ = ctx->opts["disp.occ"];
    long nwg = (long)ctx->num_cu * occ;
    if (nwg > ((long)ntask + DT / TW - 1) / (DT / TW)) nwg = ((long)ntask + DT / TW - 1) / (DT / TW);
    // first-period fast-forward (disp_bracket_kernel); off with option disp.ffwd = 0 and when the periods are handed from task to task
    // Option disp.ffwd: 1 (default) = the jump for the column's own model only, which is exact -- disp_bracket_kernel evaluated
    // every skipped grid point with the same function; 2 = the 6*nz perturbed copies jump as well, behind the gates described at
    // disp_bracket_kernel (equal to the step-by-step search on every model family tried, tools/stress_disp_ffwd.py, but a
    // statistical statement, not a proof: hence opt-in, with its statistics under dazim_last_kernel_seconds("disp.ffwd_*"));
    // 0 = no jump.
    A.ffwd = 0;
    if (A.nchunk == 1) {
      A.ffwd = 1;
      if (ctx->opts.count("disp.ffwd")) A.ffwd = ctx->opts["disp.ffwd"] < 0 ? 0 : (ctx->opts["disp.ffwd"] > 2 ? 2 : ctx->opts["disp.ffwd"]);
    }
    A.exp3 = ctx->opts.count("disp.exp3") && ctx->opts["disp.exp3"] ? 1 : 0;
    if ((rc = dz_scratch(ctx, "disp.ff_stat", 64, &p))) return rc;
    A.ff_stat = (unsigned *)p;
    DZ_HIP(hipMemsetAsync(A.ff_stat, 0, 64, ctx->stream));
    if ((rc = dz_scratch(ctx, "disp.ff_m", (size_t)ncol * 4 + 16, &p))) return rc;
    A.ff_m = (int *)p;
    if ((rc = dz_scratch(ctx, "disp.ff_c", (size_t)ncol * 8 + 16, &p))) return rc;
    A.ff_c = (double *)p;
    if ((rc = dz_scratch(ctx, "disp.ff_v", (size_t)ncol * 4 + 16, &p))) return rc;
    A.ff_v = (int *)p;
    if (async && ctx->opts["disp.async"] == 1) {
      // 1 = two streams where they pay: what the eikonal kernel can share the chip with is the copies' last, partly filled round
      // (or a launch that never fills it); over many rounds that is a small part of the launch and the two kernels only take
      // each other's LDS (S-512, four rounds: eikonal kernel +0.31 s for 0.25 s of dispersion).  2 = always.
      const double rounds = (double)(((long)ncol * (nvar - 1) + DT - 1) / DT) / (double)((long)ctx->num_cu * occ);
      if (rounds > 2.0) async = false;
    }
    ctx->ksec["disp.async"] = async ? 1.0 : 0.0;
    ctx->aux_timed = false;
    DzTimer t(ctx, "disp");
    if (A.ffwd) {
      if (rden == 1)
        hipLaunchKernelGGL(disp_bracket_kernel<1>, dim3((unsigned)ncol), dim3(64), 0, ctx->stream, A);
      else if (rden == 2)
        hipLaunchKernelGGL(disp_bracket_kernel<2>, dim3((unsigned)ncol), dim3(64), 0, ctx->stream, A);
      else
        hipLaunchKernelGGL(disp_bracket_kernel<0>, dim3((unsigned)ncol), dim3(64), 0, ctx->stream, A);
      DZ_HIP(hipGetLastError());
    }
    auto launch = [&](const DispArgs &B, long nwgB, size_t lds, hipStream_t st) {
      if (B.team > 1) {
        if (rden == 1)
          hipLaunchKernelGGL((disp_kernel<1, true>), dim3((unsigned)nwgB), dim3(DT), lds, st, B);
        else if (rden == 2)
          hipLaunchKernelGGL((disp_kernel<2, true>), dim3((unsigned)nwgB), dim3(DT), lds, st, B);
        else
          hipLaunchKernelGGL((disp_kernel<0, true>), dim3((unsigned)nwgB), dim3(DT), lds, st, B);
      } else if (rden == 1)
        hipLaunchKernelGGL(disp_kernel<1>, dim3((unsigned)nwgB), dim3(DT), lds, st, B);
      else if (rden == 2)
        hipLaunchKernelGGL(disp_kernel<2>, dim3((unsigned)nwgB), dim3(DT), lds, st, B);
      else
        hipLaunchKernelGGL(disp_kernel<0>, dim3((unsigned)nwgB), dim3(DT), lds, st, B);
    };
    const long nf = (long)ncol * kmax;
    if (!async) {
      launch(A, nwg, dyn_lds, ctx->stream);
      DZ_HIP(hipGetLastError());
      hipLaunchKernelGGL(disp_finalize, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, ctx->stream, ncol, nz, kmax, nvar,
                         vel.dev, A.cg, pv.dev, kernels ? svs.dev : nullptr, kernels ? svp.dev : nullptr,
                         kernels ? srho.dev : nullptr, d_nfail);
      DZ_HIP(hipGetLastError());
    } else {
      // the copies: auxiliary stream, behind everything enqueued so far (tables, counters, the bracket kernel)
      DispArgs C = A;
      C.var0 = 1;
      C.nvarp = nvar - 1;
      C.ngroup = (int)(((long)ncol * C.nvarp + TW - 1) / TW);
      long nwgC = (long)ctx->num_cu * occ, needC = ((long)C.ngroup + DT / TW - 1) / (DT / TW);
      if (nwgC > needC) nwgC = needC;
      // the column's own curves: a handful of workgroups on the main stream, their own task counter
      DispArgs B = A;
      B.var0 = 0;
      B.nvarp = 1;
      B.team = teams ? TEAM : 1;
      B.pv = pv.dev;              // (written by the launch itself: no kernel behind it that waits for a free SIMD beside the copies)
      B.nfail = d_nfail;
      B.ngroup = (ncol + TW / B.team - 1) / (TW / B.team);
      ctx->ksec["disp.team"] = B.team;
      if ((rc = dz_scratch(ctx, "disp.ready_b", (size_t)B.ngroup * 4 + 64, &p))) return rc;
      B.ready = (int *)p;
      B.counter = (unsigned *)((char *)p + (size_t)B.ngroup * 4 + 16 - ((size_t)B.ngroup * 4) % 16);
      DZ_HIP(hipMemsetAsync(p, 0, (size_t)B.ngroup * 4 + 64, ctx->stream));
      DZ_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
      const long nwgB = ((long)B.ngroup + DT / TW - 1) / (DT / TW);
      launch(B, nwgB, dyn_lds_base, ctx->stream);   // (first: its workgroups want a CU's LDS before the copies' have filled them)
      DZ_HIP(hipGetLastError());
      DZ_HIP(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
      DZ_HIP(hipEventRecord(ctx->ev_a0, ctx->stream2));
      // (the copies' launch forms the central differences itself -- DispArgs::svs: a separate kernel behind it would wait for a free
      // SIMD until the eikonal kernel's persistent workgroups leave, 190 ms at S-256, profiles/r3_bench_kernel_stats.md)
      C.svs = svs.dev;
      C.svp = svp.dev;
      C.srho = srho.dev;
      launch(C, nwgC, dyn_lds, ctx->stream2);
      DZ_HIP(hipGetLastError());
      DZ_HIP(hipEventRecord(ctx->ev_a1, ctx->stream2));
      ctx->aux_pending = true;
      ctx->aux_timed = true;
      ctx->aux_ranges.clear();
      ctx->aux_ranges.push_back({(const char *)vel.dev, (size_t)nz * ncol * sizeof(float)});
      for (const double *q : {svs.dev, svp.dev, srho.dev})
        if (q) ctx->aux_ranges.push_back({(const char *)q, nk * sizeof(double)});
    }
    t.stop();
  }
  if ((rc = dz_pinned(ctx, "disp.host", 64, &p))) return rc;
  int &nfail = *(int *)p;                 // (pinned: a DMA transfer, not a blit kernel that queues behind the perturbed copies)
  unsigned *hst = (unsigned *)p + 4;
  nfail = 0;
  hst[0] = hst[1] = hst[2] = hst[3] = 0;
  DZ_HIP(hipMemcpyAsync(&nfail, d_nfail, 4, hipMemcpyDeviceToHost, ctx->stream));
#ifdef DZ_DISP_STAT
  {
    unsigned long long h[4];
    DZ_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_disp_stat), sizeof h));
    fprintf(stderr, "disp stat: lane-evaluations done %llu, offered by the wavefronts %llu (%.3f used), by the workgroups %llu (%.3f used)\n", h[0], h[1],
            (double)h[0] / (double)h[1], h[2], (double)h[0] / (double)h[2]);
    unsigned long long z[4] = {0};
    DZ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_disp_stat), z, sizeof z));
  }
#endif
  if (!ctx->aux_pending) DZ_HIP(hipMemcpyAsync(hst, A.ff_stat, 16, hipMemcpyDeviceToHost, ctx->stream));   // (async: the copies' statistics are not waited for)
  if ((rc = pv.finish()) || (rc = svs.finish()) || (rc = svp.finish()) || (rc = srho.finish())) return rc;
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  if (n_failed) *n_failed = nfail;
  ctx->ksec["disp.ffwd_mode"] = A.ffwd;
  ctx->ksec["disp.ffwd_gated_columns"] = hst[0];      // mode 2: columns whose perturbed copies may not jump (roughness gate)
  ctx->ksec["disp.ffwd_dip_columns"] = hst[1];        // ... columns whose copies stop in front of a dip of |del|
  ctx->ksec["disp.ffwd_jumped_copies"] = hst[2];      // ... perturbed copies that jumped
  ctx->ksec["disp.ffwd_fallback_copies"] = hst[3];    // ... of which the arrival point had the other sign (searched step by step)
  return 0;
}
