// ti.hip -- TI Rayleigh eigenfunction partials -> azimuthal depth kernels Lsen_Gsc on the device.
//
// = depthkernelTI (inv/depthkernelTI.f90:2-106) -> tregn96 (inv/tregn96.f:52-722), live branch only: Rayleigh,
//   fundamental mode, solid layers, flattened model (iflsph=1), causal-Q phase shift (Qp=150, Qs=50, fref=1 Hz),
//   source/receiver depth 0.  The phase velocities are an input (pvRc of dazim_dispersion_kernels, which runs the same
//   surfdisp96 on the same layered columns).
//
// One lane per (column, period).  fp64 throughout, like the reference.  Three sweeps over the layers of a lane:
//   up   : Dunkin compound vector from the half-space to the surface (up, :1834).  The 5x5 layer matrix of
//          dnka_tregn (:1992) is never formed: the row vector is pushed through C2(U)*C2(H)*C2(W), the Cauchy-Binet
//          factors of the layer's Haskell matrix A = U*H*W (hska, :3477), with C2(H) in closed form so that the
//          cosh^2 - sinh^2 cancellation stays analytic.  Re(cd(m,1:5)) and the exponent exe(m) go to HBM scratch.
//   down : Haskell vector from the surface down (down, :3561), eigenfunctions at the layer tops (svfunc, :1559)
//          and the layer energy integrals / unnormalised partials (energy, :3777; intijr, :4073), streamed.
//   out  : normalisation, gammap (:3711), sprayl (:1235), fp32 rounding (chksiz) and the depthkernelTI sum.
// HBM scratch is [layer][value][lane], so every access of a wavefront is one contiguous 512-byte run.
#include "dazim_internal.h"

namespace {

constexpr int NLMAX = 200;  // inv/tregn96.f: NL
constexpr int TT = 64;      // lanes per workgroup

struct cx { double re, im; };
__device__ __forceinline__ cx C(double re, double im) { return cx{re, im}; }
__device__ __forceinline__ cx operator+(cx a, cx b) { return C(a.re + b.re, a.im + b.im); }
__device__ __forceinline__ cx operator-(cx a, cx b) { return C(a.re - b.re, a.im - b.im); }
__device__ __forceinline__ cx operator-(cx a) { return C(-a.re, -a.im); }
__device__ __forceinline__ cx operator*(cx a, cx b) { return C(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
__device__ __forceinline__ cx operator*(cx a, double s) { return C(a.re * s, a.im * s); }
__device__ __forceinline__ cx operator/(cx a, cx b) {
  const double den = b.re * b.re + b.im * b.im;
  return C((a.re * b.re + a.im * b.im) / den, (a.im * b.re - a.re * b.im) / den);
}
__device__ __forceinline__ double cabs_(cx a) { return hypot(a.re, a.im); }
__device__ __forceinline__ cx csqrt_(cx a) {
  const double r = cabs_(a);
  if (r == 0.0) return C(0.0, 0.0);
  const double s = sqrt(0.5 * (r + fabs(a.re)));
  if (a.re >= 0.0) return C(s, a.im / (2.0 * s));
  return C(fabs(a.im) / (2.0 * s), a.im >= 0.0 ? s : -s);
}
__device__ __forceinline__ cx cexp_(cx a) {
  const double e = exp(a.re);
  double sn, cs;
  sincos(a.im, &sn, &cs);
  return C(e * cs, e * sn);
}

struct Eig { cx rp, rsv, x[4][2], np, nsv; };                       // gettiegn, inv/tregn96.f:3166
struct Trig { cx cosp, rsinp, sinpr, cosq, rsinq, sinqr; double pex, svex; };   // varsv, :3363

__device__ void layer_eig(double TA, double TC, double TF, double TL, double rho, double omg, double wvn, Eig &g) {
  const double wvno2 = wvn * wvn;
  const double a = wvn * TF / TC, b = 1.0 / TC, c = -rho * omg * omg + wvn * wvn * (TA - TF * TF / TC);
  const double d = -wvn, e = 1.0 / TL, f = -rho * omg * omg;
  const double ddef = wvn * wvn - rho * omg * omg / TL, aabc = wvn * wvn * TA / TC - rho * omg * omg / TC;
  const cx bb = C(2.0 * a * d + e * c + f * b, 0.0), cc = C(ddef * aabc, 0.0);
  cx s = csqrt_(bb * bb - cc * 4.0);
  if (s.im < 0.0) s = -s;
  cx l1, l2;
  if (bb.re < 0.0 && s.re < 0.0) {
    l2 = (bb - s) * 0.5;
    l1 = cabs_(l2) > 0.0 ? cc / l2 : (bb + s) * 0.5;
  } else {
    l1 = (bb + s) * 0.5;
    l2 = cabs_(l1) > 0.0 ? cc / l1 : (bb - s) * 0.5;
  }
  if (cabs_(C(wvno2, 0) - l2) < cabs_(C(wvno2, 0) - l1)) { const cx t = l1; l1 = l2; l2 = t; }
  g.rp = csqrt_(l1);
  g.rsv = csqrt_(l2);
  if (g.rp.re < 0.0) g.rp = -g.rp;
  if (g.rsv.re < 0.0) g.rsv = -g.rsv;
  cx x12 = C(b * d - a * e, 0), x22 = l2 * b - C(e * (b * c + a * a), 0);
  cx x32 = l2 - C(a * d + c * e, 0), x42 = l2 * (-a) + C(d * (b * c + a * a), 0);
  cx x11 = l1 * (-e) + C(b * (d * d + e * f), 0), x21 = C(b * d - a * e, 0);
  cx x31 = l1 * d - C(a * (d * d + e * f), 0), x41 = -(l1 - C(a * d + b * f, 0));
  if (wvn != 0.0) {
    cx z = C(wvn, 0) / x11;
    x11 = x11 * z; x21 = x21 * z; x31 = x31 * z; x41 = x41 * z;
    z = C(wvn, 0) / x22;
    x12 = x12 * z; x22 = x22 * z; x32 = x32 * z; x42 = x42 * z;
  }
  g.x[0][0] = x11; g.x[1][0] = x21; g.x[2][0] = x31; g.x[3][0] = x41;
  g.x[0][1] = x12; g.x[1][1] = x22; g.x[2][1] = x32; g.x[3][1] = x42;
  g.np = x11 * x41 - x21 * x31;
  g.nsv = x12 * x42 - x22 * x32;
}

__device__ void layer_trig(const Eig &g, double dm, Trig &t) {
  const cx p = g.rp * dm, q = g.rsv * dm;
  t.pex = p.re;
  t.svex = q.re;
#pragma unroll
  for (int w = 0; w < 2; w++) {
    const cx arg = w ? q : p, nu = w ? g.rsv : g.rp;
    double sn, cs;
    sincos(arg.im, &sn, &cs);
    const cx epp = C(cs / 2.0, sn / 2.0), epm = C(epp.re, -epp.im);
    const double fac = arg.re < 15.0 ? exp(-2.0 * arg.re) : 0.0;
    const cx co = epp + epm * fac, si = epp - epm * fac;
    const cx rs = nu * si;
    const cx sr = (fabs(arg.re) < 1.0e-5 && cabs_(nu) < 1.0e-5) ? C(dm, 0) : si / nu;
    if (w) { t.cosq = co; t.rsinq = rs; t.sinqr = sr; } else { t.cosp = co; t.rsinp = rs; t.sinpr = sr; }
  }
}

// A = U * blockdiag(H1/np, H2/nsv) * W with U columns / W rows ordered (1a,1b,2a,2b):
//   U(1,ma)=x1m U(3,ma)=x3m U(2,mb)=x2m U(4,mb)=x4m ; W(ma,:) = [x4m,0,-x2m,0], W(mb,:) = [0,-x3m,0,x1m]
__device__ __forceinline__ cx Uel(const Eig &g, int i, int k) {
  const int md = k >> 1, b = k & 1;
  return ((i & 1) == b) ? g.x[i][md] : C(0, 0);
}
__device__ __forceinline__ cx Wel(const Eig &g, int k, int j) {
  const int md = k >> 1, b = k & 1;
  if ((j & 1) != b) return C(0, 0);
  if (!b) return j == 0 ? g.x[3][md] : -g.x[1][md];
  return j == 1 ? -g.x[2][md] : g.x[0][md];
}

// ee = cd * CA(layer) without forming CA: 6->5 reduction of :2937-2961 folded into the 6-vectors
__device__ void push_up(const Eig &g, const Trig &t, const cx (&cd)[5], cx (&ee)[5]) {
  constexpr int P0[6] = {0, 0, 0, 1, 1, 2}, P1[6] = {1, 2, 3, 2, 3, 3}, IDX[5] = {0, 1, 2, 4, 5};
  const double ex = t.pex + t.svex, dfac = ex > 35.0 ? 0.0 : exp(-ex);
  cx a6[6], b6[6], c6[6];
#pragma unroll
  for (int r = 0; r < 6; r++) a6[r] = C(0, 0);
#pragma unroll
  for (int j = 0; j < 5; j++) a6[IDX[j]] = cd[j] * (j == 2 ? 2.0 : 1.0);
  // b6 = a6 * C2(U)
#pragma unroll
  for (int c = 0; c < 6; c++) {
    cx s = C(0, 0);
#pragma unroll
    for (int r = 0; r < 6; r++) {
      const cx m = Uel(g, P0[r], P0[c]) * Uel(g, P1[r], P1[c]) - Uel(g, P0[r], P1[c]) * Uel(g, P1[r], P0[c]);
      s = s + a6[r] * m;
    }
    b6[c] = s;
  }
  // c6 = b6 * C2(H): det blocks (analytic) and H1 (x) H2
  const cx h1[2][2] = {{t.cosp / g.np, t.sinpr / g.np}, {t.rsinp / g.np, t.cosp / g.np}};
  const cx h2[2][2] = {{t.cosq / g.nsv, t.rsinq / g.nsv}, {t.sinqr / g.nsv, t.cosq / g.nsv}};
  c6[0] = b6[0] * (C(dfac, 0) / (g.np * g.np));
  c6[5] = b6[5] * (C(dfac, 0) / (g.nsv * g.nsv));
#pragma unroll
  for (int k = 0; k < 2; k++)
#pragma unroll
    for (int l = 0; l < 2; l++) {
      cx s = C(0, 0);
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) s = s + b6[1 + 2 * i + j] * (h1[i][k] * h2[j][l]);
      c6[1 + 2 * k + l] = s;
    }
  // ee = (c6 * C2(W)) restricted to the five kept pairs, minus the dfac of CA(3,3)
#pragma unroll
  for (int jo = 0; jo < 5; jo++) {
    const int c = IDX[jo];
    cx s = C(0, 0);
#pragma unroll
    for (int r = 0; r < 6; r++) {
      const cx m = Wel(g, P0[r], P0[c]) * Wel(g, P1[r], P1[c]) - Wel(g, P0[r], P1[c]) * Wel(g, P1[r], P0[c]);
      s = s + c6[r] * m;
    }
    ee[jo] = s;
  }
  ee[2] = ee[2] - cd[2] * dfac;
}

// rows of E^-1 and columns of E (evalg, :3027-3076)
__device__ void layer_E(const Eig &g, cx (&E)[4][4], cx (&EI)[4][4]) {
  const cx rp = g.rp, rsv = g.rsv;
  const cx x11 = g.x[0][0], x21 = g.x[1][0], x31 = g.x[2][0], x41 = g.x[3][0];
  const cx x12 = g.x[0][1], x22 = g.x[1][1], x32 = g.x[2][1], x42 = g.x[3][1];
  const cx dp = (rp * g.np) * 2.0, ds = (rsv * g.nsv) * 2.0;
  EI[0][0] = (x41 * rp) / dp; EI[0][1] = (-x31) / dp; EI[0][2] = (-(x21 * rp)) / dp; EI[0][3] = x11 / dp;
  EI[1][0] = x42 / ds; EI[1][1] = (-(x32 * rsv)) / ds; EI[1][2] = (-x22) / ds; EI[1][3] = (x12 * rsv) / ds;
  EI[2][0] = (-(x41 * rp)) / (-dp); EI[2][1] = (-x31) / (-dp); EI[2][2] = (x21 * rp) / (-dp); EI[2][3] = x11 / (-dp);
  EI[3][0] = x42 / (-ds); EI[3][1] = (x32 * rsv) / (-ds); EI[3][2] = (-x22) / (-ds); EI[3][3] = (-(x12 * rsv)) / (-ds);
  E[0][0] = x11; E[1][0] = x21 * rp; E[2][0] = x31; E[3][0] = x41 * rp;
  E[0][1] = x12 * rsv; E[1][1] = x22; E[2][1] = x32 * rsv; E[3][1] = x42;
  E[0][2] = x11; E[1][2] = -(x21 * rp); E[2][2] = x31; E[3][2] = -(x41 * rp);
  E[0][3] = -(x12 * rsv); E[1][3] = x22; E[2][3] = -(x32 * rsv); E[3][3] = x42;
}

// :1385-1474
__device__ cx ffunc(cx nu, double dm) {
  if (cabs_(nu) < 1.0e-8) return C(dm, 0);
  const cx arg = nu * dm;
  const cx ex = arg.re < 40.0 ? cexp_(arg * (-2.0)) : C(0, 0);
  return (C(1, 0) - ex) / (nu * 2.0);
}
__device__ cx gfunc(cx nu, double dm) {
  const cx arg = nu * dm;
  return arg.re < 75.0 ? cexp_(-arg) * dm : C(0, 0);
}
__device__ cx h1func(cx na, cx nb, double dm) {
  if (cabs_(nb + na) < 1.0e-8) return C(dm, 0);
  const cx arg = (na + nb) * dm;
  const cx ex = arg.re < 40.0 ? cexp_(-arg) : C(0, 0);
  return (C(1, 0) - ex) / (nb + na);
}
__device__ cx h2func(cx na, cx nb, double dm) {
  if (cabs_(nb - na) < 1.0e-8) return C(dm, 0);
  cx arg = na * dm;
  const cx exp_ = arg.re < 40.0 ? cexp_(-arg) : C(0, 0);
  arg = nb * dm;
  const cx exq = arg.re < 40.0 ? cexp_(-arg) : C(0, 0);
  return (exq - exp_) / (na - nb);
}

struct TiArgs {
  int ncol, nz, kmax, mmax;
  int prio;                // 1: ti_kernel raises its wavefronts' issue priority (the dispersion copies of an asynchronous call are pending)
  const float *vel;        // [nz][ncol]
  const double *pv;        // [kmax][ncol]
  const float *twopi_t;    // [kmax] fp32 periods (t_in)
  // geometry-only layer tables [mmax]
  const int *knot;         // upper knot (0-based) of each refined layer, -1 for the half-space
  const float *fm, *den;   // (2j-1), 2*nsublay
  const float *thk;        // refined thickness (fp32), 0 for the half-space
  const double *zd;        // flattened thickness as tregn96 uses it (zd(mmax)=1 after bldsph)
  const double *pw_rho, *pw_el;  // tmp**(-2.275), tmp**(-0.275) of sphere_tdisp96
  const float *vtp;        // bldsph
  const int *jlay;         // inversion layer (0-based) of each refined layer
  // per-column model [mmax][ncol]
  double *zta, *ztl, *ztf, *zrho;
  float *fTA, *fTL, *fTF, *frho, *fvp, *fvs;
  // per-lane scratch [mmax][6|4][nlane]
  double *upv, *prt;
  float *lsen;             // [nz-1][kmax][ncol]
  int *bad;                // fluid layer seen
};

__device__ __forceinline__ void brocher(float vs, float &vp, float &rho) {  // inv/depthkernelTI.f90:48-53
  vp = 0.9409f + 2.0947f * vs - 0.8206f * (vs * vs) + 0.2683f * (vs * vs * vs) - 0.0251f * (vs * vs * vs * vs);
  rho = 1.6612f * vp - 0.4721f * (vp * vp) + 0.0671f * (vp * vp * vp) - 0.0043f * (vp * vp * vp * vp) +
        0.000106f * (vp * vp * vp * vp * vp);
}

// per column: knots -> layers (refineLayerMdl, inv/CalSurfGAniso_Joint.f90:146), TI moduli (depthkernelTI :71-78) and
// their flattened fp64 copies (sphere_tdisp96, inv/tregn96.f:774; TF is not transformed there)
__global__ void ti_model_kernel(TiArgs A) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= A.ncol) return;
  for (int m = 0; m < A.mmax; m++) {
    float vp, vs, rho;
    const int i = A.knot[m];
    if (i < 0) {
      vs = A.vel[(size_t)(A.nz - 1) * A.ncol + col];
      brocher(vs, vp, rho);
    } else {
      const float vs0 = A.vel[(size_t)i * A.ncol + col], vs1 = A.vel[(size_t)(i + 1) * A.ncol + col];
      float vp0, vp1, r0, r1;
      brocher(vs0, vp0, r0);
      brocher(vs1, vp1, r1);
      vp = vp0 + A.fm[m] * (vp1 - vp0) / A.den[m];
      vs = vs0 + A.fm[m] * (vs1 - vs0) / A.den[m];
      rho = r0 + A.fm[m] * (r1 - r0) / A.den[m];
    }
    const float TA = rho * (vp * vp), TL = rho * (vs * vs), TF = 1.0f * (TA - 2 * TL);
    const size_t o = (size_t)m * A.ncol + col;
    A.fTA[o] = TA; A.fTL[o] = TL; A.fTF[o] = TF; A.frho[o] = rho; A.fvp[o] = vp; A.fvs[o] = vs;
    A.zta[o] = (double)(float)((double)TA * A.pw_el[m]);
    A.ztl[o] = (double)(float)((double)TL * A.pw_el[m]);
    A.zrho[o] = (double)(float)((double)rho * A.pw_rho[m]);
    A.ztf[o] = (double)TF;
    if (!(TL > 0.0001f * TA)) *A.bad = 1;
  }
}

__global__ __launch_bounds__(TT) void ti_kernel(TiArgs A) {
  // a small launch the eikonal solve waits for, usually beside the dispersion kernel's perturbed copies (auxiliary stream): issue
  // priority over their wavefronts (test4_Yunnan: 13.5 ms per call beside the copies, 1.6 ms alone)
  if (A.prio) __builtin_amdgcn_s_setprio(3);   // (only while such copies are pending: alone on the chip there is nobody to overtake)
  const long lane = (long)blockIdx.x * TT + threadIdx.x;
  const long nlane = (long)A.ncol * A.kmax;
  if (lane >= nlane) return;
  const int ip = (int)(lane / A.ncol), col = (int)(lane - (long)ip * A.ncol);
  const int mmax = A.mmax;
  const float twopi = 2.f * 3.141592654f;                    // inv/tregn96.f:418
  const double omega = (double)twopi / (double)A.twopi_t[ip];
  double c = (double)(float)A.pv[(size_t)ip * A.ncol + col];  // cp_in = sngl(cgRc)
  if (!(c > 0.0)) {                                            // no root at this period (pvRc = 0): no kernel
    for (int j = 0; j < A.nz - 1; j++) A.lsen[((size_t)j * A.kmax + ip) * A.ncol + col] = 0.0f;
    return;
  }
  double wvno = omega / c;
  const double wvno2 = wvno * wvno;
  double *upv = A.upv + lane, *prt = A.prt + lane;           // [m][q][nlane]
#define UPV(m, q) upv[((size_t)(m) * 6 + (q)) * nlane]
#define PRT(m, q) prt[((size_t)(m) * 4 + (q)) * nlane]
#define MOD(arr, m) A.arr[(size_t)(m) * A.ncol + col]
  Eig g;
  Trig tg;
  double ur0;
  // ---------------- up ----------------
  {
    cx cd[5], ee[5];
    {
      cx E[4][4], EI[4][4];
      layer_eig(MOD(zta, mmax - 1), MOD(zta, mmax - 1), MOD(ztf, mmax - 1), MOD(ztl, mmax - 1), MOD(zrho, mmax - 1), omega, wvno, g);
      layer_E(g, E, EI);
      constexpr int H0[5] = {0, 0, 0, 1, 2}, H1[5] = {1, 2, 3, 3, 3};   // CG(1),(2),(3),(5),(6), :3079-3088
#pragma unroll
      for (int k = 0; k < 5; k++) {
        const cx m = EI[0][H0[k]] * EI[1][H1[k]] - EI[0][H1[k]] * EI[1][H0[k]];
        cd[k] = C(m.re, 0.0);
        UPV(mmax - 1, k) = m.re;
      }
      UPV(mmax - 1, 5) = 0.0;
    }
    double exsum = 0.0;
    for (int m = mmax - 2; m >= 0; m--) {
      layer_eig(MOD(zta, m), MOD(zta, m), MOD(ztf, m), MOD(ztl, m), MOD(zrho, m), omega, wvno, g);
      layer_trig(g, A.zd[m], tg);
      push_up(g, tg, cd, ee);
      double t1 = 0.0;                                          // cnormc, :1475
#pragma unroll
      for (int i = 0; i < 5; i++) t1 = fmax(t1, cabs_(ee[i]));
      if (t1 < 1.0e-40) t1 = 1.0;
      exsum += tg.pex + tg.svex + log(t1);
#pragma unroll
      for (int i = 0; i < 5; i++) {
        cd[i] = ee[i] * (1.0 / t1);
        UPV(m, i) = cd[i].re;
      }
      UPV(m, 5) = exsum;
    }
    ur0 = (cd[2] / cd[1]).re;
  }
  // ---------------- down + eigenfunctions + energy ----------------
  const double exe0 = UPV(0, 5), f1213 = -UPV(0, 1);
  double vv[4] = {1.0, 0.0, 0.0, 0.0}, exa = 0.0;
  double cur[4] = {ur0, 1.0, 0.0, 0.0};                        // ur, uz, tz, tr at the top of layer m
  double sumi0 = 0.0, sumi1 = 0.0, sumi2 = 0.0, gam_b = 0.0, gam_a = 0.0;
  for (int m = 0; m < mmax; m++) {
    const bool last = m == mmax - 1;
    const double rho = MOD(zrho, m), TA = MOD(zta, m), TC = TA, TF = MOD(ztf, m), TL = MOD(ztl, m), dm = A.zd[m];
    layer_eig(TA, TC, TF, TL, rho, omega, wvno, g);
    double nxt[4] = {0.0, 0.0, 0.0, 0.0};
    if (!last) {
      layer_trig(g, dm, tg);
      double cpex, fp = 1.0, fs = 1.0;                          // down, :3640-3662
      if (tg.pex > tg.svex) { fs = (tg.pex - tg.svex) > 40.0 ? 0.0 : exp(-(tg.pex - tg.svex)); cpex = tg.pex; }
      else { fp = (tg.svex - tg.pex) > 40.0 ? 0.0 : exp(-(tg.svex - tg.pex)); cpex = tg.svex; }
      const cx cosp = (tg.cosp * fp) / g.np, sinpr = (tg.sinpr * fp) / g.np, rsinp = (tg.rsinp * fp) / g.np;
      const cx cosq = (tg.cosq * fs) / g.nsv, sinqr = (tg.sinqr * fs) / g.nsv, rsinq = (tg.rsinq * fs) / g.nsv;
      double AA[4][4];                                          // hska, :3477
#define X(i, md) g.x[(i) - 1][(md) - 1]
      AA[0][0] = (X(1, 1) * X(4, 1) * cosp + X(1, 2) * X(4, 2) * cosq).re;
      AA[0][1] = -(X(1, 1) * X(3, 1) * sinpr + X(1, 2) * X(3, 2) * rsinq).re;
      AA[0][2] = -(X(1, 1) * X(2, 1) * cosp + X(1, 2) * X(2, 2) * cosq).re;
      AA[0][3] = (X(1, 1) * X(1, 1) * sinpr + X(1, 2) * X(1, 2) * rsinq).re;
      AA[1][0] = (X(2, 1) * X(4, 1) * rsinp + X(2, 2) * X(4, 2) * sinqr).re;
      AA[1][1] = -(X(2, 1) * X(3, 1) * cosp + X(2, 2) * X(3, 2) * cosq).re;
      AA[1][2] = -(X(2, 1) * X(2, 1) * rsinp + X(2, 2) * X(2, 2) * sinqr).re;
      AA[2][0] = (X(3, 1) * X(4, 1) * cosp + X(3, 2) * X(4, 2) * cosq).re;
      AA[2][1] = -(X(3, 1) * X(3, 1) * sinpr + X(3, 2) * X(3, 2) * rsinq).re;
      AA[3][0] = (X(4, 1) * X(4, 1) * rsinp + X(4, 2) * X(4, 2) * sinqr).re;
#undef X
      AA[1][3] = -AA[0][2]; AA[2][2] = AA[1][1]; AA[2][3] = -AA[0][1];
      AA[3][1] = -AA[2][0]; AA[3][2] = -AA[1][0]; AA[3][3] = AA[0][0];
      double a0[4], t1 = 0.0;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < 4; j++) s += AA[i][j] * vv[j];
        a0[i] = s;
        t1 = fmax(t1, fabs(s));
      }
      if (t1 < 1.0e-40) t1 = 1.0;                               // rnormc, :1513
      exa += cpex + log(t1);
#pragma unroll
      for (int i = 0; i < 4; i++) vv[i] = a0[i] / t1;
      // eigenfunctions at the top of layer m+1 (svfunc, :1618-1650)
      const double cd1 = UPV(m + 1, 0), cd2 = UPV(m + 1, 1), cd3 = UPV(m + 1, 2), cd4 = -cd3, cd5 = UPV(m + 1, 3), cd6 = UPV(m + 1, 4);
      const double tz1 = -vv[3], tz2 = -vv[2], tz3 = vv[1], tz4 = vv[0];
      const double ext = exa + UPV(m + 1, 5) - exe0;
      if (ext > -80.0 && ext < 80.0) {
        const double fact = exp(ext);
        nxt[0] = (tz2 * cd6 - tz3 * cd5 + tz4 * cd4) * fact / f1213;
        nxt[1] = (-tz1 * cd6 + tz3 * cd3 - tz4 * cd2) * fact / f1213;
        nxt[2] = (tz1 * cd5 - tz2 * cd3 + tz4 * cd1) * fact / f1213;
        nxt[3] = (-tz1 * cd4 + tz2 * cd2 - tz3 * cd1) * fact / f1213;
      }
    }
    // ---- energy integrals of layer m (energy :3777, intijr :4073) ----
    {
      cx E[4][4], EI[4][4];
      layer_E(g, E, EI);
      const cx ra = g.rp, rb = g.rsv;
      cx k[4];   // kmpu, kmsu (amplitudes at the layer bottom), km1pd, km1sd (at its top)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const double *u = (r < 2 && !last) ? nxt : cur;
        k[r] = EI[r][0] * u[0] + EI[r][1] * u[1] + EI[r][2] * u[2] + EI[r][3] * u[3];
      }
      cx FA = C(0, 0), GA = FA, FB = FA, GB = FA, H1 = FA, H2 = FA;
      if (!last) {
        FA = ffunc(ra, dm); GA = gfunc(ra, dm); FB = ffunc(rb, dm); GB = gfunc(rb, dm);
        H1 = h1func(ra, rb, dm); H2 = h2func(ra, rb, dm);
      }
      constexpr int II[6] = {0, 0, 1, 1, 2, 3}, JJ[6] = {0, 2, 1, 3, 2, 3};
      double I[6];
#pragma unroll
      for (int q = 0; q < 6; q++) {
        const int i = II[q], j = JJ[q];
#define EE2(a, b) (E[i][a] * E[j][b] + E[i][b] * E[j][a])
        cx s;
        if (!last) {
          s = E[i][0] * E[j][0] * (k[0] * k[0]) * FA + E[i][2] * E[j][2] * (k[2] * k[2]) * FA +
              E[i][1] * E[j][1] * (k[1] * k[1]) * FB + E[i][3] * E[j][3] * (k[3] * k[3]) * FB +
              H1 * (EE2(0, 1) * (k[0] * k[1]) + EE2(2, 3) * (k[2] * k[3])) +
              H2 * (EE2(0, 3) * (k[0] * k[3]) + EE2(1, 2) * (k[2] * k[1])) + GA * EE2(0, 2) * (k[0] * k[2]) +
              GB * EE2(1, 3) * (k[1] * k[3]);
        } else {
          s = (E[i][2] * E[j][2] * (k[2] * k[2])) / (ra * 2.0) + (EE2(2, 3) * (k[2] * k[3])) / (ra + rb) +
              (E[i][3] * E[j][3] * (k[3] * k[3])) / (rb * 2.0);
        }
#undef EE2
        I[q] = s.re;
      }
      const double ah = sqrt(TA / rho), av = sqrt(TC / rho), bv = sqrt(TL / rho);
      const double eta = TF / (TA - 2. * TL), a12 = -wvno, a14 = 1.0 / TL, a21 = wvno * TF / TC, a23 = 1.0 / TC;
      const double URUR = I[0], UZUZ = I[2];
      const double DURDUR = a12 * a12 * I[2] + 2. * a12 * a14 * I[3] + a14 * a14 * I[5];
      const double DUZDUZ = a21 * a21 * I[0] + 2. * a21 * a23 * I[1] + a23 * a23 * I[4];
      const double URDUZ = a21 * I[0] + a23 * I[1], UZDUR = a12 * I[2] + a14 * I[3];
      sumi0 += rho * (URUR + UZUZ);
      sumi1 += TL * UZUZ + TA * URUR;
      sumi2 += TL * UZDUR - TF * URDUZ;
      const double fah = rho * ah * (URUR - 2. * eta * URDUZ / wvno);
      const double fav = rho * av * DUZDUZ / wvno2;
      const double fbv = rho * bv * (UZUZ + 2. * UZDUR / wvno + DURDUR / wvno2 + 4. * eta * URDUZ / wvno);
      const double fn = -TF * URDUZ / (wvno * eta);
      PRT(m, 0) = fah; PRT(m, 1) = fbv; PRT(m, 2) = fn;
      gam_b += fbv * bv;                                        // gammap sums (:3745-3757), normalised below
      gam_a += fav * av + fah * ah;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) cur[i] = nxt[i];
  }
  // ---------------- normalisation, gammap, sprayl, depthkernelTI sum ----------------
  const double ugr = (wvno * sumi1 + sumi2) / (omega * sumi0);
  const double nrm = ugr * sumi0;
  {
    const double pi = 3.141592653589493;                        // as typed in :3741
    const double zqa = (double)(1.0f / 150.0f), zqb = (double)(1.0f / 50.0f);
    const double lg = log(omega / (2.0 * pi * 1.0));
    const double dc = lg * (gam_b / nrm * zqb) / pi + lg * (gam_a / nrm * zqa) / pi;
    c = omega / wvno + dc;
  }
  const double q = c / (2. * 6370.0 * omega);
  const double tm = sqrt(1. + q * q), tm3 = tm * tm * tm;
  int m = 0;
  for (int j = 0; j < A.nz - 1; j++) {
    float acc = 0.0f;
    for (; m < mmax - 1 && A.jlay[m] == j; m++) {
      const double a = PRT(m, 0) / nrm * (double)A.vtp[m] / tm3, b = PRT(m, 1) / nrm * (double)A.vtp[m] / tm3, n = PRT(m, 2) / nrm;
      const float dah = fabs(a) < 1.0e-36 ? 0.0f : (float)a, dbv = fabs(b) < 1.0e-36 ? 0.0f : (float)b;
      const float dn = fabs(n) < 1.0e-36 ? 0.0f : (float)n;
      const float TAf = MOD(fTA, m), TLf = MOD(fTL, m), TFf = MOD(fTF, m);
      const float den = (TAf - 2.0f * TLf) * (TAf - 2.0f * TLf);
      const float dA = 0.5f / (MOD(frho, m) * MOD(fvp, m)) * dah - TFf / den * dn;
      const float dL = 0.5f / (MOD(frho, m) * MOD(fvs, m)) * dbv + 2.0f * TFf / den * dn;
      acc = acc + dA * TAf + dL * TLf;
    }
    A.lsen[((size_t)j * A.kmax + ip) * A.ncol + col] = acc;
  }
#undef UPV
#undef PRT
#undef MOD
}

}  // namespace

extern "C" int dazim_ti_kernels(dazim_ctx *ctx, int nx, int ny, int nz, const float *vel_u, const float *depz,
                                float minthk0, int kmax, const double *periods, const double *pv_u, float *lsen_u) {
  if (!ctx || !vel_u || !depz || !periods || !pv_u || !lsen_u) return dz_fail(ctx, DAZIM_E_BAD_ARG, "null argument");
  if (nz < 2 || kmax < 1 || kmax > 60 || nx < 1 || ny < 1) return dz_fail(ctx, DAZIM_E_BAD_ARG, "bad nz/kmax");
  DZ_HIP(hipSetDevice(ctx->device));
  const int ncol = nx * ny;
  // ---- geometry-only tables: refineLayerMdl, sphere_tdisp96 (radius 6371), bldsph (radius 6370) ----
  std::vector<int> knot, jlay;
  std::vector<float> fm, den, thk, vtp, tper(kmax);
  for (int i = 0; i < nz - 1; i++) {
    const float t = depz[i + 1] - depz[i];
    const float minthk = t / minthk0;
    const int nsub = (int)((t + 1.0e-4f) / minthk) + 1;
    for (int j = 1; j <= nsub; j++) {
      knot.push_back(i); jlay.push_back(i); fm.push_back((float)(2 * j - 1)); den.push_back((float)(2 * nsub));
      thk.push_back(t / (float)nsub);
    }
  }
  knot.push_back(-1); jlay.push_back(nz - 1); fm.push_back(0.f); den.push_back(1.f); thk.push_back(0.f);
  const int mmax = (int)knot.size();
  if (mmax > NLMAX) return dz_fail(ctx, DAZIM_E_BAD_ARG, "refined model has %d layers > NL=%d", mmax, NLMAX);
  std::vector<double> zd(mmax), pw_rho(mmax), pw_el(mmax);
  vtp.resize(mmax);
  {
    const double ar = (double)6371.0f;
    double r0 = ar;
    std::vector<float> d(thk);
    d[mmax - 1] = 1.0f;
    for (int i = 0; i < mmax; i++) {
      const double r1 = r0 - (double)d[i];
      const double z0 = ar * log(ar / r0), z1 = ar * log(ar / r1);
      d[i] = (float)(z1 - z0);
      const double tmp = (ar + ar) / (r0 + r1);
      pw_rho[i] = pow(tmp, -2.275);
      pw_el[i] = pow(tmp, -0.2750);
      r0 = r1;
    }
    d[mmax - 1] = 0.0f;
    for (int i = 0; i < mmax; i++) zd[i] = (double)d[i];
  }
  {
    const double ar = 6370.0;
    double r0 = ar;
    zd[mmax - 1] = 1.0;
    for (int i = 0; i < mmax; i++) {
      const double r1 = r0 * exp(-zd[i] / ar);
      vtp[i] = (float)((ar + ar) / (r0 + r1));
      r0 = r1;
    }
  }
  for (int k = 0; k < kmax; k++) tper[k] = (float)periods[k];   // t_in = sngl(tRc)

  DzBuf<float> vel, lsen;
  DzBuf<double> pv;
  int rc;
  if ((rc = vel.init(ctx, vel_u, (size_t)nz * ncol, true, false))) return rc;
  if ((rc = pv.init(ctx, pv_u, (size_t)kmax * ncol, true, false))) return rc;
  if ((rc = lsen.init(ctx, lsen_u, (size_t)(nz - 1) * kmax * ncol, false, true))) return rc;
  const size_t nlane = (size_t)ncol * kmax;
  TiArgs A;
  A.ncol = ncol; A.nz = nz; A.kmax = kmax; A.mmax = mmax;
  A.vel = vel.dev; A.pv = pv.dev; A.lsen = lsen.dev;
  // one scratch block for the small tables
  const size_t tab_bytes = (size_t)mmax * (2 * sizeof(int) + 4 * sizeof(float) + 3 * sizeof(double)) + kmax * sizeof(float) + 256;
  void *p;
  if ((rc = dz_scratch(ctx, "ti.tables", tab_bytes, &p))) return rc;
  char *base = (char *)p;
  auto put = [&](const void *src, size_t bytes) -> void * {
    void *dst = base;
    (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream);
    base += (bytes + 15) & ~(size_t)15;
    return dst;
  };
  A.zd = (const double *)put(zd.data(), mmax * 8);
  A.pw_rho = (const double *)put(pw_rho.data(), mmax * 8);
  A.pw_el = (const double *)put(pw_el.data(), mmax * 8);
  A.knot = (const int *)put(knot.data(), mmax * 4);
  A.jlay = (const int *)put(jlay.data(), mmax * 4);
  A.fm = (const float *)put(fm.data(), mmax * 4);
  A.den = (const float *)put(den.data(), mmax * 4);
  A.thk = (const float *)put(thk.data(), mmax * 4);
  A.vtp = (const float *)put(vtp.data(), mmax * 4);
  A.twopi_t = (const float *)put(tper.data(), kmax * 4);
  DZ_HIP(hipGetLastError());
  if ((rc = dz_scratch(ctx, "ti.model64", (size_t)mmax * ncol * 8 * 4, &p))) return rc;
  A.zta = (double *)p; A.ztl = A.zta + (size_t)mmax * ncol; A.ztf = A.ztl + (size_t)mmax * ncol; A.zrho = A.ztf + (size_t)mmax * ncol;
  if ((rc = dz_scratch(ctx, "ti.model32", (size_t)mmax * ncol * 4 * 6, &p))) return rc;
  A.fTA = (float *)p; A.fTL = A.fTA + (size_t)mmax * ncol; A.fTF = A.fTL + (size_t)mmax * ncol;
  A.frho = A.fTF + (size_t)mmax * ncol; A.fvp = A.frho + (size_t)mmax * ncol; A.fvs = A.fvp + (size_t)mmax * ncol;
  if ((rc = dz_scratch(ctx, "ti.up", (size_t)mmax * 6 * nlane * 8, &p))) return rc;
  A.upv = (double *)p;
  if ((rc = dz_scratch(ctx, "ti.prt", (size_t)mmax * 4 * nlane * 8, &p))) return rc;
  A.prt = (double *)p;
  if ((rc = dz_scratch(ctx, "ti.bad", 16, &p))) return rc;
  A.bad = (int *)p;
  DZ_HIP(hipMemsetAsync(A.bad, 0, 4, ctx->stream));
  DZ_HIP(hipStreamSynchronize(ctx->stream));   // host tables go out of scope after the launches are queued; keep it simple
  {
    DzTimer t(ctx, "ti");
    A.prio = ctx->aux_pending ? 1 : 0;
    hipLaunchKernelGGL(ti_model_kernel, dim3((ncol + 127) / 128), dim3(128), 0, ctx->stream, A);
    hipLaunchKernelGGL(ti_kernel, dim3((unsigned)((nlane + TT - 1) / TT)), dim3(TT), 0, ctx->stream, A);
    DZ_HIP(hipGetLastError());
    t.stop();
  }
  int bad = 0;
  DZ_HIP(hipMemcpyAsync(&bad, A.bad, 4, hipMemcpyDeviceToHost, ctx->stream));
  if ((rc = lsen.finish())) return rc;
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  if (bad) return dz_fail(ctx, DAZIM_E_BAD_ARG, "fluid layer (TL <= 1e-4 TA) in the model: not supported by dazim_ti_kernels");
  return 0;
}
