// surfdisp.hip -- the reference's surfdisp96 with EVERY value of its arguments, for a batch of layered models.
//
// The hot path (disp.hip) holds the one combination the reference's own programs use: spherical model, Rayleigh wave,
// fundamental mode, phase velocity (inv/CalSurfG.f90:1076-1077; Main_Jt.f90 STOPs on anything else).  The subroutine itself
// (inv/surfdisp96.f:52-354) also does Love waves (dltar1, :704-763), a water layer on top (dltar4's branch :844-860),
// higher modes (the loop :217-349) and group velocities (:226-233, :276-304) on a flat or a flattened spherical model;
// dazim_surfdisp96 offers all of it with the subroutine's own argument list, one lane per model.
//
// What runs where: the subroutine's prologue -- Earth flattening (sphere :480-547), the extremal velocities and the start
// value of the search (gtsolh :361-382), a few hundred host flops per model with the host libm's log / powf the reference
// calls -- is done on the host; the root searches (getsol :384-476, nevill :551-668 and the period equations) run on the
// device in plain IEEE fp64, the reference's operations in the reference's order (this file is built with
// -ffp-contract=off like the rest of the library; no reciprocal / fused forms as in disp.hip's tuned dltar4).  What can
// differ from the reference is the last bit of the device's sin / cos / exp, i.e. roots that differ by ~1e-9 km/s before
// they are rounded to fp32: tests/test_surfdisp_full_gpu.py states the resulting bars.
#include <cmath>

#include "dazim_internal.h"

namespace {

constexpr int NL = 200;  // inv/surfdisp96.f:57
constexpr int NP = 60;   // inv/surfdisp96.f:59

struct SdArgs {
  int nmodel, kmax, iwave, mode, igr;
  const float *d, *a, *b, *rho;   // [layer][model] flattened model (of this wave type)
  const int *mmax, *llw;          // [model]
  const float *betmx;             // [model]
  const double *cc;               // [model] start value of the search (= cm)
  const double *t;                // [kmax]
  double *c, *cb;                 // [k][model] scratch: roots of the current / previous mode at T (T/(1+h)) and at T/(1-h)
  double *cg;                     // [model][kmax] out
  int *nfail;                     // number of (model, period) entries left at 0
};

struct Model {
  const float *d, *a, *b, *rho;   // this lane's column of the [layer][model] arrays
  int stride, mmax, llw;
  __device__ float D(int m) const { return d[(size_t)(m - 1) * stride]; }
  __device__ float A(int m) const { return a[(size_t)(m - 1) * stride]; }
  __device__ float B(int m) const { return b[(size_t)(m - 1) * stride]; }
  __device__ float R(int m) const { return rho[(size_t)(m - 1) * stride]; }
};

__device__ __forceinline__ double sgn(double x) { return copysign(1.0, x); }

struct VarOut {
  double w, cosp, a0, cpcq, cpy, cpz, cqw, cqx, xy, xz, wy, wz;
};

// var, inv/surfdisp96.f:868-985
__device__ void var(double p, double q, double ra, double rb, double wvno, double xka, double xkb, double dpth, VarOut &o) {
  double w = 0, x = 0, y = 0, z = 0, cosp = 0, cosq = 0, sinp, sinq, fac, pex = 0.0, sex = 0.0;
  if (wvno < xka) {
    sinp = sin(p);
    w = sinp / ra;
    x = -ra * sinp;
    cosp = cos(p);
  } else if (wvno == xka) {
    cosp = 1.0;
    w = dpth;
    x = 0.0;
  } else if (wvno > xka) {
    pex = p;
    fac = 0.0;
    if (p < 16) fac = exp(-2.0 * p);
    cosp = (1.0 + fac) * 0.5;
    sinp = (1.0 - fac) * 0.5;
    w = sinp / ra;
    x = ra * sinp;
  }
  if (wvno < xkb) {
    sinq = sin(q);
    y = sinq / rb;
    z = -rb * sinq;
    cosq = cos(q);
  } else if (wvno == xkb) {
    cosq = 1.0;
    y = dpth;
    z = 0.0;
  } else if (wvno > xkb) {
    sex = q;
    fac = 0.0;
    if (q < 16) fac = exp(-2.0 * q);
    cosq = (1.0 + fac) * 0.5;
    sinq = (1.0 - fac) * 0.5;
    y = sinq / rb;
    z = rb * sinq;
  }
  const double exa = pex + sex;
  o.a0 = 0.0;
  if (exa < 60.0) o.a0 = exp(-exa);
  o.w = w;
  o.cosp = cosp;
  o.cpcq = cosp * cosq;
  o.cpy = cosp * y;
  o.cpz = cosp * z;
  o.cqw = cosq * w;
  o.cqx = cosq * x;
  o.xy = x * y;
  o.xz = x * z;
  o.wy = w * y;
  o.wz = w * z;
}

// dltar4, inv/surfdisp96.f:767-865 with dnka (:1018-1062) and normc (:989-1014)
__device__ double dltar4(const Model &M, double wvno, double omga) {
  double e[5], ee[5], ca[5][5];
  const int mmax = M.mmax;
  double omega = omga;
  if (omega < 1.0e-4) omega = 1.0e-4;
  const double wvno2 = wvno * wvno;
  double xka = omega / (double)M.A(mmax);
  double xkb = omega / (double)M.B(mmax);
  double wvnop = wvno + xka, wvnom = fabs(wvno - xka);
  double ra = sqrt(wvnop * wvnom);
  wvnop = wvno + xkb;
  wvnom = fabs(wvno - xkb);
  double rb = sqrt(wvnop * wvnom);
  double t = (double)M.B(mmax) / omega;
  double gammk = 2.0 * t * t;
  double gam = gammk * wvno2;
  double gamm1 = gam - 1.0;
  double rho1 = (double)M.R(mmax);
  e[0] = rho1 * rho1 * (gamm1 * gamm1 - gam * gammk * ra * rb);
  e[1] = -rho1 * ra;
  e[2] = rho1 * (gamm1 - gammk * ra * rb);
  e[3] = rho1 * rb;
  e[4] = wvno2 - ra * rb;
  VarOut v;
  for (int m = mmax - 1; m >= M.llw; m--) {
    const double am = (double)M.A(m), bm = (double)M.B(m);
    xka = omega / am;
    xkb = omega / bm;
    t = bm / omega;
    gammk = 2.0 * t * t;
    gam = gammk * wvno2;
    wvnop = wvno + xka;
    wvnom = fabs(wvno - xka);
    ra = sqrt(wvnop * wvnom);
    wvnop = wvno + xkb;
    wvnom = fabs(wvno - xkb);
    rb = sqrt(wvnop * wvnom);
    const double dpth = (double)M.D(m);
    rho1 = (double)M.R(m);
    const double p = ra * dpth, q = rb * dpth;
    var(p, q, ra, rb, wvno, xka, xkb, dpth, v);
    gamm1 = gam - 1.0;
    const double twgm1 = gam + gamm1, gmgmk = gam * gammk, gmgm1 = gam * gamm1, gm1sq = gamm1 * gamm1;
    const double rho2 = rho1 * rho1, a0pq = v.a0 - v.cpcq;
    ca[0][0] = v.cpcq - 2.0 * gmgm1 * a0pq - gmgmk * v.xz - wvno2 * gm1sq * v.wy;
    ca[0][1] = (wvno2 * v.cpy - v.cqx) / rho1;
    ca[0][2] = -(twgm1 * a0pq + gammk * v.xz + wvno2 * gamm1 * v.wy) / rho1;
    ca[0][3] = (v.cpz - wvno2 * v.cqw) / rho1;
    ca[0][4] = -(2.0 * wvno2 * a0pq + v.xz + wvno2 * wvno2 * v.wy) / rho2;
    ca[1][0] = (gmgmk * v.cpz - gm1sq * v.cqw) * rho1;
    ca[1][1] = v.cpcq;
    ca[1][2] = gammk * v.cpz - gamm1 * v.cqw;
    ca[1][3] = -v.wz;
    ca[1][4] = ca[0][3];
    ca[3][0] = (gm1sq * v.cpy - gmgmk * v.cqx) * rho1;
    ca[3][1] = -v.xy;
    ca[3][2] = gamm1 * v.cpy - gammk * v.cqx;
    ca[3][3] = ca[1][1];
    ca[3][4] = ca[0][1];
    ca[4][0] = -(2.0 * gmgmk * gm1sq * a0pq + gmgmk * gmgmk * v.xz + gm1sq * gm1sq * v.wy) * rho2;
    ca[4][1] = ca[3][0];
    ca[4][2] = -(gammk * gamm1 * twgm1 * a0pq + gam * gammk * gammk * v.xz + gamm1 * gm1sq * v.wy) * rho1;
    ca[4][3] = ca[1][0];
    ca[4][4] = ca[0][0];
    const double tt = -2.0 * wvno2;
    ca[2][0] = tt * ca[4][2];
    ca[2][1] = tt * ca[3][2];
    ca[2][2] = v.a0 + 2.0 * (v.cpcq - ca[0][0]);
    ca[2][3] = tt * ca[1][2];
    ca[2][4] = tt * ca[0][2];
#pragma unroll
    for (int i = 0; i < 5; i++) {
      double cr = 0.0;
#pragma unroll
      for (int j = 0; j < 5; j++) cr = cr + e[j] * ca[j][i];
      ee[i] = cr;
    }
    double t1 = 0.0;
#pragma unroll
    for (int i = 0; i < 5; i++)
      if (fabs(ee[i]) > t1) t1 = fabs(ee[i]);
    if (t1 < 1.e-40) t1 = 1.0;
#pragma unroll
    for (int i = 0; i < 5; i++) e[i] = ee[i] / t1;
  }
  if (M.llw != 1) {  // water layer on top, :844-860
    xka = omega / (double)M.A(1);
    wvnop = wvno + xka;
    wvnom = fabs(wvno - xka);
    ra = sqrt(wvnop * wvnom);
    const double dpth = (double)M.D(1);
    rho1 = (double)M.R(1);
    const double p = ra * dpth;
    const double znul = 1.0e-05;
    var(p, znul, ra, znul, wvno, xka, znul, dpth, v);
    const double w0 = -rho1 * v.w;
    return v.cosp * e[0] + w0 * e[1];
  }
  return e[0];
}

// dltar1: SH period equation, inv/surfdisp96.f:704-763
__device__ double dltar1(const Model &M, double wvno, double omega) {
  const int mmax = M.mmax;
  double beta1 = (double)M.B(mmax);
  double rho1 = (double)M.R(mmax);
  double xkb = omega / beta1;
  double wvnop = wvno + xkb, wvnom = fabs(wvno - xkb);
  double rb = sqrt(wvnop * wvnom);
  double e1 = rho1 * rb;
  double e2 = 1.0 / (beta1 * beta1);
  for (int m = mmax - 1; m >= M.llw; m--) {
    beta1 = (double)M.B(m);
    rho1 = (double)M.R(m);
    const double dm = (double)M.D(m);
    const double xmu = rho1 * beta1 * beta1;
    xkb = omega / beta1;
    wvnop = wvno + xkb;
    wvnom = fabs(wvno - xkb);
    rb = sqrt(wvnop * wvnom);
    const double q = dm * rb;
    double y, z, cosq, sinq, fac;
    if (wvno < xkb) {
      sinq = sin(q);
      y = sinq / rb;
      z = -rb * sinq;
      cosq = cos(q);
    } else if (wvno == xkb) {
      cosq = 1.0;
      y = dm;
      z = 0.0;
    } else {
      fac = 0.0;
      if (q < 16) fac = exp(-2.0 * q);
      cosq = (1.0 + fac) * 0.5;
      sinq = (1.0 - fac) * 0.5;
      y = sinq / rb;
      z = rb * sinq;
    }
    const double e10 = e1 * cosq + e2 * xmu * z;
    const double e20 = e1 * y / xmu + e2 * cosq;
    double xnor = fabs(e10);
    const double ynor = fabs(e20);
    if (ynor > xnor) xnor = ynor;
    if (xnor < 1.e-40) xnor = 1.0;
    e1 = e10 / xnor;
    e2 = e20 / xnor;
  }
  return e1;
}

__device__ double dltar(const Model &M, double wvno, double omega, int kk) {  // :684-700
  return kk == 1 ? dltar1(M, wvno, omega) : dltar4(M, wvno, omega);
}

constexpr double TWOPI = 2.0 * 3.141592653589793;

// nevill, inv/surfdisp96.f:551-668 (half :670-680 inlined)
__device__ double nevill(const Model &M, double t, double c1, double c2, double del1, double del2, int ifunc) {
  double x[20], y[20];
  const double omega = TWOPI / t;
  double c3 = 0.5 * (c1 + c2);
  double del3 = dltar(M, omega / c3, omega, ifunc);
  int nev = 1, nctrl = 1, m = 1;
  for (;;) {
    nctrl++;
    if (nctrl >= 100) break;
    if (c3 < fmin(c1, c2) || c3 > fmax(c1, c2)) {
      nev = 0;
      c3 = 0.5 * (c1 + c2);
      del3 = dltar(M, omega / c3, omega, ifunc);
    }
    const double s13 = del1 - del3, s32 = del3 - del2;
    if (sgn(del3) * sgn(del1) < 0.0) {
      c2 = c3;
      del2 = del3;
    } else {
      c1 = c3;
      del1 = del3;
    }
    if (fabs(c1 - c2) <= 1.e-6 * c1) break;
    if (sgn(s13) != sgn(s32)) nev = 0;
    const double ss1 = fabs(del1), s1 = (double)0.01f * ss1;
    const double ss2 = fabs(del2), s2 = (double)0.01f * ss2;
    bool halve = s1 > ss2 || s2 > ss1 || nev == 0;
    if (!halve) {
      if (nev == 2) {
        x[m] = c3;
        y[m] = del3;
      } else {
        x[0] = c1;
        y[0] = del1;
        x[1] = c2;
        y[1] = del2;
        m = 1;
      }
      for (int kk = 1; kk <= m; kk++) {
        const int j = m - kk + 1;
        const double denom = y[m] - y[j - 1];
        if (fabs(denom) < 1.0e-10 * fabs(y[m])) {
          halve = true;
          break;
        }
        x[j - 1] = (-y[j - 1] * x[j] + y[m] * x[j - 1]) / denom;
      }
      if (!halve) {
        c3 = x[0];
        del3 = dltar(M, omega / c3, omega, ifunc);
        nev = 2;
        m = m + 1;
        if (m > 10) m = 10;
      }
    }
    if (halve) {
      c3 = 0.5 * (c1 + c2);
      del3 = dltar(M, omega / c3, omega, ifunc);
      nev = 1;
      m = 1;
    }
  }
  return c3;
}

// getsol, inv/surfdisp96.f:384-476; del1st is its SAVE variable
__device__ int getsol(const Model &M, double t1, double &c1, double clow, double dc, double cm, float betmx, int ifunc,
                      int ifirst, double &del1st) {
  double c2, del1, del2;
  double omega = TWOPI / t1;
  del1 = dltar(M, omega / c1, omega, ifunc);
  if (ifirst == 1) del1st = del1;
  const double plmn = sgn(del1st) * sgn(del1);
  int idir = (ifirst == 1) ? 1 : (plmn >= 0.0 ? 1 : -1);
  for (;;) {
    c2 = (idir > 0) ? c1 + dc : c1 - dc;
    if (c2 <= clow) {
      idir = 1;
      c1 = clow;
      continue;
    }
    omega = TWOPI / t1;
    del2 = dltar(M, omega / c2, omega, ifunc);
    if (sgn(del1) != sgn(del2)) break;
    c1 = c2;
    del1 = del2;
    if (c1 < cm) return -1;
    if (c1 >= ((double)betmx + dc)) return -1;
  }
  c1 = nevill(M, t1, c1, c2, del1, del2, ifunc);
  if (c1 > (double)betmx) return -1;
  return 1;
}

// the loops over modes and periods, inv/surfdisp96.f:212-349, one lane per model
__global__ __launch_bounds__(64) void surfdisp_kernel(SdArgs S) {
  const int im = blockIdx.x * blockDim.x + threadIdx.x;
  if (im >= S.nmodel) return;
  Model M;
  M.d = S.d + im;
  M.a = S.a + im;
  M.b = S.b + im;
  M.rho = S.rho + im;
  M.stride = S.nmodel;
  M.mmax = S.mmax[im];
  M.llw = S.llw[im];
  const int ifunc = S.iwave, kmax = S.kmax, igr = S.igr;
  const float betmx = S.betmx[im];
  const float ddc = 0.005f, h = 0.005f, sone = 1.500f;
  const double one = 1.0e-2, onea = (double)sone;
  const double cc = S.cc[im];
  const double dc = fabs((double)ddc);
  const double cm = cc;
  double c1 = cc, clow = cc, del1st = 0.0;
  double *c = S.c + im, *cb = S.cb + im;
  double *cg = S.cg + (size_t)im * kmax;
  const size_t st = (size_t)S.nmodel;
  for (int i = 0; i < kmax; i++) {
    cb[i * st] = 0.0;
    c[i * st] = 0.0;
  }
  int ift = 999;
  for (int iq = 1; iq <= S.mode; iq++) {
    const int is = 1, ie = kmax;
    int k;
    bool failed = false;
    for (k = is; k <= ie; k++) {
      if (k >= ift) {
        failed = true;
        break;
      }
      double t1 = S.t[k - 1];
      float t1a, t1b = 0.0f;
      if (igr > 0) {
        t1a = (float)(t1 / (double)(1.f + h));
        t1b = (float)(t1 / (double)(1.f - h));
        t1 = (double)t1a;
      } else {
        t1a = (float)t1;
      }
      int ifirst;
      if (k == is && iq == 1) {
        c1 = cc;
        clow = cc;
        ifirst = 1;
      } else if (k == is && iq > 1) {
        c1 = c[(is - 1) * st] + one * dc;
        clow = c1;
        ifirst = 1;
      } else if (k > is && iq > 1) {
        ifirst = 0;
        clow = c[(k - 1) * st] + one * dc;
        c1 = c[(k - 2) * st];
        if (c1 < clow) c1 = clow;
      } else {
        ifirst = 0;
        c1 = c[(k - 2) * st] - onea * dc;
        clow = cm;
      }
      int iret = getsol(M, t1, c1, clow, dc, cm, betmx, ifunc, ifirst, del1st);
      if (iret == -1) {
        failed = true;
        break;
      }
      const double ck = c1;
      c[(k - 1) * st] = ck;
      if (igr > 0) {
        t1 = (double)t1b;
        clow = cb[(k - 1) * st] + one * dc;
        c1 = c1 - onea * dc;
        iret = getsol(M, t1, c1, clow, dc, cm, betmx, ifunc, 0, del1st);
        if (iret == -1) c1 = ck;
        cb[(k - 1) * st] = c1;
      } else {
        c1 = 0.0;
      }
      const float cc0 = (float)ck;
      const float cc1 = (float)c1;
      if (igr == 0) {
        cg[k - 1] = (double)cc0;
      } else {
        const float gvel = (1 / t1a - 1 / t1b) / (1 / (t1a * cc0) - 1 / (t1b * cc1));
        cg[k - 1] = (double)gvel;
      }
    }
    if (failed) {  // :1700-1770
      ift = k;
      for (int i = k; i <= ie; i++) cg[i - 1] = 0.0;
    }
  }
  int nz = 0;
  for (int i = 0; i < kmax; i++)
    if (cg[i] == 0.0) nz++;
  if (nz) atomicAdd(S.nfail, nz);
}

// gtsolh, inv/surfdisp96.f:361-382, all fp32 (host)
float gtsolh(float a, float b) {
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

}  // namespace

extern "C" int dazim_surfdisp96(dazim_ctx *ctx, int nmodel, int nlayer_max, const int *nlayer, const float *thk,
                                const float *vp, const float *vs, const float *rho, int iflsph, int iwave, int mode,
                                int igr, int kmax, const double *periods, double *cg, int *n_failed) {
  if (!ctx) return DAZIM_E_BAD_ARG;
  if (nmodel < 0 || !nlayer || !thk || !vp || !vs || !rho || !periods || !cg)
    return dz_fail(ctx, DAZIM_E_BAD_ARG, "dazim_surfdisp96: null argument");
  if (nlayer_max < 1 || nlayer_max > NL) return dz_fail(ctx, DAZIM_E_BAD_ARG, "nlayer_max=%d outside 1..%d (NL)", nlayer_max, NL);
  if (kmax < 1 || kmax > NP) return dz_fail(ctx, DAZIM_E_BAD_ARG, "kmax=%d outside 1..%d (NP)", kmax, NP);
  if (iwave != 1 && iwave != 2) return dz_fail(ctx, DAZIM_E_BAD_ARG, "iwave=%d: 1 (Love) or 2 (Rayleigh)", iwave);
  if (iflsph != 0 && iflsph != 1) return dz_fail(ctx, DAZIM_E_BAD_ARG, "iflsph=%d: 0 (flat) or 1 (spherical)", iflsph);
  if (mode < 1) return dz_fail(ctx, DAZIM_E_BAD_ARG, "mode=%d: 1 = fundamental, 2 = first higher, ...", mode);
  if (n_failed) *n_failed = 0;
  if (nmodel == 0) return 0;
  DZ_HIP(hipSetDevice(ctx->device));
  // ---- inputs on the host (device pointers are copied back: the prologue is host arithmetic) ----
  const size_t nl = (size_t)nmodel * nlayer_max;
  std::vector<float> h_thk(nl), h_vp(nl), h_vs(nl), h_rho(nl);
  std::vector<int> h_n(nmodel);
  auto fetch = [&](void *dst, const void *src, size_t bytes) -> int {
    if (dz_is_device_ptr(src)) {
      DZ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    } else {
      memcpy(dst, src, bytes);
    }
    return 0;
  };
  int rc;
  if ((rc = fetch(h_thk.data(), thk, nl * 4)) || (rc = fetch(h_vp.data(), vp, nl * 4)) || (rc = fetch(h_vs.data(), vs, nl * 4)) ||
      (rc = fetch(h_rho.data(), rho, nl * 4)) || (rc = fetch(h_n.data(), nlayer, (size_t)nmodel * 4)))
    return rc;
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  // ---- the subroutine's prologue per model, inv/surfdisp96.f:93-211 ----
  std::vector<float> f_d(nl, 0.0f), f_a(nl, 1.0f), f_b(nl, 1.0f), f_r(nl, 1.0f), f_betmx(nmodel);
  std::vector<int> f_llw(nmodel);
  std::vector<double> f_cc(nmodel);
  for (int im = 0; im < nmodel; im++) {
    const int mmax = h_n[im];
    if (mmax < 1 || mmax > nlayer_max) return dz_fail(ctx, DAZIM_E_BAD_ARG, "model %d: nlayer=%d outside 1..%d", im, mmax, nlayer_max);
    float d[NL], a[NL], b[NL], r[NL], rtp[NL], btp[NL];
    for (int i = 0; i < mmax; i++) {
      const size_t s = (size_t)im * nlayer_max + i;
      b[i] = h_vs[s];
      a[i] = h_vp[s];
      d[i] = h_thk[s];
      r[i] = h_rho[s];
    }
    int llw = 1;
    if (b[0] <= 0.0f) llw = 2;
    if (iflsph == 1) {  // sphere(0,0) then sphere(ifunc,1), :480-547
      const double ar = 6370.0;
      double dr = 0.0, r0 = ar;
      d[mmax - 1] = 1.0f;
      for (int i = 0; i < mmax; i++) rtp[i] = r[i];
      for (int i = 0; i < mmax; i++) {
        dr = dr + (double)d[i];
        const double r1 = ar - dr;
        const double z0 = ar * log(ar / r0), z1 = ar * log(ar / r1);
        d[i] = (float)(z1 - z0);
        const double tmp = (ar + ar) / (r0 + r1);
        a[i] = (float)((double)a[i] * tmp);
        b[i] = (float)((double)b[i] * tmp);
        btp[i] = (float)tmp;
        r0 = r1;
      }
      d[mmax - 1] = 0.0f;
    }
    float betmx = -1.e20f, betmn = 1.e20f;
    int jmn = 1, jsol = 1;
    for (int i = 0; i < mmax; i++) {
      if (b[i] > 0.01f && b[i] < betmn) {
        betmn = b[i];
        jmn = i + 1;
        jsol = 1;
      } else if (b[i] <= 0.01f && a[i] < betmn) {
        betmn = a[i];
        jmn = i + 1;
        jsol = 0;
      }
      if (b[i] > betmx) betmx = b[i];
    }
    if (iflsph == 1) {
      for (int i = 0; i < mmax; i++) {
        if (iwave == 1) {   // btp**(-5): reciprocal of the integer power, as the reference's build forms it
          const float x = btp[i];
          const float x2 = x * x;
          r[i] = rtp[i] * (1.0f / (x2 * x2 * x));
        } else {
          r[i] = rtp[i] * powf(btp[i], -2.275f);
        }
      }
      d[mmax - 1] = 0.0f;
    }
    float cc1 = (jsol == 0) ? betmn : gtsolh(a[jmn - 1], b[jmn - 1]);
    cc1 = .95f * cc1;
    cc1 = .90f * cc1;
    f_cc[im] = (double)cc1;
    f_betmx[im] = betmx;
    f_llw[im] = llw;
    for (int i = 0; i < mmax; i++) {
      const size_t s = (size_t)i * nmodel + im;
      f_d[s] = d[i];
      f_a[s] = a[i];
      f_b[s] = b[i];
      f_r[s] = r[i];
    }
  }
  // ---- device ----
  SdArgs S;
  S.nmodel = nmodel;
  S.kmax = kmax;
  S.iwave = iwave;
  S.mode = mode;
  S.igr = igr;
  void *p;
  auto up = [&](const char *name, const void *src, size_t bytes, const void **out) -> int {
    int r_ = dz_scratch(ctx, name, bytes, &p);
    if (r_) return r_;
    DZ_HIP(hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    *out = p;
    return 0;
  };
  if ((rc = up("sd.d", f_d.data(), nl * 4, (const void **)&S.d)) || (rc = up("sd.a", f_a.data(), nl * 4, (const void **)&S.a)) ||
      (rc = up("sd.b", f_b.data(), nl * 4, (const void **)&S.b)) || (rc = up("sd.rho", f_r.data(), nl * 4, (const void **)&S.rho)) ||
      (rc = up("sd.mmax", h_n.data(), (size_t)nmodel * 4, (const void **)&S.mmax)) ||
      (rc = up("sd.llw", f_llw.data(), (size_t)nmodel * 4, (const void **)&S.llw)) ||
      (rc = up("sd.betmx", f_betmx.data(), (size_t)nmodel * 4, (const void **)&S.betmx)) ||
      (rc = up("sd.cc", f_cc.data(), (size_t)nmodel * 8, (const void **)&S.cc)) ||
      (rc = up("sd.t", periods, (size_t)kmax * 8, (const void **)&S.t)))
    return rc;
  if ((rc = dz_scratch(ctx, "sd.c", (size_t)nmodel * kmax * 8, &p))) return rc;
  S.c = (double *)p;
  if ((rc = dz_scratch(ctx, "sd.cb", (size_t)nmodel * kmax * 8, &p))) return rc;
  S.cb = (double *)p;
  if ((rc = dz_scratch(ctx, "sd.nfail", 16, &p))) return rc;
  S.nfail = (int *)p;
  DZ_HIP(hipMemsetAsync(S.nfail, 0, 4, ctx->stream));
  DzBuf<double> out;
  if ((rc = out.init(ctx, cg, (size_t)nmodel * kmax, false, true))) return rc;
  S.cg = out.dev;
  {
    DzTimer t(ctx, "surfdisp96");
    hipLaunchKernelGGL(surfdisp_kernel, dim3((unsigned)((nmodel + 63) / 64)), dim3(64), 0, ctx->stream, S);
    DZ_HIP(hipGetLastError());
    t.stop();
  }
  int nfail = 0;
  DZ_HIP(hipMemcpyAsync(&nfail, S.nfail, 4, hipMemcpyDeviceToHost, ctx->stream));
  if ((rc = out.finish())) return rc;
  DZ_HIP(hipStreamSynchronize(ctx->stream));   // (also: the host vectors above were the sources of asynchronous copies)
  if (n_failed) *n_failed = nfail;
  return 0;
}
