/* disp.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * CPU restatement of the reference's Rayleigh-wave dispersion path: surfdisp96 (fundamental-mode
 * phase velocity by Thomson-Haskell / Dunkin compound matrices with bracket + Neville root search)
 * and depthkernel (finite-difference dlnc/dlnVs,Vp,rho).  Types follow the F77 implicit typing of
 * the reference exactly: the model arrays and a few scalars are fp32, the root search is fp64.
 * Integer powers are evaluated by repeated left-to-right multiplication, which is what the flang
 * build of the reference (the oracle pin) does.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int mmax;
  float d[ORC_NL], a[ORC_NL], b[ORC_NL], rho[ORC_NL], rtp[ORC_NL], dtp[ORC_NL], btp[ORC_NL];
  double del1st; /* SAVE del1st, inv/surfdisp96.f:409 */
} model;

/* inv/surfdisp96.f:868-985 + :1018-1062 + :807-843: one secular-function evaluation */
static double dltar4(const model *M, double wvno, double omga) {
  double e[5], ee[5], ca[5][5];
  int mmax = M->mmax;
  double omega = omga;
  if (omega < 1.0e-4) omega = 1.0e-4;
  double wvno2 = wvno * wvno;
  double xka = omega / (double)M->a[mmax - 1];
  double xkb = omega / (double)M->b[mmax - 1];
  double wvnop = wvno + xka, wvnom = fabs(wvno - xka);
  double ra = sqrt(wvnop * wvnom);
  wvnop = wvno + xkb;
  wvnom = fabs(wvno - xkb);
  double rb = sqrt(wvnop * wvnom);
  double t = (double)M->b[mmax - 1] / omega;
  double gammk = 2.0 * t * t;
  double gam = gammk * wvno2;
  double gamm1 = gam - 1.0;
  double rho1 = (double)M->rho[mmax - 1];
  e[0] = rho1 * rho1 * (gamm1 * gamm1 - gam * gammk * ra * rb);
  e[1] = -rho1 * ra;
  e[2] = rho1 * (gamm1 - gammk * ra * rb);
  e[3] = rho1 * rb;
  e[4] = wvno2 - ra * rb;
  for (int m = mmax - 1; m >= 1; m--) { /* llw = 1: no water layer */
    xka = omega / (double)M->a[m - 1];
    xkb = omega / (double)M->b[m - 1];
    t = (double)M->b[m - 1] / omega;
    gammk = 2.0 * t * t;
    gam = gammk * wvno2;
    wvnop = wvno + xka;
    wvnom = fabs(wvno - xka);
    ra = sqrt(wvnop * wvnom);
    wvnop = wvno + xkb;
    wvnom = fabs(wvno - xkb);
    rb = sqrt(wvnop * wvnom);
    double dpth = (double)M->d[m - 1];
    rho1 = (double)M->rho[m - 1];
    double p = ra * dpth, q = rb * dpth;
    /* ---- var ---- */
    double w, x, y, z, cosp, cosq, sinp, sinq, fac, pex = 0.0, sex = 0.0;
    if (wvno < xka) {
      sinp = sin(p);
      w = sinp / ra;
      x = -ra * sinp;
      cosp = cos(p);
    } else if (wvno == xka) {
      cosp = 1.0;
      w = dpth;
      x = 0.0;
    } else {
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
    } else {
      sex = q;
      fac = 0.0;
      if (q < 16) fac = exp(-2.0 * q);
      cosq = (1.0 + fac) * 0.5;
      sinq = (1.0 - fac) * 0.5;
      y = sinq / rb;
      z = rb * sinq;
    }
    double exa = pex + sex;
    double a0 = 0.0;
    if (exa < 60.0) a0 = exp(-exa);
    double cpcq = cosp * cosq, cpy = cosp * y, cpz = cosp * z, cqw = cosq * w, cqx = cosq * x;
    double xy = x * y, xz = x * z, wy = w * y, wz = w * z;
    /* ---- dnka ---- */
    gamm1 = gam - 1.0;
    double twgm1 = gam + gamm1, gmgmk = gam * gammk, gmgm1 = gam * gamm1, gm1sq = gamm1 * gamm1;
    double rho2 = rho1 * rho1, a0pq = a0 - cpcq;
    ca[0][0] = cpcq - 2.0 * gmgm1 * a0pq - gmgmk * xz - wvno2 * gm1sq * wy;
    ca[0][1] = (wvno2 * cpy - cqx) / rho1;
    ca[0][2] = -(twgm1 * a0pq + gammk * xz + wvno2 * gamm1 * wy) / rho1;
    ca[0][3] = (cpz - wvno2 * cqw) / rho1;
    ca[0][4] = -(2.0 * wvno2 * a0pq + xz + wvno2 * wvno2 * wy) / rho2;
    ca[1][0] = (gmgmk * cpz - gm1sq * cqw) * rho1;
    ca[1][1] = cpcq;
    ca[1][2] = gammk * cpz - gamm1 * cqw;
    ca[1][3] = -wz;
    ca[1][4] = ca[0][3];
    ca[3][0] = (gm1sq * cpy - gmgmk * cqx) * rho1;
    ca[3][1] = -xy;
    ca[3][2] = gamm1 * cpy - gammk * cqx;
    ca[3][3] = ca[1][1];
    ca[3][4] = ca[0][1];
    ca[4][0] = -(2.0 * gmgmk * gm1sq * a0pq + gmgmk * gmgmk * xz + gm1sq * gm1sq * wy) * rho2;
    ca[4][1] = ca[3][0];
    ca[4][2] = -(gammk * gamm1 * twgm1 * a0pq + gam * gammk * gammk * xz + gamm1 * gm1sq * wy) * rho1;
    ca[4][3] = ca[1][0];
    ca[4][4] = ca[0][0];
    double tt = -2.0 * wvno2;
    ca[2][0] = tt * ca[4][2];
    ca[2][1] = tt * ca[3][2];
    ca[2][2] = a0 + 2.0 * (cpcq - ca[0][0]);
    ca[2][3] = tt * ca[1][2];
    ca[2][4] = tt * ca[0][2];
    for (int i = 0; i < 5; i++) {
      double cr = 0.0;
      for (int j = 0; j < 5; j++) cr = cr + e[j] * ca[j][i];
      ee[i] = cr;
    }
    /* ---- normc (:989-1014); its log() result is never used by dltar4 ---- */
    double t1 = 0.0;
    for (int i = 0; i < 5; i++)
      if (fabs(ee[i]) > t1) t1 = fabs(ee[i]);
    if (t1 < 1.e-40) t1 = 1.0;
    for (int i = 0; i < 5; i++) e[i] = ee[i] / t1;
  }
  return e[0];
}

static const double TWOPI = 2.0 * 3.141592653589793;
static double sgn(double x) { return copysign(1.0, x); }

/* inv/surfdisp96.f:551-668 (half :670-680 inlined) */
static double nevill(const model *M, double t, double c1, double c2, double del1, double del2) {
  double x[20], y[20];
  double omega = TWOPI / t;
  double c3 = 0.5 * (c1 + c2);
  double del3 = dltar4(M, omega / c3, omega);
  int nev = 1, nctrl = 1, m = 1;
  for (;;) {
    nctrl++;
    if (nctrl >= 100) break;
    if (c3 < fmin(c1, c2) || c3 > fmax(c1, c2)) {
      nev = 0;
      c3 = 0.5 * (c1 + c2);
      del3 = dltar4(M, omega / c3, omega);
    }
    double s13 = del1 - del3, s32 = del3 - del2;
    if (sgn(del3) * sgn(del1) < 0.0) {
      c2 = c3;
      del2 = del3;
    } else {
      c1 = c3;
      del1 = del3;
    }
    if (fabs(c1 - c2) <= 1.e-6 * c1) break;
    if (sgn(s13) != sgn(s32)) nev = 0;
    double ss1 = fabs(del1), s1 = (double)0.01f * ss1;
    double ss2 = fabs(del2), s2 = (double)0.01f * ss2;
    if (s1 > ss2 || s2 > ss1 || nev == 0) {
      c3 = 0.5 * (c1 + c2);
      del3 = dltar4(M, omega / c3, omega);
      nev = 1;
      m = 1;
    } else {
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
      int bad = 0;
      for (int kk = 1; kk <= m; kk++) {
        int j = m - kk + 1;
        double denom = y[m] - y[j - 1];
        if (fabs(denom) < 1.0e-10 * fabs(y[m])) {
          bad = 1;
          break;
        }
        x[j - 1] = (-y[j - 1] * x[j] + y[m] * x[j - 1]) / denom;
      }
      if (!bad) {
        c3 = x[0];
        del3 = dltar4(M, omega / c3, omega);
        nev = 2;
        m = m + 1;
        if (m > 10) m = 10;
      } else {
        c3 = 0.5 * (c1 + c2);
        del3 = dltar4(M, omega / c3, omega);
        nev = 1;
        m = 1;
      }
    }
  }
  return c3;
}

/* inv/surfdisp96.f:384-476 ; returns iret */
static int getsol(model *M, double t1, double *c1io, double clow, double dc, double cm, float betmx,
                  int ifirst) {
  double c1 = *c1io, c2, del1, del2;
  double omega = TWOPI / t1;
  del1 = dltar4(M, omega / c1, omega);
  if (ifirst == 1) M->del1st = del1;
  double plmn = sgn(M->del1st) * sgn(del1);
  int idir = (ifirst == 1) ? 1 : (plmn >= 0.0 ? 1 : -1);
  for (;;) {
    c2 = (idir > 0) ? c1 + dc : c1 - dc;
    if (c2 <= clow) {
      idir = 1;
      c1 = clow;
      continue;
    }
    omega = TWOPI / t1;
    del2 = dltar4(M, omega / c2, omega);
    if (sgn(del1) != sgn(del2)) break;
    c1 = c2;
    del1 = del2;
    if (c1 < cm) return -1;
    if (c1 >= ((double)betmx + dc)) return -1;
  }
  c1 = nevill(M, t1, c1, c2, del1, del2);
  *c1io = c1;
  if (c1 > (double)betmx) return -1;
  return 1;
}

/* inv/surfdisp96.f:361-382, all fp32 */
static float gtsolh(float a, float b) {
  float c = 0.95f * b;
  for (int i = 0; i < 5; i++) {
    float gamma = b / a;
    float kappa = c / b;
    float k2 = kappa * kappa;
    float gk2 = (gamma * kappa) * (gamma * kappa);
    float fac1 = sqrtf(1.0f - gk2);
    float fac2 = sqrtf(1.0f - k2);
    float fr = (2.0f - k2) * (2.0f - k2) - 4.0f * fac1 * fac2;
    float frp = -4.0f * (2.0f - k2) * kappa + 4.0f * fac2 * gamma * gamma * kappa / fac1 +
                4.0f * fac1 * kappa / fac2;
    frp = frp / b;
    c = c - fr / frp;
  }
  return c;
}

/* inv/surfdisp96.f:52-354 for iflsph=1, iwave=2 (Rayleigh), mode=1, igr=0 */
int orc_surfdisp96(const float *thk, const float *vp, const float *vs, const float *rho, int nlayer,
                   int kmax, const double *t, double *cg) {
  model M;
  int mmax = nlayer;
  M.mmax = mmax;
  M.del1st = 0.0;
  for (int i = 0; i < mmax; i++) {
    M.b[i] = vs[i];
    M.a[i] = vp[i];
    M.d[i] = thk[i];
    M.rho[i] = rho[i];
  }
  /* sphere(0,0): inv/surfdisp96.f:510-534 */
  {
    double ar = 6370.0, dr = 0.0, r0 = ar;
    M.d[mmax - 1] = 1.0f;
    for (int i = 0; i < mmax; i++) {
      M.dtp[i] = M.d[i];
      M.rtp[i] = M.rho[i];
    }
    for (int i = 0; i < mmax; i++) {
      dr = dr + (double)M.d[i];
      double r1 = ar - dr;
      double z0 = ar * log(ar / r0), z1 = ar * log(ar / r1);
      M.d[i] = (float)(z1 - z0);
      double tmp = (ar + ar) / (r0 + r1);
      M.a[i] = (float)((double)M.a[i] * tmp);
      M.b[i] = (float)((double)M.b[i] * tmp);
      M.btp[i] = (float)tmp;
      r0 = r1;
    }
    M.d[mmax - 1] = 0.0f;
  }
  float betmx = -1.e20f, betmn = 1.e20f;
  int jmn = 1, jsol = 1;
  for (int i = 0; i < mmax; i++) {
    if (M.b[i] > 0.01f && M.b[i] < betmn) {
      betmn = M.b[i];
      jmn = i + 1;
      jsol = 1;
    } else if (M.b[i] <= 0.01f && M.a[i] < betmn) {
      betmn = M.a[i];
      jmn = i + 1;
      jsol = 0;
    }
    if (M.b[i] > betmx) betmx = M.b[i];
  }
  /* sphere(2,1): Rayleigh density mapping, :536-545 */
  for (int i = 0; i < mmax; i++) M.rho[i] = M.rtp[i] * powf(M.btp[i], -2.275f);
  M.d[mmax - 1] = 0.0f;
  float ddc = 0.005f, sone = 1.5f;
  double onea = (double)sone;
  float cc1 = (jsol == 0) ? betmn : gtsolh(M.a[jmn - 1], M.b[jmn - 1]);
  cc1 = .95f * cc1;
  cc1 = .90f * cc1;
  double cc = (double)cc1, dc = fabs((double)ddc), c1 = cc, cm = cc, clow;
  double c[ORC_NP];
  for (int i = 0; i < kmax; i++) c[i] = 0.0;
  int k;
  for (k = 1; k <= kmax; k++) {
    double t1 = t[k - 1];
    int ifirst;
    if (k == 1) {
      c1 = cc;
      clow = cc;
      ifirst = 1;
    } else {
      ifirst = 0;
      c1 = c[k - 2] - onea * dc;
      clow = cm;
    }
    int iret = getsol(&M, t1, &c1, clow, dc, cm, betmx, ifirst);
    if (iret == -1) break;
    c[k - 1] = c1;
    cg[k - 1] = (double)(float)c[k - 1];
  }
  int nok = k - 1;
  for (; k <= kmax; k++) cg[k - 1] = 0.0; /* :1750-1770 */
  return nok;
}

/* inv/CalSurfG.f90:2317-2376 ; returns rmax */
int orc_refine_layers(float minthk0, int mmax, const float *dep, const float *vp, const float *vs,
                      const float *rho, float *rthk, float *rvp, float *rvs, float *rrho) {
  int k = 0;
  for (int i = 1; i <= mmax - 1; i++) {
    float thk = dep[i] - dep[i - 1];
    float minthk = thk / minthk0;
    int nsub = (int)((thk + 1.0e-4f) / minthk) + 1;
    float newthk = thk / (float)nsub;
    for (int j = 1; j <= nsub; j++) {
      rthk[k] = newthk;
      rvp[k] = vp[i - 1] + (float)(2 * j - 1) * (vp[i] - vp[i - 1]) / (float)(2 * nsub);
      rvs[k] = vs[i - 1] + (float)(2 * j - 1) * (vs[i] - vs[i - 1]) / (float)(2 * nsub);
      rrho[k] = rho[i - 1] + (float)(2 * j - 1) * (rho[i] - rho[i - 1]) / (float)(2 * nsub);
      k++;
    }
  }
  rthk[k] = 0.0f;
  rvp[k] = vp[mmax - 1];
  rvs[k] = vs[mmax - 1];
  rrho[k] = rho[mmax - 1];
  return k + 1;
}

void orc_brocher(float vs, float *vp, float *rho) { /* inv/CalSurfG.f90:49-53 */
  float v2 = vs * vs, v3 = v2 * vs, v4 = v3 * vs;
  float p = 0.9409f + 2.0947f * vs - 0.8206f * v2 + 0.2683f * v3 - 0.0251f * v4;
  float p2 = p * p, p3 = p2 * p, p4 = p3 * p, p5 = p4 * p;
  *vp = p;
  *rho = 1.6612f * p - 0.4721f * p2 + 0.0671f * p3 - 0.0043f * p4 + 0.000106f * p5;
}

static void curve(float minthk, int nz, const float *depz, const float *vp, const float *vs,
                  const float *rho, int kmax, const double *t, double *cg) {
  float rthk[ORC_NL], rvp[ORC_NL], rvs[ORC_NL], rrho[ORC_NL];
  int rmax = orc_refine_layers(minthk, nz, depz, vp, vs, rho, rthk, rvp, rvs, rrho);
  orc_surfdisp96(rthk, rvp, rvs, rrho, rmax, kmax, t, cg);
}

/* inv/CalSurfG.f90:1-139 */
int orc_depthkernel(int nx, int ny, int nz, const float *vel, int kmax, const double *t,
                    const float *depz, float minthk, double *pv, double *svs, double *svp,
                    double *srho) {
  const float dln = 0.01f;
  size_t ncol = (size_t)nx * ny;
  int nfail = 0;
  for (int jj = 0; jj < ny; jj++)
    for (int ii = 0; ii < nx; ii++) {
      size_t col = (size_t)jj * nx + ii;
      float vsz[ORC_NL], vpz[ORC_NL], rhoz[ORC_NL], vsm[ORC_NL], vpm[ORC_NL], rhom[ORC_NL];
      double cg0[ORC_NP], cg1[ORC_NP], cg2[ORC_NP];
      for (int k = 0; k < nz; k++) {
        vsz[k] = vel[((size_t)k * ny + jj) * nx + ii];
        orc_brocher(vsz[k], &vpz[k], &rhoz[k]);
        vsm[k] = vsz[k];
        vpm[k] = vpz[k];
        rhom[k] = rhoz[k];
      }
      curve(minthk, nz, depz, vpz, vsz, rhoz, kmax, t, cg0);
      for (int k = 0; k < kmax; k++) {
        pv[(size_t)k * ncol + col] = cg0[k];
        if (cg0[k] == 0.0) nfail++;
      }
      if (!svs) continue;
      for (int i = 0; i < nz; i++) {
        float *arr[3] = {vsm, vpm, rhom};
        const float *base[3] = {vsz, vpz, rhoz};
        double *out[3] = {svs, svp, srho};
        for (int q = 0; q < 3; q++) {
          float b0 = base[q][i];
          arr[q][i] = b0 - 0.5f * dln * b0;
          curve(minthk, nz, depz, vpm, vsm, rhom, kmax, t, cg1);
          arr[q][i] = b0 + 0.5f * dln * b0;
          curve(minthk, nz, depz, vpm, vsm, rhom, kmax, t, cg2);
          arr[q][i] = b0;
          for (int k = 0; k < kmax; k++)
            out[q][((size_t)i * kmax + k) * ncol + col] = (cg2[k] - cg1[k]) / (double)(dln * b0);
        }
      }
    }
  return nfail;
}
