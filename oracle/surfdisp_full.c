/* surfdisp_full.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * CPU restatement of the reference's WHOLE surfdisp96 (inv/surfdisp96.f:52-354), i.e. with every value of its
 * arguments: iflsph = 0 / 1 (flat / flattened spherical model), iwave = 1 / 2 (Love: dltar1, :704-763; Rayleigh:
 * dltar4, :767-865, including the water-layer branch :844-860), mode >= 1 (the loop over higher modes :217-349)
 * and igr = 0 / > 0 (phase velocity / group velocity from the roots at T/(1+h) and T/(1-h), :226-233, :276-304).
 * disp.c holds the one combination the reference's own callers use (iflsph=1, iwave=2, mode=1, igr=0) and stays the
 * checker of the hot path; tests/test_surfdisp_full_cpu.py shows the two agree bit for bit on that combination and
 * pins this file against the flang build of the unmodified reference (oracle/_ref, ref_surfdisp96_full).
 * Types follow the F77 implicit typing of the reference: cc1, betmx, betmn, t1a, t1b, cc0, gvel, ... are fp32.
 */
#include "oracle.h"
#include <math.h>
#include <string.h>

typedef struct {
  int mmax, llw;
  float d[ORC_NL], a[ORC_NL], b[ORC_NL], rho[ORC_NL], rtp[ORC_NL], dtp[ORC_NL], btp[ORC_NL];
  float dhalf;   /* SAVE dhalf, inv/surfdisp96.f:509 */
  double del1st; /* SAVE del1st, inv/surfdisp96.f:409 */
} fmodel;

static const double TWOPI = 2.0 * 3.141592653589793;
static double sgn(double x) { return copysign(1.0, x); }

/* var, inv/surfdisp96.f:868-985 (the rescaled cosq, y, z of its last lines are never used by a caller) */
typedef struct {
  double w, cosp, exa, a0, cpcq, cpy, cpz, cqw, cqx, xy, xz, wy, wz;
} varout;

static void var(double p, double q, double ra, double rb, double wvno, double xka, double xkb, double dpth, varout *o) {
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
  o->exa = pex + sex;
  o->a0 = 0.0;
  if (o->exa < 60.0) o->a0 = exp(-o->exa);
  o->w = w;
  o->cosp = cosp;
  o->cpcq = cosp * cosq;
  o->cpy = cosp * y;
  o->cpz = cosp * z;
  o->cqw = cosq * w;
  o->cqx = cosq * x;
  o->xy = x * y;
  o->xz = x * z;
  o->wy = w * y;
  o->wz = w * z;
}

/* dltar4 with the water layer, inv/surfdisp96.f:767-865 (dnka :1018-1062, normc :989-1014) */
static double dltar4(const fmodel *M, double wvno, double omga) {
  double e[5], ee[5], ca[5][5];
  const int mmax = M->mmax;
  double omega = omga;
  if (omega < 1.0e-4) omega = 1.0e-4;
  const double wvno2 = wvno * wvno;
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
  varout v;
  for (int m = mmax - 1; m >= M->llw; m--) {
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
    const double dpth = (double)M->d[m - 1];
    rho1 = (double)M->rho[m - 1];
    const double p = ra * dpth, q = rb * dpth;
    var(p, q, ra, rb, wvno, xka, xkb, dpth, &v);
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
    for (int i = 0; i < 5; i++) {
      double cr = 0.0;
      for (int j = 0; j < 5; j++) cr = cr + e[j] * ca[j][i];
      ee[i] = cr;
    }
    double t1 = 0.0;
    for (int i = 0; i < 5; i++)
      if (fabs(ee[i]) > t1) t1 = fabs(ee[i]);
    if (t1 < 1.e-40) t1 = 1.0;
    for (int i = 0; i < 5; i++) e[i] = ee[i] / t1;
  }
  if (M->llw != 1) { /* water layer on top, :844-860 */
    xka = omega / (double)M->a[0];
    wvnop = wvno + xka;
    wvnom = fabs(wvno - xka);
    ra = sqrt(wvnop * wvnom);
    const double dpth = (double)M->d[0];
    rho1 = (double)M->rho[0];
    const double p = ra * dpth;
    const double znul = 1.0e-05;
    var(p, znul, ra, znul, wvno, xka, znul, dpth, &v);
    const double w0 = -rho1 * v.w;
    return v.cosp * e[0] + w0 * e[1];
  }
  return e[0];
}

/* dltar1: SH period equation, inv/surfdisp96.f:704-763 */
static double dltar1(const fmodel *M, double wvno, double omega) {
  const int mmax = M->mmax;
  double beta1 = (double)M->b[mmax - 1];
  double rho1 = (double)M->rho[mmax - 1];
  double xkb = omega / beta1;
  double wvnop = wvno + xkb, wvnom = fabs(wvno - xkb);
  double rb = sqrt(wvnop * wvnom);
  double e1 = rho1 * rb;
  double e2 = 1.0 / (beta1 * beta1);
  for (int m = mmax - 1; m >= M->llw; m--) {
    beta1 = (double)M->b[m - 1];
    rho1 = (double)M->rho[m - 1];
    const double xmu = rho1 * beta1 * beta1;
    xkb = omega / beta1;
    wvnop = wvno + xkb;
    wvnom = fabs(wvno - xkb);
    rb = sqrt(wvnop * wvnom);
    const double q = (double)M->d[m - 1] * rb;
    double y = 0, z = 0, cosq = 0, sinq, fac;
    if (wvno < xkb) {
      sinq = sin(q);
      y = sinq / rb;
      z = -rb * sinq;
      cosq = cos(q);
    } else if (wvno == xkb) {
      cosq = 1.0;
      y = (double)M->d[m - 1];
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

static double dltar(const fmodel *M, double wvno, double omega, int kk) { /* :684-700 */
  return kk == 1 ? dltar1(M, wvno, omega) : dltar4(M, wvno, omega);
}

/* inv/surfdisp96.f:551-668 (half :670-680 inlined) */
static double nevill(const fmodel *M, double t, double c1, double c2, double del1, double del2, int ifunc) {
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
    if (s1 > ss2 || s2 > ss1 || nev == 0) {
      c3 = 0.5 * (c1 + c2);
      del3 = dltar(M, omega / c3, omega, ifunc);
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
        const int j = m - kk + 1;
        const double denom = y[m] - y[j - 1];
        if (fabs(denom) < 1.0e-10 * fabs(y[m])) {
          bad = 1;
          break;
        }
        x[j - 1] = (-y[j - 1] * x[j] + y[m] * x[j - 1]) / denom;
      }
      if (!bad) {
        c3 = x[0];
        del3 = dltar(M, omega / c3, omega, ifunc);
        nev = 2;
        m = m + 1;
        if (m > 10) m = 10;
      } else {
        c3 = 0.5 * (c1 + c2);
        del3 = dltar(M, omega / c3, omega, ifunc);
        nev = 1;
        m = 1;
      }
    }
  }
  return c3;
}

/* inv/surfdisp96.f:384-476 ; returns iret */
static int getsol(fmodel *M, double t1, double *c1io, double clow, double dc, double cm, float betmx, int ifunc,
                  int ifirst) {
  double c1 = *c1io, c2, del1, del2;
  double omega = TWOPI / t1;
  del1 = dltar(M, omega / c1, omega, ifunc);
  if (ifirst == 1) M->del1st = del1;
  const double plmn = sgn(M->del1st) * sgn(del1);
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
    if (c1 < cm) {
      *c1io = c1;
      return -1;
    }
    if (c1 >= ((double)betmx + dc)) {
      *c1io = c1;
      return -1;
    }
  }
  c1 = nevill(M, t1, c1, c2, del1, del2, ifunc);
  *c1io = c1;
  if (c1 > (double)betmx) return -1;
  return 1;
}

/* inv/surfdisp96.f:361-382, all fp32 */
static float gtsolh(float a, float b) {
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

/* sphere, inv/surfdisp96.f:480-547 */
static void sphere(fmodel *M, int ifunc, int iflag) {
  const int mmax = M->mmax;
  const double ar = 6370.0;
  double dr = 0.0, r0 = ar;
  M->d[mmax - 1] = 1.0f;
  if (iflag == 0) {
    for (int i = 0; i < mmax; i++) {
      M->dtp[i] = M->d[i];
      M->rtp[i] = M->rho[i];
    }
    for (int i = 0; i < mmax; i++) {
      dr = dr + (double)M->d[i];
      const double r1 = ar - dr;
      const double z0 = ar * log(ar / r0), z1 = ar * log(ar / r1);
      M->d[i] = (float)(z1 - z0);
      const double tmp = (ar + ar) / (r0 + r1);
      M->a[i] = (float)((double)M->a[i] * tmp);
      M->b[i] = (float)((double)M->b[i] * tmp);
      M->btp[i] = (float)tmp;
      r0 = r1;
    }
    M->dhalf = M->d[mmax - 1];
  } else {
    M->d[mmax - 1] = M->dhalf;
    for (int i = 0; i < mmax; i++) {
      if (ifunc == 1) { /* btp**(-5): the flang build forms the reciprocal of the positive integer power */
        const float x = M->btp[i];
        const float x2 = x * x;
        M->rho[i] = M->rtp[i] * (1.0f / (x2 * x2 * x));
      } else if (ifunc == 2)
        M->rho[i] = M->rtp[i] * powf(M->btp[i], -2.275f);
    }
  }
  M->d[mmax - 1] = 0.0f;
}

/* inv/surfdisp96.f:52-354.  Returns the number of periods of the last mode computed that have a root (cg != 0 is not the
 * criterion: the reference zeroes cg(k..kmax) at the first failure, :342-348). */
int orc_surfdisp96_full(const float *thk, const float *vp, const float *vs, const float *rho, int nlayer, int iflsph,
                        int iwave, int mode, int igr, int kmax, const double *t, double *cg) {
  fmodel M;
  memset(&M, 0, sizeof M);
  const int mmax = nlayer;
  M.mmax = mmax;
  for (int i = 0; i < mmax; i++) {
    M.b[i] = vs[i];
    M.a[i] = vp[i];
    M.d[i] = thk[i];
    M.rho[i] = rho[i];
  }
  int idispl = 0, idispr = 0;
  if (iwave == 1) idispl = kmax;
  else if (iwave == 2) idispr = kmax;
  const float sone0 = 1.500f, ddc0 = 0.005f, h0 = 0.005f;
  M.llw = 1;
  if (M.b[0] <= 0.0f) M.llw = 2;
  const double one = 1.0e-2;
  if (iflsph == 1) sphere(&M, 0, 0);
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
  int nok = 0;
  for (int ifunc = 1; ifunc <= 2; ifunc++) {
    if (ifunc == 1 && idispl <= 0) continue;
    if (ifunc == 2 && idispr <= 0) continue;
    if (iflsph == 1) sphere(&M, ifunc, 1);
    const float ddc = ddc0, h = h0;
    float sone = sone0;
    if (sone < 0.01f) sone = 2.0f;
    const double onea = (double)sone;
    float cc1 = (jsol == 0) ? betmn : gtsolh(M.a[jmn - 1], M.b[jmn - 1]);
    cc1 = .95f * cc1;
    cc1 = .90f * cc1;
    const double cc = (double)cc1;
    const double dc = fabs((double)ddc);
    double c1 = cc, clow = cc;
    const double cm = cc;
    double c[ORC_NP], cb[ORC_NP];
    for (int i = 0; i < kmax; i++) cb[i] = c[i] = 0.0;
    int ift = 999;
    for (int iq = 1; iq <= mode; iq++) {
      const int is = 1, ie = kmax;
      int k;
      int failed = 0;
      for (k = is; k <= ie; k++) {
        if (k >= ift) {
          failed = 1;
          break;
        }
        double t1 = t[k - 1];
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
          c1 = c[is - 1] + one * dc;
          clow = c1;
          ifirst = 1;
        } else if (k > is && iq > 1) {
          ifirst = 0;
          clow = c[k - 1] + one * dc;
          c1 = c[k - 2];
          if (c1 < clow) c1 = clow;
        } else {
          ifirst = 0;
          c1 = c[k - 2] - onea * dc;
          clow = cm;
        }
        int iret = getsol(&M, t1, &c1, clow, dc, cm, betmx, ifunc, ifirst);
        if (iret == -1) {
          failed = 1;
          break;
        }
        c[k - 1] = c1;
        if (igr > 0) {
          t1 = (double)t1b;
          ifirst = 0;
          clow = cb[k - 1] + one * dc;
          c1 = c1 - onea * dc;
          iret = getsol(&M, t1, &c1, clow, dc, cm, betmx, ifunc, ifirst);
          if (iret == -1) c1 = c[k - 1];
          cb[k - 1] = c1;
        } else {
          c1 = 0.0;
        }
        const float cc0 = (float)c[k - 1];
        const float cc1b = (float)c1;
        if (igr == 0) {
          cg[k - 1] = (double)cc0;
        } else {
          const float gvel = (1 / t1a - 1 / t1b) / (1 / (t1a * cc0) - 1 / (t1b * cc1b));
          cg[k - 1] = (double)gvel;
        }
      }
      nok = k - 1;
      if (failed) { /* :1700-1770 */
        ift = k;
        for (int i = k; i <= ie; i++) cg[i - 1] = 0.0;
      }
    }
  }
  return nok;
}
