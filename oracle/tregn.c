/* tregn.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the TI Rayleigh-wave eigenfunction / partial-derivative path that feeds the joint
 * inversion's azimuthal depth kernels Lsen_Gsc:
 *   depthkernelTI (inv/depthkernelTI.f90:2-106)  ->  tregn96 (inv/tregn96.f:52-722), live branch only:
 *   Rayleigh, fundamental mode, solid layers (TL,TN > 0), flat-earth transformed model (iflsph=1), causal-Q
 *   phase correction on (dogam=.true., Qp=150, Qs=50, fref=1 Hz), source and receiver at depth 0.
 *
 * The per-layer 6x6 compound (Dunkin) matrix, which the reference spells out entry by entry
 * (dnka_tregn, inv/tregn96.f:1992-2988), is formed here from its factorisation instead: the 4x4 layer
 * matrix of hska (inv/tregn96.f:3477) is A = U*H*W with U, W built from the two eigenvectors and
 * H = blockdiag(H_P, H_SV) holding the cosh/sinh terms, so by Cauchy-Binet C2(A) = C2(U)*C2(H)*C2(W); C2(H)
 * is known in closed form (det H_P = det H_SV = the common scale factor, cross block = H_P (x) H_SV), which
 * keeps the cancellation analytic exactly like the reference's expanded formulas.  Parity with the
 * reference is therefore to rounding (tests: rel 1e-6 on Lsen_Gsc), not bit-for-bit.
 */
#include <math.h>
#include <string.h>

#include "oracle.h"

typedef struct { double re, im; } cx;
static cx C(double re, double im) { cx z = {re, im}; return z; }
static cx cadd(cx a, cx b) { return C(a.re + b.re, a.im + b.im); }
static cx csub(cx a, cx b) { return C(a.re - b.re, a.im - b.im); }
static cx cmul(cx a, cx b) { return C(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
static cx cscale(cx a, double s) { return C(a.re * s, a.im * s); }
static cx cneg(cx a) { return C(-a.re, -a.im); }
static double cabs_(cx a) { return hypot(a.re, a.im); }
static cx cdiv(cx a, cx b) {
  double den = b.re * b.re + b.im * b.im;
  return C((a.re * b.re + a.im * b.im) / den, (a.im * b.re - a.re * b.im) / den);
}
static cx csqrt_(cx a) { /* principal branch */
  double r = cabs_(a);
  if (r == 0.0) return C(0.0, 0.0);
  double s = sqrt(0.5 * (r + fabs(a.re)));
  if (a.re >= 0.0) return C(s, a.im / (2.0 * s));
  return C(fabs(a.im) / (2.0 * s), a.im >= 0.0 ? s : -s);
}
static cx cexp_(cx a) { double e = exp(a.re); return C(e * cos(a.im), e * sin(a.im)); }

typedef struct { /* one solid TI layer at (omega, wvno): inv/tregn96.f:3166 gettiegn */
  cx rp, rsv, x[4][2], np, nsv;
} eig;

static void layer_eig(double TA, double TC, double TF, double TL, double rho, double omg, double wvn, eig *g) {
  const double wvno2 = wvn * wvn;
  const double a = wvn * TF / TC, b = 1.0 / TC, c = -rho * omg * omg + wvn * wvn * (TA - TF * TF / TC);
  const double d = -wvn, e = 1.0 / TL, f = -rho * omg * omg;
  const double ddef = wvn * wvn - rho * omg * omg / TL, aabc = wvn * wvn * TA / TC - rho * omg * omg / TC;
  const cx bb = C(2.0 * a * d + e * c + f * b, 0.0), cc = C(ddef * aabc, 0.0);
  cx s = csqrt_(csub(cmul(bb, bb), cscale(cc, 4.0)));
  if (s.im < 0.0) s = cneg(s);
  cx l1, l2;
  if (bb.re < 0.0 && s.re < 0.0) {
    l2 = cscale(csub(bb, s), 0.5);
    l1 = cabs_(l2) > 0.0 ? cdiv(cc, l2) : cscale(cadd(bb, s), 0.5);
  } else {
    l1 = cscale(cadd(bb, s), 0.5);
    l2 = cabs_(l1) > 0.0 ? cdiv(cc, l1) : cscale(csub(bb, s), 0.5);
  }
  if (cabs_(csub(C(wvno2, 0), l2)) < cabs_(csub(C(wvno2, 0), l1))) { cx t = l1; l1 = l2; l2 = t; }
  g->rp = csqrt_(l1);
  g->rsv = csqrt_(l2);
  if (g->rp.re < 0.0) g->rp = cneg(g->rp);
  if (g->rsv.re < 0.0) g->rsv = cneg(g->rsv);
  cx x12 = C(b * d - a * e, 0), x22 = csub(cscale(l2, b), C(e * (b * c + a * a), 0));
  cx x32 = csub(l2, C(a * d + c * e, 0)), x42 = cadd(cscale(l2, -a), C(d * (b * c + a * a), 0));
  cx x11 = cadd(cscale(l1, -e), C(b * (d * d + e * f), 0)), x21 = C(b * d - a * e, 0);
  cx x31 = csub(cscale(l1, d), C(a * (d * d + e * f), 0)), x41 = cneg(csub(l1, C(a * d + b * f, 0)));
  if (wvn != 0.0) {
    cx z = cdiv(C(wvn, 0), x11);
    x11 = cmul(x11, z); x21 = cmul(x21, z); x31 = cmul(x31, z); x41 = cmul(x41, z);
    z = cdiv(C(wvn, 0), x22);
    x12 = cmul(x12, z); x22 = cmul(x22, z); x32 = cmul(x32, z); x42 = cmul(x42, z);
  }
  g->x[0][0] = x11; g->x[1][0] = x21; g->x[2][0] = x31; g->x[3][0] = x41;
  g->x[0][1] = x12; g->x[1][1] = x22; g->x[2][1] = x32; g->x[3][1] = x42;
  g->np = csub(cmul(x11, x41), cmul(x21, x31));
  g->nsv = csub(cmul(x12, x42), cmul(x22, x32));
}

typedef struct { cx cosp, rsinp, sinpr, cosq, rsinq, sinqr; double pex, svex; } trig;

/* inv/tregn96.f:3363 varsv, solid branch: cosh/sinh of nu*d scaled by exp(-Re(nu*d)) */
static void layer_trig(const eig *g, double dm, trig *t) {
  const cx p = cscale(g->rp, dm), q = cscale(g->rsv, dm);
  t->pex = p.re;
  t->svex = q.re;
  for (int w = 0; w < 2; w++) {
    const cx arg = w ? q : p, nu = w ? g->rsv : g->rp;
    const cx epp = C(cos(arg.im) / 2.0, sin(arg.im) / 2.0), epm = C(epp.re, -epp.im);
    const double fac = arg.re < 15.0 ? exp(-2.0 * arg.re) : 0.0;
    const cx co = cadd(epp, cscale(epm, fac)), si = csub(epp, cscale(epm, fac));
    const cx rs = cmul(nu, si);
    const cx sr = (fabs(arg.re) < 1.0e-5 && cabs_(nu) < 1.0e-5) ? C(dm, 0) : cdiv(si, nu);
    if (w) { t->cosq = co; t->rsinq = rs; t->sinqr = sr; } else { t->cosp = co; t->rsinp = rs; t->sinpr = sr; }
  }
}

/* factors of the layer matrix A = U * blockdiag(H1/np, H2/nsv) * W (hska, inv/tregn96.f:3477):
 * U columns (1a,1b,2a,2b), W rows (1a,1b,2a,2b) */
static void layer_factors(const eig *g, cx U[4][4], cx W[4][4]) {
  memset(U, 0, sizeof(cx) * 16);
  memset(W, 0, sizeof(cx) * 16);
  for (int md = 0; md < 2; md++) {
    const cx x1 = g->x[0][md], x2 = g->x[1][md], x3 = g->x[2][md], x4 = g->x[3][md];
    U[0][2 * md] = x1; U[2][2 * md] = x3;          /* rows 1,3 take the "a" column */
    U[1][2 * md + 1] = x2; U[3][2 * md + 1] = x4;  /* rows 2,4 the "b" column      */
    W[2 * md][0] = x4; W[2 * md][2] = cneg(x2);
    W[2 * md + 1][1] = cneg(x3); W[2 * md + 1][3] = x1;
  }
}

static const int PAIR[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};

/* 5x5 reduced compound matrix of a solid layer (= CA of dnka_tregn), scaled by exp(-(pex+svex)) */
static void layer_compound(const eig *g, const trig *t, cx CA[5][5]) {
  const double ex = t->pex + t->svex, dfac = ex > 35.0 ? 0.0 : exp(-ex);
  cx U[4][4], W[4][4], CU[6][6], CW[6][6], CH[6][6], T[6][6], R[6][6];
  layer_factors(g, U, W);
  for (int r = 0; r < 6; r++)
    for (int c = 0; c < 6; c++) {
      const int i = PAIR[r][0], j = PAIR[r][1], k = PAIR[c][0], l = PAIR[c][1];
      CU[r][c] = csub(cmul(U[i][k], U[j][l]), cmul(U[i][l], U[j][k]));
      CW[r][c] = csub(cmul(W[i][k], W[j][l]), cmul(W[i][l], W[j][k]));
      CH[r][c] = C(0, 0);
    }
  /* H1 = [[cosp, sinpr],[rsinp, cosp]]/np, H2 = [[cosq, rsinq],[sinqr, cosq]]/nsv */
  const cx H1[2][2] = {{cdiv(t->cosp, g->np), cdiv(t->sinpr, g->np)}, {cdiv(t->rsinp, g->np), cdiv(t->cosp, g->np)}};
  const cx H2[2][2] = {{cdiv(t->cosq, g->nsv), cdiv(t->rsinq, g->nsv)}, {cdiv(t->sinqr, g->nsv), cdiv(t->cosq, g->nsv)}};
  CH[0][0] = cdiv(C(dfac, 0), cmul(g->np, g->np));      /* det H1: cosh^2 - sinh^2 = 1, analytic */
  CH[5][5] = cdiv(C(dfac, 0), cmul(g->nsv, g->nsv));    /* det H2 */
  for (int i = 0; i < 2; i++)        /* pairs (1i,2j) are compound indices 1 + 2*i + j */
    for (int j = 0; j < 2; j++)
      for (int k = 0; k < 2; k++)
        for (int l = 0; l < 2; l++) CH[1 + 2 * i + j][1 + 2 * k + l] = cmul(H1[i][k], H2[j][l]);
  for (int r = 0; r < 6; r++)
    for (int c = 0; c < 6; c++) {
      cx s = C(0, 0);
      for (int k = 0; k < 6; k++) s = cadd(s, cmul(CU[r][k], CH[k][c]));
      T[r][c] = s;
    }
  for (int r = 0; r < 6; r++)
    for (int c = 0; c < 6; c++) {
      cx s = C(0, 0);
      for (int k = 0; k < 6; k++) s = cadd(s, cmul(T[r][k], CW[k][c]));
      R[r][c] = s;
    }
  /* 6 -> 5 reduction (pair 23 dropped, row 3 doubled): inv/tregn96.f:2937-2961 */
  static const int IDX[5] = {0, 1, 2, 4, 5};
  for (int i = 0; i < 5; i++)
    for (int j = 0; j < 5; j++) {
      CA[i][j] = cscale(R[IDX[i]][IDX[j]], i == 2 ? 2.0 : 1.0);
      if (i == 2 && j == 2) CA[i][j].re -= dfac;
    }
}

/* E, E^-1 of a solid layer: inv/tregn96.f:3027-3076 (evalg) */
static void layer_E(const eig *g, cx E[4][4], cx EI[4][4]) {
  const cx rp = g->rp, rsv = g->rsv;
  const cx x11 = g->x[0][0], x21 = g->x[1][0], x31 = g->x[2][0], x41 = g->x[3][0];
  const cx x12 = g->x[0][1], x22 = g->x[1][1], x32 = g->x[2][1], x42 = g->x[3][1];
  const cx dp = cscale(cmul(rp, g->np), 2.0), ds = cscale(cmul(rsv, g->nsv), 2.0);
  EI[0][0] = cdiv(cmul(x41, rp), dp); EI[0][1] = cdiv(cneg(x31), dp);
  EI[0][2] = cdiv(cneg(cmul(x21, rp)), dp); EI[0][3] = cdiv(x11, dp);
  EI[1][0] = cdiv(x42, ds); EI[1][1] = cdiv(cneg(cmul(x32, rsv)), ds);
  EI[1][2] = cdiv(cneg(x22), ds); EI[1][3] = cdiv(cmul(x12, rsv), ds);
  EI[2][0] = cdiv(cneg(cmul(x41, rp)), cneg(dp)); EI[2][1] = cdiv(cneg(x31), cneg(dp));
  EI[2][2] = cdiv(cmul(x21, rp), cneg(dp)); EI[2][3] = cdiv(x11, cneg(dp));
  EI[3][0] = cdiv(x42, cneg(ds)); EI[3][1] = cdiv(cmul(x32, rsv), cneg(ds));
  EI[3][2] = cdiv(cneg(x22), cneg(ds)); EI[3][3] = cdiv(cneg(cmul(x12, rsv)), cneg(ds));
  E[0][0] = x11; E[1][0] = cmul(x21, rp); E[2][0] = x31; E[3][0] = cmul(x41, rp);
  E[0][1] = cmul(x12, rsv); E[1][1] = x22; E[2][1] = cmul(x32, rsv); E[3][1] = x42;
  E[0][2] = x11; E[1][2] = cneg(cmul(x21, rp)); E[2][2] = x31; E[3][2] = cneg(cmul(x41, rp));
  E[0][3] = cneg(cmul(x12, rsv)); E[1][3] = x22; E[2][3] = cneg(cmul(x32, rsv)); E[3][3] = x42;
}

/* inv/tregn96.f:1385-1474 */
static cx ffunc(cx nu, double dm) {
  if (cabs_(nu) < 1.0e-8) return C(dm, 0);
  const cx arg = cscale(nu, dm);
  const cx ex = arg.re < 40.0 ? cexp_(cscale(arg, -2.0)) : C(0, 0);
  return cdiv(csub(C(1, 0), ex), cscale(nu, 2.0));
}
static cx gfunc(cx nu, double dm) {
  const cx arg = cscale(nu, dm);
  return arg.re < 75.0 ? cscale(cexp_(cneg(arg)), dm) : C(0, 0);
}
static cx h1func(cx na, cx nb, double dm) {
  if (cabs_(cadd(nb, na)) < 1.0e-8) return C(dm, 0);
  const cx arg = cscale(cadd(na, nb), dm);
  const cx ex = arg.re < 40.0 ? cexp_(cneg(arg)) : C(0, 0);
  return cdiv(csub(C(1, 0), ex), cadd(nb, na));
}
static cx h2func(cx na, cx nb, double dm) {
  if (cabs_(csub(nb, na)) < 1.0e-8) return C(dm, 0);
  cx arg = cscale(na, dm);
  const cx exp_ = arg.re < 40.0 ? cexp_(cneg(arg)) : C(0, 0);
  arg = cscale(nb, dm);
  const cx exq = arg.re < 40.0 ? cexp_(cneg(arg)) : C(0, 0);
  return cdiv(csub(exq, exp_), csub(na, nb));
}

static double normalise(double *v, int n) { /* rnormc, inv/tregn96.f:1513 */
  double t1 = 0.0;
  for (int i = 0; i < n; i++)
    if (fabs(v[i]) > t1) t1 = fabs(v[i]);
  if (t1 < 1.0e-40) t1 = 1.0;
  for (int i = 0; i < n; i++) v[i] /= t1;
  return log(t1);
}
static double cnormalise(cx *v, int n) { /* cnormc, :1475 */
  double t1 = 0.0;
  for (int i = 0; i < n; i++)
    if (cabs_(v[i]) > t1) t1 = cabs_(v[i]);
  if (t1 < 1.0e-40) t1 = 1.0;
  for (int i = 0; i < n; i++) v[i] = cscale(v[i], 1.0 / t1), v[i] = v[i];
  return log(t1);
}

/* tregn96 for one layered TI model.  d,TA,TC,TF,TL,TN,rho: fp32 [nl] as depthkernelTI passes them (the last layer is the
 * half-space).  t, cp fp32 [nt].  Outputs fp32 [nt][nl] (= dcdah_out(k,i) etc. after sprayl/chksiz).  Returns 0, or 3 if a
 * fluid layer is present (not restated). */
int orc_tregn96(int nl, const float *d_in, const float *TA_in, const float *TC_in, const float *TF_in, const float *TL_in,
                const float *TN_in, const float *rho_in, int nt, const float *t_in, const float *cp_in, float *dcdah_out,
                float *dcdbv_out, float *dcdn_out) {
  float d[ORC_NL], ta[ORC_NL], tc[ORC_NL], tl[ORC_NL], tn[ORC_NL], tf[ORC_NL], trho[ORC_NL], vtp[ORC_NL];
  double zd[ORC_NL], zta[ORC_NL], ztc[ORC_NL], ztf[ORC_NL], ztl[ORC_NL], ztn[ORC_NL], zrho[ORC_NL];
  const int mmax = nl;
  for (int i = 0; i < mmax; i++) {
    d[i] = d_in[i]; ta[i] = TA_in[i]; tc[i] = TC_in[i]; tf[i] = TF_in[i]; tl[i] = TL_in[i]; tn[i] = TN_in[i];
    trho[i] = rho_in[i];
    if (!(tn[i] > 0.0001f * ta[i]) || tl[i] == 0.0f) return 3;
  }
  const double zqa = (double)(1.0f / 150.0f), zqb = (double)(1.0f / 50.0f);   /* :267-268, Q > 1 is inverted in fp32 */
  /* ---- sphere_tdisp96, inv/tregn96.f:774 (TF is not transformed) ---- */
  {
    const double ar = (double)6371.0f;
    double r0 = ar + 0.0;
    d[mmax - 1] = 1.0f;
    for (int i = 0; i < mmax; i++) {
      const double r1 = r0 - (double)d[i];
      const double z0 = ar * log(ar / r0), z1 = ar * log(ar / r1);
      d[i] = (float)(z1 - z0);
      const double tmp = (ar + ar) / (r0 + r1);
      trho[i] = (float)((double)trho[i] * pow(tmp, -2.275));
      const double pw = pow(tmp, -0.2750);
      ta[i] = (float)((double)ta[i] * pw);
      tc[i] = (float)((double)tc[i] * pw);
      tl[i] = (float)((double)tl[i] * pw);
      tn[i] = (float)((double)tn[i] * pw);
      r0 = r1;
    }
    d[mmax - 1] = 0.0f;
  }
  for (int i = 0; i < mmax; i++) {
    zd[i] = d[i]; zta[i] = ta[i]; ztc[i] = tc[i]; ztl[i] = tl[i]; ztn[i] = tn[i]; ztf[i] = tf[i]; zrho[i] = trho[i];
  }
  /* ---- bldsph, :1301 (radius 6370 here) ---- */
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
  /* insert()/srclyr() with source and receiver at depth 0 leave the model alone (:847-997) */
  const float twopi = 2.f * 3.141592654f;
  static double cdre[ORC_NL][5], exe[ORC_NL], exa[ORC_NL], vv[ORC_NL][4];
  double ur[ORC_NL], uz[ORC_NL], tz[ORC_NL], tr[ORC_NL];
  double fah[ORC_NL], fav[ORC_NL], fbv[ORC_NL], fn[ORC_NL];
  for (int ip = 0; ip < nt; ip++) {
    const double t = (double)t_in[ip];
    const double omega = (double)twopi / t;
    double c = (double)cp_in[ip];
    double wvno = omega / c;
    const double om2 = omega * omega, wvno2 = wvno * wvno;
    eig g;
    trig tg;
    /* ---- up, :1834: compound vector from the half-space to the surface ---- */
    {
      cx EE[4][4], EI[4][4], cd[5], nx_[5];
      layer_eig(zta[mmax - 1], ztc[mmax - 1], ztf[mmax - 1], ztl[mmax - 1], zrho[mmax - 1], omega, wvno, &g);
      layer_E(&g, EE, EI);
      static const int HP[5][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 3}, {2, 3}};   /* CG(1),(2),(3),(5),(6), :3079-3088 */
      for (int k = 0; k < 5; k++) {
        const cx m = csub(cmul(EI[0][HP[k][0]], EI[1][HP[k][1]]), cmul(EI[0][HP[k][1]], EI[1][HP[k][0]]));
        cd[k] = C(m.re, 0.0);
        cdre[mmax - 1][k] = m.re;
      }
      exe[mmax - 1] = 0.0;
      double exsum = 0.0;
      for (int m = mmax - 2; m >= 0; m--) {
        cx CA[5][5];
        layer_eig(zta[m], ztc[m], ztf[m], ztl[m], zrho[m], omega, wvno, &g);
        layer_trig(&g, zd[m], &tg);
        layer_compound(&g, &tg, CA);
        for (int i = 0; i < 5; i++) {
          cx s = C(0, 0);
          for (int j = 0; j < 5; j++) s = cadd(s, cmul(cd[j], CA[j][i]));
          nx_[i] = s;
        }
        const double exn = cnormalise(nx_, 5);
        exsum += tg.pex + tg.svex + exn;
        exe[m] = exsum;
        for (int i = 0; i < 5; i++) { cd[i] = nx_[i]; cdre[m][i] = nx_[i].re; }
      }
      ur[0] = cdiv(cd[2], cd[1]).re;       /* svfunc :1606-1613 */
    }
    /* ---- down, :3561: Haskell vector from the surface down ---- */
    {
      vv[0][0] = 1.0; vv[0][1] = vv[0][2] = vv[0][3] = 0.0;
      exa[0] = 0.0;
      double exsum = 0.0;
      for (int m = 0; m < mmax - 1; m++) {
        layer_eig(zta[m], ztc[m], ztf[m], ztl[m], zrho[m], omega, wvno, &g);
        layer_trig(&g, zd[m], &tg);
        double cpex, fp = 1.0, fs = 1.0;
        if (tg.pex > tg.svex) { fs = (tg.pex - tg.svex) > 40.0 ? 0.0 : exp(-(tg.pex - tg.svex)); cpex = tg.pex; }
        else { fp = (tg.svex - tg.pex) > 40.0 ? 0.0 : exp(-(tg.svex - tg.pex)); cpex = tg.svex; }
        const cx cosp = cdiv(cscale(tg.cosp, fp), g.np), sinpr = cdiv(cscale(tg.sinpr, fp), g.np), rsinp = cdiv(cscale(tg.rsinp, fp), g.np);
        const cx cosq = cdiv(cscale(tg.cosq, fs), g.nsv), sinqr = cdiv(cscale(tg.sinqr, fs), g.nsv), rsinq = cdiv(cscale(tg.rsinq, fs), g.nsv);
        const cx *x1 = &g.x[0][0];
#define X(i, md) (x1[2 * ((i) - 1) + (md) - 1])
        double AA[4][4];
        AA[0][0] = cadd(cmul(cmul(X(1, 1), X(4, 1)), cosp), cmul(cmul(X(1, 2), X(4, 2)), cosq)).re;
        AA[0][1] = cneg(cadd(cmul(cmul(X(1, 1), X(3, 1)), sinpr), cmul(cmul(X(1, 2), X(3, 2)), rsinq))).re;
        AA[0][2] = cneg(cadd(cmul(cmul(X(1, 1), X(2, 1)), cosp), cmul(cmul(X(1, 2), X(2, 2)), cosq))).re;
        AA[0][3] = cadd(cmul(cmul(X(1, 1), X(1, 1)), sinpr), cmul(cmul(X(1, 2), X(1, 2)), rsinq)).re;
        AA[1][0] = cadd(cmul(cmul(X(2, 1), X(4, 1)), rsinp), cmul(cmul(X(2, 2), X(4, 2)), sinqr)).re;
        AA[1][1] = cneg(cadd(cmul(cmul(X(2, 1), X(3, 1)), cosp), cmul(cmul(X(2, 2), X(3, 2)), cosq))).re;
        AA[1][2] = cneg(cadd(cmul(cmul(X(2, 1), X(2, 1)), rsinp), cmul(cmul(X(2, 2), X(2, 2)), sinqr))).re;
        AA[2][0] = cadd(cmul(cmul(X(3, 1), X(4, 1)), cosp), cmul(cmul(X(3, 2), X(4, 2)), cosq)).re;
        AA[2][1] = cneg(cadd(cmul(cmul(X(3, 1), X(3, 1)), sinpr), cmul(cmul(X(3, 2), X(3, 2)), rsinq))).re;
        AA[3][0] = cadd(cmul(cmul(X(4, 1), X(4, 1)), rsinp), cmul(cmul(X(4, 2), X(4, 2)), sinqr)).re;
#undef X
        AA[1][3] = -AA[0][2]; AA[2][2] = AA[1][1]; AA[2][3] = -AA[0][1];
        AA[3][1] = -AA[2][0]; AA[3][2] = -AA[1][0]; AA[3][3] = AA[0][0];
        double a0[4];
        for (int i = 0; i < 4; i++) {
          double s = 0.0;
          for (int j = 0; j < 4; j++) s += AA[i][j] * vv[m][j];
          a0[i] = s;
        }
        const double ex2 = normalise(a0, 4);
        exsum += cpex + ex2;
        exa[m + 1] = exsum;
        for (int i = 0; i < 4; i++) vv[m + 1][i] = a0[i];
      }
    }
    /* ---- svfunc, :1559: eigenfunctions at the layer tops ---- */
    {
      const double f1213 = -cdre[0][1];
      uz[0] = 1.0; tz[0] = 0.0; tr[0] = 0.0;
      for (int i = 1; i < mmax; i++) {
        const double cd1 = cdre[i][0], cd2 = cdre[i][1], cd3 = cdre[i][2], cd4 = -cdre[i][2], cd5 = cdre[i][3], cd6 = cdre[i][4];
        const double tz1 = -vv[i][3], tz2 = -vv[i][2], tz3 = vv[i][1], tz4 = vv[i][0];
        const double uu1 = tz2 * cd6 - tz3 * cd5 + tz4 * cd4, uu2 = -tz1 * cd6 + tz3 * cd3 - tz4 * cd2;
        const double uu3 = tz1 * cd5 - tz2 * cd3 + tz4 * cd1, uu4 = -tz1 * cd4 + tz2 * cd2 - tz3 * cd1;
        const double ext = exa[i] + exe[i] - exe[0];
        if (ext > -80.0 && ext < 80.0) {
          const double fact = exp(ext);
          ur[i] = uu1 * fact / f1213; uz[i] = uu2 * fact / f1213; tz[i] = uu3 * fact / f1213; tr[i] = uu4 * fact / f1213;
        } else {
          ur[i] = uz[i] = tz[i] = tr[i] = 0.0;
        }
      }
    }
    /* ---- energy, :3777: layer integrals and unnormalised partials ---- */
    double sumi0 = 0.0, sumi1 = 0.0, sumi2 = 0.0, sumi3 = 0.0;
    for (int m = 0; m < mmax; m++) {
      const double rho = zrho[m], TA = zta[m], TC = ztc[m], TF = ztf[m], TL = ztl[m];
      const double ah = sqrt(TA / rho), av = sqrt(TC / rho), bv = sqrt(TL / rho);
      const double eta = TF / (TA - 2. * TL), a12 = -wvno, a14 = 1.0 / TL, a21 = wvno * TF / TC, a23 = 1.0 / TC;
      cx E[4][4], EI[4][4];
      layer_eig(TA, TC, TF, TL, rho, omega, wvno, &g);
      layer_E(&g, E, EI);
      const cx ra = g.rp, rb = g.rsv;
      cx k[4];   /* kmpu, kmsu (from the layer bottom), km1pd, km1sd (from its top) */
      const int last = m == mmax - 1;
      for (int r = 0; r < 4; r++) {
        const int at = (r < 2 && !last) ? m + 1 : m;
        k[r] = cadd(cadd(cscale(EI[r][0], ur[at]), cscale(EI[r][1], uz[at])), cadd(cscale(EI[r][2], tz[at]), cscale(EI[r][3], tr[at])));
      }
      cx FA = C(0, 0), GA = FA, FB = FA, GB = FA, H1 = FA, H2 = FA;
      if (!last) {
        FA = ffunc(ra, zd[m]); GA = gfunc(ra, zd[m]); FB = ffunc(rb, zd[m]); GB = gfunc(rb, zd[m]);
        H1 = h1func(ra, rb, zd[m]); H2 = h2func(ra, rb, zd[m]);
      }
      static const int IJ[6][2] = {{0, 0}, {0, 2}, {1, 1}, {1, 3}, {2, 2}, {3, 3}};
      double I[6];
      for (int q = 0; q < 6; q++) {      /* intijr, :4073 */
        const int i = IJ[q][0], j = IJ[q][1];
        cx s;
        if (!last) {
          const cx kmpu = k[0], kmsu = k[1], km1pd = k[2], km1sd = k[3];
#define EE2(a, b) cadd(cmul(E[i][a], E[j][b]), cmul(E[i][b], E[j][a]))
          s = cmul(cmul(cmul(E[i][0], E[j][0]), cmul(kmpu, kmpu)), FA);
          s = cadd(s, cmul(cmul(cmul(E[i][2], E[j][2]), cmul(km1pd, km1pd)), FA));
          s = cadd(s, cmul(cmul(cmul(E[i][1], E[j][1]), cmul(kmsu, kmsu)), FB));
          s = cadd(s, cmul(cmul(cmul(E[i][3], E[j][3]), cmul(km1sd, km1sd)), FB));
          s = cadd(s, cmul(H1, cadd(cmul(EE2(0, 1), cmul(kmpu, kmsu)), cmul(EE2(2, 3), cmul(km1pd, km1sd)))));
          s = cadd(s, cmul(H2, cadd(cmul(EE2(0, 3), cmul(kmpu, km1sd)), cmul(EE2(1, 2), cmul(km1pd, kmsu)))));
          s = cadd(s, cmul(cmul(GA, EE2(0, 2)), cmul(kmpu, km1pd)));
          s = cadd(s, cmul(cmul(GB, EE2(1, 3)), cmul(kmsu, km1sd)));
        } else {
          const cx km1pd = k[2], km1sd = k[3];
          s = cdiv(cmul(cmul(E[i][2], E[j][2]), cmul(km1pd, km1pd)), cscale(ra, 2.0));
          s = cadd(s, cdiv(cmul(EE2(2, 3), cmul(km1pd, km1sd)), cadd(ra, rb)));
          s = cadd(s, cdiv(cmul(cmul(E[i][3], E[j][3]), cmul(km1sd, km1sd)), cscale(rb, 2.0)));
#undef EE2
        }
        I[q] = s.re;
      }
      const double I11 = I[0], I13 = I[1], I22 = I[2], I24 = I[3], I33 = I[4], I44 = I[5];
      const double URUR = I11, UZUZ = I22;
      const double DURDUR = a12 * a12 * I22 + 2. * a12 * a14 * I24 + a14 * a14 * I44;
      const double DUZDUZ = a21 * a21 * I11 + 2. * a21 * a23 * I13 + a23 * a23 * I33;
      const double URDUZ = a21 * I11 + a23 * I13, UZDUR = a12 * I22 + a14 * I24;
      sumi0 += rho * (URUR + UZUZ);
      sumi1 += TL * UZUZ + TA * URUR;
      sumi2 += TL * UZDUR - TF * URDUZ;
      sumi3 += TL * DURDUR + TC * DUZDUZ;
      fah[m] = rho * ah * (URUR - 2. * eta * URDUZ / wvno);
      fav[m] = rho * av * DUZDUZ / wvno2;
      fbv[m] = rho * bv * (UZUZ + 2. * UZDUR / wvno + DURDUR / wvno2 + 4. * eta * URDUZ / wvno);
      fn[m] = -TF * URDUZ / (wvno * eta);
    }
    const double ugr = (wvno * sumi1 + sumi2) / (omega * sumi0);
    (void)sumi3; (void)om2;
    for (int m = 0; m < mmax; m++) {
      fah[m] /= ugr * sumi0; fav[m] /= ugr * sumi0; fbv[m] /= ugr * sumi0; fn[m] /= ugr * sumi0;
    }
    /* ---- gammap, :3711: causal-Q phase-velocity shift (only c is used afterwards) ---- */
    {
      const double pi = 3.141592653589493;
      double dc = 0.0;
      for (int m = 0; m < mmax; m++) {
        const double ah = sqrt(zta[m] / zrho[m]), av = sqrt(ztc[m] / zrho[m]), bv = sqrt(ztl[m] / zrho[m]);
        double x = 0.0 * zqb + fbv[m] * bv * zqb;        /* dcdbh = 0 */
        dc += log(omega / (2.0 * pi * 1.0)) * x / pi;
        x = fav[m] * av * zqa + fah[m] * ah * zqa;
        dc += log(omega / (2.0 * pi * 1.0)) * x / pi;
      }
      c = omega / wvno + dc;
      wvno = omega / c;
    }
    /* ---- sprayl, :1235 + chksiz: spherical correction of the velocity partials (dcdn is left alone) ---- */
    {
      const double ar = 6370.0;
      const double q = c / (2. * ar * omega);
      const double tm = sqrt(1. + q * q);
      const double tm3 = tm * tm * tm;
      for (int m = 0; m < mmax; m++) {
        const double a = fah[m] * (double)vtp[m] / tm3, b = fbv[m] * (double)vtp[m] / tm3, n = fn[m];
        dcdah_out[(size_t)ip * nl + m] = fabs(a) < 1.0e-36 ? 0.0f : (float)a;
        dcdbv_out[(size_t)ip * nl + m] = fabs(b) < 1.0e-36 ? 0.0f : (float)b;
        dcdn_out[(size_t)ip * nl + m] = fabs(n) < 1.0e-36 ? 0.0f : (float)n;
      }
    }
  }
  return 0;
}

/* depthkernelTI, inv/depthkernelTI.f90:2: per column Brocher model -> layers -> surfdisp96 -> tregn96 -> Lsen_Gsc.
 * vel[nz][ny][nx]; pv[kmax][nx*ny] (may be NULL); lsen[nz-1][kmax][nx*ny] fp32 */
int orc_depthkernel_ti(int nx, int ny, int nz, const float *vel, int kmax, const double *t, const float *depz, float minthk,
                       double *pv, float *lsen) {
  const size_t ncol = (size_t)nx * ny;
  static float dah[ORC_NP * ORC_NL], dbv[ORC_NP * ORC_NL], dn[ORC_NP * ORC_NL];
  memset(lsen, 0, sizeof(float) * (size_t)(nz - 1) * kmax * ncol);
  for (int jj = 0; jj < ny; jj++)
    for (int ii = 0; ii < nx; ii++) {
      const size_t col = (size_t)jj * nx + ii;
      float vsz[ORC_NL], vpz[ORC_NL], rhoz[ORC_NL], rthk[ORC_NL], rvp[ORC_NL], rvs[ORC_NL], rrho[ORC_NL];
      float TA[ORC_NL], TL[ORC_NL], TF[ORC_NL], tin[ORC_NP], cpin[ORC_NP];
      double cg[ORC_NP];
      for (int k = 0; k < nz; k++) {
        vsz[k] = vel[((size_t)k * ny + jj) * nx + ii];
        orc_brocher(vsz[k], &vpz[k], &rhoz[k]);
      }
      const int rmax = orc_refine_layers(minthk, nz, depz, vpz, vsz, rhoz, rthk, rvp, rvs, rrho);
      orc_surfdisp96(rthk, rvp, rvs, rrho, rmax, kmax, t, cg);
      for (int k = 0; k < kmax; k++) {
        if (pv) pv[(size_t)k * ncol + col] = cg[k];
        cpin[k] = (float)cg[k];
        tin[k] = (float)t[k];
      }
      for (int i = 0; i < rmax; i++) {
        TA[i] = rrho[i] * (rvp[i] * rvp[i]);
        TL[i] = rrho[i] * (rvs[i] * rvs[i]);
        TF[i] = 1.0f * (TA[i] - 2 * TL[i]);
      }
      const int rc = orc_tregn96(rmax, rthk, TA, TA, TF, TL, TL, rrho, kmax, tin, cpin, dah, dbv, dn);
      if (rc) return rc;
      for (int ip = 0; ip < kmax; ip++) {
        int k = 0;
        for (int j = 0; j < nz - 1; j++) {
          const float thk = depz[j + 1] - depz[j];
          const int nsub = (int)((thk + 1.0e-4f) / (thk / minthk)) + 1;
          float acc = 0.0f;
          for (int s = 0; s < nsub; s++, k++) {
            const float den = (TA[k] - 2.0f * TL[k]) * (TA[k] - 2.0f * TL[k]);
            const float dA = 0.5f / (rrho[k] * rvp[k]) * dah[ip * rmax + k] - TF[k] / den * dn[ip * rmax + k];
            const float dL = 0.5f / (rrho[k] * rvs[k]) * dbv[ip * rmax + k] + 2.0f * TF[k] / den * dn[ip * rmax + k];
            acc = acc + dA * TA[k] + dL * TL[k];
          }
          lsen[((size_t)j * kmax + ip) * ncol + col] = acc;
        }
      }
    }
  return 0;
}
