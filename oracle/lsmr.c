/* lsmr.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * CPU restatement of the reference's sparse solve: COO aprod, the fp32 BLAS-1 it uses and LSMR
 * (Fong & Saunders) with local reorthogonalisation, all in fp32 like the reference
 * (dp = selected_real_kind(4), inv/lsmrDataModule.f90:21).
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* inv/aprod.f90:40-55 */
void orc_aprod(int mode, int m, int n, float *x, float *y, int64_t nar, const int *irow,
               const int *icol, const float *rw) {
  (void)m;
  (void)n;
  if (mode == 1)
    for (int64_t k = 0; k < nar; k++) y[irow[k] - 1] = y[irow[k] - 1] + rw[k] * x[icol[k] - 1];
  else
    for (int64_t k = 0; k < nar; k++) x[icol[k] - 1] = x[icol[k] - 1] + rw[k] * y[irow[k] - 1];
}

/* inv/lsmrblas.f90:247-277 */
float orc_nrm2(int n, const float *x) {
  if (n < 1) return 0.0f;
  if (n == 1) return fabsf(x[0]);
  float scale = 0.0f, ssq = 1.0f;
  for (int i = 0; i < n; i++)
    if (x[i] != 0.0f) {
      float a = fabsf(x[i]);
      if (scale < a) {
        ssq = 1.0f + ssq * ((scale / a) * (scale / a));
        scale = a;
      } else
        ssq = ssq + (a / scale) * (a / scale);
    }
  return scale * sqrtf(ssq);
}

static void scal(int n, float sa, float *x) { /* inv/lsmrblas.f90:317 */
  for (int i = 0; i < n; i++) x[i] = sa * x[i];
}

static float d2norm(float a, float b) { /* inv/lsmrModule.f90:708-721 */
  float scale = fabsf(a) + fabsf(b);
  if (scale == 0.0f) return 0.0f;
  return scale * sqrtf((a / scale) * (a / scale) + (b / scale) * (b / scale));
}

/* inv/lsmrModule.f90:36-750 */
int orc_lsmr(int m, int n, int64_t nar, const int *irow, const int *icol, const float *rw,
             const float *b, float damp, float atol, float btol, float conlim, int itnlim,
             int localSize, float *x, int *istop_o, int *itn_o, float *normA_o, float *condA_o,
             float *normr_o, float *normAr_o, float *normx_o) {
  int localVecs = localSize < m ? localSize : m;
  if (n < localVecs) localVecs = n;
  float *h = calloc(n, 4), *hbar = calloc(n, 4), *u = calloc(m, 4), *v = calloc(n, 4);
  float *localV = localVecs > 0 ? malloc((size_t)n * localVecs * 4) : NULL;
  int istop = 0, itn = 0;
  float normA = 0, condA = 0, normr = 0, normAr = 0, normx = 0;
  memcpy(u, b, (size_t)m * 4);
  memset(x, 0, (size_t)n * 4);
  float alpha = 0.0f, beta = orc_nrm2(m, u);
  if (beta > 0.0f) {
    scal(m, 1.0f / beta, u);
    orc_aprod(2, m, n, v, u, nar, irow, icol, rw);
    alpha = orc_nrm2(n, v);
  }
  if (alpha > 0.0f) scal(n, 1.0f / alpha, v);
  normAr = alpha * beta;
  if (normAr == 0.0f) goto done;
  int localOrtho = 0, localPointer = 0, localVQueueFull = 0;
  if (localVecs > 0) {
    localPointer = 1;
    localOrtho = 1;
    memcpy(localV, v, (size_t)n * 4);
  }
  float zetabar = alpha * beta, alphabar = alpha, rho = 1, rhobar = 1, cbar = 1, sbar = 0;
  memcpy(h, v, (size_t)n * 4);
  float betadd = beta, betad = 0, rhodold = 1, tautildeold = 0, thetatilde = 0, zeta = 0, d = 0;
  float normA2 = alpha * alpha, maxrbar = 0, minrbar = 1e+30f, normb = beta, ctol = 0;
  if (conlim > 0.0f) ctol = 1.0f / conlim;
  normr = beta;
  for (;;) {
    itn++;
    scal(m, -alpha, u);
    orc_aprod(1, m, n, v, u, nar, irow, icol, rw);
    beta = orc_nrm2(m, u);
    if (beta > 0.0f) {
      scal(m, 1.0f / beta, u);
      if (localOrtho) { /* localVEnqueue :723-731 */
        if (localPointer < localVecs)
          localPointer++;
        else {
          localPointer = 1;
          localVQueueFull = 1;
        }
        memcpy(localV + (size_t)(localPointer - 1) * n, v, (size_t)n * 4);
      }
      scal(n, -beta, v);
      orc_aprod(2, m, n, v, u, nar, irow, icol, rw);
      if (localOrtho) { /* localVOrtho :733-748 */
        int lim = localVQueueFull ? localVecs : localPointer;
        for (int q = 0; q < lim; q++) {
          const float *lv = localV + (size_t)q * n;
          float dd = 0.0f;
          for (int i = 0; i < n; i++) dd = dd + v[i] * lv[i];
          for (int i = 0; i < n; i++) v[i] = v[i] - dd * lv[i];
        }
      }
      alpha = orc_nrm2(n, v);
      if (alpha > 0.0f) scal(n, 1.0f / alpha, v);
    }
    float alphahat = d2norm(alphabar, damp);
    float chat = alphabar / alphahat, shat = damp / alphahat;
    float rhoold = rho;
    rho = d2norm(alphahat, beta);
    float c = alphahat / rho, s = beta / rho;
    float thetanew = s * alpha;
    alphabar = c * alpha;
    float rhobarold = rhobar, zetaold = zeta;
    float thetabar = sbar * rho, rhotemp = cbar * rho;
    rhobar = d2norm(cbar * rho, thetanew);
    cbar = cbar * rho / rhobar;
    sbar = thetanew / rhobar;
    zeta = cbar * zetabar;
    zetabar = -sbar * zetabar;
    float f1 = thetabar * rho / (rhoold * rhobarold), f2 = zeta / (rho * rhobar), f3 = thetanew / rho;
    for (int i = 0; i < n; i++) hbar[i] = h[i] - f1 * hbar[i];
    for (int i = 0; i < n; i++) x[i] = x[i] + f2 * hbar[i];
    for (int i = 0; i < n; i++) h[i] = v[i] - f3 * h[i];
    float betaacute = chat * betadd, betacheck = -shat * betadd;
    float betahat = c * betaacute;
    betadd = -s * betaacute;
    float thetatildeold = thetatilde;
    float rhotildeold = d2norm(rhodold, thetabar);
    float ctildeold = rhodold / rhotildeold, stildeold = thetabar / rhotildeold;
    thetatilde = stildeold * rhobar;
    rhodold = ctildeold * rhobar;
    betad = -stildeold * betad + ctildeold * betahat;
    tautildeold = (zetaold - thetatildeold * tautildeold) / rhotildeold;
    float taud = (zeta - thetatilde * tautildeold) / rhodold;
    d = d + betacheck * betacheck;
    normr = sqrtf(d + (betad - taud) * (betad - taud) + betadd * betadd);
    normA2 = normA2 + beta * beta;
    normA = sqrtf(normA2);
    normA2 = normA2 + alpha * alpha;
    maxrbar = fmaxf(maxrbar, rhobarold);
    if (itn > 1) minrbar = fminf(minrbar, rhobarold);
    condA = fmaxf(maxrbar, rhotemp) / fminf(minrbar, rhotemp);
    normAr = fabsf(zetabar);
    normx = orc_nrm2(n, x);
    float test1 = normr / normb, test2 = normAr / (normA * normr), test3 = 1.0f / condA;
    float t1 = test1 / (1.0f + normA * normx / normb);
    float rtol = btol + atol * normA * normx / normb;
    if (itn >= itnlim) istop = 7;
    if (1.0f + test3 <= 1.0f) istop = 6;
    if (1.0f + test2 <= 1.0f) istop = 5;
    if (1.0f + t1 <= 1.0f) istop = 4;
    if (test3 <= ctol) istop = 3;
    if (test2 <= atol) istop = 2;
    if (test1 <= rtol) istop = 1;
    if (istop != 0) break;
  }
done:
  if (damp > 0.0f && istop == 2) istop = 3;
  *istop_o = istop; *itn_o = itn; *normA_o = normA; *condA_o = condA;
  *normr_o = normr; *normAr_o = normAr; *normx_o = normx;
  free(h); free(hbar); free(u); free(v); free(localV);
  return 0;
}

/* inv/TikhRegul.f90:2-61 (iso_inv branch): 7-point Laplacian rows */
int orc_tikhonov_iso(int nx, int ny, int nz, int dall, float weight, int64_t *nar_io, float *rw,
                     int *irow, int *icol) {
  int nvz = ny - 2, nvx = nx - 2, count3 = 0;
  int64_t nar = *nar_io;
  for (int k = 1; k <= nz - 1; k++)
    for (int j = 1; j <= nvz; j++)
      for (int i = 1; i <= nvx; i++) {
        int c0 = (k - 1) * nvz * nvx + (j - 1) * nvx + i;
        count3++;
        if (i == 1 || i == nvx || j == 1 || j == nvz || k == 1 || k == nz - 1) {
          icol[nar] = c0; rw[nar] = 2.0f * weight; irow[nar] = dall + count3;
          nar++;
        } else {
          int cols[7] = {c0, c0 - 1, c0 + 1, c0 - nvx, c0 + nvx, c0 - nvz * nvx, c0 + nvz * nvx};
          for (int q = 0; q < 7; q++) {
            icol[nar + q] = cols[q];
            rw[nar + q] = (q == 0 ? 6.0f : -1.0f) * weight;
            irow[nar + q] = dall + count3;
          }
          nar += 7;
        }
      }
  *nar_io = nar;
  return count3;
}
