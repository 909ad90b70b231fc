/* fmm.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * CPU restatement of the reference's 2-D spherical-shell fast-marching eikonal solver and the
 * source-grid-refinement driver around it.  fp32 throughout, no FMA (build with
 * -ffp-contract=off), operation order kept exactly as the Fortran expressions evaluate.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define GDX 5     /* inv/CalSurfG.f90:1005-1012 */
#define GDZ 5
#define SGDL 8
#define SGS 8
#define EARTH 6371.0f
static const float PI_F = 3.1415926535898f; /* inv/CalSurfG.f90:166 */

/* ---- geometry: inv/CalSurfG.f90:1017-1038 -------------------------------------------------- */
void orc_geometry(int nx, int ny, float goxd, float gozd, float dvxd, float dvzd, orc_geom *g) {
  g->nvx = nx - 2;
  g->nvz = ny - 2;
  g->dvx = dvxd * PI_F / 180.0f;
  g->dvz = dvzd * PI_F / 180.0f;
  g->gox = (90.0f - goxd) * PI_F / 180.0f;
  g->goz = gozd * PI_F / 180.0f;
  g->nnx = (g->nvx - 1) * GDX + 1;
  g->nnz = (g->nvz - 1) * GDZ + 1;
  g->dnx = g->dvx / GDX;
  g->dnz = g->dvz / GDZ;
}

/* cubic B-spline basis at u: inv/CalSurfG.f90:1472-1475 (same four expressions at :1546-1553) */
static void bspl4(float u, float w[4]) {
  float om = 1.0f - u;
  w[0] = om * om * om / 6.0f;
  w[1] = (4.0f - 6.0f * (u * u) + 3.0f * (u * u * u)) / 6.0f;
  w[2] = (1.0f + 3.0f * u + 3.0f * (u * u) - 3.0f * (u * u * u)) / 6.0f;
  w[3] = u * u * u / 6.0f;
}

/* ---- gridder: inv/CalSurfG.f90:1423-1516 --------------------------------------------------- */
void orc_gridder(const orc_geom *g, const double *pv, float *veln) {
  int nvx = g->nvx, nvz = g->nvz, nnz = g->nnz;
  int ldv = nvz + 2; /* velv(0:nvz+1,0:nvx+1): velv(i,j) = pv(i*(nvx+2)+j+1)  (:1455) */
  float *velv = (float *)malloc(sizeof(float) * (size_t)ldv * (nvx + 2));
  for (int i = 0; i <= nvz + 1; i++)
    for (int j = 0; j <= nvx + 1; j++) velv[j * ldv + i] = (float)pv[i * (nvx + 2) + j];
  float ui[GDX + 1][4], vi[GDZ + 1][4];
  for (int i = 1; i <= GDX + 1; i++) {
    float u = (float)GDX;
    u = (float)(i - 1) / u;
    bspl4(u, ui[i - 1]);
  }
  for (int i = 1; i <= GDZ + 1; i++) {
    float u = (float)GDZ;
    u = (float)(i - 1) / u;
    bspl4(u, vi[i - 1]);
  }
  for (int i = 1; i <= nvz - 1; i++) {
    int conz = (i == nvz - 1) ? GDZ + 1 : GDZ;
    for (int j = 1; j <= nvx - 1; j++) {
      int conx = (j == nvx - 1) ? GDX + 1 : GDX;
      for (int l = 1; l <= conz; l++) {
        int stz = GDZ * (i - 1) + l;
        for (int m = 1; m <= conx; m++) {
          int stx = GDX * (j - 1) + m;
          float sumi = 0.0f;
          for (int i1 = 1; i1 <= 4; i1++) {
            float sumj = 0.0f;
            for (int j1 = 1; j1 <= 4; j1++)
              sumj = sumj + ui[m - 1][j1 - 1] * velv[(j - 2 + j1) * ldv + (i - 2 + i1)];
            sumi = sumi + vi[l - 1][i1 - 1] * sumj;
          }
          veln[(size_t)(stx - 1) * nnz + (stz - 1)] = sumi;
        }
      }
    }
  }
  free(velv);
}

/* ---- bsplrefine: inv/CalSurfG.f90:1525-1591, evaluated node-by-node over the refined box ----
 * Every refined node belongs to exactly one (cell, sub-node) pair of the reference's loop nest
 * (the last cell also owns its far edge, :1563/:1566), so visiting nodes directly gives the same
 * value the reference stores. */
static void bsplrefine(const orc_geom *g, const double *pv, const orc_refbox *b, float *velnr) {
  int nvx = g->nvx, nvz = g->nvz;
  int nrxr = GDX * SGDL, nrzr = GDZ * SGDL;
  int origx = (b->vnl - 1) * SGDL + 1, origz = (b->vnt - 1) * SGDL + 1;
  for (int idm2 = 1; idm2 <= b->nnxr; idm2++) {
    int st2 = idm2 + origx - 1;
    int j = (st2 - 1) / nrxr + 1;
    if (j > nvx - 1) j = nvx - 1;
    int l = st2 - nrxr * (j - 1);
    float u = (float)nrxr;
    u = (float)(l - 1) / u;
    float ui[4];
    bspl4(u, ui);
    for (int idm1 = 1; idm1 <= b->nnzr; idm1++) {
      int st1 = idm1 + origz - 1;
      int i = (st1 - 1) / nrzr + 1;
      if (i > nvz - 1) i = nvz - 1;
      int k = st1 - nrzr * (i - 1);
      float v = (float)nrzr;
      v = (float)(k - 1) / v;
      float vi[4], sum[4];
      bspl4(v, vi);
      for (int i1 = 1; i1 <= 4; i1++) {
        float s = 0.0f;
        for (int j1 = 1; j1 <= 4; j1++)
          s = s + ui[j1 - 1] * (float)pv[(i - 2 + i1) * (nvx + 2) + (j - 2 + j1)];
        sum[i1 - 1] = vi[i1 - 1] * s;
      }
      velnr[(size_t)(idm2 - 1) * ORC_RMAX + (idm1 - 1)] = sum[0] + sum[1] + sum[2] + sum[3];
    }
  }
}

/* ---- the marching state (MODULE globalp / traveltime, inv/CalSurfG.f90:151-241) ------------- */
typedef struct {
  int nnx, nnz, ld;
  float gox, dnx, dnz;
  const float *veln;
  float *ttn;
  int *nsts;        /* -1 far, 0 alive, >0 heap slot */
  int *bpx, *bpz;   /* btg(1..ntr) */
  int ntr;
} march;

#define T(m, iz, ix) ((m)->ttn[(size_t)((ix)-1) * (m)->ld + ((iz)-1)])
#define S(m, iz, ix) ((m)->nsts[(size_t)((ix)-1) * (m)->ld + ((iz)-1)])
#define V(m, iz, ix) ((m)->veln[(size_t)((ix)-1) * (m)->ld + ((iz)-1)])
#define HT(m, p) T(m, (m)->bpz[p], (m)->bpx[p])

static void heap_swap(march *m, int a, int b) {
  int tx = m->bpx[a], tz = m->bpz[a];
  m->bpx[a] = m->bpx[b];
  m->bpz[a] = m->bpz[b];
  m->bpx[b] = tx;
  m->bpz[b] = tz;
}

/* sift a node up from slot tpc: shared tail of addtree (:760-774) and updtree (:876-890) */
static void sift_up(march *m, int iz, int ix, int tpc) {
  int tpp = tpc / 2;
  while (tpp > 0) {
    if (T(m, iz, ix) < HT(m, tpp)) {
      S(m, iz, ix) = tpp;
      S(m, m->bpz[tpp], m->bpx[tpp]) = tpc;
      heap_swap(m, tpc, tpp);
      tpc = tpp;
      tpp = tpc / 2;
    } else
      tpp = 0;
  }
}

/* inv/CalSurfG.f90:738 */
static void addtree(march *m, int iz, int ix) {
  m->ntr++;
  S(m, iz, ix) = m->ntr;
  m->bpx[m->ntr] = ix;
  m->bpz[m->ntr] = iz;
  sift_up(m, iz, ix, m->ntr);
}

/* inv/CalSurfG.f90:864 */
static void updtree(march *m, int iz, int ix) { sift_up(m, iz, ix, S(m, iz, ix)); }

/* inv/CalSurfG.f90:786 */
static void downtree(march *m) {
  if (m->ntr == 1) {
    m->ntr = 0;
    return;
  }
  S(m, m->bpz[m->ntr], m->bpx[m->ntr]) = 1;
  m->bpx[1] = m->bpx[m->ntr];
  m->bpz[1] = m->bpz[m->ntr];
  m->ntr--;
  int tpp = 1, tpc = 2;
  while (tpc < m->ntr) {
    float rd1 = HT(m, tpc), rd2 = HT(m, tpc + 1);
    if (rd1 > rd2) tpc = tpc + 1;
    rd1 = HT(m, tpc);
    rd2 = HT(m, tpp);
    if (rd1 < rd2) {
      S(m, m->bpz[tpp], m->bpx[tpp]) = tpc;
      S(m, m->bpz[tpc], m->bpx[tpc]) = tpp;
      heap_swap(m, tpc, tpp);
      tpp = tpc;
      tpc = 2 * tpp;
    } else
      tpc = m->ntr + 1;
  }
  if (tpc == m->ntr) {
    float rd1 = HT(m, tpc), rd2 = HT(m, tpp);
    if (rd1 < rd2) {
      S(m, m->bpz[tpp], m->bpx[tpp]) = tpc;
      S(m, m->bpz[tpc], m->bpx[tpc]) = tpp;
      heap_swap(m, tpc, tpp);
    }
  }
}

/* mixed-order upwind update of node (iz,ix): inv/CalSurfG.f90:557-729 */
static void fouds2(march *m, int iz, int ix) {
  int tsw1 = 0;
  float travm = 0.0f;
  float slown = 1.0f / V(m, iz, ix);
  float ri = EARTH;
  float risti = ri * sinf(m->gox + (float)(ix - 1) * m->dnx);
  float dnx = m->dnx, dnz = m->dnz;
  for (int j = ix - 1; j <= ix + 1; j += 2) {
    if (j < 1 || j > m->nnx) continue;
    int swj = -1, j2;
    if (j == ix - 1) {
      j2 = j - 1;
      if (j2 >= 1 && S(m, iz, j2) == 0) swj = 0;
    } else {
      j2 = j + 1;
      if (j2 <= m->nnx && S(m, iz, j2) == 0) swj = 0;
    }
    if (S(m, iz, j) == 0 && swj == 0) {
      swj = -1;
      if (T(m, iz, j) > T(m, iz, j2)) swj = 0;
    } else
      swj = -1;
    for (int k = iz - 1; k <= iz + 1; k += 2) {
      if (k < 1 || k > m->nnz) continue;
      int swk = -1, k2;
      if (k == iz - 1) {
        k2 = k - 1;
        if (k2 >= 1 && S(m, k2, ix) == 0) swk = 0;
      } else {
        k2 = k + 1;
        if (k2 <= m->nnz && S(m, k2, ix) == 0) swk = 0;
      }
      if (S(m, k, ix) == 0 && swk == 0) {
        swk = -1;
        if (T(m, k, ix) > T(m, k2, ix)) swk = 0;
      } else
        swk = -1;
      int swsol = 0;
      float a = 0, b = 0, c = 0, u, v, em, tref = 0, tdiv = 1.0f;
      if (swj == 0) {
        swsol = 1;
        if (swk == 0) {
          u = 2.0f * ri * dnx;
          v = 2.0f * risti * dnz;
          em = 4.0f * T(m, iz, j) - T(m, iz, j2) - 4.0f * T(m, k, ix);
          em = em + T(m, k2, ix);
          a = v * v + u * u;
          b = 2.0f * em * (u * u);
          c = (u * u) * (em * em - (slown * slown) * (v * v));
          tref = 4.0f * T(m, iz, j) - T(m, iz, j2);
          tdiv = 3.0f;
        } else if (S(m, k, ix) == 0) {
          u = risti * dnz;
          v = 2.0f * ri * dnx;
          em = 3.0f * T(m, k, ix) - 4.0f * T(m, iz, j) + T(m, iz, j2);
          a = v * v + 9.0f * (u * u);
          b = 6.0f * em * (u * u);
          c = (u * u) * (em * em - (slown * slown) * (v * v));
          tref = T(m, k, ix);
          tdiv = 1.0f;
        } else {
          u = 2.0f * ri * dnx;
          a = 1.0f;
          b = 0.0f;
          c = -(u * u) * (slown * slown);
          tref = 4.0f * T(m, iz, j) - T(m, iz, j2);
          tdiv = 3.0f;
        }
      } else if (S(m, iz, j) == 0) {
        swsol = 1;
        if (swk == 0) {
          u = ri * dnx;
          v = 2.0f * risti * dnz;
          em = 3.0f * T(m, iz, j) - 4.0f * T(m, k, ix) + T(m, k2, ix);
          a = v * v + 9.0f * (u * u);
          b = 6.0f * em * (u * u);
          c = (u * u) * (em * em - (v * v) * (slown * slown));
          tref = T(m, iz, j);
          tdiv = 1.0f;
        } else if (S(m, k, ix) == 0) {
          u = ri * dnx;
          v = risti * dnz;
          em = T(m, k, ix) - T(m, iz, j);
          a = u * u + v * v;
          b = -2.0f * (u * u) * em;
          c = (u * u) * (em * em - (v * v) * (slown * slown));
          tref = T(m, iz, j);
          tdiv = 1.0f;
        } else {
          a = 1.0f;
          b = 0.0f;
          c = -(slown * slown) * (ri * ri) * (dnx * dnx);
          tref = T(m, iz, j);
          tdiv = 1.0f;
        }
      } else {
        if (swk == 0) {
          swsol = 1;
          u = 2.0f * risti * dnz;
          a = 1.0f;
          b = 0.0f;
          c = -(u * u) * (slown * slown);
          tref = 4.0f * T(m, k, ix) - T(m, k2, ix);
          tdiv = 3.0f;
        } else if (S(m, k, ix) == 0) {
          swsol = 1;
          a = 1.0f;
          b = 0.0f;
          c = -(slown * slown) * (risti * risti) * (dnz * dnz);
          tref = T(m, k, ix);
          tdiv = 1.0f;
        }
      }
      if (swsol == 1) {
        float rd1 = b * b - 4.0f * a * c;
        if (rd1 < 0.0f) rd1 = 0.0f;
        float tdsh = (-b + sqrtf(rd1)) / (2.0f * a);
        float trav = (tref + tdsh) / tdiv;
        if (tsw1 == 1)
          travm = (trav < travm) ? trav : travm;
        else {
          travm = trav;
          tsw1 = 1;
        }
      }
    }
  }
  T(m, iz, ix) = travm;
}

/* inv/CalSurfG.f90:2293 ; nv(i,j): i along x, j along z */
static float bilinear(const float nv[2][2], float dnx, float dnz, float dsx, float dsz) {
  float biv = 0.0f;
  for (int i = 1; i <= 2; i++)
    for (int j = 1; j <= 2; j++) {
      float produ = (1.0f - fabsf(((float)(i - 1) * dnx - dsx) / dnx)) *
                    (1.0f - fabsf(((float)(j - 1) * dnz - dsz) / dnz));
      biv = biv + nv[i - 1][j - 1] * produ;
    }
  return biv;
}

/* main marching loop, inv/CalSurfG.f90:258-457.
 * urg=1: refined grid with early exit on the edges flagged in ex[4] = {x=1, x=nnx, z=1, z=nnz};
 * urg=2: restart from the nodes with nsts>0. */
static void travel(march *m, float goz, float scx, float scz, int urg, const int ex[4]) {
  m->ntr = 0;
  if (urg == 2) {
    for (int i = 1; i <= m->nnx; i++)
      for (int j = 1; j <= m->nnz; j++)
        if (S(m, j, i) > 0) addtree(m, j, i);
  } else {
    int isx = (int)((scx - m->gox) / m->dnx) + 1;
    int isz = (int)((scz - goz) / m->dnz) + 1;
    if (isx == m->nnx) isx--;
    if (isz == m->nnz) isz--;
    for (int i = 1; i <= m->nnx; i++)
      for (int j = 1; j <= m->nnz; j++) S(m, j, i) = -1;
    float vss[2][2];
    for (int i = 1; i <= 2; i++)
      for (int j = 1; j <= 2; j++) vss[i - 1][j - 1] = V(m, isz - 1 + j, isx - 1 + i);
    float dsx = (scx - m->gox) - (float)(isx - 1) * m->dnx;
    float dsz = (scz - goz) - (float)(isz - 1) * m->dnz;
    float vsrc = bilinear(vss, m->dnx, m->dnz, dsx, dsz);
    for (int i = 1; i <= 2; i++)
      for (int j = 1; j <= 2; j++) {
        float ax = dsx - (float)(i - 1) * m->dnx, az = dsz - (float)(j - 1) * m->dnz;
        float ds = sqrtf(ax * ax + az * az);
        T(m, isz - 1 + j, isx - 1 + i) = 2.0f * ds / (vss[i - 1][j - 1] + vsrc);
        addtree(m, isz - 1 + j, isx - 1 + i);
      }
  }
  while (m->ntr > 0) {
    int ix = m->bpx[1], iz = m->bpz[1];
    if (urg == 1) {
      int swrg = 0;
      if (ix == 1 && ex[0]) swrg = 1;
      if (ix == m->nnx && ex[1]) swrg = 1;
      if (iz == 1 && ex[2]) swrg = 1;
      if (iz == m->nnz && ex[3]) swrg = 1;
      if (swrg) {
        S(m, iz, ix) = 0;
        break;
      }
    }
    S(m, iz, ix) = 0;
    downtree(m);
    for (int i = ix - 1; i <= ix + 1; i += 2) {
      if (i < 1 || i > m->nnx) continue;
      if (S(m, iz, i) == -1) {
        fouds2(m, iz, i);
        addtree(m, iz, i);
      } else if (S(m, iz, i) > 0) {
        fouds2(m, iz, i);
        updtree(m, iz, i);
      }
    }
    for (int i = iz - 1; i <= iz + 1; i += 2) {
      if (i < 1 || i > m->nnz) continue;
      if (S(m, i, ix) == -1) {
        fouds2(m, i, ix);
        addtree(m, i, ix);
      } else if (S(m, i, ix) > 0) {
        fouds2(m, i, ix);
        updtree(m, i, ix);
      }
    }
  }
}

/* one (source, period): refined box -> bsplrefine -> travel(urg=1) -> inject -> travel(urg=2)
 * inv/CalSurfG.f90:1146-1314 */
int orc_fmm_field(const orc_geom *g, const double *pv, const float *veln, float scx, float scz,
                  float *ttn, float *ttnr, int *nstsr, float *velnr, orc_refbox *b) {
  int nnx = g->nnx, nnz = g->nnz;
  int isx = (int)((scx - g->gox) / g->dnx) + 1;
  int isz = (int)((scz - g->goz) / g->dnz) + 1;
  if (isx < 1 || isx > nnx || isz < 1 || isz > nnz) return 1;
  if (isx == nnx) isx--;
  if (isz == nnz) isz--;
  b->isx = isx;
  b->isz = isz;
  b->vnl = isx - SGS < 1 ? 1 : isx - SGS;
  b->vnr = isx + SGS > nnx ? nnx : isx + SGS;
  b->vnt = isz - SGS < 1 ? 1 : isz - SGS;
  b->vnb = isz + SGS > nnz ? nnz : isz + SGS;
  b->nnxr = (b->vnr - b->vnl) * SGDL + 1;
  b->nnzr = (b->vnb - b->vnt) * SGDL + 1;
  b->dnxr = g->dvx / (float)(GDX * SGDL);
  b->dnzr = g->dvz / (float)(GDZ * SGDL);
  b->goxr = g->gox + g->dnx * (float)(b->vnl - 1);
  b->gozr = g->goz + g->dnz * (float)(b->vnt - 1);

  size_t nr = (size_t)ORC_RMAX * ORC_RMAX;
  memset(velnr, 0, nr * sizeof(float));
  memset(ttnr, 0, nr * sizeof(float));
  for (size_t i = 0; i < nr; i++) nstsr[i] = -9;
  bsplrefine(g, pv, b, velnr);

  size_t maxbt = (size_t)nnx * nnz > nr ? (size_t)nnx * nnz : nr;
  int *bpx = (int *)malloc(sizeof(int) * (maxbt + 2)), *bpz = (int *)malloc(sizeof(int) * (maxbt + 2));
  march mr = {b->nnxr, b->nnzr, ORC_RMAX, b->goxr, b->dnxr, b->dnzr, velnr, ttnr, nstsr, bpx, bpz, 0};
  /* exit rule quirk kept verbatim (:366-377): vnr/vnb (coarse indices) are compared with the
   * REFINED nnx/nnz because the module variables hold the refined sizes at that point */
  int ex[4] = {b->vnl != 1, b->vnr != b->nnxr, b->vnt != 1, b->vnb != b->nnzr};
  travel(&mr, b->gozr, scx, scz, 1, ex);

  int *nsts = (int *)malloc(sizeof(int) * (size_t)nnx * nnz);
  for (size_t i = 0; i < (size_t)nnx * nnz; i++) nsts[i] = -1;
  march mc = {nnx, nnz, nnz, g->gox, g->dnx, g->dnz, veln, ttn, nsts, bpx, bpz, 0};
  for (int k = 1; k <= b->nnzr; k += SGDL) {
    int idm1 = b->vnt + (k - 1) / SGDL;
    for (int l = 1; l <= b->nnxr; l += SGDL) {
      int idm2 = b->vnl + (l - 1) / SGDL;
      int s = S(&mr, k, l);
      S(&mc, idm1, idm2) = s;
      if (s >= 0) T(&mc, idm1, idm2) = T(&mr, k, l);
    }
  }
  /* :1291-1308, in place and in this sweep order (later tests see earlier promotions to 1,
   * which is harmless since they only test for -1) */
  for (int k = 1; k <= nnx; k++)
    for (int l = 1; l <= nnz; l++)
      if (S(&mc, l, k) == 0) {
        if (l - 1 >= 1 && S(&mc, l - 1, k) == -1) S(&mc, l, k) = 1;
        if (l + 1 <= nnz && S(&mc, l + 1, k) == -1) S(&mc, l, k) = 1;
        if (k - 1 >= 1 && S(&mc, l, k - 1) == -1) S(&mc, l, k) = 1;
        if (k + 1 <= nnx && S(&mc, l, k + 1) == -1) S(&mc, l, k) = 1;
      }
  travel(&mc, g->goz, scx, scz, 2, ex);
  free(nsts);
  free(bpx);
  free(bpz);
  return 0;
}
