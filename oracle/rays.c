/* rays.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * CPU restatement of receiver traveltimes (srtimes), ray back-tracing with Frechet weights
 * (rpaths) and the isotropic G-row assembly / source loop of CalSurfG.  fp32 like the reference
 * (REAL(KIND=i10) = 4-byte real), fp64 only where the reference mixes real*8 kernels in.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define GDX 5
#define GDZ 5
#define EARTH 6371.0f

#define VC(iz, ix) veln[(size_t)((ix)-1) * nnz + ((iz)-1)]
#define TC(iz, ix) ttn[(size_t)((ix)-1) * nnz + ((iz)-1)]
#define TR(iz, ix) ttnr[(size_t)((ix)-1) * ORC_RMAX + ((iz)-1)]
#define SR(iz, ix) nstsr[(size_t)((ix)-1) * ORC_RMAX + ((iz)-1)]

/* dpl before the 0.5 factor: inv/CalSurfG.f90:1668-1672 and :1829-1833 */
static float min_cell(const orc_geom *g) {
  float dpl = g->dnx * EARTH;
  float rd1 = g->dnz * EARTH * sinf(g->gox);
  if (rd1 < dpl) dpl = rd1;
  rd1 = g->dnz * EARTH * sinf(g->gox + (float)(g->nnx - 1) * g->dnx);
  if (rd1 < dpl) dpl = rd1;
  return dpl;
}

/* bilinear velocity at offset (drx,drz) inside coarse cell (ipx,ipz), x outer / z inner as
 * inv/CalSurfG.f90:2129-2137 */
static float vel_at(const orc_geom *g, const float *veln, int ipx, int ipz, float drx, float drz) {
  int nnx = g->nnx, nnz = g->nnz;
  float vel = 0.0f;
  for (int l = 1; l <= 2; l++)
    for (int m = 1; m <= 2; m++) {
      float produ = (1.0f - fabsf(((float)(m - 1) * g->dnz - drz) / g->dnz));
      produ = produ * (1.0f - fabsf(((float)(l - 1) * g->dnx - drx) / g->dnx));
      if (ipz - 1 + m <= nnz && ipx - 1 + l <= nnx) vel = vel + VC(ipz - 1 + m, ipx - 1 + l) * produ;
    }
  return vel;
}

static void basis(float v, float b[4]) { /* inv/CalSurfG.f90:2145-2148 */
  float om = 1.0f - v;
  b[0] = om * om * om / 6.0f;
  b[1] = (4.0f - 6.0f * (v * v) + 3.0f * (v * v * v)) / 6.0f;
  b[2] = (1.0f + 3.0f * v + 3.0f * (v * v) - 3.0f * (v * v * v)) / 6.0f;
  b[3] = v * v * v / 6.0f;
}

/* inv/CalSurfG.f90:1599-1722 */
int orc_srtimes(const orc_geom *g, const float *veln, const float *ttn, float scx, float scz,
                float rcx, float rcz, float *out) {
  int nnx = g->nnx, nnz = g->nnz;
  float gox = g->gox, goz = g->goz, dnx = g->dnx, dnz = g->dnz;
  int irx = (int)((rcx - gox) / dnx) + 1;
  int irz = (int)((rcz - goz) / dnz) + 1;
  if (irx < 1 || irx > nnx || irz < 1 || irz > nnz) return 2;
  if (irx == nnx) irx--;
  if (irz == nnz) irz--;
  int isx = (int)((scx - gox) / dnx) + 1;
  int isz = (int)((scz - goz) / dnz) + 1;
  float dpl = min_cell(g);
  float sred = ((scx - rcx) * EARTH) * ((scx - rcx) * EARTH);
  float e2 = (scz - rcz) * EARTH * sinf(rcx);
  sred = sred + e2 * e2;
  sred = sqrtf(sred);
  int sw = 0;
  if (sred < dpl) sw = 1;
  if (isx == irx && isz == irz) sw = 1;
  float trr;
  if (sw) {
    float vss[2][2], vels = 0.0f, velr = 0.0f;
    for (int pass = 0; pass < 2; pass++) {
      int cx = pass ? irx : isx, cz = pass ? irz : isz;
      float px = pass ? rcx : scx, pz = pass ? rcz : scz;
      for (int k = 1; k <= 2; k++)
        for (int l = 1; l <= 2; l++) vss[k - 1][l - 1] = VC(cz - 1 + l, cx - 1 + k);
      float drx = (px - gox) - (float)(cx - 1) * dnx;
      float drz = (pz - goz) - (float)(cz - 1) * dnz;
      float biv = 0.0f; /* bilinear, inv/CalSurfG.f90:2293 */
      for (int i = 1; i <= 2; i++)
        for (int j = 1; j <= 2; j++) {
          float produ = (1.0f - fabsf(((float)(i - 1) * dnx - drx) / dnx)) *
                        (1.0f - fabsf(((float)(j - 1) * dnz - drz) / dnz));
          biv = biv + vss[i - 1][j - 1] * produ;
        }
      if (pass) velr = biv; else vels = biv;
    }
    trr = 2.0f * sred / (vels + velr);
  } else {
    float drx = (rcx - gox) - (float)(irx - 1) * dnx;
    float drz = (rcz - goz) - (float)(irz - 1) * dnz;
    trr = 0.0f;
    for (int k = 1; k <= 2; k++)
      for (int l = 1; l <= 2; l++) {
        float produ = (1.0f - fabsf(((float)(l - 1) * dnz - drz) / dnz)) *
                      (1.0f - fabsf(((float)(k - 1) * dnx - drx) / dnx));
        trr = trr + TC(irz - 1 + l, irx - 1 + k) * produ;
      }
  }
  *out = trr;
  return 0;
}

/* azdist, inv/rpathsAzim.f90:687-793, with the reference's implicit typing: the arguments, piby2 and
 * predel are REAL*4, pi is the double of a single-precision literal.  Only `az` is returned. */
static float azdist_az(float stalat, float stalon, float evtlat, float evtlon) {
  const double pi = (double)3.1415926535898f;
  const float piby2 = (float)(pi / (double)2.f);
  const double rad = (double)2.f * pi / (double)360.f;
  const double sph = (double)(1.0f / 298.257f);
  const double scolat = (double)piby2 - atan((1. - sph) * (1. - sph) * tan((double)stalat * rad));
  const double ecolat = (double)piby2 - atan((1. - sph) * (1. - sph) * tan((double)evtlat * rad));
  const double slon = (double)stalon * rad, elon = (double)evtlon * rad;
  const double a = sin(scolat) * cos(slon), b = sin(scolat) * sin(slon), c = cos(scolat);
  const double dd = sin(elon), ee = -cos(elon), cc = cos(ecolat);
  const double gg = -cc * ee, hh = cc * dd, kk = -sin(ecolat);
  const double rhs1 = (a - dd) * (a - dd) + (b - ee) * (b - ee) + c * c - (double)2.f;
  const double rhs2 = (a - gg) * (a - gg) + (b - hh) * (b - hh) + (c - kk) * (c - kk) - (double)2.f;
  double daz = atan2(rhs1, rhs2);
  if (daz < 0.0) daz = daz + 2 * pi;
  float az = (float)(daz / rad);
  if (fabsf(az - 360.f) < .00001f) az = 0.0f;
  return az;
}

/* inv/CalSurfG.f90:1735-2283 (asgr=1 branch; cfd always on); with fdmc/fdms != NULL it is
 * rpathsAzim (inv/rpathsAzim.f90:16-684): the same ray plus cos/sin(2 psi)-weighted grids */
static int rpaths_impl(const orc_geom *g, const orc_refbox *b, const float *veln, const float *ttn,
                       const float *ttnr, const int *nstsr, float scx, float scz, float rcx, float rcz,
                       float *fdm, float *fdmc, float *fdms, int *rb, float *pxz, int pcap, int *npts) {
  int nnx = g->nnx, nnz = g->nnz, nvz = g->nvz, nvx = g->nvx;
  int np = 0; /* ray points rgx/rgz(1:nrp), fwd/rpathsAzim.f90:221-380 (pxz nullable) */
#define PUSH_PT(X, Z) do { if (pxz && np < pcap) { pxz[2 * np] = (X); pxz[2 * np + 1] = (Z); } np++; } while (0)
  float gox = g->gox, goz = g->goz, dnx = g->dnx, dnz = g->dnz, dvx = g->dvx, dvz = g->dvz;
  float goxr = b->goxr, gozr = b->gozr, dnxr = b->dnxr, dnzr = b->dnzr;
  int nnxr = b->nnxr, nnzr = b->nnzr;
  int ldf = nvz + 2;
  memset(fdm, 0, sizeof(float) * (size_t)ldf * (nvx + 2));
  if (fdmc) {
    memset(fdmc, 0, sizeof(float) * (size_t)ldf * (nvx + 2));
    memset(fdms, 0, sizeof(float) * (size_t)ldf * (nvx + 2));
  }
  const float PI_F = 3.1415926535898f;
  int isx = (int)((scx - goxr) / dnxr) + 1;
  int isz = (int)((scz - gozr) / dnzr) + 1;
  float dpl = 0.5f * min_cell(g);
  int ipx = (int)((rcx - gox) / dnx) + 1;
  int ipz = (int)((rcz - goz) / dnz) + 1;
  if (ipx < 1 || ipx >= nnx || ipz < 1 || ipz >= nnz) return 3;
  float x0 = rcx, z0 = rcz; /* rgx(j), rgz(j) */
  int sw = 0;
  float sred = ((scx - x0) * EARTH) * ((scx - x0) * EARTH);
  float e2 = (scz - z0) * EARTH * sinf(x0);
  sred = sqrtf(sred + e2 * e2);
  if (sred < 2.0f * dpl) sw = 1;
  int ipxr = (int)((rcx - goxr) / dnxr) + 1;
  int ipzr = (int)((rcz - gozr) / dnzr) + 1;
  int igref = 1;
  if (ipxr < 1 || ipxr >= nnxr) igref = 0;
  if (ipzr < 1 || ipzr >= nnzr) igref = 0;
  if (igref == 1) {
    if (SR(ipzr, ipxr) != 0 || SR(ipzr + 1, ipxr) != 0) igref = 0;
    if (SR(ipzr, ipxr + 1) != 0 || SR(ipzr + 1, ipxr + 1) != 0) igref = 0;
  }
  if (sw == 0 && igref == 1 && ipxr == isx && ipzr == isz) sw = 1;
  PUSH_PT(rcx, rcz);                 /* rgx(1) = the receiver */
  if (sw == 1) PUSH_PT(scx, scz);    /* nrp = 2 */
  long maxrp = (long)nnx * nnz;
  for (long j = 1; j <= maxrp; j++) {
    if (sw == 1) break;
    float dtx, dtz;
    if (igref == 1) {
      dtx = TR(ipzr, ipxr + 1) - TR(ipzr, ipxr);
      dtx = dtx + TR(ipzr + 1, ipxr + 1) - TR(ipzr + 1, ipxr);
      dtx = dtx / (2.0f * EARTH * dnxr);
      dtz = TR(ipzr + 1, ipxr) - TR(ipzr, ipxr);
      dtz = dtz + TR(ipzr + 1, ipxr + 1) - TR(ipzr, ipxr + 1);
      dtz = dtz / (2.0f * EARTH * sinf(x0) * dnzr);
    } else {
      dtx = TC(ipz, ipx + 1) - TC(ipz, ipx);
      dtx = dtx + TC(ipz + 1, ipx + 1) - TC(ipz + 1, ipx);
      dtx = dtx / (2.0f * EARTH * dnx);
      dtz = TC(ipz + 1, ipx) - TC(ipz, ipx);
      dtz = dtz + TC(ipz + 1, ipx + 1) - TC(ipz, ipx + 1);
      dtz = dtz / (2.0f * EARTH * sinf(x0) * dnz);
    }
    float rd1 = sqrtf(dtx * dtx + dtz * dtz);
    float x1 = x0 - dpl * dtx / (EARTH * rd1);
    float z1 = z0 - dpl * dtz / (EARTH * sinf(x0) * rd1);
    int ipxo = ipx, ipzo = ipz;
    ipxr = (int)((x1 - goxr) / dnxr) + 1;
    ipzr = (int)((z1 - gozr) / dnzr) + 1;
    igref = 1;
    if (ipxr < 1 || ipxr >= nnxr) igref = 0;
    if (ipzr < 1 || ipzr >= nnzr) igref = 0;
    if (igref == 1) {
      if (SR(ipzr, ipxr) != 0 || SR(ipzr + 1, ipxr) != 0) igref = 0;
      if (SR(ipzr, ipxr + 1) != 0 || SR(ipzr + 1, ipxr + 1) != 0) igref = 0;
    }
    ipx = (int)((x1 - gox) / dnx) + 1;
    ipz = (int)((z1 - goz) / dnz) + 1;
    sred = ((scx - x1) * EARTH) * ((scx - x1) * EARTH);
    e2 = (scz - z1) * EARTH * sinf(x1);
    sred = sqrtf(sred + e2 * e2);
    sw = 0;
    if (sred < 2.0f * dpl) sw = 1;
    if (sw == 0 && igref == 1 && ipxr == isx && ipzr == isz) sw = 1;
    if (ipx < 1) { x1 = gox; ipx = 1; *rb = 1; }
    if (ipx >= nnx) { x1 = gox + (float)(nnx - 1) * dnx; ipx = nnx - 1; *rb = 1; }
    if (ipz < 1) { z1 = goz; ipz = 1; *rb = 1; }
    if (ipz >= nnz) { z1 = goz + (float)(nnz - 1) * dnz; ipz = nnz - 1; *rb = 1; }
    PUSH_PT(x1, z1);                   /* rgx(j+1), after the clipping */
    if (sw == 1) PUSH_PT(scx, scz);    /* rgx(j+2) = the source, nrp = j+2 */
    float c2psi = 0.0f, s2psi = 0.0f;
    if (fdmc) { /* inv/rpathsAzim.f90:415-423 */
      const float rgx1 = (PI_F / 2 - x0) * 180.0f / PI_F, rgz1 = z0 * 180.0f / PI_F;
      const float rgx2 = (PI_F / 2 - x1) * 180.0f / PI_F, rgz2 = z1 * 180.0f / PI_F;
      const float az = azdist_az(rgx2, rgz2, rgx1, rgz1);
      const float rgpsi = az / 180 * PI_F;
      c2psi = cosf(2.0f * rgpsi);
      s2psi = sinf(2.0f * rgpsi);
    }
    /* Frechet part, :2077-2229 */
    int ivx = (ipx - 1) / GDX + 1, ivz = (ipz - 1) / GDZ + 1;
    int ivxo = (ipxo - 1) / GDX + 1, ivzo = (ipzo - 1) / GDZ + 1;
    int nhp = 0, chp[4];
    float vrat[4];
    if (ivx != ivxo) {
      nhp++;
      float xi = (ivx > ivxo) ? gox + (float)(ivx - 1) * dvx : gox + (float)ivx * dvx;
      vrat[nhp - 1] = (xi - x0) / (x1 - x0);
      chp[nhp - 1] = 1;
    }
    if (ivz != ivzo) {
      nhp++;
      float zi = (ivz > ivzo) ? goz + (float)(ivz - 1) * dvz : goz + (float)ivz * dvz;
      float r = (zi - z0) / (z1 - z0);
      if (nhp == 1) {
        vrat[0] = r;
        chp[0] = 2;
      } else if (r >= vrat[nhp - 2]) {
        vrat[nhp - 1] = r;
        chp[nhp - 1] = 2;
      } else {
        vrat[nhp - 1] = vrat[nhp - 2];
        chp[nhp - 1] = chp[nhp - 2];
        vrat[nhp - 2] = r;
        chp[nhp - 2] = 2;
      }
    }
    nhp++;
    vrat[nhp - 1] = 1.0f;
    chp[nhp - 1] = 0;
    float drx = (x0 - gox) - (float)(ipxo - 1) * dnx;
    float drz = (z0 - goz) - (float)(ipzo - 1) * dnz;
    float vel = vel_at(g, veln, ipxo, ipzo, drx, drz);
    drx = (x0 - gox) - (float)(ivxo - 1) * dvx;
    drz = (z0 - goz) - (float)(ivzo - 1) * dvz;
    float vi[4], wi[4], vio[4], wio[4];
    basis(drx / dvx, vi);
    basis(drz / dvz, wi);
    int ivxt = ivxo, ivzt = ivzo;
    for (int k = 1; k <= nhp; k++) {
      float velo = vel;
      memcpy(vio, vi, sizeof vi);
      memcpy(wio, wi, sizeof wi);
      if (k > 1) {
        if (chp[k - 2] == 1) ivxt = ivx;
        else if (chp[k - 2] == 2) ivzt = ivz;
      }
      float rigz = z0 + vrat[k - 1] * (z1 - z0);
      float rigx = x0 + vrat[k - 1] * (x1 - x0);
      int ipxt = (int)((rigx - gox) / dnx) + 1;
      int ipzt = (int)((rigz - goz) / dnz) + 1;
      drx = (rigx - gox) - (float)(ipxt - 1) * dnx;
      drz = (rigz - goz) - (float)(ipzt - 1) * dnz;
      vel = vel_at(g, veln, ipxt, ipzt, drx, drz);
      drx = (rigx - gox) - (float)(ivxt - 1) * dvx;
      drz = (rigz - goz) - (float)(ivzt - 1) * dvz;
      basis(drx / dvx, vi);
      basis(drz / dvz, wi);
      float dinc = (k == 1) ? vrat[0] * dpl : (vrat[k - 1] - vrat[k - 2]) * dpl;
      for (int l = 1; l <= 4; l++)
        for (int m = 1; m <= 4; m++) {
          const float rdc1 = vi[m - 1] * wi[l - 1] / (vel * vel);
          const float rdc2 = vio[m - 1] * wio[l - 1] / (velo * velo);
          float r1 = -(rdc1 + rdc2) * dinc / 2.0f;
          const size_t fi = (size_t)(ivxt - 2 + m) * ldf + (ivzt - 2 + l);
          fdm[fi] = r1 + fdm[fi];
          if (fdmc) { /* inv/rpathsAzim.f90:580-586 */
            r1 = -(rdc1 * c2psi + rdc2 * c2psi) * dinc / 2.0f;
            fdmc[fi] = r1 + fdmc[fi];
            r1 = -(rdc1 * s2psi + rdc2 * s2psi) * dinc / 2.0f;
            fdms[fi] = r1 + fdms[fi];
          }
        }
    }
    x0 = x1;
    z0 = z1;
  }
  if (npts) *npts = np;
  return 0;
#undef PUSH_PT
}

/* the ray geometry alone (what the reference writes to raypath_refmdl_<T>s.dat, fwd/rpathsAzim.f90:617-625): pxz[pcap][2]
 * (colatitude, longitude in rad), *npts = nrp (may exceed pcap: then only the first pcap points were stored) */
int orc_ray_path(const orc_geom *g, const orc_refbox *b, const float *veln, const float *ttn, const float *ttnr,
                 const int *nstsr, float scx, float scz, float rcx, float rcz, float *pxz, int pcap, int *npts) {
  float *fdm = (float *)malloc(sizeof(float) * (size_t)(g->nvz + 2) * (g->nvx + 2));
  int rb = 0;
  *npts = 0;
  int rc = rpaths_impl(g, b, veln, ttn, ttnr, nstsr, scx, scz, rcx, rcz, fdm, NULL, NULL, &rb, pxz, pcap, npts);
  free(fdm);
  return rc;
}

int orc_rpaths(const orc_geom *g, const orc_refbox *b, const float *veln, const float *ttn,
               const float *ttnr, const int *nstsr, float scx, float scz, float rcx, float rcz,
               float *fdm, int *rb) {
  return rpaths_impl(g, b, veln, ttn, ttnr, nstsr, scx, scz, rcx, rcz, fdm, NULL, NULL, rb, NULL, 0, NULL);
}
int orc_rpaths_azim(const orc_geom *g, const orc_refbox *b, const float *veln, const float *ttn,
                    const float *ttnr, const int *nstsr, float scx, float scz, float rcx, float rcz,
                    float *fdm, float *fdmc, float *fdms, int *rb) {
  return rpaths_impl(g, b, veln, ttn, ttnr, nstsr, scx, scz, rcx, rcz, fdm, fdmc, fdms, rb, NULL, 0, NULL);
}

/* G-row of one ray: inv/CalSurfG.f90:1339-1364.  sen_*[nz][kmax][nx*ny]; kidx 0-based period.
 * appends (1-based) COO entries; returns #appended or -1 if it would overflow maxnar. */
static long emit_row(int nx, int ny, int nz, const float *vels, const float *fdm, const float *fdmc,
                     const float *fdms, const float *lsen, const double *svs,
                     const double *svp, const double *srho, int kmax, int kidx, int rowid,
                     float *row, int64_t nar, int64_t maxnar, float *rw, int *irow, int *icol) {
  const float ftol = 1e-4f;
  int nvx = nx - 2, nvz = ny - 2, nparpi = nvx * nvz * (nz - 1);
  const int nblk = fdmc ? 3 : 1; /* joint: dVs | Gc | Gs column blocks, inv/CalSurfGAniso_Joint.f90:728-738 */
  size_t ncol = (size_t)nx * ny;
  memset(row, 0, sizeof(float) * nparpi * nblk);
  for (int jj = 1; jj <= nvz; jj++)
    for (int kk = 1; kk <= nvx; kk++) {
      float f = fdm[(size_t)kk * (nvz + 2) + jj];
      if (!(fabsf(f) >= ftol)) continue;
      size_t cell = (size_t)jj * (nvx + 2) + kk; /* 0-based of jj*(nvx+2)+kk+1 */
      for (int k = 1; k <= nz - 1; k++) {
        float v = vels[((size_t)(k - 1) * ny + jj) * nx + kk]; /* vels(kk+1,jj+1,k) */
        float coe_a = (2.0947f - 0.8206f * 2 * v + 0.2683f * 3 * (v * v) - 0.0251f * 4 * (v * v * v));
        float vpft = 0.9409f + 2.0947f * v - 0.8206f * (v * v) + 0.2683f * (v * v * v) -
                     0.0251f * (v * v * v * v);
        float coe_rho = coe_a * (1.6612f - 0.4721f * 2 * vpft + 0.0671f * 3 * (vpft * vpft) -
                                 0.0043f * 4 * (vpft * vpft * vpft) +
                                 0.000106f * 5 * (vpft * vpft * vpft * vpft));
        size_t si = ((size_t)(k - 1) * kmax + kidx) * ncol + cell;
        double r = (svp[si] * (double)coe_a + srho[si] * (double)coe_rho + svs[si]) * (double)f;
        row[(size_t)(k - 1) * nvz * nvx + (jj - 1) * nvx + kk - 1] = (float)r;
        if (fdmc) {
          const float L = lsen[si];
          row[(size_t)nparpi + (size_t)(k - 1) * nvz * nvx + (jj - 1) * nvx + kk - 1] = L * fdmc[(size_t)kk * (nvz + 2) + jj];
          row[(size_t)2 * nparpi + (size_t)(k - 1) * nvz * nvx + (jj - 1) * nvx + kk - 1] = L * fdms[(size_t)kk * (nvz + 2) + jj];
        }
      }
    }
  long cnt = 0;
  for (int nn = 1; nn <= nparpi * nblk; nn++)
    if (fabsf(row[nn - 1]) > ftol) {
      if (nar + cnt >= maxnar) return -1;
      rw[nar + cnt] = row[nn - 1];
      irow[nar + cnt] = rowid;
      icol[nar + cnt] = nn;
      cnt++;
    }
  return cnt;
}

/* the row of one ray from its Frechet grid and the depth kernels, inv/CalSurfG.f90:1339-1364, as an entry point of its own
 * (tests that sample rays of a large batch need rows without the whole depthkernel loop of orc_calsurfg).  kidx 0-based
 * kernel slot; COO entries (1-based) are written from index 0; returns their number, -1 if more than maxnar. */
long orc_emit_row(int nx, int ny, int nz, const float *vels, const float *fdm, const double *svs, const double *svp,
                  const double *srho, int kmax, int kidx, int rowid, int64_t maxnar, float *rw, int *irow, int *icol) {
  float *row = (float *)malloc(sizeof(float) * (size_t)(nx - 2) * (ny - 2) * (nz - 1));
  long c = emit_row(nx, ny, nz, vels, fdm, NULL, NULL, NULL, svs, svp, srho, kmax, kidx, rowid, row, 0, maxnar, rw, irow, icol);
  free(row);
  return c;
}

/* The reference's DENSE copies of one ray's row, inv/CalSurfG.f90:1369-1378 (GVs) and inv/CalSurfGAniso_Joint.f90:759-775 (GVs,
 * GGc, GGs): every entry of the cells with |fdm| >= ftol, no second threshold -- and the dVs block is formed with coe_a /
 * coe_rho as the FIRST loop (:1339-1354) left them, i.e. those of the last cell with |fdm| >= ftol in (jj, kk) order: the
 * second loop does not recompute them.  gvs (ggc, ggs; nullable with fdmc) are rows of length (nx-2)(ny-2)(nz-1), overwritten
 * only where the reference assigns (the caller zeroes them like inv/Main_Jt.f90:388-390). */
void orc_dense_row(int nx, int ny, int nz, const float *vels, const float *fdm, const float *fdmc, const float *fdms,
                   const float *lsen, const double *svs, const double *svp, const double *srho, int kmax, int kidx,
                   float *gvs, float *ggc, float *ggs) {
  const float ftol = 1e-4f;
  int nvx = nx - 2, nvz = ny - 2;
  size_t ncol = (size_t)nx * ny;
  int jjL = 0, kkL = 0;
  for (int jj = 1; jj <= nvz; jj++)
    for (int kk = 1; kk <= nvx; kk++)
      if (fabsf(fdm[(size_t)kk * (nvz + 2) + jj]) >= ftol) { jjL = jj; kkL = kk; }
  if (!jjL) return; /* no cell: the reference's coe_a would be whatever the previous ray left; no entry is assigned either */
  for (int jj = 1; jj <= nvz; jj++)
    for (int kk = 1; kk <= nvx; kk++) {
      float f = fdm[(size_t)kk * (nvz + 2) + jj];
      if (!(fabsf(f) >= ftol)) continue;
      size_t cell = (size_t)jj * (nvx + 2) + kk;
      for (int k = 1; k <= nz - 1; k++) {
        float v = vels[((size_t)(k - 1) * ny + jjL) * nx + kkL]; /* the LAST cell's column, see above */
        float coe_a = (2.0947f - 0.8206f * 2 * v + 0.2683f * 3 * (v * v) - 0.0251f * 4 * (v * v * v));
        float vpft = 0.9409f + 2.0947f * v - 0.8206f * (v * v) + 0.2683f * (v * v * v) - 0.0251f * (v * v * v * v);
        float coe_rho = coe_a * (1.6612f - 0.4721f * 2 * vpft + 0.0671f * 3 * (vpft * vpft) -
                                 0.0043f * 4 * (vpft * vpft * vpft) + 0.000106f * 5 * (vpft * vpft * vpft * vpft));
        size_t si = ((size_t)(k - 1) * kmax + kidx) * ncol + cell;
        size_t o = (size_t)(k - 1) * nvz * nvx + (jj - 1) * nvx + kk - 1;
        gvs[o] = (float)((svp[si] * (double)coe_a + srho[si] * (double)coe_rho + svs[si]) * (double)f);
        if (fdmc && ggc && ggs) {
          ggc[o] = lsen[si] * fdmc[(size_t)kk * (nvz + 2) + jj];
          ggs[o] = lsen[si] * fdms[(size_t)kk * (nvz + 2) + jj];
        }
      }
    }
}

/* inv/CalSurfG.f90:909-1422 ; with lsen != NULL the source loop of CalSurfGAnisoJoint
 * (inv/CalSurfGAniso_Joint.f90:488-792): rpathsAzim and three column blocks per row.
 * lsen[nz-1][kmax][nx*ny] = Lsen_Gsc from depthkernelTI (TI kernels are an input here). */
static int calsurfg_impl(int nx, int ny, int nz, const float *vels, float goxd, float gozd, float dvxd,
                 float dvzd, int kmax, const double *tRc, const float *depz, float minthk,
                 int nsrc, int nrcf, const float *scxf, const float *sczf, const float *rcxf,
                 const float *rczf, const int *nrc1, const int *nsrc1, const int *periods,
                 const float *lsen,
                 int64_t maxnar, float *rw, int *irow, int *icol, float *dsurf, int64_t *nar_out,
                 int *nboundary) {
  orc_geom g;
  orc_geometry(nx, ny, goxd, gozd, dvxd, dvzd, &g);
  size_t ncol = (size_t)nx * ny, nk = (size_t)nz * kmax * ncol;
  double *pv = (double *)malloc(sizeof(double) * kmax * ncol);
  double *svs = (double *)malloc(sizeof(double) * nk), *svp = (double *)malloc(sizeof(double) * nk),
         *srho = (double *)malloc(sizeof(double) * nk);
  orc_depthkernel(nx, ny, nz, vels, kmax, tRc, depz, minthk, pv, svs, svp, srho);
  size_t nn = (size_t)g.nnx * g.nnz, nr = (size_t)ORC_RMAX * ORC_RMAX;
  float *veln = (float *)malloc(sizeof(float) * nn), *ttn = (float *)malloc(sizeof(float) * nn);
  float *ttnr = (float *)malloc(sizeof(float) * nr), *velnr = (float *)malloc(sizeof(float) * nr);
  int *nstsr = (int *)malloc(sizeof(int) * nr);
  size_t nfd = (size_t)(g.nvx + 2) * (g.nvz + 2);
  float *fdm = (float *)malloc(sizeof(float) * nfd * 3), *fdmc = lsen ? fdm + nfd : NULL, *fdms = lsen ? fdm + 2 * nfd : NULL;
  float *row = (float *)malloc(sizeof(float) * (size_t)g.nvx * g.nvz * (nz - 1) * 3);
  int64_t nar = 0;
  int count1 = 0, rc = 0, rbindex = 0;
  int rb = 0; /* rbint is never reset inside CalSurfG (:1061), so it latches */
  for (int knumi = 0; knumi < kmax && !rc; knumi++)
    for (int s = 0; s < nsrc1[knumi] && !rc; s++) {
      size_t si = (size_t)knumi * nsrc + s;
      const double *pvk = pv + (size_t)(periods[si] - 1) * ncol;
      orc_gridder(&g, pvk, veln);
      orc_refbox box;
      float x = scxf[si], z = sczf[si];
      rc = orc_fmm_field(&g, pvk, veln, x, z, ttn, ttnr, nstsr, velnr, &box);
      if (rc) break;
      for (int r = 0; r < nrc1[si]; r++) {
        float rx = rcxf[si * nrcf + r], rz = rczf[si * nrcf + r];
        float t;
        if ((rc = orc_srtimes(&g, veln, ttn, x, z, rx, rz, &t))) break;
        count1++;
        dsurf[count1 - 1] = t;
        if ((rc = rpaths_impl(&g, &box, veln, ttn, ttnr, nstsr, x, z, rx, rz, fdm, fdmc, fdms, &rb, NULL, 0, NULL))) break;
        long c = emit_row(nx, ny, nz, vels, fdm, fdmc, fdms, lsen, svs, svp, srho, kmax, knumi, count1, row, nar, maxnar, rw, irow, icol);
        if (c < 0) { rc = 4; break; }
        nar += c;
      }
      if (rb) rbindex++;
    }
  *nar_out = nar;
  *nboundary = rbindex;
  free(pv); free(svs); free(svp); free(srho); free(veln); free(ttn); free(ttnr); free(velnr);
  free(nstsr); free(fdm); free(row);
  return rc;
}

int orc_calsurfg(int nx, int ny, int nz, const float *vels, float goxd, float gozd, float dvxd,
                 float dvzd, int kmax, const double *tRc, const float *depz, float minthk,
                 int nsrc, int nrcf, const float *scxf, const float *sczf, const float *rcxf,
                 const float *rczf, const int *nrc1, const int *nsrc1, const int *periods,
                 int64_t maxnar, float *rw, int *irow, int *icol, float *dsurf, int64_t *nar_out,
                 int *nboundary) {
  return calsurfg_impl(nx, ny, nz, vels, goxd, gozd, dvxd, dvzd, kmax, tRc, depz, minthk, nsrc, nrcf, scxf, sczf, rcxf,
                       rczf, nrc1, nsrc1, periods, NULL, maxnar, rw, irow, icol, dsurf, nar_out, nboundary);
}
int orc_calsurfg_joint(int nx, int ny, int nz, const float *vels, float goxd, float gozd, float dvxd,
                       float dvzd, int kmax, const double *tRc, const float *depz, float minthk,
                       int nsrc, int nrcf, const float *scxf, const float *sczf, const float *rcxf,
                       const float *rczf, const int *nrc1, const int *nsrc1, const int *periods,
                       const float *lsen, int64_t maxnar, float *rw, int *irow, int *icol, float *dsurf,
                       int64_t *nar_out, int *nboundary) {
  return calsurfg_impl(nx, ny, nz, vels, goxd, gozd, dvxd, dvzd, kmax, tRc, depz, minthk, nsrc, nrcf, scxf, sczf, rcxf,
                       rczf, nrc1, nsrc1, periods, lsen, maxnar, rw, irow, icol, dsurf, nar_out, nboundary);
}
