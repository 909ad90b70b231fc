/* oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the DAzimSurfTomo hot path (reference: Chuanming-Liu/DAzimSurfTomo,
 * src/src_inv_iso_joint/, cited per function as inv/<file>:<line>).  It is the checker for the HIP
 * product path and the "port" CPU baseline of bench.py.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load liboracle.so; the product library never links or calls it.
 *
 * Parity status: PINNED.  Every function here is checked (tests/test_oracle_*.py, -m "not gpu")
 * against golden vectors generated in the build container by the unmodified reference Fortran
 * compiled with AMD flang (oracle/_ref/libdazim_ref.so, recipe in oracle/Makefile, generator
 * tests/golden/make_golden.py) and against the reference-authored fixture
 * example/test1_syn_foward/output/period_Azm_tomo.real (column 4).
 *
 * Conventions: all 2-D grids are stored in the reference's (Fortran) memory order, i.e. a node
 * (iz,ix) of an nnz x nnx grid lives at [(ix-1)*ld + (iz-1)] ("[ix][iz]" in C terms, z fastest).
 * Arithmetic is fp32 for the eikonal/ray path and fp64 for the dispersion path, exactly as in the
 * reference; compile with -ffp-contract=off (the reference build has no FMA).
 */
#ifndef DAZIM_ORACLE_H
#define DAZIM_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NL 200      /* inv/surfdisp96.f:57  */
#define ORC_NP 60       /* inv/surfdisp96.f:59  */
#define ORC_RMAX 129    /* (2*sgs*sgdl+1), inv/CalSurfG.f90:1187-1196 with sgs=sgdl=8 */

/* ---- dispersion (disp.c) ------------------------------------------------------------------- */
/* inv/surfdisp96.f:52 with iflsph=1, iwave=2, mode=1, igr=0. returns #periods with a root. */
int orc_surfdisp96(const float *thk, const float *vp, const float *vs, const float *rho,
                   int nlayer, int kmax, const double *t, double *cg);
/* inv/surfdisp96.f:52 with every argument: iflsph 0/1, iwave 1 (Love) / 2 (Rayleigh), mode >= 1, igr 0 / >0 (surfdisp_full.c) */
int orc_surfdisp96_full(const float *thk, const float *vp, const float *vs, const float *rho, int nlayer, int iflsph,
                        int iwave, int mode, int igr, int kmax, const double *t, double *cg);
/* inv/CalSurfG.f90:2317 */
int orc_refine_layers(float minthk0, int mmax, const float *dep, const float *vp, const float *vs,
                      const float *rho, float *rthk, float *rvp, float *rvs, float *rrho);
/* inv/CalSurfG.f90:1 ; vel[nz][ny][nx]; pv[kmax][nx*ny]; sen_*[nz][kmax][nx*ny] (may be NULL) */
int orc_depthkernel(int nx, int ny, int nz, const float *vel, int kmax, const double *t,
                    const float *depz, float minthk, double *pv, double *svs, double *svp,
                    double *srho);

/* inv/CalSurfG.f90:49-53 (Brocher Vp(Vs), rho(Vp), fp32) */
void orc_brocher(float vs, float *vp, float *rho);

/* ---- TI eigenfunction kernels (tregn.c) ------------------------------------------------------ */
/* inv/tregn96.f:52 as depthkernelTI calls it (Rayleigh, mode 1, iflsph=1, dogam, hs=hr=0, solid layers):
 * d,TA,TC,TF,TL,TN,rho fp32[nl]; t,cp fp32[nt]; outputs fp32 [nt][nl].  returns 0 / 3 (fluid layer) */
int orc_tregn96(int nl, const float *d, const float *TA, const float *TC, const float *TF, const float *TL,
                const float *TN, const float *rho, int nt, const float *t, const float *cp, float *dcdah,
                float *dcdbv, float *dcdn);
/* inv/depthkernelTI.f90:2 ; pv[kmax][nx*ny] (nullable), lsen[nz-1][kmax][nx*ny] fp32 (= Lsen_Gsc) */
int orc_depthkernel_ti(int nx, int ny, int nz, const float *vel, int kmax, const double *t,
                       const float *depz, float minthk, double *pv, float *lsen);

/* ---- eikonal (fmm.c) ----------------------------------------------------------------------- */
typedef struct {
  int nvx, nvz;             /* B-spline vertices (nx-2, ny-2)              */
  int nnx, nnz;             /* coarse propagation grid                      */
  float gox, goz, dnx, dnz; /* coarse origin (colat, lon in rad) + spacing  */
  float dvx, dvz;           /* vertex spacing (rad)                         */
} orc_geom;

typedef struct {
  int vnl, vnr, vnt, vnb;   /* refined box bounds in coarse node indices (1-based)           */
  int nnxr, nnzr;           /* refined grid size                                              */
  int isx, isz;             /* coarse source cell                                             */
  float goxr, gozr, dnxr, dnzr;
} orc_refbox;

/* inv/CalSurfG.f90:1017-1038 */
void orc_geometry(int nx, int ny, float goxd, float gozd, float dvxd, float dvzd, orc_geom *g);
/* inv/CalSurfG.f90:1423 ; pv[(nvz+2)*(nvx+2)] doubles -> veln[nnx][nnz] */
void orc_gridder(const orc_geom *g, const double *pv, float *veln);
/* one (source,period) field: gridder'd coarse veln in, writes ttn[nnx][nnz], ttnr/nstsr/velnr
 * [129][129] (ld 129), box.  scratch-free.  returns 0, or 1 if the source is outside the grid
 * (reference STOPs, inv/CalSurfG.f90:1174). pv is the period's phase-velocity map. */
int orc_fmm_field(const orc_geom *g, const double *pv, const float *veln, float scx, float scz,
                  float *ttn, float *ttnr, int *nstsr, float *velnr, orc_refbox *box);

/* ---- rays + G rows (rays.c) ----------------------------------------------------------------- */
/* inv/CalSurfG.f90:1599 */
int orc_srtimes(const orc_geom *g, const float *veln, const float *ttn, float scx, float scz,
                float rcx, float rcz, float *t);
/* inv/CalSurfG.f90:1735 ; fdm[(nvx+2)][(nvz+2)] (Fortran fdm(0:nvz+1,0:nvx+1)); *rb |= rbint */
int orc_rpaths(const orc_geom *g, const orc_refbox *box, const float *veln, const float *ttn,
               const float *ttnr, const int *nstsr, float scx, float scz, float rcx, float rcz,
               float *fdm, int *rb);
/* inv/rpathsAzim.f90:16 ; as orc_rpaths plus the cos/sin(2 psi)-weighted grids fdmc, fdms */
int orc_rpaths_azim(const orc_geom *g, const orc_refbox *box, const float *veln, const float *ttn,
                    const float *ttnr, const int *nstsr, float scx, float scz, float rcx, float rcz,
                    float *fdm, float *fdmc, float *fdms, int *rb);
/* ray geometry rgx/rgz(1:nrp) of one ray (receiver first, source last), fwd/rpathsAzim.f90:221-380 = inv/CalSurfG.f90:1818-2066:
 * pxz[pcap][2] colatitude, longitude (rad); *npts = nrp */
int orc_ray_path(const orc_geom *g, const orc_refbox *box, const float *veln, const float *ttn, const float *ttnr,
                 const int *nstsr, float scx, float scz, float rcx, float rcz, float *pxz, int pcap, int *npts);
/* inv/CalSurfG.f90:1339-1364: G row of one ray from fdm and sen_*[nz][kmax][nx*ny]; kidx 0-based kernel slot; entries from
 * index 0 of rw/irow/icol (1-based ids); returns their number or -1 (more than maxnar) */
long orc_emit_row(int nx, int ny, int nz, const float *vels, const float *fdm, const double *svs, const double *svp,
                  const double *srho, int kmax, int kidx, int rowid, int64_t maxnar, float *rw, int *irow, int *icol);
/* the reference's dense copies of one ray's row (GVs; GGc, GGs when fdmc/fdms/lsen are given), inv/CalSurfG.f90:1369-1378,
 * inv/CalSurfGAniso_Joint.f90:759-775: all entries of the |fdm| >= ftol cells, dVs with the Brocher derivatives of the LAST
 * such cell (the reference's second loop reuses coe_a / coe_rho); rows of length (nx-2)(ny-2)(nz-1), assigned entries only */
void orc_dense_row(int nx, int ny, int nz, const float *vels, const float *fdm, const float *fdmc, const float *fdms,
                   const float *lsen, const double *svs, const double *svp, const double *srho, int kmax, int kidx,
                   float *gvs, float *ggc, float *ggs);
/* source loop of inv/CalSurfGAniso_Joint.f90:209 given Lsen_Gsc (lsen[nz-1][kmax][nx*ny], fp32):
 * rows have three column blocks dVs | Gc | Gs, n = 3*(nx-2)*(ny-2)*(nz-1) */
int orc_calsurfg_joint(int nx, int ny, int nz, const float *vels, float goxd, float gozd, float dvxd,
                       float dvzd, int kmax, const double *tRc, const float *depz, float minthk,
                       int nsrc, int nrcf, const float *scxf, const float *sczf, const float *rcxf,
                       const float *rczf, const int *nrc1, const int *nsrc1, const int *periods,
                       const float *lsen, int64_t maxnar, float *rw, int *irow, int *icol, float *dsurf,
                       int64_t *nar, int *nboundary);
/* inv/CalSurfG.f90:909 (whole iso G assembly).  Layouts as the Fortran arrays:
 * vels[nz][ny][nx]; scxf[kmax][nsrc]; rcxf[kmax][nsrc][nrcf]; nrc1/periods[kmax][nsrc]; nsrc1[kmax].
 * COO out: rw/irow/icol (1-based like iw/col), dsurf[dall]. returns 0 or reference-STOP code. */
int orc_calsurfg(int nx, int ny, int nz, const float *vels, float goxd, float gozd, float dvxd,
                 float dvzd, int kmax, const double *tRc, const float *depz, float minthk,
                 int nsrc, int nrcf, const float *scxf, const float *sczf, const float *rcxf,
                 const float *rczf, const int *nrc1, const int *nsrc1, const int *periods,
                 int64_t maxnar, float *rw, int *irow, int *icol, float *dsurf, int64_t *nar,
                 int *nboundary);

/* ---- solver (lsmr.c) ------------------------------------------------------------------------ */
/* inv/aprod.f90:7 (COO, 1-based indices) */
void orc_aprod(int mode, int m, int n, float *x, float *y, int64_t nar, const int *irow,
               const int *icol, const float *rw);
/* inv/lsmrblas.f90:247 */
float orc_nrm2(int n, const float *x);
/* inv/lsmrModule.f90:36 */
int orc_lsmr(int m, int n, int64_t nar, const int *irow, const int *icol, const float *rw,
             const float *b, float damp, float atol, float btol, float conlim, int itnlim,
             int localSize, float *x, int *istop, int *itn, float *normA, float *condA,
             float *normr, float *normAr, float *normx);
/* inv/TikhRegul.f90:2 (iso branch) appends rows; returns count3 */
int orc_tikhonov_iso(int nx, int ny, int nz, int dall, float weight, int64_t *nar, float *rw,
                     int *irow, int *icol);

#ifdef __cplusplus
}
#endif
#endif
