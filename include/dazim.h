/* dazim.h -- C ABI of the MI355X-native DAzimSurfTomo hot path (libdazim_hip.so).
 *
 * The reference (Chuanming-Liu/DAzimSurfTomo) has no FFI; its seams are Fortran procedure calls
 * with module-global state.  Each entry point below replaces one seam and
 * cites it as inv/<file>:<line> (inv/ = src/src_inv_iso_joint/).  The Fortran host binds them
 * through ISO_C_BINDING (host/dazim_mod.f90).
 *
 * Conventions
 *  - every function returns 0 on success, >0 for a reference-STOP-equivalent condition
 *    (DAZIM_E_*), <0 for a HIP/RCCL runtime failure; dazim_last_error(ctx) holds the message.
 *  - every bulk pointer may be a HOST pointer or a DEVICE (hipMalloc) pointer; the library
 *    detects which.  Host pointers are staged through device buffers (PCIe-inclusive); device
 *    pointers are used in place (zero-copy, what bench.py times).
 *  - 2-D grids keep the reference's Fortran memory order: node (iz,ix) of an nnz x nnx grid is at
 *    [(ix-1)*nnz + (iz-1)], i.e. "[ix][iz]" in C terms.  Model cubes are vel(nx,ny,nz) =
 *    [k][j][i] in C terms.  Indices that cross the ABI (period_idx, COO/CSR columns) are 1-based
 *    exactly where the reference's are, and say so.
 *  - no global state: everything hangs off dazim_ctx; one ctx per GPU / per host thread.
 */
#ifndef DAZIM_H
#define DAZIM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DAZIM_RMAX 129 /* refined source grid is at most (2*sgs*sgdl+1)^2, inv/CalSurfG.f90:1187-1196 */

enum {
  DAZIM_OK = 0,
  DAZIM_E_SOURCE_OUTSIDE = 1,   /* inv/CalSurfG.f90:287-293, 1174-1180 */
  DAZIM_E_RECEIVER_OUTSIDE = 2, /* inv/CalSurfG.f90:1649-1655, 1863-1869 */
  DAZIM_E_NNZ_OVERFLOW = 4,     /* inv/Main_Jt.f90:523 "increase sparsity fraction" */
  DAZIM_E_BAD_ARG = 5,
  DAZIM_E_ROOT_NOT_FOUND = 6    /* inv/surfdisp96.f:307-348 (warning there; count returned here) */
};

typedef struct dazim_ctx dazim_ctx;

/* propagation-grid geometry derived from the inversion grid, inv/CalSurfG.f90:1017-1038 */
typedef struct {
  int nvx, nvz;             /* B-spline vertices nx-2, ny-2                      */
  int nnx, nnz;             /* propagation nodes (nvx-1)*5+1, (nvz-1)*5+1         */
  float gox, goz, dnx, dnz; /* origin (colatitude, longitude; rad) and node step  */
  float dvx, dvz;           /* vertex step (rad)                                  */
} dazim_geom;

/* refined source box of one field, inv/CalSurfG.f90:1187-1206,1266-1271 */
typedef struct {
  int vnl, vnr, vnt, vnb; /* box bounds in coarse node indices, 1-based */
  int nnxr, nnzr;         /* refined grid size                           */
  int isx, isz;           /* coarse source cell                          */
  float goxr, gozr, dnxr, dnzr;
} dazim_refbox;

/* ---- context / memory ---------------------------------------------------------------------- */
int dazim_create(dazim_ctx **ctx, int device);
void dazim_destroy(dazim_ctx *ctx);
const char *dazim_last_error(const dazim_ctx *ctx);
int dazim_malloc(dazim_ctx *ctx, void **dptr, size_t bytes);
int dazim_free(dazim_ctx *ctx, void *dptr);
/* blocking copies on the context's stream.  After an asynchronous dazim_dispersion_kernels call (option disp.async) a copy that
 * touches the model or one of the three kernel tables waits for the perturbed copies on the auxiliary stream first; any other
 * copy leaves them running (dazim_get_stat "aux.pending") */
int dazim_memcpy_h2d(dazim_ctx *ctx, void *dst, const void *src, size_t bytes);
int dazim_memcpy_d2h(dazim_ctx *ctx, void *dst, const void *src, size_t bytes);
int dazim_sync(dazim_ctx *ctx);
/* 64-bit hash of every byte of a host array (host/dazim_mod.f90's aprod keys its cached device matrix on iw and rw with it,
 * replaces nothing in the reference: inv/aprod.f90:7 reads the arrays afresh on every call, which this makes observable) */
unsigned long long dazim_hash64(const void *data, size_t bytes);
void *dazim_stream(dazim_ctx *ctx); /* the hipStream_t every kernel of this ctx is launched on */
/* seconds spent in the last call's kernels, measured with HIP events on the ctx stream; name
 * selects the kernel ("fmm", "gridder", "disp", "ti", "rays", "spmv", "spmvt", "lsmr"); <0 if unknown */
double dazim_last_kernel_seconds(const dazim_ctx *ctx, const char *name);
/* the same table with a status: times of the last call's kernels AND the counts / choices the library reports ("fmm.wg_per_cu",
 * "spmv.kind", "lsmr.nranks", "fmm.field_pops" ...; docs/OPTIONS.md).  0 and *value, or DAZIM_E_BAD_ARG for an unknown name.      */
int dazim_get_stat(const dazim_ctx *ctx, const char *name, double *value);
/* tuning / test knobs, none of which changes a result unless its entry says so: the catalogue is docs/OPTIONS.md (heap forms and
 * time slicing of the eikonal kernel "fmm.*", two-stream dispersion "disp.*", ray cell lists "rays.*", product kernels "spmv.*",
 * "lsmr.*", CSR reservations "csr.*").  The environment variable DAZIM_OPTS=name=value,... sets options when a context is made. */
int dazim_set_option(dazim_ctx *ctx, const char *name, int value);

/* ---- geometry (host only; replaces the constant block inv/CalSurfG.f90:1005-1038) ---------- */
int dazim_geometry(int nx, int ny, float goxd, float gozd, float dvxd, float dvzd, dazim_geom *g);

/* ---- K1: dispersion + depth kernels -------------------------------------------------------------
 * = depthkernel (inv/CalSurfG.f90:1-139): per model column Brocher Vp(Vs), rho(Vp), knot -> layer
 *   refinement (refineGrid2LayerMdl, :2317), surfdisp96 (inv/surfdisp96.f:52; iflsph=1, Rayleigh,
 *   fundamental mode, phase velocity) for the column and its 6*nz perturbed copies, central
 *   differences.  With sen_* == NULL only pvRc is produced (= CalRayleighPhase,
 *   fwd/FwdTraveltimeCPS.f90:4).
 *  vel   [nz][ny][nx] fp32 (= Fortran vel(nx,ny,nz));  depz [nz] (host);  periods [kmax] (host)
 *  pvRc  [kmax][nx*ny] fp64 holding fp32-rounded values (cg(k)=sngl(c(k)), inv/surfdisp96.f:292)
 *  sen_vs, sen_vp, sen_rho [nz][kmax][nx*ny] fp64, nullable (all three or none)
 *  n_failed  number of (column, period) entries for which no root was found (pvRc = 0 there,
 *            inv/surfdisp96.f:342-348); the reference only prints a warning                    */
int dazim_dispersion_kernels(dazim_ctx *ctx, int nx, int ny, int nz, const float *vel,
                             const float *depz, float sublayers, int kmax, const double *periods,
                             double *pvRc, double *sen_vs, double *sen_vp, double *sen_rho,
                             int *n_failed);

/* ---- surfdisp96 itself, every argument -------------------------------------------------------------
 * = surfdisp96(thkm,vpm,vsm,rhom,nlayer,iflsph,iwave,mode,igr,kmax,t,cg) (inv/surfdisp96.f:52-354) for a batch of
 *   layered models: iflsph 0 flat / 1 spherical (sphere, :480); iwave 1 Love (dltar1, :704) / 2 Rayleigh (dltar4,
 *   :767, with the water-layer branch :844 when vs of the first layer is <= 0); mode 1 = fundamental, 2 = first
 *   higher, ... (:217); igr 0 phase velocity / > 0 group velocity from the roots at T/(1+h), T/(1-h), h = 0.005
 *   (:226-233, :276-304).  dazim_dispersion_kernels above is the tuned form of the one combination the reference's
 *   programs call (1, 2, 1, 0); this entry is the subroutine as it stands, one lane per model.
 *  nlayer [nmodel] layers of each model (<= nlayer_max <= 200 = NL); thk, vp, vs, rho [nmodel][nlayer_max] fp32
 *  (thickness of the last layer = half-space, ignored); periods [kmax <= 60 = NP] fp64 (host)
 *  cg [nmodel][kmax] fp64 holding fp32-rounded values, 0 from the first period without a root on (:342-348)
 *  n_failed  number of (model, period) entries left at 0, nullable                                        */
int dazim_surfdisp96(dazim_ctx *ctx, int nmodel, int nlayer_max, const int *nlayer, const float *thk, const float *vp,
                     const float *vs, const float *rho, int iflsph, int iwave, int mode, int igr, int kmax,
                     const double *periods, double *cg, int *n_failed);

/* ---- N1: TI eigenfunction partials -> azimuthal depth kernels ---------------------------------------
 * = depthkernelTI (inv/depthkernelTI.f90:2) calling tregn96 (inv/tregn96.f:52) once per column: Rayleigh fundamental-mode
 *   eigenfunctions of the flattened TI column (A=C=rho*Vp^2, L=N=rho*Vs^2, F=A-2L), partials dc/dah, dc/dbv, dc/dn with the
 *   causal-Q shift (Qp=150, Qs=50) and sprayl's sphericity factors, combined over the sub-layers of each inversion layer into
 *   Lsen_Gsc = dc/dA*A + dc/dL*L, the sensitivity of c to Gc/L, Gs/L.  Solid layers only (the reference's models are).
 *  vel, depz, sublayers, periods as for dazim_dispersion_kernels; pvRc [kmax][nx*ny] = its output (the reference recomputes
 *  the same surfdisp96 curve inside depthkernelTI); Lsen_Gsc out [nz-1][kmax][nx*ny] fp32 = Fortran Lsen_Gsc(nx*ny,kmax,nz-1),
 *  0 where pvRc is 0.                                                                                             */
int dazim_ti_kernels(dazim_ctx *ctx, int nx, int ny, int nz, const float *vel, const float *depz,
                     float sublayers, int kmax, const double *periods, const double *pvRc, float *Lsen_Gsc);

/* ---- K2+K3: batched eikonal fields -----------------------------------------------------------
 * = gridder (inv/CalSurfG.f90:1423) once per period + per (source,period): bsplrefine (:1525),
 *   travel on the refined box (:258, urg=1), injection + band completion (:1246-1308) and
 *   travel on the coarse grid (urg=2), i.e. the body of the source loop inv/CalSurfG.f90:1146-1314.
 *  pv         [kmax][(nvz+2)*(nvx+2)] phase velocity per period and inversion cell (= pvRc(:,k))
 *  period_idx [nfield] 1-based period of each field (= periods(srcnum,knumi))
 *  scx, scz   [nfield] source colatitude / longitude in radians (fp32, as the reference holds them)
 *  veln       [kmax][nnx][nnz]  out, nullable: gridded velocity per period
 *  ttn        [nfield][nnx][nnz] out: coarse traveltime field
 *  ttnr       [nfield][129][129] out, nullable: refined field (0 where the node is not alive/close)
 *  nstsr      [nfield][129][129] out, nullable: refined node status (-1 far, 0 alive, >0 close,
 *             -9 outside the nnzr x nnxr box)
 *  boxes      [nfield] out, nullable
 *  status     [nfield] out, nullable: DAZIM_OK or DAZIM_E_SOURCE_OUTSIDE per field; the call
 *             returns the first non-zero status (the reference STOPs there)                    */
/* ttn nullable (round 6): the coarse fields then stay inside the library, in the eikonal kernel's own layout, for the
 * dazim_rays_build_G* call that follows with ttn = NULL -- the reference's CalSurfG returns no field either (inv/CalSurfG.f90:909-912).
 * With option "fmm.async" = 1 on top (and every array device-resident) the call returns when its launch is enqueued; the ray call
 * that follows runs beside the launch's tail and collects this call's statuses and errors (docs/OPTIONS.md).                     */
int dazim_fmm_batch(dazim_ctx *ctx, int nx, int ny, float goxd, float gozd, float dvxd, float dvzd,
                    int kmax, const double *pv, int nfield, const float *scx, const float *scz,
                    const int *period_idx, float *veln, float *ttn, float *ttnr, int *nstsr,
                    dazim_refbox *boxes, int *status);

/* ---- K6/K7: sparse matrix + LSMR ----------------------------------------------------------------
 * The reference stores G as COO triplets rw / iw(2:nar+1) rows / iw(nar+2:2nar+1) cols
 * (inv/aprod.f90:20-24).  dazim_csr keeps the same matrix on the device twice: CSR for A*x and the
 * stable transpose (CSC) for A^T*y, fp32 values + int32 indices, int64 pointers.               */
typedef struct dazim_csr dazim_csr;

/* build from the reference's COO arrays (1-based irow/icol, rows need not be sorted).  nnz order
 * inside a row is kept, so sums run over the row in the order the reference appended it.        */
int dazim_csr_from_coo(dazim_ctx *ctx, int64_t m, int64_t n, int64_t nnz, const int *irow,
                       const int *icol, const float *rw, dazim_csr **A);
int dazim_csr_free(dazim_ctx *ctx, dazim_csr *A);
int dazim_csr_dims(const dazim_csr *A, int64_t *m, int64_t *n, int64_t *nnz);
/* multiply every stored entry of row i by w[i] (rw(i)=rw(i)*datweight(iw(1+i)), inv/Main_Jt.f90:467) */
int dazim_csr_scale_rows(dazim_ctx *ctx, dazim_csr *A, const float *w);

/* out[n] = column sums of |A|: the reference's DWS (norm(col(i))+=abs(rw(i)), inv/Main_Jt.f90:477-481) */
int dazim_csr_col_abs_sums(dazim_ctx *ctx, const dazim_csr *A, float *out);

/* device arrays in, ownership taken: rowptr[m+1] int64, col[nnz] 0-based int32, val[nnz] fp32   */
int dazim_csr_adopt(dazim_ctx *ctx, int64_t m, int64_t n, int64_t nnz, int64_t *rowptr, int *col,
                    float *val, dazim_csr **A);
/* append rows m+1..m+extra_m given as COO with absolute 1-based row ids: how the reference adds its
 * Tikhonov rows to the same triplet arrays (inv/TikhRegul.f90:2, inv/Main_Jt.f90:513-532)       */
int dazim_csr_append_coo(dazim_ctx *ctx, dazim_csr *A, int64_t extra_m, int64_t nnz, const int *irow,
                         const int *icol, const float *rw);
/* the matrix back as the reference's triplets (1-based, rows ascending); any pointer may be NULL  */
int dazim_csr_to_coo(dazim_ctx *ctx, const dazim_csr *A, int *irow, int *icol, float *rw);
/* Ray geometries (the reference's raypath_refmdl_<T>s.dat, written when `writepath` is set: fwd/rpathsAzim.f90:617-625,
 * fwd/FwdTraveltimeCPS.f90:673-691).  With option "rays.keep_paths" = 1 dazim_rays_build_G[_joint] keeps the points rgx/rgz(1:nrp)
 * of every ray -- receiver first, one point per half-cell step (after the clipping to the grid), source last -- on the device.
 * dazim_ray_paths_dims: rays and the capacity in points per ray of the last such call; dazim_ray_paths_copy: xz[nray][cap][2]
 * (colatitude, longitude in rad) and nrp[nray] (points of each ray; -1 if it needed more than cap) into host or device arrays. */
int dazim_ray_paths_dims(dazim_ctx *ctx, int64_t *nray, int *cap);
int dazim_ray_paths_copy(dazim_ctx *ctx, float *xz, int *nrp);
/* The reference keeps every ray row twice: the triplets rw/iw/col with |row| > ftol, which LSMR sees (inv/CalSurfG.f90:1358), and
 * the dense GVs (GGc, GGs) its residual diagnostics multiply with (inv/CalSigamNorm.f90:73): every entry of the cells with
 * |fdm| >= ftol, the dVs block formed with the Brocher derivatives coe_a / coe_rho left over from the LAST such cell of the ray
 * (the second loop, inv/CalSurfG.f90:1369-1378 / inv/CalSurfGAniso_Joint.f90:759-775, does not recompute them).  With option
 * "rays.dense_twin" = 1 dazim_rays_build_G[_joint] builds that second matrix as well (one more pass over the saved cell lists,
 * no second ray trace) and attaches it to G; this call hands it out (NULL if G has none).  The caller frees it. */
int dazim_csr_take_twin(dazim_ctx *ctx, dazim_csr *G, dazim_csr **twin);
/* B = the entries of A with |value| > tol (same shape, same order): the second threshold of the row assembly
 * (inv/CalSurfG.f90:1358) applied to a matrix that was built with option "rays.keep_small" = 1 (every non-zero entry of the
 * |fdm| >= ftol cells, the forward program's GGc/GGs, fwd/FwdTraveltimeCPS.f90:694-712) -- two streaming passes on the device.
 * reserve_rows / reserve_nnz: room for regularisation rows appended to B in place. */
int dazim_csr_threshold(dazim_ctx *ctx, const dazim_csr *A, float tol, int64_t reserve_rows, int64_t reserve_nnz,
                        dazim_csr **B);

/* ---- K4+K5: receiver times, ray tracing, Frechet weights, G rows ---------------------------------
 * = the receiver loop of CalSurfG (inv/CalSurfG.f90:1326-1364): srtimes (:1599), rpaths (:1735) and
 *   the row assembly with the double ftol=1e-4 threshold, for every ray of a batch whose eikonal
 *   fields (outputs of dazim_fmm_batch) are resident.  Row i of G belongs to ray i (the caller orders
 *   rays period -> source -> receiver like the reference's count1); columns are the reference's
 *   (k-1)*nvx*nvz+(jj-1)*nvx+kk, stored 0-based.
 *  vels [nz][ny][nx]; scx,scz,period_idx[nfield] as for dazim_fmm_batch; kernel_idx[nfield] 1-based
 *  period slot of sen_* (the reference's knumi; NULL = period_idx); veln,ttn,ttnr,nstsr,boxes: outputs
 *  of dazim_fmm_batch; field_of_ray[nray] 0-based; rcx,rcz[nray] receiver colatitude/longitude (rad);
 *  sen_vs,sen_vp,sen_rho [nz][kmax][nx*ny] from dazim_dispersion_kernels;
 *  tpred[nray] out = dsurf; G out; n_boundary out = rays clipped at the model edge (rbint)       */
int dazim_rays_build_G(dazim_ctx *ctx, int nx, int ny, int nz, float goxd, float gozd, float dvxd,
                       float dvzd, int kmax, const float *vels, int nfield, const float *scx,
                       const float *scz, const int *period_idx, const int *kernel_idx,
                       const float *veln, const float *ttn, const float *ttnr, const int *nstsr,
                       const dazim_refbox *boxes, int64_t nray, const int *field_of_ray,
                       const float *rcx, const float *rcz, const double *sen_vs, const double *sen_vp,
                       const double *sen_rho, float *tpred, dazim_csr **G, int64_t *nnz,
                       int *n_boundary);

/* joint Vsv + 2-psi mode: = the receiver loop of CalSurfGAnisoJoint (inv/CalSurfGAniso_Joint.f90:680-752):
 * rpathsAzim (inv/rpathsAzim.f90:16: the same ray plus per-step azimuth psi from azdist :687 and
 * cos/sin(2 psi)-weighted Frechet grids) and rows with three column blocks dVs | Gc | Gs, so
 * n = 3*(nx-2)*(ny-2)*(nz-1).  Lsen_Gsc [nz-1][kmax][nx*ny] fp32 (= Fortran Lsen_Gsc(nx*ny,kmax,nz-1))
 * are the TI depth kernels of depthkernelTI/tregn96 (inv/depthkernelTI.f90:2), an INPUT here
 * (SURVEY 8f N1).  Other arguments as dazim_rays_build_G.                                        */
int dazim_rays_build_G_joint(dazim_ctx *ctx, int nx, int ny, int nz, float goxd, float gozd, float dvxd,
                             float dvzd, int kmax, const float *vels, int nfield, const float *scx,
                             const float *scz, const int *period_idx, const int *kernel_idx,
                             const float *veln, const float *ttn, const float *ttnr, const int *nstsr,
                             const dazim_refbox *boxes, int64_t nray, const int *field_of_ray,
                             const float *rcx, const float *rcz, const double *sen_vs,
                             const double *sen_vp, const double *sen_rho, const float *Lsen_Gsc,
                             float *tpred, dazim_csr **G, int64_t *nnz, int *n_boundary);

/* ---- N4: what surrounds the solve in the outer iteration, on the device ------------------------------
 * = TikhonovRegularization / TikhRegul_joint (inv/TikhRegul.f90:2-104, :107-209): appends nblock * maxvp rows, maxvp =
 *   (nx-2)(ny-2)(nz-1); block b regularises columns b*maxvp+1 .. (b+1)*maxvp with weight w[b] (host array): a cell on a face
 *   of the block gets the single entry 2w, an inner cell the 7-point Laplacian 6w, -w x 6.  Rows in the reference's k, j, i order. */
int dazim_csr_append_tikhonov(dazim_ctx *ctx, dazim_csr *A, int nx, int ny, int nz, int nblock, const float *w);
/* = residual cbst = obst - dsyn, CalDdatSigma (inv/CalSigamNorm.f90:2-41), datweight = 1/sigmaT, cbst *= datweight and
 *   rw(i) = rw(i)*datweight(iw(1+i)) on the first dall rows of G (inv/Main_Jt.f90:432-469).  obst, dsyn [dall] in; res
 *   (the reference's Tdata), datweight, rhs [dall] out; host or device arrays; G nullable.  stats (host, 8 floats, nullable):
 *   mean, std, mean |.|, rms of res; meandeltaT, stddeltaT; mean datweight; mean |rhs|.  The two sums of CalDdatSigma run in
 *   the reference's sequential fp32 order, so the weights are the reference's bit for bit.                              */
int dazim_weight_data(dazim_ctx *ctx, dazim_csr *G, int64_t dall, const float *obst, const float *dsyn, float *res,
                      float *datweight, float *rhs, float *stats);
/* = the clamped model update (inv/Main_Jt.f90:582-620): dv [maxvp, or 3*maxvp when joint] in/out (dVs clamped to +-0.5, zeroed
 *   below 1e-5), vs [nz][ny][nx] in/out (inner cells += dVs, clamped to [minvel, maxvel]), gc, gs [nz-1][ny-2][nx-2] out
 *   (joint; nullable).  stats (host, nullable): [nblock][nz-1][3] = min, max, sum |.| of the update per block and depth.   */
int dazim_model_update(dazim_ctx *ctx, int nx, int ny, int nz, int joint, float *vs, float *dv, float minvel, float maxvel,
                       float *gc, float *gs, float *stats);

/* = aprod (inv/aprod.f90:7): mode 1: y(m) += A*x(n) ; mode 2: x(n) += A^T*y(m)                    */
int dazim_aprod(dazim_ctx *ctx, int mode, const dazim_csr *A, float *x, float *y);

/* ---- multi-GPU solve: rows of [G; L] sharded over ranks (one process per GPU), SURVEY 8e -----------------------
 * dazim_comm_unique_id: rank 0 makes the 128-byte RCCL id and hands it to the other ranks (any transport: MPI,
 * torch.distributed, a file); dazim_comm_init: every rank joins with the same id (ncclCommInitRank on the ctx's device).
 * While a communicator is attached, dazim_lsmr treats A and b as THIS rank's row shard of one global system: per iteration
 * ONE collective on the ctx stream -- an all-gather of the n fp32 of A_p^T u_p with the double ||u_p||^2 (each rank scales its
 * shard of u by its own norm first; the sums over the ranks are formed in rank order, beta follows) --; x, v, h, hbar and the
 * reorthogonalisation window are replicated, so every rank returns the same x.  dazim_comm_free detaches.             */
int dazim_comm_unique_id(void *id128);
int dazim_comm_init(dazim_ctx *ctx, int nranks, int rank, const void *id128);
int dazim_comm_free(dazim_ctx *ctx);
/* The same communicator over FILES in a directory every rank sees (each collective staged through the host): for tests -- the
 * whole multi-rank path with two or three processes on ONE GPU, which RCCL refuses -- not for production.                      */
int dazim_comm_init_files(dazim_ctx *ctx, int nranks, int rank, const char *dir);
/* recv[r*count .. (r+1)*count) = rank r's `count` values at send, on every rank; send, recv: host or device pointers (each on its
 * own); dtype as for dazim_comm_allreduce.  No communicator attached: recv = send.  Both transports move bytes only (RCCL:
 * ncclAllGather on the ctx stream); every SUM over the ranks in this library -- dazim_comm_allreduce, the row-sharded LSMR -- is an
 * all-gather followed by one kernel that adds the ranks' values in RANK ORDER, so that RCCL and the file transport, and every rank,
 * return the same bits (SURVEY 8e "fix reduction order"; the reference's sums are sequential, inv/aprod.f90:40-55).  Option
 * "comm.allreduce" = 1: ncclAllReduce instead (RCCL's own order). */
int dazim_comm_allgather(dazim_ctx *ctx, const void *send, void *recv, int64_t count, int dtype);
/* sum (op 0) / max (op 1) over the ranks of `count` values in place; buf: host or device pointer; dtype 0 fp32, 1 fp64, 2 int64.
 * No communicator attached: nothing happens.  What a sharded host program reduces its statistics and outputs with
 * (host/dazim_main.f90, DAZIM_NGPU; the reference has no counterpart: inv/Main_Jt.f90 is one process).                         */
int dazim_comm_allreduce(dazim_ctx *ctx, void *buf, int64_t count, int dtype, int op);
/* The two N4 calls for one rank's share of a row-sharded system (host/dazim_main.f90 with DAZIM_NGPU): the regularisation rows
 * [row_lo, row_hi) of the nblock*maxvp (inv/TikhRegul.f90:2-209), and the data weights of the data rows [row0, row0 + dall) of
 * dall_glob -- meandeltaT / stddeltaT (inv/CalSigamNorm.f90:20-31) are taken over ALL data in the reference's summation order on
 * every rank, the statistics returned are the whole data set's.                                                                */
int dazim_csr_append_tikhonov_rows(dazim_ctx *ctx, dazim_csr *A, int nx, int ny, int nz, int nblock, const float *w,
                                   int64_t row_lo, int64_t row_hi);
int dazim_weight_data_sharded(dazim_ctx *ctx, dazim_csr *G, int64_t dall, int64_t row0, int64_t dall_glob, const float *obst,
                              const float *dsyn, float *res, float *wgt, float *rhs, float *stats);

/* The model's tables with the model's rows sharded over the ranks: the reference's one parallel loop (OpenMP over the columns jj,
 * inv/CalSurfG.f90:39-43 called at :1078; depthkernelTI inv/depthkernelTI.f90:2-112).  Same arguments and results as
 * dazim_dispersion_kernels / dazim_ti_kernels.  With a communicator attached, this rank computes the model rows [lo, hi) of the even
 * contiguous split of ny (columns are numbered jj*nx+ii: a block of rows is a block of columns) with the same kernels, and
 * all-gathers join the blocks: every rank returns the complete tables, bit-identical to the single-rank call.  pvRc is joined
 * before the call returns; with option "disp.async" and device-resident sen_* the perturbed copies of this rank's block run on the
 * auxiliary stream as in the single-rank call and the gather of the three depth-kernel tables follows them when that stream is
 * joined (dazim_rays_build_G*, dazim_sync, a copy that touches the tables).  Every rank must make the same calls in the same order.
 * Without a communicator: the plain calls.                                                                                     */
int dazim_dispersion_kernels_sharded(dazim_ctx *ctx, int nx, int ny, int nz, const float *vel, const float *depz, float sublayers,
                                     int kmax, const double *periods, double *pvRc, double *sen_vs, double *sen_vp,
                                     double *sen_rho, int *n_failed);
int dazim_ti_kernels_sharded(dazim_ctx *ctx, int nx, int ny, int nz, const float *vel, const float *depz, float sublayers,
                             int kmax, const double *periods, const double *pvRc, float *Lsen_Gsc);

/* = LSMR (inv/lsmrModule.f90:36), fp32 like the reference; b[m] in, x[n] out.                     */
int dazim_lsmr(dazim_ctx *ctx, const dazim_csr *A, const float *b, float damp, float atol,
               float btol, float conlim, int itnlim, int localSize, float *x, int *istop, int *itn,
               float *normA, float *condA, float *normr, float *normAr, float *normx);

/* One line of the reference's iteration log (unit nout, inv/lsmrModule.f90:667-682, format 1500:
 * itn, x(1), normr, normAr, test1, test2, normA, condA) plus test3 and rtol, which together with ctol/atol decide
 * whether the reference prints the line (:653-661).  Record 0 is the line printed before the loop (:468-471).  */
typedef struct {
  int itn;
  float x1, normr, normAr, test1, test2, test3, rtol, normA, condA;
} dazim_lsmr_rec;
/* dazim_lsmr that also returns the iteration log: trace[0..*trace_n) (HOST array of trace_cap records; iterations
 * beyond trace_cap-1 are not recorded).  The Fortran drop-in LSMR prints it to `nout` in the reference's formats.  */
int dazim_lsmr_traced(dazim_ctx *ctx, const dazim_csr *A, const float *b, float damp, float atol,
                      float btol, float conlim, int itnlim, int localSize, float *x, int *istop, int *itn,
                      float *normA, float *condA, float *normr, float *normAr, float *normx,
                      dazim_lsmr_rec *trace, int trace_cap, int *trace_n);

#ifdef __cplusplus
}
#endif
#endif
