// rays.hip -- K4 + K5 of SURVEY.md: receiver traveltimes (srtimes, inv/CalSurfG.f90:1599), ray
// back-tracing with B-spline Frechet weights (rpaths, :1735) and G-row assembly (:1339-1364) for a
// batch of rays whose eikonal fields are already resident in HBM (output of dazim_fmm_batch).
//
// Eight rays per wavefront, one per 8-lane group (the kernel is bound by instruction issue and by the dependent
// loads of a step, so SIMT across groups divides the instruction count per ray and multiplies the loads in flight;
// measured: 16 lanes per ray 0.094 s, 8 lanes 0.073 s, 4 lanes 0.084 s on the S-256 batch).  The half-cell stepping
// is inherently serial and is executed redundantly by the lanes of a group, while the 4x4 B-spline scatter of every
// sub-segment is spread over them: each lane keeps two cells of the current 4x4 block of the Frechet grid(s) in
// registers and the block is written back to the ray's grid in HBM scratch only when the ray leaves it.  Touched cells are collected in an
// LDS cell list.  Rows are emitted straight into CSR in the reference's column order (depth-major, then jj,
// kk) by a count pass, an exclusive scan and an emit pass that reuses the saved cell lists -- no atomics, so G
// is reproducible.  fp32 without FMA like the reference.
#include <cmath>
#include <chrono>
#include <unistd.h>

#include "dazim_internal.h"

#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_radix_sort.hpp>

namespace {

constexpr int GDX = 5, GDZ = 5;
constexpr int RM = DAZIM_RMAX;
constexpr float EARTH = 6371.0f;
constexpr float FTOL = 1e-4f;  // inv/CalSurfG.f90:999

struct RayArgs {
  dazim_geom g;
  int nx, ny, nz, kmax;
  long nray;
  const int *field;     // [nray] field of each ray
  const float *rcx, *rcz;
  const float *scx, *scz;  // [nfield]
  const int *period;       // [nfield], 1-based (velocity map of the field)
  const int *kidx;         // [nfield], 1-based period slot of the depth kernels (knumi), may equal period
  const float *veln;       // [kmax][nnx][nnz]
  const float *ttn;        // [nfield][nnx][nnz]; TILED kernels: the fields as dazim_fmm_batch(ttn = NULL) left them, node (x0, z0) of field f
                           // at ttn[(tslot ? tslot[f] : f) * fstride + dz_tile_x(x0, tsh) + dz_tile_z(z0)] (4 x 4 tiles: a ray that
                           // moves in x stays in one 64-byte tile for four nodes instead of touching a new 128-byte line per node)
  const int *tslot;
  int tsh;
  long fstride;
  // beside an asynchronous eikonal launch (option fmm.async): defer_mark[quad] != 0 = not traced yet; a pass traces the marked quads
  // whose fields are finished (fdone[f] = 1; 0 still marching, 2 band overflow: the host reruns the field first) and unmarks them
  const int *fdone;
  int *defer_mark;
  int max_quads;           // > 0: a workgroup leaves after tracing that many quads
  int sweeps;              // > 1: the pass offers every quad of a range that many times (see the queue of rays_kernel)
  const float *ttnr;       // [nfield][RM][RM]
  const int *nstsr;        // [nfield][RM][RM]
  const dazim_refbox *boxes;
  const float *vels;       // [nz][ny][nx]
  const double *svs, *svp, *srho;  // [nz][kmax][nx*ny]
  unsigned nvx_magic;      // ceil(2^32 / nvx): cell id / nvx = __umulhi(id, nvx_magic) for every id < 65 536 (checked on the host)
  const double *skern;     // [nz][kmax][nx*ny] svp*coe_a + srho*coe_rho + svs per model cell and period (k_row_kernels), or nullptr
  const float *lsen;       // joint mode: Lsen_Gsc [nz-1][kmax][nx*ny] (fp32, inv/CalSurfGAniso_Joint.f90:337)
  // double-precision reciprocals RN(1/d) of the loop-invariant fp32 divisors (grid spacings and 2*EARTH*spacing): see divr()
  double r_dnx, r_dnz, r_dnxr, r_dnzr, r_dvx, r_dvz, r_e2dnx, r_e2dnxr;
  float dplh;              // min cell size (before the 0.5 factor), host libm
  float *dsurf;            // [nray]
  int *status;             // [nray]
  int *rbflag;             // [nray]
  long *count;             // [nray]  (count pass out)
  int *nlist;              // [nray]  cells with |fdm| >= ftol saved by the count pass (-1: did not fit, retrace)
  unsigned short *lcell;   // [nray][LK] their (jj,kk) cell ids, ascending
  float *lval;             // [nray][LK] (x3 in joint mode) their fdm (, fdmc, fdms) values
  float *fdm_scratch;      // [nwg*4][(nvx+2)*(nvz+2)] (x3 in joint mode): one Frechet grid slot per 16-lane group
  int LK;
  int lcap;                // LDS cell-list capacity per ray
  const unsigned *perm;    // [nray] order in which the rays are dealt to the lane groups: by field, then by source-receiver
                           // distance, so that the rays marching in lockstep in one wavefront have similar lengths (speed only:
                           // everything a ray produces is stored under its own index)
  unsigned *qcount;        // [16] task counters of the two passes (8 ranges each, one per XCD); [16] rays whose cell list outgrew
                           // the LDS capacity (full-grid sweep), [17] rays whose list outgrew LK (traced again by the emit pass);
                           // [18] quads a pass beside an asynchronous eikonal launch left for a later pass
  float2 *pts;             // option rays.keep_paths: [nray][pcap] ray-path points (colatitude, longitude in rad) as the reference's
  int *npts;               //   rgx/rgz(1:nrp) (receiver first, source last; fwd/rpathsAzim.f90:221-380); npts = nrp, or -1 if > pcap
  int pcap;
  int dense;               // the reference's dense copies GVs/GGc/GGs as a second matrix ("twin", option rays.dense_twin):
                           // 0 off; 2 (count pass): also count the twin's entries into countd; 1 (a second emit pass): write the
                           // twin -- every non-zero entry of the |fdm| >= ftol cells, the dVs block with the Brocher derivatives
                           // coe_a / coe_rho of the LAST such cell of the ray, which is what the reference's second loop uses
                           // (inv/CalSurfG.f90:1369-1378, inv/CalSurfGAniso_Joint.f90:759-775 do not recompute them)
  long *countd;            // [nray] twin entries per row (count pass out when dense = 2)
  int keep_small;          // 1: keep every non-zero row entry of the |fdm| >= ftol cells (the forward program's dense GGc/GGs,
                           // fwd/FwdTraveltimeCPS.f90:694-712); 0: the inversion's second |row| > ftol threshold
  const long *rowptr;      // [nray+1] (emit pass in)
  float *val;
  int *col;
};

// sin of a colatitude x in (0, pi): sin x = cos(x - pi/2), an even Taylor polynomial to y^18 in fp64
// (remainder < 3e-14 for |y| <= pi/2), rounded to fp32 -- the correctly rounded fp32 sine up to ties
// nobody will hit, at a tenth of the cost of the general fp64 sin.  Outside (0, pi) fall back to it.
__device__ __forceinline__ float dz_sinf(float x) {
  if (!(x > 0.0f && x < 3.1415927f)) return (float)sin((double)x);
  const double y = (double)x - 1.5707963267948966192;
  const double y2 = y * y;
  double p = -1.5619206968586226e-16;            // -1/18!
  p = fma(p, y2, 4.7794773323873853e-14);        //  1/16!
  p = fma(p, y2, -1.1470745597729725e-11);       // -1/14!
  p = fma(p, y2, 2.0876756987868099e-09);        //  1/12!
  p = fma(p, y2, -2.7557319223985891e-07);       // -1/10!
  p = fma(p, y2, 2.4801587301587302e-05);        //  1/8!
  p = fma(p, y2, -1.3888888888888889e-03);       // -1/6!
  p = fma(p, y2, 4.1666666666666664e-02);        //  1/4!
  p = fma(p, y2, -0.5);
  p = fma(p, y2, 1.0);
  return (float)p;
}

// azimuth of the step (x0,z0) -> (x1,z1): azdist (inv/rpathsAzim.f90:687-793) with the reference's
// implicit typing, called as at inv/rpathsAzim.f90:415-423; returns cos(2 psi), sin(2 psi)
__device__ __forceinline__ void step_azimuth(float x0, float z0, float x1, float z1, float &c2psi, float &s2psi) {
  const float PI_F = 3.1415926535898f;
  const float evtlat = (PI_F / 2 - x0) * 180.0f / PI_F, evtlon = z0 * 180.0f / PI_F;
  const float stalat = (PI_F / 2 - x1) * 180.0f / PI_F, stalon = z1 * 180.0f / PI_F;
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
  const float rgpsi = az / 180 * PI_F;
  c2psi = (float)cos((double)(2.0f * rgpsi));
  s2psi = (float)sin((double)(2.0f * rgpsi));
}

// x / d, correctly rounded, for a divisor whose double-precision reciprocal rd = RN(1/d) is at hand: three instructions
// (convert, multiply, convert) instead of the ~11 of the IEEE fp32 division sequence -- a fifth to a third of this kernel was
// divisions by grid spacings.  Exact for EVERY x: the double product is within 2^-52 (relative) of x/d, and the quotient of two
// 24-bit numbers X/D is either a 24-bit number itself or at least 1/(D*M) >= 2^-49 (relative) away from every rounding
// boundary M of the 24-bit format (|X*2^k - D*M| is a non-zero integer; a quotient cannot sit exactly on a boundary, because
// X = D'*M with an odd 25-bit M would need more than 24 bits), so rounding the double to fp32 rounds where the exact quotient
// rounds.  Infinities, NaNs and zeros come out as the division's.
__device__ __forceinline__ float divr(float x, double rd) { return (float)((double)x * rd); }

__device__ __forceinline__ void basis(float v, float b[4]) {  // inv/CalSurfG.f90:2145-2148
  const float om = 1.0f - v;
  b[0] = om * om * om / 6.0f;
  b[1] = (4.0f - 6.0f * (v * v) + 3.0f * (v * v * v)) / 6.0f;
  b[2] = (1.0f + 3.0f * v + 3.0f * (v * v) - 3.0f * (v * v * v)) / 6.0f;
  b[3] = v * v * v / 6.0f;
}

// x / 6.0f through divr (the reciprocal is a compile-time constant)
__device__ __forceinline__ float div6(float x) { return divr(x, 1.0 / 6.0); }
// element i (0..3) of the cubic B-spline basis at v, inv/CalSurfG.f90:2145-2148: only the selected numerator is divided
// (the kernel is bound by instruction issue; the four divisions of the plain form were a fifth of a step)
__device__ __forceinline__ float basis1(float v, int i) {
  const float om = 1.0f - v;
  const float n0 = om * om * om;
  const float n1 = 4.0f - 6.0f * (v * v) + 3.0f * (v * v * v);
  const float n2 = 1.0f + 3.0f * v + 3.0f * (v * v) - 3.0f * (v * v * v);
  const float n3 = v * v * v;
  return div6(i == 0 ? n0 : (i == 1 ? n1 : (i == 2 ? n2 : n3)));
}

// bilinear velocity inside coarse cell (ipx,ipz), inv/CalSurfG.f90:2129-2137
__device__ __forceinline__ float vel_at(const dazim_geom &g, const float *veln, int ipx, int ipz, float drx, float drz,
                                        double rdnx, double rdnz) {
  float vel = 0.0f;
#pragma unroll
  for (int l = 1; l <= 2; l++)
#pragma unroll
    for (int m = 1; m <= 2; m++) {
      float produ = (1.0f - fabsf(divr((float)(m - 1) * g.dnz - drz, rdnz)));
      produ = produ * (1.0f - fabsf(divr((float)(l - 1) * g.dnx - drx, rdnx)));
      if (ipz - 1 + m <= g.nnz && ipx - 1 + l <= g.nnx && ipz - 1 + m >= 1 && ipx - 1 + l >= 1)
        vel = vel + veln[(size_t)(ipx - 2 + l) * g.nnz + (ipz - 2 + m)] * produ;
    }
  return vel;
}

// bilinear(nv,dsx,dsz), inv/CalSurfG.f90:2293 -- velocity at a point of cell (cx,cz)
__device__ __forceinline__ float bilin_cell(const dazim_geom &g, const float *veln, int cx, int cz, float px, float pz,
                                            double rdnx, double rdnz) {
  const float drx = (px - g.gox) - (float)(cx - 1) * g.dnx;
  const float drz = (pz - g.goz) - (float)(cz - 1) * g.dnz;
  float biv = 0.0f;
#pragma unroll
  for (int i = 1; i <= 2; i++)
#pragma unroll
    for (int j = 1; j <= 2; j++) {
      const float produ = (1.0f - fabsf(divr((float)(i - 1) * g.dnx - drx, rdnx))) *
                          (1.0f - fabsf(divr((float)(j - 1) * g.dnz - drz, rdnz)));
      biv = biv + veln[(size_t)(cx - 2 + i) * g.nnz + (cz - 2 + j)] * produ;
    }
  return biv;
}

// The factor of a dVs row entry that does not depend on the ray (inv/CalSurfG.f90:1339-1364): the Brocher derivatives of the cell's
// velocity and the three depth kernels, (svp*coe_a + srho*coe_rho + svs) in the reference's order and precision -- once per model
// cell, layer and period instead of once per ray that crosses the cell (both passes of rays_kernel; the emit pass is little else).
__global__ void k_row_kernels(long n, int kmax, long ncol, const float *__restrict__ vels, const double *__restrict__ svs,
                              const double *__restrict__ svp, const double *__restrict__ srho, double *__restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // (layer, period, column)
  if (i >= n) return;
  const long k = i / (kmax * ncol), c = i % ncol;
  const float v = vels[k * ncol + c];
  const float coe_a = (2.0947f - 0.8206f * 2 * v + 0.2683f * 3 * (v * v) - 0.0251f * 4 * (v * v * v));
  const float vpft = 0.9409f + 2.0947f * v - 0.8206f * (v * v) + 0.2683f * (v * v * v) - 0.0251f * (v * v * v * v);
  const float coe_rho = coe_a * (1.6612f - 0.4721f * 2 * vpft + 0.0671f * 3 * (vpft * vpft) -
                                 0.0043f * 4 * (vpft * vpft * vpft) + 0.000106f * 5 * (vpft * vpft * vpft * vpft));
  out[i] = svp[i] * (double)coe_a + srho[i] * (double)coe_rho + svs[i];
}

// sort key of a ray for the order in which rays are dealt to the wavefronts: field in the high bits, quantised source-receiver
// distance (a proxy for the number of steps of the ray) in the low `dbits`
__global__ void k_ray_keys(long nray, const int *__restrict__ field, const float *__restrict__ scx, const float *__restrict__ scz,
                           const float *__restrict__ rcx, const float *__restrict__ rcz, float inv_dmax, int dbits,
                           unsigned *__restrict__ keys, unsigned *__restrict__ iota, const int *__restrict__ order) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nray) return;
  const int f = field[i];
  const float dx = scx[f] - rcx[i], dz = (scz[f] - rcz[i]) * __sinf(rcx[i]);
  const float d = sqrtf(dx * dx + dz * dz) * inv_dmax;
  const unsigned dmaxq = (1u << dbits) - 1u;
  unsigned q = d >= 1.0f ? dmaxq : (unsigned)(d * (float)dmaxq);
  // (order: the fields' places in the eikonal launch's queue -- beside an asynchronous launch the rays come in the order their fields finish)
  keys[i] = ((unsigned)(order ? order[f] : f) << dbits) | (dmaxq - q);          // longest rays of a field first
  iota[i] = (unsigned)i;
}

// lanes per ray: 8 in the count pass, which traces every ray (measured: 16 lanes 94 ms, 8 lanes 73 ms, 4 lanes 84 ms on the S-256
// batch); 16 in the emit pass, which only walks the saved cell lists (lane-parallel work: 8.2 ms with 16 lanes, 12.5 with 8)
#ifndef DZ_GP_COUNT
#define DZ_GP_COUNT 8   // (experiment switch, tools/exp_rays_ab.sh: 4 and 16 lanes per ray are both 17 % slower at S-256)
#endif
constexpr int GP_COUNT = DZ_GP_COUNT, GP_EMIT = 16;
constexpr int RPW_MAX = 64 / GP_COUNT;   // rays per wavefront (most of the two passes: sizes the scratch slots)
__device__ __forceinline__ void cbar() { asm volatile("" ::: "memory"); }  // compiler-only barrier (same-wave ops are in order)

// One GP-lane group per ray (~330 k instructions per ray when one wavefront traced one ray, all of it per-ray
// scalar work except the 4x4 B-spline scatter).  The Frechet grid(s) of a ray live in an HBM scratch slot; the
// 4x4 block currently being updated is cached in registers, LPR cells per lane, and written back when the ray
// moves to another B-spline cell (every ~10 steps), so every cell still sees its contributions in the
// reference's order (fdm = r1 + fdm).
template <bool EMIT, bool AZIM, bool TILED>
#ifndef DZ_RAYS_MINW
#define DZ_RAYS_MINW 4
#endif
__global__ __launch_bounds__(64, AZIM ? 2 : DZ_RAYS_MINW) void rays_kernel(RayArgs A_) {
  // (arguments through an opaque pointer to the kernarg segment, like fmm_kernel: taken by value, the ~50 scalar registers of
  // RayArgs stay alive across the stepping loop and the spill code moved 93 of its 1 227 VALU instructions per step through
  // VGPR lanes -- v_readlane / v_writelane --, 17 more through scratch memory)
#ifndef DZ_RAYS_ARGS_BYVALUE
  using ArgP = const __attribute__((address_space(4))) RayArgs *;
  auto launder = [](ArgP p) {
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return (ArgP)((const __attribute__((address_space(4))) char *)p + __builtin_amdgcn_readfirstlane(z));
  };
  ArgP Ap = launder((ArgP)__builtin_amdgcn_kernarg_segment_ptr());
#define A (*Ap)
#define RAYS_ARGS_FRESH Ap = launder(Ap)
#else
  const RayArgs &A = A_;
#define RAYS_ARGS_FRESH
#endif
  constexpr int GP = EMIT ? GP_EMIT : GP_COUNT;   // lanes per ray
  constexpr int RPW = 64 / GP;                    // rays per wavefront
  constexpr int LPR = 16 / GP;                    // cells of the 4x4 B-spline block per lane
  constexpr int LSTEP = 4 / LPR;                  // ... which are LSTEP rows apart
  constexpr unsigned GMASK = (1u << GP) - 1u;
  extern __shared__ __attribute__((aligned(16))) unsigned short s_lists[];  // [RPW][lcap] cell lists
  dazim_geom g;
  g.nvx = A.g.nvx; g.nvz = A.g.nvz; g.nnx = A.g.nnx; g.nnz = A.g.nnz;
  g.gox = A.g.gox; g.goz = A.g.goz; g.dnx = A.g.dnx; g.dnz = A.g.dnz; g.dvx = A.g.dvx; g.dvz = A.g.dvz;
  const int lane = threadIdx.x, grp = lane / GP, gl = lane & (GP - 1);
  const int lm = gl & 3, l0 = gl >> 2;  // this lane's cells of the 4x4 scatter: (m, l) = (lm, l0 + q*LSTEP), q < LPR
  const int nnx = g.nnx, nnz = g.nnz, nvx = g.nvx, nvz = g.nvz, ldf = nvz + 2, nf = ldf * (nvx + 2);
  constexpr int NG = AZIM ? 3 : 1;
  const unsigned nvx_magic = A.nvx_magic;   // (cell ids are decoded in the innermost loops of the row assembly: two instructions instead of the ~20 of an integer division)
  const int LC = A.lcap;   // list capacity (<= 1024 so that 12 wavefronts fit a CU); longer lists fall back to a full-grid sweep
  unsigned short *s_list = s_lists + (size_t)grp * LC;
  float *gfdm = A.fdm_scratch + ((size_t)blockIdx.x * RPW + grp) * nf * NG;
  float *gfdmc = gfdm + nf, *gfdms = gfdm + 2 * nf;
  const float gox = g.gox, goz = g.goz, dnx = g.dnx, dnz = g.dnz, dvx = g.dvx, dvz = g.dvz;
  const double rdnx = A.r_dnx, rdnz = A.r_dnz, rdnxr = A.r_dnxr, rdnzr = A.r_dnzr, rdvx = A.r_dvx, rdvz = A.r_dvz;
  const unsigned gmask_shift = grp * GP;
  // XCD-aware order (speed only): workgroup b runs on XCD b % 8, and the rays of one field read the
  // same traveltime grids, so each XCD gets one contiguous eighth of the ray quads.  Within its eighth a workgroup takes
  // the next quad from a counter (rays differ in length by an order of magnitude: equal shares of quads are not equal
  // shares of work); a workgroup whose eighth is drained helps with the next ones.
  const long nquad = (A.nray + RPW - 1) / RPW;
  const int nxcd = (gridDim.x % 8 == 0) ? 8 : 1;
  unsigned *qcnt = A.qcount + (EMIT ? 8 : 0);
  int chunk = (int)(blockIdx.x % nxcd);
  int traced = 0;
  for (;;) {
    RAYS_ARGS_FRESH;
    long quad = -1;
    if (lane == 0) {
      for (int tried = 0; tried < nxcd; tried++) {
        const long lo = nquad * chunk / nxcd, hi = nquad * (chunk + 1) / nxcd;
        const long b = lo < hi ? (long)atomicAdd(&qcnt[chunk], 1u) : hi;
        if (lo + b < hi) {
          quad = lo + b;
          break;
        }
        // (a pass beside the eikonal launch sweeps its range several times: a quad whose field was not finished when a workgroup
        // came by is offered again to the workgroups that arrive later -- no waiting, only A.sweeps cheap looks per quad)
        if (!EMIT && A.sweeps > 1 && lo < hi && b < (hi - lo) * A.sweeps) {
          quad = lo + b % (hi - lo);
          break;
        }
        chunk = (chunk + 1) % nxcd;
      }
    }
    quad = ((long)__shfl((int)(quad >> 32), 0) << 32) | (unsigned)__shfl((int)quad, 0);
    chunk = __shfl(chunk, 0);
    if (quad < 0) break;
    const long slot = quad * RPW + grp;
    if (!EMIT && A.defer_mark) {
      // Beside an asynchronous eikonal launch the count pass is a sequence of short NON-BLOCKING passes over the quads that are
      // still marked: a quad is traced (and unmarked) if the fields of all its rays are finished, left for a later pass otherwise,
      // and a workgroup leaves after max_quads of them.  A ray workgroup NEVER waits for the other kernel: waiting wavefronts were
      // seen to hold the whole chip while the eikonal launch stood still (4 071 of 4 096 ray workgroups resident and waiting, 19
      // eikonal workgroups gone, for as long as the wait was allowed to last: profiles/r6_tail_fill.md) -- which launch's
      // wavefronts are resident is the hardware scheduler's business, and nothing here may depend on it.
      if (!A.defer_mark[quad]) continue;                      // traced by an earlier pass
      if (A.fdone) {
        // (ACQUIRE: the load invalidates this XCD's cached copy of the flag -- the XCDs' L2s are not coherent with each other)
        const int fq = slot < A.nray ? A.field[A.perm ? (long)A.perm[slot] : slot] : -1;
        const int st = fq >= 0 ? __hip_atomic_load(A.fdone + fq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : 1;
        if (__ballot(st != 1) != 0) continue;                 // a field still marching (0) or waiting for its spill rerun (2)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");    // the fields' words and refined outputs, written by another kernel on other XCDs
      }
      if (A.max_quads > 0 && traced >= A.max_quads) break;    // this pass's share is done (the quad stays marked)
      // claim the quad: a pass offers it several times, perhaps to workgroups on different XCDs, whose L2s do not see each other's
      // plain stores -- the exchange is a device-scope atomic, exactly one workgroup gets the 1
      int mine = 0;
      if (lane == 0) mine = __hip_atomic_exchange(A.defer_mark + quad, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      mine = __shfl(mine, 0);
      if (!mine) continue;
      traced++;
      if (lane == 0) atomicAdd(&A.qcount[18], 1u);
    }
    if (slot >= A.nray) continue;
    const long ray = A.perm ? (long)A.perm[slot] : slot;
    const int f = A.field[ray];
    const float scx = A.scx[f], scz = A.scz[f], rcx = A.rcx[ray], rcz = A.rcz[ray];
    const float *veln = A.veln + (size_t)(A.period[f] - 1) * nnx * nnz;
    const float *ttn = TILED ? A.ttn + (size_t)(A.tslot ? A.tslot[f] : f) * A.fstride : A.ttn + (size_t)f * nnx * nnz;
    const int tsh = TILED ? A.tsh : 0;
    // node (x0, z0), 0-based, of this field's coarse times
    auto tnode = [&](int x0, int z0) -> float {
      return TILED ? ttn[dz_tile_x(x0, tsh) + dz_tile_z(z0)] : ttn[(size_t)x0 * nnz + z0];
    };
    const float *ttnr = A.ttnr + (size_t)f * RM * RM;
    const int *nstsr = A.nstsr + (size_t)f * RM * RM;
    const dazim_refbox bx = A.boxes[f];
    const int saved = EMIT ? A.nlist[ray] : -1;   // EMIT: reuse the count pass's Frechet cells when they fit
    if (saved < 0)
      for (int i = gl; i < nf * NG; i += GP) gfdm[i] = 0.0f;
    cbar();
    int status = 0, rb = 0;
    // ---------------- srtimes, inv/CalSurfG.f90:1644-1711 ----------------
    if (!EMIT) {
      int irx = (int)divr(rcx - gox, rdnx) + 1, irz = (int)divr(rcz - goz, rdnz) + 1;
      if (irx < 1 || irx > nnx || irz < 1 || irz > nnz) status = DAZIM_E_RECEIVER_OUTSIDE;
      if (!status) {
        if (irx == nnx) irx--;
        if (irz == nnz) irz--;
        const int isx = (int)divr(scx - gox, rdnx) + 1, isz = (int)divr(scz - goz, rdnz) + 1;
        float sred = ((scx - rcx) * EARTH) * ((scx - rcx) * EARTH);
        const float e2 = (scz - rcz) * EARTH * dz_sinf(rcx);
        sred = sqrtf(sred + e2 * e2);
        bool sw = sred < A.dplh;
        if (isx == irx && isz == irz) sw = true;
        float trr;
        if (sw) {
          const float vs = bilin_cell(g, veln, isx, isz, scx, scz, rdnx, rdnz);
          const float vr = bilin_cell(g, veln, irx, irz, rcx, rcz, rdnx, rdnz);
          trr = 2.0f * sred / (vs + vr);
        } else {
          const float drx = (rcx - gox) - (float)(irx - 1) * dnx;
          const float drz = (rcz - goz) - (float)(irz - 1) * dnz;
          trr = 0.0f;
#pragma unroll
          for (int k = 1; k <= 2; k++)
#pragma unroll
            for (int l = 1; l <= 2; l++) {
              const float produ = (1.0f - fabsf(divr((float)(l - 1) * dnz - drz, rdnz))) *
                                  (1.0f - fabsf(divr((float)(k - 1) * dnx - drx, rdnx)));
              trr = trr + tnode(irx - 2 + k, irz - 2 + l) * produ;
            }
        }
        if (gl == 0) A.dsurf[ray] = trr;
      }
    }
    // ---------------- rpaths, inv/CalSurfG.f90:1818-2236 ----------------
    const float goxr = bx.goxr, gozr = bx.gozr, dnxr = bx.dnxr, dnzr = bx.dnzr;
    const int nnxr = bx.nnxr, nnzr = bx.nnzr;
    const int isx = (int)divr(scx - goxr, rdnxr) + 1, isz = (int)divr(scz - gozr, rdnzr) + 1;
    const float dpl = 0.5f * A.dplh;
    int ipx = (int)divr(rcx - gox, rdnx) + 1, ipz = (int)divr(rcz - goz, rdnz) + 1;
    if (ipx < 1 || ipx >= nnx || ipz < 1 || ipz >= nnz) status = DAZIM_E_RECEIVER_OUTSIDE;
    if (!status && saved < 0) {
      float x0 = rcx, z0 = rcz;
      float sinx0 = dz_sinf(x0);
      int sw = 0;
      float sred = ((scx - x0) * EARTH) * ((scx - x0) * EARTH);
      float e2 = (scz - z0) * EARTH * sinx0;
      sred = sqrtf(sred + e2 * e2);
      if (sred < 2.0f * dpl) sw = 1;
      int ipxr = (int)divr(rcx - goxr, rdnxr) + 1, ipzr = (int)divr(rcz - gozr, rdnzr) + 1;
      auto in_refined = [&](int px, int pz) -> int {
        if (px < 1 || px >= nnxr || pz < 1 || pz >= nnzr) return 0;
        const int *sp = nstsr + (size_t)(px - 1) * RM + (pz - 1);
        if (sp[0] != 0 || sp[1] != 0) return 0;
        if (sp[RM] != 0 || sp[RM + 1] != 0) return 0;
        return 1;
      };
      int igref = in_refined(ipxr, ipzr);
      if (sw == 0 && igref == 1 && ipxr == isx && ipzr == isz) sw = 1;
      // ray geometry for the path files (writepath): point 1 = the receiver; a ray that ends at once gets the source as point 2
      const bool keep_pts = !EMIT && A.pts != nullptr && gl == 0;
      float2 *rp = A.pts ? A.pts + (size_t)ray * A.pcap : nullptr;
      int np = 1;
      if (keep_pts) {
        rp[0] = make_float2(rcx, rcz);
        if (sw == 1) { if (A.pcap > 1) rp[1] = make_float2(scx, scz); np = 2; }
      }
      // register-cached 4x4 block of the Frechet grid(s): this lane's cell is (bz+ll, bx+lm)
      int cbx = -100, cbz = -100;
      float acc[LPR], accc[LPR], accs[LPR];
#pragma unroll
      for (int q = 0; q < LPR; q++) acc[q] = accc[q] = accs[q] = 0.0f;
      auto flush = [&]() {
        if (cbx > -100) {
#pragma unroll
          for (int q = 0; q < LPR; q++) {
            const int fi = (cbx + lm) * ldf + (cbz + l0 + q * LSTEP);
            gfdm[fi] = acc[q];
            if (AZIM) { gfdmc[fi] = accc[q]; gfdms[fi] = accs[q]; }
          }
        }
      };
      // The corner times of the cell the NEXT step starts in are requested as soon as that cell is known (on both grids: which
      // one applies depends on node states that are still being loaded), so that they travel together with the velocity and
      // node-state loads of the current step instead of costing a second dependent round trip per step.
      float tc00, tc01, tc10, tc11, tr00 = 0.0f, tr01 = 0.0f, tr10 = 0.0f, tr11 = 0.0f;
      auto load_corner_times = [&]() {
        if (TILED) {
          const int x0 = dz_tile_x(ipx - 1, tsh), x1 = dz_tile_x(ipx, tsh), z0 = dz_tile_z(ipz - 1), z1 = dz_tile_z(ipz);
          tc00 = ttn[x0 + z0]; tc01 = ttn[x0 + z1]; tc10 = ttn[x1 + z0]; tc11 = ttn[x1 + z1];
        } else {
          const float *t = ttn + (size_t)(ipx - 1) * nnz + (ipz - 1);
          tc00 = t[0]; tc01 = t[1]; tc10 = t[nnz]; tc11 = t[nnz + 1];
        }
        if (ipxr >= 1 && ipxr < nnxr && ipzr >= 1 && ipzr < nnzr) {   // (outside the refined box igref is 0 and these are not used)
          const float *u = ttnr + (size_t)(ipxr - 1) * RM + (ipzr - 1);
          tr00 = u[0]; tr01 = u[1]; tr10 = u[RM]; tr11 = u[RM + 1];
        }
      };
      load_corner_times();
      // What a step needs at its starting point -- velocity and B-spline weights at (x0, z0) -- is what the step before computed at
      // its end point, bit for bit, whenever that end point was the step's (x1, z1) itself in the same cells: the last sub-segment
      // ends at x0 + 1.0f * (x1 - x0), which is x1 exactly unless the two differ by more than a factor of two (Sterbenz), and the
      // cell indices agree unless the point was clipped to the grid.  Checked per ray after every step; the values are carried
      // over only while every ray of the wavefront may (a sixth of the step's instructions otherwise repeated).
      bool carry = false;
      float vel_c = 0.0f, vi_c = 0.0f, wi_c[LPR];
#pragma unroll
      for (int q = 0; q < LPR; q++) wi_c[q] = 0.0f;
      const long maxrp = (long)nnx * nnz;
      for (long j = 1; j <= maxrp; j++) {
        if (sw == 1) break;
        float dtx, dtz;
        if (igref == 1) {
          dtx = tr10 - tr00;
          dtx = dtx + tr11 - tr01;
          dtx = divr(dtx, A.r_e2dnxr);
          dtz = tr01 - tr00;
          dtz = dtz + tr11 - tr10;
          dtz = dtz / (2.0f * EARTH * sinx0 * dnzr);
        } else {
          dtx = tc10 - tc00;
          dtx = dtx + tc11 - tc01;
          dtx = divr(dtx, A.r_e2dnx);
          dtz = tc01 - tc00;
          dtz = dtz + tc11 - tc10;
          dtz = dtz / (2.0f * EARTH * sinx0 * dnz);
        }
        const float rd1 = sqrtf(dtx * dtx + dtz * dtz);
        float x1 = x0 - dpl * dtx / (EARTH * rd1);
        float z1 = z0 - dpl * dtz / (EARTH * sinx0 * rd1);
        const int ipxo = ipx, ipzo = ipz;
        ipxr = (int)divr(x1 - goxr, rdnxr) + 1;
        ipzr = (int)divr(z1 - gozr, rdnzr) + 1;
        igref = in_refined(ipxr, ipzr);
        ipx = (int)divr(x1 - gox, rdnx) + 1;
        ipz = (int)divr(z1 - goz, rdnz) + 1;
        float sinx1 = dz_sinf(x1);
        sred = ((scx - x1) * EARTH) * ((scx - x1) * EARTH);
        e2 = (scz - z1) * EARTH * sinx1;
        sred = sqrtf(sred + e2 * e2);
        sw = 0;
        if (sred < 2.0f * dpl) sw = 1;
        if (sw == 0 && igref == 1 && ipxr == isx && ipzr == isz) sw = 1;
        bool clipx = false;
        if (ipx < 1) { x1 = gox; ipx = 1; rb = 1; clipx = true; }
        if (ipx >= nnx) { x1 = gox + (float)(nnx - 1) * dnx; ipx = nnx - 1; rb = 1; clipx = true; }
        if (ipz < 1) { z1 = goz; ipz = 1; rb = 1; }
        if (ipz >= nnz) { z1 = goz + (float)(nnz - 1) * dnz; ipz = nnz - 1; rb = 1; }
        load_corner_times();
        if (keep_pts) {   // rgx(j+1) after the clipping, then rgx(j+2) = the source if this was the last step (:352-400)
          if (np < A.pcap) rp[np] = make_float2(x1, z1);
          np++;
          if (sw == 1) {
            if (np < A.pcap) rp[np] = make_float2(scx, scz);
            np++;
          }
        }
        if (clipx) sinx1 = dz_sinf(x1);   // the next step starts from the clipped point
        float c2psi = 0.0f, s2psi = 0.0f;
        if (AZIM) step_azimuth(x0, z0, x1, z1, c2psi, s2psi);
        // ---- Frechet weights, :2077-2229 ----
        const int ivx = (ipx - 1) / GDX + 1, ivz = (ipz - 1) / GDZ + 1;
        const int ivxo = (ipxo - 1) / GDX + 1, ivzo = (ipzo - 1) / GDZ + 1;
        int nhp = 0, chp0 = 0, chp1 = 0;
        float vr0 = 1.0f, vr1 = 1.0f, vr2 = 1.0f;
        if (ivx != ivxo) {
          nhp = 1;
          const float xi = (ivx > ivxo) ? gox + (float)(ivx - 1) * dvx : gox + (float)ivx * dvx;
          vr0 = (xi - x0) / (x1 - x0);
          chp0 = 1;
        }
        if (ivz != ivzo) {
          const float zi = (ivz > ivzo) ? goz + (float)(ivz - 1) * dvz : goz + (float)ivz * dvz;
          const float r = (zi - z0) / (z1 - z0);
          if (nhp == 0) {
            vr0 = r;
            chp0 = 2;
          } else if (r >= vr0) {
            vr1 = r;
            chp1 = 2;
          } else {
            vr1 = vr0;
            chp1 = chp0;
            vr0 = r;
            chp0 = 2;
          }
          nhp++;
        }
        nhp++;  // the closing sub-segment with vrat = 1, chp = 0
        if (nhp == 1) vr0 = 1.0f;
        if (nhp == 2) vr1 = 1.0f;
        float drx, drz;
        float vel = vel_c, vi = vi_c, wi[LPR];            // this lane's vi(m), wi(l)
#pragma unroll
        for (int q = 0; q < LPR; q++) wi[q] = wi_c[q];
#ifdef DZ_RAYS_NOCARRY   // experiment: every step computes its starting point
        if (true) {
#else
        if (__builtin_expect(__ballot(!carry) != 0, 0)) {  // (first step, clipped points: rare -- and wave-uniform, so out of line)
#endif
          drx = (x0 - gox) - (float)(ipxo - 1) * dnx;
          drz = (z0 - goz) - (float)(ipzo - 1) * dnz;
          vel = vel_at(g, veln, ipxo, ipzo, drx, drz, rdnx, rdnz);
          drx = (x0 - gox) - (float)(ivxo - 1) * dvx;
          drz = (z0 - goz) - (float)(ivzo - 1) * dvz;
          vi = basis1(divr(drx, rdvx), lm);
#pragma unroll
          for (int q = 0; q < LPR; q++) wi[q] = basis1(divr(drz, rdvz), l0 + q * LSTEP);
        }
        int ivxt = ivxo, ivzt = ivzo;
        bool endsame = false;
        for (int k = 1; k <= nhp; k++) {
          const float velo = vel, vio = vi;
          float wio[LPR];
#pragma unroll
          for (int q = 0; q < LPR; q++) wio[q] = wi[q];
          if (k > 1) {
            const int cp = (k == 2) ? chp0 : chp1;
            if (cp == 1) ivxt = ivx;
            else if (cp == 2) ivzt = ivz;
          }
          const float vrk = (k == 1) ? vr0 : (k == 2 ? vr1 : vr2);
          const float vrp = (k == 2) ? vr0 : vr1;
          const float rigz = z0 + vrk * (z1 - z0);
          const float rigx = x0 + vrk * (x1 - x0);
          const int ipxt = (int)divr(rigx - gox, rdnx) + 1, ipzt = (int)divr(rigz - goz, rdnz) + 1;
          endsame = rigx == x1 && rigz == z1 && ipxt == ipx && ipzt == ipz && ivxt == ivx && ivzt == ivz;   // (of the last sub-segment)
          drx = (rigx - gox) - (float)(ipxt - 1) * dnx;
          drz = (rigz - goz) - (float)(ipzt - 1) * dnz;
          vel = vel_at(g, veln, ipxt, ipzt, drx, drz, rdnx, rdnz);
          drx = (rigx - gox) - (float)(ivxt - 1) * dvx;
          drz = (rigz - goz) - (float)(ivzt - 1) * dvz;
          vi = basis1(divr(drx, rdvx), lm);
#pragma unroll
          for (int q = 0; q < LPR; q++) wi[q] = basis1(divr(drz, rdvz), l0 + q * LSTEP);
          const float dinc = (k == 1) ? vrk * dpl : (vrk - vrp) * dpl;
          // block of this sub-segment: cells (ivzt-2+l, ivxt-2+m), l,m = 1..4
          const int nbx = ivxt - 1, nbz = ivzt - 1;
          if (nbx != cbx || nbz != cbz) {
            flush();
            cbar();
            cbx = nbx;
            cbz = nbz;
#pragma unroll
            for (int q = 0; q < LPR; q++) {
              const int fi = (cbx + lm) * ldf + (cbz + l0 + q * LSTEP);
              acc[q] = gfdm[fi];
              if (AZIM) { accc[q] = gfdmc[fi]; accs[q] = gfdms[fi]; }
            }
          }
#pragma unroll
          for (int q = 0; q < LPR; q++) {
            const float rdc1 = vi * wi[q] / (vel * vel);
            const float rdc2 = vio * wio[q] / (velo * velo);
            float r1 = -(rdc1 + rdc2) * dinc / 2.0f;
            acc[q] = r1 + acc[q];
            if (AZIM) {   // inv/rpathsAzim.f90:580-586
              r1 = -(rdc1 * c2psi + rdc2 * c2psi) * dinc / 2.0f;
              accc[q] = r1 + accc[q];
              r1 = -(rdc1 * s2psi + rdc2 * s2psi) * dinc / 2.0f;
              accs[q] = r1 + accs[q];
            }
          }
        }
        carry = endsame;
        vel_c = vel;
        vi_c = vi;
#pragma unroll
        for (int q = 0; q < LPR; q++) wi_c[q] = wi[q];
        x0 = x1;
        z0 = z1;
        sinx0 = sinx1;
      }
      flush();
      if (keep_pts) A.npts[ray] = np <= A.pcap ? np : -1;
    }
    cbar();
    if (!EMIT && gl == 0) {
      if (A.pts && status) A.npts[ray] = 0;
      A.status[ray] = status;
      A.rbflag[ray] = rb;
    }
    // ---------------- G row, inv/CalSurfG.f90:1339-1364 ----------------
    // cells with |fdm| >= ftol in (jj,kk) order -> this group's LDS list
    int nlist = 0;
    if (saved >= 0) {
      nlist = saved;
      const size_t o = (size_t)ray * A.LK, ov = o * NG;
      for (int i = gl; i < nlist; i += GP) {
        const int c = A.lcell[o + i];
        const int jj = (int)__umulhi((unsigned)c, nvx_magic) + 1, kk = c - (jj - 1) * nvx + 1;
        s_list[i] = (unsigned short)c;
        gfdm[kk * ldf + jj] = A.lval[ov + i];
        if (AZIM) {
          gfdmc[kk * ldf + jj] = A.lval[ov + A.LK + i];
          gfdms[kk * ldf + jj] = A.lval[ov + 2 * A.LK + i];
        }
      }
    } else if (!status) {
      for (int base = 0; base < nvz * nvx; base += GP) {
        const int c = base + gl;
        bool keep = false;
        if (c < nvz * nvx) {
          const int jj = (int)__umulhi((unsigned)c, nvx_magic) + 1, kk = c - (jj - 1) * nvx + 1;
          keep = fabsf(gfdm[kk * ldf + jj]) >= FTOL;
        }
        const unsigned m = (unsigned)((__ballot(keep) >> gmask_shift) & GMASK);
        const int pos = nlist + __popc(m & ((1u << gl) - 1u));
        if (keep && pos < LC) s_list[pos] = (unsigned short)c;
        nlist += __popc(m);
      }
    }
    cbar();
    if (!EMIT) {   // hand the Frechet cells to the emit pass so that it need not trace the ray again
      const size_t o = (size_t)ray * A.LK, ov = o * NG;
      if (nlist <= A.LK)
        for (int i = gl; i < nlist; i += GP) {
          const int c = s_list[i];
          const int jj = (int)__umulhi((unsigned)c, nvx_magic) + 1, kk = c - (jj - 1) * nvx + 1;
          A.lcell[o + i] = (unsigned short)c;
          A.lval[ov + i] = gfdm[kk * ldf + jj];
          if (AZIM) {
            A.lval[ov + A.LK + i] = gfdmc[kk * ldf + jj];
            A.lval[ov + 2 * A.LK + i] = gfdms[kk * ldf + jj];
          }
        }
      if (gl == 0) A.nlist[ray] = (status || nlist <= A.LK) ? (status ? 0 : nlist) : -1;
    }
    const size_t ncol = (size_t)A.nx * A.ny;
    const int kslot = A.kidx[f] - 1;
    long cnt = 0;
    const long rstart = EMIT ? A.rowptr[ray] : 0;
    const int nparpi = nvx * nvz * (A.nz - 1);
    const bool lovf = nlist > LC;                       // list did not fit: sweep the whole grid instead (rare)
    const int ntot = lovf ? nvz * nvx : nlist;
    long cntd = 0;
    int jjL = 1, kkL = 1;                               // dense twin: the last cell with |fdm| >= ftol in (jj, kk) order
    if (A.dense) {
      int clast = -1;
      if (!lovf) {
        if (nlist > 0) clast = s_list[nlist - 1];
      } else {
        for (int base = ((nvz * nvx - 1) / GP) * GP; base >= 0 && clast < 0; base -= GP) {
          const int c = base + gl;
          bool k2 = false;
          if (c < nvz * nvx) {
            const int jj = (int)__umulhi((unsigned)c, nvx_magic) + 1, kk = c - (jj - 1) * nvx + 1;
            k2 = fabsf(gfdm[kk * ldf + jj]) >= FTOL;
          }
          const unsigned m2 = (unsigned)((__ballot(k2) >> gmask_shift) & GMASK);
          if (m2) clast = base + (31 - __clz(m2));
        }
      }
      if (clast >= 0) { jjL = (int)__umulhi((unsigned)clast, nvx_magic) + 1; kkL = clast - (jjL - 1) * nvx + 1; }
    }
    // The usual ray (its cell list fits RC cells per lane, the combined kernels are there, no dense twin): every cell's indices and
    // Frechet values are decoded and loaded ONCE into registers and the layers loop over them, where the general loop below decodes
    // the cell id and reloads its Frechet value for each of the nz - 1 layers.  Same entries in the same order.
    constexpr int RC = 8;   // (128 cells in the emit pass, where a ray of the S-256 batch has ~100; 64 in the count pass: sixteen per lane there cost the tracing loop registers, 38.1 -> 40.1 ms)
#ifdef DZ_RAYS_NOROWCACHE
    const bool rowcache = false;
#else
    const bool rowcache = !lovf && ntot <= RC * GP && A.skern != nullptr && !A.dense;
#endif
    if (rowcache) {
      int sidx[RC], cbase[RC], fidx[RC];
#pragma unroll
      for (int i = 0; i < RC; i++) {
        const int li = i * GP + gl;
        const int c = li < ntot ? (int)s_list[li] : 0;
        const int jj = (int)__umulhi((unsigned)c, nvx_magic) + 1, kk = c - (jj - 1) * nvx + 1;
        sidx[i] = jj * (nvx + 2) + kk;
        cbase[i] = (jj - 1) * nvx + kk;
        fidx[i] = kk * ldf + jj;
      }
      for (int blk = 0; blk < NG; blk++) {
        float fdv[RC];
        const float *fsrc = blk == 0 ? gfdm : (blk == 1 ? gfdmc : gfdms);
#pragma unroll
        for (int i = 0; i < RC; i++) fdv[i] = (i * GP + gl < ntot) ? fsrc[fidx[i]] : 0.0f;
        for (int k = 1; k <= A.nz - 1; k++) {
          const size_t sk = ((size_t)(k - 1) * A.kmax + kslot) * ncol;
          const int nk = blk * nparpi + (k - 1) * nvz * nvx;
#pragma unroll
          for (int i = 0; i < RC; i++) {
            if (i * GP >= ntot) break;                    // (per ray: the lanes of a group leave together)
            const bool cell = i * GP + gl < ntot;
            float rowv = 0.0f;
            bool keep = false;
            if (cell) {
              if (blk == 0) rowv = (float)(A.skern[sk + sidx[i]] * (double)fdv[i]);
              else rowv = A.lsen[sk + sidx[i]] * fdv[i];
              keep = A.keep_small ? (rowv != 0.0f) : (fabsf(rowv) > FTOL);
            }
            const unsigned m = (unsigned)((__ballot(keep) >> gmask_shift) & GMASK);
            if (EMIT && keep) {
              const long pos = rstart + cnt + __popc(m & ((1u << gl) - 1u));
              A.val[pos] = rowv;
              A.col[pos] = nk + cbase[i] - 1;
            }
            cnt += __popc(m);
          }
        }
      }
    } else
    for (int blk = 0; blk < NG; blk++)   // dVs | Gc | Gs column blocks (inv/CalSurfGAniso_Joint.f90:728-738)
      for (int k = 1; k <= A.nz - 1; k++) {
        for (int base = 0; base < ntot; base += GP) {
          const int li = base + gl;
          bool keep = false, keepd = false;
          float rowv = 0.0f;
          int nn = 0;
          bool cell = li < ntot;
          int c = 0;
          if (cell) c = lovf ? li : s_list[li];
          const int jj = (int)__umulhi((unsigned)c, nvx_magic) + 1, kk = c - (jj - 1) * nvx + 1;
          if (cell && lovf) cell = fabsf(gfdm[kk * ldf + jj]) >= FTOL;
          if (cell) {
            const size_t si = ((size_t)(k - 1) * A.kmax + kslot) * ncol + (size_t)jj * (nvx + 2) + kk;
            if (blk == 0) {
              const float fd = gfdm[kk * ldf + jj];
              double r;
              if (A.skern) {
                r = A.skern[si] * (double)fd;           // (the cell's factor from k_row_kernels: same operations, same order)
              } else {
                const float v = A.vels[((size_t)(k - 1) * A.ny + jj) * A.nx + kk];
                const float coe_a = (2.0947f - 0.8206f * 2 * v + 0.2683f * 3 * (v * v) - 0.0251f * 4 * (v * v * v));
                const float vpft = 0.9409f + 2.0947f * v - 0.8206f * (v * v) + 0.2683f * (v * v * v) - 0.0251f * (v * v * v * v);
                const float coe_rho = coe_a * (1.6612f - 0.4721f * 2 * vpft + 0.0671f * 3 * (vpft * vpft) -
                                               0.0043f * 4 * (vpft * vpft * vpft) + 0.000106f * 5 * (vpft * vpft * vpft * vpft));
                r = (A.svp[si] * (double)coe_a + A.srho[si] * (double)coe_rho + A.svs[si]) * (double)fd;
              }
              rowv = (float)r;
              if (A.dense) {   // the same expression with the derivatives left over from the last cell of the first loop
                const float vL = A.vels[((size_t)(k - 1) * A.ny + jjL) * A.nx + kkL];
                const float caL = (2.0947f - 0.8206f * 2 * vL + 0.2683f * 3 * (vL * vL) - 0.0251f * 4 * (vL * vL * vL));
                const float vpL = 0.9409f + 2.0947f * vL - 0.8206f * (vL * vL) + 0.2683f * (vL * vL * vL) - 0.0251f * (vL * vL * vL * vL);
                const float crL = caL * (1.6612f - 0.4721f * 2 * vpL + 0.0671f * 3 * (vpL * vpL) -
                                         0.0043f * 4 * (vpL * vpL * vpL) + 0.000106f * 5 * (vpL * vpL * vpL * vpL));
                const double rd = (A.svp[si] * (double)caL + A.srho[si] * (double)crL + A.svs[si]) * (double)fd;
                keepd = (float)rd != 0.0f;
                if (EMIT && A.dense == 1) rowv = (float)rd;
              }
            } else {
              rowv = A.lsen[si] * (blk == 1 ? gfdmc : gfdms)[kk * ldf + jj];
              keepd = rowv != 0.0f;
            }
            keep = A.keep_small ? (rowv != 0.0f) : (fabsf(rowv) > FTOL);
            if (EMIT && A.dense == 1) keep = keepd;
            nn = blk * nparpi + (k - 1) * nvz * nvx + (jj - 1) * nvx + kk;  // 1-based column of the reference
          }
          if (!EMIT && A.dense == 2) cntd += __popc((unsigned)((__ballot(keepd) >> gmask_shift) & GMASK));
          const unsigned m = (unsigned)((__ballot(keep) >> gmask_shift) & GMASK);
          if (EMIT && keep) {
            const long pos = rstart + cnt + __popc(m & ((1u << gl) - 1u));
            A.val[pos] = rowv;
            A.col[pos] = nn - 1;
          }
          cnt += __popc(m);
        }
      }
    if (!EMIT && gl == 0) {
      A.count[ray] = cnt;
      if (A.dense == 2) A.countd[ray] = cntd;
      if (lovf) atomicAdd(&A.qcount[16], 1u);           // (statistics only: dazim_last_kernel_seconds("rays.list_sweeps"))
      if (!status && nlist > A.LK) atomicAdd(&A.qcount[17], 1u);
    }
  }
}
#undef A
#undef RAYS_ARGS_FRESH

}  // namespace

struct dazim_csr;  // sparse.hip
extern "C" int dazim_csr_adopt(dazim_ctx *ctx, int64_t m, int64_t n, int64_t nnz, int64_t *rowptr, int *col,
                               float *val, dazim_csr **out);

static int rays_build_impl(dazim_ctx *ctx, int nx, int ny, int nz, float goxd, float gozd, float dvxd,
                                  float dvzd, int kmax, const float *vels_u, int nfield, const float *scx_u,
                                  const float *scz_u, const int *period_u, const int *kidx_u, const float *veln_u,
                                  const float *ttn_u, const float *ttnr_u, const int *nstsr_u,
                                  const dazim_refbox *boxes_u, int64_t nray, const int *field_u, const float *rcx_u,
                                  const float *rcz_u, const double *svs_u, const double *svp_u, const double *srho_u,
                                  const float *lsen_u, float *dsurf_u, dazim_csr **G, int64_t *nnz_out, int *n_boundary) {
  if (!ctx || !G) return DAZIM_E_BAD_ARG;
  const bool joint = lsen_u != nullptr;
  dazim_geom g;
  if (dazim_geometry(nx, ny, goxd, gozd, dvxd, dvzd, &g)) return dz_fail(ctx, DAZIM_E_BAD_ARG, "bad grid");
  if (nray < 0 || nfield < 1 || nz < 2 || kmax < 1 || (size_t)g.nvx * g.nvz > 65535u) return dz_fail(ctx, DAZIM_E_BAD_ARG, "bad arguments to dazim_rays_build_G");
  // every array the kernels dereference must be there: dazim_fmm_batch can be called without the refined outputs (ttnr, nstsr,
  // boxes nullable there), but the ray tracer reads them next to the source (inv/CalSurfG.f90:1941-1952)
  // ttn == NULL: the coarse fields are the ones the last dazim_fmm_batch call (made with ttn == NULL) kept inside the library
  const bool tiled = ttn_u == nullptr;
  if (tiled && (!ctx->fields.tiled || ctx->fields.nfield != nfield || ctx->fields.nnx != g.nnx || ctx->fields.nnz != g.nnz))
    return dz_fail(ctx, DAZIM_E_BAD_ARG, "dazim_rays_build_G: ttn is NULL and the last dazim_fmm_batch call did not keep %d fields of this grid inside the library (call it with ttn = NULL)", nfield);
  if (!vels_u || !scx_u || !scz_u || !period_u || !veln_u || !ttnr_u || !nstsr_u || !boxes_u || !svs_u || !svp_u ||
      !srho_u || !dsurf_u || (nray > 0 && (!field_u || !rcx_u || !rcz_u)))
    return dz_fail(ctx, DAZIM_E_BAD_ARG, "dazim_rays_build_G: NULL array (the refined fields ttnr, nstsr and boxes of dazim_fmm_batch are required)");
  DZ_HIP(hipSetDevice(ctx->device));
  // An asynchronous eikonal call (option fmm.async) is still marching: the count pass goes to the context's third stream, where its
  // workgroups are dispatched as the eikonal launch's persistent workgroups leave and every quad of rays waits for its fields'
  // completion flags -- the ray kernel fills the tail of the eikonal launch.  Only if nothing here has to wait on the host for the
  // main stream (every array device-resident); otherwise the eikonal call is completed first.  (A pending gather of sharded
  // dispersion tables -- dz_join_aux below -- runs on the third stream too, behind the perturbed copies it follows.)
  bool overlap = (bool)ctx->fmm_finish && tiled && ctx->fields.fdone && nray > 0;
  // (a pending gather of sharded dispersion tables would run on the third stream too -- dz_join_aux below --: fine with the file
  // transport, whose collectives are host-staged; an RCCL communicator is kept to ONE stream, the main one, so with RCCL the
  // eikonal call is completed first.  Option comm.gather_stream3 = 1 lifts that.)
  if (overlap && ctx->aux_epilogue && ctx->comm && ((DzComm *)ctx->comm)->nccl &&
      !(ctx->opts.count("comm.gather_stream3") && ctx->opts["comm.gather_stream3"]))
    overlap = false;
  if (overlap)
    for (const void *q : {(const void *)vels_u, (const void *)scx_u, (const void *)scz_u, (const void *)period_u, (const void *)veln_u,
                          (const void *)ttnr_u, (const void *)nstsr_u, (const void *)boxes_u, (const void *)field_u, (const void *)rcx_u,
                          (const void *)rcz_u, (const void *)svs_u, (const void *)svp_u, (const void *)srho_u, (const void *)dsurf_u,
                          (const void *)kidx_u, (const void *)lsen_u})
      if (q && !dz_is_device_ptr(q)) overlap = false;
  if (!overlap) {
    const int rcf = dz_fmm_finish(ctx);
    if (rcf) return rcf;
  }
  struct StreamSwap {   // while the count pass is prepared and launched, "the context's stream" is the third stream
    dazim_ctx *c; hipStream_t main; bool on = false;
    void restore() { if (on) { c->stream = main; on = false; } }
    ~StreamSwap() {
      if (on) (void)hipStreamSynchronize(c->stream);   // (an early return: nothing of this call may still run when its buffers go)
      restore();
    }
  } swap{ctx, ctx->stream};
  if (overlap) {
    DZ_HIP(hipStreamWaitEvent(ctx->stream3, ctx->ev_pre, 0));   // the velocity grids and the cleared flags, enqueued before the launch
    ctx->stream = ctx->stream3;
    swap.on = true;
  }
  {   // the depth kernels may still be in the making on the auxiliary stream (dazim_dispersion_kernels with disp.async)
    const int rcj = dz_join_aux(ctx);
    if (rcj) return rcj;
  }
  const size_t nn = (size_t)g.nnx * g.nnz, nr = (size_t)RM * RM, ncol = (size_t)nx * ny;
  DzBuf<float> vels, scx, scz, veln, ttn, ttnr, rcx, rcz, dsurf;
  DzBuf<int> period, kidx, nstsr, field;
  DzBuf<dazim_refbox> boxes;
  DzBuf<double> svs, svp, srho;
  int rc;
  if ((rc = vels.init(ctx, vels_u, (size_t)nz * ncol, true, false))) return rc;
  if ((rc = scx.init(ctx, scx_u, nfield, true, false))) return rc;
  if ((rc = scz.init(ctx, scz_u, nfield, true, false))) return rc;
  if ((rc = period.init(ctx, period_u, nfield, true, false))) return rc;
  if ((rc = kidx.init(ctx, kidx_u ? kidx_u : period_u, nfield, true, false))) return rc;
  if ((rc = veln.init(ctx, veln_u, nn * kmax, true, false))) return rc;
  if (!tiled && (rc = ttn.init(ctx, ttn_u, nn * nfield, true, false))) return rc;
  if ((rc = ttnr.init(ctx, ttnr_u, nr * nfield, true, false))) return rc;
  if ((rc = nstsr.init(ctx, nstsr_u, nr * nfield, true, false))) return rc;
  if ((rc = boxes.init(ctx, boxes_u, nfield, true, false))) return rc;
  if ((rc = field.init(ctx, field_u, nray, true, false))) return rc;
  if ((rc = rcx.init(ctx, rcx_u, nray, true, false))) return rc;
  if ((rc = rcz.init(ctx, rcz_u, nray, true, false))) return rc;
  const size_t nk = (size_t)nz * kmax * ncol;
  if ((rc = svs.init(ctx, svs_u, nk, true, false))) return rc;
  if ((rc = svp.init(ctx, svp_u, nk, true, false))) return rc;
  if ((rc = srho.init(ctx, srho_u, nk, true, false))) return rc;
  if ((rc = dsurf.init(ctx, dsurf_u, nray, false, true))) return rc;
  DzBuf<float> lsen;
  if (joint && (rc = lsen.init(ctx, lsen_u, (size_t)(nz - 1) * kmax * ncol, true, false))) return rc;
  // index conventions follow the reference: periods and kernel slots 1-based (periods(srcnum,knumi), knumi), field_of_ray 0-based
  if ((rc = dz_check_range(ctx, period.dev, nfield, 1, kmax, "period_idx"))) return rc;
  if ((rc = dz_check_range(ctx, kidx.dev, nfield, 1, kmax, "kernel_idx"))) return rc;
  if ((rc = dz_check_range(ctx, field.dev, nray, 0, nfield - 1, "field_of_ray"))) return rc;

  RayArgs A;
  A.g = g;
  A.nx = nx; A.ny = ny; A.nz = nz; A.kmax = kmax;
  A.nray = nray;
  A.field = field.dev; A.rcx = rcx.dev; A.rcz = rcz.dev; A.scx = scx.dev; A.scz = scz.dev;
  A.period = period.dev; A.kidx = kidx.dev; A.veln = veln.dev; A.ttn = ttn.dev; A.ttnr = ttnr.dev;
  A.tslot = nullptr; A.tsh = 0; A.fstride = (long)nn;
  if (tiled) {
    A.ttn = reinterpret_cast<const float *>(ctx->fields.tiled);   // (the node word of a finished node is its time)
    A.tslot = ctx->fields.tslot; A.tsh = ctx->fields.tsh; A.fstride = ctx->fields.stride;
  }
  ctx->ksec["rays.tiled_fields"] = tiled ? 1.0 : 0.0;
  ctx->ksec["rays.overlap"] = overlap ? 1.0 : 0.0;
  A.fdone = overlap ? ctx->fields.fdone : nullptr;
  A.defer_mark = nullptr;
  A.max_quads = 0;
  A.sweeps = 1;
  A.nstsr = nstsr.dev; A.boxes = boxes.dev; A.vels = vels.dev; A.svs = svs.dev; A.svp = svp.dev; A.srho = srho.dev;
  A.lsen = joint ? lsen.dev : nullptr;
  A.skern = nullptr;
  {
    const unsigned d = (unsigned)g.nvx;
    A.nvx_magic = (unsigned)((0x100000000ull + d - 1) / d);
    for (unsigned c = 0; c < 65536u; c++)   // (cell ids are 16-bit: the identity is checked for all of them, once per call)
      if ((unsigned)(((unsigned long long)c * A.nvx_magic) >> 32) != c / d) return dz_fail(ctx, DAZIM_E_BAD_ARG, "internal: reciprocal of nvx");
  }
  if (!(ctx->opts.count("rays.skern") && !ctx->opts["rays.skern"])) {   // (option rays.skern = 0: every entry from the three kernels)
    void *pk;
    if ((rc = dz_scratch(ctx, "rays.skern", nk * sizeof(double), &pk))) return rc;
    hipLaunchKernelGGL(k_row_kernels, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, ctx->stream, (long)nk, kmax, (long)ncol,
                       vels.dev, svs.dev, svp.dev, srho.dev, (double *)pk);
    DZ_HIP(hipGetLastError());
    A.skern = (const double *)pk;
  }
  {  // dpl, inv/CalSurfG.f90:1829-1833 (host libm sin, geometry only)
    float dpl = g.dnx * EARTH;
    float rd1 = g.dnz * EARTH * sinf(g.gox);
    if (rd1 < dpl) dpl = rd1;
    rd1 = g.dnz * EARTH * sinf(g.gox + (float)(g.nnx - 1) * g.dnx);
    if (rd1 < dpl) dpl = rd1;
    A.dplh = dpl;
  }
  {  // reciprocals of the loop-invariant divisors of the ray kernel (divr): the divisors as fp32 values the kernel would divide by
    const float dnxr = g.dvx / (float)(GDX * 8), dnzr = g.dvz / (float)(GDZ * 8);   // = dazim_refbox::dnxr/dnzr (sgdl = 8, fmm.hip)
    const float e2dnx = 2.0f * EARTH * g.dnx, e2dnxr = 2.0f * EARTH * dnxr;
    A.r_dnx = 1.0 / (double)g.dnx;   A.r_dnz = 1.0 / (double)g.dnz;
    A.r_dnxr = 1.0 / (double)dnxr;   A.r_dnzr = 1.0 / (double)dnzr;
    A.r_dvx = 1.0 / (double)g.dvx;   A.r_dvz = 1.0 / (double)g.dvz;
    A.r_e2dnx = 1.0 / (double)e2dnx; A.r_e2dnxr = 1.0 / (double)e2dnxr;
  }
  void *p;
  const int64_t m = nray;
  const size_t nr1 = (size_t)(nray > 0 ? nray : 1);
  if ((rc = dz_scratch(ctx, "rays.status", nr1 * 4, &p))) return rc;
  A.status = (int *)p;
  if ((rc = dz_scratch(ctx, "rays.rb", nr1 * 4, &p))) return rc;
  A.rbflag = (int *)p;
  if ((rc = dz_scratch(ctx, "rays.count", (size_t)(m + 1) * 8, &p))) return rc;
  A.count = (long *)p;
  // LDS cell lists: 512 entries keep 16 wavefronts (128 rays) on a CU, which is worth 25 % on the S-256 grid where longer lists
  // are rare; large inversion grids (S-512: rays cross > 100 cells) get 1024.  Longer lists fall back to a full-grid sweep.
  A.lcap = g.nvx * g.nvz <= 4096 ? 512 : 1024;
  if (ctx->opts.count("rays.lcap") && ctx->opts["rays.lcap"] >= 16 && ctx->opts["rays.lcap"] <= 8192)   // tuning / test knob: small
    A.lcap = ctx->opts["rays.lcap"];                                                                      // values force the fallbacks
  if (A.lcap > g.nvx * g.nvz) A.lcap = g.nvx * g.nvz;
  A.LK = A.lcap;   // cell lists handed from the count pass to the emit pass (longer ones are traced again)
  A.keep_small = ctx->opts.count("rays.keep_small") && ctx->opts["rays.keep_small"] ? 1 : 0;
  A.pts = nullptr;
  A.npts = nullptr;
  A.pcap = 0;
  if (ctx->opts.count("rays.keep_paths") && ctx->opts["rays.keep_paths"]) {
    // a ray advances half a cell per step: a few times (nnx + nnz) points even for a path that wanders; longer ones are flagged
    A.pcap = 4 * (g.nnx + g.nnz) + 16;
    if ((rc = dz_scratch(ctx, "rays.pts", nr1 * (size_t)A.pcap * sizeof(float2), &p))) return rc;
    A.pts = (float2 *)p;
    if ((rc = dz_scratch(ctx, "rays.npts", nr1 * 4, &p))) return rc;
    A.npts = (int *)p;
    DZ_HIP(hipMemsetAsync(A.npts, 0, nr1 * 4, ctx->stream));
  }
  ctx->ksec["rays.path_cap"] = A.pcap;
  ctx->ksec["rays.path_rays"] = A.pts ? (double)nray : 0.0;
  const bool twin = ctx->opts.count("rays.dense_twin") && ctx->opts["rays.dense_twin"] && !A.keep_small;
  A.dense = twin ? 2 : 0;
  A.countd = nullptr;
  if (twin) {
    if ((rc = dz_scratch(ctx, "rays.countd", (size_t)(m + 1) * 8, &p))) return rc;
    A.countd = (long *)p;
    DZ_HIP(hipMemsetAsync(A.countd, 0, (size_t)(m + 1) * 8, ctx->stream));
  }
  if ((rc = dz_scratch(ctx, "rays.nlist", nr1 * 4, &p))) return rc;
  A.nlist = (int *)p;
  if ((rc = dz_scratch(ctx, "rays.lcell", nr1 * A.LK * 2, &p))) return rc;
  A.lcell = (unsigned short *)p;
  if ((rc = dz_scratch(ctx, "rays.lval", nr1 * A.LK * 4 * (joint ? 3 : 1), &p))) return rc;
  A.lval = (float *)p;
  int64_t *rowptr = nullptr;
  float *val = nullptr;
  int *col = nullptr;
  // the caller may announce rows it is going to append (regularisation): the CSR arrays then get that much slack and
  // dazim_csr_append_coo writes behind the ray rows instead of reallocating and copying the matrix
  // By default: one regularisation row per model parameter with the 7-point stencil of inv/TikhRegul.f90 (a few MB).
  int64_t res_rows = (int64_t)g.nvx * g.nvz * (nz - 1) * (joint ? 3 : 1), res_nnz = 7 * res_rows;
  if (ctx->opts.count("csr.reserve_rows") && ctx->opts["csr.reserve_rows"] > res_rows) res_rows = ctx->opts["csr.reserve_rows"];
  if (ctx->opts.count("csr.reserve_nnz") && ctx->opts["csr.reserve_nnz"] > res_nnz) res_nnz = ctx->opts["csr.reserve_nnz"];
  { void *pp; if ((rc = dz_big_get(ctx, (size_t)(m + res_rows + 1) * 8, &pp))) return rc; rowptr = (int64_t *)pp; }
  struct Arrays {   // the matrix arrays go back to the cache on every early return (until the matrix has adopted them)
    dazim_ctx *c; int64_t *&rp; float *&v; int *&cl; bool keep = false;
    ~Arrays() { if (!keep) { dz_big_put(c, rp); dz_big_put(c, v); dz_big_put(c, cl); } }
  } arrays{ctx, rowptr, val, col};
  A.dsurf = dsurf.dev;
  A.rowptr = (const long *)rowptr;
  A.val = nullptr;
  A.col = nullptr;
  const size_t lds = (size_t)A.lcap * 2 * RPW_MAX + 16;   // one cell list per ray of the wavefront
  // the kernel of each pass: count / emit x iso / joint x column-major / tiled fields
  auto kern = [&](bool emit) -> const void * {
    if (emit) {
      if (joint) return tiled ? (const void *)rays_kernel<true, true, true> : (const void *)rays_kernel<true, true, false>;
      return tiled ? (const void *)rays_kernel<true, false, true> : (const void *)rays_kernel<true, false, false>;
    }
    if (joint) return tiled ? (const void *)rays_kernel<false, true, true> : (const void *)rays_kernel<false, true, false>;
    return tiled ? (const void *)rays_kernel<false, false, true> : (const void *)rays_kernel<false, false, false>;
  };
  auto launch = [&](bool emit, const RayArgs &R, long nwg_) -> int {
    RayArgs args = R;
    void *params[] = {&args};
    DZ_HIP(hipLaunchKernel(kern(emit), dim3((unsigned)nwg_), dim3(64), params, lds, ctx->stream));
    return 0;
  };
  DZ_HIP(hipFuncSetAttribute(kern(false), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  DZ_HIP(hipFuncSetAttribute(kern(true), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = (int)(160 * 1024 / (lds + 256));
  {   // resident workgroups per CU are limited by LDS or by registers (3 wavefronts per SIMD): persistent workgroups beyond
      // that only queue up behind the resident ones and unbalance the XCD-ordered ray ranges
    int occ = 0;
    const void *kf = kern(false);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kf, 64, lds) == hipSuccess && occ > 0 && occ < per_cu) per_cu = occ;
  }
  if (per_cu > 16) per_cu = 16;
  if (per_cu < 1) return dz_fail(ctx, DAZIM_E_BAD_ARG, "inversion grid too large for the LDS cell lists");
  if (ctx->opts.count("rays.wg_per_cu") && ctx->opts["rays.wg_per_cu"] > 0 && ctx->opts["rays.wg_per_cu"] < per_cu) per_cu = ctx->opts["rays.wg_per_cu"];
  long nwg = (long)ctx->num_cu * per_cu;
  if (nwg > (nray + RPW_MAX - 1) / RPW_MAX) nwg = (nray + RPW_MAX - 1) / RPW_MAX;
  if (nwg >= 8) nwg -= nwg % 8;   // the XCD-aware ray order wants a multiple of 8
  if (nwg < 1) nwg = 1;
  if ((rc = dz_scratch(ctx, "rays.fdm", (size_t)nwg * RPW_MAX * (g.nvx + 2) * (g.nvz + 2) * 4 * (joint ? 3 : 1), &p))) return rc;
  A.fdm_scratch = (float *)p;
  A.perm = nullptr;
  if (nray >= 64 && nray < (1ll << 32) && !(ctx->opts.count("rays.sort") && !ctx->opts["rays.sort"])) {
    int fbits = 1;
    while ((1ll << fbits) < nfield) fbits++;
    const int dbits = 32 - fbits > 12 ? 12 : 32 - fbits;
    if (dbits >= 4) {
      unsigned *k0, *k1, *v0, *v1;
      if ((rc = dz_scratch(ctx, "rays.sortbuf", (size_t)nray * 16 + 64, &p))) return rc;
      k0 = (unsigned *)p; k1 = k0 + nray; v0 = k1 + nray; v1 = v0 + nray;
      const float ex = (float)g.nnx * g.dnx, ez = (float)g.nnz * g.dnz;
      const float inv_dmax = 1.0f / sqrtf(ex * ex + ez * ez);
      hipLaunchKernelGGL(k_ray_keys, dim3((unsigned)((nray + 255) / 256)), dim3(256), 0, ctx->stream, (long)nray, field.dev, scx.dev,
                         scz.dev, rcx.dev, rcz.dev, inv_dmax, dbits, k0, v0, overlap ? A.tslot : (const int *)nullptr);
      size_t tb = 0;
      DZ_HIP(rocprim::radix_sort_pairs(nullptr, tb, k0, k1, v0, v1, (size_t)nray, 0, fbits + dbits, ctx->stream));
      void *tmp;
      if ((rc = dz_scratch(ctx, "rays.sorttmp", tb + 256, &tmp))) return rc;
      DZ_HIP(rocprim::radix_sort_pairs(tmp, tb, k0, k1, v0, v1, (size_t)nray, 0, fbits + dbits, ctx->stream));
      A.perm = v1;
    }
  }
  if ((rc = dz_scratch(ctx, "rays.qcount", 128, &p))) return rc;
  A.qcount = (unsigned *)p;
  DZ_HIP(hipMemsetAsync(A.qcount, 0, 128, ctx->stream));
  int64_t nnz = 0;
  double overlap_tail_s = 0.0;
  if (overlap) {
    const size_t nq = (size_t)((nray + RPW_MAX - 1) / RPW_MAX) + 8;
    if ((rc = dz_scratch(ctx, "rays.defer", nq * 4, &p))) return rc;
    A.defer_mark = (int *)p;
    DZ_HIP(hipMemsetAsync(A.defer_mark, 0, nq * 4, ctx->stream));
  }
  DzTimer t(ctx, "rays");
  DZ_HIP(hipMemsetAsync(A.count, 0, (size_t)(m + 1) * 8, ctx->stream));
  if (!overlap) {
    if (nray > 0 && (rc = launch(false, A, nwg))) return rc;
  } else {
    // ---- the count pass beside the eikonal launch's tail: a sequence of NON-BLOCKING passes on the third stream ----
    // A pass traces every quad of rays whose fields are finished and marks the others; the host launches the next pass a moment
    // later for the marked ones, and a last one after the eikonal call is complete (spill reruns included).  The first pass goes
    // out when the eikonal launch's task queue is nearly empty (the kernel reports the tasks it has handed out through
    // host-mapped words): from then on its workgroups leave and the passes' workgroups take their slots.  No ray workgroup ever
    // waits for the eikonal kernel, so nothing depends on which launch's wavefronts the hardware keeps resident.
    const auto t_poll0 = std::chrono::steady_clock::now();
    auto fmm_over = [&]() {
      const bool over = hipEventQuery(ctx->ev_f1) == hipSuccess;
      if (!over) (void)hipGetLastError();
      return over;
    };
    if (ctx->fields.hprog && ctx->fields.total_tasks > 0) {
      unsigned slack = ctx->fields.nwg / 4 + 8 * 16;
      if (ctx->opts.count("rays.start_slack") && ctx->opts["rays.start_slack"] > 0) slack = (unsigned)ctx->opts["rays.start_slack"];   // (tuning)
      const unsigned total = ctx->fields.total_tasks;
      const unsigned want = total > slack ? total - slack : 0;
      for (;;) {
        unsigned have = 0;
        for (int c = 0; c < 8; c++) have += ctx->fields.hprog[c];
        if (have >= want || fmm_over()) break;
        if (std::chrono::steady_clock::now() - t_poll0 > std::chrono::seconds(120)) break;
        usleep(50);
      }
    }
    const unsigned nquad_all = (unsigned)((nray + RPW_MAX - 1) / RPW_MAX);
    unsigned pending = nquad_all, *h_traced = nullptr;
    { void *hp; if ((rc = dz_pinned(ctx, "rays.host_pending", 64, &hp))) return rc; h_traced = (unsigned *)hp; }
    DZ_HIP(hipMemsetAsync(A.defer_mark, 1, (size_t)nquad_all * 4, ctx->stream));   // every quad marked (any non-zero value)
    int npass = 0;
    bool collected = false, ev0_set = false;
    unsigned left_by_first = 0;
    int rcf = 0;
    while (pending > 0) {
      const bool last = fmm_over();
      if (last && !collected) {   // the eikonal call's own end: statuses, spill reruns (main stream), before the pass that needs every field
        swap.restore();
        DZ_HIP(hipStreamSynchronize(ctx->stream3));   // (the preparation and the passes so far ran there: the last pass runs on the main stream)
        rcf = dz_fmm_finish(ctx);
        collected = true;
        if (rcf) break;
        if (npass == 0) (void)hipEventRecord(ctx->ev_r1, ctx->stream);   // (nothing ran beside the launch)
        (void)hipEventRecord(ctx->ev0, ctx->stream);                     // what follows the eikonal launch's end is what the step pays
        ev0_set = true;
      }
      long nwg_pass = nwg;
      A.max_quads = 0;
      A.sweeps = collected ? 1 : (ctx->opts.count("rays.sweeps") && ctx->opts["rays.sweeps"] > 0 ? ctx->opts["rays.sweeps"] : 3);   // (measured: 1 -> 300-302 ms, 2-4 -> 297.6-299.6, 6 -> 301-302, 12 -> 305.6, 32 -> 319.6: every look costs)
      if (collected) A.fdone = nullptr;
      DZ_HIP(hipMemsetAsync(A.qcount, 0, 32, ctx->stream));        // the count pass's task counters
      if ((rc = launch(false, A, nwg_pass))) return rc;
      if (!collected) DZ_HIP(hipEventRecord(ctx->ev_r1, ctx->stream));
      DZ_HIP(hipMemcpyAsync(h_traced, A.qcount + 18, 4, hipMemcpyDeviceToHost, ctx->stream));
      DZ_HIP(hipStreamSynchronize(ctx->stream));
      const unsigned before = pending;
      pending = collected ? 0u : nquad_all - *h_traced;
      if (npass == 0) left_by_first = pending;
      npass++;
      if (pending > 0 && pending == before) usleep(100);   // (nothing was ready: let a few more fields finish)
    }
    swap.restore();
    if (!collected) rcf = dz_fmm_finish(ctx);
    DZ_HIP(hipStreamSynchronize(ctx->stream3));
    if (rcf) return rcf;
    ctx->ksec["rays.passes"] = npass;
    ctx->ksec["rays.deferred_quads"] = left_by_first;   // quads the first pass had to leave (fields still marching, or waiting for their rerun)
    A.fdone = nullptr;
    A.defer_mark = nullptr;
    A.max_quads = 0;
    A.sweeps = 1;
    // what the step pays for the rays: from the end of the eikonal launch on (the passes' share beside it is free) = the last
    // pass beside the launch, as far as it outlasted it, + everything below
    float ms_tail = 0;
    if (hipEventElapsedTime(&ms_tail, ctx->ev_f1, ctx->ev_r1) == hipSuccess && ms_tail > 0) overlap_tail_s = ms_tail * 1e-3;
    if (!ev0_set) (void)hipEventRecord(ctx->ev0, ctx->stream);
  }
  {  // exclusive scan of the row counts -> rowptr
    size_t tb = 0;
    DZ_HIP(rocprim::exclusive_scan(nullptr, tb, A.count, (long *)rowptr, 0l, (size_t)(m + 1), rocprim::plus<long>(), ctx->stream));
    if ((rc = dz_scratch(ctx, "rays.scan", tb + 256, &p))) return rc;
    DZ_HIP(rocprim::exclusive_scan(p, tb, A.count, (long *)rowptr, 0l, (size_t)(m + 1), rocprim::plus<long>(), ctx->stream));
    DZ_HIP(hipMemcpyAsync(&nnz, rowptr + m, 8, hipMemcpyDeviceToHost, ctx->stream));
    DZ_HIP(hipStreamSynchronize(ctx->stream));
  }
  const int64_t cap_nnz = nnz + res_nnz;   // (options csr.reserve_rows / csr.reserve_nnz: room for rows appended later)
  { void *pp; if ((rc = dz_big_get(ctx, (size_t)(cap_nnz > 0 ? cap_nnz : 1) * 4, &pp))) return rc; val = (float *)pp; }
  { void *pp; if ((rc = dz_big_get(ctx, (size_t)(cap_nnz > 0 ? cap_nnz : 1) * 4, &pp))) return rc; col = (int *)pp; }
  A.val = val;
  A.col = col;
  if (nray > 0 && (rc = launch(true, A, nwg))) return rc;
  // ---- the dense twin: a second emit pass over the saved cell lists (no ray is traced again unless its list did not fit) ----
  int64_t *rowptr_d = nullptr;
  float *val_d = nullptr;
  int *col_d = nullptr;
  Arrays arrays_d{ctx, rowptr_d, val_d, col_d};
  int64_t nnz_d = 0;
  if (twin) {
    { void *pp; if ((rc = dz_big_get(ctx, (size_t)(m + 1) * 8, &pp))) return rc; rowptr_d = (int64_t *)pp; }
    size_t tb = 0;
    DZ_HIP(rocprim::exclusive_scan(nullptr, tb, A.countd, (long *)rowptr_d, 0l, (size_t)(m + 1), rocprim::plus<long>(), ctx->stream));
    if ((rc = dz_scratch(ctx, "rays.scan", tb + 256, &p))) return rc;
    DZ_HIP(rocprim::exclusive_scan(p, tb, A.countd, (long *)rowptr_d, 0l, (size_t)(m + 1), rocprim::plus<long>(), ctx->stream));
    DZ_HIP(hipMemcpyAsync(&nnz_d, rowptr_d + m, 8, hipMemcpyDeviceToHost, ctx->stream));
    DZ_HIP(hipStreamSynchronize(ctx->stream));
    { void *pp; if ((rc = dz_big_get(ctx, (size_t)(nnz_d > 0 ? nnz_d : 1) * 4, &pp))) return rc; val_d = (float *)pp; }
    { void *pp; if ((rc = dz_big_get(ctx, (size_t)(nnz_d > 0 ? nnz_d : 1) * 4, &pp))) return rc; col_d = (int *)pp; }
    RayArgs D = A;
    D.dense = 1;
    D.rowptr = (const long *)rowptr_d;
    D.val = val_d;
    D.col = col_d;
    DZ_HIP(hipMemsetAsync(A.qcount + 8, 0, 32, ctx->stream));   // the emit pass's task counters
    if (nray > 0 && (rc = launch(true, D, nwg))) return rc;
  }
  t.stop();
  if (overlap) {
    ctx->ksec["rays.after_fmm_count"] = overlap_tail_s;
    ctx->ksec["rays"] += overlap_tail_s;
  }
  // statuses: first failing ray is the reference's STOP
  std::vector<int> hs(nr1), hb(nr1);
  unsigned hq[2] = {0, 0};
  if (nray > 0) DZ_HIP(hipMemcpyAsync(hq, A.qcount + 16, 8, hipMemcpyDeviceToHost, ctx->stream));
  if (nray > 0) {
    DZ_HIP(hipMemcpyAsync(hs.data(), A.status, nray * 4, hipMemcpyDeviceToHost, ctx->stream));
    DZ_HIP(hipMemcpyAsync(hb.data(), A.rbflag, nray * 4, hipMemcpyDeviceToHost, ctx->stream));
    DZ_HIP(hipStreamSynchronize(ctx->stream));
  }
  int nb = 0, err = 0;
  for (int64_t i = 0; i < nray; i++) {
    nb += hb[i];
    if (hs[i] && !err) err = dz_fail(ctx, hs[i], "ray %ld: receiver lies outside model", (long)i);
  }
  if (n_boundary) *n_boundary = nb;
  if (nnz_out) *nnz_out = nnz;
  ctx->ksec["rays.list_sweeps"] = hq[0];     // rays that took the full-grid sweep instead of the LDS cell list
  ctx->ksec["rays.list_retraced"] = hq[1];   // rays the emit pass traced a second time
  ctx->ksec["rays.lcap"] = A.lcap;
  if ((rc = dsurf.finish())) return rc;
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  if (err) return err;
  const int64_t n = (int64_t)g.nvx * g.nvz * (nz - 1) * (joint ? 3 : 1);
  arrays.keep = true;   // adopted (dz_csr_adopt_cap frees them itself if it fails)
  if ((rc = dz_csr_adopt_cap(ctx, m, n, nnz, rowptr, col, val, m + res_rows, cap_nnz, G))) return rc;
  if (twin) {
    dazim_csr *Gd = nullptr;
    arrays_d.keep = true;
    if ((rc = dz_csr_adopt_cap(ctx, m, n, nnz_d, rowptr_d, col_d, val_d, 0, 0, &Gd)) || (rc = dz_csr_set_twin(ctx, *G, Gd))) {
      dazim_csr_free(ctx, *G);
      *G = nullptr;
      return rc;
    }
  }
  return 0;
}

// = the receiver loop of CalSurfG (inv/CalSurfG.f90:1326-1364) for every ray of a batch of fields
extern "C" int dazim_rays_build_G(dazim_ctx *ctx, int nx, int ny, int nz, float goxd, float gozd, float dvxd,
                                  float dvzd, int kmax, const float *vels, int nfield, const float *scx,
                                  const float *scz, const int *period, const int *kidx, const float *veln,
                                  const float *ttn, const float *ttnr, const int *nstsr, const dazim_refbox *boxes,
                                  int64_t nray, const int *field, const float *rcx, const float *rcz,
                                  const double *svs, const double *svp, const double *srho, float *dsurf,
                                  dazim_csr **G, int64_t *nnz_out, int *n_boundary) {
  return rays_build_impl(ctx, nx, ny, nz, goxd, gozd, dvxd, dvzd, kmax, vels, nfield, scx, scz, period, kidx, veln, ttn, ttnr,
                         nstsr, boxes, nray, field, rcx, rcz, svs, svp, srho, nullptr, dsurf, G, nnz_out, n_boundary);
}
// = the receiver loop of CalSurfGAnisoJoint (inv/CalSurfGAniso_Joint.f90:680-752): rpathsAzim and rows
// with the three column blocks dVs | Gc | Gs; lsen = Lsen_Gsc from depthkernelTI (TI kernels, an input)
extern "C" int dazim_rays_build_G_joint(dazim_ctx *ctx, int nx, int ny, int nz, float goxd, float gozd, float dvxd,
                                        float dvzd, int kmax, const float *vels, int nfield, const float *scx,
                                        const float *scz, const int *period, const int *kidx, const float *veln,
                                        const float *ttn, const float *ttnr, const int *nstsr,
                                        const dazim_refbox *boxes, int64_t nray, const int *field, const float *rcx,
                                        const float *rcz, const double *svs, const double *svp, const double *srho,
                                        const float *lsen, float *dsurf, dazim_csr **G, int64_t *nnz_out,
                                        int *n_boundary) {
  if (!lsen) return dz_fail(ctx, DAZIM_E_BAD_ARG, "joint mode needs Lsen_Gsc");
  return rays_build_impl(ctx, nx, ny, nz, goxd, gozd, dvxd, dvzd, kmax, vels, nfield, scx, scz, period, kidx, veln, ttn, ttnr,
                         nstsr, boxes, nray, field, rcx, rcz, svs, svp, srho, lsen, dsurf, G, nnz_out, n_boundary);
}

// The ray geometries of the last dazim_rays_build_G[_joint] call made with option "rays.keep_paths" = 1: what the reference
// writes to raypath_refmdl_<T>s.dat when writepath is set (fwd/rpathsAzim.f90:617-625, fwd/FwdTraveltimeCPS.f90:673-691).
// dims: *nray rays, *cap points per ray at most.  copy: xz[nray][cap][2] (colatitude, longitude in rad) and nrp[nray]
// (number of points, receiver first, source last; -1: more than cap points) into host or device arrays.
extern "C" int dazim_ray_paths_dims(dazim_ctx *ctx, int64_t *nray, int *cap) {
  if (!ctx || !nray || !cap) return DAZIM_E_BAD_ARG;
  *nray = (int64_t)dazim_last_kernel_seconds(ctx, "rays.path_rays");
  *cap = (int)dazim_last_kernel_seconds(ctx, "rays.path_cap");
  if (*nray < 0) *nray = 0;
  if (*cap < 0) *cap = 0;
  return 0;
}
extern "C" int dazim_ray_paths_copy(dazim_ctx *ctx, float *xz, int *nrp) {
  int64_t nray;
  int cap;
  if (!ctx || !xz || !nrp || dazim_ray_paths_dims(ctx, &nray, &cap)) return DAZIM_E_BAD_ARG;
  if (nray == 0 || cap == 0) return dz_fail(ctx, DAZIM_E_BAD_ARG, "no ray paths were kept (option rays.keep_paths)");
  auto a = ctx->scratch.find("rays.pts"), b = ctx->scratch.find("rays.npts");
  if (a == ctx->scratch.end() || b == ctx->scratch.end()) return dz_fail(ctx, DAZIM_E_BAD_ARG, "no ray paths were kept");
  DZ_HIP(hipMemcpyAsync(xz, a->second.first, (size_t)nray * cap * 8, hipMemcpyDefault, ctx->stream));
  DZ_HIP(hipMemcpyAsync(nrp, b->second.first, (size_t)nray * 4, hipMemcpyDefault, ctx->stream));
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}
