// sparse.hip -- K6/K7 of SURVEY.md: the sensitivity matrix on the device (CSR + stable transpose)
// its two products (aprod, inv/aprod.f90:7) and the fp32 LSMR solver (inv/lsmrModule.f90:36).
//
// The products are HBM-bound: 8 B per stored entry (fp32 value + int32 index) are streamed once per
// product with 16-byte loads, one wavefront per row (CSR, A*x) or per column (CSC, A^T*y), and a
// 64-lane shuffle reduction; the gathered vector stays in L2.  Both products write their result in
// a fixed order (no atomics) so that LSMR is reproducible run to run.
#include "dazim_internal.h"

#include <rccl/rccl.h>
#include <unistd.h>

#include <cstdio>
#include <string>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

struct dazim_csr {
  int64_t m = 0, n = 0, nnz = 0;
  int64_t *rowptr = nullptr, *colptr = nullptr;  // [m+1], [n+1]
  int *col = nullptr, *row = nullptr;            // CSR column / CSC row of each entry, 0-based
  float *val = nullptr, *tval = nullptr;         // CSR / CSC values
  unsigned *tperm = nullptr;                     // CSC entry -> CSR entry (for value rescaling)
  // column-blocked view of the (canonical: columns ascending inside a row) CSR for the scatter form
  // of A^T*y: row r's entries with column in block b are [cbptr[r*(ncb+1)+b], cbptr[r*(ncb+1)+b+1])
  int ncb = 0, cbw = 0;                          // number of column blocks, block width
  int64_t *cbptr = nullptr;
  float vmax = 0.0f;                             // max |val|, sets the fixed-point scale
  // rows [0, split_row) hold the long rows, the rows from split_row on are all shorter than SPLIT_SHORT entries (G: the ray rows,
  // then the seven-entry regularisation rows) -- the blocked products give each part the lane grouping it wants.  m: no short tail.
  int64_t split_row = 0;
  double long_avg = 0.0;                         // entries per row in [0, split_row)
  // the same column indices in 16 bits: the two products of an LSMR iteration stream 6 instead of 8 bytes per stored entry.
  // col16_mod = 0: the column itself (n <= 65536, the S-256 matrix); col16_mod = 2*cbw > 0: the column relative to the first
  // column of its PAIR of column blocks (larger n: the scatter kernel works on one block, the blocked A*x on a pair).
  // Built with the column blocks; nullptr when not used.
  unsigned short *col16 = nullptr;
  int64_t col16_cap = 0;                         // entries col16 can hold
  int col16_mod = 0;
  // rows / entries the arrays rowptr (cap_m + 1), col and val (cap_nnz) can hold: rays_build_G allocates them with the slack
  // the options csr.reserve_rows / csr.reserve_nnz ask for, so that the regularisation rows are appended in place
  // (0: exactly m / nnz)
  int64_t cap_m = 0, cap_nnz = 0;
  dazim_csr *twin = nullptr;   // option rays.dense_twin: the reference's dense copies GVs | GGc | GGs of the same rows
};

namespace {

constexpr int WPB = 4;  // wavefronts per workgroup in the row kernels

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// out[r] = beta*out[r] + sum_k val[k]*x[idx[k]], k in [ptr[r], ptr[r+1]) ; one wavefront per row.
// If sumsq != nullptr the workgroup also writes its partial sum of out[r]^2 (double) for a norm.
__global__ __launch_bounds__(64 * WPB) void spmv_rows(int64_t nrows, const int64_t *__restrict__ ptr,
                                                        const int *__restrict__ idx,
                                                        const float *__restrict__ val,
                                                        const float *__restrict__ x, float *__restrict__ out,
                                                        const float *__restrict__ beta_p, float beta_sign,
                                                        double *__restrict__ sumsq, const int *__restrict__ guard) {
  if (guard && *guard) return;   // LSMR has stopped (or skips this half-step): see LsmrState
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float beta = beta_p ? beta_sign * beta_p[0] : beta_sign;
  double sq = 0.0;
  for (int64_t r = (int64_t)blockIdx.x * WPB + w; r < nrows; r += (int64_t)gridDim.x * WPB) {
    const int64_t s = ptr[r], e = ptr[r + 1];
    float acc = 0.0f;
    int64_t s4 = (s + 3) & ~(int64_t)3;
    if (s4 > e) s4 = e;
    for (int64_t i = s + lane; i < s4; i += 64) acc += val[i] * x[idx[i]];
    const int64_t e4 = s4 + ((e - s4) & ~(int64_t)3);
    for (int64_t i = s4 + 4 * lane; i < e4; i += 256) {
      const float4 v = *reinterpret_cast<const float4 *>(val + i);
      const int4 c = *reinterpret_cast<const int4 *>(idx + i);
      acc += v.x * x[c.x];
      acc += v.y * x[c.y];
      acc += v.z * x[c.z];
      acc += v.w * x[c.w];
    }
    for (int64_t i = e4 + lane; i < e; i += 64) acc += val[i] * x[idx[i]];
    acc = wave_sum(acc);
    if (lane == 0) {
      const float o = beta * out[r] + acc;
      out[r] = o;
      sq += (double)o * (double)o;
    }
  }
  if (sumsq) {
    __shared__ double s_sq[WPB];
    if (lane == 0) s_sq[w] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int i = 0; i < WPB; i++) t += s_sq[i];
      sumsq[blockIdx.x] = t;
    }
  }
}

// Same product with the dense vector staged in LDS: used when 4*len(x) fits (<= 152 KB), which
// removes the per-entry cache-line gather that otherwise bounds the product (~1 line/clk/CU).
// One workgroup of 16 wavefronts per CU, each wavefront a row at a time, two 16-byte loads of
// values and of indices in flight per lane.
constexpr int LWPB = 16;
#ifndef DZ_LDSX_UNROLL4
#define DZ_LDSX_UNROLL4 1
#endif
// four consecutive column indices with one load: int4 (16 bytes) or ushort4 (8 bytes)
template <class IT> struct Idx4;
template <> struct Idx4<int> { using type = int4; };
template <> struct Idx4<unsigned short> { using type = ushort4; };
template <class IT>
__global__ __launch_bounds__(64 * LWPB) void spmv_rows_ldsx(int64_t nrows, int64_t nx, const int64_t *__restrict__ ptr,
                                                           const IT *__restrict__ idx, const float *__restrict__ val,
                                                           const float *__restrict__ x, float *__restrict__ out,
                                                           const float *__restrict__ beta_p, float beta_sign,
                                                           double *__restrict__ sumsq, const int *__restrict__ guard) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  if (guard && *guard) return;
  for (int64_t i = threadIdx.x * 4; i < nx; i += 64 * LWPB * 4) {
    if (i + 3 < nx)
      *reinterpret_cast<float4 *>(xs + i) = *reinterpret_cast<const float4 *>(x + i);
    else
      for (int64_t j = i; j < nx; j++) xs[j] = x[j];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float beta = beta_p ? beta_sign * beta_p[0] : beta_sign;
  double sq = 0.0;
  for (int64_t r = (int64_t)blockIdx.x * LWPB + w; r < nrows; r += (int64_t)gridDim.x * LWPB) {
    const int64_t s = ptr[r], e = ptr[r + 1];
    float acc = 0.0f;
    int64_t s4 = (s + 3) & ~(int64_t)3;
    if (s4 > e) s4 = e;
    for (int64_t i = s + lane; i < s4; i += 64) acc += val[i] * xs[idx[i]];
    const int64_t e4 = s4 + ((e - s4) & ~(int64_t)3);
    int64_t i = s4 + 4 * lane;
#if DZ_LDSX_UNROLL4
    for (; i + 768 < e4; i += 1024) {   // four groups in flight (16-bit indices leave 24 instead of 32 bytes per lane and group)
      using I4 = typename Idx4<IT>::type;
      float4 v[4];
      I4 c[4];
#pragma unroll
      for (int g = 0; g < 4; g++) {
        v[g] = *reinterpret_cast<const float4 *>(val + i + 256 * g);
        c[g] = *reinterpret_cast<const I4 *>(idx + i + 256 * g);
      }
#pragma unroll
      for (int g = 0; g < 4; g++) {
        acc += v[g].x * xs[c[g].x];
        acc += v[g].y * xs[c[g].y];
        acc += v[g].z * xs[c[g].z];
        acc += v[g].w * xs[c[g].w];
      }
    }
#endif
    for (; i + 256 < e4; i += 512) {
      using I4 = typename Idx4<IT>::type;
      const float4 v0 = *reinterpret_cast<const float4 *>(val + i);
      const I4 c0 = *reinterpret_cast<const I4 *>(idx + i);
      const float4 v1 = *reinterpret_cast<const float4 *>(val + i + 256);
      const I4 c1 = *reinterpret_cast<const I4 *>(idx + i + 256);
      acc += v0.x * xs[c0.x];
      acc += v0.y * xs[c0.y];
      acc += v0.z * xs[c0.z];
      acc += v0.w * xs[c0.w];
      acc += v1.x * xs[c1.x];
      acc += v1.y * xs[c1.y];
      acc += v1.z * xs[c1.z];
      acc += v1.w * xs[c1.w];
    }
    for (; i < e4; i += 256) {
      using I4 = typename Idx4<IT>::type;
      const float4 v = *reinterpret_cast<const float4 *>(val + i);
      const I4 c = *reinterpret_cast<const I4 *>(idx + i);
      acc += v.x * xs[c.x];
      acc += v.y * xs[c.y];
      acc += v.z * xs[c.z];
      acc += v.w * xs[c.w];
    }
    for (int64_t k = e4 + lane; k < e; k += 64) acc += val[k] * xs[idx[k]];
    acc = wave_sum(acc);
    if (lane == 0) {
      const float o = beta * out[r] + acc;
      out[r] = o;
      sq += (double)o * (double)o;
    }
  }
  if (sumsq) {
    __shared__ double s_sq[LWPB];
    if (lane == 0) s_sq[w] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int k = 0; k < LWPB; k++) t += s_sq[k];
      sumsq[blockIdx.x] = t;
    }
  }
}

// ---- small vector kernels (all O(m+n), negligible next to the products) ----------------------
constexpr int VB = 256;   // threads per block
constexpr int NPART = 256;  // partial sums per reduction

__device__ __forceinline__ void block_partial(double v, double *part) {
  __shared__ double s[VB / 64];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < VB / 64; i++) t += s[i];
    part[blockIdx.x] = t;
  }
}
// res[0] = sqrt(sum part[0..np)) as fp32 (the reference's dnrm2 result type), res[1] = the sum
__global__ void finish_norm(const double *part, int np, float *res, double *res_d) {
  double t = 0.0;
  for (int i = threadIdx.x; i < np; i += 64) t += part[i];
  t = wave_sum(t);
  if (threadIdx.x == 0) {
    res[0] = (float)sqrt(t);
    if (res_d) res_d[0] = t;
  }
}
// res[0] = sqrt(*sum) (after the all-reduce of a distributed norm)
__global__ void k_sqrt_sum(const double *sum, float *res) { res[0] = (float)sqrt(sum[0]); }
// v = w + sign*beta*v with the partial of ||v||^2 (distributed A^T u: w is the all-reduced product)
__global__ void k_axpby_norm(int64_t n, const float *w, float *v, const float *beta_p, float beta_sign, double *part,
                             const int *guard) {
  if (guard && *guard) return;
  const float beta = beta_p ? beta_sign * beta_p[0] : beta_sign;
  double sq = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB) {
    const float o = beta * v[i] + w[i];
    v[i] = o;
    sq += (double)o * o;
  }
  block_partial(sq, part);
}
__global__ void k_sumsq(int64_t n, const float *x, double *part) {
  double v = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB) v += (double)x[i] * x[i];
  block_partial(v, part);
}
// x *= sign/ (*d)   or  x *= sign * (*d)
__global__ void k_scal_inv(int64_t n, float *x, const float *d, float sign) {
  const float a = sign * (1.0f / d[0]);
  for (int64_t i = (int64_t)blockIdx.x * VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB) x[i] = a * x[i];
}
__global__ void k_copy(int64_t n, const float *a, float *b) {
  for (int64_t i = (int64_t)blockIdx.x * VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB) b[i] = a[i];
}
// hbar = h - f1*hbar ; x = x + f2*hbar ; h = v - f3*h ; partial of ||x||^2   (inv/lsmrModule.f90:539-541,590)
__global__ void k_update(int64_t n, float f1, float f2, float f3, float *h, float *hbar, float *x,
                         const float *v, double *part) {
  double sq = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB) {
    const float hb = h[i] - f1 * hbar[i];
    const float xn = x[i] + f2 * hb;
    hbar[i] = hb;
    x[i] = xn;
    h[i] = v[i] - f3 * h[i];
    sq += (double)xn * xn;
  }
  block_partial(sq, part);
}
// ---- LSMR with the scalar recurrences on the device (inv/lsmrModule.f90:484-616) ----------------------------------------
// The host only enqueues kernels: alpha, beta, the plane rotations, the norm estimates and the stopping tests live in one
// LsmrState in HBM, so an iteration needs no host round trip.  The stop flag is looked at every few iterations; the
// iterations enqueued behind the one that stopped return at once (every kernel of the loop starts with `if (*guard) return`),
// so x, itn and the norm estimates are exactly those of the stopping iteration.
struct LsmrState {
  float alpha, beta, alphabar, zetabar, rho, rhobar, cbar, sbar;
  float betadd, betad, rhodold, tautildeold, thetatilde, zeta, d;
  float normA2, maxrbar, minrbar, normb, ctol;
  float normA, condA, normr, normAr, normx;
  float damp, atol, btol;
  float alpha_new;   // alpha of the running iteration: k_alpha_update -> k_tests
  int itn, istop, itnlim;
  int stop;          // != 0: LSMR has stopped; guard of the first half-step (u = A v - alpha u)
  int stop2;         // stop, or beta == 0 in the running iteration: guard of the second half-step (skipped as a block, :490-503)
};
__device__ __forceinline__ float dz_d2norm(float a, float bb) {   // d2norm, :708-721
  const float scale = fabsf(a) + fabsf(bb);
  if (scale == 0.0f) return 0.0f;
  return scale * sqrtf((a / scale) * (a / scale) + (bb / scale) * (bb / scale));
}
// one pass of the scalar recurrences, statement for statement :506-588 (fp32, no contraction); s -> state after the
// iteration that produced (alpha, beta); f1..f3 are the coefficients of the hbar / x / h updates (:539-541)
__device__ __forceinline__ void lsmr_recur(LsmrState &s, float alpha, float beta, float &f1, float &f2, float &f3) {
  const float damp = s.damp;
  float alphabar = s.alphabar, zetabar = s.zetabar, rho = s.rho, rhobar = s.rhobar, cbar = s.cbar, sbar = s.sbar;
  float betadd = s.betadd, betad = s.betad, rhodold = s.rhodold, tautildeold = s.tautildeold, thetatilde = s.thetatilde;
  float zeta = s.zeta, d = s.d, normA2 = s.normA2, maxrbar = s.maxrbar, minrbar = s.minrbar;
  const int itn = s.itn + 1;
  const float alphahat = dz_d2norm(alphabar, damp);
  const float chat = alphabar / alphahat, shat = damp / alphahat;
  const float rhoold = rho;
  rho = dz_d2norm(alphahat, beta);
  const float c = alphahat / rho, sn = beta / rho;
  const float thetanew = sn * alpha;
  alphabar = c * alpha;
  const float rhobarold = rhobar, zetaold = zeta;
  const float thetabar = sbar * rho, rhotemp = cbar * rho;
  rhobar = dz_d2norm(cbar * rho, thetanew);
  cbar = cbar * rho / rhobar;
  sbar = thetanew / rhobar;
  zeta = cbar * zetabar;
  zetabar = -sbar * zetabar;
  f1 = thetabar * rho / (rhoold * rhobarold);
  f2 = zeta / (rho * rhobar);
  f3 = thetanew / rho;
  const float betaacute = chat * betadd, betacheck = -shat * betadd;
  const float betahat = c * betaacute;
  betadd = -sn * betaacute;
  const float thetatildeold = thetatilde;
  const float rhotildeold = dz_d2norm(rhodold, thetabar);
  const float ctildeold = rhodold / rhotildeold, stildeold = thetabar / rhotildeold;
  thetatilde = stildeold * rhobar;
  rhodold = ctildeold * rhobar;
  betad = -stildeold * betad + ctildeold * betahat;
  tautildeold = (zetaold - thetatildeold * tautildeold) / rhotildeold;
  const float taud = (zeta - thetatilde * tautildeold) / rhodold;
  d = d + betacheck * betacheck;
  s.normr = sqrtf(d + (betad - taud) * (betad - taud) + betadd * betadd);
  normA2 = normA2 + beta * beta;
  s.normA = sqrtf(normA2);
  normA2 = normA2 + alpha * alpha;
  maxrbar = fmaxf(maxrbar, rhobarold);
  if (itn > 1) minrbar = fminf(minrbar, rhobarold);
  s.condA = fmaxf(maxrbar, rhotemp) / fminf(minrbar, rhotemp);
  s.normAr = fabsf(zetabar);
  s.alpha = alpha; s.beta = beta; s.alphabar = alphabar; s.zetabar = zetabar; s.rho = rho; s.rhobar = rhobar; s.cbar = cbar;
  s.sbar = sbar; s.betadd = betadd; s.betad = betad; s.rhodold = rhodold; s.tautildeold = tautildeold;
  s.thetatilde = thetatilde; s.zeta = zeta; s.d = d; s.normA2 = normA2; s.maxrbar = maxrbar; s.minrbar = minrbar;
  s.itn = itn;
}
// sum of np partials by the first wavefront of the block in a fixed order: every block, every launch gets the same bits
__device__ __forceinline__ double block_total(const double *part, int np) {
  __shared__ double s_t;
  if (threadIdx.x < 64) {
    double t = 0.0;
    for (int i = threadIdx.x; i < np; i += 64) t += part[i];
    t = wave_sum(t);
    if (threadIdx.x == 0) s_t = t;
  }
  __syncthreads();
  return s_t;
}
// beta = ||u|| from the partials of the product that wrote u (or from the all-reduced sum); u /= beta; localVEnqueue(v)
// (:487-492).  beta == 0 skips the second half-step of this iteration (stop2).
__global__ void k_beta_scal_u(int64_t m, float *u, const double *part, int np, const double *sum_in, int64_t n, const float *v,
                              float *lv_slot, LsmrState *S) {
  if (S->stop) return;
  const double t = sum_in ? sum_in[0] : block_total(part, np);
  const float beta = (float)sqrt(t);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    S->beta = beta;
    S->stop2 = !(beta > 0.0f);
  }
  if (!(beta > 0.0f)) return;
  const float a = 1.0f / beta;
  for (int64_t i = (int64_t)blockIdx.x * VB + threadIdx.x; i < m; i += (int64_t)gridDim.x * VB) u[i] = a * u[i];
  if (lv_slot)
    for (int64_t i = (int64_t)blockIdx.x * VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB) lv_slot[i] = v[i];
}
// Row-sharded solve, ONE collective per iteration (round 5).  The two all-reduces of an iteration used to depend on each other:
// ||u||^2 had to be summed over the ranks before u could be scaled, and only the scaled u went into A_p^T u_p.  A^T is linear, so
// each rank now scales its shard by its OWN norm (u_p / beta_p: entries <= 1, which the fixed-point scatter relies on), forms
// w_p = beta_p A_p^T (u_p / beta_p) = A_p^T u_p, and ONE collective carries the n floats of w and the double beta_p^2 (round 6: an
// all-gather of the ranks' buffers, summed in rank order by k_beta_axpby; a grouped ncclAllReduce with option comm.allreduce);
// afterwards beta = sqrt(sum beta_p^2), v = w / beta - beta v and u_p <- (u_p / beta_p) (beta_p / beta).
// k_local_norm_scal: beta_p^2 -> sum[0], beta_p -> bp[0], u_p /= beta_p.
__global__ void k_local_norm_scal(int64_t m, float *u, const double *part, int np, double *sum, float *bp, const LsmrState *S) {
  if (S->stop) return;
  const double t = block_total(part, np);
  const float b = (float)sqrt(t);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    sum[0] = t;
    bp[0] = b;
  }
  if (!(b > 0.0f)) return;
  const float a = 1.0f / b;
  for (int64_t i = (int64_t)blockIdx.x * VB + threadIdx.x; i < m; i += (int64_t)gridDim.x * VB) u[i] = a * u[i];
}
__global__ void k_scale_by(int64_t n, float *w, const float *f, const int *guard) {
  if (guard && *guard) return;
  const float a = f[0];
  for (int64_t i = (int64_t)blockIdx.x * VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB) w[i] = a * w[i];
}
// after the collective: beta (:487), localVEnqueue(v) (:490-492), u_p = u / beta, v = A^T u - beta v (:496-497) from w = sum_p A_p^T u_p,
// partials of ||v||^2.  beta == 0 skips the second half-step (stop2), as in k_beta_scal_u.  The collective is an all-gather: rank r's
// n floats of w_r and its double beta_r^2 (at byte offset sum_off) sit at gathered + r*stride, and the sums over the ranks are formed
// HERE, in rank order -- the same bits on every rank and with every transport (SURVEY 8e "fix reduction order").  nr = 1: `gathered`
// holds sums already (option comm.allreduce).
__global__ void k_beta_axpby(int64_t m, float *u, int64_t n, float *v, const char *__restrict__ gathered, int nr, size_t stride,
                             size_t sum_off, const float *bp, float *lv_slot, double *part, LsmrState *S) {
  if (S->stop) return;
  double t = *reinterpret_cast<const double *>(gathered + sum_off);
  for (int r = 1; r < nr; r++) t += *reinterpret_cast<const double *>(gathered + (size_t)r * stride + sum_off);
  const float beta = (float)sqrt(t);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    S->beta = beta;
    S->stop2 = !(beta > 0.0f);
  }
  double sq = 0.0;
  if (beta > 0.0f) {
    const float rb = 1.0f / beta, ru = bp[0] * rb;
    for (int64_t i = (int64_t)blockIdx.x * VB + threadIdx.x; i < m; i += (int64_t)gridDim.x * VB) u[i] = ru * u[i];
    for (int64_t i = (int64_t)blockIdx.x * VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB) {
      const float vi = v[i];
      if (lv_slot) lv_slot[i] = vi;
      float w = reinterpret_cast<const float *>(gathered)[i];
      for (int r = 1; r < nr; r++) w += reinterpret_cast<const float *>(gathered + (size_t)r * stride)[i];
      const float o = -beta * vi + rb * w;
      v[i] = o;
      sq += (double)o * o;
    }
  }
  block_partial(sq, part);
}
// local reorthogonalisation step q (localVOrtho, inv/lsmrModule.f90:733-748), modified Gram-Schmidt:
// d = sum(part_in) (the dot of v with lv_prev computed by the previous launch); v -= d*lv_prev;
// part_out = partial dots of the updated v with lv_next, or -- last step, lv_next null -- partials of ||v||^2.
__global__ void k_reorth(int64_t n, float *v, const float *lv_prev, const double *part_in, int np,
                         const float *lv_next, double *part_out, const int *guard) {
  if (guard && *guard) return;
  __shared__ float s_d;
  if (lv_prev) {
    if (threadIdx.x < 64) {
      double t = 0.0;
      for (int i = threadIdx.x; i < np; i += 64) t += part_in[i];
      t = wave_sum(t);
      if (threadIdx.x == 0) s_d = (float)t;
    }
    __syncthreads();
  }
  const float d = lv_prev ? s_d : 0.0f;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB) {
    float vi = v[i];
    if (lv_prev) {
      vi = vi - d * lv_prev[i];
      v[i] = vi;
    }
    acc += lv_next ? (double)vi * lv_next[i] : (double)vi * vi;
  }
  if (part_out) block_partial(acc, part_out);
}
// The same chain in ONE launch (round 5; the chain above is lim + 1 launches of ~5 us each on an n-float vector -- 54 of the 255 us
// of a test4_Yunnan iteration).  At most RC_BLOCKS workgroups of 1024 threads hold v in registers (E elements per thread) and walk
// the window in the reference's order -- d = v . lv_q, v -= d lv_q, modified Gram-Schmidt: each dot sees the subtractions before it
// -- with a grid barrier between a step's partial dots and its subtraction.  The sums are taken in a fixed order (per block, then
// over the blocks by every block alike), so the result does not depend on arrival order; it differs from the chain's only in the
// grouping of the partial sums.  (Taking all dots of the window at once -- classical Gram-Schmidt, two launches -- was tried first:
// the iterates leave the reference's within eight iterations, 1.2e-2 relative on the test system of tests/test_sparse_gpu.py.)
// The barrier: one counter per solve, never reset, target = (barriers so far) x blocks; release / acquire at agent scope
// (MI355X_MICROARCH.md, inter-workgroup visibility).  All blocks must be resident at once: the host launches at most as many as the
// occupancy query allows on the device and takes the chain below otherwise (or on a CU-masked stream).
// Option lsmr.reorth_chain = 1: the chain; lsmr.reorth_blocks: at most that many workgroups.
constexpr int RC_BLOCKS = 64, RC_THREADS = 1024, RC_EMAX = 16;
template <int E>
__global__ __launch_bounds__(RC_THREADS) void k_reorth_coop(int64_t n, float *__restrict__ v, const float *__restrict__ lv, int lim,
                                                             double *__restrict__ part2, double *__restrict__ part_out,
                                                             unsigned *bar, unsigned bar_base, const int *guard) {
  const int G = gridDim.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (guard && *guard) {   // skipped half-step: the counter still advances by what the host has booked for this launch
    if (tid == 0 && G > 1) __hip_atomic_fetch_add(bar, (unsigned)lim, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  __shared__ double s_w[RC_THREADS / 64];
  __shared__ float s_d;
  float vr[E];
  int64_t idx[E];
#pragma unroll
  for (int e = 0; e < E; e++) {
    idx[e] = (int64_t)blockIdx.x * RC_THREADS + tid + (int64_t)e * G * RC_THREADS;
    vr[e] = idx[e] < n ? v[idx[e]] : 0.0f;
  }
  for (int q = 0; q < lim; q++) {
    float lr[E];
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < E; e++) {
      lr[e] = idx[e] < n ? lv[(size_t)q * n + idx[e]] : 0.0f;
      acc += (double)vr[e] * lr[e];
    }
    acc = wave_sum(acc);
    if (lane == 0) s_w[w] = acc;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int i = 0; i < RC_THREADS / 64; i++) t += s_w[i];
      part2[(size_t)(q & 1) * RC_BLOCKS + blockIdx.x] = t;
      if (G > 1) {   // (one release fence, relaxed polls, one acquire fence: an acquiring load per poll invalidates caches every time)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = bar_base + (unsigned)(q + 1) * (unsigned)G;
        while ((int)(__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
    }
    __syncthreads();
    if (tid < 64) {   // the dot: the blocks' partials in block order, by every block alike
      double t = 0.0;
      for (int i = lane; i < G; i += 64) t += part2[(size_t)(q & 1) * RC_BLOCKS + i];
      t = wave_sum(t);
      if (tid == 0) s_d = (float)t;
    }
    __syncthreads();
    const float d = s_d;
#pragma unroll
    for (int e = 0; e < E; e++) vr[e] = vr[e] - d * lr[e];
  }
  double sq = 0.0;
#pragma unroll
  for (int e = 0; e < E; e++)
    if (idx[e] < n) {
      v[idx[e]] = vr[e];
      sq += (double)vr[e] * vr[e];
    }
  sq = wave_sum(sq);
  __syncthreads();
  if (lane == 0) s_w[w] = sq;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int i = 0; i < RC_THREADS / 64; i++) t += s_w[i];
    part_out[blockIdx.x] = t;
  }
}
// alpha = ||v|| (:499); v /= alpha; rotations; hbar = h - f1*hbar ; x += f2*hbar ; h = v - f3*h (:539-541); partials of ||x||^2.
// Every block evaluates the recurrences from the (read-only here) state; k_tests commits them.
__global__ void k_alpha_update(int64_t n, float *v, float *h, float *hbar, float *x, const double *part, int np,
                               double *partx, LsmrState *S) {
  if (S->stop) return;
  __shared__ float s_f[4];
  const bool half2 = !S->stop2;                  // beta > 0: v was renewed and alpha with it (else both keep their values)
  const double t = half2 ? block_total(part, np) : 0.0;
  if (threadIdx.x == 0) {
    LsmrState st = *S;
    const float alpha = half2 ? (float)sqrt(t) : st.alpha;
    float f1, f2, f3;
    lsmr_recur(st, alpha, st.beta, f1, f2, f3);
    s_f[0] = f1; s_f[1] = f2; s_f[2] = f3; s_f[3] = alpha;
    if (blockIdx.x == 0) S->alpha_new = alpha;
  }
  __syncthreads();
  const float f1 = s_f[0], f2 = s_f[1], f3 = s_f[2], alpha = s_f[3];
  const bool scal = half2 && alpha > 0.0f;
  const float a = scal ? 1.0f / alpha : 1.0f;
  double sq = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB) {
    float vi = v[i];
    if (scal) {
      vi = a * vi;
      v[i] = vi;
    }
    const float hb = h[i] - f1 * hbar[i];
    const float xn = x[i] + f2 * hb;
    hbar[i] = hb;
    x[i] = xn;
    h[i] = vi - f3 * h[i];
    sq += (double)xn * xn;
  }
  block_partial(sq, partx);
}
// normx (:590), stopping tests (:595-616), commit of the state, one trace record per iteration (the columns of the
// reference's iteration log, format 1500 at :679, plus test3 and rtol which decide whether the line is printed)
__global__ void k_tests(const double *partx, int npx, const float *x, LsmrState *S, dazim_lsmr_rec *trace, int trace_cap) {
  if (S->stop) return;
  const double t = block_total(partx, npx);
  if (threadIdx.x != 0) return;
  LsmrState st = *S;
  float f1, f2, f3;
  lsmr_recur(st, st.alpha_new, st.beta, f1, f2, f3);
  const float normx = (float)sqrt(t);
  st.normx = normx;
  const float test1 = st.normr / st.normb, test2 = st.normAr / (st.normA * st.normr), test3 = 1.0f / st.condA;
  const float t1 = test1 / (1.0f + st.normA * normx / st.normb);
  const float rtol = st.btol + st.atol * st.normA * normx / st.normb;
  int istop = 0;
  if (st.itn >= st.itnlim) istop = 7;
  if (1.0f + test3 <= 1.0f) istop = 6;
  if (1.0f + test2 <= 1.0f) istop = 5;
  if (1.0f + t1 <= 1.0f) istop = 4;
  if (test3 <= st.ctol) istop = 3;
  if (test2 <= st.atol) istop = 2;
  if (test1 <= rtol) istop = 1;
  st.istop = istop;
  st.stop = istop != 0;
  st.stop2 = st.stop;
  if (trace && st.itn < trace_cap) {
    dazim_lsmr_rec r;
    r.itn = st.itn; r.x1 = x[0]; r.normr = st.normr; r.normAr = st.normAr; r.test1 = test1; r.test2 = test2;
    r.test3 = test3; r.rtol = rtol; r.normA = st.normA; r.condA = st.condA;
    trace[st.itn] = r;
  }
  *S = st;
}
__global__ void k_scale_rows(int64_t nrows, const int64_t *ptr, float *val, const float *w) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int64_t r = (int64_t)blockIdx.x * WPB + wv; r < nrows; r += (int64_t)gridDim.x * WPB) {
    const float a = w[r];
    for (int64_t i = ptr[r] + lane; i < ptr[r + 1]; i += 64) val[i] *= a;
  }
}
// out[col[i]] += |val[i]|: the reference's DWS, norm(col(i))=norm(col(i))+abs(rw(i)), inv/Main_Jt.f90:477-481
// Accumulated in 64-bit fixed point (integer addition is associative: the result does not depend on the order in which the
// atomics land, so DWS is reproducible run to run), then converted.
__global__ void k_col_abs_sums(int64_t n, const int *col, const float *val, double scale, unsigned long long *acc) {
  for (int64_t i = (int64_t)blockIdx.x * VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB)
    atomicAdd(&acc[col[i]], (unsigned long long)__double2ll_rn((double)fabsf(val[i]) * scale));
}
__global__ void k_fixed_to_float(int64_t n, const unsigned long long *acc, double inv_scale, float *out) {
  for (int64_t i = (int64_t)blockIdx.x * VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB) out[i] = (float)((double)acc[i] * inv_scale);
}
__global__ void k_gather_f(int64_t n, const unsigned *perm, const float *src, float *dst) {
  for (int64_t i = (int64_t)blockIdx.x * VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB) dst[i] = src[perm[i]];
}
__global__ void k_gather_i(int64_t n, const unsigned *perm, const int *src, int *dst, int add) {
  for (int64_t i = (int64_t)blockIdx.x * VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB) dst[i] = src[perm ? perm[i] : (unsigned)i] + add;
}
__global__ void k_iota_keys(int64_t n, const int *one_based, unsigned *keys, unsigned *iota) {
  for (int64_t i = (int64_t)blockIdx.x * VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB) {
    if (keys) keys[i] = (unsigned)(one_based[i] - 1);
    iota[i] = (unsigned)i;
  }
}
// ptr[r] = first position in the sorted key array whose key >= r  (r = 0..nrows)
__global__ void k_lower_bound(int64_t nrows, int64_t nnz, const unsigned *keys, int64_t *ptr) {
  for (int64_t r = (int64_t)blockIdx.x * VB + threadIdx.x; r <= nrows; r += (int64_t)gridDim.x * VB) {
    int64_t lo = 0, hi = nnz;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)keys[mid] < r) lo = mid + 1; else hi = mid;
    }
    ptr[r] = lo;
  }
}
__global__ void k_check_range(int64_t n, const int *a, int lo, int hi, int *bad) {
  for (int64_t i = (int64_t)blockIdx.x * VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB)
    if (a[i] < lo || a[i] > hi) atomicOr(bad, 1);
}

int spmv_blocks(dazim_ctx *ctx, int64_t nrows, int64_t nx);
// ---- A^T*y in scatter form ---------------------------------------------------------------------
// The gather form (one wavefront per CSC column) is bound by one cache-line fetch of y per entry.
// The scatter form streams the CSR rows instead -- y[r] is a per-row scalar, no gather at all -- and
// accumulates val*y[r] into per-column accumulators in LDS.  To stay reproducible the accumulators
// are 64-bit fixed point (integer addition is associative, so the order in which wavefronts arrive
// does not matter; the quantum is 2^-40 of the largest |val*y|, far below fp32 round-off).  Columns
// are split into blocks that fit LDS; a workgroup owns (row chunk, column block) and per-chunk
// partials are combined in a second, equally order-free, pass.
constexpr int SCW = 16;             // wavefronts per workgroup
constexpr int CBW_MAX = 19 * 1024;  // int64 accumulators per column block (152 KB)

__global__ void k_colblock_ptr(int64_t nrows, int ncb, int cbw, const int64_t *__restrict__ ptr,
                               const int *__restrict__ col, int64_t *__restrict__ cbptr) {
  const int64_t t = (int64_t)blockIdx.x * VB + threadIdx.x;
  if (t >= nrows * (ncb + 1)) return;
  const int64_t r = t / (ncb + 1);
  const int b = (int)(t - r * (ncb + 1));
  int64_t lo = ptr[r], hi = ptr[r + 1];
  const int target = b * cbw;  // first entry with col >= target
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (col[mid] < target) lo = mid + 1; else hi = mid;
  }
  cbptr[t] = lo;
}
// 1 + the last row with at least `thresh` entries (0: none): behind it the matrix has only short rows
constexpr int SPLIT_SHORT = 64;
__global__ void k_last_long_row(int64_t nrows, const int64_t *__restrict__ ptr, int thresh, unsigned long long *res) {
  unsigned long long best = 0;
  for (int64_t r = (int64_t)blockIdx.x * VB + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * VB)
    if (ptr[r + 1] - ptr[r] >= thresh) best = (unsigned long long)(r + 1);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long b = __shfl_xor(best, o);
    best = b > best ? b : best;
  }
  if ((threadIdx.x & 63) == 0 && best) atomicMax(res, best);
}
constexpr int APART = 2048;   // partial maxima (enough workgroups to stream at HBM rate)
// 32-bit -> 16-bit column indices, four per thread step; mod > 0: relative to the pair of column blocks (column mod 2*cbw)
__global__ void k_narrow_cols(int64_t n, const int *__restrict__ col, unsigned short *__restrict__ col16, int mod) {
  for (int64_t i = ((int64_t)blockIdx.x * VB + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * VB * 4) {
    if (i + 3 < n) {
      int4 c = *reinterpret_cast<const int4 *>(col + i);
      if (mod > 0) { c.x %= mod; c.y %= mod; c.z %= mod; c.w %= mod; }
      *reinterpret_cast<ushort4 *>(col16 + i) = make_ushort4((unsigned short)c.x, (unsigned short)c.y, (unsigned short)c.z, (unsigned short)c.w);
    } else {
      for (int64_t j = i; j < n; j++) col16[j] = (unsigned short)(mod > 0 ? col[j] % mod : col[j]);
    }
  }
}
__global__ void k_absmax(int64_t n, const float *x, float *part) {
  float v = 0.0f;
  const int64_t n4 = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? n / 4 : 0;   // 16-byte loads when the array allows
  for (int64_t i = (int64_t)blockIdx.x * VB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * VB) {
    const float4 q = reinterpret_cast<const float4 *>(x)[i];
    v = fmaxf(fmaxf(v, fmaxf(fabsf(q.x), fabsf(q.y))), fmaxf(fabsf(q.z), fabsf(q.w)));
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB) v = fmaxf(v, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  __shared__ float s[VB / 64];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.0f;
    for (int i = 0; i < VB / 64; i++) t = fmaxf(t, s[i]);
    part[blockIdx.x] = t;
  }
}
__global__ void k_absmax_finish(const float *part, int np, float *res) {
  float v = 0.0f;
  for (int i = threadIdx.x; i < np; i += 64) v = fmaxf(v, part[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  if (threadIdx.x == 0) res[0] = v;
}

// ---- software-pipelined walk over CSR row segments (A^T*y scatter and column-blocked A*x) ----
// These kernels are bound by the dependent loads of a row (its pointers -> its entries -> the arithmetic), not by bandwidth, so
// each lane group keeps three rows in flight: while the entries of row i are consumed, those of row i+1 are arriving and the
// pointers of row i+2 are requested.  All prefetch loads are unconditional (a lane with nothing to fetch reads entry 0), so that
// the compiler can count them and wait only for the oldest ones; rows with more than 4*NG*GL aligned entries finish in a plain
// loop.  GL = lanes that share one row (64, or 16 for short segments: four rows per wavefront at a time); rows r_first,
// r_first + stride, ... below nrows are visited; ptrf(r) -> {s, e, aux} gives the entry range and one per-row float,
// elemf(column, value, aux) is called for every entry (head, aligned groups, tail: the order of a plain loop over this lane's
// entries), endf(r, aux) once per row.
struct RowPtr { int64_t s, e; float aux; };
template <int GL, int NG, class IT, class PtrF, class ElemF, class EndF>
__device__ __forceinline__ void walk_rows(int64_t r_first, int64_t stride, int64_t nrows, int lane, const IT *__restrict__ col,
                                          const float *__restrict__ val, PtrF ptrf, ElemF elemf, EndF endf) {
  using I4 = typename Idx4<IT>::type;
  struct Ent { int64_t r, s4, e4; float aux; float4 v[NG]; I4 k[NG]; float hv, tv; int hk, tk; bool hh, ht; };
  auto load_ptr = [&](int64_t r) {
    RowPtr p = ptrf(r < nrows ? r : 0);                // (a dummy read past the end)
    if (r >= nrows) p.e = p.s;                         // nothing to do
    return p;
  };
  auto issue = [&](int64_t r, const RowPtr &p) {
    Ent t;
    t.r = r;
    t.aux = p.aux;
    t.s4 = (p.s + 3) & ~(int64_t)3;
    if (t.s4 > p.e) t.s4 = p.e;
    t.e4 = t.s4 + ((p.e - t.s4) & ~(int64_t)3);
#pragma unroll
    for (int g = 0; g < NG; g++) {
      const int64_t i = t.s4 + 4 * lane + (int64_t)g * 4 * GL;
      const int64_t j = i < t.e4 ? i : 0;
      t.v[g] = *reinterpret_cast<const float4 *>(val + j);
      t.k[g] = *reinterpret_cast<const I4 *>(col + j);
    }
    const int64_t ih = p.s + lane, it = t.e4 + lane;   // <= 3 unaligned entries at either end
    t.hh = ih < t.s4;
    t.ht = it < p.e;
    t.hv = val[t.hh ? ih : 0];
    t.hk = (int)col[t.hh ? ih : 0];
    t.tv = val[t.ht ? it : 0];
    t.tk = (int)col[t.ht ? it : 0];
    return t;
  };
  auto consume = [&](const Ent &t) {
    if (t.hh) elemf(t.hk, t.hv, t.aux);
#pragma unroll
    for (int g = 0; g < NG; g++) {
      const int64_t i = t.s4 + 4 * lane + (int64_t)g * 4 * GL;
      if (i < t.e4) {
        elemf((int)t.k[g].x, t.v[g].x, t.aux); elemf((int)t.k[g].y, t.v[g].y, t.aux);
        elemf((int)t.k[g].z, t.v[g].z, t.aux); elemf((int)t.k[g].w, t.v[g].w, t.aux);
      }
    }
    for (int64_t i = t.s4 + 4 * lane + (int64_t)NG * 4 * GL; i < t.e4; i += 4 * GL) {   // long rows
      const float4 v = *reinterpret_cast<const float4 *>(val + i);
      const I4 k = *reinterpret_cast<const I4 *>(col + i);
      elemf((int)k.x, v.x, t.aux); elemf((int)k.y, v.y, t.aux); elemf((int)k.z, v.z, t.aux); elemf((int)k.w, v.w, t.aux);
    }
    if (t.ht) elemf(t.tk, t.tv, t.aux);
    endf(t.r, t.aux);
  };
  int64_t r = r_first;
  Ent eC = issue(r, load_ptr(r));
  RowPtr pB = load_ptr(r + stride);
  for (; r < nrows; r += stride) {   // (rows differ per 16-lane group when GL = 16: the groups simply diverge at the end)
    const RowPtr pA = load_ptr(r + 2 * stride);
    const Ent eB = issue(r + stride, pB);
    consume(eC);
    eC = eB;
    pB = pA;
  }
}

// A^T*y: GL lanes share one (row, column block) segment and walk it with walk_rows.
// Row steps are dealt round-robin over all wavefronts of the launch (step k goes to workgroup k mod nchunk): G's ray rows hold
// hundreds to thousands of entries and its Tikhonov rows seven, so contiguous chunks of equal rows or equal entries leave CUs
// idle, while a counter that hands rows out dynamically costs more in same-address atomics than it saves (both measured).
// (The same pipeline applied to the LDS-staged A*x kernel made it 2 % slower -- that kernel streams whole rows and is bandwidth
// bound already -- and pipelining fixed-size segments instead of rows made this one 6 % slower; same-box A/B runs.)
template <int GL, int NG, class IT>
__global__ __launch_bounds__(64 * SCW) void spmvT_scatter(int64_t nsplit, int64_t nrows, int nchunk, int ncb, int cbw, int64_t ncols,
                                                          const int64_t *__restrict__ cbptr, const IT *__restrict__ col,
                                                          const float *__restrict__ val, const float *__restrict__ y,
                                                          double scale, long long *__restrict__ part, const int *__restrict__ guard,
                                                          int pairlocal) {
  extern __shared__ __attribute__((aligned(16))) long long acc[];
  if (guard && *guard) return;
  const int chunk = blockIdx.x / ncb, cb = blockIdx.x - chunk * ncb;
  const int c0 = cb * cbw;
  const int csub = pairlocal ? (cb & 1) * cbw : c0;    // what to take off a stored index to get the accumulator
  const int width = (int)((ncols - c0) < cbw ? (ncols - c0) : cbw);
  for (int i = threadIdx.x; i < width; i += 64 * SCW) acc[i] = 0;
  __syncthreads();
  constexpr int RPWV = 64 / GL;                        // rows per wavefront and step
  const int lane = (threadIdx.x & 63) % GL, grp = (threadIdx.x & 63) / GL;
  // fixed point by the magic-number trick: for |t| < 2^51, the low mantissa bits of t + 1.5*2^52 hold round-to-nearest-even(t);
  // scale is a power of two, so the fused multiply-add rounds exactly like (v*y*scale) + magic would
  constexpr double MAGIC = 6755399441055744.0;
  // (two walks, round 5: the long rows [0, nsplit) with GL lanes per row, the short tail [nsplit, nrows) -- G's seven-entry
  // regularisation rows behind its ray rows -- four rows per wavefront; one launch, one set of accumulators)
  auto ptrf = [&](int64_t r) { return RowPtr{cbptr[r * (ncb + 1) + cb], cbptr[r * (ncb + 1) + cb + 1], y[r]}; };
  auto elemf = [&](int c, float v, float yr) {
    atomicAdd((unsigned long long *)&acc[c - csub],
              (unsigned long long)(__double_as_longlong(fma((double)(v * yr), scale, MAGIC)) - __double_as_longlong(MAGIC)));
  };
  // (contiguous row ranges of equal entry counts per wavefront instead of this round-robin deal were measured in round 5: A*x 83 ->
  // 108 us, A^T*y 103 -> 135 us on test4_Yunnan's system -- dealt round-robin, the wavefronts of the launch stream one moving
  // window of the matrix together, which the memory system likes better than 2 048 separate streams)
  walk_rows<GL, NG, IT>(((int64_t)chunk + (int64_t)nchunk * (threadIdx.x >> 6)) * RPWV + grp, (int64_t)nchunk * SCW * RPWV, nsplit, lane,
                        col, val, ptrf, elemf, [](int64_t, float) {});
  if (nsplit < nrows) {
    constexpr int GS = 16, RS = 64 / GS;
    const int lane_s = (threadIdx.x & 63) % GS, grp_s = (threadIdx.x & 63) / GS;
    walk_rows<GS, 1, IT>(nsplit + ((int64_t)chunk + (int64_t)nchunk * (threadIdx.x >> 6)) * RS + grp_s, (int64_t)nchunk * SCW * RS, nrows,
                         lane_s, col, val, ptrf, elemf, [](int64_t, float) {});
  }
  __syncthreads();
  long long *dst = part + (size_t)chunk * ncols + c0;
  for (int i = threadIdx.x; i < width; i += 64 * SCW) dst[i] = acc[i];
}
// out[c] = beta*out[c] + sum_chunks part[chunk][c] / scale ; partial ||out||^2
__global__ void k_scatter_combine(int64_t ncols, int nchunk, const long long *__restrict__ part, double inv_scale,
                                  float *__restrict__ out, const float *__restrict__ beta_p, float beta_sign,
                                  double *__restrict__ sumsq, const int *__restrict__ guard) {
  if (guard && *guard) return;
  const float beta = beta_p ? beta_sign * beta_p[0] : beta_sign;
  double sq = 0.0;
  for (int64_t c = (int64_t)blockIdx.x * VB + threadIdx.x; c < ncols; c += (int64_t)gridDim.x * VB) {
    // eight independent partial sums: the loads of a column are in flight together (integer sums: any order gives the same bits)
    long long t8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int k = 0;
    for (; k + 8 <= nchunk; k += 8)
#pragma unroll
      for (int j = 0; j < 8; j++) t8[j] += part[(size_t)(k + j) * ncols + c];
    for (; k < nchunk; k++) t8[0] += part[(size_t)k * ncols + c];
    const long long t = ((t8[0] + t8[1]) + (t8[2] + t8[3])) + ((t8[4] + t8[5]) + (t8[6] + t8[7]));
    const float o = beta * out[c] + (float)((double)t * inv_scale);
    out[c] = o;
    sq += (double)o * o;
  }
  if (sumsq) block_partial(sq, sumsq);
}

// A*x for matrices whose x does not fit the LDS (n > 38 K: joint inversions, S-512): the columns are cut at the same block
// pointers as the scatter kernel, two scatter blocks (<= 38 K floats of x) per workgroup, which removes the per-entry
// cache-line gather that bounds the plain kernel.  A workgroup owns (row set x column-block pair) and writes the partial
// dot product of each of its rows to part[pair][row]; k_rows_combine adds the pairs in a fixed order.
template <int GL, int NG, class IT>
__global__ __launch_bounds__(64 * SCW) void spmv_rows_blocked(int64_t nsplit, int64_t nrows, int nset, int npair, int ncb, int cbw, int64_t ncols,
                                                              const int64_t *__restrict__ cbptr, const IT *__restrict__ col,
                                                              const float *__restrict__ val, const float *__restrict__ x,
                                                              float *__restrict__ part, const int *__restrict__ guard, int pairlocal) {
  extern __shared__ __attribute__((aligned(16))) float xblk[];
  if (guard && *guard) return;
  const int set = blockIdx.x / npair, pr = blockIdx.x - set * npair;
  const int cb0 = 2 * pr, cb1 = (cb0 + 2 < ncb) ? cb0 + 2 : ncb;
  const int c0 = cb0 * cbw;
  const int csub = pairlocal ? 0 : c0;                 // (16-bit indices of large matrices are already relative to the pair)
  const int width = (int)((ncols - c0) < 2 * (int64_t)cbw ? (ncols - c0) : 2 * (int64_t)cbw);
  for (int i = threadIdx.x; i < width; i += 64 * SCW) xblk[i] = x[c0 + i];
  __syncthreads();
  constexpr int RPWV = 64 / GL;
  const int lane = (threadIdx.x & 63) % GL, grp = (threadIdx.x & 63) / GL;
  float acc = 0.0f;
  float *dst = part + (size_t)pr * nrows;
  auto ptrf = [&](int64_t r) { return RowPtr{cbptr[r * (ncb + 1) + cb0], cbptr[r * (ncb + 1) + cb1], 0.0f}; };
  auto elemf = [&](int c, float v, float) { acc += v * xblk[c - csub]; };
  walk_rows<GL, NG, IT>(((int64_t)set + (int64_t)nset * (threadIdx.x >> 6)) * RPWV + grp, (int64_t)nset * SCW * RPWV, nsplit, lane, col, val,
                        ptrf, elemf, [&](int64_t r, float) {
                          float a = acc;
#pragma unroll
                          for (int o = GL / 2; o > 0; o >>= 1) a += __shfl_xor(a, o);
                          if (lane == 0 && r < nsplit) dst[r] = a;
                          acc = 0.0f;
                        });
  if (nsplit < nrows) {   // the short tail of the matrix (see spmvT_scatter): four rows per wavefront
    constexpr int GS = 16, RS = 64 / GS;
    const int lane_s = (threadIdx.x & 63) % GS, grp_s = (threadIdx.x & 63) / GS;
    acc = 0.0f;
    walk_rows<GS, 1, IT>(nsplit + ((int64_t)set + (int64_t)nset * (threadIdx.x >> 6)) * RS + grp_s, (int64_t)nset * SCW * RS, nrows, lane_s,
                         col, val, ptrf, elemf, [&](int64_t r, float) {
                           float a = acc;
#pragma unroll
                           for (int o = GS / 2; o > 0; o >>= 1) a += __shfl_xor(a, o);
                           if (lane_s == 0 && r < nrows) dst[r] = a;
                           acc = 0.0f;
                         });
  }
}
// out[r] = beta*out[r] + sum_pairs part[pair][r] ; partial ||out||^2
__global__ void k_rows_combine(int64_t nrows, int npair, const float *__restrict__ part, float *__restrict__ out,
                               const float *__restrict__ beta_p, float beta_sign, double *__restrict__ sumsq,
                               const int *__restrict__ guard) {
  if (guard && *guard) return;
  const float beta = beta_p ? beta_sign * beta_p[0] : beta_sign;
  double sq = 0.0;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x) {
    float t = part[r];
    for (int k = 1; k < npair; k++) t += part[(size_t)k * nrows + r];
    const float o = beta * out[r] + t;
    out[r] = o;
    sq += (double)o * o;
  }
  if (sumsq) block_partial(sq, sumsq);
}

inline int nblk(int64_t n, int cap = 2048) {
  int64_t b = (n + VB - 1) / VB;
  if (b < 1) b = 1;
  return (int)(b > cap ? cap : b);
}

constexpr int64_t LDSX_MAX = 38 * 1024;  // floats of the dense vector that fit the 160 KB LDS next to the reduction scratch
bool use_ldsx(dazim_ctx *ctx, int64_t nrows, int64_t nx) {
  if (ctx->opts.count("spmv.ldsx") && !ctx->opts["spmv.ldsx"]) return false;
  return nx <= LDSX_MAX && nrows >= (int64_t)ctx->num_cu * LWPB * 4;
}
int spmv_blocks(dazim_ctx *ctx, int64_t nrows, int64_t nx = -1) {
  if (nx >= 0 && use_ldsx(ctx, nrows, nx)) return ctx->num_cu;  // one 16-wave workgroup per CU
  int64_t b = (nrows + WPB - 1) / WPB;
  const int64_t cap = (int64_t)ctx->num_cu * 8;  // 8 workgroups (32 waves) per CU, grid-stride the rest
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}
// nx = length of the gathered vector; nblocks must come from spmv_blocks(ctx, nrows, nx)
int launch_spmv(dazim_ctx *ctx, int64_t nrows, int64_t nx, const int64_t *ptr, const int *idx, const float *val,
                const float *x, float *out, const float *beta_p, float beta_sign, double *sumsq,
                int nblocks, const int *guard = nullptr, const unsigned short *idx16 = nullptr) {
  if (use_ldsx(ctx, nrows, nx)) {
    const size_t lds = (size_t)((nx + 3) & ~(int64_t)3) * 4;
    if (idx16) {   // 16-bit column indices: 6 bytes per stored entry
      DZ_HIP(hipFuncSetAttribute((const void *)spmv_rows_ldsx<unsigned short>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(spmv_rows_ldsx<unsigned short>, dim3(nblocks), dim3(64 * LWPB), lds, ctx->stream, nrows, nx, ptr, idx16, val,
                         x, out, beta_p, beta_sign, sumsq, guard);
    } else {
      DZ_HIP(hipFuncSetAttribute((const void *)spmv_rows_ldsx<int>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(spmv_rows_ldsx<int>, dim3(nblocks), dim3(64 * LWPB), lds, ctx->stream, nrows, nx, ptr, idx, val, x, out,
                         beta_p, beta_sign, sumsq, guard);
    }
  } else {
    hipLaunchKernelGGL(spmv_rows, dim3(nblocks), dim3(64 * WPB), 0, ctx->stream, nrows, ptr, idx, val, x, out,
                       beta_p, beta_sign, sumsq, guard);
  }
  DZ_HIP(hipGetLastError());
  return 0;
}

// row id of every CSR entry (one wavefront per row)
__global__ void k_expand_rows(int64_t nrows, const int64_t *ptr, unsigned *rowid) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int64_t r = (int64_t)blockIdx.x * WPB + wv; r < nrows; r += (int64_t)gridDim.x * WPB)
    for (int64_t i = ptr[r] + lane; i < ptr[r + 1]; i += 64) rowid[i] = (unsigned)r;
}
__global__ void k_offset_ptr(int64_t n, const int64_t *src, int64_t add, int64_t *dst) {
  for (int64_t i = (int64_t)blockIdx.x * VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB) dst[i] = src[i] + add;
}

// the CSC copy is only needed by the gather form of A^T*y: it is built on first use
int invalidate_transpose(dazim_csr *A) {
  for (void **pp : {(void **)&A->colptr, (void **)&A->row, (void **)&A->tval, (void **)&A->tperm})
    if (*pp) { (void)hipFree(*pp); *pp = nullptr; }
  return 0;
}
// build the stable transpose (colptr,row,tval,tperm) of A's CSR arrays
int build_transpose(dazim_ctx *ctx, dazim_csr *A) {
  const int64_t nnz = A->nnz, n = A->n, m = A->m;
  const size_t nz = (size_t)(nnz > 0 ? nnz : 1);
  for (void **pp : {(void **)&A->colptr, (void **)&A->row, (void **)&A->tval, (void **)&A->tperm})
    if (*pp) { (void)hipFree(*pp); *pp = nullptr; }
  DZ_HIP(dz_malloc_retry(ctx, (void **)&A->colptr, (n + 1) * 8));
  DZ_HIP(dz_malloc_retry(ctx, (void **)&A->row, nz * 4));
  DZ_HIP(dz_malloc_retry(ctx, (void **)&A->tval, nz * 4));
  DZ_HIP(dz_malloc_retry(ctx, (void **)&A->tperm, nz * 4));
  if (nnz == 0) {
    DZ_HIP(hipMemsetAsync(A->colptr, 0, (n + 1) * 8, ctx->stream));
    return 0;
  }
  int rc;
  void *p;
  unsigned *ck, *cks, *iota, *rowid;
  if ((rc = dz_scratch(ctx, "csr.k0", nz * 4, &p))) return rc;
  ck = (unsigned *)p;
  if ((rc = dz_scratch(ctx, "csr.k1", nz * 4, &p))) return rc;
  cks = (unsigned *)p;
  if ((rc = dz_scratch(ctx, "csr.v0", nz * 4, &p))) return rc;
  iota = (unsigned *)p;
  if ((rc = dz_scratch(ctx, "csr.perm", nz * 4, &p))) return rc;
  rowid = (unsigned *)p;
  const int nb = nblk(nnz);
  int cbits = 1;
  while (((int64_t)1 << cbits) < n) cbits++;
  // keys = column (0-based) + 1 so that k_iota_keys' "-1" applies
  hipLaunchKernelGGL(k_gather_i, dim3(nb), dim3(VB), 0, ctx->stream, nnz, (const unsigned *)nullptr, A->col, (int *)ck, 0);
  hipLaunchKernelGGL(k_iota_keys, dim3(nb), dim3(VB), 0, ctx->stream, nnz, (const int *)nullptr, (unsigned *)nullptr, iota);
  hipLaunchKernelGGL(k_expand_rows, dim3(spmv_blocks(ctx, m, -1)), dim3(64 * WPB), 0, ctx->stream, m, A->rowptr, rowid);
  size_t tb = 0;
  DZ_HIP(rocprim::radix_sort_pairs(nullptr, tb, ck, cks, iota, A->tperm, (size_t)nnz, 0, cbits, ctx->stream));
  void *tmp;
  if ((rc = dz_scratch(ctx, "csr.tmp", tb + 256, &tmp))) return rc;
  DZ_HIP(rocprim::radix_sort_pairs(tmp, tb, ck, cks, iota, A->tperm, (size_t)nnz, 0, cbits, ctx->stream));
  hipLaunchKernelGGL(k_lower_bound, dim3(nblk(n + 1)), dim3(VB), 0, ctx->stream, n, nnz, cks, A->colptr);
  hipLaunchKernelGGL(k_gather_f, dim3(nb), dim3(VB), 0, ctx->stream, nnz, A->tperm, A->val, A->tval);
  hipLaunchKernelGGL(k_gather_i, dim3(nb), dim3(VB), 0, ctx->stream, nnz, A->tperm, (const int *)rowid, A->row, 0);
  DZ_HIP(hipGetLastError());
  return 0;
}

// column-block pointers + max|val| for the scatter form of A^T*y (needs canonical CSR)
// changed_from: first entry whose column index is new (0: all of them; < 0: only values changed, e.g. row scaling) -- the
// 16-bit copy of the column indices is extended / kept accordingly
int build_colblocks(dazim_ctx *ctx, dazim_csr *A, int64_t changed_from = 0) {
  if (A->cbptr) { dz_big_put(ctx, A->cbptr); A->cbptr = nullptr; }
  A->ncb = (int)((A->n + CBW_MAX - 1) / CBW_MAX);
  A->cbw = (int)(((A->n + A->ncb - 1) / A->ncb + 3) & ~3ll);
  const int mod16 = A->n <= 65536 ? 0 : 2 * A->cbw;       // 16-bit indices: the column, or the column within its block pair
  const bool want16 = mod16 <= 65536 && A->nnz > 0 && !(ctx->opts.count("spmv.col16") && !ctx->opts["spmv.col16"]);
  if (A->col16 && (!want16 || A->col16_cap < A->nnz || changed_from == 0 || A->col16_mod != mod16)) {
    dz_big_put(ctx, A->col16);
    A->col16 = nullptr;
    A->col16_cap = 0;
  }
  if (want16 && (!A->col16 || changed_from >= 0)) {
    int64_t from = 0;
    if (!A->col16) {
      A->col16_cap = ((A->cap_nnz > A->nnz ? A->cap_nnz : A->nnz) + 3) & ~(int64_t)3;
      { void *pp; int rcp = dz_big_get(ctx, (size_t)A->col16_cap * 2, &pp); if (rcp) return rcp; A->col16 = (unsigned short *)pp; }
      A->col16_mod = mod16;
    } else {
      from = changed_from & ~(int64_t)3;
    }
    const int64_t cnt = A->nnz - from;
    if (cnt > 0)
      hipLaunchKernelGGL(k_narrow_cols, dim3(nblk((cnt + 3) / 4)), dim3(VB), 0, ctx->stream, cnt, A->col + from, A->col16 + from, mod16);
    DZ_HIP(hipGetLastError());
  }
  const int64_t np = A->m * (A->ncb + 1);
  if (np > 0) {   // a matrix without rows (an empty ray batch) has no block pointers
    { void *pp; int rcp = dz_big_get(ctx, (size_t)np * 8, &pp); if (rcp) return rcp; A->cbptr = (int64_t *)pp; }
    hipLaunchKernelGGL(k_colblock_ptr, dim3((unsigned)((np + VB - 1) / VB)), dim3(VB), 0, ctx->stream, A->m, A->ncb, A->cbw,
                       A->rowptr, A->col, A->cbptr);
    DZ_HIP(hipGetLastError());
  }
  int rc;
  void *p;
  if ((rc = dz_scratch(ctx, "csr.absmax", (APART + 4) * 4, &p))) return rc;
  float *pm = (float *)p;
  A->vmax = 0.0f;
  A->split_row = A->m;
  A->long_avg = A->m > 0 ? (double)A->nnz / (double)A->m : 0.0;
  if (A->m > 0 && A->nnz > 0) {   // where the short tail of the matrix begins (see dazim_csr::split_row)
    unsigned long long *d_last = reinterpret_cast<unsigned long long *>(pm + APART + 2), h_last = 0;
    DZ_HIP(hipMemsetAsync(d_last, 0, 8, ctx->stream));
    hipLaunchKernelGGL(k_last_long_row, dim3(nblk(A->m)), dim3(VB), 0, ctx->stream, A->m, A->rowptr, SPLIT_SHORT, d_last);
    DZ_HIP(hipMemcpyAsync(&h_last, d_last, 8, hipMemcpyDeviceToHost, ctx->stream));
    DZ_HIP(hipStreamSynchronize(ctx->stream));
    if ((int64_t)h_last < A->m && (int64_t)h_last > 0) {
      int64_t at = 0;
      DZ_HIP(hipMemcpy(&at, A->rowptr + h_last, 8, hipMemcpyDeviceToHost));
      A->split_row = (int64_t)h_last;
      A->long_avg = (double)at / (double)h_last;
    }
  }
  if (A->nnz > 0) {
    const int nb = nblk((A->nnz + 3) / 4, APART);
    hipLaunchKernelGGL(k_absmax, dim3(nb), dim3(VB), 0, ctx->stream, A->nnz, A->val, pm);
    hipLaunchKernelGGL(k_absmax_finish, dim3(1), dim3(64), 0, ctx->stream, pm, nb, pm + APART);
    DZ_HIP(hipMemcpyAsync(&A->vmax, pm + APART, 4, hipMemcpyDeviceToHost, ctx->stream));
  }
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}
bool use_scatter(dazim_ctx *ctx, const dazim_csr *A) {
  if (ctx->opts.count("spmv.scatter") && !ctx->opts["spmv.scatter"]) return false;
  return A->cbptr && A->nnz >= (1 << 22) && A->m >= (int64_t)ctx->num_cu * SCW;
}
// x(out, n) = beta*x + A^T y ; returns the number of ||out||^2 partials written to sumsq in *npart
int launch_spmvT(dazim_ctx *ctx, const dazim_csr *A, const float *y, float ymax, float *out, const float *beta_p,
                 float beta_sign, double *sumsq, int *npart, const int *guard = nullptr) {
  const double pm = (double)A->vmax * (double)ymax;
  ctx->ksec["spmvt.kind"] = (use_scatter(ctx, A) && std::isfinite(pm)) ? 1 : 0;
  ctx->ksec["spmvt.idx_bytes"] = (A->col16 && use_scatter(ctx, A) && std::isfinite(pm)) ? 2 : 4;
  // non-finite values (NaN / Inf in G or y) cannot be put on the fixed-point grid: the gather form propagates them like
  // the reference's plain loop would
  if (!use_scatter(ctx, A) || !std::isfinite(pm)) {
    if (!A->colptr) {
      int rc0 = build_transpose(ctx, const_cast<dazim_csr *>(A));
      if (rc0) return rc0;
    }
    const int gn = spmv_blocks(ctx, A->n, A->m);
    if (npart) *npart = gn;
    return launch_spmv(ctx, A->n, A->m, A->colptr, A->row, A->tval, y, out, beta_p, beta_sign, sumsq, gn, guard);
  }
  int nchunk = ctx->num_cu / A->ncb;
  if (nchunk < 1) nchunk = 1;
  int rc;
  void *p;
  if ((rc = dz_scratch(ctx, "spmvt.part", (size_t)nchunk * A->n * 8, &p))) return rc;
  long long *part = (long long *)p;
  int e = 0;
  if (pm > 0) (void)frexp(pm, &e);   // pm = f * 2^e, 0.5 <= f < 1  ->  pm < 2^e
  // fractional bits: every term is below 2^fb in fixed point and a column holds at most m of them, so the int64 sum needs
  // fb + ceil(log2 m) <= 62.  40 bits up to 4 M rows (quantum 2^-40 of the largest term), fewer beyond (still < fp32 round-off)
  int lgm = 0;
  while (((int64_t)1 << lgm) < A->m) lgm++;
  const int fb = 62 - lgm < 40 ? 62 - lgm : 40;
  const double scale = ldexp(1.0, fb - e);
  const size_t lds = (size_t)A->cbw * 8;
  // short (row, column block) segments: four rows per wavefront (16 lanes each), else a whole wavefront per row
  // (entries per segment of the LONG rows: the short tail, if any, is walked four rows per wavefront anyway -- test4_Yunnan's joint
  // matrix is 20 877 ray rows of 2 548 entries and 73 440 regularisation rows of seven, 284 per segment on average, 637 in the ray rows)
  const bool split = !(ctx->opts.count("spmv.split") && !ctx->opts["spmv.split"]) && A->split_row < A->m;
  const int64_t nsplit = split ? A->split_row : A->m;
  bool shortseg = (split ? A->long_avg : (double)A->nnz / (double)A->m) < 400.0 * A->ncb;   // measured: 16 lanes win at 141 and 296 entries per segment, 64 at 553
  if (ctx->opts.count("spmv.gl16") && ctx->opts["spmv.gl16"] >= 0) shortseg = ctx->opts["spmv.gl16"] != 0;
  ctx->ksec["spmvt.split_row"] = (double)nsplit;
  const dim3 sgrid(nchunk * A->ncb), sblock(64 * SCW);
#define DZ_SCATTER(GL_, NG_, IT_, COLP_)                                                                                        \
  do {                                                                                                                          \
    DZ_HIP(hipFuncSetAttribute((const void *)spmvT_scatter<GL_, NG_, IT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((spmvT_scatter<GL_, NG_, IT_>), sgrid, sblock, lds, ctx->stream, nsplit, A->m, nchunk, A->ncb, A->cbw, A->n, \
                       A->cbptr, COLP_, A->val, y, scale, part, guard, (sizeof(IT_) == 2 && A->col16_mod > 0) ? 1 : 0);               \
  } while (0)
  // (more or fewer 256-entry groups in flight per row segment -- 3 or 6 instead of 4 -- measured in round 5: no difference)
  if (A->col16) {   // 16-bit column indices: 6 bytes per stored entry
    if (shortseg) DZ_SCATTER(16, 2, unsigned short, A->col16); else DZ_SCATTER(64, 4, unsigned short, A->col16);
  } else {
    if (shortseg) DZ_SCATTER(16, 2, int, A->col); else DZ_SCATTER(64, 4, int, A->col);
  }
#undef DZ_SCATTER
  const int nb = nblk(A->n, NPART);
  hipLaunchKernelGGL(k_scatter_combine, dim3(nb), dim3(VB), 0, ctx->stream, A->n, nchunk, part, 1.0 / scale, out, beta_p,
                     beta_sign, sumsq, guard);
  DZ_HIP(hipGetLastError());
  if (npart) *npart = nb;
  return 0;
}

bool use_blocked(dazim_ctx *ctx, const dazim_csr *A) {
  if (ctx->opts.count("spmv.blocked") && !ctx->opts["spmv.blocked"]) return false;
  return A->cbptr && A->n > LDSX_MAX && A->nnz >= (1 << 22) && A->m >= (int64_t)ctx->num_cu * SCW;
}
// y(out, m) = beta*y + A x ; the number of ||out||^2 partials written to sumsq goes to *npart
int launch_spmvA(dazim_ctx *ctx, const dazim_csr *A, const float *x, float *out, const float *beta_p, float beta_sign,
                 double *sumsq, int *npart, const int *guard = nullptr) {
  ctx->ksec["spmv.kind"] = use_blocked(ctx, A) ? 2 : (use_ldsx(ctx, A->m, A->n) ? 1 : 0);
  // index bytes streamed per entry (the whole-x LDS kernel needs the column itself, the blocked one takes either form)
  ctx->ksec["spmv.idx_bytes"] = (A->col16 && (use_blocked(ctx, A) || (use_ldsx(ctx, A->m, A->n) && A->col16_mod == 0))) ? 2 : 4;
  if (!use_blocked(ctx, A)) {
    const int gm = spmv_blocks(ctx, A->m, A->n);
    if (npart) *npart = gm;
    return launch_spmv(ctx, A->m, A->n, A->rowptr, A->col, A->val, x, out, beta_p, beta_sign, sumsq, gm, guard,
                       A->col16_mod == 0 ? A->col16 : nullptr);
  }
  const int npair = (A->ncb + 1) / 2;
  int nset = ctx->num_cu / npair;
  if (nset < 1) nset = 1;
  int rc;
  void *p;
  if ((rc = dz_scratch(ctx, "spmv.part", (size_t)npair * A->m * 4, &p))) return rc;
  float *part = (float *)p;
  const size_t lds = (size_t)A->cbw * 2 * 4;
  const bool split = !(ctx->opts.count("spmv.split") && !ctx->opts["spmv.split"]) && A->split_row < A->m;
  const int64_t nsplit = split ? A->split_row : A->m;
  bool shortseg = (split ? A->long_avg : (double)A->nnz / (double)A->m) < 600.0 * npair;    // measured: 16 lanes win at 282 and 519 entries per segment
  if (ctx->opts.count("spmv.gl16") && ctx->opts["spmv.gl16"] >= 0) shortseg = ctx->opts["spmv.gl16"] != 0;
  ctx->ksec["spmv.split_row"] = (double)nsplit;
  const dim3 bgrid(nset * npair), bblock(64 * SCW);
#define DZ_BLOCKED(GL_, NG_, IT_, COLP_)                                                                                        \
  do {                                                                                                                          \
    DZ_HIP(hipFuncSetAttribute((const void *)spmv_rows_blocked<GL_, NG_, IT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((spmv_rows_blocked<GL_, NG_, IT_>), bgrid, bblock, lds, ctx->stream, nsplit, A->m, nset, npair, A->ncb, A->cbw, A->n, \
                       A->cbptr, COLP_, A->val, x, part, guard, (sizeof(IT_) == 2 && A->col16_mod > 0) ? 1 : 0);                 \
  } while (0)
  if (A->col16) {   // 16-bit column indices: 6 bytes per stored entry
    if (shortseg) DZ_BLOCKED(16, 2, unsigned short, A->col16); else DZ_BLOCKED(64, 4, unsigned short, A->col16);
  } else {
    if (shortseg) DZ_BLOCKED(16, 2, int, A->col); else DZ_BLOCKED(64, 4, int, A->col);
  }
#undef DZ_BLOCKED
  const int nb = nblk(A->m, NPART);
  hipLaunchKernelGGL(k_rows_combine, dim3(nb), dim3(VB), 0, ctx->stream, A->m, npair, part, out, beta_p, beta_sign, sumsq, guard);
  DZ_HIP(hipGetLastError());
  if (npart) *npart = nb;
  return 0;
}


// ---- N4: regularisation rows, data weights and the clamped model update on the device ---------------------------------------
// One thread per regularisation row r = blk*maxvp + cell (cell in the reference's k, j, i loop order, inv/TikhRegul.f90:20-23):
// a cell on a face of the block holds one entry 2w, an interior cell the 7-point stencil 6w, -w x 6 (inv/TikhRegul.f90:24-58).
__device__ __forceinline__ bool tikh_face(int cell, int nvx, int nvz, int nzm1, int &i, int &j, int &k) {
  k = cell / (nvx * nvz);
  const int r = cell - k * nvx * nvz;
  j = r / nvx;
  i = r - j * nvx;
  return i == 0 || i == nvx - 1 || j == 0 || j == nvz - 1 || k == 0 || k == nzm1 - 1;
}
// dazim_csr_threshold: one wavefront per row; entries with |val| > tol keep their order (ballot prefix)
template <bool FILL>
__global__ __launch_bounds__(256) void k_threshold_rows(int64_t m, const int64_t *__restrict__ rowptr, const int *__restrict__ col,
                                                        const float *__restrict__ val, float tol, long *__restrict__ cnt,
                                                        const long *__restrict__ rowptr2, int *__restrict__ col2,
                                                        float *__restrict__ val2) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r > m) return;
  if (r == m) {
    if (!FILL && lane == 0) cnt[m] = 0;
    return;
  }
  const int64_t s = rowptr[r], e = rowptr[r + 1];
  long kept = 0;
  const long o = FILL ? rowptr2[r] : 0;
  for (int64_t i = s; i < e; i += 64) {
    const int64_t k = i + lane;
    float v = 0.0f;
    int c = 0;
    if (k < e) { v = val[k]; c = col[k]; }
    const bool keep = k < e && fabsf(v) > tol;
    const unsigned long long b = __ballot(keep);
    if (FILL && keep) {
      const long q = o + kept + __popcll(b & ((1ull << lane) - 1ull));
      val2[q] = v;
      col2[q] = c;
    }
    kept += __popcll(b);
  }
  if (!FILL && lane == 0) cnt[r] = kept;
}

// (r0: the first regularisation row of this call -- a rank of a row-sharded run appends its share [r0, r0 + nrow) of them)
__global__ void k_tikh_count(int64_t r0, int64_t nrow, int maxvp, int nvx, int nvz, int nzm1, long *cnt) {
  const int64_t r = (int64_t)blockIdx.x * VB + threadIdx.x;
  if (r > nrow) return;
  int i, j, k;
  cnt[r] = r == nrow ? 0 : (tikh_face((int)((r0 + r) % maxvp), nvx, nvz, nzm1, i, j, k) ? 1 : 7);
}
// entries written with ascending columns (the canonical order every other row of the matrix has)
__global__ void k_tikh_fill(int64_t r0, int64_t nrow, int maxvp, int nvx, int nvz, int nzm1, const long *off, int64_t nnz0,
                            const float *__restrict__ w, int64_t *__restrict__ rowptr, int *__restrict__ col,
                            float *__restrict__ val) {
  const int64_t r = (int64_t)blockIdx.x * VB + threadIdx.x;
  if (r > nrow) return;
  rowptr[r] = nnz0 + off[r];
  if (r == nrow) return;
  const int blk = (int)((r0 + r) / maxvp), cell = (int)((r0 + r) - (int64_t)blk * maxvp);
  int i, j, k;
  const bool face = tikh_face(cell, nvx, nvz, nzm1, i, j, k);
  const float wt = w[blk];
  const int c = blk * maxvp + cell;
  const int64_t p = nnz0 + off[r];
  if (face) {
    col[p] = c;
    val[p] = 2.0f * wt;
  } else {
    const int d[7] = {-nvz * nvx, -nvx, -1, 0, 1, nvx, nvz * nvx};
#pragma unroll
    for (int q = 0; q < 7; q++) {
      col[p + q] = c + d[q];
      val[p + q] = q == 3 ? 6.0f * wt : -1.0f * wt;
    }
  }
}
// res = obst - dsyn ; rel = |res / obst|   (inv/Main_Jt.f90:432-435, inv/CalSigamNorm.f90:20-23)
__global__ void k_residual(int64_t n, const float *obst, const float *dsyn, float *res, float *rel) {
  for (int64_t i = (int64_t)blockIdx.x * VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB) {
    const float r = obst[i] - dsyn[i];
    res[i] = r;
    rel[i] = fabsf(r / obst[i]);
  }
}
// meandeltaT and stddeltaT of CalDdatSigma (inv/CalSigamNorm.f90:20-31): two sequential fp32 sums, kept sequential (one lane) so
// that the weights are the reference's bit for bit -- 2 n dependent additions, 0.1 ms at test4's 20 877 rays
__global__ void k_sigma_stats(int64_t n, const float *rel, float *out) {
  // the workgroup stages chunks of rel in LDS (coalesced loads); lane 0 adds them in index order
  constexpr int CH = 8192;
  __shared__ float s_c[CH];
  __shared__ float s_mean;
  float acc = 0.0f;
  for (int pass = 0; pass < 2; pass++) {
    acc = 0.0f;
    const float mean = pass ? s_mean : 0.0f;
    for (int64_t base = 0; base < n; base += CH) {
      const int len = (int)((n - base) < CH ? (n - base) : CH);
      __syncthreads();
      for (int i = threadIdx.x; i < len; i += blockDim.x) s_c[i] = rel[base + i];
      __syncthreads();
      if (threadIdx.x == 0) {
        if (pass == 0)
          for (int i = 0; i < len; i++) acc = acc + s_c[i];
        else
          for (int i = 0; i < len; i++) acc = acc + (s_c[i] - mean) * (s_c[i] - mean);
      }
    }
    if (threadIdx.x == 0) {
      if (pass == 0) {
        s_mean = acc / (float)n;
        out[0] = s_mean;
      } else {
        out[1] = sqrtf(acc / (float)n);
      }
    }
    __syncthreads();
  }
}
// sigmaT (inv/CalSigamNorm.f90:32-40), datweight = 1/sigmaT, cbst = res*datweight (inv/Main_Jt.f90:462-466)
__global__ void k_sigma_weights(int64_t n, const float *obst, const float *res, const float *rel, const float *ms, float *wgt,
                                float *rhs) {
  const float sd = ms[1];
  for (int64_t i = (int64_t)blockIdx.x * VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB) {
    const float ratio = fabsf(rel[i] / (1.5f * sd));
    float sigma = sd * obst[i];
    if (ratio > 1.0f) sigma = sigma * (float)exp((double)(ratio - 1.0f));   // correctly rounded expf like the host libm's
    const float wt = 1.0f / sigma;
    wgt[i] = wt;
    rhs[i] = res[i] * wt;
  }
}
// sums for the log lines: part[b][0..4] = sum res, sum |res|, sum res^2, sum wgt, sum |rhs|
__global__ void k_weight_sums(int64_t n, const float *res, const float *wgt, const float *rhs, double *part) {
  double a[5] = {0, 0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * VB + threadIdx.x; i < n; i += (int64_t)gridDim.x * VB) {
    const double r = res[i];
    a[0] += r; a[1] += fabs(r); a[2] += r * r; a[3] += wgt[i]; a[4] += fabsf(rhs[i]);
  }
  __shared__ double s[5][VB / 64];
#pragma unroll
  for (int q = 0; q < 5; q++) {
    const double t = wave_sum(a[q]);
    if ((threadIdx.x & 63) == 0) s[q][threadIdx.x >> 6] = t;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    double t = 0.0;
    for (int i = 0; i < VB / 64; i++) t += s[threadIdx.x][i];
    part[blockIdx.x * 5 + threadIdx.x] = t;
  }
}
// clamped update of the shear velocities and the Gc, Gs maps (inv/Main_Jt.f90:582-620); one thread per inner cell
__global__ void k_model_update(int nx, int ny, int nzm1, int joint, float *vs, float *dv, float minvel, float maxvel, float *gc,
                               float *gs) {
  const int nvx = nx - 2, nvz = ny - 2, maxvp = nvx * nvz * nzm1;
  const int ii = blockIdx.x * VB + threadIdx.x;
  if (ii >= maxvp) return;
  const int k = ii / (nvx * nvz), r = ii - k * nvx * nvz, j = r / nvx, i = r - j * nvx;
  float p = dv[ii];
  if (p >= 0.500f) p = 0.500f;
  if (p <= -0.500f) p = -0.500f;
  if (fabsf(p) < 1e-5f) p = 0.0f;
  dv[ii] = p;
  const size_t iv = ((size_t)k * ny + (j + 1)) * nx + (i + 1);
  float v = vs[iv] + p;
  if (v < minvel) v = minvel;
  if (v > maxvel) v = maxvel;
  vs[iv] = v;
  if (joint) {
    if (gc) gc[ii] = dv[maxvp + ii];
    if (gs) gs[ii] = dv[2 * maxvp + ii];
  }
}
// per (block, depth) min, max and sum |.| of the update (the log lines of inv/Main_Jt.f90:621-666): one workgroup each
__global__ void k_update_stats(int ncell, const float *dv, float *out) {
  const float *x = dv + (size_t)blockIdx.x * ncell;
  float mn = INFINITY, mx = -INFINITY;
  double sa = 0.0;
  for (int i = threadIdx.x; i < ncell; i += VB) {
    const float v = x[i];
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
    sa += fabsf(v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o));
    mx = fmaxf(mx, __shfl_xor(mx, o));
  }
  sa = wave_sum(sa);
  __shared__ float s_mn[VB / 64], s_mx[VB / 64];
  __shared__ double s_sa[VB / 64];
  if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx; s_sa[threadIdx.x >> 6] = sa; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < VB / 64; i++) { mn = fminf(mn, s_mn[i]); mx = fmaxf(mx, s_mx[i]); sa += s_sa[i]; }
    out[blockIdx.x * 3 + 0] = mn; out[blockIdx.x * 3 + 1] = mx; out[blockIdx.x * 3 + 2] = (float)sa;
  }
}

}  // namespace

extern "C" {

int dz_csr_set_twin(dazim_ctx *ctx, dazim_csr *A, dazim_csr *B) {
  if (!A || !B || A->twin) return dz_fail(ctx, DAZIM_E_BAD_ARG, "bad arguments to dz_csr_set_twin");
  A->twin = B;
  return 0;
}
// hand out the dense twin of a matrix built with option rays.dense_twin = 1 (NULL if there is none); the caller frees it
int dazim_csr_take_twin(dazim_ctx *ctx, dazim_csr *A, dazim_csr **twin) {
  if (!A || !twin) return dz_fail(ctx, DAZIM_E_BAD_ARG, "bad arguments to dazim_csr_take_twin");
  *twin = A->twin;
  A->twin = nullptr;
  return 0;
}

int dazim_csr_free(dazim_ctx *ctx, dazim_csr *A) {
  if (!A) return 0;
  if (A->twin) { dazim_csr *t = A->twin; A->twin = nullptr; (void)dazim_csr_free(ctx, t); }
  if (ctx) DZ_HIP(hipStreamSynchronize(ctx->stream));   // (a matrix may outlive its context: the arrays are still freed)
  else (void)hipDeviceSynchronize();
  void *ps[] = {A->rowptr, A->colptr, A->col, A->row, A->val, A->tval, A->tperm, A->cbptr, A->col16};
  for (void *p : ps)
    if (p) dz_big_put(ctx, p);
  delete A;
  return 0;
}

int dazim_csr_dims(const dazim_csr *A, int64_t *m, int64_t *n, int64_t *nnz) {
  if (!A) return DAZIM_E_BAD_ARG;
  if (m) *m = A->m;
  if (n) *n = A->n;
  if (nnz) *nnz = A->nnz;
  return 0;
}

int dazim_csr_from_coo(dazim_ctx *ctx, int64_t m, int64_t n, int64_t nnz, const int *irow_u,
                       const int *icol_u, const float *rw_u, dazim_csr **out) {
  if (!ctx || !out || m < 1 || n < 1 || nnz < 0 || nnz > 0xfffffff0ll || m > 0x7ffffff0 || n > 0x7ffffff0)
    return dz_fail(ctx, DAZIM_E_BAD_ARG, "bad arguments to dazim_csr_from_coo");
  DZ_HIP(hipSetDevice(ctx->device));
  DzBuf<int> irow, icol;
  DzBuf<float> rw;
  int rc;
  if ((rc = irow.init(ctx, irow_u, nnz, true, false))) return rc;
  if ((rc = icol.init(ctx, icol_u, nnz, true, false))) return rc;
  if ((rc = rw.init(ctx, rw_u, nnz, true, false))) return rc;
  dazim_csr *A = new dazim_csr;
  auto fail = [&](int r) { dazim_csr_free(ctx, A); return r; };   // no leak on the error paths below
  A->m = m;
  A->n = n;
  A->nnz = nnz;
  const size_t nz = (size_t)(nnz > 0 ? nnz : 1);
  { void *pp; if ((rc = dz_big_get(ctx, (m + 1) * 8, &pp))) return fail(rc); A->rowptr = (int64_t *)pp; }
  { void *pp; if ((rc = dz_big_get(ctx, nz * 4, &pp))) return fail(rc); A->col = (int *)pp; }
  { void *pp; if ((rc = dz_big_get(ctx, nz * 4, &pp))) return fail(rc); A->val = (float *)pp; }
  unsigned *k0, *k1, *v0, *perm;
  int *bad;
  void *p;
  if ((rc = dz_scratch(ctx, "csr.k0", nz * 4, &p))) return fail(rc);
  k0 = (unsigned *)p;
  if ((rc = dz_scratch(ctx, "csr.k1", nz * 4, &p))) return fail(rc);
  k1 = (unsigned *)p;
  if ((rc = dz_scratch(ctx, "csr.v0", nz * 4, &p))) return fail(rc);
  v0 = (unsigned *)p;
  if ((rc = dz_scratch(ctx, "csr.perm", nz * 4, &p))) return fail(rc);
  perm = (unsigned *)p;
  if ((rc = dz_scratch(ctx, "csr.bad", 16, &p))) return fail(rc);
  bad = (int *)p;
  DZ_HIP(hipMemsetAsync(bad, 0, 4, ctx->stream));
  const int nb = nblk(nnz);
  if (nnz > 0) {
    hipLaunchKernelGGL(k_check_range, dim3(nb), dim3(VB), 0, ctx->stream, nnz, irow.dev, 1, (int)m, bad);
    hipLaunchKernelGGL(k_check_range, dim3(nb), dim3(VB), 0, ctx->stream, nnz, icol.dev, 1, (int)n, bad);
    int hbad = 0;
    DZ_HIP(hipMemcpyAsync(&hbad, bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    DZ_HIP(hipStreamSynchronize(ctx->stream));
    if (hbad) {
      dazim_csr_free(ctx, A);
      return dz_fail(ctx, DAZIM_E_BAD_ARG, "COO index outside 1..m / 1..n");
    }
    int rbits = 1, cbits = 1;
    while (((int64_t)1 << rbits) < m) rbits++;
    while (((int64_t)1 << cbits) < n) cbits++;
    // ---- canonical CSR: stable sort by column, then stable sort by row -> rows ascending, columns
    // ascending inside a row (entries of equal (row,col) keep the caller's order) ----
    unsigned *permc;
    if ((rc = dz_scratch(ctx, "csr.permc", nz * 4, &p))) return fail(rc);
    permc = (unsigned *)p;
    hipLaunchKernelGGL(k_iota_keys, dim3(nb), dim3(VB), 0, ctx->stream, nnz, icol.dev, k0, v0);
    size_t tb = 0, tb1 = 0;
    DZ_HIP(rocprim::radix_sort_pairs(nullptr, tb, k0, k1, v0, permc, (size_t)nnz, 0, cbits, ctx->stream));
    DZ_HIP(rocprim::radix_sort_pairs(nullptr, tb1, k0, k1, permc, perm, (size_t)nnz, 0, rbits, ctx->stream));
    if (tb1 > tb) tb = tb1;
    void *tmp;
    if ((rc = dz_scratch(ctx, "csr.tmp", tb + 256, &tmp))) return fail(rc);
    DZ_HIP(rocprim::radix_sort_pairs(tmp, tb, k0, k1, v0, permc, (size_t)nnz, 0, cbits, ctx->stream));
    hipLaunchKernelGGL(k_gather_i, dim3(nb), dim3(VB), 0, ctx->stream, nnz, permc, irow.dev, (int *)k0, -1);
    DZ_HIP(rocprim::radix_sort_pairs(tmp, tb, k0, k1, permc, perm, (size_t)nnz, 0, rbits, ctx->stream));
    hipLaunchKernelGGL(k_lower_bound, dim3(nblk(m + 1)), dim3(VB), 0, ctx->stream, m, nnz, k1, A->rowptr);
    hipLaunchKernelGGL(k_gather_f, dim3(nb), dim3(VB), 0, ctx->stream, nnz, perm, rw.dev, A->val);
    hipLaunchKernelGGL(k_gather_i, dim3(nb), dim3(VB), 0, ctx->stream, nnz, perm, icol.dev, A->col, -1);
    DZ_HIP(hipGetLastError());
  } else {
    DZ_HIP(hipMemsetAsync(A->rowptr, 0, (m + 1) * 8, ctx->stream));
  }
  if ((rc = build_colblocks(ctx, A))) return fail(rc);
  if ((rc = invalidate_transpose(A))) return fail(rc);
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  *out = A;
  return 0;
}

// take ownership of device CSR arrays (hipMalloc'ed: rowptr[m+1], col[nnz] 0-based, val[nnz])
int dazim_csr_adopt(dazim_ctx *ctx, int64_t m, int64_t n, int64_t nnz, int64_t *rowptr, int *col, float *val,
                    dazim_csr **out) {
  return dz_csr_adopt_cap(ctx, m, n, nnz, rowptr, col, val, 0, 0, out);
}

// (library-internal) the same for arrays that are larger than m / nnz (room for rows appended later, dazim_csr::cap_m): the
// capacities are known BEFORE the column blocks are built, so the 16-bit column copy is sized for the reserved entries once
// and an append only narrows its own tail
int dz_csr_adopt_cap(dazim_ctx *ctx, int64_t m, int64_t n, int64_t nnz, int64_t *rowptr, int *col, float *val,
                     int64_t cap_m, int64_t cap_nnz, dazim_csr **out) {
  if (!ctx || !out || !rowptr) return DAZIM_E_BAD_ARG;
  dazim_csr *A = new dazim_csr;
  A->m = m; A->n = n; A->nnz = nnz;
  A->rowptr = rowptr; A->col = col; A->val = val;
  if (cap_m > m) A->cap_m = cap_m;
  if (cap_nnz > nnz) A->cap_nnz = cap_nnz;
  int rc;
  if ((rc = build_colblocks(ctx, A)) || (rc = invalidate_transpose(A))) {
    dazim_csr_free(ctx, A);   // ownership was taken: the arrays go with it
    return rc;
  }
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  *out = A;
  return 0;
}

// B = the entries of A with |value| > tol, same shape and order.  The reference keeps two copies of every ray row: the
// triplets rw/iw/col hold the entries with |row| > ftol (inv/CalSurfG.f90:1358), the dense GVs/GGc/GGs every entry of the cells
// with |fdm| >= ftol (:1369-1378), and its residual diagnostics multiply with the dense ones (inv/CalSigamNorm.f90:73).  A
// program that wants both builds the matrix once with option rays.keep_small and derives the solver's matrix here: two
// streaming passes instead of tracing the rays twice.  reserve_rows / reserve_nnz: room for rows appended to B later.
int dazim_csr_threshold(dazim_ctx *ctx, const dazim_csr *A, float tol, int64_t reserve_rows, int64_t reserve_nnz, dazim_csr **out) {
  if (!ctx || !A || !out || reserve_rows < 0 || reserve_nnz < 0) return dz_fail(ctx, DAZIM_E_BAD_ARG, "bad arguments to dazim_csr_threshold");
  DZ_HIP(hipSetDevice(ctx->device));
  const int64_t m = A->m;
  int rc;
  void *p;
  if ((rc = dz_scratch(ctx, "thr.cnt", (size_t)(m + 1) * 8, &p))) return rc;
  long *cnt = (long *)p;
  int64_t *rowptr = nullptr;
  float *val = nullptr;
  int *col = nullptr;
  struct Arrays {
    dazim_ctx *c; int64_t *&rp; float *&v; int *&cl; bool keep = false;
    ~Arrays() { if (!keep) { dz_big_put(c, rp); dz_big_put(c, v); dz_big_put(c, cl); } }
  } arrays{ctx, rowptr, val, col};
  { void *pp; if ((rc = dz_big_get(ctx, (size_t)(m + reserve_rows + 1) * 8, &pp))) return rc; rowptr = (int64_t *)pp; }
  const unsigned nb = (unsigned)((m + 1 + 3) / 4);
  hipLaunchKernelGGL((k_threshold_rows<false>), dim3(nb), dim3(256), 0, ctx->stream, m, A->rowptr, A->col, A->val, tol, cnt,
                     (const long *)nullptr, (int *)nullptr, (float *)nullptr);
  size_t tb = 0;
  DZ_HIP(rocprim::exclusive_scan(nullptr, tb, cnt, (long *)rowptr, 0l, (size_t)(m + 1), rocprim::plus<long>(), ctx->stream));
  if ((rc = dz_scratch(ctx, "thr.scan", tb + 256, &p))) return rc;
  DZ_HIP(rocprim::exclusive_scan(p, tb, cnt, (long *)rowptr, 0l, (size_t)(m + 1), rocprim::plus<long>(), ctx->stream));
  long nnz = 0;
  DZ_HIP(hipMemcpyAsync(&nnz, rowptr + m, 8, hipMemcpyDeviceToHost, ctx->stream));
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  const int64_t cap = nnz + reserve_nnz;
  { void *pp; if ((rc = dz_big_get(ctx, (size_t)(cap > 0 ? cap : 1) * 4, &pp))) return rc; val = (float *)pp; }
  { void *pp; if ((rc = dz_big_get(ctx, (size_t)(cap > 0 ? cap : 1) * 4, &pp))) return rc; col = (int *)pp; }
  hipLaunchKernelGGL((k_threshold_rows<true>), dim3(nb), dim3(256), 0, ctx->stream, m, A->rowptr, A->col, A->val, tol, cnt,
                     (const long *)rowptr, col, val);
  DZ_HIP(hipGetLastError());
  arrays.keep = true;
  return dz_csr_adopt_cap(ctx, m, A->n, nnz, rowptr, col, val, m + reserve_rows, cap, out);
}

// append rows m+1..m+extra_m given as COO (1-based absolute row ids, any order) -- the reference
// appends its Tikhonov rows to the same rw/iw/col arrays (inv/TikhRegul.f90:2)
int dazim_csr_append_coo(dazim_ctx *ctx, dazim_csr *A, int64_t extra_m, int64_t nnz2, const int *irow_u,
                         const int *icol_u, const float *rw_u) {
  if (!ctx || !A || extra_m < 0 || nnz2 < 0) return DAZIM_E_BAD_ARG;
  if (extra_m == 0 && nnz2 == 0) return 0;
  // build the block as its own matrix with rows shifted to 1..extra_m
  DzBuf<int> irow;
  int rc;
  if ((rc = irow.init(ctx, irow_u, nnz2, true, false))) return rc;
  int *shifted;
  void *p;
  if ((rc = dz_scratch(ctx, "csr.shift", (size_t)(nnz2 > 0 ? nnz2 : 1) * 4, &p))) return rc;
  shifted = (int *)p;
  if (nnz2 > 0) hipLaunchKernelGGL(k_gather_i, dim3(nblk(nnz2)), dim3(VB), 0, ctx->stream, nnz2, (const unsigned *)nullptr, irow.dev, shifted, (int)-A->m);
  dazim_csr *B = nullptr;
  if ((rc = dazim_csr_from_coo(ctx, extra_m > 0 ? extra_m : 1, A->n, nnz2, shifted, icol_u, rw_u, &B))) return rc;
  const int64_t m2 = A->m + extra_m, nz2 = A->nnz + nnz2;
  if (A->cap_m >= m2 && A->cap_nnz >= nz2) {   // the arrays were allocated with room for these rows: append in place
    const int64_t nnz1 = A->nnz;
    if (extra_m > 0)
      hipLaunchKernelGGL(k_offset_ptr, dim3(nblk(extra_m + 1)), dim3(VB), 0, ctx->stream, extra_m + 1, B->rowptr, A->nnz, A->rowptr + A->m);
    DZ_HIP(hipMemcpyAsync(A->col + nnz1, B->col, (size_t)nnz2 * 4, hipMemcpyDeviceToDevice, ctx->stream));
    DZ_HIP(hipMemcpyAsync(A->val + nnz1, B->val, (size_t)nnz2 * 4, hipMemcpyDeviceToDevice, ctx->stream));
    DZ_HIP(hipStreamSynchronize(ctx->stream));
    dazim_csr_free(ctx, B);
    A->m = m2;
    A->nnz = nz2;
    if ((rc = build_colblocks(ctx, A, nnz1 > 0 ? nnz1 : 0))) return rc;
    if ((rc = invalidate_transpose(A))) return rc;
    DZ_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
  }
  int64_t *rowptr;
  int *col;
  float *val;
  { void *pp; if ((rc = dz_big_get(ctx, (size_t)(m2 + 1) * 8, &pp))) return rc; rowptr = (int64_t *)pp; }
  { void *pp; if ((rc = dz_big_get(ctx, (size_t)(nz2 > 0 ? nz2 : 1) * 4, &pp))) return rc; col = (int *)pp; }
  { void *pp; if ((rc = dz_big_get(ctx, (size_t)(nz2 > 0 ? nz2 : 1) * 4, &pp))) return rc; val = (float *)pp; }
  DZ_HIP(hipMemcpyAsync(rowptr, A->rowptr, (A->m + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream));
  if (extra_m > 0)
    hipLaunchKernelGGL(k_offset_ptr, dim3(nblk(extra_m + 1)), dim3(VB), 0, ctx->stream, extra_m + 1, B->rowptr, A->nnz, rowptr + A->m);
  DZ_HIP(hipMemcpyAsync(col, A->col, (size_t)A->nnz * 4, hipMemcpyDeviceToDevice, ctx->stream));
  DZ_HIP(hipMemcpyAsync(val, A->val, (size_t)A->nnz * 4, hipMemcpyDeviceToDevice, ctx->stream));
  DZ_HIP(hipMemcpyAsync(col + A->nnz, B->col, (size_t)nnz2 * 4, hipMemcpyDeviceToDevice, ctx->stream));
  DZ_HIP(hipMemcpyAsync(val + A->nnz, B->val, (size_t)nnz2 * 4, hipMemcpyDeviceToDevice, ctx->stream));
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  dazim_csr_free(ctx, B);
  dz_big_put(ctx, A->rowptr);
  dz_big_put(ctx, A->col);
  dz_big_put(ctx, A->val);
  A->rowptr = rowptr; A->col = col; A->val = val;
  A->m = m2; A->nnz = nz2;
  A->cap_m = A->cap_nnz = 0;
  if ((rc = build_colblocks(ctx, A))) return rc;
  if ((rc = invalidate_transpose(A))) return rc;
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}


// copy the matrix out as the reference's COO triplets (1-based), rows ascending
int dazim_csr_to_coo(dazim_ctx *ctx, const dazim_csr *A, int *irow_u, int *icol_u, float *rw_u) {
  if (!ctx || !A) return DAZIM_E_BAD_ARG;
  DzBuf<int> irow, icol;
  DzBuf<float> rw;
  int rc;
  if ((rc = irow.init(ctx, irow_u, A->nnz, false, true))) return rc;
  if ((rc = icol.init(ctx, icol_u, A->nnz, false, true))) return rc;
  if ((rc = rw.init(ctx, rw_u, A->nnz, false, true))) return rc;
  if (A->nnz > 0) {
    void *p;
    if ((rc = dz_scratch(ctx, "csr.perm", (size_t)A->nnz * 4, &p))) return rc;
    unsigned *rowid = (unsigned *)p;
    hipLaunchKernelGGL(k_expand_rows, dim3(spmv_blocks(ctx, A->m, -1)), dim3(64 * WPB), 0, ctx->stream, A->m, A->rowptr, rowid);
    const int nb = nblk(A->nnz);
    if (irow.dev) hipLaunchKernelGGL(k_gather_i, dim3(nb), dim3(VB), 0, ctx->stream, A->nnz, (const unsigned *)nullptr, (const int *)rowid, irow.dev, 1);
    if (icol.dev) hipLaunchKernelGGL(k_gather_i, dim3(nb), dim3(VB), 0, ctx->stream, A->nnz, (const unsigned *)nullptr, A->col, icol.dev, 1);
    if (rw.dev) DZ_HIP(hipMemcpyAsync(rw.dev, A->val, (size_t)A->nnz * 4, hipMemcpyDeviceToDevice, ctx->stream));
    DZ_HIP(hipGetLastError());
  }
  if ((rc = irow.finish()) || (rc = icol.finish()) || (rc = rw.finish())) return rc;
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

int dazim_csr_scale_rows(dazim_ctx *ctx, dazim_csr *A, const float *w_u) {
  if (!ctx || !A || !w_u) return DAZIM_E_BAD_ARG;
  DzBuf<float> w;
  int rc;
  if ((rc = w.init(ctx, w_u, A->m, true, false))) return rc;
  hipLaunchKernelGGL(k_scale_rows, dim3(spmv_blocks(ctx, A->m, -1)), dim3(64 * WPB), 0, ctx->stream, A->m, A->rowptr, A->val, w.dev);
  if (A->tperm) hipLaunchKernelGGL(k_gather_f, dim3(nblk(A->nnz)), dim3(VB), 0, ctx->stream, A->nnz, A->tperm, A->val, A->tval);
  DZ_HIP(hipGetLastError());
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  return build_colblocks(ctx, A, -1);
}

int dazim_csr_col_abs_sums(dazim_ctx *ctx, const dazim_csr *A, float *out_u) {
  if (!ctx || !A || !out_u) return DAZIM_E_BAD_ARG;
  DzBuf<float> out;
  int rc;
  if ((rc = out.init(ctx, out_u, A->n, false, true))) return rc;
  void *p;
  if ((rc = dz_scratch(ctx, "csr.colacc", (size_t)A->n * 8, &p))) return rc;
  unsigned long long *acc = (unsigned long long *)p;
  DZ_HIP(hipMemsetAsync(acc, 0, (size_t)A->n * 8, ctx->stream));
  int e = 0, lgm = 0;
  if (A->vmax > 0) (void)frexp((double)A->vmax, &e);
  while (((int64_t)1 << lgm) < A->m) lgm++;
  const double scale = ldexp(1.0, (62 - lgm < 40 ? 62 - lgm : 40) - e);
  if (A->nnz) hipLaunchKernelGGL(k_col_abs_sums, dim3(nblk(A->nnz)), dim3(VB), 0, ctx->stream, A->nnz, A->col, A->val, scale, acc);
  hipLaunchKernelGGL(k_fixed_to_float, dim3(nblk(A->n)), dim3(VB), 0, ctx->stream, A->n, acc, 1.0 / scale, out.dev);
  DZ_HIP(hipGetLastError());
  if ((rc = out.finish())) return rc;
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

// aprod, inv/aprod.f90:7
int dazim_aprod(dazim_ctx *ctx, int mode, const dazim_csr *A, float *x_u, float *y_u) {
  if (!ctx || !A || !x_u || !y_u || (mode != 1 && mode != 2)) return dz_fail(ctx, DAZIM_E_BAD_ARG, "bad arguments to dazim_aprod");
  DZ_HIP(hipSetDevice(ctx->device));
  DzBuf<float> x, y;
  int rc;
  if ((rc = x.init(ctx, x_u, A->n, true, mode == 2))) return rc;
  if ((rc = y.init(ctx, y_u, A->m, true, mode == 1))) return rc;
  if (mode == 1) {
    DzTimer t(ctx, "spmv");
    if ((rc = launch_spmvA(ctx, A, x.dev, y.dev, nullptr, 1.0f, nullptr, nullptr))) return rc;
    t.stop();
  } else {
    float ymax = 1.0f;
    if (use_scatter(ctx, A)) {   // fixed-point scale needs max|y| (inside LSMR it is 1: u is normalised)
      void *p;
      if ((rc = dz_scratch(ctx, "csr.absmax", (APART + 4) * 4, &p))) return rc;
      float *pm = (float *)p;
      const int nb = nblk((A->m + 3) / 4, APART);
      hipLaunchKernelGGL(k_absmax, dim3(nb), dim3(VB), 0, ctx->stream, A->m, y.dev, pm);
      hipLaunchKernelGGL(k_absmax_finish, dim3(1), dim3(64), 0, ctx->stream, pm, nb, pm + APART);
      DZ_HIP(hipMemcpyAsync(&ymax, pm + APART, 4, hipMemcpyDeviceToHost, ctx->stream));
      DZ_HIP(hipStreamSynchronize(ctx->stream));
    }
    DzTimer t(ctx, "spmvt");
    if ((rc = launch_spmvT(ctx, A, y.dev, ymax, x.dev, nullptr, 1.0f, nullptr, nullptr))) return rc;
    t.stop();
  }
  if ((rc = x.finish()) || (rc = y.finish())) return rc;
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

static int dz_files_or_stage_allreduce(dazim_ctx *ctx, double *host, int count) {   // sum of a few host doubles over the ranks
  return dazim_comm_allreduce(ctx, host, count, DZ_F64, DZ_SUM);
}

// LSMR, inv/lsmrModule.f90:36-750.  Vectors AND scalars live on the device (LsmrState above); the host enqueues
// iterations and looks at the stop flag every CHECK iterations, one batch behind the one being enqueued, so the GPU never waits
// for the host.  With a communicator attached (dazim_comm_init) A and b are this rank's rows of one global system: one scalar
// all-reduce for ||u||^2 and one all-reduce of the n floats of A_p^T u_p per iteration, the state is replicated.
int dazim_lsmr_traced(dazim_ctx *ctx, const dazim_csr *A, const float *b_u, float damp, float atol, float btol,
                      float conlim, int itnlim, int localSize, float *x_u, int *istop_o, int *itn_o,
                      float *normA_o, float *condA_o, float *normr_o, float *normAr_o, float *normx_o,
                      dazim_lsmr_rec *trace, int trace_cap, int *trace_n) {
  if (!ctx || !A || !b_u || !x_u || (trace && trace_cap < 1)) return dz_fail(ctx, DAZIM_E_BAD_ARG, "bad arguments to dazim_lsmr");
  DZ_HIP(hipSetDevice(ctx->device));
  const int64_t m = A->m, n = A->n;
  DzBuf<float> b, x;
  int rc;
  DzComm *comm = (DzComm *)ctx->comm;   // non-null: A, b are this rank's rows of one global system
  void *p;
  double *d_sum = nullptr;
  long long *d_cons = nullptr;
  float *wbuf = nullptr;
  char *gbuf = nullptr;
  const size_t w_sum_off = (((size_t)n * 4 + 7) / 8) * 8, w_bytes = w_sum_off + 8;
  int64_t m_glob = m;
  int localVecs = 0;
  float *u = nullptr, *v = nullptr, *h = nullptr, *hbar = nullptr, *localV = nullptr, *d_scal = nullptr;
  double *part = nullptr, *part2 = nullptr, *partx = nullptr;
  LsmrState *S = nullptr;
  dazim_lsmr_rec *d_trace = nullptr;
  const int gm = spmv_blocks(ctx, m, n), gn = spmv_blocks(ctx, n, m);
  // ---- everything that can fail locally comes first, so that a row-sharded solve can agree on it before any rank waits in
  // a collective for a rank that has already returned ----
  auto setup = [&]() -> int {
    int r;
    if ((r = b.init(ctx, b_u, m, true, false))) return r;
    if ((r = x.init(ctx, x_u, n, false, true))) return r;
    if (comm) {
      if ((r = dz_scratch(ctx, "lsmr.cons", 64, &p))) return r;   // consensus words: from the scratch pool, inside the voted set-up
      d_cons = (long long *)p;
      if ((r = dz_scratch(ctx, "lsmr.sum", 64, &p))) return r;
      d_sum = (double *)p;
      if ((r = dz_scratch(ctx, "lsmr.w", w_bytes, &p))) return r;            // this rank's n floats of A_p^T u_p | its double beta_p^2
      wbuf = (float *)p;
      if ((r = dz_scratch(ctx, "lsmr.gather", w_bytes * (size_t)comm->nranks, &p))) return r;   // ... of every rank
      gbuf = (char *)p;
    }
    if ((r = dz_scratch(ctx, "lsmr.u", m * 4, &p))) return r;
    u = (float *)p;
    if ((r = dz_scratch(ctx, "lsmr.v", n * 4, &p))) return r;
    v = (float *)p;
    if ((r = dz_scratch(ctx, "lsmr.h", n * 4, &p))) return r;
    h = (float *)p;
    if ((r = dz_scratch(ctx, "lsmr.hbar", n * 4, &p))) return r;
    hbar = (float *)p;
    const int npart = gm > gn ? (gm > NPART ? gm : NPART) : (gn > NPART ? gn : NPART);
    if ((r = dz_scratch(ctx, "lsmr.part", (size_t)npart * 8, &p))) return r;
    part = (double *)p;
    if ((r = dz_scratch(ctx, "lsmr.part2", (size_t)NPART * 8 * 2 + 64, &p))) return r;
    part2 = (double *)p;
    if ((r = dz_scratch(ctx, "lsmr.partx", (size_t)NPART * 8, &p))) return r;
    partx = (double *)p;
    if ((r = dz_scratch(ctx, "lsmr.scal", 64, &p))) return r;
    d_scal = (float *)p;
    if ((r = dz_scratch(ctx, "lsmr.state", sizeof(LsmrState), &p))) return r;
    S = (LsmrState *)p;
    if (trace) {
      if ((r = dz_scratch(ctx, "lsmr.trace", (size_t)trace_cap * sizeof(dazim_lsmr_rec), &p))) return r;
      d_trace = (dazim_lsmr_rec *)p;
    }
    if (!use_scatter(ctx, A) && !A->colptr && (r = build_transpose(ctx, const_cast<dazim_csr *>(A)))) return r;
    // the reorthogonalisation window, sized by its upper bound min(localSize, n) (the global row count, known after the
    // consensus, can only make it smaller): allocated here so that its failure is part of the vote
    const int64_t lv = localSize < 0 ? 0 : (localSize < n ? localSize : n);
    if (lv > 0) {
      if ((r = dz_scratch(ctx, "lsmr.localV", (size_t)n * lv * 4, &p))) return r;
      localV = (float *)p;
    }
    return 0;
  };
  // after the consensus a rank that fails on its own must not leave the others waiting in a collective: abort the communicator
  // (every pending and future collective on it returns an error on every rank) and detach it
  auto leave = [&](int code) -> int {
    if (comm) {
      dz_comm_abort(ctx);
      comm = nullptr;
    }
    return code;
  };
  rc = setup();
  if (comm) {   // agree on (failure, n, m_total): every rank leaves together or none does
    long long hv[4] = {rc != 0 ? 1 : 0, (long long)n, -(long long)n, 0}, *dv = d_cons;
    double hm = (double)m;
    if (!dv) return leave(rc ? rc : dz_fail(ctx, -3, "row-sharded LSMR: no memory for the consensus buffer"));
    (void)hipMemcpyAsync(dv, hv, sizeof hv, hipMemcpyHostToDevice, ctx->stream);
    (void)hipMemcpyAsync(dv + 4, &hm, 8, hipMemcpyHostToDevice, ctx->stream);
    const int r1 = dz_allreduce(ctx, comm, dv, 3, DZ_I64, DZ_MAX);
    const int r2 = dz_allreduce(ctx, comm, dv + 4, 1, DZ_F64, DZ_SUM);
    (void)hipMemcpyAsync(hv, dv, sizeof hv, hipMemcpyDeviceToHost, ctx->stream);
    (void)hipMemcpyAsync(&hm, dv + 4, 8, hipMemcpyDeviceToHost, ctx->stream);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (rc) return rc;
    if (r1 != 0 || r2 != 0 || e != hipSuccess) return dz_fail(ctx, -2000, "row-sharded LSMR: consensus all-reduce failed");
    if (hv[0]) return dz_fail(ctx, -2001, "row-sharded LSMR: another rank failed during set-up");
    if (hv[1] != -hv[2]) return dz_fail(ctx, DAZIM_E_BAD_ARG, "row-sharded LSMR: the ranks disagree on the number of columns (%lld here, %lld elsewhere)", (long long)n, hv[1]);
    m_glob = (int64_t)hm;
  } else if (rc) {
    return rc;
  }
  // ---- from here on a local failure (a launch, a copy, a collective) aborts the communicator: see `leave` ----
  auto solve = [&]() -> int {
  localVecs = localSize < 0 ? 0 : localSize;
  if (m_glob < localVecs) localVecs = (int)m_glob;
  if (n < localVecs) localVecs = (int)n;
  constexpr int CHECK = 8, NSLOT = 2;
  struct Guard {   // pinned state copies + events, released on every exit path
    LsmrState *h = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr, done[NSLOT] = {}, ta[NSLOT][4] = {};
    ~Guard() {
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
      for (int i = 0; i < NSLOT; i++) {
        if (done[i]) (void)hipEventDestroy(done[i]);
        for (int j = 0; j < 4; j++)
          if (ta[i][j]) (void)hipEventDestroy(ta[i][j]);
      }
      if (h) (void)hipHostFree(h);
    }
  } guard;
  DZ_HIP(hipHostMalloc((void **)&guard.h, sizeof(LsmrState) * (NSLOT + 1) + 64));
  LsmrState *h_state = guard.h;
  float *h_scal = (float *)(guard.h + NSLOT + 1);
  DZ_HIP(hipEventCreate(&guard.e0));
  DZ_HIP(hipEventCreate(&guard.e1));
  for (int i = 0; i < NSLOT; i++) {
    DZ_HIP(hipEventCreate(&guard.done[i]));
    for (int j = 0; j < 4; j++) DZ_HIP(hipEventCreate(&guard.ta[i][j]));
  }
  // rowwise = the vector is sharded by rows (u): its squared norm is summed over the ranks first
  auto norm_to_host = [&](const double *pp, int np, float *res, bool rowwise = false) -> int {
    if (comm && rowwise) {
      hipLaunchKernelGGL(finish_norm, dim3(1), dim3(64), 0, ctx->stream, pp, np, d_scal, d_sum);
      { const int rr = dz_allreduce(ctx, comm, d_sum, 1, DZ_F64, DZ_SUM); if (rr) return rr; }
      hipLaunchKernelGGL(k_sqrt_sum, dim3(1), dim3(1), 0, ctx->stream, d_sum, d_scal);
    } else {
      hipLaunchKernelGGL(finish_norm, dim3(1), dim3(64), 0, ctx->stream, pp, np, d_scal, (double *)nullptr);
    }
    DZ_HIP(hipMemcpyAsync(h_scal, d_scal, 4, hipMemcpyDeviceToHost, ctx->stream));
    DZ_HIP(hipStreamSynchronize(ctx->stream));
    *res = h_scal[0];
    return 0;
  };
  const int bn = nblk(n, NPART), bm = nblk(m, NPART);
  DZ_HIP(hipEventRecord(guard.e0, ctx->stream));
  int gm_t = gm, gn_t = gn;   // partial counts of the last products
  // v(out) = A^T u + sign*beta*v with partials of ||v||^2 in `part` (row-sharded: local product, all-reduce, then the axpby)
  auto spmvT = [&](const float *beta_p, float sign, const int *g) -> int {
    if (!comm) return launch_spmvT(ctx, A, u, 1.0f, v, beta_p, sign, part, &gn_t, g);
    DZ_HIP(hipMemsetAsync(wbuf, 0, n * 4, ctx->stream));
    int r = launch_spmvT(ctx, A, u, 1.0f, wbuf, nullptr, 1.0f, nullptr, nullptr, g);
    if (r) return r;
    { const int rr = dz_allreduce(ctx, comm, wbuf, (size_t)n, DZ_F32, DZ_SUM); if (rr) return rr; }
    hipLaunchKernelGGL(k_axpby_norm, dim3(bn), dim3(VB), 0, ctx->stream, n, wbuf, v, beta_p, sign, part, g);
    gn_t = bn;
    return 0;
  };

  int istop = 0, itn = 0;
  float normA = 0, condA = 0, normr = 0, normAr = 0, normx = 0, normb = 0;
  int ntrace = 0;
  double t_spmv = 0, t_spmvt = 0;
  int n_spmv = 0, n_spmvt = 0;
  long n_enq = 0;       // iterations enqueued
  long n_coll = 0;      // collectives issued inside the iteration loop (row-sharded solve)
  bool rccl_allreduce = false;
  int host_syncs = 0;   // host waits on the device inside the iteration loop (one per examined batch of CHECK iterations)
  // u = b ; beta = ||u|| ; u /= beta ; v = A^T u ; alpha = ||v|| ; v /= alpha   (:355-372)
  hipLaunchKernelGGL(k_copy, dim3(bm), dim3(VB), 0, ctx->stream, m, b.dev, u);
  DZ_HIP(hipMemsetAsync(v, 0, n * 4, ctx->stream));
  DZ_HIP(hipMemsetAsync(x.dev, 0, n * 4, ctx->stream));
  DZ_HIP(hipMemsetAsync(hbar, 0, n * 4, ctx->stream));
  hipLaunchKernelGGL(k_sumsq, dim3(bm), dim3(VB), 0, ctx->stream, m, u, part);
  float alpha = 0.0f, beta = 0.0f;
  if ((rc = norm_to_host(part, bm, &beta, true))) return rc;
  if (beta > 0.0f) {
    hipLaunchKernelGGL(k_scal_inv, dim3(bm), dim3(VB), 0, ctx->stream, m, u, d_scal, 1.0f);
    if ((rc = spmvT(nullptr, 1.0f, nullptr))) return rc;  // v = 1*v(=0) + A^T u
    if ((rc = norm_to_host(part, gn_t, &alpha))) return rc;
  }
  if (alpha > 0.0f) hipLaunchKernelGGL(k_scal_inv, dim3(bn), dim3(VB), 0, ctx->stream, n, v, d_scal, 1.0f);
  normAr = alpha * beta;
  normb = beta;
  if (trace) {   // the line the reference prints before the loop (:468-471): itn 0, x(1) = 0, test1 = 1, test2 = alpha/beta
    memset(&trace[0], 0, sizeof trace[0]);
    trace[0].normr = beta; trace[0].normAr = normAr; trace[0].test1 = 1.0f; trace[0].test2 = beta > 0.0f ? alpha / beta : 0.0f;
    ntrace = 1;
  }
  if (normAr != 0.0f) {
    if (localVecs > 0) hipLaunchKernelGGL(k_copy, dim3(bn), dim3(VB), 0, ctx->stream, n, v, localV);   // localV(:,1) = v
    hipLaunchKernelGGL(k_copy, dim3(bn), dim3(VB), 0, ctx->stream, n, v, h);
    LsmrState &s0 = h_state[NSLOT];
    memset(&s0, 0, sizeof s0);
    s0.alpha = alpha; s0.beta = beta; s0.alphabar = alpha; s0.zetabar = alpha * beta; s0.rho = 1; s0.rhobar = 1; s0.cbar = 1;
    s0.betadd = beta; s0.rhodold = 1; s0.normA2 = alpha * alpha; s0.minrbar = 1e+30f; s0.normb = beta;
    s0.ctol = conlim > 0.0f ? 1.0f / conlim : 0.0f;
    s0.normr = beta; s0.normAr = normAr; s0.damp = damp; s0.atol = atol; s0.btol = btol; s0.itnlim = itnlim;
    DZ_HIP(hipMemcpyAsync(S, &s0, sizeof s0, hipMemcpyHostToDevice, ctx->stream));
    if (d_trace) DZ_HIP(hipMemsetAsync(d_trace, 0, (size_t)trace_cap * sizeof(dazim_lsmr_rec), ctx->stream));
    const int *g1 = &S->stop, *g2 = &S->stop2;
    // one iteration, enqueued without any host synchronisation; k = its number (the reorthogonalisation window is a function
    // of k alone: localVEnqueue advances once per iteration, :723-731)
    int coop_max = 0;
    if (!getenv("DAZIM_CU_MASK")) {
      int occ = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_reorth_coop<16>, RC_THREADS, 0) == hipSuccess && occ > 0) coop_max = occ * ctx->num_cu;
      else (void)hipGetLastError();
    }
    unsigned reorth_barriers = 0;
    rccl_allreduce = comm && comm->nccl && ctx->opts.count("comm.allreduce") && ctx->opts["comm.allreduce"] == 1;   // arrivals booked at the grid barrier of k_reorth_coop so far (its counter is zeroed here, once)
    DZ_HIP(hipMemsetAsync(part2 + 2 * NPART, 0, 64, ctx->stream));
    auto enqueue_iteration = [&](int k, hipEvent_t *tev) -> int {
      int r;
      if (tev) DZ_HIP(hipEventRecord(tev[0], ctx->stream));
      if ((r = launch_spmvA(ctx, A, v, u, &S->alpha, -1.0f, part, &gm_t, g1))) return r;   // u = A v - alpha u (:484-486)
      if (tev) DZ_HIP(hipEventRecord(tev[1], ctx->stream));
      float *slot = nullptr;
      int lim = 0;
      if (localVecs > 0) {
        const int ptr = k % localVecs + 1;             // localPointer after this iteration's enqueue
        slot = localV + (size_t)(ptr - 1) * n;
        lim = k >= localVecs ? localVecs : k + 1;     // localVQueueFull ? localVecs : localPointer (:738-742)
      }
      if (comm) {   // ONE collective per iteration (see k_local_norm_scal / k_beta_axpby)
        float *d_bp = reinterpret_cast<float *>(d_sum + 4);
        double *w_sum = reinterpret_cast<double *>(reinterpret_cast<char *>(wbuf) + w_sum_off);
        const int bmn = bm > bn ? bm : bn;
        hipLaunchKernelGGL(k_local_norm_scal, dim3(bm), dim3(VB), 0, ctx->stream, m, u, part, gm_t, w_sum, d_bp, S);
        if (tev) DZ_HIP(hipEventRecord(tev[2], ctx->stream));
        DZ_HIP(hipMemsetAsync(wbuf, 0, n * 4, ctx->stream));
        if ((r = launch_spmvT(ctx, A, u, 1.0f, wbuf, nullptr, 1.0f, nullptr, nullptr, g1))) return r;
        hipLaunchKernelGGL(k_scale_by, dim3(bn), dim3(VB), 0, ctx->stream, n, wbuf, d_bp, g1);
        if (tev) DZ_HIP(hipEventRecord(tev[3], ctx->stream));
        if (rccl_allreduce) {   // option comm.allreduce: RCCL's own sums (its order), the two buffers in one group
          DZ_NCCL(ncclGroupStart());
          const ncclResult_t ra = ncclAllReduce(wbuf, wbuf, (size_t)n, ncclFloat, ncclSum, comm->nccl, ctx->stream);
          const ncclResult_t rb = ncclAllReduce(w_sum, w_sum, 1, ncclDouble, ncclSum, comm->nccl, ctx->stream);
          const ncclResult_t rg = ncclGroupEnd();   // (inside a group the calls above only enqueue: launch errors surface here)
          if (ra != ncclSuccess || rb != ncclSuccess || rg != ncclSuccess) {
            const ncclResult_t bad = ra != ncclSuccess ? ra : (rb != ncclSuccess ? rb : rg);
            return dz_fail(ctx, -2000 - (int)bad, "row-sharded LSMR: grouped ncclAllReduce -> %s", ncclGetErrorString(bad));
          }
          hipLaunchKernelGGL(k_beta_axpby, dim3(bmn), dim3(VB), 0, ctx->stream, m, u, n, v, (const char *)wbuf, 1, w_bytes, w_sum_off, d_bp,
                             slot, part, S);
        } else {
          if ((r = dz_allgather(ctx, comm, wbuf, gbuf, w_bytes))) return r;
          hipLaunchKernelGGL(k_beta_axpby, dim3(bmn), dim3(VB), 0, ctx->stream, m, u, n, v, (const char *)gbuf, comm->nranks, w_bytes,
                             w_sum_off, d_bp, slot, part, S);
        }
        n_coll++;
        gn_t = bmn;
      } else {
        hipLaunchKernelGGL(k_beta_scal_u, dim3(bm > bn ? bm : bn), dim3(VB), 0, ctx->stream, m, u, part, gm_t,
                           (const double *)nullptr, n, v, slot, S);
        if (tev) DZ_HIP(hipEventRecord(tev[2], ctx->stream));
        if ((r = spmvT(&S->beta, -1.0f, g2))) return r;                                     // v = A^T u - beta v (:496-497)
        if (tev) DZ_HIP(hipEventRecord(tev[3], ctx->stream));
      }
      const double *pa = part;
      int npa = gn_t;
      const bool chain = ctx->opts.count("lsmr.reorth_chain") && ctx->opts["lsmr.reorth_chain"];
      // localVOrtho :733-748 in one launch (k_reorth_coop) when v fits the registers of RC_BLOCKS workgroups
      // as few workgroups as hold v with <= RC_EMAX elements per thread, eight where that is enough: a step's grid barrier is
      // atomics on one word across XCDs (whose L2s do not share it), and its cost grows with the arrivals -- test4_Yunnan's
      // 73 440-float v, ten vectors: 52 us with 64 workgroups (= the chain's eleven launches), 33 us with 8
      int rcb = (int)((n + (int64_t)RC_THREADS * RC_EMAX - 1) / ((int64_t)RC_THREADS * RC_EMAX));
      if (rcb < 8) rcb = 8;
      if (rcb > (int)((n + RC_THREADS - 1) / RC_THREADS)) rcb = (int)((n + RC_THREADS - 1) / RC_THREADS);
      if (rcb > RC_BLOCKS) rcb = RC_BLOCKS;
      if (ctx->opts.count("lsmr.reorth_blocks") && ctx->opts["lsmr.reorth_blocks"] > 0 && ctx->opts["lsmr.reorth_blocks"] < rcb) rcb = ctx->opts["lsmr.reorth_blocks"];
      const int64_t per_thread = (n + (int64_t)rcb * RC_THREADS - 1) / ((int64_t)rcb * RC_THREADS);
      // the grid barrier of k_reorth_coop needs every workgroup resident at once: bounded by what the occupancy query allows on this
      // device (coop_max, taken once per solve) -- never on a stream restricted to some CUs (DAZIM_CU_MASK), where that bound does not hold
      if (localVecs > 0 && lim > 0 && per_thread <= RC_EMAX && !chain && rcb <= coop_max) {
        unsigned *bar = reinterpret_cast<unsigned *>(part2 + 2 * NPART);
        const unsigned base = reorth_barriers;
        reorth_barriers += (unsigned)lim * (unsigned)rcb;
#define DZ_RC(E_) hipLaunchKernelGGL(k_reorth_coop<E_>, dim3(rcb), dim3(RC_THREADS), 0, ctx->stream, n, v, localV, lim, part2, part, bar, base, g2)
        if (per_thread <= 1) DZ_RC(1); else if (per_thread <= 2) DZ_RC(2); else if (per_thread <= 4) DZ_RC(4); else if (per_thread <= 8) DZ_RC(8); else DZ_RC(16);
#undef DZ_RC
        npa = rcb;
      } else if (localVecs > 0) {   // ... or as the chain: modified Gram-Schmidt, one launch per vector; the last one leaves ||v||^2
        for (int q = 0; q <= lim; q++) {
          const float *prev = q > 0 ? localV + (size_t)(q - 1) * n : nullptr;
          const float *next = q < lim ? localV + (size_t)q * n : nullptr;
          hipLaunchKernelGGL(k_reorth, dim3(bn), dim3(VB), 0, ctx->stream, n, v, prev, part2 + ((q + 1) & 1) * NPART, bn, next,
                             q < lim ? part2 + (q & 1) * NPART : part, g2);
        }
        npa = bn;
      }
      hipLaunchKernelGGL(k_alpha_update, dim3(bn), dim3(VB), 0, ctx->stream, n, v, h, hbar, x.dev, pa, npa, partx, S);
      hipLaunchKernelGGL(k_tests, dim3(1), dim3(64), 0, ctx->stream, partx, bn, x.dev, S, d_trace, trace_cap);
      DZ_HIP(hipGetLastError());
      return 0;
    };
    // batches of CHECK iterations; the state after batch j is copied to pinned slot j % NSLOT and examined while batch j+1
    // is already running
    const int limit = itnlim > 1 ? itnlim : 1;   // (the reference tests itn >= itnlim after its first iteration)
    int launched = 0, nbatch = 0, examined = 0;
    bool stopped = false;
    while (!stopped) {
      if (launched < limit) {
        const int sl = nbatch % NSLOT;
        for (int i = 0; i < CHECK && launched < limit; i++) {
          launched++;
          n_enq++;
          if ((rc = enqueue_iteration(launched, i == 0 ? guard.ta[sl] : nullptr))) return rc;
        }
        DZ_HIP(hipMemcpyAsync(&h_state[sl], S, sizeof(LsmrState), hipMemcpyDeviceToHost, ctx->stream));
        DZ_HIP(hipEventRecord(guard.done[sl], ctx->stream));
        nbatch++;
      }
      const int keep = launched < limit ? 1 : 0;   // one batch stays unexamined while more can be enqueued behind it
      while (examined < nbatch - keep && !stopped) {
        const int ls = examined % NSLOT;
        DZ_HIP(hipEventSynchronize(guard.done[ls]));
        host_syncs++;
        float ms = 0;
        if (hipEventElapsedTime(&ms, guard.ta[ls][0], guard.ta[ls][1]) == hipSuccess) { t_spmv += ms * 1e-3; n_spmv++; }
        if (hipEventElapsedTime(&ms, guard.ta[ls][2], guard.ta[ls][3]) == hipSuccess) { t_spmvt += ms * 1e-3; n_spmvt++; }
        s0 = h_state[ls];
        stopped = s0.stop != 0;
        examined++;
      }
      if (!stopped && launched >= limit && examined == nbatch) stopped = true;   // (istop = 7 sets the flag at itn >= itnlim)
    }
    DZ_HIP(hipStreamSynchronize(ctx->stream));
    istop = s0.istop; itn = s0.itn; normA = s0.normA; condA = s0.condA; normr = s0.normr; normAr = s0.normAr; normx = s0.normx;
    if (trace) {
      const int cnt = itn + 1 < trace_cap ? itn + 1 : trace_cap;
      if (cnt > 1) {
        DZ_HIP(hipMemcpyAsync(trace + 1, d_trace + 1, (size_t)(cnt - 1) * sizeof(dazim_lsmr_rec), hipMemcpyDeviceToHost, ctx->stream));
        DZ_HIP(hipStreamSynchronize(ctx->stream));
      }
      ntrace = cnt;
    }
  }
  if (damp > 0.0f && istop == 2) istop = 3;  // :686
  DZ_HIP(hipEventRecord(guard.e1, ctx->stream));
  DZ_HIP(hipEventSynchronize(guard.e1));
  float ms = 0;
  DZ_HIP(hipEventElapsedTime(&ms, guard.e0, guard.e1));
  ctx->ksec["lsmr"] = ms * 1e-3;
  ctx->ksec["spmv"] = n_spmv ? t_spmv / n_spmv : -1.0;
  ctx->ksec["spmvt"] = n_spmvt ? t_spmvt / n_spmvt : -1.0;
  ctx->ksec["lsmr.normb"] = normb;
  ctx->ksec["lsmr.host_syncs"] = host_syncs;
  // counted, not assumed: collectives issued by the loop / iterations enqueued (the n floats of A_p^T u_p with the double ||u_p||^2)
  ctx->ksec["lsmr.collectives_per_iteration"] = comm && n_enq > 0 ? (double)n_coll / (double)n_enq : 0.0;
  ctx->ksec["lsmr.collective_kind"] = comm ? (rccl_allreduce ? 2.0 : 1.0) : 0.0;   // 1 all-gather + rank-ordered sums, 2 ncclAllReduce
  {
    int nr = 1;
    if (comm && comm->nccl) (void)ncclCommCount(comm->nccl, &nr);
    else if (comm) nr = comm->nranks;
    ctx->ksec["lsmr.nranks"] = nr;     // ranks the communicator of this solve really has
    ctx->ksec["lsmr.transport"] = comm ? (comm->nccl ? 1.0 : 2.0) : 0.0;   // 1 RCCL, 2 files (tests)
  }
  if (istop_o) *istop_o = istop;
  if (itn_o) *itn_o = itn;
  if (normA_o) *normA_o = normA;
  if (condA_o) *condA_o = condA;
  if (normr_o) *normr_o = normr;
  if (normAr_o) *normAr_o = normAr;
  if (normx_o) *normx_o = normx;
  if (trace_n) *trace_n = ntrace;
  if ((rc = x.finish())) return rc;
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
  };
  rc = solve();
  return rc ? leave(rc) : 0;
}

int dazim_lsmr(dazim_ctx *ctx, const dazim_csr *A, const float *b_u, float damp, float atol, float btol,
               float conlim, int itnlim, int localSize, float *x_u, int *istop_o, int *itn_o,
               float *normA_o, float *condA_o, float *normr_o, float *normAr_o, float *normx_o) {
  return dazim_lsmr_traced(ctx, A, b_u, damp, atol, btol, conlim, itnlim, localSize, x_u, istop_o, itn_o, normA_o, condA_o,
                           normr_o, normAr_o, normx_o, nullptr, 0, nullptr);
}

// ---- N4 ------------------------------------------------------------------------------------------------------------------------
// = TikhonovRegularization / TikhRegul_joint (inv/TikhRegul.f90:2-104, :107-209): nblock*maxvp rows appended to the resident
// matrix, generated on the device (block b regularises columns b*maxvp+1.., weight w[b])
int dazim_csr_append_tikhonov(dazim_ctx *ctx, dazim_csr *A, int nx, int ny, int nz, int nblock, const float *w_host) {
  if (nblock < 1 || nx < 3 || ny < 3 || nz < 2) return dz_fail(ctx, DAZIM_E_BAD_ARG, "bad arguments to dazim_csr_append_tikhonov");
  return dazim_csr_append_tikhonov_rows(ctx, A, nx, ny, nz, nblock, w_host, 0, (int64_t)(nx - 2) * (ny - 2) * (nz - 1) * nblock);
}
// rows [row_lo, row_hi) of the same nblock*maxvp regularisation rows: the share of one rank of a row-sharded system
int dazim_csr_append_tikhonov_rows(dazim_ctx *ctx, dazim_csr *A, int nx, int ny, int nz, int nblock, const float *w_host,
                                   int64_t row_lo, int64_t row_hi) {
  if (!ctx || !A || !w_host || nblock < 1 || nx < 3 || ny < 3 || nz < 2) return dz_fail(ctx, DAZIM_E_BAD_ARG, "bad arguments to dazim_csr_append_tikhonov");
  const int nvx = nx - 2, nvz = ny - 2, nzm1 = nz - 1;
  const int64_t maxvp = (int64_t)nvx * nvz * nzm1, nrow = row_hi - row_lo;
  if (row_lo < 0 || row_hi < row_lo || row_hi > maxvp * nblock) return dz_fail(ctx, DAZIM_E_BAD_ARG, "bad row range of the regularisation block");
  if (maxvp * nblock > A->n || maxvp > 0x7ffffff0) return dz_fail(ctx, DAZIM_E_BAD_ARG, "regularisation blocks do not fit the %lld columns", (long long)A->n);
  DZ_HIP(hipSetDevice(ctx->device));
  int rc;
  void *p;
  if ((rc = dz_scratch(ctx, "tikh.cnt", (size_t)(nrow + 1) * 8, &p))) return rc;
  long *cnt = (long *)p;
  if ((rc = dz_scratch(ctx, "tikh.off", (size_t)(nrow + 1) * 8, &p))) return rc;
  long *off = (long *)p;
  if ((rc = dz_scratch(ctx, "tikh.w", 64 * 4, &p))) return rc;
  float *dw = (float *)p;
  if (nblock > 64) return dz_fail(ctx, DAZIM_E_BAD_ARG, "too many regularisation blocks");
  DZ_HIP(hipMemcpyAsync(dw, w_host, (size_t)nblock * 4, hipMemcpyHostToDevice, ctx->stream));
  const unsigned nb = (unsigned)((nrow + 1 + VB - 1) / VB);
  hipLaunchKernelGGL(k_tikh_count, dim3(nb), dim3(VB), 0, ctx->stream, row_lo, nrow, (int)maxvp, nvx, nvz, nzm1, cnt);
  size_t tb = 0;
  DZ_HIP(rocprim::exclusive_scan(nullptr, tb, cnt, off, 0l, (size_t)(nrow + 1), rocprim::plus<long>(), ctx->stream));
  if ((rc = dz_scratch(ctx, "tikh.scan", tb + 256, &p))) return rc;
  DZ_HIP(rocprim::exclusive_scan(p, tb, cnt, off, 0l, (size_t)(nrow + 1), rocprim::plus<long>(), ctx->stream));
  long nnz2 = 0;
  DZ_HIP(hipMemcpyAsync(&nnz2, off + nrow, 8, hipMemcpyDeviceToHost, ctx->stream));
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  const int64_t m2 = A->m + nrow, nz2 = A->nnz + nnz2;
  if (nz2 > 0xfffffff0ll) return dz_fail(ctx, DAZIM_E_NNZ_OVERFLOW, "too many stored entries");
  if (A->cap_m >= m2 && A->cap_nnz >= nz2) {   // dazim_rays_build_G left room for these rows: generate them behind the ray rows
    const int64_t nnz1 = A->nnz;
    hipLaunchKernelGGL(k_tikh_fill, dim3(nb), dim3(VB), 0, ctx->stream, row_lo, nrow, (int)maxvp, nvx, nvz, nzm1, off, A->nnz, dw,
                       A->rowptr + A->m, A->col, A->val);
    DZ_HIP(hipGetLastError());
    A->m = m2;
    A->nnz = nz2;
    if ((rc = build_colblocks(ctx, A, nnz1 > 0 ? nnz1 : 0))) return rc;
    if ((rc = invalidate_transpose(A))) return rc;
    DZ_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
  }
  int64_t *rowptr;
  int *col;
  float *val;
  { void *pp; if ((rc = dz_big_get(ctx, (size_t)(m2 + 1) * 8, &pp))) return rc; rowptr = (int64_t *)pp; }
  { void *pp; if ((rc = dz_big_get(ctx, (size_t)(nz2 > 0 ? nz2 : 1) * 4, &pp))) return rc; col = (int *)pp; }
  { void *pp; if ((rc = dz_big_get(ctx, (size_t)(nz2 > 0 ? nz2 : 1) * 4, &pp))) return rc; val = (float *)pp; }
  DZ_HIP(hipMemcpyAsync(rowptr, A->rowptr, (size_t)A->m * 8, hipMemcpyDeviceToDevice, ctx->stream));
  DZ_HIP(hipMemcpyAsync(col, A->col, (size_t)A->nnz * 4, hipMemcpyDeviceToDevice, ctx->stream));
  DZ_HIP(hipMemcpyAsync(val, A->val, (size_t)A->nnz * 4, hipMemcpyDeviceToDevice, ctx->stream));
  hipLaunchKernelGGL(k_tikh_fill, dim3(nb), dim3(VB), 0, ctx->stream, row_lo, nrow, (int)maxvp, nvx, nvz, nzm1, off, A->nnz, dw,
                     rowptr + A->m, col, val);
  DZ_HIP(hipGetLastError());
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  dz_big_put(ctx, A->rowptr);
  dz_big_put(ctx, A->col);
  dz_big_put(ctx, A->val);
  A->rowptr = rowptr; A->col = col; A->val = val;
  A->m = m2; A->nnz = nz2;
  A->cap_m = A->cap_nnz = 0;
  if ((rc = build_colblocks(ctx, A))) return rc;
  if ((rc = invalidate_transpose(A))) return rc;
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

// = residuals + CalDdatSigma + data weights + weighted right-hand side + row scaling of G (inv/Main_Jt.f90:432-469,
// inv/CalSigamNorm.f90:2-41) on the device.  obst, dsyn in; res (= Tdata), datweight, rhs (= cbst weighted) out, dall each
// (host or device); G nullable.  stats (host, 8 floats): mean, std, mean |.|, rms of the residual; meandeltaT, stddeltaT;
// mean weight; mean |weighted residual|.
int dazim_weight_data(dazim_ctx *ctx, dazim_csr *G, int64_t dall, const float *obst_u, const float *dsyn_u, float *res_u,
                      float *wgt_u, float *rhs_u, float *stats) {
  return dazim_weight_data_sharded(ctx, G, dall, 0, dall, obst_u, dsyn_u, res_u, wgt_u, rhs_u, stats);
}
// The same for one rank's data rows [row0, row0 + dall) of dall_glob (communicator attached): meandeltaT / stddeltaT are the
// reference's two sequential fp32 sums over ALL data, so the relative residuals of all ranks are put together first (an
// all-reduce of the zero-padded vector: exact, every other rank adds zeros) and every rank runs the same sums; the statistics
// returned are those of the whole data set.
int dazim_weight_data_sharded(dazim_ctx *ctx, dazim_csr *G, int64_t dall, int64_t row0, int64_t dall_glob, const float *obst_u,
                              const float *dsyn_u, float *res_u, float *wgt_u, float *rhs_u, float *stats) {
  if (!ctx || dall < 1 || !obst_u || !dsyn_u || !res_u || !wgt_u || !rhs_u || (G && G->m < dall) || row0 < 0 || row0 + dall > dall_glob)
    return dz_fail(ctx, DAZIM_E_BAD_ARG, "bad arguments to dazim_weight_data");
  const bool sharded = ctx->comm && dall_glob > dall;
  DZ_HIP(hipSetDevice(ctx->device));
  DzBuf<float> obst, dsyn, res, wgt, rhs;
  int rc;
  if ((rc = obst.init(ctx, obst_u, dall, true, false)) || (rc = dsyn.init(ctx, dsyn_u, dall, true, false)) ||
      (rc = res.init(ctx, res_u, dall, false, true)) || (rc = wgt.init(ctx, wgt_u, dall, false, true)) ||
      (rc = rhs.init(ctx, rhs_u, dall, false, true)))
    return rc;
  void *p;
  if ((rc = dz_scratch(ctx, "wd.rel", (size_t)dall * 4, &p))) return rc;
  float *rel = (float *)p;
  if ((rc = dz_scratch(ctx, "wd.ms", 64, &p))) return rc;
  float *ms = (float *)p;
  if ((rc = dz_scratch(ctx, "wd.part", (size_t)NPART * 5 * 8, &p))) return rc;
  double *part = (double *)p;
  const int nb = nblk(dall, NPART);
  hipLaunchKernelGGL(k_residual, dim3(nb), dim3(VB), 0, ctx->stream, dall, obst.dev, dsyn.dev, res.dev, rel);
  if (sharded) {
    if ((rc = dz_scratch(ctx, "wd.relg", (size_t)dall_glob * 4, &p))) return rc;
    float *relg = (float *)p;
    DZ_HIP(hipMemsetAsync(relg, 0, (size_t)dall_glob * 4, ctx->stream));
    DZ_HIP(hipMemcpyAsync(relg + row0, rel, (size_t)dall * 4, hipMemcpyDeviceToDevice, ctx->stream));
    if ((rc = dz_allreduce(ctx, (DzComm *)ctx->comm, relg, (size_t)dall_glob, DZ_F32, DZ_SUM))) return rc;
    hipLaunchKernelGGL(k_sigma_stats, dim3(1), dim3(VB), 0, ctx->stream, dall_glob, relg, ms);
  } else {
    hipLaunchKernelGGL(k_sigma_stats, dim3(1), dim3(VB), 0, ctx->stream, dall, rel, ms);
  }
  hipLaunchKernelGGL(k_sigma_weights, dim3(nb), dim3(VB), 0, ctx->stream, dall, obst.dev, res.dev, rel, ms, wgt.dev, rhs.dev);
  hipLaunchKernelGGL(k_weight_sums, dim3(nb), dim3(VB), 0, ctx->stream, dall, res.dev, wgt.dev, rhs.dev, part);
  DZ_HIP(hipGetLastError());
  std::vector<double> hp((size_t)nb * 5);
  float hms[2];
  DZ_HIP(hipMemcpyAsync(hp.data(), part, hp.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
  DZ_HIP(hipMemcpyAsync(hms, ms, 8, hipMemcpyDeviceToHost, ctx->stream));
  if (G) {   // rw(i) = rw(i)*datweight(iw(1+i)) for the data rows; rows beyond dall (none yet in the reference's order) untouched
    hipLaunchKernelGGL(k_scale_rows, dim3(spmv_blocks(ctx, dall, -1)), dim3(64 * WPB), 0, ctx->stream, dall, G->rowptr, G->val, wgt.dev);
    if (G->tperm) hipLaunchKernelGGL(k_gather_f, dim3(nblk(G->nnz)), dim3(VB), 0, ctx->stream, G->nnz, G->tperm, G->val, G->tval);
    DZ_HIP(hipGetLastError());
  }
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  if (G && (rc = build_colblocks(ctx, G, -1))) return rc;
  if (stats) {
    double a[5] = {0, 0, 0, 0, 0};
    for (int b = 0; b < nb; b++)
      for (int q = 0; q < 5; q++) a[q] += hp[(size_t)b * 5 + q];
    if (sharded && (rc = dz_files_or_stage_allreduce(ctx, a, 5))) return rc;
    const double n = (double)dall_glob, mean = a[0] / n;
    stats[0] = (float)mean;
    stats[1] = (float)sqrt(fmax(a[2] / n - mean * mean, 0.0));
    stats[2] = (float)(a[1] / n);
    stats[3] = (float)sqrt(a[2] / n);
    stats[4] = hms[0];
    stats[5] = hms[1];
    stats[6] = (float)(a[3] / n);
    stats[7] = (float)(a[4] / n);
  }
  if ((rc = res.finish()) || (rc = wgt.finish()) || (rc = rhs.finish())) return rc;
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

// = the clamped model update (inv/Main_Jt.f90:582-620) on the device: dv (n = maxvp or 3*maxvp, in/out: the dVs block is clamped
// to +-0.5 and zeroed below 1e-5), vs[nz][ny][nx] in/out (+= dVs on the inner cells, clamped to [minvel, maxvel]), gc, gs
// [nz-1][ny-2][nx-2] out (joint, nullable).  stats (host, nullable): per block (dVs, Gc, Gs) and depth k: min, max, sum |.| of
// the update -> [nblock][nz-1][3].
int dazim_model_update(dazim_ctx *ctx, int nx, int ny, int nz, int joint, float *vs_u, float *dv_u, float minvel, float maxvel,
                       float *gc_u, float *gs_u, float *stats) {
  if (!ctx || !vs_u || !dv_u || nx < 3 || ny < 3 || nz < 2) return dz_fail(ctx, DAZIM_E_BAD_ARG, "bad arguments to dazim_model_update");
  DZ_HIP(hipSetDevice(ctx->device));
  const int nzm1 = nz - 1, ncell = (nx - 2) * (ny - 2), maxvp = ncell * nzm1, nblock = joint ? 3 : 1;
  DzBuf<float> vs, dv, gc, gs;
  int rc;
  if ((rc = vs.init(ctx, vs_u, (size_t)nx * ny * nz, true, true)) || (rc = dv.init(ctx, dv_u, (size_t)maxvp * nblock, true, true)) ||
      (rc = gc.init(ctx, gc_u, joint ? maxvp : 0, false, true)) || (rc = gs.init(ctx, gs_u, joint ? maxvp : 0, false, true)))
    return rc;
  hipLaunchKernelGGL(k_model_update, dim3((maxvp + VB - 1) / VB), dim3(VB), 0, ctx->stream, nx, ny, nzm1, joint, vs.dev, dv.dev,
                     minvel, maxvel, joint ? gc.dev : nullptr, joint ? gs.dev : nullptr);
  DZ_HIP(hipGetLastError());
  if (stats) {
    void *p;
    if ((rc = dz_scratch(ctx, "mu.stats", (size_t)nblock * nzm1 * 3 * 4, &p))) return rc;
    hipLaunchKernelGGL(k_update_stats, dim3(nblock * nzm1), dim3(VB), 0, ctx->stream, ncell, dv.dev, (float *)p);
    DZ_HIP(hipGetLastError());
    DZ_HIP(hipMemcpyAsync(stats, p, (size_t)nblock * nzm1 * 3 * 4, hipMemcpyDeviceToHost, ctx->stream));
  }
  if ((rc = vs.finish()) || (rc = dv.finish()) || (rc = gc.finish()) || (rc = gs.finish())) return rc;
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

}  // extern "C"
