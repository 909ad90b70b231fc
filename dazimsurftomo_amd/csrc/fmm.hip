// fmm.hip -- batched fast-marching eikonal solver for gfx950 (K2 + K3 of SURVEY.md section 2).
//
// One 64-lane wavefront (= one workgroup) marches one (source, period) field at a time and pulls
// fields from an atomic queue until the batch is drained.  Per accepted node:
//   * the narrow-band binary heap lives in LDS as {key, node} pairs (keys cached next to the node
//     id, so sifting never touches HBM); slots >= CAP spill to a per-workgroup HBM array;
//   * lanes 0..15 evaluate the 4 neighbours x 4 quadrants of the mixed-order upwind stencil
//     (fouds2, inv/CalSurfG.f90:557-729) in parallel; their {T, status} loads are issued before
//     the root is sifted down so the L2 latency hides under the LDS heap work;
//   * heap updates are applied in the reference's order (x-1, x+1, z-1, z+1) so that the heap --
//     and therefore the acceptance order, including ties -- is bit-identical to the reference.
// A node's heap slot is not kept in HBM; a small LDS hint table (verified, with a wave-parallel
// search as fallback) recovers it for "close" neighbours.  fp32 without FMA contraction
// (this file is built with -ffp-contract=off); sin() of the colatitude comes from host tables so
// that it is the same libm value the CPU reference uses.
#include <cmath>

#include "dazim_internal.h"

namespace {

constexpr int GDX = 5, GDZ = 5, SGDL = 8, SGS = 8;  // inv/CalSurfG.f90:1005-1012
constexpr int RM = DAZIM_RMAX;
constexpr float EARTH = 6371.0f;
constexpr int TAB = 4096;  // slot-hint table entries

struct __align__(8) Node {
  float t;
  int s;  // during the march: -1 far, 0 alive, 1 close.  (the slot is written only at the end)
};
struct __align__(8) HEnt {
  float key;
  int node;  // (ix << 16) | iz, both 1-based (the reference's int16 px,pz: inv/CalSurfG.f90:238)
};

struct FmmArgs {
  dazim_geom g;
  int nfield, kmax;
  const double *pv;
  const float *veln;  // [kmax][nnx][nnz]
  const float *scx, *scz;
  const int *period;
  const float *risti_c;  // [nnx]       EARTH*sin(gox+(ix-1)*dnx)
  const float *risti_r;  // [nnx][RM]   same on the refined lattice for every possible vnl
  float *ttn, *ttnr;
  int *nstsr;
  dazim_refbox *boxes;
  int *status;
  Node *rec_c;   // [nwg][nnx*nnz]
  Node *rec_r;   // [nwg][RM*RM]
  float *velnr;  // [nwg][RM*RM]
  HEnt *ovf;     // [nwg][ovfcap]
  int ovfcap;
  unsigned *counter;
};

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float unif(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// cubic B-spline basis, inv/CalSurfG.f90:1472-1475
__device__ __forceinline__ void bspl4(float u, float w[4]) {
  float om = 1.0f - u;
  w[0] = om * om * om / 6.0f;
  w[1] = (4.0f - 6.0f * (u * u) + 3.0f * (u * u * u)) / 6.0f;
  w[2] = (1.0f + 3.0f * u + 3.0f * (u * u) - 3.0f * (u * u * u)) / 6.0f;
  w[3] = u * u * u / 6.0f;
}

// ---- gridder: inv/CalSurfG.f90:1423-1516, one thread per propagation node -------------------
__global__ void gridder_kernel(dazim_geom g, int kmax, const double *__restrict__ pv,
                               float *__restrict__ veln) {
  const int nn = g.nnx * g.nnz;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= nn * kmax) return;
  const int k = tid / nn, r = tid - k * nn;
  const int stx = r / g.nnz + 1, stz = r - (stx - 1) * g.nnz + 1;
  int i = (stz - 1) / GDZ + 1;
  if (i > g.nvz - 1) i = g.nvz - 1;
  const int l = stz - GDZ * (i - 1);
  int j = (stx - 1) / GDX + 1;
  if (j > g.nvx - 1) j = g.nvx - 1;
  const int m = stx - GDX * (j - 1);
  float ui[4], vi[4];
  bspl4((float)(m - 1) / (float)GDX, ui);
  bspl4((float)(l - 1) / (float)GDZ, vi);
  const double *p = pv + (size_t)k * (g.nvz + 2) * (g.nvx + 2);
  float sumi = 0.0f;
#pragma unroll
  for (int i1 = 1; i1 <= 4; i1++) {
    float sumj = 0.0f;
#pragma unroll
    for (int j1 = 1; j1 <= 4; j1++)
      sumj = sumj + ui[j1 - 1] * (float)p[(i - 2 + i1) * (g.nvx + 2) + (j - 2 + j1)];
    sumi = sumi + vi[i1 - 1] * sumj;
  }
  veln[tid] = sumi;
}

// ---- narrow-band heap (addtree/downtree/updtree, inv/CalSurfG.f90:738-891) -------------------
// "hole" formulation: the moving element stays in registers while displaced entries are copied;
// the comparisons, and hence the final array, are those of the reference's swap formulation.
template <int CAP>
struct Heap {
  HEnt *lds;            // [CAP], indexed by slot (slot 0 unused)
  unsigned short *tab;  // [TAB] slot hints
  HEnt *ovf;            // HBM spill for slots >= CAP
  int ntr;
  bool lane0;
  int lane;

  __device__ __forceinline__ static int hash(int node) { return ((node >> 16) & 63) << 6 | (node & 63); }

  __device__ __forceinline__ HEnt get(int slot) const {
    HEnt e = slot < CAP ? lds[slot] : ovf[slot - CAP];
    e.key = unif(e.key);
    e.node = uni(e.node);
    return e;
  }
  __device__ __forceinline__ void get2(int slot, HEnt &a, HEnt &b) const {  // slot even
    if (slot < CAP) {
      const int4 v = *reinterpret_cast<const int4 *>(&lds[slot]);
      a.key = __int_as_float(uni(v.x));
      a.node = uni(v.y);
      b.key = __int_as_float(uni(v.z));
      b.node = uni(v.w);
    } else {
      a = get(slot);
      b = get(slot + 1);
    }
  }
  __device__ __forceinline__ void put(int slot, float key, int node) {
    if (lane0) {
      HEnt e{key, node};
      if (slot < CAP)
        lds[slot] = e;
      else
        ovf[slot - CAP] = e;
      tab[hash(node)] = (unsigned short)slot;
    }
  }
  __device__ void sift_up(int c, float key, int node) {
    while (c > 1) {
      const int p = c >> 1;
      const HEnt pe = get(p);
      if (key < pe.key) {
        put(c, pe.key, pe.node);
        c = p;
      } else
        break;
    }
    put(c, key, node);
  }
  __device__ void pop_root() {  // downtree
    if (ntr == 1) {
      ntr = 0;
      return;
    }
    const HEnt mv = get(ntr);
    ntr--;
    int p = 1;
    bool broke = false;
    while (2 * p < ntr) {
      HEnt c0, c1;
      get2(2 * p, c0, c1);
      int c = 2 * p;
      float ck = c0.key;
      int cn = c0.node;
      if (c0.key > c1.key) {
        c = 2 * p + 1;
        ck = c1.key;
        cn = c1.node;
      }
      if (ck < mv.key) {
        put(p, ck, cn);
        p = c;
      } else {
        broke = true;
        break;
      }
    }
    if (!broke && 2 * p == ntr) {
      const HEnt c = get(2 * p);
      if (c.key < mv.key) {
        put(p, c.key, c.node);
        p = 2 * p;
      }
    }
    put(p, mv.key, mv.node);
  }
  // slot of a node that is known to be in the heap
  __device__ int find(int node) const {
    int s = uni((int)tab[hash(node)]);
    bool ok = s >= 1 && s <= ntr;
    if (ok) ok = get(s).node == node;
    if (ok) return s;
    for (int base = 1; base <= ntr; base += 64) {  // rare: hint overwritten by a colliding node
      const int sl = base + lane;
      int nd = -1;
      if (sl <= ntr) nd = (sl < CAP ? lds[sl] : ovf[sl - CAP]).node;
      const unsigned long long m = __ballot(nd == node);
      if (m) return base + (int)__builtin_ctzll(m);
    }
    return 1;  // unreachable for a consistent heap
  }
};

// ---- one marching run (travel, inv/CalSurfG.f90:356-456) -------------------------------------
// REFINED: urg=1 early-exit rule on the edges flagged in `ex` (bit0 x=1, bit1 x=nnx, bit2 z=1,
// bit3 z=nnz).  Returns the packed node the run exited on, or 0.
template <int CAP, bool REFINED>
__device__ int march(Heap<CAP> &H, Node *__restrict__ rec, const float *__restrict__ veln,
                     const float *__restrict__ risti_tab, int nnx, int nnz, int ld, float dnx,
                     float dnz, int ex) {
  const int lane = H.lane;
  const int nb = lane >> 2, q = lane & 3;
  const int dix = nb == 0 ? -1 : (nb == 1 ? 1 : 0);
  const int diz = nb == 2 ? -1 : (nb == 3 ? 1 : 0);
  const int jd = (q & 2) ? 1 : -1, kd = (q & 1) ? 1 : -1;
  const float ri = EARTH;
  while (H.ntr > 0) {
    wave_sync();
    const HEnt root = H.get(1);
    const int ix = root.node >> 16, iz = root.node & 0xffff;
    if (REFINED) {
      bool swrg = false;
      if (ix == 1 && (ex & 1)) swrg = true;
      if (ix == nnx && (ex & 2)) swrg = true;
      if (iz == 1 && (ex & 4)) swrg = true;
      if (iz == nnz && (ex & 8)) swrg = true;
      if (swrg) return root.node;
    }
    if (H.lane0) rec[(ix - 1) * ld + (iz - 1)].s = 0;
    wave_sync();
    // ---- issue the stencil loads (lanes 0..15), then sift while they are in flight ----
    const int nix = ix + dix, niz = iz + diz;
    const bool nvalid = lane < 16 && nix >= 1 && nix <= nnx && niz >= 1 && niz <= nnz;
    const int j = nix + jd, j2 = nix + 2 * jd, k = niz + kd, k2 = niz + 2 * kd;
    const bool vj = nvalid && j >= 1 && j <= nnx, vj2 = vj && j2 >= 1 && j2 <= nnx;
    const bool vk = nvalid && k >= 1 && k <= nnz, vk2 = vk && k2 >= 1 && k2 <= nnz;
    Node nself{0.0f, 0}, nj{0.0f, -1}, nj2{0.0f, -1}, nk{0.0f, -1}, nk2{0.0f, -1};
    float vel = 1.0f, risti = 0.0f;
    if (nvalid) {
      nself = rec[(nix - 1) * ld + (niz - 1)];
      vel = veln[(nix - 1) * ld + (niz - 1)];
      risti = risti_tab[nix - 1];
    }
    if (vj) nj = rec[(j - 1) * ld + (niz - 1)];
    if (vj2) nj2 = rec[(j2 - 1) * ld + (niz - 1)];
    if (vk) nk = rec[(nix - 1) * ld + (k - 1)];
    if (vk2) nk2 = rec[(nix - 1) * ld + (k2 - 1)];

    H.pop_root();

    // ---- fouds2 for (neighbour nb, quadrant q): inv/CalSurfG.f90:586-723 ----
    float trav = INFINITY;
    if (vj && vk && nself.s != 0) {
      const float slown = 1.0f / vel;
      const bool aj = nj.s == 0, ak = nk.s == 0;
      const bool so2j = vj2 && nj2.s == 0 && aj && nj.t > nj2.t;
      const bool so2k = vk2 && nk2.s == 0 && ak && nk.t > nk2.t;
      const float tj = nj.t, tj2 = nj2.t, tk = nk.t, tk2 = nk2.t;
      float a = 1.0f, b = 0.0f, c = 0.0f, tref = 0.0f, tdiv = 1.0f, u, v, em;
      bool sol = true;
      if (so2j) {
        if (so2k) {
          u = 2.0f * ri * dnx;
          v = 2.0f * risti * dnz;
          em = 4.0f * tj - tj2 - 4.0f * tk;
          em = em + tk2;
          a = v * v + u * u;
          b = 2.0f * em * (u * u);
          c = (u * u) * (em * em - (slown * slown) * (v * v));
          tref = 4.0f * tj - tj2;
          tdiv = 3.0f;
        } else if (ak) {
          u = risti * dnz;
          v = 2.0f * ri * dnx;
          em = 3.0f * tk - 4.0f * tj + tj2;
          a = v * v + 9.0f * (u * u);
          b = 6.0f * em * (u * u);
          c = (u * u) * (em * em - (slown * slown) * (v * v));
          tref = tk;
        } else {
          u = 2.0f * ri * dnx;
          c = -(u * u) * (slown * slown);
          tref = 4.0f * tj - tj2;
          tdiv = 3.0f;
        }
      } else if (aj) {
        if (so2k) {
          u = ri * dnx;
          v = 2.0f * risti * dnz;
          em = 3.0f * tj - 4.0f * tk + tk2;
          a = v * v + 9.0f * (u * u);
          b = 6.0f * em * (u * u);
          c = (u * u) * (em * em - (v * v) * (slown * slown));
          tref = tj;
        } else if (ak) {
          u = ri * dnx;
          v = risti * dnz;
          em = tk - tj;
          a = u * u + v * v;
          b = -2.0f * (u * u) * em;
          c = (u * u) * (em * em - (v * v) * (slown * slown));
          tref = tj;
        } else {
          c = -(slown * slown) * (ri * ri) * (dnx * dnx);
          tref = tj;
        }
      } else {
        if (so2k) {
          u = 2.0f * risti * dnz;
          c = -(u * u) * (slown * slown);
          tref = 4.0f * tk - tk2;
          tdiv = 3.0f;
        } else if (ak) {
          c = -(slown * slown) * (risti * risti) * (dnz * dnz);
          tref = tk;
        } else
          sol = false;
      }
      if (sol) {
        float rd1 = b * b - 4.0f * a * c;
        if (rd1 < 0.0f) rd1 = 0.0f;
        const float tdsh = (-b + sqrtf(rd1)) / (2.0f * a);
        trav = (tref + tdsh) / tdiv;
      }
    }
    trav = fminf(trav, __shfl_xor(trav, 1));
    trav = fminf(trav, __shfl_xor(trav, 2));

    // ---- apply the (up to) four heap updates in the reference's order ----
#pragma unroll
    for (int n = 0; n < 4; n++) {
      const int ux = ix + (n == 0 ? -1 : (n == 1 ? 1 : 0));
      const int uz = iz + (n == 2 ? -1 : (n == 3 ? 1 : 0));
      if (ux < 1 || ux > nnx || uz < 1 || uz > nnz) continue;
      const int cls = __builtin_amdgcn_readlane(nself.s, 4 * n);
      if (cls == 0) continue;
      const float tn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(trav), 4 * n));
      const int node = (ux << 16) | uz;
      if (H.lane0) rec[(ux - 1) * ld + (uz - 1)] = Node{tn, 1};
      if (cls < 0) {
        H.ntr++;
        H.sift_up(H.ntr, tn, node);
      } else {
        H.sift_up(H.find(node), tn, node);
      }
    }
  }
  return 0;
}

template <int CAP>
__global__ __launch_bounds__(64) void fmm_kernel(FmmArgs A) {
  __shared__ __attribute__((aligned(16))) HEnt s_heap[CAP];
  __shared__ unsigned short s_tab[TAB];
  __shared__ unsigned s_field;
  const int lane = threadIdx.x;
  const dazim_geom g = A.g;
  const int nnx = g.nnx, nnz = g.nnz, nn = nnx * nnz;
  Node *rec_c = A.rec_c + (size_t)blockIdx.x * nn;
  Node *rec_r = A.rec_r + (size_t)blockIdx.x * RM * RM;
  float *velnr = A.velnr + (size_t)blockIdx.x * RM * RM;
  Heap<CAP> H;
  H.lds = s_heap;
  H.tab = s_tab;
  H.ovf = A.ovf + (size_t)blockIdx.x * A.ovfcap;
  H.lane = lane;
  H.lane0 = lane == 0;

  for (;;) {
    __syncthreads();
    if (lane == 0) s_field = atomicAdd(A.counter, 1u);
    __syncthreads();
    const int f = (int)s_field;
    if (f >= A.nfield) break;
    const float scx = A.scx[f], scz = A.scz[f];
    const int per = A.period[f] - 1;
    const double *pv = A.pv + (size_t)per * (g.nvz + 2) * (g.nvx + 2);
    const float *veln = A.veln + (size_t)per * nn;
    float *ttn = A.ttn + (size_t)f * nn;

    // ---- refined source box, inv/CalSurfG.f90:1169-1206 ----
    int isx = (int)((scx - g.gox) / g.dnx) + 1;
    int isz = (int)((scz - g.goz) / g.dnz) + 1;
    const bool outside = isx < 1 || isx > nnx || isz < 1 || isz > nnz || per < 0 || per >= A.kmax;
    if (A.status && lane == 0) A.status[f] = outside ? DAZIM_E_SOURCE_OUTSIDE : 0;
    if (outside) {
      for (int i = lane; i < nn; i += 64) ttn[i] = 0.0f;
      continue;
    }
    if (isx == nnx) isx--;
    if (isz == nnz) isz--;
    dazim_refbox bx;
    bx.isx = isx;
    bx.isz = isz;
    bx.vnl = max(isx - SGS, 1);
    bx.vnr = min(isx + SGS, nnx);
    bx.vnt = max(isz - SGS, 1);
    bx.vnb = min(isz + SGS, nnz);
    bx.nnxr = (bx.vnr - bx.vnl) * SGDL + 1;
    bx.nnzr = (bx.vnb - bx.vnt) * SGDL + 1;
    bx.dnxr = g.dvx / (float)(GDX * SGDL);
    bx.dnzr = g.dvz / (float)(GDZ * SGDL);
    bx.goxr = g.gox + g.dnx * (float)(bx.vnl - 1);
    bx.gozr = g.goz + g.dnz * (float)(bx.vnt - 1);
    if (A.boxes && lane == 0) A.boxes[f] = bx;
    const int nnxr = bx.nnxr, nnzr = bx.nnzr;

    // ---- bsplrefine (inv/CalSurfG.f90:1525-1591) + status reset, one node per lane ----
    {
      const int nrxr = GDX * SGDL, nrzr = GDZ * SGDL;
      const int origx = (bx.vnl - 1) * SGDL + 1, origz = (bx.vnt - 1) * SGDL + 1;
      for (int idx = lane; idx < nnxr * RM; idx += 64) {
        const int idm2 = idx / RM + 1, idm1 = idx - (idm2 - 1) * RM + 1;
        if (idm1 > nnzr) continue;
        const int st2 = idm2 + origx - 1, st1 = idm1 + origz - 1;
        int jc = (st2 - 1) / nrxr + 1;
        if (jc > g.nvx - 1) jc = g.nvx - 1;
        const int l = st2 - nrxr * (jc - 1);
        int ic = (st1 - 1) / nrzr + 1;
        if (ic > g.nvz - 1) ic = g.nvz - 1;
        const int kk = st1 - nrzr * (ic - 1);
        float ui[4], vi[4], sum[4];
        bspl4((float)(l - 1) / (float)nrxr, ui);
        bspl4((float)(kk - 1) / (float)nrzr, vi);
#pragma unroll
        for (int i1 = 1; i1 <= 4; i1++) {
          float s = 0.0f;
#pragma unroll
          for (int j1 = 1; j1 <= 4; j1++)
            s = s + ui[j1 - 1] * (float)pv[(ic - 2 + i1) * (g.nvx + 2) + (jc - 2 + j1)];
          sum[i1 - 1] = vi[i1 - 1] * s;
        }
        velnr[idx] = sum[0] + sum[1] + sum[2] + sum[3];
        rec_r[idx] = Node{0.0f, -1};
      }
    }
    __syncthreads();

    // ---- travel(urg=1) source initialisation, inv/CalSurfG.f90:324-345 (uniform) ----
    H.ntr = 0;
    int rsx = (int)((scx - bx.goxr) / bx.dnxr) + 1;
    int rsz = (int)((scz - bx.gozr) / bx.dnzr) + 1;
    if (rsx == nnxr) rsx--;
    if (rsz == nnzr) rsz--;
    {
      const float dnx = bx.dnxr, dnz = bx.dnzr;
      float vss[2][2];
#pragma unroll
      for (int i = 1; i <= 2; i++)
#pragma unroll
        for (int jj = 1; jj <= 2; jj++) vss[i - 1][jj - 1] = unif(velnr[(rsx - 2 + i) * RM + (rsz - 2 + jj)]);
      const float dsx = (scx - bx.goxr) - (float)(rsx - 1) * dnx;
      const float dsz = (scz - bx.gozr) - (float)(rsz - 1) * dnz;
      float vsrc = 0.0f;  // bilinear, inv/CalSurfG.f90:2293
#pragma unroll
      for (int i = 1; i <= 2; i++)
#pragma unroll
        for (int jj = 1; jj <= 2; jj++) {
          const float produ = (1.0f - fabsf(((float)(i - 1) * dnx - dsx) / dnx)) *
                              (1.0f - fabsf(((float)(jj - 1) * dnz - dsz) / dnz));
          vsrc = vsrc + vss[i - 1][jj - 1] * produ;
        }
#pragma unroll
      for (int i = 1; i <= 2; i++)
#pragma unroll
        for (int jj = 1; jj <= 2; jj++) {
          const float ax = dsx - (float)(i - 1) * dnx, az = dsz - (float)(jj - 1) * dnz;
          const float ds = sqrtf(ax * ax + az * az);
          const float t0 = 2.0f * ds / (vss[i - 1][jj - 1] + vsrc);
          const int ux = rsx - 1 + i, uz = rsz - 1 + jj;
          if (lane == 0) rec_r[(ux - 1) * RM + (uz - 1)] = Node{t0, 1};
          H.ntr++;
          H.sift_up(H.ntr, t0, (ux << 16) | uz);
        }
    }
    // exit-rule quirk kept verbatim (inv/CalSurfG.f90:366-377): vnr/vnb (coarse indices) are
    // compared with the REFINED nnx/nnz, which is what the module variables hold at that point
    const int ex = (bx.vnl != 1 ? 1 : 0) | (bx.vnr != nnxr ? 2 : 0) | (bx.vnt != 1 ? 4 : 0) |
                   (bx.vnb != nnzr ? 8 : 0);
    const int exnode = march<CAP, true>(H, rec_r, velnr, A.risti_r + (size_t)(bx.vnl - 1) * RM,
                                        nnxr, nnzr, RM, bx.dnxr, bx.dnzr, ex);
    // nstsr carries the heap slot of every node still in the band (nstsr=nsts, :1247)
    __syncthreads();
    for (int sl = 1 + lane; sl <= H.ntr; sl += 64) {
      const HEnt e = sl < CAP ? s_heap[sl] : H.ovf[sl - CAP];
      rec_r[((e.node >> 16) - 1) * RM + ((e.node & 0xffff) - 1)].s = sl;
    }
    __syncthreads();
    if (exnode && lane == 0) rec_r[((exnode >> 16) - 1) * RM + ((exnode & 0xffff) - 1)].s = 0;
    __syncthreads();

    // ---- refined outputs + reset of the coarse records ----
    {
      float *ttnr = A.ttnr ? A.ttnr + (size_t)f * RM * RM : nullptr;
      int *nstsr = A.nstsr ? A.nstsr + (size_t)f * RM * RM : nullptr;
      if (ttnr || nstsr)
        for (int idx = lane; idx < RM * RM; idx += 64) {
          const int c = idx / RM, r = idx - c * RM;
          Node nd{0.0f, -9};
          if (c < nnxr && r < nnzr) nd = rec_r[idx];
          if (nstsr) nstsr[idx] = nd.s;
          if (ttnr) ttnr[idx] = nd.s >= 0 ? nd.t : 0.0f;
        }
      for (int i = lane; i < nn; i += 64) rec_c[i] = Node{0.0f, -1};
    }
    __syncthreads();
    // ---- inject every sgdl-th refined node (inv/CalSurfG.f90:1252-1262) ----
    const int bw = bx.vnr - bx.vnl + 1, bh = bx.vnb - bx.vnt + 1, nbox = bw * bh;
    for (int i = lane; i < nbox; i += 64) {
      const int bxi = i / bh, bzi = i - bxi * bh;  // column-major inside the box
      const Node nd = rec_r[(bxi * SGDL) * RM + bzi * SGDL];
      Node o{0.0f, nd.s};
      if (nd.s >= 0) o.t = nd.t;
      rec_c[(bx.vnl - 1 + bxi) * nnz + (bx.vnt - 1 + bzi)] = o;
    }
    __syncthreads();
    // ---- alive nodes touching a far node rejoin the band (inv/CalSurfG.f90:1291-1308).  Only the
    // tests against -1 matter and promotions never create a -1, so the sweep is order-free. ----
    for (int i = lane; i < nbox; i += 64) {
      const int bxi = i / bh, bzi = i - bxi * bh;
      const int cx = bx.vnl + bxi, cz = bx.vnt + bzi;
      Node *p = &rec_c[(cx - 1) * nnz + (cz - 1)];
      if (p->s == 0) {
        bool far = false;
        if (cz - 1 >= 1 && p[-1].s == -1) far = true;
        if (cz + 1 <= nnz && p[1].s == -1) far = true;
        if (cx - 1 >= 1 && p[-nnz].s == -1) far = true;
        if (cx + 1 <= nnx && p[nnz].s == -1) far = true;
        if (far) p->s = 1;
      }
    }
    __syncthreads();
    // ---- travel(urg=2): rebuild the band in column-major node order (inv/CalSurfG.f90:311-317) ----
    H.ntr = 0;
    for (int base = 0; base < nbox; base += 64) {
      const int i = base + lane;
      Node nd{0.0f, -1};
      int node = 0;
      if (i < nbox) {
        const int bxi = i / bh, bzi = i - bxi * bh;
        const int cx = bx.vnl + bxi, cz = bx.vnt + bzi;
        nd = rec_c[(cx - 1) * nnz + (cz - 1)];
        node = (cx << 16) | cz;
      }
      unsigned long long m = __ballot(nd.s > 0);
      while (m) {
        const int b = (int)__builtin_ctzll(m);
        m &= m - 1;
        const float t0 = __shfl(nd.t, b);
        const int n0 = __shfl(node, b);
        if (lane == 0) rec_c[((n0 >> 16) - 1) * nnz + ((n0 & 0xffff) - 1)].s = 1;
        H.ntr++;
        H.sift_up(H.ntr, unif(t0), uni(n0));
      }
    }
    march<CAP, false>(H, rec_c, veln, A.risti_c, nnx, nnz, nnz, g.dnx, g.dnz, 0);
    __syncthreads();
    for (int i = lane; i < nn; i += 64) ttn[i] = rec_c[i].t;  // coalesced traveltime-grid write
  }
}

}  // namespace

extern "C" int dazim_fmm_batch(dazim_ctx *ctx, int nx, int ny, float goxd, float gozd, float dvxd,
                               float dvzd, int kmax, const double *pv_u, int nfield,
                               const float *scx_u, const float *scz_u, const int *period_u,
                               float *veln_u, float *ttn_u, float *ttnr_u, int *nstsr_u,
                               dazim_refbox *boxes_u, int *status_u) {
  if (!ctx) return DAZIM_E_BAD_ARG;
  dazim_geom g;
  if (dazim_geometry(nx, ny, goxd, gozd, dvxd, dvzd, &g)) return dz_fail(ctx, DAZIM_E_BAD_ARG, "bad grid %dx%d", nx, ny);
  if (kmax < 1 || nfield < 0 || !pv_u || !ttn_u || g.nnx > 32767 || g.nnz > 32767)
    return dz_fail(ctx, DAZIM_E_BAD_ARG, "bad arguments to dazim_fmm_batch");
  DZ_HIP(hipSetDevice(ctx->device));
  const size_t nn = (size_t)g.nnx * g.nnz, npv = (size_t)(g.nvz + 2) * (g.nvx + 2), nr = (size_t)RM * RM;
  DzBuf<double> pv;
  DzBuf<float> scx, scz, veln, ttn, ttnr;
  DzBuf<int> period, nstsr, status;
  DzBuf<dazim_refbox> boxes;
  int rc;
  if ((rc = pv.init(ctx, pv_u, npv * kmax, true, false))) return rc;
  if ((rc = scx.init(ctx, scx_u, nfield, true, false))) return rc;
  if ((rc = scz.init(ctx, scz_u, nfield, true, false))) return rc;
  if ((rc = period.init(ctx, period_u, nfield, true, false))) return rc;
  if ((rc = veln.init(ctx, veln_u, nn * kmax, false, true))) return rc;
  if ((rc = ttn.init(ctx, ttn_u, nn * nfield, false, true))) return rc;
  if ((rc = ttnr.init(ctx, ttnr_u, nr * nfield, false, true))) return rc;
  if ((rc = nstsr.init(ctx, nstsr_u, nr * nfield, false, true))) return rc;
  if ((rc = boxes.init(ctx, boxes_u, nfield, false, true))) return rc;
  if ((rc = status.init(ctx, status_u, nfield, false, true))) return rc;

  // host tables of EARTH*sin(colatitude): same libm sinf the CPU reference calls (fouds2 :585)
  std::vector<float> rc_tab(g.nnx), rr_tab((size_t)g.nnx * RM);
  const float dnxr = g.dvx / (float)(GDX * SGDL);
  for (int i = 1; i <= g.nnx; i++) rc_tab[i - 1] = EARTH * sinf(g.gox + (float)(i - 1) * g.dnx);
  for (int vnl = 1; vnl <= g.nnx; vnl++) {
    const float goxr = g.gox + g.dnx * (float)(vnl - 1);
    for (int i = 1; i <= RM; i++) rr_tab[(size_t)(vnl - 1) * RM + i - 1] = EARTH * sinf(goxr + (float)(i - 1) * dnxr);
  }
  float *d_rc, *d_rr, *d_veln;
  void *p;
  if ((rc = dz_scratch(ctx, "fmm.risti_c", rc_tab.size() * 4, &p))) return rc;
  d_rc = (float *)p;
  if ((rc = dz_scratch(ctx, "fmm.risti_r", rr_tab.size() * 4, &p))) return rc;
  d_rr = (float *)p;
  DZ_HIP(hipMemcpyAsync(d_rc, rc_tab.data(), rc_tab.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  DZ_HIP(hipMemcpyAsync(d_rr, rr_tab.data(), rr_tab.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  DZ_HIP(hipStreamSynchronize(ctx->stream));  // the vectors above die at scope exit
  if (veln.dev)
    d_veln = veln.dev;
  else {
    if ((rc = dz_scratch(ctx, "fmm.veln", nn * kmax * 4, &p))) return rc;
    d_veln = (float *)p;
  }
  {
    DzTimer t(ctx, "gridder");
    const int total = (int)(nn * kmax);
    hipLaunchKernelGGL(gridder_kernel, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, g, kmax, pv.dev, d_veln);
    DZ_HIP(hipGetLastError());
    t.stop();
  }
  if (nfield > 0) {
    // workgroups: enough single-wave groups to fill every CU's LDS, never more than fields
    constexpr int CAP = 1024;
    int per_cu = 16;
    int nwg = ctx->num_cu * per_cu;
    if (nwg > nfield) nwg = nfield;
    const int ovfcap = (int)((nn > nr ? nn : nr) / 2 + 64);  // maxbt = nint(snb*nnx*nnz), :1068
    FmmArgs A;
    A.g = g;
    A.nfield = nfield;
    A.kmax = kmax;
    A.pv = pv.dev;
    A.veln = d_veln;
    A.scx = scx.dev;
    A.scz = scz.dev;
    A.period = period.dev;
    A.risti_c = d_rc;
    A.risti_r = d_rr;
    A.ttn = ttn.dev;
    A.ttnr = ttnr.dev;
    A.nstsr = nstsr.dev;
    A.boxes = boxes.dev;
    A.status = status.dev;
    A.ovfcap = ovfcap;
    if ((rc = dz_scratch(ctx, "fmm.rec_c", (size_t)nwg * nn * sizeof(Node), &p))) return rc;
    A.rec_c = (Node *)p;
    if ((rc = dz_scratch(ctx, "fmm.rec_r", (size_t)nwg * nr * sizeof(Node), &p))) return rc;
    A.rec_r = (Node *)p;
    if ((rc = dz_scratch(ctx, "fmm.velnr", (size_t)nwg * nr * 4, &p))) return rc;
    A.velnr = (float *)p;
    if ((rc = dz_scratch(ctx, "fmm.ovf", (size_t)nwg * ovfcap * sizeof(HEnt), &p))) return rc;
    A.ovf = (HEnt *)p;
    if ((rc = dz_scratch(ctx, "fmm.counter", 256, &p))) return rc;
    A.counter = (unsigned *)p;
    int *d_status = status.dev;
    if (!d_status) {
      if ((rc = dz_scratch(ctx, "fmm.status", (size_t)nfield * 4, &p))) return rc;
      d_status = (int *)p;
      A.status = d_status;
    }
    DZ_HIP(hipMemsetAsync(A.counter, 0, 256, ctx->stream));
    DzTimer t(ctx, "fmm");
    hipLaunchKernelGGL(fmm_kernel<CAP>, dim3(nwg), dim3(64), 0, ctx->stream, A);
    DZ_HIP(hipGetLastError());
    t.stop();
    // first failing field, like the reference's STOP
    std::vector<int> hs(nfield);
    DZ_HIP(hipMemcpyAsync(hs.data(), d_status, (size_t)nfield * 4, hipMemcpyDeviceToHost, ctx->stream));
    DZ_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < nfield; i++)
      if (hs[i]) {
        rc = dz_fail(ctx, hs[i], "field %d: source lies outside bounds of model", i);
        break;
      }
  }
  int rc2;
  if ((rc2 = veln.finish()) || (rc2 = ttn.finish()) || (rc2 = ttnr.finish()) || (rc2 = nstsr.finish()) ||
      (rc2 = boxes.finish()) || (rc2 = status.finish()))
    return rc2;
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  return rc;
}
