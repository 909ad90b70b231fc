// fmm.hip -- batched fast-marching eikonal solver for gfx950 (K2 + K3 of SURVEY.md section 2).
//
// Fast marching is a strict dependence chain per field (one heap pop after the other), so all the
// parallelism is across fields: a wavefront marches FOUR (source, period) fields at once, one per
// 16-lane group, and workgroups (= one wavefront) pull batches of four fields from an atomic queue
// until the batch is drained.  Per accepted node, inside a group:
//   * the narrow-band binary heap lives in LDS as {key, node} pairs (keys cached next to the node
//     id, so sifting never reads HBM); slots >= CAP spill to a per-group HBM array;
//   * the 16 lanes are the 4 neighbours x 4 quadrants of the mixed-order upwind stencil (fouds2,
//     inv/CalSurfG.f90:557-729): each lane loads its {T, status} records and solves one quadratic,
//     two __shfl_xor give the minimum per neighbour;
//   * heap updates are applied in the reference's order (x-1, x+1, z-1, z+1) so that the heap --
//     and therefore the acceptance order, including ties -- is bit-identical to the reference;
//   * node status in HBM carries the heap slot of band nodes exactly like the reference's nsts, so
//     "close" neighbours find their entry without a search (slots of the other pending neighbours
//     are tracked in registers while one of them sifts).
// The four groups execute the same instruction stream (SIMT), which cuts the instruction count per
// field four-fold compared with one field per wavefront -- the kernel is issue-bound, not HBM-bound.
// fp32 without FMA contraction (this file is built with -ffp-contract=off); sin() of the colatitude
// comes from host tables so that it is the same libm value the CPU reference uses.
#include <algorithm>
#include <cmath>

#include "dazim_internal.h"

namespace {

constexpr int GDX = 5, GDZ = 5, SGDL = 8, SGS = 8;  // inv/CalSurfG.f90:1005-1012
constexpr int RM = DAZIM_RMAX;
constexpr float EARTH = 6371.0f;

struct __align__(8) HEnt {
  float key;
  int node;  // index of the node's record in the tiled layout (tile_x + tile_z below; the reference keeps int16 px,pz: inv/CalSurfG.f90:238)
};

// Node word (round 3): ONE 32-bit word per node instead of the reference's {time, status} pair.  An alive node holds its time
// (a non-negative float: sign bit clear); a band node holds 0x80000000 | heap slot -- its trial time is the heap key, and nothing
// reads a band node's time from the grid (fouds2 replaces it, the stencil only uses alive neighbours, acceptance takes the key);
// a far node holds 0xffffffff.  Between the refined and the coarse march a close node that waits to be put into the heap holds
// its time with the sign bit set (-T; T = 0 gives 0x80000000, "slot 0", which never is a slot).  The refined grid's band, which
// nstsr / ttnr and the injection into the coarse grid want with slots AND trial times, is read back through the heap that is
// still in LDS when the refined march stops (the key at the slot the word holds).  4 x 4 tiles: two tiles that are neighbours in
// z share a 128-byte line, so a line covers 4 x 8 nodes.
constexpr unsigned W_FAR = 0xffffffffu, W_BAND = 0x80000000u;
__device__ __forceinline__ unsigned w_alive(float t) { return (unsigned)__float_as_int(t); }
__device__ __forceinline__ unsigned w_band(int slot) { return W_BAND | (unsigned)slot; }
__device__ __forceinline__ unsigned w_pending(float t) { return W_BAND | (unsigned)__float_as_int(t); }
__device__ __forceinline__ bool w_is_alive(unsigned w) { return (int)w >= 0; }
__device__ __forceinline__ bool w_is_pending(unsigned w) { return (int)w < 0 && w != W_FAR; }   // (before the coarse march only)
__device__ __forceinline__ float w_time(unsigned w) { return __int_as_float((int)(w & 0x7fffffffu)); }

struct FmmArgs {
  dazim_geom g;
  int nfield, kmax;
  const double *pv;
  const float *veln;  // [kmax][nnx][nnz]
  const float2 *slown; // [kmax][tiled nnx x nnz]  {1/veln, EARTH*sin(colatitude of the node's column)}: the slown = 1.0/vel of fouds2 :583
                      // (one IEEE division per node instead of one per update) and its risti (:585), in the same 4 x 4 tiles as the
                      // node records -- one index and one 8-byte load serve both (round 5; risti used to be a table look-up by column)
  const float *scx, *scz;
  const int *period;
  const float *risti_c;  // [nnx]       EARTH*sin(gox+(ix-1)*dnx)
  const float *risti_r;  // [nnx][RM]   same on the refined lattice for every possible vnl
  float *ttn, *ttnr;   // ttn nullable: the coarse fields stay in the kernel's own 4 x 4-tile layout (ttn_tiled) for the ray kernel
  unsigned *ttn_tiled; // [nfield][tiled nnx x nnz] finished fields (time bits of every node), field f at index tslot[f] (or f)
  const int *tslot;    // nullable.  Time-sliced batches: ttn_tiled IS rec_c and tslot the field's place in the queue -- nothing is copied
  unsigned *hprog;     // nullable (option fmm.async): host-mapped progress words, one per XCD range: tasks handed out so far
  int *fdone;          // nullable (option fmm.async): [nfield] completion flags for a ray kernel that runs beside this launch -- 1 once the
                       // field's times (and, long before, its refined outputs) are in memory, 2 if its band outgrew the heap (rerun pending)
  int *nstsr;
  dazim_refbox *boxes;
  int *status;
  unsigned *rec_c;   // [nwg][tiled nnx x nnz] node words of the coarse grid (see w_alive ...)
  unsigned *rec_r;   // [nwg][tiled RM x RM] node words of the refined grid
  float *velnr;  // [nwg][RM*RM]
  float2 *slownr; // [nwg][tiled RM x RM]  {1/velnr, risti} of the refined grid
  HEnt *ovf;     // [nwg][ovfcap]
  int ovfcap;
  unsigned *counter;     // [0..7] the XCD ranges' queue positions; [16..17] (as one 64-bit word) nodes accepted by the launch
  const int *flist;  // nullable: indirection used by the spill rerun
  int prio;          // 1: the wavefronts raise their issue priority (small batches beside the dispersion copies, see run_fmm)
  int fastm;         // grid steps within the range in which the short exact division / square root may run (see div_exact)
  const int *vflag;  // [1] set by gridder_kernel when a phase velocity lies outside that range
  int fpw;           // fields a wavefront takes per batch (1, 2 or FPW = 4 of its 16-lane groups are active): see run_fmm
  // time slicing (see fmm_kernel): a field is marched in ts_nstage tasks -- stage 0 = refined march + injection, stages 1.. =
  // ts_pops accepted nodes of the coarse march each (the last one: to the end) -- that may run on different workgroups
  int ts_nstage;     // 1: the whole field in one task (rec_c / ovf are per resident slot); > 1: rec_c / ovf / the arrays below per field
  int ts_pops;       // accepted nodes per coarse stage (the last stage runs to the end)
  unsigned *ts_flag; // [batches] stages of the batch that are complete
  float *ts_keys;    // [nfield][CAP] heap image between two stages (LDS part)
  int *ts_nodes;     // [nfield][CAP]; [..][0] = entries in that heap, <= 0: the field is finished.  (Everything a stage hands over
                     // lives in lines that belong to ONE field: fields change hands between XCDs, each with its own L2, and a
                     // line that two of them write -- a packed array of sizes was the first form -- can sit partly dirty in one L2
                     // while the other's update is in memory: a stale size, one wrong field in 10^5 hand-overs)
};

// cubic B-spline basis, inv/CalSurfG.f90:1472-1475
__device__ __forceinline__ void bspl4(float u, float w[4]) {
  float om = 1.0f - u;
  w[0] = om * om * om / 6.0f;
  w[1] = (4.0f - 6.0f * (u * u) + 3.0f * (u * u * u)) / 6.0f;
  w[2] = (1.0f + 3.0f * u + 3.0f * (u * u) - 3.0f * (u * u * u)) / 6.0f;
  w[3] = u * u * u / 6.0f;
}

// Node words live in HBM in 4 x 4 tiles (16 words of 4 bytes, see w_alive: two z-neighbouring tiles per 128-byte line): node
// (ix0, iz0), 0-based, of a grid
// with ntz tiles per column of tiles (ceil(nz/4) rounded up to a power of two) is record
// ((ix0>>2)*ntz + (iz0>>2))*16 + (ix0&3)*4 + (iz0&3).  The stencil of a pop reaches +-3 nodes in both directions: in the
// reference's column-major order that is 7 columns = 7-8 lines, tiled it is 4-6.  The two coordinates contribute separately, so
// the five addresses of a lane cost three X and three Z parts.  That record index is also the node's name in the heap (16 bits
// on grids up to 256 x 256, 32 bits otherwise): entries that move need no decoding to find their record, and only the root of a
// pop is turned back into coordinates (rid_x0 / rid_z0).  tsh = log2 of the record stride between columns of tiles.
__host__ __device__ constexpr int tile_shift(int nz) { int l = 0; while ((1 << l) < ((nz + 3) >> 2)) l++; return l + 4; }
__host__ __device__ constexpr int tile_stride(int nz) { return 1 << tile_shift(nz); }
__host__ __device__ constexpr int tile_records(int nx, int nz) { return ((nx + 3) >> 2) * tile_stride(nz); }
__host__ __device__ __forceinline__ constexpr int tile_x(int x0, int tsh) { return ((x0 >> 2) << tsh) + ((x0 & 3) << 2); }
__host__ __device__ __forceinline__ constexpr int tile_z(int z0) { return ((z0 & ~3) << 2) | (z0 & 3); }
__device__ __forceinline__ int rid_x0(int rid, int tsh) { return ((rid >> tsh) << 2) | ((rid >> 2) & 3); }
__device__ __forceinline__ int rid_z0(int rid, int tsh) { return (((rid & ((1 << tsh) - 1)) >> 4) << 2) | (rid & 3); }
static_assert(tile_shift(256) == dz_tile_shift(256) && tile_shift(511) == dz_tile_shift(511) && tile_shift(71) == dz_tile_shift(71) &&
              tile_x(37, 10) == dz_tile_x(37, 10) && tile_x(510, 11) == dz_tile_x(510, 11) && tile_z(37) == dz_tile_z(37) &&
              tile_z(510) == dz_tile_z(510), "dazim_internal.h restates this layout for the ray kernel");
constexpr int TSH_R = tile_shift(DAZIM_RMAX);                       // refined grid: 33 tiles per column of tiles, stride 64 tiles
constexpr int NREC_R = tile_records(DAZIM_RMAX, DAZIM_RMAX);        // 33 792 record slots per refined field

// ---- gridder: inv/CalSurfG.f90:1423-1516, one thread per propagation node -------------------
// FAST_V*: the velocities between which fmm_kernel may use the short exact division / square root (div_exact); veln and velnr are
// convex combinations of the pv values (the cubic B-spline weights are non-negative and sum to one)
constexpr float FAST_VMIN = 0.125f, FAST_VMAX = 16.0f, FAST_STEP_MIN = 2.0f, FAST_STEP_MAX = 4096.0f;
__global__ void gridder_kernel(dazim_geom g, int kmax, const double *__restrict__ pv,
                               float *__restrict__ veln, float2 *__restrict__ slown, const float *__restrict__ risti_c,
                               int *__restrict__ vflag) {
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
  bool bad = false;
#pragma unroll
  for (int i1 = 1; i1 <= 4; i1++) {
    float sumj = 0.0f;
#pragma unroll
    for (int j1 = 1; j1 <= 4; j1++) {
      const float pf = (float)p[(i - 2 + i1) * (g.nvx + 2) + (j - 2 + j1)];
      if (!(pf >= FAST_VMIN && pf <= FAST_VMAX)) bad = true;
      sumj = sumj + ui[j1 - 1] * pf;
    }
    sumi = sumi + vi[i1 - 1] * sumj;
  }
  if (bad) *vflag = 1;
  veln[tid] = sumi;
  slown[(size_t)k * tile_records(g.nnx, g.nnz) + tile_x(stx - 1, tile_shift(g.nnz)) + tile_z(stz - 1)] = make_float2(1.0f / sumi, risti_c[stx - 1]);   // 4 x 4 tiles
}

// ---- narrow-band heap (addtree/downtree/updtree, inv/CalSurfG.f90:738-891) -------------------
// One heap per 16-lane group.  "Hole" formulation: the moving element stays in registers while
// displaced entries are copied; the comparisons, and hence the final array, are those of the
// reference's swap formulation.  Like the reference, the node status in HBM carries a heap slot
// for every band node (nsts>0) -- exact in the spill kernel, written whenever an entry moves; in the kernels with the parallel
// sift-down (all-in-LDS and hybrid heaps) an ancestor-or-self relation of the true slot (lazy back-pointers, see march()).
constexpr int GP = 16;   // lanes per field
constexpr int FPW = 4;   // fields per wavefront

// __ballot(int) of the HIP headers turns its predicate into 0/1 and compares it with zero again (two VALU instructions per ballot
// whenever the predicate is a combination of lane masks); the builtin takes the lane mask as it stands.
__device__ __forceinline__ unsigned long long wballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// Lane masks (round 5).  A ballot of anything but a plain comparison -- a && b, a loop-carried flag, a flag set under a branch -- is
// compiled as v_cndmask 0/1 + v_cmp_ne (two four-cycle VALU slots), 17 times per pop in the round-4 loop.  The marching loop
// therefore forms such masks from the ballots of the plain comparisons with scalar and / or / andn2 (free: the scalar unit), and
// turns a mask back into a per-lane predicate with the inverse ballot (no instruction at all).  ~m may set bits of inactive
// lanes; every mask that is tested or shifted has at least one un-negated ballot factor, which is zero there.
using lmask = unsigned long long;
__device__ __forceinline__ bool lanes(lmask m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
// min with the lane's xor-1 / xor-2 partner in ONE instruction (v_min_*_dpp; fminf() through a v_mov_dpp costs the move, two
// canonicalising v_max and the v_min).  IEEE mode: a quiet NaN operand gives the other operand, like fminf; arithmetic cannot
// produce a signalling one.  s_nop 1 = the two wait states a DPP read needs after the VALU write of its source.
__device__ __forceinline__ float fmin_xor1(float v) { float r; asm("s_nop 1\n\tv_min_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(r) : "v"(v)); return r; }
__device__ __forceinline__ float fmin_xor2(float v) { float r; asm("s_nop 1\n\tv_min_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=&v"(r) : "v"(v)); return r; }
__device__ __forceinline__ int imin_xor1(int v) { int r; asm("s_nop 1\n\tv_min_i32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(r) : "v"(v)); return r; }
__device__ __forceinline__ int imin_xor2(int v) { int r; asm("s_nop 1\n\tv_min_i32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=&v"(r) : "v"(v)); return r; }

__device__ __forceinline__ void cbar() { asm volatile("" ::: "memory"); }  // compiler-only barrier:
// same-wave LDS/VMEM operations execute in program order, so no s_waitcnt is needed for lane 0's
// stores to be seen by the group's later loads.

// Cross-lane moves inside a 16-lane group as DPP modifiers on a v_mov (no LDS crossbar round trip):
// quad_perm swaps for the xor-1 / xor-2 exchanges and row_newbcast (gfx90a+) to broadcast lane L of each row.
// (mov_dpp = update_dpp with an undefined `old`: every lane is written by these controls, so the destination needs no
// initialising v_mov -- one instruction per move instead of two)
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) { return __int_as_float(dpp_i<CTRL>(__float_as_int(v))); }
constexpr int DPP_XOR1 = 0xB1;   // quad_perm:[1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;   // quad_perm:[2,3,0,1]
constexpr int DPP_BCAST0 = 0x150;  // row_newbcast:0 (+L for lane L of the row)
// value held by the OWNER lane of neighbour N (the first lane of its quad / pair) in every lane of the field's group.  16 lanes per
// field: one row_newbcast.  8 lanes per field: a row holds two fields -- lanes 0-7 take lane 2N, lanes 8-15 lane 8 + 2N (two DPP
// moves with complementary bank masks; a bank = four lanes of the row)
template <int GPL, int N> __device__ __forceinline__ int own_i(int v) {
  if constexpr (GPL == 16) {
    return dpp_i<DPP_BCAST0 + 4 * N>(v);
  } else {
    const int a = __builtin_amdgcn_mov_dpp(v, DPP_BCAST0 + 2 * N, 0xf, 0x3, false);
    return __builtin_amdgcn_update_dpp(a, v, DPP_BCAST0 + 8 + 2 * N, 0xf, 0xc, false);
  }
}
template <int GPL, int N> __device__ __forceinline__ float own_f(float v) { return __int_as_float(own_i<GPL, N>(__float_as_int(v))); }

// LDS layout is structure-of-arrays: fp32 keys + node ids (the node's record index, see tile_x above): NT = unsigned short when
// both grids of a field have at most 65 536 record slots (sides <= 256, the S-256 case: 6 bytes per entry, a third more fields
// in flight per CU), NT = int otherwise.

// HYB: the upper levels of the heap (slots < CAP, a power of two) live in LDS and are sifted by the parallel routines exactly as in
// the all-LDS heap; the NH levels below (slots CAP .. (CAP << NH) - 1) live in the HBM array `ovf` and are reached by one
// sequential step of the sift-down per level, by the owner lanes' direct writes and by rising entries.  Three uses (run_fmm's
// dispatch): 512 slots + one HBM level with 16-bit ids on 171..256-node grids when the batch is large (S-256: a third wavefront
// per SIMD), 512 slots + TWO HBM levels with 32-bit ids on 257..682-node grids (S-512: ten workgroups per CU; these grids wait
// on latencies, and twice the wavefronts are worth one or two more dependent memory accesses per pop), 1024 slots + one level
// (the round-2 form of S-512, option fmm.hyb2 = 2).
template <int CAP, bool SPILL, class NT, bool HYB = false, int GPL = 16>
struct Heap {
  static constexpr int GP = GPL;                          // lanes per field: 16 (four fields per wavefront) or 8 (eight)
  static constexpr int LV = GPL == 16 ? 4 : 3;            // levels per parallel sift-down step: GPL - 1 parent positions
  static constexpr unsigned GMASK = (1u << GPL) - 1u;
  // HBM levels of the hybrid heap: one with 16-bit node ids, two with 32-bit ids (grids of 257..682 nodes a side: levels 1-9 in
  // LDS, levels 10 and 11 in HBM; above that levels 1-10 in LDS, 11 and 12 in HBM -- see run_fmm's dispatch)
  static constexpr int NH = (HYB && (sizeof(NT) == 4 || CAP <= 256)) ? 2 : 1;
  static constexpr int TOT = HYB ? (CAP << NH) : CAP;   // slots the fast kernel can hold before the field is handed to the spill kernel
  // interior-root table of march(): 288 bytes of LDS, which cost no kernel form a workgroup per CU (LDS comes in 1 280-byte granules:
  // 12 288 + 288 B still give twelve, 16 384 + 288 nine like 16 384 alone -- profiles/r5_lds_granule.md)
#ifdef DZ_FMM_NOTAB   // experiment: the coordinate path for every root
  static constexpr bool TAB = false;
#else
  static constexpr bool TAB = !SPILL && GPL == 16;
#endif
  unsigned long long *popcnt;   // nodes accepted by the launch (nullable)
  short *tab;   // [144] the wavefront's table (16-bit: the S-256 form must stay within 12 800 bytes of LDS -- allocation comes in
                // 1 280-byte granules on this chip, and with 12 864 bytes a CU held eleven workgroups instead of twelve, -6 %)
  float *keys;  // this group's [CAP] keys (slot 0 unused)
  NT *nodes;    // this group's [CAP] node ids
  HEnt *ovf;   // HBM spill for slots >= CAP
  unsigned *rec;   // node words of the grid being marched (4 x 4 tiles, see w_alive)
  __device__ __forceinline__ void set_rec(unsigned *r) { rec = r; }
  __device__ __forceinline__ unsigned ldw(unsigned node) const { return rec[node]; }
  __device__ __forceinline__ void stw(unsigned node, unsigned w) { rec[node] = w; }
  // back-pointer store: a band node's word is its slot
  __device__ __forceinline__ void set_slot(unsigned node, int slot) { stw(node, w_band(slot)); }
  int tsh;     // log2 of the record stride between columns of tiles of that grid
  int ntr;
  bool g0;     // lane 0 of the group

  // SPILL=false: the whole band lives in LDS (no VMEM load inside the sift loops, so the back-pointer
  // stores never have to be waited for); a field whose band outgrows CAP is flagged and redone by
  // the SPILL=true instantiation, which keeps slots >= CAP in HBM.
  __device__ __forceinline__ HEnt get(int slot) const {
    if ((SPILL || HYB) && slot >= CAP) return ovf[slot - CAP];
    return HEnt{keys[slot], (int)nodes[slot]};
  }
  __device__ __forceinline__ void get2(int slot, HEnt &a, HEnt &b) const {  // slot even
    if (SPILL && slot >= CAP) {
      a = ovf[slot - CAP];
      b = ovf[slot + 1 - CAP];
    } else {
      const float2 k2 = *reinterpret_cast<const float2 *>(&keys[slot]);
      a.key = k2.x;
      b.key = k2.y;
      if (sizeof(NT) == 2) {
        const unsigned v = *reinterpret_cast<const unsigned *>(&nodes[slot]);
        a.node = (int)(v & 0xffffu);
        b.node = (int)(v >> 16);
      } else {
        a.node = (int)nodes[slot];
        b.node = (int)nodes[slot + 1];
      }
    }
  }
  // All 16 lanes of the group hold the same (slot, key, node): the LDS entry is written by all of them
  // (same address, same value -- no exec-mask branch inside the sift loops); the HBM back-pointer
  // and spill stores are issued by lane 0 only.
  __device__ __forceinline__ void put(int slot, float key, int node) {
    if ((SPILL || HYB) && slot >= CAP) {
      if (g0) ovf[slot - CAP] = HEnt{key, node};
    } else {
      keys[slot] = key;
      nodes[slot] = (NT)node;
    }
    if (g0) set_slot((unsigned)node, slot);
  }
  // sift (key,node) up from slot c.  If `track`, entries that move down are compared with the
  // pending neighbours' node ids so that their slots stay current (nbs[m] for m > from).
  template <bool TRACK>
  __device__ __forceinline__ void sift_up(int c, float key, int node, const int (&nbn)[4], int (&nbs)[4], int from) {
    while (c > 1) {
      const int p = c >> 1;
      const HEnt pe = get(p);
      if (key < pe.key) {
        put(c, pe.key, pe.node);
        if (TRACK) {
#pragma unroll
          for (int m = 1; m < 4; m++)
            if (m > from && pe.node == nbn[m]) nbs[m] = c;
        }
        c = p;
      } else
        break;
    }
    put(c, key, node);
  }
  // addtree / updtree (:738-783, :872-890) for the all-in-LDS heap in ONE LDS round instead of a loop over the levels: lane i of
  // the group reads ancestor c >> (i+1) of the rising entry (a heap of < 4096 slots has at most 11); the entry rises past the
  // leading run of ancestors whose key is larger (strict <, like the sequential loop), and each of those moves down one level --
  // lane i writes its ancestor into slot c >> i and that node's back-pointer, lane L places the entry itself at c >> L.  Same
  // comparisons, same final array.  Returns L, the number of levels risen (the caller shifts the slots of pending neighbours
  // that sat on the path).  Groups with live == false read and write the dummy slot 0.
  __device__ __forceinline__ int rise_par(bool live, int gl, int gbase, int c, float key, int node) {
    const int a = c >> (gl + 1);
    const bool valid = live && a >= 1;
    const bool ahi = HYB && NH > 1 && valid && a >= CAP;   // (two HBM levels: the parent of a slot of the lower one lies in the upper one)
    const int rs = (valid && !ahi) ? a : 0;
    float ak = keys[rs];
    NT an = nodes[rs];
    if (HYB && NH > 1 && __builtin_expect(wballot(ahi) != 0, 0)) {
      if (ahi) {
        const HEnt e = ovf[a - CAP];
        ak = e.key;
        an = (NT)e.node;
      }
    }
    const unsigned mb = (unsigned)(wballot(valid && key < ak) >> gbase) & GMASK;
    const int L = __builtin_ctz(~mb);                      // (bit GPL of ~mb is set: L <= GPL; <= 11 by the heap depth)
    const bool mover = live && gl < L;
    const int dst = mover ? (c >> gl) : 0;
    // (HYB: slot c itself -- lane 0's destination, or the entry's own if it does not rise -- and, with two HBM levels, its parent)
    const bool dhi = HYB && dst >= CAP;
    const int ldst = dhi ? 0 : dst;
    keys[ldst] = ak;
    nodes[ldst] = an;
    if (mover) set_slot((unsigned)an, dst);
    const bool last = live && gl == L;
    const int fin = c >> L;
    const bool fhi = HYB && last && fin >= CAP;
    if (last && !fhi) {
      keys[fin] = key;
      nodes[fin] = (NT)node;
    }
    if (last) set_slot((unsigned)node, fin);
    if (HYB && __builtin_expect(wballot(dhi || fhi) != 0, 0)) {
      if (dhi) ovf[dst - CAP] = HEnt{ak, (int)an};
      if (fhi) ovf[fin - CAP] = HEnt{key, node};
    }
    if constexpr (GPL < 16) {   // eight lanes see eight ancestors: an entry that passed all of them goes on from slot c >> GPL
      if (live && L == GPL && (c >> GPL) == 1) {     // ... which is the root: nobody had gl == L to place it
        keys[1] = key;
        nodes[1] = (NT)node;
        if (g0) set_slot((unsigned)node, 1);
      }
      const bool more = live && L == GPL && (c >> GPL) > 1;
      if (wballot(more) != 0) {
        cbar();
        return L + rise_par(more, gl, gbase, more ? (c >> GPL) : 0, key, node);   // (a group that is done reads and writes slot 0)
      }
    }
    return L;
  }
  __device__ __forceinline__ bool full() const { return !SPILL && ntr + 1 >= TOT; }
  // keys of the LDS slots from `from` on = +inf: the parallel sift-down reads children without asking whether they exist (slots
  // beyond the heap's end must lose every comparison); pop_root_par keeps the invariant when the heap shrinks, growing overwrites
  __device__ __forceinline__ void pad(int gl, int from) {
    for (int i = from + gl; i < CAP; i += GPL) keys[i] = INFINITY;
  }
  __device__ __forceinline__ void add(float key, int node) {
    const int nbn[4] = {0, 0, 0, 0};
    int nbs[4] = {0, 0, 0, 0};
    ntr++;
    sift_up<false>(ntr, key, node, nbn, nbs, 0);
  }
  // LDS-only write of a heap entry (the HBM back-pointer is deferred by the caller)
  __device__ __forceinline__ void put_lds(int slot, float key, int node) {
    if (SPILL && slot >= CAP) {
      if (g0) ovf[slot - CAP] = HEnt{key, node};
    } else {
      keys[slot] = key;
      nodes[slot] = (NT)node;
    }
  }
  // downtree.  The back-pointer stores of the entries that move are NOT issued here: move #i is
  // captured by lane i of the group (mynode/myslot, nmoves) and flushed by the caller after the
  // stencil loads have been consumed, so that those loads never queue behind this loop's stores.
  // nbs[n] receives the new slot of pending neighbour n if its entry moved.
  __device__ __forceinline__ void pop_root(int gl, const int (&nbn)[4], int (&nbm)[4], int &mynode, int &myslot,
                                           int &nmoves) {
    nmoves = 0;
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
        put_lds(p, ck, cn);
        if (gl == nmoves) { mynode = cn; myslot = p; }
#pragma unroll
        for (int n = 0; n < 4; n++)
          if (cn == nbn[n]) nbm[n] = p;
        nmoves++;
        p = c;
      } else {
        broke = true;
        break;
      }
    }
    if (!broke && 2 * p == ntr) {
      const HEnt c = get(2 * p);
      if (c.key < mv.key) {
        put_lds(p, c.key, c.node);
        if (gl == nmoves) { mynode = c.node; myslot = p; }
#pragma unroll
        for (int n = 0; n < 4; n++)
          if (c.node == nbn[n]) nbm[n] = p;
        nmoves++;
        p = 2 * p;
      }
    }
    put_lds(p, mv.key, mv.node);
    if (gl == nmoves) { mynode = mv.node; myslot = p; }
#pragma unroll
    for (int n = 0; n < 4; n++)
      if (mv.node == nbn[n]) nbm[n] = p;
    nmoves++;
  }

  // downtree for the heap in LDS, four levels per step.  Which child a hole descends to does not depend on the moving
  // entry (always the smaller child, ties to the left, :857-866), only the stopping depth does.  Lane i < 15 of the group is
  // parent position q = i + 1 of the 4-level subtree below the hole (q = 1: the hole itself): it reads ITS TWO CHILDREN (one
  // 8-byte key pair + one node pair), picks the smaller one, and two ballots tell every lane which way each of the 15 parents
  // would send the hole and whether the child going up is smaller than the moving key.  A lane whose ancestors all point at
  // it and all move (per-lane constant masks G/E/A) copies its smaller child into its own slot: same comparisons, same final
  // array as the sequential loop.  A missing child (slot > ntr) reads as +inf, which reproduces the reference's single-child
  // tail.  Two steps empty a heap below 512 slots (every refined grid, grids up to ~170 nodes a side), three one below 8192.
  // The words of the entries that move up are left alone (lazy back-pointers, see march()); the caller stores the slot of the
  // dropped entry (fin_node at fin_slot).
  static constexpr int clog2(int v) { int l = 0; while ((1 << l) < v) l++; return l; }
  static constexpr int NSTEP = (clog2(CAP) - 1 + LV - 1) / LV;   // the hole goes from level 0 to at most level clog2(CAP) - 1, LV levels per step
  __device__ __forceinline__ void pop_root_par(int lane, int &fin_node, int &fin_slot) {
    static_assert(CAP <= 8192 && NSTEP <= 4, "pop_root_par covers 13 levels");
    static_assert((CAP & 1) == 0, "pairs of children are read together");
    static_assert(!HYB || (CAP & (CAP - 1)) == 0, "HYB: the LDS part of the heap is whole levels");
    const int gl = lane & (GP - 1), gsh = lane & ~(GP - 1);
    fin_slot = 0;
    fin_node = 0;
    if (ntr == 1) {
      ntr = 0;
      return;
    }
    const bool mhi = HYB && ntr >= CAP;                   // the last entry sits in the HBM level
    const int mls = mhi ? 0 : ntr;
    float mvk = keys[mls];
    NT mvc = nodes[mls];
    keys[mls] = INFINITY;                                 // the slot leaves the heap: +inf beyond the end (see pad())
    if (HYB && __builtin_expect(wballot(mhi) != 0, 0)) {
      if (mhi) {
        const HEnt e = ovf[ntr - CAP];
        mvk = e.key;
        int mvi = e.node;
        // the wait for this load belongs inside the branch: left to the join below it would be executed by every pop and
        // drain the stencil loads that were issued before the sift-down (vmcnt is in-order)
        asm volatile("" : "+v"(mvk), "+v"(mvi));
        mvc = (NT)mvi;
      }
    }
    ntr--;
    // lane constants: position q, its depth dq and offset oq in the subtree; G = its proper ancestors' bits (bit a-1 per
    // position a), E = the directions those ancestors must choose to reach q, A = G + its own bit (all of them must move)
    const int q = gl == GP - 1 ? 1 : gl + 1;              // (the idle lane looks where lane 0 looks and never matches)
    const int dq = 31 - __clz(q), oq = q - (1 << dq);
    unsigned G = 0, E = 0;
    if (dq >= 1) { G |= 1u << ((q >> 1) - 1); E |= (unsigned)(q & 1) << ((q >> 1) - 1); }
    if (dq >= 2) { G |= 1u << ((q >> 2) - 1); E |= (unsigned)((q >> 1) & 1) << ((q >> 2) - 1); }
    if (dq >= 3) { G |= 1u; E |= (unsigned)((q >> 2) & 1); }
    unsigned A = G | (1u << (q - 1));
    if (gl == GP - 1) A = 1u << GP;                        // the idle lane never matches (lt is cut to GP bits)
    int p = 1;
    lmask actm = ~0ull, deepm = 0;                        // lanes of the groups that go on sifting / whose hole has children in HBM
#pragma unroll
    for (int b = 0; b < NSTEP; b++) {
      if (b > 0 && actm == 0) break;                      // wave-uniform: no group of this wavefront goes deeper
      // Straight-line code without validity tests (round 5): the keys beyond the heap's end are +inf (pad()), so a missing child
      // loses every comparison by itself; a group whose hole has stopped looks at the children of the hole again, finds the same
      // "not smaller than the moving key" and moves nothing; lanes with nothing to move write their entry to slot 0.  Only the
      // last step, and only where its parents' children can lie beyond the LDS array (c0 < 2^(LV NSTEP + 1); the 512-slot forms
      // never: 2^9), tests for that: such children do not exist, or live in the HBM level of a hybrid heap (below).
      const int s = (p << dq) + oq;                        // this lane's parent slot
      const int c0 = 2 * s;
      constexpr bool CHK = (1 << (LV * NSTEP + 1)) > CAP;
      const bool lds = !(CHK && b == NSTEP - 1) || c0 < CAP;
      const int rs = lds ? c0 : 0;
      const float2 kk = *reinterpret_cast<const float2 *>(&keys[rs]);
      int n0, n1;
      if (sizeof(NT) == 2) {
        const unsigned v = *reinterpret_cast<const unsigned *>(&nodes[rs]);
        n0 = (int)(v & 0xffffu);
        n1 = (int)(v >> 16);
      } else {
        const int2 v = *reinterpret_cast<const int2 *>(&nodes[rs]);
        n0 = v.x;
        n1 = v.y;
      }
      const float k0 = lds ? kk.x : INFINITY, k1 = lds ? kk.y : INFINITY;
      const bool right = k0 > k1;                          // left child strictly greater -> the hole goes right
      const float ck = right ? k1 : k0;
      const int cn = right ? n1 : n0;
      const unsigned long long gtm = wballot(right);
      const unsigned long long ltm = wballot(ck < mvk);
      const unsigned gt = (unsigned)(gtm >> gsh), lt = (unsigned)(ltm >> gsh) & GMASK;
      const lmask minem = wballot((gt & G) == E) & wballot((lt & A) == A);   // (an inactive group has ck = +inf everywhere: nobody moves)
      const bool mine = lanes(minem);
      const int dst = mine ? s : 0;
      keys[dst] = ck;
      nodes[dst] = (NT)cn;
      const unsigned mb = (unsigned)(minem >> gsh) & GMASK;   // <= one lane per level, levels 1..nm
      // (the flags of the next step as ballots of comparisons made in THIS block: a flag set under the branch below would come back
      // as a 0 / 1 integer.)  The hole went down all LV levels <=> a parent of the subtree's last level moved its child up
      constexpr unsigned LASTLV = ((1u << (GP - 1)) - 1u) & ~((1u << (GP / 2 - 1)) - 1u);
      const lmask mvd = wballot(mb != 0), full = wballot((mb & LASTLV) != 0);
      if (mb != 0) {
        const int qs = 32 - __clz(mb);                    // deepest parent whose child moved up: the hole is at that child now
        const int rt = 2 * qs + (int)((gt >> (qs - 1)) & 1u);
        const int dt = 31 - __clz(rt);
        p = (p << dt) + rt - (1 << dt);
      }
      const lmask kids = wballot(2 * p <= ntr);
      actm = full & kids;
      if (HYB) deepm = (deepm & ~mvd) | (mvd & kids & wballot(p >= CAP / 2));   // on the last LDS level, with children: they are in HBM
    }
    if (HYB) {
      // the hole reached the last level of the LDS part and has children: they live in HBM.  One sequential step of downtree
      // (:857-866) per HBM level: the smaller child, ties to the left, moves up if it is smaller than the moving key.  A child
      // that moves up INTO the LDS part keeps its word (lazy, like every move of the parallel steps: march() finds it one level
      // above the recorded slot); one that moves from the lower to the upper HBM level stores its slot, so that the words of the
      // entries in the HBM levels are exact -- march() cannot search there.
#pragma unroll
      for (int h = 0; h < NH; h++) {
        if (__builtin_expect(deepm == 0, 1)) break;   // (the usual pop of a 256 x 256 field ends in the LDS part)
        int again = 0;
        if (lanes(deepm)) {
          const HEnt *ch = ovf + (2 * p - CAP);
          const HEnt c0 = ch[0];
          HEnt c1 = HEnt{INFINITY, 0};
          if (2 * p + 1 <= ntr) c1 = ch[1];
          const bool right = c0.key > c1.key;
          const float ck = right ? c1.key : c0.key;
          const int cn = right ? c1.node : c0.node;
          if (ck < mvk) {
            if (h == 0) {
              keys[p] = ck;
              nodes[p] = (NT)cn;
            } else if (g0) {
              ovf[p - CAP] = HEnt{ck, cn};
              set_slot((unsigned)cn, p);
            }
            p = 2 * p + (right ? 1 : 0);
            again = (NH > h + 1 && 2 * p <= ntr) ? 1 : 0;
          }
        }
        deepm = NH > h + 1 ? wballot(again != 0) : 0;
      }
    }
    const bool phi = HYB && p >= CAP;
    const int lp = phi ? 0 : p;
    keys[lp] = mvk;
    nodes[lp] = mvc;
    if (HYB && __builtin_expect(wballot(phi) != 0, 0)) {
      if (phi && g0) ovf[p - CAP] = HEnt{mvk, (int)mvc};
    }
    fin_node = (int)mvc;
    fin_slot = p;
  }
};

// fouds2 for one quadrant: inv/CalSurfG.f90:586-723.  (tj,sj) = neighbour along x, (tj2,sj2) the
// node behind it, (tk,..) along z; vj2/vk2 = second node inside the grid.
//
// Straight-line form (round 3).  The reference distinguishes eight cases by which of the two directions has an alive neighbour
// and whether that direction is second order; the sixteen lanes of a group are in different cases at almost every pop, so a
// branch per case costs the wavefront the sum of all eight bodies plus the exec-mask bookkeeping (~100 VALU + ~45 scalar
// instructions of the 364 + 150 per wave-pop).  Here every lane evaluates ONE set of expressions whose operands are selected
// per case -- each fp32 operation is the one the reference's case performs, in its order:
//   two-sided (both directions alive), reference cases with u, v, em as below:
//     a = v*v + k9*(u*u)            k9 = 1 (both second order / both first order: v*v + u*u), 9 (mixed: v*v + 9*(u*u))
//     b = (kb*em)*(u*u)             kb = 2 (2*em*(u*u)), 6 (6*em*(u*u)), -2 (-2*(u*u)*em: x2 is exact, so the order of the two
//                                   factors does not change the one rounding)
//     c = (u*u)*(em*em - (s*s)*(v*v))
//     both second order : u = 2 ri dnx, v = 2 risti dnz, em = ((4tj - tj2) - 4tk) + tk2, tref = 4tj - tj2, tdiv = 3
//     j second, k first : u = risti dnz, v = 2 ri dnx,  em = (3tk - 4tj) + tj2,          tref = tk
//     j first, k second : u = ri dnx,    v = 2 risti dnz, em = (3tj - 4tk) + tk2,        tref = tj
//     both first order  : u = ri dnx,    v = risti dnz,  em = tk - tj,                   tref = tj
//     (2*ri*dnx = 2*(ri*dnx) exactly: scaling by two commutes with rounding)
//   one-sided: a = 1, b = 0 and
//     second order: c = -(u*u)*(s*s) with u = 2 ri dnx (tref = 4tj - tj2) or 2 risti dnz (tref = 4tk - tk2), tdiv = 3
//     first order : c = -(s*s)*(ri*ri)*(dnx*dnx), tref = tj, or -(s*s)*(risti*risti)*(dnz*dnz), tref = tk
// then the common tail tdsh = (-b + sqrt(max(b*b - 4*a*c, 0))) / (2*a), (tref + tdsh) / tdiv.  Bit-identical fields
// (tests/test_fmm_gpu.py against the oracle on every grid size; option-free, so the old form is kept below for reference only
// under DZ_FMM_QUADRANT_BRANCHES).
#ifndef DZ_FMM_QUADRANT_BRANCHES
// Exact fp32 division and square root without the range handling the compiler wraps around them (round 4).  x / y is compiled as
// v_div_scale x 2, v_rcp, the Newton / residual steps below, v_div_fmas, v_div_fixup: the two scale instructions and the fixup only
// act on operands near the ends of the exponent range (|y| or |x / y| below 2^-126 or above 2^126, |x| < 2^-103), zeros, infinities
// and NaNs; in between they pass their operands through and the quotient is the one of the FMA steps, which are repeated here as
// they stand.  run_fmm only lets this form run when the grid's steps and velocities keep every operand within 2^+-30 of one
// (`fast` false: the compiler's sequences).  Division by three: q = RN(x / 3), then one residual step -- equal to x / 3.0f for every
// finite float (2^32 cases, tools/check_div3.c).  Square root: v_sqrt_f32 is within one ulp; the two residual tests pick the
// correctly rounded neighbour (the compiler's own sequence minus its rescaling of arguments below 2^-96 and its class test).
__device__ __forceinline__ float div_exact(float x, float y, int fast) {
  if (!fast) return x / y;
  float r = __builtin_amdgcn_rcpf(y);
  const float e = __builtin_fmaf(-y, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  float q = x * r;
  const float e2 = __builtin_fmaf(-y, q, x);
  q = __builtin_fmaf(e2, r, q);
  const float e3 = __builtin_fmaf(-y, q, x);
  return __builtin_fmaf(e3, r, q);
}
__device__ __forceinline__ float div3_exact(float x) {
  const float third = 1.0f / 3.0f;
  const float q = x * third;
  return __builtin_fmaf(__builtin_fmaf(-3.0f, q, x), third, q);
}
__device__ __forceinline__ float sqrt_exact(float x, int fast) {
  if (!fast) return sqrtf(x);
  const float s = __builtin_amdgcn_sqrtf(x);
  const float sm = __int_as_float(__float_as_int(s) - 1), sp = __int_as_float(__float_as_int(s) + 1);
  const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
  float o = rm <= 0.0f ? sm : s;
  o = rp > 0.0f ? sp : o;
  return o;
}

// tj, tj2, tk, tk2: the words of the four nodes read as floats (the time, when the node is alive); aj .. ak2: alive and inside the grid
__device__ __forceinline__ float quadrant_time(float slown, float risti, float dnx, float dnz, float tj, float tj2, float tk,
                                               float tk2, bool aj, bool aj2, bool ak, bool ak2, int fast) {
  const float ri = EARTH;
  const bool so2j = aj2 && aj && tj > tj2;
  const bool so2k = ak2 && ak && tk > tk2;
  const bool two = aj && ak, both2 = so2j && so2k, mixed = so2j != so2k, onlyj = aj && !ak;
  const float U1 = ri * dnx, U2 = U1 + U1, V1 = risti * dnz, V2 = V1 + V1;
  const float fourtj = 4.0f * tj, fourtk = 4.0f * tk;
  const float X2 = fourtj - tj2, Z2 = fourtk - tk2;
  const float ss = slown * slown;
  // ---- two-sided ----
  const float emA = (X2 - fourtk) + tk2;
  const float emBD = (3.0f * (so2j ? tk : tj) - (so2j ? fourtj : fourtk)) + (so2j ? tj2 : tk2);
  const float emE = tk - tj;
  const float em = both2 ? emA : (mixed ? emBD : emE);
  const float u = both2 ? U2 : (so2j ? V1 : U1);
  const float v = both2 ? V2 : (so2j ? U2 : (so2k ? V2 : V1));
  const float uu = u * u, vv = v * v;
  const float a2 = vv + (mixed ? 9.0f : 1.0f) * uu;
  const float b2 = ((both2 ? 2.0f : (mixed ? 6.0f : -2.0f)) * em) * uu;
  const float c2 = uu * (em * em - ss * vv);
  const float tref2 = both2 ? X2 : (so2j ? tk : tj);
  // ---- one-sided ----
  // so1 = onlyj ? so2j : so2k.  Only used when exactly one direction is alive, and then the other direction's flag is false anyway
  // (so2j needs aj, so2k needs ak): an OR, where a select between two predicates would go through 0 / 1 integers
  const bool so1 = so2j || so2k;
  const float u1 = onlyj ? U2 : V2;
  const float c1s = -(u1 * u1) * ss;
  const float c1f = (-ss * (onlyj ? ri * ri : risti * risti)) * (onlyj ? dnx * dnx : dnz * dnz);
  const float c1 = so1 ? c1s : c1f;
  const float tref1 = onlyj ? (so2j ? X2 : tj) : (so2k ? Z2 : tk);
  // ---- common tail ----
  const float a = two ? a2 : 1.0f, b = two ? b2 : 0.0f, c = two ? c2 : c1;
  const float tref = two ? tref2 : tref1;
  const bool third = both2 || (!two && so1);       // two ? both2 : so1 -- tdiv = 3 (else 1)
  float rd1 = b * b - 4.0f * a * c;
  if (rd1 < 0.0f) rd1 = 0.0f;
  // (the short forms first, the compiler's sequences as a rare override behind ONE not-taken branch: written as the two arms of an
  // if / else -- or as a test inside sqrt_exact and another inside div_exact -- the loop got both arms inline behind a flag, two
  // taken branches per pop; `fast` is a per-lane copy of the launch's flag so that the test does not fetch a spilled scalar)
  float tdsh = div_exact(-b + sqrt_exact(rd1, 1), 2.0f * a, 1);
  int fast_here = fast;
  asm volatile("" : "+v"(fast_here));   // (compared here: hoisted out of the loop, the lane mask of the comparison is a spilled scalar pair again)
  if (__builtin_expect(wballot(fast_here == 0) != 0, 0)) tdsh = div_exact(-b + sqrt_exact(rd1, 0), 2.0f * a, 0);
  const float tsum = tref + tdsh;
  const float t3 = div3_exact(tsum);
  const float t = third ? t3 : tsum;
  return (aj || ak) ? t : INFINITY;
}
#endif

#ifdef DZ_FMM_LAZYSTAT
__device__ unsigned long long g_lazy_stat[4];
#endif
#ifdef DZ_FMM_PROF   // experiment-only build: per-phase shader-clock totals of the marching loop
__device__ unsigned long long g_fmm_prof[8];
#define PROF_DECL unsigned long long pt_ = __builtin_amdgcn_s_memtime(), pa_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PROF(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pa_[i] += n_ - pt_; pt_ = n_; } while (0)
#define PROF_WAIT asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define PROF_FLUSH do { if (lane == 0) for (int i_ = 0; i_ < 8; i_++) atomicAdd(&g_fmm_prof[i_], pa_[i_]); } while (0)
#elif defined(DZ_FMM_MARK)   // experiment-only build: phase markers in the assembly (tools/fmm_phase_count.py counts the instructions between them)
#define PROF_DECL
#define PROF(i) asm volatile("; MARK " #i ::: "memory")
#define PROF_WAIT
#define PROF_FLUSH
#else
#define PROF_DECL
#define PROF(i)
#define PROF_WAIT
#define PROF_FLUSH
#endif

// (Round 3 measured cache-policy hints on the node accesses of the marching loop -- non-temporal loads -17 %, stores -32 %,
// both -51 %, tools/exp_fmm_nt.sh on the 8-byte records of that time: L2 / Infinity Cache residency carries them.)

// ---- one marching run (travel, inv/CalSurfG.f90:356-456), executed by a 16-lane group --------
// REFINED: urg=1 early-exit rule on the edges flagged in `ex` (bit0 x=1, bit1 x=nnx, bit2 z=1,
// bit3 z=nnz).
template <int CAP, bool SPILL, class NT, bool HYB, bool REFINED, int GPL>
__device__ __forceinline__ bool march(Heap<CAP, SPILL, NT, HYB, GPL> &H, const float2 *__restrict__ slow,
                                      int nnx, int nnz, float dnx, float dnz,
                                      int ex, int lane, int fastm, int maxpop = 0x7fffffff) {
  // Lanes of a field's group: 16 = 4 neighbours x 4 quadrants (jd, kd); 8 = 4 neighbours x 2 (jd), each lane solving the
  // quadrants kd = -1 and kd = +1 one after the other (eight fields per wavefront: the per-pop bookkeeping is shared by twice the
  // fields)
  static_assert(GPL == 16 || (GPL == 8 && !SPILL), "lanes per field");
  constexpr int GP = GPL, NBL = GPL / 4;                   // lanes per neighbour
  constexpr unsigned GMASK = (1u << GPL) - 1u, OWNERS = GPL == 16 ? 0x1111u : 0x55u;
  const int gl = lane & (GP - 1), gbase = lane & ~(GP - 1);
  const int nb = gl / NBL, q = gl & (NBL - 1);
  const int dix = nb == 0 ? -1 : (nb == 1 ? 1 : 0);
  const int diz = nb == 2 ? -1 : (nb == 3 ? 1 : 0);
  const int jd = GPL == 16 ? ((q & 2) ? 1 : -1) : (q ? 1 : -1), kd = GPL == 16 ? ((q & 1) ? 1 : -1) : -1;
  const int tsh = H.tsh;
  bool overflow = false;
  // lazy back-pointers (see below) wherever the sift-down is the parallel one: the all-in-LDS heap and, since the late round 3,
  // the hybrid heap of the 342..682-node grids too (S-512: 6 790 -> 7 530 fields/s in a same-box A/B, bit-identical; an entry of
  // the HBM level that moves up into LDS is found one level above the slot its word holds, one that stays is not found and keeps it)
  constexpr bool LAZY = !SPILL;
  // the interior-root table (see the loop): record-index differences by lane class and coordinate residue
  constexpr bool TAB = Heap<CAP, SPILL, NT, HYB, GPL>::TAB;
  const bool tab_ok = TAB && nnx >= 8 && nnz >= 8 && tsh <= 14;   // (16-bit table entries: tile strides up to 2^14 records)
  const unsigned XMASK = ~((1u << tsh) - 1u) | 0xCu, ZMASK = ((1u << tsh) - 1u) & ~0xCu;   // the x and z bits of a record index
  // (both parts are monotone in their coordinate: one unsigned range test each; a grid too small to have an interior never passes)
  const unsigned xlo = tab_ok ? (unsigned)tile_x(3, tsh) : 0xffffffffu, xspan = tab_ok ? (unsigned)tile_x(nnx - 4, tsh) - xlo : 0u;
  const unsigned zlo = (unsigned)tile_z(3), zspan = tab_ok ? (unsigned)tile_z(nnz - 4) - zlo : 0u;
  const short *xt = nullptr, *zt = nullptr;
  if constexpr (TAB) {
    short *tab = H.tab;
    cbar();
    for (int e = gl; e < 48; e += GP) {   // (every group writes the whole table: any of them may be the only one that marches)
      const int l = e < 24 ? e : e - 24;
      const int c = l >> 2, m = l & 3, d = (c >> 1) - 1, sg = (c & 1) ? 1 : -1;   // class = (dix or diz, jd or kd), residue
      const int c0 = 4 + m, cn = c0 + d;
      if (e < 24) {
        tab[l] = (short)(tile_x(cn, tsh) - tile_x(c0, tsh));
        tab[24 + l] = (short)(tile_x(cn + sg, tsh) - tile_x(cn, tsh));
        tab[48 + l] = (short)(tile_x(cn + 2 * sg, tsh) - tile_x(cn, tsh));
      } else {
        tab[72 + l] = (short)(tile_z(cn) - tile_z(c0));
        tab[96 + l] = (short)(tile_z(cn + sg) - tile_z(cn));
        tab[120 + l] = (short)(tile_z(cn + 2 * sg) - tile_z(cn));
      }
    }
    cbar();
    xt = tab + ((dix + 1) * 2 + (jd > 0 ? 1 : 0)) * 4;
    zt = tab + 72 + ((diz + 1) * 2 + (kd > 0 ? 1 : 0)) * 4;
  }
  PROF_DECL;
  int npop = 0;
  while (H.ntr > 0 && !overflow && npop < maxpop) {
    npop++;
    cbar();
    PROF(7);
    const HEnt root = H.get(1);
    const int iroot = root.node;                            // record index of the node being accepted
    const unsigned uroot = (unsigned)iroot;
    // ---- stencil loads first: lane (nb,q) of the group reads its neighbour and the 4 nodes behind
    // it.  All loads are issued unconditionally (invalid lanes read the root's own record) and
    // stay in flight while the root is sifted down in LDS. ----
    unsigned uself, aj_i, aj2_i, ak_i, ak2_i;               // record indices of the neighbour and of the nodes behind it along x and z
    lmask nvalidm, vjm, vj2m, vkm, vk2m;                     // ... and whether they lie inside the grid
    bool vkp = false, vk2p = false;
    unsigned akp_i = uroot, ak2p_i = uroot;                  // (8 lanes per field: the lane's second quadrant, kd = +1)
    // Interior roots (round 5): a root at least three nodes from every edge has its whole stencil inside the grid, and the record
    // index of node (x + d) differs from that of node x by an amount that depends on x & 3 and d alone (4 x 4 tiles).  Those
    // differences sit in a 288-byte LDS table per wavefront (built at the start of the march: 6 lane classes x 4 residues for
    // each of the three x parts and the three z parts), so the usual pop forms its five indices with six LDS reads and six
    // additions instead of decoding the root, offsetting, range-testing and re-tiling five coordinates (66 -> ~30 VALU
    // instructions in this phase).  One edge root among the four fields (17 % of the pops of a 256 x 256 grid) sends the wavefront
    // through the coordinate path.
    bool interior = false;
    if constexpr (TAB) {
      const lmask inm = wballot(((uroot & XMASK) - xlo) <= xspan) & wballot(((uroot & ZMASK) - zlo) <= zspan);
      interior = inm == wballot(true);
    }
    // (the table path is written as straight-line code in front of an `if` without `else`: as the two arms of an if / else the
    // compiler placed BOTH out of line behind a flag, two taken branches and three scalar instructions per pop; an edge root
    // reads the table for nothing)
    if constexpr (TAB) {
      const short *xa = xt + ((uroot >> 2) & 3u), *za = zt + (uroot & 3u);
      const int dnx_ = xa[0], dj_ = xa[24], dj2_ = xa[48];
      const int dnz_ = za[0], dk_ = za[24], dk2_ = za[48];
      uself = uroot + (unsigned)dnx_ + (unsigned)dnz_;
      aj_i = uself + (unsigned)dj_;
      aj2_i = uself + (unsigned)dj2_;
      ak_i = uself + (unsigned)dk_;
      ak2_i = uself + (unsigned)dk2_;
      nvalidm = vjm = vj2m = vkm = vk2m = ~0ull;
    }
    if (!TAB || __builtin_expect(!interior, 0)) {
      const int ix = rid_x0(iroot, tsh) + 1, iz = rid_z0(iroot, tsh) + 1;
      if (REFINED) {
        bool swrg = false;
        if (ix == 1 && (ex & 1)) swrg = true;
        if (ix == nnx && (ex & 2)) swrg = true;
        if (iz == 1 && (ex & 4)) swrg = true;
        if (iz == nnz && (ex & 8)) swrg = true;
        if (swrg) {  // nsts(iz,ix)=0 ; EXIT  -- the band keeps its slots in nstsr (:378-381)
          if (H.g0) H.stw(uroot, w_alive(root.key));            // (it stays at slot 1 of the heap; its time is its key)
          if (LAZY)   // lazy back-pointers (below): the words of entries that only moved up are behind; nstsr wants them exact
            for (int i = 2 + gl; i <= H.ntr; i += GP) H.stw((unsigned)H.get(i).node, w_band(i));
          break;
        }
      }
      const int nix = ix + dix, niz = iz + diz;
      // 0-based coordinates and one unsigned comparison per range test ((unsigned)(x) < n  <=>  0 <= x < n)
      const int nx0 = nix - 1, nz0 = niz - 1;
      const unsigned unx = (unsigned)nnx, unz = (unsigned)nnz;
      const bool nvalid = (unsigned)nx0 < unx && (unsigned)nz0 < unz;
      const int j0 = nx0 + jd, j20 = nx0 + 2 * jd, k0 = nz0 + kd, k20 = nz0 + 2 * kd;
      const bool vj = nvalid && (unsigned)j0 < unx, vj2 = vj && (unsigned)j20 < unx;
      const bool vk = nvalid && (unsigned)k0 < unz, vk2 = vk && (unsigned)k20 < unz;
      // tiled record indices: an X part per column (neighbour, +-1, +-2) and a Z part per row; out-of-grid coordinates give
      // garbage that the validity flags replace by the root's own record.  Unsigned indices: no sign extension per address.
      const int xn = tile_x(nx0, tsh), xj = tile_x(j0, tsh), xj2 = tile_x(j20, tsh);
      const int zn = tile_z(nz0), zk = tile_z(k0), zk2 = tile_z(k20);
      uself = nvalid ? (unsigned)(xn + zn) : uroot;
      aj_i = vj ? (unsigned)(xj + zn) : uroot;
      aj2_i = vj2 ? (unsigned)(xj2 + zn) : uroot;
      ak_i = vk ? (unsigned)(xn + zk) : uroot;
      ak2_i = vk2 ? (unsigned)(xn + zk2) : uroot;
      nvalidm = wballot((unsigned)nx0 < unx) & wballot((unsigned)nz0 < unz);
      vjm = nvalidm & wballot((unsigned)j0 < unx);
      vj2m = vjm & wballot((unsigned)j20 < unx);
      vkm = nvalidm & wballot((unsigned)k0 < unz);
      vk2m = vkm & wballot((unsigned)k20 < unz);
      if constexpr (GPL == 8) {
        const int kp0 = nz0 + 1, kp20 = nz0 + 2;
        vkp = nvalid && (unsigned)kp0 < unz;
        vk2p = vkp && (unsigned)kp20 < unz;
        akp_i = vkp ? (unsigned)(xn + tile_z(kp0)) : uroot;
        ak2p_i = vk2p ? (unsigned)(xn + tile_z(kp20)) : uroot;
      }
    }
    const bool nvalid = lanes(nvalidm), vj = lanes(vjm), vj2 = lanes(vj2m), vk = lanes(vkm), vk2 = lanes(vk2m);
    // accepted: the word becomes the time (= the heap key, the trial time).  Stored by all sixteen lanes of the group (same address,
    // same value: one request) -- a store by lane 0 alone costs the pop an exec save / test / restore
    H.stw(uroot, w_alive(root.key));
    cbar();
    // (the node words as they are: an alive node's word is its time, sign bit clear; see w_alive)
    unsigned wself = H.ldw(uself);
    unsigned wj = H.ldw(aj_i);
    unsigned wj2 = H.ldw(aj2_i);
    unsigned wk = H.ldw(ak_i);
    unsigned wk2 = H.ldw(ak2_i);
    unsigned wkp = 0, wk2p = 0;
    if constexpr (GPL == 8) {
      wkp = H.ldw(akp_i);
      wk2p = H.ldw(ak2p_i);
    }
    const float2 slri = slow[uself];                        // slowness of the neighbour (1/velocity, precomputed, same tiling) and its risti
    const float vel = slri.x, risti = slri.y;
    int nbn[4], nbs[4], nbm[4];
    float nbt[4];
    // the four neighbours' record indices, from their owner lanes (nb, q = 0); -1 outside the grid.  Needed before the pop only
    // by the spill kernel's sequential sift-down (which entry moved where); the parallel kernel fetches them in its slow path.
    const int nrid = nvalid ? (int)uself : -1;
    if (SPILL) {
      nbn[0] = own_i<GPL, 0>(nrid); nbn[1] = own_i<GPL, 1>(nrid);
      nbn[2] = own_i<GPL, 2>(nrid); nbn[3] = own_i<GPL, 3>(nrid);
    } else {
      nbn[0] = nbn[1] = nbn[2] = nbn[3] = 0;
    }
#pragma unroll
    for (int n = 0; n < 4; n++) nbm[n] = 0;
    int mynode = 0, myslot = 0, nmoves = 0;
    constexpr int TOT = Heap<CAP, SPILL, NT, HYB, GPL>::TOT;
    int fin_node = 0, fin_slot = 0;
    PROF(0);
    if (SPILL) {
      H.pop_root(gl, nbn, nbm, mynode, myslot, nmoves);
    } else {
      H.pop_root_par(lane, fin_node, fin_slot);
      cbar();
    }
    PROF(1);
    PROF_WAIT;
    PROF(2);
    // keep the compiler from sinking the loads behind a test of the first one and from hoisting
    // the deferred stores above this point: all results are "used" here, together
    if constexpr (GPL == 8)
      asm volatile("" : "+v"(wself), "+v"(wj), "+v"(wj2), "+v"(wk), "+v"(wk2), "+v"(wkp), "+v"(wk2p) : "v"(vel), "v"(risti) : "memory");
    else
      asm volatile("" : "+v"(wself), "+v"(wj), "+v"(wj2), "+v"(wk), "+v"(wk2) : "v"(vel), "v"(risti) : "memory");
    if (SPILL) {
      if (gl < nmoves) H.set_slot((unsigned)mynode, myslot);   // deferred back-pointers of the sift-down
    } else {
      // (the entries the sift-down moved UP get no store, see below; the dropped entry moved down and gets one)
      if (H.g0 && fin_slot > 0) H.set_slot((unsigned)fin_node, fin_slot);
    }
    const lmask fslotm = wballot(fin_slot > 0);            // (ballots are taken in the block that makes the comparison, see lanes())
    // status of the neighbour: -1 far, 0 alive / outside, > 0 heap slot.  A word that is not alive has its sign bit set, and its low
    // 31 bits sign-extended are the status: 0xffffffff -> -1, 0x80000000 | slot -> slot
    const bool nopen = nvalid && !w_is_alive(wself);
    int stfix = nopen ? ((int)(wself << 1) >> 1) : 0;
    const bool aj = vj && w_is_alive(wj), aj2 = vj2 && w_is_alive(wj2), ak = vk && w_is_alive(wk), ak2 = vk2 && w_is_alive(wk2);
    // Lazy back-pointers (round 3).  A sift-down moves ~9 entries up one level each, and the reference stores the new slot of
    // every one of them in its node's status -- nine 4-byte writes into nine random lines per pop, half of this kernel's HBM
    // traffic.  Here the record of an entry that moves UP (child slot -> parent slot) is left alone, so a band node's record
    // holds a slot whose ancestor-or-self is the entry's true slot; every other movement still stores at once (the dropped
    // last entry, entries pushed down by a rising one, new and rising entries themselves).  The true slot is looked up when it
    // is needed, after the sift-down: the four lanes of the neighbour's quad read the node ids at srec >> k, k = q, q+4, q+8(,
    // q+12) -- ids are unique in the heap, so the one that matches is it.  The entry the pop dropped into the hole is
    // recognised by its id (its record was loaded before it moved).  The heap array itself evolves exactly as before; the
    // refined march restores exact slots when it leaves its band in nstsr (above).
    // (the reads go out here and are looked at after the quadrant solve: their latency hides behind it)
    const int srec = stfix;
    const bool band = srec > 0;
    const lmask bandm = wballot(srec > 0);
    constexpr int LEV = 31 - __builtin_clz((unsigned)(TOT - 1));   // deepest level of the heap (root = level 0)
    constexpr int LT = LEV / 4 + 1;
    int lz_id0 = 0, lz_id1 = 0;
    // (only slots 1 .. ntr are entries: a recorded slot may lie beyond the heap's present end, and what sits there is stale --
    // after a time-sliced hand-over even another field's entries, whose ids can coincide with this neighbour's)
    auto lz_ok = [&](int a) { return (unsigned)(a - 1) < (unsigned)H.ntr && (!HYB || a < CAP); };
    if (LAZY) {   // k = q first: an entry rarely moves up more than three levels between two stores of its slot
      // (8 lanes per field: the two lanes of the neighbour read two ancestors each, k = 2q and 2q + 1)
      const int a = band ? (srec >> (GPL == 16 ? q : 2 * q)) : 0;
      lz_id0 = (int)H.nodes[lz_ok(a) ? a : 0];
      if constexpr (GPL == 8) lz_id1 = (int)H.nodes[lz_ok(a >> 1) ? (a >> 1) : 0];
    }
    float trav = INFINITY;
    PROF(3);
    if (vj && vk && nopen)
      trav = quadrant_time(vel, risti, dnx, dnz, __int_as_float((int)wj), __int_as_float((int)wj2), __int_as_float((int)wk),
                           __int_as_float((int)wk2), aj, aj2, ak, ak2, fastm);
    if constexpr (GPL == 8) {
      const bool akp = vkp && w_is_alive(wkp), ak2p = vk2p && w_is_alive(wk2p);
      float travp = INFINITY;
      if (vj && vkp && nopen)
        travp = quadrant_time(vel, risti, dnx, dnz, __int_as_float((int)wj), __int_as_float((int)wj2), __int_as_float((int)wkp),
                              __int_as_float((int)wk2p), aj, aj2, akp, ak2p, fastm);
      trav = fminf(trav, travp);
    }
    trav = fmin_xor1(trav);
    if constexpr (GPL == 16) trav = fmin_xor2(trav);
    if (LAZY) {   // the true slot of a band neighbour (lazy back-pointers, above)
      int found = 64;
      {
        const int k0 = GPL == 16 ? q : 2 * q;
        const int a = band ? (srec >> k0) : 0;
        if constexpr (GPL == 8)
          if (lz_ok(a >> 1) && lz_id1 == (int)uself) found = k0 + 1;
        if (lz_ok(a) && lz_id0 == (int)uself) found = k0;
      }
      found = imin_xor1(found);
      if constexpr (GPL == 16) found = imin_xor2(found);
      const lmask dropm = wballot((int)uself == fin_node) & fslotm;
      const bool isdrop = lanes(dropm);
#ifdef DZ_FMM_LAZYSTAT   // experiment build: how often the first four ancestors do not hold the entry (lanes), and pops
      if (q == 0 && band) atomicAdd(&g_lazy_stat[found == 64 && !isdrop ? 1 : 0], 1ull);
      if (q == 0 && band && isdrop) atomicAdd(&g_lazy_stat[2], 1ull);
#endif
      if (__builtin_expect((bandm & wballot(found == 64) & ~dropm) != 0, 0)) {   // (wave-uniform, rare) the higher ancestors
#pragma unroll
        for (int t = 1; t < LT; t++) {
#pragma unroll
          for (int u = 0; u < 4 / NBL; u++) {                // (8 lanes per field: two ancestors per lane and round)
            const int k = (4 / NBL) * q + u + 4 * t;
            const int a = band ? (srec >> k) : 0;
            const bool ok = lz_ok(a);
            const int id = (int)H.nodes[ok ? a : 0];
            if (ok && id == (int)uself) found = found < k ? found : k;
          }
        }
        found = imin_xor1(found);
        if constexpr (GPL == 16) found = imin_xor2(found);
      }
      if (band) {
        if (isdrop) stfix = fin_slot;
        else if (found < 64) stfix = srec >> found;        // (else, hybrid heap: still at srec in the HBM level)
        // (... unless THIS pop's sift-down took it from the lower HBM level up to the upper one: the word was loaded before)
        else if (HYB && Heap<CAP, SPILL, NT, HYB, GPL>::NH > 1 && srec >= 2 * CAP && srec == fin_slot) stfix = srec >> 1;
      }
    }
    PROF(4);
    // ---- the (up to) four heap updates in the reference's order (x-1, x+1, z-1, z+1) ----
    // Lane (nb, q=0) "owns" neighbour nb: it holds that node's status (nself.s) and new time (trav).
    bool fast = false;
    int n0 = 0;                                            // first neighbour handled by the sequential code
    if (!SPILL) {
      // all-in-LDS heap.  Usual case: none of the four entries has to rise above its parent (the new times
      // lie at the far side of the band), so the four addtree/updtree calls reduce to independent writes,
      // done by the owner lanes.  A parent that is itself an earlier neighbour (m < nb) is compared with
      // its new key, as the sequential order would.  Any rise in the group -> the sequential code below.
      constexpr lmask OWN64 = GPL == 16 ? 0x1111111111111111ull : 0x5555555555555555ull;   // the owner lanes (q == 0)
      const bool owner = q == 0;
      const bool act = stfix != 0, isnew = stfix < 0;          // (stfix: the neighbour's true slot, resolved above)
      const lmask actm = wballot(stfix != 0), ownact = OWN64 & actm;
      const unsigned newb = (unsigned)((wballot(stfix < 0) & OWN64) >> gbase) & GMASK;
      const int cnt = __popc(newb);
      const int c = isnew ? H.ntr + 1 + __popc(newb & ((1u << gl) - 1u)) : stfix;
      const lmask roomm = wballot(H.ntr + cnt < TOT);
      const bool room = lanes(roomm);
      const int pc = c >> 1;
      constexpr int NH = Heap<CAP, SPILL, NT, HYB, GPL>::NH;
      const bool pchi = HYB && NH > 1 && act && room && pc >= CAP;   // (two HBM levels: the parent of a slot of the lower one)
      float pk = H.keys[(act && room && !pchi) ? pc : 0];
      // (for the one-level rise below: the grandparent's key and the parent's node, requested with the parent's key so that
      // they cost no LDS round trip of their own)
      const int gp = pc >> 1;
      const float gk = H.keys[(act && room && gp >= 1) ? gp : 0];
      NT pn = H.nodes[(act && room && !pchi) ? pc : 0];
      if (HYB && NH > 1 && __builtin_expect(wballot(pchi) != 0, 0)) {
        if (pchi) {
          const HEnt e = H.ovf[pc - CAP];
          pk = e.key;
          pn = (NT)e.node;
        }
      }
      const int cact = act ? c : 0;
      const int c0 = own_i<GPL, 0>(cact), c1 = own_i<GPL, 1>(cact), c2 = own_i<GPL, 2>(cact);
      const float t0 = own_f<GPL, 0>(trav), t1 = own_f<GPL, 1>(trav), t2 = own_f<GPL, 2>(trav);
      if (nb > 0 && pc == c0) pk = t0;
      if (nb > 1 && pc == c1) pk = t1;
      if (nb > 2 && pc == c2) pk = t2;
      const lmask risem = ownact & wballot(c > 1) & wballot(trav < pk);
      const bool rise = lanes(risem);
      const unsigned riseb = (unsigned)(risem >> gbase) & GMASK;
      // neighbours before the first rising one (n < n0) are written directly; from n0 on, sequentially
      n0 = !room ? 0 : (riseb ? (__builtin_ctz(riseb) / NBL) : 4);
      // One-level rise in place (round 3).  41 % of the wave-pops have a rising entry in some group, and in 92 % of those every
      // such group has exactly ONE riser that stops after one level and whose move touches no slot another neighbour of the pop
      // reads or writes (measured, DZ_FMM_PROF2): then the sequential addtree/updtree calls still reduce to independent
      // writes -- the riser swaps with its parent, everybody else writes as above -- instead of up to four parallel rounds.
      // Conditions, per group (r = the riser, at slot c_r with parent slot p_r and grandparent slot g_r):
      //   * one riser, and its key is not smaller than the key at g_r (it stops at p_r);
      //   * no other active neighbour m sits at p_r or g_r (c_m) or has p_r or c_r as its parent slot: their comparisons, made
      //     against the array as it was, are then the ones the sequential order makes (neighbours before r see r's old
      //     entry, neighbours after r see slots r did not touch), and r's own comparisons see no slot an earlier neighbour wrote;
      //   * (hybrid heap) the riser's slot lies in the LDS part.
      lmask f2m = 0;
      const lmask anyrise = wballot(riseb != 0u) & roomm;
      if (anyrise != 0) {                                  // wave-uniform: some group has a rising entry
        const int c3 = own_i<GPL, 3>(cact);
        const int nbr = __builtin_ctz(riseb | (1u << GPL)) / NBL;
        const int cr = nbr == 0 ? c0 : (nbr == 1 ? c1 : (nbr == 2 ? c2 : c3));
        const int pr = cr >> 1, gr = cr >> 2;
        // okr = rise && !(gp >= 1 && trav < gk) && (!HYB || c < CAP);  clash = owner && act && !rise && (c == pr || c == gr || pc == cr || pc == pr)
        lmask okm = risem & ~(wballot(gp >= 1) & wballot(trav < gk));
        if (HYB) okm &= wballot(c < CAP);
        const lmask clm = ownact & ~risem & (wballot(c == pr) | wballot(c == gr) | wballot(pc == cr) | wballot(pc == pr));
        const unsigned okb = (unsigned)(okm >> gbase) & GMASK;
        const unsigned clb = (unsigned)(clm >> gbase) & GMASK;
        // f2 = room && riseb != 0 && exactly one riser && okb == riseb && clb == 0
        f2m = anyrise & wballot(((riseb & (riseb - 1u)) | (okb ^ riseb) | clb) == 0u);
        if (lanes(f2m)) n0 = 4;
      }
      fast = n0 == 4;
      const lmask f2rise = f2m & risem;
      if (f2rise != 0) {                                   // the risers: entry to the parent's slot, parent down to the entry's
        if (lanes(f2rise)) {
          H.keys[pc] = trav;
          H.nodes[pc] = (NT)uself;
          H.stw(uself, w_band(pc));
          H.keys[c] = pk;
          H.nodes[c] = pn;
          H.set_slot((unsigned)pn, c);
        }
      }
      {
        const lmask wrm = ownact & wballot(nb < n0) & ~f2rise;   // wr = owner && act && nb < n0 && !(f2 && rise)
        const lmask whim = HYB ? (wrm & wballot(c >= CAP)) : 0;   // (HYB) the entry's slot lies in the HBM level
        const bool wr = lanes(wrm);
        const int dst = lanes(wrm & ~whim) ? c : 0;        // slot 0 is never a heap entry
        H.keys[dst] = trav;
        H.nodes[dst] = (NT)uself;
        if (wr) H.stw(uself, w_band(c));
        if (HYB && __builtin_expect(whim != 0, 0)) {
          if (lanes(whim)) H.ovf[c - CAP] = HEnt{trav, (int)uself};
        }
        H.ntr += __popc(newb & ((1u << (NBL * n0)) - 1u));
      }
    }
    PROF(5);
#ifdef DZ_FMM_PROF
    if (wballot(!fast)) pa_[3] += 1000000;   // "fix" slot doubles as a counter of slow-path iterations (x1e6)
    pa_[7] += 1000000;                        // iterations (x1e6) on top of the loop-top ticks
#endif
#ifdef DZ_FMM_PROF2   // experiment: how many slow-path pops have, in every group that needs the rounds, exactly ONE rising entry (any
    // number of levels) whose path no other neighbour of the pop touches?  (counted x1e6 on the "setup+loads" slot; slow pops on "fix")
    if (!SPILL && wballot(!fast)) {
      const bool owner = q == 0;
      const bool act = stfix != 0, isnew = stfix < 0;
      const unsigned newb = (unsigned)(wballot(owner && isnew) >> gbase) & 0x1111u;
      const int ntr0 = H.ntr - __popc(newb & ((1u << (4 * n0)) - 1u));            // (H.ntr was advanced for the neighbours < n0)
      const int c = isnew ? ntr0 + 1 + __popc(newb & ((1u << gl) - 1u)) : stfix;
      const int pc = c >> 1;
      const float pk = H.keys[(act && pc < CAP) ? pc : 0];
      const bool rise = owner && act && c > 1 && trav < pk;
      const unsigned rb = (unsigned)(wballot(rise) >> gbase) & 0xffffu;
      const int rl = rb ? __builtin_ctz(rb) : 0;
      const int cr = __shfl(c, gbase + rl);
      const bool one = __popc(rb) == 1;
      auto anc = [&](int x) { if (x < 1 || x > cr) return false; const int d = __clz(x) - __clz(cr); return (cr >> d) == x; };
      const bool clash = owner && act && gl != rl && (anc(c) || anc(pc));
      const unsigned cl = (unsigned)(wballot(clash) >> gbase) & 0xffffu;
      const bool room = ntr0 + __popc(newb) < Heap<CAP, SPILL, NT, HYB, GPL>::TOT;
      const bool ok = fast || (room && one && cl == 0 && (!HYB || cr < CAP));
      if (wballot(!ok) == 0) pa_[0] += 1000000;
      if (wballot(!fast && !room) != 0) pa_[1] += 1000000;       // pops with a group out of room ("popdown" slot)
      if (wballot(!fast && room && !one) != 0) pa_[2] += 1000000; // ... with several risers in a group ("loadwait" slot)
    }
#endif
    if (!fast) {
      nbs[0] = own_i<GPL, 0>(stfix); nbs[1] = own_i<GPL, 1>(stfix);
      nbs[2] = own_i<GPL, 2>(stfix); nbs[3] = own_i<GPL, 3>(stfix);
      nbt[0] = own_f<GPL, 0>(trav); nbt[1] = own_f<GPL, 1>(trav);
      nbt[2] = own_f<GPL, 2>(trav); nbt[3] = own_f<GPL, 3>(trav);
      if (!SPILL) {
        nbn[0] = own_i<GPL, 0>(nrid); nbn[1] = own_i<GPL, 1>(nrid);
        nbn[2] = own_i<GPL, 2>(nrid); nbn[3] = own_i<GPL, 3>(nrid);
      }
      if (SPILL) {
#pragma unroll
        for (int n = 0; n < 4; n++)
          if (nbs[n] > 0 && nbm[n] > 0) nbs[n] = nbm[n];   // its entry moved during the sift-down
#pragma unroll
        for (int n = 0; n < 4; n++) {
          if (nbs[n] == 0 || n < n0) continue;
          const int node = nbn[n];
          int c = nbs[n];
          if (c < 0) {   // far -> close: appended at the bottom (addtree), else its key dropped in place (updtree)
            if (H.full()) {
              overflow = true;
              break;
            }
            H.ntr++;
            c = H.ntr;
          }
          H.template sift_up<true>(c, nbt[n], node, nbn, nbs, n);
        }
      } else {
        // all-in-LDS heap: one parallel round per remaining neighbour, in the reference's order (the four groups of the
        // wavefront run their n-th neighbour together; a round nobody needs is skipped)
#pragma unroll
        for (int n = 0; n < 4; n++) {
          bool live = nbs[n] != 0 && n >= n0 && !overflow;
          if (wballot(live) == 0) continue;
          int c = nbs[n];
          if (live && c < 0) {   // far -> close: appended at the bottom (addtree), else its key dropped in place (updtree)
            if (H.full()) {
              overflow = true;
              live = false;
            } else {
              H.ntr++;
              c = H.ntr;
            }
          }
          const int L = H.rise_par(live, gl, gbase, c, nbt[n], nbn[n]);
          // a pending neighbour whose entry sat on the path moved down one level with it
#pragma unroll
          for (int m = n + 1; m < 4; m++) {
            const int sm = nbs[m];
            if (live && sm > 0 && sm < c) {
              const int d = __clz(sm) - __clz(c);              // levels between the two slots
              if (d >= 1 && d <= L && (c >> d) == sm) nbs[m] = c >> (d - 1);
            }
          }
        }
      }
    }
    PROF(6);
  }
  PROF_FLUSH;
  if (H.g0 && H.popcnt) atomicAdd(H.popcnt, (unsigned long long)npop);   // (one per field and march: what bench.py prices the launch by)
  return overflow;
}

// Register budget: the wavefronts per SIMD that the kernel's LDS allows (one wavefront per workgroup, 160 KB per CU, four SIMDs) --
// the S-256 form (12 864 bytes) fits twelve workgroups per CU only while it stays within 168 registers, and the compiler does not
// know about that cliff unless it is told.
template <int CAP, class NT, int GPL, bool TAB>
constexpr int fmm_waves_per_simd() {
  const int lds = (64 / GPL) * CAP * (4 + (int)sizeof(NT)) + (TAB ? 288 : 4);
  const int w = (163840 / ((lds + 1279) / 1280 * 1280)) / 4;
  return w < 1 ? 1 : (w > 3 ? 3 : w);   // (never fewer than 168 registers: the loop needs ~150-170)
}
#ifdef DZ_FMM_WPE   // experiment: register budget for DZ_FMM_WPE wavefronts per SIMD
#define FMM_WPE_ATTR __attribute__((amdgpu_waves_per_eu(DZ_FMM_WPE, DZ_FMM_WPE)))
#else
#define FMM_WPE_ATTR __attribute__((amdgpu_waves_per_eu(fmm_waves_per_simd<CAP, NT, GPL, Heap<CAP, SPILL, NT, HYB, GPL>::TAB>())))
#endif
#ifdef DZ_TS_WAITSTAT   // experiment build: clocks the workgroups spend waiting for the previous stage of their task
__device__ unsigned long long g_ts_wait[2];
#endif
// The kernel reads its arguments through a pointer to the kernarg segment that the compiler cannot see through (round 5).  Taken
// as a by-value struct, the compiler loads all of FmmArgs into ~60 scalar registers at entry and keeps them alive across the
// marching loop; with the loop's own lane masks that is more than the 102 a wavefront has, and the spill code reloaded a whole
// 16-register tuple of arguments from VGPR lanes in EVERY pop (23 v_readlane_b32 of 380 VALU instructions per pop, each a
// four-cycle issue slot: profiles/r5_fmm_phase_split.md).  Behind the laundered pointer an argument is one s_load_dword where it is
// used (scalar memory, no VALU slot), and nothing but the pointer is alive across a march.
using FmmArgP = const __attribute__((address_space(4))) FmmArgs *;
__device__ __forceinline__ FmmArgP arg_launder(FmmArgP p) {
  int z;   // a zero the compiler cannot fold, made wavefront-uniform again (the result of an asm counts as divergent)
  asm volatile("v_mov_b32 %0, 0" : "=v"(z));
  return (FmmArgP)((const __attribute__((address_space(4))) char *)p + __builtin_amdgcn_readfirstlane(z));
}
#ifdef DZ_FMM_ARGS_BYVALUE   // experiment: the round-4 form
#define FMM_ARGS_DECL(A_)
#define FMM_ARGS_FRESH
#else
#define FMM_ARGS_DECL(A_) FmmArgP Ap = arg_launder((FmmArgP)__builtin_amdgcn_kernarg_segment_ptr())
#define FMM_ARGS_FRESH Ap = arg_launder(Ap)
#endif
template <int CAP, bool SPILL, class NT, bool HYB, int GPL = 16>
__global__ __launch_bounds__(64) FMM_WPE_ATTR void fmm_kernel(FmmArgs A_) {
#ifdef DZ_FMM_ARGS_BYVALUE
  const FmmArgs &A = A_;
#else
  FMM_ARGS_DECL(A_);
#define A (*Ap)
#endif
  constexpr int GP = GPL, FPW = 64 / GPL;   // lanes per field, fields per wavefront (these shadow the 16-lane constants above)
  // The four fields of a wavefront walk their heaps in step: the lanes of groups 0 and 1 (one 32-lane half of every LDS access:
  // MI355X_MICROARCH.md, LDS) read the SAME slot of two rows, and rows CAP entries apart start in the same bank -- a two-way
  // conflict on every access of the sift-down (SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS = 0.94).  -DDZ_FMM_SKEW lays the rows out
  // 0, 1, 3, 2 with a skew of half the banks in front of rows 1 and 2 (what is left of the 1 280-byte LDS granule: 12 768 of
  // 12 800 B at S-256, still twelve workgroups per CU): the ratio drops to 0.48 and NOTHING else moves -- SQ_WAIT_ANY 124.1 ->
  // 125.1 G of 315 G wave cycles, 80.9 k fields/s without against 80.4 k with in four same-box pairs (profiles/
  // r6_fmm_lds_conflicts.md).  The two extra LDS cycles of a conflicting access hide behind the ~100-cycle round trip they belong
  // to; the conflicts are not on the pop's critical chain.  Default: the plain layout.
#ifndef DZ_FMM_SKEW
  constexpr int KSKEW = 0, NSKEW = 0;
#else
  constexpr int KSKEW = FPW == 4 ? 16 : 0, NSKEW = FPW == 4 ? 16 : 0;   // entries (16-bit node ids: 8 banks)
#endif
  __shared__ __attribute__((aligned(16))) float s_keys_all[FPW * CAP + 2 * KSKEW];
  __shared__ __attribute__((aligned(16))) NT s_nodes_all[FPW * CAP + 2 * NSKEW];
  auto row_of = [](int g_) { return FPW == 4 ? (g_ == 0 ? 0 : (g_ == 1 ? 1 : (g_ == 3 ? 2 : 3))) : g_; };   // position of group g_'s row
  auto key_row = [&](int g_) { const int r = row_of(g_); return s_keys_all + r * CAP + (r >= 1 ? KSKEW : 0) + (r >= 3 ? KSKEW : 0); };
  auto node_row = [&](int g_) { const int r = row_of(g_); return s_nodes_all + r * CAP + (r >= 1 ? NSKEW : 0) + (r >= 3 ? NSKEW : 0); };
  // the queue position is handed to the wavefront through slot 0 of the first field's keys (the dummy slot of the marching
  // loop, idle between fields): the kernel's LDS is exactly the heaps, so five 32 KB workgroups of the hybrid heap fill 160 KB
  __shared__ short s_tab[Heap<CAP, SPILL, NT, HYB, GPL>::TAB ? 144 : 2];
  unsigned &s_base = *reinterpret_cast<unsigned *>(key_row(0));
  const int lane = threadIdx.x, grp = lane / GP, gl = lane & (GP - 1);
  dazim_geom g;   // (member by member: the arguments live in the constant address space)
  g.nvx = A.g.nvx; g.nvz = A.g.nvz; g.nnx = A.g.nnx; g.nnz = A.g.nnz;
  g.gox = A.g.gox; g.goz = A.g.goz; g.dnx = A.g.dnx; g.dnz = A.g.dnz; g.dvx = A.g.dvx; g.dvz = A.g.dvz;
  const int nnx = g.nnx, nnz = g.nnz, nn = nnx * nnz;
  const int tsh_c = tile_shift(nnz), nrec_c = tile_records(nnx, nnz);
  const size_t slot = (size_t)blockIdx.x * FPW + grp;
  unsigned *rec_r = A.rec_r + slot * NREC_R;
  float *velnr = A.velnr + slot * RM * RM;
  float2 *slownr = A.slownr + slot * NREC_R;
  Heap<CAP, SPILL, NT, HYB, GPL> H;
  H.keys = key_row(grp);
  H.nodes = node_row(grp);
  H.g0 = gl == 0;
  H.tab = s_tab;
  H.popcnt = reinterpret_cast<unsigned long long *>(A.counter + 16);
  // Time slicing (round 3, late).  A field is one serial chain of pops and all fields are equally long, so a launch lasts a whole
  // number of rounds of one field's latency at the occupancy of that round: S-256's 16 000 fields on 13 312 resident slots would
  // run one full round and a second one with a fifth of the chip busy.  With ts_nstage > 1 a field is marched in stages that
  // any workgroup may pick up -- stage 0 = refined march, injection and the coarse band; stages 1.. = ts_pops accepted nodes of
  // the coarse march each, the last one to the end.  Between two stages the LDS part of the heap and its size go to HBM, node
  // words and the heap's HBM level are per field anyway then.  Tasks are handed out stage-major inside each XCD range, so a
  // task's predecessor (same batch, previous stage) was handed out a whole generation earlier and a workgroup only ever
  // waits for a task that is running (flag per batch, release / acquire at agent scope: the two may run on different XCDs).
  // Which workgroup runs which stage has no influence on any result: the state handed over is exact.
  // (an integer in a scalar register: a wavefront-uniform bool is still a lane mask to the compiler, and a branch on it costs VALU work)
  // a finished field leaves the kernel: un-tiled into the reference's column-major ttn, or -- ttn == NULL, the fields stay inside the
  // library for the ray kernel (inv/CalSurfG.f90:909-912 never returns them) -- as it stands, in tiles (copied only if its node
  // words are not already at their final place); `zero`: a field whose source lies outside the grid
  auto store_field = [&](const unsigned *rec, int f, bool zero) {
    if (A.ttn) {
      float *ttn = A.ttn + (size_t)f * nn;
      if (zero) {
        for (int i = gl; i < nn; i += GP) ttn[i] = 0.0f;
        return;
      }
      for (int cx = 0; cx < nnx; cx++) {   // traveltime-grid write, back in the reference's column-major order
        const int tx = tile_x(cx, tsh_c);
        for (int cz = gl; cz < nnz; cz += GP) ttn[(size_t)cx * nnz + cz] = __int_as_float((int)rec[tx + tile_z(cz)]);   // all alive
      }
    } else {
      unsigned *dst = A.ttn_tiled + (size_t)(A.tslot ? A.tslot[f] : f) * nrec_c;
      if (zero) {
        for (int i = gl; i < nrec_c; i += GP) dst[i] = 0u;
      } else if (dst != rec) {
        for (int i = gl; i < nrec_c; i += GP) dst[i] = rec[i];
      }
    }
    if (A.fdone) {   // release: every store of this wavefront (the field's words; the refined outputs went out with an earlier fence), then the flag
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (gl == 0) __hip_atomic_store(A.fdone + f, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto flag_overflow = [&](int f) {
    if (A.fdone && gl == 0) __hip_atomic_store(A.fdone + f, 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  };
  int fastm = __builtin_amdgcn_readfirstlane(*A.vflag) == 0 ? A.fastm : 0;
  asm volatile("" : "+v"(fastm));   // (kept in a vector register: as a scalar it is spilled and comes back through v_readlane in every pop)
  if (A.prio) __builtin_amdgcn_s_setprio(3);   // issue priority over another kernel's wavefronts on the same SIMD (see run_fmm)
  const int nstage = SPILL ? 1 : A.ts_nstage;
  const bool ts = nstage > 1;
  unsigned &s_stage = *reinterpret_cast<unsigned *>(key_row(1));   // (the dummy slot of the second field, like s_base)

  // Work queue: the field list (sorted by period on the host) is cut into eight contiguous ranges, one per XCD (workgroup b
  // runs on XCD b % 8), so that the fields an XCD marches share one or two velocity grids and these stay in that XCD's L2;
  // a workgroup whose range is drained steals from the next ranges.
  const unsigned fpw = (unsigned)A.fpw;
  const unsigned nquad = ((unsigned)A.nfield + fpw - 1) / fpw;
  int chunk = (int)(blockIdx.x & 7);
  for (;;) {
    FMM_ARGS_FRESH;
    __syncthreads();
    if (lane == 0) {
      unsigned found = 0xffffffffu;
      for (int tried = 0; tried < 8; tried++) {
        const unsigned c0 = (nquad * (unsigned)chunk >> 3) * fpw, c1 = (nquad * (unsigned)(chunk + 1) >> 3) * fpw;
        const unsigned nbr = (c1 - c0) / fpw;                       // batches of this range; tasks: stage-major
        const unsigned b = c0 < c1 ? atomicAdd(&A.counter[chunk], 1u) : 0xffffffffu;
        if (b < nbr * (unsigned)nstage) {
          if (A.hprog && (b & 15u) == 15u) __hip_atomic_store(A.hprog + chunk, b + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          const unsigned stg = b / nbr;
          found = c0 + (b - stg * nbr) * fpw;
          s_stage = stg;
          if (stg > 0) {   // its predecessor (same batch, previous stage) must have handed its state over
            const unsigned *flag = A.ts_flag + found / fpw;
#ifdef DZ_TS_WAITSTAT
            const unsigned long long w0_ = __builtin_amdgcn_s_memtime();
#endif
            while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < stg) __builtin_amdgcn_s_sleep(20);
#ifdef DZ_TS_WAITSTAT
            atomicAdd(&g_ts_wait[0], __builtin_amdgcn_s_memtime() - w0_);
            atomicAdd(&g_ts_wait[1], 1ull);
#endif
          }
          break;
        }
        chunk = (chunk + 1) & 7;
      }
      s_base = found;
    }
    __syncthreads();
    const unsigned fbase = s_base;
    if (fbase == 0xffffffffu) break;
    const int stage = ts ? (int)s_stage : 0;
#ifdef DZ_TS_LIGHT_ACQ   // measurement only (not coherent across XCDs): what the L2 invalidate of the acquire costs
    if (stage > 0) asm volatile("buffer_inv sc0" ::: "memory");
#else
    if (stage > 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // acquire side for the lanes that did not spin
#endif
    const int q = (int)fbase + grp;
    // node words of the coarse grid and the heap's HBM level: per resident slot, or per field when stages change hands
    unsigned *rec_c = A.rec_c + (ts ? (size_t)q : slot) * nrec_c;
    H.ovf = A.ovf + (ts ? (size_t)q : slot) * A.ovfcap;
    if (grp < (int)fpw && q < A.nfield && stage > 0) {
      // ---- a later stage: take the heap over, go on marching, hand it on or finish ----
      const int f = A.flist ? A.flist[q] : q;
      const int n0 = A.ts_nodes[(size_t)q * CAP];
      if (n0 > 0) {
        const int per = A.period[f] - 1;
        const int nl = n0 < CAP ? n0 : CAP - 1;
        for (int i = 1 + gl; i <= nl; i += GP) {
          H.keys[i] = A.ts_keys[(size_t)q * CAP + i];
          H.nodes[i] = (NT)A.ts_nodes[(size_t)q * CAP + i];
        }
        H.ntr = n0;
        if (!SPILL) H.pad(gl, nl + 1);
        H.set_rec(rec_c);
        H.tsh = tsh_c;
        cbar();
        const bool ovf = march<CAP, SPILL, NT, HYB, false, GPL>(H, A.slown + (size_t)per * nrec_c, nnx, nnz, g.dnx, g.dnz, 0, lane, fastm,
                                                            stage == nstage - 1 ? 0x7fffffff : A.ts_pops);
        FMM_ARGS_FRESH;
        cbar();
        if (ovf) {
          flag_overflow(f);
          if (gl == 0) { A.status[f] = -2; A.ts_nodes[(size_t)q * CAP] = -1; }
        } else if (H.ntr == 0) {
          store_field(rec_c, f, false);
          if (gl == 0) A.ts_nodes[(size_t)q * CAP] = -1;
        } else {
          const int ns = H.ntr < CAP ? H.ntr : CAP - 1;
          for (int i = 1 + gl; i <= ns; i += GP) {
            A.ts_keys[(size_t)q * CAP + i] = H.keys[i];
            A.ts_nodes[(size_t)q * CAP + i] = (int)H.nodes[i];
          }
          if (gl == 0) A.ts_nodes[(size_t)q * CAP] = H.ntr;
        }
      }
    }
    if (grp < (int)fpw && q < A.nfield && stage == 0) {
      const int f = A.flist ? A.flist[q] : q;   // the four groups run the same phases on their own field (SIMT across groups)
      const float scx = A.scx[f], scz = A.scz[f];
      const int per = A.period[f] - 1;
      // ---- refined source box, inv/CalSurfG.f90:1169-1206 ----
      int isx = (int)((scx - g.gox) / g.dnx) + 1;
      int isz = (int)((scz - g.goz) / g.dnz) + 1;
      const bool outside = isx < 1 || isx > nnx || isz < 1 || isz > nnz || per < 0 || per >= A.kmax;
      if (gl == 0) A.status[f] = outside ? DAZIM_E_SOURCE_OUTSIDE : 0;
      if (outside) {
        store_field(rec_c, f, true);
        if (ts && gl == 0) A.ts_nodes[(size_t)q * CAP] = -1;
      } else {
        const double *pv = A.pv + (size_t)per * (g.nvz + 2) * (g.nvx + 2);
        const float *veln = A.veln + (size_t)per * nn;
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
        if (A.boxes && gl == 0) A.boxes[f] = bx;
        const int nnxr = bx.nnxr, nnzr = bx.nnzr;

        // ---- bsplrefine (inv/CalSurfG.f90:1525-1591) + status reset ----
        {
          const float *risti_rr = A.risti_r + (size_t)(bx.vnl - 1) * RM;   // EARTH*sin(colatitude) of the refined columns of this box
          const int nrxr = GDX * SGDL, nrzr = GDZ * SGDL;
          const int origx = (bx.vnl - 1) * SGDL + 1, origz = (bx.vnt - 1) * SGDL + 1;
          for (int idx = gl; idx < nnxr * RM; idx += GP) {
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
            const float vr = sum[0] + sum[1] + sum[2] + sum[3];
            velnr[idx] = vr;
            const int ti = tile_x(idm2 - 1, TSH_R) + tile_z(idm1 - 1);
            slownr[ti] = make_float2(1.0f / vr, risti_rr[idm2 - 1]);
            rec_r[ti] = W_FAR;
          }
        }
        cbar();
        // ---- travel(urg=1) source initialisation, inv/CalSurfG.f90:324-345 ----
        H.ntr = 0;
        if (!SPILL) H.pad(gl, 1);
        H.set_rec(rec_r);
        H.tsh = TSH_R;
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
            for (int jj = 1; jj <= 2; jj++) vss[i - 1][jj - 1] = velnr[(rsx - 2 + i) * RM + (rsz - 2 + jj)];
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
              const int urid = tile_x(ux - 1, TSH_R) + tile_z(uz - 1);
              H.add(t0, urid);
            }
        }
        // exit-rule quirk kept verbatim (inv/CalSurfG.f90:366-377): vnr/vnb (coarse indices) are
        // compared with the REFINED nnx/nnz, which is what the module variables hold at that point
        const int ex = (bx.vnl != 1 ? 1 : 0) | (bx.vnr != nnxr ? 2 : 0) | (bx.vnt != 1 ? 4 : 0) |
                       (bx.vnb != nnzr ? 8 : 0);
        bool ovf = march<CAP, SPILL, NT, HYB, true, GPL>(H, slownr, nnxr, nnzr, bx.dnxr, bx.dnzr, ex, lane, fastm);
        FMM_ARGS_FRESH;
        cbar();
        // ---- refined outputs (ttnr=ttn, nstsr=nsts, :1246-1247) + reset of the coarse records ----
        {
          float *ttnr = A.ttnr ? A.ttnr + (size_t)f * RM * RM : nullptr;
          int *nstsr = A.nstsr ? A.nstsr + (size_t)f * RM * RM : nullptr;
          if (ttnr || nstsr)
            for (int idx = gl; idx < RM * RM; idx += GP) {
              const int c = idx / RM, r = idx - c * RM;
              // (a band node's word is its slot; its trial time is the key at that slot of the heap, which is still in place)
              int st = -9;
              float tt = 0.0f;
              if (c < nnxr && r < nnzr) {
                const unsigned w = rec_r[tile_x(c, TSH_R) + tile_z(r)];
                if (w_is_alive(w)) { st = 0; tt = __int_as_float((int)w); }
                else if (w == W_FAR) st = -1;
                else { st = (int)(w & 0x7fffffffu); tt = H.get(st).key; }
              }
              if (nstsr) nstsr[idx] = st;
              if (ttnr) ttnr[idx] = tt;
            }
          for (int i = gl; i < nrec_c; i += GP) rec_c[i] = W_FAR;
        }
        cbar();
        // ---- inject every sgdl-th refined node (inv/CalSurfG.f90:1252-1262) ----
        const int bw = bx.vnr - bx.vnl + 1, bh = bx.vnb - bx.vnt + 1, nbox = bw * bh;
        for (int i = gl; i < nbox; i += GP) {
          const int bxi = i / bh, bzi = i - bxi * bh;  // column-major inside the box
          const unsigned w = rec_r[tile_x(bxi * SGDL, TSH_R) + tile_z(bzi * SGDL)];
          // alive: its time; close (in the refined band when the march stopped): its trial time = the key at its slot, to be put
          // into the coarse heap; far
          rec_c[tile_x(bx.vnl - 1 + bxi, tsh_c) + tile_z(bx.vnt - 1 + bzi)] =
              (w_is_alive(w) || w == W_FAR) ? w : w_pending(H.get((int)(w & 0x7fffffffu)).key);
        }
        cbar();
        // ---- alive nodes touching a far node rejoin the band (inv/CalSurfG.f90:1291-1308).  Only
        // the tests against -1 matter and promotions never create a -1, so the sweep is order-free.
        // Two passes (decide, then write) keep the 16 lanes from racing on each other's nodes. ----
        for (int base = 0; base < nbox; base += GP) {
          const int i = base + gl;
          bool promote = false;
          unsigned *p = nullptr;
          unsigned w = W_FAR;
          if (i < nbox) {
            const int bxi = i / bh, bzi = i - bxi * bh;
            const int cx = bx.vnl + bxi, cz = bx.vnt + bzi;
            const int tx = tile_x(cx - 1, tsh_c), tz = tile_z(cz - 1);
            p = &rec_c[tx + tz];
            w = *p;
            if (w_is_alive(w)) {
              if (cz - 1 >= 1 && rec_c[tx + tile_z(cz - 2)] == W_FAR) promote = true;
              if (cz + 1 <= nnz && rec_c[tx + tile_z(cz)] == W_FAR) promote = true;
              if (cx - 1 >= 1 && rec_c[tile_x(cx - 2, tsh_c) + tz] == W_FAR) promote = true;
              if (cx + 1 <= nnx && rec_c[tile_x(cx, tsh_c) + tz] == W_FAR) promote = true;
            }
          }
          cbar();
          if (promote) *p = w_pending(w_time(w));               // alive -> close, same time
        }
        cbar();
        // ---- travel(urg=2): rebuild the band in column-major node order (inv/CalSurfG.f90:311-317) ----
        H.ntr = 0;
        if (!SPILL) H.pad(gl, 1);
        H.set_rec(rec_c);
        H.tsh = tsh_c;
        for (int base = 0; base < nbox; base += GP) {
          const int i = base + gl;
          unsigned w = W_FAR;
          int node = 0;
          if (i < nbox) {
            const int bxi = i / bh, bzi = i - bxi * bh;
            const int cx = bx.vnl + bxi, cz = bx.vnt + bzi;
            node = tile_x(cx - 1, tsh_c) + tile_z(cz - 1);
            w = rec_c[node];
          }
          unsigned m = (unsigned)((wballot(w_is_pending(w)) >> (grp * GP)) & ((1ull << GP) - 1ull));
          while (m) {
            const int b = __builtin_ctz(m);
            m &= m - 1;
            const float t0 = __shfl(w_time(w), grp * GP + b);
            const int n0 = __shfl(node, grp * GP + b);
            if (!H.full()) H.add(t0, n0); else ovf = true;
          }
        }
        if (ts && !ovf) {   // the coarse band is built: hand the heap to stage 1
          const int ns = H.ntr < CAP ? H.ntr : CAP - 1;
          for (int i = 1 + gl; i <= ns; i += GP) {
            A.ts_keys[(size_t)q * CAP + i] = H.keys[i];
            A.ts_nodes[(size_t)q * CAP + i] = (int)H.nodes[i];
          }
          if (gl == 0) A.ts_nodes[(size_t)q * CAP] = H.ntr;
        }
        if (!ts && !ovf) ovf = march<CAP, SPILL, NT, HYB, false, GPL>(H, A.slown + (size_t)per * nrec_c, nnx, nnz, g.dnx, g.dnz, 0, lane, fastm);
        FMM_ARGS_FRESH;
        cbar();
        if (ovf) {
          flag_overflow(f);
          if (gl == 0) A.status[f] = -2;  // band outgrew the LDS heap: host reruns this field with SPILL
          if (ts && gl == 0) A.ts_nodes[(size_t)q * CAP] = -1;
        } else if (!ts || H.ntr == 0) {   // (time-sliced: only if the box left no band at all -- the later stages find nothing to do)
          store_field(rec_c, f, false);
        }
      }
    }
    if (ts) {   // hand the batch to its next stage: everything this task stored, then the flag
#ifdef DZ_TS_LIGHT_REL   // measurement only (not coherent across XCDs): what the L2 write-back of the release costs
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
      __syncthreads();
      if (lane == 0) __hip_atomic_store(A.ts_flag + fbase / fpw, (unsigned)stage + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#undef A
}


template <int CAP, class NT, bool HYB = false, int GPL = 16>
int run_fmm(dazim_ctx *ctx, FmmArgs A, int nfield, size_t nn, size_t nr, int *d_status, std::shared_ptr<std::vector<int>> hsp, bool async,
            std::function<int()> *finish_out) {
  constexpr int FPW = 64 / GPL;   // fields per wavefront of the fast kernel (the spill rerun keeps 16 lanes per field)
  int rc;
  void *p;
  // workgroups (one wavefront, FPW fields each): as many as the LDS heaps allow per CU
  int per_cu = 0;
  DZ_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fmm_kernel<CAP, false, NT, HYB, GPL>, 64, 0));
  {   // the API rounds LDS differently from the allocator, which hands out 1 280-byte granules of the CU's 160 KB: at 12 288, 16 384
      // and 32 768 bytes it answers one workgroup too many (measured: tools/lds_granule_probe.hip, profiles/r5_lds_granule.md)
    hipFuncAttributes fa;
    DZ_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&fmm_kernel<CAP, false, NT, HYB, GPL>)));
    const int by_lds = (int)(163840 / ((fa.sharedSizeBytes + 1279) / 1280 * 1280));
    if (per_cu > by_lds) per_cu = by_lds;
  }
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 16) per_cu = 16;
  ctx->ksec["fmm.wg_per_cu"] = (double)per_cu;
  if (ctx->opts.count("fmm.wg_per_cu") && ctx->opts["fmm.wg_per_cu"] > 0 && ctx->opts["fmm.wg_per_cu"] < per_cu) per_cu = ctx->opts["fmm.wg_per_cu"];
  int nwg = ctx->num_cu * per_cu;
  // Small batches: a launch lasts at least as long as ONE field takes alone, and with fewer wavefronts than SIMDs most of the
  // chip idles; fewer fields per wavefront then put every field on a SIMD of its own sooner.  Option fmm.fpw forces 1, 2 or 4.
  A.fpw = FPW;
  if (ctx->opts.count("fmm.fpw") && (ctx->opts["fmm.fpw"] == 1 || ctx->opts["fmm.fpw"] == 2 || ctx->opts["fmm.fpw"] == 4) && ctx->opts["fmm.fpw"] < FPW) A.fpw = ctx->opts["fmm.fpw"];
  ctx->ksec["fmm.lanes_per_field"] = GPL;
  ctx->ksec["fmm.fpw"] = A.fpw;
  // Issue priority (round 4).  A batch that leaves most of the chip empty lasts as long as one field's serial chain, and every cycle
  // one of its wavefronts waits for an issue slot behind another kernel's wavefronts -- the dispersion kernel's perturbed copies on
  // the auxiliary stream (disp.async) -- lengthens that chain: s_setprio 3 gives them the slot first.  S-128: eikonal launch 52.0 ->
  // 45.0 ms (= alone on the chip), step 64.0 -> 56.9 ms; test4_Yunnan program: assembly 1.015 -> 0.875 s.  A batch that fills the chip
  // gains nothing among its own wavefronts and only starves the copies the ray kernel then waits for (S-256: step 368 -> 380 ms), so
  // the rule is: at most half of the resident workgroups.  Option fmm.prio = 1 / 2 forces it on / off.
  A.prio = (nfield + A.fpw - 1) / A.fpw <= nwg / 2 ? 1 : 0;
  if (ctx->opts.count("fmm.prio") && ctx->opts["fmm.prio"] == 1) A.prio = 1;
  if (ctx->opts.count("fmm.prio") && ctx->opts["fmm.prio"] == 2) A.prio = 0;
  ctx->ksec["fmm.prio"] = A.prio;
  if (nwg > (nfield + A.fpw - 1) / A.fpw) nwg = (nfield + A.fpw - 1) / A.fpw;
  const int nslot = nwg * FPW;
  const int ovfcap = (int)((nn > nr ? nn : nr) / 2 + 64);  // maxbt = nint(snb*nnx*nnz), inv/CalSurfG.f90:1068
  A.ovfcap = HYB ? Heap<CAP, false, NT, HYB, GPL>::TOT - CAP : 0;   // the fast kernel: only the HYB heap has HBM levels
  // Time slicing (see fmm_kernel): on when the batch does not fit the resident slots (more than one round) and the per-field node
  // words fit comfortably; option fmm.ts = 1 / 2 forces it on / off (0: this rule), fmm.ts_stages sets the number of coarse stages.
  // Few stages are best (not because of the hand-over fences: tools/exp_ts_fences.sh; with many short stages handed out
  // stage-major the fields march in step again and the mixture of phases on a CU is lost): S-256's 16 000
  // fields on the 512-slot hybrid heap take 0.240 / 0.253 / 0.248 / 0.249 / 0.252 s with 2 / 3 / 4 / 8 / 12 coarse stages
  // (0.288 s unsliced on the 768-slot heap, same box), the 768-slot heap 0.292 / 0.264 / 0.262 s with 2 / 4 / 8-12; S-512's
  // 32 000 fields (1024-slot hybrid heap) 4.29 s unsliced, 4.03 / 3.96 / 3.97 s with 2 / 4 / 8.  Defaults: 2 on the 512-slot heaps
  // with 16-bit ids (round 5, the all-LDS one on a 166 x 166 grid, 16 000 fields: 156 k fields/s with 2 stages, 122 k with 4), 8 on the one with two HBM levels (S-512: 2.68 / 2.63 / 2.61 s with 2 / 4 / 8), 4 elsewhere.
  bool ts = nfield > nslot;
  if (ctx->opts.count("fmm.ts") && ctx->opts["fmm.ts"] == 1) ts = true;
  if (ctx->opts.count("fmm.ts") && ctx->opts["fmm.ts"] == 2) ts = false;
  const size_t rec_field_bytes = (size_t)tile_records(A.g.nnx, A.g.nnz) * sizeof(unsigned);
  {
    size_t mfree = 0, mtot = 0;
    if (hipMemGetInfo(&mfree, &mtot) != hipSuccess || (size_t)nfield * rec_field_bytes > mfree / 4) ts = false;
  }
  int nseg = ctx->opts.count("fmm.ts_stages") && ctx->opts["fmm.ts_stages"] > 0 ? ctx->opts["fmm.ts_stages"] : (CAP <= 512 ? (sizeof(NT) == 2 ? 2 : (HYB ? 8 : 4)) : 4);
  A.ts_nstage = ts ? 1 + nseg : 1;
  // (stage lengths that shrink towards the end -- a shorter tail -- were measured and lose: 0.251-0.265 s against 0.247 s)
  A.ts_pops = (int)((nn + nseg - 1) / nseg);
  // (stages of unequal length -- odd batches longer, even ones shorter, to put the batches out of step -- lose badly: +19 % of
  // the time at +-20 %, +44 % at +-40 %: the stage-major hand-out relies on equal tasks, a workgroup that takes a task whose
  // predecessor is still running waits; with equal stages that wait is 0.15 % of the workgroups' time at 2 stages, 2 % at 8:
  // tools/exp_ts_wait.sh)
  ctx->ksec["fmm.ts_stages"] = ts ? (double)nseg : 0.0;
  // owners of node words / HBM heap levels: fields or resident slots.  + 8: the idle lane groups of the last wavefront of a batch
  // (owner index up to nfield + fields per wavefront - 1, at most 8 fields per wavefront) address their own, unused, state
  const size_t nown = (ts ? (size_t)nfield : (size_t)nslot) + 8;
  if ((rc = dz_scratch(ctx, "fmm.rec_c", nown * rec_field_bytes, &p))) return rc;
  A.rec_c = (unsigned *)p;
  const bool keep_tiled = A.ttn == nullptr;   // the fields stay inside the library, in tiles (dazim_fmm_batch with ttn == NULL)
  A.ttn_tiled = nullptr;
  A.tslot = nullptr;
  if (keep_tiled && ts) {
    A.ttn_tiled = A.rec_c;   // every field's node words end where they were marched: nothing is copied (tslot below)
  } else if (keep_tiled) {
    if ((rc = dz_scratch(ctx, "fmm.ttn_tiled", (size_t)nfield * rec_field_bytes, &p))) return rc;
    A.ttn_tiled = (unsigned *)p;
  }
  A.ts_flag = nullptr; A.ts_keys = nullptr; A.ts_nodes = nullptr;
  if (ts) {
    if ((rc = dz_scratch(ctx, "fmm.ts_flag", ((size_t)nfield / A.fpw + 2) * 4, &p))) return rc;
    A.ts_flag = (unsigned *)p;
    DZ_HIP(hipMemsetAsync(A.ts_flag, 0, ((size_t)nfield / A.fpw + 2) * 4, ctx->stream));
    if ((rc = dz_scratch(ctx, "fmm.ts_keys", (size_t)nfield * CAP * 4, &p))) return rc;
    A.ts_keys = (float *)p;
    if ((rc = dz_scratch(ctx, "fmm.ts_nodes", (size_t)nfield * CAP * 4, &p))) return rc;
    A.ts_nodes = (int *)p;
  }
  if ((rc = dz_scratch(ctx, "fmm.rec_r", (size_t)nslot * NREC_R * sizeof(unsigned), &p))) return rc;
  A.rec_r = (unsigned *)p;
  if ((rc = dz_scratch(ctx, "fmm.velnr", (size_t)nslot * nr * 4, &p))) return rc;
  A.velnr = (float *)p;
  if ((rc = dz_scratch(ctx, "fmm.slownr", (size_t)nslot * NREC_R * sizeof(float2), &p))) return rc;
  A.slownr = (float2 *)p;
  if ((rc = dz_scratch(ctx, "fmm.ovf", nown * A.ovfcap * sizeof(HEnt) + 64, &p))) return rc;
  A.ovf = (HEnt *)p;
  if ((rc = dz_scratch(ctx, "fmm.counter", 256, &p))) return rc;
  A.counter = (unsigned *)p;
  A.status = d_status;
  A.flist = nullptr;
  std::vector<int> order(nfield);
  {   // stable counting sort of the fields by period (the list the XCD ranges are cut from)
    // (the host's reads of the step land in pinned memory -- dz_pinned: DMA transfers instead of blit kernels that queue behind
    // whatever holds the chip)
    if ((rc = dz_pinned(ctx, "fmm.host", (size_t)nfield * 12 + 64, &p))) return rc;
    int *hper = (int *)p;
    float *hx = (float *)p + nfield, *hz = (float *)p + 2 * (size_t)nfield;
    const bool sorted = !(ctx->opts.count("fmm.sort") && !ctx->opts["fmm.sort"]);
    DZ_HIP(hipMemcpyAsync(hper, A.period, (size_t)nfield * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (sorted) {
      DZ_HIP(hipMemcpyAsync(hx, A.scx, (size_t)nfield * 4, hipMemcpyDeviceToHost, ctx->stream));
      DZ_HIP(hipMemcpyAsync(hz, A.scz, (size_t)nfield * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    DZ_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < nfield; i++)
      if (hper[i] < 1 || hper[i] > A.kmax)
        return dz_fail(ctx, DAZIM_E_BAD_ARG, "period_idx[%d] = %d outside 1..%d (1-based, like periods(srcnum,knumi))", i, hper[i], A.kmax);
    std::vector<int> cnt(A.kmax + 2, 0);
    auto key = [&](int i) { const int k = hper[i]; return k < 1 || k > A.kmax ? A.kmax + 1 : k; };
    for (int i = 0; i < nfield; i++) cnt[key(i)]++;
    int run = 0;
    for (size_t k = 0; k < cnt.size(); k++) { const int c = cnt[k]; cnt[k] = run; run += c; }
    for (int i = 0; i < nfield; i++) order[cnt[key(i)]++] = i;
    // Within a period: by the distance of the source from the middle of the grid.  The four fields of a wavefront march in
    // lockstep and the sift-down runs as many 4-level steps as the largest of their bands needs (three from 512 entries on,
    // else two); the band of a field grows with the room the front has, i.e. with how central the source is, so fields with
    // similar bands are put together.  (Speed only: which fields share a wavefront has no influence on any field.)
    if (sorted) {
      const float cx = A.g.gox + 0.5f * (float)(A.g.nnx - 1) * A.g.dnx, cz = A.g.goz + 0.5f * (float)(A.g.nnz - 1) * A.g.dnz;
      const float wx = 1.0f / ((float)A.g.nnx * A.g.dnx), wz = 1.0f / ((float)A.g.nnz * A.g.dnz);
      auto dist = [&](int i) {   // Chebyshev distance from the centre in units of the grid size: the nearest edge decides the band
        const float ax = fabsf(hx[i] - cx) * wx, az = fabsf(hz[i] - cz) * wz;
        return ax > az ? ax : az;
      };
      int b0 = 0;
      for (int i = 1; i <= nfield; i++)
        if (i == nfield || hper[order[i]] != hper[order[b0]]) {
          std::stable_sort(order.begin() + b0, order.begin() + i, [&](int a, int b) { return dist(a) < dist(b); });
          b0 = i;
        }
    }
    void *po;
    if ((rc = dz_pinned(ctx, "fmm.host_order", (size_t)nfield * 8 + 16, &po))) return rc;
    memcpy(po, order.data(), (size_t)nfield * 4);
    if (keep_tiled && ts) {   // field -> its place in the queue = its block of node words
      int *inv = (int *)po + nfield;
      for (int i = 0; i < nfield; i++) inv[order[i]] = i;
      if ((rc = dz_scratch(ctx, "fmm.tslot", (size_t)nfield * 4 + 16, &p))) return rc;
      DZ_HIP(hipMemcpyAsync(p, inv, (size_t)nfield * 4, hipMemcpyHostToDevice, ctx->stream));
      A.tslot = (const int *)p;
    }
    if ((rc = dz_scratch(ctx, "fmm.order", (size_t)nfield * 4 + 16, &p))) return rc;
    DZ_HIP(hipMemcpyAsync(p, po, (size_t)nfield * 4, hipMemcpyHostToDevice, ctx->stream));
    A.flist = (const int *)p;
  }
  const bool force_spill = ctx->opts.count("fmm.force_spill") && ctx->opts["fmm.force_spill"];
  if ((rc = dz_pinned(ctx, "fmm.host_status", (size_t)nfield * 4 + 64, &p))) return rc;
  int *hs_pin = (int *)p;
  if ((rc = dz_async_init(ctx))) return rc;
  DZ_HIP(hipMemsetAsync(A.counter, 0, 256, ctx->stream));
  if (A.fdone) DZ_HIP(hipMemsetAsync(A.fdone, 0, (size_t)nfield * 4, ctx->stream));
  A.hprog = nullptr;
  unsigned total_tasks = 0;
  if (A.fdone && ctx->hprog) {
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, ctx->hprog, 0) == hipSuccess && dp) {
      for (int c = 0; c < 8; c++) ctx->hprog[c] = 0;
      A.hprog = (unsigned *)dp;
      const unsigned nq = ((unsigned)nfield + (unsigned)A.fpw - 1) / (unsigned)A.fpw;
      for (unsigned c = 0; c < 8; c++) total_tasks += ((nq * (c + 1) >> 3) - (nq * c >> 3)) * (unsigned)A.ts_nstage;
    } else {
      (void)hipGetLastError();
    }
  }
  // (what a ray kernel on another stream must see complete before it starts: the gridder's velocity grids, the cleared flags)
  DZ_HIP(hipEventRecord(ctx->ev_pre, ctx->stream));
  DZ_HIP(hipEventRecord(ctx->ev_f0, ctx->stream));
  if (!force_spill) {
    hipLaunchKernelGGL((fmm_kernel<CAP, false, NT, HYB, GPL>), dim3(nwg), dim3(64), 0, ctx->stream, A);
    DZ_HIP(hipGetLastError());
  }
  DZ_HIP(hipEventRecord(ctx->ev_f1, ctx->stream));
  // ---- everything after the launch: statuses, spill reruns, timers.  At once, or -- option fmm.async -- when the ray call (or
  // dazim_sync ...) asks for it, so that the ray kernel's count pass can be enqueued beside the launch ----
  auto fin = [=]() mutable -> int {
  int rc;
  void *p;
  std::vector<int> &hs = *hsp;
  std::vector<int> redo;
  if (!force_spill) {
    DZ_HIP(hipMemcpyAsync(hs_pin, d_status, (size_t)nfield * 4, hipMemcpyDeviceToHost, ctx->stream));
    DZ_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(hs.data(), hs_pin, (size_t)nfield * 4);
    for (int i = 0; i < nfield; i++)
      if (hs[i] == -2) redo.push_back(i);
  } else {
    for (int i = 0; i < nfield; i++) redo.push_back(i);
  }
  ctx->ksec["fmm.spilled_fields"] = (double)redo.size();
  if (!redo.empty()) {  // fields whose narrow band outgrew the LDS heap: same kernel, HBM spill enabled
    if ((rc = dz_scratch(ctx, "fmm.flist", redo.size() * 4, &p))) return rc;
    DZ_HIP(hipMemcpyAsync(p, redo.data(), redo.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    A.flist = (const int *)p;
    A.nfield = (int)redo.size();
    DZ_HIP(hipMemsetAsync(A.counter, 0, 64, ctx->stream));   // (the queue positions; the count of accepted nodes goes on)
    A.fpw = 4;                                        // (16 lanes per field: the spill kernel's sequential sift-down is written for them)
    A.ts_nstage = 1;
    int nwg2 = ((int)redo.size() + 3) / 4;
    if (nwg2 > nwg) nwg2 = nwg;
    // the spill kernel keeps every slot >= CAP in HBM: maxbt entries per resident field, allocated only when a field needs it
    A.ovfcap = ovfcap;
    if ((rc = dz_scratch(ctx, "fmm.ovf_spill", (size_t)nwg2 * 4 * ovfcap * sizeof(HEnt), &p))) return rc;
    A.ovf = (HEnt *)p;
    if (keep_tiled && ts) {   // (the first launch's node words ARE the results: the rerun marches in blocks of its own and copies)
      if ((rc = dz_scratch(ctx, "fmm.rec_spill", (size_t)(nwg2 * 4 + 8) * rec_field_bytes, &p))) return rc;
      A.rec_c = (unsigned *)p;
    }
    hipLaunchKernelGGL((fmm_kernel<CAP, true, NT, false>), dim3(nwg2), dim3(64), 0, ctx->stream, A);
    DZ_HIP(hipGetLastError());
    DZ_HIP(hipEventRecord(ctx->ev_f1, ctx->stream));
    DZ_HIP(hipMemcpyAsync(hs_pin, d_status, (size_t)nfield * 4, hipMemcpyDeviceToHost, ctx->stream));
    DZ_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(hs.data(), hs_pin, (size_t)nfield * 4);
  }
  {
    float ms = 0;
    DZ_HIP(hipEventSynchronize(ctx->ev_f1));
    DZ_HIP(hipEventElapsedTime(&ms, ctx->ev_f0, ctx->ev_f1));
    ctx->ksec["fmm"] = ms * 1e-3;
  }
  {
    unsigned long long &hp = *(unsigned long long *)(hs_pin + nfield + 2 - (nfield & 1));
    hp = 0;
    DZ_HIP(hipMemcpyAsync(&hp, A.counter + 16, 8, hipMemcpyDeviceToHost, ctx->stream));
    DZ_HIP(hipStreamSynchronize(ctx->stream));
    ctx->ksec["fmm.field_pops"] = (double)hp;   // nodes accepted by this call (all fields, refined + coarse marches, incl. spill reruns)
  }
#ifdef DZ_TS_WAITSTAT
  {
    unsigned long long h[2];
    DZ_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ts_wait), sizeof h));
    fprintf(stderr, "ts wait: %llu clocks (100 MHz) in %llu waits of later-stage tasks; %d workgroups\n", h[0], h[1], nwg);
    unsigned long long z[2] = {0, 0};
    DZ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_ts_wait), z, sizeof z));
  }
#endif
#ifdef DZ_FMM_LAZYSTAT
  {
    unsigned long long h[4];
    DZ_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_lazy_stat), sizeof h));
    fprintf(stderr, "lazy look-ups: %llu within three levels, %llu higher (second round), %llu the dropped entry\n", h[0], h[1], h[2]);
    unsigned long long z[4] = {0};
    DZ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_lazy_stat), z, sizeof z));
  }
#endif
#ifdef DZ_FMM_PROF
  {
    unsigned long long h[8];
    DZ_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fmm_prof), sizeof h));
    fprintf(stderr, "fmm prof (memtime ticks): setup+loads %llu popdown %llu loadwait %llu fix %llu quadrant %llu siftups-fast %llu siftups-slow %llu - looptop %llu\n",
            h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    unsigned long long z[8] = {0};
    DZ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_fmm_prof), z, sizeof z));
  }
#endif
  return 0;
  };
  ctx->fields.tiled = keep_tiled ? A.ttn_tiled : nullptr;
  ctx->fields.tslot = keep_tiled ? A.tslot : nullptr;
  ctx->fields.nfield = nfield;
  ctx->fields.nnx = A.g.nnx;
  ctx->fields.nnz = A.g.nnz;
  ctx->fields.stride = tile_records(A.g.nnx, A.g.nnz);
  ctx->fields.tsh = tile_shift(A.g.nnz);
  // (a batch that fits the resident slots has no tail worth filling -- its workgroups all end together -- and a ray pass beside it
  // would only wait: such a call completes at once)
  // ... and with eight coarse stages (the two-level hybrid heaps of the 257..682-node grids: S-512) the tail is a ninth of a field's
  // march: the ray passes beside it only stretch the launch by what they save afterwards (3.00 s either way) -- not there
  async = async && ts && A.hprog != nullptr && nseg <= 4;
  ctx->fields.fdone = async ? A.fdone : nullptr;
  ctx->fields.nwg = (unsigned)nwg;
  ctx->fields.hprog = A.hprog ? ctx->hprog : nullptr;
  ctx->fields.total_tasks = total_tasks;
  ctx->ksec["fmm.async"] = async ? 1.0 : 0.0;
  if (!async) return fin();
  *finish_out = fin;
  return 0;
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
  if (kmax < 1 || nfield < 0 || !pv_u || (nfield > 0 && (!scx_u || !scz_u || !period_u)) || g.nnx > 32767 || g.nnz > 32767)
    return dz_fail(ctx, DAZIM_E_BAD_ARG, "bad arguments to dazim_fmm_batch");
  DZ_HIP(hipSetDevice(ctx->device));
  { const int rcf = dz_fmm_finish(ctx); if (rcf) return rcf; }   // (an earlier asynchronous call nobody has collected yet)
  struct Busy {   // (the multi-GB scratch of a time-sliced batch may be freed by dz_trim_caches when another call runs out of memory)
    dazim_ctx *c;
    bool keep = false;
    explicit Busy(dazim_ctx *c_) : c(c_) { c->fmm_busy = true; }
    ~Busy() { if (!keep) c->fmm_busy = false; }
  } busy(ctx);
  ctx->fields = dazim_ctx::TiledFields();   // (whatever an earlier call left for the ray kernel is gone: the scratch is reused)
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
  if ((rc = dz_scratch(ctx, "fmm.slown", (size_t)tile_records(g.nnx, g.nnz) * kmax * sizeof(float2), &p))) return rc;
  float2 *d_slown = (float2 *)p;
  if ((rc = dz_scratch(ctx, "fmm.vflag", 64, &p))) return rc;
  int *d_vflag = (int *)p;
  // the short exact division / square root of the quadrant solve (div_exact): node spacings of 2 .. 4096 km on the coarse grid
  // (0.25 km on the refined one) and, checked by gridder_kernel, velocities of 0.125 .. 16 km/s.  Option fmm.ieee = 1: never.
  bool fastm = !(ctx->opts.count("fmm.ieee") && ctx->opts["fmm.ieee"]);
  {
    const float u1 = fabsf(EARTH * g.dnx);
    if (!(u1 >= FAST_STEP_MIN && u1 <= FAST_STEP_MAX)) fastm = false;
    for (float r : rc_tab) {
      const float v1 = fabsf(r * g.dnz);
      if (!(v1 >= FAST_STEP_MIN && v1 <= FAST_STEP_MAX)) fastm = false;
    }
    for (float r : rr_tab) {   // (the refined lattice: its steps are an eighth of these)
      const float v1 = fabsf(r * g.dnz);
      if (!(v1 >= FAST_STEP_MIN && v1 <= FAST_STEP_MAX)) fastm = false;
    }
  }
  ctx->ksec["fmm.fast_math"] = fastm ? 1.0 : 0.0;
  {
    DzTimer t(ctx, "gridder");
    const int total = (int)(nn * kmax);
    DZ_HIP(hipMemsetAsync(d_vflag, 0, 4, ctx->stream));
    hipLaunchKernelGGL(gridder_kernel, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, g, kmax, pv.dev, d_veln, d_slown, d_rc, d_vflag);
    DZ_HIP(hipGetLastError());
    t.stop();
  }
  if (nfield > 0) {
    FmmArgs A0;
    A0.g = g;
    A0.nfield = nfield;
    A0.kmax = kmax;
    A0.pv = pv.dev;
    A0.veln = d_veln;
    A0.slown = d_slown;
    A0.scx = scx.dev;
    A0.scz = scz.dev;
    A0.period = period.dev;
    A0.risti_c = d_rc;
    A0.risti_r = d_rr;
    A0.fastm = fastm ? 1 : 0;
    A0.vflag = d_vflag;
    A0.ttn = ttn.dev;
    A0.ttnr = ttnr.dev;
    A0.nstsr = nstsr.dev;
    A0.boxes = boxes.dev;
    int *d_status = status.dev;
    if (!d_status) {
      if ((rc = dz_scratch(ctx, "fmm.status", (size_t)nfield * 4, &p))) return rc;
      d_status = (int *)p;
    }
    // Option fmm.async: return once the launch is enqueued (see dazim_ctx::fmm_finish).  Only when nothing of this call waits for
    // the launch on the host: the coarse fields stay inside the library (ttn == NULL) and every array is device-resident.
    const bool async = ctx->opts.count("fmm.async") && ctx->opts["fmm.async"] && !ttn_u && !pv.staged && !scx.staged && !scz.staged &&
                       !period.staged && !veln.staged && !ttnr.staged && !nstsr.staged && !boxes.staged && !status.staged &&
                       ttnr.dev && nstsr.dev && boxes.dev;
    A0.fdone = nullptr;
    if (async) {
      if ((rc = dz_scratch(ctx, "fmm.fdone", (size_t)nfield * 4 + 16, &p))) return rc;
      A0.fdone = (int *)p;
    }
    // LDS heap slots per field: the narrow band of an N x M grid peaks near 3*max(N,M) entries (and the
    // 129 x 129 refined grid near 400); the smallest instantiation above that maximises the number of
    // fields in flight per CU.  A field whose band still outgrows it is redone by the spill kernel.
    int cap = 3 * (g.nnx > g.nnz ? g.nnx : g.nnz);
    if (cap < 3 * RM) cap = 3 * RM;
    if (ctx->opts.count("fmm.cap") && ctx->opts["fmm.cap"] > 0) cap = ctx->opts["fmm.cap"];
    auto hsp = std::make_shared<std::vector<int>>(nfield);
    std::function<int()> fin;
    const bool small = g.nnx <= 256 && g.nnz <= 256;   // node id fits 16 bits
    bool use_hyb512 = cap > 512 && nfield > ctx->num_cu * 8 * FPW;
    if (ctx->opts.count("fmm.hyb512") && ctx->opts["fmm.hyb512"] == 1) use_hyb512 = true;
    if (ctx->opts.count("fmm.hyb512") && ctx->opts["fmm.hyb512"] == 2) use_hyb512 = false;
    // (fmm.no_hybrid = 1 or an explicit fmm.cap: the one-level hybrid / all-LDS heaps of the branches below)
    // The small-LDS forms trade latency for wavefronts: a batch that leaves the chip half empty anyway (fewer than 2.5 workgroups
    // per CU) marches faster on the heaps with more levels in LDS -- 1 600 fields: 511 x 511 nodes 0.60 against 0.71 s, 341 x 341
    // 0.21 against 0.29 s, 701 x 701 0.86 against 1.33 s; 4 800 fields: 1.08 / 0.82 s and 0.40 / 0.33 s the other way round --
    // unless the bands would outgrow those (grids above 768 nodes a side: the all-LDS 2048-slot heap hands them to the spill kernel).
    // fmm.hyb2 = 1 / 2 forces the small-LDS forms on / off.
    bool use_hyb2 = cap > 768 && ((long)nfield > (long)ctx->num_cu * 10 || cap > 2304);
    if (ctx->opts.count("fmm.hyb2") && ctx->opts["fmm.hyb2"] == 1) use_hyb2 = cap > 768;
    if (ctx->opts.count("fmm.hyb2") && ctx->opts["fmm.hyb2"] == 2) use_hyb2 = false;
    if ((ctx->opts.count("fmm.no_hybrid") && ctx->opts["fmm.no_hybrid"]) || (ctx->opts.count("fmm.cap") && ctx->opts["fmm.cap"] > 0)) use_hyb2 = false;
    // Eight fields per wavefront (8 lanes per field, two quadrants per lane; round 4, option fmm.gp8): 1 = on the heaps the batch
    // would take anyway (512 LDS slots, all-LDS up to 170-node grids, + one HBM level up to 256), 2 = 255 LDS slots + two HBM
    // levels (1.5 KB of LDS per field).  Grids with 16-bit node ids only.  Measurements: DESIGN.md section 4.
    const int gp8 = ctx->opts.count("fmm.gp8") ? ctx->opts["fmm.gp8"] : 0;
    if (gp8 && small && cap <= 768) {
      if (gp8 == 2) rc = run_fmm<256, unsigned short, true, 8>(ctx, A0, nfield, nn, nr, d_status, hsp, async, &fin);
      else if (cap <= 512) rc = run_fmm<512, unsigned short, false, 8>(ctx, A0, nfield, nn, nr, d_status, hsp, async, &fin);
      else rc = run_fmm<512, unsigned short, true, 8>(ctx, A0, nfield, nn, nr, d_status, hsp, async, &fin);
    } else
    if (cap <= 64) rc = small ? run_fmm<64, unsigned short>(ctx, A0, nfield, nn, nr, d_status, hsp, async, &fin) : run_fmm<64, int>(ctx, A0, nfield, nn, nr, d_status, hsp, async, &fin);
    else if (cap <= 512) rc = small ? run_fmm<512, unsigned short>(ctx, A0, nfield, nn, nr, d_status, hsp, async, &fin) : run_fmm<512, int>(ctx, A0, nfield, nn, nr, d_status, hsp, async, &fin);
    // grids of 171 .. 256 nodes a side (S-256) with more fields than the 768-slot heaps hold at once (8 workgroups of 4 per CU):
    // levels 1-9 in LDS + level 10 in HBM -- 12 workgroups per CU, a third wavefront per SIMD, and time slicing (run_fmm) keeps
    // them all busy to the end.  The 13 % of the fields whose band outgrows 511 entries pay for the HBM level (-17 % at equal
    // occupancy), so batches that fit the 768-slot heaps stay there.  Option fmm.hyb512 = 1 / 2 forces it on / off.
    else if (cap <= 768 && small && use_hyb512) rc = run_fmm<512, unsigned short, true>(ctx, A0, nfield, nn, nr, d_status, hsp, async, &fin);
    else if (cap <= 768) rc = small ? run_fmm<768, unsigned short>(ctx, A0, nfield, nn, nr, d_status, hsp, async, &fin) : run_fmm<768, int>(ctx, A0, nfield, nn, nr, d_status, hsp, async, &fin);
    // grids of 257 .. 682 nodes a side (S-512): levels 1-9 in LDS, levels 10 and 11 in HBM -- 16 KB of LDS per workgroup, ten
    // workgroups per CU instead of five.  These kernels wait on latencies (1.25 wavefronts per SIMD with 1024 LDS slots), so twice
    // the wavefronts for one or two more dependent memory accesses per pop is a good trade: 341 x 341 nodes 20.2 -> 31.3 k
    // fields/s against the all-LDS 1024-slot heap, S-512 7 950 -> 11 300 against levels 1-10 in LDS (same box, bit-identical).
    // (option fmm.hyb2 = 2: the forms below)
    else if (cap <= 2048 && use_hyb2) rc = run_fmm<512, int, true>(ctx, A0, nfield, nn, nr, d_status, hsp, async, &fin);
    else if (cap <= 1024) rc = run_fmm<1024, int>(ctx, A0, nfield, nn, nr, d_status, hsp, async, &fin);
    // (fmm.hyb2 = 2) grids of 342 .. 682 nodes a side: levels 1-10 in LDS + levels 11 (and, never reached there, 12) in HBM
    else if (cap <= 2048 && !(ctx->opts.count("fmm.no_hybrid") && ctx->opts["fmm.no_hybrid"])) rc = run_fmm<1024, int, true>(ctx, A0, nfield, nn, nr, d_status, hsp, async, &fin);
    // grids above 682 nodes a side: levels 1-10 in LDS, 11 and 12 in HBM (bands up to 4 095 entries = 1 365 nodes a side without
    // the spill kernel, five workgroups per CU instead of the two of the all-LDS 2048-slot heap)
    else if (use_hyb2) rc = run_fmm<1024, int, true>(ctx, A0, nfield, nn, nr, d_status, hsp, async, &fin);
    else if (cap <= 1536) rc = run_fmm<1536, int>(ctx, A0, nfield, nn, nr, d_status, hsp, async, &fin);
    else rc = run_fmm<2048, int>(ctx, A0, nfield, nn, nr, d_status, hsp, async, &fin);
    if (rc) return rc;
    // first failing field, like the reference's STOP
    auto first_error = [ctx, hsp, nfield]() -> int {
      const std::vector<int> &hs = *hsp;
      for (int i = 0; i < nfield; i++)
        if (hs[i]) return dz_fail(ctx, hs[i], "field %d: source lies outside bounds of model", i);
      return 0;
    };
    if (fin) {   // the launch is on its way: the rest when the ray call (or dazim_sync, dazim_free, the next eikonal call) asks for it
      ctx->fmm_busy = true;
      busy.keep = true;
      ctx->fmm_finish = [ctx, fin, first_error]() -> int {
        const int r = fin();
        ctx->fmm_busy = false;
        return r ? r : first_error();
      };
      return 0;
    }
    rc = first_error();
  }
  int rc2;
  if ((rc2 = veln.finish()) || (rc2 = ttn.finish()) || (rc2 = ttnr.finish()) || (rc2 = nstsr.finish()) ||
      (rc2 = boxes.finish()) || (rc2 = status.finish()))
    return rc2;
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  return rc;
}
