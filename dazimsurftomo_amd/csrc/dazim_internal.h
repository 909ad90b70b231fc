// dazim_internal.h -- shared plumbing of libdazim_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/dazim.h"

struct dazim_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::string err;
  std::map<std::string, double> ksec;  // last measured kernel seconds by name
  int num_cu = 256;
  std::map<std::string, int> opts;     // dazim_set_option
  // reusable device scratch, grown on demand (never shrunk) so that repeated calls do not hipMalloc
  std::map<std::string, std::pair<void *, size_t>> scratch;
  // pinned host buffers by name (dz_pinned): where the step's small device-to-host reads land -- a copy into pageable memory is
  // staged by a blit KERNEL, which waits for a free wavefront slot behind persistent workgroups (the dispersion copies held the
  // chip for 20 ms while a 4-byte read waited, profiles/r5); into pinned memory it is a DMA transfer
  std::map<std::string, std::pair<void *, size_t>> pinned;
  bool fmm_busy = false;   // an eikonal call is using its scratch blocks (dz_trim_caches must not free them)
  // RCCL communicator of a row-sharded solve (dazim_comm_init); nullptr = single GPU, RCCL never touched
  // staging blocks for host-pointer arguments (DzBuf): released blocks are kept and handed out again, because a
  // hipMalloc + hipFree pair per staged array costs milliseconds in a program that calls the library once per outer iteration
  struct StageBlock { void *p; size_t bytes; bool busy; };
  std::vector<StageBlock> stage;
  // matrix arrays (CSR values / columns / row pointers of G, hundreds of MB to GB each): a program builds and frees one G per
  // outer iteration, and a hipMalloc + hipFree pair of that size costs about a millisecond each -- freed arrays are kept
  // (a few, best fit) and handed out again
  std::vector<StageBlock> big;
  // auxiliary stream (option disp.async): the perturbed copies of the dispersion kernel run there while the main stream goes on
  // with the column's own curves, the gridder and the eikonal fields; whoever consumes the depth kernels joins first (dz_join_aux)
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_a0 = nullptr, ev_a1 = nullptr;
  bool aux_pending = false, aux_timed = false;
  // device ranges the auxiliary stream's pending work reads or writes (the model and the three kernel tables): a host copy that
  // touches none of them does not have to wait for it (dazim_memcpy_h2d / _d2h)
  struct AuxRange { const char *p; size_t bytes; };
  std::vector<AuxRange> aux_ranges;
  // what the join of the auxiliary stream still has to do on the MAIN stream (dazim_dispersion_kernels_sharded: the all-gather
  // of the ranks' blocks of the depth-kernel tables); run once by dz_join_aux / dazim_sync, dropped by dazim_destroy
  int (*aux_epilogue)(dazim_ctx *) = nullptr;
  struct ShardPending {
    int nx = 0, ny = 0, nz = 0, kmax = 0;
    const double *send = nullptr;                     // this rank's blocks of sen_vs | sen_vp | sen_rho
    double *svs = nullptr, *svp = nullptr, *srho = nullptr;   // the complete tables (device)
  } shard;
  // eikonal fields a dazim_fmm_batch call with ttn == NULL left inside the library for dazim_rays_build_G* with ttn == NULL: the
  // finished node words of every field in the eikonal kernel's own 4 x 4 tiles (scratch "fmm.rec_c" / "fmm.ttn_tiled", which
  // dz_trim_caches leaves alone while `tiled` is set); field f at tiled + (tslot ? tslot[f] : f) * stride
  struct TiledFields {
    const unsigned *tiled = nullptr;
    const int *tslot = nullptr;
    int nfield = 0, nnx = 0, nnz = 0, stride = 0, tsh = 0;
    const int *fdone = nullptr;   // asynchronous eikonal call: per field 0 = still marching, 1 = finished, 2 = band overflow (rerun pending)
    unsigned nwg = 0;                    // workgroups of that launch
    volatile unsigned *hprog = nullptr;  // host-mapped: tasks handed out so far per XCD range of the asynchronous launch, of total_tasks:
    unsigned total_tasks = 0;            // the ray call launches its count pass when the queue is nearly empty (not before: see rays.hip)
  } fields;
  // Option fmm.async (with ttn == NULL, device-resident arguments and a time-sliced batch): dazim_fmm_batch returns when its launch
  // is enqueued, and the dazim_rays_build_G* call that follows runs its count pass on a third stream as non-blocking passes over
  // the quads of rays whose fields' completion flags are set -- its workgroups are dispatched as the eikonal launch's persistent
  // workgroups leave: the ray kernel fills the TAIL of the eikonal launch (profiles/r6_tail_fill.md).  fmm_finish = what the
  // eikonal call still owes (statuses, spill reruns, timers); run by the ray call, by dazim_sync / dazim_free / a copy / the next
  // eikonal or dispersion call, whichever comes first (dz_fmm_finish).
  std::function<int()> fmm_finish;
  unsigned *hprog = nullptr;           // pinned, device-visible: 8 progress words of an asynchronous eikonal launch (dz_async_init)
  hipStream_t stream3 = nullptr;
  hipEvent_t ev_f0 = nullptr, ev_f1 = nullptr, ev_pre = nullptr, ev_r0 = nullptr, ev_r1 = nullptr;
  void *comm = nullptr;
  void (*comm_release)(dazim_ctx *) = nullptr;   // set by dazim_comm_init: dazim_destroy must not leak the communicator
  int nranks = 1, rank = 0;
};

int dz_fail(dazim_ctx *c, int code, const char *fmt, ...);

// ---- the communicator of a multi-rank run (comm.hip) ---------------------------------------------------------------------------
struct DzComm {
  ncclComm_t nccl = nullptr;
  std::string dir;        // non-empty: the file transport (tests: several ranks on ONE GPU)
  std::string tag;        // ... the nonce every file of this communicator carries in its name
  unsigned seq = 0;
  int nranks = 1, rank = 0;
};
enum { DZ_F32 = 0, DZ_F64 = 1, DZ_I64 = 2, DZ_SUM = 0, DZ_MAX = 1 };
// recv[r*bytes .. (r+1)*bytes) = rank r's `bytes` bytes at send, every rank the same; DEVICE buffers, on the context's stream
int dz_allgather(dazim_ctx *ctx, DzComm *c, const void *send_dev, void *recv_dev, size_t bytes);
// in-place sum / max over the ranks of a DEVICE buffer on the context's stream: all-gather + one kernel that combines the ranks'
// values in RANK ORDER (the same bits on every rank, with every transport and every rank count's grouping)
int dz_allreduce(dazim_ctx *ctx, DzComm *c, void *dbuf, size_t count, int dtype, int op);
// a rank that fails on its own after the others may already wait in a collective: abort the communicator, detach it
void dz_comm_abort(dazim_ctx *ctx);
// even contiguous split of n items over the ranks (dazim_shard_rows of host/dazim_mod.f90, shard_rows of distributed.py)
static inline void dz_shard_even(int64_t n, int world, int rank, int64_t *lo, int64_t *hi) {
  const int64_t base = n / world, rem = n % world;
  *lo = rank * base + (rank < rem ? rank : rem);
  *hi = *lo + base + (rank < rem ? 1 : 0);
}

#define DZ_NCCL(call)                                                                                        \
  do {                                                                                                        \
    ncclResult_t r_ = (call);                                                                                 \
    if (r_ != ncclSuccess) return dz_fail(ctx, -2000 - (int)r_, "%s:%d %s -> %s", __FILE__, __LINE__, #call, ncclGetErrorString(r_)); \
  } while (0)
int dz_aux_init(dazim_ctx *ctx);                  // creates the auxiliary stream and its events on first use
int dz_async_init(dazim_ctx *ctx);                // ... the third stream and the events of an asynchronous eikonal call
int dz_fmm_finish(dazim_ctx *ctx);                // completes a pending asynchronous eikonal call (no-op without one)
int dz_join_aux(dazim_ctx *ctx);                  // main stream waits for what the auxiliary stream was given (no host wait)
int dz_join_aux_if_touched(dazim_ctx *ctx, const void *dev, size_t bytes);   // ... only if [dev, dev + bytes) overlaps what it works on

#define DZ_HIP(call)                                                                           \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return dz_fail(ctx, -(int)e_ - 1000, "%s:%d %s -> %s", __FILE__, __LINE__, #call,       \
                     hipGetErrorString(e_));                                                   \
  } while (0)

// The eikonal kernel's 4 x 4-tile layout of a grid (fmm.hip: tile_shift / tile_x / tile_z, which these restate for the ray kernel):
// node (x0, z0), 0-based, is word dz_tile_x(x0, tsh) + dz_tile_z(z0), tsh = dz_tile_shift(nnz)
__host__ __device__ constexpr int dz_tile_shift(int nz) { int l = 0; while ((1 << l) < ((nz + 3) >> 2)) l++; return l + 4; }
__host__ __device__ __forceinline__ constexpr int dz_tile_x(int x0, int tsh) { return ((x0 >> 2) << tsh) + ((x0 & 3) << 2); }
__host__ __device__ __forceinline__ constexpr int dz_tile_z(int z0) { return ((z0 & ~3) << 2) | (z0 & 3); }

// named scratch buffer of at least `bytes` bytes
int dz_scratch(dazim_ctx *ctx, const char *name, size_t bytes, void **out);
// named PINNED host buffer of at least `bytes` bytes (option ctx.pinned = 0: plain malloc'ed memory, the behaviour before round 6)
int dz_pinned(dazim_ctx *ctx, const char *name, size_t bytes, void **out);

size_t dz_trim_caches(dazim_ctx *ctx);                                // free all idle cached blocks; bytes released
hipError_t dz_malloc_retry(dazim_ctx *ctx, void **p, size_t bytes);   // hipMalloc; on out-of-memory trim the caches and retry once
bool dz_is_device_ptr(const void *p);
int dz_stage_get(dazim_ctx *ctx, size_t bytes, void **out);   // a device block of >= bytes from the context's staging cache
void dz_stage_put(dazim_ctx *ctx, void *p);                   // give it back (kept for reuse; freed by dazim_destroy)
int dz_big_get(dazim_ctx *ctx, size_t bytes, void **out);     // a device array for a matrix (see dazim_ctx::big)
void dz_big_put(dazim_ctx *ctx, void *p);                     // return it (any hipMalloc'ed pointer is accepted; ctx may be null)

// take ownership of device CSR arrays whose allocations hold cap_m rows / cap_nnz entries (0 = exactly m / nnz); sparse.hip
extern "C" int dz_csr_adopt_cap(dazim_ctx *ctx, int64_t m, int64_t n, int64_t nnz, int64_t *rowptr, int *col, float *val,
                                int64_t cap_m, int64_t cap_nnz, dazim_csr **out);

// attach B as the "dense twin" of A (A owns it until dazim_csr_take_twin hands it out; dazim_csr_free(A) frees an attached twin)
extern "C" int dz_csr_set_twin(dazim_ctx *ctx, dazim_csr *A, dazim_csr *B);

// every a[i] of a DEVICE array inside lo..hi?  Returns 0, or DAZIM_E_BAD_ARG with "<what> outside lo..hi" as the message.
int dz_check_range(dazim_ctx *ctx, const int *a_dev, int64_t n, int lo, int hi, const char *what);

// Staging helper: wraps a user pointer that may live on the host or on the device.
template <class T>
struct DzBuf {
  dazim_ctx *ctx = nullptr;
  T *user = nullptr;
  T *dev = nullptr;
  size_t n = 0;
  bool staged = false, out = false;
  int init(dazim_ctx *c, const T *p, size_t count, bool copy_in, bool copy_out) {
    ctx = c;
    user = const_cast<T *>(p);
    n = count;
    out = copy_out;
    if (!p || count == 0) return 0;
    if (dz_is_device_ptr(p)) {
      dev = user;
      return 0;
    }
    staged = true;
    {
      void *blk = nullptr;
      int rc_ = dz_stage_get(ctx, n * sizeof(T), &blk);
      if (rc_) return rc_;
      dev = (T *)blk;
    }
    if (copy_in) DZ_HIP(hipMemcpyAsync(dev, user, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return 0;
  }
  int finish() {
    if (staged && out && dev)
      DZ_HIP(hipMemcpyAsync(user, dev, n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    return 0;
  }
  ~DzBuf() {
    if (staged && dev) {
      (void)hipStreamSynchronize(ctx->stream);
      dz_stage_put(ctx, dev);
    }
  }
};

// time a region on the ctx stream with HIP events and record it under `name`
struct DzTimer {
  dazim_ctx *ctx;
  const char *name;
  DzTimer(dazim_ctx *c, const char *n) : ctx(c), name(n) { (void)hipEventRecord(ctx->ev0, ctx->stream); }
  void stop() {
    (void)hipEventRecord(ctx->ev1, ctx->stream);
    (void)hipEventSynchronize(ctx->ev1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->ksec[name] = ms * 1e-3;
  }
};
