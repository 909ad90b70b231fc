// comm.hip -- what the ranks of a multi-GPU run exchange: the communicator, its two transports, the two collectives
// built on it (all-gather; all-reduce = all-gather + a rank-ordered combine) and the model-sharded forms of the dispersion / TI
// depth-kernel calls (the reference's only parallel loop: OMP over the model's columns, inv/CalSurfG.f90:39-43 called at :1078,
// inv/depthkernelTI.f90:2-112).
//
// Transports.  RCCL (dazim_comm_init): ncclAllGather on the context's stream, over xGMI -- the product.  Files
// (dazim_comm_init_files): every collective is staged through the host and a directory all ranks can see -- there so that the
// WHOLE multi-rank path (sharded tables, row-sharded LSMR, the sharded host program) runs with several processes on a box with ONE
// GPU, where RCCL refuses a device used twice (tests/test_multirank_files_gpu.py, tests/test_rehearsal_gpu.py).
//
// Determinism (SURVEY 8e "fix reduction order").  Both transports only MOVE bytes (all-gather).  Every sum over the ranks is then
// formed on the device by ONE kernel, k_rank_reduce (or LSMR's k_beta_axpby), which adds the ranks' values in rank order:
// ((r0 + r1) + r2) + ...  So a run over RCCL returns the bits of the same run over files, on every rank, whatever ring or tree RCCL
// would have picked for an all-reduce.  Option comm.allreduce = 1 takes ncclAllReduce instead (RCCL's own order).
#include "dazim_internal.h"

#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <random>

namespace {

// out[i] = v_0[i] (+ | max) v_1[i] (+ | max) ... in rank order, v_r = gathered + r*stride
template <class T>
__global__ void k_rank_reduce(const T *__restrict__ gathered, int nranks, size_t count, size_t stride, T *__restrict__ out, int op) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    T a = gathered[i];
    for (int r = 1; r < nranks; r++) {
      const T b = gathered[(size_t)r * stride + i];
      a = op == DZ_SUM ? a + b : (a > b ? a : b);
    }
    out[i] = a;
  }
}

// table[a][ncol] <-> block[a][cb] (columns c0 .. c0 + cb of the table); to_block: table -> block, else block -> table
template <class T>
__global__ void k_cols_copy(T *__restrict__ table, size_t ncol, T *__restrict__ block, size_t cb, size_t c0, size_t na, bool to_block) {
  const size_t total = na * cb;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t a = i / cb, c = i - a * cb;
    if (to_block) block[i] = table[a * ncol + c0 + c];
    else table[a * ncol + c0 + c] = block[i];
  }
}

// The gathered blocks of all ranks -> complete tables.  Rank r computed the model rows [lo_r, hi_r) (dz_shard_even of ny), i.e. the
// columns lo_r*nx .. hi_r*nx, and stored table q (0 .. ntab) as [na][cb_r] at gathered + r*stride + q*na*cbmax.
template <class T>
__global__ void k_join_blocks(const T *__restrict__ gathered, size_t stride, int nranks, int nx, int ny, size_t na, size_t cbmax,
                              int ntab, T *__restrict__ t0, T *__restrict__ t1, T *__restrict__ t2) {
  const size_t ncol = (size_t)nx * ny, total = (size_t)ntab * na * ncol;
  const int base = ny / nranks, rem = ny % nranks;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t q = i / (na * ncol), j = i - q * na * ncol, a = j / ncol, c = j - a * ncol;
    const int row = (int)(c / nx);
    int r;
    if (row < rem * (base + 1)) r = row / (base + 1);
    else r = rem + (row - rem * (base + 1)) / (base > 0 ? base : 1);
    const int lo = r * base + (r < rem ? r : rem), cb = (base + (r < rem ? 1 : 0)) * nx;
    const T v = gathered[(size_t)r * stride + q * na * cbmax + a * cb + (c - (size_t)lo * nx)];
    T *t = q == 0 ? t0 : (q == 1 ? t1 : t2);
    t[j] = v;
  }
}

inline unsigned grid_for(size_t n) {
  size_t b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// ---- file transport ------------------------------------------------------------------------------------------------------------
constexpr int FILE_WAIT_US = 200, FILE_WAIT_SPINS = 600000;   // <= 120 s per file

bool write_atomic(const std::string &path, const void *data, size_t bytes) {
  const std::string tmp = path + ".tmp";
  FILE *f = fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const bool ok = fwrite(data, 1, bytes, f) == bytes;
  fclose(f);
  return ok && rename(tmp.c_str(), path.c_str()) == 0;
}
bool read_whole(const std::string &path, std::string &out) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return false;
  out.clear();
  char buf[4096];
  size_t g;
  while ((g = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, g);
  fclose(f);
  return true;
}

// every rank publishes `bytes` bytes as <dir>/<tag>.x<seq>.<rank> (temporary name, then renamed) and reads the others': recv =
// [nranks][bytes].  A rank's file of collective seq - 1 goes once it has read everybody's file of collective seq: whoever published
// seq had finished reading seq - 1.
int files_exchange(dazim_ctx *ctx, DzComm *c, const void *send, size_t bytes, void *recv) {
  const unsigned seq = ++c->seq;
  auto path = [&](unsigned s_, int r) { return c->dir + "/" + c->tag + ".x" + std::to_string(s_) + "." + std::to_string(r); };
  if (!write_atomic(path(seq, c->rank), send, bytes)) return dz_fail(ctx, -2100, "file transport: cannot publish %s", path(seq, c->rank).c_str());
  for (int r = 0; r < c->nranks; r++) {
    char *dst = (char *)recv + (size_t)r * bytes;
    if (r == c->rank) { memcpy(dst, send, bytes); continue; }
    const std::string pr = path(seq, r);
    FILE *f = nullptr;
    for (int spin = 0; spin < FILE_WAIT_SPINS && !(f = fopen(pr.c_str(), "rb")); spin++) usleep(FILE_WAIT_US);
    if (!f) return dz_fail(ctx, -2101, "file transport: rank %d never published collective %u (%s)", r, seq, pr.c_str());
    const size_t got = fread(dst, 1, bytes, f);
    char extra;
    const bool more = fread(&extra, 1, 1, f) == 1;
    fclose(f);
    if (got != bytes || more) return dz_fail(ctx, -2102, "file transport: %s does not hold the %zu bytes expected (the ranks disagree on a size)", pr.c_str(), bytes);
  }
  if (seq > 1) (void)remove(path(seq - 1, c->rank).c_str());
  return 0;
}

}  // namespace

int dz_allgather(dazim_ctx *ctx, DzComm *c, const void *send_dev, void *recv_dev, size_t bytes) {
  if (bytes == 0) return 0;
  if (c->dir.empty()) {
    DZ_NCCL(ncclAllGather(send_dev, recv_dev, bytes, ncclChar, c->nccl, ctx->stream));
    return 0;
  }
  std::vector<char> hs(bytes), hr(bytes * (size_t)c->nranks);
  DZ_HIP(hipMemcpyAsync(hs.data(), send_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  const int rc = files_exchange(ctx, c, hs.data(), bytes, hr.data());
  if (rc) return rc;
  DZ_HIP(hipMemcpyAsync(recv_dev, hr.data(), hr.size(), hipMemcpyHostToDevice, ctx->stream));
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

int dz_allreduce(dazim_ctx *ctx, DzComm *c, void *dbuf, size_t count, int dtype, int op) {
  if (count == 0) return 0;
  const size_t esz = dtype == DZ_F32 ? 4 : 8, bytes = count * esz;
  if (c->dir.empty() && ctx->opts.count("comm.allreduce") && ctx->opts["comm.allreduce"] == 1) {   // RCCL's own all-reduce (its order)
    const ncclDataType_t t = dtype == DZ_F32 ? ncclFloat : (dtype == DZ_F64 ? ncclDouble : ncclInt64);
    DZ_NCCL(ncclAllReduce(dbuf, dbuf, count, t, op == DZ_SUM ? ncclSum : ncclMax, c->nccl, ctx->stream));
    return 0;
  }
  void *g;
  int rc;
  if ((rc = dz_scratch(ctx, "comm.gather", bytes * (size_t)c->nranks, &g))) return rc;
  if ((rc = dz_allgather(ctx, c, dbuf, g, bytes))) return rc;
  if (dtype == DZ_F32)
    hipLaunchKernelGGL(k_rank_reduce<float>, dim3(grid_for(count)), dim3(256), 0, ctx->stream, (const float *)g, c->nranks, count, count, (float *)dbuf, op);
  else if (dtype == DZ_F64)
    hipLaunchKernelGGL(k_rank_reduce<double>, dim3(grid_for(count)), dim3(256), 0, ctx->stream, (const double *)g, c->nranks, count, count, (double *)dbuf, op);
  else
    hipLaunchKernelGGL(k_rank_reduce<long long>, dim3(grid_for(count)), dim3(256), 0, ctx->stream, (const long long *)g, c->nranks, count, count, (long long *)dbuf, op);
  DZ_HIP(hipGetLastError());
  return 0;
}

void dz_comm_abort(dazim_ctx *ctx) {
  if (!ctx || !ctx->comm) return;
  DzComm *c = (DzComm *)ctx->comm;
  if (c->nccl) (void)ncclCommAbort(c->nccl);
  delete c;
  ctx->comm = nullptr;
  ctx->nranks = 1;
  ctx->rank = 0;
}

extern "C" {

int dazim_comm_unique_id(void *id128) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  if (!id128) return DAZIM_E_BAD_ARG;
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return -2000;
  memcpy(id128, &id, sizeof id);
  return 0;
}

int dazim_comm_init(dazim_ctx *ctx, int nranks, int rank, const void *id128) {
  if (!ctx || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return dz_fail(ctx, DAZIM_E_BAD_ARG, "bad arguments to dazim_comm_init");
  if (ctx->comm) return dz_fail(ctx, DAZIM_E_BAD_ARG, "a communicator is already attached");
  DZ_HIP(hipSetDevice(ctx->device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  ncclComm_t comm;
  DZ_NCCL(ncclCommInitRank(&comm, nranks, id, rank));
  DzComm *c = new DzComm;
  c->nccl = comm;
  c->nranks = nranks;
  c->rank = rank;
  ctx->comm = (void *)c;
  ctx->comm_release = [](dazim_ctx *cx) { (void)dazim_comm_free(cx); };
  ctx->nranks = nranks;
  ctx->rank = rank;
  return 0;
}

// The file communicator agrees on a nonce first, so that a directory used before -- the files of an earlier communicator, of a
// run that died -- cannot feed this one stale data: every rank publishes join.<rank> = "<nranks> <token>" (a fresh random token);
// rank 0 collects them, checks the rank counts, and publishes `nonce` = "<tag> <nranks> <token_0> ... <token_N-1>"; a rank
// accepts a nonce file only if it lists ITS token at ITS position (a stale one cannot), and every later file carries <tag> in its
// name.  One empty exchange ends the set-up (everybody has read the nonce before rank 0 may remove it again).
int dazim_comm_init_files(dazim_ctx *ctx, int nranks, int rank, const char *dir) {
  if (!ctx || !dir || !dir[0] || nranks < 1 || rank < 0 || rank >= nranks) return dz_fail(ctx, DAZIM_E_BAD_ARG, "bad arguments to dazim_comm_init_files");
  if (ctx->comm) return dz_fail(ctx, DAZIM_E_BAD_ARG, "a communicator is already attached");
  struct stat sb;
  if (stat(dir, &sb) != 0 || !S_ISDIR(sb.st_mode)) return dz_fail(ctx, DAZIM_E_BAD_ARG, "file transport: %s is not a directory", dir);
  std::random_device rd;
  const unsigned long long token = ((unsigned long long)rd() << 32) ^ rd() ^ ((unsigned long long)getpid() << 20) ^
                                   (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
  const std::string d(dir);
  auto join = [&](int r) { return d + "/join." + std::to_string(r); };
  char line[128];
  snprintf(line, sizeof line, "%d %llx", nranks, token);
  if (!write_atomic(join(rank), line, strlen(line))) return dz_fail(ctx, -2100, "file transport: cannot write %s", join(rank).c_str());
  std::string tag;
  if (rank == 0) {
    std::string all;
    for (int r = 0; r < nranks; r++) {
      std::string s;
      int nr = 0;
      unsigned long long tk = 0;
      bool got = false;
      for (int spin = 0; spin < FILE_WAIT_SPINS && !got; spin++) {
        got = read_whole(join(r), s) && sscanf(s.c_str(), "%d %llx", &nr, &tk) == 2;
        if (!got) usleep(FILE_WAIT_US);
      }
      if (!got) return dz_fail(ctx, -2101, "file transport: rank %d never joined (%s)", r, join(r).c_str());
      if (nr != nranks) return dz_fail(ctx, DAZIM_E_BAD_ARG, "file transport: rank %d was started with %d ranks, rank 0 with %d", r, nr, nranks);
      snprintf(line, sizeof line, " %llx", tk);
      all += line;
    }
    snprintf(line, sizeof line, "c%llx", token);
    tag = line;
    const std::string body = tag + " " + std::to_string(nranks) + all;
    if (!write_atomic(d + "/nonce", body.data(), body.size())) return dz_fail(ctx, -2100, "file transport: cannot write %s/nonce", dir);
  } else {
    bool ok = false;
    for (int spin = 0; spin < FILE_WAIT_SPINS && !ok; spin++) {
      std::string s;
      if (read_whole(d + "/nonce", s)) {
        char tg[64];
        int nr = 0, off = 0;
        if (sscanf(s.c_str(), "%63s %d%n", tg, &nr, &off) == 2) {
          const char *q = s.c_str() + off;
          unsigned long long tk = 0;
          int r = 0, adv = 0;
          bool mine = false;
          while (sscanf(q, " %llx%n", &tk, &adv) == 1) {
            if (r == rank && tk == token) mine = true;
            r++;
            q += adv;
          }
          if (mine && r == nr) {
            if (nr != nranks) return dz_fail(ctx, DAZIM_E_BAD_ARG, "file transport: rank 0 was started with %d ranks, this rank with %d", nr, nranks);
            tag = tg;
            ok = true;
          }
        }
      }
      if (!ok) usleep(FILE_WAIT_US);
    }
    if (!ok) return dz_fail(ctx, -2101, "file transport: no nonce from rank 0 that lists this rank's token (stale files in %s? use an empty directory)", dir);
  }
  DzComm *c = new DzComm;
  c->dir = d;
  c->tag = tag;
  c->nranks = nranks;
  c->rank = rank;
  char one = 0, *all1 = new char[nranks];
  const int rcx = files_exchange(ctx, c, &one, 1, all1);   // everybody has the nonce
  delete[] all1;
  (void)remove(join(rank).c_str());
  if (rcx) { delete c; return rcx; }
  ctx->comm = (void *)c;
  ctx->comm_release = [](dazim_ctx *cx) { (void)dazim_comm_free(cx); };
  ctx->nranks = nranks;
  ctx->rank = rank;
  return 0;
}

int dazim_comm_free(dazim_ctx *ctx) {
  if (!ctx) return DAZIM_E_BAD_ARG;
  if (ctx->comm && ctx->aux_epilogue) {   // (tables of a sharded dispersion call still wait for their gather: it needs this communicator)
    const int rcj = dz_join_aux(ctx);
    if (rcj) return rcj;
  }
  if (ctx->comm) {
    DzComm *c = (DzComm *)ctx->comm;
    DZ_HIP(hipStreamSynchronize(ctx->stream));
    if (c->nccl) DZ_NCCL(ncclCommDestroy(c->nccl));
    if (!c->dir.empty()) {
      // leave the directory as it was found: every rank says goodbye; rank 0 waits (a while) for all of them -- nobody reads
      // any more then -- and removes this communicator's files and the nonce.  Ranks that died leave files behind, which the
      // nonce in their names keeps out of every later communicator's way.
      const std::string bye = c->dir + "/" + c->tag + ".bye.";
      (void)write_atomic(bye + std::to_string(c->rank), "", 0);
      if (c->rank == 0) {
        bool all = false;
        for (int spin = 0; spin < 50000 && !all; spin++) {   // <= 10 s
          all = true;
          for (int r = 0; r < c->nranks && all; r++) all = access((bye + std::to_string(r)).c_str(), F_OK) == 0;
          if (!all) usleep(FILE_WAIT_US);
        }
        if (all) {
          if (DIR *dp = opendir(c->dir.c_str())) {
            std::vector<std::string> names;
            while (struct dirent *e = readdir(dp))
              if (strncmp(e->d_name, c->tag.c_str(), c->tag.size()) == 0 && e->d_name[c->tag.size()] == '.') names.push_back(e->d_name);
            closedir(dp);
            for (const auto &nm : names) (void)remove((c->dir + "/" + nm).c_str());
          }
          (void)remove((c->dir + "/nonce").c_str());
        }
      }
    }
    delete c;
  }
  ctx->comm = nullptr;
  ctx->nranks = 1;
  ctx->rank = 0;
  return 0;
}

// sum / max over the ranks of `count` values, in place; buf is a host or a device pointer.  dtype: 0 fp32, 1 fp64, 2 int64; op: 0 sum,
// 1 max.  Without a communicator (one rank) nothing happens.  What the sharded host program reduces with: residual statistics,
// column sums, the predicted times it writes out.
int dazim_comm_allreduce(dazim_ctx *ctx, void *buf, int64_t count, int dtype, int op) {
  if (!ctx || !buf || count < 0 || dtype < 0 || dtype > 2 || op < 0 || op > 1) return dz_fail(ctx, DAZIM_E_BAD_ARG, "bad arguments to dazim_comm_allreduce");
  if (!ctx->comm || count == 0) return 0;
  DzComm *c = (DzComm *)ctx->comm;
  DZ_HIP(hipSetDevice(ctx->device));
  const size_t bytes = (size_t)count * (dtype == DZ_F32 ? 4 : 8);
  int rc;
  if (dz_is_device_ptr(buf)) {
    if ((rc = dz_allreduce(ctx, c, buf, (size_t)count, dtype, op))) return rc;
    DZ_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
  }
  void *p;
  if ((rc = dz_scratch(ctx, "comm.stage", bytes, &p))) return rc;
  DZ_HIP(hipMemcpyAsync(p, buf, bytes, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = dz_allreduce(ctx, c, p, (size_t)count, dtype, op))) return rc;
  DZ_HIP(hipMemcpyAsync(buf, p, bytes, hipMemcpyDeviceToHost, ctx->stream));
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

// recv[r*count .. (r+1)*count) = rank r's `count` values at send, on every rank; send / recv: host or device pointers (each on its
// own); dtype as above.  Without a communicator recv = send.
int dazim_comm_allgather(dazim_ctx *ctx, const void *send, void *recv, int64_t count, int dtype) {
  if (!ctx || !send || !recv || count < 0 || dtype < 0 || dtype > 2) return dz_fail(ctx, DAZIM_E_BAD_ARG, "bad arguments to dazim_comm_allgather");
  if (count == 0) return 0;
  DZ_HIP(hipSetDevice(ctx->device));
  const size_t bytes = (size_t)count * (dtype == DZ_F32 ? 4 : 8);
  DzComm *c = (DzComm *)ctx->comm;
  const int nr = c ? c->nranks : 1;
  const bool sd = dz_is_device_ptr(send), rd = dz_is_device_ptr(recv);
  if (!c) {
    DZ_HIP(hipMemcpyAsync(recv, send, bytes, hipMemcpyDefault, ctx->stream));
    DZ_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
  }
  int rc;
  void *ps = const_cast<void *>(send), *pr = recv;
  if (!sd) {
    if ((rc = dz_scratch(ctx, "comm.stage", bytes, &ps))) return rc;
    DZ_HIP(hipMemcpyAsync(ps, send, bytes, hipMemcpyHostToDevice, ctx->stream));
  }
  if (!rd && (rc = dz_scratch(ctx, "comm.gather", bytes * (size_t)nr, &pr))) return rc;
  if ((rc = dz_allgather(ctx, c, ps, pr, bytes))) return rc;
  if (!rd) DZ_HIP(hipMemcpyAsync(recv, pr, bytes * (size_t)nr, hipMemcpyDeviceToHost, ctx->stream));
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

// ---- the model's tables, sharded over the ranks --------------------------------------------------------------------------------
// The dispersion curves and depth kernels belong to the MODEL, which every rank holds; its columns are independent (the reference's
// OMP loop over jj, inv/CalSurfG.f90:39-43).  Rank r computes the model rows [lo_r, hi_r) = dz_shard_even(ny) -- columns are numbered
// jj*nx + ii, so a block of rows is a block of columns -- with the SAME kernels as the unsharded call (every column is computed by
// exactly one rank from the same numbers: the joined tables are bit-identical to the single-rank tables), and all-gathers join the
// blocks: pvRc at once (the eikonal solve needs every column's curve), the three depth-kernel tables when the auxiliary stream is
// joined (dz_join_aux: dazim_rays_build_G*, dazim_sync, a copy that touches them ...) -- with option disp.async the perturbed copies
// of this rank's block run beside the eikonal kernel as in the single-rank call, and their gather follows them on the main stream.
// Every rank must make the same calls in the same order (SPMD); the tables passed for sen_* must be device arrays on every rank or
// host arrays on every rank.
static int shard_join_sen(dazim_ctx *ctx) {
  DzComm *c = (DzComm *)ctx->comm;
  const auto &P = ctx->shard;
  if (!c) return dz_fail(ctx, DAZIM_E_BAD_ARG, "sharded dispersion tables: the communicator was freed before the tables were joined");
  int64_t lo, hi;
  dz_shard_even(P.ny, c->nranks, 0, &lo, &hi);
  const size_t cbmax = (size_t)(hi - lo) * P.nx, na = (size_t)P.nz * P.kmax, per = 3 * na * cbmax;
  void *g;
  int rc;
  if ((rc = dz_scratch(ctx, "shard.recv_sen", per * 8 * (size_t)c->nranks, &g))) return rc;
  if ((rc = dz_allgather(ctx, c, P.send, g, per * 8))) return rc;
  hipLaunchKernelGGL(k_join_blocks<double>, dim3(grid_for(3 * na * (size_t)P.nx * P.ny)), dim3(256), 0, ctx->stream, (const double *)g, per,
                     c->nranks, P.nx, P.ny, na, cbmax, 3, P.svs, P.svp, P.srho);
  DZ_HIP(hipGetLastError());
  return 0;
}

int dazim_dispersion_kernels_sharded(dazim_ctx *ctx, int nx, int ny, int nz, const float *vel_u, const float *depz, float minthk0,
                                     int kmax, const double *periods, double *pv_u, double *svs_u, double *svp_u, double *srho_u,
                                     int *n_failed) {
  if (!ctx) return DAZIM_E_BAD_ARG;
  DzComm *c = (DzComm *)ctx->comm;
  if (!c || c->nranks <= 1)
    return dazim_dispersion_kernels(ctx, nx, ny, nz, vel_u, depz, minthk0, kmax, periods, pv_u, svs_u, svp_u, srho_u, n_failed);
  if (!vel_u || !depz || !periods || !pv_u || nx < 1 || ny < 1 || nz < 2 || kmax < 1) return dz_fail(ctx, DAZIM_E_BAD_ARG, "null argument");
  DZ_HIP(hipSetDevice(ctx->device));
  int rc;
  if ((rc = dz_join_aux(ctx))) return rc;   // (an earlier call's blocks may still wait for their gather: the send buffer is reused below)
  const bool kernels = svs_u && svp_u && srho_u;
  const size_t ncol = (size_t)nx * ny;
  int64_t lo, hi, lo0, hi0;
  dz_shard_even(ny, c->nranks, c->rank, &lo, &hi);
  dz_shard_even(ny, c->nranks, 0, &lo0, &hi0);
  const int nyb = (int)(hi - lo);
  const size_t cb = (size_t)nyb * nx, cbmax = (size_t)(hi0 - lo0) * nx, na = (size_t)nz * kmax;
  const size_t per_pv = (size_t)kmax * cbmax, per_sen = 3 * na * cbmax;
  DzBuf<float> vel;
  DzBuf<double> pv, svs, svp, srho;
  if ((rc = vel.init(ctx, vel_u, (size_t)nz * ncol, true, false))) return rc;
  if ((rc = pv.init(ctx, pv_u, (size_t)kmax * ncol, false, true))) return rc;
  if (kernels) {
    if ((rc = svs.init(ctx, svs_u, na * ncol, false, true))) return rc;
    if ((rc = svp.init(ctx, svp_u, na * ncol, false, true))) return rc;
    if ((rc = srho.init(ctx, srho_u, na * ncol, false, true))) return rc;
  }
  void *p;
  if ((rc = dz_scratch(ctx, "shard.vel", (size_t)nz * cbmax * 4 + 16, &p))) return rc;
  float *vel_b = (float *)p;
  if ((rc = dz_scratch(ctx, "shard.send", (per_pv + (kernels ? per_sen : 0)) * 8 + 16, &p))) return rc;
  double *send = (double *)p;
  if ((rc = dz_scratch(ctx, "shard.recv_pv", per_pv * 8 * (size_t)c->nranks, &p))) return rc;
  double *recv_pv = (double *)p;
  if ((rc = dz_scratch(ctx, "shard.nfail", 16, &p))) return rc;
  long long *d_nf = (long long *)p;
  DZ_HIP(hipMemsetAsync(send, 0, (per_pv + (kernels ? per_sen : 0)) * 8, ctx->stream));   // (the padding of a shorter block travels too)
  int nf = 0;
  if (nyb > 0) {
    hipLaunchKernelGGL(k_cols_copy<float>, dim3(grid_for((size_t)nz * cb)), dim3(256), 0, ctx->stream, vel.dev, ncol, vel_b, cb, (size_t)lo * nx,
                       (size_t)nz, true);
    DZ_HIP(hipGetLastError());
    DZ_HIP(hipStreamSynchronize(ctx->stream));
    double *sen_b = send + per_pv;
    rc = dazim_dispersion_kernels(ctx, nx, nyb, nz, vel_b, depz, minthk0, kmax, periods, send, kernels ? sen_b : nullptr,
                                  kernels ? sen_b + na * cbmax : nullptr, kernels ? sen_b + 2 * na * cbmax : nullptr, &nf);
  }
  {   // a failure on this rank's block is every rank's: agree before anybody waits in the gather for a rank that has returned
    long long flag = rc ? 1 : 0;
    int rc2;
    DZ_HIP(hipMemcpyAsync(d_nf, &flag, 8, hipMemcpyHostToDevice, ctx->stream));
    if ((rc2 = dz_allreduce(ctx, c, d_nf, 1, DZ_I64, DZ_MAX))) return rc2;
    DZ_HIP(hipMemcpyAsync(&flag, d_nf, 8, hipMemcpyDeviceToHost, ctx->stream));
    DZ_HIP(hipStreamSynchronize(ctx->stream));
    if (rc) return rc;
    if (flag) return dz_fail(ctx, DAZIM_E_BAD_ARG, "dazim_dispersion_kernels failed on another rank's block of the model");
  }
  // the curves: now, on the main stream
  if ((rc = dz_allgather(ctx, c, send, recv_pv, per_pv * 8))) return rc;
  hipLaunchKernelGGL(k_join_blocks<double>, dim3(grid_for((size_t)kmax * ncol)), dim3(256), 0, ctx->stream, (const double *)recv_pv, per_pv,
                     c->nranks, nx, ny, (size_t)kmax, cbmax, 1, pv.dev, (double *)nullptr, (double *)nullptr);
  DZ_HIP(hipGetLastError());
  long long hnf = nf;
  DZ_HIP(hipMemcpyAsync(d_nf, &hnf, 8, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = dz_allreduce(ctx, c, d_nf, 1, DZ_I64, DZ_SUM))) return rc;
  DZ_HIP(hipMemcpyAsync(&hnf, d_nf, 8, hipMemcpyDeviceToHost, ctx->stream));
  if ((rc = pv.finish())) return rc;
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  if (n_failed) *n_failed = (int)hnf;
  if (!kernels) return 0;
  // the depth kernels: when the auxiliary stream is joined -- by every rank alike, whether or not its own block's copies went
  // there (a rank without rows, a block the library declined to run on two streams), so that all ranks issue the gather at the
  // same point of the program
  ctx->shard.nx = nx; ctx->shard.ny = ny; ctx->shard.nz = nz; ctx->shard.kmax = kmax;
  ctx->shard.send = send + per_pv;
  ctx->shard.svs = svs.dev; ctx->shard.svp = svp.dev; ctx->shard.srho = srho.dev;
  // ... unless an asynchronous eikonal call is going to follow (option fmm.async) over RCCL with blocks small enough (>= 4 ranks)
  // that waiting for this rank's copies costs less than the ray kernel beside the eikonal tail brings: an RCCL communicator is
  // kept to the main stream, where a deferred gather would sit behind the whole eikonal launch and keep the ray call from
  // starting beside it (rays.hip).  The same decision on every rank: options and rank count are the ranks' own.
  const bool gather_now = (c->nccl && c->nranks >= 4 && ctx->opts.count("fmm.async") && ctx->opts["fmm.async"] &&
                           !(ctx->opts.count("comm.gather_stream3") && ctx->opts["comm.gather_stream3"])) ||
                          (ctx->opts.count("comm.gather_now") && ctx->opts["comm.gather_now"]);   // (test knob: the same path over files)
  const bool defer = !gather_now && !svs.staged && !svp.staged && !srho.staged && ctx->opts.count("disp.async") && ctx->opts["disp.async"];
  if (defer) {
    if (!ctx->aux_pending) {   // nothing of this rank on the auxiliary stream: an event that is already complete
      if ((rc = dz_aux_init(ctx))) return rc;
      DZ_HIP(hipEventRecord(ctx->ev_a1, ctx->stream2));
      ctx->aux_pending = true;
      ctx->aux_ranges.clear();
    }
    for (const double *q : {svs.dev, svp.dev, srho.dev}) ctx->aux_ranges.push_back({(const char *)q, na * ncol * sizeof(double)});
    ctx->aux_epilogue = shard_join_sen;
    return 0;
  }
  if ((rc = dz_join_aux(ctx))) return rc;
  if ((rc = shard_join_sen(ctx))) return rc;
  if ((rc = svs.finish()) || (rc = svp.finish()) || (rc = srho.finish())) return rc;
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

// depthkernelTI / tregn96 (dazim_ti_kernels) on this rank's block of model rows, Lsen_Gsc joined by one all-gather
int dazim_ti_kernels_sharded(dazim_ctx *ctx, int nx, int ny, int nz, const float *vel_u, const float *depz, float minthk0, int kmax,
                             const double *periods, const double *pv_u, float *lsen_u) {
  if (!ctx) return DAZIM_E_BAD_ARG;
  DzComm *c = (DzComm *)ctx->comm;
  if (!c || c->nranks <= 1) return dazim_ti_kernels(ctx, nx, ny, nz, vel_u, depz, minthk0, kmax, periods, pv_u, lsen_u);
  if (!vel_u || !depz || !periods || !pv_u || !lsen_u || nx < 1 || ny < 1 || nz < 2 || kmax < 1) return dz_fail(ctx, DAZIM_E_BAD_ARG, "null argument");
  DZ_HIP(hipSetDevice(ctx->device));
  int rc;
  const size_t ncol = (size_t)nx * ny;
  int64_t lo, hi, lo0, hi0;
  dz_shard_even(ny, c->nranks, c->rank, &lo, &hi);
  dz_shard_even(ny, c->nranks, 0, &lo0, &hi0);
  const int nyb = (int)(hi - lo);
  const size_t cb = (size_t)nyb * nx, cbmax = (size_t)(hi0 - lo0) * nx, na = (size_t)(nz - 1) * kmax;
  DzBuf<float> vel, lsen;
  DzBuf<double> pv;
  if ((rc = vel.init(ctx, vel_u, (size_t)nz * ncol, true, false))) return rc;
  if ((rc = pv.init(ctx, pv_u, (size_t)kmax * ncol, true, false))) return rc;
  if ((rc = lsen.init(ctx, lsen_u, na * ncol, false, true))) return rc;
  void *p;
  if ((rc = dz_scratch(ctx, "shard.ti_vel", (size_t)nz * cbmax * 4 + 16, &p))) return rc;
  float *vel_b = (float *)p;
  if ((rc = dz_scratch(ctx, "shard.ti_pv", (size_t)kmax * cbmax * 8 + 16, &p))) return rc;
  double *pv_b = (double *)p;
  if ((rc = dz_scratch(ctx, "shard.ti_send", na * cbmax * 4 + 16, &p))) return rc;
  float *send = (float *)p;
  if ((rc = dz_scratch(ctx, "shard.ti_recv", na * cbmax * 4 * (size_t)c->nranks, &p))) return rc;
  float *recv = (float *)p;
  DZ_HIP(hipMemsetAsync(send, 0, na * cbmax * 4, ctx->stream));
  if (nyb > 0) {
    hipLaunchKernelGGL(k_cols_copy<float>, dim3(grid_for((size_t)nz * cb)), dim3(256), 0, ctx->stream, vel.dev, ncol, vel_b, cb, (size_t)lo * nx,
                       (size_t)nz, true);
    hipLaunchKernelGGL(k_cols_copy<double>, dim3(grid_for((size_t)kmax * cb)), dim3(256), 0, ctx->stream, pv.dev, ncol, pv_b, cb, (size_t)lo * nx,
                       (size_t)kmax, true);
    DZ_HIP(hipGetLastError());
    DZ_HIP(hipStreamSynchronize(ctx->stream));
    rc = dazim_ti_kernels(ctx, nx, nyb, nz, vel_b, depz, minthk0, kmax, periods, pv_b, send);
  }
  // a failure of this rank's block (a fluid layer ...) is every rank's: agree before anybody returns
  long long flag = rc ? 1 : 0;
  void *pf;
  int rc2;
  if ((rc2 = dz_scratch(ctx, "shard.nfail", 16, &pf))) return rc2;
  DZ_HIP(hipMemcpyAsync(pf, &flag, 8, hipMemcpyHostToDevice, ctx->stream));
  if ((rc2 = dz_allreduce(ctx, c, pf, 1, DZ_I64, DZ_MAX))) return rc2;
  DZ_HIP(hipMemcpyAsync(&flag, pf, 8, hipMemcpyDeviceToHost, ctx->stream));
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  if (rc) return rc;
  if (flag) return dz_fail(ctx, DAZIM_E_BAD_ARG, "dazim_ti_kernels failed on another rank's block of the model");
  if ((rc = dz_allgather(ctx, c, send, recv, na * cbmax * 4))) return rc;
  hipLaunchKernelGGL(k_join_blocks<float>, dim3(grid_for(na * ncol)), dim3(256), 0, ctx->stream, (const float *)recv, na * cbmax, c->nranks, nx, ny,
                     na, cbmax, 1, lsen.dev, (float *)nullptr, (float *)nullptr);
  DZ_HIP(hipGetLastError());
  if ((rc = lsen.finish())) return rc;
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

}  // extern "C"
