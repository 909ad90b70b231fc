// ctx.hip -- context, memory helpers and host-side geometry of libdazim_hip.so.
#include <cstring>

#include "dazim_internal.h"

int dz_fail(dazim_ctx *c, int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  return code;
}

bool dz_is_device_ptr(const void *p) {
  hipPointerAttribute_t a;
  memset(&a, 0, sizeof a);
  hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // plain malloc'ed host memory: clear the sticky error
    return false;
  }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// Free every cached block nobody is using (staging blocks and matrix arrays).  Returns the bytes released.
size_t dz_trim_caches(dazim_ctx *ctx) {
  if (!ctx) return 0;
  size_t freed = 0;
  (void)hipStreamSynchronize(ctx->stream);
  for (size_t i = ctx->stage.size(); i-- > 0;)
    if (!ctx->stage[i].busy) { freed += ctx->stage[i].bytes; (void)hipFree(ctx->stage[i].p); ctx->stage.erase(ctx->stage.begin() + i); }
  for (size_t i = ctx->big.size(); i-- > 0;)
    if (!ctx->big[i].busy) { freed += ctx->big[i].bytes; (void)hipFree(ctx->big[i].p); ctx->big.erase(ctx->big.begin() + i); }
  // ... and the multi-GB per-field state of a time-sliced eikonal batch (S-512: 34 GB), which is kept between calls because
  // giving it back and asking for it again costs 0.7 s per step -- unless an eikonal call is the one that ran out of memory
  if (!ctx->fmm_busy)
    for (const char *nm : {"fmm.rec_c", "fmm.ts_keys", "fmm.ts_nodes", "fmm.ovf", "fmm.ttn_tiled"}) {
      if (ctx->fields.tiled && (std::string(nm) == "fmm.rec_c" || std::string(nm) == "fmm.ttn_tiled")) continue;   // (fields waiting for the ray kernel)
      auto it = ctx->scratch.find(nm);
      if (it != ctx->scratch.end() && it->second.first && it->second.second >= ((size_t)1 << 30)) {
        freed += it->second.second;
        (void)hipFree(it->second.first);
        ctx->scratch.erase(it);
      }
    }
  return freed;
}

// hipMalloc that gives the caches' idle memory back to the device before it reports out-of-memory: up to 24 GiB of freed
// matrix arrays and 2 GiB of staging blocks may be parked there, which a failing allocation must be able to use.
hipError_t dz_malloc_retry(dazim_ctx *ctx, void **p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipErrorOutOfMemory && dz_trim_caches(ctx) > 0) {
    (void)hipGetLastError();
    e = hipMalloc(p, bytes);
  }
  return e;
}

int dz_scratch(dazim_ctx *ctx, const char *name, size_t bytes, void **out) {
  auto &s = ctx->scratch[name];
  if (s.second < bytes) {
    if (s.first) {
      DZ_HIP(hipStreamSynchronize(ctx->stream));
      DZ_HIP(hipFree(s.first));
    }
    s.first = nullptr;
    s.second = 0;
    // a little slack so slowly growing batches do not thrash; none on the multi-GB blocks (the per-field state of time-sliced
    // eikonal marches: an eighth of 34 GB would be 4 GB that nothing ever touches)
    // ... but rounded up to the next 256 MB, so that a batch that grows by one field does not free and allocate gigabytes each time
    size_t want = bytes >= ((size_t)1 << 30) ? ((bytes + ((size_t)1 << 28) - 1) >> 28) << 28 : bytes + bytes / 8;
    DZ_HIP(dz_malloc_retry(ctx, &s.first, want));
    s.second = want;
  }
  *out = s.first;
  return 0;
}

int dz_pinned(dazim_ctx *ctx, const char *name, size_t bytes, void **out) {
  auto &s = ctx->pinned[name];
  if (s.second < bytes) {
    const bool pin = !(ctx->opts.count("ctx.pinned") && !ctx->opts["ctx.pinned"]);
    if (s.first) {
      DZ_HIP(hipStreamSynchronize(ctx->stream));
      if (hipHostFree(s.first) != hipSuccess) { (void)hipGetLastError(); free(s.first); }
    }
    s.first = nullptr;
    s.second = 0;
    const size_t want = bytes + bytes / 4 + 64;
    if (pin) DZ_HIP(hipHostMalloc(&s.first, want));
    else if (!(s.first = malloc(want))) return dz_fail(ctx, -3, "out of host memory");
    s.second = want;
  }
  *out = s.first;
  return 0;
}

// Staging cache: best fit among the free blocks, else a new allocation; at most 64 blocks / 2 GiB are kept, larger or surplus
// blocks are released on return.
int dz_stage_get(dazim_ctx *ctx, size_t bytes, void **out) {
  if (bytes == 0) bytes = 16;
  int best = -1;
  for (size_t i = 0; i < ctx->stage.size(); i++) {
    auto &b = ctx->stage[i];
    if (!b.busy && b.bytes >= bytes && (best < 0 || b.bytes < ctx->stage[best].bytes)) best = (int)i;
  }
  if (best >= 0 && ctx->stage[best].bytes <= 4 * bytes + (1 << 20)) {
    ctx->stage[best].busy = true;
    *out = ctx->stage[best].p;
    return 0;
  }
  void *p = nullptr;
  DZ_HIP(dz_malloc_retry(ctx, &p, bytes));
  ctx->stage.push_back({p, bytes, true});
  *out = p;
  return 0;
}
void dz_stage_put(dazim_ctx *ctx, void *p) {
  size_t kept = 0, nfree = 0;
  for (auto &b : ctx->stage)
    if (!b.busy) { kept += b.bytes; nfree++; }
  for (size_t i = 0; i < ctx->stage.size(); i++) {
    auto &b = ctx->stage[i];
    if (b.p != p) continue;
    if (nfree >= 64 || kept + b.bytes > ((size_t)2 << 30)) {
      (void)hipFree(b.p);
      ctx->stage.erase(ctx->stage.begin() + i);
    } else {
      b.busy = false;
    }
    return;
  }
  (void)hipFree(p);   // not from the cache
}

// Matrix-array cache: best fit among the free blocks if it is not more than half again as large, else a new allocation; at
// most 12 free blocks / 24 GiB are kept.
int dz_big_get(dazim_ctx *ctx, size_t bytes, void **out) {
  if (bytes == 0) bytes = 16;
  int best = -1;
  for (size_t i = 0; i < ctx->big.size(); i++) {
    auto &b = ctx->big[i];
    if (!b.busy && b.bytes >= bytes && (best < 0 || b.bytes < ctx->big[best].bytes)) best = (int)i;
  }
  if (best >= 0 && ctx->big[best].bytes <= bytes + bytes / 2 + (1 << 20)) {
    ctx->big[best].busy = true;
    *out = ctx->big[best].p;
    return 0;
  }
  void *p = nullptr;
  DZ_HIP(dz_malloc_retry(ctx, &p, bytes));
  ctx->big.push_back({p, bytes, true});
  *out = p;
  return 0;
}
void dz_big_put(dazim_ctx *ctx, void *p) {
  if (!p) return;
  if (ctx) {
    size_t kept = 0, nfree = 0;
    for (auto &b : ctx->big)
      if (!b.busy) { kept += b.bytes; nfree++; }
    for (size_t i = 0; i < ctx->big.size(); i++) {
      auto &b = ctx->big[i];
      if (b.p != p) continue;
      if (nfree >= 12 || kept + b.bytes > ((size_t)24 << 30)) {
        (void)hipFree(b.p);
        ctx->big.erase(ctx->big.begin() + i);
      } else {
        b.busy = false;
      }
      return;
    }
  }
  (void)hipFree(p);   // not from the cache (or no context left)
}

namespace {
__global__ void k_range_flag(int64_t n, const int *a, int lo, int hi, int *bad) {
  bool b = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) b |= a[i] < lo || a[i] > hi;
  if (b) atomicOr(bad, 1);
}
}  // namespace

// Integer index arrays that cross the ABI (periods, kernel slots, field-of-ray, COO rows / columns) are validated on the device
// before any kernel dereferences them: an index outside its range would otherwise read another period's grid or fault.
int dz_check_range(dazim_ctx *ctx, const int *a_dev, int64_t n, int lo, int hi, const char *what) {
  if (n <= 0) return 0;
  if (!a_dev) return dz_fail(ctx, DAZIM_E_BAD_ARG, "%s is NULL", what);
  void *p;
  int rc;
  if ((rc = dz_scratch(ctx, "ctx.rangeflag", 16, &p))) return rc;
  int *bad = (int *)p, hbad = 0;
  DZ_HIP(hipMemsetAsync(bad, 0, 4, ctx->stream));
  int64_t nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(k_range_flag, dim3((unsigned)nb), dim3(256), 0, ctx->stream, n, a_dev, lo, hi, bad);
  DZ_HIP(hipGetLastError());
  DZ_HIP(hipMemcpyAsync(&hbad, bad, 4, hipMemcpyDeviceToHost, ctx->stream));
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  if (hbad) return dz_fail(ctx, DAZIM_E_BAD_ARG, "%s outside %d..%d", what, lo, hi);
  return 0;
}

int dz_aux_init(dazim_ctx *ctx) {
  if (ctx->stream2) return 0;
  DZ_HIP(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
  DZ_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
  DZ_HIP(hipEventCreate(&ctx->ev_a0));
  DZ_HIP(hipEventCreate(&ctx->ev_a1));
  return 0;
}

int dz_async_init(dazim_ctx *ctx) {
  if (ctx->stream3) return 0;
  DZ_HIP(hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking));
  DZ_HIP(hipEventCreate(&ctx->ev_f0));
  DZ_HIP(hipEventCreate(&ctx->ev_f1));
  DZ_HIP(hipEventCreateWithFlags(&ctx->ev_pre, hipEventDisableTiming));
  DZ_HIP(hipEventCreate(&ctx->ev_r0));
  DZ_HIP(hipEventCreate(&ctx->ev_r1));
  DZ_HIP(hipHostMalloc((void **)&ctx->hprog, 64, hipHostMallocMapped));
  memset(ctx->hprog, 0, 64);
  return 0;
}
int dz_fmm_finish(dazim_ctx *ctx) {
  if (!ctx->fmm_finish) return 0;
  auto f = std::move(ctx->fmm_finish);
  ctx->fmm_finish = nullptr;
  return f();
}

// everything the main stream enqueues from here on comes after what the auxiliary stream was given (device-side wait only)
int dz_join_aux(dazim_ctx *ctx) {
  if (!ctx->aux_pending) return 0;
  DZ_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_a1, 0));
  ctx->aux_pending = false;
  ctx->aux_ranges.clear();
  if (ctx->aux_epilogue) {   // (sharded dispersion tables: the all-gather that follows the perturbed copies, on the main stream)
    auto f = ctx->aux_epilogue;
    ctx->aux_epilogue = nullptr;
    return f(ctx);
  }
  return 0;
}
int dz_join_aux_if_touched(dazim_ctx *ctx, const void *dev, size_t bytes) {
  if (!ctx->aux_pending) return 0;
  const char *a = (const char *)dev;
  bool hit = ctx->aux_ranges.empty();   // (no ranges recorded: assume the worst)
  for (const auto &r : ctx->aux_ranges)
    if (a < r.p + r.bytes && r.p < a + bytes) hit = true;
  return hit ? dz_join_aux(ctx) : 0;
}

extern "C" {

int dazim_create(dazim_ctx **out, int device) {
  if (!out) return DAZIM_E_BAD_ARG;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return -1;  // no GPU: the product path refuses to run
  if (device < 0 || device >= n) return DAZIM_E_BAD_ARG;
  dazim_ctx *ctx = new dazim_ctx;
  ctx->device = device;
  // Experiments only (tools/exp_cumask.sh): DAZIM_CU_MASK=<hex words, least significant first, comma separated> restricts the
  // library's stream to a subset of the compute units -- "does a kernel's throughput follow the CUs it may use or the memory
  // system behind them?"
  bool masked = false;
  if (const char *mk = getenv("DAZIM_CU_MASK")) {
    std::vector<uint32_t> words;
    for (const char *q = mk; *q;) {
      words.push_back((uint32_t)strtoul(q, nullptr, 16));
      const char *c = strchr(q, ',');
      if (!c) break;
      q = c + 1;
    }
    if (hipSetDevice(device) == hipSuccess && !words.empty() &&
        hipExtStreamCreateWithCUMask(&ctx->stream, (uint32_t)words.size(), words.data()) == hipSuccess)
      masked = true;
  }
  if (hipSetDevice(device) != hipSuccess || (!masked && hipStreamCreate(&ctx->stream) != hipSuccess) ||
      hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
    delete ctx;
    return -2;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cu = prop.multiProcessorCount;
  // DAZIM_OPTS=name=value,name=value: tuning options (include/dazim.h, dazim_set_option) for callers that do not set them
  // themselves -- the Fortran programs under tools/run_test4_program.sh; speed only, like the options
  if (const char *ev = getenv("DAZIM_OPTS")) {
    std::string str(ev);
    size_t pos = 0;
    while (pos < str.size()) {
      size_t end = str.find(',', pos);
      if (end == std::string::npos) end = str.size();
      const std::string kv = str.substr(pos, end - pos);
      const size_t eq = kv.find('=');
      if (eq != std::string::npos && eq > 0) ctx->opts[kv.substr(0, eq)] = atoi(kv.c_str() + eq + 1);
      pos = end + 1;
    }
  }
  *out = ctx;
  return 0;
}

void dazim_destroy(dazim_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  ctx->aux_epilogue = nullptr;   // (no collective on the way out)
  if (ctx->stream3) (void)hipStreamSynchronize(ctx->stream3);
  (void)hipStreamSynchronize(ctx->stream);
  (void)dz_fmm_finish(ctx);
  if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->comm && ctx->comm_release) ctx->comm_release(ctx);
  for (auto &b : ctx->stage) (void)hipFree(b.p);
  ctx->stage.clear();
  for (auto &b : ctx->big)
    if (!b.busy) (void)hipFree(b.p);   // (arrays of matrices that outlive the context stay theirs: dazim_csr_free(NULL, A) frees them)
  ctx->big.clear();
  for (auto &kv : ctx->scratch)
    if (kv.second.first) (void)hipFree(kv.second.first);
  for (auto &kv : ctx->pinned)
    if (kv.second.first && hipHostFree(kv.second.first) != hipSuccess) { (void)hipGetLastError(); free(kv.second.first); }
  (void)hipEventDestroy(ctx->ev0);
  (void)hipEventDestroy(ctx->ev1);
  if (ctx->stream3) {
    for (hipEvent_t e : {ctx->ev_f0, ctx->ev_f1, ctx->ev_pre, ctx->ev_r0, ctx->ev_r1}) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(ctx->stream3);
    if (ctx->hprog) (void)hipHostFree(ctx->hprog);
  }
  if (ctx->stream2) {
    (void)hipEventDestroy(ctx->ev_fork);
    (void)hipEventDestroy(ctx->ev_a0);
    (void)hipEventDestroy(ctx->ev_a1);
    (void)hipStreamDestroy(ctx->stream2);
  }
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char *dazim_last_error(const dazim_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int dazim_malloc(dazim_ctx *ctx, void **dptr, size_t bytes) {
  DZ_HIP(hipSetDevice(ctx->device));
  DZ_HIP(dz_malloc_retry(ctx, dptr, bytes));
  return 0;
}
int dazim_free(dazim_ctx *ctx, void *dptr) {
  { int rcf = dz_fmm_finish(ctx); if (rcf) return rcf; }
  if (ctx->aux_epilogue) { int rcj = dz_join_aux(ctx); if (rcj) return rcj; }
  if (ctx->stream2) DZ_HIP(hipStreamSynchronize(ctx->stream2));
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  DZ_HIP(hipFree(dptr));
  return 0;
}
// (both copies come after whatever the auxiliary stream still has to do -- the depth kernels of a disp.async call: a caller that
// reads sen_* straight after dazim_dispersion_kernels gets the complete arrays, one that overwrites vel does not race the copies)
// (a copy into or out of the arrays the perturbed copies of an asynchronous dazim_dispersion_kernels call still work on waits for
// them; any other copy leaves the auxiliary stream alone, so that staging the next inputs does not cost the overlap)
int dazim_memcpy_h2d(dazim_ctx *ctx, void *dst, const void *src, size_t bytes) {
  { int rcf = dz_fmm_finish(ctx); if (rcf) return rcf; }
  int rcj = dz_join_aux_if_touched(ctx, dst, bytes);
  if (rcj) return rcj;
  DZ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}
int dazim_memcpy_d2h(dazim_ctx *ctx, void *dst, const void *src, size_t bytes) {
  { int rcf = dz_fmm_finish(ctx); if (rcf) return rcf; }
  int rcj = dz_join_aux_if_touched(ctx, src, bytes);
  if (rcj) return rcj;
  DZ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}
int dazim_sync(dazim_ctx *ctx) {
  { int rcf = dz_fmm_finish(ctx); if (rcf) return rcf; }
  if (ctx->aux_epilogue) { int rcj = dz_join_aux(ctx); if (rcj) return rcj; }
  if (ctx->stream2) DZ_HIP(hipStreamSynchronize(ctx->stream2));
  ctx->aux_pending = false;
  ctx->aux_ranges.clear();
  DZ_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}
void *dazim_stream(dazim_ctx *ctx) { return (void *)ctx->stream; }

// 64-bit hash of a host array (every byte; four independent multiply-xorshift lanes so that one core streams it at memory
// speed).  The Fortran aprod drop-in keys its cached device matrix on it: the reference's aprod uses iw / rw as they are at call
// time, so an in-place edit anywhere in them must be seen.
unsigned long long dazim_hash64(const void *data, size_t bytes) {
  const unsigned char *p = (const unsigned char *)data;
  const unsigned long long K = 0x9E3779B97F4A7C15ull;
  unsigned long long h[4] = {0x243F6A8885A308D3ull, 0x13198A2E03707344ull, 0xA4093822299F31D0ull, 0x082EFA98EC4E6C89ull};
  size_t i = 0;
  for (; i + 32 <= bytes; i += 32) {
    unsigned long long w[4];
    memcpy(w, p + i, 32);
    for (int l = 0; l < 4; l++) {
      h[l] = (h[l] ^ w[l]) * K;
      h[l] ^= h[l] >> 29;
    }
  }
  unsigned long long t = bytes;
  for (; i < bytes; i++) t = (t ^ p[i]) * K;
  unsigned long long r = t;
  for (int l = 0; l < 4; l++) {
    r = (r ^ h[l]) * K;
    r ^= r >> 32;
  }
  return r;
}

int dazim_set_option(dazim_ctx *ctx, const char *name, int value) {
  if (!ctx || !name) return DAZIM_E_BAD_ARG;
  ctx->opts[name] = value;
  return 0;
}

double dazim_last_kernel_seconds(const dazim_ctx *ctx, const char *name) {
  if (ctx->aux_timed && std::string(name) == "disp.copies") {   // the auxiliary stream's part of the last dispersion call (waits for it)
    float ms = -1.0f;
    if (hipEventSynchronize(ctx->ev_a1) == hipSuccess && hipEventElapsedTime(&ms, ctx->ev_a0, ctx->ev_a1) == hipSuccess) return ms * 1e-3;
    return -1.0;
  }
  auto it = ctx->ksec.find(name);
  return it == ctx->ksec.end() ? -1.0 : it->second;
}
int dazim_get_stat(const dazim_ctx *ctx, const char *name, double *value) {
  if (!ctx || !name || !value) return DAZIM_E_BAD_ARG;
  if (std::string(name) == "disp.copies") {
    const double v = dazim_last_kernel_seconds(ctx, name);
    if (v < 0) return DAZIM_E_BAD_ARG;
    *value = v;
    return 0;
  }
  if (std::string(name) == "aux.pending") {   // 1 while work handed to the auxiliary stream has not been joined by the main stream
    *value = ctx->aux_pending ? 1.0 : 0.0;
    return 0;
  }
  if (std::string(name) == "comm.nranks" || std::string(name) == "comm.rank") {   // the attached communicator (1, 0 without one)
    *value = name[5] == 'n' ? ctx->nranks : ctx->rank;
    return 0;
  }
  auto it = ctx->ksec.find(name);
  if (it == ctx->ksec.end()) return DAZIM_E_BAD_ARG;
  *value = it->second;
  return 0;
}

// inv/CalSurfG.f90:1005-1038 (gdx = gdz = 5, fp32 pi = 3.1415926535898 as in MODULE globalp :166)
int dazim_geometry(int nx, int ny, float goxd, float gozd, float dvxd, float dvzd, dazim_geom *g) {
  if (!g || nx < 4 || ny < 4) return DAZIM_E_BAD_ARG;
  const float pi = 3.1415926535898f;
  g->nvx = nx - 2;
  g->nvz = ny - 2;
  g->dvx = dvxd * pi / 180.0f;
  g->dvz = dvzd * pi / 180.0f;
  g->gox = (90.0f - goxd) * pi / 180.0f;
  g->goz = gozd * pi / 180.0f;
  g->nnx = (g->nvx - 1) * 5 + 1;
  g->nnz = (g->nvz - 1) * 5 + 1;
  g->dnx = g->dvx / 5;
  g->dnz = g->dvz / 5;
  return 0;
}

}  // extern "C"
