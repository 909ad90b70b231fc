// valu_issue_calib.hip -- how many wave64 VALU instructions a gfx950 SIMD issues per cycle.
//
// bench.py prices the eikonal kernel against the VALU issue rate, and the two sources disagree about it:
// /opt/skills/guides/MI355X_MICROARCH.md says "SIMD-32, each VALU instruction over 2 cycles" (1 228.8 G wave-instructions/s on
// 256 CUs x 4 SIMDs x 2.4 GHz), the SQ counters of the kernel itself say SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.005 quad-cycles
// (4 cycles, 614.4 G/s).  This program measures it: streams of INDEPENDENT instructions (16 accumulators, no dependence closer
// than 16 instructions) of one opcode, W wavefronts per SIMD (one workgroup of 256*W threads per CU, 80 KB of LDS so that a CU
// holds exactly one), timed three ways -- s_memtime ticks inside the wavefront, HIP events around the launch, and (under
// rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE, a separate pass: tools/valu_calib.sh) the SQ counters.
//
//   hipcc -O2 --offload-arch=gfx950 tools/valu_issue_calib.hip -o gpurun_out/valu_issue_calib && gpurun_out/valu_issue_calib
//
// Output: one markdown row per (opcode, W): cycles per wave64 instruction per SIMD by the shader clock, G wave-instructions/s of
// the whole chip by wall time, and the clock the chip ran at (ticks / wall time; s_memtime counts at a constant 100 MHz on
// gfx950, so the shader clock is taken from GRBM_GUI_ACTIVE in the counter pass instead -- the program prints both).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int UNROLL = 8;     // 16 instructions x 8 per loop iteration
constexpr int ITERS = 4096;   // 16 * 8 * 4096 = 524 288 instructions of the opcode per wavefront

// sixteen independent accumulators a0..a15; OP is the instruction text with %0 = accumulator, %16 / %17 = loop-invariant operands
#define STREAM16(OP)                                                                                                      \
  asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)     \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),          \
                 "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])     \
               : "v"(x), "v"(y) : "vcc", "scc", "s20", "s21", "s22", "s23", "memory")

#define OP_ADD_F32(i) "v_add_f32 %" #i ", %" #i ", %16\n"
#define OP_MUL_F32(i) "v_mul_f32 %" #i ", %" #i ", %16\n"
#define OP_FMA_F32(i) "v_fma_f32 %" #i ", %" #i ", %16, %17\n"
#define OP_ADD_U32(i) "v_add_u32 %" #i ", %" #i ", %16\n"
#define OP_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %16, vcc\n"
#define OP_MOV_DPP(i) "v_mov_b32_dpp %" #i ", %16 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define OP_RCP_F32(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define OP_SQRT_F32(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define OP_PK_ADD_F32(i) "v_pk_add_f32 %" #i ", %" #i ", %16\n"
#define OP_PK_MUL_F32(i) "v_pk_mul_f32 %" #i ", %" #i ", %16\n"
#define OP_PK_FMA_F32(i) "v_pk_fma_f32 %" #i ", %" #i ", %16, %17\n"
#define OP_ADD_F64(i) "v_add_f64 %" #i ", %" #i ", %16\n"
#define OP_MUL_F64(i) "v_mul_f64 %" #i ", %" #i ", %16\n"
#define OP_FMA_F64(i) "v_fma_f64 %" #i ", %" #i ", %16, %17\n"
#define OP_RCP_F64(i) "v_rcp_f64 %" #i ", %" #i "\n"
// second table (round 5): encodings and operand counts -- what makes an instruction a 2-cycle or a 4-cycle one
#define OP_ADD_F32_E64(i) "v_add_f32_e64 %" #i ", %" #i ", %16\n"
#define OP_FMAC_F32(i) "v_fmac_f32 %" #i ", %16, %17\n"
#define OP_FMA_F32_2SRC(i) "v_fma_f32 %" #i ", %" #i ", %16, %16\n"
#define OP_FMA_F32_K(i) "v_fma_f32 %" #i ", %" #i ", %16, 1.0\n"
#define OP_CNDMASK_E64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %16, s[20:21]\n"
#define OP_CNDMASK_ALT(i) "v_cndmask_b32 %" #i ", %16, %17, vcc\n"
#define OP_MOV(i) "v_mov_b32 %" #i ", %16\n"
#define OP_AND(i) "v_and_b32 %" #i ", %" #i ", %16\n"
#define OP_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define OP_LSHL_ADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %16\n"
#define OP_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %16, %17\n"
#define OP_AND_OR(i) "v_and_or_b32 %" #i ", %" #i ", %16, %17\n"
#define OP_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 2, 5\n"
#define OP_BFI(i) "v_bfi_b32 %" #i ", %16, %17, %" #i "\n"
#define OP_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %16, %17\n"
#define OP_MUL_LO(i) "v_mul_lo_u32 %" #i ", %" #i ", %16\n"
#define OP_MIN_F32(i) "v_min_f32 %" #i ", %" #i ", %16\n"
#define OP_MAX3_F32(i) "v_max3_f32 %" #i ", %" #i ", %16, %17\n"
#define OP_CMP_VCC(i) "v_cmp_lt_f32 vcc, %" #i ", %16\n"
#define OP_CMP_SGPR(i) "v_cmp_lt_f32_e64 s[20:21], %" #i ", %16\n"
#define OP_CMP_CND(i) "v_cmp_lt_f32 vcc, %" #i ", %16\nv_cndmask_b32 %" #i ", %" #i ", %17, vcc\n"
#define OP_MIN_DPP(i) "v_min_f32_dpp %" #i ", %" #i ", %16 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define OP_MOV_BCAST(i) "v_mov_b32_dpp %" #i ", %16 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
#define OP_ADD_SALU(i) "v_add_f32 %" #i ", %" #i ", %16\ns_add_u32 s20, s20, 1\n"
#define OP_SALU(i) "s_add_u32 s20, s20, 1\n"
#define OP_READLANE(i) "v_readlane_b32 s20, %" #i ", 3\n"
#define OP_DS_READ(i) "ds_read_b32 %" #i ", %16\n"
#define OP_DS_READ64(i) "ds_read_b64 %" #i ", %16\n"
#define OP_DS_WRITE(i) "ds_write_b32 %16, %" #i "\n"
#define OP_SAND64(i) "s_and_b64 s[20:21], s[22:23], exec\n"
#define OP_SNOP(i) "s_nop 0\n"
#define OP_SWAIT(i) "s_waitcnt lgkmcnt(0)\n"
#define OP_V4_SALU(i) "v_min_f32 %" #i ", %" #i ", %16\ns_and_b64 s[20:21], s[22:23], exec\n"
#define OP_V4_SALU2(i) "v_min_f32 %" #i ", %" #i ", %16\ns_and_b64 s[20:21], s[22:23], exec\ns_or_b64 s[22:23], s[20:21], exec\n"
#define OP_CMP_SAND_CND(i) "v_cmp_lt_f32_e64 s[20:21], %" #i ", %16\ns_and_b64 s[22:23], s[20:21], exec\nv_cndmask_b32_e64 %" #i ", %" #i ", %17, s[22:23]\n"
#define OP_DEP_ADD(i) "v_add_f32 %0, %0, %16\n"
#define OP_DEP_MIN(i) "v_min_f32 %0, %0, %16\n"
#define OP_ADD_MUL_PAIR(i) "v_add_f32 %" #i ", %" #i ", %16\nv_fma_f32 %" #i ", %" #i ", %16, %17\n"

struct Out { unsigned long long ticks; unsigned long long clocks; };

template <class T, int KIND>
__global__ __launch_bounds__(1024) void k_stream(Out *out, T seed, int iters) {
  __shared__ float lds_hold[21 * 1024];   // 84 KB (only reserves LDS: one workgroup per CU)
  if (iters < 0) lds_hold[threadIdx.x] = (float)seed;
  T a[16];
  T x = seed, y = seed;
#pragma unroll
  for (int i = 0; i < 16; i++) a[i] = seed + (T)(threadIdx.x + i);
  if (KIND == 4 || KIND == 20) asm volatile("v_cmp_gt_u32 vcc, %0, %1" ::"v"(threadIdx.x), "v"(17u) : "vcc");
  if (KIND == 19) asm volatile("v_cmp_gt_u32_e64 s[20:21], %0, %1" ::"v"(threadIdx.x), "v"(17u) : "s20", "s21");
  if (KIND == 41 || KIND == 42) x = (T)((threadIdx.x * 4) & 8191);   // LDS byte address
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      if constexpr (KIND == 0) STREAM16(OP_ADD_F32);
      if constexpr (KIND == 1) STREAM16(OP_MUL_F32);
      if constexpr (KIND == 2) STREAM16(OP_FMA_F32);
      if constexpr (KIND == 3) STREAM16(OP_ADD_U32);
      if constexpr (KIND == 4) STREAM16(OP_CNDMASK);
      if constexpr (KIND == 5) STREAM16(OP_MOV_DPP);
      if constexpr (KIND == 6) STREAM16(OP_RCP_F32);
      if constexpr (KIND == 7) STREAM16(OP_SQRT_F32);
      if constexpr (KIND == 8) STREAM16(OP_PK_ADD_F32);
      if constexpr (KIND == 9) STREAM16(OP_PK_MUL_F32);
      if constexpr (KIND == 10) STREAM16(OP_PK_FMA_F32);
      if constexpr (KIND == 11) STREAM16(OP_ADD_F64);
      if constexpr (KIND == 12) STREAM16(OP_MUL_F64);
      if constexpr (KIND == 13) STREAM16(OP_FMA_F64);
      if constexpr (KIND == 14) STREAM16(OP_RCP_F64);
      if constexpr (KIND == 15) STREAM16(OP_ADD_F32_E64);
      if constexpr (KIND == 16) STREAM16(OP_FMAC_F32);
      if constexpr (KIND == 17) STREAM16(OP_FMA_F32_2SRC);
      if constexpr (KIND == 18) STREAM16(OP_FMA_F32_K);
      if constexpr (KIND == 19) STREAM16(OP_CNDMASK_E64);
      if constexpr (KIND == 20) STREAM16(OP_CNDMASK_ALT);
      if constexpr (KIND == 21) STREAM16(OP_MOV);
      if constexpr (KIND == 22) STREAM16(OP_AND);
      if constexpr (KIND == 23) STREAM16(OP_LSHL);
      if constexpr (KIND == 24) STREAM16(OP_LSHL_ADD);
      if constexpr (KIND == 25) STREAM16(OP_ADD3);
      if constexpr (KIND == 26) STREAM16(OP_AND_OR);
      if constexpr (KIND == 27) STREAM16(OP_BFE);
      if constexpr (KIND == 28) STREAM16(OP_BFI);
      if constexpr (KIND == 29) STREAM16(OP_MAD24);
      if constexpr (KIND == 30) STREAM16(OP_MUL_LO);
      if constexpr (KIND == 31) STREAM16(OP_MIN_F32);
      if constexpr (KIND == 32) STREAM16(OP_MAX3_F32);
      if constexpr (KIND == 33) STREAM16(OP_CMP_VCC);
      if constexpr (KIND == 34) STREAM16(OP_CMP_SGPR);
      if constexpr (KIND == 35) STREAM16(OP_CMP_CND);
      if constexpr (KIND == 36) STREAM16(OP_MIN_DPP);
      if constexpr (KIND == 37) STREAM16(OP_MOV_BCAST);
      if constexpr (KIND == 38) STREAM16(OP_ADD_SALU);
      if constexpr (KIND == 39) STREAM16(OP_SALU);
      if constexpr (KIND == 40) STREAM16(OP_READLANE);
      if constexpr (KIND == 41) STREAM16(OP_DS_READ);
      if constexpr (KIND == 42) STREAM16(OP_DS_WRITE);
      if constexpr (KIND == 43) STREAM16(OP_ADD_MUL_PAIR);
      if constexpr (KIND == 44) STREAM16(OP_SAND64);
      if constexpr (KIND == 45) STREAM16(OP_SNOP);
      if constexpr (KIND == 46) STREAM16(OP_SWAIT);
      if constexpr (KIND == 47) STREAM16(OP_V4_SALU);
      if constexpr (KIND == 48) STREAM16(OP_V4_SALU2);
      if constexpr (KIND == 49) STREAM16(OP_CMP_SAND_CND);
      if constexpr (KIND == 50) STREAM16(OP_DEP_ADD);
      if constexpr (KIND == 51) STREAM16(OP_DEP_MIN);
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  T s = a[0];
#pragma unroll
  for (int i = 1; i < 16; i++) s = s + a[i];
  if (s == (T)123456789) out[0].ticks = (unsigned long long)lds_hold[threadIdx.x ^ 1];   // keeps the accumulators (and the LDS block) alive
  if ((threadIdx.x & 63) == 0) {
    Out &o = out[1 + blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64];
    o.ticks = t1 - t0;
    o.clocks = c1 - c0;
  }
}

struct Case { const char *name; int kind; bool wide; const char *note; };

int main(int argc, char **argv) {
  const char *only = argc > 1 ? argv[1] : nullptr;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  int clk_khz = 0;
  CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
  printf("# VALU issue rate of %s (%s): %d CUs, reported clock %.0f MHz\n\n", prop.name, prop.gcnArchName, ncu, clk_khz / 1e3);
  printf("Streams of independent instructions (16 accumulators), %d instructions per wavefront, one workgroup of 256*W threads per CU.\n",
         16 * UNROLL * ITERS);
  printf("`cyc/inst/SIMD` = shader-clock cycles (s_memtime of the slowest wavefront x the clock ratio below) x 1 / (W x instructions);\n"
         "`G inst/s` = all wavefronts' instructions / wall time (HIP events).\n\n");
  const Case cases[] = {
      {"v_add_f32", 0, false, ""},          {"v_mul_f32", 1, false, ""},          {"v_fma_f32", 2, false, ""},
      {"v_add_u32", 3, false, ""},          {"v_cndmask_b32", 4, false, ""},      {"v_mov_b32_dpp quad_perm", 5, false, ""},
      {"v_rcp_f32", 6, false, "transcendental"}, {"v_sqrt_f32", 7, false, "transcendental"},
      {"v_pk_add_f32", 8, true, "2 fp32 per lane"}, {"v_pk_mul_f32", 9, true, "2 fp32 per lane"}, {"v_pk_fma_f32", 10, true, "2 fp32 per lane"},
      {"v_add_f64", 11, true, ""},          {"v_mul_f64", 12, true, ""},          {"v_fma_f64", 13, true, ""},
      {"v_rcp_f64", 14, true, "transcendental"},
      {"v_add_f32_e64 (VOP3 encoding)", 15, false, "64-bit encoding"},
      {"v_fmac_f32 (VOP2)", 16, false, "32-bit encoding, 3 register reads"},
      {"v_fma_f32 a,a,x,x", 17, false, "2 distinct source registers"},
      {"v_fma_f32 a,a,x,1.0", 18, false, "inline constant"},
      {"v_cndmask_b32_e64 sgpr mask", 19, false, ""},
      {"v_cndmask_b32 d,x,y,vcc (independent of d)", 20, false, ""},
      {"v_mov_b32", 21, false, ""}, {"v_and_b32", 22, false, ""}, {"v_lshlrev_b32", 23, false, ""},
      {"v_lshl_add_u32", 24, false, "VOP3"}, {"v_add3_u32", 25, false, "VOP3, 3 reads"}, {"v_and_or_b32", 26, false, "VOP3, 3 reads"},
      {"v_bfe_u32", 27, false, "VOP3, inline constants"}, {"v_bfi_b32", 28, false, "VOP3, 3 reads"},
      {"v_mad_u32_u24", 29, false, "VOP3, 3 reads"}, {"v_mul_lo_u32", 30, false, ""},
      {"v_min_f32", 31, false, ""}, {"v_max3_f32", 32, false, "VOP3, 3 reads"},
      {"v_cmp_lt_f32 -> vcc", 33, false, ""}, {"v_cmp_lt_f32_e64 -> sgpr", 34, false, ""},
      {"v_cmp + v_cndmask pair (per pair)", 35, false, "2 instructions per count"},
      {"v_min_f32_dpp quad_perm", 36, false, ""}, {"v_mov_b32_dpp row_newbcast", 37, false, ""},
      {"v_add_f32 + s_add_u32 (per pair)", 38, false, "VALU + SALU of one wavefront"},
      {"s_add_u32", 39, false, "SALU"}, {"v_readlane_b32", 40, false, ""},
      {"ds_read_b32", 41, false, "LDS, conflict-free"}, {"ds_write_b32", 42, false, "LDS, conflict-free"},
      {"v_add_f32 + v_fma_f32 (per pair)", 43, false, "2 instructions per count"},
      {"s_and_b64", 44, false, "SALU"}, {"s_nop 0", 45, false, ""}, {"s_waitcnt lgkmcnt(0) (nothing pending)", 46, false, ""},
      {"v_min_f32 + s_and_b64 (per pair)", 47, false, "4-cycle VALU + SALU of one wavefront"},
      {"v_min_f32 + 2 SALU (per triple)", 48, false, ""},
      {"v_cmp_e64 -> s_and_b64 -> v_cndmask_e64 (per triple, dependent through SGPRs)", 49, false, ""},
      {"v_add_f32 dependent chain", 50, false, "every instruction needs the previous result"},
      {"v_min_f32 dependent chain", 51, false, "every instruction needs the previous result"},
  };
  Out *d_out;
  const int maxw = ncu * 16 + 1;
  CK(hipMalloc(&d_out, sizeof(Out) * maxw));
  std::vector<Out> h(maxw);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("| instruction | W (wavefronts / SIMD) | ticks (slowest wave, s_memtime) | cycle counter (s_memrealtime-free: readcyclecounter) | wall ms | G wave-inst/s (chip) | cyc/inst/SIMD at wall x 2.4 GHz | inst/cycle/SIMD by the cycle counter | note |\n");
  printf("|---|---|---|---|---|---|---|---|---|\n");
  for (const Case &c : cases) {
    if (only && !strstr(c.name, only)) continue;
    for (int W = 1; W <= 4; W++) {
      const int threads = 256 * W;
      const size_t lds = 0;
      auto launch = [&](int iters) {
        switch (c.kind) {
#define L32(K) case K: hipLaunchKernelGGL((k_stream<float, K>), dim3(ncu), dim3(threads), lds, 0, d_out, 1.0f, iters); break;
#define LU32(K) case K: hipLaunchKernelGGL((k_stream<unsigned, K>), dim3(ncu), dim3(threads), lds, 0, d_out, 1u, iters); break;
#define L64(K) case K: hipLaunchKernelGGL((k_stream<double, K>), dim3(ncu), dim3(threads), lds, 0, d_out, 1.0, iters); break;
          L32(0) L32(1) L32(2) LU32(3) LU32(4) LU32(5) L32(6) L32(7) L64(8) L64(9) L64(10) L64(11) L64(12) L64(13) L64(14)
          L32(15) L32(16) L32(17) L32(18) LU32(19) LU32(20) LU32(21) LU32(22) LU32(23) LU32(24) LU32(25) LU32(26) LU32(27) LU32(28) LU32(29) LU32(30)
          L32(31) L32(32) L32(33) L32(34) L32(35) L32(36) LU32(37) L32(38) LU32(39) LU32(40) LU32(41) LU32(42) L32(43)
          LU32(44) LU32(45) LU32(46) L32(47) L32(48) L32(49) L32(50) L32(51)
        }
      };
      launch(64);   // warm-up (code object load, clocks)
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      launch(ITERS);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const int nw = ncu * 4 * W;
      CK(hipMemcpy(h.data(), d_out, sizeof(Out) * (nw + 1), hipMemcpyDeviceToHost));
      unsigned long long tmax = 0, cmax = 0;
      for (int i = 1; i <= nw; i++) { if (h[i].ticks > tmax) tmax = h[i].ticks; if (h[i].clocks > cmax) cmax = h[i].clocks; }
      const double ninst = 16.0 * UNROLL * ITERS;
      const double ginst = ninst * nw / (ms * 1e-3) / 1e9;
      const double cyc_wall = (ms * 1e-3) * 2.4e9 / (ninst * W);
      const double ipc = ninst * W / (double)cmax;
      printf("| `%s` | %d | %llu | %llu | %.3f | %.1f | %.2f | %.3f | %s |\n", c.name, W, tmax, cmax, ms, ginst, cyc_wall, ipc, c.note);
      fflush(stdout);
    }
  }
  printf("\n(s_memtime runs at a constant 100 MHz on gfx950; __builtin_readcyclecounter is the same counter on this target when the two columns agree.  The\n"
         "shader clock during a pass is GRBM_GUI_ACTIVE / wall time of the counter pass, tools/valu_calib.sh.)\n");
  return 0;
}
