// lds_granule_probe.hip -- how many one-wavefront workgroups a gfx950 CU really holds for a given LDS size.
//
// The eikonal kernel's workgroups per CU are set by their LDS (one wavefront each, heaps in LDS), and the occupancy API
// (hipOccupancyMaxActiveBlocksPerMultiprocessor) rounds the LDS size differently from the hardware allocator: with 12 864 bytes it
// answered 12 while the kernel ran as if on 11 (-6 %, round 5).  This program measures residency directly: workgroups of 64
// threads with S bytes of dynamic LDS sleep ~200 us each; every workgroup adds one to its CU's counter when it starts and takes it
// off when it ends, and the largest value any CU's counter reached is the residency.  (CU identity: HW_REG_HW_ID se/sh/cu + XCC_ID.)
//   hipcc -O2 --offload-arch=gfx950 tools/lds_granule_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(64) void probe(int *cur, int *peak, int sleeps) {
  extern __shared__ int lds[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
  const unsigned id = ((xcc & 0xf) * 8 + se) * 32 + sh * 16 + cu;
  if (threadIdx.x == 0) {
    lds[0] = 1;
    const int c = atomicAdd(&cur[id], 1) + 1;
    atomicMax(&peak[id], c);
  }
  for (int i = 0; i < sleeps; i++) __builtin_amdgcn_s_sleep(127);
  if (threadIdx.x == 0) atomicSub(&cur[id], lds[0]);
}

int main() {
  int *cur, *peak;
  const int NID = 4096;
  CK(hipMalloc(&cur, NID * 4));
  CK(hipMalloc(&peak, NID * 4));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  CK(hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  printf("| LDS bytes per workgroup (64 threads) | occupancy API | measured workgroups per CU (max over CUs) | CUs seen |\n|---:|---:|---:|---:|\n");
  const int sizes[] = {8192, 10240, 12288, 12576, 12800, 12804, 12864, 13312, 14080, 16384, 16640, 16388, 18432, 18720, 19200, 24576, 32768};
  for (int S : sizes) {
    CK(hipMemset(cur, 0, NID * 4));
    CK(hipMemset(peak, 0, NID * 4));
    int api = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, probe, 64, S));
    hipLaunchKernelGGL(probe, dim3(prop.multiProcessorCount * 40), dim3(64), S, 0, cur, peak, 60);
    CK(hipDeviceSynchronize());
    std::vector<int> h(NID);
    CK(hipMemcpy(h.data(), peak, NID * 4, hipMemcpyDeviceToHost));
    int mx = 0, seen = 0;
    for (int v : h) { if (v > mx) mx = v; if (v) seen++; }
    printf("| %d | %d | %d | %d |\n", S, api, mx, seen);
  }
  return 0;
}
