// How much LDS can the workgroups that share a CU really use together?  (gfx950: 160 KB per CU.)  Asks the runtime's occupancy calculator
// for 512-thread workgroups of a given dynamic LDS size, and verifies with a launch that measures how many workgroups of two different
// kernels are resident on one CU at the same time.     hipcc --offload-arch=gfx950 -O2 tools/exp/lds_fit_probe.hip -o /tmp/lds_fit && /tmp/lds_fit
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(512) void k(int* out) {
  extern __shared__ char smem[];
  smem[threadIdx.x] = (char)threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0 && out) out[blockIdx.x] = smem[5];
}

int main() {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int sizes[] = {32768, 40960, 53248, 54272, 54784, 65536, 65600, 66048, 80 * 1024, 80 * 1024 + 64, 81408, 96 * 1024, 97 * 1024, 98304 + 512, 160 * 1024};
  for (int s : sizes) {
    int n = -1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 512, s);
    printf("dynamic LDS %6d B: %d workgroup(s) of 512 threads per CU (%s)\n", s, n, hipGetErrorString(e));
  }
  return 0;
}
