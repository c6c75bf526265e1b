// Probe: what is the per-CU L2 -> LDS streaming rate of global_load_lds_dwordx4 (LDS-DMA), and of plain global_load_dwordx4,
// as a function of waves per CU and requests in flight?  The GEMM / attention K loops stream their operands this way; their
// measured K-loop times correspond to ~46 GB/s per CU.  Is that the hardware's ceiling or the loops' own limit?
//
//   ./lds_dma_rate_probe
// Each workgroup (1 per CU, 256 WGs) streams `span` bytes of ITS OWN region repeatedly (region re-read => L2 / MALL resident
// after the first pass), every wave keeping `depth` 1-KiB requests in flight (counted vmcnt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); return 1; } } while (0)

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// mode 0: LDS-DMA (global_load_lds_dwordx4); mode 1: global_load_dwordx4 -> VGPR (xor-accumulated so the loads stay live);
// mode 2: LDS-DMA through a buffer descriptor (buffer_load_dwordx4 ... offen lds: 32-bit per-lane offset + SGPR offset)
template <int DEPTH, int MODE>
__global__ __launch_bounds__(1024) void stream_kernel(const char* __restrict__ src, size_t span, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int nw = blockDim.x >> 6;
  const char* base = src + (size_t)blockIdx.x * span;
  const size_t pieces = span / 1024;          // 1 KiB pieces of this WG's region; wave w takes pieces w, w + nw, ...
  unsigned acc = 0;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  size_t p = wave;
  const long total = (long)iters * (long)(pieces / nw);
  if (MODE == 2) {
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    const int voff = lane * 16;
    for (long i = 0; i < total; ++i) {
      char* l = smem + (wave * DEPTH + (int)(i % DEPTH)) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)l, 16, voff, (int)(p * 1024), 0, 0);
      wait_vmcnt<DEPTH - 1>();
      p += nw;
      if (p >= pieces) p = wave;
    }
  } else if (MODE == 0) {
    for (long i = 0; i < total; ++i) {
      const char* g = base + p * 1024 + lane * 16;
      char* l = smem + (wave * DEPTH + (int)(i % DEPTH)) * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
      wait_vmcnt<DEPTH - 1>();
      p += nw;
      if (p >= pieces) p = wave;
    }
  } else {
    for (long i = 0; i < total; i += DEPTH) {     // DEPTH independent loads in flight, then consumed together
      u32x4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        v[d] = *reinterpret_cast<const u32x4*>(base + p * 1024 + lane * 16);
        p += nw;
        if (p >= pieces) p = wave;
      }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc ^= v[d][0] ^ v[d][3];
    }
  }
  wait_vmcnt<0>();
  if (MODE != 1) acc = *reinterpret_cast<unsigned*>(smem + threadIdx.x * 4);
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int DEPTH, int MODE>
int run(const char* buf, size_t span, int waves, int iters, unsigned* sink, const char* label) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int lds = waves * DEPTH * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel<DEPTH, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL((stream_kernel<DEPTH, MODE>), dim3(256), dim3(64 * waves), lds, 0, buf, span, 2, sink);   // warm the caches
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((stream_kernel<DEPTH, MODE>), dim3(256), dim3(64 * waves), lds, 0, buf, span, iters, sink);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = 256.0 * (double)span * iters;
  printf("%-10s span %5zu KB/WG  waves %2d  depth %2d : %7.1f us  %6.1f GB/s per CU  %6.2f TB/s chip\n", label, span >> 10, waves, DEPTH, 1e3 * ms,
         bytes / 256.0 / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12);
  return 0;
}

// cycles to ISSUE 8 LDS-DMA pieces back to back from one wave (s_memtime around the issue, nothing else running on the CU)
template <int MODE>
__global__ void issue_cost_kernel(const char* src, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* g = src + (size_t)blockIdx.x * 65536 + wave * 8192 + lane * 16;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)blockIdx.x * 65536), 0, 0x7fffffff, 0x00020000);
  unsigned long long t0, t1, t2;
  for (int rep = 0; rep < 3; ++rep) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      char* l = smem + (wave * 8 + q) * 1024;
      if (MODE == 0) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + q * 1024), (__attribute__((address_space(3))) void*)l, 16, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)l, 16, wave * 8192 + lane * 16, q * 1024, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2));
  }
  if (lane == 0) { out[(blockIdx.x * 16 + wave) * 2] = t1 - t0; out[(blockIdx.x * 16 + wave) * 2 + 1] = t2 - t0; }
}

template <int MODE>
int issue_cost(const char* buf, int waves, const char* label) {
  unsigned long long* out;
  CK(hipMalloc((void**)&out, 256 * 16 * 2 * 8));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(issue_cost_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL((issue_cost_kernel<MODE>), dim3(256), dim3(64 * waves), waves * 8192, 0, buf, out);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(256 * 16 * 2);
  CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
  double a = 0, b = 0;
  for (int w = 0; w < waves; ++w) { a += h[w * 2]; b += h[w * 2 + 1]; }
  printf("%-10s waves %2d: issue of 8 pieces %6.0f cycles (%4.0f per piece), issue -> all landed %6.0f cycles (WG 0, third repetition, L2 warm)\n", label, waves,
         a / waves, a / waves / 8, b / waves);
  CK(hipFree(out));
  return 0;
}

int main() {
  char* buf; unsigned* sink;
  const size_t maxspan = 1 << 20;
  CK(hipMalloc((void**)&buf, 256 * maxspan));
  CK(hipMemset(buf, 1, 256 * maxspan));
  CK(hipMalloc((void**)&sink, 64));
  CK(hipDeviceSynchronize());
  for (int waves : {1, 4, 8}) { issue_cost<0>(buf, waves, "global-lds"); issue_cost<1>(buf, waves, "buffer-lds"); }
  for (size_t span : {(size_t)64 << 10, (size_t)512 << 10}) {        // 64 KB/WG: 16 MB total (L2: 4 MB per XCD -> partly L2, MALL); 512 KB/WG: 128 MB (MALL)
    const int iters = span == (64 << 10) ? 64 : 8;
    for (int waves : {4, 8, 16}) {
      run<2, 0>(buf, span, waves, iters, sink, "lds-dma");
      run<4, 0>(buf, span, waves, iters, sink, "lds-dma");
      run<8, 0>(buf, span, waves, iters, sink, "lds-dma");
      run<8, 1>(buf, span, waves, iters, sink, "vgpr-load");
      run<2, 2>(buf, span, waves, iters, sink, "buf-lds");
      run<4, 2>(buf, span, waves, iters, sink, "buf-lds");
    }
  }
  return 0;
}
