// Probe: is hipMemset (NULL stream) complete when it returns, as seen from a NON-BLOCKING stream?
// DevBuf::ensure() relied on that: hipMalloc + hipMemset(NULL stream), then work on the engine's non-blocking stream.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill_kernel(float* p, size_t n, float v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  // 1) timing: does a 2 GiB hipMemset return before the device finished?
  {
    void* p; size_t n = (size_t)2 << 30;
    CK(hipMalloc(&p, n));
    CK(hipDeviceSynchronize());
    for (int r = 0; r < 3; ++r) {
      auto t0 = std::chrono::steady_clock::now();
      CK(hipMemset(p, 0, n));
      auto t1 = std::chrono::steady_clock::now();
      CK(hipDeviceSynchronize());
      auto t2 = std::chrono::steady_clock::now();
      printf("hipMemset 2 GiB: call %.1f us, then deviceSync %.1f us\n", std::chrono::duration<double, std::micro>(t1 - t0).count(),
             std::chrono::duration<double, std::micro>(t2 - t1).count());
    }
    CK(hipFree(p));
  }
  // 2) race: small buffers, memset then H2D copy on the non-blocking stream
  int lost_small = 0, lost_big = 0;
  const int iters = 2000;
  std::vector<float> h(64, 1.0f);
  for (int it = 0; it < iters; ++it) {
    float* p;
    CK(hipMalloc((void**)&p, 256));
    CK(hipMemset(p, 0, 256));
    CK(hipMemcpyAsync(p, h.data(), 12, hipMemcpyHostToDevice, s));
    CK(hipStreamSynchronize(s));
    CK(hipDeviceSynchronize());
    float back[3];
    CK(hipMemcpy(back, p, 12, hipMemcpyDeviceToHost));
    if (back[0] != 1.0f || back[1] != 1.0f || back[2] != 1.0f) ++lost_small;
    CK(hipFree(p));
  }
  // 3) race: big buffer, memset then a kernel on the non-blocking stream writing 1.0 everywhere
  for (int it = 0; it < 200; ++it) {
    float* p; size_t n = (size_t)32 << 20;   // 128 MiB
    CK(hipMalloc((void**)&p, n * 4));
    CK(hipMemset(p, 0, n * 4));
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, s, p, n, 1.0f);
    CK(hipStreamSynchronize(s));
    CK(hipDeviceSynchronize());
    std::vector<float> back(1024);
    int bad = 0;
    for (size_t off : {(size_t)0, n / 2, n - 1024}) {
      CK(hipMemcpy(back.data(), p + off, 4096, hipMemcpyDeviceToHost));
      for (float v : back) bad += v != 1.0f;
    }
    if (bad) ++lost_big;
    CK(hipFree(p));
  }
  // 4) the fix used by DevBuf::ensure: hipMemsetAsync on the NULL stream + hipStreamSynchronize(NULL) before anything else
  int lost_fixed = 0;
  for (int it = 0; it < 200; ++it) {
    float* p; size_t n = (size_t)32 << 20;
    CK(hipMalloc((void**)&p, n * 4));
    CK(hipMemsetAsync(p, 0, n * 4, nullptr));
    CK(hipStreamSynchronize(nullptr));
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, s, p, n, 1.0f);
    CK(hipStreamSynchronize(s));
    CK(hipDeviceSynchronize());
    std::vector<float> back(1024);
    int bad = 0;
    for (size_t off : {(size_t)0, n / 2, n - 1024}) {
      CK(hipMemcpy(back.data(), p + off, 4096, hipMemcpyDeviceToHost));
      for (float v : back) bad += v != 1.0f;
    }
    if (bad) ++lost_fixed;
    CK(hipFree(p));
  }
  printf("fixed (memsetAsync + streamSync(NULL)): %d / 200 iterations lost\n", lost_fixed);
  printf("small: %d / %d iterations lost the stream's write to a late memset\n", lost_small, iters);
  printf("big:   %d / 200 iterations lost the stream's write to a late memset\n", lost_big);
  return 0;
}
