// Can a stream capture fork into MORE than two streams on this runtime?  Round 1 saw the HIP runtime crash while capturing a
// four-stream pattern (two CFG lanes, each with a side stream).  This probe captures, from one origin stream, a two-lane graph in
// which every "block" of a lane forks a side stream beside a kernel and joins it again -- the shape a concurrent LN-mod branch needs.
//   hipcc --offload-arch=gfx950 -O2 tools/exp/capture_fork_probe.hip -o /tmp/capture_fork_probe && /tmp/capture_fork_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void work(float* p, int n, int iters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = p[i];
  for (int k = 0; k < iters; ++k) v = v * 1.0001f + 0.5f;
  p[i] = v;
}

int main() {
  const int n = 1 << 20, blocks = 22;
  float* buf[4];
  for (auto& b : buf) { CK(hipMalloc(&b, n * 4)); CK(hipMemset(b, 0, n * 4)); }
  hipStream_t s0, lane1, side0, side1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&lane1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&side0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&side1, hipStreamNonBlocking));
  std::vector<hipEvent_t> ev(4 * blocks + 4);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  int ne = 0;
  CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
  CK(hipEventRecord(ev[ne], s0)); CK(hipStreamWaitEvent(lane1, ev[ne], 0)); ++ne;      // fork lane 1
  hipStream_t lane[2] = {s0, lane1}, side[2] = {side0, side1};
  for (int b = 0; b < blocks; ++b)
    for (int l = 0; l < 2; ++l) {
      CK(hipEventRecord(ev[ne], lane[l])); CK(hipStreamWaitEvent(side[l], ev[ne], 0)); ++ne;   // fork the side stream
      hipLaunchKernelGGL(work, dim3(n / 256), dim3(256), 0, lane[l], buf[l], n, 200);
      hipLaunchKernelGGL(work, dim3(n / 256), dim3(256), 0, side[l], buf[2 + l], n, 100);
      CK(hipEventRecord(ev[ne], side[l])); CK(hipStreamWaitEvent(lane[l], ev[ne], 0)); ++ne;   // join it
      hipLaunchKernelGGL(work, dim3(n / 256), dim3(256), 0, lane[l], buf[l], n, 200);
    }
  CK(hipEventRecord(ev[ne], lane1)); CK(hipStreamWaitEvent(s0, ev[ne], 0)); ++ne;          // join lane 1
  hipGraph_t g;
  CK(hipStreamEndCapture(s0, &g));
  size_t nodes = 0;
  CK(hipGraphGetNodes(g, nullptr, &nodes));
  hipGraphExec_t ge;
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t t0, t1;
  CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s0));
  CK(hipEventRecord(t0, s0));
  for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, s0));
  CK(hipEventRecord(t1, s0));
  CK(hipStreamSynchronize(s0));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, t0, t1));
  float h = 0;
  CK(hipMemcpy(&h, buf[0], 4, hipMemcpyDeviceToHost));
  printf("captured %zu nodes from 4 streams; 10 replays %.3f ms (%.1f us per replay); buf[0][0] = %g\nOK\n", nodes, ms, 100.f * ms, h);
  return 0;
}
