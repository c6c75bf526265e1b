// Where does a bf16 GEMM K loop lose the matrix pipe?  8 waves per CU, each iteration = R x ds_read_b128 + 4 x v_mfma_f32_32x32x16_bf16
// (4 independent accumulators), no barriers, no DMA.  Variants: accumulators in VGPRs or AGPRs; 0 / 4 / 8 LDS reads per 4 MFMAs;
// optional s_barrier per 16 MFMAs.   hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_lds_probe.hip -o tools/exp/mfma_lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int AGPR, int READS, int BARRIER>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += 512) ((float*)lds)[i] = 1e-3f * i;
  __syncthreads();
  const unsigned addr = base + wave * 4096 + ((lane & 31) * 128 + (((lane >> 5) ^ ((lane >> 1) & 7)) << 4));
  u32x4 fa = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, fb = fa, r0 = fa, r1 = fa, r2 = fa, r3 = fa, r4 = fa, r5 = fa, r6 = fa, r7 = fa;
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int it = 0; it < iters; ++it) {
    if (READS >= 4) {
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\tds_read_b128 %2, %4 offset:32768\n\tds_read_b128 %3, %4 offset:36864"
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr) : "memory");
    }
    if (READS >= 8) {
      asm volatile("ds_read_b128 %0, %4 offset:8192\n\tds_read_b128 %1, %4 offset:12288\n\tds_read_b128 %2, %4 offset:40960\n\tds_read_b128 %3, %4 offset:45056"
                   : "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(addr) : "memory");
    }
    if (AGPR) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %6, %1\n\tv_mfma_f32_32x32x16_bf16 %2, %5, %6, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %6, %4, %3"
                   : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : "v"(fa), "v"(fb), "v"(r0));
    } else {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %6, %1\n\tv_mfma_f32_32x32x16_bf16 %2, %5, %6, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %6, %4, %3"
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(fa), "v"(fb), "v"(r0));
    }
    if (READS >= 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));   // consume last iteration's data next time round
    if (READS >= 8) asm volatile("" : "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
    fa = r1; fb = r2;
    if (BARRIER && (it & 3) == 3) __builtin_amdgcn_s_barrier();
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  out[blockIdx.x * 512 + threadIdx.x] = s + (float)(r3[0] + r5[0] + r7[1]);
}

template <int AGPR, int READS, int BARRIER>
void run(float* out, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4096, blocks = 256;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    probe<AGPR, READS, BARRIER><<<blocks, 512>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double fl = 2.0 * 32 * 32 * 16 * 4.0 * iters * 8 * blocks;
  printf("%-54s %7.1f TFLOP/s  (%.3f ms)\n", name, fl / (ms * 1e-3) / 1e12, ms);
}

int main() {
  float* out; hipMalloc(&out, 4 * 512 * 256);
  run<0, 0, 0>(out, "VGPR acc, no LDS reads");
  run<1, 0, 0>(out, "AGPR acc, no LDS reads");
  run<0, 4, 0>(out, "VGPR acc, 4 ds_read_b128 per 4 MFMA (wait same iter)");
  run<1, 4, 0>(out, "AGPR acc, 4 ds_read_b128 per 4 MFMA");
  run<0, 8, 0>(out, "VGPR acc, 8 ds_read_b128 per 4 MFMA");
  run<1, 8, 0>(out, "AGPR acc, 8 ds_read_b128 per 4 MFMA");
  run<0, 4, 1>(out, "VGPR acc, 4 reads per 4 MFMA, s_barrier per 16 MFMA");
  run<1, 4, 1>(out, "AGPR acc, 4 reads per 4 MFMA, s_barrier per 16 MFMA");
  return 0;
}
