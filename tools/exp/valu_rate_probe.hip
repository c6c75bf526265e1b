// Issue-rate probe for the VALU ops of the softmax inner loop on gfx950 (wave64).  clk/instr per wave from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/valu_rate_probe.hip -o tools/exp/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void probe(float* out, long long* clk, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  const f32x2 c = {0.999f, 0.999f}, d = {1e-3f, 1e-3f};
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) {   // v_exp_f32 x8, independent
      a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3);
      a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7);
    } else if (OP == 1) {   // v_fma_f32 x8
      a0 = fmaf(a0, 0.999f, 1e-3f); a1 = fmaf(a1, 0.999f, 1e-3f); a2 = fmaf(a2, 0.999f, 1e-3f); a3 = fmaf(a3, 0.999f, 1e-3f);
      a4 = fmaf(a4, 0.999f, 1e-3f); a5 = fmaf(a5, 0.999f, 1e-3f); a6 = fmaf(a6, 0.999f, 1e-3f); a7 = fmaf(a7, 0.999f, 1e-3f);
    } else if (OP == 2) {   // v_pk_fma_f32 x4 (8 values)
      p0 = p0 * c + d; p1 = p1 * c + d; p2 = p2 * c + d; p3 = p3 * c + d;
    } else if (OP == 3) {   // v_max3_f32 x8
      a0 = fmaxf(fmaxf(a0, a1), a2); a1 = fmaxf(fmaxf(a1, a2), a3); a2 = fmaxf(fmaxf(a2, a3), a4); a3 = fmaxf(fmaxf(a3, a4), a5);
      a4 = fmaxf(fmaxf(a4, a5), a6); a5 = fmaxf(fmaxf(a5, a6), a7); a6 = fmaxf(fmaxf(a6, a7), a0); a7 = fmaxf(fmaxf(a7, a0), a1);
    } else {   // v_cvt_pk_bf16_f32 x4
      typedef __bf16 b2 __attribute__((ext_vector_type(2)));
      b2 r0 = {(__bf16)a0, (__bf16)a1}, r1 = {(__bf16)a2, (__bf16)a3}, r2 = {(__bf16)a4, (__bf16)a5}, r3 = {(__bf16)a6, (__bf16)a7};
      a0 = (float)r0[0] + 1.f; a1 = (float)r0[1]; a2 = (float)r1[0]; a3 = (float)r1[1]; a4 = (float)r2[0]; a5 = (float)r2[1]; a6 = (float)r3[0]; a7 = (float)r3[1];
    }
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1];
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

int main() {
  float* out; long long* clk; hipMalloc(&out, 4 * 1024 * 256); hipMalloc(&clk, 8);
  const char* names[] = {"v_exp_f32 x8", "v_fma_f32 x8", "v_pk_fma_f32 x4 (8 values)", "v_max3_f32 x8", "cvt bf16 round trip"};
  const int iters = 4096;
  for (int waves = 1; waves <= 4; waves *= 2)        // waves per SIMD: block of 256*waves threads on one CU
    for (int op = 0; op < 4; ++op) {
      long long h = 0;
      for (int rep = 0; rep < 2; ++rep) {
        if (op == 0) probe<0><<<1, 256 * waves>>>(out, clk, iters);
        if (op == 1) probe<1><<<1, 256 * waves>>>(out, clk, iters);
        if (op == 2) probe<2><<<1, 256 * waves>>>(out, clk, iters);
        if (op == 3) probe<3><<<1, 256 * waves>>>(out, clk, iters);
        hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
      }
      printf("%d wave(s)/SIMD  %-28s %.2f clk per loop of 8 values per wave -> %.2f clk per wave-instruction (SIMD time / waves)\n", waves, names[op],
             (double)h / iters, (double)h / iters / (op == 2 ? 4 : 8) / 1.0);
    }
  return 0;
}
