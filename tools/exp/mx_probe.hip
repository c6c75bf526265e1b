// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 on gfx950: operand layout, block-scale mapping, issue rate.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/mx_probe.hip -o tools/exp/mx_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline uint32_t pack4(float a, float b, float c, float d) {
  int v = 0;
  v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
  return (uint32_t)v;
}

// A [32][64] f32 row-major, B^T [32][64] (row j = column j of B), out D [32][32].  mode 0: scales 0 (unscaled);
// mode 1: scale_a byte0 = 127 + (lane>>5) (x2 on the upper-half lanes' K block), scale_b = 127.
__global__ void layout_probe(const float* A, const float* Bt, float* D, int mode, uint8_t* a8) {
  const int l = threadIdx.x, i = l & 31, h = l >> 5;
  i32x8 a, b;
  for (int v = 0; v < 8; ++v) {
    const float* pa = A + i * 64 + h * 32 + v * 4;
    const float* pb = Bt + i * 64 + h * 32 + v * 4;
    a[v] = (int)pack4(pa[0], pa[1], pa[2], pa[3]);
    b[v] = (int)pack4(pb[0], pb[1], pb[2], pb[3]);
    ((uint32_t*)a8)[(i * 64 + h * 32) / 4 + v] = (uint32_t)a[v];
  }
  f32x16 c = {0};
  if (mode == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
  else if (mode == 1) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 127 + h, 0, 127);
  else if (mode == 2) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127 + h);
  else if (mode == 3) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 127 + (i & 1), 0, 127);   // odd rows x2
  else if (mode == 4) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 2, (127 + h) << 16, 0, 127);  // opsel byte 2
  else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;      // standard 32x32 C map: D[row][col = lane&31]
    D[row * 32 + i] = c[r];
  }
}

template <int SCALED>
__global__ void __launch_bounds__(256) rate_probe(float* out, int iters) {
  i32x8 a, b;
  for (int v = 0; v < 8; ++v) { a[v] = 0x38383838 + threadIdx.x; b[v] = 0x38303830 + v; }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int it = 0; it < iters; ++it) {
    if (SCALED) {
      c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 0, 0, 0, 127, 0, 127);
      c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 0, 0, 0, 127, 0, 127);
      c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 0, 0, 0, 127, 0, 127);
      c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 0, 0, 0, 127, 0, 127);
    } else {
      c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 0, 0, 0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 0, 0, 0, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 0, 0, 0, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 0, 0, 0, 0, 0, 0);
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static float e4m3(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + m / 8.f, e - 7);
  return s ? -x : x;
}

int main() {
  std::vector<float> A(32 * 64), Bt(32 * 64), D(32 * 32);
  uint32_t rng = 7;
  auto rnd = [&] { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) / 8388608.0f - 1.0f); };
  for (auto& x : A) x = 3.f * rnd();
  for (auto& x : Bt) x = 2.f * rnd() + 0.25f;
  float *dA, *dB, *dD; uint8_t* da8;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, Bt.size() * 4); hipMalloc(&dD, D.size() * 4); hipMalloc(&da8, 32 * 64);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, Bt.data(), Bt.size() * 4, hipMemcpyHostToDevice);
  // host reference uses the device's own fp8 roundings of A (read back) and a host RNE for B via the same kernel trick
  for (int mode = 0; mode < 6; ++mode) {
    layout_probe<<<1, 64>>>(dA, dB, dD, mode, da8);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    std::vector<uint8_t> a8(32 * 64), b8(32 * 64);
    hipMemcpy(a8.data(), da8, 32 * 64, hipMemcpyDeviceToHost);
    layout_probe<<<1, 64>>>(dB, dA, dD, 0, da8);     // reuse to fetch B's fp8 bytes
    hipMemcpy(b8.data(), da8, 32 * 64, hipMemcpyDeviceToHost);
    double err_contig = 0, err_plain = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double lo = 0, hi = 0;
      for (int k = 0; k < 32; ++k) lo += (double)e4m3(a8[i * 64 + k]) * e4m3(b8[j * 64 + k]);
      for (int k = 32; k < 64; ++k) hi += (double)e4m3(a8[i * 64 + k]) * e4m3(b8[j * 64 + k]);
      err_plain = fmax(err_plain, fabs(D[i * 32 + j] - (lo + hi)));
      err_contig = fmax(err_contig, fabs(D[i * 32 + j] - (lo + 2 * hi)));
    }
    {  // least-squares fit D = sum_q c_q Q_q over the four 16-wide K quarters (separately for even / odd rows)
      for (int par = 0; par < 2; ++par) {
        double N[4][5] = {{0}};
        for (int i = par; i < 32; i += 2) for (int j = 0; j < 32; ++j) {
          double Q[4] = {0, 0, 0, 0};
          for (int k = 0; k < 64; ++k) Q[k / 16] += (double)e4m3(a8[i * 64 + k]) * e4m3(b8[j * 64 + k]);
          for (int x = 0; x < 4; ++x) { for (int y = 0; y < 4; ++y) N[x][y] += Q[x] * Q[y]; N[x][4] += Q[x] * D[i * 32 + j]; }
        }
        for (int x = 0; x < 4; ++x) {   // Gauss-Jordan
          double pv = N[x][x];
          for (int y = 0; y < 5; ++y) N[x][y] /= pv;
          for (int z = 0; z < 4; ++z) if (z != x) { double f = N[z][x]; for (int y = 0; y < 5; ++y) N[z][y] -= f * N[x][y]; }
        }
        printf("  mode %d rows %s: K-quarter coefficients %.3f %.3f %.3f %.3f\n", mode, par ? "odd " : "even", N[0][4], N[1][4], N[2][4], N[3][4]);
      }
    }
    printf("mode %d: max|D - A.B| = %.3g   max|D - (lo + 2 hi)| = %.3g   (fp8 sample: %g -> %g)\n", mode, err_plain, err_contig, A[5], e4m3(a8[5]));
  }
  float* dO; hipMalloc(&dO, 256 * 2048 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int sc = 0; sc < 2; ++sc) {
    const int iters = 20000, blocks = 256 * 4;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (sc) rate_probe<1><<<blocks, 256>>>(dO, iters); else rate_probe<0><<<blocks, 256>>>(dO, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * 32 * 32 * 64 * 4.0 * iters * 4 /*waves*/ * blocks;
    printf("%s 32x32x64 fp8: %.1f TFLOP/s (%.3f ms)\n", sc ? "scaled" : "unscaled(scale=0)", fl / (ms * 1e-3) / 1e12, ms);
  }
  return 0;
}
