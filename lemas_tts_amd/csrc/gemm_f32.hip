// Exact-fp32 GEMM on the f32-input matrix cores (v_mfma_f32_16x16x4_f32), with small fused epilogues (gfx950).
//
//   out[M, N] = A[M, K] . W[N, K]^T (+ bias ...)        fp32 in, fp32 accumulate == an fmaf chain
//
// Used where the path must stay fp32 and the FLOPs are negligible next to the bf16 step loop:
//   * once per utterance: text ConvNeXtV2 linears (dit.py:73-77), the hoisted cond/text part of the input
//     projection (dit.py:97), prosody projections, the time MLP + all AdaLN modulation vectors for every ODE
//     step (modules.py:311,332,727-731) -- the t-grid is known up front;
//   * once per ODE step: the K=100 "x" part of the input projection (keeps the ODE state path in fp32);
//   * the Vocos vocoder (all of it: embed conv as im2col GEMM, pointwise convs, head, inverse DFT).
// Tile 64x64x16, 4 waves as 2x2, each wave 32x32 = 2x2 MFMA tiles.  LDS tiles are stored K-major so the
// one-float-per-lane fragments are consecutive words (conflict-free ds_read_b32).
#include "common.h"

namespace {

constexpr int TM = 64, TN = 64, TK = 16, LD = TM + 4;

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmF32Params p) {
  __shared__ float As[TK][LD];
  __shared__ float Bs[TK][LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lk = lane >> 4;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;

  const int lrow = tid >> 2, lk4 = (tid & 3) * 4;
  const bool a_ok = (m0 + lrow) < p.M, b_ok = (n0 + lrow) < p.N;
  const float* Wz = p.nbatch > 0 ? p.Wv[blockIdx.z] : p.W;
  const float* biasz = p.nbatch > 0 ? p.biasv[blockIdx.z] : p.bias;
  float* outz = p.out + (p.nbatch > 0 ? (size_t)blockIdx.z * p.out_bstride : 0);
  const float* ap = p.A + (size_t)(a_ok ? m0 + lrow : 0) * p.lda + lk4;
  const float* bp = Wz + (size_t)(b_ok ? n0 + lrow : 0) * p.ldw + lk4;

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + TK - 1) / TK;
  float4 ra, rb;
  auto gload = [&](int kt) {
    const int k = kt * TK + lk4;
    ra = (a_ok && k < p.K) ? *reinterpret_cast<const float4*>(ap + kt * TK) : make_float4(0.f, 0.f, 0.f, 0.f);
    rb = (b_ok && k < p.K) ? *reinterpret_cast<const float4*>(bp + kt * TK) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  gload(0);
  for (int kt = 0; kt < nk; ++kt) {
    As[lk4 + 0][lrow] = ra.x; As[lk4 + 1][lrow] = ra.y; As[lk4 + 2][lrow] = ra.z; As[lk4 + 3][lrow] = ra.w;
    Bs[lk4 + 0][lrow] = rb.x; Bs[lk4 + 1][lrow] = rb.y; Bs[lk4 + 2][lrow] = rb.z; Bs[lk4 + 3][lrow] = rb.w;
    __syncthreads();
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float a[2], b[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = As[ks * 4 + lk][wm * 32 + t * 16 + l15];
        b[t] = Bs[ks * 4 + lk][wn * 32 + t * 16 + l15];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], a[i], acc[i][j], 0, 0, 0);   // C^T: see the epilogue
    }
    __syncthreads();
  }

  // The MFMA runs with its operands swapped (C^T = W . A^T; same products, same k order, same sums), so a lane owns ONE output row and
  // FOUR CONSECUTIVE columns: row = lane & 15, columns (lane >> 4) * 4 + reg.  Bias / residual / add operands and the result move as
  // 16-B vectors wherever the row pitch and the column count allow it (a quarter of the load / store instructions of the scalar form,
  // which is what the K = 100 input-projection launch of every ODE step was made of); otherwise element by element.
  auto finish = [&](float v, int m, int n, float bias, float cs) -> float {
    v += bias;
    if (EPI == F32_BIAS_GELU) v = gelu_erf_f(v);
    if (EPI == F32_BIAS_SILU) v = silu_f(v);
    if (EPI == F32_BIAS_RELU) v = fmaxf(v, 0.f);
    if (EPI == F32_BIAS_SIGMOID) v = 1.0f / (1.0f + expf(-v));
    if (EPI == F32_BIAS_RES_SCALE) v = p.res[(size_t)m * p.ldres + n] + cs * v;
    return v;
  };
  const bool vec_ok = (p.ldc & 3) == 0 && (reinterpret_cast<size_t>(outz) & 15) == 0 &&
                      (EPI != F32_BIAS_ADD2 || (reinterpret_cast<size_t>(p.add) & 15) == 0);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 32 + i * 16 + l15;
    if (m >= p.M) continue;
    const bool masked = EPI != F32_BIAS_ADD2 && p.rowmask && p.rowmask[m];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nb = n0 + wn * 32 + j * 16 + lk * 4;
      if (nb >= p.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nb + r;
        const bool in = n < p.N;
        const float bias = (biasz && in) ? biasz[n] : 0.f;
        const float cs = (EPI == F32_BIAS_RES_SCALE && p.colscale && in) ? p.colscale[n] : 1.f;
        v[r] = in ? finish(acc[i][j][r], m, n, bias, cs) : 0.f;
        if (masked) v[r] = 0.f;
      }
      if (vec_ok && nb + 3 < p.N) {
        if (EPI == F32_BIAS_ADD2) {
          const float4 a0 = *reinterpret_cast<const float4*>(p.add + (size_t)m * p.ldc + nb);
          const float4 a1 = *reinterpret_cast<const float4*>(p.add + (size_t)(m + p.M) * p.ldc + nb);
          *reinterpret_cast<float4*>(outz + (size_t)m * p.ldc + nb) = make_float4(v[0] + a0.x, v[1] + a0.y, v[2] + a0.z, v[3] + a0.w);
          *reinterpret_cast<float4*>(outz + (size_t)(m + p.M) * p.ldc + nb) = make_float4(v[0] + a1.x, v[1] + a1.y, v[2] + a1.z, v[3] + a1.w);
        } else {
          *reinterpret_cast<float4*>(outz + (size_t)m * p.ldc + nb) = make_float4(v[0], v[1], v[2], v[3]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = nb + r;
          if (n >= p.N) continue;
          if (EPI == F32_BIAS_ADD2) {
            outz[(size_t)m * p.ldc + n] = v[r] + p.add[(size_t)m * p.ldc + n];
            outz[(size_t)(m + p.M) * p.ldc + n] = v[r] + p.add[(size_t)(m + p.M) * p.ldc + n];
          } else {
            outz[(size_t)m * p.ldc + n] = v[r];
          }
        }
      }
    }
  }
}

template <int EPI>
hipError_t launch(const GemmF32Params& p, hipStream_t s) {
  dim3 grid((p.N + TN - 1) / TN, (p.M + TM - 1) / TM, p.nbatch > 0 ? p.nbatch : 1);
  hipLaunchKernelGGL(gemm_f32_kernel<EPI>, grid, dim3(256), 0, s, p);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_gemm_f32(int epi, const GemmF32Params& p, hipStream_t s) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K & 3) || (p.lda & 3) || (p.ldw & 3)) return hipErrorInvalidValue;
  if (p.nbatch > 0 && (!p.Wv || !p.biasv)) return hipErrorInvalidValue;
  switch (epi) {
    case F32_BIAS: return launch<F32_BIAS>(p, s);
    case F32_BIAS_GELU: return launch<F32_BIAS_GELU>(p, s);
    case F32_BIAS_SILU: return launch<F32_BIAS_SILU>(p, s);
    case F32_BIAS_RES_SCALE: return launch<F32_BIAS_RES_SCALE>(p, s);
    case F32_BIAS_ADD2: return launch<F32_BIAS_ADD2>(p, s);
    case F32_BIAS_RELU: return launch<F32_BIAS_RELU>(p, s);
    case F32_BIAS_SIGMOID: return launch<F32_BIAS_SIGMOID>(p, s);
  }
  return hipErrorInvalidValue;
}
