// LayerNorm-modulate of rows of 1024 held in registers: shared by the stand-alone ln_mod_kernel (norm_elementwise.hip) and by the
// LN tail of the gate+residual GEMM (gemm_bf16.hip), so that both forms produce the same bits for the same row.
//   out[c] = (x[c] - mean) * rstd * (1 + scale[c]) + shift[c],  LayerNorm without affine, eps 1e-6   (modules.py:314,637 / :335)
// Register image of one row (D = 1024 = 64 lanes x PER groups of 8 consecutive columns, each group two float4):
//   v[i][h] = row[8 (lane + 64 i) + 4 h .. + 3],  i < PER = 2, h < 2; the modulation vectors a (scale) / b (shift) use the same image.
#pragma once
#include "common.h"

constexpr int LN_D = 1024, LN_PER = LN_D / 512;

__device__ __forceinline__ void ln_load_vec(const float* __restrict__ vec, int lane, float4 (&a)[LN_PER][2]) {
  const float4* p = reinterpret_cast<const float4*>(vec);
#pragma unroll
  for (int i = 0; i < LN_PER; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) a[i][h] = p[(lane + 64 * i) * 2 + h];
}

// statistics of one row (two dependent wave reductions) and its modulated bf16 image as 16-B write-through stores.
// Every multiply-add is written out (fmaf or a separately rounded product under contract(off)): left to -ffp-contract=fast the two
// call sites were contracted differently and a handful of rows differed in the last bf16 bit between them.
__device__ __forceinline__ void ln_row_store(const float4 (&v)[LN_PER][2], const float4 (&a)[LN_PER][2], const float4 (&b)[LN_PER][2],
                                             bf16_t* __restrict__ orow, int lane, bool store) {
#pragma clang fp contract(off)
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_PER; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) s += (v[i][h].x + v[i][h].y) + (v[i][h].z + v[i][h].w);
  const float mean = wave_sum(s) * (1.0f / LN_D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_PER; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float c0 = v[i][h].x - mean, c1 = v[i][h].y - mean, c2 = v[i][h].z - mean, c3 = v[i][h].w - mean;
      q = __builtin_fmaf(c0, c0, q); q = __builtin_fmaf(c1, c1, q); q = __builtin_fmaf(c2, c2, q); q = __builtin_fmaf(c3, c3, q);
    }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / LN_D) + 1e-6f);
  if (store) {
#pragma unroll
    for (int i = 0; i < LN_PER; ++i) {
      bf16x8 o;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        o[4 * h + 0] = (bf16_t)__builtin_fmaf((v[i][h].x - mean) * rstd, 1.0f + a[i][h].x, b[i][h].x);
        o[4 * h + 1] = (bf16_t)__builtin_fmaf((v[i][h].y - mean) * rstd, 1.0f + a[i][h].y, b[i][h].y);
        o[4 * h + 2] = (bf16_t)__builtin_fmaf((v[i][h].z - mean) * rstd, 1.0f + a[i][h].z, b[i][h].z);
        o[4 * h + 3] = (bf16_t)__builtin_fmaf((v[i][h].w - mean) * rstd, 1.0f + a[i][h].w, b[i][h].w);
      }
      store_wt_b128(orow + (lane + 64 * i) * 8, __builtin_bit_cast(u32x4, o));
    }
  }
}
