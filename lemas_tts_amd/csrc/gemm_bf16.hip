// bf16 x bf16 -> fp32 GEMM on v_mfma_f32_32x32x16_bf16 with fused epilogues (gfx950).
//
//   C[M, N] = A[M, K] . W[N, K]^T          (nn.Linear layout: both operands K-contiguous)
//
// This is the workhorse of the DiT step loop: QKV (+bias +RoPE, scatter to head-major q/k and v^T),
// attention out-proj and FF2 (+bias, x += gate * .), FF1 (+bias, GELU-tanh), proj_out (+bias -> fp32).
// Reference semantics: lemas_tts/model/modules.py:452-461,470-480 (QKV + RoPE), :495,:635 (out-proj + gated
// residual), :349-350,:638-639 (FF), backbones/dit.py:252 (proj_out).
//
// Tiling: 128x128x64 block tile, 256 threads = 4 waves as 2(M) x 2(N), each wave 64x64 = 2x2 MFMA tiles of 32x32.
// Both tiles are staged global -> registers -> LDS (16 B per lane, 8 lanes cover one 128 B row: coalesced),
// issued one K-tile ahead of use (loads fly during the MFMAs), double-buffered LDS, one barrier per K-tile.
// LDS rows are 128 B; the 16-B chunk index is XOR-swizzled with (row>>1)&7 so that both the ds_write_b128
// (8 contiguous lanes = one row) and the ds_read_b128 fragment reads (16-lane groups = 16 different rows at
// one chunk) hit 16 distinct 16-B slots of the 256-B bank row: conflict-free (MI355X_MICROARCH.md, LDS).
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ void store_bf16(bf16_t* p, float v) { *p = (bf16_t)v; }

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A 16K | B 16K]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile order: consecutive logical ids (same A row-panel) land on one XCD's L2 (blocks are
  // dispatched round-robin over the 8 XCDs).  Bijective for any grid size.
  const int tiles_n = p.N / BN;
  const int nwg = gridDim.x;
  int lid;
  {
    const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;

  // per-thread staging coordinates: 4 x (row, chunk) for A and the same for W
  const int srow = tid >> 3, schunk = tid & 7;
  const bf16_t* ag[4];
  const bf16_t* wg[4];
  int soff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = srow + 32 * i;
    int am = m0 + r;
    am = am < p.M ? am : p.M - 1;  // clamp: out-of-range rows are computed and discarded
    ag[i] = p.A + (size_t)am * p.K + schunk * 8;
    wg[i] = p.W + (size_t)(n0 + r) * p.K + schunk * 8;
    soff[i] = lds_off(r, schunk);
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x4 ra[4], rb[4];
  const int nk = p.K / BK;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ra[i] = *reinterpret_cast<const u32x4*>(ag[i]);
    rb[i] = *reinterpret_cast<const u32x4*>(wg[i]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    *reinterpret_cast<u32x4*>(smem + soff[i]) = ra[i];
    *reinterpret_cast<u32x4*>(smem + TILE_BYTES + soff[i]) = rb[i];
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = *reinterpret_cast<const u32x4*>(ag[i] + (kt + 1) * BK);
        rb[i] = *reinterpret_cast<const u32x4*>(wg[i] + (kt + 1) * BK);
      }
    }
    const char* sA = smem + buf * 2 * TILE_BYTES;
    const char* sB = sA + TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 af[2], bf[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        af[t] = *reinterpret_cast<const bf16x8*>(sA + lds_off(wm * 64 + t * 32 + l31, kk * 2 + hi));
        bf[t] = *reinterpret_cast<const bf16x8*>(sB + lds_off(wn * 64 + t * 32 + l31, kk * 2 + hi));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      char* d = smem + (buf ^ 1) * 2 * TILE_BYTES;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<u32x4*>(d + soff[i]) = ra[i];
        *reinterpret_cast<u32x4*>(d + TILE_BYTES + soff[i]) = rb[i];
      }
    }
    __syncthreads();
  }

  // ---------------------------------------------------------------- epilogue
  // C fragment: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  const float* gate = nullptr;
  if (EPI == EPI_GATE_RES) gate = p.tab + (size_t)p.step_idx[0] * p.tab_stride + p.gate_off;

#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + l31;
    const float bias = p.bias ? p.bias[n] : 0.f;
    float g = 0.f;
    if (EPI == EPI_GATE_RES) g = gate[n];
    // qkv bookkeeping (uniform per 128-column tile because dim % 128 == 0)
    int which = 0, head = 0, d = 0;
    if (EPI == EPI_QKV_ROPE) {
      const int inner = p.heads * 64;
      which = n / inner;
      head = (n % inner) >> 6;
      d = n & 63;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        float v = acc[i][j][r] + bias;
        if (EPI == EPI_BIAS_BF16) {
          if (m < p.M && n < p.n_valid) store_bf16(p.out_bf16 + (size_t)m * p.ldc + n, v);
        } else if (EPI == EPI_BIAS_GELU_BF16) {
          if (m < p.M && n < p.n_valid) store_bf16(p.out_bf16 + (size_t)m * p.ldc + n, gelu_tanh_f(v));
        } else if (EPI == EPI_BIAS_F32) {
          if (m < p.M && n < p.n_valid) p.out_f32[(size_t)m * p.ldc + n] = v;
        } else if (EPI == EPI_GATE_RES) {
          if (m < p.M && n < p.n_valid) {
            bool live = true;
            if (p.kv_len) {
              const int b = (m / p.seq_len) % p.batch;
              live = (m % p.seq_len) < p.kv_len[b];
            }
            if (live) p.out_f32[(size_t)m * p.ldc + n] += g * v;
          }
        } else if (EPI == EPI_QKV_ROPE) {
          const float partner = __shfl_xor(v, 1, 64);  // the other half of the (2i, 2i+1) rotary pair
          if (m < p.M) {
            const int b2 = m / p.seq_len, pos = m % p.seq_len;
            if (which < 2) {
              const float c = p.rope_cos[pos * 32 + (d >> 1)], s = p.rope_sin[pos * 32 + (d >> 1)];
              const float o = (d & 1) ? (v * c + partner * s) : (v * c - partner * s);
              bf16_t* dst = (which == 0 ? p.q : p.k) + ((size_t)(b2 * p.heads + head) * p.seq_len + pos) * 64 + d;
              store_bf16(dst, o);
            } else {
              store_bf16(p.vt + ((size_t)(b2 * p.heads + head) * 64 + d) * p.npad + pos, v);
            }
          }
        }
      }
    }
  }
}

template <int EPI>
hipError_t launch(const GemmParams& p, hipStream_t s) {
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;
  hipLaunchKernelGGL(gemm_bf16_kernel<EPI>, dim3(tiles_m * tiles_n), dim3(256), 4 * TILE_BYTES, s, p);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_gemm_bf16(int epi, const GemmParams& p, hipStream_t s) {
  if (p.K % BK != 0 || p.N % BN != 0 || p.M <= 0) return hipErrorInvalidValue;
  switch (epi) {
    case EPI_BIAS_BF16: return launch<EPI_BIAS_BF16>(p, s);
    case EPI_BIAS_GELU_BF16: return launch<EPI_BIAS_GELU_BF16>(p, s);
    case EPI_BIAS_F32: return launch<EPI_BIAS_F32>(p, s);
    case EPI_GATE_RES: return launch<EPI_GATE_RES>(p, s);
    case EPI_QKV_ROPE: return launch<EPI_QKV_ROPE>(p, s);
  }
  return hipErrorInvalidValue;
}
