// ConvPositionEmbedding: grouped Conv1d(k=31, groups=16, pad=15) + Mish, twice, + residual (gfx950).
// Reference: lemas_tts/model/modules.py:167-190, called WITHOUT a mask from backbones/dit.py:98
// (x = conv_pos_embed(x) + x), so every sample is zero-padded at its own sequence ends and padded frames of
// a batch DO leak into their neighbours -- reproduced as is.
//
// Formulation: per group (64 in / 64 out channels) the convolution is 31 accumulating GEMMs
//     out[f, co] += in[f + tap - 15, ci] * W[tap][co][ci]            (K = 64 per tap)
// on v_mfma_f32_32x32x16_bf16.  A workgroup owns 128 frames x one group of one sample; the input slab
// (128 + 30 frames x 64 ch, bf16) is staged once in LDS and re-read at a row offset per tap; tap weights stream
// through LDS four taps at a time, fetched to registers one stage ahead.  Rows are 128 B with the same
// XOR swizzle as the GEMM (conflict-free b128 reads for any row shift, since 16 consecutive rows always cover
// all residues mod 16).
#include "common.h"

namespace {

constexpr int FB = 128;        // frames per workgroup
constexpr int CG = 64;         // channels per group
constexpr int TPS = 4;         // taps per weight stage
constexpr int MAXTAPS = 31;
constexpr int SLAB_ROWS = FB + MAXTAPS - 1;
constexpr int SLAB_BYTES = SLAB_ROWS * 128;
constexpr int WTAP_BYTES = CG * 128;  // 8 KiB per tap

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <bool FIRST>
__global__ __launch_bounds__(256, 2) void convpos_kernel(const ConvPosParams p) {
  __shared__ __attribute__((aligned(16))) char smem[SLAB_BYTES + TPS * WTAP_BYTES];
  char* sIn = smem;
  char* sW = smem + SLAB_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int g = blockIdx.y, b = blockIdx.z;
  const int f0 = blockIdx.x * FB;
  const int N = p.n, C = p.channels, taps = p.taps, half = taps / 2;

  // ---- stage the input slab (frames f0-half .. f0+FB+half-1), zero outside [0, N).  All of a thread's chunks are requested before the
  // first one is converted: every load is unconditional from a clamped frame and zeroed by a select at the LDS write (with the load inside
  // the `if`, each of the five trips waited for its own round trip to L2 / the fabric: ~5 us of a 20 us launch).
  constexpr int NCH = (SLAB_ROWS * 8 + 255) / 256;
  u32x4 sv[NCH][FIRST ? 2 : 1];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int t = tid + 256 * i, r = t >> 3, ch = t & 7;
    int f = f0 - half + r;
    f = f < 0 ? 0 : f >= N ? N - 1 : f;
    const size_t off = ((size_t)b * p.pitch + f) * C + g * CG + ch * 8;
    if (FIRST) {
      sv[i][0] = *reinterpret_cast<const u32x4*>(p.in_f32 + off);
      sv[i][FIRST ? 1 : 0] = *reinterpret_cast<const u32x4*>(p.in_f32 + off + 4);
    } else {
      sv[i][0] = *reinterpret_cast<const u32x4*>(p.in_bf16 + off);
    }
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int t = tid + 256 * i, r = t >> 3, ch = t & 7;
    const int f = f0 - half + r;
    const bool ok = f >= 0 && f < N && r < FB + taps - 1;
    u32x4 v;
    if (FIRST) {
      const float4 a = __builtin_bit_cast(float4, sv[i][0]), c = __builtin_bit_cast(float4, sv[i][FIRST ? 1 : 0]);
      bf16x8 w;
      w[0] = (bf16_t)a.x; w[1] = (bf16_t)a.y; w[2] = (bf16_t)a.z; w[3] = (bf16_t)a.w;
      w[4] = (bf16_t)c.x; w[5] = (bf16_t)c.y; w[6] = (bf16_t)c.z; w[7] = (bf16_t)c.w;
      v = __builtin_bit_cast(u32x4, w);
    } else {
      v = sv[i][0];
    }
    v = ok ? v : u32x4{0u, 0u, 0u, 0u};
    if (t < SLAB_ROWS * 8) *reinterpret_cast<u32x4*>(sIn + lds_off(r, ch)) = v;
  }

  const bf16_t* wg = p.w + (size_t)g * taps * CG * CG;
  // weight stage: TPS taps x 64 rows x 8 chunks = 2048 chunks, 8 per thread
  u32x4 rw[8];
  auto wload = [&](int stage) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int id = tid + 256 * i;        // 0 .. 2047
      const int tp = id >> 9;              // tap within stage
      const int tap = stage * TPS + tp;
      // (taps past the last one of the final stage re-read it: they are never multiplied -- the tap loop below skips them)
      const int tc = tap < taps ? tap : taps - 1;
      rw[i] = *reinterpret_cast<const u32x4*>(wg + (size_t)tc * CG * CG + (id & 511) * 8);
    }
  };
  auto wwrite = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int id = tid + 256 * i;
      const int tp = id >> 9, row = (id & 511) >> 3, ch = id & 7;
      *reinterpret_cast<u32x4*>(sW + tp * WTAP_BYTES + lds_off(row, ch)) = rw[i];
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }

  const int nstages = (taps + TPS - 1) / TPS;
  wload(0);
  wwrite();
  __syncthreads();
  for (int st = 0; st < nstages; ++st) {
    if (st + 1 < nstages) wload(st + 1);
#pragma unroll
    for (int tp = 0; tp < TPS; ++tp) {
      const int tap = st * TPS + tp;
      if (tap < taps) {
        const int arow = wave * 32 + l31 + tap;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(sIn + lds_off(arow, kk * 2 + hi));
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const bf16x8 w = *reinterpret_cast<const bf16x8*>(sW + tp * WTAP_BYTES + lds_off(j * 32 + l31, kk * 2 + hi));
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a, acc[j], 0, 0, 0);   // swapped: a lane owns one frame
          }
        }
      }
    }
    __syncthreads();
    if (st + 1 < nstages) {
      wwrite();
      __syncthreads();
    }
  }

  // ---- epilogue: bias + Mish (+ residual).  Swapped operands: lane -> frame f0 + 32 wave + (lane & 31), register r ->
  // channel 32 j + (r & 3) + 8 (r >> 2) + 4 hi.  Each wave parks its 32 x 64 tile in a private LDS slab (the input slab is
  // dead after the last barrier) and writes whole rows: 128 B (bf16) / 256 B (fp32) contiguous per frame instead of 2048
  // scattered 2- or 4-byte stores per wave.
  constexpr int PITCH = CG * 4 + 16;
  char* slab = smem + wave * (32 * PITCH);
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cl = j * 32 + 8 * q + 4 * hi;
      const float4 bias = *reinterpret_cast<const float4*>(p.bias + g * CG + cl);
      *reinterpret_cast<float4*>(slab + l31 * PITCH + cl * 4) =
          make_float4(mish_f(acc[j][4 * q + 0] + bias.x), mish_f(acc[j][4 * q + 1] + bias.y), mish_f(acc[j][4 * q + 2] + bias.z),
                      mish_f(acc[j][4 * q + 3] + bias.w));
    }
  if (FIRST) {
    const int rr = lane >> 3, ch = lane & 7;                 // 8 rows x 8 chunks (8 bf16 = 16 B) per pass
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int fr = it * 8 + rr, f = f0 + wave * 32 + fr;
      const float4 a = *reinterpret_cast<const float4*>(slab + fr * PITCH + ch * 32);
      const float4 c = *reinterpret_cast<const float4*>(slab + fr * PITCH + ch * 32 + 16);
      bf16x8 o;
      o[0] = (bf16_t)a.x; o[1] = (bf16_t)a.y; o[2] = (bf16_t)a.z; o[3] = (bf16_t)a.w;
      o[4] = (bf16_t)c.x; o[5] = (bf16_t)c.y; o[6] = (bf16_t)c.z; o[7] = (bf16_t)c.w;
      if (f < N) store_wt_b128(p.out_bf16 + ((size_t)b * p.pitch + f) * C + g * CG + ch * 8, __builtin_bit_cast(u32x4, o));
    }
  } else {
    const int rr = lane >> 4, ch = lane & 15;                // 4 rows x 16 chunks (4 fp32 = 16 B) per pass
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int fr = it * 4 + rr, f = f0 + wave * 32 + fr;
      if (f < N) {
        const size_t off = ((size_t)b * p.pitch + f) * C + g * CG + ch * 4;
        const float4 a = *reinterpret_cast<const float4*>(slab + fr * PITCH + ch * 16);
        const float4 r4 = *reinterpret_cast<const float4*>(p.residual + off);
        const float4 o = make_float4(a.x + r4.x, a.y + r4.y, a.z + r4.z, a.w + r4.w);
        store_wt_b128(p.out_f32 + off, __builtin_bit_cast(u32x4, o));
      }
    }
  }
}

}  // namespace

hipError_t launch_convpos(const ConvPosParams& p, hipStream_t s) {
  if (p.channels / p.groups != CG || p.taps > MAXTAPS || (p.taps & 1) == 0 || p.pitch < p.n) return hipErrorInvalidValue;
  dim3 grid((p.n + FB - 1) / FB, p.groups, p.b2);
  if (p.in_f32) hipLaunchKernelGGL(convpos_kernel<true>, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(convpos_kernel<false>, grid, dim3(256), 0, s, p);
  return hipGetLastError();
}
