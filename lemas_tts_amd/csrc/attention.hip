// Non-causal multi-head attention with key-padding mask, head_dim 64, flash-style online softmax (gfx950).
//
// Reference semantics: lemas_tts/model/modules.py:483-491 -- F.scaled_dot_product_attention(q, k, v,
// attn_mask = key-padding mask [B,1,1,N], is_causal=False), scale 1/sqrt(64); q,k already rotated.
//
// Layout contract (produced by the QKV GEMM epilogue): q, k [B2, H, pitch, 64] bf16; v^T [B2, H, 64, npad] bf16
// (npad % 64 == 0, the tail is finite); out [B2*pitch, H*64] bf16 (token-major, feeds the out-proj GEMM); pitch >= N
// is the per-sample row pitch of the activation row space (a multiple of 128).
//
// Work split: one workgroup = 128 queries of one (batch, head) = 4 waves x 32 queries; K and V^T tiles of 64 keys
// go global -> registers -> LDS one tile ahead (double-buffered, one barrier per tile).
//
// Both matmuls are computed TRANSPOSED so that everything the softmax needs is lane-local:
//   S^T[key, q] = K . Q^T   (A = K rows from LDS, B = Q fragment held in registers)
//   O^T[d,  q] = V^T . P^T  (A = V^T rows from LDS, B = P^T built in registers from S^T)
// With v_mfma_f32_32x32x16_bf16 the C fragment has col = lane&31 (= query) and 16 rows per lane, so each lane
// owns ONE query: row max / row sum / rescale are per-lane scalars plus a single exchange with lane^32.
// The S^T row index i is mapped to key kappa(i) = i with bits 2 and 3 swapped (the A operand simply reads LDS
// row kappa(lane&31)); then the 8 accumulator registers 8s..8s+7 of a lane hold 8 CONSECUTIVE keys
// 16s + 8*(lane>>5) + 0..7, which is exactly the k-slot order of the B operand of the second MFMA -- P never
// leaves registers and V^T is read with one ds_read_b128 per MFMA.
#include "common.h"

namespace {

constexpr int QB = 128;  // queries per workgroup
constexpr int KB = 64;   // keys per tile
constexpr int TILE = KB * 64 * 2;  // 8 KiB per operand tile

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const AttnParams p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE];  // [2][K 8K | V^T 8K]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int bh = blockIdx.y;
  const int b2 = bh / p.heads, h = bh - b2 * p.heads;
  const int N = p.n;
  const int kvlen = p.kv_len ? p.kv_len[b2 % p.batch] : N;
  const int q_base = blockIdx.x * QB + wave * 32;

  const bf16_t* Qg = p.q + (size_t)bh * p.pitch * 64;
  const bf16_t* Kg = p.k + (size_t)bh * p.pitch * 64;
  const bf16_t* Vg = p.vt + (size_t)bh * 64 * p.npad;

  // Q fragment (B operand of S^T = K.Q^T): lane (q = l31, hi) holds Q[q][kk*16 + hi*8 .. +7]
  bf16x8 qf[4];
  {
    int qrow = q_base + l31;
    qrow = qrow < N ? qrow : N - 1;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      qf[kk] = *reinterpret_cast<const bf16x8*>(Qg + (size_t)qrow * 64 + kk * 16 + hi * 8);
  }
  // kappa: swap bits 2 and 3 of the MFMA row index
  const int krow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);

  // staging coordinates: 512 16-B chunks per tile, 2 per thread per operand
  const int srow0 = tid >> 3, schunk = tid & 7;  // rows srow0 and srow0 + 32
  u32x4 rk[2], rv[2];
  const int ntiles = (kvlen + KB - 1) / KB;

  auto gload = [&](int j) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = srow0 + 32 * i;
      int key = j * KB + r;
      key = key < N ? key : N - 1;
      rk[i] = *reinterpret_cast<const u32x4*>(Kg + (size_t)key * 64 + schunk * 8);
      rv[i] = *reinterpret_cast<const u32x4*>(Vg + (size_t)r * p.npad + j * KB + schunk * 8);
    }
  };
  auto lwrite = [&](int buf) {
    char* d = smem + buf * 2 * TILE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = srow0 + 32 * i;
      *reinterpret_cast<u32x4*>(d + lds_off(r, schunk)) = rk[i];
      *reinterpret_cast<u32x4*>(d + TILE + lds_off(r, schunk)) = rv[i];
    }
  };

  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;
  const float c = p.scale * 1.4426950408889634f;  // softmax in base 2

  gload(0);
  lwrite(0);
  __syncthreads();

  for (int j = 0; j < ntiles; ++j) {
    const int buf = j & 1;
    if (j + 1 < ntiles) gload(j + 1);
    const char* sK = smem + buf * 2 * TILE;
    const char* sV = sK + TILE;

    // ---- S^T = K . Q^T for two 32-key sub-tiles
    f32x16 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(sK + lds_off(t * 32 + krow, kk * 2 + hi));
        s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[kk], s[t], 0, 0, 0);
      }
    }
    // lane's register r of sub-tile t is key  j*64 + 32 t + 16 (r>>3) + 8 hi + (r&7)
    if ((j + 1) * KB > kvlen) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = j * KB + 32 * t + 16 * (r >> 3) + 8 * hi + (r & 7);
          if (key >= kvlen) s[t][r] = -INFINITY;
        }
    }
    // ---- online softmax (per-lane query; partner lane^32 holds the other half of the keys)
    float mx = s[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx * c);
    const float alpha = exp2f(m_run - m_new);  // first tile: exp2(-inf) = 0
    m_run = m_new;
    float psum = 0.f;
    bf16x8 pb[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = exp2f(fmaf(s[t][r], c, -m_new));
        psum += pv;
        pb[t][r >> 3][r & 7] = (bf16_t)pv;
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }

    // ---- O^T += V^T . P^T
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(sV + lds_off(dt * 32 + l31, 4 * t + 2 * ss + hi));
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pb[t][ss], o[dt], 0, 0, 0);
        }

    if (j + 1 < ntiles) lwrite(buf ^ 1);
    __syncthreads();
  }

  // ---- normalise and store: lane owns query q_base + l31; rows of O^T are d = 32 dt + (r&3) + 8 (r>>2) + 4 hi
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int q = q_base + l31;
  if (q < N) {
    bf16_t* dst = p.out + ((size_t)b2 * p.pitch + q) * (p.heads * 64) + h * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (bf16_t)(o[dt][g * 4 + e] * inv);
        *reinterpret_cast<bf16x4*>(dst + dt * 32 + 8 * g + 4 * hi) = v;
      }
  }
}

}  // namespace

hipError_t launch_attention(const AttnParams& p, hipStream_t s) {
  if (p.npad % 64 != 0 || p.n <= 0 || p.pitch < p.n) return hipErrorInvalidValue;
  dim3 grid((p.n + QB - 1) / QB, p.b2 * p.heads);
  hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(256), 0, s, p);
  return hipGetLastError();
}
