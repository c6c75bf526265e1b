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
// Tile TM x 64 x 32 (TM = 64, or 32 for small grids: launch()), 4 waves as 2x2, each wave (TM/2) x 32 outputs.
// LDS tiles are row-major [row][32 k + 4 pad]: a thread parks the float4 it loaded with ONE ds_write_b128 and a lane fetches the four k
// values of its (row, k-group) with ONE ds_read_b128 (the 16 lanes of a read group hit 16 different 4-bank groups: row pitch 36 words), i.e.
// MFMA j of a 16-k half takes k = 4 lk + j instead of 4 j + lk -- A and W use the same assignment, so every product meets its partner and only
// the order of the fp32 additions inside a 16-k group differs from the textbook one.  Two LDS buffers and TWO register sets: while tile t is
// multiplied, tile t+1 sits in the other buffer and the global loads of tiles t+2 and t+3 are in flight (asm loads, exact wait counts), one
// __syncthreads per K-tile.  Round-4 history of the vocoder decode at L = 938 (profiles/r04/r04_vocoder_f32_gemm.txt): 0.94 ms with every
// load waited for where it was issued (a select behind the load), 0.79 with the loads overlapping the MFMAs, 0.77 with exact wait counts,
// 0.59 with 32-row tiles on the under-filled grids.
#include "common.h"

namespace {

constexpr int TN = 64, TK = 32, LDK = TK + 4;

template <int EPI, int TM>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmF32Params p) {
  constexpr int TI = TM / 32;                       // 16-row MFMA tiles per wave (wave tile = TM/2 x 32)
  constexpr int AQ = TM / 32;                       // float4 loads per thread and K-tile for the A tile (TM rows x 8 float4)
  static_assert(TM == 32 || TM == 64, "tile heights");
  __shared__ __attribute__((aligned(16))) float As[2][TM][LDK];
  __shared__ __attribute__((aligned(16))) float Bs[2][TN][LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lk = lane >> 4;
  // Row-major tile order, NOT the XCD-blocked one of the bf16 GEMMs (common.h xcd_tile_coords): measured on the vocoder (16 of these launches,
  // 938 rows): the blocked order cuts the fabric-side bytes of a decode from 810 to 529 MB (every XCD no longer fetches every panel) and makes
  // the decode 13 % SLOWER (0.95 -> 1.07 ms), the K = 100 input projection of the step loop 6 % slower (profiles/r04/r04_f32_gemm_xcd_order.txt).
  // These launches are latency-bound chains of K-tiles, the panels come from the die-level cache either way, and eight XCDs asking for the
  // same lines at the same time is the cheaper pattern.
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;

  const int lrow = tid >> 3, lk4 = (tid & 7) * 4;   // loader: rows lrow (+ 32), k offset lk4 inside the tile
  const float* Wz = p.nbatch > 0 ? p.Wv[blockIdx.z] : p.W;
  const float* biasz = p.nbatch > 0 ? p.biasv[blockIdx.z] : p.bias;
  float* outz = p.out + (p.nbatch > 0 ? (size_t)blockIdx.z * p.out_bstride : 0);
  // (named scalars, not arrays: per-thread arrays touched from a lambda ended up in scratch memory -- 64-96 B per lane, and 35 % slower)
  const bool a_ok0 = (m0 + lrow) < p.M, a_ok1 = AQ == 2 && (m0 + lrow + 32) < p.M;
  const bool b_ok0 = (n0 + lrow) < p.N, b_ok1 = (n0 + lrow + 32) < p.N;
  const float* ap0 = p.A + (size_t)(a_ok0 ? m0 + lrow : 0) * p.lda + lk4;
  const float* ap1 = p.A + (size_t)(a_ok1 ? m0 + lrow + 32 : 0) * p.lda + lk4;
  const float* bp0 = Wz + (size_t)(b_ok0 ? n0 + lrow : 0) * p.ldw + lk4;
  const float* bp1 = Wz + (size_t)(b_ok1 ? n0 + lrow + 32 : 0) * p.ldw + lk4;

  f32x4 acc00 = {0.f, 0.f, 0.f, 0.f}, acc01 = acc00, acc10 = acc00, acc11 = acc00;      // [16-row tile of the wave][16-column tile]

  const int nk = (p.K + TK - 1) / TK;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // TWO register sets (x, y): the global loads of K-tiles t+2 and t+3 are in flight while tile t is multiplied and tile t+1 sits in the other
  // LDS buffer.  With one set (round 4's first form, and the 16-k kernel before it) a workgroup had ONE K-tile of loads in flight and every
  // iteration waited for an L2 / fabric round trip: 2.3 us per 32-k tile for the vocoder's pointwise convolutions against 0.43 us of MFMA work.
  const f32x4 zv = {0.f, 0.f, 0.f, 0.f};
  f32x4 ra0x = zv, ra1x = zv, rb0x = zv, rb1x = zv, ra0y = zv, ra1y = zv, rb0y = zv, rb1y = zv;
  // every load is unconditional from a clamped (always valid) address; rows / k groups outside the problem are zeroed when the set is
  // PARKED, not where it is loaded: a select right behind the load is a use, and the compiler put `s_waitcnt vmcnt(0)` there -- the loads
  // never overlapped the MFMAs at all.  (A `cond ? *p : 0` form is worse still: a pointer select against a zero parked in scratch, flat loads.)
  // The loads are asm (invisible to the compiler's wait-count bookkeeping, like the fragment reads of the bf16 GEMM) and every park waits
  // with an exact count: only for ITS set, the other set's NLD loads -- requested after it -- stay in flight.  Left to the compiler the park
  // at the head of a trip waits for everything (vmcnt 0 at the loop edge) and the younger set loses half of its look-ahead.
#define LEMAS_F32_ASMLD(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(ptr) : "memory")
#define LEMAS_F32_GLOAD(S, kt)                                                                      \
  do {                                                                                              \
    const bool kin = (kt) * TK + lk4 < p.K; /* K % 4 == 0: a float4 is inside or outside as a whole */ \
    const int koff = kin ? (kt) * TK : -lk4; /* outside: the row's first float4 (valid), zeroed at the park */ \
    LEMAS_F32_ASMLD(ra0##S, ap0 + koff);                                                            \
    if (AQ == 2) LEMAS_F32_ASMLD(ra1##S, ap1 + koff);                                               \
    LEMAS_F32_ASMLD(rb0##S, bp0 + koff);                                                            \
    LEMAS_F32_ASMLD(rb1##S, bp1 + koff);                                                            \
  } while (0)
#define LEMAS_F32_SEL(v, ok) ((ok) ? (v) : zv)
  // (the wait names the set's registers as read-write operands: without that data dependence the compiler hoists the selects below ABOVE
  // the wait -- seen in the ISA -- and parks registers the loads have not reached yet)
#define LEMAS_F32_PARK(S, buf, kt, INFLIGHT)                                                        \
  do {                                                                                              \
    const bool kin = (kt) * TK + lk4 < p.K;                                                         \
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(ra0##S), "+v"(ra1##S), "+v"(rb0##S), "+v"(rb1##S) : "n"(INFLIGHT) : "memory"); \
    *reinterpret_cast<f32x4*>(&As[buf][lrow][lk4]) = LEMAS_F32_SEL(ra0##S, a_ok0 && kin);           \
    if (AQ == 2) *reinterpret_cast<f32x4*>(&As[buf][(lrow + 32) % TM][lk4]) = LEMAS_F32_SEL(ra1##S, a_ok1 && kin); \
    *reinterpret_cast<f32x4*>(&Bs[buf][lrow][lk4]) = LEMAS_F32_SEL(rb0##S, b_ok0 && kin);           \
    *reinterpret_cast<f32x4*>(&Bs[buf][lrow + 32][lk4]) = LEMAS_F32_SEL(rb1##S, b_ok1 && kin);      \
  } while (0)
#define LEMAS_F32_MFMA4(av, bv0, bv1, c0, c1)                                  \
  c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(bv0, av, c0, 0, 0, 0); /* C^T: see the epilogue */ \
  c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(bv1, av, c1, 0, 0, 0);
#define LEMAS_F32_TILE(cur)                                                                                                          \
  _Pragma("unroll") for (int h = 0; h < 2; ++h) { /* two 16-k halves of the tile */                                                  \
    const float4 a0 = *reinterpret_cast<const float4*>(&As[cur][wm * (TM / 2) + l15][h * 16 + lk * 4]);                             \
    const float4 a1 = TI == 2 ? *reinterpret_cast<const float4*>(&As[cur][(wm * (TM / 2) + 16 + l15) % TM][h * 16 + lk * 4]) : z4;  \
    const float4 b0 = *reinterpret_cast<const float4*>(&Bs[cur][wn * 32 + l15][h * 16 + lk * 4]);                                   \
    const float4 b1 = *reinterpret_cast<const float4*>(&Bs[cur][wn * 32 + 16 + l15][h * 16 + lk * 4]);                              \
    LEMAS_F32_MFMA4(a0.x, b0.x, b1.x, acc00, acc01)                                                                                  \
    if (TI == 2) { LEMAS_F32_MFMA4(a1.x, b0.x, b1.x, acc10, acc11) }                                                                 \
    LEMAS_F32_MFMA4(a0.y, b0.y, b1.y, acc00, acc01)                                                                                  \
    if (TI == 2) { LEMAS_F32_MFMA4(a1.y, b0.y, b1.y, acc10, acc11) }                                                                 \
    LEMAS_F32_MFMA4(a0.z, b0.z, b1.z, acc00, acc01)                                                                                  \
    if (TI == 2) { LEMAS_F32_MFMA4(a1.z, b0.z, b1.z, acc10, acc11) }                                                                 \
    LEMAS_F32_MFMA4(a0.w, b0.w, b1.w, acc00, acc01)                                                                                  \
    if (TI == 2) { LEMAS_F32_MFMA4(a1.w, b0.w, b1.w, acc10, acc11) }                                                                 \
  }
  constexpr int NLD = AQ == 2 ? 4 : 3;      // loads per set
  LEMAS_F32_GLOAD(x, 0);
  LEMAS_F32_PARK(x, 0, 0, 0);
  const int last = nk - 1;
  LEMAS_F32_GLOAD(x, min(1, last));
  LEMAS_F32_GLOAD(y, min(2, last));
  __syncthreads();
  // Two K-tiles per trip: tile kt from buffer 0 while set x (tile kt+1) is parked in buffer 1 and refilled with tile kt+3, then tile kt+1
  // from buffer 1 while set y (tile kt+2) goes to buffer 0 and is refilled with tile kt+4.  A buffer is rewritten one barrier after its last
  // read; a set is parked two half-trips after it was requested.  The body is branch-free on purpose -- requests past the last tile re-read
  // the last one, parks past it write zeros nobody reads: with conditionals inside, the compiler's wait-count bookkeeping gives up at the
  // loop edge and waits for EVERY load in flight (vmcnt 0) at each park, which halves the look-ahead again.
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    LEMAS_F32_PARK(x, 1, kt + 1, NLD);
    LEMAS_F32_GLOAD(x, min(kt + 3, last));
    LEMAS_F32_TILE(0)
    __syncthreads();
    LEMAS_F32_PARK(y, 0, kt + 2, NLD);
    LEMAS_F32_GLOAD(y, min(kt + 4, last));
    LEMAS_F32_TILE(1)
    __syncthreads();
  }
  // The tail's redundant requests must land before their registers mean anything else: the compiler sees both sets dead past the loop and
  // would hand the registers out while the loads are still in flight (the operands keep them alive up to the wait).
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra0x), "+v"(ra1x), "+v"(rb0x), "+v"(rb1x), "+v"(ra0y), "+v"(ra1y), "+v"(rb0y), "+v"(rb1y)::"memory");
  if (nk & 1) { LEMAS_F32_TILE(0) }        // an odd tile count: the last tile was parked in buffer 0 by the final trip (or is tile 0)
#undef LEMAS_F32_TILE
#undef LEMAS_F32_GLOAD
#undef LEMAS_F32_SEL
#undef LEMAS_F32_ASMLD
#undef LEMAS_F32_PARK
#undef LEMAS_F32_MFMA4
  const f32x4 acc[2][2] = {{acc00, acc01}, {acc10, acc11}};

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
  (void)finish;
  const bool vec_ok = (p.ldc & 3) == 0 && (reinterpret_cast<size_t>(outz) & 15) == 0 &&
                      (EPI != F32_BIAS_ADD2 || (reinterpret_cast<size_t>(p.add) & 15) == 0);
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int m = m0 + wm * (TM / 2) + i * 16 + l15;
    if (m >= p.M) continue;
    const bool masked = EPI != F32_BIAS_ADD2 && p.rowmask && p.rowmask[m];
    constexpr bool ROWAFF = EPI == F32_ROWAFF_RELU || EPI == F32_ROWAFF_RELU_RES;
    float rs = 1.f, rh = 0.f;
    if (ROWAFF && p.rowscale) {
      const int ch = (m / p.rows_per_ch) % p.nch;
      rs = p.rowscale[ch]; rh = p.rowshift[ch];
    }
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
        if (ROWAFF) {
          float t = fmaxf(fmaf(rs, acc[i][j][r] + bias, rh), 0.f);
          if (EPI == F32_ROWAFF_RELU_RES && in) t += p.res[(size_t)m * p.ldres + n];
          v[r] = in ? t : 0.f;
        } else {
          v[r] = in ? finish(acc[i][j][r], m, n, bias, cs) : 0.f;
        }
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
  const int gn = (p.N + TN - 1) / TN, nb = p.nbatch > 0 ? p.nbatch : 1;
  // 64-row tiles for big grids; the 32-row form for problems of at most 32 rows (the per-utterance GEMVs: prosody projections, the time MLP)
  // and for grids of fewer than 400 64-row workgroups, i.e. under ~1.5 per CU (the vocoder at L ~ 900: N = 512 convolutions 120 -> 240
  // workgroups, 43.9 -> 27.9 us; N = 1536 360 -> 720, 28.9 -> 23.9 us).  Measured both ways: while a workgroup's loads did not overlap its
  // MFMAs the taller tile won (59 vs 66 us); with the loads in flight across two K-tiles the bound is what a CU can pull out of L2
  // (~11 B per cycle) and how evenly the workgroups cover the CUs, and the smaller tile wins on both counts at these sizes.
  const long g64 = (long)gn * ((p.M + 63) / 64) * nb;
  if (p.M <= 32 || g64 < 400) {
    hipLaunchKernelGGL((gemm_f32_kernel<EPI, 32>), dim3(gn, (p.M + 31) / 32, nb), dim3(256), 0, s, p);
  } else {
    hipLaunchKernelGGL((gemm_f32_kernel<EPI, 64>), dim3(gn, (p.M + 63) / 64, nb), dim3(256), 0, s, p);
  }
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
    case F32_ROWAFF_RELU: return launch<F32_ROWAFF_RELU>(p, s);
    case F32_ROWAFF_RELU_RES: return launch<F32_ROWAFF_RELU_RES>(p, s);
  }
  return hipErrorInvalidValue;
}
