// bf16 x bf16 -> fp32 GEMM on v_mfma_f32_32x32x16_bf16 with fused epilogues (gfx950).
//
//   C[M, N] = A[M, K] . W[N, K]^T          (nn.Linear layout: both operands K-contiguous)
//
// This is the workhorse of the DiT step loop: QKV (+bias +RoPE, scatter to head-major q/k and v^T),
// attention out-proj and FF2 (+bias, x += gate * .), FF1 (+bias, GELU-tanh), proj_out (+bias -> fp32).
// Reference semantics: lemas_tts/model/modules.py:452-461,470-480 (QKV + RoPE), :495,:635 (out-proj + gated
// residual), :349-350,:638-639 (FF), backbones/dit.py:252 (proj_out).
//
// Structure: BM x BN x 64 block tile, 4 waves as 2(M) x 2(N); operand tiles stream HBM/L2 -> LDS with
// global_load_lds_dwordx4 (LDS-DMA: no VGPR round trip, no ds_write pass) into an NSTAGE ring, one s_barrier per
// K-tile, counted s_waitcnt vmcnt so that NSTAGE-2 tiles stay in flight across the barrier.  LDS rows are 128 B
// with the 16-B chunk index XOR-swizzled by (row>>1)&7: conflict-free ds_read_b128 for the 32x32x16 fragments
// (16-lane groups hit 16 distinct slots of the 256-B bank row; SQ_LDS_BANK_CONFLICT = 0 measured).  The DMA writes
// LDS lane-linearly, so the swizzle is applied to each lane's global SOURCE address (the 8 lanes of a row still
// cover the same 128-B line).
//
// Epilogue: the MFMA is issued with SWAPPED operands (C^T = W . A^T) so a lane owns ONE output row m and, per
// accumulator group, FOUR CONSECUTIVE output columns: bias/gate are float4 loads, the fp32 residual update is a
// 16-B read-modify-write, bf16 results leave as 8-B stores, and the rotary pair (2i, 2i+1) is lane-local.  (With the
// plain orientation every lane stores 64 scattered 2/4-byte elements: measured, that epilogue cost 2x the whole
// K loop at these shapes.)  Only the V projection (its own launch, EPI_V_T) keeps the plain orientation, because v^T
// wants four consecutive POSITIONS per lane.
//
// Row space: activations are laid out as [sample][seq_pitch rows][...] with seq_pitch a multiple of 128, so a
// 128-row tile never straddles two samples and v^T stores are 8-B aligned; rows >= seq_valid are padding
// (computed, never stored where it matters).
#include "common.h"

namespace {

constexpr int BK = 64;

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ int xcd_tile_id() {
  // XCD-aware tile order: consecutive logical ids (same A row-panel) land on one XCD's L2 (blocks are dispatched
  // round-robin over the 8 XCDs).  Bijective for any grid size.
  const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ bf16x4 pack4(float a, float b, float c, float d) {
  bf16x4 o;
  o[0] = (bf16_t)a; o[1] = (bf16_t)b; o[2] = (bf16_t)c; o[3] = (bf16_t)d;
  return o;
}

// ---- swapped orientation: acc[i][j] = C^T tile; lane -> row m = mw + 32 i + (lane&31);
//      register r -> column n = nw + 32 j + (r&3) + 8 (r>>2) + 4 (lane>>5)
template <int EPI, int TI, int TJ>
__device__ __forceinline__ void epilogue_rows(const GemmParams& p, f32x16 (&acc)[TI][TJ], int mw, int nw, int l31, int hi) {
  const float* gate = nullptr;
  if (EPI == EPI_GATE_RES) gate = p.tab + (size_t)p.step_idx[0] * p.tab_stride + p.gate_off;
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int m = mw + i * 32 + l31;
    const int b2 = m / p.seq_pitch, pos = m - b2 * p.seq_pitch;
    bool live = m < p.M && pos < p.seq_valid;
    if (EPI == EPI_GATE_RES && p.kv_len) live = live && pos < p.kv_len[b2 % p.batch];
    // issue every global READ of this row before the first store: the compiler cannot hoist loads over the
    // (may-alias) stores itself, and a load -> wait -> store chain per 16-B chunk is pure exposed latency
    float4 xin[TJ][4];
    if (EPI == EPI_GATE_RES) {
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = nw + j * 32 + 8 * g + 4 * hi;
          xin[j][g] = (live && n < p.n_valid) ? *reinterpret_cast<const float4*>(p.out_f32 + (size_t)m * p.ldc + n)
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float2 rc[TJ][4], rs[TJ][4];
    if (EPI == EPI_QK_ROPE) {
      const int ps = live ? pos : 0;
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dh = ((nw + j * 32 + 8 * g + 4 * hi) & 63) >> 1;
          rc[j][g] = *reinterpret_cast<const float2*>(p.rope_cos + ps * 32 + dh);
          rs[j][g] = *reinterpret_cast<const float2*>(p.rope_sin + ps * 32 + dh);
        }
    }
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = nw + j * 32 + 8 * g + 4 * hi;
        const float4 bias = *reinterpret_cast<const float4*>(p.bias + n);
        float v0 = acc[i][j][4 * g + 0] + bias.x, v1 = acc[i][j][4 * g + 1] + bias.y;
        float v2 = acc[i][j][4 * g + 2] + bias.z, v3 = acc[i][j][4 * g + 3] + bias.w;
        if (EPI == EPI_BIAS_BF16) {
          if (live && n < p.n_valid) *reinterpret_cast<bf16x4*>(p.out_bf16 + (size_t)m * p.ldc + n) = pack4(v0, v1, v2, v3);
        } else if (EPI == EPI_BIAS_GELU_BF16) {
          if (live && n < p.n_valid)
            *reinterpret_cast<bf16x4*>(p.out_bf16 + (size_t)m * p.ldc + n) =
                pack4(gelu_tanh_f(v0), gelu_tanh_f(v1), gelu_tanh_f(v2), gelu_tanh_f(v3));
        } else if (EPI == EPI_BIAS_F32) {
          if (live && n < p.n_valid) *reinterpret_cast<float4*>(p.out_f32 + (size_t)m * p.ldc + n) = make_float4(v0, v1, v2, v3);
        } else if (EPI == EPI_GATE_RES) {
          if (live && n < p.n_valid) {
            const float4 gt = *reinterpret_cast<const float4*>(gate + n);
            float4 x = xin[j][g];
            x.x += gt.x * v0; x.y += gt.y * v1; x.z += gt.z * v2; x.w += gt.w * v3;
            *reinterpret_cast<float4*>(p.out_f32 + (size_t)m * p.ldc + n) = x;
          }
        } else if (EPI == EPI_QK_ROPE) {
          // q | k columns (v has its own launch, EPI_V_T): rotary pairs (d, d+1), (d+2, d+3) are lane-local
          if (live) {
            const int inner = p.heads * 64;
            const int which = n / inner, head = (n % inner) >> 6, d = n & 63;
            const float2 c = rc[j][g], s = rs[j][g];
            bf16_t* dst = (which == 0 ? p.q : p.k) + ((size_t)(b2 * p.heads + head) * p.seq_pitch + pos) * 64 + d;
            *reinterpret_cast<bf16x4*>(dst) = pack4(v0 * c.x - v1 * s.x, v1 * c.x + v0 * s.x, v2 * c.y - v3 * s.y, v3 * c.y + v2 * s.y);
          }
        }
      }
    }
  }
}

// ---- plain orientation, used only for the V projection (EPI_V_T): lane -> column n (= head dim d),
//      register r -> row m = mw + 32 i + (r&3) + 8 (r>>2) + 4 (lane>>5): four consecutive positions -> 8-B v^T store
template <int TI, int TJ>
__device__ __forceinline__ void epilogue_vt(const GemmParams& p, f32x16 (&acc)[TI][TJ], int mw, int nw, int l31, int hi) {
  const int inner = p.heads * 64;
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int n = nw + j * 32 + l31;
    const float bias = p.bias[n];
    const int head = (n % inner) >> 6, d = n & 63;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int m = mw + i * 32 + 8 * g + 4 * hi;
        const int b2 = m / p.seq_pitch, pos = m - b2 * p.seq_pitch;
        if (m < p.M && pos < p.seq_valid) {   // pos % 4 == 0; the row tail past seq_valid is padding inside v^T's pitch
          bf16_t* dst = p.vt + ((size_t)(b2 * p.heads + head) * 64 + d) * p.npad + pos;
          *reinterpret_cast<bf16x4*>(dst) = pack4(acc[i][j][4 * g + 0] + bias, acc[i][j][4 * g + 1] + bias,
                                                  acc[i][j][4 * g + 2] + bias, acc[i][j][4 * g + 3] + bias);
        }
      }
    }
  }
}

template <int EPI, int TBM, int TBN, int NSTAGE, int NWM, int NWN, bool SPREAD, bool SWAP>
__device__ __forceinline__ void gemm_body(const GemmParams& p, char* smem, int m0, int n0) {
  constexpr int NW = NWM * NWN;                       // waves per workgroup
  constexpr int WTM = TBM / NWM, WTN = TBN / NWN;     // wave tile
  constexpr int TI = WTM / 32, TJ = WTN / 32;         // 32x32 MFMA tiles per wave
  constexpr int A_BYTES = TBM * 128, STAGE = (TBM + TBN) * 128;
  constexpr int A_PW = TBM / 8 / NW, B_PW = TBN / 8 / NW;  // 1-KiB DMA instructions per wave per tile
  constexpr int PW = A_PW + B_PW;
  static_assert((TBM / 8) % NW == 0 && (TBN / 8) % NW == 0, "DMA pieces must divide over the waves");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wave / NWN, wn = wave % NWN;

  // DMA source pointers: instruction q of this wave covers LDS rows 8 (wave + NW q) .. +7; lane -> (row, phys chunk)
  const int lr = lane >> 3, lp = lane & 7;
  const bf16_t* asrc[A_PW];
  const bf16_t* wsrc[B_PW];
#pragma unroll
  for (int q = 0; q < A_PW; ++q) {
    const int r = 8 * (wave + NW * q) + lr;
    int am = m0 + r;
    am = am < p.M ? am : p.M - 1;  // clamp: out-of-range rows are computed and discarded
    asrc[q] = p.A + (size_t)am * p.K + ((lp ^ ((r >> 1) & 7)) << 3);
  }
#pragma unroll
  for (int q = 0; q < B_PW; ++q) {
    const int r = 8 * (wave + NW * q) + lr;
    wsrc[q] = p.W + (size_t)(n0 + r) * p.K + ((lp ^ ((r >> 1) & 7)) << 3);
  }
  // piece x of a tile: x < A_PW -> A rows, else W rows; each piece is one 1-KiB global_load_lds_dwordx4
  auto issue_piece = [&](int stage, int kt, int x) {
    char* base = smem + stage * STAGE + wave * 1024;
    if (x < A_PW)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[x] + kt * BK),
                                       (__attribute__((address_space(3))) void*)(base + x * (NW * 1024)), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[x - A_PW] + kt * BK),
                                       (__attribute__((address_space(3))) void*)(base + A_BYTES + (x - A_PW) * (NW * 1024)), 16, 0, 0);
  };
  auto issue = [&](int stage, int kt) {
#pragma unroll
    for (int x = 0; x < PW; ++x) issue_piece(stage, kt, x);
  };

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK;
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) issue(s, s);

  int stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt must have landed; in steady state the NSTAGE-2 younger tiles stay in flight across the barrier
    if (kt + NSTAGE - 2 < nk) wait_vmcnt<PW * (NSTAGE - 2)>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    // every wave is past its reads of the stage consumed in iteration kt-1: refill it with tile kt+NSTAGE-1
    const int nt = kt + NSTAGE - 1;
    int ns = stage + NSTAGE - 1;
    ns = ns >= NSTAGE ? ns - NSTAGE : ns;
    const bool refill = nt < nk;
    if (!SPREAD && refill) issue(ns, nt);
    const char* sA = smem + stage * STAGE;
    const char* sB = sA + A_BYTES;
    // fragments are double-buffered in registers: the ds_read_b128 of k-step kk+1 are in flight under the MFMAs of kk;
    // sched_barrier pins that order (the scheduler otherwise sinks the reads next to their consumers)
    bf16x8 af[2][TI], bf[2][TJ];
#pragma unroll
    for (int t = 0; t < TI; ++t) af[0][t] = *reinterpret_cast<const bf16x8*>(sA + lds_off(wm * WTM + t * 32 + l31, hi));
#pragma unroll
    for (int t = 0; t < TJ; ++t) bf[0][t] = *reinterpret_cast<const bf16x8*>(sB + lds_off(wn * WTN + t * 32 + l31, hi));
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk < 3) {
#pragma unroll
        for (int t = 0; t < TI; ++t)
          af[(kk + 1) & 1][t] = *reinterpret_cast<const bf16x8*>(sA + lds_off(wm * WTM + t * 32 + l31, (kk + 1) * 2 + hi));
#pragma unroll
        for (int t = 0; t < TJ; ++t)
          bf[(kk + 1) & 1][t] = *reinterpret_cast<const bf16x8*>(sB + lds_off(wn * WTN + t * 32 + l31, (kk + 1) * 2 + hi));
      }
      if (SPREAD && refill) {   // DMA issue slots hidden behind the MFMAs instead of a burst after the barrier
#pragma unroll
        for (int x = (PW * kk) / 4; x < (PW * (kk + 1)) / 4; ++x) issue_piece(ns, nt, x);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[kk & 1][j], af[kk & 1][i], acc[i][j], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1][i], bf[kk & 1][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    stage = stage + 1 == NSTAGE ? 0 : stage + 1;
  }
  if (SWAP) epilogue_rows<EPI, TI, TJ>(p, acc, m0 + wm * WTM, n0 + wn * WTN, l31, hi);
  else epilogue_vt<TI, TJ>(p, acc, m0 + wm * WTM, n0 + wn * WTN, l31, hi);
}

template <int EPI, int TBM, int TBN, int NSTAGE, int NWM, int NWN, bool SPREAD>
__global__ __launch_bounds__(64 * NWM * NWN) void gemm_bf16_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_n = p.N / TBN;
  const int lid = xcd_tile_id();
  const int m0 = (lid / tiles_n) * TBM, n0 = (lid % tiles_n) * TBN;
  gemm_body<EPI, TBM, TBN, NSTAGE, NWM, NWN, SPREAD, EPI != EPI_V_T>(p, smem, m0, n0);
}

template <int EPI, int TBM, int TBN, int NSTAGE, int NWM, int NWN, bool SPREAD>
hipError_t launch_cfg(const GemmParams& p, hipStream_t s) {
  if (p.N % TBN != 0) return hipErrorInvalidValue;
  constexpr int lds = NSTAGE * (TBM + TBN) * 128;
  static bool attr_set = false;
  if (lds > 65536 && !attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<EPI, TBM, TBN, NSTAGE, NWM, NWN, SPREAD>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int tiles_m = (p.M + TBM - 1) / TBM, tiles_n = p.N / TBN;
  hipLaunchKernelGGL((gemm_bf16_kernel<EPI, TBM, TBN, NSTAGE, NWM, NWN, SPREAD>), dim3(tiles_m * tiles_n),
                     dim3(64 * NWM * NWN), lds, s, p);
  return hipGetLastError();
}

template <int EPI>
hipError_t dispatch(const GemmParams& p, int variant, hipStream_t s) {
  if (variant == 0) {
    // measured on MI355X at M = 3840 (tools/kbench.py): wide GEMMs (QK, FF1; N >= 2048) are fastest with 256x128 tiles,
    // 8 waves, 3-stage ring (one round of 240 tiles); the N = 1024 GEMMs (V, out-proj, FF2) with 128x128 tiles, 8 waves
    variant = p.N >= 2048 ? 6 : 10;
  }
  switch (variant) {
    //                              BM   BN  ST WM WN spread
    case 2: return launch_cfg<EPI, 128, 128, 2, 2, 2, false>(p, s);
    case 3: return launch_cfg<EPI, 128, 128, 3, 2, 2, true>(p, s);
    case 4: return launch_cfg<EPI, 128, 64, 3, 2, 2, true>(p, s);
    case 5: return launch_cfg<EPI, 128, 64, 2, 2, 2, false>(p, s);
    case 6: return launch_cfg<EPI, 256, 128, 3, 4, 2, true>(p, s);
    case 7: return launch_cfg<EPI, 256, 128, 2, 4, 2, false>(p, s);
    case 10: return launch_cfg<EPI, 128, 128, 3, 2, 4, true>(p, s);
    case 11: return launch_cfg<EPI, 128, 128, 3, 2, 2, false>(p, s);
    case 12: return launch_cfg<EPI, 256, 128, 3, 4, 2, false>(p, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

hipError_t launch_gemm_bf16_variant(int epi, const GemmParams& p, int variant, hipStream_t s) {
  if (p.K % BK != 0 || p.N % 128 != 0 || p.M <= 0 || p.seq_pitch <= 0) return hipErrorInvalidValue;
  switch (epi) {
    case EPI_BIAS_BF16: return dispatch<EPI_BIAS_BF16>(p, variant, s);
    case EPI_BIAS_GELU_BF16: return dispatch<EPI_BIAS_GELU_BF16>(p, variant, s);
    case EPI_BIAS_F32: return dispatch<EPI_BIAS_F32>(p, variant, s);
    case EPI_GATE_RES: return dispatch<EPI_GATE_RES>(p, variant, s);
    case EPI_QK_ROPE: return dispatch<EPI_QK_ROPE>(p, variant, s);
    case EPI_V_T: return dispatch<EPI_V_T>(p, variant, s);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_gemm_bf16(int epi, const GemmParams& p, hipStream_t s) { return launch_gemm_bf16_variant(epi, p, 0, s); }
