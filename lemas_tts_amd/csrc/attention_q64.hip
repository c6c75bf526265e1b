// Attention, 64 queries per wave (gfx950): the round-4 form of attention.hip's split-KV kernel for the step loop.
//
// Same contract, layouts and arithmetic as attn_fwd_splitkv_kernel<19> (reference: lemas_tts/model/modules.py:483-491; q arrives
// multiplied by softmax_scale * log2(e) from the QK GEMM epilogue, P = exp2(S) with no running max, one range check per workgroup and a
// two-pass softmax for a workgroup that trips it).  What changes is the work split:
//
//   attention.hip      workgroup = 128 queries, 8 waves = 2 key-parity groups x 4 sub-blocks of 32 queries, <= 128 VGPRs, 4 waves / SIMD
//   here               workgroup = 256 queries, 8 waves = 2 key-parity groups x 4 sub-blocks of 64 queries, <= 256 VGPRs, 2 waves / SIMD
//
// A wave owns TWO 32-query blocks (A, B) and feeds both from every K / V^T fragment it reads, so per MFMA the kernel reads half the LDS
// bytes and moves half the K / V^T bytes from L2 (256 queries share a tile instead of 128), and a wave always has an independent
// chain at hand: the exponentials of one block issue under the matrix instructions of the other.  At configs[1] a lane launch is
// 8 x 16 = 128 workgroups, the two CFG lanes together 256 = one per CU in ONE round (the 128-query kernel: 480 workgroups, two per CU).
// The ring is three stages of tile PAIRS (96 KB), filled two pairs ahead with a counted vmcnt: a pair's DMA has two pairs of compute
// to land in instead of one.
#include <type_traits>

#include "common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int QB2 = 256;           // queries per workgroup
constexpr int KB = 64;             // keys per tile
constexpr int TILE = KB * 64 * 2;  // 8 KiB per operand tile
constexpr int STAGE = 4 * TILE;    // [K0 | V0^T | K1 | V1^T]
constexpr int NSTG = 3;
constexpr int XCH_F = 68;          // floats a lane parks in the merge: 2 x (m, l, 32 x O^T)
constexpr int LDS_BYTES = NSTG * STAGE;

__device__ __forceinline__ int xcd_block_id() {
  const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int OFF>
__device__ __forceinline__ void lds_read_b128(u32x4& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(OFF) : "memory");
}
// both 32-row halves of one k-step of a K tile (attention.hip): rows r and r + 32 share the swizzle, the step flips bits 5-6
template <int X>
__device__ __forceinline__ void lds_read_kstep(u32x4& d0, u32x4& d1, unsigned base) {
  unsigned t;
  asm volatile("v_xor_b32 %2, %4, %3\n\tds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096"
               : "=&v"(d0), "=&v"(d1), "=&v"(t) : "v"(base), "n"(X) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_lgkm_frags(u32x4& a, u32x4& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}

// SV (bit mask): 1 = blocks A and B run SKEWED (S_A | S_B + exp A | P.V_A + exp B | P.V_B; every fragment of the tile read once into
//                    registers and used by both blocks) instead of phase by phase (S_A+S_B | exp A, exp B | P.V_A+P.V_B)
//                2 = static priority for the younger half of the workgroup (waves 4-7)
template <int SV>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_fwd_q64_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wq = wave & 3;     // key-tile parity, query sub-block of 64
  const int l31 = lane & 31, hi = lane >> 5;
  const int nqb = (p.n + QB2 - 1) / QB2;
  const int lid = xcd_block_id();
  const int bh = lid / nqb, qblk = lid - bh * nqb;
  const int b2 = bh / p.heads, h = bh - b2 * p.heads;
  const int N = p.n;
  // ragged batch: both 128-row blocks of this 256-query block are padding nobody reads, and with skip_masked their q rows were never
  // written -- return as attention.hip does for its 128-query blocks (a half-live block runs: its dead half's rows are finite garbage
  // that no live row depends on, stored to rows nobody reads)
  if (p.live_len && qblk * QB2 >= ((p.live_len[b2 % p.batch] + 127) & ~127)) return;
  const int kvlen = p.kv_len ? p.kv_len[b2 % p.batch] : N;
  const int ntiles = (kvlen + KB - 1) / KB, nsup = (ntiles + 1) >> 1;
  const char* kg = reinterpret_cast<const char*>(p.k + (size_t)bh * p.pitch * 64);
  const char* vg = reinterpret_cast<const char*>(p.vt + (size_t)bh * 64 * p.npad);
  const int q_base = qblk * QB2 + wq * 64;

  // DMA: wave w moves piece w (rows 8w..8w+7) of K0, V0^T, K1, V1^T of every tile pair (four 1-KiB instructions per wave and pair)
  unsigned koff, voff;
  {
    const int r = 8 * wave + (lane >> 3), lp = lane & 7;
    const int cc = (lp ^ ((r >> 1) & 7)) << 3;
    koff = (unsigned)((r * 64 + cc) * 2);
    voff = (unsigned)((r * p.npad + cc) * 2);
  }
  auto issue = [&](int stage, int i) __attribute__((always_inline)) {
    char* base = smem + stage * STAGE + wave * 1024;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      int j = 2 * i + g;
      j = j < ntiles ? j : ntiles - 1;   // odd tile count: the missing tile aliases the last one (never consumed)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kg + (size_t)j * (KB * 64 * 2) + koff),
                                       (__attribute__((address_space(3))) void*)(base + g * 2 * TILE), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vg + (size_t)j * (KB * 2) + voff),
                                       (__attribute__((address_space(3))) void*)(base + g * 2 * TILE + TILE), 16, 0, 0);
    }
  };

  // fragment addresses: as attention.hip (S^T rows are keys permuted so that 8 consecutive accumulator registers are 8 consecutive keys)
  const int krow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
  const int ksw = (krow >> 1) & 7, vsw = (l31 >> 1) & 7;
  const unsigned kx0 = krow * 128 + ((hi ^ ksw) << 4);   // k-step kk: kx0 ^ (kk << 5)
  const unsigned vx0 = l31 * 128 + ((hi ^ vsw) << 4);    // 16-key step e: vx0 ^ (e << 5)

  bf16x8 qf[2][4];
  {
    const bf16_t* Qg = p.q + (size_t)bh * p.pitch * 64;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      int qrow = q_base + 32 * qb + l31;
      qrow = qrow < N ? qrow : N - 1;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) qf[qb][kk] = *reinterpret_cast<const bf16x8*>(Qg + (size_t)qrow * 64 + kk * 16 + hi * 8);
    }
  }
  f32x16 o[2][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[qb][0][r] = 0.f; o[qb][1][r] = 0.f; }
  float m_run[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f};
  if constexpr ((SV & 2) != 0) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }

  // pair i of the ring: its DMA has landed for every wave (counted: the pair after it may still be in flight), pair i + 2 is requested
  auto sync = [&](int i, int sg) __attribute__((always_inline)) {
    if (i + 1 < nsup) wait_vmcnt<4>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    int s2 = sg + 2;
    s2 = s2 >= NSTG ? s2 - NSTG : s2;
    if (i + 2 < nsup) issue(s2, i + 2);       // that stage held pair i - 1: every wave finished it before this barrier
  };

  // MODE: 1 fast (P = exp2(S)) | 2 max-only sweep (m_run = running row maximum, nothing accumulated) | 3 fast, P = exp2(S - m_run)
  auto compute = [&](int i, auto sg_c, auto mode_c) __attribute__((always_inline)) {
    const int SG = sg_c;
    constexpr int MODE = decltype(mode_c)::value;
    constexpr bool SUB = MODE == 3;
    const int j = 2 * i + grp;
    if (j >= ntiles) return;
    const unsigned ka = lds_base + SG * STAGE + grp * 2 * TILE + kx0;
    const unsigned va = lds_base + SG * STAGE + grp * 2 * TILE + TILE + vx0;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool last = __builtin_expect((j + 1) * KB > kvlen, 0);
    f32x16 s[2][2];
    auto mask = [&](int qb) __attribute__((always_inline)) {
      if (last) {
        asm volatile("" ::: "memory");      // a real, rarely taken branch (attention.hip)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = j * KB + 32 * t + 16 * (r >> 3) + 8 * hi + (r & 7);
            if (key >= kvlen) s[qb][t][r] = -INFINITY;
          }
      }
    };
    // keys 16 e .. 16 e + 15 of block qb -> one bf16 B-operand fragment; adds this lane's share of the row sum
    bf16x8 pb[2][4];
    f32x2 ps[2] = {{0.f, 0.f}, {0.f, 0.f}};
    auto pchunk = [&](int qb, int e) __attribute__((always_inline)) {
      const f32x2 m2 = {m_run[qb], m_run[qb]};
#pragma unroll
      for (int r = 0; r < 8; r += 2) {
        const int q = 8 * (e & 1) + r;
        f32x2 a = {s[qb][e >> 1][q], s[qb][e >> 1][q + 1]};
        if constexpr (SUB) a = a - m2;
        f32x2 pv = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
        ps[qb] += pv;
        pb[qb][e][r] = (bf16_t)pv[0];
        pb[qb][e][r + 1] = (bf16_t)pv[1];
      }
    };
    auto rowmax = [&](int qb) __attribute__((always_inline)) {
      float mx = s[qb][0][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[qb][0][r]);
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qb][1][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      m_run[qb] = fmaxf(m_run[qb], mx);
    };

    if constexpr ((SV & 1) == 0 || MODE == 2) {
      // ---- phase by phase: every K fragment feeds both blocks as it arrives (double-buffered k-steps), then the exponentials, then P.V
      {
        u32x4 fk[2][2];
        lds_read_kstep<0>(fk[0][0], fk[0][1], ka);
        lds_read_kstep<32>(fk[1][0], fk[1][1], ka);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          u32x4 (&f)[2] = fk[kk & 1];
          if (kk < 3) wait_lgkm_frags<2>(f[0], f[1]);
          else wait_lgkm_frags<0>(f[0], f[1]);
#pragma unroll
          for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int t = 0; t < 2; ++t)
              s[qb][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[t]), qf[qb][kk], kk == 0 ? zero : s[qb][t], 0, 0, 0);
          if (kk == 0) lds_read_kstep<64>(f[0], f[1], ka);
          if (kk == 1) lds_read_kstep<96>(f[0], f[1], ka);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      mask(0); mask(1);
      if constexpr (MODE == 2) { rowmax(0); rowmax(1); return; }
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int e = 0; e < 4; ++e) pchunk(qb, e);
      const char* sV = smem + SG * STAGE + grp * 2 * TILE + TILE;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bf16x8 a[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) a[dt] = *reinterpret_cast<const bf16x8*>(sV + dt * 4096 + (vx0 ^ (unsigned)(e << 5)));
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) o[qb][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[dt], pb[qb][e], o[qb][dt], 0, 0, 0);
      }
    } else {
      // ---- skewed: all eight K fragments into registers, S_A, then S_B with the exponentials of A between its MFMAs; all eight V^T
      //      fragments into the same registers' successors, P.V_A with the exponentials of B between its MFMAs, then P.V_B
      u32x4 fk[4][2];
      lds_read_kstep<0>(fk[0][0], fk[0][1], ka);
      lds_read_kstep<32>(fk[1][0], fk[1][1], ka);
      lds_read_kstep<64>(fk[2][0], fk[2][1], ka);
      lds_read_kstep<96>(fk[3][0], fk[3][1], ka);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (kk == 0) wait_lgkm_frags<6>(fk[0][0], fk[0][1]);
        if (kk == 1) wait_lgkm_frags<4>(fk[1][0], fk[1][1]);
        if (kk == 2) wait_lgkm_frags<2>(fk[2][0], fk[2][1]);
        if (kk == 3) wait_lgkm_frags<0>(fk[3][0], fk[3][1]);
#pragma unroll
        for (int t = 0; t < 2; ++t)
          s[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fk[kk][t]), qf[0][kk], kk == 0 ? zero : s[0][t], 0, 0, 0);
      }
      mask(0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
          s[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fk[kk][t]), qf[1][kk], kk == 0 ? zero : s[1][t], 0, 0, 0);
        pchunk(0, kk);
      }
      mask(1);
      u32x4 fv[4][2];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned ad = va ^ (unsigned)(e << 5);       // the stage / tile offsets are multiples of 8 KiB: bits 5-6 belong to vx0
        lds_read_b128<0>(fv[e][0], ad);
        lds_read_b128<4096>(fv[e][1], ad);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (e == 0) wait_lgkm_frags<6>(fv[0][0], fv[0][1]);
        if (e == 1) wait_lgkm_frags<4>(fv[1][0], fv[1][1]);
        if (e == 2) wait_lgkm_frags<2>(fv[2][0], fv[2][1]);
        if (e == 3) wait_lgkm_frags<0>(fv[3][0], fv[3][1]);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          o[0][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fv[e][dt]), pb[0][e], o[0][dt], 0, 0, 0);
        pchunk(1, e);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          o[1][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fv[e][dt]), pb[1][e], o[1][dt], 0, 0, 0);
    }
    l_run[0] += ps[0][0] + ps[0][1];
    l_run[1] += ps[1][0] + ps[1][1];
  };

  using c0_t = std::integral_constant<int, 0>;
  using c1_t = std::integral_constant<int, 1>;
  using c2_t = std::integral_constant<int, 2>;
  using fast_t = std::integral_constant<int, 1>;
  using maxonly_t = std::integral_constant<int, 2>;
  using fastsub_t = std::integral_constant<int, 3>;
  // the key loop, unrolled by the ring depth: every LDS address is base register + immediate
  auto key_loop = [&](auto mode_c) __attribute__((always_inline)) {
    issue(0, 0);
    if (1 < nsup) issue(1, 1);
    for (int i = 0; i < nsup; i += 3) {
      sync(i, 0);
      compute(i, c0_t{}, mode_c);
      if (i + 1 < nsup) { sync(i + 1, 1); compute(i + 1, c1_t{}, mode_c); }
      if (i + 2 < nsup) { sync(i + 2, 2); compute(i + 2, c2_t{}, mode_c); }
    }
    __syncthreads();                                   // every wave is done with the ring
  };
  // the (rare) fallback sweeps: one instance of the tile body, ring stage as a run-time value
  auto key_loop_cold = [&](auto mode_c) __attribute__((always_inline)) {
    issue(0, 0);
    if (1 < nsup) issue(1, 1);
    int sg = 0;
#pragma clang loop unroll(disable)
    for (int i = 0; i < nsup; ++i) {
      sync(i, sg);
      compute(i, sg, mode_c);
      sg = sg + 1 == NSTG ? 0 : sg + 1;
    }
    __syncthreads();
  };
  // merge the two key-parity partials: group 1 parks (m, l, O^T) of both blocks in LDS, group 0 folds them in
  float* xch = reinterpret_cast<float*>(smem) + ((size_t)wq * 64 + lane) * XCH_F;
  auto merge = [&]() __attribute__((always_inline)) {
    if (grp == 1) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        float* x = xch + 34 * qb;
        x[0] = m_run[qb];
        x[1] = l_run[qb];
#pragma unroll
        for (int r = 0; r < 16; ++r) { x[2 + r] = o[qb][0][r]; x[18 + r] = o[qb][1][r]; }
      }
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const float* x = xch + 34 * qb;
        const float m2 = x[0], l2 = x[1];
        const float m = fmaxf(m_run[qb], m2);
        const float a1 = __builtin_amdgcn_exp2f(m_run[qb] - m), a2 = __builtin_amdgcn_exp2f(m2 - m);   // m2 = -inf (no odd tile) -> a2 = 0
        l_run[qb] = l_run[qb] * a1 + l2 * a2;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          o[qb][0][r] = o[qb][0][r] * a1 + x[2 + r] * a2;
          o[qb][1][r] = o[qb][1][r] * a1 + x[18 + r] * a2;
        }
      }
    }
  };
  auto restart = [&](float m0) __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { o[qb][0][r] = 0.f; o[qb][1][r] = 0.f; }
      m_run[qb] = m0;
      l_run[qb] = 0.f;
    }
  };
  // behind the merge's exchange area (4 x 64 x 68 floats = 69 632 B), inside the ring; written after the key loop's last barrier
  constexpr int FLAGS_OFF = 73728;
  static_assert(4 * 64 * XCH_F * 4 <= FLAGS_OFF && FLAGS_OFF + 64 <= LDS_BYTES, "flags sit between the exchange area and the end of the ring");
  int* flags = reinterpret_cast<int*>(smem + FLAGS_OFF);

  key_loop(fast_t{});
  merge();
  if (grp == 0) {                                      // range check on the merged rows (attention.hip VAR & 16)
    bool bad = false;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const float lt = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
      float t = lt;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += fabsf(o[qb][0][r]) + fabsf(o[qb][1][r]);
      bad = bad || !(t < 3.0e38f) || !(lt >= 8.0e-28f);     // 2^-90
    }
    const bool anybad = __any(bad);
    if (lane == 0) flags[wq] = anybad ? 1 : 0;
  }
  __syncthreads();                                     // xch consumed, flags visible
  if (__builtin_expect((flags[0] | flags[1] | flags[2] | flags[3]) != 0, 0)) {
    __syncthreads();                                   // every wave has read the flags before any refill DMA may overwrite them
    restart(-INFINITY);                                // two-pass softmax: every row's maximum over this wave's tiles ...
    key_loop_cold(maxonly_t{});
    key_loop_cold(fastsub_t{});                        // ... then P = exp2(S - max); the merge reconciles the two key groups' maxima
    merge();
    __syncthreads();
  }
  if (grp == 0) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      // normalise, park O as [32 q][64 d] bf16 (144-B pitch) and write whole 128-B rows
      const float inv = 1.0f / (l_run[qb] + __shfl_xor(l_run[qb], 32, 64));
      char* slab = smem + (wq * 2 + qb) * (32 * 144);
      const int qb0 = q_base + 32 * qb;
      if (p.out8) {
        // fp8 path (attention.hip): MXFP8 rows for the out-projection, [32 q][64 B] slab (80-B pitch)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          float amax = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) { o[qb][dt][r] *= inv; amax = fmaxf(amax, fabsf(o[qb][dt][r])); }
          amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
          const int e = mx_exponent(amax);
          const float sc = mx_inv_scale(e);
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<unsigned int*>(slab + l31 * 80 + dt * 32 + 8 * g + 4 * hi) =
                pack_fp8x4(o[qb][dt][g * 4 + 0] * sc, o[qb][dt][g * 4 + 1] * sc, o[qb][dt][g * 4 + 2] * sc, o[qb][dt][g * 4 + 3] * sc);
          const int q = qb0 + l31;
          if (hi == 0 && q < N) p.out_mx[((size_t)b2 * p.pitch + q) * (p.heads * 2) + h * 2 + dt] = (uint8_t)(e + 127);
        }
        const int rr = lane >> 2, ch = lane & 3;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int q = qb0 + it * 16 + rr;
          const u32x4 d = *reinterpret_cast<const u32x4*>(slab + (it * 16 + rr) * 80 + ch * 16);
          if (q < N) store_wt_b128(p.out8 + ((size_t)b2 * p.pitch + q) * (p.heads * 64) + h * 64 + ch * 16, d);
        }
      } else {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            bf16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (bf16_t)(o[qb][dt][g * 4 + e] * inv);
            *reinterpret_cast<bf16x4*>(slab + l31 * 144 + (dt * 32 + 8 * g + 4 * hi) * 2) = v;
          }
        const int rr = lane >> 3, ch = lane & 7;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int q = qb0 + it * 8 + rr;
          const u32x4 d = *reinterpret_cast<const u32x4*>(slab + (it * 8 + rr) * 144 + ch * 16);
          if (q < N) store_wt_b128(p.out + ((size_t)b2 * p.pitch + q) * (p.heads * 64) + h * 64 + ch * 8, d);
        }
      }
    }
  }
}

template <int SV>
hipError_t launch_sv(const AttnParams& p, hipStream_t s) {
  const dim3 grid(((p.n + QB2 - 1) / QB2) * p.b2 * p.heads);
  if (p.ev_start) hipExtLaunchKernelGGL(attn_fwd_q64_kernel<SV>, grid, dim3(512), LDS_BYTES, s, p.ev_start, p.ev_stop, 0, p);
  else hipLaunchKernelGGL(attn_fwd_q64_kernel<SV>, grid, dim3(512), LDS_BYTES, s, p);
  return hipGetLastError();
}

}  // namespace

// the > 64 KB dynamic-LDS opt-in (once per device, from kernels_init(): never on a launch path)
hipError_t attention_q64_init() {
  hipError_t e;
#define LEMAS_Q64_INIT(SV) \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_q64_kernel<SV>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES)) != hipSuccess) return e;
  LEMAS_Q64_INIT(0) LEMAS_Q64_INIT(1) LEMAS_Q64_INIT(2) LEMAS_Q64_INIT(3)
#undef LEMAS_Q64_INIT
  return hipSuccess;
}

// variant = ATTN_Q64 | 16 | SV  (common.h): q prescaled, 64 queries per wave
hipError_t launch_attention_q64(const AttnParams& p, hipStream_t s) {
  if (p.npad % 64 != 0 || p.n <= 0 || p.pitch < ((p.n + 63) & ~63) || p.npad < ((p.n + 63) & ~63)) return hipErrorInvalidValue;
  if (p.out8 && !p.out_mx) return hipErrorInvalidValue;
  switch (p.variant & 3) {
    case 0: return launch_sv<0>(p, s);
    case 1: return launch_sv<1>(p, s);
    case 2: return launch_sv<2>(p, s);
    default: return launch_sv<3>(p, s);
  }
}
