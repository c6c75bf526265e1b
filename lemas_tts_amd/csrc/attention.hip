// Non-causal multi-head attention with key-padding mask, head_dim 64, flash-style online softmax (gfx950).
//
// Reference semantics: lemas_tts/model/modules.py:483-491 -- F.scaled_dot_product_attention(q, k, v,
// attn_mask = key-padding mask [B,1,1,N], is_causal=False), scale 1/sqrt(64); q,k already rotated.
//
// Layout contract (produced by the QK / V GEMM epilogues): q, k [B2, H, pitch, 64] bf16; v^T [B2, H, 64, npad] bf16
// (npad % 64 == 0); out [B2*pitch, H*64] bf16 (token-major, feeds the out-proj GEMM); pitch >= N is the per-sample
// row pitch of the activation row space (a multiple of 128).  Rows / columns past N inside the pitch hold finite
// stale values and are masked, so tiles are loaded without clamping.
//
// Work split: one workgroup = 128 queries of one (batch, head), 8 waves = 2 key-parity groups x 4 query sub-blocks of 32
// (see the kernel).  K and V^T tiles of 64 keys stream L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, wave-uniform base +
// per-lane 32-bit offset) into a 2-stage ring of tile PAIRS: one s_barrier per pair.
//
// Both matmuls are computed TRANSPOSED so that everything the softmax needs is lane-local:
//   S^T[key, q] = K . Q^T   (A = K rows from LDS, B = Q fragment held in registers)
//   O^T[d,  q] = V^T . P^T  (A = V^T rows from LDS, B = P^T built in registers from S^T)
// With v_mfma_f32_32x32x16_bf16 the C fragment has col = lane&31 (= query) and 16 rows per lane, so each lane
// owns ONE query: row max / row sum / rescale are per-lane scalars plus a single exchange with lane^32.
// The S^T row index i is mapped to key kappa(i) = i with bits 2 and 3 swapped (the A operand simply reads LDS row
// kappa(lane&31)); then the 8 accumulator registers 8s..8s+7 of a lane hold 8 CONSECUTIVE keys
// 16s + 8*(lane>>5) + 0..7, which is exactly the k-slot order of the B operand of the second MFMA -- P never
// leaves registers and V^T is read with one ds_read_b128 per MFMA.
//
// This kernel is VALU-bound, not MFMA-bound (PMC: at head_dim 64 the softmax costs 2x the matrix-pipe time), so the
// structure minimises VALU instructions per key: the pair loop is unrolled by the ring depth (every LDS address is base
// register + immediate), the exponent argument and the row sum use packed fp32 math (v_pk_fma_f32 / v_pk_add_f32),
// v_exp_f32 is issued directly, and l / O are only rescaled when some query's running max grew by more than 2^8.
//
// fp8 path (variant bit ATTN_F8QK = 8192, round 6; with bits 1 and 16): q and k arrive as MXFP8 ([B2, H, pitch, 64] e4m3 + one E8M0 scale per
// 32-wide half of a head, written by the QK GEMM epilogue, GemmParams::q8) and S^T of a 64-key tile is TWO v_mfma_scale_f32_32x32x64_f8f6f4 (64
// matrix-pipe clocks each) instead of eight v_mfma_f32_32x32x16_bf16 (32 each).  A K tile is then 4 KiB (rows of 64 B, the 16-B chunk index
// XOR-swizzled by (row >> 2) & 3 in the DMA's SOURCE address: conflict-free ds_read_b128), the pair's 128 key scales (256 B) ride along as one
// dword LDS-DMA of wave 0 into the unused half of K0's slot, and the operand layout is the fp8 GEMM body's (gemm_bf16.hip; probed on hardware,
// tools/exp/mx_probe.hip): lane (i = l & 31, h = l >> 5) holds row i, registers 0-3 = K 16h .. 16h+15, registers 4-7 = K 32+16h ..; the scale
// of K-block beta comes from lane i + 32 beta.  P . V, the softmax, the merge, the fallback sweeps and the output forms are unchanged.
// Measured (profiles/r06/r06j_*): 22.0 -> 18.1 us at N = 1875, 47.0 -> 36.9 us at N = 2814 (BH 16).
#include <type_traits>

#include "common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifdef LEMAS_PHASE_TIMESTAMPS      // measurement builds only (tools/kbench.py in a -DLEMAS_PHASE_TIMESTAMPS build); compiled out of the product
#define ATTN_STAMP(k) do { if (p.dbg && threadIdx.x == 0 && blockIdx.x < 1024) p.dbg[blockIdx.x * 4 + (k)] = wall_clock64(); } while (0)
#else
#define ATTN_STAMP(k) do { } while (0)
#endif
constexpr int QB = 128;            // queries per workgroup
constexpr int KB = 64;             // keys per tile
constexpr int TILE = KB * 64 * 2;  // 8 KiB per operand tile

// XCD-aware block order: workgroups are dispatched round-robin over the 8 XCDs (private L2s).  Consecutive LOGICAL ids
// are mapped onto one XCD so that all query blocks of a (batch, head) share that XCD's L2 copy of K and V^T; with the
// plain order the 15 query blocks of a head sit on 8 different XCDs and K/V are fetched 8 times (PMC: FETCH_SIZE 65 MB
// per launch against 23 MB algorithmic at B=1, N=1875).  Bijective for any grid size.
__device__ __forceinline__ int xcd_block_id() {
  const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Hand-scheduled LDS fragment reads (split-KV kernel): asm reads the compiler does not track, and waits with exact counts
// that are tied to the fragment registers so that the consuming MFMAs cannot be scheduled above them.
template <int OFF>
__device__ __forceinline__ void lds_read_b128(u32x4& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(OFF) : "memory");
}
// both 32-row halves of one k-step: rows r and r + 32 share the swizzle, the k-step only flips bits 5-6 of the address
// (chunk = (2 kk) ^ hi ^ swizzle), so one base register serves all four steps instead of four address registers
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

template <int OFF>
__device__ __forceinline__ void lds_read_u8(int& d, unsigned addr) {
  asm volatile("ds_read_u8 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_lgkm_frags8(u32x4& a, u32x4& b, int& c) {
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N) : "memory");
}

// q, k rows (bf16, [rows, 64]) -> MXFP8, one thread per (row, 32-wide block); blockIdx.y: 0 = q, 1 = k.  The arithmetic of the QK GEMM epilogue's
// q8 / k8 output on rows that already exist in bf16 (test library, kbench).
__global__ __launch_bounds__(256) void qk_mx8_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, uint8_t* __restrict__ q8,
                                                     uint8_t* __restrict__ k8, uint8_t* __restrict__ qs, uint8_t* __restrict__ ks, size_t rows) {
  const bf16_t* src = blockIdx.y ? k : q;
  uint8_t* dst = blockIdx.y ? k8 : q8;
  uint8_t* sc = blockIdx.y ? ks : qs;
  const size_t nb = rows * 2;
  for (size_t b = (size_t)blockIdx.x * 256 + threadIdx.x; b < nb; b += (size_t)gridDim.x * 256) {
    const bf16x8* s = reinterpret_cast<const bf16x8*>(src + b * 32);
    bf16x8 v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = s[c];
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf((float)v[c][e]));
    const int ex = mx_exponent(amax);
    const float inv = mx_inv_scale(ex);
    u32x4 o[2];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      o[c >> 1][(c & 1) * 2 + 0] = pack_fp8x4((float)v[c][0] * inv, (float)v[c][1] * inv, (float)v[c][2] * inv, (float)v[c][3] * inv);
      o[c >> 1][(c & 1) * 2 + 1] = pack_fp8x4((float)v[c][4] * inv, (float)v[c][5] * inv, (float)v[c][6] * inv, (float)v[c][7] * inv);
    }
    u32x4* d = reinterpret_cast<u32x4*>(dst + b * 32);
    d[0] = o[0];
    d[1] = o[1];
    sc[b] = (uint8_t)(ex + 127);
  }
}

// Split-KV: a workgroup owns 128 queries of one (batch, head) and runs 8 waves: waves 0-3 take the even key tiles, waves
// 4-7 the odd ones (same queries), and the two partial results (m, l, O^T) are merged through LDS at the end.  Four waves
// per SIMD (<= 128 VGPRs each) hide the long per-tile dependency chain (LDS read -> MFMA -> row max -> exchange -> exp ->
// MFMA); a 4-wave one-group kernel measured slower at every size (B=1: 53 vs 47 us; BH=256: 154 vs 136 us) and was removed.
// Ring: 2 stages of [K0 | V0^T | K1 | V1^T] (32 KiB each), one barrier per tile pair.
constexpr int STAGE2 = 4 * TILE;
constexpr int F8_SC_OFF = 4096;      // fp8 path: the pair's 256 B of key scales sit in the unused half of K0's 8-KiB slot

// VAR (bit mask), variants kept selectable for A/B measurements (AttnParams::variant; 0 = the original schedule):
//   1  max-free softmax: a wave's FIRST tile goes the classical way and fixes the running max m of each query; every later tile is
//      P = exp2(s c - m) with no max tree, no cross-half exchange, no deferral logic and no rescale of l and O (fp32 and bf16 keep
//      their RELATIVE precision at any magnitude, so P = 2^40 is as exact as P = 1 after the division by l; scores far below m
//      underflow to the zero weight they deserve).  What can go wrong is overflow -- a score more than ~127 octaves (88 nats) above
//      the first tile's maximum, or P.V beyond 3.4e38 -- and it cannot go unnoticed: l or O is then inf / NaN.  The workgroup
//      checks that once, after its key loop, and if any wave saw it the whole workgroup runs the loop again classically.  (A
//      per-tile guard on the row sum was measured first: correct, but the branch sits between the exponentials and the P.V MFMAs
//      and stops the compiler from overlapping them -- no faster than the classical kernel.)
//  16  (with 1) NO running max at all: the caller hands q already multiplied by scale * log2(e) (the QK GEMM epilogue does it in
//      fp32 before the bf16 rounding, GemmParams::q_scale) and P = exp2(S) is taken as it comes out of the MFMA -- softmax is
//      invariant to the per-query offset, so the offset only ever served the number range.  The range is checked once per workgroup
//      on the merged rows: l and O finite and l >= 2^-90 (a row whose largest weight sits near the bottom of the fp32 range would
//      lose its small weights to underflow); otherwise the workgroup goes round again as a TWO-PASS softmax: a max-only sweep over
//      the keys (QK MFMAs + max tree, nothing else) fixes every row's true maximum, then the same fast loop with P = exp2(S - max).
//      |logit| < ~60 nats never trips it.  Per tile the fast loop has 32 v_exp + 16 v_cvt_pk + the row-sum adds: no multiply-add, no
//      max tree, no exchange.  (The fallback used to be the classical online-softmax loop: correct, but its live ranges sat on the
//      fast loop's register allocation -- with the hand-scheduled P.V of bit 1024 that cost 3 us per launch; the two sweeps of the
//      fallback have the fast loop's own register shape.)
// 1024 (with 16) P.V hand-scheduled like the K side: V^T fragments are asm reads into a double buffer (the K fragments' dead registers),
//      requested two 16-key steps ahead, each step = wait for ITS two fragments (exact lgkmcnt) -> 2 MFMAs -> request step e+2 ->
//      exponentials / bf16 packing of step e+1: the VALU work of the next step issues while the matrix pipe runs this one.
//  32 / 64 / 128 / 256 / 512  ABLATIONS for measurement only (wrong results): no v_exp (one FMA instead) / no P.V MFMAs / no K,V DMA after
//      the first pair / no s_barrier in the pair hand-off / no vmcnt wait either.  What each removes is that resource's share of the loop.
// 8192 (ATTN_F8QK; with 1 and 16) q, k as MXFP8, S^T on the fp8 matrix path (see the top of the file)
//   2  static priority for the younger half of the workgroup (waves 4-7), no per-cluster flips
//   4  s_setprio 1 around the MFMA clusters
template <int VAR>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_fwd_splitkv_kernel(const AttnParams p) {
  // exactly 64 KB (the ring; the 8 range-check flags live inside it once the key loop is over), so that a workgroup of this kernel and a
  // 96 KB workgroup of the other lane's 128 x 128 GEMM fit a CU's 160 KB together (measured: no effect either way)
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE2];
  ATTN_STAMP(0);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wq = wave & 3;     // key-tile parity, query sub-block
  const int l31 = lane & 31, hi = lane >> 5;
  const int nqb = (p.n + QB - 1) / QB;
  const int lid = xcd_block_id();
  const int bh = lid / nqb, qblk = lid - bh * nqb;
  const int b2 = bh / p.heads, h = bh - b2 * p.heads;
  const int N = p.n;
  const int kvlen = p.kv_len ? p.kv_len[b2 % p.batch] : N;
  static_assert(QB == 128, "dead query blocks are the 128-row blocks of GemmParams::live_len");
  if (p.live_len && qblk * QB >= p.live_len[b2 % p.batch]) return;      // ragged batch: this block's queries are all padding nobody reads
  const int ntiles = (kvlen + KB - 1) / KB, nsup = (ntiles + 1) >> 1;
  const float c = (VAR & 16) != 0 ? 1.0f : p.scale * 1.4426950408889634f;    // VAR & 16: q arrives prescaled
  constexpr bool F8 = (VAR & ATTN_F8QK) != 0;
  // ABLATION (measurement builds, wrong results): P . V on the fp8 MFMA as well, with NO block maximum for P's scale and unit scales for V^T -- the
  // work of the cheapest conceivable fp8 P . V (half the V^T bytes, 4 instead of 8 fragment reads, 2 instead of 8 MFMAs per tile): an UPPER bound
  // on what a real one (which needs a per-(query, 32-key) maximum for P's E8M0 scale and MXFP8 v^T from the V epilogue) could gain
  constexpr bool PV8 = (VAR & 16384) != 0;
  static_assert(!PV8 || F8, "the fp8 P.V ablation sits on the fp8 QK^T path");
  static_assert(!F8 || ((VAR & 17) == 17 && (VAR & (8 | 1024)) == 0), "the fp8 QK^T path is built on the no-running-max softmax (bits 1 and 16)");
  const char* kg = F8 ? reinterpret_cast<const char*>(p.k8 + (size_t)bh * p.pitch * 64) : reinterpret_cast<const char*>(p.k + (size_t)bh * p.pitch * 64);
  const char* ksg = F8 ? reinterpret_cast<const char*>(p.k8_mx + (size_t)bh * p.pitch * 2) : nullptr;
  const char* vg = reinterpret_cast<const char*>(p.vt + (size_t)bh * 64 * p.npad);
  const int q_base = qblk * QB + wq * 32;

  // DMA: wave w moves piece w (rows 8w..8w+7) of K0, V0^T, K1, V1^T of every tile pair
  unsigned koff, voff;
  {
    const int r = 8 * wave + (lane >> 3), lp = lane & 7;
    const int cc = (lp ^ ((r >> 1) & 7)) << 3;
    koff = (unsigned)((r * 64 + cc) * 2);
    voff = (unsigned)((r * p.npad + cc) * 2);
    if constexpr (F8) {      // 4-KiB K tile: ONE 1-KiB piece (16 rows of 64 B) per wave -- waves 0-3 of K0, waves 4-7 of K1
      const int kr = 16 * (wave & 3) + (lane >> 2);
      koff = (unsigned)(kr * 64 + (((lane & 3) ^ ((kr >> 2) & 3)) << 4));
    }
  }
  auto issue = [&](int stage, int i) __attribute__((always_inline)) {
    char* base = smem + stage * STAGE2 + wave * 1024;
    if constexpr (F8) {
      if constexpr (PV8) {      // half the V^T bytes: one 1-KiB piece per wave (any finite bytes do for the timing)
        int j = 2 * i + grp;
        j = j < ntiles ? j : ntiles - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vg + (size_t)j * (KB * 2) + voff),
                                         (__attribute__((address_space(3))) void*)(smem + stage * STAGE2 + grp * 2 * TILE + TILE + (wave & 3) * 1024), 16, 0, 0);
      } else {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        int j = 2 * i + g;
        j = j < ntiles ? j : ntiles - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vg + (size_t)j * (KB * 2) + voff),
                                         (__attribute__((address_space(3))) void*)(base + g * 2 * TILE + TILE), 16, 0, 0);
      }
      }
      int j = 2 * i + grp;
      j = j < ntiles ? j : ntiles - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kg + (size_t)j * (KB * 64) + koff),
                                       (__attribute__((address_space(3))) void*)(smem + stage * STAGE2 + grp * 2 * TILE + (wave & 3) * 1024), 16, 0, 0);
      if (wave == 0)     // scales of keys 128 i .. 128 i + 127 (the row pitch is a multiple of 128: always inside the sample's rows)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ksg + (size_t)i * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(smem + stage * STAGE2 + F8_SC_OFF), 4, 0, 0);
      return;
    }
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

  const int krow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
  const int ksw = (krow >> 1) & 7, vsw = (l31 >> 1) & 7;
  const unsigned kx0 = krow * 128 + ((hi ^ ksw) << 4);   // k-step kk: kx0 ^ (kk << 5)
  // fp8 path: K fragment of 32-key block t = LDS row 32 t + krow (64-B rows), 16-B chunks h and 2 + h, physical slot = chunk ^ ((row >> 2) & 3);
  // scale byte of (key row, K-block h) at F8_SC_OFF + (64 grp + 32 t + krow) * 2 + h
  const unsigned kx8 = krow * 64 + ((hi ^ ((krow >> 2) & 3)) << 4);
  const unsigned sx8 = F8_SC_OFF + (grp * 64 + krow) * 2 + hi;
  // V^T fragment of 16-key step e: row l31, 16-B chunk (2 e + hi) ^ vsw.  2 e and hi occupy disjoint bits, so the chunk is
  // (hi ^ vsw) ^ 2 e and the byte offset vx0 ^ (e << 5): ONE register serves the four steps (the K side's trick)
  const int vx0 = l31 * 128 + ((hi ^ vsw) << 4);
  auto vx = [&](int e) __attribute__((always_inline)) { return vx0 ^ (e << 5); };

  bf16x8 qf[F8 ? 1 : 4];
  i32x8 qf8;
  int qsc = 0;
  {
    int qrow = q_base + l31;
    qrow = qrow < N ? qrow : N - 1;
    if constexpr (F8) {
      const uint8_t* Qg = p.q8 + ((size_t)bh * p.pitch + qrow) * 64;
      const u32x4 a = *reinterpret_cast<const u32x4*>(Qg + 16 * hi), b = *reinterpret_cast<const u32x4*>(Qg + 32 + 16 * hi);
      qf8[0] = a[0]; qf8[1] = a[1]; qf8[2] = a[2]; qf8[3] = a[3]; qf8[4] = b[0]; qf8[5] = b[1]; qf8[6] = b[2]; qf8[7] = b[3];
      qsc = p.q8_mx[((size_t)bh * p.pitch + qrow) * 2 + hi];
    } else {
      const bf16_t* Qg = p.q + (size_t)bh * p.pitch * 64;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(Qg + (size_t)qrow * 64 + kk * 16 + hi * 8);
    }
  }
  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;
  if constexpr ((VAR & 2) != 0) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }
  auto prio_rest = [&]() __attribute__((always_inline)) {     // priority outside the MFMA clusters (s_setprio takes an immediate)
    if ((VAR & 2) != 0 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
  };

  // the ring stage is a compile-time constant inside `pair` (the loop below is unrolled by the ring depth), so every
  // LDS address is base register + immediate
  // One pair of the ring = sync (the pair's DMA has landed for every wave; the next pair is requested) + compute (this wave's tile).
  auto sync = [&](int i, int sg) __attribute__((always_inline)) {
    if constexpr ((VAR & 512) == 0) wait_vmcnt<0>();
    if constexpr ((VAR & (256 | 512)) == 0) __builtin_amdgcn_s_barrier();
    if constexpr ((VAR & 128) != 0) { if (i >= 1) return; }
    if (i + 1 < nsup) issue(sg ^ 1, i + 1);
  };
  // sg_c: ring stage, an integral_constant inside the unrolled loops (every LDS address is then base register + immediate) or a
  // plain int.  mode_c: 0 classical online softmax | 1 fast: P = exp2(S c - m_run) with m_run fixed (see VAR & 1 / 16) | 2 max-only
  // sweep (m_run = running row maximum, nothing accumulated) | 3 = 1 for VAR & 16 but subtracting m_run (second sweep of the fallback)
  auto compute = [&](int i, auto sg_c, auto mode_c) __attribute__((always_inline)) -> bool {
    const int SG = sg_c;
    constexpr int MODE = decltype(mode_c)::value;
    constexpr bool FAST = MODE == 1 || MODE == 3;
    constexpr bool SUB = MODE == 3;
    const int j = 2 * i + grp;
    if (j >= ntiles) return true;
    const unsigned ka = lds_base + SG * STAGE2 + grp * 2 * TILE + kx0;
    f32x16 s[2];
    if constexpr (F8) {
      // S^T = K . Q^T of this tile on the fp8 matrix path: per 32-key block one scale byte + two ds_read_b128, one MFMA
      const unsigned kb8 = lds_base + SG * STAGE2 + grp * 2 * TILE, sb8 = lds_base + SG * STAGE2 + sx8;
      u32x4 f[2][2];
      int sc[2];
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      lds_read_u8<0>(sc[0], sb8);
      lds_read_b128<0>(f[0][0], kb8 + kx8);
      lds_read_b128<0>(f[0][1], kb8 + (kx8 ^ 32u));      // chunk 2 + h: bit 1 of the slot
      lds_read_u8<64>(sc[1], sb8);
      lds_read_b128<2048>(f[1][0], kb8 + kx8);
      lds_read_b128<2048>(f[1][1], kb8 + (kx8 ^ 32u));
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (t == 0) wait_lgkm_frags8<3>(f[0][0], f[0][1], sc[0]);
        else wait_lgkm_frags8<0>(f[1][0], f[1][1], sc[1]);
        i32x8 kf;
        kf[0] = f[t][0][0]; kf[1] = f[t][0][1]; kf[2] = f[t][0][2]; kf[3] = f[t][0][3];
        kf[4] = f[t][1][0]; kf[5] = f[t][1][1]; kf[6] = f[t][1][2]; kf[7] = f[t][1][3];
        s[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qf8, zero, 0, 0, 0, sc[t], 0, qsc);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (__builtin_expect((j + 1) * KB > kvlen, 0)) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = j * KB + 32 * t + 16 * (r >> 3) + 8 * hi + (r & 7);
            if (key >= kvlen) s[t][r] = -INFINITY;      // (also replaces what stale bytes of a masked key row produced, NaN included)
          }
      }
    } else {
      // S^T = K . Q^T of this tile.  K fragment schedule (2 x ds_read_b128 per k-step, double-buffered in fk[2][2]): the reads of step
      // kk+1 are in flight under the MFMAs of step kk.  Left to the compiler every step was read -> s_waitcnt lgkmcnt(0) -> MFMA.
      u32x4 fk[2][2];
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      lds_read_kstep<0>(fk[0][0], fk[0][1], ka);
      lds_read_kstep<32>(fk[1][0], fk[1][1], ka);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        u32x4 (&f)[2] = fk[kk & 1];
        if (kk < 3) wait_lgkm_frags<2>(f[0], f[1]);      // the two youngest outstanding reads are the next step's
        else wait_lgkm_frags<0>(f[0], f[1]);
        if constexpr ((VAR & 4) != 0) { if (kk == 0) __builtin_amdgcn_s_setprio(1); }
#pragma unroll
        for (int t = 0; t < 2; ++t)   // first k-step accumulates onto the inline constant 0: no register zeroing
          s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[t]), qf[kk], kk == 0 ? zero : s[t], 0, 0, 0);
        if constexpr ((VAR & 4) != 0) { if (kk == 3) prio_rest(); }
        if (kk == 0) lds_read_kstep<64>(f[0], f[1], ka);
        if (kk == 1) lds_read_kstep<96>(f[0], f[1], ka);
        __builtin_amdgcn_sched_barrier(0);   // keep wait -> MFMAs -> refill in this order (sinking the MFMAs costs a third buffer)
      }
      if (__builtin_expect((j + 1) * KB > kvlen, 0)) {
        // only the last (partial) tile of a sample masks keys.  The empty asm keeps this a real, rarely taken branch: left
        // to itself the compiler if-converts it into 32 compare/select pairs (+ the key-index adds) executed for EVERY tile
        asm volatile("" ::: "memory");
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = j * KB + 32 * t + 16 * (r >> 3) + 8 * hi + (r & 7);
            if (key >= kvlen) s[t][r] = -INFINITY;
          }
      }
    }
    if constexpr (MODE == 2) {            // max-only sweep: the row maximum over this wave's tiles, in the units of S c
      float mx = s[0][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      m_run = fmaxf(m_run, mx * c);
      return true;
    }
    const f32x2 c2 = {c, c};
    bf16x8 pb[2][2];
    // P = exp2(s c - m) as bf16 B-operand fragments; returns this lane's partial row sum (the other 32 keys sit in lane ^ 32)
    auto exps = [&](float m) __attribute__((always_inline)) -> float {
      const f32x2 m2 = {m, m};
      f32x2 ps = {0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          f32x2 e = {s[t][r], s[t][r + 1]};
          e = e * c2 - m2;
          f32x2 pv = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
          ps += pv;
          pb[t][r >> 3][r & 7] = (bf16_t)pv[0];
          pb[t][r >> 3][(r & 7) + 1] = (bf16_t)pv[1];
        }
      return ps[0] + ps[1];
    };
    if constexpr ((VAR & 8) != 0) {         // ABLATION ONLY (wrong on large scores): no running max at all
      l_run += exps(0.f);
      m_run = 0.f;
    } else if constexpr (FAST && (VAR & 16) != 0 && (VAR & 1024) != 0) {
      const unsigned va = lds_base + SG * STAGE2 + grp * 2 * TILE + TILE;
      u32x4 fv[2][2];
      auto vread = [&](int e, u32x4 (&f)[2]) __attribute__((always_inline)) {
        const unsigned ad = (va + (unsigned)vx0) ^ (unsigned)(e << 5);   // va is a multiple of 8 KiB
        lds_read_b128<0>(f[0], ad);
        lds_read_b128<4096>(f[1], ad);
      };
      vread(0, fv[0]);
      vread(1, fv[1]);
      f32x2 ps = {0.f, 0.f};
      const f32x2 m2 = {m_run, m_run};
      bf16x8 pe[2];
      auto pchunk = [&](int e, bf16x8& dst) __attribute__((always_inline)) {   // keys 16 e .. 16 e + 15: registers 8 (e & 1) .. + 7 of s[e >> 1]
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
          const int q = 8 * (e & 1) + r;
          f32x2 a = {s[e >> 1][q], s[e >> 1][q + 1]};
          if constexpr (SUB) a = a - m2;
          f32x2 pv = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
          ps += pv;
          dst[r] = (bf16_t)pv[0];
          dst[r + 1] = (bf16_t)pv[1];
        }
      };
      pchunk(0, pe[0]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        u32x4 (&f)[2] = fv[e & 1];
        if (e < 3) wait_lgkm_frags<2>(f[0], f[1]);     // the two youngest outstanding reads belong to the next step
        else wait_lgkm_frags<0>(f[0], f[1]);
        o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[0]), pe[e & 1], o[0], 0, 0, 0);
        o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[1]), pe[e & 1], o[1], 0, 0, 0);
        if (e + 2 < 4) vread(e + 2, f);
        if (e + 1 < 4) pchunk(e + 1, pe[(e + 1) & 1]);
      }
      l_run += ps[0] + ps[1];
      return true;
    } else if constexpr (FAST && PV8) {
      f32x2 ps = {0.f, 0.f};
      i32x8 pf8;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
          const f32x2 p0 = {__builtin_amdgcn_exp2f(s[t][r]), __builtin_amdgcn_exp2f(s[t][r + 1])};
          const f32x2 p1 = {__builtin_amdgcn_exp2f(s[t][r + 2]), __builtin_amdgcn_exp2f(s[t][r + 3])};
          ps += p0;
          ps += p1;
          int w = 0;
          w = __builtin_amdgcn_cvt_pk_fp8_f32(p0[0], p0[1], w, false);
          w = __builtin_amdgcn_cvt_pk_fp8_f32(p1[0], p1[1], w, true);
          pf8[t * 4 + (r >> 2)] = w;
        }
      l_run += ps[0] + ps[1];
      const char* sV8 = smem + SG * STAGE2 + grp * 2 * TILE + TILE;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int row = dt * 32 + l31;
        const u32x4 a0 = *reinterpret_cast<const u32x4*>(sV8 + row * 64 + ((hi ^ ((row >> 2) & 3)) << 4));
        const u32x4 a1 = *reinterpret_cast<const u32x4*>(sV8 + row * 64 + (((2 + hi) ^ ((row >> 2) & 3)) << 4));
        i32x8 vf;
        vf[0] = a0[0]; vf[1] = a0[1]; vf[2] = a0[2]; vf[3] = a0[3]; vf[4] = a1[0]; vf[5] = a1[1]; vf[6] = a1[2]; vf[7] = a1[3];
        o[dt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf, pf8, o[dt], 0, 0, 0, 127, 0, 127);
      }
      return true;
    } else if constexpr (FAST && (VAR & 16) != 0) {
      f32x2 ps = {0.f, 0.f};                // P = exp2(S) (second sweep of the fallback: exp2(S - m_run)): see VAR & 16
      const f32x2 m2 = {m_run, m_run};
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          f32x2 pv;
          f32x2 a = {s[t][r], s[t][r + 1]};
          if constexpr (SUB) a = a - m2;
          if constexpr ((VAR & 32) != 0) pv = f32x2{a[0] * 1.0e-3f + 1.0f, a[1] * 1.0e-3f + 1.0f};
          else pv = f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
          ps += pv;
          pb[t][r >> 3][r & 7] = (bf16_t)pv[0];
          pb[t][r >> 3][(r & 7) + 1] = (bf16_t)pv[1];
        }
      l_run += ps[0] + ps[1];
    } else if constexpr (FAST) {
      l_run += exps(m_run);
    } else {
      float mx = s[0][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float m_new = fmaxf(m_run, mx * c);
      // deferred max: while no query of the wave outgrows its running max by more than 2^8, keep the old max (P <= 256, exact
      // after the final division by l) -- no alpha, no rescale of l and O on that tile
      // (measured: 45.0 -> 43.2 us at N = 1875, +1.4 % end to end; on most tiles SOME query's max grows a little, so the plain
      // "did any max grow" test rescaled almost every time)
      const bool defer = __all(m_new - m_run <= 8.0f);
      if (defer) m_new = m_run;
      const float alpha = defer ? 1.0f : __builtin_amdgcn_exp2f(m_run - m_new);
      const bool grew = m_new > m_run;
      m_run = m_new;
      l_run = l_run * alpha + exps(m_new);
      if (__any(grew)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
      }
    }
    // P.V keeps the compiler's schedule (single-buffered V^T fragments, softmax tail interleaved with the MFMAs): a second
    // fragment buffer here lifts the kernel to 138 VGPRs and costs the fourth wave per SIMD (measured 51.8 vs 45.7 us)
    const char* sV = smem + SG * STAGE2 + grp * 2 * TILE + TILE;
    if constexpr ((VAR & 64) != 0) {       // ablation: P stays live, no V reads, no P.V MFMAs
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) asm volatile("" ::"v"(pb[t][h]));
      return true;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bf16x8 a[2];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) a[dt] = *reinterpret_cast<const bf16x8*>(sV + dt * 4096 + vx(e));
      if constexpr ((VAR & 4) != 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[dt], pb[e >> 1][e & 1], o[dt], 0, 0, 0);
      if constexpr ((VAR & 4) != 0) prio_rest();
    }
    return true;
  };
  using c0_t = std::integral_constant<int, 0>;
  using c1_t = std::integral_constant<int, 1>;
  ATTN_STAMP(1);
  // the key loop: pairs of tiles through the 2-stage ring, unrolled by the ring depth.  first_c / later_c: the mode (see compute) of
  // the wave's first tile / of every later one
  using classic_t = std::integral_constant<int, 0>;
  using fast_t = std::integral_constant<int, 1>;
  using maxonly_t = std::integral_constant<int, 2>;
  using fastsub_t = std::integral_constant<int, 3>;
  auto key_loop = [&](auto first_c, auto later_c) __attribute__((always_inline)) {
    issue(0, 0);
    sync(0, 0);
    compute(0, c0_t{}, first_c);
    for (int i = 1; i < nsup; i += 2) {
      sync(i, 1);
      compute(i, c1_t{}, later_c);
      if (i + 1 < nsup) { sync(i + 1, 0); compute(i + 1, c0_t{}, later_c); }
    }
    __syncthreads();                                   // every wave is done with the ring
  };
  // the same loop for the (rare) fallback sweeps: ONE instance of the tile body with the ring stage as a run-time value -- small code
  // that shares nothing with the unrolled loop above
  auto key_loop_cold = [&](auto mode_c) __attribute__((always_inline)) {
    issue(0, 0);
#pragma clang loop unroll(disable)
    for (int i = 0; i < nsup; ++i) {
      sync(i, i & 1);
      compute(i, i & 1, mode_c);
    }
    __syncthreads();
  };
  // merge the two key-parity partials: group 1 parks (m, l, O^T) in LDS, group 0 folds it in
  float* xch = reinterpret_cast<float*>(smem) + (size_t)wq * 64 * 36 + lane * 36;   // 34 floats used per lane, 36 pitch
  auto merge = [&]() __attribute__((always_inline)) {
    if (grp == 1) {
      xch[0] = m_run;
      xch[1] = l_run;
#pragma unroll
      for (int r = 0; r < 16; ++r) { xch[2 + r] = o[0][r]; xch[18 + r] = o[1][r]; }
    }
    __syncthreads();
    if (grp == 0) {
      const float m2 = xch[0], l2 = xch[1];
      const float m = fmaxf(m_run, m2);
      const float a1 = __builtin_amdgcn_exp2f(m_run - m), a2 = __builtin_amdgcn_exp2f(m2 - m);   // m2 = -inf (no odd tile) -> a2 = 0
      l_run = l_run * a1 + l2 * a2;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o[0][r] = o[0][r] * a1 + xch[2 + r] * a2;
        o[1][r] = o[1][r] * a1 + xch[18 + r] * a2;
      }
    }
  };
  auto restart = [&]() __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    m_run = -INFINITY;
    l_run = 0.f;
  };
  // inside the ring, behind the merge's exchange area (36 KB): written after the key loop's last barrier, when no tile is in flight; a
  // workgroup that goes round again passes a barrier first, so no wave's refill DMA can land here before every wave has read the flags
  int* flags = reinterpret_cast<int*>(smem + 40960);
  static_assert(4 * 64 * 36 * 4 <= 40960 && 40960 + 64 <= 2 * STAGE2, "flags sit between the exchange area and the end of the ring");
  if constexpr ((VAR & 17) == 17) {
    m_run = 0.f;                                       // P = exp2(S) for every tile, the first included (see VAR & 16)
    key_loop(fast_t{}, fast_t{});
    ATTN_STAMP(2);
    merge();
    if (grp == 0) {                                    // range check on the merged rows
      const float lt = l_run + __shfl_xor(l_run, 32, 64);
      float t = lt;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += fabsf(o[0][r]) + fabsf(o[1][r]);
      const bool bad = __any(!(t < 3.0e38f) || !(lt >= 8.0e-28f));     // 2^-90
      if (lane == 0) flags[wq] = bad ? 1 : 0;
    }
    __syncthreads();                                   // xch consumed, flags visible
    if (__builtin_expect((VAR & (32 | 64 | 128 | 256 | 512 | 2048 | 16384)) == 0 && (flags[0] | flags[1] | flags[2] | flags[3]) != 0, 0)) {     // (ablation builds never redo)
      __syncthreads();                                 // every wave has read the flags before any refill DMA may overwrite them
      restart();                                       // two-pass softmax: every row's maximum over this wave's tiles ...
      key_loop_cold(maxonly_t{});
      key_loop_cold(fastsub_t{});                      // ... then P = exp2(S - max); the merge below reconciles the two key groups' maxima
      merge();
      __syncthreads();
    }
  } else if constexpr ((VAR & 1) != 0) {
    key_loop(classic_t{}, fast_t{});
    ATTN_STAMP(2);
    // overflow check of the max-free pass (see VAR & 1): anything non-finite in l or O of any wave sends the WORKGROUP round again
    float t = l_run;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += fabsf(o[0][r]) + fabsf(o[1][r]);
    const bool bad = __any(!(t < 3.0e38f));
    if (lane == 0) flags[wave] = bad ? 1 : 0;
    __syncthreads();
    int redo = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) redo |= flags[w];
    if (__builtin_expect(redo != 0, 0)) {
      __syncthreads();                                 // (as above)
      restart();
      key_loop(classic_t{}, classic_t{});
    }
    merge();
    __syncthreads();   // xch fully consumed before the slabs below reuse the LDS
  } else {
    key_loop(classic_t{}, classic_t{});
    ATTN_STAMP(2);
    merge();
    __syncthreads();   // xch fully consumed before the slabs below reuse the LDS
  }
  if (grp == 0) {
    // normalise, park O as [32 q][64 d] bf16 (144-B pitch) and write whole 128-B rows
    const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32, 64));
    char* slab = smem + wq * (32 * 144);
    if (p.out8) {
      // fp8 path: the out-projection consumes MXFP8 -- each 32-wide half of the head (dt) is one scale block, held by
      // lanes l and l ^ 32; [32 q][64 B] slab (80-B pitch), whole 64-B head rows leave as 4 x 16 B
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        float amax = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[dt][r] *= inv; amax = fmaxf(amax, fabsf(o[dt][r])); }
        amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
        const int e = mx_exponent(amax);
        const float sc = mx_inv_scale(e);
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<unsigned int*>(slab + l31 * 80 + dt * 32 + 8 * g + 4 * hi) =
              pack_fp8x4(o[dt][g * 4 + 0] * sc, o[dt][g * 4 + 1] * sc, o[dt][g * 4 + 2] * sc, o[dt][g * 4 + 3] * sc);
        const int q = q_base + l31;
        if (hi == 0 && q < N) p.out_mx[((size_t)b2 * p.pitch + q) * (p.heads * 2) + h * 2 + dt] = (uint8_t)(e + 127);
      }
      const int rr = lane >> 2, ch = lane & 3;
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int q = q_base + it * 16 + rr;
        const u32x4 d = *reinterpret_cast<const u32x4*>(slab + (it * 16 + rr) * 80 + ch * 16);
        if (q < N) store_wt_b128(p.out8 + ((size_t)b2 * p.pitch + q) * (p.heads * 64) + h * 64 + ch * 16, d);
      }
      return;
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (bf16_t)(o[dt][g * 4 + e] * inv);
        *reinterpret_cast<bf16x4*>(slab + l31 * 144 + (dt * 32 + 8 * g + 4 * hi) * 2) = v;
      }
    const int rr = lane >> 3, ch = lane & 7;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int q = q_base + it * 8 + rr;
      const u32x4 d = *reinterpret_cast<const u32x4*>(slab + (it * 8 + rr) * 144 + ch * 16);
      if (q < N) store_wt_b128(p.out + ((size_t)b2 * p.pitch + q) * (p.heads * 64) + h * 64 + ch * 8, d);
    }
  }
#ifdef LEMAS_PHASE_TIMESTAMPS
  if (p.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); ATTN_STAMP(3); }
#endif
}

}  // namespace

bool attention_variant_ok(int v) {
  if ((v & ATTN_F8QK) != 0) return false;    // the fp8 QK^T bit is the engine's to set (option "attn_f8qk": it also needs the QK epilogue's MXFP8 output)
  if ((v & ATTN_Q64) != 0) return false;     // the 64-queries-per-wave kernel lives in the test library (lemas_k_attention), not in an engine
  switch (v) {
    case 0: case 1: case 2: case 3: case 4: case 5: case 7: case 17: case 19: case 17 + 1024: case 19 + 1024: return true;
    default: return false;
  }
}

hipError_t launch_qk_mx8(const bf16_t* q, const bf16_t* k, uint8_t* q8, uint8_t* k8, uint8_t* q8_mx, uint8_t* k8_mx, size_t rows, hipStream_t s) {
  const size_t nb = rows * 2;
  const unsigned gx = (unsigned)((nb + 255) / 256 < 4096 ? (nb + 255) / 256 : 4096);
  hipLaunchKernelGGL(qk_mx8_kernel, dim3(gx, 2), dim3(256), 0, s, q, k, q8, k8, q8_mx, k8_mx, rows);
  return hipGetLastError();
}

hipError_t launch_attention(const AttnParams& p, hipStream_t s) {
  // K rows are loaded up to the next multiple of 64 without clamping: the row pitch must cover them
  if (p.npad % 64 != 0 || p.n <= 0 || p.pitch < ((p.n + 63) & ~63) || p.npad < ((p.n + 63) & ~63)) return hipErrorInvalidValue;
  if (p.out8 && !p.out_mx) return hipErrorInvalidValue;
  // fp8 QK^T: K rows and key scales are loaded in whole 128-key pairs without clamping, so the row pitch is a multiple of 128
  if ((p.variant & ATTN_F8QK) != 0 && (!p.q8 || !p.k8 || !p.q8_mx || !p.k8_mx || p.pitch % 128 != 0)) return hipErrorInvalidValue;
  dim3 grid(((p.n + QB - 1) / QB) * p.b2 * p.heads);
#define LEMAS_ATTN_LAUNCH(V)                                                                                                   \
  case V:                                                                                                                      \
    if (p.ev_start) hipExtLaunchKernelGGL(attn_fwd_splitkv_kernel<V>, grid, dim3(512), 0, s, p.ev_start, p.ev_stop, 0, p);    \
    else hipLaunchKernelGGL(attn_fwd_splitkv_kernel<V>, grid, dim3(512), 0, s, p);                                             \
    break;
  if ((p.variant & ATTN_Q64) != 0) return hipErrorInvalidValue;      // attention_q64.hip is a measurement kernel: liblemas_hip_test.so only
  switch (p.variant) {
    LEMAS_ATTN_LAUNCH(0) LEMAS_ATTN_LAUNCH(1) LEMAS_ATTN_LAUNCH(2) LEMAS_ATTN_LAUNCH(3) LEMAS_ATTN_LAUNCH(4) LEMAS_ATTN_LAUNCH(5)
    LEMAS_ATTN_LAUNCH(7) LEMAS_ATTN_LAUNCH(17) LEMAS_ATTN_LAUNCH(19) LEMAS_ATTN_LAUNCH(17 + 1024) LEMAS_ATTN_LAUNCH(19 + 1024)
    LEMAS_ATTN_LAUNCH(17 + ATTN_F8QK) LEMAS_ATTN_LAUNCH(19 + ATTN_F8QK)
#ifdef LEMAS_PHASE_TIMESTAMPS      // measurement builds only: ablations (wrong results by construction) and the forms without a fallback pass
    LEMAS_ATTN_LAUNCH(8) LEMAS_ATTN_LAUNCH(10)
    LEMAS_ATTN_LAUNCH(17 + 32) LEMAS_ATTN_LAUNCH(17 + 64) LEMAS_ATTN_LAUNCH(17 + 128) LEMAS_ATTN_LAUNCH(17 + 256) LEMAS_ATTN_LAUNCH(17 + 256 + 512)
    LEMAS_ATTN_LAUNCH(17 + 128 + 256 + 512) LEMAS_ATTN_LAUNCH(19 + 2048) LEMAS_ATTN_LAUNCH(19 + 1024 + 2048)
    LEMAS_ATTN_LAUNCH(19 + ATTN_F8QK + 16384)
#endif
    default: return hipErrorInvalidValue;
  }
#undef LEMAS_ATTN_LAUNCH
  return hipGetLastError();
}
