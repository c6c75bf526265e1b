// bf16 x bf16 -> fp32 GEMM on v_mfma_f32_32x32x16_bf16 with fused epilogues (gfx950).
//
//   C[M, N] = A[M, K] . W[N, K]^T          (nn.Linear layout: both operands K-contiguous)
//
// This is the workhorse of the DiT step loop: QKV (+bias +RoPE, scatter to head-major q/k and v^T),
// attention out-proj and FF2 (+bias, x += gate * .), FF1 (+bias, GELU-tanh), proj_out (+bias -> fp32).
// Reference semantics: lemas_tts/model/modules.py:452-461,470-480 (QKV + RoPE), :495,:635 (out-proj + gated
// residual), :349-350,:638-639 (FF), backbones/dit.py:252 (proj_out).
//
// Structure: BM x BN x 64 block tile, 4 or 8 waves as NWM(M) x NWN(N); operand tiles stream HBM/L2 -> LDS with
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
#include <mutex>
#include <type_traits>

#include "common.h"
#include "ln_core.h"

namespace {

constexpr int BK = 64;
constexpr int LN_NP = LN_D / 32;      // ln fold: 32-column statistics slots per row (GemmParams::ln_np)

// Measurement builds only (-DLEMAS_PHASE_TIMESTAMPS, see profiles/r02/r02_kbench_phases.txt): thread 0 of every workgroup stamps the 100 MHz
// wall clock at entry, after the prologue, after the K loop and after its last store has drained.  Compiled out of the product.
#ifdef LEMAS_PHASE_TIMESTAMPS
// a timeline slot holds 1024 workgroups x 4 stamps (engine_dit.hip TL_SLOT): larger grids stamp their first 1024 workgroups only
#define PHASE_STAMP(k) do { if (p.dbg && threadIdx.x == 0 && blockIdx.x < 1024) p.dbg[blockIdx.x * 4 + (k)] = wall_clock64(); } while (0)
#define PHASE_STAMP_END() do { if (p.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); PHASE_STAMP(3); } } while (0)
#else
#define PHASE_STAMP(k) do { } while (0)
#define PHASE_STAMP_END() do { } while (0)
#endif

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// XCD-aware tile order.  Workgroups are dispatched round-robin over the 8 XCDs (private, non-coherent L2s), so workgroup `bid` of
// `nwg` runs on XCD bid & 7.  Step 1 (bijective for any grid size): give every XCD a CONTIGUOUS run of the logical sequence.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
// Step 2: the logical sequence walks the tile grid BLOCK by block, the grid cut into gx x gy blocks (gx * gy = 8), so one XCD's run
// is (about) one block: it fetches 1/gx of the A panels and 1/gy of the W panels.  With the plain row-major sequence (gx = 8,
// gy = 1) every XCD streamed ALL of W: PMC FETCH 45.8 MB for the QK GEMM against 15.8 MB algorithmic (profiles/r01/r01o_traffic.json).
// The host picks (gx, gy) to minimise gy * |A| + gx * |W| (GemmParams::xcd_gx).  Ragged grids: blocks at the edges are smaller.
// Round 4: the run of an XCD and its block are the SAME thing.  xcd_remap hands every XCD an equal share of the sequence, but the blocks of a
// ragged grid are not equal (15 x 8 tiles cut 4 x 2: blocks of 16, 16, 16, 16, 16, 16, 12, 12), so an XCD's run started inside one block and
// ended in the next and fetched both blocks' panels (out-projection / FF2 at configs[1]: 41.9 MB of fabric reads per launch against 31.3 MB
// for aligned blocks).  Here workgroup `bid` belongs to XCD bid & 7 = block bid & 7 and takes tile (bid >> 3) of that block; the launch holds
// 8 x (largest block) workgroups and the few beyond a smaller block's end return at once (xcd_grid / xcd_tile_coords).
// (xcd_grid / xcd_tile_coords live in common.h: the fp32 GEMM uses the same mapping)
// tile of workgroup `bid`: one block per XCD (default) or, for A/B measurements, round 3's equal runs (GemmParams::xcd_runs); false = surplus workgroup
// Request the operand pointers together with the first kernel arguments: left alone, the compiler loads them (s_load from the kernarg segment)
// where they are first used, i.e. AFTER the tile coordinates are known, and a second scalar-memory round trip sits in front of the first
// operand request of every workgroup.
__device__ __forceinline__ void kernargs_early(const GemmParams& p) {
  asm volatile("" ::"s"(p.A), "s"(p.W), "s"(p.K), "s"(p.M), "s"(p.N), "s"(p.live_len), "s"(p.seq_pitch), "s"(p.batch));
  // (live_len / seq_pitch / batch: tile_dead stands before the first operand request; left to the compiler, the load of the pitch was hoisted
  // above the test of live_len and its wait sat in front of every workgroup's first request, ragged batch or not: -1.3 % on configs[1])
}
__device__ __forceinline__ void tile_coords(int seq, int tiles_m, int tiles_n, int gx, int& tm, int& tn);
__device__ __forceinline__ bool tile_of(int bid, int tiles_m, int tiles_n, int gx, int runs, int& tm, int& tn) {
#ifndef LEMAS_MEASUREMENT_BUILD
  runs = 0;      // round 3's order is an A/B arm of measurement builds (engine option "xcd_runs")
#endif
  if (runs == 1) {      // (2 = the tail-skip ablation of the ping-pong launches, see tail_skip_grid: the default order)
    const int nwg = tiles_m * tiles_n;
    if (bid >= nwg) return false;
    tile_coords(xcd_remap(bid, nwg), tiles_m, tiles_n, gx, tm, tn);
    return true;
  }
  return xcd_tile_coords(bid, tiles_m, tiles_n, gx, tm, tn);
}
// ragged batches (GemmParams::live_len): a tile is dead when every 128-row block it covers lies in some sample's padding
__device__ __forceinline__ bool tile_dead(const GemmParams& p, int m0, int tbm) {
  if (!p.live_len) return false;
  for (int r = m0 & ~127; r < m0 + tbm && r < p.M; r += 128)
    if (!row_block_dead(p.live_len, r, p.seq_pitch, p.batch)) return false;
  return true;
}
static inline int grid_of(int tiles_m, int tiles_n, int gx, int runs) {
#ifndef LEMAS_MEASUREMENT_BUILD
  runs = 0;
#endif
  return runs == 1 ? tiles_m * tiles_n : xcd_grid(tiles_m, tiles_n, gx);
}
// ABLATION, measurement builds only (engine option xcd_runs = 2; WRONG RESULTS by construction): a multi-round ping-pong launch drops the workgroups
// of its partial last round, i.e. the work a stream-K / tail-splitting scheme would have to redistribute simply vanishes.  What that buys end to end is
// an UPPER bound on what any such scheme can buy in situ (round-5 review, item 2c: measured instead of argued).
static inline int tail_skip_grid(const GemmParams& p, int grid) {
#ifdef LEMAS_MEASUREMENT_BUILD
  if (p.xcd_runs == 2 && grid > 256) return grid / 256 * 256;
#endif
  (void)p;
  return grid;
}
__device__ __forceinline__ void tile_coords(int seq, int tiles_m, int tiles_n, int gx, int& tm, int& tn) {
  const int gy = 8 / gx;
  const int bm = (tiles_m + gx - 1) / gx, bn = (tiles_n + gy - 1) / gy;
  tm = tn = 0;
  for (int bi = 0; bi < gx; ++bi) {
    const int rows = min(bm, tiles_m - bi * bm);
    if (rows <= 0) break;
    for (int bj = 0; bj < gy; ++bj) {
      const int cols = min(bn, tiles_n - bj * bn);
      if (cols <= 0) break;
      const int cnt = rows * cols;
      if (seq < cnt) { tm = bi * bm + seq / cols; tn = bj * bn + seq % cols; return; }
      seq -= cnt;
    }
  }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int N>
__device__ __forceinline__ void wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

// LDS reads the compiler does not track (hand-scheduled K loop): the data is only valid after an explicit wait_lgkmcnt
template <int OFF>
__device__ __forceinline__ void lds_read_b128(u32x4& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read_b32(int& d, unsigned addr) {
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(OFF) : "memory");
}
// T .. N-1 fragments of one operand: row offset 32 t -> +4096 t bytes (the XOR swizzle depends on (row >> 1) & 7 only)
template <int T, int N, int RPF>
struct ReadFrags {
  static __device__ __forceinline__ void run(u32x4 (*d)[RPF], const unsigned (&addr)[RPF], unsigned base) {
    if constexpr (T < N) {
#pragma unroll
      for (int h = 0; h < RPF; ++h) lds_read_b128<T * 4096>(d[T][h], base + addr[h]);
      ReadFrags<T + 1, N, RPF>::run(d, addr, base);
    }
  }
};
template <int T, int N>
struct ReadScales {
  static __device__ __forceinline__ void run(int* d, unsigned addr) {
    if constexpr (T < N) {
      lds_read_b32<T * 128>(d[T], addr);
      ReadScales<T + 1, N>::run(d, addr);
    }
  }
};

// the value of lane perm(l) inside each row of 16 lanes (DPP control CTRL: 0xB1 = quad_perm [1,0,3,2], 0x4E = [2,3,0,1], 0x141 = row_half_mirror,
// 0x140 = row_mirror)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

__device__ __forceinline__ bf16x4 pack4(float a, float b, float c, float d) {
  bf16x4 o;
  o[0] = (bf16_t)a; o[1] = (bf16_t)b; o[2] = (bf16_t)c; o[3] = (bf16_t)d;
  return o;
}

// ---------------------------------------------------------------- row windows (epilogue addressing)
// A wave tile is at most 128 rows, aligned to its own height, and seq_pitch is a multiple of 128 (or the launch is one sample, seq_pitch >= M):
// all its rows belong to ONE sample, so the sample index, the position of its first row and the number of LIVE rows counted from that
// row are wave-uniform.  The epilogues used to derive (sample, position) per lane and per stored row -- an integer division by a run-time
// pitch, a second one for the kv_len index, a 64-bit address multiply and an exec-mask branch around every store: ~25 VALU slots per
// row, and the epilogues are VALU-issue bound (two waves per SIMD: 2400-3800 wave instructions per 128 x 64 wave tile measured as
// 9-16 us per 256 x 256 tile).  Here the two divisions happen once per wave, on uniform values, and every row access goes through a
// raw buffer descriptor whose base is the block's first row and whose size is the block's live rows: the hardware range check drops
// the dead rows (loads return 0), the byte offset is 32 bits, and no store sits behind a branch.
struct RowWin {
  int b2, pos0;      // sample (of the doubled batch) and position of the wave tile's first row
  int kvl;           // kv_len of that sample (INT_MAX: none); folded into `rl` by row_window_kv()
  int lvl;           // live_len of that sample (GemmParams::live_len; large: none); folded into `rl` by row_window_kv()
  int rl, rows_m;    // rows of the wave tile, counted from its first one, that are stored (pos < seq_valid, m < M) / that exist (m < M)
};
__device__ __forceinline__ RowWin row_window(const GemmParams& p, int mw_uniform, bool kv) {
  const int mw = __builtin_amdgcn_readfirstlane(mw_uniform);
  RowWin w;
  w.b2 = mw / p.seq_pitch;
  w.pos0 = mw - w.b2 * p.seq_pitch;
  w.kvl = 0x7fffffff;
  if (kv && p.kv_len) w.kvl = p.kv_len[w.b2 % p.batch];
  const int left = p.M - mw;
  w.rows_m = left < 0 ? 0 : left > 128 ? 128 : left;
  // ragged batch with dead blocks skipped: a half-live 256-row tile computes its dead half from stale rows -- never store them.  Like kvl the
  // value is only REQUESTED here (this runs one K-tile before the loop ends) and folded into rl after the loop (row_window_kv).
  w.lvl = p.live_len ? p.live_len[w.b2 % p.batch] : 0x7fffff00;
  const int rl = p.seq_valid - w.pos0;
  w.rl = rl < 0 ? 0 : rl > w.rows_m ? w.rows_m : rl;
  return w;
}
// (separate from row_window: the kv_len load is requested one K-tile before the loop ends and first looked at after it)
__device__ __forceinline__ void row_window_kv(RowWin& w) {
  const int lv = ((w.lvl + 127) & ~127) - w.pos0;      // rows of 128-row blocks past the sample's last live block
  int rl = w.kvl - w.pos0;
  rl = lv < rl ? lv : rl;
  w.rl = rl < 0 ? 0 : rl < w.rl ? rl : w.rl;
}
constexpr int BUF_WORD3 = 0x00020000;      // gfx950 raw buffer descriptor, dword 3: 32-bit data format, no swizzle, no stride
constexpr int BUF_OOB = 0x40000000;        // a byte offset past every window (windows are < 4 MB): the access is dropped
constexpr int BUF_SC1 = 16;                // aux: write-through, the policy of store_wt_b128 (common.h)
// `rows` rows of `row_bytes` each, starting `first_row` rows into the array at `base` (all uniform)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rows(const void* base, long long first_row, int rows, int row_bytes) {
  char* b = const_cast<char*>(static_cast<const char*>(base)) + first_row * row_bytes;
  // (the explicit readfirstlane: a clamp the compiler evaluates on the vector ALU would otherwise put the whole descriptor in VGPRs and
  // wrap every access in a waterfall loop)
  const int bytes = __builtin_amdgcn_readfirstlane(rows > 0 ? rows * row_bytes : 0);
  return __builtin_amdgcn_make_buffer_rsrc(b, 0, bytes, BUF_WORD3);
}
__device__ __forceinline__ float4 buf_load_f4(__amdgpu_buffer_rsrc_t r, int voff) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}

// ---------------------------------------------------------------- epilogues (staged through LDS)
// After the K loop the ring's LDS is free.  Every wave parks its finished sub-tile in a private LDS slab and reads it
// back row-wise, so that each global store instruction writes whole 128/256-B contiguous segments (8 or 16 lanes x
// 16 B per row).  Storing straight from the MFMA fragment layout instead touches 32 different rows per instruction
// with 8-16 B each: measured, the partial-line writes made the epilogue cost as much as the entire K = 1024 loop.
//
// swapped orientation (C^T fragments): lane -> row m = mw + 32 i + (lane&31);
//      register r -> column n = nw + 32 j + (r&3) + 8 (r>>2) + 4 (lane>>5)
template <int WTM, int WTN>
struct SlabBf16 {   // [WTM rows][WTN bf16] + 16 B pad per row
  static constexpr int PITCH = WTN * 2 + 16, BYTES = WTM * PITCH, CPR = WTN * 2 / 16, RPI = 64 / CPR, ITERS = WTM / RPI;
};
template <int WTM, int WTN>
struct SlabF32 {    // [WTM rows][WTN f32] + 16 B pad per row
  static constexpr int PITCH = WTN * 4 + 16, BYTES = WTM * PITCH, CPR = WTN * 4 / 16, RPI = 64 / CPR, ITERS = WTM / RPI;
};

template <int WTM, int WTN>
struct SlabF8 {     // [WTM rows][WTN e4m3] + 16 B pad per row
  static constexpr int PITCH = WTN + 16, BYTES = WTM * PITCH, CPR = WTN / 16, RPI = 64 / CPR, ITERS = WTM / RPI;
};

// MXFP8 output (EPI_BIAS_GELU_F8; the other row epilogues are epilogue_row_blocks below): one 32-row block.
template <int EPI, int TI, int TJ>
__device__ __forceinline__ void epilogue_rows(const GemmParams& p, f32x16 (&acc)[TI][TJ], char* slab, int mw, int nw,
                                              int lane) {
  static_assert(EPI == EPI_BIAS_GELU_F8, "the MXFP8 epilogue only");
  constexpr int WTM = 32 * TI, WTN = 32 * TJ;
  const int l31 = lane & 31, hi = lane >> 5;
  // a 32-column block of one row lives in two lanes (l, l ^ 32) x 16 registers
  using S = SlabF8<WTM, WTN>;
  const int mxld = p.ldc >> 5;
  RowWin win = row_window(p, mw, false);
  row_window_kv(win);
  const int mwu = __builtin_amdgcn_readfirstlane(mw);
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int m = mw + i * 32 + l31;
    const bool live = i * 32 + l31 < win.rl;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      float v[16];
      float amax = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bias = *reinterpret_cast<const float4*>(p.bias + nw + j * 32 + 8 * g + 4 * hi);
        const f32x2 g01 = gelu_tanh_f2(f32x2{acc[i][j][4 * g + 0] + bias.x, acc[i][j][4 * g + 1] + bias.y});
        const f32x2 g23 = gelu_tanh_f2(f32x2{acc[i][j][4 * g + 2] + bias.z, acc[i][j][4 * g + 3] + bias.w});
        v[4 * g + 0] = g01.x; v[4 * g + 1] = g01.y; v[4 * g + 2] = g23.x; v[4 * g + 3] = g23.y;
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[4 * g + 0]), fabsf(v[4 * g + 1]))), fmaxf(fabsf(v[4 * g + 2]), fabsf(v[4 * g + 3])));
      }
      amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
      const int e = mx_exponent(amax);
      const float inv = mx_inv_scale(e);
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<unsigned int*>(slab + (i * 32 + l31) * S::PITCH + j * 32 + 8 * g + 4 * hi) =
            pack_fp8x4(v[4 * g + 0] * inv, v[4 * g + 1] * inv, v[4 * g + 2] * inv, v[4 * g + 3] * inv);
      if (hi == 0 && live) p.out_mx[(size_t)m * mxld + ((nw + j * 32) >> 5)] = (uint8_t)(e + 127);
    }
  }
  const int rr = lane / S::CPR, ch = lane % S::CPR;
  const __amdgpu_buffer_rsrc_t wst = buf_rows(p.out_f8, mwu, win.rl, p.ldc);
  const int vo = rr * p.ldc + nw + ch * 16, rstep = S::RPI * p.ldc;
#pragma unroll
  for (int it = 0; it < S::ITERS; ++it) {
    const u32x4 d = *reinterpret_cast<const u32x4*>(slab + (it * S::RPI + rr) * S::PITCH + ch * 16);
    __builtin_amdgcn_raw_buffer_store_b128(d, wst, vo + it * rstep, 0, BUF_SC1);
  }
}

// The same epilogues for a wave tile of TI 32-row blocks, software-pipelined over the blocks.  Calling epilogue_rows once per
// block serialises, per block, the global loads it starts with (bias / gate vectors, the fp32 residual rows, the RoPE table rows)
// with the slab round trip behind them -- measured with in-kernel timestamps: 3.9 us of a 14.7 us out-projection launch at batch 1,
// 11-17 us per 256 x 256 tile at full chip.  Here the column vectors are loaded once per wave tile (COLS_ONCE; off where the
// registers do not allow it) and the row-dependent loads of block i+1 are in flight while block i goes through the slab.
// What a wave's row epilogue reads from global memory before it can touch its accumulators: the column vectors (bias, gate) and the
// row-dependent operands of its FIRST 32-row block (fp32 residual rows / RoPE table rows).  load() is called one K-tile before the
// end of the main loop (the operands are in registers when the loop ends); every load is unconditional with a clamped address,
// so the number of vector-memory operations in flight does not depend on the data (the loops count them with vmcnt).
// epilogues that can consume the LayerNorm-folded operand (GemmParams::ln_part): acc + bias -> r_m acc - r_m mu_m c1[n] + c2[n]
template <int EPI>
constexpr bool epi_lna() { return EPI == EPI_QK_ROPE || EPI == EPI_BIAS_GELU_BF16 || EPI == EPI_V_T; }

// LNA: 0 = the instantiation never consumes a folded LayerNorm (fp8 bodies, gate/residual, fp32 outputs); 1 = it may, c1 columns held
// in registers beside the bias; 2 = it may, c1 columns re-read per 32-row block (the 256 x 256 tile has no registers to spare)
template <int EPI, int TJ, bool COLS_ONCE, int LNA = (epi_lna<EPI>() ? 1 : 0)>
struct EpiPre {
  static constexpr int WTN = 32 * TJ;
  static constexpr int NC = COLS_ONCE ? TJ * 4 : 1;
  static constexpr int NR = EPI == EPI_GATE_RES ? SlabF32<32, WTN>::ITERS : EPI == EPI_QK_ROPE ? 32 / (64 / (WTN / 8)) : 1;
  static_assert(LNA == 0 || (epi_lna<EPI>() && COLS_ONCE), "ln-fold consumers keep their column vectors in registers");
  float4 bias_c[NC], gate_c[EPI == EPI_GATE_RES ? NC : 1];
  float4 c1_c[LNA == 1 ? NC : 1];               // ln fold: c1 columns (zero when the launch is not a ln-fold consumer)
  const float* c1src;                           // LNA == 2: where to re-read them (nullptr: not a consumer)
  float4 r0[NR], r1[EPI == EPI_QK_ROPE ? NR : 1];     // block 0: residual rows | cos rows, sin rows
  const float* gate;
  RowWin win;                                         // the wave tile's rows (uniform)

  // rows of 32-row block i of the wave tile at (mw, nw): residual / RoPE operands into (x, y)
  __device__ __forceinline__ void load_rows(const GemmParams& p, int mw, int nw, int lane, int i, float4 (&x)[NR], float4 (&y)[EPI == EPI_QK_ROPE ? NR : 1]) {
    const int mwu = __builtin_amdgcn_readfirstlane(mw);
    if constexpr (EPI == EPI_GATE_RES) {
      using S = SlabF32<32, WTN>;
      const int rr = lane / S::CPR, ch = lane % S::CPR;
      int col = nw + ch * 4;
      col = col < p.ldc - 3 ? col : 0;
      // every row that exists (m < M), live or not: the ln-fold producer re-publishes the rows it does not update
      const __amdgpu_buffer_rsrc_t w = buf_rows(p.out_f32, (long long)mwu + 32 * i, win.rows_m - 32 * i, p.ldc * 4);
      const int vo = (rr * p.ldc + col) * 4, rstep = S::RPI * p.ldc * 4;
#pragma unroll
      for (int it = 0; it < NR; ++it) x[it] = buf_load_f4(w, vo + it * rstep);
    } else if constexpr (EPI == EPI_QK_ROPE) {
      constexpr int CPR = WTN / 8, RPI = 64 / CPR;
      const int rr = lane / CPR, ch = lane % CPR;
      const int d = (nw + ch * 8) & 63;
      // table rows pos0 + 32 i ... of the live rows (the table has seq_valid of them); dead rows are rotated by 0 and never stored
      const int rows = win.rl - 32 * i;
      const __amdgpu_buffer_rsrc_t wc = buf_rows(p.rope_cos, (long long)win.pos0 + 32 * i, rows, 128);
      const __amdgpu_buffer_rsrc_t ws = buf_rows(p.rope_sin, (long long)win.pos0 + 32 * i, rows, 128);
      const int vo = (rr * 32 + (d >> 1)) * 4;
#pragma unroll
      for (int it = 0; it < NR; ++it) {
        x[it] = buf_load_f4(wc, vo + it * RPI * 128);
        y[it] = buf_load_f4(ws, vo + it * RPI * 128);
      }
    }
  }
  __device__ __forceinline__ void load(const GemmParams& p, int mw, int nw, int lane) {
    const int hi = lane >> 5;
    gate = nullptr;
    if (EPI == EPI_GATE_RES) gate = p.tab + (size_t)p.step_idx[0] * p.tab_stride + p.gate_off;
    const float* bsrc = p.bias;
    c1src = nullptr;
    if constexpr (LNA != 0) {
      if (p.ln_part) {       // ln fold: c2 (which holds the bias) in place of the bias vector, c1 beside it
        const float* row = p.tab + (size_t)p.step_idx[0] * p.tab_stride;
        bsrc = row + p.lnc2_off;
        c1src = row + p.lnc1_off;
      }
    }
    if (COLS_ONCE) {
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = j * 32 + 8 * g + 4 * hi;
          bias_c[j * 4 + g] = *reinterpret_cast<const float4*>(bsrc + nw + nl);
          if (EPI == EPI_GATE_RES) gate_c[j * 4 + g] = *reinterpret_cast<const float4*>(gate + nw + nl);
          if constexpr (LNA == 1)
            c1_c[j * 4 + g] = c1src ? *reinterpret_cast<const float4*>(c1src + nw + nl) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    win = row_window(p, mw, EPI == EPI_GATE_RES);
    load_rows(p, mw, nw, lane, 0, r0, r1);
  }
};

// The same epilogues for a wave tile of TI 32-row blocks, software-pipelined over the blocks.  Calling epilogue_rows once per
// block serialises, per block, the global loads it starts with (bias / gate vectors, the fp32 residual rows, the RoPE table rows)
// with the slab round trip behind them -- measured with in-kernel timestamps: 3.9 us of a 14.7 us out-projection launch at batch 1,
// 11-17 us per 256 x 256 tile at full chip.  Here the column vectors are loaded once per wave tile (COLS_ONCE; off where the
// registers do not allow it), block 0's row operands arrive with `pre`, and those of block i+1 are in flight while block i goes
// through the slab.
// `lnx`: ln fold (GemmParams), consumer epilogues: the (r, -r mu) pairs of this wave tile's rows in LDS ([row][2], written by
// ln_rowstat_store at the top of the kernel).  The producer side (EPI_GATE_RES with xs_out) needs nothing from the caller.
// QK8 (EPI_QK_ROPE in the fp8 bodies only): the instantiation can write q / k as MXFP8 (GemmParams::q8); the bf16 kernels do not carry the code
// (it cost the 256 x 128 and 256 x 256 QK kernels 16-36 B of scratch each)
template <int EPI, int TI, int TJ, bool COLS_ONCE, int AHEAD = 1, int LNA = (epi_lna<EPI>() ? 1 : 0), bool QK8 = false>
__device__ __forceinline__ void epilogue_row_blocks(const GemmParams& p, f32x16 (&acc)[TI][TJ], char* slab, int mw, int nw, int lane,
                                                    EpiPre<EPI, TJ, COLS_ONCE, LNA>& pre, const float* lnx = nullptr) {
  static_assert(EPI != EPI_BIAS_GELU_F8 && EPI != EPI_V_T, "bf16-path row epilogues only");
  constexpr int WTN = 32 * TJ;
  using Pre = EpiPre<EPI, TJ, COLS_ONCE, LNA>;
  const int l31 = lane & 31, hi = lane >> 5;
  const float* gate = pre.gate;
  // ln-fold consumer: v = r acc + (nrm c1 + c2); without it r = 1, nrm = 0, c1 = 0 and the same expression is acc + bias exactly
  bool lna = false;
  if constexpr (LNA != 0) lna = p.ln_part != nullptr;
  auto c1_of = [&](int j, int g) -> float4 {
    if constexpr (LNA == 2) return pre.c1src ? *reinterpret_cast<const float4*>(pre.c1src + nw + j * 32 + 8 * g + 4 * hi) : make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (LNA == 1) return pre.c1_c[j * 4 + g];
    return make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto row_rn = [&](int i, float& r, float& nrm) {
    r = 1.0f; nrm = 0.f;
    if (lna) { const float2 t = *reinterpret_cast<const float2*>(lnx + 2 * (32 * i + l31)); r = t.x; nrm = t.y; }
  };
  auto bias_of = [&](int j, int g) -> float4 {
    if (COLS_ONCE) return pre.bias_c[j * 4 + g];
    return *reinterpret_cast<const float4*>(p.bias + nw + j * 32 + 8 * g + 4 * hi);
  };
  auto gate_of = [&](int j, int g) -> float4 {
    if (COLS_ONCE) return pre.gate_c[EPI == EPI_GATE_RES ? j * 4 + g : 0];
    return *reinterpret_cast<const float4*>(gate + nw + j * 32 + 8 * g + 4 * hi);
  };

  const int mwu = __builtin_amdgcn_readfirstlane(mw);
  if constexpr (EPI == EPI_GATE_RES) {
    using S = SlabF32<32, WTN>;
    const int rr = lane / S::CPR, ch = lane % S::CPR;
    float4 xnext[Pre::NR], unused[1];
    row_window_kv(pre.win);
    const int rl = pre.win.rl;
    int vo = (rr * p.ldc + nw + ch * 4) * 4;
    if (nw + ch * 4 >= p.n_valid) vo = BUF_OOB;
    const int rstep = S::RPI * p.ldc * 4;
    // one block of 32 rows: gate (acc + bias) through the slab, added to the residual rows, stored through the block's window of
    // live rows.  PROD = the ln-fold producer (xs_out): scaled bf16 image + row partial sums of the new rows, every row that exists
    auto blocks = [&](auto prod_c) {
      constexpr bool PROD = decltype(prod_c)::value;
      float4 sc1p = make_float4(1.f, 1.f, 1.f, 1.f);
      if constexpr (PROD) {
        const float4 t = *reinterpret_cast<const float4*>(gate - p.gate_off + p.xs_scale_off + nw + ch * 4);
        sc1p = make_float4(1.0f + t.x, 1.0f + t.y, 1.0f + t.z, 1.0f + t.w);
      }
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        float4 xin[Pre::NR];
        // residual rows of block i + 1: requested before this block goes through the slab (AHEAD 1), or right after its accumulators
        // were parked there (AHEAD 2: the 256 x 256 tile, where the 32 registers of the next rows only exist once a block's are free)
        if (AHEAD != 0 || i == 0) {
#pragma unroll
          for (int it = 0; it < Pre::NR; ++it) xin[it] = i == 0 ? pre.r0[it] : xnext[it];
          if (AHEAD == 1 && i + 1 < TI) pre.load_rows(p, mw, nw, lane, i + 1, xnext, unused);
        } else {
          pre.load_rows(p, mw, nw, lane, i, xin, unused);
        }
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int nl = j * 32 + 8 * g + 4 * hi;
            const float4 bias = bias_of(j, g), gt = gate_of(j, g);
            *reinterpret_cast<float4*>(slab + l31 * S::PITCH + nl * 4) =
                make_float4(gt.x * (acc[i][j][4 * g + 0] + bias.x), gt.y * (acc[i][j][4 * g + 1] + bias.y),
                            gt.z * (acc[i][j][4 * g + 2] + bias.z), gt.w * (acc[i][j][4 * g + 3] + bias.w));
          }
        if (AHEAD == 2 && i + 1 < TI) pre.load_rows(p, mw, nw, lane, i + 1, xnext, unused);
        const __amdgpu_buffer_rsrc_t wst = buf_rows(p.out_f32, (long long)mwu + 32 * i, rl - 32 * i, p.ldc * 4);
        float k1 = 0.f, k2 = 0.f;      // ln fold: the (sum, sum of squares) pair this lane will publish for the block
#pragma unroll
        for (int it = 0; it < S::ITERS; ++it) {
          const float4 d = *reinterpret_cast<const float4*>(slab + (it * S::RPI + rr) * S::PITCH + ch * 16);
          float4 x = xin[it];
          if constexpr (!PROD) {
            x.x += d.x; x.y += d.y; x.z += d.z; x.w += d.w;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), wst, vo + it * rstep, 0, BUF_SC1);
          } else {
            const int row = 32 * i + it * S::RPI + rr, m = mw + row;
            if (row < rl && vo != BUF_OOB) { x.x += d.x; x.y += d.y; x.z += d.z; x.w += d.w; }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), wst, vo + it * rstep, 0, BUF_SC1);
            if (m < p.M) {     // every row of the activation space (updated or not): the consumer GEMM reads all of them
              const bf16x4 o = pack4(x.x * sc1p.x, x.y * sc1p.y, x.z * sc1p.z, x.w * sc1p.w);
              store_wt_b64(p.xs_out + (size_t)m * p.ldc + nw + ch * 4, __builtin_bit_cast(unsigned int __attribute__((ext_vector_type(2))), o));
            }
            float s1 = (x.x + x.y) + (x.z + x.w);
            float s2 = __builtin_fmaf(x.x, x.x, __builtin_fmaf(x.y, x.y, __builtin_fmaf(x.z, x.z, x.w * x.w)));
            // butterfly over the 8 lanes (32 columns = one statistics slot) of the row on DPP: v_add_f32 with a lane-permuting source
            // modifier (the __shfl_xor form is a ds_bpermute round trip per step: 2 us of a 16 us launch).  xor 1, xor 2 (quad
            // permutes), then the other quad of the 8 (row_half_mirror), whose lanes all hold their quad's sum by then
            s1 += dpp_f<0xB1>(s1); s2 += dpp_f<0xB1>(s2);
            s1 += dpp_f<0x4E>(s1); s2 += dpp_f<0x4E>(s2);
            s1 += dpp_f<0x141>(s1); s2 += dpp_f<0x141>(s2);
            // lane (rr, ch) keeps the pair of iteration it = ch % 8: after the loop ITERS * RPI * (CPR / 8) = 32 * WTN / 32 lanes hold one
            // (row, slot) pair each and ONE store instruction publishes the block's statistics
            if ((ch & 7) == it) { k1 = s1; k2 = s2; }
          }
        }
        if constexpr (PROD) {
          static_assert(S::ITERS <= 8 && (S::CPR == 8 || S::CPR == 16), "one kept pair per lane");
          const int m = mw + 32 * i + (ch & 7) * S::RPI + rr;
          if ((ch & 7) < S::ITERS && m < p.M)
            *reinterpret_cast<float2*>(p.ln_part_out + ((size_t)m * LN_NP + ((nw + ch * 4) >> 5)) * 2) = make_float2(k1, k2);
        }
      }
    };
    if (p.xs_out != nullptr) blocks(std::true_type{}); else blocks(std::false_type{});
  } else if constexpr (EPI == EPI_QK_ROPE) {
    using S = SlabF32<32, WTN>;
    constexpr int CPR = WTN / 8, RPI = 64 / CPR, ITERS = 32 / RPI;   // 8 head dims (16 B of bf16) per lane
    static_assert(ITERS == Pre::NR, "row operand count");
    static_assert(WTN <= 64, "a wave tile lies inside one head");
    const int rr = lane / CPR, ch = lane % CPR;
    const int nwu = __builtin_amdgcn_readfirstlane(nw);
    const int inner = p.heads * 64;
    const int which = nwu / inner, head = (nwu % inner) >> 6, d = (nw + ch * 8) & 63;     // uniform: q or k, the head
    const bf16_t* qk = which == 0 ? p.q : p.k;
    const long long row0 = ((long long)pre.win.b2 * p.heads + head) * p.seq_pitch + pre.win.pos0;    // [b2][head][pos][64]
    row_window_kv(pre.win);
    const int rl = pre.win.rl, vo = (rr * 64 + d) * 2;
    const float qs = (which == 0 && p.q_scale != 0.f) ? p.q_scale : 1.0f;
    // MXFP8 output (GemmParams::q8): the same bf16-rounded values, quantised per 32-wide half of the head = the four lanes of this lane's quad
    uint8_t* qk8 = which == 0 ? p.q8 : p.k8;
    uint8_t* qk8mx = which == 0 ? p.q8_mx : p.k8_mx;
    const bool f8out = QK8 && p.q8 != nullptr;
    const int vo8 = rr * 64 + d, vomx = (lane & 3) == 0 ? rr * 2 + (d >> 5) : BUF_OOB;
    float4 cnext[ITERS], snext[ITERS];
    auto blocks = [&](auto lna_c) {      // two copies of the unrolled blocks: with and without the folded LayerNorm (a run-time property of the launch)
    constexpr bool L = LNA != 0 && decltype(lna_c)::value;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const __amdgpu_buffer_rsrc_t wst = buf_rows(qk, row0 + 32 * i, rl - 32 * i, 128);
      float4 cs[ITERS], sn[ITERS];
#pragma unroll
      for (int it = 0; it < ITERS; ++it) { cs[it] = i == 0 ? pre.r0[it] : cnext[it]; sn[it] = i == 0 ? pre.r1[it] : snext[it]; }
      // the RoPE rows of block i + 1: requested here, or (wave tiles of four blocks: 128 accumulator registers) once this block's
      // accumulators are parked in the slab and their registers are free
      constexpr bool LATE = false;      // (tried for four-block wave tiles: the compiler spills MORE, 148 vs 124 B of scratch in the 256 x 256 QK kernel)
      if (!LATE && i + 1 < TI) pre.load_rows(p, mw, nw, lane, i + 1, cnext, snext);
      float r = 1.0f, nrm = 0.f;
      if constexpr (L) row_rn(i, r, nrm);
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = j * 32 + 8 * g + 4 * hi;
          const float4 bias = bias_of(j, g);
          float4 v;
          if constexpr (L) {
            const float4 c1 = c1_of(j, g);
            v = make_float4(__builtin_fmaf(r, acc[i][j][4 * g + 0], __builtin_fmaf(nrm, c1.x, bias.x)),
                            __builtin_fmaf(r, acc[i][j][4 * g + 1], __builtin_fmaf(nrm, c1.y, bias.y)),
                            __builtin_fmaf(r, acc[i][j][4 * g + 2], __builtin_fmaf(nrm, c1.z, bias.z)),
                            __builtin_fmaf(r, acc[i][j][4 * g + 3], __builtin_fmaf(nrm, c1.w, bias.w)));
          } else {
            v = make_float4(acc[i][j][4 * g + 0] + bias.x, acc[i][j][4 * g + 1] + bias.y, acc[i][j][4 * g + 2] + bias.z,
                            acc[i][j][4 * g + 3] + bias.w);
          }
          *reinterpret_cast<float4*>(slab + l31 * S::PITCH + nl * 4) = v;
        }
      if (LATE && i + 1 < TI) pre.load_rows(p, mw, nw, lane, i + 1, cnext, snext);
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const float4 c = cs[it], sv = sn[it];
        const float4 a = *reinterpret_cast<const float4*>(slab + (it * RPI + rr) * S::PITCH + ch * 32);
        const float4 b = *reinterpret_cast<const float4*>(slab + (it * RPI + rr) * S::PITCH + ch * 32 + 16);
        bf16x8 o;
        o[0] = (bf16_t)((a.x * c.x - a.y * sv.x) * qs); o[1] = (bf16_t)((a.y * c.x + a.x * sv.x) * qs);
        o[2] = (bf16_t)((a.z * c.y - a.w * sv.y) * qs); o[3] = (bf16_t)((a.w * c.y + a.z * sv.y) * qs);
        o[4] = (bf16_t)((b.x * c.z - b.y * sv.z) * qs); o[5] = (bf16_t)((b.y * c.z + b.x * sv.z) * qs);
        o[6] = (bf16_t)((b.z * c.w - b.w * sv.w) * qs); o[7] = (bf16_t)((b.w * c.w + b.z * sv.w) * qs);
        if (QK8 && f8out) {
          float v[8], amax = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) { v[e] = (float)o[e]; amax = fmaxf(amax, fabsf(v[e])); }
          amax = fmaxf(amax, dpp_f<0xB1>(amax));      // the quad's other three lanes hold the rest of the 32 values
          amax = fmaxf(amax, dpp_f<0x4E>(amax));
          const int ex = mx_exponent(amax);
          const float inv = mx_inv_scale(ex);
          const unsigned int __attribute__((ext_vector_type(2))) w8 = {pack_fp8x4(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv),
                                                                      pack_fp8x4(v[4] * inv, v[5] * inv, v[6] * inv, v[7] * inv)};
          const __amdgpu_buffer_rsrc_t w8st = buf_rows(qk8, row0 + 32 * i, rl - 32 * i, 64);
          const __amdgpu_buffer_rsrc_t wmx = buf_rows(qk8mx, row0 + 32 * i, rl - 32 * i, 2);
          __builtin_amdgcn_raw_buffer_store_b64(w8, w8st, vo8 + it * RPI * 64, 0, BUF_SC1);
          __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(ex + 127), wmx, vomx == BUF_OOB ? BUF_OOB : vomx + it * RPI * 2, 0, BUF_SC1);
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), wst, vo + it * RPI * 128, 0, BUF_SC1);
        }
      }
    }
    };
    if constexpr (LNA != 0) { if (lna) blocks(std::true_type{}); else blocks(std::false_type{}); } else blocks(std::false_type{});
  } else if constexpr (EPI == EPI_BIAS_F32) {
    using S = SlabF32<32, WTN>;
    const int rr = lane / S::CPR, ch = lane % S::CPR;
    row_window_kv(pre.win);
    int vo = (rr * p.ldc + nw + ch * 4) * 4;
    if (nw + ch * 4 >= p.n_valid) vo = BUF_OOB;
    const int rstep = S::RPI * p.ldc * 4;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = j * 32 + 8 * g + 4 * hi;
          const float4 bias = bias_of(j, g);
          *reinterpret_cast<float4*>(slab + l31 * S::PITCH + nl * 4) =
              make_float4(acc[i][j][4 * g + 0] + bias.x, acc[i][j][4 * g + 1] + bias.y, acc[i][j][4 * g + 2] + bias.z,
                          acc[i][j][4 * g + 3] + bias.w);
        }
      const __amdgpu_buffer_rsrc_t wst = buf_rows(p.out_f32, (long long)mwu + 32 * i, pre.win.rl - 32 * i, p.ldc * 4);
#pragma unroll
      for (int it = 0; it < S::ITERS; ++it) {
        const float4 d = *reinterpret_cast<const float4*>(slab + (it * S::RPI + rr) * S::PITCH + ch * 16);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, d), wst, vo + it * rstep, 0, 0);
      }
    }
  } else {   // bf16 outputs: plain, GELU-tanh
    using S = SlabBf16<32, WTN>;
    const int rr = lane / S::CPR, ch = lane % S::CPR;
    row_window_kv(pre.win);
    int vo = (rr * p.ldc + nw + ch * 8) * 2;
    if (nw + ch * 8 >= p.n_valid) vo = BUF_OOB;
    const int rstep = S::RPI * p.ldc * 2;
    auto blocks = [&](auto lna_c) {
    constexpr bool L = LNA != 0 && decltype(lna_c)::value;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      float r = 1.0f, nrm = 0.f;
      if constexpr (L) row_rn(i, r, nrm);
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = j * 32 + 8 * g + 4 * hi;
          const float4 bias = bias_of(j, g);
          float v0, v1, v2, v3;
          if constexpr (L) {
            const float4 c1 = c1_of(j, g);
            v0 = __builtin_fmaf(r, acc[i][j][4 * g + 0], __builtin_fmaf(nrm, c1.x, bias.x));
            v1 = __builtin_fmaf(r, acc[i][j][4 * g + 1], __builtin_fmaf(nrm, c1.y, bias.y));
            v2 = __builtin_fmaf(r, acc[i][j][4 * g + 2], __builtin_fmaf(nrm, c1.z, bias.z));
            v3 = __builtin_fmaf(r, acc[i][j][4 * g + 3], __builtin_fmaf(nrm, c1.w, bias.w));
          } else {
            v0 = acc[i][j][4 * g + 0] + bias.x; v1 = acc[i][j][4 * g + 1] + bias.y;
            v2 = acc[i][j][4 * g + 2] + bias.z; v3 = acc[i][j][4 * g + 3] + bias.w;
          }
          bf16x4 o;
          if (EPI == EPI_BIAS_GELU_BF16) {
            const f32x2 g01 = gelu_tanh_f2(f32x2{v0, v1}), g23 = gelu_tanh_f2(f32x2{v2, v3});
            o = pack4(g01.x, g01.y, g23.x, g23.y);
          } else {
            o = pack4(v0, v1, v2, v3);
          }
          *reinterpret_cast<bf16x4*>(slab + l31 * S::PITCH + nl * 2) = o;
        }
      const __amdgpu_buffer_rsrc_t wst = buf_rows(p.out_bf16, (long long)mwu + 32 * i, pre.win.rl - 32 * i, p.ldc * 2);
#pragma unroll
      for (int it = 0; it < S::ITERS; ++it) {
        const u32x4 d = *reinterpret_cast<const u32x4*>(slab + (it * S::RPI + rr) * S::PITCH + ch * 16);
        __builtin_amdgcn_raw_buffer_store_b128(d, wst, vo + it * rstep, 0, BUF_SC1);
      }
    }
    };
    if constexpr (LNA != 0) { if (lna) blocks(std::true_type{}); else blocks(std::false_type{}); } else blocks(std::false_type{});
  }
}

// ---- plain orientation, used only for the V projection (EPI_V_T): lane -> column n (= head dim d),
//      register r -> row m = mw + 32 i + (r&3) + 8 (r>>2) + 4 (lane>>5).  The slab is [d][pos] so that v^T rows
//      (contiguous positions) leave as whole 64/128-B segments.
// ln fold (p.ln_part): `rs` = the (r, -r mu) pairs of the rows mw .. mw + 32 TI - 1 in LDS; four consecutive rows per register group.
template <int TI, int TJ>
__device__ __forceinline__ void epilogue_vt(const GemmParams& p, f32x16 (&acc)[TI][TJ], char* slab, int mw, int nw, int lane,
                                            const float* rs = nullptr) {
  constexpr int WTM = 32 * TI, WTN = 32 * TJ;
  using S = SlabBf16<WTN, WTM>;   // rows = d (WTN of them), columns = positions (WTM)
  const int l31 = lane & 31, hi = lane >> 5;
  const bool lna = p.ln_part != nullptr;
  const float* bsrc = p.bias;
  const float* c1src = nullptr;
  if (lna) {
    const float* row = p.tab + (size_t)p.step_idx[0] * p.tab_stride;
    bsrc = row + p.lnc2_off;
    c1src = row + p.lnc1_off;
  }
  auto park = [&](auto lna_c) {      // with / without the folded LayerNorm: a run-time property of the launch, two copies of the unrolled code
    constexpr bool L = decltype(lna_c)::value;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const float bias = bsrc[nw + j * 32 + l31];
      const float c1 = L ? c1src[nw + j * 32 + l31] : 0.f;
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 o;
          if constexpr (L) {
            // (r, -r mu) of rows +0, +1 | +2, +3
            const float4 ra = *reinterpret_cast<const float4*>(rs + 2 * (i * 32 + 8 * g + 4 * hi));
            const float4 rb = *reinterpret_cast<const float4*>(rs + 2 * (i * 32 + 8 * g + 4 * hi) + 4);
            o = pack4(__builtin_fmaf(ra.x, acc[i][j][4 * g + 0], __builtin_fmaf(ra.y, c1, bias)),
                      __builtin_fmaf(ra.z, acc[i][j][4 * g + 1], __builtin_fmaf(ra.w, c1, bias)),
                      __builtin_fmaf(rb.x, acc[i][j][4 * g + 2], __builtin_fmaf(rb.y, c1, bias)),
                      __builtin_fmaf(rb.z, acc[i][j][4 * g + 3], __builtin_fmaf(rb.w, c1, bias)));
          } else {
            o = pack4(acc[i][j][4 * g + 0] + bias, acc[i][j][4 * g + 1] + bias, acc[i][j][4 * g + 2] + bias, acc[i][j][4 * g + 3] + bias);
          }
          *reinterpret_cast<bf16x4*>(slab + (j * 32 + l31) * S::PITCH + (i * 32 + 8 * g + 4 * hi) * 2) = o;
        }
    }
  };
  if (lna) park(std::true_type{}); else park(std::false_type{});
  static_assert(WTN <= 64, "a wave tile lies inside one head");
  const int rr = lane / S::CPR, ch = lane % S::CPR;
  const int mwu = __builtin_amdgcn_readfirstlane(mw), nwu = __builtin_amdgcn_readfirstlane(nw);
  const int b2 = mwu / p.seq_pitch, pos0 = mwu - b2 * p.seq_pitch;   // a wave tile never straddles samples (pitch % 128 == 0)
  const int head = (nwu % (p.heads * 64)) >> 6;
  // v^T is [b2][head][64 d][npad]: the window is this wave tile's WTN d-rows (none when the tile lies past M), a lane stores 8 positions of
  // one of them.  Positions past seq_valid inside the pitch are padding columns of v^T (masked keys): storing them is harmless
  const __amdgpu_buffer_rsrc_t wst = buf_rows(p.vt, ((long long)b2 * p.heads + head) * 64 + (nwu & 63), mwu < p.M ? WTN : 0, p.npad * 2);
  int vo = (rr * p.npad + pos0 + ch * 8) * 2;
  if (pos0 + ch * 8 >= p.npad) vo = BUF_OOB;
  const int rstep = S::RPI * p.npad * 2;
#pragma unroll
  for (int it = 0; it < S::ITERS; ++it) {
    const u32x4 d = *reinterpret_cast<const u32x4*>(slab + (it * S::RPI + rr) * S::PITCH + ch * 16);
    __builtin_amdgcn_raw_buffer_store_b128(d, wst, vo + it * rstep, 0, BUF_SC1);
  }
}

template <int EPI, int WTM, int WTN>
constexpr int slab_bytes() {
  return (EPI == EPI_GATE_RES || EPI == EPI_BIAS_F32 || EPI == EPI_QK_ROPE) ? SlabF32<WTM, WTN>::BYTES
         : (EPI == EPI_V_T)                           ? SlabBf16<WTN, WTM>::BYTES
         : (EPI == EPI_BIAS_GELU_F8)                  ? SlabF8<WTM, WTN>::BYTES
                                                      : SlabBf16<WTM, WTN>::BYTES;
}

// (measurement builds only: the LayerNorm-modulate tail of the gate + residual launch, engine option "ln_fused")
#ifdef LEMAS_MEASUREMENT_BUILD
#include "gemm_bf16_exp_ln_tail.h"
#endif

// ---------------------------------------------------------------- ln fold: row statistics on both sides of a GEMM (GemmParams)
// consumer: the first TBM * TPR threads turn the LN_NP partial (sum, sum of squares) pairs of the tile's rows into (r, -r mu) in LDS.
// The loads are issued BEFORE the K loop's first LDS-DMA (so the counted vmcnt waits of the loops only ever see them as older
// operations) and consumed after the K loop (256 x 256 body: in the prologue); the table lives behind the ring / slab area and is read
// by the epilogues.
// TPR adjacent threads share a row (each takes LN_NP / TPR slots): thread t of the first TBM * TPR threads of the workgroup.
template <int TPR> struct RowStatLoad { f32x4 v[LN_NP / 2 / TPR]; };
template <int TPR>
__device__ __forceinline__ void ln_rowstat_load(const GemmParams& p, int m0, int t, RowStatLoad<TPR>& L) {
  int m = m0 + t / TPR;
  m = m < p.M ? m : p.M - 1;
  const float4* q = reinterpret_cast<const float4*>(p.ln_part + (size_t)m * (LN_NP * 2)) + (t % TPR) * (LN_NP / 2 / TPR);
  // asm loads (invisible to the compiler's vmcnt bookkeeping, like the loops' fragment reads): the caller waits with an exact count
  // -- the number of LDS-DMA instructions it issued after this -- instead of the vmcnt(0) the compiler would put in front of the first
  // use, which also drains every prologue DMA
#pragma unroll
  for (int j = 0; j < LN_NP / 2 / TPR; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(L.v[j]) : "v"(q + j) : "memory");
}
template <int TPR>
__device__ __forceinline__ void ln_rowstat_store(RowStatLoad<TPR>& L, float* rs, int t) {
#pragma clang fp contract(off)
  // The loads must have landed: every caller has passed a volatile s_waitcnt vmcnt that covers them before it gets here (they are the
  // OLDEST vector-memory operations of the wave, so the first counted wait of the K loop already retires them, and the loop's last
  // wait is vmcnt(0); the 256 x 256 body waits explicitly in its prologue).  These empty volatile asms stay behind that wait in
  // program order and make every use of the loaded registers depend on them: nothing that reads L.v can be scheduled above it.
#pragma unroll
  for (int j = 0; j < LN_NP / 2 / TPR; ++j) asm volatile("" : "+v"(L.v[j]));
  // one balanced binary tree over the 16 slots whatever TPR is (pairs inside a float4, then inside the thread, then across the TPR
  // threads): the statistics of a row do not depend on the tile that consumes it (single-lane and two-lane runs stay bit-identical)
  constexpr int NV = LN_NP / 2 / TPR;
  float a[NV], b[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) { a[j] = L.v[j][0] + L.v[j][2]; b[j] = L.v[j][1] + L.v[j][3]; }
#pragma unroll
  for (int w = 1; w < NV; w <<= 1)
#pragma unroll
    for (int j = 0; j + w < NV; j += 2 * w) { a[j] += a[j + w]; b[j] += b[j + w]; }
  float s = a[0], ss = b[0];
#pragma unroll
  for (int o = 1; o < TPR; o <<= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); }
  const float mean = s * (1.0f / LN_D);
  float var = ss * (1.0f / LN_D) - mean * mean;
  var = var > 0.f ? var : 0.f;
  const float r = rsqrtf(var + 1e-6f);
  if (t % TPR == 0) *reinterpret_cast<float2*>(rs + 2 * (t / TPR)) = make_float2(r, -r * mean);
}
template <bool F8> struct FragT { using type = bf16x8; };
template <> struct FragT<true> { using type = i32x8; };

// F8: operands are e4m3 bytes (a 128-B LDS row = 128 K elements), the MFMA is v_mfma_scale_f32_32x32x64_f8f6f4 (2x the
// bf16 rate) and the activation tile carries MX block scales: one dword (4 E8M0 bytes = the 4 K-blocks of the tile)
// per row, brought in by one extra 4-B-per-lane LDS-DMA per wave and tile.  Operand layout of the instruction
// (probed on the hardware, tools/exp/mx_probe.hip): lane (i = l&31, h = l>>5) holds row i; registers 0-3 are K
// 16h..16h+15 and registers 4-7 are K 32+16h..; the scale of K-block beta (32 K) comes from lane i + 32 beta.
// The WEIGHT pieces of the first NSTAGE - 1 K-tiles of tile column n0, requested ahead of a gemm_body<..., WPRE = true> call (measurement builds:
// the persistent block kernel issues them before it waits at a grid barrier -- weights do not depend on the previous stage).  Same pieces, same
// LDS addresses as the body's own prologue.
template <int TBM, int TBN, int NSTAGE, int NWM, int NWN>
__device__ __forceinline__ void gemm_prefetch_w(const GemmParams& p, char* smem, int n0) {
  constexpr int NW = NWM * NWN, A_BYTES = TBM * 128, STAGE = (TBM + TBN) * 128, B_PW = TBN / 8 / NW;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane >> 3, lp = lane & 7;
  const size_t rowb = (size_t)p.K * 2;
  const int nk = (int)(rowb >> 7);
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s) {
    if (s >= nk) break;
#pragma unroll
    for (int q = 0; q < B_PW; ++q) {
      const int r = 8 * (wave + NW * q) + lr;
      const char* src = reinterpret_cast<const char*>(p.W) + (size_t)(n0 + r) * rowb + ((lp ^ ((r >> 1) & 7)) << 4) + s * 128;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(smem + s * STAGE + wave * 1024 + A_BYTES + q * (NW * 1024)), 16, 0, 0);
    }
  }
}

// WPRE (measurement builds, bf16 only): the weight pieces of the prologue's stages are already in flight (gemm_prefetch_w): the prologue requests
// the activation pieces only, and the first wait of the loop counts accordingly.
// NL (round 6, tiles T128x128L / T128x128W4L): that many LOADER waves behind the NWM x NWN compute waves.  A 1-KiB LDS-DMA instruction holds its
// wave's issue for 60-185 cycles (MI355X_MICROARCH.md price list), and in the plain form every compute wave places 4 (8 waves) or 8 (4 waves) of
// them per K-tile between its own MFMAs -- the K-loop ablations price that at 0.085 us of a 0.425 us K-tile (profiles/r04/r04g_kloop_ablations_128x128.txt:
// 0.34 without the refill).  Here the compute waves issue no vector-memory instruction inside the loop at all: waves NW .. NW + NL - 1 (one per
// SIMD) request every piece of every K-tile, wait for them (counted vmcnt) and meet the compute waves at the loop's one barrier per K-tile; they
// leave at the barrier that hands the ring's LDS to the epilogue.  Same ring, same LDS image, same barriers, same MFMA order: bit-identical output.
template <int EPI, int TBM, int TBN, int NSTAGE, int NWM, int NWN, bool SWAP, bool F8, bool WPRE = false, int NL = 0>
__device__ __forceinline__ void gemm_body(const GemmParams& p, char* smem, int m0, int n0, float* rs_lds) {
  static_assert(!(WPRE && F8), "the weight prefetch form exists for the bf16 tiles");
  static_assert(NL == 0 || (!F8 && !WPRE), "loader waves exist for the bf16 tiles");
  using frag_t = typename FragT<F8>::type;
  constexpr int NW = NWM * NWN;                       // waves per workgroup
  constexpr int WTM = TBM / NWM, WTN = TBN / NWN;     // wave tile
  constexpr int TI = WTM / 32, TJ = WTN / 32;         // 32x32 MFMA tiles per wave
  constexpr int A_BYTES = TBM * 128, SC_OFF = (TBM + TBN) * 128, STAGE = SC_OFF + (F8 ? TBM * 4 : 0);
  constexpr int A_PW = TBM / 8 / NW, B_PW = TBN / 8 / NW;  // 1-KiB DMA instructions per wave per tile
  constexpr int PW = A_PW + B_PW + (F8 ? 1 : 0);           // + the scale piece
  constexpr int KS = F8 ? 2 : 4;                           // MFMA k-steps per 128-B K tile
  static_assert((TBM / 8) % NW == 0 && (TBN / 8) % NW == 0, "DMA pieces must divide over the waves");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wave / NWN, wn = wave % NWN;

  // DMA source pointers: instruction q of this wave covers LDS rows 8 (wave + NW q) .. +7; lane -> (row, phys chunk)
  const int lr = lane >> 3, lp = lane & 7;
  const size_t rowb = F8 ? (size_t)p.K : (size_t)p.K * 2;   // bytes per operand row
  if constexpr (NL > 0) {
    if (wave >= NW) {
      // ---- a loader wave: piece x of a K-tile = LDS rows 8 x .. 8 x + 7 of [A (TBM rows) | W (TBN rows)]; loader lw takes pieces lw + NL q
      constexpr int A_P = TBM / 8, P = (TBM + TBN) / 8, PL = P / NL;
      static_assert(A_P % NL == 0 && P % NL == 0, "pieces must divide over the loader waves");
      const int lw = wave - NW;
      const char* src[PL];
#pragma unroll
      for (int q = 0; q < PL; ++q) {
        const int x = lw + NL * q;
        if (q < A_P / NL) {
          const int r = 8 * x + lr;
          int am = m0 + r;
          am = am < p.M ? am : p.M - 1;
          src[q] = reinterpret_cast<const char*>(p.A) + (size_t)am * rowb + ((lp ^ ((r >> 1) & 7)) << 4);
        } else {
          const int r = 8 * (x - A_P) + lr;
          src[q] = reinterpret_cast<const char*>(p.W) + (size_t)(n0 + r) * rowb + ((lp ^ ((r >> 1) & 7)) << 4);
        }
      }
      auto issue_l = [&](int stage, int kt) {
#pragma unroll
        for (int q = 0; q < PL; ++q)      // (the LDS image is [A rows | W rows], row r at r * 128 B: piece x at x * 1024 for both operands)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[q] + kt * 128),
                                           (__attribute__((address_space(3))) void*)(smem + stage * STAGE + (lw + NL * q) * 1024), 16, 0, 0);
      };
      const int nkl = (int)(rowb >> 7);
#pragma unroll
      for (int st = 0; st < NSTAGE - 1; ++st)
        if (st < nkl) issue_l(st, st);
      int stage = 0;
      for (int kt = 0; kt < nkl; ++kt) {
        if (kt + NSTAGE - 2 < nkl) wait_vmcnt<PL * (NSTAGE - 2)>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        const int nt = kt + NSTAGE - 1;
        int ns = stage + NSTAGE - 1;
        ns = ns >= NSTAGE ? ns - NSTAGE : ns;
        if (nt < nkl) issue_l(ns, nt);
        stage = stage + 1 == NSTAGE ? 0 : stage + 1;
      }
      __syncthreads();      // the barrier behind the K loop (compute waves: "every wave is done with the ring")
      return;
    }
  }
  const char* asrc[A_PW];
  const char* wsrc[B_PW];
#pragma unroll
  for (int q = 0; q < A_PW; ++q) {
    const int r = 8 * (wave + NW * q) + lr;
    int am = m0 + r;
    am = am < p.M ? am : p.M - 1;  // clamp: out-of-range rows are computed and discarded
    asrc[q] = reinterpret_cast<const char*>(p.A) + (size_t)am * rowb + ((lp ^ ((r >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int q = 0; q < B_PW; ++q) {
    const int r = 8 * (wave + NW * q) + lr;
    wsrc[q] = reinterpret_cast<const char*>(p.W) + (size_t)(n0 + r) * rowb + ((lp ^ ((r >> 1) & 7)) << 4);
  }
  // MX scales: every wave copies the dwords of 64 rows (row group wave % (TBM/64); duplicates write identical bytes)
  const int sgrp = wave % (TBM / 64);
  const char* ssrc = nullptr;
  if (F8) {
    int am = m0 + sgrp * 64 + lane;
    am = am < p.M ? am : p.M - 1;
    ssrc = reinterpret_cast<const char*>(p.a_mx) + (size_t)am * (p.K >> 5);
  }
  // piece x of a tile: x < A_PW -> A rows, then W rows (one 1-KiB global_load_lds_dwordx4 each), then the scale dwords
  auto issue_piece = [&](int stage, int kt, int x) {
    char* base = smem + stage * STAGE + wave * 1024;
    if (x < A_PW)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[x] + kt * 128),
                                       (__attribute__((address_space(3))) void*)(base + x * (NW * 1024)), 16, 0, 0);
    else if (x < A_PW + B_PW)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[x - A_PW] + kt * 128),
                                       (__attribute__((address_space(3))) void*)(base + A_BYTES + (x - A_PW) * (NW * 1024)), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ssrc + kt * 4),
                                       (__attribute__((address_space(3))) void*)(smem + stage * STAGE + SC_OFF + sgrp * 256), 4, 0, 0);
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
  constexpr bool PREFETCH_EPI = SWAP && EPI != EPI_BIAS_GELU_F8;
  constexpr int LNA = (epi_lna<EPI>() && !F8) ? 1 : 0;      // the fp8 path keeps its MXFP8 LayerNorm launch
  EpiPre<EPI, TJ, true, LNA> pre;

  const int nk = (int)(rowb >> 7);
  PHASE_STAMP(0); PHASE_STAMP(1);
  constexpr int TPR = (64 * NW / TBM) >= 4 ? 4 : (64 * NW / TBM) >= 2 ? 2 : 1;
  RowStatLoad<TPR> rsl;
  bool rs_mine = false;
  if constexpr (LNA != 0) {
    rs_mine = p.ln_part != nullptr && tid < TBM * TPR;
    if (rs_mine) ln_rowstat_load<TPR>(p, m0, tid, rsl);
  }
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (NL == 0 && s < nk) {
      if constexpr (WPRE) {
#pragma unroll
        for (int x = 0; x < A_PW; ++x) issue_piece(s, s, x);
      } else {
        issue(s, s);
      }
    }
  // (the loaded statistics stay in registers through the K loop -- they are older than every LDS-DMA, so the loop's counted waits cover
  // them -- and become the (r, -r mu) table just before the epilogue: waiting for them here costs ~1-2 us of every launch)

  // one MFMA of k-step kk; asc = this lane's scale dword for the A row-fragment, already shifted by 8 hi
  auto mfma1 = [&](const frag_t& fa, const frag_t& fb, f32x16& c, int kk, int asc) {
    if constexpr (F8) {
      if (SWAP) c = kk == 0 ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb, fa, c, 0, 0, 0, 127, 0, asc)
                            : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb, fa, c, 0, 0, 0, 127, 2, asc);
      else c = kk == 0 ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, c, 0, 0, 0, asc, 0, 127)
                       : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, c, 0, 0, 2, asc, 0, 127);
    } else {
      c = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fa, c, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c, 0, 0, 0);
    }
  };
  {
    // Hand-scheduled K loop.  Left to the compiler, the waitcnt pass cannot count past the LDS-DMA / branch structure and
    // drops an s_waitcnt lgkmcnt(0) in front of every MFMA group, i.e. it also waits for the fragment reads of the NEXT
    // k-step that were issued a few instructions earlier (measured: matrix pipe 52 % busy while resident).  Here the
    // fragment reads are asm (untracked) and each k-step waits with an exact count: only for its own operands.  The refill
    // DMA of tile kt+NSTAGE-1 is spread over the k-steps so that its issue slots hide behind the MFMAs.
    constexpr int RPF = F8 ? 2 : 1;                 // ds_read_b128 per fragment
    constexpr int NR = (TI + TJ) * RPF;             // LDS reads per k-step (F8: + TI scale dwords in step 0, all older)
    static_assert(NR + (F8 ? TI : 0) <= 15, "lgkmcnt is a 4-bit counter");
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    unsigned aA[KS][RPF], aB[KS][RPF];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
      for (int h = 0; h < RPF; ++h) {
        const int chunk = F8 ? kk * 4 + 2 * h + hi : kk * 2 + hi;
        aA[kk][h] = lds_off(wm * WTM + l31, chunk);
        aB[kk][h] = A_BYTES + lds_off(wn * WTN + l31, chunk);
      }
    const unsigned aS = SC_OFF + (wm * WTM + l31) * 4;
    u32x4 fa[2][TI][RPF], fb[2][TJ][RPF];
    int asc[TI];
    auto as_frag = [&](const u32x4 (&f)[RPF]) -> frag_t {
      if constexpr (F8) {
        i32x8 r;
        r[0] = f[0][0]; r[1] = f[0][1]; r[2] = f[0][2]; r[3] = f[0][3]; r[4] = f[1][0]; r[5] = f[1][1]; r[6] = f[1][2]; r[7] = f[1][3];
        return r;
      } else {
        return __builtin_bit_cast(bf16x8, f[0]);
      }
    };
    // measurement builds only (-DLEMAS_ABLATE=bits; compile-time so that the loop keeps its shape): 1 = no MFMAs, 2 = no refill LDS-DMA after
    // the prologue's stages, 4 = no fragment reads from LDS (wrong results; what each removes is that pipe's share of the loop)
#ifdef LEMAS_ABLATE
    constexpr int abl = LEMAS_ABLATE;
#else
    constexpr int abl = 0;
#endif
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
      if constexpr (NL == 0) {
        if (WPRE && kt == 0 && NSTAGE - 2 < nk) wait_vmcnt<A_PW * (NSTAGE - 2)>();     // the youngest requests are the activation pieces of stage 1 only
        else if (kt + NSTAGE - 2 < nk) wait_vmcnt<PW * (NSTAGE - 2)>();
        else wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();
      // last K-tile: nothing is in flight and no vmcnt wait follows -- request the epilogue's global operands under its MFMAs
      if constexpr (PREFETCH_EPI) { if (kt == nk - 1) pre.load(p, m0 + wm * WTM, n0 + wn * WTN, lane); }
      const int nt = kt + NSTAGE - 1;
      int ns = stage + NSTAGE - 1;
      ns = ns >= NSTAGE ? ns - NSTAGE : ns;
      const bool refill = NL == 0 && nt < nk && !(abl & 2);
      const unsigned sb = lds_base + stage * STAGE;
      if constexpr (F8) ReadScales<0, TI>::run(asc, sb + aS);
      if (!(abl & 4)) {
        ReadFrags<0, TI, RPF>::run(fa[0], aA[0], sb);
        ReadFrags<0, TJ, RPF>::run(fb[0], aB[0], sb);
      }
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        if (kk < KS - 1 && !(abl & 4)) {
          ReadFrags<0, TI, RPF>::run(fa[(kk + 1) & 1], aA[kk + 1], sb);
          ReadFrags<0, TJ, RPF>::run(fb[(kk + 1) & 1], aB[kk + 1], sb);
        }
        if (refill) {
#pragma unroll
          for (int x = (PW * kk) / KS; x < (PW * (kk + 1)) / KS; ++x) issue_piece(ns, nt, x);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (abl & 4) wait_lgkmcnt<0>();
        else if (kk < KS - 1) wait_lgkmcnt<NR>();
        else wait_lgkmcnt<0>();
        if (F8 && kk == 0) {
#pragma unroll
          for (int t = 0; t < TI; ++t) asc[t] >>= (8 * hi);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(abl & 1)) {
#pragma unroll
          for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) mfma1(as_frag(fa[kk & 1][i]), as_frag(fb[kk & 1][j]), acc[i][j], kk, F8 ? asc[i] : 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      stage = stage + 1 == NSTAGE ? 0 : stage + 1;
    }
  }
  if constexpr (F8) {   // per-output-channel weight scale
    const float* ws = p.w_scale + n0 + wn * WTN;
    if (SWAP) {
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 w4 = *reinterpret_cast<const float4*>(ws + j * 32 + 8 * g + 4 * hi);
#pragma unroll
          for (int i = 0; i < TI; ++i) {
            acc[i][j][4 * g + 0] *= w4.x; acc[i][j][4 * g + 1] *= w4.y; acc[i][j][4 * g + 2] *= w4.z; acc[i][j][4 * g + 3] *= w4.w;
          }
        }
    } else {
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const float w1 = ws[j * 32 + l31];
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] *= w1;
      }
    }
  }
  if constexpr (LNA != 0) { if (rs_mine) ln_rowstat_store<TPR>(rsl, rs_lds, tid); }
  __syncthreads();   // every wave is done with the ring: its LDS becomes the epilogue slabs
  PHASE_STAMP(2);
  // one 32-row block of the wave tile at a time through a small wave-private slab (LDS ops of a wave execute in order, so
  // the slab can be rewritten right after it was read): keeps the kernel's LDS footprint = the ring, not ring + big slabs
  char* slab = smem + wave * slab_bytes<EPI, 32, WTN>();
  if constexpr (SWAP && EPI != EPI_BIAS_GELU_F8) {
    epilogue_row_blocks<EPI, TI, TJ, true, 1, LNA, F8 && EPI == EPI_QK_ROPE>(p, acc, slab, m0 + wm * WTM, n0 + wn * WTN, lane, pre, rs_lds + 2 * (wm * WTM));
  } else {
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      f32x16 (&blk)[1][TJ] = *reinterpret_cast<f32x16 (*)[1][TJ]>(&acc[i]);
      if constexpr (SWAP) epilogue_rows<EPI, 1, TJ>(p, blk, slab, m0 + wm * WTM + 32 * i, n0 + wn * WTN, lane);
      else epilogue_vt<1, TJ>(p, blk, slab, m0 + wm * WTM + 32 * i, n0 + wn * WTN, lane, rs_lds + 2 * (wm * WTM + 32 * i));
    }
  }
#ifdef LEMAS_MEASUREMENT_BUILD
  if constexpr (EPI == EPI_GATE_RES && SWAP && !F8) {
    if (p.ln_out) ln_tail<TBM, TBN, NW>(p, m0, n0, smem);      // (its first barrier retires every wave's epilogue slab: LDS is free behind it)
  }
#endif
  PHASE_STAMP_END();
}

// ================================================================================================================
// 256 x 256 tile, 8 waves, "ping-pong" schedule for the batched shapes (bf16).
//
// The loop above has one barrier per K-tile with every wave in the same phase: all waves read LDS together, then all issue
// MFMAs together, and the matrix pipe idles while the fragment reads are in flight (PMC: busy 52 % of a wave's residency;
// 0.6-0.75 PFLOP/s at M = 18432).  Here the K-tile is cut into FOUR phases of one 64 x 32 accumulator quadrant each, and the two
// wave rows (waves 0-3 / 4-7: one of each per SIMD) run ONE BARRIER APART: while one group issues its 8 MFMAs of a phase (256
// cycles of matrix pipe), the other group does the LDS reads and the LDS-DMA issue of its next phase.  Every phase is
//      L: fragment reads of this phase's quadrant, LDS-DMA of one half-tile a full K-tile ahead, counted vmcnt for the half-tile
//         the NEXT phase reads;   s_barrier;   lgkmcnt(0);   M: 8 MFMAs;   s_barrier
// so each barrier flips the two groups between L and M.  Operands live in LDS as eight 16-KiB half-tiles [K-tile parity][A|W][half]:
// A half h = the h-th 64 rows of BOTH wave rows, W half h = the h-th 32 columns of all four wave columns, so the quadrant
// order (A0,W0) (A0,W1) (A1,W1) (A1,W0) needs exactly one new half-tile per phase (two for the first) and each half-tile is
// read from LDS once per K-tile (24 ds_read_b128 for 32 MFMAs per wave).  Half-tiles arrive by LDS-DMA (two 1-KiB pieces per wave),
// up to four in flight; vmcnt never drains to 0 inside the loop.
//
// Ordering rules (both are what makes the schedule race-free, not just fast):
//   RAW  a half-tile is waited for (vmcnt) in the L segment of phase p by every wave and read in phase p+1 at the earliest: the
//        reader has then passed a barrier that follows every wave's wait, for either group;
//   WAR  the DMA that overwrites buffer [parity][op][half] is issued a full K-tile (>= 2 phases, 4 barriers) after the last
//        read of its previous content, and those reads were retired by an lgkmcnt(0) before their own MFMA segment.
template <int EPI, bool SWAP>
__device__ __forceinline__ void gemm_body_pp(const GemmParams& p, char* smem, int m0, int n0, float* rs_lds) {
  constexpr int HT = 16384;                 // bytes per half-tile buffer: 128 rows x 128 B
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wave >> 2, wn = wave & 3;  // 2 (M) x 4 (N) waves, 128 x 64 outputs each
  const size_t rowb = (size_t)p.K * 2;
  const int nk = p.K >> 6;

  // ---- DMA sources.  Piece q of a half-tile = local rows 8q .. 8q+7; this wave moves pieces wave and wave + 8.
  //      A half h, local row r -> tile row (r >> 6) * 128 + h * 64 + (r & 63);  W half h, local row r -> tile col (r >> 5) * 64 + h * 32 + (r & 31)
  const int lr = lane >> 3, lp = lane & 7;
  const char* asrc[2][2];
  const char* wsrc[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = 8 * (wave + 8 * q) + lr;
      const int sw = (lp ^ ((r >> 1) & 7)) << 4;          // the swizzle sits in the SOURCE address (LDS image is lane-linear)
      int am = m0 + (r >> 6) * 128 + h * 64 + (r & 63);
      am = am < p.M ? am : p.M - 1;                        // rows past M are computed and discarded
      asrc[h][q] = reinterpret_cast<const char*>(p.A) + (size_t)am * rowb + sw;
      wsrc[h][q] = reinterpret_cast<const char*>(p.W) + (size_t)(n0 + (r >> 5) * 64 + h * 32 + (r & 31)) * rowb + sw;
    }
  auto issue_half = [&](int op, int h, int t) {            // op 0 = A, 1 = W; t = K-tile
    char* dst = smem + (((t & 1) * 2 + op) * 2 + h) * HT + wave * 1024;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const char* src = (op == 0 ? asrc[h][q] : wsrc[h][q]) + (size_t)t * 128;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dst + q * 8192), 16, 0, 0);
    }
  };

  // ---- fragment read addresses (byte offsets inside a half-tile buffer); +32 rows = +4096 B, the swizzle is unchanged
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  unsigned aoff[4], boff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    aoff[kk] = lds_off(wm * 64 + l31, kk * 2 + hi);
    boff[kk] = lds_off(wn * 32 + l31, kk * 2 + hi);
  }
  u32x4 fa[2][4], fb[2][4];      // fa[row block of the half][k-step]; fb[W half][k-step] (both W halves stay resident)
  auto read_a = [&](int t, int h) {
    const unsigned b = lds_base + (((t & 1) * 2 + 0) * 2 + h) * HT;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      lds_read_b128<0>(fa[0][kk], b + aoff[kk]);
      lds_read_b128<4096>(fa[1][kk], b + aoff[kk]);
    }
  };
  auto read_b = [&](int t, int h, u32x4 (&f)[4]) {
    const unsigned b = lds_base + (((t & 1) * 2 + 1) * 2 + h) * HT;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) lds_read_b128<0>(f[kk], b + boff[kk]);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto mma = [&](int ha, int jb, const u32x4 (&f)[4]) {     // the quadrant rows {2 ha, 2 ha + 1} x column block jb, K = 64
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bf16x8 a = __builtin_bit_cast(bf16x8, fa[i][kk]), b = __builtin_bit_cast(bf16x8, f[kk]);
        acc[ha * 2 + i][jb] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc[ha * 2 + i][jb], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[ha * 2 + i][jb], 0, 0, 0);
      }
    __builtin_amdgcn_s_setprio(0);
  };
  // end of an L segment: barrier, then this wave's fragment reads must have landed before its MFMAs
  auto l_to_m = [&]() {
    __builtin_amdgcn_s_barrier();
    wait_lgkmcnt<0>();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto m_to_l = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };

  // ---- prologue: the four half-tiles of K-tile 0 in consumption order (W0, A0, W1, A1); W0 and A0 must have landed
  PHASE_STAMP(0);
  RowStatLoad<2> rsl;      // 512 threads, 256 rows
  bool rs_mine = false;
  if constexpr (epi_lna<EPI>()) {
    rs_mine = p.ln_part != nullptr;
    if (rs_mine) ln_rowstat_load<2>(p, m0, tid, rsl);
  }
  issue_half(1, 0, 0); issue_half(0, 0, 0); issue_half(1, 1, 0); issue_half(0, 1, 0);
  if constexpr (epi_lna<EPI>()) { if (rs_mine) { wait_vmcnt<8>(); ln_rowstat_store<2>(rsl, rs_lds, tid); } }
  wait_vmcnt<4>();
  __builtin_amdgcn_s_barrier();
  PHASE_STAMP(1);
  if (wm == 1) __builtin_amdgcn_s_barrier();     // stagger: the second wave row runs one barrier behind the first

  for (int t = 0; t < nk; ++t) {
    const bool more = t + 1 < nk;
    // phase 0: quadrant (A0, W0); prefetch W0 of the next K-tile; W1 of this K-tile must land for phase 1
    read_b(t, 0, fb[0]);
    read_a(t, 0);
    if (more) { issue_half(1, 0, t + 1); wait_vmcnt<4>(); } else wait_vmcnt<2>();
    l_to_m();
    mma(0, 0, fb[0]);
    m_to_l();
    // phase 1: (A0, W1); prefetch A0'; A1 must land for phase 2
    read_b(t, 1, fb[1]);
    if (more) { issue_half(0, 0, t + 1); wait_vmcnt<4>(); } else wait_vmcnt<0>();
    l_to_m();
    mma(0, 1, fb[1]);
    m_to_l();
    // phase 2: (A1, W1); prefetch W1'
    read_a(t, 1);
    if (more) issue_half(1, 1, t + 1);
    l_to_m();
    mma(1, 1, fb[1]);
    m_to_l();
    // phase 3: (A1, W0), no LDS reads; prefetch A1'; W0' and A0' must land for the next phase 0
    if (more) { issue_half(0, 1, t + 1); wait_vmcnt<4>(); }
    l_to_m();
    mma(1, 0, fb[0]);
    m_to_l();
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();     // the first wave row catches up: equal barrier counts for all waves
  __syncthreads();   // every wave is done with the half-tiles: the LDS becomes the epilogue slabs
  PHASE_STAMP(2);
  char* slab = smem + wave * slab_bytes<EPI, 32, 64>();
  if constexpr (SWAP) {
    // 128 accumulator registers are still live here: the gate + residual epilogue has no room for its column vectors or a second
    // set of residual rows (and gained nothing from them: at full chip it is bound by the fp32 read-modify-write traffic), and
    // the loop (224-256 VGPRs) none for an early request.  (Round 4, AHEAD 2: the next block's rows requested once a block's
    // accumulators are parked -- 56 B of scratch and SLOWER: 88.2 -> 91.2 us at M = 30720, K = 1024, 141.6 -> 152.5 at K = 2048,
    // configs[3] -1.1 %; profiles/r04/r04_gate_epilogue_prefetch_256x256.txt)
    constexpr int LNA = epi_lna<EPI>() ? 2 : 0;
    EpiPre<EPI, 2, EPI != EPI_GATE_RES, LNA> pre;
    pre.load(p, m0 + wm * 128, n0 + wn * 64, lane);
    if constexpr (EPI == EPI_GATE_RES) {
      epilogue_row_blocks<EPI, 4, 2, false, 0, LNA>(p, acc, slab, m0 + wm * 128, n0 + wn * 64, lane, pre);
    } else {
      epilogue_row_blocks<EPI, 4, 2, true, 1, LNA>(p, acc, slab, m0 + wm * 128, n0 + wn * 64, lane, pre, rs_lds + 2 * (wm * 128));
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x16 (&blk)[1][2] = *reinterpret_cast<f32x16 (*)[1][2]>(&acc[i]);
      epilogue_vt<1, 2>(p, blk, slab, m0 + wm * 128 + 32 * i, n0 + wn * 64, lane, rs_lds + 2 * (wm * 128 + 32 * i));
    }
  }
  PHASE_STAMP_END();
}

// ================================================================================================================

// ================================================================================================================
// The same ping-pong idea on the 256 x 128 tile of the batch-1 shapes (bf16): 8 waves as 4 (M) x 2 (N), 64 x 64 outputs each, the two
// wave-row PAIRS {0,1} / {2,3} (one wave of each per SIMD) run one barrier apart.  A K-tile is two phases of two k-steps (8 MFMAs =
// 256 matrix-pipe cycles per wave); the ring is the lock-step kernel's (3 stages of [A 256 rows | W 128 rows] x 128 B, the same
// DMA pieces), filled two K-tiles ahead, three pieces per wave and phase.  Ordering: a K-tile is waited for (vmcnt) in the L
// segment of the second phase of its predecessor and read one phase later; every wave retires its fragment reads (lgkmcnt 0)
// BEFORE the barrier that ends its L segment, so the stage of K-tile t-1 may be restaged by either group right after that barrier.
template <int EPI, bool SWAP>
__device__ __forceinline__ void gemm_body_pp2(const GemmParams& p, char* smem, int m0, int n0, float* rs_lds) {
  constexpr int A_BYTES = 256 * 128, STAGE = (256 + 128) * 128;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane >> 3, lp = lane & 7;
  const size_t rowb = (size_t)p.K * 2;
  const int nk = p.K >> 6;
  const char* asrc[4];
  const char* wsrc[2];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = 8 * (wave + 8 * q) + lr;
    int am = m0 + r;
    am = am < p.M ? am : p.M - 1;
    asrc[q] = reinterpret_cast<const char*>(p.A) + (size_t)am * rowb + ((lp ^ ((r >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int r = 8 * (wave + 8 * q) + lr;
    wsrc[q] = reinterpret_cast<const char*>(p.W) + (size_t)(n0 + r) * rowb + ((lp ^ ((r >> 1) & 7)) << 4);
  }
  // pieces 0-3: A rows, 4-5: W rows (one 1-KiB global_load_lds_dwordx4 each) of K-tile kt into ring stage `stage`
  auto issue_piece = [&](int stage, int kt, int x) {
    char* base = smem + stage * STAGE + wave * 1024;
    if (x < 4)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[x] + (size_t)kt * 128),
                                       (__attribute__((address_space(3))) void*)(base + x * 8192), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[x - 4] + (size_t)kt * 128),
                                       (__attribute__((address_space(3))) void*)(base + A_BYTES + (x - 4) * 8192), 16, 0, 0);
  };
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  unsigned aoff[4], boff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    aoff[kk] = lds_off(wm * 64 + l31, kk * 2 + hi);
    boff[kk] = A_BYTES + lds_off(wn * 64 + l31, kk * 2 + hi);
  }
  u32x4 fa[2][2], fb[2][2];      // [k-step of the phase][32-row / 32-column block]
  auto read_phase = [&](int stage, int ph) {
    const unsigned b = lds_base + stage * STAGE;
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      lds_read_b128<0>(fa[k2][0], b + aoff[ph * 2 + k2]);
      lds_read_b128<4096>(fa[k2][1], b + aoff[ph * 2 + k2]);
      lds_read_b128<0>(fb[k2][0], b + boff[ph * 2 + k2]);
      lds_read_b128<4096>(fb[k2][1], b + boff[ph * 2 + k2]);
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto mma = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bf16x8 a = __builtin_bit_cast(bf16x8, fa[k2][i]), b = __builtin_bit_cast(bf16x8, fb[k2][j]);
          acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc[i][j], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i][j], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
  };
  auto l_to_m = [&]() {          // fragment reads retired BEFORE the barrier (see the WAR note above)
    wait_lgkmcnt<0>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
  auto m_to_l = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };

  // prologue: K-tiles 0 and 1 requested, K-tile 0 landed
  PHASE_STAMP(0);
  RowStatLoad<2> rsl;      // 512 threads, 256 rows
  bool rs_mine = false;
  if constexpr (epi_lna<EPI>()) {
    rs_mine = p.ln_part != nullptr;
    if (rs_mine) ln_rowstat_load<2>(p, m0, tid, rsl);
  }
#pragma unroll
  for (int x = 0; x < 6; ++x) issue_piece(0, 0, x);
  if (nk > 1) {
#pragma unroll
    for (int x = 0; x < 6; ++x) issue_piece(1, 1, x);
    wait_vmcnt<6>();
  } else {
    wait_vmcnt<0>();
  }
  __builtin_amdgcn_s_barrier();
  PHASE_STAMP(1);
  if (wm >= 2) __builtin_amdgcn_s_barrier();     // stagger: wave rows 2-3 run one barrier behind rows 0-1

  EpiPre<EPI, 2, true> pre;
  int stage = 0;
  for (int t = 0; t < nk; ++t) {
    int s2 = stage + 2;
    s2 = s2 >= 3 ? s2 - 3 : s2;
    const bool more = t + 2 < nk;
    // phase 0: k-steps 0, 1
    read_phase(stage, 0);
    if (more) {
#pragma unroll
      for (int x = 0; x < 3; ++x) issue_piece(s2, t + 2, x);
    }
    // last K-tile: nothing is in flight and no vmcnt wait follows -- request the epilogue's global operands under its MFMAs
    if constexpr (SWAP) { if (t == nk - 1) pre.load(p, m0 + wm * 64, n0 + wn * 64, lane); }
    l_to_m();
    mma();
    m_to_l();
    // phase 1: k-steps 2, 3; K-tile t+1 must have landed for the next phase 0
    read_phase(stage, 1);
    if (more) {
#pragma unroll
      for (int x = 3; x < 6; ++x) issue_piece(s2, t + 2, x);
      wait_vmcnt<6>();
    } else if (t + 1 < nk) {
      wait_vmcnt<0>();
    }
    l_to_m();
    mma();
    m_to_l();
    stage = stage == 2 ? 0 : stage + 1;
  }
  if (wm < 2) __builtin_amdgcn_s_barrier();      // rows 0-1 catch up: equal barrier counts
  // ln fold: the statistics loaded at the top (older than every LDS-DMA: long since landed) become the (r, -r mu) table now
  if constexpr (epi_lna<EPI>()) { if (rs_mine) ln_rowstat_store<2>(rsl, rs_lds, tid); }
  __syncthreads();
  PHASE_STAMP(2);
  char* slab = smem + wave * slab_bytes<EPI, 32, 64>();
  if constexpr (SWAP) {
    epilogue_row_blocks<EPI, 2, 2, true>(p, acc, slab, m0 + wm * 64, n0 + wn * 64, lane, pre, rs_lds + 2 * (wm * 64));
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x16 (&blk)[1][2] = *reinterpret_cast<f32x16 (*)[1][2]>(&acc[i]);
      epilogue_vt<1, 2>(p, blk, slab, m0 + wm * 64 + 32 * i, n0 + wn * 64, lane, rs_lds + 2 * (wm * 64 + 32 * i));
    }
  }
  PHASE_STAMP_END();
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm_pp2_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tm, tn;
  kernargs_early(p);
  if (!tile_of(blockIdx.x, (p.M + 255) / 256, p.N / 128, p.xcd_gx, p.xcd_runs, tm, tn) || tile_dead(p, tm * 256, 256)) return;
  gemm_body_pp2<EPI, EPI != EPI_V_T>(p, smem, tm * 256, tn * 128, reinterpret_cast<float*>(smem + 3 * (256 + 128) * 128));
}

template <int EPI>
struct LaunchPP2 {
  static constexpr int lds = 3 * (256 + 128) * 128 + 256 * 8;     // ring + the ln-fold (r, -r mu) table of the tile's rows
  static hipError_t init() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pp2_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  static hipError_t run(const GemmParams& p, hipStream_t s) {
    if (p.N % 128 != 0 || p.K % 64 != 0) return hipErrorInvalidValue;
    const dim3 grid(tail_skip_grid(p, grid_of((p.M + 255) / 256, p.N / 128, p.xcd_gx, p.xcd_runs))), block(512);
    // the (r, -r mu) table behind the ring is only there for a ln-fold consumer: every other launch keeps the ring's own footprint
    const int lds_now = p.ln_part ? lds : lds - 256 * 8;
    if (p.ev_start) hipExtLaunchKernelGGL((gemm_pp2_kernel<EPI>), grid, block, lds_now, s, p.ev_start, p.ev_stop, 0, p);
    else hipLaunchKernelGGL((gemm_pp2_kernel<EPI>), grid, block, lds_now, s, p);
    return hipGetLastError();
  }
};

template <int EPI>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tm, tn;
  kernargs_early(p);
  if (!tile_of(blockIdx.x, (p.M + 255) / 256, p.N / 256, p.xcd_gx, p.xcd_runs, tm, tn) || tile_dead(p, tm * 256, 256)) return;
  gemm_body_pp<EPI, EPI != EPI_V_T>(p, smem, tm * 256, tn * 256, reinterpret_cast<float*>(smem + 8 * 16384));
}

template <int EPI>
struct LaunchPP {
  static constexpr int lds = 8 * 16384 + 256 * 8;                 // half-tile buffers + the ln-fold (r, -r mu) table
  static_assert(8 * slab_bytes<EPI, 32, 64>() <= 8 * 16384, "epilogue slabs must fit the half-tile buffers");
  static hipError_t init() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pp_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  static hipError_t run(const GemmParams& p, hipStream_t s) {
    if (p.N % 256 != 0 || p.K % 64 != 0) return hipErrorInvalidValue;
    const dim3 grid(tail_skip_grid(p, grid_of((p.M + 255) / 256, p.N / 256, p.xcd_gx, p.xcd_runs))), block(512);
    const int lds_now = p.ln_part ? lds : lds - 256 * 8;
    if (p.ev_start) hipExtLaunchKernelGGL((gemm_pp_kernel<EPI>), grid, block, lds_now, s, p.ev_start, p.ev_stop, 0, p);
    else hipLaunchKernelGGL((gemm_pp_kernel<EPI>), grid, block, lds_now, s, p);
    return hipGetLastError();
  }
};

// LDS of the lock-step body: the ring (reused by the epilogue slabs), then the ln-fold (r, -r mu) table of the tile's rows
template <int EPI, int TBM, int TBN, int NSTAGE, int NWM, int NWN, bool F8>
constexpr int body_lds_base() {
  constexpr int ring = NSTAGE * ((TBM + TBN) * 128 + (F8 ? TBM * 4 : 0));
  constexpr int slabs = NWM * NWN * slab_bytes<EPI, 32, TBN / NWN>();
  return ring > slabs ? ring : slabs;
}

template <int EPI, int TBM, int TBN, int NSTAGE, int NWM, int NWN, bool F8, int NL = 0>
__global__ __launch_bounds__(64 * (NWM * NWN + NL)) void gemm_bf16_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tm, tn;
  kernargs_early(p);
  if (!tile_of(blockIdx.x, (p.M + TBM - 1) / TBM, p.N / TBN, p.xcd_gx, p.xcd_runs, tm, tn) || tile_dead(p, tm * TBM, TBM)) return;
  gemm_body<EPI, TBM, TBN, NSTAGE, NWM, NWN, EPI != EPI_V_T, F8, false, NL>(p, smem, tm * TBM, tn * TBN,
      reinterpret_cast<float*>(smem + body_lds_base<EPI, TBM, TBN, NSTAGE, NWM, NWN, F8>()));
}

// The tile shapes in use.
//   T256x128: 8 waves (4 x 2), 3-stage ring, 147 KB LDS -- N >= 2048 GEMMs at batch 1: bf16 on the ping-pong schedule of
//             gemm_body_pp2 (K loop 0.68 vs 0.78 us per K-tile), fp8 on the lock-step hand-scheduled loop of gemm_body
//   T128x128: 8 waves (2 x 4), 3-stage ring,  96 KB      -- mid-size shapes
//   T128x64 : 4 waves (2 x 2), 3-stage ring,  72 KB      -- short utterances (2 workgroups per CU)
//   T128x128W4: the same 128 x 128 tile with FOUR waves (2 x 2, 64 x 64 outputs each: 16 instead of 24 fragment reads per wave and K-tile, one
//             wave per SIMD), 96 KB.  Alone it is 5-10 % SLOWER than the 8-wave form (13.3 -> 14.6 us for the out-projection at M = 1875); with
//             two CFG lanes in flight it is the better FF1 tile end to end (+2.3 % on configs[1] against the 256 x 128 ping-pong tile, which is
//             itself 5 % better there than the 8-wave 128 x 128): 240 workgroups of half the register / wave-slot footprint pack with the other
//             lane's kernels where 128 eight-wave workgroups of 147 KB do not
//   T64x64  : 4 waves (2 x 2), 3-stage ring,  48 KB      -- N <= 1024 GEMMs of short utterances (M = 768: 8.0 vs 10.2 us, 12.1 vs 15.5 at K = 2048)
//   T256x256: 8 waves (2 x 4), ping-pong schedule of gemm_body_pp, 128 KB -- batched shapes (several rounds of tiles per CU);
//             bf16 only.  Its K loop runs at 1.45 PFLOP/s-equivalent per CU (K = 4096: 89 us for 64 tiles); with K = 1024 the
//             ~12 us of launch + prologue + epilogue per round leave 0.86-1.0 PFLOP/s at M = 30720 (tools/kbench.py)
//   T128x128L / T128x128W4L: the two 128 x 128 forms with FOUR LOADER WAVES behind the compute waves (gemm_body NL; bf16 only): 12 / 8 waves
enum GemmTile : int { T256x128 = 16, T128x128 = 17, T128x64 = 18, T64x64 = 19, T256x256 = 22, T128x128W4 = 26, T128x128L = 27, T128x128W4L = 28, T128x128W4L4 = 29, T128x64L = 30, T64x64L = 31 };

template <int TILE> struct TileCfg;
template <> struct TileCfg<T256x128> { static constexpr int BM = 256, BN = 128, ST = 3, WM = 4, WN = 2, NL = 0; };
template <> struct TileCfg<T128x128> { static constexpr int BM = 128, BN = 128, ST = 3, WM = 2, WN = 4, NL = 0; };
template <> struct TileCfg<T128x64>  { static constexpr int BM = 128, BN = 64,  ST = 3, WM = 2, WN = 2, NL = 0; };
template <> struct TileCfg<T64x64>   { static constexpr int BM = 64,  BN = 64,  ST = 3, WM = 2, WN = 2, NL = 0; };
template <> struct TileCfg<T128x128W4> { static constexpr int BM = 128, BN = 128, ST = 3, WM = 2, WN = 2, NL = 0; };
template <> struct TileCfg<T128x128L> { static constexpr int BM = 128, BN = 128, ST = 3, WM = 2, WN = 4, NL = 4; };
template <> struct TileCfg<T128x128W4L> { static constexpr int BM = 128, BN = 128, ST = 3, WM = 2, WN = 2, NL = 4; };
template <> struct TileCfg<T128x64L> { static constexpr int BM = 128, BN = 64, ST = 3, WM = 2, WN = 2, NL = 4; };
template <> struct TileCfg<T64x64L>  { static constexpr int BM = 64,  BN = 64, ST = 3, WM = 2, WN = 2, NL = 4; };
// ... and a FOUR-stage ring (128 KB): three K-tiles = 96 KB in flight per CU instead of two (the loop's slope follows bytes in flight / latency)
template <> struct TileCfg<T128x128W4L4> { static constexpr int BM = 128, BN = 128, ST = 4, WM = 2, WN = 2, NL = 4; };
// Tried and dropped in round 3 (profiles/r03/r03_structural_attempts.txt): the 8 waves as 4 x 2 (whole 128-B lines for the ln-fold image:
// +0.7 us plain, -0.7 us as fold producer), a TWO-stage ring (64 KB: two workgroups per CU; -2 ... -8 % end to end), and the K-tile
// split over two groups of 2 x 2 waves with 64 x 64 outputs that swap halves through LDS before the epilogue (a third less LDS read
// traffic, parity-green: the K loop is no faster -- 0.43 vs 0.44 us per K-tile, it is the lock-step structure, not LDS bandwidth --
// and the swap costs 1.5 us: 15.05 vs 13.43 us for the out-projection).

template <int EPI, int TILE, bool F8>
struct Launch {
  using C = TileCfg<TILE>;
  static constexpr int lds = body_lds_base<EPI, C::BM, C::BN, C::ST, C::WM, C::WN, F8>() + C::BM * 8;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static_assert(C::NL == 0 || !F8, "loader-wave tiles are bf16 tiles");
  static const void* fn() { return reinterpret_cast<const void*>(gemm_bf16_kernel<EPI, C::BM, C::BN, C::ST, C::WM, C::WN, F8, C::NL>); }
  // the > 64 KB dynamic-LDS opt-in; done once from lemas_kernels_init(), never on a launch path (a launch may sit inside a
  // stream capture)
  static hipError_t init() { return hipFuncSetAttribute(fn(), hipFuncAttributeMaxDynamicSharedMemorySize, lds); }
  static hipError_t run(const GemmParams& p, hipStream_t s) {
    if (p.N % C::BN != 0) return hipErrorInvalidValue;
    const int tiles_m = (p.M + C::BM - 1) / C::BM, tiles_n = p.N / C::BN;
    const dim3 grid(grid_of(tiles_m, tiles_n, p.xcd_gx, p.xcd_runs)), block(64 * (C::WM * C::WN + C::NL));
    // without the ln-fold table the launch keeps the ring's own footprint (128 x 128: exactly 96 KB, which with the attention kernel's
    // exact 64 KB is a CU's 160 KB -- measured: sharing or not sharing a CU that way changes nothing, profiles/r03/r03_structural_attempts.txt)
    const int lds_now = p.ln_part ? lds : lds - C::BM * 8;
    if (p.ev_start)
      hipExtLaunchKernelGGL((gemm_bf16_kernel<EPI, C::BM, C::BN, C::ST, C::WM, C::WN, F8, C::NL>), grid, block, lds_now, s, p.ev_start, p.ev_stop, 0, p);
    else
      hipLaunchKernelGGL((gemm_bf16_kernel<EPI, C::BM, C::BN, C::ST, C::WM, C::WN, F8, C::NL>), grid, block, lds_now, s, p);
    return hipGetLastError();
  }
};

// Largest tile that still yields about one workgroup per CU (measured at M = 1920 / 3840 / 18432 with tools/kbench.py).
// With two CFG lanes in flight each launch only needs half the chip (p.concurrency = 2: +1.8 % end to end for the larger tiles).
// the tile without loader waves that a loader-wave tile is built on (the LayerNorm-tail experiments run on those)
int plain_tile(int tile) {
  return tile == T128x128L ? T128x128 : (tile == T128x128W4L || tile == T128x128W4L4) ? T128x128W4 : tile == T128x64L ? T128x64 : tile == T64x64L ? T64x64 : tile;
}

int pick_tile_plain(const GemmParams& p);
// Round 6: where the rules below pick a lock-step tile for bf16 operands, its loader-wave form runs instead (bit-identical output).  Measured
// (profiles/r06/r06s-r06x): the 4 + 4-wave 128 x 128 form 11.1 vs 11.8 us for the out-projection and 18.0 vs 19.4 for FF2 at M = 1920 against
// the 8-wave tile it replaces, configs[1] +1.6 ... +2.1 % end to end; 64 x 64: 6.2 vs 7.2 / 9.4 vs 11.0 us at M = 768, 128 x 64: 7.9 vs 8.7, `short` +3.6 %.
// FF1's 4-wave 128 x 128 tile stays as it is (its loader form is 8 % faster alone and 0.8 % slower end to end, r06v).
// The 128 x 128 form only while both lanes' workgroups of the launch fit the chip at once (M = 2816, 176 tiles per lane: 1.0 % SLOWER end to end
// than the 8-wave tile, r06w); the fused QK + V launch keeps its plain tiles (128 x 64 at M = 768: the loader form costs `short` 3.5 %, r06x).
int pick_tile(const GemmParams& p) {
  const int tile = pick_tile_plain(p);
  if (p.tile || p.f8 || p.ln_out) return tile;
  const long conc = p.concurrency > 1 ? p.concurrency : 1;
  if (tile == T128x128) return (long)((p.M + 127) / 128) * (p.N / 128) * conc <= 256 ? T128x128W4L : tile;
  return tile == T128x64 ? T128x64L : tile == T64x64 ? T64x64L : tile;
}

int pick_tile_plain(const GemmParams& p) {
  if (p.tile) return p.tile;      // explicit tile: unit tests, kbench, the engine's measurement options (per engine, never process-global)
  const long conc = p.concurrency > 1 ? p.concurrency : 1;
  const long t256 = (long)((p.M + 255) / 256) * (p.N / 128), t128 = (long)((p.M + 127) / 128) * (p.N / 128);
  const long want = 200 / conc;
  int tile = t256 >= want ? T256x128 : t128 >= want ? T128x128 : T128x64;
  // short utterances: when even 128x64 tiles leave most CUs without work, halve the tile height (N <= 1024: out-proj, FF2, V)
  if (tile == T128x64 && p.N <= 1024 && (long)((p.M + 127) / 128) * (p.N / 64) < 130) tile = T64x64;
  // Two lanes in flight and a stand-alone N >= 2048 GEMM (FF1) of at most 256 128 x 128 tiles: the 4-wave form of that tile instead of the
  // 256 x 128 ping-pong tile.  End to end on configs[1] (M = 1875 per lane; tools/e2e_ab.py, ONE engine re-captured per arm, 6 interleaved
  // rounds, profiles/r02/r02_e2e_ab_w4_tile.txt): 87.1 -> 89.1 audio-s/s (+2.3 %); M = 1152: ties; M = 2304 (288 tiles): -2 %, M = 3456: -5 % (not
  // chosen there).  For the N = 1024 GEMMs it LOSES 1 % against the 8-wave 128 x 128 tile, for the fused QK+V launch 7 %.
  if (!p.f8 && conc >= 2 && p.N >= 2048 && (tile == T256x128 || tile == T128x128) && t128 <= 256) tile = T128x128W4;
  if (!p.f8 && p.N % 256 == 0 && p.N >= 1024) {
    // batched workloads: the ping-pong 256x256 tile once its tiles cover the CUs of this launch 1.5 times (N >= 2048) / 1.1 times
    // (N = 1024).  Measured end to end with two lanes in flight (tools/e2e_ab.py --batch B, profiles/r02/r02_e2e_ab_pingpong_batch.txt):
    // N = 2048 GEMMs: -3 % at M = 2304 per lane, +1 % at 4608, +8 % at 6912 and 9216; adding the N = 1024 GEMMs: -24 %, -3 %, -5 %,
    // +1.3 % (configs[3]'s share: 155 -> 169 audio-s/s).  The tile count, not a round-quantisation model, is what predicts it: with
    // two lanes sharing the chip a partial last round of one lane is filled by the other.
    const long cus = 256 / conc, tbig = (long)((p.M + 255) / 256) * (p.N / 256);
    if (p.N >= 2048 ? 2 * tbig >= 3 * cus : 10 * tbig >= 11 * cus) tile = T256x256;
  }
  return tile;
}

// the ping-pong bodies share the block-pipelined row epilogues, which do not write MXFP8: bf16 operands with EPI_BIAS_GELU_F8 (the fp8 path's
// FF1 under the outlier guard) run on the lock-step tiles
template <int EPI> constexpr bool pp_ok() { return EPI != EPI_BIAS_GELU_F8; }

template <int EPI, bool F8>
hipError_t dispatch(const GemmParams& p, int tile, hipStream_t s) {
  if (tile == 0) tile = pick_tile(p);
  if (p.ln_out) tile = plain_tile(tile);      // (measurement builds: the LayerNorm tail's barriers count the compute waves of the plain tiles)
  if (!F8 && !pp_ok<EPI>() && tile == T256x256) tile = T256x128;
  switch (tile) {
    case T256x128:
      if constexpr (!F8 && pp_ok<EPI>()) return LaunchPP2<EPI>::run(p, s);       // bf16: the ping-pong form (5-9 % faster alone, +0.9 % end to end for FF1)
      else return Launch<EPI, T256x128, F8>::run(p, s);
    case T128x128: return Launch<EPI, T128x128, F8>::run(p, s);
    case T128x64: return Launch<EPI, T128x64, F8>::run(p, s);
    case T64x64: return Launch<EPI, T64x64, F8>::run(p, s);
    case T128x128W4: return Launch<EPI, T128x128W4, F8>::run(p, s);
    case T128x128L:
      if constexpr (!F8) return Launch<EPI, T128x128L, false>::run(p, s);
      else return hipErrorInvalidValue;
    case T128x128W4L:
      if constexpr (!F8) return Launch<EPI, T128x128W4L, false>::run(p, s);
      else return hipErrorInvalidValue;
    case T128x128W4L4:
      if constexpr (!F8) return Launch<EPI, T128x128W4L4, false>::run(p, s);
      else return hipErrorInvalidValue;
    case T128x64L:
      if constexpr (!F8) return Launch<EPI, T128x64L, false>::run(p, s);
      else return hipErrorInvalidValue;
    case T64x64L:
      if constexpr (!F8) return Launch<EPI, T64x64L, false>::run(p, s);
      else return hipErrorInvalidValue;
    case T256x256:
      if constexpr (!F8 && pp_ok<EPI>()) return LaunchPP<EPI>::run(p, s);
      else return hipErrorInvalidValue;
    default: return hipErrorInvalidValue;
  }
}

template <int EPI, bool F8>
hipError_t init_epi() {
  hipError_t e;
  if ((e = Launch<EPI, T256x128, F8>::init()) != hipSuccess) return e;
  if ((e = Launch<EPI, T128x128, F8>::init()) != hipSuccess) return e;
  if ((e = Launch<EPI, T128x64, F8>::init()) != hipSuccess) return e;
  if ((e = Launch<EPI, T64x64, F8>::init()) != hipSuccess) return e;
  if ((e = Launch<EPI, T128x128W4, F8>::init()) != hipSuccess) return e;
  if constexpr (!F8) {
    if ((e = Launch<EPI, T128x128L, false>::init()) != hipSuccess) return e;
    if ((e = Launch<EPI, T128x128W4L, false>::init()) != hipSuccess) return e;
    if ((e = Launch<EPI, T128x128W4L4, false>::init()) != hipSuccess) return e;
    if ((e = Launch<EPI, T128x64L, false>::init()) != hipSuccess) return e;
    if ((e = Launch<EPI, T64x64L, false>::init()) != hipSuccess) return e;
  }
  if constexpr (!F8 && pp_ok<EPI>()) {
    if ((e = LaunchPP<EPI>::init()) != hipSuccess) return e;
    if ((e = LaunchPP2<EPI>::init()) != hipSuccess) return e;
  }
  return hipSuccess;
}

// QK (+RoPE) and V^T projections of one lane in ONE launch: they only share their input, so instead of two launches of ~half
// a chip each, run back to back, the first tiles_q workgroups take QK tiles and the remaining ones V tiles of the same shape:
// 256x128 at configs[1] (128 + 64 workgroups; measured 0.6 % faster end to end than 128x128 V tiles), 128x128 / 128x64 for short
// utterances, where 256-row tiles leave most of the chip idle (N = 750: 72 workgroups, 19.6 us -- as long as at N = 1875).
template <bool F8, int TILE>
struct QkvLds {
  using C = TileCfg<TILE>;
  static constexpr int NW = C::WM * C::WN;
  static constexpr int ring = C::ST * ((C::BM + C::BN) * 128 + (F8 ? C::BM * 4 : 0));
  static constexpr int slab_q = NW * slab_bytes<EPI_QK_ROPE, 32, C::BN / C::WN>(), slab_v = NW * slab_bytes<EPI_V_T, 32, C::BN / C::WN>();
  static constexpr int slab = slab_q > slab_v ? slab_q : slab_v;
  static constexpr int base = ring > slab ? ring : slab;      // then the ln-fold (r, -r mu) table of the tile's rows
  static constexpr int lds = base + C::BM * 8;
};

template <bool F8, int TILE>
__global__ __launch_bounds__(64 * (TileCfg<TILE>::WM * TileCfg<TILE>::WN + TileCfg<TILE>::NL)) void gemm_qkv_fused_kernel(const GemmParams pq, const GemmParams pv,
                                                                                                      int tiles_q, int tiles_v) {
  using C = TileCfg<TILE>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int bid = blockIdx.x;
  float* rs = reinterpret_cast<float*>(smem + QkvLds<F8, TILE>::base);
  int tm, tn;
  if (bid < tiles_q) {       // tiles_q / tiles_v: the PADDED workgroup counts of the two parts (multiples of 8: both parts keep the XCD phase)
    kernargs_early(pq);
    if (!tile_of(bid, (pq.M + C::BM - 1) / C::BM, pq.N / C::BN, pq.xcd_gx, pq.xcd_runs, tm, tn) || tile_dead(pq, tm * C::BM, C::BM)) return;
    if constexpr (!F8 && TILE == T256x128) gemm_body_pp2<EPI_QK_ROPE, true>(pq, smem, tm * 256, tn * 128, rs);
    else gemm_body<EPI_QK_ROPE, C::BM, C::BN, C::ST, C::WM, C::WN, true, F8, false, C::NL>(pq, smem, tm * C::BM, tn * C::BN, rs);
  } else {
    kernargs_early(pv);
    if (!tile_of(bid - tiles_q, (pv.M + C::BM - 1) / C::BM, pv.N / C::BN, pv.xcd_gx, pv.xcd_runs, tm, tn) || tile_dead(pv, tm * C::BM, C::BM)) return;
    if constexpr (!F8 && TILE == T256x128) gemm_body_pp2<EPI_V_T, false>(pv, smem, tm * 256, tn * 128, rs);
    else gemm_body<EPI_V_T, C::BM, C::BN, C::ST, C::WM, C::WN, false, F8, false, C::NL>(pv, smem, tm * C::BM, tn * C::BN, rs);
  }
}

template <bool F8, int TILE>
struct LaunchQkv {
  using C = TileCfg<TILE>;
  static constexpr int NW = C::WM * C::WN;
  static constexpr int lds = QkvLds<F8, TILE>::lds;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static hipError_t init() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_qkv_fused_kernel<F8, TILE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  static hipError_t run(const GemmParams& pq, const GemmParams& pv, hipStream_t s) {
    const int tiles_q = grid_of((pq.M + C::BM - 1) / C::BM, pq.N / C::BN, pq.xcd_gx, pq.xcd_runs), tiles_v = grid_of((pv.M + C::BM - 1) / C::BM, pv.N / C::BN, pv.xcd_gx, pv.xcd_runs);
    const dim3 grid(tiles_q + tiles_v), block(64 * (NW + C::NL));
    const int lds_now = (pq.ln_part || pv.ln_part) ? lds : lds - C::BM * 8;
    if (pq.ev_start)
      hipExtLaunchKernelGGL((gemm_qkv_fused_kernel<F8, TILE>), grid, block, lds_now, s, pq.ev_start, pq.ev_stop, 0, pq, pv, tiles_q, tiles_v);
    else
      hipLaunchKernelGGL((gemm_qkv_fused_kernel<F8, TILE>), grid, block, lds_now, s, pq, pv, tiles_q, tiles_v);
    return hipGetLastError();
  }
};

// Several independent GEMMs of the same epilogue in ONE launch (blockIdx.y picks a parameter block from device memory, 128 x 128
// lock-step tiles): the 2 x depth small ln-fold table GEMMs of a prepare() ([4 S rows] x [3 inner | ff] x dim each) would otherwise be
// 44 launches of ~20 workgroups.
template <int EPI>
__global__ __launch_bounds__(512) void gemm_group_kernel(const GemmParams* __restrict__ ps) {
  using C = TileCfg<T128x128>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const GemmParams& p = ps[blockIdx.y];
  const int tiles_n = p.N / C::BN, tiles = ((p.M + C::BM - 1) / C::BM) * tiles_n;
  const int bid = blockIdx.x;
  if (bid >= tiles) return;
  gemm_body<EPI, C::BM, C::BN, C::ST, C::WM, C::WN, true, false>(p, smem, (bid / tiles_n) * C::BM, (bid % tiles_n) * C::BN,
      reinterpret_cast<float*>(smem + body_lds_base<EPI, C::BM, C::BN, C::ST, C::WM, C::WN, false>()));
}


// (measurement builds only: the persistent FF-half kernel, engine option "block_persist")
#ifdef LEMAS_MEASUREMENT_BUILD
#include "gemm_bf16_exp_chain.h"
#endif

// tile of the fused QK+V launch: the largest whose QK part alone still gives ~100 workgroups per lane (two lanes share the chip)
int pick_qkv_tile(const GemmParams& pq) {
  if (pq.tile == T256x128 || pq.tile == T128x128 || pq.tile == T128x64 || pq.tile == T128x128W4) return pq.tile;
  if (!pq.f8 && (pq.tile == T128x128W4L || pq.tile == T128x64L)) return pq.tile;
  const long t256 = (long)((pq.M + 255) / 256) * (pq.N / 128), t128 = (long)((pq.M + 127) / 128) * (pq.N / 128);
  // (the loader-wave forms of the lock-step tiles are reachable by explicit tile only: measured slower in this launch, see pick_tile)
  return t256 >= 100 ? T256x128 : t128 >= 100 ? T128x128 : T128x64;
}

}  // namespace

hipError_t gemm_bf16_init() {
  hipError_t e;
#define LEMAS_INIT(EPI)                                                   \
  if ((e = init_epi<EPI, false>()) != hipSuccess) return e;               \
  if ((e = init_epi<EPI, true>()) != hipSuccess) return e;
  LEMAS_INIT(EPI_BIAS_BF16) LEMAS_INIT(EPI_BIAS_GELU_BF16) LEMAS_INIT(EPI_BIAS_F32) LEMAS_INIT(EPI_GATE_RES) LEMAS_INIT(EPI_QK_ROPE)
  LEMAS_INIT(EPI_V_T)
#undef LEMAS_INIT
  if ((e = init_epi<EPI_BIAS_GELU_F8, true>()) != hipSuccess) return e;
  if ((e = init_epi<EPI_BIAS_GELU_F8, false>()) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_group_kernel<EPI_BIAS_F32>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               Launch<EPI_BIAS_F32, T128x128, false>::lds)) != hipSuccess) return e;
  if ((e = LaunchQkv<false, T256x128>::init()) != hipSuccess) return e;
  if ((e = LaunchQkv<false, T128x128>::init()) != hipSuccess) return e;
  if ((e = LaunchQkv<false, T128x64>::init()) != hipSuccess) return e;
  if ((e = LaunchQkv<false, T128x128W4>::init()) != hipSuccess) return e;
  if ((e = LaunchQkv<false, T128x128W4L>::init()) != hipSuccess) return e;
  if ((e = LaunchQkv<false, T128x64L>::init()) != hipSuccess) return e;
  if ((e = LaunchQkv<true, T128x128W4>::init()) != hipSuccess) return e;
  if ((e = LaunchQkv<true, T256x128>::init()) != hipSuccess) return e;
  if ((e = LaunchQkv<true, T128x128>::init()) != hipSuccess) return e;
  return LaunchQkv<true, T128x64>::init();
}

// ln fold: what a launch that consumes (ln_part) or produces (xs_out) the folded LayerNorm must look like
static bool ln_fold_ok(int epi, const GemmParams& p) {
  if (p.ln_part) {
    if (!(epi == EPI_QK_ROPE || epi == EPI_V_T || epi == EPI_BIAS_GELU_BF16)) return false;
    if (p.f8 || p.K != LN_D || p.ln_np != LN_NP || !p.tab || !p.step_idx) return false;
  }
  if (p.xs_out) {
    if (epi != EPI_GATE_RES || p.f8 || p.N != LN_D || p.ldc != LN_D || p.n_valid != LN_D || p.ln_np != LN_NP || !p.ln_part_out || !p.tab ||
        !p.step_idx)
      return false;
  }
  return true;
}

// measurement builds only: the FF half of a block as one persistent launch (gemm_chain_ffhalf_kernel); hipErrorNotSupported in the product
hipError_t launch_gemm_chain_ffhalf(const GemmParams& out_in, const GemmParams& ff1_in, const GemmParams& ff2_in, int ln_scale_off, int ln_shift_off,
                                    unsigned int* sync, unsigned int* err, int prefetch, int* grid_out, hipStream_t s) {
#ifdef LEMAS_MEASUREMENT_BUILD
  ChainParams c{};
  c.out = out_in; c.ff1 = ff1_in; c.ff2 = ff2_in;
  for (GemmParams* p : {&c.out, &c.ff1, &c.ff2})
    if (p->xcd_gx != 8 && p->xcd_gx != 4 && p->xcd_gx != 2 && p->xcd_gx != 1) p->xcd_gx = pick_xcd_gx(p->M, p->N);
  if (c.out.f8 || c.ff1.f8 || c.ff2.f8 || c.out.N != LN_D || c.ff2.N != LN_D || c.ff1.K != LN_D || c.out.M != c.ff1.M || c.out.M != c.ff2.M ||
      c.out.K % BK || c.ff2.K % BK || c.ff1.N % 128 || !sync || !err || c.out.ln_part || c.ff1.ln_part || c.out.xs_out || c.ff2.xs_out)
    return hipErrorInvalidValue;
  c.ln_scale_off = ln_scale_off; c.ln_shift_off = ln_shift_off; c.sync = sync; c.err = err; c.prefetch = prefetch;
  using C = TileCfg<T128x128>;
  const int grid = xcd_grid((c.out.M + 127) / 128, c.out.N / 128, c.out.xcd_gx);       // the N = 1024 stages' grid; FF1 walks its tiles in strides of it
  if (grid_out) *grid_out = grid;
  constexpr int lds = body_lds_base<EPI_GATE_RES, C::BM, C::BN, C::ST, C::WM, C::WN, false>() + C::BM * 8 + 16;
  static std::once_flag once;
  static hipError_t init_err = hipSuccess;
  std::call_once(once, []() {
    init_err = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_chain_ffhalf_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  });
  if (init_err != hipSuccess) return init_err;
  if (s == nullptr) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gemm_chain_ffhalf_kernel, dim3(grid), dim3(512), lds, s, c);
  return hipGetLastError();
#else
  (void)out_in; (void)ff1_in; (void)ff2_in; (void)ln_scale_off; (void)ln_shift_off; (void)sync; (void)err; (void)prefetch; (void)grid_out; (void)s;
  return hipErrorNotSupported;
#endif
}

hipError_t launch_gemm_bf16_group(const GemmParams* dev_params, int groups, int max_tiles, hipStream_t s) {
  if (groups <= 0 || max_tiles <= 0) return hipErrorInvalidValue;
  constexpr int lds = Launch<EPI_BIAS_F32, T128x128, false>::lds;
  hipLaunchKernelGGL((gemm_group_kernel<EPI_BIAS_F32>), dim3(max_tiles, groups), dim3(512), lds, s, dev_params);
  return hipGetLastError();
}

hipError_t launch_gemm_qkv_fused(const GemmParams& pq_in, const GemmParams& pv_in, hipStream_t s) {
  GemmParams pq = pq_in, pv = pv_in;
  if (!ln_fold_ok(EPI_QK_ROPE, pq) || !ln_fold_ok(EPI_V_T, pv) || pq.xs_out || pv.xs_out) return hipErrorInvalidValue;
  // only {8, 4, 2, 1} cut the tile grid into 8 XCD blocks (anything else would make tile_coords divide by zero or skip tiles)
  if (pq.xcd_gx != 8 && pq.xcd_gx != 4 && pq.xcd_gx != 2 && pq.xcd_gx != 1) pq.xcd_gx = pick_xcd_gx(pq.M, pq.N);
  if (pv.xcd_gx != 8 && pv.xcd_gx != 4 && pv.xcd_gx != 2 && pv.xcd_gx != 1) pv.xcd_gx = pick_xcd_gx(pv.M, pv.N);
  if (pq.K % 128 != 0 || pq.N % 128 != 0 || pv.N % 128 != 0 || pq.M != pv.M || pq.f8 != pv.f8 || pq.M <= 0) return hipErrorInvalidValue;
  if (pq.f8 && (!pq.a_mx || !pq.w_scale || !pv.w_scale)) return hipErrorInvalidValue;
  switch (pick_qkv_tile(pq)) {
    case T256x128: return pq.f8 ? LaunchQkv<true, T256x128>::run(pq, pv, s) : LaunchQkv<false, T256x128>::run(pq, pv, s);
    case T128x128: return pq.f8 ? LaunchQkv<true, T128x128>::run(pq, pv, s) : LaunchQkv<false, T128x128>::run(pq, pv, s);
    case T128x128W4: return pq.f8 ? LaunchQkv<true, T128x128W4>::run(pq, pv, s) : LaunchQkv<false, T128x128W4>::run(pq, pv, s);
    case T128x128W4L: return LaunchQkv<false, T128x128W4L>::run(pq, pv, s);      // (pick_qkv_tile returns the loader forms for bf16 operands only)
    case T128x64L: return LaunchQkv<false, T128x64L>::run(pq, pv, s);
    default: return pq.f8 ? LaunchQkv<true, T128x64>::run(pq, pv, s) : LaunchQkv<false, T128x64>::run(pq, pv, s);
  }
}

// tiles whose gate + residual kernel carries the LayerNorm tail (the lock-step body; the ping-pong tiles serve batched shapes,
// whose grids do not fit the chip at once anyway): rows x columns of the tile and how many of its workgroups share a CU
static bool ln_tile_shape(int tile, int* bm, int* bn, int* per_cu) {
  switch (tile) {
    case T128x128: case T128x128W4: *bm = 128; *bn = 128; *per_cu = 1; return true;   // 96 KB of LDS each (the loader-wave forms carry no tail)
    case T128x64: *bm = 128; *bn = 64; *per_cu = 2; return true;                      // 72 KB
    case T64x64: *bm = 64; *bn = 64; *per_cu = 2; return true;                        // 48 KB (3 would fit: counted as 2)
    default: return false;
  }
}

int gemm_bf16_ln_fusable(const GemmParams& p, int* panels, int* per_cu) {
#ifndef LEMAS_MEASUREMENT_BUILD
  return 0;
#endif
  int bm, bn;
  if (p.f8 || p.N != LN_D || p.ldc != LN_D || p.n_valid != LN_D || p.M <= 0) return 0;
  if (!ln_tile_shape(plain_tile(pick_tile(p)), &bm, &bn, per_cu) || p.M % bm != 0) return 0;
  *panels = p.M / bm;
  return (p.M / bm) * (p.N / bn);
}

hipError_t launch_gemm_bf16_tile(int epi, const GemmParams& p_in, int tile, hipStream_t s) {
  GemmParams p = p_in;
  if (p.xcd_gx != 8 && p.xcd_gx != 4 && p.xcd_gx != 2 && p.xcd_gx != 1) p.xcd_gx = pick_xcd_gx(p.M, p.N);
  if (p.K % BK != 0 || p.N % 128 != 0 || p.M <= 0 || p.seq_pitch <= 0) return hipErrorInvalidValue;
  if (!ln_fold_ok(epi, p)) return hipErrorInvalidValue;
  if (p.ln_out) {   // LayerNorm tail: only on the tiles that carry it, with complete row panels and the counters in place
#ifndef LEMAS_MEASUREMENT_BUILD
    return hipErrorInvalidValue;      // the tail's device code is compiled into measurement builds only (common.h)
#endif
    int bm, bn, per_cu;
    if (epi != EPI_GATE_RES || p.f8 || p.N != LN_D || p.ldc != LN_D || p.n_valid != LN_D || !p.ln_cnt || !p.ln_err || !p.tab || !p.step_idx)
      return hipErrorInvalidValue;
    if (!ln_tile_shape(plain_tile(tile ? tile : pick_tile(p)), &bm, &bn, &per_cu) || p.M % bm != 0) return hipErrorInvalidValue;
  }
  // the row-wise epilogues store whole 16-B chunks: 4 fp32 / 8 bf16 / 16 e4m3 columns, so the stored width and the row
  // pitch must be multiples of that (every shape of the path is: 100, 1024, 2048)
  if (epi == EPI_BIAS_F32 || epi == EPI_GATE_RES) { if ((p.n_valid > 0 && p.n_valid % 4) || p.ldc % 4) return hipErrorInvalidValue; }
  if (epi == EPI_BIAS_BF16 || epi == EPI_BIAS_GELU_BF16) { if (p.n_valid % 8 || p.ldc % 8) return hipErrorInvalidValue; }
  if (epi == EPI_BIAS_GELU_F8) { if (p.n_valid % 128 || p.ldc % 128) return hipErrorInvalidValue; }
  if (p.f8) {
    if (p.K % 128 != 0 || !p.a_mx || !p.w_scale) return hipErrorInvalidValue;
    switch (epi) {
      case EPI_BIAS_GELU_BF16: return dispatch<EPI_BIAS_GELU_BF16, true>(p, tile, s);
      case EPI_BIAS_GELU_F8: return dispatch<EPI_BIAS_GELU_F8, true>(p, tile, s);
      case EPI_BIAS_F32: return dispatch<EPI_BIAS_F32, true>(p, tile, s);
      case EPI_GATE_RES: return dispatch<EPI_GATE_RES, true>(p, tile, s);
      case EPI_QK_ROPE: return dispatch<EPI_QK_ROPE, true>(p, tile, s);
      case EPI_V_T: return dispatch<EPI_V_T, true>(p, tile, s);
    }
    return hipErrorInvalidValue;
  }
  switch (epi) {
    case EPI_BIAS_GELU_F8:      // bf16 operands, MXFP8 output: FF1 of the fp8 path when the outlier guard keeps its operands bf16
      if (!p.out_f8 || !p.out_mx) return hipErrorInvalidValue;
      return dispatch<EPI_BIAS_GELU_F8, false>(p, tile, s);
    case EPI_BIAS_BF16: return dispatch<EPI_BIAS_BF16, false>(p, tile, s);
    case EPI_BIAS_GELU_BF16: return dispatch<EPI_BIAS_GELU_BF16, false>(p, tile, s);
    case EPI_BIAS_F32: return dispatch<EPI_BIAS_F32, false>(p, tile, s);
    case EPI_GATE_RES: return dispatch<EPI_GATE_RES, false>(p, tile, s);
    case EPI_QK_ROPE: return dispatch<EPI_QK_ROPE, false>(p, tile, s);
    case EPI_V_T: return dispatch<EPI_V_T, false>(p, tile, s);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_gemm_bf16(int epi, const GemmParams& p, hipStream_t s) { return launch_gemm_bf16_tile(epi, p, p.tile, s); }
