// Shared device helpers and internal launch prototypes for liblemas_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

#define LEMAS_WAVE 64

// Measurement builds (LEMAS_EXTRA_HIPCC_FLAGS=-DLEMAS_MEASUREMENT_BUILD; implied by -DLEMAS_PHASE_TIMESTAMPS): the experiments that were measured,
// lost and are kept reproducible -- the LayerNorm tail inside the gate + residual GEMM launch (engine option "ln_fused"), the lanes one stage apart
// ("lane_skew"), round 3's XCD tile order ("xcd_runs") -- exist only there.  The product library carries neither their device code nor their options
// (lemas_dit_set_option refuses them).
#if defined(LEMAS_PHASE_TIMESTAMPS) && !defined(LEMAS_MEASUREMENT_BUILD)
#define LEMAS_MEASUREMENT_BUILD 1
#endif

// 16-byte global store with the sc1 (write-through) cache policy: the line goes to memory now instead of sitting dirty
// in this XCD's L2 until the end-of-kernel write-back, which otherwise serialises ~B/6 TB/s behind every producer
// kernel (MI355X_MICROARCH.md price list, rows 'boundary' and 'publish-large').  Consumers run on other XCDs anyway
// (private, non-coherent L2s), so nothing is lost by not keeping the line.  The asm store is invisible to the
// compiler's waitcnt bookkeeping; s_nop 1 keeps the data registers intact until the store has read them.
__device__ __forceinline__ void store_wt_b128(void* p, unsigned int __attribute__((ext_vector_type(4))) v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ void store_wt_b64(void* p, unsigned int __attribute__((ext_vector_type(2))) v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// Wave-wide sum / max, every lane gets the result.  The wave must be fully active (all call sites branch wave-uniformly).
// Four DPP steps fold the 16 lanes of a row (xor 1, xor 2 as quad permutes, then row_half_mirror and row_mirror: by then every lane of a
// quad / half row holds the same partial), four v_readlane + three adds fold the rows.  The __shfl_xor butterfly this replaces is six
// DEPENDENT ds_bpermute round trips per reduction: 24 of them in a row were 1.4 us of the 5 us LayerNorm launch.
template <int CTRL>
__device__ __forceinline__ float dpp_lanes_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_lanes_f<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_lanes_f<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_lanes_f<0x141>(v);    // row_half_mirror
  v += dpp_lanes_f<0x140>(v);    // row_mirror
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_lanes_f<0xB1>(v));
  v = fmaxf(v, dpp_lanes_f<0x4E>(v));
  v = fmaxf(v, dpp_lanes_f<0x141>(v));
  v = fmaxf(v, dpp_lanes_f<0x140>(v));
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// Activations: v_exp_f32 / v_rcp_f32 directly (1 ulp class) -- a precise fp32 division costs ~10 VALU instructions and
// these sit in GEMM / conv epilogues that run 64 values per lane.
__device__ __forceinline__ float fast_sigmoid(float z) {   // 1 / (1 + e^-z); z -> -inf gives rcp(inf) = 0
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
// GELU, tanh form, two values at a time: 0.5 x (1 + tanh(u)) = x * sigmoid(2u) = x / (1 + 2^(-log2(e) 2u)),  u = sqrt(2/pi) (x + 0.044715 x^3),
// with -log2(e) folded into the polynomial: 2^(x (c1 + c2 x^2)).  Written on 2-vectors so that the five non-transcendental steps are packed
// fp32 instructions (v_pk_mul / v_pk_fma / v_pk_add: one issue slot per PAIR): the GEMM epilogues this sits in are bound by VALU issue, and
// the two quarter-rate instructions per value (v_exp, v_rcp) are 8 of its slots either way.  x -> -inf: 2^(+inf) = inf, rcp = 0, x * 0 = -0.
__device__ __forceinline__ f32x2 gelu_tanh_f2(f32x2 x) {
  const f32x2 c1 = {-2.302208198144325f, -2.302208198144325f}, c2 = {-0.1029432395800235f, -0.1029432395800235f}, one = {1.0f, 1.0f};
  const f32x2 z = x * (x * x * c2 + c1);
  f32x2 d;
  d.x = __builtin_amdgcn_exp2f(z.x); d.y = __builtin_amdgcn_exp2f(z.y);
  d = d + one;
  f32x2 r;
  r.x = __builtin_amdgcn_rcpf(d.x); r.y = __builtin_amdgcn_rcpf(d.y);
  return x * r;
}
__device__ __forceinline__ float gelu_tanh_f(float x) {
  const f32x2 v = {x, x};
  return gelu_tanh_f2(v).x;
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }
__device__ __forceinline__ float silu_f(float x) { return x * fast_sigmoid(x); }
__device__ __forceinline__ float mish_f(float x) {
  // x * tanh(softplus(x)); with n = e^x (e^x + 2): tanh(log(1+e^x)) = n / (n + 2).  softplus threshold 20 as torch.
  const float e = __builtin_amdgcn_exp2f(1.4426950408889634f * fminf(x, 20.0f));
  const float n = e * (e + 2.0f);
  return x * n * __builtin_amdgcn_rcpf(n + 2.0f);
}

// ---- MXFP8 activations (OCP microscaling: fp8 e4m3 elements, one E8M0 power-of-two scale per 32 consecutive K) ----
// smallest e with amax * 2^-e <= 448 (e4m3 max), clamped so 2^e and 2^-e stay normal fp32; byte stored = e + 127.
// oracle/mxfp8.py restates exactly this arithmetic in torch.
__device__ __forceinline__ int mx_exponent(float amax) {
  const unsigned int b = __float_as_uint(amax * (1.0f / 448.0f));
  int e = (int)((b >> 23) & 255u) - 127 + ((b & 0x7fffffu) ? 1 : 0);
  return e < -120 ? -120 : (e > 120 ? 120 : e);
}
__device__ __forceinline__ float mx_inv_scale(int e) { return __uint_as_float((unsigned int)(127 - e) << 23); }
// four fp32 -> four e4m3 bytes (round to nearest even), clamped to +-448 first
__device__ __forceinline__ unsigned int pack_fp8x4(float a, float b, float c, float d) {
  a = __builtin_amdgcn_fmed3f(a, -448.0f, 448.0f); b = __builtin_amdgcn_fmed3f(b, -448.0f, 448.0f);
  c = __builtin_amdgcn_fmed3f(c, -448.0f, 448.0f); d = __builtin_amdgcn_fmed3f(d, -448.0f, 448.0f);
  int v = 0;
  v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
  return (unsigned int)v;
}

// ---- XCD-aware tile order of the GEMM launches (gemm_bf16.hip, gemm_f32.hip) -------------------------------------------------------
// Workgroups are dispatched round-robin over the 8 XCDs (private, non-coherent L2s): workgroup `bid` runs on XCD bid & 7.  The tile grid is cut
// into gx x gy blocks (gx * gy = 8), one per XCD, so that an XCD fetches 1 / gx of the A panels and 1 / gy of the W panels instead of all of
// both; workgroup `bid` takes tile (bid >> 3) of block (bid & 7), the launch holds 8 x (largest block) workgroups and the few beyond a
// smaller block's end return at once.
__host__ __device__ __forceinline__ int xcd_grid(int tiles_m, int tiles_n, int gx) {
  const int gy = 8 / gx;
  const int bm = (tiles_m + gx - 1) / gx, bn = (tiles_n + gy - 1) / gy;
  return 8 * bm * bn;
}
__device__ __forceinline__ bool xcd_tile_coords(int bid, int tiles_m, int tiles_n, int gx, int& tm, int& tn) {
  // gx is 1, 2, 4 or 8: shifts, not divisions (every workgroup runs this before it can request its first operand tile, and a division by a
  // run-time value is ~30 dependent scalar instructions: five of them were ~0.2 us of every GEMM workgroup's start)
  const int lg = 31 - __builtin_clz((unsigned)gx), lgy = 3 - lg, gy = 1 << lgy;
  const int bm = (tiles_m + gx - 1) >> lg, bn = (tiles_n + gy - 1) >> lgy;
  const int blk = bid & 7, idx = bid >> 3;
  const int bi = blk >> lgy, bj = blk & (gy - 1);
  const int rows = min(bm, tiles_m - bi * bm), cols = min(bn, tiles_n - bj * bn);
  if (rows <= 0 || cols <= 0 || idx >= rows * cols) return false;
  const int r = idx / cols;
  tm = bi * bm + r;
  tn = bj * bn + (idx - r * cols);
  return true;
}
// XCD block grid: fabric-side fetch ~ gy * |A| + gx * |W| -> minimise gy * M + gx * N
static inline int pick_xcd_gx(int M, int N) {
  int best = 8;
  long cost = -1;
  for (int gx : {8, 4, 2, 1}) {
    const long c = (long)(8 / gx) * M + (long)gx * N;
    if (cost < 0 || c < cost) { cost = c; best = gx; }
  }
  return best;
}

// ---- epilogue selectors of the bf16 MFMA GEMM -------------------------------------------------
enum GemmEpi : int {
  EPI_BIAS_BF16 = 0,       // out_bf16[m][n] = acc + bias
  EPI_BIAS_GELU_BF16 = 1,  // out_bf16[m][n] = gelu_tanh(acc + bias)
  EPI_BIAS_F32 = 2,        // out_f32[m][n] = acc + bias      (n < n_valid)
  EPI_GATE_RES = 3,        // res_f32[m][n] += gate[n] * (acc + bias)   (rows past kv_len contribute 0)
  EPI_QK_ROPE = 4,         // N = 2*inner: +bias, RoPE, scatter to q / k [B2,H,pitch,64]
  EPI_V_T = 5,             // N = inner:   +bias, scatter to v^T [B2,H,64,npad]
  EPI_BIAS_GELU_F8 = 7,    // out_f8 / out_mx = MXFP8(gelu_tanh(acc + bias))   (fp8 path only)
};

struct GemmParams {
  const bf16_t* A;  // [M, K] row-major activations
  const bf16_t* W;  // [Nw, K] row-major weights (nn.Linear layout), Nw padded to a multiple of 128
  const float* bias;  // [Nw]
  int M, N, K;        // N = padded Nw
  int n_valid;        // columns actually stored
  // outputs
  bf16_t* out_bf16;
  float* out_f32;
  int ldc;
  // gate/residual epilogue: gate vector lives in the per-step AdaLN table
  const float* tab;      // table base (step 0)
  int tab_stride;        // floats per step
  int gate_off;          // offset of the gate vector inside a step row
  const int* step_idx;   // device scalar: current ODE step
  const int* kv_len;     // [B] valid frames per sample or nullptr
  // ragged batches: [B] valid frames per sample; a tile whose 128-row blocks all start at or past their sample's length is not computed at
  // all (row_block_dead below).  nullptr = every tile is computed.  Every kernel of the block chain uses the same 128-row granularity, so a
  // live block's rows are produced by every kernel exactly as without the switch and a dead block's rows are never read by a live one
  // (GEMM rows are independent; attention reads keys < kv_len only).  The reference computes those rows and throws them away
  // (cfm.py:336-339 mask, utils_infer.py:579-585 trims each sample to its duration).
  const int* live_len;
  int seq_pitch;         // rows per sample in the activation row space (multiple of 128): m -> (m / pitch, m % pitch)
  int seq_valid;         // real frames per sample (rows with m % pitch >= seq_valid are padding)
  int batch;             // B (kv_len index = (m / seq_pitch) % batch)
  // qkv epilogue
  bf16_t* q;
  bf16_t* k;
  bf16_t* vt;
  const float* rope_cos;  // [seq_valid, 32]
  const float* rope_sin;
  int heads, npad;
  float q_scale;          // EPI_QK_ROPE: the q half is multiplied by this in fp32 before the bf16 rounding (0 = 1.0): the attention
                          // kernel's "prescaled q" variants take softmax_scale * log2(e) here (attention.hip VAR & 16)
  // EPI_QK_ROPE of the fp8 bodies (f8 != 0), optional: when q8 is set, q and k leave as MXFP8 instead of bf16 -- the rotated (and prescaled) values are rounded to bf16
  // exactly as before and THEN quantised, one E8M0 scale per 32-wide half of a head (attention.hip VAR & ATTN_F8QK consumes them)
  uint8_t* q8;            // [B2, H, seq_pitch, 64] e4m3 (nullptr: bf16 q / k as above)
  uint8_t* k8;
  uint8_t* q8_mx;         // [B2, H, seq_pitch, 2] E8M0
  uint8_t* k8_mx;
  // fp8 path (f8 != 0): A and W point at e4m3 bytes (same [rows][K] layouts, K % 128 == 0); activations carry MX block
  // scales, weights one fp32 scale per output channel (applied in the epilogue)
  int f8;
  const uint8_t* a_mx;    // [M][K/32] E8M0 bytes
  const float* w_scale;   // [Nw]
  uint8_t* out_f8;        // EPI_BIAS_GELU_F8: [M][ldc] e4m3
  uint8_t* out_mx;        //                   [M][ldc/32]
  // profiling: when set, the launch goes through hipExtLaunchKernelGGL, which stamps these events with the dispatch's own
  // begin / end times (what rocprofv3 --kernel-trace reports), instead of bracketing the launch with stream events
  hipEvent_t ev_start, ev_stop;
  // how many launches of this size run concurrently (the CFG lanes): the tile heuristic aims at ~256 / concurrency
  // workgroups, i.e. larger, more efficient tiles when another lane's kernel fills the other half of the chip.  0 = 1.
  int concurrency;
  // XCD block grid of the tile order: the tile grid is cut into xcd_gx x (8 / xcd_gx) blocks, one per XCD (gemm_bf16.hip
  // tile_coords).  0 = let the launcher choose (minimises the fabric-side fetch gy * |A| + gx * |W|); 8 = the row-major order.
  int xcd_gx;
  // measurement switch: != 0 = round 3's tile order (every XCD takes an equal-length RUN of the blocked tile sequence, which straddles blocks of
  // a ragged grid) instead of one block per XCD; for A/B runs of the two orders in one process (engine option "xcd_runs")
  int xcd_runs;
  // explicit tile shape (ids as for launch_gemm_bf16_tile; 0 = the production choice).  Set per launch by whoever builds the
  // parameters -- unit tests, kbench, the engine's per-engine measurement options -- never by process-global state.
  int tile;
  // LayerNorm-modulate tail of EPI_GATE_RES (ln_out != nullptr; bf16 path, N == ldc == 1024): the AdaLN-modulated LayerNorm that
  // follows the residual update (modules.py:637 / the next block's :314 / the final :335) is computed by the SAME launch.  Every
  // workgroup publishes its x_res tile (write-through stores, drained), bumps its row panel's arrival counter, waits until the
  // panel's N / BN column tiles have arrived and then normalises BM / (N / BN) rows of the panel -> ln_out [M][1024] bf16.
  // Needs every workgroup of the launch co-resident (gemm_bf16_ln_fusable); ln_cnt must be zero when the launch starts.
  bf16_t* ln_out;
  int ln_scale_off, ln_shift_off;   // offsets of the scale / shift vectors inside the AdaLN table row of the current step
  unsigned int* ln_cnt;             // [ceil(M / BM)] arrival counters
  unsigned int* ln_err;             // sticky error word (host-visible): set when a wait gives up instead of hanging
  // ---- LayerNorm folded ACROSS the GEMMs ("ln fold"; bf16 path, row width D = 32 * ln_np) -------------------------------------------
  // The AdaLN-modulated LayerNorm between a gated residual update and the GEMM that consumes it is linear in everything but the two
  // row statistics:   ((x - mu) r (1 + s) + b) . W^T + bias  =  r (x (1 + s)) . W^T  -  r mu c1  +  c2,
  //   c1[n] = sum_k (1 + s_k) W[n][k],   c2[n] = sum_k b_k W[n][k] + bias[n]      (per ODE step and site: hoisted into the AdaLN table)
  // so the producer writes the SCALED bf16 image of its new rows plus per-row partial sums, the consumer applies r / mu in its
  // epilogue, and the separate LayerNorm launch (with its two dependent-launch boundaries) disappears from the lane's chain.
  // producer (EPI_GATE_RES, xs_out != nullptr): xs_out[m][n] = bf16(x_new[m][n] * (1 + tab[step][xs_scale_off + n])) and
  //   ln_part_out[m][slot][2] = (sum x_new, sum x_new^2) over the 32 columns of slot n / 32 (every row m < M, updated or not)
  bf16_t* xs_out;
  int xs_scale_off;
  float* ln_part_out;
  // consumer (any row epilogue and EPI_V_T; A = that xs image): ln_part != nullptr -> acc + bias becomes
  //   r_m acc - r_m mu_m c1[n] + c2[n],  c1 = tab[step][lnc1_off + n], c2 = tab[step][lnc2_off + n]  (p.bias is not read)
  const float* ln_part;
  int ln_np;                        // 32-column slots per row (D / 32) of ln_part / ln_part_out
  int lnc1_off, lnc2_off;
#ifdef LEMAS_PHASE_TIMESTAMPS
  unsigned long long* dbg;   // measurement builds: per-workgroup phase timestamps [grid][4] (gemm_bf16.hip PHASE_STAMP)
#endif
};

// ---- internal launchers (one per .hip translation unit) ----------------------------------------
hipError_t launch_gemm_bf16(int epi, const GemmParams& p, hipStream_t s);
// explicit tile shape (16 = 256x128, 17 = 128x128, 18 = 128x64, 19 = 64x64, 22 = 256x256, 26 = 128x128 with 4 waves; 0 = the production
// choice): unit tests and kbench
hipError_t launch_gemm_bf16_tile(int epi, const GemmParams& p, int tile, hipStream_t s);
// one-time > 64 KB dynamic-LDS opt-in of every GEMM instantiation (called from lemas_kernels_init, never on a launch path)
hipError_t gemm_bf16_init();
// `groups` independent EPI_BIAS_F32 GEMMs (parameter blocks in DEVICE memory, 128 x 128 tiles, at most max_tiles tiles each) as one launch
hipError_t launch_gemm_bf16_group(const GemmParams* dev_params, int groups, int max_tiles, hipStream_t s);
// measurement builds only (engine option "block_persist"): out-projection -> ff_norm -> FF1 -> FF2 of one lane as ONE persistent launch with
// grid barriers between the stages; sync = 16 zeroed words, err = the engine's sticky error word; *grid_out = workgroups of the launch
hipError_t launch_gemm_chain_ffhalf(const GemmParams& out, const GemmParams& ff1, const GemmParams& ff2, int ln_scale_off, int ln_shift_off,
                                    unsigned int* sync, unsigned int* err, int prefetch, int* grid_out, hipStream_t s);
// one launch for a lane's QK (+RoPE) and V^T projections (same A, different W / bias / epilogue)
hipError_t launch_gemm_qkv_fused(const GemmParams& pq, const GemmParams& pv, hipStream_t s);
// EPI_GATE_RES with the LayerNorm-modulate tail (GemmParams::ln_out): number of workgroups the launch would have if the production
// tile for this shape supports the tail, else 0.  The caller fuses only when all those workgroups (times the lanes running the
// same kind of launch concurrently) are co-resident; *panels = arrival counters the launch needs, *per_cu = how many of its
// workgroups share a CU.
int gemm_bf16_ln_fusable(const GemmParams& p, int* panels, int* per_cu);

// true when the 128-row block starting at row r0 of the [sample][pitch] row space lies entirely in a sample's padding
__device__ __forceinline__ bool row_block_dead(const int* __restrict__ live_len, int r0, int pitch, int batch) {
  const int b2 = r0 / pitch;
  return r0 - b2 * pitch >= live_len[b2 % batch];
}

struct AttnParams {
  const bf16_t* q;   // [B2, H, pitch, 64]
  const bf16_t* k;   // [B2, H, pitch, 64]
  const bf16_t* vt;  // [B2, H, 64, npad]
  // variants with bit ATTN_F8QK: q and k as MXFP8 (q, k above unused) -- S^T = K . Q^T on v_mfma_scale_f32_32x32x64_f8f6f4
  const uint8_t* q8;     // [B2, H, pitch, 64] e4m3 (q prescaled by softmax_scale * log2(e) before the quantisation: bit 16 is required)
  const uint8_t* k8;
  const uint8_t* q8_mx;  // [B2, H, pitch, 2] E8M0: one scale per 32-wide half of a head
  const uint8_t* k8_mx;
  bf16_t* out;       // [B2*pitch, H*64]
  const int* kv_len; // [B] or nullptr
  const int* live_len;  // [B] or nullptr: 128-query blocks that start at or past live_len of their sample are not computed (GemmParams::live_len)
  int b2, batch, heads, n, npad;
  int pitch;         // rows per sample of q / k / out (>= n)
  float scale;
  uint8_t* out8;     // fp8 path: when set, the output is written as MXFP8 here ([B2*pitch, H*64] e4m3) instead of `out`
  uint8_t* out_mx;   //           [B2*pitch, H*2] E8M0
  hipEvent_t ev_start, ev_stop;   // profiling: kernel begin / end stamps (see GemmParams)
  int variant;       // schedule variant (attention.hip, bit mask); set per launch by the engine / kbench
#ifdef LEMAS_PHASE_TIMESTAMPS
  unsigned long long* dbg;   // measurement builds: per-workgroup phase timestamps [grid][4]
#endif
};
hipError_t launch_attention(const AttnParams& p, hipStream_t s);
// AttnParams::variant with this bit (and bit 16: q prescaled): the 64-queries-per-wave kernel of attention_q64.hip; bits 0-1 pick its
// schedule (1 = blocks skewed, 2 = static priority for the younger half-workgroup)
constexpr int ATTN_Q64 = 4096;
// AttnParams::variant with this bit (and bit 16): QK^T on the fp8 matrix path from MXFP8 q / k (AttnParams::q8 ...), P . V unchanged (attention.hip)
constexpr int ATTN_F8QK = 8192;
// q, k bf16 [rows, 64] -> MXFP8 (the arithmetic of the QK GEMM epilogue's q8 / k8 output; test library and kbench use it to feed the kernel alone)
hipError_t launch_qk_mx8(const bf16_t* q, const bf16_t* k, uint8_t* q8, uint8_t* k8, uint8_t* q8_mx, uint8_t* k8_mx, size_t rows, hipStream_t s);
hipError_t launch_attention_q64(const AttnParams& p, hipStream_t s);
hipError_t attention_q64_init();
// the variants a product engine may select (lemas_dit_set_option "attn_variant"); the measurement-only instantiations of attention.hip
// (ablations, the no-fallback forms) exist in -DLEMAS_PHASE_TIMESTAMPS builds only
bool attention_variant_ok(int variant);

// out_bf16[m][c] = LN(x[m][:])[c] * (1 + scale[c]) + shift[c]; scale/shift read from the AdaLN table row of the current step
// live_len / pitch / batch (optional): rows of 128-row blocks that lie in a sample's padding are skipped (GemmParams::live_len)
hipError_t launch_ln_mod(const float* x, bf16_t* out, int M, int D, const float* tab, int tab_stride,
                         int scale_off, int shift_off, const int* step_idx, hipStream_t s,
                         const int* live_len = nullptr, int pitch = 0, int batch = 0);

// ---- ln fold (see GemmParams): chain entry, and the c1 / c2 table rows of a t-grid ------------------------------------------------
// xs[m][c] = bf16(x[m][c] (1 + scale[c])), part[m][D / 32][2] = (sum, sum of squares) of x over each 32-column slot
hipError_t launch_ln_prep(const float* x, bf16_t* xs, float* part, int M, int D, const float* tab, int tab_stride, int scale_off,
                          const int* step_idx, hipStream_t s);
struct LnFoldSite {          // one LayerNorm + the GEMM behind it; offsets are positions inside an AdaLN table row
  const float* bias;         // [N] bias of that GEMM
  const float* tmp;          // [4 S][N] fp32: the split-term GEMM's output for this site
  int N, scale_off, shift_off, c1_off, c2_off;
};
// A[site][4 S][d] bf16 = hi / lo terms of (1 + scale) and of shift, per step (the A operands of the sites' table GEMMs)
hipError_t launch_ln_fold_split(const float* tab, int tab_stride, int S, int d, const LnFoldSite* sites, int nsites, bf16_t* A, hipStream_t s);
// tab[step][c1_off + n] = tmp rows 0+1, tab[step][c2_off + n] = tmp rows 2+3 + bias
hipError_t launch_ln_fold_combine(const LnFoldSite* sites, int nsites, int max_n, int S, float* tab, int tab_stride, hipStream_t s);

// same LayerNorm-modulate, written as MXFP8 (out8 [M][D] e4m3 + mx [M][D/32] E8M0) for the fp8 GEMMs
hipError_t launch_ln_mod_f8(const float* x, uint8_t* out8, uint8_t* mx, int M, int D, const float* tab, int tab_stride,
                            int scale_off, int shift_off, const int* step_idx, hipStream_t s);
// fp32 rows -> MXFP8 (K % 32 == 0)
hipError_t launch_mx_quant_rows(const float* x, int M, int K, uint8_t* out8, uint8_t* mx, hipStream_t s);
// weights [N][K] fp32 -> e4m3 with one fp32 scale per row (scale = amax / 448; 0-rows get scale 1)
hipError_t launch_w_quant_f8(const float* w, int N, int K, uint8_t* out8, float* scale, hipStream_t s);
// and back: bf16(e4m3 * scale[row]) -- the weights-only-fp8 accuracy point runs the bf16 kernels on these
hipError_t launch_f8_to_bf16(const uint8_t* w8, const float* scale, int N, int K, bf16_t* out, hipStream_t s);

// ---- fp8 path on outlier checkpoints: the flagged output channels of a residual-writing projection from bf16 operands (outlier_rows.hip) ----
struct OutlierRowsParams {
  const bf16_t* A;        // [M][K] bf16 activations (attention output / FF1 output)
  const bf16_t* W;        // [32][K] bf16: row j = the weight row of flagged channel chan[j]; rows past nf are zero
  const float* bias;      // [32]: bias of channel chan[j]
  const int* chan;        // [32] device: flagged channel indices (ascending), -1 past nf
  int nf;                 // flagged channels (1 .. 32)
  float* x;               // residual stream [M][ldx] fp32, updated in place
  int ldx, M, K;
  const float* tab; int tab_stride, gate_off; const int* step_idx;   // gate vector of the current ODE step (GemmParams)
  const int* kv_len;      // [batch] or nullptr: rows at or past their sample's length contribute 0 (EPI_GATE_RES)
  int seq_pitch, seq_valid, batch;
  uint8_t* a8;            // optional: the MXFP8 image of A ([M][K] e4m3 + amx [M][K/32] E8M0) for the fp8 GEMM of the same site, written while
  uint8_t* amx;           // the rows stream through (every row m < M, live or not)
};
hipError_t launch_outlier_rows(const OutlierRowsParams& p, hipStream_t s);

struct ConvPosParams {
  const float* in_f32;    // conv1 input  [B2*N, C] fp32 (nullptr when in_bf16 is used)
  const bf16_t* in_bf16;  // conv2 input  [B2*N, C] bf16
  const bf16_t* w;        // [G][taps][C/G (co)][C/G (ci)] bf16
  const float* bias;      // [C]
  bf16_t* out_bf16;       // conv1 output: mish(conv)                bf16
  float* out_f32;         // conv2 output: mish(conv) + residual     fp32
  const float* residual;  // [B2*N, C] fp32
  int b2, n, channels, groups, taps;
  int pitch;              // rows per sample of every [B2*pitch, C] operand (>= n)
};
hipError_t launch_convpos(const ConvPosParams& p, hipStream_t s);

// y += dt * clamp(pred + (pred - null) * cfg_t, -20, 20); optionally records y into traj[step+1]; step counter++ by one thread
hipError_t launch_cfg_euler(float* y, const float* pred /*[2B*N,100] cond rows then uncond rows*/, int rows, int cols,
                            const float* dt_tab, const float* cfg_tab, int* step_idx, float* traj, int use_cfg, hipStream_t s);

// ---- fp32 (exact) GEMM on f32 MFMA for the once-per-utterance and vocoder work --------------------
enum F32Epi : int {
  F32_BIAS = 0,        // out = acc + bias
  F32_BIAS_GELU = 1,   // out = gelu_erf(acc + bias)
  F32_BIAS_SILU = 2,   // out = silu(acc + bias)
  F32_BIAS_RES_SCALE = 3,  // out = res + colscale[n] * (acc + bias)      (colscale may be null => 1)
  F32_BIAS_ADD2 = 4,   // out[m] = acc + bias? + add[m] and out[m + M] = acc + add[m + M]  (input-proj x-part broadcast to cond/uncond)
  F32_BIAS_RELU = 5,   // out = max(acc + bias, 0)         (prosody encoder TDNN / SE)
  F32_BIAS_SIGMOID = 6,  // out = 1 / (1 + exp(-(acc + bias)))  exact expf (SE gate)
  F32_ROWAFF_RELU = 7,     // out = max(rowscale[ch] * (acc + bias) + rowshift[ch], 0), ch = (m / rows_per_ch) % nch: a Linear over the last axis
                           // of [b, c, t, f] followed by an inference BatchNorm2d over c (MDX-Net TDF, uvr5/lib_v5/modules.py:61-70)
  F32_ROWAFF_RELU_RES = 8, // the same + res[m, n]  (x + tdf(x), modules.py:74)
};
struct GemmF32Params {
  const float* A; int lda;   // [M, K]
  const float* W; int ldw;   // [N, K]
  const float* bias;         // [N] or null
  float* out; int ldc;
  int M, N, K;
  const float* res; int ldres;
  const float* colscale;
  const float* add;          // for F32_BIAS_ADD2: [2M, N]
  const uint8_t* rowmask;    // optional: rows with rowmask[m]!=0 are written as 0
  // batched form (nbatch > 0): the same A against nbatch weight / bias tensors (device arrays of pointers), output z at
  // out + z * out_bstride -- the 22 AdaLN table GEMMs of a prepare() as one launch
  int nbatch;
  const float* const* Wv;
  const float* const* biasv;
  size_t out_bstride;
  // F32_ROWAFF_*: per-row-group affine (null rowscale => 1, 0)
  const float* rowscale;
  const float* rowshift;
  int rows_per_ch, nch;
};
hipError_t launch_gemm_f32(int epi, const GemmF32Params& p, hipStream_t s);

// ---- UVR5 MDX-Net (mdx_kernels.hip): planar [b][c][t][f] fp32, implicit-GEMM convolutions on the f32 MFMA --------------------------
enum MdxConvKind : int { MDX_CONV3 = 0 /* 3x3 s1 p1 */, MDX_DOWN2 = 1 /* 2x2 s2 */, MDX_UP2 = 2 /* transposed 2x2 s2 (+ skip product) */,
                          MDX_CONV3_BX = 3 /* 3x3 s1 p1 on split-bf16 operands: weights [ntiles][ceil(Cin / 16)][10 taps][2][48][8] packed (hi << 16 | lo) */ };
struct MdxConvParams {
  const float* x;        // [B][Cin][Ti][Fi]
  const float* w;        // re-laid weights [ntiles][nchunks][8][mdx_conv_ciw(kind)] (norm folded in)
  const float* bias;     // [Cout] (norm folded in)
  float* out;            // CONV3 / DOWN2: [B][Cout][Tg][Fg]; UP2: [B][Cout][2 Tg][2 Fg]
  const float* skip;     // UP2: multiplied into the result (same shape as out) or null
  int B, Cin, Cout;
  int Ti, Fi;            // input image
  int Tg, Fg;            // GEMM position grid: the output image (CONV3, DOWN2) or the input image (UP2)
  int nchunks, ntiles;   // ceil(Cin / 8); ceil(columns / 48), columns = Cout (UP2: 4 Cout)
  int relu;
};
hipError_t launch_mdx_conv(int kind, const MdxConvParams& p, hipStream_t s);
int mdx_conv_ciw(int kind);
void mdx_conv_set_bx_products(int n);   // bf16 MFMAs per product of the split-bf16 kernel: 3 or 4
void mdx_conv_set_ck(int ck);      // input channels per K chunk of the exact 3x3 kernel: 4 (default) or 8
hipError_t launch_mdx_first(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout, int F, int T, int relu,
                            hipStream_t s);
hipError_t launch_mdx_final(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout, int F, int T, hipStream_t s);
hipError_t launch_mdx_groupnorm(const float* x, int ld, int cols, int B, int C, int T, const float* gamma, const float* beta, float eps,
                                const float* other, int mode, float* out, double* part, float* stats, hipStream_t s);
