// fp8 path on checkpoints with OUTLIER residual channels: the few output channels of a residual-writing
// projection (attn.to_out, ff.2: modules.py:495, :350) whose weight rows are far above the typical row are computed HERE from bf16 operands
// instead of by the fp8 GEMM (whose e4m3 image of those rows is zeroed, bias included):
//
//     x[m][c_j] += gate[c_j] * ( sum_k A[m][k] * W[c_j][k] + bias[c_j] )        for the nf <= 32 flagged channels c_j
//
// Why: on such weights every error that reaches a flagged channel is handed, x30, to every LayerNorm behind it.  The error of an e4m3 WEIGHT is
// the same perturbation at every frame and every ODE step and integrates coherently; measured on the reference's own outputs
// (profiles/r05/r05_fp8_outlier_decomposition.txt), the flagged rows need bf16 operands on BOTH sides, everything else of the three fp8 sites does not.
// The kernel runs BEFORE the fp8 GEMM of its site (the two touch disjoint columns: the GEMM adds gate * 0 to the flagged ones), and since it
// streams every activation row anyway it can also write the row's MXFP8 image (`a8` / `amx`) that the fp8 GEMM consumes: FF1, which runs
// on bf16 operands in this mode, then keeps its fast bf16 kernel instead of a two-output epilogue.
//
// Shape: M rows x (nf <= 32 channels, padded to 32) x K in {1024, 2048}: 0.1-0.4 GFLOP and 6-12 MB -- a latency / bandwidth problem.  One
// workgroup = 32 rows, EIGHT waves that split K; a wave's K slice is cut in two halves and lane (row i, half h) owns a CONTIGUOUS run of K / 16
// values of row i -- whole 128-B lines (K = 1024) per lane, every fetched sector used, all loads of a wave in flight at once.  The MFMA's k index
// is only a summation index: step t multiplies lane-owned values [8 t, 8 t + 8) of the activation row with the SAME k positions of the weight
// row, so the permutation cancels.  A lane's run is whole 32-value MX blocks, so the MXFP8 image needs no cross-lane maximum.  Partial tiles
// are summed through LDS in a fixed order by wave 0, which also does the gated read-modify-write.  Swapped operands as in gemm_bf16.hip: a
// lane owns ONE row m and 16 of the 32 channel slots.
// MEASUREMENT BUILDS ONLY since round 6 (-DLEMAS_MEASUREMENT_BUILD): the decomposition holds the 1e-4 target but is slower than the all-bf16
// fallback it replaces (profiles/r05/r05h_outlier_throughput.txt), so the product library does not carry this kernel or its engine option.
#include "common.h"      // (defines LEMAS_MEASUREMENT_BUILD for -DLEMAS_PHASE_TIMESTAMPS builds too)
#ifdef LEMAS_MEASUREMENT_BUILD
#include "common.h"

namespace {

constexpr int OR_ROWS = 32, OR_WAVES = 8;

template <int KPL>      // K values per lane = K / 16: 64 (K = 1024) or 128 (K = 2048)
__global__ __launch_bounds__(64 * OR_WAVES) void outlier_rows_kernel(const OutlierRowsParams p) {
  constexpr int STEPS = KPL / 8;                               // MFMA k-steps per wave
  __shared__ float part[OR_WAVES - 1][64][16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int m0 = blockIdx.x * OR_ROWS;
  const int m = m0 + l31;
  const int mc = m < p.M ? m : p.M - 1;                        // clamp: rows past M are computed and discarded
  const int k0 = (wave * 2 + hi) * KPL;                        // first K value this lane owns
  const u32x4* arow = reinterpret_cast<const u32x4*>(p.A + (size_t)mc * p.K + k0);
  const u32x4* wrow = reinterpret_cast<const u32x4*>(p.W + (size_t)l31 * p.K + k0);      // side weights: [32][K] bf16, rows past nf are zero
  u32x4 a[STEPS], w[STEPS];
#pragma unroll
  for (int t = 0; t < STEPS; ++t) a[t] = arow[t];
#pragma unroll
  for (int t = 0; t < STEPS; ++t) w[t] = wrow[t];
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int t = 0; t < STEPS; ++t)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[t]), __builtin_bit_cast(bf16x8, a[t]), acc, 0, 0, 0);
  // optional: the MXFP8 image of this lane's run (KPL / 32 blocks; arithmetic of common.h mx_exponent / pack_fp8x4 = oracle/mxfp8.py)
  if (p.a8 && m < p.M) {
#pragma unroll
    for (int blk = 0; blk < KPL / 32; ++blk) {
      float v[32];
      float amax = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned int u = a[blk * 4 + q][e];
          v[q * 8 + 2 * e] = __uint_as_float(u << 16);
          v[q * 8 + 2 * e + 1] = __uint_as_float(u & 0xffff0000u);
          amax = fmaxf(amax, fmaxf(fabsf(v[q * 8 + 2 * e]), fabsf(v[q * 8 + 2 * e + 1])));
        }
      const int ex = mx_exponent(amax);
      const float inv = mx_inv_scale(ex);
      u32x4 o[2];
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q >> 2][q & 3] = pack_fp8x4(v[4 * q] * inv, v[4 * q + 1] * inv, v[4 * q + 2] * inv, v[4 * q + 3] * inv);
      uint8_t* dst = p.a8 + (size_t)m * p.K + k0 + blk * 32;
      store_wt_b128(dst, o[0]);
      store_wt_b128(dst + 16, o[1]);
      p.amx[(size_t)m * (p.K >> 5) + (k0 >> 5) + blk] = (uint8_t)(ex + 127);
    }
  }
  // K-split partial sums: waves 1..7 park theirs, wave 0 adds them in a fixed order (deterministic)
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave - 1][lane][r] = acc[r];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int q = 0; q < OR_WAVES - 1; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += part[q][lane][r];
  // epilogue: lane -> row m = m0 + (lane & 31); register r -> channel slot j = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  if (m >= p.M) return;
  const int b2 = m / p.seq_pitch, pos = m - b2 * p.seq_pitch;
  int limit = p.seq_valid;
  if (p.kv_len) { const int kv = p.kv_len[b2 % p.batch]; limit = kv < limit ? kv : limit; }       // rows past kv_len contribute 0 (EPI_GATE_RES)
  if (pos >= limit) return;
  const float* gate = p.tab + (size_t)p.step_idx[0] * p.tab_stride + p.gate_off;
  float* xrow = p.x + (size_t)m * p.ldx;
  float xv[16];
  int ch[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = (r & 3) + 8 * (r >> 2) + 4 * hi;
    ch[r] = j < p.nf ? p.chan[j] : -1;
    xv[r] = ch[r] >= 0 ? xrow[ch[r]] : 0.f;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = (r & 3) + 8 * (r >> 2) + 4 * hi;
    if (ch[r] >= 0) xrow[ch[r]] = xv[r] + gate[ch[r]] * (acc[r] + p.bias[j]);
  }
}

}  // namespace

hipError_t launch_outlier_rows(const OutlierRowsParams& p, hipStream_t s) {
  if (p.M <= 0 || p.nf <= 0 || p.nf > 32 || !p.A || !p.W || !p.bias || !p.chan || !p.x || !p.tab || !p.step_idx || p.seq_pitch <= 0 || p.batch <= 0 ||
      (p.a8 && !p.amx))
    return hipErrorInvalidValue;
  const dim3 grid((p.M + OR_ROWS - 1) / OR_ROWS), block(64 * OR_WAVES);
  if (p.K == 1024) hipLaunchKernelGGL(outlier_rows_kernel<64>, grid, block, 0, s, p);
  else if (p.K == 2048) hipLaunchKernelGGL(outlier_rows_kernel<128>, grid, block, 0, s, p);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

#endif  // LEMAS_MEASUREMENT_BUILD
