// Row / column kernels of the prosody encoder (ECAPA-TDNN) and its kaldi-fbank front end (gfx950).  SURVEY.md 8f-2.
// Reference: lemas_tts/model/backbones/prosody_encoder.py (TDNNBlock :136-161, Res2NetBlock :164-202, SEBlock :205-230,
// AttentiveStatisticsPooling :233-280, extract_fbank_16k :334-361).  Runs once per utterance on ~1000 frames: these are
// plain HBM-bound kernels; the contractions go through the exact-fp32 MFMA GEMM (gemm_f32.hip).
// Layout: activations are time-major [T][ld] fp32 (channels contiguous), so every 1x1 conv is a GEMM as is and a
// dilated k-tap conv is im2col + GEMM with the reference's [out][in][k] weight viewed as [out][in*k].
#include "common.h"
#include "kernels.h"

namespace {

inline unsigned grid_for(size_t n, int block = 256) {
  size_t g = (n + block - 1) / block;
  return (unsigned)(g < 65535u * 16 ? (g ? g : 1) : 65535u * 16);
}

// col[t][c * k + j] = (x + add)[t + (j - (k-1)/2) * dil][c], zero outside [0, T)   (Conv1d padding = dil (k-1) / 2)
__global__ void im2col_dil_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ add, int ldadd, int T, int C, int k,
                                  int dil, float* __restrict__ col) {
  const size_t total = (size_t)T * C * k;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % k);
    const int c = (int)((i / k) % C);
    const int t = (int)(i / ((size_t)k * C));
    const int ts = t + (j - (k - 1) / 2) * dil;
    float v = 0.f;
    if (ts >= 0 && ts < T) {
      v = x[(size_t)ts * ldx + c];
      if (add) v += add[(size_t)ts * ldadd + c];
    }
    col[i] = v;
  }
}

// LayerNorm over the C channels of each row (affine, eps inside the sqrt), optional tanh; one wave per row.  C % 4 == 0, C <= 4096 and
// 16-B aligned rows (launch_ln_rows checks): the row is read ONCE, as float4, and stays in registers for the two-pass statistics and the output (the first
// form read it three times with scalar loads: 30 us for the 998 x 1536 MFA rows, profiles/r06/r06fe_kernel_stats_frontend.txt)
template <int NV>
__global__ __launch_bounds__(256) void ln_rows_kernel(const float* __restrict__ x, int ldx, int T, int C, const float* __restrict__ w,
                                                      const float* __restrict__ b, float eps, int act_tanh, float* __restrict__ out, int ldo) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= T) return;
  const int lane = threadIdx.x & 63, nv = C >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * ldx);
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = lane + 64 * i;
    v[i] = c4 < nv ? xr[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (lane + 64 * i < nv) {
      const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
      q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
  float4* o = reinterpret_cast<float4*>(out + (size_t)row * ldo);
  const float4* w4 = reinterpret_cast<const float4*>(w);
  const float4* b4 = reinterpret_cast<const float4*>(b);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < nv) {
      const float4 ww = w4[c4], bb = b4[c4];
      float4 r = make_float4((v[i].x - mean) * rstd * ww.x + bb.x, (v[i].y - mean) * rstd * ww.y + bb.y,
                             (v[i].z - mean) * rstd * ww.z + bb.z, (v[i].w - mean) * rstd * ww.w + bb.w);
      if (act_tanh) r = make_float4(tanhf(r.x), tanhf(r.y), tanhf(r.z), tanhf(r.w));
      o[c4] = r;
    }
  }
}

// One Res2Net chunk of an SE-Res2Net block at the published widths (64 channels, 3 taps; prosody_encoder.py:188-202 + TDNNBlock :136-161) as ONE
// launch: y = LayerNorm(relu(conv_3,dil(x + add) + bias)).  The three launches it replaces (im2col 3.2 us, a 998 x 64 x 192 GEMM 5.6 us, the row
// LayerNorm 2.7 us: profiles/r06/r06fh_kernel_stats_frontend.txt) run 21 times per prompt, each waiting for the one before.
// Block = 16 rows x 64 output channels on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, operand roles as gemm_f32.hip): wave w owns output channels
// 16 w .. 16 w + 15 with their 192 weights in 48 registers per lane; the (x + add) rows with their halo sit in LDS ([row][68]: the 16 rows of a
// fragment read start in 16 different 4-bank groups) and tap j of row m is simply row m + j dil of that tile -- no im2col; the relu'd 16 x 64
// tile goes through LDS once so that a wave can normalise whole rows.  K order: (tap, 16-channel group, 4 lk + e), the same on both operands.
// A first form on the vector ALU (lane = output channel, x rows read as wave-wide LDS broadcasts) took 8.5 us: 192 broadcast ds_read_b128 per
// lane still move 1 KB each into registers (profiles/r06/r06fj_kernel_stats_frontend_valu_form.txt).
constexpr int R2_C = 64, R2_K = 3, R2_ROWS = 16, R2_PITCH = R2_C + 4;
// the reference's [out][in * k] weight as [tap][16-channel group][out][lk][4]: the lanes of a wave read their operand float4s as coalesced 1-KB requests
__global__ void res2net_weight_image_kernel(const float* __restrict__ W, float* __restrict__ img) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R2_C * R2_C * R2_K) return;
  const int e = i & 3, lk = (i >> 2) & 3, n = (i >> 4) % R2_C, jg = i / (16 * R2_C), g = jg & 3, j = jg >> 2;
  img[i] = W[(size_t)n * (R2_C * R2_K) + (16 * g + 4 * lk + e) * R2_K + j];
}
__global__ __launch_bounds__(256) void res2net_step_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ add, int ldadd, int T, int dil,
                                                           const float* __restrict__ Wimg, const float* __restrict__ bias, const float* __restrict__ lnw,
                                                           const float* __restrict__ lnb, float eps, float* __restrict__ out, int ldo) {
  extern __shared__ __attribute__((aligned(16))) float xs[];     // [R2_ROWS + 2 dil][R2_PITCH], then ys [R2_ROWS][R2_C + 1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lk = lane >> 4;
  const int t0 = blockIdx.x * R2_ROWS, nrows = R2_ROWS + 2 * dil;
  float* ys = xs + nrows * R2_PITCH;
  float4 w[R2_K * 4];
  const float4* w4 = reinterpret_cast<const float4*>(Wimg) + (wave * 16 + l15) * 4 + lk;
#pragma unroll
  for (int q = 0; q < R2_K * 4; ++q) w[q] = w4[q * (R2_C * 4)];
  for (int i = tid; i < nrows * (R2_C / 4); i += 256) {
    const int r = i / (R2_C / 4), c4 = i % (R2_C / 4), t = t0 - dil + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t >= 0 && t < T) {
      v = *reinterpret_cast<const float4*>(x + (size_t)t * ldx + 4 * c4);
      if (add) {
        const float4 a = *reinterpret_cast<const float4*>(add + (size_t)t * ldadd + 4 * c4);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
      }
    }
    *reinterpret_cast<float4*>(xs + r * R2_PITCH + 4 * c4) = v;
  }
  __syncthreads();
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < R2_K; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 a = *reinterpret_cast<const float4*>(xs + (l15 + j * dil) * R2_PITCH + 16 * g + 4 * lk);
      const float4 b = w[j * 4 + g];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b.x, a.x, acc, 0, 0, 0);      // C^T: lane holds out[row l15][channel 16 wave + 4 lk + i]
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b.y, a.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b.z, a.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b.w, a.w, acc, 0, 0, 0);
    }
  {
    const int n = wave * 16 + 4 * lk;
    const float4 bv = *reinterpret_cast<const float4*>(bias + n);
    float* yr = ys + l15 * (R2_C + 1) + n;
    yr[0] = fmaxf(acc[0] + bv.x, 0.f); yr[1] = fmaxf(acc[1] + bv.y, 0.f); yr[2] = fmaxf(acc[2] + bv.z, 0.f); yr[3] = fmaxf(acc[3] + bv.w, 0.f);
  }
  __syncthreads();
  const float gw = lnw[lane], gb = lnb[lane];
#pragma unroll
  for (int q = 0; q < R2_ROWS / 4; ++q) {
    const int r = wave * (R2_ROWS / 4) + q, t = t0 + r;
    if (t >= T) break;                            // (uniform per wave)
    const float v = ys[r * (R2_C + 1) + lane];
    const float mean = wave_sum(v) / (float)R2_C;
    const float d = v - mean;
    const float rstd = 1.0f / sqrtf(wave_sum(d * d) / (float)R2_C + eps);
    out[(size_t)t * ldo + lane] = d * rstd * gw + gb;
  }
}

__global__ void copy_cols_kernel(const float* __restrict__ x, int ldx, int T, int C, float* __restrict__ out, int ldo) {
  const size_t total = (size_t)T * C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const size_t t = i / C;
    out[t * ldo + c] = x[t * ldx + c];
  }
}

// per-channel mean and std over time with uniform weights 1/T: std = sqrt(clamp(sum w (x - mean)^2, eps)).
// block = COL_CX channels x COL_RY row lanes (1024 threads): a column of ~1000 frames is ~16 rows per thread, requested eight at a time.  (The
// first form -- 32 channels x 8 row lanes -- walked 125 dependent loads per thread: 50 us for 2 MB, profiles/r06/r06fe_kernel_stats_frontend.txt.)
constexpr int COL_CX = 16, COL_RY = 64;

template <bool IS_MAX>
__device__ __forceinline__ float col_reduce(float (&red)[COL_RY][COL_CX + 1], float v, int ry, int cx) {
  red[ry][cx] = v;
  __syncthreads();
  for (int h = COL_RY / 2; h >= 1; h >>= 1) {
    if (ry < h) red[ry][cx] = IS_MAX ? fmaxf(red[ry][cx], red[ry + h][cx]) : red[ry][cx] + red[ry + h][cx];
    __syncthreads();
  }
  const float r = red[0][cx];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(COL_CX * COL_RY) void col_stats_kernel(const float* __restrict__ x, int ldx, int T, int C, float eps,
                                                                    float* __restrict__ mean, float* __restrict__ stdv) {
  __shared__ float red[COL_RY][COL_CX + 1];
  const int cx = threadIdx.x % COL_CX, ry = threadIdx.x / COL_CX, c = blockIdx.x * COL_CX + cx;
  const bool ok = c < C;
  const float* xc = x + (ok ? c : 0);
  float s = 0.f;
#pragma unroll 8
  for (int t = ry; t < T; t += COL_RY) s += xc[(size_t)t * ldx];
  const float m = col_reduce<false>(red, s, ry, cx) / (float)T;
  float q = 0.f;
#pragma unroll 8
  for (int t = ry; t < T; t += COL_RY) { const float d = xc[(size_t)t * ldx] - m; q += d * d; }
  const float v = col_reduce<false>(red, q, ry, cx);
  if (ry == 0 && ok) {
    mean[c] = m;
    if (stdv) stdv[c] = sqrtf(fmaxf(v / (float)T, eps));
  }
}

// out[t][c] = scale[c] * x[t][c] + res[t][c]     (SE gate + block residual)
__global__ void scale_cols_add_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ scale, const float* __restrict__ res,
                                      int ldr, int T, int C, float* __restrict__ out, int ldo) {
  const size_t total = (size_t)T * C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const size_t t = i / C;
    out[t * ldo + c] = scale[c] * x[t * ldx + c] + res[t * ldr + c];
  }
}

// attentive statistics: a = softmax over time of att[:, c]; mean = sum a x; std = sqrt(clamp(sum a (x - mean)^2, eps))
__global__ __launch_bounds__(COL_CX * COL_RY) void softmax_pool_kernel(const float* __restrict__ att, int lda, const float* __restrict__ x, int ldx, int T,
                                                                       int C, float eps, float* __restrict__ mean, float* __restrict__ stdv) {
  __shared__ float red[COL_RY][COL_CX + 1];
  const int cx = threadIdx.x % COL_CX, ry = threadIdx.x / COL_CX, c = blockIdx.x * COL_CX + cx;
  const bool ok = c < C;
  const float* ac = att + (ok ? c : 0);
  const float* xc = x + (ok ? c : 0);
  float mx = -INFINITY;
#pragma unroll 8
  for (int t = ry; t < T; t += COL_RY) mx = fmaxf(mx, ac[(size_t)t * lda]);
  mx = col_reduce<true>(red, mx, ry, cx);
  float se = 0.f, sx = 0.f;
#pragma unroll 8
  for (int t = ry; t < T; t += COL_RY) { const float e = expf(ac[(size_t)t * lda] - mx); se += e; sx += e * xc[(size_t)t * ldx]; }
  se = col_reduce<false>(red, se, ry, cx);
  sx = col_reduce<false>(red, sx, ry, cx);
  const float m = sx / se;
  float q = 0.f;
#pragma unroll 8
  for (int t = ry; t < T; t += COL_RY) { const float e = expf(ac[(size_t)t * lda] - mx); const float d = xc[(size_t)t * ldx] - m; q += e * d * d; }
  q = col_reduce<false>(red, q, ry, cx);
  if (ry == 0 && ok) { mean[c] = m; stdv[c] = sqrtf(fmaxf(q / se, eps)); }
}

// one-row Linear (the SE gates, the global-context bias, the final fc): out[n] = act(bias[n] + sum_k a[k] W[n][k]); one wave per output,
// float4 reads of the weight row.  As a 32 x 64 tile of the matrix GEMM a single row walked K = 3072 in 96 dependent K-tiles on 2-8
// workgroups: 47-48 us per call (profiles/r06/r06fg_sequence_frontend.txt).  EPI: F32_BIAS / F32_BIAS_RELU / F32_BIAS_SIGMOID
template <int EPI>
__global__ __launch_bounds__(256) void gemv_f32_kernel(const float* __restrict__ a, const float* __restrict__ W, int ldw, const float* __restrict__ bias,
                                                       int N, int K, float* __restrict__ out) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int lane = threadIdx.x & 63, k4 = K >> 2;
  const float4* w4 = reinterpret_cast<const float4*>(W + (size_t)n * ldw);
  const float4* a4 = reinterpret_cast<const float4*>(a);
  float acc = 0.f;
#pragma unroll 4
  for (int i = lane; i < k4; i += 64) {
    const float4 w = w4[i], x = a4[i];
    acc = fmaf(w.x, x.x, fmaf(w.y, x.y, fmaf(w.z, x.z, fmaf(w.w, x.w, acc))));
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    float v = acc + (bias ? bias[n] : 0.f);
    if (EPI == F32_BIAS_RELU) v = fmaxf(v, 0.f);
    if (EPI == F32_BIAS_SIGMOID) v = 1.0f / (1.0f + expf(-v));
    out[n] = v;
  }
}

// F.normalize: x / max(||x||, eps); one wave
__global__ void l2_normalize_kernel(const float* __restrict__ x, int n, float eps, float* __restrict__ out) {
  const int lane = threadIdx.x;
  float q = 0.f;
  for (int i = lane; i < n; i += 64) q += x[i] * x[i];
  const float nrm = fmaxf(sqrtf(wave_sum(q)), eps);
  for (int i = lane; i < n; i += 64) out[i] = x[i] / nrm;
}

// kaldi framing (snip_edges): remove DC, pre-emphasis with the first sample replicated, povey window, zero-pad; one block per frame
__global__ __launch_bounds__(256) void kaldi_frames_kernel(const float* __restrict__ wav, int n, int frames, int win, int shift, int padded,
                                                           float preemph, float* __restrict__ out) {
  __shared__ float buf[1024];
  __shared__ float part[4];
  const int f = blockIdx.x, tid = threadIdx.x;
  const float* src = wav + (size_t)f * shift;
  float s = 0.f;
  for (int i = tid; i < win; i += 256) { const float v = src[i]; buf[i] = v; s += v; }
  s = wave_sum(s);
  if ((tid & 63) == 0) part[tid >> 6] = s;
  __syncthreads();
  const float mean = (part[0] + part[1] + part[2] + part[3]) / (float)win;
  for (int i = tid; i < padded; i += 256) {
    float v = 0.f;
    if (i < win) {
      const float cur = buf[i] - mean, prev = buf[i > 0 ? i - 1 : 0] - mean;
      const float w = powf(0.5f - 0.5f * cospif(2.0f * (float)i / (float)(win - 1)), 0.85f);
      v = (cur - preemph * prev) * w;
    }
    out[(size_t)f * padded + i] = v;
  }
}

// power spectrum from [re | im] rows, padded to ldp columns with zeros
__global__ void power_kernel(const float* __restrict__ spec, int rows, int nb, int lds_, int ldp, float* __restrict__ pw) {
  const size_t total = (size_t)rows * ldp;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % ldp);
    const size_t r = i / ldp;
    float v = 0.f;
    if (k < nb) { const float re = spec[r * lds_ + k], im = spec[r * lds_ + nb + k]; v = re * re + im * im; }
    pw[i] = v;
  }
}

}  // namespace

#define LAUNCH1(kern, n, ...)                                                             \
  hipLaunchKernelGGL(kern, dim3(grid_for(n)), dim3(256), 0, s, __VA_ARGS__);             \
  return hipGetLastError();

hipError_t launch_im2col_dil(const float* x, int ldx, const float* add, int ldadd, int T, int C, int k, int dil, float* col, hipStream_t s) {
  if ((k & 1) == 0) return hipErrorInvalidValue;
  LAUNCH1(im2col_dil_kernel, (size_t)T * C * k, x, ldx, add, ldadd, T, C, k, dil, col)
}
hipError_t launch_ln_rows(const float* x, int ldx, int T, int C, const float* w, const float* b, float eps, int act_tanh, float* out, int ldo,
                          hipStream_t s) {
  const dim3 grid((T + 3) / 4), block(256);
  // float4 rows: every row start and the width on 16 B (the engine's buffers are; a width that is not a multiple of 4 never gets here --
  // the GEMM in front of every LayerNorm refuses K % 4 != 0)
  if ((C & 3) || (ldx & 3) || (ldo & 3) || C > 4096 || ((uintptr_t)x & 15) || ((uintptr_t)out & 15) || ((uintptr_t)w & 15) || ((uintptr_t)b & 15))
    return hipErrorInvalidValue;
  if (C <= 256) hipLaunchKernelGGL(ln_rows_kernel<1>, grid, block, 0, s, x, ldx, T, C, w, b, eps, act_tanh, out, ldo);
  else if (C <= 512) hipLaunchKernelGGL(ln_rows_kernel<2>, grid, block, 0, s, x, ldx, T, C, w, b, eps, act_tanh, out, ldo);
  else if (C <= 1024) hipLaunchKernelGGL(ln_rows_kernel<4>, grid, block, 0, s, x, ldx, T, C, w, b, eps, act_tanh, out, ldo);
  else if (C <= 2048) hipLaunchKernelGGL(ln_rows_kernel<8>, grid, block, 0, s, x, ldx, T, C, w, b, eps, act_tanh, out, ldo);
  else hipLaunchKernelGGL(ln_rows_kernel<16>, grid, block, 0, s, x, ldx, T, C, w, b, eps, act_tanh, out, ldo);
  return hipGetLastError();
}
hipError_t launch_copy_cols(const float* x, int ldx, int T, int C, float* out, int ldo, hipStream_t s) {
  LAUNCH1(copy_cols_kernel, (size_t)T * C, x, ldx, T, C, out, ldo)
}
hipError_t launch_col_stats(const float* x, int ldx, int T, int C, float eps, float* mean, float* stdv, hipStream_t s) {
  hipLaunchKernelGGL(col_stats_kernel, dim3((C + COL_CX - 1) / COL_CX), dim3(COL_CX * COL_RY), 0, s, x, ldx, T, C, eps, mean, stdv);
  return hipGetLastError();
}
hipError_t launch_scale_cols_add(const float* x, int ldx, const float* scale, const float* res, int ldr, int T, int C, float* out, int ldo,
                                 hipStream_t s) {
  LAUNCH1(scale_cols_add_kernel, (size_t)T * C, x, ldx, scale, res, ldr, T, C, out, ldo)
}
hipError_t launch_softmax_pool(const float* att, int lda, const float* x, int ldx, int T, int C, float eps, float* mean, float* stdv,
                               hipStream_t s) {
  hipLaunchKernelGGL(softmax_pool_kernel, dim3((C + COL_CX - 1) / COL_CX), dim3(COL_CX * COL_RY), 0, s, att, lda, x, ldx, T, C, eps, mean, stdv);
  return hipGetLastError();
}
size_t res2net_weight_image_floats() { return (size_t)R2_C * R2_C * R2_K; }
hipError_t launch_res2net_weight_image(const float* W, float* img, hipStream_t s) {
  hipLaunchKernelGGL(res2net_weight_image_kernel, dim3((R2_C * R2_C * R2_K + 255) / 256), dim3(256), 0, s, W, img);
  return hipGetLastError();
}
bool res2net_step_fits(int cin, int cout, int k, int dil) { return cin == R2_C && cout == R2_C && k == R2_K && dil >= 1 && dil <= 16; }
hipError_t launch_res2net_step(const float* x, int ldx, const float* add, int ldadd, int T, int dil, const float* Wimg, const float* bias,
                               const float* lnw, const float* lnb, float eps, float* out, int ldo, hipStream_t s) {
  const float* W = Wimg;
  if (T <= 0 || dil < 1 || dil > 16 || (ldx & 3) || (add && (ldadd & 3)) || ((uintptr_t)x & 15) || ((uintptr_t)add & 15) || ((uintptr_t)W & 15))
    return hipErrorInvalidValue;
  if (((uintptr_t)bias & 15)) return hipErrorInvalidValue;
  const size_t lds = ((size_t)(R2_ROWS + 2 * dil) * R2_PITCH + (size_t)R2_ROWS * (R2_C + 1)) * 4;
  hipLaunchKernelGGL(res2net_step_kernel, dim3((T + R2_ROWS - 1) / R2_ROWS), dim3(256), lds, s, x, ldx, add, ldadd, T, dil, W, bias, lnw, lnb, eps, out, ldo);
  return hipGetLastError();
}
hipError_t launch_gemv_f32(int epi, const GemmF32Params& p, hipStream_t s) {
  if (p.M != 1 || p.N <= 0 || p.K <= 0 || (p.K & 3) || (p.ldw & 3) || ((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15) || p.nbatch > 0 || p.rowmask)
    return hipErrorInvalidValue;
  const dim3 grid((p.N + 3) / 4), block(256);
  switch (epi) {
    case F32_BIAS: hipLaunchKernelGGL(gemv_f32_kernel<F32_BIAS>, grid, block, 0, s, p.A, p.W, p.ldw, p.bias, p.N, p.K, p.out); break;
    case F32_BIAS_RELU: hipLaunchKernelGGL(gemv_f32_kernel<F32_BIAS_RELU>, grid, block, 0, s, p.A, p.W, p.ldw, p.bias, p.N, p.K, p.out); break;
    case F32_BIAS_SIGMOID: hipLaunchKernelGGL(gemv_f32_kernel<F32_BIAS_SIGMOID>, grid, block, 0, s, p.A, p.W, p.ldw, p.bias, p.N, p.K, p.out); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t launch_l2_normalize(const float* x, int n, float eps, float* out, hipStream_t s) {
  hipLaunchKernelGGL(l2_normalize_kernel, dim3(1), dim3(64), 0, s, x, n, eps, out);
  return hipGetLastError();
}
hipError_t launch_kaldi_frames(const float* wav, int n, int frames, int win, int shift, int padded, float preemph, float* out, hipStream_t s) {
  if (win > 1024 || frames <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(kaldi_frames_kernel, dim3(frames), dim3(256), 0, s, wav, n, frames, win, shift, padded, preemph, out);
  return hipGetLastError();
}
hipError_t launch_power(const float* spec, int rows, int nb, int lds_, int ldp, float* pw, hipStream_t s) {
  LAUNCH1(power_kernel, (size_t)rows * ldp, spec, rows, nb, lds_, ldp, pw)
}
