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

// LayerNorm over the C channels of each row (affine, eps inside the sqrt), optional tanh; one wave per row
__global__ __launch_bounds__(256) void ln_rows_kernel(const float* __restrict__ x, int ldx, int T, int C, const float* __restrict__ w,
                                                      const float* __restrict__ b, float eps, int act_tanh, float* __restrict__ out, int ldo) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= T) return;
  const int lane = threadIdx.x & 63;
  const float* xr = x + (size_t)row * ldx;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c];
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
  for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
  float* o = out + (size_t)row * ldo;
  for (int c = lane; c < C; c += 64) {
    float v = (xr[c] - mean) * rstd * w[c] + b[c];
    if (act_tanh) v = tanhf(v);
    o[c] = v;
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
// block = 32 channels x 8 row lanes
__global__ __launch_bounds__(256) void col_stats_kernel(const float* __restrict__ x, int ldx, int T, int C, float eps,
                                                        float* __restrict__ mean, float* __restrict__ stdv) {
  __shared__ float red[8][33];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5, c = blockIdx.x * 32 + cx;
  const bool ok = c < C;
  float s = 0.f;
  if (ok) for (int t = ry; t < T; t += 8) s += x[(size_t)t * ldx + c];
  red[ry][cx] = s;
  __syncthreads();
  float m = 0.f;
  for (int r = 0; r < 8; ++r) m += red[r][cx];
  m /= (float)T;
  __syncthreads();
  float q = 0.f;
  if (ok) for (int t = ry; t < T; t += 8) { const float d = x[(size_t)t * ldx + c] - m; q += d * d; }
  red[ry][cx] = q;
  __syncthreads();
  if (ry == 0 && ok) {
    float v = 0.f;
    for (int r = 0; r < 8; ++r) v += red[r][cx];
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
__global__ __launch_bounds__(256) void softmax_pool_kernel(const float* __restrict__ att, int lda, const float* __restrict__ x, int ldx, int T,
                                                           int C, float eps, float* __restrict__ mean, float* __restrict__ stdv) {
  __shared__ float red[8][33];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5, c = blockIdx.x * 32 + cx;
  const bool ok = c < C;
  auto reduce = [&](float v, bool is_max) {
    red[ry][cx] = v;
    __syncthreads();
    float r = red[0][cx];
    for (int i = 1; i < 8; ++i) r = is_max ? fmaxf(r, red[i][cx]) : r + red[i][cx];
    __syncthreads();
    return r;
  };
  float mx = -INFINITY;
  if (ok) for (int t = ry; t < T; t += 8) mx = fmaxf(mx, att[(size_t)t * lda + c]);
  mx = reduce(mx, true);
  float se = 0.f, sx = 0.f;
  if (ok) for (int t = ry; t < T; t += 8) { const float e = expf(att[(size_t)t * lda + c] - mx); se += e; sx += e * x[(size_t)t * ldx + c]; }
  se = reduce(se, false);
  sx = reduce(sx, false);
  const float m = sx / se;
  float q = 0.f;
  if (ok) for (int t = ry; t < T; t += 8) { const float e = expf(att[(size_t)t * lda + c] - mx); const float d = x[(size_t)t * ldx + c] - m; q += e * d * d; }
  q = reduce(q, false);
  if (ry == 0 && ok) { mean[c] = m; stdv[c] = sqrtf(fmaxf(q / se, eps)); }
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
  hipLaunchKernelGGL(ln_rows_kernel, dim3((T + 3) / 4), dim3(256), 0, s, x, ldx, T, C, w, b, eps, act_tanh, out, ldo);
  return hipGetLastError();
}
hipError_t launch_copy_cols(const float* x, int ldx, int T, int C, float* out, int ldo, hipStream_t s) {
  LAUNCH1(copy_cols_kernel, (size_t)T * C, x, ldx, T, C, out, ldo)
}
hipError_t launch_col_stats(const float* x, int ldx, int T, int C, float eps, float* mean, float* stdv, hipStream_t s) {
  hipLaunchKernelGGL(col_stats_kernel, dim3((C + 31) / 32), dim3(256), 0, s, x, ldx, T, C, eps, mean, stdv);
  return hipGetLastError();
}
hipError_t launch_scale_cols_add(const float* x, int ldx, const float* scale, const float* res, int ldr, int T, int C, float* out, int ldo,
                                 hipStream_t s) {
  LAUNCH1(scale_cols_add_kernel, (size_t)T * C, x, ldx, scale, res, ldr, T, C, out, ldo)
}
hipError_t launch_softmax_pool(const float* att, int lda, const float* x, int ldx, int T, int C, float eps, float* mean, float* stdv,
                               hipStream_t s) {
  hipLaunchKernelGGL(softmax_pool_kernel, dim3((C + 31) / 32), dim3(256), 0, s, att, lda, x, ldx, T, C, eps, mean, stdv);
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
