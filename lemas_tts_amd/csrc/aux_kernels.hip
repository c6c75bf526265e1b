// Once-per-utterance kernels (text embedding, conditioning setup, AdaLN/time tables) and the Vocos vocoder's
// non-GEMM kernels.  All fp32, all HBM-bound, channel-last layouts so consecutive lanes touch consecutive
// channels (coalesced).  Reference lines are cited at each kernel.
#include "common.h"
#include "kernels.h"

namespace {

inline int grid_for(size_t n, int block = 256) {
  size_t g = (n + block - 1) / block;
  return (int)(g < 4096 ? (g ? g : 1) : 4096);
}

// ---- TextEmbedding front (backbones/dit.py:51-70): token+1, truncate/pad to N, pad-mask BEFORE the cfg drop,
// embedding lookup + sinusoid position table, masked_fill(0).  Rows [0,B*N) = text branch, [B*N,2B*N) = dropped text.
__global__ void text_gather_kernel(const int64_t* __restrict__ text, int B, int Nt, int N, int td, int branches,
                                   const float* __restrict__ table, int vocab_rows, const float* __restrict__ freqs_cis, int max_pos,
                                   float* __restrict__ out, uint8_t* __restrict__ rowmask) {
  const size_t total = (size_t)branches * B * N * td;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % td);
    const size_t row = i / td;
    const int n = (int)(row % N);
    const int bb = (int)(row / N);
    const int b = bb % B, branch = bb / B;
    int tok = 0;
    if (n < Nt) tok = (int)text[(size_t)b * Nt + n] + 1;
    // ids outside [-1, vocab) raise in the reference's nn.Embedding; the Python mirror checks them on the host
    // (model/cfm.py), here they are clamped so that a raw C-ABI caller cannot read outside the table
    tok = tok < 0 ? 0 : (tok >= vocab_rows ? vocab_rows - 1 : tok);
    const bool pad = tok == 0;
    if (branch == 1) tok = 0;
    const int pos = n < max_pos ? n : max_pos - 1;
    float v = table[(size_t)tok * td + c] + freqs_cis[(size_t)pos * td + c];
    if (pad) v = 0.f;
    out[i] = v;
    if (c == 0) rowmask[row] = pad ? 1 : 0;
  }
}

// ---- depthwise Conv1d k=7 pad=3 over the sequence, channel-last [B, N, C] (modules.py:250-252,262; vocos ConvNeXtBlock)
__global__ void dwconv7_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                               float* __restrict__ out, int B, int N, int C) {
  const size_t total = (size_t)B * N * C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int n = (int)((i / C) % N);
    const size_t base = i - (size_t)n * C;  // (b, 0, c)
    float acc = bias[c];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int nn = n + j - 3;
      if (nn >= 0 && nn < N) acc = fmaf(w[c * 7 + j], x[base + (size_t)nn * C], acc);
    }
    out[i] = acc;
  }
}

// ---- LayerNorm with affine weight/bias, eps 1e-6, one wave per row (modules.py:253,264; vocos LayerNorms)
template <int D>
__global__ __launch_bounds__(256) void ln_affine_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ out, int M) {
  constexpr int PER = D / 256;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = threadIdx.x & 63;
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * D);
  float4 v[PER];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) { v[i] = xr[lane + 64 * i]; s += v[i].x + v[i].y + v[i].z + v[i].w; }
  const float mean = wave_sum(s) * (1.0f / D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
    q += a * a + bb * bb + c * c + d * d;
  }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + 1e-6f);
  float4* orow = reinterpret_cast<float4*>(out + (size_t)row * D);
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const float4 ww = reinterpret_cast<const float4*>(w)[lane + 64 * i], bv = reinterpret_cast<const float4*>(b)[lane + 64 * i];
    float4 o;
    o.x = (v[i].x - mean) * rstd * ww.x + bv.x;
    o.y = (v[i].y - mean) * rstd * ww.y + bv.y;
    o.z = (v[i].z - mean) * rstd * ww.z + bv.z;
    o.w = (v[i].w - mean) * rstd * ww.w + bv.w;
    orow[lane + 64 * i] = o;
  }
}

// ---- depthwise Conv1d k = 7 followed by the affine LayerNorm, one wave per row (the first two operations of a vocos ConvNeXtBlock): the
// arithmetic of dwconv7_kernel and ln_affine_kernel, statement for statement, without the [rows, C] round trip between them
template <int D>
__global__ __launch_bounds__(256) void dwconv7_ln_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                         const float* __restrict__ lnw, const float* __restrict__ lnb, float* __restrict__ out,
                                                         int B, int N) {
  constexpr int PER = D / 256;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B * N) return;
  const int lane = threadIdx.x & 63;
  const int n = row % N;
  float4 v[PER];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = (lane + 64 * i) * 4;
    float acc[4];
    const float4 b4 = *reinterpret_cast<const float4*>(bias + c);
    acc[0] = b4.x; acc[1] = b4.y; acc[2] = b4.z; acc[3] = b4.w;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int nn = n + j - 3;
      if (nn >= 0 && nn < N) {
        const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)(row + j - 3) * D + c);
        acc[0] = fmaf(w[(c + 0) * 7 + j], xv.x, acc[0]); acc[1] = fmaf(w[(c + 1) * 7 + j], xv.y, acc[1]);
        acc[2] = fmaf(w[(c + 2) * 7 + j], xv.z, acc[2]); acc[3] = fmaf(w[(c + 3) * 7 + j], xv.w, acc[3]);
      }
    }
    v[i] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    s += v[i].x + v[i].y + v[i].z + v[i].w;
  }
  const float mean = wave_sum(s) * (1.0f / D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
    q += a * a + bb * bb + c * c + d * d;
  }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + 1e-6f);
  float4* orow = reinterpret_cast<float4*>(out + (size_t)row * D);
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const float4 ww = reinterpret_cast<const float4*>(lnw)[lane + 64 * i], bv = reinterpret_cast<const float4*>(lnb)[lane + 64 * i];
    float4 o;
    o.x = (v[i].x - mean) * rstd * ww.x + bv.x;
    o.y = (v[i].y - mean) * rstd * ww.y + bv.y;
    o.z = (v[i].z - mean) * rstd * ww.z + bv.z;
    o.w = (v[i].w - mean) * rstd * ww.w + bv.w;
    orow[lane + 64 * i] = o;
  }
}

// ---- GRN (modules.py:225-234): Gx[b,c] = ||x[b,:,c]||_2 over the SEQUENCE; Nx = Gx / (mean_c Gx + 1e-6);
// out = gamma * (x * Nx) + beta + x.   Kernel 1: column norms; kernel 2: apply.
// partial sums of squares: grid (C/64, B, GRN_SPLIT); chunk z covers rows z, z + GRN_SPLIT*4, ... (fixed order: deterministic)
constexpr int GRN_SPLIT = 16;
__global__ __launch_bounds__(256) void grn_norm_kernel(const float* __restrict__ x, float* __restrict__ gx_part, int N, int C) {
  __shared__ float part[4][64];
  const int b = blockIdx.y, z = blockIdx.z, c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  float s = 0.f;
  for (int n = z * 4 + rg; n < N; n += 4 * GRN_SPLIT) {
    const float v = x[((size_t)b * N + n) * C + c];
    s = fmaf(v, v, s);
  }
  part[rg][threadIdx.x & 63] = s;
  __syncthreads();
  if (rg == 0)
    gx_part[((size_t)b * GRN_SPLIT + z) * C + c] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}
// a wave applies GRN_ROWS consecutive rows (of one sample: N % GRN_ROWS is not required, the sample index is re-derived per row):
// the 16 x C/64 partial-norm loads that rebuild Nx are amortised over them instead of repeated for every row
constexpr int GRN_ROWS = 4;
template <int C>
__global__ __launch_bounds__(256) void grn_apply_kernel(float* __restrict__ x, const float* __restrict__ gx,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int M, int N) {
  constexpr int PER = C / 64;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * GRN_ROWS;
  if (row0 >= M) return;
  const int lane = threadIdx.x & 63;
  float g[PER], gm[PER], bt[PER], denom = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) { gm[i] = gamma[lane + 64 * i]; bt[i] = beta[lane + 64 * i]; }
  int bcur = -1;
  for (int r = 0; r < GRN_ROWS; ++r) {
    const int row = row0 + r;
    if (row >= M) break;
    const int b = row / N;
    if (b != bcur) {   // wave-uniform
      bcur = b;
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        float q = 0.f;
#pragma unroll
        for (int z = 0; z < GRN_SPLIT; ++z) q += gx[((size_t)b * GRN_SPLIT + z) * C + lane + 64 * i];
        g[i] = sqrtf(q);
        s += g[i];
      }
      denom = wave_sum(s) * (1.0f / C) + 1e-6f;
    }
    float* xr = x + (size_t)row * C;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = lane + 64 * i;
      const float v = xr[c];
      xr[c] = gm[i] * (v * (g[i] / denom)) + bt[i] + v;
    }
  }
}

// text_embed[bb, n, :] += vec[b, :] for n < nlim (prosody text conditioning, dit.py:225-233, both CFG branches)
__global__ void add_rowvec_kernel(float* __restrict__ x, const float* __restrict__ vec, int BB, int B, int N, int C, int nlim) {
  const size_t total = (size_t)BB * N * C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int n = (int)((i / C) % N);
    const int bb = (int)(i / ((size_t)C * N));
    if (n < nlim) x[i] += vec[(size_t)(bb % B) * C + c];
  }
}

// conditioning setup (cfm.py:311-318,326-327,388-390): cond_eff = pad(cond) [+ Linear(pad(prosody))]; step_cond = mask ? cond_eff : 0
__global__ void cond_prepare_kernel(const float* __restrict__ cond, const uint8_t* __restrict__ mask,
                                    const float* __restrict__ pm /*[B,100] W.e (no bias) or null*/,
                                    const float* __restrict__ pbias, int B, int N, int F, int md, int crows /* rows per sample in cond */,
                                    float* __restrict__ cond_eff, float* __restrict__ step_cond) {
  const size_t total = (size_t)B * N * md;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % md);
    const size_t row = i / md;
    const int n = (int)(row % N), b = (int)(row / N);
    float v = n < crows ? cond[((size_t)b * crows + n) * md + c] : 0.f;      // rows past crows: the reference's zero right-padding (cfm.py:311)
    if (pm) v += (n < F ? pm[(size_t)b * md + c] : 0.f) + pbias[c];
    cond_eff[i] = v;
    step_cond[i] = mask[row] ? v : 0.f;
  }
}

// CT[bb, n, :] = [ step_cond (branch 0) or 0 (branch 1) | text_embed[bb, n, :] ]   (dit.py:94-97, x part hoisted out)
// output rows live in the padded row space [bb][pitch]; rows n >= N are zero
__global__ void concat_ct_kernel(const float* __restrict__ step_cond, const float* __restrict__ te, int B, int N, int md,
                                 int td, int branches, int pitch, float* __restrict__ ct) {
  const int W = md + td;
  const size_t total = (size_t)branches * B * pitch * W;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % W);
    const size_t prow = i / W;
    const int bb = (int)(prow / pitch), n = (int)(prow % pitch);
    float v = 0.f;
    if (n < N) {
      const size_t row = (size_t)bb * N + n;
      if (c < md) v = bb < B ? step_cond[row * md + c] : 0.f;
      else v = te[row * td + (c - md)];
    }
    ct[i] = v;
  }
}

// SinusPositionEmbedding(256)(t, scale=1000) (modules.py:149-161): [S, 256] = [sin(1000 t f_i) | cos(1000 t f_i)]
__global__ void time_sinus_kernel(const float* __restrict__ t, const float* __restrict__ freqs, int S, int half,
                                  float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S * half) return;
  const int s = i / half, j = i - s * half;
  const float a = (1000.0f * t[s]) * freqs[j];
  out[(size_t)s * 2 * half + j] = sinf(a);
  out[(size_t)s * 2 * half + half + j] = cosf(a);
}
__global__ void silu_kernel(const float* __restrict__ x, float* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = silu_f(x[i]);
}

// ---- Vocos: mel [B, C, L] -> im2col rows for Conv1d(C -> 512, k=7, pad=3): col[b*L+n][ci*7+j] = mel[b][ci][n+j-3]
// mel element (b, ci, n) at mel[b * sb + ci * sc + n * sl]: [B, C, L] is (C L, L, 1); a frames-first slice [B][L][C] of the sampler's output is
// (rows per sample * C, 1, C) -- the vocoder then reads the generated frames in place, no permute copy in between
__global__ void im2col7_kernel(const float* __restrict__ mel, int B, int C, int L, long sb, long sc, long sl, float* __restrict__ col) {
  const int W = C * 7;
  const size_t total = (size_t)B * L * W;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % W);
    const size_t row = i / W;
    const int n = (int)(row % L), b = (int)(row / L);
    const int ci = k / 7, j = k - ci * 7;
    const int nn = n + j - 3;
    col[i] = (nn >= 0 && nn < L) ? mel[(long)b * sb + (long)ci * sc + (long)nn * sl] : 0.f;
  }
}

// ISTFTHead (vocos heads.py): x[.., :513] -> mag = clip(exp(.), max 100); x[.., 513:] -> phase; S = mag (cos p + i sin p)
// spec row layout: [re_0..re_512 | im_0..im_512 | 0 0]  (ld = 1028, K padded to a multiple of 4 for the DFT GEMM)
__global__ void spec_kernel(const float* __restrict__ head, int rows, int nb, int ldh, int lds_, float* __restrict__ spec) {
  const size_t total = (size_t)rows * nb;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % nb);
    const size_t r = i / nb;
    const float mag = fminf(expf(head[r * ldh + k]), 100.0f);
    float sn, cs;
    sincosf(head[r * ldh + nb + k], &sn, &cs);
    spec[r * lds_ + k] = mag * cs;
    spec[r * lds_ + nb + k] = mag * sn;
    if (k < lds_ - 2 * nb) spec[r * lds_ + 2 * nb + k] = 0.f;
  }
}

// windowed inverse real DFT as a matrix: frame[n] = w[n]/N * sum_k c_k (re_k cos(2 pi k n/N) - im_k sin(2 pi k n/N))
__global__ void dft_basis_kernel(const float* __restrict__ window, int nfft, int ld, float* __restrict__ basis) {
  const int nb = nfft / 2 + 1;
  const size_t total = (size_t)nfft * ld;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % ld), n = (int)(i / ld);
    float v = 0.f;
    if (col < 2 * nb) {
      const int k = col < nb ? col : col - nb;
      const float ck = (k == 0 || k == nfft / 2) ? 1.0f : 2.0f;
      const int ph = (int)(((long long)k * n) % nfft);       // exact argument reduction
      const float x = 2.0f * (float)ph / (float)nfft;         // angle / pi
      const float tr = col < nb ? cospif(x) : -sinpif(x);
      v = window[n] * ck * tr / (float)nfft;
    }
    basis[i] = v;
  }
}

// overlap-add + window-envelope normalisation + center trim (torch.istft(center=True)): wav[b][t], t in [0, hop*(L-1))
__global__ void overlap_add_kernel(const float* __restrict__ frames, const float* __restrict__ window, int B, int L,
                                   int nfft, int hop, float* __restrict__ wav) {
  const int T = hop * (L - 1);
  const size_t total = (size_t)B * T;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % T), b = (int)(i / T);
    const int tt = t + nfft / 2;
    int f_hi = tt / hop;
    if (f_hi > L - 1) f_hi = L - 1;
    int f_lo = (tt - nfft + hop) / hop;  // smallest f with tt - hop f < nfft
    if (f_lo < 0) f_lo = 0;
    float acc = 0.f, env = 0.f;
    for (int f = f_lo; f <= f_hi; ++f) {
      const int n = tt - hop * f;
      if (n >= 0 && n < nfft) {
        acc += frames[((size_t)b * L + f) * nfft + n];
        const float w = window[n];
        env = fmaf(w, w, env);
      }
    }
    wav[i] = acc / env;
  }
}

__global__ void scale_kernel(float* __restrict__ x, float s, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] *= s;
}

}  // namespace

#define LAUNCH(kern, total, ...) \
  hipLaunchKernelGGL(kern, dim3(grid_for(total)), dim3(256), 0, s, __VA_ARGS__); \
  return hipGetLastError();

hipError_t launch_text_gather(const int64_t* text, int B, int Nt, int N, int td, int branches, const float* table, int vocab_rows,
                              const float* freqs_cis, int max_pos, float* out, uint8_t* rowmask, hipStream_t s) {
  LAUNCH(text_gather_kernel, (size_t)branches * B * N * td, text, B, Nt, N, td, branches, table, vocab_rows, freqs_cis, max_pos, out, rowmask)
}
hipError_t launch_dwconv7(const float* x, const float* w, const float* bias, float* out, int B, int N, int C, hipStream_t s) {
  LAUNCH(dwconv7_kernel, (size_t)B * N * C, x, w, bias, out, B, N, C)
}
hipError_t launch_ln_affine(const float* x, const float* w, const float* b, float* out, int M, int D, hipStream_t s) {
  if (D == 512) hipLaunchKernelGGL(ln_affine_kernel<512>, dim3((M + 3) / 4), dim3(256), 0, s, x, w, b, out, M);
  else if (D == 1024) hipLaunchKernelGGL(ln_affine_kernel<1024>, dim3((M + 3) / 4), dim3(256), 0, s, x, w, b, out, M);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
hipError_t launch_grn(float* x, float* gx_scratch, const float* gamma, const float* beta, int B, int N, int C, hipStream_t s) {
  if (C != 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(grn_norm_kernel, dim3(C / 64, B, GRN_SPLIT), dim3(256), 0, s, x, gx_scratch, N, C);   // gx_scratch: [B][GRN_SPLIT][C]
  hipLaunchKernelGGL(grn_apply_kernel<1024>, dim3((B * N + 4 * GRN_ROWS - 1) / (4 * GRN_ROWS)), dim3(256), 0, s, x, gx_scratch, gamma, beta, B * N, N);
  return hipGetLastError();
}
hipError_t launch_add_rowvec(float* x, const float* vec, int BB, int B, int N, int C, int nlim, hipStream_t s) {
  LAUNCH(add_rowvec_kernel, (size_t)BB * N * C, x, vec, BB, B, N, C, nlim)
}
hipError_t launch_cond_prepare(const float* cond, const uint8_t* mask, const float* pm, const float* pbias, int B, int N,
                               int F, int md, int crows, float* cond_eff, float* step_cond, hipStream_t s) {
  LAUNCH(cond_prepare_kernel, (size_t)B * N * md, cond, mask, pm, pbias, B, N, F, md, crows, cond_eff, step_cond)
}
hipError_t launch_concat_ct(const float* step_cond, const float* te, int B, int N, int md, int td, int branches, int pitch,
                            float* ct, hipStream_t s) {
  LAUNCH(concat_ct_kernel, (size_t)branches * B * pitch * (md + td), step_cond, te, B, N, md, td, branches, pitch, ct)
}
hipError_t launch_time_sinus(const float* t, const float* freqs, int S, int half, float* out, hipStream_t s) {
  hipLaunchKernelGGL(time_sinus_kernel, dim3((S * half + 255) / 256), dim3(256), 0, s, t, freqs, S, half, out);
  return hipGetLastError();
}
hipError_t launch_silu(const float* x, float* out, size_t n, hipStream_t s) { LAUNCH(silu_kernel, n, x, out, n) }
hipError_t launch_im2col7(const float* mel, int B, int C, int L, long sb, long sc, long sl, float* col, hipStream_t s) {
  LAUNCH(im2col7_kernel, (size_t)B * L * C * 7, mel, B, C, L, sb, sc, sl, col)
}
hipError_t launch_dwconv7_ln(const float* x, const float* w, const float* bias, const float* lnw, const float* lnb, float* out, int B, int N,
                             int C, hipStream_t s) {
  if (C != 512) return hipErrorInvalidValue;
  hipLaunchKernelGGL(dwconv7_ln_kernel<512>, dim3((B * N + 3) / 4), dim3(256), 0, s, x, w, bias, lnw, lnb, out, B, N);
  return hipGetLastError();
}
hipError_t launch_spec(const float* head, int rows, int nb, int ldh, int lds_, float* spec, hipStream_t s) {
  LAUNCH(spec_kernel, (size_t)rows * nb, head, rows, nb, ldh, lds_, spec)
}
hipError_t launch_dft_basis(const float* window, int nfft, int ld, float* basis, hipStream_t s) {
  LAUNCH(dft_basis_kernel, (size_t)nfft * ld, window, nfft, ld, basis)
}
hipError_t launch_overlap_add(const float* frames, const float* window, int B, int L, int nfft, int hop, float* wav,
                              hipStream_t s) {
  LAUNCH(overlap_add_kernel, (size_t)B * hop * (L - 1), frames, window, B, L, nfft, hop, wav)
}
hipError_t launch_scale(float* x, float sc, size_t n, hipStream_t s) { LAUNCH(scale_kernel, n, x, sc, n) }

// ================================================================================================================
// wav -> log-mel front edge ("next" row f-1): lemas_tts/model/modules.py:75-101 get_vocos_mel_spectrogram
// (torchaudio MelSpectrogram: reflect pad n_fft/2, periodic Hann, |STFT| (power 1), HTK mel filterbank, then
// clamp(1e-5).log()).  The DFT is a GEMM against a precomputed basis, like the vocoder's inverse.
namespace {

// frames[b*F + f][n] = window[n] * reflect(wav[b])[f*hop + n - nfft/2]
__global__ void stft_frames_kernel(const float* __restrict__ wav, const float* __restrict__ window, int B, int nw, int F,
                                   int nfft, int hop, float* __restrict__ frames) {
  const size_t total = (size_t)B * F * nfft;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % nfft);
    const size_t row = i / nfft;
    const int f = (int)(row % F), b = (int)(row / F);
    int t = f * hop + n - nfft / 2;
    if (t < 0) t = -t;                         // reflect (no edge repeat), as torch.stft(center=True, pad_mode="reflect")
    if (t >= nw) t = 2 * (nw - 1) - t;
    frames[i] = window[n] * wav[(size_t)b * nw + t];
  }
}
// forward real-DFT basis rows: k < nb -> cos(2 pi k n / N), nb <= k < 2 nb -> -sin(2 pi (k-nb) n / N), rest 0
__global__ void rdft_basis_kernel(int nfft, int rows, float* __restrict__ basis) {
  const int nb = nfft / 2 + 1;
  const size_t total = (size_t)rows * nfft;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % nfft), r = (int)(i / nfft);
    float v = 0.f;
    if (r < 2 * nb) {
      const int k = r < nb ? r : r - nb;
      const float x = 2.0f * (float)(int)(((long long)k * n) % nfft) / (float)nfft;
      v = r < nb ? cospif(x) : -sinpif(x);
    }
    basis[i] = v;
  }
}
// mag[row][k] = sqrt(re^2 + im^2), padded to ldm columns with zeros
__global__ void magnitude_kernel(const float* __restrict__ spec, int rows, int nb, int lds_, int ldm, float* __restrict__ mag) {
  const size_t total = (size_t)rows * ldm;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % ldm);
    const size_t r = i / ldm;
    float v = 0.f;
    if (k < nb) {
      const float re = spec[r * lds_ + k], im = spec[r * lds_ + nb + k];
      v = sqrtf(re * re + im * im);
    }
    mag[i] = v;
  }
}
__global__ void log_clamp_kernel(float* __restrict__ x, size_t n, float lo) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    x[i] = logf(fmaxf(x[i], lo));
}

}  // namespace

hipError_t launch_stft_frames(const float* wav, const float* window, int B, int nw, int F, int nfft, int hop, float* frames,
                              hipStream_t s) {
  LAUNCH(stft_frames_kernel, (size_t)B * F * nfft, wav, window, B, nw, F, nfft, hop, frames)
}
hipError_t launch_rdft_basis(int nfft, int rows, float* basis, hipStream_t s) {
  LAUNCH(rdft_basis_kernel, (size_t)rows * nfft, nfft, rows, basis)
}
hipError_t launch_magnitude(const float* spec, int rows, int nb, int lds_, int ldm, float* mag, hipStream_t s) {
  LAUNCH(magnitude_kernel, (size_t)rows * ldm, spec, rows, nb, lds_, ldm, mag)
}
hipError_t launch_log_clamp(float* x, size_t n, float lo, hipStream_t s) { LAUNCH(log_clamp_kernel, n, x, n, lo) }
