// Real FFT / inverse real FFT of the rows of an STFT, two rows per workgroup, entirely in LDS (gfx950).  Behind lemas_stft_forward / _inverse
// (engine_stft.hip) for the UVR5 denoiser's n_fft = 7680 = 4^4 * 2 * 3 * 5 (Kim_Vocal_1; 6144 for other models) (uvr5/multiprocess_cuda_infer.py:206-223: torch.stft /
// torch.istft).  Round 6 measured that transform as a GEMM against a precomputed DFT basis -- 60 GFLOP per 256-frame stereo chunk, 576 us per
// launch at 0.43 of the fp32 matrix peak, 2.3 ms of a 31.7 ms denoise (profiles/r06/r06mx_kernel_stats_mdx_denoise.txt) -- for arithmetic an FFT
// does in well under 1 GFLOP: this is integer-indexed shuffling of 60 KB per row pair, LDS-bound, not matrix work.
//   * Stockham autosort, mixed radix 4 / 2 / 3 / 5 (any N = 2^a 3^b 5^c, even, <= 8192): stage with radix R and Ns = product of the radices before it,
//       j in [0, N / R):  k = j mod Ns;  v_r = x[j + r N / R] * w^(r k N / (Ns R));  V = DFT_R(v);  y[(j div Ns) Ns R + k + r Ns] = V_r
//     two LDS buffers (2 x 8 N bytes), one barrier per stage, output in natural order; w^t = exp(-2 pi i t / N) from a table built in double
//     precision on the host (conjugated for the inverse).
//   * two REAL rows ride one complex transform (row a in the real part, row b in the imaginary part):
//       forward   X_a[k] = (Z[k] + conj Z[N-k]) / 2,  X_b[k] = (Z[k] - conj Z[N-k]) / (2 i)
//       inverse   Z[k] = X_a[k] + i X_b[k],  Z[N-k] = conj X_a[k] + i conj X_b[k]   (imaginary parts of the DC and Nyquist bins dropped, as irfft)
// Layouts are engine_stft.hip's: frames [rows][N] (window already applied / applied here with 1 / N for the inverse), spectra [rows][ld] as
// [re(0..nb-1) | im(0..nb-1) | zero padding], nb = N / 2 + 1.
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// the transform of the N values in `x` (LDS); returns the buffer that holds the result
template <bool INV>
__device__ __forceinline__ float2* fft_in_lds(const FftPlan& plan, float2* x, float2* y, const float2* __restrict__ tw) {
  const int N = plan.n;
  int Ns = 1;
  for (int st = 0; st < plan.nrad; ++st) {
    const int R = plan.rad[st], M = N / R, step = N / (Ns * R);
    for (int j = threadIdx.x; j < M; j += blockDim.x) {
      const int q = j / Ns, k = j - q * Ns;
      const int j0 = q * Ns * R + k;
      auto twid = [&](int r) {                    // w^(r k step); r k step < N
        float2 w = tw[r * k * step];
        if (INV) w.y = -w.y;
        return w;
      };
      const float2 v0 = x[j];
      if (R == 2) {
        const float2 v1 = cmul(x[j + M], twid(1));
        y[j0] = cadd(v0, v1);
        y[j0 + Ns] = csub(v0, v1);
      } else if (R == 4) {
        const float2 v1 = cmul(x[j + M], twid(1)), v2 = cmul(x[j + 2 * M], twid(2)), v3 = cmul(x[j + 3 * M], twid(3));
        const float2 a0 = cadd(v0, v2), a1 = csub(v0, v2), b0 = cadd(v1, v3), d = csub(v1, v3);
        const float2 b1 = INV ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);      // (v1 - v3) * (+-i)
        y[j0] = cadd(a0, b0);
        y[j0 + Ns] = cadd(a1, b1);
        y[j0 + 2 * Ns] = csub(a0, b0);
        y[j0 + 3 * Ns] = csub(a1, b1);
      } else if (R == 5) {
        const float2 v1 = cmul(x[j + M], twid(1)), v2 = cmul(x[j + 2 * M], twid(2)), v3 = cmul(x[j + 3 * M], twid(3)), v4 = cmul(x[j + 4 * M], twid(4));
        constexpr float c1 = 0.30901699437494745f, c2 = -0.8090169943749475f, s1 = 0.9510565162951535f, s2 = 0.5877852522924731f;
        const float2 t1 = cadd(v1, v4), t2 = cadd(v2, v3), t3 = csub(v1, v4), t4 = csub(v2, v3);
        const float2 a1 = make_float2(v0.x + c1 * t1.x + c2 * t2.x, v0.y + c1 * t1.y + c2 * t2.y);
        const float2 a2 = make_float2(v0.x + c2 * t1.x + c1 * t2.x, v0.y + c2 * t1.y + c1 * t2.y);
        const float2 b1 = make_float2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y);
        const float2 b2 = make_float2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y);
        // forward: V1 = a1 - i b1, V4 = a1 + i b1, V2 = a2 - i b2, V3 = a2 + i b2; the inverse swaps the signs.  -i (x + i y) = y - i x
        const float2 ib1 = INV ? make_float2(-b1.y, b1.x) : make_float2(b1.y, -b1.x);
        const float2 ib2 = INV ? make_float2(-b2.y, b2.x) : make_float2(b2.y, -b2.x);
        y[j0] = make_float2(v0.x + t1.x + t2.x, v0.y + t1.y + t2.y);
        y[j0 + Ns] = cadd(a1, ib1);
        y[j0 + 2 * Ns] = cadd(a2, ib2);
        y[j0 + 3 * Ns] = csub(a2, ib2);
        y[j0 + 4 * Ns] = csub(a1, ib1);
      } else {                                    // R == 3
        const float2 v1 = cmul(x[j + M], twid(1)), v2 = cmul(x[j + 2 * M], twid(2));
        const float s = INV ? 0.8660254037844386f : -0.8660254037844386f;
        const float2 t1 = cadd(v1, v2), d = csub(v1, v2);
        const float2 t2 = make_float2(v0.x - 0.5f * t1.x, v0.y - 0.5f * t1.y);
        const float2 t3 = make_float2(-d.y * s, d.x * s);                              // (v1 - v2) * (i s)
        y[j0] = cadd(v0, t1);
        y[j0 + Ns] = cadd(t2, t3);
        y[j0 + 2 * Ns] = csub(t2, t3);
      }
    }
    __syncthreads();
    float2* t = x; x = y; y = t;
    Ns *= R;
  }
  return x;
}

// frames [rows][N] -> spec [rows][ld]
__global__ __launch_bounds__(256) void rfft_rows_kernel(const FftPlan plan, const float2* __restrict__ tw, const float* __restrict__ frames, int rows,
                                                        float* __restrict__ spec, int ld) {
  extern __shared__ __attribute__((aligned(16))) float2 fft_lds[];
  const int N = plan.n, nb = N / 2 + 1;
  const int ra = 2 * blockIdx.x, rb = ra + 1;
  const bool hb = rb < rows;
  float2* x = fft_lds;
  float2* y = fft_lds + N;
  const float* fa = frames + (size_t)ra * N;
  const float* fb = frames + (size_t)(hb ? rb : ra) * N;
  for (int n = threadIdx.x; n < N; n += blockDim.x) x[n] = make_float2(fa[n], hb ? fb[n] : 0.f);
  __syncthreads();
  const float2* z = fft_in_lds<false>(plan, x, y, tw);
  float* sa = spec + (size_t)ra * ld;
  float* sb = spec + (size_t)rb * ld;
  for (int k = threadIdx.x; k < nb; k += blockDim.x) {
    const float2 zk = z[k], zn = z[k == 0 ? 0 : N - k];
    // X_a = (zk + conj zn) / 2, X_b = (zk - conj zn) / (2 i)
    sa[k] = 0.5f * (zk.x + zn.x);
    sa[nb + k] = 0.5f * (zk.y - zn.y);
    if (hb) {
      sb[k] = 0.5f * (zk.y + zn.y);
      sb[nb + k] = -0.5f * (zk.x - zn.x);
    }
  }
  for (int c = 2 * nb + threadIdx.x; c < ld; c += blockDim.x) {      // the padding columns a GEMM against zero basis rows left at zero
    sa[c] = 0.f;
    if (hb) sb[c] = 0.f;
  }
}

// spec [rows][ld] -> frames [rows][N] = window[n] / N * irfft(spec row)[n]
__global__ __launch_bounds__(256) void irfft_rows_kernel(const FftPlan plan, const float2* __restrict__ tw, const float* __restrict__ spec, int ld, int rows,
                                                         const float* __restrict__ window, float* __restrict__ frames) {
  extern __shared__ __attribute__((aligned(16))) float2 fft_lds[];
  const int N = plan.n, nb = N / 2 + 1;
  const int ra = 2 * blockIdx.x, rb = ra + 1;
  const bool hb = rb < rows;
  float2* x = fft_lds;
  float2* y = fft_lds + N;
  const float* sa = spec + (size_t)ra * ld;
  const float* sb = spec + (size_t)(hb ? rb : ra) * ld;
  for (int k = threadIdx.x; k < nb; k += blockDim.x) {
    const bool edge = k == 0 || 2 * k == N;
    const float ar = sa[k], ai = edge ? 0.f : sa[nb + k];
    const float br = hb ? sb[k] : 0.f, bi = (edge || !hb) ? 0.f : sb[nb + k];
    x[k] = make_float2(ar - bi, ai + br);                     // X_a + i X_b
    if (!edge) x[N - k] = make_float2(ar + bi, br - ai);      // conj X_a + i conj X_b
  }
  __syncthreads();
  const float2* z = fft_in_lds<true>(plan, x, y, tw);
  const float inv_n = 1.0f / (float)N;
  float* fa = frames + (size_t)ra * N;
  float* fb = frames + (size_t)rb * N;
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    const float w = window[n] * inv_n;
    const float2 v = z[n];
    fa[n] = v.x * w;
    if (hb) fb[n] = v.y * w;
  }
}

}  // namespace

bool fft_plan_make(int n, FftPlan* plan) {
  if (n < 4 || (n & 1) || n > 8192) return false;
  FftPlan p{};
  p.n = n;
  int m = n;
  while (m % 4 == 0) { p.rad[p.nrad++] = 4; m /= 4; }
  while (m % 2 == 0) { p.rad[p.nrad++] = 2; m /= 2; }
  while (m % 3 == 0) { if (p.nrad >= 14) return false; p.rad[p.nrad++] = 3; m /= 3; }
  while (m % 5 == 0) { if (p.nrad >= 14) return false; p.rad[p.nrad++] = 5; m /= 5; }
  if (m != 1) return false;
  *plan = p;
  return true;
}
// the > 64 KB dynamic-LDS opt-in of both kernels: once per engine, never on a launch path.  Always the largest plan's footprint (the attribute
// belongs to the kernel, not to the engine: a second engine with a shorter transform must not lower it under the first one's launches)
hipError_t fft_kernels_init(const FftPlan&) {
  const int lds = 8192 * 16;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rfft_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(irfft_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}
hipError_t launch_rfft_rows(const FftPlan& plan, const float* tw, const float* frames, int rows, float* spec, int ld, hipStream_t s) {
  if (rows <= 0 || ld < plan.n + 2) return hipErrorInvalidValue;
  hipLaunchKernelGGL(rfft_rows_kernel, dim3((rows + 1) / 2), dim3(256), (size_t)plan.n * 16, s, plan, reinterpret_cast<const float2*>(tw), frames, rows, spec, ld);
  return hipGetLastError();
}
hipError_t launch_irfft_rows(const FftPlan& plan, const float* tw, const float* spec, int ld, int rows, const float* window, float* frames, hipStream_t s) {
  if (rows <= 0 || ld < plan.n + 2) return hipErrorInvalidValue;
  hipLaunchKernelGGL(irfft_rows_kernel, dim3((rows + 1) / 2), dim3(256), (size_t)plan.n * 16, s, plan, reinterpret_cast<const float2*>(tw), spec, ld, rows, window, frames);
  return hipGetLastError();
}
