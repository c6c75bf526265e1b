// lemas_mel: reference wav -> log-mel (the "next" row f-1 of SURVEY.md section 8f), the front edge that feeds CFM.sample.
// Replaces lemas_tts/model/modules.py:75-101 get_vocos_mel_spectrogram / :104-143 MelSpec.forward (call site
// lemas_tts/model/cfm.py:232-236).  The arithmetic is torchaudio's MelSpectrogram (third party, not in the tree and not
// installed: PARITY UNPINNED): reflect-pad n_fft/2, periodic Hann window, |rDFT| (power 1), HTK-scale triangular
// filterbank without normalisation over [0, sr/2], then clamp(min=1e-5).log().
#include <cmath>
#include <vector>

#include "engine_common.h"

using namespace lemas;

struct lemas_mel {
  int nfft = 1024, hop = 256, nmels = 100, sr = 24000;
  int nb = 513, ldk = 1028, ldm = 516;
  DevBuf window, basis, fbt;   // [nfft], [ldk][nfft], [nmels][ldm]
  DevBuf d_frames, d_spec, d_mag;

  ~lemas_mel() {
    for (DevBuf* b : {&window, &basis, &fbt, &d_frames, &d_spec, &d_mag}) b->release();
  }
  int init() {
    nb = nfft / 2 + 1;
    ldk = (2 * nb + 3) & ~3;
    ldm = (nb + 3) & ~3;
    // periodic Hann (torch.hann_window default) and the HTK filterbank of torchaudio.functional.melscale_fbanks
    std::vector<float> w(nfft), fb((size_t)nmels * ldm, 0.f);
    for (int n = 0; n < nfft; ++n) w[n] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * n / nfft));
    auto hz2mel = [](double f) { return 2595.0 * std::log10(1.0 + f / 700.0); };
    auto mel2hz = [](double m) { return 700.0 * (std::pow(10.0, m / 2595.0) - 1.0); };
    const double fmax = sr / 2, mmin = hz2mel(0.0), mmax = hz2mel(fmax);
    std::vector<double> fpts(nmels + 2);
    for (int i = 0; i < nmels + 2; ++i) fpts[i] = mel2hz(mmin + (mmax - mmin) * i / (nmels + 1));
    for (int k = 0; k < nb; ++k) {
      const double f = fmax * k / (nb - 1);      // linspace(0, sr // 2, n_freqs)
      for (int m = 0; m < nmels; ++m) {
        const double down = (f - fpts[m]) / (fpts[m + 1] - fpts[m]);
        const double up = (fpts[m + 2] - f) / (fpts[m + 2] - fpts[m + 1]);
        const double v = std::fmax(0.0, std::fmin(down, up));
        fb[(size_t)m * ldm + k] = (float)v;
      }
    }
    RC_TRY(window.ensure((size_t)nfft * 4));
    RC_TRY(fbt.ensure(fb.size() * 4));
    RC_TRY(basis.ensure((size_t)ldk * nfft * 4));
    HIP_TRY(hipMemcpy(window.p, w.data(), (size_t)nfft * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(fbt.p, fb.data(), fb.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(launch_rdft_basis(nfft, ldk, basis.as<float>(), nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return 0;
  }
  int forward(const float* wav, int B, int nw, float* mel, hipStream_t s) {
    if (!wav || !mel || B <= 0 || nw <= nfft / 2) { set_error("lemas_mel_forward: bad arguments (B=%d nw=%d)", B, nw); return LEMAS_E_ARG; }
    const int F = nw / hop + 1, rows = B * F;
    RC_TRY(d_frames.ensure((size_t)rows * nfft * 4));
    RC_TRY(d_spec.ensure((size_t)rows * ldk * 4));
    RC_TRY(d_mag.ensure((size_t)rows * ldm * 4));
    HIP_TRY(launch_stft_frames(wav, window.as<float>(), B, nw, F, nfft, hop, d_frames.as<float>(), s));
    GemmF32Params g{};
    g.A = d_frames.as<float>(); g.lda = nfft; g.W = basis.as<float>(); g.ldw = nfft; g.out = d_spec.as<float>(); g.ldc = ldk;
    g.M = rows; g.N = ldk; g.K = nfft;
    HIP_TRY(launch_gemm_f32(F32_BIAS, g, s));
    HIP_TRY(launch_magnitude(d_spec.as<float>(), rows, nb, ldk, ldm, d_mag.as<float>(), s));
    GemmF32Params h{};
    h.A = d_mag.as<float>(); h.lda = ldm; h.W = fbt.as<float>(); h.ldw = ldm; h.out = mel; h.ldc = nmels; h.M = rows; h.N = nmels; h.K = ldm;
    HIP_TRY(launch_gemm_f32(F32_BIAS, h, s));
    HIP_TRY(launch_log_clamp(mel, (size_t)rows * nmels, 1e-5f, s));
    return 0;
  }
};

extern "C" {

int lemas_mel_create(int32_t n_fft, int32_t hop_length, int32_t n_mels, int32_t sample_rate, lemas_mel** out) {
  if (!out || n_fft <= 0 || (n_fft & 3) || hop_length <= 0 || n_mels <= 0 || (n_mels & 3)) { set_error("lemas_mel_create: bad arguments"); return LEMAS_E_ARG; }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    set_error("lemas_mel_create: no HIP device (this library has no CPU path)");
    return e != hipSuccess ? -(int)e : LEMAS_E_STATE;
  }
  lemas_mel* m = new lemas_mel();
  m->nfft = n_fft; m->hop = hop_length; m->nmels = n_mels; m->sr = sample_rate;
  int rc = m->init();
  if (rc != 0) { delete m; return rc; }
  *out = m;
  return 0;
}
void lemas_mel_destroy(lemas_mel* m) { delete m; }
int lemas_mel_forward(lemas_mel* m, const float* wav, int32_t batch, int32_t samples, float* mel, void* stream) {
  if (!m) return LEMAS_E_ARG;
  return m->forward(wav, batch, samples, mel, (hipStream_t)stream);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------
// lemas_resample: prompt resampling to 24 kHz, the other half of the front edge.  Replaces the
// torchaudio.transforms.Resample call at lemas_tts/infer/utils_infer.py:494-496 (also cfm.py:254).  Arithmetic: torchaudio's
// sinc_interp_hann polyphase resampler with its defaults (lowpass_filter_width 6, rolloff 0.99) -- third party, not in
// the tree, not installed: PARITY UNPINNED (restated from the published algorithm; oracle: resample_sinc_hann).
//   o = orig / gcd, n = new / gcd, base = min(o, n) * rolloff, width = ceil(lpw * o / base)
//   kernel[p][k] = sinc(pi t) * cos^2(pi t / (2 lpw)) * base / o,  t = clamp((-p / n + (k - width) / o) * base, +-lpw)
//   out[m n + p] = sum_k kernel[p][k] * xpad[m o + k],  xpad = x padded (width, width + o), length ceil(n * len / o)
namespace {
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, int B, int len, const float* __restrict__ kern,
                                                       int o, int n, int width, int klen, float* __restrict__ out, int out_len) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= (size_t)B * out_len) return;
  const int b = (int)(i / out_len), j = (int)(i - (size_t)b * out_len);
  const int m = j / n, p = j - m * n;
  const float* xr = x + (size_t)b * len;
  const float* kr = kern + (size_t)p * klen;
  const int base = m * o - width;          // xpad[m o + k] = x[m o + k - width]
  float acc = 0.f;
  for (int k = 0; k < klen; ++k) {
    const int idx = base + k;
    const float v = (idx >= 0 && idx < len) ? xr[idx] : 0.f;
    acc = fmaf(kr[k], v, acc);
  }
  out[i] = acc;
}
}  // namespace

struct lemas_resample {
  int o = 1, n = 1, width = 0, klen = 0;
  DevBuf kern;
  ~lemas_resample() { kern.release(); }
};

extern "C" {

int lemas_resample_create(int32_t orig_freq, int32_t new_freq, lemas_resample** out) {
  if (!out || orig_freq <= 0 || new_freq <= 0) { set_error("lemas_resample_create: bad arguments"); return LEMAS_E_ARG; }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    set_error("lemas_resample_create: no HIP device (this library has no CPU path)");
    return e != hipSuccess ? -(int)e : LEMAS_E_STATE;
  }
  int a = orig_freq, b = new_freq;
  while (b) { const int t = a % b; a = b; b = t; }
  lemas_resample* r = new lemas_resample();
  r->o = orig_freq / a; r->n = new_freq / a;
  const double lpw = 6.0, rolloff = 0.99;
  const double base = (double)(r->o < r->n ? r->o : r->n) * rolloff;
  r->width = (int)std::ceil(lpw * r->o / base);
  r->klen = 2 * r->width + r->o;
  std::vector<float> k((size_t)r->n * r->klen);
  for (int p = 0; p < r->n; ++p)
    for (int j = 0; j < r->klen; ++j) {
      double t = (-(double)p / r->n + (double)(j - r->width) / r->o) * base;
      t = t < -lpw ? -lpw : (t > lpw ? lpw : t);
      const double c = std::cos(t * M_PI / lpw / 2.0);
      const double tp = t * M_PI;
      const double sinc = tp == 0.0 ? 1.0 : std::sin(tp) / tp;
      k[(size_t)p * r->klen + j] = (float)(sinc * c * c * (base / r->o));
    }
  int rc = r->kern.ensure(k.size() * 4);
  if (rc == 0 && hipMemcpy(r->kern.p, k.data(), k.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { set_error("lemas_resample_create: upload failed"); rc = LEMAS_E_STATE; }
  if (rc != 0) { delete r; return rc; }
  *out = r;
  return 0;
}
void lemas_resample_destroy(lemas_resample* r) { delete r; }
int64_t lemas_resample_out_len(const lemas_resample* r, int64_t samples) {
  if (!r || samples < 0) return -1;
  return (samples * r->n + r->o - 1) / r->o;
}
int lemas_resample_forward(lemas_resample* r, const float* wav, int32_t batch, int32_t samples, float* out, void* stream) {
  if (!r || !wav || !out || batch <= 0 || samples <= 0) { set_error("lemas_resample_forward: bad arguments"); return LEMAS_E_ARG; }
  const int out_len = (int)lemas_resample_out_len(r, samples);
  const size_t total = (size_t)batch * out_len;
  hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wav, batch, samples,
                     r->kern.as<float>(), r->o, r->n, r->width, r->klen, out, out_len);
  HIP_TRY(hipGetLastError());
  return 0;
}

}  // extern "C"
