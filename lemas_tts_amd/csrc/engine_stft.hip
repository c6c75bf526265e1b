// lemas_stft: the STFT / inverse STFT pair around the UVR5 MDX-Net denoiser of the reference prompt ("next" row f-4 of SURVEY.md
// section 8f).  Replaces uvr5/multiprocess_cuda_infer.py:206-223 Inference.stft / .istft, whose arithmetic is
//   torch.stft (n_fft, hop_length, window, center=True [reflect pad n_fft/2], onesided, return_complex)        and
//   torch.istft(n_fft, hop_length, window, center=True): windowed inverse real DFT of every frame, overlap-add, division by the
//   overlap-added squared window, trim of n_fft/2 at both ends -> hop * (frames - 1) samples.
// A transform length of the form 2^a 3^b 5^c <= 8192 (the denoiser's 7680) runs as an in-LDS FFT, two rows per workgroup (fft_kernels.hip; 40-50 us per
// launch where the DFT-as-GEMM of rounds 4-6 took 576 us and two 236 MB bases); any other length keeps the fp32 GEMM on the f32 MFMA against
// precomputed bases (like the vocoder's head and the mel front edge).  The network between the two is lemas_mdx_* (engine_mdx.hip).
// Spectrogram layout of this interface: [batch][frames][ld] fp32, a frame = [re(0..nb-1) | im(0..nb-1) | padding], nb = n_fft/2+1,
// ld = lemas_stft_ld().
#include <cmath>
#include <vector>

#include "engine_common.h"

using namespace lemas;

struct lemas_stft {
  int nfft = 0, hop = 0, nb = 0, ld = 0;
  DevBuf window, fwd, inv;   // [nfft], forward basis [ld][nfft], inverse basis [nfft][ld] (window and 1/N folded in): GEMM form only
  DevBuf tw;                 // FFT form: exp(-2 pi i t / nfft), t < nfft
  FftPlan plan{};
  bool use_fft = false;
  DevBuf d_frames;
  ~lemas_stft() {
    for (DevBuf* b : {&window, &fwd, &inv, &tw, &d_frames}) b->release();
  }
};

extern "C" {

int lemas_stft_create(int32_t n_fft, int32_t hop_length, const float* window, lemas_stft** out) {
  if (!out || !window || n_fft <= 0 || (n_fft & 3) || hop_length <= 0 || hop_length > n_fft) { set_error("lemas_stft_create: bad arguments"); return LEMAS_E_ARG; }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    set_error("lemas_stft_create: no HIP device (this library has no CPU path)");
    return e != hipSuccess ? -(int)e : LEMAS_E_STATE;
  }
  lemas_stft* m = new lemas_stft();
  m->nfft = n_fft; m->hop = hop_length; m->nb = n_fft / 2 + 1; m->ld = (2 * m->nb + 3) & ~3;
  m->use_fft = fft_plan_make(n_fft, &m->plan);
  int rc = m->window.ensure((size_t)n_fft * 4);
  if (rc == 0 && hipMemcpy(m->window.p, window, (size_t)n_fft * 4, hipMemcpyHostToDevice) != hipSuccess) { set_error("lemas_stft_create: upload failed"); rc = LEMAS_E_STATE; }
  if (m->use_fft) {
    std::vector<float> t((size_t)2 * n_fft);
    for (int i = 0; i < n_fft; ++i) {
      const double a = -2.0 * M_PI * (double)i / (double)n_fft;
      t[2 * i] = (float)std::cos(a);
      t[2 * i + 1] = (float)std::sin(a);
    }
    if (rc == 0) rc = m->tw.ensure(t.size() * 4);
    if (rc == 0 && hipMemcpy(m->tw.p, t.data(), t.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { set_error("lemas_stft_create: upload failed"); rc = LEMAS_E_STATE; }
    if (rc == 0 && fft_kernels_init(m->plan) != hipSuccess) { set_error("lemas_stft_create: the FFT kernels' LDS opt-in failed"); rc = LEMAS_E_STATE; }
  } else {
    if (rc == 0) rc = m->fwd.ensure((size_t)m->ld * n_fft * 4);
    if (rc == 0) rc = m->inv.ensure((size_t)n_fft * m->ld * 4);
    if (rc == 0 && launch_rdft_basis(n_fft, m->ld, m->fwd.as<float>(), nullptr) != hipSuccess) { set_error("lemas_stft_create: basis kernel failed"); rc = LEMAS_E_STATE; }
    if (rc == 0 && launch_dft_basis(m->window.as<float>(), n_fft, m->ld, m->inv.as<float>(), nullptr) != hipSuccess) { set_error("lemas_stft_create: basis kernel failed"); rc = LEMAS_E_STATE; }
  }
  if (rc == 0 && hipStreamSynchronize(nullptr) != hipSuccess) { set_error("lemas_stft_create: device error"); rc = LEMAS_E_STATE; }
  if (rc != 0) { delete m; return rc; }
  *out = m;
  return 0;
}
void lemas_stft_destroy(lemas_stft* m) { delete m; }
int32_t lemas_stft_ld(const lemas_stft* m) { return m ? m->ld : -1; }
int64_t lemas_stft_frames(const lemas_stft* m, int64_t samples) { return (m && samples >= 0) ? samples / m->hop + 1 : -1; }

// wav device [batch][samples] -> spec device [batch][samples / hop + 1][ld]
int lemas_stft_forward(lemas_stft* m, const float* wav, int32_t batch, int32_t samples, float* spec, void* stream) {
  if (!m || !wav || !spec || batch <= 0 || samples <= m->nfft / 2) { set_error("lemas_stft_forward: bad arguments (reflect padding needs samples > n_fft / 2)"); return LEMAS_E_ARG; }
  hipStream_t s = (hipStream_t)stream;
  const int F = samples / m->hop + 1, rows = batch * F;
  RC_TRY(m->d_frames.ensure((size_t)rows * m->nfft * 4));
  HIP_TRY(launch_stft_frames(wav, m->window.as<float>(), batch, samples, F, m->nfft, m->hop, m->d_frames.as<float>(), s));
  if (m->use_fft) {
    HIP_TRY(launch_rfft_rows(m->plan, m->tw.as<float>(), m->d_frames.as<float>(), rows, spec, m->ld, s));
    return 0;
  }
  GemmF32Params g{};
  g.A = m->d_frames.as<float>(); g.lda = m->nfft; g.W = m->fwd.as<float>(); g.ldw = m->nfft; g.out = spec; g.ldc = m->ld;
  g.M = rows; g.N = m->ld; g.K = m->nfft;
  HIP_TRY(launch_gemm_f32(F32_BIAS, g, s));
  return 0;
}

// spec device [batch][frames][ld] -> wav device [batch][hop * (frames - 1)]
int lemas_stft_inverse(lemas_stft* m, const float* spec, int32_t batch, int32_t frames, float* wav, void* stream) {
  if (!m || !spec || !wav || batch <= 0 || frames < 2) { set_error("lemas_stft_inverse: bad arguments"); return LEMAS_E_ARG; }
  if ((long long)m->hop * (frames - 1) < m->nfft / 2) { set_error("lemas_stft_inverse: too few frames for the centre trim"); return LEMAS_E_ARG; }
  hipStream_t s = (hipStream_t)stream;
  const int rows = batch * frames;
  RC_TRY(m->d_frames.ensure((size_t)rows * m->nfft * 4));
  if (m->use_fft) {
    HIP_TRY(launch_irfft_rows(m->plan, m->tw.as<float>(), spec, m->ld, rows, m->window.as<float>(), m->d_frames.as<float>(), s));
    HIP_TRY(launch_overlap_add(m->d_frames.as<float>(), m->window.as<float>(), batch, frames, m->nfft, m->hop, wav, s));
    return 0;
  }
  GemmF32Params g{};
  g.A = spec; g.lda = m->ld; g.W = m->inv.as<float>(); g.ldw = m->ld; g.out = m->d_frames.as<float>(); g.ldc = m->nfft;
  g.M = rows; g.N = m->nfft; g.K = m->ld;
  HIP_TRY(launch_gemm_f32(F32_BIAS, g, s));
  HIP_TRY(launch_overlap_add(m->d_frames.as<float>(), m->window.as<float>(), batch, frames, m->nfft, m->hop, wav, s));
  return 0;
}

}  // extern "C"
