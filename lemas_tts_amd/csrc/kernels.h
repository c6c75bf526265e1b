// Internal launcher prototypes for the elementwise / setup / vocoder kernels (see norm_elementwise.hip, aux_kernels.hip).
#pragma once
#include "common.h"

hipError_t launch_step_set(int* step_idx, int value, hipStream_t s);
hipError_t launch_rope_table(float* cs, float* sn, int n, int half, const float* inv_freq, hipStream_t s);
hipError_t launch_f32_to_bf16(const float* src, bf16_t* dst, size_t n, hipStream_t s);
hipError_t launch_convpos_weight(const float* src, bf16_t* dst, int C, int cg, int taps, hipStream_t s);
hipError_t launch_select_rows(float* out, const float* a, const float* b_padded, const uint8_t* mask, int B, int N, int pitch, int cols, hipStream_t s);

hipError_t launch_text_gather(const int64_t* text, int B, int Nt, int N, int td, int branches, const float* table, int vocab_rows,
                              const float* freqs_cis, int max_pos, float* out, uint8_t* rowmask, hipStream_t s);
hipError_t launch_dwconv7(const float* x, const float* w, const float* bias, float* out, int B, int N, int C, hipStream_t s);
hipError_t launch_ln_affine(const float* x, const float* w, const float* b, float* out, int M, int D, hipStream_t s);
hipError_t launch_grn(float* x, float* gx_scratch, const float* gamma, const float* beta, int B, int N, int C, hipStream_t s);
hipError_t launch_add_rowvec(float* x, const float* vec, int BB, int B, int N, int C, int nlim, hipStream_t s);
hipError_t launch_cond_prepare(const float* cond, const uint8_t* mask, const float* pm, const float* pbias, int B, int N,
                               int F, int md, int crows /* rows per sample present in cond, <= N */, float* cond_eff, float* step_cond,
                               hipStream_t s);
hipError_t launch_concat_ct(const float* step_cond, const float* te, int B, int N, int md, int td, int branches, int pitch,
                            float* ct, hipStream_t s);
hipError_t launch_time_sinus(const float* t, const float* freqs, int S, int half, float* out, hipStream_t s);
hipError_t launch_silu(const float* x, float* out, size_t n, hipStream_t s);
// mel element (b, ci, n) at mel[b * sb + ci * sc + n * sl] (strides in elements): [B, C, L] or a frames-first slice in place
hipError_t launch_im2col7(const float* mel, int B, int C, int L, long sb, long sc, long sl, float* col, hipStream_t s);
// dwconv7 + affine LayerNorm of one ConvNeXt block in one launch (C == 512)
hipError_t launch_dwconv7_ln(const float* x, const float* w, const float* bias, const float* lnw, const float* lnb, float* out, int B, int N,
                             int C, hipStream_t s);
hipError_t launch_spec(const float* head, int rows, int nb, int ldh, int lds_, float* spec, hipStream_t s);
hipError_t launch_dft_basis(const float* window, int nfft, int ld, float* basis, hipStream_t s);
hipError_t launch_overlap_add(const float* frames, const float* window, int B, int L, int nfft, int hop, float* wav, hipStream_t s);
hipError_t launch_scale(float* x, float sc, size_t n, hipStream_t s);

// wav -> log-mel front edge
hipError_t launch_stft_frames(const float* wav, const float* window, int B, int nw, int F, int nfft, int hop, float* frames, hipStream_t s);
hipError_t launch_rdft_basis(int nfft, int rows, float* basis, hipStream_t s);
hipError_t launch_magnitude(const float* spec, int rows, int nb, int lds_, int ldm, float* mag, hipStream_t s);
hipError_t launch_log_clamp(float* x, size_t n, float lo, hipStream_t s);

// prosody encoder (ECAPA-TDNN, prosody_kernels.hip) -- activations are time-major [T][ld] fp32
hipError_t launch_im2col_dil(const float* x, int ldx, const float* add, int ldadd, int T, int C, int k, int dil, float* col, hipStream_t s);
hipError_t launch_ln_rows(const float* x, int ldx, int T, int C, const float* w, const float* b, float eps, int act_tanh, float* out, int ldo, hipStream_t s);
hipError_t launch_copy_cols(const float* x, int ldx, int T, int C, float* out, int ldo, hipStream_t s);
// one Res2Net chunk (64 -> 64 channels, 3 taps) as one launch: y = LayerNorm(relu(conv_dil(x + add) + bias)); Wimg = the lane-major image
// launch_res2net_weight_image makes of the reference's [out][in][k] weight (res2net_weight_image_floats() floats)
bool res2net_step_fits(int cin, int cout, int k, int dil);
size_t res2net_weight_image_floats();
hipError_t launch_res2net_weight_image(const float* W, float* img, hipStream_t s);
hipError_t launch_res2net_step(const float* x, int ldx, const float* add, int ldadd, int T, int dil, const float* Wimg, const float* bias,
                               const float* lnw, const float* lnb, float eps, float* out, int ldo, hipStream_t s);
// one-row Linear (M == 1; F32_BIAS / F32_BIAS_RELU / F32_BIAS_SIGMOID; A, W rows on 16 B): a wave per output instead of a GEMM tile
hipError_t launch_gemv_f32(int epi, const GemmF32Params& p, hipStream_t s);
hipError_t launch_col_stats(const float* x, int ldx, int T, int C, float eps, float* mean, float* stdv, hipStream_t s);
hipError_t launch_scale_cols_add(const float* x, int ldx, const float* scale, const float* res, int ldr, int T, int C, float* out, int ldo, hipStream_t s);
hipError_t launch_softmax_pool(const float* att, int lda, const float* x, int ldx, int T, int C, float eps, float* mean, float* stdv, hipStream_t s);
hipError_t launch_l2_normalize(const float* x, int n, float eps, float* out, hipStream_t s);
hipError_t launch_kaldi_frames(const float* wav, int n, int frames, int win, int shift, int padded, float preemph, float* out, hipStream_t s);
hipError_t launch_power(const float* spec, int rows, int nb, int lds_, int ldp, float* pw, hipStream_t s);
// real FFT / inverse real FFT of STFT rows in LDS (fft_kernels.hip): N = 2^a 3^b 5^c, even, <= 8192; `tw` = N float2 exp(-2 pi i t / N)
struct FftPlan { int n; int nrad; int rad[14]; };
bool fft_plan_make(int n, FftPlan* plan);
hipError_t fft_kernels_init(const FftPlan& plan);
hipError_t launch_rfft_rows(const FftPlan& plan, const float* tw, const float* frames, int rows, float* spec, int ld, hipStream_t s);
hipError_t launch_irfft_rows(const FftPlan& plan, const float* tw, const float* spec, int ld, int rows, const float* window, float* frames, hipStream_t s);
