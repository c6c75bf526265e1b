/* liblemas_hip_test.so -- test and measurement entry points (lemas_k_*).
 *
 * NOT part of the drop-in surface (include/lemas_hip.h) and NOT in the product library: liblemas_hip_test.so is engine_ktests.hip
 * linked against liblemas_hip.so (one copy of every kernel and of the error state in the process).  Thin drivers that feed fp32 device arrays through ONE production
 * kernel so that the parity tests can localise a failure, plus a micro-benchmark.  They allocate scratch with hipMalloc
 * and synchronise the stream before returning.  Conventions as in lemas_hip.h: all pointers are device fp32 unless noted,
 * `stream` is a hipStream_t, return 0 or a negative code.
 */
#ifndef LEMAS_HIP_TEST_H
#define LEMAS_HIP_TEST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- single-kernel entry points (parity tests) ---- all pointers device fp32 unless noted */
/* out[M,N] = act(A[M,K] . W[N,K]^T + bias) through the bf16 MFMA GEMM (inputs rounded to bf16); act: 0 none, 1 gelu-tanh.
 * K % 64 == 0; N % 4 == 0 (act 0) / N % 8 == 0 (act 1): the epilogues store whole 16-byte chunks; anything else is refused. */
int lemas_k_linear_bf16(const float* A, const float* W, const float* bias, float* out, int32_t M, int32_t N, int32_t K,
                        int32_t act, void* stream);
/* out[M,N] = A . W^T + bias through the exact-fp32 MFMA GEMM; act: 0 none, 1 gelu-erf, 2 silu */
int lemas_k_linear_f32(const float* A, const float* W, const float* bias, float* out, int32_t M, int32_t N, int32_t K,
                       int32_t act, void* stream);
/* fp8 (MXFP8) path of the GEMMs -- BASELINE config 5 "fp8 MFMA weights".  Activations: e4m3 bytes + one E8M0 scale per
 * 32 consecutive K (OCP MX); weights: e4m3 + one fp32 scale per output channel.  All pointers device. */
int lemas_k_mx_quant(const float* x, int32_t M, int32_t K, uint8_t* out8, uint8_t* mx, void* stream);
int lemas_k_w_quant_f8(const float* w, int32_t N, int32_t K, uint8_t* out8, float* scale, void* stream);
/* the side product of the fp8 path on outlier checkpoints (csrc/outlier_rows.hip): x[m][chan[j]] += gate[chan[j]] * (bf16(A)[m] . bf16(Wside)[j] + bias_side[j])
 * for j < nf <= 32 and every row m = (sample, position < min(frames, seq_len[sample])) of the [batch][pitch] row space; A [batch * pitch][K], Wside [nf][K],
 * gate [ldx], x [batch * pitch][ldx] fp32 in place; a8 / amx (optional): the MXFP8 image of bf16(A), [M][K] e4m3 + [M][K / 32] E8M0; K is 1024 or 2048 */
int lemas_k_outlier_rows(const float* A, const float* Wside, const float* bias_side, const int32_t* chan, int32_t nf, const float* gate,
                         const int32_t* seq_len, float* x, int32_t batch, int32_t frames, int32_t pitch, int32_t K, int32_t ldx, uint8_t* a8, uint8_t* amx,
                         void* stream);
int lemas_k_ln_mod_f8(const float* x, const float* scale, const float* shift, uint8_t* out8, uint8_t* mx, int32_t M, int32_t D,
                      void* stream);
/* out = act(MXFP8(A) . FP8(W)^T + bias); act 0 none (fp32 out), 1 GELU-tanh (bf16-rounded out), 2 GELU-tanh written as
 * MXFP8 into out8 [M,N] / outmx [M,N/32] (out unused) */
int lemas_k_linear_f8(const float* A, const float* W, const float* bias, float* out, int32_t M, int32_t N, int32_t K, int32_t act,
                      uint8_t* out8, uint8_t* outmx, void* stream);
/* q,k,v [B,H,N,64] (already rotated) -> out [B,N,H*64]; seq_len device int32 [B] or NULL (modules.py:483-491) */
int lemas_k_attention(const float* q, const float* k, const float* v, const int32_t* seq_len, float* out, int32_t B,
                      int32_t H, int32_t N, void* stream);
/* the same through schedule variant `variant` of the attention kernel (csrc/attention.hip; 0 = lemas_k_attention) */
int lemas_k_attention_variant(const float* q, const float* k, const float* v, const int32_t* seq_len, float* out, int32_t B,
                              int32_t H, int32_t N, int32_t variant, void* stream);
/* out = LayerNorm(x; eps 1e-6) * (1 + scale) + shift, rows of 1024; result rounded to bf16 then widened */
int lemas_k_ln_mod(const float* x, const float* scale, const float* shift, float* out, int32_t M, int32_t D, void* stream);
/* out = conv_pos_embed(x) + x for x [B,N,C]; w1,w2 [C, C/groups, taps], b1,b2 [C] */
int lemas_k_convpos(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* out,
                    int32_t B, int32_t N, int32_t C, int32_t groups, int32_t taps, void* stream);

/* one production bf16 GEMM launch with an explicit tile and epilogue in the engine's row space: rows = batch x pitch
 * (pitch % 128 == 0), `frames` valid rows per sample.  A [batch*pitch, K], W [N, K], bias [N] fp32 (rounded to bf16 inside).
 *   tile: 0 production choice | 16 = 256x128 | 17 = 128x128 | 18 = 128x64 | 19 = 64x64 | 22 = 256x256 | 26 = 128x128 (4 waves)
 *         | bf16 only, the lock-step tiles with four LOADER waves behind the compute waves (bit-identical to the plain form): 27 = 17 + loaders,
 *           28 = 26 + loaders, 29 = 28 with a 4-stage ring (measurement), 30 = 18 + loaders, 31 = 19 + loaders
 *   epi 0: out[M,N] = bf16(acc + bias)        1: out = bf16(gelu_tanh(acc + bias))       2: out = acc + bias (fp32)
 *   epi 3: out[M,N] += aux[n] * (acc + bias) for rows with pos < frames (and pos < seq_len[b] when given); aux = gate [N]
 *   epi 4: N = 2*H*64: +bias, RoPE with aux = [cos | sin] ([frames][32] each) -> out = q then k, each [batch][H][pitch][64]
 *   epi 5: N = H*64:   +bias -> out = v^T [batch][H][64][pitch]
 * (modules.py:452-461,470-480,495,635,349-350) */
int lemas_k_gemm_epi(int32_t epi, int32_t tile, const float* A, const float* W, const float* bias, const float* aux,
                     const int32_t* seq_len, float* out, int32_t batch, int32_t pitch, int32_t frames, int32_t N, int32_t K, void* stream);

/* the gate + residual GEMM (epi 3 above, N = 1024) WITH its LayerNorm-modulate tail, as the sampler launches it (gemm_bf16.hip ln_tail):
 * x [batch*pitch, 1024] in/out = residual stream; h [batch*pitch, 1024] out = bf16(LN(x_new; eps 1e-6) * (1 + scale) + shift) widened;
 * gate / scale / shift [1024].  tile: 17 / 18 / 19 / 26 (the tiles that carry the tail; 0 = production choice).  concurrent (1..4):
 * that many independent copies of the launch run at once on separate streams (the two CFG lanes' situation); their results must
 * be bit-identical, copy 0 is returned.  (modules.py:635-637) */
int lemas_k_gemm_gate_ln(int32_t tile, const float* A, const float* W, const float* bias, const float* gate, const float* scale,
                         const float* shift, const int32_t* seq_len, float* x, float* h, int32_t batch, int32_t pitch, int32_t frames,
                         int32_t K, int32_t concurrent, void* stream);

/* "ln fold": the AdaLN-modulated LayerNorm between a gated residual update and the GEMM behind it, folded across the two GEMMs as the
 * sampler runs it (csrc/common.h GemmParams).  producer = the gate + residual GEMM (A [M,Kp], Wp [1024,Kp], bias_p, gate; tile prod_tile:
 * 16 / 17 / 18 / 19 / 22 / 26, 0 = production choice) updating x [M,1024] in place -- or, with use_prep != 0, the chain-entry kernel on x
 * as given; then the c1 / c2 rows through the production table builders; then the consumer GEMM on Wc [Nc,1024], bias_c with epilogue
 * cons_epi (1: y [M,Nc] = gelu_tanh(.); 4: y = q then k [batch][H][pitch][64] with rope = [cos | sin]; 5: y = v^T [batch][H][64][pitch])
 * on tile cons_tile.  Mathematically y = epi(LN(x_new; eps 1e-6) * (1 + scale) + shift) . Wc^T + bias_c).  (modules.py:627-641) */
int lemas_k_ln_fold_pair(int32_t prod_tile, int32_t cons_epi, int32_t cons_tile, const float* A, const float* Wp, const float* bias_p,
                         const float* gate, const float* scale, const float* shift, const float* Wc, const float* bias_c, const float* rope,
                         const int32_t* seq_len, float* x, float* y, int32_t batch, int32_t pitch, int32_t frames, int32_t Kp, int32_t Nc,
                         int32_t use_prep, void* stream);

/* Measurement builds only (-DLEMAS_PHASE_TIMESTAMPS, tools/timeline_step.py): every block GEMM and attention launch the engines enqueue from now
 * on stamps its workgroups' start / end (100 MHz wall clock) into slot k of `buf` (u64 [slots][4096]: [workgroup][4]), k counting the
 * launches of one forward pass in enqueue order; buf = NULL switches it off.  In a product build this returns LEMAS_E_STATE. */
int lemas_k_timeline(void* buf, int32_t slots);
/* how both libraries were built: bit 0 = -DLEMAS_MEASUREMENT_BUILD (the kept-reproducible experiments exist: engine options "ln_fused", "lane_skew",
 * "xcd_runs", lemas_k_gemm_gate_ln), bit 1 = -DLEMAS_PHASE_TIMESTAMPS (lemas_k_timeline, phase stamps).  0 = the product build. */
int lemas_k_build_flags(void);

/* micro-benchmark of one step-loop kernel on synthetic operands: what = "gemm_gelu" | "gemm_gate" | "gemm_qk" | "gemm_v" |
 * "gemm_f32out" | "gemm_gate_ln" (gemm_gate with its LayerNorm tail, N = 1024) (M,N,K = GEMM shape; prefix "f8_" for the MXFP8 path) or "attention" (M = frames, N = batch*heads); returns
 * the average launch duration in microseconds over `iters` back-to-back launches (HIP events).  `variant` = GEMM tile as
 * in lemas_k_gemm_epi (0 = production choice). */
int lemas_k_bench(const char* what, int32_t M, int32_t N, int32_t K, int32_t iters, int32_t variant, double* avg_us);


#ifdef __cplusplus
}
#endif
#endif /* LEMAS_HIP_TEST_H */
