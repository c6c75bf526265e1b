/* liblemas_hip.so -- C ABI of the MI355X-native LEMAS-TTS acoustic-generation hot path.
 *
 * The reference (LEMAS-Project/LEMAS-TTS) is pure Python with no FFI/plugin/operator interface for this path
 * (SURVEY.md section 8b): its boundary is the Python call surface.  This header therefore declares the entry
 * points a maintainer of the reference would bind (ctypes stub in INTEGRATION.md) to replace, one for one:
 *
 *   lemas_dit_create/load_weight/finalize   <- lemas_tts/infer/utils_infer.py:252-303 load_model,
 *                                              :204-246 load_checkpoint (same key names, strict)
 *   lemas_dit_sample                        <- lemas_tts/model/cfm.py:206-473 CFM.sample, i.e. the odeint loop
 *                                              (:456) over fn (:382-425) = 2 x DiT.forward
 *                                              (lemas_tts/model/backbones/dit.py:194-254) + CFG + clamp
 *   lemas_dit_forward                       <- lemas_tts/model/backbones/dit.py:194-254 DiT.forward (both CFG branches)
 *   lemas_vocos_create/load_weight/finalize <- lemas_tts/infer/utils_infer.py:120-143 load_vocoder
 *   lemas_vocos_decode                      <- vocoder.decode call, lemas_tts/infer/utils_infer.py:549 and
 *                                              lemas_tts/scripts/speech_edit_multilingual.py:198
 *   lemas_mel_create/forward                <- lemas_tts/model/modules.py:104-143 MelSpec.forward (cfm.py:232-236)
 *   lemas_resample_create/forward           <- torchaudio Resample call, lemas_tts/infer/utils_infer.py:494-496
 *   lemas_prosody_*                         <- lemas_tts/model/backbones/prosody_encoder.py ProsodyEncoder / extract_fbank_16k,
 *                                              called per sample at lemas_tts/model/cfm.py:248-262
 *   lemas_stft_create/forward/inverse       <- uvr5/multiprocess_cuda_infer.py:206-223 Inference.stft / .istft (the torch.stft /
 *                                              torch.istft pair around the MDX-Net prompt denoiser; tts_multilingual.py:38-86)
 *   lemas_mdx_create/load_weight/finalize   <- uvr5/multiprocess_cuda_infer.py:225-238 Inference.load_model (the onnxruntime session built
 *                                              from Kim_Vocal_1.onnx, an export of uvr5/lib_v5/mdxnet.py:36-101 ConvTDFNet)
 *   lemas_mdx_forward                       <- uvr5/multiprocess_cuda_infer.py:262-272 model_run inside Inference.run_model, i.e.
 *                                              uvr5/lib_v5/mdxnet.py:103-127 ConvTDFNet.forward
 * Single-kernel entry points for the parity tests and micro-benchmarks (lemas_k_*) are declared in lemas_hip_test.h; they
 * live in a separate library, liblemas_hip_test.so, and are not part of the drop-in surface.
 *
 * Conventions: plain pointers and sizes only.  "device" pointers are HIP device addresses on the current device
 * (e.g. torch.Tensor.data_ptr()); "host" pointers are ordinary memory.  `stream` is a hipStream_t passed as void*
 * (NULL = default stream); all work is stream-ordered and the library never synchronises the device in the
 * sample/decode paths.  The caller owns every buffer it passes; the library owns its weights, workspaces and
 * captured hipGraphs.  Every function returns 0 on success or a negative code (-hipError_t for HIP failures,
 * LEMAS_E_* otherwise); lemas_last_error() returns a thread-local description.  An object is not thread-safe
 * (the reference's module is not re-entrant either: dit.py:140,191-192 text cache).
 */
#ifndef LEMAS_HIP_H
#define LEMAS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LEMAS_E_ARG (-10001)      /* bad argument / unsupported shape */
#define LEMAS_E_STATE (-10002)    /* call order (e.g. sample before finalize) */
#define LEMAS_E_WEIGHT (-10003)   /* unknown / missing / mis-shaped tensor (strict load) */

typedef struct lemas_dit lemas_dit;
typedef struct lemas_vocos lemas_vocos;
typedef struct lemas_mel lemas_mel;
typedef struct lemas_resample lemas_resample;
typedef struct lemas_prosody lemas_prosody;

/* model.arch of lemas_tts/configs/multilingual_grl.yaml:48-58 (+ derived sizes) */
typedef struct {
  int32_t dim, depth, heads, dim_head, ff_mult, text_dim, conv_layers, mel_dim;
  int32_t vocab_rows;       /* text embedding rows = vocab_size + 1 (dit.py:37) */
  int32_t conv_pos_kernel;  /* 31 */
  int32_t conv_pos_groups;  /* 16 */
  int32_t time_freq_dim;    /* 256 */
  int32_t has_prosody;      /* prosody_to_mel / prosody_text_proj present */
} lemas_dit_config;

/* ABI version 200 (lemas_version()): lemas_sample_args opens with its own size, checked by every entry point that takes it -- a client
 * compiled against another layout of the struct is refused (LEMAS_E_ARG) instead of being read past its end.  Initialise with
 * `lemas_sample_args a = {0}; a.struct_size = sizeof a;`. */
typedef struct {
  uint32_t struct_size; /* sizeof(lemas_sample_args) */
  int32_t batch;        /* B */
  int32_t frames;       /* N  = max duration (cfm.py:305) */
  int32_t cond_frames;  /* F  = frames of the reference mel before padding (cfm.py:240) */
  int32_t text_len;     /* Nt = padded token count */
  int32_t steps;        /* NFE */
  float cfg_strength;   /* < 1e-5 => single branch, no clamp (cfm.py:404-405) */
  const float* cond;          /* device [B,N,mel]  reference mel, zero right-padded to N (cfm.py:311) */
  const uint8_t* cond_mask;   /* device [B,N]      1 = conditioning frame (lens & edit mask, cfm.py:293-295,326) */
  const int64_t* text;        /* device [B,Nt]     token ids, -1 padded (model/utils.py:87-94) */
  const int32_t* seq_len;     /* device [B] valid frames per sample = duration (cfm.py:337), or NULL when B == 1 */
  const float* prosody;       /* device [B,512] utterance prosody embedding or NULL (cfm.py:261-263) */
  const float* t_grid;        /* HOST   [steps+1] ODE time grid (cfm.py:445-453) */
  float* y;                   /* device [B,N,mel]  in: y0 noise (cfm.py:430-435); out: trajectory[-1] */
  float* out;                 /* device [B,N,mel]  where(cond_mask, cond, y) (cfm.py:459-461); may be NULL */
  float* trajectory;          /* device [steps+1,B,N,mel] or NULL (callers discard it, utils_infer.py:543) */
  const float* step_cond;     /* device [B,N,mel] or NULL: what the flow is conditioned on where cond_mask is set, when that is
                               * not `cond` (+ prosody projection) itself -- the accent-GRL path conditions on cond_grl, built
                               * from the RAW prompt mel before the prosody projection / no_ref_audio substitution and
                               * optionally clipped-and-shuffled (cfm.py:266-283, 329-330, 387-388).  `out` still uses `cond`. */
  int32_t prosody_text_only;  /* != 0: `prosody` conditions the TEXT side only (dit.py:225-233) and `cond` is final as given.  The
                               * no_ref_audio path of the reference overwrites the prosody-shifted mel with its random conditioning
                               * (cfm.py:313-324), so the prosody-to-mel projection has no effect there. */
  int32_t cond_rows;          /* rows per sample present in `cond` / `step_cond`: [B,cond_rows,mel], 1 <= cond_rows <= N; rows beyond it
                               * read as zero, which is the reference's right-padding (cfm.py:311) done by the library instead of a
                               * caller-side pad copy.  0 = N (the buffers are already padded). */
  const float* y_init;        /* device [B,N,mel] or NULL.  When set, the ODE starts from y_init (left untouched) and `y` is output
                               * only; NULL = `y` is in/out as described above. */
} lemas_sample_args;

const char* lemas_last_error(void);
int lemas_version(void);

/* ---- DiT / CFM sampler ---- */
int lemas_dit_create(const lemas_dit_config* cfg, lemas_dit** out);
void lemas_dit_destroy(lemas_dit* m);
/* fp32 HOST tensor under its checkpoint key (prefix "ema_model." already stripped).  Auxiliary tables the
 * reference keeps as non-persistent buffers / recomputes are loaded the same way:
 *   "transformer.text_embed.freqs_cis" [4096, text_dim]  (modules.py:196-207)
 *   "transformer.time_embed.freqs"     [time_freq_dim/2] (modules.py:157-158) */
int lemas_dit_load_weight(lemas_dit* m, const char* name, const float* host_data, const int64_t* shape, int32_t ndim);
/* the same with the tensor already in DEVICE memory on the current device (fp32, contiguous; complete on entry): the form the
 * data-parallel launcher uses after the RCCL weight broadcast, so that non-source ranks never stage weights through the host
 * (precedent for per-GPU workers: uvr5/multiprocess_cuda_infer.py:404-420) */
int lemas_dit_load_weight_device(lemas_dit* m, const char* name, const float* device_data, const int64_t* shape, int32_t ndim);
int lemas_dit_finalize(lemas_dit* m);
/* options: "graph" (1 = replay one captured hipGraph per ODE step, default 1), "profile" (1 = per-kernel events),
 * "table_cache" (1 = keep the time/AdaLN tables while the t-grid is unchanged, default 1),
 * "dual" (1 = run the two CFG branches as concurrent lanes on two streams / graph branches, default 1),
 * "fp8" (1 = the DiT block GEMMs run on fp8-e4m3 MFMA: e4m3 weights with one fp32 scale per output channel, MXFP8
 *        activations (one E8M0 scale per 32 K); attention, norms, residual stream and ODE state unchanged; default 0;
 *        takes effect at the next prepare()/sample().  2 = accuracy point, not a speed path: the same e4m3 weights with bf16
 *        ACTIVATIONS (weights-only fp8), computed by the bf16 kernels on the dequantised weights),
 * "attn_f8qk" (what attention does WHILE the block GEMMs run on fp8 operands, i.e. under "fp8" = 1 with no tripped outlier guard: 1 (default) =
 *        the fp8 QK GEMM's epilogue writes the rotated q (prescaled by softmax_scale * log2 e) and k as MXFP8, one E8M0 scale per 32-wide half of a
 *        head, and Q K^T runs on v_mfma_scale_f32_32x32x64_f8f6f4 (csrc/attention.hip, variant bit 8192); P . V stays bf16.  0 = bf16 q / k as on
 *        the bf16 path.  2 = as 1 with a side launch quantising bf16 rows (the same bytes: A/B form).  + 4 = also on the bf16 path (a measurement:
 *        BASELINE's bf16 configurations never use it).  Needs attn_variant 17 or 19 (q prescaled by the QK epilogue), else it is inert),
 * "ln_fold" (1 = those LayerNorms folded ACROSS the GEMMs on either side -- the gate + residual epilogues write the scaled bf16 rows and
 *        per-row partial sums, the QKV / FF1 epilogues apply the row statistics, c1 / c2 rows per ODE step in the AdaLN table: no
 *        LayerNorm launch after a step's first; bf16 activations only; a different rounding of the same arithmetic, inside the
 *        sampler's tolerance; default 0: measured neutral to 2 % slower),
 * measurement options (0 = the production choice; each drops the cached graphs): "tile_n1024", "tile_n2048", "tile_qkv" = explicit
 *        GEMM tile ids (include/lemas_hip_test.h) for the block GEMMs of that width / the fused QK+V launch, "xcd_gx" = XCD block
 *        grid of the tile order (8, 4, 2, 1), "attn_variant" = schedule variant of the attention kernel (csrc/attention.hip; default
 *        19, 0 = classical online softmax).  Per engine: there is no process-global dispatch switch.
 * NOT in this library: the experiments that were measured, lost and are kept reproducible -- "ln_fused" (LayerNorm as the tail of the gate +
 *        residual GEMM launch), "lane_skew" (the CFG lanes one stage apart), "xcd_runs" (round 3's tile order) and the 64-queries-per-wave
 *        attention kernel.  The first three exist in builds with -DLEMAS_MEASUREMENT_BUILD only (this library refuses a non-zero value), the
 *        attention kernel in liblemas_hip_test.so only (include/lemas_hip_test.h: lemas_k_attention_variant, lemas_k_bench). */
int lemas_dit_set_option(lemas_dit* m, const char* key, int64_t value);
/* ragged batches (lemas_sample_args.seq_len set; cfm.py:336-339): "skip_masked" (default 1: the ATTENTION half of every block -- attn_norm, the
 * QK / V projections, attention, the out-projection -- skips the 128-row blocks that lie wholly in a sample's padding.  Exact: the reference zeroes
 * that half's output for rows past a sample's length (modules.py AttnProcessor) and nothing else reads what it computes for them; the output is
 * bit-identical, padding rows included.  0 = compute them, for A/B runs), "skip_dead" (what the FF half does with those blocks: 0 = computes them
 * (default, the reference's arithmetic: its unmasked position-embedding conv, dit.py:98, lets a sample's last ~30 frames see the padding rows
 * behind it); 2 = skips all but ONE block behind every sample, results at the reference's own error level; 1 = skips them all, a sample's last ~30
 * frames then differ from the reference's by 1e-5 instead of 2e-6 mel-MSE).  bf16 chain only.
 * further options: "graph_cache" (step-graph buckets kept, least recently used evicted; default 16), "graph_update" (1 = a new
 * frame count inside a cached bucket -- same batch, same 128-row pitch -- patches one of the bucket's instantiated graphs with
 * hipGraphExecUpdate instead of instantiating another; default 1), "fp8_outlier_guard" (default 1: with option "fp8" = 1, a checkpoint
 * whose residual-writing projections (attn.to_out, ff.2) show outlier output channels -- per-channel weight scale > 8x the median --
 * runs every block GEMM on its bf16 operands after all: such channels make the solve five to seven times more sensitive to e4m3 noise at
 * EVERY GEMM site (2.9e-4 mel-MSE at full depth / NFE 32 against the 1e-4 target; no single site stays under it with margin), so fp8 does
 * not ship for such weights.  0 = run fp8 regardless; round 5's mixed-precision decomposition, option "fp8_outlier_mode", met the target
 * but ran slower than this fallback and lives in measurement builds only since round 6), "fp8_sites" (mask of the GEMM sites of a block that take fp8 operands when
 * "fp8" = 1: 1 QKV, 2 out-projection, 4 FF1, 8 FF2; default 15).
 * counters since creation, for tools and tests: "graph_captures", "graph_instantiates", "graph_updates", "graph_update_failures",
 * "graph_evictions", "graph_buckets", "fp8_outlier_channels" (residual channels flagged by the guard; -1 before the first fp8 prepare),
 * "fp8_gemms_kept_bf16" (GEMMs per DiT block that run on bf16 operands while "fp8" = 1: 0; 4 when the guard tripped) */
int lemas_dit_get_stat(lemas_dit* m, const char* key, int64_t* value);
/* synchronises the device and returns 0, or LEMAS_E_STATE when a device-side wait of this engine gave up (fused LayerNorm tail):
 * every result since then is invalid and prepare() / solve() refuse to run */
int lemas_dit_health(lemas_dit* m);
/* full sampler: hoists + NFE Euler steps (+ final where) */
int lemas_dit_sample(lemas_dit* m, const lemas_sample_args* a, void* stream);
/* split form: prepare() runs the per-utterance hoists, solve() the step loop on the prepared state */
int lemas_dit_prepare(lemas_dit* m, const lemas_sample_args* a, void* stream);
int lemas_dit_solve(lemas_dit* m, const lemas_sample_args* a, void* stream);
/* one DiT forward at step index k of the prepared grid: x device [B,N,mel] -> pred device [BB*N, mel]
 * (BB = 2B with CFG: conditional rows first, then unconditional) */
int lemas_dit_forward(lemas_dit* m, const float* x, int32_t step_index, float* pred, void* stream);
/* per-kernel-class timings collected while option "profile" = 1: fills up to `cap` entries, returns the count.
 * Synchronises the device. */
int lemas_dit_profile_read(lemas_dit* m, char (*names)[32], double* total_ms, int64_t* launches, int32_t cap);

/* ---- Vocos vocoder ---- */
int lemas_vocos_create(int32_t input_channels, int32_t dim, int32_t intermediate_dim, int32_t num_layers, int32_t n_fft,
                       int32_t hop_length, lemas_vocos** out);
void lemas_vocos_destroy(lemas_vocos* v);
int lemas_vocos_load_weight(lemas_vocos* v, const char* name, const float* host_data, const int64_t* shape, int32_t ndim);
int lemas_vocos_load_weight_device(lemas_vocos* v, const char* name, const float* device_data, const int64_t* shape, int32_t ndim);
int lemas_vocos_finalize(lemas_vocos* v);
/* mel device [B, C, L] fp32 -> wav device [B, hop*(L-1)] fp32; `gain` multiplies the waveform (rms rescale,
 * utils_infer.py:552-553; pass 1.0 for none) */
int lemas_vocos_decode(lemas_vocos* v, const float* mel, int32_t batch, int32_t frames, float gain, float* wav, void* stream);
/* the same on the FRAMES-FIRST layout the sampler produces: mel_rows device [B][frames][C] fp32, sample b starting at mel_rows + b *
 * batch_stride (elements) -- e.g. `out + (F - 1) * C` with batch_stride N * C decodes frames F-1.. of lemas_dit_sample's `out` in
 * place, which is what the reference's `vocoder.decode(generated[:, F-1:, :].permute(0, 2, 1))` computes after its permute copy
 * (lemas_tts/infer/utils_infer.py:546-549) */
int lemas_vocos_decode_rows(lemas_vocos* v, const float* mel_rows, int32_t batch, int32_t frames, int64_t batch_stride, float gain, float* wav,
                            void* stream);
/* options: "graph" (1 = the backbone + head of a decode shape seen twice is replayed as one hipGraph, default 1), "graph_cache" (decode
 * shapes kept, least recently used evicted; default 8) */
int lemas_vocos_set_option(lemas_vocos* v, const char* key, int64_t value);

/* ---- reference wav -> log-mel front edge (lemas_tts/model/modules.py:75-143 MelSpec, call site cfm.py:232-236) ----
 * wav device [B, samples] fp32 at `sample_rate` -> mel device [B, samples/hop + 1, n_mels] fp32 (already in the
 * [b, frames, channels] layout CFM.sample uses after its permute, cfm.py:235) */
int lemas_mel_create(int32_t n_fft, int32_t hop_length, int32_t n_mels, int32_t sample_rate, lemas_mel** out);
void lemas_mel_destroy(lemas_mel* m);
int lemas_mel_forward(lemas_mel* m, const float* wav, int32_t batch, int32_t samples, float* mel, void* stream);

/* ---- prompt resampling to the model rate (torchaudio.transforms.Resample(sr, 24000) at lemas_tts/infer/utils_infer.py:494-496
 * and lemas_tts/model/cfm.py:254): sinc_interp_hann polyphase filter, lowpass_filter_width 6, rolloff 0.99.
 * wav device [B, samples] fp32 -> out device [B, lemas_resample_out_len(samples)] fp32 ---- */
int lemas_resample_create(int32_t orig_freq, int32_t new_freq, lemas_resample** out);
void lemas_resample_destroy(lemas_resample* r);
int64_t lemas_resample_out_len(const lemas_resample* r, int64_t samples);   /* ceil(new * samples / orig) */
int lemas_resample_forward(lemas_resample* r, const float* wav, int32_t batch, int32_t samples, float* out, void* stream);

/* ---- STFT / inverse STFT around the UVR5 MDX-Net prompt denoiser (uvr5/multiprocess_cuda_infer.py:206-223): torch.stft(n_fft, hop,
 * window, center=True, onesided) and torch.istft(..., center=True): an in-LDS fp32 FFT for n_fft = 2^a 3^b 5^c <= 8192 (the denoiser's 7680),
 * fp32 GEMMs against DFT bases for any other length.  `window` host [n_fft] (the reference passes
 * hann_window(n_fft, periodic=False)).  Spectrogram frames are [re(0..n_fft/2) | im(0..n_fft/2) | padding], ld = lemas_stft_ld().
 *   forward: wav device [B, samples] -> spec device [B, samples / hop + 1, ld]
 *   inverse: spec device [B, frames, ld] -> wav device [B, hop * (frames - 1)]  (imaginary parts of bins 0 and n_fft/2 are ignored) ---- */
typedef struct lemas_stft lemas_stft;
int lemas_stft_create(int32_t n_fft, int32_t hop_length, const float* window, lemas_stft** out);
void lemas_stft_destroy(lemas_stft* m);
int32_t lemas_stft_ld(const lemas_stft* m);
int64_t lemas_stft_frames(const lemas_stft* m, int64_t samples);
int lemas_stft_forward(lemas_stft* m, const float* wav, int32_t batch, int32_t samples, float* spec, void* stream);
int lemas_stft_inverse(lemas_stft* m, const float* spec, int32_t batch, int32_t frames, float* wav, void* stream);

/* ---- the MDX-Net separation network of the UVR5 prompt denoiser: ConvTDFNet (uvr5/lib_v5/mdxnet.py:36-127, blocks
 * uvr5/lib_v5/modules.py:5-74), which the reference runs as an onnxruntime session (uvr5/multiprocess_cuda_infer.py:225-238,262-272).
 * Configuration = the constructor arguments of ConvTDFNet that shape the network (mdxnet.py:37-49).  Weight names = the state-dict keys of
 * that module ("first_conv.0.weight", "encoding_blocks.0.tfc.H.0.1.running_var", "ds.0.0.weight", "us.0.0.weight", "final_conv.0.bias"
 * ...), strict: unknown names are refused, finalize() names the first missing one; `window`, `freq_pad` and `*.num_batches_tracked` are
 * accepted and ignored (forward never reads them).  Inference only: BatchNorm uses its running statistics (eps 1e-5).
 *   forward: spek device [batch, dim_c, dim_f, dim_t] fp32 (frames innermost, what Inference.stft returns) -> out, same shape.
 * All arithmetic is fp32 (f32-input MFMA); workspaces grow to the largest batch seen (~0.75 GB per sample at the Kim_Vocal_1 shape). ---- */
typedef struct lemas_mdx lemas_mdx;
typedef struct lemas_mdx_config {
  int32_t dim_c;        /* 4: (left re, left im, right re, right im) */
  int32_t dim_f;        /* frequency bins kept (3072); must be divisible by 2^(num_blocks/2), and by 4 after that division if bn >= 0 */
  int32_t dim_t;        /* frames per chunk (256); divisible by 2^(num_blocks/2) */
  int32_t num_blocks;   /* 11: n = num_blocks / 2 encoder and decoder stages around the bottleneck */
  int32_t l;            /* convolutions per TFC (3) */
  int32_t g;            /* channel growth per stage (48) */
  int32_t k;            /* TFC kernel size; only 3 is built */
  int32_t bn;           /* TDF bottleneck factor: f -> f / bn -> f; 0: a single Linear(f, f); -1: no TDF branch (the reference's None) */
  int32_t bias;         /* TDF linears carry a bias */
  int32_t norm;         /* 0: BatchNorm2d (optimizer 'rmsprop', folded into the convolutions); 1: GroupNorm(2, c) (optimizer 'adamw') */
} lemas_mdx_config;
int lemas_mdx_create(const lemas_mdx_config* cfg, lemas_mdx** out);
void lemas_mdx_destroy(lemas_mdx* m);
int lemas_mdx_load_weight(lemas_mdx* m, const char* name, const float* host_data, const int64_t* shape, int32_t ndim);
/* options.  BEFORE finalize(): "bf16x3" (default 0 = every product exact fp32 on the f32-input MFMA; 1 = the 3x3 convolutions -- 77 % of the
 * FLOPs -- on split-bf16 operands: x = hi + lo, three bf16 MFMAs per product, ~2^-16 relative precision per product with fp32 accumulation:
 * rms error 2e-5 of the output's rms against 2e-6 for the exact path, 1.45x the speed; BatchNorm variant only, the GroupNorm variant ignores
 * it).  Any time, process-wide: "conv_chunk" (4, default, or 8 input channels per K chunk of the exact 3x3 kernel), "bf16x3_products" (3,
 * default, or 4 bf16 MFMAs per product of the split kernel: measured 9 % slower for 10 % less error). */
int lemas_mdx_set_option(lemas_mdx* m, const char* key, int64_t value);
int lemas_mdx_finalize(lemas_mdx* m);
int lemas_mdx_forward(lemas_mdx* m, const float* spek, int32_t batch, float* out, void* stream);
/* Verification hook: the next forwards also copy the activation after stage `name` ("first", "enc<i>", "ds<i>", "bottleneck", "us<i>",
 * "dec<i>"; [batch, c, t, f]) to `dev_out` (device, caller-sized); NULL removes the tap. */
int lemas_mdx_tap(lemas_mdx* m, const char* name, float* dev_out);
/* multiply-add FLOPs (2 per MAC) of one forward at `batch`: the algorithmic work of the roofline line */
int64_t lemas_mdx_flops(const lemas_mdx* m, int32_t batch);

/* ---- prosody encoder (ECAPA-TDNN), the prompt's global prosody embedding ----
 * Architecture numbers = the reference's pretssel_cfg.json "model.prosody_*" keys (prosody_encoder.py:390-403). */
typedef struct lemas_prosody_config {
  int32_t n_layers;            /* len(prosody_channels) */
  int32_t channels[8];
  int32_t kernel_sizes[8];
  int32_t dilations[8];
  int32_t groups[8];           /* must be 1 */
  int32_t attention_channels, res2net_scale, se_channels, global_context, embed_dim, input_dim;
} lemas_prosody_config;
int lemas_prosody_create(const lemas_prosody_config* cfg, lemas_prosody** out);
void lemas_prosody_destroy(lemas_prosody* p);
/* weight names as in the reference module's state dict after its loader strips "prosody_encoder." (prosody_encoder.py:406-424) */
int lemas_prosody_load_weight(lemas_prosody* p, const char* name, const float* host_data, const int64_t* shape, int32_t ndim);
int lemas_prosody_finalize(lemas_prosody* p);
/* kaldi fbank of extract_fbank_16k (prosody_encoder.py:334-361): wav16k device [samples >= 400] -> fbank device [frames, 80] */
int64_t lemas_prosody_fbank_frames(int64_t samples_16k);      /* 1 + (samples - 400) / 160, 0 if samples < 400 */
int lemas_prosody_fbank(lemas_prosody* p, const float* wav16k, int32_t samples, float* fbank, void* stream);
/* one sample, padding_mask=None (how cfm.py:259 calls it): fbank device [frames, input_dim] -> emb device [embed_dim], L2-normalised */
int lemas_prosody_encode(lemas_prosody* p, const float* fbank, int32_t frames, float* emb, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LEMAS_HIP_H */
