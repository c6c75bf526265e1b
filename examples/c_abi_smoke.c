/* Plain-C client of liblemas_hip.so: proves the boundary is a C ABI (no C++/torch types cross it).
 *
 *   gcc -O2 -I include examples/c_abi_smoke.c -o examples/c_abi_smoke -L lemas_tts_amd/lib -llemas_hip \
 *       -Wl,-rpath,$PWD/lemas_tts_amd/lib -L/opt/rocm/lib -lamdhip64 -lm
 *
 * It runs the Vocos vocoder (lemas_vocos_*) on a synthetic mel with synthetic weights and checks an invariant that
 * needs no oracle: decode is linear in `gain`, deterministic, finite, and has the right length (256 (L-1) samples).
 * Then the sampler (lemas_dit_*): a depth-1 DiT of the shipped width is created, every tensor of the reference's checkpoint
 * layout is loaded under its key (utils_infer.py:223-237), the weights are finalized and lemas_dit_sample runs 3 Euler steps
 * with CFG on a 100-frame utterance; checked without an oracle: finite, repeatable bit for bit, the conditioning frames of
 * `out` are the prompt itself (cfm.py:459-461), the generated frames are not, and a wrong argument is refused with a message.
 * Last the STFT pair of the prompt denoiser's shell (lemas_stft_*): inverse(forward(x)) = x for a Hann window at hop = n_fft / 2,
 * Parseval-sized spectra, right frame count, and a too-short signal is refused.
 * Device memory comes from the HIP runtime's C API.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lemas_hip.h"

/* minimal HIP runtime C prototypes (avoid hip headers: this file is compiled by gcc) */
extern int hipMalloc(void** p, size_t n);
extern int hipFree(void* p);
extern int hipMemcpy(void* dst, const void* src, size_t n, int kind);
extern int hipDeviceSynchronize(void);
enum { H2D = 1, D2H = 2 };

static uint32_t rng = 12345u;
static float frand(void) { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) / 8388608.0f - 1.0f); }

static int load(lemas_vocos* v, const char* name, int nd, int64_t d0, int64_t d1, int64_t d2, float scale, float offset) {
  int64_t shape[3] = {d0, d1, d2};
  size_t n = (size_t)d0 * (nd > 1 ? d1 : 1) * (nd > 2 ? d2 : 1);
  float* h = (float*)malloc(n * sizeof(float));
  for (size_t i = 0; i < n; ++i) h[i] = offset + scale * frand();
  if (!strcmp(name, "head.istft.window"))
    for (size_t i = 0; i < n; ++i) h[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)i / (double)n));
  int rc = lemas_vocos_load_weight(v, name, h, shape, nd);
  free(h);
  if (rc) fprintf(stderr, "load %s: %s\n", name, lemas_last_error());
  return rc;
}

/* ---- the sampler through the C ABI ------------------------------------------------------------------------------------ */
static int dload(lemas_dit* m, const char* name, int nd, int64_t d0, int64_t d1, int64_t d2, float scale, float offset) {
  int64_t shape[3] = {d0, d1, d2};
  size_t n = (size_t)d0 * (nd > 1 ? d1 : 1) * (nd > 2 ? d2 : 1);
  float* h = (float*)malloc(n * sizeof(float));
  for (size_t i = 0; i < n; ++i) h[i] = offset + scale * frand();
  int rc = lemas_dit_load_weight(m, name, h, shape, nd);
  free(h);
  if (rc) fprintf(stderr, "load %s: %s\n", name, lemas_last_error());
  return rc;
}

static int dit_section(void) {
  const int D = 1024, TD = 512, MD = 100, V = 60, FD = 256;
  lemas_dit_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.dim = D; cfg.depth = 1; cfg.heads = 16; cfg.dim_head = 64; cfg.ff_mult = 2; cfg.text_dim = TD; cfg.conv_layers = 4; cfg.mel_dim = MD;
  cfg.vocab_rows = V + 1; cfg.conv_pos_kernel = 31; cfg.conv_pos_groups = 16; cfg.time_freq_dim = FD; cfg.has_prosody = 0;
  lemas_dit* m = NULL;
  if (lemas_dit_create(&cfg, &m)) { fprintf(stderr, "dit create: %s\n", lemas_last_error()); return 10; }
  char name[128];
  int rc = 0;
  const float w = 0.02f;
  rc |= dload(m, "transformer.time_embed.time_mlp.0.weight", 2, D, FD, 0, w, 0.f);
  rc |= dload(m, "transformer.time_embed.time_mlp.0.bias", 1, D, 0, 0, w, 0.f);
  rc |= dload(m, "transformer.time_embed.time_mlp.2.weight", 2, D, D, 0, w, 0.f);
  rc |= dload(m, "transformer.time_embed.time_mlp.2.bias", 1, D, 0, 0, w, 0.f);
  rc |= dload(m, "transformer.text_embed.text_embed.weight", 2, V + 1, TD, 0, w, 0.f);
  for (int i = 0; i < 4; ++i) {
    const char* p = "transformer.text_embed.text_blocks";
    snprintf(name, sizeof name, "%s.%d.dwconv.weight", p, i);  rc |= dload(m, name, 3, TD, 1, 7, w, 0.f);
    snprintf(name, sizeof name, "%s.%d.dwconv.bias", p, i);    rc |= dload(m, name, 1, TD, 0, 0, w, 0.f);
    snprintf(name, sizeof name, "%s.%d.norm.weight", p, i);    rc |= dload(m, name, 1, TD, 0, 0, w, 1.f);
    snprintf(name, sizeof name, "%s.%d.norm.bias", p, i);      rc |= dload(m, name, 1, TD, 0, 0, w, 0.f);
    snprintf(name, sizeof name, "%s.%d.pwconv1.weight", p, i); rc |= dload(m, name, 2, 2 * TD, TD, 0, w, 0.f);
    snprintf(name, sizeof name, "%s.%d.pwconv1.bias", p, i);   rc |= dload(m, name, 1, 2 * TD, 0, 0, w, 0.f);
    snprintf(name, sizeof name, "%s.%d.grn.gamma", p, i);      rc |= dload(m, name, 3, 1, 1, 2 * TD, w, 0.f);
    snprintf(name, sizeof name, "%s.%d.grn.beta", p, i);       rc |= dload(m, name, 3, 1, 1, 2 * TD, w, 0.f);
    snprintf(name, sizeof name, "%s.%d.pwconv2.weight", p, i); rc |= dload(m, name, 2, TD, 2 * TD, 0, w, 0.f);
    snprintf(name, sizeof name, "%s.%d.pwconv2.bias", p, i);   rc |= dload(m, name, 1, TD, 0, 0, w, 0.f);
  }
  rc |= dload(m, "transformer.input_embed.proj.weight", 2, D, 2 * MD + TD, 0, w, 0.f);
  rc |= dload(m, "transformer.input_embed.proj.bias", 1, D, 0, 0, w, 0.f);
  rc |= dload(m, "transformer.input_embed.conv_pos_embed.conv1d.0.weight", 3, D, D / 16, 31, w, 0.f);
  rc |= dload(m, "transformer.input_embed.conv_pos_embed.conv1d.0.bias", 1, D, 0, 0, w, 0.f);
  rc |= dload(m, "transformer.input_embed.conv_pos_embed.conv1d.2.weight", 3, D, D / 16, 31, w, 0.f);
  rc |= dload(m, "transformer.input_embed.conv_pos_embed.conv1d.2.bias", 1, D, 0, 0, w, 0.f);
  {
    const char* p = "transformer.transformer_blocks.0";
    snprintf(name, sizeof name, "%s.attn_norm.linear.weight", p); rc |= dload(m, name, 2, 6 * D, D, 0, w, 0.f);
    snprintf(name, sizeof name, "%s.attn_norm.linear.bias", p);   rc |= dload(m, name, 1, 6 * D, 0, 0, w, 0.f);
    const char* qkv[3] = {"to_q", "to_k", "to_v"};
    for (int j = 0; j < 3; ++j) {
      snprintf(name, sizeof name, "%s.attn.%s.weight", p, qkv[j]); rc |= dload(m, name, 2, D, D, 0, w, 0.f);
      snprintf(name, sizeof name, "%s.attn.%s.bias", p, qkv[j]);   rc |= dload(m, name, 1, D, 0, 0, w, 0.f);
    }
    snprintf(name, sizeof name, "%s.attn.to_out.0.weight", p); rc |= dload(m, name, 2, D, D, 0, w, 0.f);
    snprintf(name, sizeof name, "%s.attn.to_out.0.bias", p);   rc |= dload(m, name, 1, D, 0, 0, w, 0.f);
    snprintf(name, sizeof name, "%s.ff.ff.0.0.weight", p);     rc |= dload(m, name, 2, 2 * D, D, 0, w, 0.f);
    snprintf(name, sizeof name, "%s.ff.ff.0.0.bias", p);       rc |= dload(m, name, 1, 2 * D, 0, 0, w, 0.f);
    snprintf(name, sizeof name, "%s.ff.ff.2.weight", p);       rc |= dload(m, name, 2, D, 2 * D, 0, w, 0.f);
    snprintf(name, sizeof name, "%s.ff.ff.2.bias", p);         rc |= dload(m, name, 1, D, 0, 0, w, 0.f);
  }
  rc |= dload(m, "transformer.norm_out.linear.weight", 2, 2 * D, D, 0, w, 0.f);
  rc |= dload(m, "transformer.norm_out.linear.bias", 1, 2 * D, 0, 0, w, 0.f);
  rc |= dload(m, "transformer.proj_out.weight", 2, MD, D, 0, w, 0.f);
  rc |= dload(m, "transformer.proj_out.bias", 1, MD, 0, 0, w, 0.f);
  /* tables the reference keeps as non-persistent buffers (lemas_hip.h): rotary inverse frequencies, the text position table
   * cat(cos, sin) of outer(pos, 1 / 10000^(2i / text_dim)) and the sinusoid frequencies of the time embedding */
  {
    float inv[32];
    for (int j = 0; j < 32; ++j) inv[j] = (float)pow(10000.0, -2.0 * j / 64.0);
    int64_t sh1[1] = {32};
    rc |= lemas_dit_load_weight(m, "transformer.rotary_embed.inv_freq", inv, sh1, 1);
    float* fc = (float*)malloc(sizeof(float) * 4096 * TD);
    for (int pos = 0; pos < 4096; ++pos)
      for (int i = 0; i < TD / 2; ++i) {
        const float a = (float)pos * (float)(1.0 / pow(10000.0, 2.0 * i / TD));
        fc[(size_t)pos * TD + i] = cosf(a);
        fc[(size_t)pos * TD + TD / 2 + i] = sinf(a);
      }
    int64_t sh2[2] = {4096, TD};
    rc |= lemas_dit_load_weight(m, "transformer.text_embed.freqs_cis", fc, sh2, 2);
    free(fc);
    float tf[128];
    for (int i = 0; i < FD / 2; ++i) tf[i] = expf(-(float)i * (logf(10000.0f) / (FD / 2 - 1)));
    int64_t sh3[1] = {FD / 2};
    rc |= lemas_dit_load_weight(m, "transformer.time_embed.freqs", tf, sh3, 1);
  }
  if (rc) { fprintf(stderr, "dit load: %s\n", lemas_last_error()); return 11; }
  /* strict loading (utils_infer.py:237): an unknown key and a wrong shape are refused */
  {
    float z[4] = {0};
    int64_t s4[1] = {4};
    if (lemas_dit_load_weight(m, "transformer.no_such_tensor", z, s4, 1) == 0) return 12;
    if (lemas_dit_load_weight(m, "transformer.proj_out.bias", z, s4, 1) == 0) return 12;
  }
  if (lemas_dit_finalize(m)) { fprintf(stderr, "dit finalize: %s\n", lemas_last_error()); return 13; }

  const int B = 1, N = 100, F = 40, NT = 12, S = 3;
  float *hcond = (float*)calloc((size_t)B * N * MD, sizeof(float)), *hy0 = (float*)malloc(sizeof(float) * B * N * MD);
  unsigned char* hmask = (unsigned char*)calloc((size_t)B * N, 1);
  int64_t htext[12];
  for (int i = 0; i < F * MD; ++i) hcond[i] = -3.0f + 2.0f * frand();          /* reference mel, zero right-padded to N */
  for (int i = 0; i < F; ++i) hmask[i] = 1;
  for (int i = 0; i < B * N * MD; ++i) hy0[i] = 1.7f * frand();
  for (int i = 0; i < NT; ++i) htext[i] = 1 + (int64_t)((frand() * 0.5f + 0.5f) * (V - 2));
  float tgrid[4];
  for (int k = 0; k <= S; ++k) tgrid[k] = powf((float)k / S, 1.0f + 2.0f);         /* a sway-warped grid, strictly increasing */
  void *dcond = NULL, *dmask = NULL, *dtext = NULL, *dy = NULL, *dout = NULL;
  if (hipMalloc(&dcond, sizeof(float) * B * N * MD) || hipMalloc(&dmask, (size_t)B * N) || hipMalloc(&dtext, sizeof htext) ||
      hipMalloc(&dy, sizeof(float) * B * N * MD) || hipMalloc(&dout, sizeof(float) * B * N * MD)) return 14;
  hipMemcpy(dcond, hcond, sizeof(float) * B * N * MD, H2D);
  hipMemcpy(dmask, hmask, (size_t)B * N, H2D);
  hipMemcpy(dtext, htext, sizeof htext, H2D);
  lemas_sample_args a;
  memset(&a, 0, sizeof a);
  a.struct_size = sizeof a;                 /* ABI 200: the library refuses a struct of another layout */
  a.batch = B; a.frames = N; a.cond_frames = F; a.text_len = NT; a.steps = S; a.cfg_strength = 2.0f;
  a.cond = (const float*)dcond; a.cond_mask = (const uint8_t*)dmask; a.text = (const int64_t*)dtext; a.t_grid = tgrid;
  a.y = (float*)dy; a.out = (float*)dout;
  float *o1 = (float*)malloc(sizeof(float) * B * N * MD), *o2 = (float*)malloc(sizeof(float) * B * N * MD);
  for (int rep = 0; rep < 2; ++rep) {
    hipMemcpy(dy, hy0, sizeof(float) * B * N * MD, H2D);
    if (lemas_dit_sample(m, &a, NULL)) { fprintf(stderr, "dit sample: %s\n", lemas_last_error()); return 15; }
    hipDeviceSynchronize();
    hipMemcpy(rep ? o2 : o1, dout, sizeof(float) * B * N * MD, D2H);
  }
  if (lemas_dit_health(m)) { fprintf(stderr, "dit health: %s\n", lemas_last_error()); return 16; }
  double gen = 0;
  int exact = 1, same = 1;
  for (int i = 0; i < B * N * MD; ++i) {
    if (!isfinite(o1[i])) { fprintf(stderr, "dit: non-finite output\n"); return 17; }
    if (o1[i] != o2[i]) same = 0;
    if (i < F * MD) { if (o1[i] != hcond[i]) exact = 0; }
    else gen += fabs(o1[i] - hy0[i]);
  }
  a.steps = 0;
  const int bad = lemas_dit_sample(m, &a, NULL);
  printf("c_abi_smoke: dit depth 1: %d frames x %d mel, repeatable %d, conditioning frames copied %d, mean |out - y0| over generated frames %.4g, "
         "bad-arg rc %d (%s)\n", N, MD, same, exact, gen / ((N - F) * MD), bad, lemas_last_error());
  lemas_dit_destroy(m);
  hipFree(dcond); hipFree(dmask); hipFree(dtext); hipFree(dy); hipFree(dout);
  free(hcond); free(hy0); free(hmask); free(o1); free(o2);
  return (same && exact && gen > 0 && bad == LEMAS_E_ARG) ? 0 : 18;
}

static int stft_section(void) {
  const int nfft = 1024, hop = 512, n = 8192, B = 2;
  float* w = (float*)malloc(nfft * sizeof(float));
  for (int i = 0; i < nfft; ++i) w[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * i / (nfft - 1)));    /* hann_window(periodic=False) */
  lemas_stft* st = NULL;
  if (lemas_stft_create(nfft, hop, w, &st) != 0) { fprintf(stderr, "stft create: %s\n", lemas_last_error()); return 1; }
  const int frames = (int)lemas_stft_frames(st, n), ld = lemas_stft_ld(st);
  if (frames != n / hop + 1 || ld < nfft + 2) { fprintf(stderr, "stft: frames %d ld %d\n", frames, ld); return 1; }
  float* x = (float*)malloc((size_t)B * n * sizeof(float));
  for (int i = 0; i < B * n; ++i) x[i] = 0.5f * frand();
  float *dx = NULL, *dspec = NULL, *dy = NULL;
  hipMalloc((void**)&dx, (size_t)B * n * 4);
  hipMalloc((void**)&dspec, (size_t)B * frames * ld * 4);
  hipMalloc((void**)&dy, (size_t)B * hop * (frames - 1) * 4);
  hipMemcpy(dx, x, (size_t)B * n * 4, H2D);
  if (lemas_stft_forward(st, dx, B, n, dspec, NULL) != 0 || lemas_stft_inverse(st, dspec, B, frames, dy, NULL) != 0) {
    fprintf(stderr, "stft: %s\n", lemas_last_error());
    return 1;
  }
  hipDeviceSynchronize();
  float* y = (float*)malloc((size_t)B * n * sizeof(float));
  hipMemcpy(y, dy, (size_t)B * n * 4, D2H);         /* hop * (frames - 1) == n */
  double err = 0.0;
  for (int i = 0; i < B * n; ++i) { const double d = fabs((double)y[i] - x[i]); if (d > err) err = d; }
  if (!(err < 2e-5)) { fprintf(stderr, "stft round trip: max |err| %g\n", err); return 1; }
  if (lemas_stft_forward(st, dx, B, nfft / 2, dspec, NULL) == 0) { fprintf(stderr, "stft: a signal too short for the reflect padding was accepted\n"); return 1; }
  printf("stft round trip OK (max |err| %.2e, %d frames, ld %d); refusal: %s\n", err, frames, ld, lemas_last_error());
  lemas_stft_destroy(st);
  hipFree(dx); hipFree(dspec); hipFree(dy);
  free(w); free(x); free(y);
  return 0;
}

int main(void) {
  lemas_vocos* v = NULL;
  if (lemas_vocos_create(100, 512, 1536, 8, 1024, 256, &v)) { fprintf(stderr, "create: %s\n", lemas_last_error()); return 2; }
  char name[96];
  int rc = 0;
  rc |= load(v, "backbone.embed.weight", 3, 512, 100, 7, 0.03f, 0.f);
  rc |= load(v, "backbone.embed.bias", 1, 512, 0, 0, 0.03f, 0.f);
  rc |= load(v, "backbone.norm.weight", 1, 512, 0, 0, 0.03f, 1.f);
  rc |= load(v, "backbone.norm.bias", 1, 512, 0, 0, 0.03f, 0.f);
  for (int i = 0; i < 8; ++i) {
    snprintf(name, sizeof name, "backbone.convnext.%d.dwconv.weight", i); rc |= load(v, name, 3, 512, 1, 7, 0.03f, 0.f);
    snprintf(name, sizeof name, "backbone.convnext.%d.dwconv.bias", i);   rc |= load(v, name, 1, 512, 0, 0, 0.03f, 0.f);
    snprintf(name, sizeof name, "backbone.convnext.%d.norm.weight", i);   rc |= load(v, name, 1, 512, 0, 0, 0.03f, 1.f);
    snprintf(name, sizeof name, "backbone.convnext.%d.norm.bias", i);     rc |= load(v, name, 1, 512, 0, 0, 0.03f, 0.f);
    snprintf(name, sizeof name, "backbone.convnext.%d.pwconv1.weight", i); rc |= load(v, name, 2, 1536, 512, 0, 0.03f, 0.f);
    snprintf(name, sizeof name, "backbone.convnext.%d.pwconv1.bias", i);  rc |= load(v, name, 1, 1536, 0, 0, 0.03f, 0.f);
    snprintf(name, sizeof name, "backbone.convnext.%d.pwconv2.weight", i); rc |= load(v, name, 2, 512, 1536, 0, 0.03f, 0.f);
    snprintf(name, sizeof name, "backbone.convnext.%d.pwconv2.bias", i);  rc |= load(v, name, 1, 512, 0, 0, 0.03f, 0.f);
    snprintf(name, sizeof name, "backbone.convnext.%d.gamma", i);         rc |= load(v, name, 1, 512, 0, 0, 0.02f, 0.1f);
  }
  rc |= load(v, "backbone.final_layer_norm.weight", 1, 512, 0, 0, 0.03f, 1.f);
  rc |= load(v, "backbone.final_layer_norm.bias", 1, 512, 0, 0, 0.03f, 0.f);
  rc |= load(v, "head.out.weight", 2, 1026, 512, 0, 0.03f, 0.f);
  rc |= load(v, "head.out.bias", 1, 1026, 0, 0, 0.03f, 0.f);
  rc |= load(v, "head.istft.window", 1, 1024, 0, 0, 0.f, 0.f);
  if (rc || lemas_vocos_finalize(v)) { fprintf(stderr, "finalize: %s\n", lemas_last_error()); return 3; }

  const int B = 2, L = 75, T = 256 * (L - 1);
  float* hmel = (float*)malloc(sizeof(float) * B * 100 * L);
  for (int i = 0; i < B * 100 * L; ++i) hmel[i] = -3.0f + 2.0f * frand();
  void *dmel = NULL, *dwav = NULL;
  if (hipMalloc(&dmel, sizeof(float) * B * 100 * L) || hipMalloc(&dwav, sizeof(float) * B * T)) return 4;
  hipMemcpy(dmel, hmel, sizeof(float) * B * 100 * L, H2D);
  float *w1 = (float*)malloc(sizeof(float) * B * T), *w2 = (float*)malloc(sizeof(float) * B * T), *w3 = (float*)malloc(sizeof(float) * B * T);
  if (lemas_vocos_decode(v, (const float*)dmel, B, L, 1.0f, (float*)dwav, NULL)) { fprintf(stderr, "decode: %s\n", lemas_last_error()); return 5; }
  hipDeviceSynchronize(); hipMemcpy(w1, dwav, sizeof(float) * B * T, D2H);
  lemas_vocos_decode(v, (const float*)dmel, B, L, 0.25f, (float*)dwav, NULL);
  hipDeviceSynchronize(); hipMemcpy(w2, dwav, sizeof(float) * B * T, D2H);
  lemas_vocos_decode(v, (const float*)dmel, B, L, 1.0f, (float*)dwav, NULL);
  hipDeviceSynchronize(); hipMemcpy(w3, dwav, sizeof(float) * B * T, D2H);
  double maxabs = 0, lin = 0, det = 0;
  for (int i = 0; i < B * T; ++i) {
    if (!isfinite(w1[i])) { fprintf(stderr, "non-finite sample\n"); return 6; }
    if (fabs(w1[i]) > maxabs) maxabs = fabs(w1[i]);
    if (fabs(0.25 * w1[i] - w2[i]) > lin) lin = fabs(0.25 * w1[i] - w2[i]);
    if (fabs(w1[i] - w3[i]) > det) det = fabs(w1[i] - w3[i]);
  }
  /* a bad argument must be refused with a message, not crash */
  int bad = lemas_vocos_decode(v, (const float*)dmel, B, 1, 1.0f, (float*)dwav, NULL);
  printf("c_abi_smoke: %d samples, max|wav| %.4g, gain-linearity err %.3g, repeat err %.3g, bad-arg rc %d (%s)\n", B * T, maxabs, lin, det, bad,
         lemas_last_error());
  lemas_vocos_destroy(v);
  hipFree(dmel); hipFree(dwav);
  if (!(maxabs > 0 && lin <= 1e-6 * (maxabs + 1) && det == 0.0 && bad == LEMAS_E_ARG)) return 7;
  if (dit_section() != 0) return 1;
  return stft_section();
}
