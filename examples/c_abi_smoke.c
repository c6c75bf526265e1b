/* Plain-C client of liblemas_hip.so: proves the boundary is a C ABI (no C++/torch types cross it).
 *
 *   gcc -O2 -I include examples/c_abi_smoke.c -o examples/c_abi_smoke -L lemas_tts_amd/lib -llemas_hip \
 *       -Wl,-rpath,$PWD/lemas_tts_amd/lib -L/opt/rocm/lib -lamdhip64 -lm
 *
 * It runs the Vocos vocoder (lemas_vocos_*) on a synthetic mel with synthetic weights and checks an invariant that
 * needs no oracle: decode is linear in `gain`, deterministic, finite, and has the right length (256 (L-1) samples).
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
  return (maxabs > 0 && lin <= 1e-6 * (maxabs + 1) && det == 0.0 && bad == LEMAS_E_ARG) ? 0 : 7;
}
