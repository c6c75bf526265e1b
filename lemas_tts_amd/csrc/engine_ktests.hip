// Single-kernel entry points of the C ABI (lemas_k_*): thin drivers that feed fp32 device arrays through the
// production kernels so the parity tests can localise a failure to one kernel.  They allocate scratch with
// hipMalloc and synchronise -- test infrastructure, never on the sampling path.
#include <cmath>
#include <vector>

#include "engine_common.h"

using namespace lemas;

namespace {

struct Scratch {
  std::vector<void*> ptrs;
  ~Scratch() {
    for (void* p : ptrs) (void)hipFree(p);
  }
  template <typename T>
  T* get(size_t n) {
    void* p = nullptr;
    if (hipMalloc(&p, n * sizeof(T) + 256) != hipSuccess) return nullptr;
    if (zero_fill_sync(p, n * sizeof(T) + 256) != hipSuccess) { (void)hipFree(p); return nullptr; }
    ptrs.push_back(p);
    return reinterpret_cast<T*>(p);
  }
};

__global__ void transpose_v_kernel(const float* v, bf16_t* vt, int BH, int N, int npad) {
  // v [BH, N, 64] fp32 -> vt [BH, 64, npad] bf16
  const size_t total = (size_t)BH * N * 64;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % 64);
    const int n = (int)((i / 64) % N);
    const int bh = (int)(i / ((size_t)64 * N));
    vt[((size_t)bh * 64 + d) * npad + n] = (bf16_t)v[i];
  }
}
// [BH][N][64] fp32 -> [BH][pitch][64] bf16 (rows >= N stay zero)
__global__ void pad_rows_kernel(const float* src, bf16_t* dst, int BH, int N, int pitch) {
  const size_t total = (size_t)BH * N * 64;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % 64);
    const int n = (int)((i / 64) % N);
    const int bh = (int)(i / ((size_t)64 * N));
    dst[((size_t)bh * pitch + n) * 64 + d] = (bf16_t)src[i];
  }
}
// [B][pitch][C] bf16 -> [B][N][C] fp32
__global__ void unpad_widen_kernel(const bf16_t* src, float* dst, int B, int N, int pitch, int C) {
  const size_t total = (size_t)B * N * C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int n = (int)((i / C) % N);
    const int b = (int)(i / ((size_t)C * N));
    dst[i] = (float)src[((size_t)b * pitch + n) * C + c];
  }
}
__global__ void widen_kernel(const bf16_t* src, float* dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (float)src[i];
}
__global__ void fill_i32_kernel(int* p, int v) { p[0] = v; }
// pseudo-random bf16 in [-1, 1): benchmarks must not run on zero-filled operands (clock/power artefacts)
__global__ void fill_pattern_kernel(bf16_t* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed * 40503u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = (bf16_t)((float)(x & 0xffff) / 32768.0f - 1.0f);
  }
}

}  // namespace

static int g_attn_variant = 0;

extern "C" {

int lemas_k_set_attention_variant(int32_t v) { g_attn_variant = v; return 0; }

int lemas_k_linear_bf16(const float* A, const float* W, const float* bias, float* out, int32_t M, int32_t N, int32_t K,
                        int32_t act, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (K % 64 != 0) { set_error("lemas_k_linear_bf16: K must be a multiple of 64"); return LEMAS_E_ARG; }
  Scratch sc;
  const int Np = (N + 127) & ~127;
  bf16_t* a = sc.get<bf16_t>((size_t)M * K);
  bf16_t* w = sc.get<bf16_t>((size_t)Np * K);
  float* b = sc.get<float>(Np);
  bf16_t* ob = sc.get<bf16_t>((size_t)M * N);
  if (!a || !w || !b || !ob) { set_error("lemas_k_linear_bf16: out of memory"); return LEMAS_E_STATE; }
  HIP_TRY(launch_f32_to_bf16(A, a, (size_t)M * K, s));
  HIP_TRY(launch_f32_to_bf16(W, w, (size_t)N * K, s));
  if (bias) HIP_TRY(hipMemcpyAsync(b, bias, (size_t)N * 4, hipMemcpyDeviceToDevice, s));
  GemmParams p{};
  p.A = a; p.W = w; p.bias = b; p.M = M; p.N = Np; p.K = K; p.n_valid = N; p.ldc = N; p.seq_pitch = M; p.seq_valid = M; p.batch = 1;
  if (act == 1) {
    p.out_bf16 = ob;
    HIP_TRY(launch_gemm_bf16(EPI_BIAS_GELU_BF16, p, s));
    hipLaunchKernelGGL(widen_kernel, dim3(1024), dim3(256), 0, s, ob, out, (size_t)M * N);
  } else {
    p.out_f32 = out;
    HIP_TRY(launch_gemm_bf16(EPI_BIAS_F32, p, s));
  }
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

int lemas_k_linear_f32(const float* A, const float* W, const float* bias, float* out, int32_t M, int32_t N, int32_t K,
                       int32_t act, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  GemmF32Params p{};
  p.A = A; p.lda = K; p.W = W; p.ldw = K; p.bias = bias; p.out = out; p.ldc = N; p.M = M; p.N = N; p.K = K;
  HIP_TRY(launch_gemm_f32(act == 1 ? F32_BIAS_GELU : act == 2 ? F32_BIAS_SILU : F32_BIAS, p, s));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

int lemas_k_mx_quant(const float* x, int32_t M, int32_t K, uint8_t* out8, uint8_t* mx, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(launch_mx_quant_rows(x, M, K, out8, mx, s));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

int lemas_k_w_quant_f8(const float* w, int32_t N, int32_t K, uint8_t* out8, float* scale, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(launch_w_quant_f8(w, N, K, out8, scale, s));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

int lemas_k_ln_mod_f8(const float* x, const float* scale, const float* shift, uint8_t* out8, uint8_t* mx, int32_t M, int32_t D,
                      void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Scratch sc;
  float* tab = sc.get<float>((size_t)2 * D);
  if (!tab) { set_error("lemas_k_ln_mod_f8: out of memory"); return LEMAS_E_STATE; }
  HIP_TRY(hipMemcpyAsync(tab, scale, (size_t)D * 4, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(tab + D, shift, (size_t)D * 4, hipMemcpyDeviceToDevice, s));
  HIP_TRY(launch_ln_mod_f8(x, out8, mx, M, D, tab, 0, 0, D, nullptr, s));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

int lemas_k_linear_f8(const float* A, const float* W, const float* bias, float* out, int32_t M, int32_t N, int32_t K, int32_t act,
                      uint8_t* out8, uint8_t* outmx, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (K % 128 != 0) { set_error("lemas_k_linear_f8: K must be a multiple of 128"); return LEMAS_E_ARG; }
  if (act == 2 && (N % 128 != 0 || !out8 || !outmx)) { set_error("lemas_k_linear_f8: act 2 needs N %% 128 == 0 and out8/outmx"); return LEMAS_E_ARG; }
  Scratch sc;
  const int Np = (N + 127) & ~127;
  uint8_t* a8 = sc.get<uint8_t>((size_t)M * K);
  uint8_t* amx = sc.get<uint8_t>((size_t)M * (K / 32));
  uint8_t* w8 = sc.get<uint8_t>((size_t)Np * K);
  float* wsc = sc.get<float>(Np);
  float* b = sc.get<float>(Np);
  bf16_t* ob = sc.get<bf16_t>((size_t)M * N);
  if (!a8 || !amx || !w8 || !wsc || !b || !ob) { set_error("lemas_k_linear_f8: out of memory"); return LEMAS_E_STATE; }
  HIP_TRY(launch_mx_quant_rows(A, M, K, a8, amx, s));
  HIP_TRY(launch_w_quant_f8(W, N, K, w8, wsc, s));
  if (bias) HIP_TRY(hipMemcpyAsync(b, bias, (size_t)N * 4, hipMemcpyDeviceToDevice, s));
  GemmParams p{};
  p.A = reinterpret_cast<const bf16_t*>(a8); p.W = reinterpret_cast<const bf16_t*>(w8); p.bias = b; p.M = M; p.N = Np; p.K = K;
  p.n_valid = N; p.ldc = N; p.seq_pitch = M; p.seq_valid = M; p.batch = 1;
  p.f8 = 1; p.a_mx = amx; p.w_scale = wsc;
  if (act == 1) {
    p.out_bf16 = ob;
    HIP_TRY(launch_gemm_bf16(EPI_BIAS_GELU_BF16, p, s));
    hipLaunchKernelGGL(widen_kernel, dim3(1024), dim3(256), 0, s, ob, out, (size_t)M * N);
  } else if (act == 2) {
    p.out_f8 = out8; p.out_mx = outmx;
    HIP_TRY(launch_gemm_bf16(EPI_BIAS_GELU_F8, p, s));
  } else {
    p.out_f32 = out;
    HIP_TRY(launch_gemm_bf16(EPI_BIAS_F32, p, s));
  }
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

int lemas_k_attention(const float* q, const float* k, const float* v, const int32_t* seq_len, float* out, int32_t B,
                      int32_t H, int32_t N, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Scratch sc;
  const int pitch = (N + 127) & ~127, npad = pitch;
  const size_t np = (size_t)B * H * pitch * 64;
  bf16_t* qb = sc.get<bf16_t>(np);
  bf16_t* kb = sc.get<bf16_t>(np);
  bf16_t* vt = sc.get<bf16_t>((size_t)B * H * 64 * npad);
  bf16_t* ob = sc.get<bf16_t>(np);
  if (!qb || !kb || !vt || !ob) { set_error("lemas_k_attention: out of memory"); return LEMAS_E_STATE; }
  hipLaunchKernelGGL(pad_rows_kernel, dim3(2048), dim3(256), 0, s, q, qb, B * H, N, pitch);
  hipLaunchKernelGGL(pad_rows_kernel, dim3(2048), dim3(256), 0, s, k, kb, B * H, N, pitch);
  hipLaunchKernelGGL(transpose_v_kernel, dim3(2048), dim3(256), 0, s, v, vt, B * H, N, npad);
  AttnParams p{};
  p.q = qb; p.k = kb; p.vt = vt; p.out = ob; p.kv_len = seq_len; p.b2 = B; p.batch = B; p.heads = H; p.n = N; p.npad = npad; p.pitch = pitch; p.variant = g_attn_variant;
  p.scale = 0.125f;
  HIP_TRY(launch_attention(p, s));
  hipLaunchKernelGGL(unpad_widen_kernel, dim3(2048), dim3(256), 0, s, ob, out, B, N, pitch, H * 64);
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

int lemas_k_ln_mod(const float* x, const float* scale, const float* shift, float* out, int32_t M, int32_t D, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Scratch sc;
  float* tab = sc.get<float>((size_t)2 * D);
  bf16_t* ob = sc.get<bf16_t>((size_t)M * D);
  if (!tab || !ob) { set_error("lemas_k_ln_mod: out of memory"); return LEMAS_E_STATE; }
  HIP_TRY(hipMemcpyAsync(tab, scale, (size_t)D * 4, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(tab + D, shift, (size_t)D * 4, hipMemcpyDeviceToDevice, s));
  HIP_TRY(launch_ln_mod(x, ob, M, D, tab, 0, 0, D, nullptr, s));
  hipLaunchKernelGGL(widen_kernel, dim3(1024), dim3(256), 0, s, ob, out, (size_t)M * D);
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

int lemas_k_convpos(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* out,
                    int32_t B, int32_t N, int32_t C, int32_t groups, int32_t taps, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Scratch sc;
  const int cg = C / groups;
  bf16_t* wa = sc.get<bf16_t>((size_t)C * cg * taps);
  bf16_t* wb = sc.get<bf16_t>((size_t)C * cg * taps);
  bf16_t* mid = sc.get<bf16_t>((size_t)B * N * C);
  if (!wa || !wb || !mid) { set_error("lemas_k_convpos: out of memory"); return LEMAS_E_STATE; }
  HIP_TRY(launch_convpos_weight(w1, wa, C, cg, taps, s));
  HIP_TRY(launch_convpos_weight(w2, wb, C, cg, taps, s));
  ConvPosParams c{};
  c.b2 = B; c.n = N; c.channels = C; c.groups = groups; c.taps = taps; c.pitch = N;
  c.in_f32 = x; c.w = wa; c.bias = b1; c.out_bf16 = mid;
  HIP_TRY(launch_convpos(c, s));
  c.in_f32 = nullptr; c.in_bf16 = mid; c.w = wb; c.bias = b2; c.out_bf16 = nullptr; c.out_f32 = out; c.residual = x;
  HIP_TRY(launch_convpos(c, s));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

}  // extern "C"

// ---- micro-benchmarks of the step-loop kernels at arbitrary shapes (development aid + bench.py roofline cross-check)
extern "C" int lemas_k_bench(const char* what, int32_t M, int32_t N, int32_t K, int32_t iters, int32_t variant, double* avg_us) {
  hipStream_t s = nullptr;
  hipStream_t own = nullptr;
  HIP_TRY(hipStreamCreate(&own));
  s = own;
  Scratch sc;
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  std::string w(what);
  const bool f8 = w.rfind("f8_", 0) == 0;
  if (f8) w = w.substr(3);
  float ms = 0.f;
  int rc = 0;
  auto time_it = [&](auto&& fn) -> int {
    for (int i = 0; i < 3; ++i) { hipError_t e = fn(); if (e != hipSuccess) return hip_fail(e, "bench warmup", __FILE__, __LINE__); }
    HIP_TRY(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) (void)fn();
    HIP_TRY(hipEventRecord(e1, s));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    return 0;
  };
  if (w == "gemm_gelu" || w == "gemm_gelu8" || w == "gemm_gate" || w == "gemm_qk" || w == "gemm_v" || w == "gemm_f32out" || w == "gemm_none" || w == "gemm_nodma") {
    const int Np = (N + 127) & ~127;
    bf16_t* a = sc.get<bf16_t>((size_t)M * K);
    bf16_t* wt = sc.get<bf16_t>((size_t)Np * K);
    float* b = sc.get<float>(Np);
    bf16_t* ob = sc.get<bf16_t>((size_t)M * Np);
    float* of = sc.get<float>((size_t)M * Np);
    float* tab = sc.get<float>(Np);
    int* step = sc.get<int>(16);
    const int npad = (M + 127) & ~127;
    bf16_t* q = sc.get<bf16_t>((size_t)M * 1024);
    bf16_t* k = sc.get<bf16_t>((size_t)M * 1024);
    bf16_t* vt = sc.get<bf16_t>((size_t)16 * 64 * npad);
    float* rc_ = sc.get<float>((size_t)M * 32);
    float* rs_ = sc.get<float>((size_t)M * 32);
    if (!a || !wt || !b || !ob || !of || !tab || !step || !q || !k || !vt || !rc_ || !rs_) { set_error("bench: out of memory"); return LEMAS_E_STATE; }
    // non-trivial operand bits (DVFS: zero-filled operands clock higher and overstate throughput)
    hipLaunchKernelGGL(fill_pattern_kernel, dim3(1024), dim3(256), 0, s, a, (size_t)M * K, 1u);
    hipLaunchKernelGGL(fill_pattern_kernel, dim3(1024), dim3(256), 0, s, wt, (size_t)Np * K, 2u);
    GemmParams p{};
    p.A = a; p.W = wt; p.bias = b; p.M = M; p.N = Np; p.K = K; p.n_valid = N; p.ldc = N; p.out_bf16 = ob; p.out_f32 = of;
    p.tab = tab; p.tab_stride = 0; p.gate_off = 0; p.step_idx = step; p.seq_pitch = M; p.seq_valid = M; p.batch = 1; p.heads = 16; p.npad = npad;
    p.q = q; p.k = k; p.vt = vt; p.rope_cos = rc_; p.rope_sin = rs_;
    if (f8) {
      uint8_t* amx = sc.get<uint8_t>((size_t)M * (K / 32));
      float* wsc = sc.get<float>(Np);
      uint8_t* o8 = sc.get<uint8_t>((size_t)M * Np);
      uint8_t* omx = sc.get<uint8_t>((size_t)M * (Np / 32));
      if (!amx || !wsc || !o8 || !omx) { set_error("bench: out of memory"); return LEMAS_E_STATE; }
      HIP_TRY(hipMemsetAsync(amx, 127, (size_t)M * (K / 32), s));
      p.f8 = 1; p.a_mx = amx; p.w_scale = wsc; p.out_f8 = o8; p.out_mx = omx;
    }
    const int epi = w == "gemm_gelu8" ? EPI_BIAS_GELU_F8 : w == "gemm_gelu" ? EPI_BIAS_GELU_BF16 : w == "gemm_gate" ? EPI_GATE_RES : w == "gemm_qk" ? EPI_QK_ROPE : w == "gemm_v" ? EPI_V_T : (w == "gemm_none" || w == "gemm_nodma") ? EPI_NONE : EPI_BIAS_F32;
    if (w == "gemm_nodma") p.n_valid = -1;
    if ((epi == EPI_QK_ROPE && N != 2048) || (epi == EPI_V_T && N != 1024)) { set_error("bench: gemm_qk needs N = 2048, gemm_v N = 1024"); return LEMAS_E_ARG; }
    rc = time_it([&]() { return launch_gemm_bf16_variant(epi, p, variant, s); });
  } else if (w == "attention") {
    // M = sequence length, N = batch*heads
    const int n = M, bh = N, npad = (n + 127) & ~127, pitch = npad;
    bf16_t* q = sc.get<bf16_t>((size_t)bh * pitch * 64);
    bf16_t* k = sc.get<bf16_t>((size_t)bh * pitch * 64);
    bf16_t* vt = sc.get<bf16_t>((size_t)bh * 64 * npad);
    bf16_t* o = sc.get<bf16_t>((size_t)bh * pitch * 64);
    if (!q || !k || !vt || !o) { set_error("bench: out of memory"); return LEMAS_E_STATE; }
    hipLaunchKernelGGL(fill_pattern_kernel, dim3(1024), dim3(256), 0, s, q, (size_t)bh * pitch * 64, 3u);
    hipLaunchKernelGGL(fill_pattern_kernel, dim3(1024), dim3(256), 0, s, k, (size_t)bh * pitch * 64, 4u);
    hipLaunchKernelGGL(fill_pattern_kernel, dim3(1024), dim3(256), 0, s, vt, (size_t)bh * 64 * npad, 5u);
    AttnParams p{};
    p.q = q; p.k = k; p.vt = vt; p.out = o; p.kv_len = nullptr; p.b2 = bh / 16; p.batch = bh / 16; p.heads = 16; p.n = n; p.npad = npad; p.pitch = pitch; p.variant = variant;
    p.scale = 0.125f;
    rc = time_it([&]() { return launch_attention(p, s); });
  } else {
    set_error("bench: unknown kernel '%s'", what);
    rc = LEMAS_E_ARG;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipStreamDestroy(own);
  if (rc == 0 && avg_us) *avg_us = 1e3 * ms / iters;
  return rc;
}

// ---- experiment: how well does the VALU-bound attention kernel overlap with the MFMA-bound GEMMs of the OTHER CFG lane
// when they run concurrently on two streams?  mode 0 = serial on one stream, 1 = concurrent.  Returns us per (attention +
// 5 GEMMs) group.
extern "C" int lemas_k_bench_overlap(int32_t mode, int32_t iters, int32_t gemm_variant_wide, int32_t gemm_variant_narrow, double* avg_us) {
  hipStream_t sa = nullptr, sb = nullptr;
  HIP_TRY(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  Scratch sc;
  const int M = 1920, n = 1875, bh = 16, pitch = 1920;
  bf16_t* a = sc.get<bf16_t>((size_t)M * 2048);
  bf16_t* wt = sc.get<bf16_t>((size_t)2048 * 2048);
  float* b = sc.get<float>(2048);
  bf16_t* ob = sc.get<bf16_t>((size_t)M * 2048);
  float* of = sc.get<float>((size_t)M * 2048);
  float* tab = sc.get<float>(2048);
  int* step = sc.get<int>(16);
  bf16_t* q = sc.get<bf16_t>((size_t)bh * pitch * 64);
  bf16_t* k = sc.get<bf16_t>((size_t)bh * pitch * 64);
  bf16_t* vt = sc.get<bf16_t>((size_t)bh * 64 * pitch);
  bf16_t* o = sc.get<bf16_t>((size_t)bh * pitch * 64);
  bf16_t* q2 = sc.get<bf16_t>((size_t)bh * pitch * 64);
  bf16_t* k2 = sc.get<bf16_t>((size_t)bh * pitch * 64);
  bf16_t* vt2 = sc.get<bf16_t>((size_t)bh * 64 * pitch);
  float* rc_ = sc.get<float>((size_t)M * 32);
  float* rs_ = sc.get<float>((size_t)M * 32);
  if (!a || !wt || !b || !ob || !of || !tab || !step || !q || !k || !vt || !o || !q2 || !k2 || !vt2 || !rc_ || !rs_) { set_error("bench: out of memory"); return LEMAS_E_STATE; }
  hipLaunchKernelGGL(fill_pattern_kernel, dim3(1024), dim3(256), 0, sa, a, (size_t)M * 2048, 1u);
  hipLaunchKernelGGL(fill_pattern_kernel, dim3(1024), dim3(256), 0, sa, wt, (size_t)2048 * 2048, 2u);
  hipLaunchKernelGGL(fill_pattern_kernel, dim3(1024), dim3(256), 0, sa, q, (size_t)bh * pitch * 64, 3u);
  hipLaunchKernelGGL(fill_pattern_kernel, dim3(1024), dim3(256), 0, sa, k, (size_t)bh * pitch * 64, 4u);
  hipLaunchKernelGGL(fill_pattern_kernel, dim3(1024), dim3(256), 0, sa, vt, (size_t)bh * 64 * pitch, 5u);
  HIP_TRY(hipStreamSynchronize(sa));
  AttnParams at{};
  at.q = q; at.k = k; at.vt = vt; at.out = o; at.b2 = 1; at.batch = 1; at.heads = 16; at.n = n; at.npad = pitch; at.pitch = pitch; at.scale = 0.125f;
  GemmParams g{};
  g.A = a; g.W = wt; g.bias = b; g.M = M; g.tab = tab; g.step_idx = step; g.seq_pitch = pitch; g.seq_valid = n; g.batch = 1; g.heads = 16; g.npad = pitch;
  g.q = q2; g.k = k2; g.vt = vt2; g.rope_cos = rc_; g.rope_sin = rs_; g.out_bf16 = ob; g.out_f32 = of;
  auto gemms = [&](hipStream_t s) {
    g.N = 2048; g.K = 1024; g.n_valid = 2048; g.ldc = 2048; (void)launch_gemm_bf16_variant(EPI_QK_ROPE, g, gemm_variant_wide, s);
    g.N = 1024; g.n_valid = 1024; (void)launch_gemm_bf16_variant(EPI_V_T, g, gemm_variant_narrow, s);
    g.ldc = 1024; (void)launch_gemm_bf16_variant(EPI_GATE_RES, g, gemm_variant_narrow, s);
    g.N = 2048; g.n_valid = 2048; g.ldc = 2048; (void)launch_gemm_bf16_variant(EPI_BIAS_GELU_BF16, g, gemm_variant_wide, s);
    g.N = 1024; g.K = 2048; g.n_valid = 1024; g.ldc = 1024; (void)launch_gemm_bf16_variant(EPI_GATE_RES, g, gemm_variant_narrow, s);
  };
  hipEvent_t e0, e1, ej;
  HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); HIP_TRY(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  for (int w = 0; w < 2; ++w) { (void)launch_attention(at, sa); gemms(mode ? sb : sa); }
  HIP_TRY(hipStreamSynchronize(sa)); HIP_TRY(hipStreamSynchronize(sb));
  HIP_TRY(hipEventRecord(e0, sa));
  if (mode == 1 || mode == 2) { HIP_TRY(hipStreamWaitEvent(sb, e0, 0)); }
  for (int i = 0; i < iters; ++i) {
    if (mode >= 2) {   // control experiment: the SAME half-size attention on both streams (mode 2) or twice on one (mode 3)
      (void)launch_attention(at, sa);
      (void)launch_attention(at, mode == 2 ? sb : sa);
      continue;
    }
    (void)launch_attention(at, sa);
    gemms(mode ? sb : sa);
  }
  if (mode == 1 || mode == 2) { HIP_TRY(hipEventRecord(ej, sb)); HIP_TRY(hipStreamWaitEvent(sa, ej, 0)); }
  HIP_TRY(hipEventRecord(e1, sa));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  if (avg_us) *avg_us = 1e3 * ms / iters;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(ej);
  (void)hipStreamDestroy(sa); (void)hipStreamDestroy(sb);
  return 0;
}
