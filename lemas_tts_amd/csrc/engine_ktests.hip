// Single-kernel entry points of the C ABI (lemas_k_*): thin drivers that feed fp32 device arrays through the
// production kernels so the parity tests can localise a failure to one kernel.  They allocate scratch with
// hipMalloc and synchronise -- test infrastructure, never on the sampling path.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <mutex>
#include "engine_common.h"
#include "../../include/lemas_hip_test.h"

using namespace lemas;

int lemas_internal_timeline(void* buf, int slots);   // engine_dit.hip: live in a -DLEMAS_PHASE_TIMESTAMPS build, an error otherwise

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
// [BH][N][64] fp32 -> [BH][pitch][64] bf16 (rows >= N stay zero), scaled in fp32 before the rounding
__global__ void pad_rows_kernel(const float* src, bf16_t* dst, int BH, int N, int pitch, float scale = 1.0f) {
  const size_t total = (size_t)BH * N * 64;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % 64);
    const int n = (int)((i / 64) % N);
    const int bh = (int)(i / ((size_t)64 * N));
    dst[((size_t)bh * pitch + n) * 64 + d] = (bf16_t)(src[i] * scale);
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

extern "C" {

int lemas_k_timeline(void* buf, int32_t slots) { return lemas_internal_timeline(buf, slots); }
int lemas_k_build_flags(void) {
  int f = 0;
#ifdef LEMAS_MEASUREMENT_BUILD
  f |= 1;
#endif
#ifdef LEMAS_PHASE_TIMESTAMPS
  f |= 2;
#endif
  return f;
}

int lemas_k_linear_bf16(const float* A, const float* W, const float* bias, float* out, int32_t M, int32_t N, int32_t K,
                        int32_t act, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  RC_TRY(kernels_init());
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

// One production GEMM launch with an explicit tile shape and epilogue in the engine's row space ([batch][pitch] rows, `frames`
// valid per sample), so that every (tile, epilogue) pair the sampler can pick is reachable from a unit test.
int lemas_k_gemm_epi(int32_t epi, int32_t tile, const float* A, const float* W, const float* bias, const float* aux,
                     const int32_t* seq_len, float* out, int32_t batch, int32_t pitch, int32_t frames, int32_t N, int32_t K, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  RC_TRY(kernels_init());
  if (batch <= 0 || pitch % 128 != 0 || frames <= 0 || frames > pitch || K % 64 != 0 || N % 128 != 0) {
    set_error("lemas_k_gemm_epi: need pitch %% 128 == 0, frames <= pitch, K %% 64 == 0, N %% 128 == 0");
    return LEMAS_E_ARG;
  }
  const int M = batch * pitch;
  Scratch sc;
  bf16_t* a = sc.get<bf16_t>((size_t)M * K);
  bf16_t* w = sc.get<bf16_t>((size_t)N * K);
  int* step = sc.get<int>(16);
  if (!a || !w || !step) { set_error("lemas_k_gemm_epi: out of memory"); return LEMAS_E_STATE; }
  HIP_TRY(launch_f32_to_bf16(A, a, (size_t)M * K, s));
  HIP_TRY(launch_f32_to_bf16(W, w, (size_t)N * K, s));
  GemmParams p{};
  p.A = a; p.W = w; p.bias = bias; p.M = M; p.N = N; p.K = K; p.n_valid = N; p.ldc = N;
  p.seq_pitch = pitch; p.seq_valid = frames; p.batch = batch; p.step_idx = step;
  if (epi == EPI_BIAS_F32) {
    p.out_f32 = out;
    HIP_TRY(launch_gemm_bf16_tile(epi, p, tile, s));
  } else if (epi == EPI_BIAS_BF16 || epi == EPI_BIAS_GELU_BF16) {
    bf16_t* ob = sc.get<bf16_t>((size_t)M * N);
    if (!ob) { set_error("lemas_k_gemm_epi: out of memory"); return LEMAS_E_STATE; }
    p.out_bf16 = ob;
    HIP_TRY(launch_gemm_bf16_tile(epi, p, tile, s));
    hipLaunchKernelGGL(widen_kernel, dim3(2048), dim3(256), 0, s, ob, out, (size_t)M * N);
  } else if (epi == EPI_GATE_RES) {     // out is the residual stream (in/out); aux = gate[N]
    p.out_f32 = out; p.tab = aux; p.tab_stride = 0; p.gate_off = 0; p.kv_len = seq_len;
    HIP_TRY(launch_gemm_bf16_tile(epi, p, tile, s));
  } else if (epi == EPI_QK_ROPE) {      // aux = [cos | sin], each [frames][32]; out = q then k, each [batch][H][pitch][64]
    const int heads = N / 128;
    const size_t per = (size_t)M * heads * 64;
    bf16_t* qk = sc.get<bf16_t>(2 * per);
    if (!qk) { set_error("lemas_k_gemm_epi: out of memory"); return LEMAS_E_STATE; }
    p.q = qk; p.k = qk + per; p.heads = heads; p.npad = pitch; p.rope_cos = aux; p.rope_sin = aux + (size_t)frames * 32;
    HIP_TRY(launch_gemm_bf16_tile(epi, p, tile, s));
    hipLaunchKernelGGL(widen_kernel, dim3(2048), dim3(256), 0, s, qk, out, 2 * per);
  } else if (epi == EPI_V_T) {          // out = v^T [batch][H][64][pitch]
    const int heads = N / 64;
    const size_t per = (size_t)M * heads * 64;
    bf16_t* vt = sc.get<bf16_t>(per);
    if (!vt) { set_error("lemas_k_gemm_epi: out of memory"); return LEMAS_E_STATE; }
    p.vt = vt; p.heads = heads; p.npad = pitch;
    HIP_TRY(launch_gemm_bf16_tile(epi, p, tile, s));
    hipLaunchKernelGGL(widen_kernel, dim3(2048), dim3(256), 0, s, vt, out, per);
  } else {
    set_error("lemas_k_gemm_epi: unknown epilogue %d", epi);
    return LEMAS_E_ARG;
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

// The gate + residual GEMM with its LayerNorm-modulate tail (GemmParams::ln_out): x (in/out) is the residual stream [batch*pitch, 1024],
// h (out) the modulated LayerNorm of the UPDATED rows, widened to fp32.  `concurrent` > 1 launches that many independent copies on
// as many streams at once (what the two CFG lanes do), each with its own operands' copies and counters.
int lemas_k_gemm_gate_ln(int32_t tile, const float* A, const float* W, const float* bias, const float* gate, const float* scale,
                         const float* shift, const int32_t* seq_len, float* x, float* h, int32_t batch, int32_t pitch, int32_t frames,
                         int32_t K, int32_t concurrent, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  RC_TRY(kernels_init());
  const int N = 1024;
  if (batch <= 0 || pitch % 128 != 0 || frames <= 0 || frames > pitch || K % 64 != 0 || concurrent < 1 || concurrent > 4) {
    set_error("lemas_k_gemm_gate_ln: need pitch %% 128 == 0, frames <= pitch, K %% 64 == 0, 1 <= concurrent <= 4");
    return LEMAS_E_ARG;
  }
  const int M = batch * pitch;
  Scratch sc;
  bf16_t* a = sc.get<bf16_t>((size_t)M * K);
  bf16_t* w = sc.get<bf16_t>((size_t)N * K);
  float* tab = sc.get<float>((size_t)3 * N);
  int* step = sc.get<int>(16);
  unsigned int* cnt = sc.get<unsigned int>((size_t)concurrent * (M / 64 + 1));
  float* xs = sc.get<float>((size_t)concurrent * M * N);          // one residual stream per concurrent copy
  bf16_t* hs = sc.get<bf16_t>((size_t)concurrent * M * N);
  unsigned int* err_host = nullptr;
  unsigned int* err_dev = nullptr;
  if (!a || !w || !tab || !step || !cnt || !xs || !hs) { set_error("lemas_k_gemm_gate_ln: out of memory"); return LEMAS_E_STATE; }
  HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&err_host), 64, hipHostMallocMapped));
  *err_host = 0;
  HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&err_dev), err_host, 0));
  HIP_TRY(launch_f32_to_bf16(A, a, (size_t)M * K, s));
  HIP_TRY(launch_f32_to_bf16(W, w, (size_t)N * K, s));
  HIP_TRY(hipMemcpyAsync(tab, gate, (size_t)N * 4, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(tab + N, scale, (size_t)N * 4, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(tab + 2 * N, shift, (size_t)N * 4, hipMemcpyDeviceToDevice, s));
  for (int c = 0; c < concurrent; ++c) HIP_TRY(hipMemcpyAsync(xs + (size_t)c * M * N, x, (size_t)M * N * 4, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  GemmParams p{};
  p.A = a; p.W = w; p.bias = bias; p.M = M; p.N = N; p.K = K; p.n_valid = N; p.ldc = N;
  p.seq_pitch = pitch; p.seq_valid = frames; p.batch = batch; p.step_idx = step;
  p.tab = tab; p.tab_stride = 0; p.gate_off = 0; p.kv_len = seq_len;
  p.ln_scale_off = N; p.ln_shift_off = 2 * N; p.ln_err = err_dev; p.concurrency = concurrent;
  std::vector<hipStream_t> st(concurrent, nullptr);
  int rc = 0;
  for (int c = 0; c < concurrent && rc == 0; ++c)
    if (hipStreamCreateWithFlags(&st[c], hipStreamNonBlocking) != hipSuccess) rc = LEMAS_E_STATE;
  for (int c = 0; c < concurrent && rc == 0; ++c) {
    p.out_f32 = xs + (size_t)c * M * N; p.ln_out = hs + (size_t)c * M * N; p.ln_cnt = cnt + (size_t)c * (M / 64 + 1);
    hipError_t e = launch_gemm_bf16_tile(EPI_GATE_RES, p, tile, st[c]);
    if (e != hipSuccess) rc = hip_fail(e, "launch_gemm_bf16_tile(gate + LayerNorm tail)", __FILE__, __LINE__);
  }
  for (int c = 0; c < concurrent; ++c)
    if (st[c]) { (void)hipStreamSynchronize(st[c]); (void)hipStreamDestroy(st[c]); }
  const unsigned int gave_up = *err_host;
  (void)hipHostFree(err_host);
  if (rc != 0) return rc;
  if (gave_up) { set_error("lemas_k_gemm_gate_ln: a LayerNorm tail gave up waiting for its row panel"); return LEMAS_E_STATE; }
  // every concurrent copy computed the same thing: compare them bit for bit, return copy 0
  if (concurrent > 1) {
    std::vector<float> h0((size_t)M * N), hc((size_t)M * N);
    std::vector<unsigned short> b0((size_t)M * N), bc((size_t)M * N);
    HIP_TRY(hipMemcpy(h0.data(), xs, h0.size() * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(b0.data(), hs, b0.size() * 2, hipMemcpyDeviceToHost));
    for (int c = 1; c < concurrent; ++c) {
      HIP_TRY(hipMemcpy(hc.data(), xs + (size_t)c * M * N, hc.size() * 4, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(bc.data(), hs + (size_t)c * M * N, bc.size() * 2, hipMemcpyDeviceToHost));
      if (memcmp(h0.data(), hc.data(), h0.size() * 4) != 0 || memcmp(b0.data(), bc.data(), b0.size() * 2) != 0) {
        set_error("lemas_k_gemm_gate_ln: concurrent copy %d differs from copy 0", c);
        return LEMAS_E_STATE;
      }
    }
  }
  HIP_TRY(hipMemcpyAsync(x, xs, (size_t)M * N * 4, hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(widen_kernel, dim3(2048), dim3(256), 0, s, hs, h, (size_t)M * N);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

// ln fold at kernel level (common.h GemmParams): producer (the gate + residual GEMM on tile prod_tile, or with use_prep the chain-entry
// kernel on x as given) -> the c1 / c2 rows through the production table builders (one site, one step) -> consumer GEMM with epilogue
// cons_epi (1 GELU -> y [M][Nc]; 4 QK+RoPE -> y = q then k [batch][H][pitch][64]; 5 -> y = v^T [batch][H][64][pitch]) on tile cons_tile.
int lemas_k_ln_fold_pair(int32_t prod_tile, int32_t cons_epi, int32_t cons_tile, const float* A, const float* Wp, const float* bias_p,
                         const float* gate, const float* scale, const float* shift, const float* Wc, const float* bias_c, const float* rope,
                         const int32_t* seq_len, float* x, float* y, int32_t batch, int32_t pitch, int32_t frames, int32_t Kp, int32_t Nc,
                         int32_t use_prep, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  RC_TRY(kernels_init());
  const int D = 1024;
  if (batch <= 0 || pitch % 128 != 0 || frames <= 0 || frames > pitch || Kp % 64 != 0 || Nc % 128 != 0 ||
      !(cons_epi == EPI_BIAS_GELU_BF16 || cons_epi == EPI_QK_ROPE || cons_epi == EPI_V_T) || (cons_epi == EPI_QK_ROPE && !rope)) {
    set_error("lemas_k_ln_fold_pair: bad arguments");
    return LEMAS_E_ARG;
  }
  const int M = batch * pitch, stride = 3 * D + 2 * Nc;
  Scratch sc;
  bf16_t* a = sc.get<bf16_t>((size_t)M * Kp);
  bf16_t* wp = sc.get<bf16_t>((size_t)D * Kp);
  bf16_t* wc = sc.get<bf16_t>((size_t)Nc * D);
  bf16_t* xs = sc.get<bf16_t>((size_t)M * D);
  float* part = sc.get<float>((size_t)M * 64);
  float* tab = sc.get<float>((size_t)stride);
  float* zero = sc.get<float>((size_t)Nc);
  float* tmp = sc.get<float>((size_t)4 * Nc);
  bf16_t* fa = sc.get<bf16_t>((size_t)4 * D);
  LnFoldSite* site_d = sc.get<LnFoldSite>(1);
  GemmParams* gp_d = sc.get<GemmParams>(1);
  int* step = sc.get<int>(16);
  bf16_t* ob = sc.get<bf16_t>((size_t)M * Nc);
  if (!a || !wp || !wc || !xs || !part || !tab || !zero || !tmp || !fa || !site_d || !gp_d || !step || !ob) {
    set_error("lemas_k_ln_fold_pair: out of memory");
    return LEMAS_E_STATE;
  }
  HIP_TRY(launch_f32_to_bf16(A, a, (size_t)M * Kp, s));
  HIP_TRY(launch_f32_to_bf16(Wp, wp, (size_t)D * Kp, s));
  HIP_TRY(launch_f32_to_bf16(Wc, wc, (size_t)Nc * D, s));
  HIP_TRY(hipMemcpyAsync(tab, gate, (size_t)D * 4, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(tab + D, scale, (size_t)D * 4, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(tab + 2 * D, shift, (size_t)D * 4, hipMemcpyDeviceToDevice, s));
  // the table rows, exactly as lemas_dit::build_fold_tables does it
  LnFoldSite site{};
  site.bias = bias_c; site.tmp = tmp; site.N = Nc; site.scale_off = D; site.shift_off = 2 * D; site.c1_off = 3 * D; site.c2_off = 3 * D + Nc;
  GemmParams gf{};
  gf.A = fa; gf.W = wc; gf.bias = zero; gf.M = 4; gf.N = Nc; gf.K = D; gf.n_valid = Nc; gf.out_f32 = tmp; gf.ldc = Nc;
  gf.seq_pitch = 128; gf.seq_valid = 4; gf.batch = 1; gf.xcd_gx = 8;
  HIP_TRY(hipMemcpyAsync(site_d, &site, sizeof site, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(gp_d, &gf, sizeof gf, hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(launch_ln_fold_split(tab, stride, 1, D, site_d, 1, fa, s));
  HIP_TRY(launch_gemm_bf16_group(gp_d, 1, Nc / 128, s));
  HIP_TRY(launch_ln_fold_combine(site_d, 1, Nc, 1, tab, stride, s));
  GemmParams p{};
  p.M = M; p.seq_pitch = pitch; p.seq_valid = frames; p.batch = batch; p.step_idx = step; p.tab = tab; p.tab_stride = stride;
  if (use_prep) {
    HIP_TRY(launch_ln_prep(x, xs, part, M, D, tab, stride, D, step, s));
  } else {
    p.A = a; p.W = wp; p.bias = bias_p; p.N = D; p.K = Kp; p.n_valid = D; p.ldc = D; p.out_f32 = x; p.gate_off = 0; p.kv_len = seq_len;
    p.xs_out = xs; p.xs_scale_off = D; p.ln_part_out = part; p.ln_np = D / 32;
    HIP_TRY(launch_gemm_bf16_tile(EPI_GATE_RES, p, prod_tile, s));
    p.xs_out = nullptr; p.ln_part_out = nullptr; p.kv_len = nullptr;
  }
  p.A = xs; p.W = wc; p.bias = bias_c; p.N = Nc; p.K = D; p.n_valid = Nc; p.ldc = Nc;
  p.ln_part = part; p.ln_np = D / 32; p.lnc1_off = 3 * D; p.lnc2_off = 3 * D + Nc;
  size_t nout = (size_t)M * Nc;
  if (cons_epi == EPI_BIAS_GELU_BF16) {
    p.out_bf16 = ob;
  } else if (cons_epi == EPI_QK_ROPE) {
    p.heads = Nc / 128; p.q = ob; p.k = ob + (size_t)M * p.heads * 64; p.npad = pitch; p.rope_cos = rope; p.rope_sin = rope + (size_t)frames * 32;
  } else {
    p.heads = Nc / 64; p.vt = ob; p.npad = pitch;
  }
  HIP_TRY(launch_gemm_bf16_tile(cons_epi, p, cons_tile, s));
  hipLaunchKernelGGL(widen_kernel, dim3(2048), dim3(256), 0, s, ob, y, nout);
  HIP_TRY(hipGetLastError());
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

int lemas_k_outlier_rows(const float* A, const float* Wside, const float* bias_side, const int32_t* chan, int32_t nf, const float* gate,
                         const int32_t* seq_len, float* x, int32_t batch, int32_t frames, int32_t pitch, int32_t K, int32_t ldx, uint8_t* a8, uint8_t* amx,
                         void* stream) {
#ifndef LEMAS_MEASUREMENT_BUILD
  set_error("lemas_k_outlier_rows: csrc/outlier_rows.hip is compiled into measurement builds only (-DLEMAS_MEASUREMENT_BUILD)");
  return LEMAS_E_STATE;
#else
  hipStream_t s = (hipStream_t)stream;
  Scratch sc;
  const int M = batch * pitch;
  bf16_t* a = sc.get<bf16_t>((size_t)M * K);
  bf16_t* w = sc.get<bf16_t>((size_t)32 * K);
  float* b = sc.get<float>(32);
  int* ch = sc.get<int>(32);
  int* step = sc.get<int>(1);
  if (!a || !w || !b || !ch || !step) { set_error("lemas_k_outlier_rows: out of memory"); return LEMAS_E_STATE; }
  if (nf < 1 || nf > 32) { set_error("lemas_k_outlier_rows: nf is 1 .. 32"); return LEMAS_E_ARG; }
  HIP_TRY(launch_f32_to_bf16(A, a, (size_t)M * K, s));
  HIP_TRY(launch_f32_to_bf16(Wside, w, (size_t)nf * K, s));      // rows past nf stay zero (Scratch zero-fills)
  HIP_TRY(hipMemcpyAsync(b, bias_side, (size_t)nf * 4, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemsetAsync(ch, 0xff, 32 * 4, s));
  HIP_TRY(hipMemcpyAsync(ch, chan, (size_t)nf * 4, hipMemcpyDeviceToDevice, s));
  OutlierRowsParams o{};
  o.A = a; o.W = w; o.bias = b; o.chan = ch; o.nf = nf; o.x = x; o.ldx = ldx; o.M = M; o.K = K;
  o.tab = gate; o.tab_stride = 0; o.gate_off = 0; o.step_idx = step; o.kv_len = seq_len; o.seq_pitch = pitch; o.seq_valid = frames; o.batch = batch;
  o.a8 = a8; o.amx = amx;
  HIP_TRY(launch_outlier_rows(o, s));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
#endif
}

int lemas_k_linear_f8(const float* A, const float* W, const float* bias, float* out, int32_t M, int32_t N, int32_t K, int32_t act,
                      uint8_t* out8, uint8_t* outmx, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  RC_TRY(kernels_init());
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

// attention by variant: the product kernel (attention.hip), or -- variants with bit ATTN_Q64 -- the 64-queries-per-wave measurement kernel of
// attention_q64.hip, which is compiled into THIS library only (no engine can select it)
static hipError_t launch_attention_any(const AttnParams& p, hipStream_t s) {
  if ((p.variant & ATTN_Q64) == 0) return launch_attention(p, s);
  if ((p.variant & 16) == 0 || (p.variant & ~(ATTN_Q64 | 16 | 3)) != 0) return hipErrorInvalidValue;
  static std::once_flag once;
  static hipError_t init_err = hipSuccess;
  std::call_once(once, []() { init_err = attention_q64_init(); });     // > 64 KB dynamic LDS opt-in (one device per test process)
  if (init_err != hipSuccess) return init_err;
  return launch_attention_q64(p, s);
}

int lemas_k_attention(const float* q, const float* k, const float* v, const int32_t* seq_len, float* out, int32_t B,
                      int32_t H, int32_t N, void* stream) {
  return lemas_k_attention_variant(q, k, v, seq_len, out, B, H, N, 0, stream);
}

int lemas_k_attention_variant(const float* q, const float* k, const float* v, const int32_t* seq_len, float* out, int32_t B,
                              int32_t H, int32_t N, int32_t variant, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Scratch sc;
  const int pitch = (N + 127) & ~127, npad = pitch;
  const size_t np = (size_t)B * H * pitch * 64;
  bf16_t* qb = sc.get<bf16_t>(np);
  bf16_t* kb = sc.get<bf16_t>(np);
  bf16_t* vt = sc.get<bf16_t>((size_t)B * H * 64 * npad);
  bf16_t* ob = sc.get<bf16_t>(np);
  if (!qb || !kb || !vt || !ob) { set_error("lemas_k_attention: out of memory"); return LEMAS_E_STATE; }
  // the "prescaled q" variants (attention.hip VAR & 16) take q * softmax_scale * log2(e), as the QK GEMM epilogue hands it over
  hipLaunchKernelGGL(pad_rows_kernel, dim3(2048), dim3(256), 0, s, q, qb, B * H, N, pitch, (variant & 16) ? 0.125f * 1.4426950408889634f : 1.0f);
  hipLaunchKernelGGL(pad_rows_kernel, dim3(2048), dim3(256), 0, s, k, kb, B * H, N, pitch, 1.0f);
  hipLaunchKernelGGL(transpose_v_kernel, dim3(2048), dim3(256), 0, s, v, vt, B * H, N, npad);
  AttnParams p{};
  p.q = qb; p.k = kb; p.vt = vt; p.out = ob; p.kv_len = seq_len; p.b2 = B; p.batch = B; p.heads = H; p.n = N; p.npad = npad; p.pitch = pitch;
  p.scale = 0.125f; p.variant = variant;
  if (variant & ATTN_F8QK) {      // q, k quantised to MXFP8 by their own launch (in the engine: by the QK GEMM epilogue), then QK^T on the fp8 MFMA
    if ((variant & 17) != 17 || (variant & ~(ATTN_F8QK | 19)) != 0) { set_error("lemas_k_attention: the fp8 QK^T path is variant 8192 + 17 / 19"); return LEMAS_E_ARG; }
    const size_t rows = (size_t)B * H * pitch;
    uint8_t* q8 = sc.get<uint8_t>(rows * 64);
    uint8_t* k8 = sc.get<uint8_t>(rows * 64);
    uint8_t* qs = sc.get<uint8_t>(rows * 2);
    uint8_t* ks = sc.get<uint8_t>(rows * 2);
    if (!q8 || !k8 || !qs || !ks) { set_error("lemas_k_attention: out of memory"); return LEMAS_E_STATE; }
    HIP_TRY(launch_qk_mx8(qb, kb, q8, k8, qs, ks, rows, s));
    p.q8 = q8; p.k8 = k8; p.q8_mx = qs; p.k8_mx = ks;
    HIP_TRY(launch_attention(p, s));
  } else
  HIP_TRY(launch_attention_any(p, s));
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
  RC_TRY(kernels_init());
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
  const bool with_ln = w == "gemm_gate_ln";    // the gate + residual GEMM with its LayerNorm tail (N = 1024)
  if (with_ln) w = "gemm_gate";
  // "<name>_fold": the same launch as a ln-fold producer (gemm_gate) / consumer (gemm_gelu, gemm_qk, gemm_v); K or N = 1024 as the fold needs
  const bool fold = w.size() > 5 && w.compare(w.size() - 5, 5, "_fold") == 0;
  if (fold) w = w.substr(0, w.size() - 5);
  if (w == "gemm_gelu" || w == "gemm_gelu8" || w == "gemm_gate" || w == "gemm_qk" || w == "gemm_v" || w == "gemm_f32out") {
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
    const int epi = w == "gemm_gelu8" ? EPI_BIAS_GELU_F8 : w == "gemm_gelu" ? EPI_BIAS_GELU_BF16 : w == "gemm_gate" ? EPI_GATE_RES : w == "gemm_qk" ? EPI_QK_ROPE : w == "gemm_v" ? EPI_V_T : EPI_BIAS_F32;
    if ((epi == EPI_QK_ROPE && N != 2048) || (epi == EPI_V_T && N != 1024)) { set_error("bench: gemm_qk needs N = 2048, gemm_v N = 1024"); return LEMAS_E_ARG; }
    if (fold) {
      float* tabf = sc.get<float>((size_t)4096 + 2 * Np);
      float* part = sc.get<float>((size_t)M * 64);
      bf16_t* xs = sc.get<bf16_t>((size_t)M * 1024);
      if (!tabf || !part || !xs) { set_error("bench: out of memory"); return LEMAS_E_STATE; }
      p.tab = tabf; p.ln_np = 32;
      if (epi == EPI_GATE_RES) { p.xs_out = xs; p.xs_scale_off = 1024; p.ln_part_out = part; }
      else { p.ln_part = part; p.lnc1_off = 4096; p.lnc2_off = 4096 + Np; }
    }
    unsigned int* err_host = nullptr;
    if (with_ln) {
      if (f8 || N != 1024) { set_error("bench: gemm_gate_ln is the bf16 N = 1024 launch"); return LEMAS_E_ARG; }
      float* tab3 = sc.get<float>(3 * 1024);
      bf16_t* lnout = sc.get<bf16_t>((size_t)M * 1024);
      unsigned int* cnt = sc.get<unsigned int>((size_t)M / 64 + 1);
      if (!tab3 || !lnout || !cnt) { set_error("bench: out of memory"); return LEMAS_E_STATE; }
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&err_host), 64, hipHostMallocMapped));
      *err_host = 0;
      unsigned int* err_dev = nullptr;
      HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&err_dev), err_host, 0));
      p.tab = tab3; p.ln_out = lnout; p.ln_scale_off = 1024; p.ln_shift_off = 2048; p.ln_cnt = cnt; p.ln_err = err_dev;
      const size_t cb = ((size_t)M / 64 + 1) * 4;
      rc = time_it([&]() { hipError_t e = hipMemsetAsync(cnt, 0, cb, s); return e != hipSuccess ? e : launch_gemm_bf16_tile(epi, p, variant, s); });
      if (rc == 0 && *err_host) { set_error("bench: a LayerNorm tail gave up"); rc = LEMAS_E_STATE; }
      (void)hipHostFree(err_host);
    } else
    rc = time_it([&]() { return launch_gemm_bf16_tile(epi, p, variant, s); });
#ifdef LEMAS_PHASE_TIMESTAMPS
    if (rc == 0 && variant >= 16 && !with_ln) {   // phase timestamps of one more launch
      const int bm_ = variant == 17 || variant == 18 ? 128 : variant == 19 ? 64 : 256, bn_ = variant == 22 ? 256 : variant == 16 || variant == 17 ? 128 : 64;
      const int grid = ((M + bm_ - 1) / bm_) * (Np / bn_);
      unsigned long long* d = sc.get<unsigned long long>((size_t)grid * 4);
      p.dbg = d;
      (void)launch_gemm_bf16_tile(epi, p, variant, s);
      HIP_TRY(hipStreamSynchronize(s));
      std::vector<unsigned long long> h((size_t)grid * 4);
      HIP_TRY(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull, t3 = 0;
      for (int i = 0; i < grid; ++i) { t0 = std::min(t0, h[i * 4]); t3 = std::max(t3, h[i * 4 + 3]); }
      double sp[2] = {0, 0}, sl[2] = {0, 0}, se[2] = {0, 0}, st[2] = {0, 0}; int cnt[2] = {0, 0};
      for (int i = 0; i < grid; ++i) {
        const int g = i < 256 ? 0 : 1;
        sp[g] += (h[i * 4 + 1] - h[i * 4]) * 0.01; sl[g] += (h[i * 4 + 2] - h[i * 4 + 1]) * 0.01; se[g] += (h[i * 4 + 3] - h[i * 4 + 2]) * 0.01;
        st[g] += (h[i * 4] - t0) * 0.01; cnt[g]++;
      }
      fprintf(stderr, "  phases %s M=%d N=%d K=%d grid=%d span %.1f us | first round: start +%.1f prologue %.1f loop %.1f epilogue %.1f | later: n=%d start +%.1f prologue %.1f loop %.1f epilogue %.1f\n",
              w.c_str(), M, N, K, grid, (t3 - t0) * 0.01, st[0] / std::max(cnt[0], 1), sp[0] / std::max(cnt[0], 1), sl[0] / std::max(cnt[0], 1), se[0] / std::max(cnt[0], 1),
              cnt[1], st[1] / std::max(cnt[1], 1), sp[1] / std::max(cnt[1], 1), sl[1] / std::max(cnt[1], 1), se[1] / std::max(cnt[1], 1));
    }
#endif
  } else if (w == "attention" || w == "attention_qkquant") {
    // M = sequence length, N = batch*heads
    const bool quant_only = w == "attention_qkquant";
    if (quant_only && !(variant & ATTN_F8QK)) { set_error("bench: attention_qkquant needs variant bit 8192"); return LEMAS_E_ARG; }
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
    p.q = q; p.k = k; p.vt = vt; p.out = o; p.kv_len = nullptr; p.b2 = bh / 16; p.batch = bh / 16; p.heads = 16; p.n = n; p.npad = npad; p.pitch = pitch;
    p.scale = 0.125f; p.variant = variant;
    if (variant & ATTN_F8QK) {     // "attention_qkquant" times the quantising launch alone, "attention" the kernel on quantised rows
      const size_t rows = (size_t)bh * pitch;
      uint8_t* q8 = sc.get<uint8_t>(rows * 64);
      uint8_t* k8 = sc.get<uint8_t>(rows * 64);
      uint8_t* qs = sc.get<uint8_t>(rows * 2);
      uint8_t* ks = sc.get<uint8_t>(rows * 2);
      if (!q8 || !k8 || !qs || !ks) { set_error("bench: out of memory"); return LEMAS_E_STATE; }
      HIP_TRY(launch_qk_mx8(q, k, q8, k8, qs, ks, rows, s));
      p.q8 = q8; p.k8 = k8; p.q8_mx = qs; p.k8_mx = ks;
      if (quant_only) rc = time_it([&]() { return launch_qk_mx8(q, k, q8, k8, qs, ks, rows, s); });
      else rc = time_it([&]() { return launch_attention(p, s); });
    } else
    rc = time_it([&]() { return launch_attention_any(p, s); });
#ifdef LEMAS_PHASE_TIMESTAMPS
    if (rc == 0 && !(variant & ATTN_F8QK)) {   // phase timestamps of one more launch
      const int grid = ((n + 127) / 128) * bh;
      unsigned long long* d = sc.get<unsigned long long>((size_t)grid * 4);
      p.dbg = d;
      (void)launch_attention_any(p, s);
      HIP_TRY(hipStreamSynchronize(s));
      std::vector<unsigned long long> h((size_t)grid * 4);
      HIP_TRY(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull, t3 = 0;
      for (int i = 0; i < grid; ++i) { t0 = std::min(t0, h[i * 4]); t3 = std::max(t3, h[i * 4 + 3]); }
      double sp = 0, sl = 0, se = 0, st = 0;
      for (int i = 0; i < grid; ++i) {
        sp += (h[i * 4 + 1] - h[i * 4]) * 0.01; sl += (h[i * 4 + 2] - h[i * 4 + 1]) * 0.01; se += (h[i * 4 + 3] - h[i * 4 + 2]) * 0.01; st += (h[i * 4] - t0) * 0.01;
      }
      fprintf(stderr, "  phases attention N=%d BH=%d grid=%d span %.1f us | start +%.1f prologue (Q fragments) %.1f loop %.1f merge + epilogue %.1f\n",
              n, bh, grid, (t3 - t0) * 0.01, st / grid, sp / grid, sl / grid, se / grid);
    }
#endif
  } else if (w == "ln_mod") {
    // M rows of D = N fp32 -> bf16 (the AdaLN-modulated LayerNorm in front of QKV and FF1)
    float* x = sc.get<float>((size_t)M * N);
    bf16_t* o = sc.get<bf16_t>((size_t)M * N);
    float* tab = sc.get<float>((size_t)2 * N);
    int* step = sc.get<int>(16);
    if (!x || !o || !tab || !step) { set_error("bench: out of memory"); return LEMAS_E_STATE; }
    hipLaunchKernelGGL(fill_pattern_kernel, dim3(1024), dim3(256), 0, s, reinterpret_cast<bf16_t*>(x), (size_t)M * N * 2, 6u);
    rc = time_it([&]() { return launch_ln_mod(x, o, M, N, tab, 0, 0, N, step, s); });
  } else if (w == "gemm_qkv") {
    // the fused QK (N = 2048, RoPE epilogue) + V (N = 1024, transposed store) launch of a lane; K = 1024
    const int in = 1024, npad = (M + 127) & ~127;
    bf16_t* a = sc.get<bf16_t>((size_t)M * K);
    bf16_t* wt = sc.get<bf16_t>((size_t)3 * in * K);
    float* b = sc.get<float>(3 * in);
    bf16_t* q = sc.get<bf16_t>((size_t)M * in);
    bf16_t* k = sc.get<bf16_t>((size_t)M * in);
    bf16_t* vt = sc.get<bf16_t>((size_t)16 * 64 * npad);
    float* rc_ = sc.get<float>((size_t)M * 32);
    float* rs_ = sc.get<float>((size_t)M * 32);
    int* step = sc.get<int>(16);
    if (!a || !wt || !b || !q || !k || !vt || !rc_ || !rs_ || !step) { set_error("bench: out of memory"); return LEMAS_E_STATE; }
    hipLaunchKernelGGL(fill_pattern_kernel, dim3(1024), dim3(256), 0, s, a, (size_t)M * K, 1u);
    hipLaunchKernelGGL(fill_pattern_kernel, dim3(1024), dim3(256), 0, s, wt, (size_t)3 * in * K, 2u);
    GemmParams gq{};
    gq.A = a; gq.W = wt; gq.bias = b; gq.M = M; gq.N = 2 * in; gq.K = K; gq.n_valid = 2 * in; gq.ldc = 2 * in; gq.step_idx = step;
    gq.seq_pitch = M; gq.seq_valid = M; gq.batch = 1; gq.heads = 16; gq.npad = npad; gq.q = q; gq.k = k; gq.vt = vt; gq.rope_cos = rc_; gq.rope_sin = rs_;
    GemmParams gv = gq;
    gv.W = wt + (size_t)2 * in * K; gv.bias = b + 2 * in; gv.N = in; gv.n_valid = in; gv.ldc = in;
    gq.tile = gv.tile = variant;
    rc = time_it([&]() { return launch_gemm_qkv_fused(gq, gv, s); });
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

