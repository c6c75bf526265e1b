// lemas_prosody: the prompt's global prosody embedding, behind the C ABI.
//   lemas_prosody_fbank   <- extract_fbank_16k, lemas_tts/model/backbones/prosody_encoder.py:334-361 (kaldi fbank, 80 bins, 16 kHz;
//                            third party arithmetic, PARITY UNPINNED -- oracle/prosody_oracle.py kaldi_fbank_80)
//   lemas_prosody_encode  <- ProsodyEncoder.forward / ECAPA_TDNN.forward, prosody_encoder.py:103-133, called per sample with
//                            padding_mask=None at lemas_tts/model/cfm.py:248-262
// One utterance = ~1000 frames x 7 M parameters (~10 GFLOP): everything is exact fp32 (f32 MFMA GEMM + row/column kernels),
// time-major [T][C] activations, dilated convolutions as im2col + GEMM on the reference's own [out][in][k] weights.
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "engine_common.h"

using namespace lemas;

struct lemas_prosody {
  lemas_prosody_config cfg{};
  WeightStore ws;
  bool finalized = false;
  // fbank constants
  DevBuf basis, banks;                 // [ldk][512], [80][ldp]
  DevBuf d_frames, d_spec, d_pow;
  // encoder workspaces
  DevBuf d_col, d_a, d_b, d_r, d_tmp, d_cat, d_m, d_att1, d_att2, d_vec, d_small;
  DevBuf d_r2w;                        // lane-major weight images of the Res2Net chunks the fused step kernel takes (finalize())
  std::map<std::string, const float*> r2w;
  int nb = 257, ldk = 516, ldp = 260;

  ~lemas_prosody() {
    for (DevBuf* b : {&basis, &banks, &d_frames, &d_spec, &d_pow, &d_col, &d_a, &d_b, &d_r, &d_tmp, &d_cat, &d_m, &d_att1, &d_att2,
                      &d_vec, &d_small, &d_r2w})
      b->release();
    ws.release();
  }
  int nblk() const { return cfg.n_layers; }

  void tdnn_schema(const std::string& p, int cin, int cout, int k) {
    ws.declare(p + "conv.weight", {cout, cin, k});
    ws.declare(p + "conv.bias", {cout});
    ws.declare(p + "norm.weight", {cout});
    ws.declare(p + "norm.bias", {cout});
  }
  void declare_schema() {
    const int* ch = cfg.channels;
    const int L = cfg.n_layers;
    tdnn_schema("blocks.0.", cfg.input_dim, ch[0], cfg.kernel_sizes[0]);
    for (int i = 1; i < L - 1; ++i) {
      const std::string p = "blocks." + std::to_string(i) + ".";
      tdnn_schema(p + "tdnn1.", ch[i - 1], ch[i], 1);
      const int sub = ch[i] / cfg.res2net_scale;
      for (int j = 0; j < cfg.res2net_scale - 1; ++j) tdnn_schema(p + "res2net_block.blocks." + std::to_string(j) + ".", sub, sub, cfg.kernel_sizes[i]);
      tdnn_schema(p + "tdnn2.", ch[i], ch[i], 1);
      ws.declare(p + "se_block.conv1.weight", {cfg.se_channels, ch[i], 1});
      ws.declare(p + "se_block.conv1.bias", {cfg.se_channels});
      ws.declare(p + "se_block.conv2.weight", {ch[i], cfg.se_channels, 1});
      ws.declare(p + "se_block.conv2.bias", {ch[i]});
      if (ch[i - 1] != ch[i]) {
        ws.declare(p + "shortcut.weight", {ch[i], ch[i - 1], 1});
        ws.declare(p + "shortcut.bias", {ch[i]});
      }
    }
    tdnn_schema("mfa.", ch[L - 1], ch[L - 1], cfg.kernel_sizes[L - 1]);
    tdnn_schema("asp.tdnn.", ch[L - 1] * (cfg.global_context ? 3 : 1), cfg.attention_channels, 1);
    ws.declare("asp.conv.weight", {ch[L - 1], cfg.attention_channels, 1});
    ws.declare("asp.conv.bias", {ch[L - 1]});
    ws.declare("asp_norm.weight", {2 * ch[L - 1]});
    ws.declare("asp_norm.bias", {2 * ch[L - 1]});
    ws.declare("fc.weight", {cfg.embed_dim, 2 * ch[L - 1], 1});
    ws.declare("fc.bias", {cfg.embed_dim});
  }

  int init_fbank() {
    // kaldi mel banks (torchaudio.compliance.kaldi.get_mel_banks: 80 bins, 20 Hz .. Nyquist, 512-point FFT, Nyquist column 0)
    const int bins = 80, padded = 512;
    nb = padded / 2 + 1; ldk = (2 * nb + 3) & ~3; ldp = (nb + 3) & ~3;
    std::vector<float> fb((size_t)bins * ldp, 0.f);
    auto mel = [](double f) { return 1127.0 * std::log(1.0 + f / 700.0); };
    const double sr = 16000.0, width = sr / padded, ml = mel(20.0), mh = mel(0.5 * sr), delta = (mh - ml) / (bins + 1);
    for (int b = 0; b < bins; ++b) {
      const double left = ml + b * delta, center = left + delta, right = center + delta;
      for (int i = 0; i < padded / 2; ++i) {
        const double m = mel(width * i);
        if (m > left && m < right) fb[(size_t)b * ldp + i] = (float)(m <= center ? (m - left) / (center - left) : (right - m) / (right - center));
      }
    }
    RC_TRY(banks.ensure(fb.size() * 4));
    HIP_TRY(hipMemcpy(banks.p, fb.data(), fb.size() * 4, hipMemcpyHostToDevice));
    RC_TRY(basis.ensure((size_t)ldk * padded * 4));
    HIP_TRY(launch_rdft_basis(padded, ldk, basis.as<float>(), nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return 0;
  }

  int finalize() {
    RC_TRY(ws.check_complete());
    if (!basis.p) RC_TRY(init_fbank());
    // Res2Net chunks at the fused kernel's shape: their weights once more, lane-major
    r2w.clear();
    std::vector<std::string> names;
    for (int i = 1; i < cfg.n_layers - 1; ++i) {
      const int sub = cfg.channels[i] / cfg.res2net_scale;
      if (!res2net_step_fits(sub, sub, cfg.kernel_sizes[i], cfg.dilations[i])) continue;
      for (int j = 0; j < cfg.res2net_scale - 1; ++j) names.push_back("blocks." + std::to_string(i) + ".res2net_block.blocks." + std::to_string(j) + ".");
    }
    if (!names.empty()) {
      const size_t n = res2net_weight_image_floats();
      RC_TRY(d_r2w.ensure(names.size() * n * 4));
      for (size_t q = 0; q < names.size(); ++q) {
        float* img = d_r2w.as<float>() + q * n;
        HIP_TRY(launch_res2net_weight_image(ws.ptr(names[q] + "conv.weight"), img, nullptr));
        r2w[names[q]] = img;
      }
      HIP_TRY(hipStreamSynchronize(nullptr));
    }
    finalized = true;
    return 0;
  }

  static int64_t fbank_frames(int64_t samples) { return samples < 400 ? 0 : 1 + (samples - 400) / 160; }

  int fbank(const float* wav, int n, float* out, hipStream_t s) {
    if (!basis.p) RC_TRY(init_fbank());
    const int frames = (int)fbank_frames(n);
    if (!wav || !out || frames <= 0) { set_error("lemas_prosody_fbank: needs >= 400 samples at 16 kHz (got %d); tile short prompts first", n); return LEMAS_E_ARG; }
    RC_TRY(d_frames.ensure((size_t)frames * 512 * 4));
    RC_TRY(d_spec.ensure((size_t)frames * ldk * 4));
    RC_TRY(d_pow.ensure((size_t)frames * ldp * 4));
    HIP_TRY(launch_kaldi_frames(wav, n, frames, 400, 160, 512, 0.97f, d_frames.as<float>(), s));
    GemmF32Params g{};
    g.A = d_frames.as<float>(); g.lda = 512; g.W = basis.as<float>(); g.ldw = 512; g.out = d_spec.as<float>(); g.ldc = ldk; g.M = frames; g.N = ldk; g.K = 512;
    HIP_TRY(launch_gemm_f32(F32_BIAS, g, s));
    HIP_TRY(launch_power(d_spec.as<float>(), frames, nb, ldk, ldp, d_pow.as<float>(), s));
    GemmF32Params h{};
    h.A = d_pow.as<float>(); h.lda = ldp; h.W = banks.as<float>(); h.ldw = ldp; h.out = out; h.ldc = 80; h.M = frames; h.N = 80; h.K = ldp;
    HIP_TRY(launch_gemm_f32(F32_BIAS, h, s));
    HIP_TRY(launch_log_clamp(out, (size_t)frames * 80, 1.1920928955078125e-07f, s));
    return 0;
  }

  // a Linear on ONE row (SE gates, global-context bias, fc): the wave-per-output kernel when the operands allow its float4 reads
  static hipError_t row_linear(int epi, const GemmF32Params& g, hipStream_t s) {
    if (((uintptr_t)g.A & 15) || ((uintptr_t)g.W & 15) || (g.ldw & 3) || (g.K & 3)) return launch_gemm_f32(epi, g, s);
    return launch_gemv_f32(epi, g, s);
  }

  static bool add_ok_fused(const float* x, const float* add, int ldx, int ldadd) {
    return !((uintptr_t)x & 15) && !((uintptr_t)add & 15) && !(ldx & 3) && (!add || !(ldadd & 3));
  }
  // y = LayerNorm(relu(conv_k,dil(x (+ add)))) ; x [T][ldx] with cin channels -> out [T][ldo] with cout channels
  int tdnn(const std::string& p, const float* x, int ldx, const float* add, int ldadd, int T, int cin, int cout, int k, int dil, float* scratch,
           float* out, int ldo, int act_tanh, const float* bias_override, int w_cols, hipStream_t s) {
    const auto img = r2w.find(p);
    if (img != r2w.end() && add_ok_fused(x, add, ldx, ldadd) && !bias_override && !w_cols && !act_tanh && res2net_step_fits(cin, cout, k, dil)) {
      HIP_TRY(launch_res2net_step(x, ldx, add, ldadd, T, dil, img->second, ws.ptr(p + "conv.bias"), ws.ptr(p + "norm.weight"),
                                  ws.ptr(p + "norm.bias"), 1e-12f, out, ldo, s));
      return 0;
    }
    GemmF32Params g{};
    g.W = ws.ptr(p + "conv.weight"); g.ldw = w_cols ? w_cols : cin * k; g.bias = bias_override ? bias_override : ws.ptr(p + "conv.bias");
    g.M = T; g.N = cout; g.out = scratch; g.ldc = cout;
    if (k == 1 && !add) {
      g.A = x; g.lda = ldx; g.K = cin;
    } else {
      HIP_TRY(launch_im2col_dil(x, ldx, add, ldadd, T, cin, k, dil, d_col.as<float>(), s));
      g.A = d_col.as<float>(); g.lda = cin * k; g.K = cin * k;
    }
    HIP_TRY(launch_gemm_f32(F32_BIAS_RELU, g, s));
    HIP_TRY(launch_ln_rows(scratch, cout, T, cout, ws.ptr(p + "norm.weight"), ws.ptr(p + "norm.bias"), 1e-12f, act_tanh, out, ldo, s));
    return 0;
  }

  int encode(const float* fb, int T, float* emb, hipStream_t s) {
    if (!finalized) { set_error("lemas_prosody_encode: weights not finalized"); return LEMAS_E_STATE; }
    if (!fb || !emb || T <= 0) { set_error("lemas_prosody_encode: bad arguments (frames=%d)", T); return LEMAS_E_ARG; }
    const int L = cfg.n_layers;
    const int* ch = cfg.channels;
    int cmax = cfg.input_dim * cfg.kernel_sizes[0], ccat = 0, cwide = ch[L - 1];
    for (int i = 1; i < L - 1; ++i) {
      ccat += ch[i];
      cmax = std::max(cmax, ch[i] / cfg.res2net_scale * cfg.kernel_sizes[i]);
      cwide = std::max(cwide, ch[i]);
    }
    if (ccat != ch[L - 1]) { set_error("lemas_prosody_encode: sum of the SE-Res2Net widths (%d) must equal the last width (%d)", ccat, ch[L - 1]); return LEMAS_E_ARG; }
    cwide = std::max(cwide, ch[0]);
    RC_TRY(d_col.ensure((size_t)T * cmax * 4));
    RC_TRY(d_a.ensure((size_t)T * cwide * 4));
    RC_TRY(d_b.ensure((size_t)T * cwide * 4));
    RC_TRY(d_r.ensure((size_t)T * cwide * 4));
    RC_TRY(d_tmp.ensure((size_t)T * cwide * 4));
    RC_TRY(d_cat.ensure((size_t)T * ccat * 4));
    RC_TRY(d_m.ensure((size_t)T * ch[L - 1] * 4));
    RC_TRY(d_att1.ensure((size_t)T * cfg.attention_channels * 4));
    RC_TRY(d_att2.ensure((size_t)T * ch[L - 1] * 4));
    RC_TRY(d_vec.ensure((size_t)4 * ch[L - 1] * 4));
    RC_TRY(d_small.ensure((size_t)(cwide + cfg.se_channels + cwide + cfg.attention_channels + cfg.embed_dim) * 4));
    float *A = d_a.as<float>(), *Bf = d_b.as<float>(), *R = d_r.as<float>(), *tmp = d_tmp.as<float>(), *cat = d_cat.as<float>();

    // blocks[0]: TDNN(input_dim -> ch0)                                                        prosody_encoder.py:59-67
    RC_TRY(tdnn("blocks.0.", fb, cfg.input_dim, nullptr, 0, T, cfg.input_dim, ch[0], cfg.kernel_sizes[0], cfg.dilations[0], tmp, A, ch[0], 0, nullptr, 0, s));
    const float* xin = A;
    int ldin = ch[0], off = 0;
    for (int i = 1; i < L - 1; ++i) {                                                         // SERes2NetBlock :283-331
      const std::string p = "blocks." + std::to_string(i) + ".";
      const int c = ch[i], sub = c / cfg.res2net_scale, k = cfg.kernel_sizes[i], dil = cfg.dilations[i];
      RC_TRY(tdnn(p + "tdnn1.", xin, ldin, nullptr, 0, T, ch[i - 1], c, 1, 1, tmp, Bf, c, 0, nullptr, 0, s));
      // Res2Net: chunk 0 passes through, chunk j = TDNN(x_j (+ y_{j-1}))                       :188-202
      HIP_TRY(launch_copy_cols(Bf, c, T, sub, R, c, s));
      for (int j = 1; j < cfg.res2net_scale; ++j)
        RC_TRY(tdnn(p + "res2net_block.blocks." + std::to_string(j - 1) + ".", Bf + j * sub, c, j >= 2 ? R + (j - 1) * sub : nullptr, c, T, sub, sub, k, dil,
                    tmp, R + j * sub, c, 0, nullptr, 0, s));
      RC_TRY(tdnn(p + "tdnn2.", R, c, nullptr, 0, T, c, c, 1, 1, tmp, Bf, c, 0, nullptr, 0, s));
      // SE gate (padding_mask=None: plain mean over time)                                      :224-230
      float* sm = d_small.as<float>();
      float *mean = sm, *s1 = sm + cwide, *s2 = s1 + cfg.se_channels;
      HIP_TRY(launch_col_stats(Bf, c, T, c, 0.f, mean, nullptr, s));
      GemmF32Params g{};
      g.A = mean; g.lda = c; g.W = ws.ptr(p + "se_block.conv1.weight"); g.ldw = c; g.bias = ws.ptr(p + "se_block.conv1.bias"); g.out = s1; g.ldc = cfg.se_channels;
      g.M = 1; g.N = cfg.se_channels; g.K = c;
      HIP_TRY(row_linear(F32_BIAS_RELU, g, s));
      g.A = s1; g.lda = cfg.se_channels; g.W = ws.ptr(p + "se_block.conv2.weight"); g.ldw = cfg.se_channels; g.bias = ws.ptr(p + "se_block.conv2.bias"); g.out = s2;
      g.ldc = c; g.N = c; g.K = cfg.se_channels;
      HIP_TRY(row_linear(F32_BIAS_SIGMOID, g, s));
      const float* res = xin;
      int ldr = ldin;
      if (ch[i - 1] != c) {                                                                    // projection shortcut :318-324
        GemmF32Params h{};
        h.A = xin; h.lda = ldin; h.W = ws.ptr(p + "shortcut.weight"); h.ldw = ch[i - 1]; h.bias = ws.ptr(p + "shortcut.bias"); h.out = tmp; h.ldc = c;
        h.M = T; h.N = c; h.K = ch[i - 1];
        HIP_TRY(launch_gemm_f32(F32_BIAS, h, s));
        res = tmp; ldr = c;
      }
      HIP_TRY(launch_scale_cols_add(Bf, c, s2, res, ldr, T, c, cat + off, ccat, s));         // block output lands in its slice of the concat (:123)
      xin = cat + off; ldin = ccat; off += c;
    }
    // MFA TDNN over the concatenated block outputs                                              :123-124
    const int cl = ch[L - 1];
    float* M = d_m.as<float>();
    RC_TRY(tdnn("mfa.", cat, ccat, nullptr, 0, T, ccat, cl, cfg.kernel_sizes[L - 1], cfg.dilations[L - 1], d_att2.as<float>(), M, cl, 0, nullptr, 0, s));
    // attentive statistics pooling                                                              :245-280
    float* vec = d_vec.as<float>();            // [mean | std | mean2 | std2]
    float* sm = d_small.as<float>();
    float* bias2 = sm + 2 * cwide + cfg.se_channels;
    const float* tb = nullptr;
    int wcols = 0;
    if (cfg.global_context) {
      HIP_TRY(launch_col_stats(M, cl, T, cl, 1e-12f, vec, vec + cl, s));
      // the [x, mean, std] concat is never built: its mean/std columns are constant over time and fold into the bias
      GemmF32Params g{};
      g.A = vec; g.lda = 2 * cl; g.W = ws.ptr("asp.tdnn.conv.weight") + cl; g.ldw = 3 * cl; g.bias = ws.ptr("asp.tdnn.conv.bias"); g.out = bias2;
      g.ldc = cfg.attention_channels; g.M = 1; g.N = cfg.attention_channels; g.K = 2 * cl;
      HIP_TRY(row_linear(F32_BIAS, g, s));
      tb = bias2; wcols = 3 * cl;
    }
    RC_TRY(tdnn("asp.tdnn.", M, cl, nullptr, 0, T, cl, cfg.attention_channels, 1, 1, tmp, d_att1.as<float>(), cfg.attention_channels, 1 /*tanh*/, tb, wcols, s));
    {
      GemmF32Params g{};
      g.A = d_att1.as<float>(); g.lda = cfg.attention_channels; g.W = ws.ptr("asp.conv.weight"); g.ldw = cfg.attention_channels; g.bias = ws.ptr("asp.conv.bias");
      g.out = d_att2.as<float>(); g.ldc = cl; g.M = T; g.N = cl; g.K = cfg.attention_channels;
      HIP_TRY(launch_gemm_f32(F32_BIAS, g, s));
    }
    HIP_TRY(launch_softmax_pool(d_att2.as<float>(), cl, M, cl, T, cl, 1e-12f, vec + 2 * cl, vec + 3 * cl, s));
    // asp_norm -> fc -> L2 normalise                                                            :127-133
    HIP_TRY(launch_ln_rows(vec + 2 * cl, 2 * cl, 1, 2 * cl, ws.ptr("asp_norm.weight"), ws.ptr("asp_norm.bias"), 1e-12f, 0, vec, 2 * cl, s));
    float* raw = sm + 2 * cwide + cfg.se_channels + cfg.attention_channels;
    {
      GemmF32Params g{};
      g.A = vec; g.lda = 2 * cl; g.W = ws.ptr("fc.weight"); g.ldw = 2 * cl; g.bias = ws.ptr("fc.bias"); g.out = raw; g.ldc = cfg.embed_dim;
      g.M = 1; g.N = cfg.embed_dim; g.K = 2 * cl;
      HIP_TRY(row_linear(F32_BIAS, g, s));
    }
    HIP_TRY(launch_l2_normalize(raw, cfg.embed_dim, 1e-12f, emb, s));
    return 0;
  }
};

extern "C" {

int lemas_prosody_create(const lemas_prosody_config* cfg, lemas_prosody** out) {
  if (!cfg || !out) { set_error("lemas_prosody_create: null argument"); return LEMAS_E_ARG; }
  if (cfg->n_layers < 3 || cfg->n_layers > 8 || cfg->res2net_scale < 2) { set_error("lemas_prosody_create: bad architecture"); return LEMAS_E_ARG; }
  for (int i = 0; i < cfg->n_layers; ++i) {
    if (cfg->groups[i] != 1) { set_error("lemas_prosody_create: grouped TDNN convolutions are not built"); return LEMAS_E_ARG; }
    if ((cfg->kernel_sizes[i] & 1) == 0 || cfg->channels[i] % 4) { set_error("lemas_prosody_create: odd kernels / widths %% 4 == 0 only"); return LEMAS_E_ARG; }
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    set_error("lemas_prosody_create: no HIP device (this library has no CPU path)");
    return e != hipSuccess ? -(int)e : LEMAS_E_STATE;
  }
  lemas_prosody* p = new lemas_prosody();
  p->cfg = *cfg;
  p->declare_schema();
  *out = p;
  return 0;
}
void lemas_prosody_destroy(lemas_prosody* p) { delete p; }
int lemas_prosody_load_weight(lemas_prosody* p, const char* name, const float* host, const int64_t* shape, int32_t ndim) {
  if (!p || !name || !host) return LEMAS_E_ARG;
  p->finalized = false;
  return p->ws.load(name, host, shape, ndim);
}
int lemas_prosody_finalize(lemas_prosody* p) { return p ? p->finalize() : LEMAS_E_ARG; }
int64_t lemas_prosody_fbank_frames(int64_t samples_16k) { return lemas_prosody::fbank_frames(samples_16k); }
int lemas_prosody_fbank(lemas_prosody* p, const float* wav16k, int32_t samples, float* fbank, void* stream) {
  if (!p) return LEMAS_E_ARG;
  return p->fbank(wav16k, samples, fbank, (hipStream_t)stream);
}
int lemas_prosody_encode(lemas_prosody* p, const float* fbank, int32_t frames, float* emb, void* stream) {
  if (!p) return LEMAS_E_ARG;
  return p->encode(fbank, frames, emb, (hipStream_t)stream);
}

}  // extern "C"
