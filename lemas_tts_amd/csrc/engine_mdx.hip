// lemas_mdx: the MDX-Net separation network of the UVR5 prompt denoiser, behind the C ABI.
//   lemas_mdx_create / load_weight / finalize  <- uvr5/multiprocess_cuda_infer.py:225-238 Inference.load_model: the reference builds an
//                                                 onnxruntime session from Kim_Vocal_1.onnx, an export of ConvTDFNet
//                                                 (uvr5/lib_v5/mdxnet.py:36-101); here the module's own state dict is loaded, strictly
//   lemas_mdx_forward                          <- model_run (multiprocess_cuda_infer.py:238, called at :269) = ConvTDFNet.forward,
//                                                 uvr5/lib_v5/mdxnet.py:103-127, with TFC / TFC_TDF of uvr5/lib_v5/modules.py:5-74
// Everything is exact fp32: 3x3 / 2x2 / transposed 2x2 convolutions as halo-staged implicit GEMMs (mdx_kernels.hip), the TDF linears as
// row-major GEMMs with the per-channel BatchNorm + ReLU (+ residual) in the epilogue (gemm_f32.hip F32_ROWAFF_*).  Inference BatchNorm is
// folded at finalize(): w' = w s, b' = s b + (beta - mean s), s = gamma / sqrt(var + eps).  The GroupNorm variant cannot be folded (its
// statistics depend on the data): the producers then write the raw sums and three small kernels normalise in place.
// Activations stay [b][c][t][f]; one forward at the Kim_Vocal_1 shape is 0.74 TFLOP per sample.
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "engine_common.h"

using namespace lemas;

namespace {

constexpr float kBnEps = 1e-5f, kGnEps = 1e-5f;

struct DevVec {
  DevBuf b;
  int upload(const std::vector<float>& v) {
    RC_TRY(b.ensure(std::max<size_t>(v.size(), 1) * 4));
    if (!v.empty()) HIP_TRY(hipMemcpy(b.p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    return 0;
  }
  const float* f() const { return b.as<float>(); }
};

struct NormP {            // GroupNorm affine (device) -- BatchNorm lives folded in the producer, or as scale / shift for the TDF epilogue
  DevVec gamma, beta, scale, shift;
};

// fp32 -> (bf16(x) << 16) | bf16(x - bf16(x)), both round-to-nearest-even: the split-operand format of mdx_conv3_bx_kernel
inline uint32_t bf16_rne(float x) {
  uint32_t u;
  std::memcpy(&u, &x, 4);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
inline uint32_t split_pack(float x) {
  const uint32_t h = bf16_rne(x), hb = h << 16;
  float hf;
  std::memcpy(&hf, &hb, 4);
  return hb | (bf16_rne(x - hf) & 0xFFFFu);
}

struct ConvP {
  int kind = 0, cin = 0, cout = 0, nchunks = 0, ntiles = 0;
  DevVec w, bias;
  NormP norm;
};

struct TdfP {
  bool present = false, single = false;
  int f = 0, h = 0, ldh = 0;
  DevVec w1, b1, w2, b2;
  bool has_bias = false;
  NormP n1, n2;
};

struct BlockP {
  int c = 0, T = 0, F = 0;
  std::vector<ConvP> convs;
  TdfP tdf;
};

}  // namespace

struct lemas_mdx {
  lemas_mdx_config cfg{};
  int n = 0;
  bool finalized = false;
  bool bf16x3 = false;      // option "bf16x3": the 3x3 convolutions on split-bf16 operands (mdx_kernels.hip), set before finalize()
  std::map<std::string, std::vector<int64_t>> schema;
  std::map<std::string, std::vector<float>> host;
  std::map<std::string, float*> taps;

  DevVec first_w, first_b, final_w, final_b;
  NormP first_norm;
  std::vector<BlockP> enc, dec;
  BlockP mid;
  std::vector<ConvP> ds, us;

  DevBuf S[3], hidden, gn_part, gn_stats;
  std::vector<DevBuf> skip;

  ~lemas_mdx() {
    // DevVec / DevBuf have no destructors: release everything reachable
    auto rn = [](NormP& q) { q.gamma.b.release(); q.beta.b.release(); q.scale.b.release(); q.shift.b.release(); };
    auto rc = [&](ConvP& c) { c.w.b.release(); c.bias.b.release(); rn(c.norm); };
    auto rb = [&](BlockP& b) {
      for (ConvP& c : b.convs) rc(c);
      for (DevVec* v : {&b.tdf.w1, &b.tdf.b1, &b.tdf.w2, &b.tdf.b2}) v->b.release();
      rn(b.tdf.n1); rn(b.tdf.n2);
    };
    for (DevVec* v : {&first_w, &first_b, &final_w, &final_b}) v->b.release();
    rn(first_norm);
    for (BlockP& b : enc) rb(b);
    for (BlockP& b : dec) rb(b);
    rb(mid);
    for (ConvP& c : ds) rc(c);
    for (ConvP& c : us) rc(c);
    for (DevBuf& b : S) b.release();
    for (DevBuf* b : {&hidden, &gn_part, &gn_stats}) b->release();
    for (DevBuf& b : skip) b.release();
  }

  bool group_norm() const { return cfg.norm == 1; }

  // ---- schema: the state-dict keys of ConvTDFNet (mdxnet.py:57-101; TFC_TDF modules.py:43-70) ----------------------------------
  void norm_schema(const std::string& p, int c) {
    schema[p + "weight"] = {c};
    schema[p + "bias"] = {c};
    if (!group_norm()) { schema[p + "running_mean"] = {c}; schema[p + "running_var"] = {c}; }
  }
  void block_schema(const std::string& p, int c, int f) {
    for (int j = 0; j < cfg.l; ++j) {
      const std::string q = p + "tfc.H." + std::to_string(j) + ".";
      schema[q + "0.weight"] = {c, c, cfg.k, cfg.k};
      schema[q + "0.bias"] = {c};
      norm_schema(q + "1.", c);
    }
    if (cfg.bn < 0) return;
    const int h = cfg.bn == 0 ? f : f / cfg.bn;
    schema[p + "tdf.0.weight"] = {h, f};
    if (cfg.bias) schema[p + "tdf.0.bias"] = {h};
    norm_schema(p + "tdf.1.", c);
    if (cfg.bn != 0) {
      schema[p + "tdf.3.weight"] = {f, h};
      if (cfg.bias) schema[p + "tdf.3.bias"] = {f};
      norm_schema(p + "tdf.4.", c);
    }
  }
  void declare_schema() {
    const int g = cfg.g;
    schema["first_conv.0.weight"] = {g, cfg.dim_c, 1, 1};
    schema["first_conv.0.bias"] = {g};
    norm_schema("first_conv.1.", g);
    int f = cfg.dim_f, c = g;
    for (int i = 0; i < n; ++i) {
      const std::string si = std::to_string(i);
      block_schema("encoding_blocks." + si + ".", c, f);
      schema["ds." + si + ".0.weight"] = {c + g, c, 2, 2};
      schema["ds." + si + ".0.bias"] = {c + g};
      norm_schema("ds." + si + ".1.", c + g);
      f /= 2; c += g;
    }
    block_schema("bottleneck_block.", c, f);
    for (int i = 0; i < n; ++i) {
      const std::string si = std::to_string(i);
      schema["us." + si + ".0.weight"] = {c, c - g, 2, 2};
      schema["us." + si + ".0.bias"] = {c - g};
      norm_schema("us." + si + ".1.", c - g);
      f *= 2; c -= g;
      block_schema("decoding_blocks." + si + ".", c, f);
    }
    schema["final_conv.0.weight"] = {cfg.dim_c, c, 1, 1};
    schema["final_conv.0.bias"] = {cfg.dim_c};
  }

  const std::vector<float>& H(const std::string& name) const { return host.at(name); }

  // per-channel (scale, shift) of the norm at prefix p: the folded inference BatchNorm, or (1, 0) for GroupNorm
  void norm_fold(const std::string& p, int c, std::vector<float>& sc, std::vector<float>& sh) const {
    sc.assign(c, 1.f); sh.assign(c, 0.f);
    if (group_norm()) return;
    const std::vector<float>&ga = H(p + "weight"), &be = H(p + "bias"), &mu = H(p + "running_mean"), &va = H(p + "running_var");
    for (int i = 0; i < c; ++i) {
      const float s = ga[i] / std::sqrt(va[i] + kBnEps);
      sc[i] = s; sh[i] = be[i] - mu[i] * s;
    }
  }
  int norm_upload(const std::string& p, int c, NormP& q, bool want_scale) const {
    if (group_norm()) { RC_TRY(q.gamma.upload(H(p + "weight"))); RC_TRY(q.beta.upload(H(p + "bias"))); }
    if (want_scale) {
      std::vector<float> sc, sh;
      norm_fold(p, c, sc, sh);
      RC_TRY(q.scale.upload(sc)); RC_TRY(q.shift.upload(sh));
    }
    return 0;
  }

  // conv weight `wname` ([co][ci][k][k], or [ci][co][2][2] for the transposed one) + norm at `nname` -> the kernel's K-slab layout
  int conv_build(ConvP& c, int kind, int cin, int cout, const std::string& wname, const std::string& bname, const std::string& nname) const {
    c.kind = kind; c.cin = cin; c.cout = cout;
    if (kind == MDX_CONV3 && bf16x3 && !group_norm()) return conv_build_bx(c, cin, cout, wname, bname, nname);
    const int taps = kind == MDX_CONV3 ? 9 : kind == MDX_DOWN2 ? 4 : 1;
    const int cols = kind == MDX_UP2 ? 4 * cout : cout;
    const int ciw = mdx_conv_ciw(kind);
    c.nchunks = (cin + 7) / 8; c.ntiles = (cols + 47) / 48;
    std::vector<float> sc, sh;
    norm_fold(nname, cout, sc, sh);
    const std::vector<float>&w = H(wname), &b = H(bname);
    std::vector<float> slab((size_t)c.ntiles * c.nchunks * 8 * ciw, 0.f), bias(cout);
    for (int co = 0; co < cout; ++co) bias[co] = sc[co] * b[co] + sh[co];
    for (int co = 0; co < cout; ++co)
      for (int ci = 0; ci < cin; ++ci)
        for (int tap = 0; tap < (kind == MDX_UP2 ? 4 : taps); ++tap) {
          float v;
          int col, ktap;
          if (kind == MDX_UP2) {
            v = w[((size_t)ci * cout + co) * 4 + tap];          // ConvTranspose2d weight [in][out][dt][df]
            col = co * 4 + tap; ktap = 0;
          } else {
            v = w[((size_t)co * cin + ci) * taps + tap];
            col = co; ktap = tap;
          }
          const int nt = col / 48, nn = col % 48, ch = ci / 8, c8 = ci % 8;
          slab[(((size_t)nt * c.nchunks + ch) * 8 + c8) * ciw + ktap * 48 + nn] = v * sc[co];
        }
    RC_TRY(c.w.upload(slab));
    RC_TRY(c.bias.upload(bias));
    RC_TRY(norm_upload(nname, cout, c.norm, false));
    return 0;
  }

  // the split-bf16 slab: [ntile][chunk of 16 ci][10 taps][2 channel blocks][48 co][8 ci] packed dwords, tap 9 zero
  int conv_build_bx(ConvP& c, int cin, int cout, const std::string& wname, const std::string& bname, const std::string& nname) const {
    c.kind = MDX_CONV3_BX;
    c.nchunks = (cin + 15) / 16; c.ntiles = (cout + 47) / 48;
    std::vector<float> sc, sh;
    norm_fold(nname, cout, sc, sh);
    const std::vector<float>&w = H(wname), &b = H(bname);
    const size_t per_chunk = (size_t)10 * 2 * 48 * 8;
    std::vector<float> slab((size_t)c.ntiles * c.nchunks * per_chunk, 0.f), bias(cout);
    uint32_t* bits = reinterpret_cast<uint32_t*>(slab.data());
    for (int co = 0; co < cout; ++co) bias[co] = sc[co] * b[co] + sh[co];
    for (int co = 0; co < cout; ++co)
      for (int ci = 0; ci < cin; ++ci)
        for (int tap = 0; tap < 9; ++tap) {
          const float v = w[((size_t)co * cin + ci) * 9 + tap] * sc[co];
          const int nt = co / 48, nn = co % 48, ch = ci / 16, cb = (ci % 16) / 8, k = ci % 8;
          bits[((size_t)nt * c.nchunks + ch) * per_chunk + (((size_t)tap * 2 + cb) * 48 + nn) * 8 + k] = split_pack(v);
        }
    RC_TRY(c.w.upload(slab));
    RC_TRY(c.bias.upload(bias));
    return 0;
  }

  int block_build(BlockP& blk, const std::string& p, int c, int T, int F) const {
    blk.c = c; blk.T = T; blk.F = F;
    blk.convs.resize(cfg.l);
    for (int j = 0; j < cfg.l; ++j) {
      const std::string q = p + "tfc.H." + std::to_string(j) + ".";
      RC_TRY(conv_build(blk.convs[j], MDX_CONV3, c, c, q + "0.weight", q + "0.bias", q + "1."));
    }
    TdfP& t = blk.tdf;
    t.present = cfg.bn >= 0;
    if (!t.present) return 0;
    t.single = cfg.bn == 0; t.f = F; t.h = t.single ? F : F / cfg.bn; t.ldh = (t.h + 3) & ~3; t.has_bias = cfg.bias != 0;
    {  // W1 [h][f] -> [ldh][f] with zero rows (the pad columns of the hidden tensor then meet zero columns of W2)
      std::vector<float> w1((size_t)t.ldh * F, 0.f), b1(t.ldh, 0.f);
      const std::vector<float>& src = H(p + "tdf.0.weight");
      std::memcpy(w1.data(), src.data(), src.size() * 4);
      if (t.has_bias) std::memcpy(b1.data(), H(p + "tdf.0.bias").data(), (size_t)t.h * 4);
      RC_TRY(t.w1.upload(w1)); RC_TRY(t.b1.upload(b1));
      RC_TRY(norm_upload(p + "tdf.1.", c, t.n1, true));
    }
    if (!t.single) {
      std::vector<float> w2((size_t)F * t.ldh, 0.f), b2(F, 0.f);
      const std::vector<float>& src = H(p + "tdf.3.weight");
      for (int r = 0; r < F; ++r) std::memcpy(&w2[(size_t)r * t.ldh], &src[(size_t)r * t.h], (size_t)t.h * 4);
      if (t.has_bias) std::memcpy(b2.data(), H(p + "tdf.3.bias").data(), (size_t)F * 4);
      RC_TRY(t.w2.upload(w2)); RC_TRY(t.b2.upload(b2));
      RC_TRY(norm_upload(p + "tdf.4.", c, t.n2, true));
    }
    return 0;
  }

  int finalize() {
    for (const auto& kv : schema)
      if (host.find(kv.first) == host.end()) { set_error("missing tensor '%s' (strict load)", kv.first.c_str()); return LEMAS_E_WEIGHT; }
    const int g = cfg.g;
    {  // first 1x1 convolution: fold its norm
      std::vector<float> sc, sh, w = H("first_conv.0.weight"), b = H("first_conv.0.bias");
      norm_fold("first_conv.1.", g, sc, sh);
      for (int co = 0; co < g; ++co) {
        for (int ci = 0; ci < cfg.dim_c; ++ci) w[(size_t)co * cfg.dim_c + ci] *= sc[co];
        b[co] = sc[co] * b[co] + sh[co];
      }
      RC_TRY(first_w.upload(w)); RC_TRY(first_b.upload(b));
      RC_TRY(norm_upload("first_conv.1.", g, first_norm, false));
    }
    enc.assign(n, BlockP()); dec.assign(n, BlockP()); ds.assign(n, ConvP()); us.assign(n, ConvP());
    int f = cfg.dim_f, T = cfg.dim_t, c = g;
    for (int i = 0; i < n; ++i) {
      const std::string si = std::to_string(i);
      RC_TRY(block_build(enc[i], "encoding_blocks." + si + ".", c, T, f));
      RC_TRY(conv_build(ds[i], MDX_DOWN2, c, c + g, "ds." + si + ".0.weight", "ds." + si + ".0.bias", "ds." + si + ".1."));
      f /= 2; T /= 2; c += g;
    }
    RC_TRY(block_build(mid, "bottleneck_block.", c, T, f));
    for (int i = 0; i < n; ++i) {
      const std::string si = std::to_string(i);
      RC_TRY(conv_build(us[i], MDX_UP2, c, c - g, "us." + si + ".0.weight", "us." + si + ".0.bias", "us." + si + ".1."));
      f *= 2; T *= 2; c -= g;
      RC_TRY(block_build(dec[i], "decoding_blocks." + si + ".", c, T, f));
    }
    RC_TRY(final_w.upload(H("final_conv.0.weight"))); RC_TRY(final_b.upload(H("final_conv.0.bias")));
    HIP_TRY(hipDeviceSynchronize());
    host.clear();                          // 70 MB of host copies are not needed any more
    finalized = true;
    return 0;
  }

  // ---- forward ------------------------------------------------------------------------------------------------------------------
  int tap(const char* name, const float* src, size_t floats, hipStream_t s) {
    auto it = taps.find(name);
    if (it == taps.end() || !it->second) return 0;
    HIP_TRY(hipMemcpyAsync(it->second, src, floats * 4, hipMemcpyDeviceToDevice, s));
    return 0;
  }

  int gn(const float* x, int ld, int cols, int B, int C, int T, const NormP& q, const float* other, int mode, float* out, hipStream_t s) {
    HIP_TRY(launch_mdx_groupnorm(x, ld, cols, B, C, T, q.gamma.f(), q.beta.f(), kGnEps, other, mode, out, gn_part.as<double>(), gn_stats.as<float>(), s));
    return 0;
  }

  int conv(const ConvP& c, const float* x, float* out, const float* skipmul, int B, int Ti, int Fi, hipStream_t s) {
    MdxConvParams p{};
    p.x = x; p.w = c.w.f(); p.bias = c.bias.f(); p.out = out; p.B = B; p.Cin = c.cin; p.Cout = c.cout; p.Ti = Ti; p.Fi = Fi;
    p.Tg = c.kind == MDX_DOWN2 ? Ti / 2 : Ti; p.Fg = c.kind == MDX_DOWN2 ? Fi / 2 : Fi;
    p.nchunks = c.nchunks; p.ntiles = c.ntiles;
    p.relu = group_norm() ? 0 : 1;
    p.skip = group_norm() ? nullptr : skipmul;
    HIP_TRY(launch_mdx_conv(c.kind, p, s));
    if (group_norm()) {
      const int To = c.kind == MDX_DOWN2 ? Ti / 2 : c.kind == MDX_UP2 ? 2 * Ti : Ti, Fo = c.kind == MDX_DOWN2 ? Fi / 2 : c.kind == MDX_UP2 ? 2 * Fi : Fi;
      RC_TRY(gn(out, Fo, Fo, B, c.cout, To, c.norm, skipmul, skipmul ? 1 : 0, out, s));
    }
    return 0;
  }

  // TFC_TDF (modules.py:72-74).  `in` is one of S[]; `dest` (an encoder's skip tensor) or null; returns where the result lives.
  int block(const BlockP& blk, float* in, float* dest, int B, float** result, hipStream_t s) {
    float* sa = nullptr; float* sb = nullptr;
    for (DevBuf& b : S) {
      float* q = b.as<float>();
      if (q == in) continue;
      (sa ? sb : sa) = q;
    }
    float* src = in;
    const int L = (int)blk.convs.size();
    for (int j = 0; j < L; ++j) {
      float* dst = (j == L - 1 && !blk.tdf.present && dest) ? dest : (src == sa ? sb : sa);
      RC_TRY(conv(blk.convs[j], src, dst, nullptr, B, blk.T, blk.F, s));
      src = dst;
    }
    if (!blk.tdf.present) { *result = src; return 0; }
    const TdfP& t = blk.tdf;
    float* out = dest ? dest : in;          // `in` is free again: src is a scratch buffer by now
    const int M = B * blk.c * blk.T;
    const bool gnm = group_norm();
    GemmF32Params g1{};
    g1.A = src; g1.lda = t.f; g1.W = t.w1.f(); g1.ldw = t.f; g1.bias = t.has_bias ? t.b1.f() : nullptr; g1.M = M; g1.K = t.f;
    g1.rowscale = gnm ? nullptr : t.n1.scale.f(); g1.rowshift = gnm ? nullptr : t.n1.shift.f(); g1.rows_per_ch = blk.T; g1.nch = blk.c;
    if (t.single) {
      g1.N = t.f; g1.out = out; g1.ldc = t.f; g1.res = src; g1.ldres = t.f;
      HIP_TRY(launch_gemm_f32(gnm ? F32_BIAS : F32_ROWAFF_RELU_RES, g1, s));
      if (gnm) RC_TRY(gn(out, t.f, t.f, B, blk.c, blk.T, t.n1, src, 2, out, s));
    } else {
      float* hid = hidden.as<float>();
      g1.N = t.ldh; g1.out = hid; g1.ldc = t.ldh;
      HIP_TRY(launch_gemm_f32(gnm ? F32_BIAS : F32_ROWAFF_RELU, g1, s));
      if (gnm) RC_TRY(gn(hid, t.ldh, t.h, B, blk.c, blk.T, t.n1, nullptr, 0, hid, s));
      GemmF32Params g2{};
      g2.A = hid; g2.lda = t.ldh; g2.W = t.w2.f(); g2.ldw = t.ldh; g2.bias = t.has_bias ? t.b2.f() : nullptr; g2.M = M; g2.N = t.f; g2.K = t.ldh;
      g2.out = out; g2.ldc = t.f; g2.res = src; g2.ldres = t.f;
      g2.rowscale = gnm ? nullptr : t.n2.scale.f(); g2.rowshift = gnm ? nullptr : t.n2.shift.f(); g2.rows_per_ch = blk.T; g2.nch = blk.c;
      HIP_TRY(launch_gemm_f32(gnm ? F32_BIAS : F32_ROWAFF_RELU_RES, g2, s));
      if (gnm) RC_TRY(gn(out, t.f, t.f, B, blk.c, blk.T, t.n2, src, 2, out, s));
    }
    *result = out;
    return 0;
  }

  int ensure(int B) {
    const size_t lvl0 = (size_t)B * cfg.g * cfg.dim_t * cfg.dim_f * 4;
    for (DevBuf& b : S) RC_TRY(b.ensure(lvl0));
    skip.resize(n);
    size_t hmax = 4;
    auto hid = [&](const BlockP& b) { if (b.tdf.present && !b.tdf.single) hmax = std::max(hmax, (size_t)B * b.c * b.T * b.tdf.ldh * 4); };
    for (int i = 0; i < n; ++i) {
      RC_TRY(skip[i].ensure((size_t)B * enc[i].c * enc[i].T * enc[i].F * 4));
      hid(enc[i]); hid(dec[i]);
    }
    hid(mid);
    RC_TRY(hidden.ensure(hmax));
    if (group_norm()) { RC_TRY(gn_part.ensure((size_t)B * 2 * 64 * 2 * 8)); RC_TRY(gn_stats.ensure((size_t)B * 2 * 2 * 4)); }
    return 0;
  }

  int forward(const float* x, int B, float* out, hipStream_t s) {
    RC_TRY(ensure(B));
    const int g = cfg.g, T0 = cfg.dim_t, F0 = cfg.dim_f;
    float* cur = S[0].as<float>();
    HIP_TRY(launch_mdx_first(x, first_w.f(), first_b.f(), cur, B, cfg.dim_c, g, F0, T0, group_norm() ? 0 : 1, s));
    if (group_norm()) RC_TRY(gn(cur, F0, F0, B, g, T0, first_norm, nullptr, 0, cur, s));
    RC_TRY(tap("first", cur, (size_t)B * g * T0 * F0, s));
    for (int i = 0; i < n; ++i) {
      const BlockP& e = enc[i];
      float* sk = skip[i].as<float>();
      float* r = nullptr;
      RC_TRY(block(e, cur, sk, B, &r, s));
      RC_TRY(tap(("enc" + std::to_string(i)).c_str(), sk, (size_t)B * e.c * e.T * e.F, s));
      cur = S[0].as<float>();
      RC_TRY(conv(ds[i], sk, cur, nullptr, B, e.T, e.F, s));
      RC_TRY(tap(("ds" + std::to_string(i)).c_str(), cur, (size_t)B * (e.c + g) * (e.T / 2) * (e.F / 2), s));
    }
    float* r = nullptr;
    RC_TRY(block(mid, cur, nullptr, B, &r, s));
    cur = r;
    RC_TRY(tap("bottleneck", cur, (size_t)B * mid.c * mid.T * mid.F, s));
    int Tc = mid.T, Fc = mid.F;
    for (int i = 0; i < n; ++i) {
      const BlockP& d = dec[i];
      float* nxt = cur == S[0].as<float>() ? S[1].as<float>() : S[0].as<float>();
      RC_TRY(conv(us[i], cur, nxt, skip[n - 1 - i].as<float>(), B, Tc, Fc, s));
      Tc *= 2; Fc *= 2;
      RC_TRY(tap(("us" + std::to_string(i)).c_str(), nxt, (size_t)B * d.c * d.T * d.F, s));
      RC_TRY(block(d, nxt, nullptr, B, &r, s));
      cur = r;
      RC_TRY(tap(("dec" + std::to_string(i)).c_str(), cur, (size_t)B * d.c * d.T * d.F, s));
    }
    HIP_TRY(launch_mdx_final(cur, final_w.f(), final_b.f(), out, B, g, cfg.dim_c, F0, T0, s));
    return 0;
  }

  int64_t flops(int B) const {
    const int64_t g = cfg.g, k = cfg.k;
    int64_t T = cfg.dim_t, f = cfg.dim_f, c = g, total = 2 * cfg.dim_c * g * T * f;
    auto blk = [&](int64_t c, int64_t T, int64_t f) {
      int64_t w = cfg.l * 2 * c * c * k * k * T * f;
      if (cfg.bn >= 0) {
        const int64_t h = cfg.bn == 0 ? f : f / cfg.bn;
        w += 2 * c * T * f * h * (cfg.bn == 0 ? 1 : 2);
      }
      return w;
    };
    for (int i = 0; i < n; ++i) {
      total += blk(c, T, f) + 2 * c * (c + g) * 4 * (T / 2) * (f / 2);
      T /= 2; f /= 2; c += g;
    }
    total += blk(c, T, f);
    for (int i = 0; i < n; ++i) {
      total += 2 * c * (c - g) * 4 * T * f;
      T *= 2; f *= 2; c -= g;
      total += blk(c, T, f);
    }
    total += 2 * c * cfg.dim_c * T * f;
    return total * B;
  }
};

extern "C" {

int lemas_mdx_create(const lemas_mdx_config* cfg, lemas_mdx** out) {
  if (!cfg || !out) { set_error("lemas_mdx_create: null argument"); return LEMAS_E_ARG; }
  const int n = cfg->num_blocks / 2;
  if (cfg->dim_c <= 0 || cfg->dim_c > 8 || cfg->dim_f <= 0 || cfg->dim_t <= 0 || cfg->num_blocks < 1 || n > 8 || cfg->l < 1 || cfg->g <= 0 ||
      cfg->bn < -1 || (cfg->norm != 0 && cfg->norm != 1)) {
    set_error("lemas_mdx_create: bad configuration"); return LEMAS_E_ARG;
  }
  if (cfg->k != 3) { set_error("lemas_mdx_create: TFC kernel size %d is not built (only 3, the value every published MDX-Net uses)", cfg->k); return LEMAS_E_ARG; }
  if ((cfg->dim_f % (1 << n)) || (cfg->dim_t % (1 << n))) { set_error("lemas_mdx_create: dim_f / dim_t must be divisible by 2^(num_blocks/2)"); return LEMAS_E_ARG; }
  if (cfg->bn >= 0 && ((cfg->dim_f >> n) & 3)) { set_error("lemas_mdx_create: dim_f / 2^(num_blocks/2) must be a multiple of 4 for the TDF linears"); return LEMAS_E_ARG; }
  if (cfg->bn > 0 && (cfg->dim_f >> n) / cfg->bn < 1) { set_error("lemas_mdx_create: bn larger than the bottleneck's frequency bins"); return LEMAS_E_ARG; }
  if (cfg->norm == 1 && (cfg->g & 1)) { set_error("lemas_mdx_create: GroupNorm(2, c) needs an even g"); return LEMAS_E_ARG; }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    set_error("lemas_mdx_create: no HIP device (this library has no CPU path)");
    return e != hipSuccess ? -(int)e : LEMAS_E_STATE;
  }
  RC_TRY(kernels_init());
  lemas_mdx* m = new lemas_mdx();
  m->cfg = *cfg; m->n = n;
  m->declare_schema();
  *out = m;
  return 0;
}

void lemas_mdx_destroy(lemas_mdx* m) { delete m; }

int lemas_mdx_load_weight(lemas_mdx* m, const char* name, const float* host_data, const int64_t* shape, int32_t ndim) {
  if (!m || !name || !host_data || (ndim > 0 && !shape)) { set_error("lemas_mdx_load_weight: null argument"); return LEMAS_E_ARG; }
  const std::string nm(name);
  auto ends = [&](const char* suf) { const size_t l = std::strlen(suf); return nm.size() >= l && nm.compare(nm.size() - l, l, suf) == 0; };
  if (nm == "window" || nm == "freq_pad" || ends("num_batches_tracked")) return 0;      // module state the forward never reads
  auto it = m->schema.find(nm);
  if (it == m->schema.end()) { set_error("unexpected tensor '%s' (strict load)", name); return LEMAS_E_WEIGHT; }
  bool ok = (int)it->second.size() == ndim;
  size_t numel = 1;
  for (int i = 0; i < ndim && ok; ++i) { ok = it->second[i] == shape[i]; numel *= (size_t)shape[i]; }
  if (!ok) { set_error("tensor '%s' has the wrong shape (ndim %d)", name, ndim); return LEMAS_E_WEIGHT; }
  if (m->finalized) { set_error("lemas_mdx_load_weight: already finalized (create a new object to reload)"); return LEMAS_E_STATE; }
  m->host[nm].assign(host_data, host_data + numel);
  return 0;
}

int lemas_mdx_finalize(lemas_mdx* m) {
  if (!m) { set_error("lemas_mdx_finalize: null"); return LEMAS_E_ARG; }
  if (m->finalized) return 0;
  return m->finalize();
}

int lemas_mdx_forward(lemas_mdx* m, const float* spek, int32_t batch, float* out, void* stream) {
  if (!m || !spek || !out || batch <= 0) { set_error("lemas_mdx_forward: bad arguments"); return LEMAS_E_ARG; }
  if (!m->finalized) { set_error("lemas_mdx_forward: finalize() first"); return LEMAS_E_STATE; }
  return m->forward(spek, batch, out, (hipStream_t)stream);
}

int lemas_mdx_set_option(lemas_mdx* m, const char* key, int64_t value) {
  if (!m || !key) { set_error("lemas_mdx_set_option: null argument"); return LEMAS_E_ARG; }
  if (!std::strcmp(key, "bf16x3")) {
    if (m->finalized) { set_error("lemas_mdx_set_option: 'bf16x3' decides the weight layout and must be set before finalize()"); return LEMAS_E_STATE; }
    m->bf16x3 = value != 0;
    return 0;
  }
  if (!std::strcmp(key, "bf16x3_products")) {      // 3 (default) or 4 bf16 MFMAs per product in the split-bf16 convolutions; process-wide
    if (value != 3 && value != 4) { set_error("lemas_mdx_set_option: bf16x3_products is 3 or 4"); return LEMAS_E_ARG; }
    mdx_conv_set_bx_products((int)value);
    return 0;
  }
  if (!std::strcmp(key, "conv_chunk")) {      // channels per K chunk of the exact 3x3 kernel (8 or 4); process-wide, takes effect at the next forward
    if (value != 4 && value != 8) { set_error("lemas_mdx_set_option: conv_chunk is 4 or 8"); return LEMAS_E_ARG; }
    mdx_conv_set_ck((int)value);
    return 0;
  }
  set_error("lemas_mdx_set_option: unknown option '%s'", key);
  return LEMAS_E_ARG;
}

int lemas_mdx_tap(lemas_mdx* m, const char* name, float* dev_out) {
  if (!m || !name) { set_error("lemas_mdx_tap: null argument"); return LEMAS_E_ARG; }
  if (dev_out) m->taps[name] = dev_out; else m->taps.erase(name);
  return 0;
}

int64_t lemas_mdx_flops(const lemas_mdx* m, int32_t batch) { return (m && batch > 0) ? m->flops(batch) : -1; }

}  // extern "C"
