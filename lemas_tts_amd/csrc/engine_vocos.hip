// lemas_vocos: Vocos (charactr/vocos-mel-24khz) decode on MI355X -- mel [B,100,L] -> waveform [B, 256 (L-1)].
// Replaces the `vocoder.decode(mel)` call of lemas_tts/infer/utils_infer.py:549 (loader :120-143).
// The arithmetic is the third-party `vocos` package's (VocosBackbone + ISTFTHead(padding="center")); all of it
// runs in fp32: the vocoder is < 0.1 % of an utterance's FLOPs, so exactness is free (f32 MFMA GEMMs).
//   embed Conv1d(100->512,k7)  = im2col + GEMM          ConvNeXt block = dwconv7 -> LN -> GEMM+GELU -> GEMM*gamma + res
//   ISTFT                      = head GEMM -> (mag, phase) -> spectrum -> windowed inverse-rDFT as a GEMM against a
//                                precomputed [1024 x 1026] basis -> overlap-add / window-envelope normalisation.
#include <cstring>
#include <utility>

#include "engine_common.h"

using namespace lemas;

struct lemas_vocos {
  int C = 100, dim = 512, idim = 1536, layers = 8, nfft = 1024, hop = 256;
  WeightStore ws;
  bool finalized = false;
  DevBuf basis;  // [nfft][ldk]
  int ldk = 0;
  DevBuf d_col, d_a, d_b, d_c, d_mid, d_head, d_spec, d_frames;
  // The backbone + head (everything between the im2col of the caller's mel and the overlap-add into the caller's waveform: ~36 launches on
  // the engine's own buffers) is replayed as ONE hipGraph per (batch, frames) shape.  A shape is captured the SECOND time it is seen (a
  // serving process decodes a different length almost every utterance: a capture costs about what the decode does); at most `graph_cap`
  // shapes are kept, least recently used evicted; a (re)allocated workspace or reloaded weight drops them all.
  // `done` is recorded behind every launch of `exec` on the stream it went to: a graph is destroyed only after its last launch finished,
  // whichever stream that was (the caller may decode on several)
  struct Slot { hipGraphExec_t exec = nullptr; hipEvent_t done = nullptr; unsigned long long used = 0; int seen = 0; };
  static void drop_slot(Slot& sl) {
    if (sl.done) { (void)hipEventSynchronize(sl.done); (void)hipEventDestroy(sl.done); }
    if (sl.exec) (void)hipGraphExecDestroy(sl.exec);
    sl = Slot{};
  }
  std::map<std::pair<int, int>, Slot> graphs;
  unsigned long long tick = 0, moved = 1, generation = 0;
  int graph_cap = 8;
  bool use_graph = true;

  lemas_vocos() {
    for (DevBuf* b : {&basis, &d_col, &d_a, &d_b, &d_c, &d_mid, &d_head, &d_spec, &d_frames}) b->moved = &moved;
  }
  void drop_graphs() {
    for (auto& g : graphs) drop_slot(g.second);
    graphs.clear();
  }
  ~lemas_vocos() {
    (void)hipDeviceSynchronize();
    drop_graphs();
    for (DevBuf* b : {&basis, &d_col, &d_a, &d_b, &d_c, &d_mid, &d_head, &d_spec, &d_frames}) b->release();
    ws.release();
  }
  void declare_schema() {
    ws.declare("backbone.embed.weight", {dim, C, 7});
    ws.declare("backbone.embed.bias", {dim});
    ws.declare("backbone.norm.weight", {dim});
    ws.declare("backbone.norm.bias", {dim});
    for (int i = 0; i < layers; ++i) {
      const std::string p = "backbone.convnext." + std::to_string(i) + ".";
      ws.declare(p + "dwconv.weight", {dim, 1, 7});
      ws.declare(p + "dwconv.bias", {dim});
      ws.declare(p + "norm.weight", {dim});
      ws.declare(p + "norm.bias", {dim});
      ws.declare(p + "pwconv1.weight", {idim, dim});
      ws.declare(p + "pwconv1.bias", {idim});
      ws.declare(p + "pwconv2.weight", {dim, idim});
      ws.declare(p + "pwconv2.bias", {dim});
      ws.declare(p + "gamma", {dim});
    }
    ws.declare("backbone.final_layer_norm.weight", {dim});
    ws.declare("backbone.final_layer_norm.bias", {dim});
    ws.declare("head.out.weight", {nfft + 2, dim});
    ws.declare("head.out.bias", {nfft + 2});
    ws.declare("head.istft.window", {nfft});
  }
  int finalize() {
    RC_TRY(ws.check_complete());
    if (dim != 512) { set_error("lemas_vocos: LayerNorm kernel is specialised for dim 512"); return LEMAS_E_ARG; }
    HIP_TRY(hipDeviceSynchronize());
    drop_graphs();
    ldk = ((nfft + 2) + 3) & ~3;
    RC_TRY(basis.ensure((size_t)nfft * ldk * 4));
    HIP_TRY(launch_dft_basis(ws.ptr("head.istft.window"), nfft, ldk, basis.as<float>(), nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    finalized = true;
    return 0;
  }
  // d_col (im2col of the mel) -> d_frames (windowed inverse-rDFT frames): the part of a decode that only touches the engine's buffers
  int body(int B, int L, hipStream_t s) {
    const int rows = B * L, kcol = C * 7;
    float *xa = d_a.as<float>(), *xb = d_b.as<float>(), *xc = d_c.as<float>();
    GemmF32Params g{};
    g.A = d_col.as<float>(); g.lda = kcol; g.W = ws.ptr("backbone.embed.weight"); g.ldw = kcol; g.bias = ws.ptr("backbone.embed.bias");
    g.out = xb; g.ldc = dim; g.M = rows; g.N = dim; g.K = kcol;
    HIP_TRY(launch_gemm_f32(F32_BIAS, g, s));
    HIP_TRY(launch_ln_affine(xb, ws.ptr("backbone.norm.weight"), ws.ptr("backbone.norm.bias"), xa, rows, dim, s));
    for (int i = 0; i < layers; ++i) {
      const std::string p = "backbone.convnext." + std::to_string(i) + ".";
      // dwconv -> LayerNorm in one launch (round 4; same statements, no [rows, 512] round trip)
      HIP_TRY(launch_dwconv7_ln(xa, ws.ptr(p + "dwconv.weight"), ws.ptr(p + "dwconv.bias"), ws.ptr(p + "norm.weight"), ws.ptr(p + "norm.bias"), xc, B, L,
                                dim, s));
      GemmF32Params g1{};
      g1.A = xc; g1.lda = dim; g1.W = ws.ptr(p + "pwconv1.weight"); g1.ldw = dim; g1.bias = ws.ptr(p + "pwconv1.bias");
      g1.out = d_mid.as<float>(); g1.ldc = idim; g1.M = rows; g1.N = idim; g1.K = dim;
      HIP_TRY(launch_gemm_f32(F32_BIAS_GELU, g1, s));
      GemmF32Params g2{};
      g2.A = d_mid.as<float>(); g2.lda = idim; g2.W = ws.ptr(p + "pwconv2.weight"); g2.ldw = idim; g2.bias = ws.ptr(p + "pwconv2.bias");
      g2.out = xa; g2.ldc = dim; g2.M = rows; g2.N = dim; g2.K = idim; g2.res = xa; g2.ldres = dim; g2.colscale = ws.ptr(p + "gamma");
      HIP_TRY(launch_gemm_f32(F32_BIAS_RES_SCALE, g2, s));
    }
    HIP_TRY(launch_ln_affine(xa, ws.ptr("backbone.final_layer_norm.weight"), ws.ptr("backbone.final_layer_norm.bias"), xb, rows, dim, s));
    GemmF32Params gh{};
    gh.A = xb; gh.lda = dim; gh.W = ws.ptr("head.out.weight"); gh.ldw = dim; gh.bias = ws.ptr("head.out.bias");
    gh.out = d_head.as<float>(); gh.ldc = nfft + 2; gh.M = rows; gh.N = nfft + 2; gh.K = dim;
    HIP_TRY(launch_gemm_f32(F32_BIAS, gh, s));
    HIP_TRY(launch_spec(d_head.as<float>(), rows, nfft / 2 + 1, nfft + 2, ldk, d_spec.as<float>(), s));
    GemmF32Params gd{};
    gd.A = d_spec.as<float>(); gd.lda = ldk; gd.W = basis.as<float>(); gd.ldw = ldk; gd.bias = nullptr;
    gd.out = d_frames.as<float>(); gd.ldc = nfft; gd.M = rows; gd.N = nfft; gd.K = ldk;
    HIP_TRY(launch_gemm_f32(F32_BIAS, gd, s));
    return 0;
  }
  // mel element (b, c, n) at mel[b * sb + c * sc + n * sl]
  int decode(const float* mel, long sb, long sc, long sl, int B, int L, float gain, float* wav, hipStream_t s) {
    if (!finalized) { set_error("lemas_vocos_decode: weights not finalized"); return LEMAS_E_STATE; }
    if (B <= 0 || L < 2 || !mel || !wav) { set_error("lemas_vocos_decode: bad arguments (B=%d L=%d)", B, L); return LEMAS_E_ARG; }
    const int rows = B * L, kcol = C * 7;
    RC_TRY(d_col.ensure((size_t)rows * kcol * 4));
    RC_TRY(d_a.ensure((size_t)rows * dim * 4));
    RC_TRY(d_b.ensure((size_t)rows * dim * 4));
    RC_TRY(d_c.ensure((size_t)rows * dim * 4));
    RC_TRY(d_mid.ensure((size_t)rows * idim * 4));
    RC_TRY(d_head.ensure((size_t)rows * (nfft + 2) * 4));
    RC_TRY(d_spec.ensure((size_t)rows * ldk * 4));
    RC_TRY(d_frames.ensure((size_t)rows * nfft * 4));
    if (generation != moved) { drop_graphs(); generation = moved; }     // a workspace moved: every captured address is stale

    HIP_TRY(launch_im2col7(mel, B, C, L, sb, sc, sl, d_col.as<float>(), s));
    hipGraphExec_t exec = nullptr;
    if (use_graph && s != nullptr) {               // the legacy NULL stream cannot be captured
      Slot& slot = graphs[{B, L}];
      slot.used = ++tick;
      if (!slot.exec && ++slot.seen >= 2) {
        hipGraph_t graph = nullptr;
        HIP_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        const int rc = body(B, L, s);
        const hipError_t e = hipStreamEndCapture(s, &graph);
        if (rc != 0) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        HIP_TRY(e);
        const hipError_t ei = hipGraphInstantiate(&slot.exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        HIP_TRY(ei);
      }
      exec = slot.exec;
      while ((int)graphs.size() > graph_cap) {     // evict the least recently used shape (never the one just touched)
        auto lru = graphs.begin();
        for (auto j = graphs.begin(); j != graphs.end(); ++j)
          if (j->second.used < lru->second.used) lru = j;
        drop_slot(lru->second);                    // waits for that graph's own last launch, on whatever stream it ran
        graphs.erase(lru);
      }
    }
    if (exec) {
      Slot& slot = graphs[{B, L}];
      if (!slot.done) HIP_TRY(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
      const hipError_t el = hipGraphLaunch(exec, s);
      (void)hipEventRecord(slot.done, s);          // also when the launch failed: whatever did get enqueued is fenced
      HIP_TRY(el);
    } else {
      RC_TRY(body(B, L, s));
    }
    HIP_TRY(launch_overlap_add(d_frames.as<float>(), ws.ptr("head.istft.window"), B, L, nfft, hop, wav, s));
    if (gain != 1.0f) HIP_TRY(launch_scale(wav, gain, (size_t)B * hop * (L - 1), s));
    return 0;
  }
};

extern "C" {

int lemas_vocos_create(int32_t input_channels, int32_t dim, int32_t intermediate_dim, int32_t num_layers, int32_t n_fft,
                       int32_t hop_length, lemas_vocos** out) {
  if (!out) return LEMAS_E_ARG;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    set_error("lemas_vocos_create: no HIP device (this library has no CPU path)");
    return e != hipSuccess ? -(int)e : LEMAS_E_STATE;
  }
  lemas_vocos* v = new lemas_vocos();
  v->C = input_channels; v->dim = dim; v->idim = intermediate_dim; v->layers = num_layers; v->nfft = n_fft; v->hop = hop_length;
  v->declare_schema();
  *out = v;
  return 0;
}
void lemas_vocos_destroy(lemas_vocos* v) { delete v; }
static int vocos_load(lemas_vocos* v, const char* name, const float* src, const int64_t* shape, int32_t ndim, bool on_device) {
  if (!v || !name || !src) return LEMAS_E_ARG;
  if (strncmp(name, "feature_extractor.", 18) == 0) return 0;  // mel front-end of the vocos checkpoint: not used by decode
  if (v->finalized || !v->graphs.empty()) HIP_TRY(hipDeviceSynchronize());   // a reload while a decode is in flight: wait before touching what it uses
  v->finalized = false;
  v->drop_graphs();                                // captured launches baked this tensor's address
  return v->ws.load(name, src, shape, ndim, on_device);
}
int lemas_vocos_load_weight(lemas_vocos* v, const char* name, const float* host, const int64_t* shape, int32_t ndim) {
  return vocos_load(v, name, host, shape, ndim, false);
}
int lemas_vocos_load_weight_device(lemas_vocos* v, const char* name, const float* dev, const int64_t* shape, int32_t ndim) {
  return vocos_load(v, name, dev, shape, ndim, true);
}
int lemas_vocos_finalize(lemas_vocos* v) { return v ? v->finalize() : LEMAS_E_ARG; }
int lemas_vocos_decode(lemas_vocos* v, const float* mel, int32_t batch, int32_t frames, float gain, float* wav, void* stream) {
  if (!v) return LEMAS_E_ARG;
  return v->decode(mel, (long)v->C * frames, frames, 1, batch, frames, gain, wav, (hipStream_t)stream);
}
int lemas_vocos_decode_rows(lemas_vocos* v, const float* mel_rows, int32_t batch, int32_t frames, int64_t batch_stride, float gain, float* wav,
                            void* stream) {
  if (!v) return LEMAS_E_ARG;
  if (batch_stride < (int64_t)v->C * frames && batch > 1) { set_error("lemas_vocos_decode_rows: batch_stride %lld is smaller than frames * channels", (long long)batch_stride); return LEMAS_E_ARG; }
  return v->decode(mel_rows, (long)batch_stride, 1, v->C, batch, frames, gain, wav, (hipStream_t)stream);
}
int lemas_vocos_set_option(lemas_vocos* v, const char* key, int64_t value) {
  if (!v || !key) return LEMAS_E_ARG;
  if (!strcmp(key, "graph")) { v->use_graph = value != 0; return 0; }
  if (!strcmp(key, "graph_cache")) {
    if (value < 1 || value > 1024) { set_error("lemas_vocos_set_option: graph_cache is the number of decode shapes kept, 1 .. 1024"); return LEMAS_E_ARG; }
    v->graph_cap = (int)value;
    return 0;
  }
  set_error("lemas_vocos_set_option: unknown option '%s'", key);
  return LEMAS_E_ARG;
}

}  // extern "C"
