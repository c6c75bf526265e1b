// lemas_dit: the flow-matching sampler engine (CFM.sample, lemas_tts/model/cfm.py:206-473) on MI355X.
//
// Structure of one utterance batch (B samples, N frames, S = NFE steps):
//   prepare():  everything that does not depend on the ODE state, computed ONCE in fp32
//       - conditioning: cond (+ prosody_to_mel), step_cond = where(mask, cond, 0)        cfm.py:311-318,388-390
//       - text embedding for the text and dropped-text CFG branches                       dit.py:51-81 (reference caches per branch, :212-220)
//       - prosody text conditioning added to both branches                               dit.py:225-233
//       - the [cond | text] part of the input projection + bias (exact algebraic split)  dit.py:97
//       - time MLP and ALL AdaLN modulation vectors for all S steps (depend on t only)   modules.py:311,332,727-731
//       - rotary cos/sin table for N                                                     dit.py:236
//   solve():    S Euler steps; one step = one DiT forward over 2B rows-batches (CFG folded into the batch, the
//               reference runs the two branches sequentially, cfm.py:393-417) + fused CFG/clamp/Euler update.
//               The step's ~165 launches are captured once per shape into a hipGraph and replayed; the step
//               index lives in device memory so the same graph serves every step.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "engine_common.h"


namespace lemas {

enum ProfClass { PC_INPROJ, PC_CONVPOS, PC_LN, PC_GEMM_QKV, PC_GEMM_QK, PC_GEMM_V, PC_ATTN, PC_GEMM_OUT, PC_GEMM_FF1, PC_GEMM_FF2, PC_GEMM_FINAL,
                 PC_CFG_EULER, PC_COUNT };
static const char* kProfNames[PC_COUNT] = {"inproj_f32", "convpos", "ln_mod", "gemm_qkv_fused", "gemm_qk_rope", "gemm_v_t", "attention", "gemm_attn_out",
                                           "gemm_ff1_gelu", "gemm_ff2", "gemm_proj_out", "cfg_euler"};

struct BlockW {
  DevBuf wqkv, wo, w1, w2;  // bf16
  DevBuf wqkv8, wo8, w18, w28;   // e4m3 copies for the fp8 path (built on first use of option "fp8")
  DevBuf wqkvq, woq, w1q, w2q;   // option "fp8" = 2: the e4m3 weights dequantised back to bf16 (weights-only fp8, bf16 activations)
  DevBuf sqkv, so, s1, s2;       // fp32 per-output-channel scales
  DevBuf bqkv;              // fp32 [3*inner]
  const float *bo = nullptr, *b1 = nullptr, *b2 = nullptr;
  // fp8 path on outlier checkpoints (quantize_fp8): bf16 weight rows / biases of the flagged output channels of attn.to_out and ff.2
  // ([32][K] bf16, [32] fp32; outlier_rows.hip) and copies of the two bias vectors with those channels zeroed (the fp8 GEMM, whose e4m3 image of
  // those rows is zeroed too, must contribute nothing there)
  DevBuf wo_side, w2_side, bo_side, b2_side, bo_z, b2_z;
};

}  // namespace lemas

using namespace lemas;

struct lemas_dit {
  lemas_dit_config cfg{};
  WeightStore ws;
  bool finalized = false;
  bool use_graph = true;
  bool profile = false;
  bool table_cache = true;  // reuse the AdaLN/time tables while the t-grid is unchanged
  bool dual = true;         // run the two CFG branches as concurrent lanes (second stream / parallel graph branch)
  bool qkv_fused = true;    // QK and V projections of a lane in one launch
  bool fp8 = false;         // block GEMMs on the MXFP8 path (BASELINE config 5); weights quantised on first use
  bool fp8_ready = false;
  // option "fp8" = 2 (an ACCURACY point, not a speed path): BASELINE configs[4] read literally -- "fp8 MFMA weights": the block GEMM weights
  // are e4m3 with one fp32 scale per output channel, the activations stay bf16.  Run as the bf16 kernels on the dequantised weights: the
  // same numbers a weights-only fp8 kernel would produce, next to the MXFP8 path that also quantises the activations
  bool fp8_wonly = false, fp8_wonly_ready = false;
  // fp8 OUTLIER GUARD (option "fp8_outlier_guard", default on).  Trained DiT checkpoints carry a few residual-stream channels tens of times
  // larger than the rest; synth.synth_cfm_state_dict(outlier=...) reproduces the mechanism (rows of attn.to_out / ff.2, weight and bias,
  // scaled up).  Measured against the reference's own output at full depth and NFE 32 (tests/golden/configs0_outlier_nfe32.npz, 1 % of the
  // channels x30; profiles/r04/r04_fp8_outlier_points.txt): bf16 3.9e-6, the fp8 path 2.9e-4 (weights-only fp8 2.2e-4) against the 1e-4
  // target -- and no part of it is safe to keep: ONE GEMM site on fp8 with the other three on bf16 gives 6.4e-5 (QKV), 8.4e-5
  // (out-projection), 1.15e-4 (FF1), 1.05e-4 (FF2), five to seven times what the same site costs on weights without outliers.  So fp8 does
  // not ship for such checkpoints: quantize_fp8() already computes the per-output-channel scales of the residual-writing projections,
  // which show the outlier channels (scale > 8x the median over channels), and when there are any the engine keeps every block GEMM on its
  // bf16 operands although "fp8" is on (the result is then the bf16 path's, bit for bit).  `fp8_outlier_channels` / `fp8_gemms_kept_bf16`
  // (lemas_dit_get_stat) tell the caller.  What the guard watches is this one mechanism; outliers made elsewhere (a modulation scale, the
  // text embedding) are not seen -- a first-contact item for real checkpoints.
  bool fp8_guard = true, fp8_guard_tripped = false;
  int fp8_outlier_channels = 0;
  // option "fp8_outlier_mode" (round 5; default 0; MEASUREMENT BUILDS ONLY since round 6 -- slower than the bf16 fallback it replaces, 125 vs 136
  // audio-s/s at configs[4], so the product does not carry it): 1 = when the guard trips on at most 32 channels, fp8 is NOT given up: QKV, out-projection and FF2
  // keep their fp8 operands (three of the four sites), FF1 runs on bf16 operands, and the flagged OUTPUT channels of out-projection / FF2 are
  // computed from bf16 operands by outlier_rows.hip (their rows of the e4m3 images and their biases are zeroed), which also writes the MXFP8 image
  // of its input rows for the fp8 GEMM of the site.  Accuracy against the reference's own outputs: 8.4e-5 at NFE 32 and 9.6e-5 on the 8-step
  // fixture (target 1e-4; unguarded 2.9e-4 / 6.8e-4; tests/test_gpu_02_fp8.py asserts the tolerance).  Throughput: 125 audio-s/s at configs[4]'s
  // shape against 136 for 0 = every block GEMM on bf16 (round 4's behaviour, 3.9e-6): at these shapes the three fp8 sites save ~16 us per block
  // and lane, the two side launches cost ~39 (profiles/r05/r05_fp8_outlier_decomposition.txt) -- which is why 0 stays the default.
  bool fp8_outlier_mode = false;
  bool fp8_outlier_split = false;            // set by quantize_fp8(): the decomposition is in force
  std::vector<int> h_flagged;
  DevBuf d_flagged;                          // int [32]: flagged channels (ascending), -1 past the count
  // which of a block's four GEMM sites take fp8 operands (bit 0 QKV, 1 out-projection, 2 FF1, 3 FF2): option "fp8_sites" (default all),
  // narrowed by the guard.  Each site's input comes from its own producer (LayerNorm 1, attention, LayerNorm 2, FF1's epilogue), so the
  // four choices are independent.
  int fp8_sites_opt = 15;
  int fp8_sites() const { return !fp8 ? 0 : (fp8_guard && fp8_guard_tripped) ? (fp8_outlier_split ? (fp8_sites_opt & 0b1011) : 0) : fp8_sites_opt; }
  bool outlier_rows_on() const { return fp8 && fp8_guard && fp8_guard_tripped && fp8_outlier_split; }
  // measurement options (per engine; changing one drops the cached graphs): explicit tile ids for the block GEMMs with N == 1024 /
  // N == 2048, for the fused QK+V launch, and the XCD block grid of the tile order; 0 = the production choice
  // option "skip_dead" (0 by default): what the FF HALF of a block does with a ragged batch's padding blocks (the attention half skips them
  // always and exactly: skip_masked below).  0 = computes them, as the reference does -- its unmasked position-embedding conv (dit.py:98 ->
  // modules.py:167-190, kernel 31 twice) lets the last ~30 valid frames of a sample see the padding rows' ODE state.  1 = skips them all: those
  // frames differ from the reference's by 1e-5 mel-MSE instead of 2e-6 (tolerance 1e-4; profiles/r04/r04g_skip_dead_ragged_batches.txt).
  // 2 = skips all but ONE block behind every sample's last live block (d_live = min(len + 128, N)): the padding rows the position conv
  // reaches into are then evolved by the chain as the reference evolves them -- the reference's own error level -- at most of the saving
  int skip_dead = 0;
  bool skip_masked = true;    // option "skip_masked": the attention half of every block skips padding blocks (exact; see run_step); 0 for A/B runs
  int opt_tile_n1024 = 0, opt_tile_n2048 = 0, opt_tile_qkv = 0, opt_xcd_gx = 0, opt_xcd_runs = 0;
  // attention schedule variant (attention.hip VAR).  19 = no running max (P = exp2(S) on q prescaled by the QK epilogue, one range check
  // per workgroup with a classical second pass if it trips) + static priority for the younger half-workgroup: 25.9 -> 22.5 us per lane
  // launch at configs[1], +4.6 % end to end (profiles/r03/r03_attention_variants.txt); 0 = the classical online softmax
  int attn_variant = 19;
  // measurement option: lane 1 launches stage k of a block only after lane 0's stage k has completed (the lanes run one stage apart
  // instead of in lock-step, so unlike kernels share the chip)
  int lane_skew = 0;
  int ln_skip = 0;            // ABLATION, measurement builds only (wrong results): 1 = no LayerNorm launch after block 0's first -- the upper bound of any LayerNorm fusion
  // fp8 QK^T in attention (attention.hip VAR & ATTN_F8QK): bits 0-1 = 0 off | 1 the QK GEMM epilogue writes q, k as MXFP8 | 2 a side launch
  // quantises the bf16 rows (same bits, one more launch: the A/B form); bit 2 = also while the block GEMMs run on bf16 operands (measurements:
  // BASELINE's bf16 configurations must not use it).  Takes effect with a prescaled-q attention variant (17 / 19) only.
  int attn_f8qk = 1;
  int f8qk_mode() const {
    if ((attn_f8qk & 3) == 0 || (attn_variant & 17) != 17 || (attn_variant & ~19) != 0) return 0;
    return ((attn_f8qk & 4) != 0 || fp8_sites() != 0) ? (attn_f8qk & 3) : 0;
  }
  hipEvent_t ev_skew[8] = {};
  // measurement option (measurement builds only): the FF half of every block -- out-projection, ff_norm, FF1, FF2 -- as ONE persistent launch per
  // lane with grid barriers between the stages (gemm_bf16.hip gemm_chain_ffhalf_kernel); 2 = with the next stage's weights prefetched across the barrier
  int block_persist = 0;
  // the AdaLN LayerNorms behind the gated residual updates as the tail of those GEMM launches (gemm_bf16.hip ln_tail).  OFF: measured on
  // configs[1] it is 1.5x SLOWER end to end (88.9 -> 59.8 audio-s/s, profiles/r03/r03_ln_tail_experiment.txt): with two lanes sharing the chip a
  // panel's column tiles do not run at the same time, so finished workgroups sit on their CUs waiting for panel-mates that have not started
  bool ln_fused = false;
  // "ln fold" (common.h GemmParams): the AdaLN LayerNorms of the block chain folded across the GEMMs on either side -- the gate +
  // residual epilogues write the scaled bf16 rows and per-row partial sums, the QKV / FF1 epilogues apply the row statistics, and c1 / c2
  // rows per ODE step live in the AdaLN table.  Two of the seven launches per block and lane disappear.  bf16 activations only (the
  // MXFP8 path keeps its quantising LayerNorm launch).  OFF: end to end it is a wash at configs[1] (95.9 = 95.9 audio-s/s) and 2-2.5 %
  // slower at the batched and the short workload (profiles/r03/r03_structural_attempts.txt).  What the two removed launches cost a lane's
  // chain the other lane was already hiding; what counts with two lanes on the chip is workgroup-time, and there the fold adds (1-2 us per
  // producer launch for the bf16 image and the statistics, 1-2.5 us per consumer launch for re-reading 256 B of statistics per row by
  // every column tile) about what the two small LayerNorm launches took.
  bool ln_fold = false;
  bool fold_on() const { return ln_fold && !fp8; }
  unsigned int* ln_err_host = nullptr;   // pinned, device-visible: set by a device-side wait that gave up (checked at every entry point)
  unsigned int* ln_err_dev = nullptr;
  hipStream_t s2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // option "lane_split" (k = 1, 2, 4; default: see lanes_for()): each CFG branch is cut into k groups of B / k samples, every group an independent
  // chain of launches on its own stream / graph branch -- 2 k lanes.  Batched shapes only (k divides B): more, smaller launches in flight fill the
  // CUs a lone multi-round launch leaves idle in its last round and at its boundaries.  Results do not depend on k (rows are independent).
  int lane_split = 0;                       // 0 = automatic
  static constexpr int MAX_LANES = 8;
  hipStream_t sx[MAX_LANES - 2] = {};       // lanes 2 .. 7 (lane 0 runs on the caller's stream, lane 1 on s2)
  hipEvent_t ev_joinx[MAX_LANES - 2] = {};
  int lanes_for(hipStream_t s) const {
    if (!(dual && use_cfg && s != nullptr)) return 1;
    int k = lane_split;
    if (k <= 0) k = 1;                      // automatic: one lane per CFG branch (what every workload measured best with so far)
    while (k > 1 && (B % k != 0 || 2 * k > MAX_LANES)) k >>= 1;
    return 2 * k;
  }

  std::vector<BlockW> blocks;
  DevBuf wproj_out, bproj_out;  // padded to 128 rows
  DevBuf d_tabW, d_tabB;        // device arrays of the per-block attn_norm.linear weight / bias pointers (batched table GEMM)
  DevBuf wconv[2];              // [G][taps][64][64] bf16

  // --- per-shape state (valid after prepare)
  int B = 0, N = 0, F = 0, Nt = 0, S = 0, BB = 0, npad = 0;
  int pitch = 0;  // rows per sample in the activation row space: N rounded up to 128 (tiles never straddle samples)
  bool use_cfg = true, has_len = false, prepared = false;
  std::vector<float> tgrid_cached;
  int rope_n = 0;
  std::vector<float> h_dt, h_cfg, h_t;

  DevBuf d_step, d_dt, d_cfg, d_t, d_tab, d_rope_cos, d_rope_sin, d_len, d_live;
  DevBuf d_sin, d_h1, d_temb, d_st;                       // time path scratch
  DevBuf d_cond_eff, d_step_cond, d_pm, d_pt;             // conditioning
  DevBuf d_te, d_rowmask, d_t1, d_t2, d_t3, d_gx, d_ct;   // text embedding scratch
  DevBuf d_pconst, d_y, d_xres, d_hbf, d_q, d_k, d_vt, d_abf, d_ff, d_cmid, d_pred;
  DevBuf d_q8, d_k8, d_qs8, d_ks8;      // option attn_f8qk: the MXFP8 images of q and k ([B2, H, pitch, 64] e4m3, [B2, H, pitch, 2] E8M0)
  DevBuf d_h8, d_hmx, d_a8, d_amx, d_ff8, d_ffmx;         // MXFP8 activations of the fp8 path (bytes + E8M0 scales)
  DevBuf d_lncnt;                                         // arrival counters of the fused LayerNorm tails: [block][site][lane][panel]
  DevBuf d_lnpart;                                        // ln fold: [rows][32][2] (sum, sum of squares) per 32-column slot
  DevBuf d_foldA, d_foldtmp, d_foldsites, d_foldparams, d_zero;   // ln-fold table build: split A operands, GEMM outputs, site / launch descriptors
  std::vector<LnFoldSite> h_foldsites;
  std::vector<GemmParams> h_foldparams;
  int tab_stride = 0;
  int n_cus = 0;

  // Step graphs are cached per BUCKET: everything that fixes the graph's topology and buffer addresses (batch, the 128-row pitch of the
  // activation row space, CFG / length / lane / precision switches).  The frame count N inside a bucket only changes kernel ARGUMENTS
  // (mask bounds, valid rows), so a never-seen N in a seen bucket re-captures the launches (host work, ~1 ms) and patches one of the
  // bucket's instantiated graphs in place with hipGraphExecUpdate instead of instantiating a new one.  A bucket holds up to two
  // instantiated graphs, used alternately for new lengths, so that the one being patched is never the one still in flight (its last
  // launch is fenced by an event).  Buckets are evicted least-recently-used beyond `graph_cap`.
  struct StepGraph { hipGraphExec_t exec = nullptr; int N = 0; unsigned long long used = 0; hipEvent_t done = nullptr; };
  struct GraphBucket { StepGraph g[2]; unsigned long long used = 0; };
  std::map<std::string, GraphBucket> graphs;
  unsigned long long graph_tick = 0;
  int graph_cap = 16;                       // option "graph_cache": buckets kept (LRU)
  bool graph_update = true;                 // option "graph_update": 0 = always instantiate (measurement)
  long long n_capture = 0, n_instantiate = 0, n_update = 0, n_update_fail = 0, n_evict = 0;   // lemas_dit_get_stat
  unsigned long long moved = 1;             // bumped by this engine's DevBufs when one of them is (re)allocated
  unsigned long long graph_generation = 0;  // value of `moved` the cached graphs were captured under

  struct ProfRec { hipEvent_t a, b; int cls; };
  std::vector<ProfRec> prof;

  std::vector<DevBuf*> own_bufs() {
    return {&wproj_out, &bproj_out, &d_tabW, &d_tabB, &wconv[0], &wconv[1], &d_step, &d_dt, &d_cfg, &d_t, &d_tab, &d_rope_cos,
            &d_rope_sin, &d_len, &d_live, &d_sin, &d_h1, &d_temb, &d_st, &d_cond_eff, &d_step_cond, &d_pm, &d_pt, &d_te,
            &d_rowmask, &d_t1, &d_t2, &d_t3, &d_gx, &d_ct, &d_pconst, &d_y, &d_xres, &d_hbf, &d_q, &d_k, &d_vt, &d_q8, &d_k8, &d_qs8, &d_ks8,
            &d_abf, &d_ff, &d_cmid, &d_pred, &d_h8, &d_hmx, &d_a8, &d_amx, &d_ff8, &d_ffmx, &d_lncnt, &d_lnpart,
            &d_foldA, &d_foldtmp, &d_foldsites, &d_foldparams, &d_zero, &d_flagged};
  }
  static std::vector<DevBuf*> block_bufs(BlockW& b) {
    return {&b.wqkv, &b.wo, &b.w1, &b.w2, &b.bqkv, &b.wqkv8, &b.wo8, &b.w18, &b.w28, &b.sqkv, &b.so, &b.s1, &b.s2, &b.wqkvq, &b.woq, &b.w1q, &b.w2q,
            &b.wo_side, &b.w2_side, &b.bo_side, &b.b2_side, &b.bo_z, &b.b2_z};
  }
  static void drop_bucket(GraphBucket& b) {
    for (auto& g : b.g) {
      if (g.done) { (void)hipEventSynchronize(g.done); (void)hipEventDestroy(g.done); }
      if (g.exec) (void)hipGraphExecDestroy(g.exec);
      g = StepGraph{};
    }
  }
  void drop_graphs() {
    for (auto& b : graphs) drop_bucket(b.second);
    graphs.clear();
  }
  int capture_step(hipStream_t s, hipGraph_t* out);
  int step_graph(hipStream_t s, hipGraphExec_t* exec, hipEvent_t* done);
  lemas_dit() {
    for (DevBuf* b : own_bufs()) b->moved = &moved;
  }
  lemas_dit(const lemas_dit&) = delete;
  lemas_dit& operator=(const lemas_dit&) = delete;
  ~lemas_dit() {
    (void)hipDeviceSynchronize();   // nothing of this engine may still be running when its graphs and buffers go
    drop_graphs();
    if (s2) (void)hipStreamDestroy(s2);
    for (auto& q : sx) if (q) (void)hipStreamDestroy(q);
    for (auto& e : ev_joinx) if (e) (void)hipEventDestroy(e);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    for (auto& e : ev_skew) if (e) (void)hipEventDestroy(e);
    for (auto& r : prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (DevBuf* b : own_bufs()) b->release();
    for (auto& b : blocks)
      for (DevBuf* w : block_bufs(b)) w->release();
    ws.release();
    if (ln_err_host) (void)hipHostFree(ln_err_host);
  }
  // a device-side wait that gave up left its mark in pinned host memory: every later call on this engine fails loudly
  int health() const {
    if (ln_err_host && *(volatile unsigned int*)ln_err_host) {
      set_error("lemas_dit: a fused LayerNorm tail gave up waiting for its row panel (results since then are invalid); "
                "set option ln_fused=0 and report the launch shape");
      return LEMAS_E_STATE;
    }
    return 0;
  }

  int inner() const { return cfg.heads * cfg.dim_head; }
  std::string T(const std::string& s) const { return "transformer." + s; }

  void declare_schema();
  int finalize();
  int quantize_fp8();
  int dequantize_fp8();
  int prepare(const lemas_sample_args* a, hipStream_t s);
  int solve(const lemas_sample_args* a, hipStream_t s);
  int enqueue_forward(hipStream_t s);
  int enqueue_update(float* traj, hipStream_t s);
  int build_tables(const lemas_sample_args* a, hipStream_t s);
  int build_fold_tables(int Snew, hipStream_t s);
  // AdaLN table row: [depth x 6 d modulation | 2 d final norm | depth x (c1_qkv, c2_qkv [3 inner each], c1_ff1, c2_ff1 [ff each])]
  int fold_off(int l) const { return cfg.depth * 6 * cfg.dim + 2 * cfg.dim + l * (6 * inner() + 2 * cfg.ff_mult * cfg.dim); }
  int text_embed(const lemas_sample_args* a, hipStream_t s);

  // profiling helpers
  int pbegin(int cls, hipStream_t s) {
    if (!profile) return 0;
    ProfRec r;
    r.cls = cls;
    HIP_TRY(hipEventCreate(&r.a));
    HIP_TRY(hipEventCreate(&r.b));
    HIP_TRY(hipEventRecord(r.a, s));
    prof.push_back(r);
    return 0;
  }
  int pend(hipStream_t s) {
    if (!profile) return 0;
    HIP_TRY(hipEventRecord(prof.back().b, s));
    return 0;
  }
  // single-kernel classes: the launcher stamps the pair with the dispatch's own begin / end (hipExtLaunchKernelGGL)
  int pkernel(int cls, hipEvent_t* a, hipEvent_t* b) {
    *a = *b = nullptr;
    if (!profile) return 0;
    ProfRec r;
    r.cls = cls;
    HIP_TRY(hipEventCreate(&r.a));
    HIP_TRY(hipEventCreate(&r.b));
    prof.push_back(r);
    *a = r.a; *b = r.b;
    return 0;
  }
};

void lemas_dit::declare_schema() {
  const int64_t d = cfg.dim, td = cfg.text_dim, in = inner(), md = cfg.mel_dim;
  auto D = [&](const std::string& n, std::vector<int64_t> sh) { ws.declare(n, std::move(sh)); };
  D(T("time_embed.time_mlp.0.weight"), {d, cfg.time_freq_dim});
  D(T("time_embed.time_mlp.0.bias"), {d});
  D(T("time_embed.time_mlp.2.weight"), {d, d});
  D(T("time_embed.time_mlp.2.bias"), {d});
  D(T("time_embed.freqs"), {cfg.time_freq_dim / 2});
  D(T("text_embed.text_embed.weight"), {cfg.vocab_rows, td});
  D(T("text_embed.freqs_cis"), {4096, td});
  for (int i = 0; i < cfg.conv_layers; ++i) {
    const std::string p = T("text_embed.text_blocks." + std::to_string(i) + ".");
    D(p + "dwconv.weight", {td, 1, 7});
    D(p + "dwconv.bias", {td});
    D(p + "norm.weight", {td});
    D(p + "norm.bias", {td});
    D(p + "pwconv1.weight", {2 * td, td});
    D(p + "pwconv1.bias", {2 * td});
    D(p + "grn.gamma", {1, 1, 2 * td});
    D(p + "grn.beta", {1, 1, 2 * td});
    D(p + "pwconv2.weight", {td, 2 * td});
    D(p + "pwconv2.bias", {td});
  }
  if (cfg.has_prosody) {
    D(T("prosody_text_proj.weight"), {td, 512});
    D(T("prosody_text_proj.bias"), {td});
    D("prosody_to_mel.weight", {md, 512});
    D("prosody_to_mel.bias", {md});
  }
  D(T("input_embed.proj.weight"), {d, 2 * md + td});
  D(T("input_embed.proj.bias"), {d});
  for (int j : {0, 2}) {
    D(T("input_embed.conv_pos_embed.conv1d." + std::to_string(j) + ".weight"), {d, d / cfg.conv_pos_groups, cfg.conv_pos_kernel});
    D(T("input_embed.conv_pos_embed.conv1d." + std::to_string(j) + ".bias"), {d});
  }
  D(T("rotary_embed.inv_freq"), {cfg.dim_head / 2});
  for (int i = 0; i < cfg.depth; ++i) {
    const std::string p = T("transformer_blocks." + std::to_string(i) + ".");
    D(p + "attn_norm.linear.weight", {6 * d, d});
    D(p + "attn_norm.linear.bias", {6 * d});
    for (const char* n : {"to_q", "to_k", "to_v"}) {
      D(p + "attn." + n + ".weight", {in, d});
      D(p + "attn." + n + ".bias", {in});
    }
    D(p + "attn.to_out.0.weight", {d, in});
    D(p + "attn.to_out.0.bias", {d});
    D(p + "ff.ff.0.0.weight", {(int64_t)cfg.ff_mult * d, d});
    D(p + "ff.ff.0.0.bias", {(int64_t)cfg.ff_mult * d});
    D(p + "ff.ff.2.weight", {d, (int64_t)cfg.ff_mult * d});
    D(p + "ff.ff.2.bias", {d});
  }
  D(T("norm_out.linear.weight"), {2 * d, d});
  D(T("norm_out.linear.bias"), {2 * d});
  D(T("proj_out.weight"), {md, d});
  D(T("proj_out.bias"), {md});
}

// e4m3 copies of the block GEMM weights with one fp32 scale per output channel
int lemas_dit::quantize_fp8() {
  if (fp8_ready) return 0;
  const int d = cfg.dim, in = inner(), ffd = cfg.ff_mult * d;
  hipStream_t s = nullptr;
  for (int i = 0; i < cfg.depth; ++i) {
    const std::string p = T("transformer_blocks." + std::to_string(i) + ".");
    BlockW& b = blocks[i];
    RC_TRY(b.wqkv8.ensure((size_t)3 * in * d));
    RC_TRY(b.sqkv.ensure((size_t)3 * in * 4));
    int j = 0;
    for (const char* n : {"to_q", "to_k", "to_v"}) {
      HIP_TRY(launch_w_quant_f8(ws.ptr(p + "attn." + n + ".weight"), in, d, b.wqkv8.as<uint8_t>() + (size_t)j * in * d,
                                b.sqkv.as<float>() + (size_t)j * in, s));
      ++j;
    }
    RC_TRY(b.wo8.ensure((size_t)d * in));
    RC_TRY(b.so.ensure((size_t)d * 4));
    HIP_TRY(launch_w_quant_f8(ws.ptr(p + "attn.to_out.0.weight"), d, in, b.wo8.as<uint8_t>(), b.so.as<float>(), s));
    RC_TRY(b.w18.ensure((size_t)ffd * d));
    RC_TRY(b.s1.ensure((size_t)ffd * 4));
    HIP_TRY(launch_w_quant_f8(ws.ptr(p + "ff.ff.0.0.weight"), ffd, d, b.w18.as<uint8_t>(), b.s1.as<float>(), s));
    RC_TRY(b.w28.ensure((size_t)d * ffd));
    RC_TRY(b.s2.ensure((size_t)d * 4));
    HIP_TRY(launch_w_quant_f8(ws.ptr(p + "ff.ff.2.weight"), d, ffd, b.w28.as<uint8_t>(), b.s2.as<float>(), s));
  }
  HIP_TRY(hipStreamSynchronize(s));
  // outlier guard: a residual channel written with weights far above the typical row is an activation outlier in every later LayerNorm
  {
    std::vector<float> chan(d, 0.f), tmp(d);
    for (auto& b : blocks)
      for (const DevBuf* sc : {&b.so, &b.s2}) {
        HIP_TRY(hipMemcpy(tmp.data(), sc->p, (size_t)d * 4, hipMemcpyDeviceToHost));
        for (int c = 0; c < d; ++c) chan[c] = tmp[c] > chan[c] ? tmp[c] : chan[c];
      }
    std::vector<float> sorted(chan);
    std::sort(sorted.begin(), sorted.end());
    const float med = sorted[d / 2];
    fp8_outlier_channels = 0;
    h_flagged.clear();
    for (int c = 0; c < d; ++c)
      if (chan[c] > 8.0f * med) { ++fp8_outlier_channels; h_flagged.push_back(c); }
    fp8_guard_tripped = fp8_outlier_channels > 0;
  }
  // mixed-precision decomposition for outlier checkpoints: side operands of the flagged channels, zeroed rows / biases for the fp8 GEMMs
  fp8_outlier_split = false;
#ifdef LEMAS_MEASUREMENT_BUILD
  // (outlier_rows.hip is built for K = 1024 and 2048 only: any other inner / FF width keeps the all-bf16 guard behaviour, decided BEFORE a row is zeroed)
  const bool side_ok = (in == 1024 || in == 2048) && (ffd == 1024 || ffd == 2048);
  if (fp8_guard && fp8_guard_tripped && fp8_outlier_mode && fp8_outlier_channels <= 32 && side_ok) {
    const int nf = fp8_outlier_channels;
    std::vector<int> pad(32, -1);
    for (int j = 0; j < nf; ++j) pad[j] = h_flagged[j];
    RC_TRY(d_flagged.ensure(32 * sizeof(int)));
    HIP_TRY(hipMemcpy(d_flagged.p, pad.data(), 32 * sizeof(int), hipMemcpyHostToDevice));
    for (auto& b : blocks) {
      struct Site { DevBuf* wbf; DevBuf* w8; const float* bias; DevBuf* wside; DevBuf* bside; DevBuf* bz; int K; };
      for (const Site& st : {Site{&b.wo, &b.wo8, b.bo, &b.wo_side, &b.bo_side, &b.bo_z, in}, Site{&b.w2, &b.w28, b.b2, &b.w2_side, &b.b2_side, &b.b2_z, ffd}}) {
        RC_TRY(st.wside->ensure((size_t)32 * st.K * 2));       // zero-filled by ensure(): rows past nf stay zero
        RC_TRY(st.bside->ensure(32 * 4));
        RC_TRY(st.bz->ensure((size_t)d * 4));
        HIP_TRY(hipMemset(st.wside->p, 0, (size_t)32 * st.K * 2));
        HIP_TRY(hipMemset(st.bside->p, 0, 32 * 4));
        HIP_TRY(hipMemcpy(st.bz->p, st.bias, (size_t)d * 4, hipMemcpyDeviceToDevice));
        for (int j = 0; j < nf; ++j) {
          const int c = h_flagged[j];
          HIP_TRY(hipMemcpy(st.wside->as<bf16_t>() + (size_t)j * st.K, st.wbf->as<bf16_t>() + (size_t)c * st.K, (size_t)st.K * 2, hipMemcpyDeviceToDevice));
          HIP_TRY(hipMemcpy(st.bside->as<float>() + j, st.bias + c, 4, hipMemcpyDeviceToDevice));
          HIP_TRY(hipMemset(st.w8->as<uint8_t>() + (size_t)c * st.K, 0, (size_t)st.K));        // e4m3 0x00 = +0
          HIP_TRY(hipMemset(st.bz->as<float>() + c, 0, 4));
        }
      }
    }
    HIP_TRY(hipDeviceSynchronize());
    fp8_outlier_split = true;
  }
#endif
  fp8_ready = true;
  return 0;
}

// option "fp8" = 2: bf16 images of the e4m3-quantised weights (value = e4m3 * per-channel scale, rounded to bf16)
int lemas_dit::dequantize_fp8() {
  if (fp8_wonly_ready) return 0;
  // the weights-only accuracy point dequantises the COMPLETE e4m3 images: if the outlier decomposition zeroed the flagged rows, quantise afresh
  // without it (and leave the images marked stale, so that the next fp8 = 1 prepare() builds the decomposition again)
  const bool split_mode = fp8_outlier_mode;
  if (fp8_ready && fp8_outlier_split) fp8_ready = false;
  fp8_outlier_mode = false;
  const int qrc = quantize_fp8();
  fp8_outlier_mode = split_mode;
  if (split_mode) fp8_ready = false;
  RC_TRY(qrc);
  const int d = cfg.dim, in = inner(), ffd = cfg.ff_mult * d;
  hipStream_t s = nullptr;
  for (auto& b : blocks) {
    RC_TRY(b.wqkvq.ensure((size_t)3 * in * d * 2));
    HIP_TRY(launch_f8_to_bf16(b.wqkv8.as<uint8_t>(), b.sqkv.as<float>(), 3 * in, d, b.wqkvq.as<bf16_t>(), s));
    RC_TRY(b.woq.ensure((size_t)d * in * 2));
    HIP_TRY(launch_f8_to_bf16(b.wo8.as<uint8_t>(), b.so.as<float>(), d, in, b.woq.as<bf16_t>(), s));
    RC_TRY(b.w1q.ensure((size_t)ffd * d * 2));
    HIP_TRY(launch_f8_to_bf16(b.w18.as<uint8_t>(), b.s1.as<float>(), ffd, d, b.w1q.as<bf16_t>(), s));
    RC_TRY(b.w2q.ensure((size_t)d * ffd * 2));
    HIP_TRY(launch_f8_to_bf16(b.w28.as<uint8_t>(), b.s2.as<float>(), d, ffd, b.w2q.as<bf16_t>(), s));
  }
  HIP_TRY(hipStreamSynchronize(s));
  fp8_wonly_ready = true;
  return 0;
}

int lemas_dit::finalize() {
  RC_TRY(ws.check_complete());
  const int d = cfg.dim, in = inner(), ffd = cfg.ff_mult * d;
  hipStream_t s = nullptr;
  HIP_TRY(hipDeviceSynchronize());   // a solve() may still be running on the caller's stream: nothing below may free memory under it
  drop_graphs();   // weights may have been reloaded: every cached graph baked the old tensors' addresses
  prepared = false;
  tgrid_cached.clear();
  for (auto& b : blocks)
    for (DevBuf* w : block_bufs(b)) w->release();
  blocks.clear();
  blocks.resize(cfg.depth);
  for (auto& b : blocks)
    for (DevBuf* w : block_bufs(b)) w->moved = &moved;
  fp8_ready = false;
  fp8_wonly_ready = false;
  for (int i = 0; i < cfg.depth; ++i) {
    const std::string p = T("transformer_blocks." + std::to_string(i) + ".");
    BlockW& b = blocks[i];
    RC_TRY(b.wqkv.ensure((size_t)3 * in * d * 2));
    RC_TRY(b.bqkv.ensure((size_t)3 * in * 4));
    int j = 0;
    for (const char* n : {"to_q", "to_k", "to_v"}) {
      HIP_TRY(launch_f32_to_bf16(ws.ptr(p + "attn." + n + ".weight"), b.wqkv.as<bf16_t>() + (size_t)j * in * d, (size_t)in * d, s));
      HIP_TRY(hipMemcpyAsync(b.bqkv.as<float>() + (size_t)j * in, ws.ptr(p + "attn." + n + ".bias"), (size_t)in * 4,
                             hipMemcpyDeviceToDevice, s));
      ++j;
    }
    RC_TRY(b.wo.ensure((size_t)d * in * 2));
    HIP_TRY(launch_f32_to_bf16(ws.ptr(p + "attn.to_out.0.weight"), b.wo.as<bf16_t>(), (size_t)d * in, s));
    RC_TRY(b.w1.ensure((size_t)ffd * d * 2));
    HIP_TRY(launch_f32_to_bf16(ws.ptr(p + "ff.ff.0.0.weight"), b.w1.as<bf16_t>(), (size_t)ffd * d, s));
    RC_TRY(b.w2.ensure((size_t)d * ffd * 2));
    HIP_TRY(launch_f32_to_bf16(ws.ptr(p + "ff.ff.2.weight"), b.w2.as<bf16_t>(), (size_t)d * ffd, s));
    b.bo = ws.ptr(p + "attn.to_out.0.bias");
    b.b1 = ws.ptr(p + "ff.ff.0.0.bias");
    b.b2 = ws.ptr(p + "ff.ff.2.bias");
  }
  // proj_out padded to 128 output rows (zero rows beyond mel_dim; DevBuf memsets to 0)
  RC_TRY(wproj_out.ensure((size_t)128 * d * 2));
  RC_TRY(bproj_out.ensure((size_t)128 * 4));
  HIP_TRY(launch_f32_to_bf16(ws.ptr(T("proj_out.weight")), wproj_out.as<bf16_t>(), (size_t)cfg.mel_dim * d, s));
  HIP_TRY(hipMemcpyAsync(bproj_out.p, ws.ptr(T("proj_out.bias")), (size_t)cfg.mel_dim * 4, hipMemcpyDeviceToDevice, s));
  const int cg = d / cfg.conv_pos_groups;
  for (int j = 0; j < 2; ++j) {
    RC_TRY(wconv[j].ensure((size_t)d * cg * cfg.conv_pos_kernel * 2));
    HIP_TRY(launch_convpos_weight(ws.ptr(T("input_embed.conv_pos_embed.conv1d." + std::to_string(j * 2) + ".weight")),
                                  wconv[j].as<bf16_t>(), d, cg, cfg.conv_pos_kernel, s));
  }
  {
    std::vector<const float*> hw(cfg.depth), hb(cfg.depth);
    for (int l = 0; l < cfg.depth; ++l) {
      const std::string p = T("transformer_blocks." + std::to_string(l) + ".attn_norm.linear.");
      hw[l] = ws.ptr(p + "weight");
      hb[l] = ws.ptr(p + "bias");
    }
    RC_TRY(d_tabW.ensure((size_t)cfg.depth * sizeof(float*)));
    RC_TRY(d_tabB.ensure((size_t)cfg.depth * sizeof(float*)));
    HIP_TRY(hipMemcpy(d_tabW.p, hw.data(), (size_t)cfg.depth * sizeof(float*), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_tabB.p, hb.data(), (size_t)cfg.depth * sizeof(float*), hipMemcpyHostToDevice));
  }
  RC_TRY(d_step.ensure(64));
  if (!ln_err_host) {
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&ln_err_host), 64, hipHostMallocMapped));
    *ln_err_host = 0;
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&ln_err_dev), ln_err_host, 0));
    int dev = 0;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    n_cus = prop.multiProcessorCount;
  }
  if (!s2) {
    HIP_TRY(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    // the extra lanes of "lane_split" too: created here, never inside enqueue_forward(), which also runs under stream capture
    for (int i = 0; i < MAX_LANES - 2; ++i) {
      HIP_TRY(hipStreamCreateWithFlags(&sx[i], hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&ev_joinx[i], hipEventDisableTiming));
    }
  }
  HIP_TRY(hipStreamSynchronize(s));
  tab_stride = fold_off(cfg.depth);
  finalized = true;
  return 0;
}

// time MLP + AdaLN vectors for every step, dt / cfg_t tables, rotary table
int lemas_dit::build_tables(const lemas_sample_args* a, hipStream_t s) {
  const int d = cfg.dim;
  const int Snew = a->steps;
  std::vector<float> tg(a->t_grid, a->t_grid + Snew + 1);
  h_dt.resize(Snew);
  h_cfg.resize(Snew);
  for (int k = 0; k < Snew; ++k) {
    h_dt[k] = tg[k + 1] - tg[k];                 // torchdiffeq: dt = t1 - t0 in the grid dtype (fp32)
    const float omt = 1.0f - tg[k];
    h_cfg[k] = a->cfg_strength * (omt * omt);    // cfm.py:420
  }
  RC_TRY(d_dt.ensure((size_t)Snew * 4));
  RC_TRY(d_cfg.ensure((size_t)Snew * 4));
  HIP_TRY(hipMemcpyAsync(d_dt.p, h_dt.data(), (size_t)Snew * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_cfg.p, h_cfg.data(), (size_t)Snew * 4, hipMemcpyHostToDevice, s));

  if (!table_cache || tg != tgrid_cached) {
    h_t = tg;
    RC_TRY(d_t.ensure((size_t)(Snew + 1) * 4));
    HIP_TRY(hipMemcpyAsync(d_t.p, h_t.data(), (size_t)(Snew + 1) * 4, hipMemcpyHostToDevice, s));
    const int fd = cfg.time_freq_dim;
    RC_TRY(d_sin.ensure((size_t)Snew * fd * 4));
    RC_TRY(d_h1.ensure((size_t)Snew * d * 4));
    RC_TRY(d_temb.ensure((size_t)Snew * d * 4));
    RC_TRY(d_st.ensure((size_t)Snew * d * 4));
    RC_TRY(d_tab.ensure((size_t)Snew * tab_stride * 4));
    HIP_TRY(launch_time_sinus(d_t.as<float>(), ws.ptr(T("time_embed.freqs")), Snew, fd / 2, d_sin.as<float>(), s));
    GemmF32Params g{};
    g.A = d_sin.as<float>(); g.lda = fd; g.W = ws.ptr(T("time_embed.time_mlp.0.weight")); g.ldw = fd;
    g.bias = ws.ptr(T("time_embed.time_mlp.0.bias")); g.out = d_h1.as<float>(); g.ldc = d; g.M = Snew; g.N = d; g.K = fd;
    HIP_TRY(launch_gemm_f32(F32_BIAS_SILU, g, s));
    g.A = d_h1.as<float>(); g.lda = d; g.W = ws.ptr(T("time_embed.time_mlp.2.weight")); g.ldw = d;
    g.bias = ws.ptr(T("time_embed.time_mlp.2.bias")); g.out = d_temb.as<float>(); g.K = d;
    HIP_TRY(launch_gemm_f32(F32_BIAS, g, s));
    HIP_TRY(launch_silu(d_temb.as<float>(), d_st.as<float>(), (size_t)Snew * d, s));
    g.A = d_st.as<float>(); g.lda = d; g.ldw = d; g.K = d; g.ldc = tab_stride;
    // the depth attn_norm.linear GEMMs [S, d] x [6 d, d]^T as ONE launch (grid.z = block): 96 workgroups each would leave the
    // chip mostly idle 22 times in a row
    g.N = 6 * d; g.out = d_tab.as<float>();
    g.nbatch = cfg.depth; g.Wv = d_tabW.as<const float*>(); g.biasv = d_tabB.as<const float*>(); g.out_bstride = (size_t)6 * d;
    HIP_TRY(launch_gemm_f32(F32_BIAS, g, s));
    g.nbatch = 0; g.Wv = nullptr; g.biasv = nullptr;
    g.W = ws.ptr(T("norm_out.linear.weight")); g.bias = ws.ptr(T("norm_out.linear.bias")); g.N = 2 * d;
    g.out = d_tab.as<float>() + (size_t)cfg.depth * 6 * d;
    HIP_TRY(launch_gemm_f32(F32_BIAS, g, s));
    if (fold_on()) RC_TRY(build_fold_tables(Snew, s));
    tgrid_cached = tg;
  }
  if (rope_n != a->frames) {
    const int half = cfg.dim_head / 2;
    // sized by the 128-row bucket (as every per-shape buffer below): a new length inside a bucket must not move a buffer, which would
    // invalidate the engine's cached step graphs
    const size_t rope_rows = ((size_t)a->frames + 127) & ~(size_t)127;
    RC_TRY(d_rope_cos.ensure(rope_rows * half * 4));
    RC_TRY(d_rope_sin.ensure(rope_rows * half * 4));
    HIP_TRY(launch_rope_table(d_rope_cos.as<float>(), d_rope_sin.as<float>(), a->frames, half, ws.ptr(T("rotary_embed.inv_freq")), s));
    rope_n = a->frames;
  }
  return 0;
}

// ln fold: c1 / c2 rows of every (block, LayerNorm site, ODE step) -- see ln_fold_split_kernel (norm_elementwise.hip).  One split
// launch, one grouped GEMM launch over the 2 x depth sites against the bf16 weights the step loop uses, one combine launch.
int lemas_dit::build_fold_tables(int Snew, hipStream_t s) {
  const int d = cfg.dim, in = inner(), ffd = cfg.ff_mult * d, depth = cfg.depth;
  const int nsites = 2 * depth, ra = 4 * Snew;
  RC_TRY(d_foldA.ensure((size_t)nsites * ra * d * 2));
  RC_TRY(d_foldtmp.ensure((size_t)depth * ra * (3 * in + ffd) * 4));
  RC_TRY(d_foldsites.ensure((size_t)nsites * sizeof(LnFoldSite)));
  RC_TRY(d_foldparams.ensure((size_t)nsites * sizeof(GemmParams)));
  RC_TRY(d_zero.ensure((size_t)(3 * in > ffd ? 3 * in : ffd) * 4));     // zero-filled at allocation, never written
  h_foldsites.assign(nsites, LnFoldSite{});
  h_foldparams.assign(nsites, GemmParams{});
  size_t off = 0;
  int max_n = 0;
  for (int l = 0; l < depth; ++l)
    for (int site = 0; site < 2; ++site) {
      const BlockW& w = blocks[l];
      const int base = l * 6 * d, n = site == 0 ? 3 * in : ffd, idx = 2 * l + site;
      LnFoldSite& st = h_foldsites[idx];
      st.bias = site == 0 ? w.bqkv.as<float>() : w.b1;
      st.tmp = d_foldtmp.as<float>() + off;
      st.N = n;
      st.scale_off = base + (site == 0 ? d : 4 * d);        // [shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp]
      st.shift_off = base + (site == 0 ? 0 : 3 * d);
      st.c1_off = fold_off(l) + (site == 0 ? 0 : 6 * in);
      st.c2_off = st.c1_off + n;
      GemmParams& g = h_foldparams[idx];
      g.A = d_foldA.as<bf16_t>() + (size_t)idx * ra * d;
      const DevBuf& wb = site == 0 ? (fp8_wonly ? w.wqkvq : w.wqkv) : (fp8_wonly ? w.w1q : w.w1);
      g.W = wb.as<bf16_t>();
      g.bias = d_zero.as<float>();
      g.M = ra; g.N = n; g.K = d; g.n_valid = n; g.out_f32 = d_foldtmp.as<float>() + off; g.ldc = n;
      g.seq_pitch = (ra + 127) & ~127; g.seq_valid = ra; g.batch = 1; g.xcd_gx = 8;
      off += (size_t)ra * n;
      max_n = n > max_n ? n : max_n;
    }
  HIP_TRY(hipMemcpyAsync(d_foldsites.p, h_foldsites.data(), (size_t)nsites * sizeof(LnFoldSite), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_foldparams.p, h_foldparams.data(), (size_t)nsites * sizeof(GemmParams), hipMemcpyHostToDevice, s));
  HIP_TRY(launch_ln_fold_split(d_tab.as<float>(), tab_stride, Snew, d, d_foldsites.as<LnFoldSite>(), nsites, d_foldA.as<bf16_t>(), s));
  HIP_TRY(launch_gemm_bf16_group(d_foldparams.as<GemmParams>(), nsites, ((ra + 127) / 128) * (max_n / 128), s));
  HIP_TRY(launch_ln_fold_combine(d_foldsites.as<LnFoldSite>(), nsites, max_n, Snew, d_tab.as<float>(), tab_stride, s));
  return 0;
}

// TextEmbedding for both CFG branches (dit.py:51-81) + prosody text conditioning (dit.py:225-233)
int lemas_dit::text_embed(const lemas_sample_args* a, hipStream_t s) {
  const int td = cfg.text_dim, rows = BB * N, branches = BB / B;
  const size_t arows = (size_t)BB * pitch;        // allocation size: the bucket's rows (see build_tables)
  RC_TRY(d_te.ensure(arows * td * 4));
  RC_TRY(d_rowmask.ensure(arows));
  HIP_TRY(launch_text_gather(a->text, B, Nt, N, td, branches, ws.ptr(T("text_embed.text_embed.weight")), cfg.vocab_rows,
                             ws.ptr(T("text_embed.freqs_cis")), 4096, d_te.as<float>(), d_rowmask.as<uint8_t>(), s));
  if (cfg.conv_layers > 0) {
    RC_TRY(d_t1.ensure(arows * td * 4));
    RC_TRY(d_t2.ensure(arows * td * 4));
    RC_TRY(d_t3.ensure(arows * 2 * td * 4));
    RC_TRY(d_gx.ensure((size_t)BB * 16 * 2 * td * 4));   // [BB][GRN_SPLIT = 16][2 td] partial sums of squares
    for (int i = 0; i < cfg.conv_layers; ++i) {
      const std::string p = T("text_embed.text_blocks." + std::to_string(i) + ".");
      HIP_TRY(launch_dwconv7(d_te.as<float>(), ws.ptr(p + "dwconv.weight"), ws.ptr(p + "dwconv.bias"), d_t1.as<float>(), BB, N, td, s));
      HIP_TRY(launch_ln_affine(d_t1.as<float>(), ws.ptr(p + "norm.weight"), ws.ptr(p + "norm.bias"), d_t2.as<float>(), rows, td, s));
      GemmF32Params g{};
      g.A = d_t2.as<float>(); g.lda = td; g.W = ws.ptr(p + "pwconv1.weight"); g.ldw = td; g.bias = ws.ptr(p + "pwconv1.bias");
      g.out = d_t3.as<float>(); g.ldc = 2 * td; g.M = rows; g.N = 2 * td; g.K = td;
      HIP_TRY(launch_gemm_f32(F32_BIAS_GELU, g, s));
      HIP_TRY(launch_grn(d_t3.as<float>(), d_gx.as<float>(), ws.ptr(p + "grn.gamma"), ws.ptr(p + "grn.beta"), BB, N, 2 * td, s));
      GemmF32Params h{};
      h.A = d_t3.as<float>(); h.lda = 2 * td; h.W = ws.ptr(p + "pwconv2.weight"); h.ldw = 2 * td; h.bias = ws.ptr(p + "pwconv2.bias");
      h.out = d_te.as<float>(); h.ldc = td; h.M = rows; h.N = td; h.K = 2 * td;
      h.res = d_te.as<float>(); h.ldres = td; h.rowmask = d_rowmask.as<uint8_t>();
      HIP_TRY(launch_gemm_f32(F32_BIAS_RES_SCALE, h, s));
    }
  }
  if (a->prosody && cfg.has_prosody) {
    RC_TRY(d_pt.ensure((size_t)B * td * 4));
    GemmF32Params g{};
    g.A = a->prosody; g.lda = 512; g.W = ws.ptr(T("prosody_text_proj.weight")); g.ldw = 512; g.bias = ws.ptr(T("prosody_text_proj.bias"));
    g.out = d_pt.as<float>(); g.ldc = td; g.M = B; g.N = td; g.K = 512;
    HIP_TRY(launch_gemm_f32(F32_BIAS, g, s));
    const int nlim = Nt < N ? Nt : N;
    HIP_TRY(launch_add_rowvec(d_te.as<float>(), d_pt.as<float>(), BB, B, N, td, nlim, s));
  }
  return 0;
}

// skip_dead 2: the rows the block chain keeps alive = the sample's own frames and one more 128-row block behind them
__global__ void live_len_kernel(const int* __restrict__ len, int* __restrict__ live, int B, int halo, int N) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b < B) live[b] = min(((len[b] + 127) & ~127) + halo, N);
}

int lemas_dit::prepare(const lemas_sample_args* a, hipStream_t s) {
  if (!finalized) { set_error("lemas_dit_prepare: weights not finalized"); return LEMAS_E_STATE; }
  RC_TRY(health());
  if (a->struct_size != sizeof(lemas_sample_args)) {
    set_error("lemas_dit_prepare: lemas_sample_args.struct_size is %u, this library (ABI %d) expects %zu -- set struct_size = sizeof(lemas_sample_args) "
              "and rebuild the client against include/lemas_hip.h", a->struct_size, lemas_version(), sizeof(lemas_sample_args));
    return LEMAS_E_ARG;
  }
  if (a->cond_rows < 0 || a->cond_rows > a->frames) { set_error("lemas_dit_prepare: cond_rows %d outside [0, frames = %d]", a->cond_rows, a->frames); return LEMAS_E_ARG; }
  if (a->batch <= 0 || a->frames <= 0 || a->frames > 4096 || a->steps <= 0 || a->text_len <= 0 || !a->cond || !a->cond_mask ||
      !a->text || !a->t_grid || a->cond_frames <= 0 || a->cond_frames > a->frames) {
    set_error("lemas_dit_prepare: bad arguments (B=%d N=%d F=%d Nt=%d S=%d)", a->batch, a->frames, a->cond_frames, a->text_len, a->steps);
    return LEMAS_E_ARG;
  }
  for (int k = 0; k < a->steps; ++k)      // torchdiffeq asserts a strictly monotone grid (SURVEY.md 8a row a-O); the sampler's is increasing
    if (!(a->t_grid[k + 1] > a->t_grid[k])) {
      set_error("lemas_dit_prepare: t_grid must be strictly monotone increasing (t[%d] = %g, t[%d] = %g)", k, (double)a->t_grid[k], k + 1,
                (double)a->t_grid[k + 1]);
      return LEMAS_E_ARG;
    }
  if (cfg.dim != 1024 || cfg.dim_head != 64 || cfg.dim / cfg.conv_pos_groups != 64) {
    set_error("lemas_dit: kernels are specialised for dim 1024, dim_head 64, 64 channels per conv group");
    return LEMAS_E_ARG;
  }
  B = a->batch; N = a->frames; F = a->cond_frames; Nt = a->text_len; S = a->steps;
  use_cfg = !(a->cfg_strength < 1e-5f);
  BB = use_cfg ? 2 * B : B;
  has_len = a->seq_len != nullptr;
  pitch = (N + 127) & ~127;
  npad = pitch;
  const int d = cfg.dim, md = cfg.mel_dim, td = cfg.text_dim, rows = BB * pitch, in = inner();

  if (fp8_wonly) RC_TRY(dequantize_fp8());      // before the tables: the ln-fold rows are sums over the weights the step loop multiplies with
  RC_TRY(build_tables(a, s));
  if (has_len) {
    RC_TRY(d_len.ensure((size_t)B * 4));
    RC_TRY(d_live.ensure((size_t)B * 4));
    HIP_TRY(hipMemcpyAsync(d_len.p, a->seq_len, (size_t)B * 4, hipMemcpyDeviceToDevice, s));
    // skip_dead == 2 reads these; filled HERE, whatever the option says now, so that every path behind prepare() -- solve(), a bare
    // lemas_dit_forward(), an option change between the two -- finds them
    hipLaunchKernelGGL(live_len_kernel, dim3((B + 63) / 64), dim3(64), 0, s, d_len.as<int>(), d_live.as<int>(), B, 128, N);
    HIP_TRY(hipGetLastError());
  }
  // conditioning
  RC_TRY(d_cond_eff.ensure((size_t)B * pitch * md * 4));
  RC_TRY(d_step_cond.ensure((size_t)B * pitch * md * 4));
  const float* pm = nullptr;
  if (a->prosody && cfg.has_prosody && !a->prosody_text_only) {   // text-only: cond is final as given (cfm.py:320-324 overwrote the shifted mel)
    RC_TRY(d_pm.ensure((size_t)B * md * 4));
    GemmF32Params g{};
    g.A = a->prosody; g.lda = 512; g.W = ws.ptr("prosody_to_mel.weight"); g.ldw = 512; g.bias = nullptr;
    g.out = d_pm.as<float>(); g.ldc = md; g.M = B; g.N = md; g.K = 512;
    HIP_TRY(launch_gemm_f32(F32_BIAS, g, s));
    pm = d_pm.as<float>();
  }
  const int crows = a->cond_rows > 0 ? a->cond_rows : N;
  HIP_TRY(launch_cond_prepare(a->cond, a->cond_mask, pm, pm ? ws.ptr("prosody_to_mel.bias") : nullptr, B, N, F, md, crows,
                              d_cond_eff.as<float>(), d_step_cond.as<float>(), s));
  if (a->step_cond) {   // accent-GRL conditioning: step_cond = where(cond_mask, cond_grl, 0)  (cfm.py:387-388)
    RC_TRY(d_t1.ensure((size_t)B * pitch * md * 4));
    HIP_TRY(launch_cond_prepare(a->step_cond, a->cond_mask, nullptr, nullptr, B, N, F, md, crows, d_t1.as<float>(), d_step_cond.as<float>(), s));
  }
  RC_TRY(text_embed(a, s));
  // hoisted [cond | text] part of the input projection
  RC_TRY(d_ct.ensure((size_t)rows * (md + td) * 4));
  RC_TRY(d_pconst.ensure((size_t)rows * d * 4));
  HIP_TRY(launch_concat_ct(d_step_cond.as<float>(), d_te.as<float>(), B, N, md, td, BB / B, pitch, d_ct.as<float>(), s));
  {
    GemmF32Params g{};
    g.A = d_ct.as<float>(); g.lda = md + td; g.W = ws.ptr(T("input_embed.proj.weight")) + md; g.ldw = 2 * md + td;
    g.bias = ws.ptr(T("input_embed.proj.bias")); g.out = d_pconst.as<float>(); g.ldc = d; g.M = rows; g.N = d; g.K = md + td;
    HIP_TRY(launch_gemm_f32(F32_BIAS, g, s));
  }
  // step-loop workspaces
  RC_TRY(d_y.ensure((size_t)B * pitch * md * 4));
  RC_TRY(d_xres.ensure((size_t)rows * d * 4));
  RC_TRY(d_hbf.ensure((size_t)rows * d * 2));
  RC_TRY(d_abf.ensure((size_t)rows * in * 2));
  RC_TRY(d_q.ensure((size_t)rows * in * 2));
  RC_TRY(d_k.ensure((size_t)rows * in * 2));
  RC_TRY(d_vt.ensure((size_t)BB * cfg.heads * 64 * npad * 2));
  if (f8qk_mode() != 0) {
    RC_TRY(d_q8.ensure((size_t)rows * in)); RC_TRY(d_k8.ensure((size_t)rows * in));
    RC_TRY(d_qs8.ensure((size_t)rows * (in / 32))); RC_TRY(d_ks8.ensure((size_t)rows * (in / 32)));
  }
  RC_TRY(d_ff.ensure((size_t)rows * cfg.ff_mult * d * 2));
  RC_TRY(d_cmid.ensure((size_t)rows * d * 2));
  RC_TRY(d_pred.ensure((size_t)rows * md * 4));
  RC_TRY(d_lncnt.ensure((size_t)cfg.depth * 2 * 2 * (rows / 64 + 1) * sizeof(unsigned int)));   // >= [block][site][lane][panel of >= 64 rows]
  RC_TRY(d_lnpart.ensure((size_t)rows * (d / 32) * 2 * 4));
  if (fp8) {
    RC_TRY(quantize_fp8());
    RC_TRY(d_h8.ensure((size_t)rows * d));
    RC_TRY(d_hmx.ensure((size_t)rows * (d / 32)));
    RC_TRY(d_a8.ensure((size_t)rows * in));
    RC_TRY(d_amx.ensure((size_t)rows * (in / 32)));
    RC_TRY(d_ff8.ensure((size_t)rows * cfg.ff_mult * d));
    RC_TRY(d_ffmx.ensure((size_t)rows * (cfg.ff_mult * d / 32)));
  }
  prepared = true;
  return 0;
}

#ifdef LEMAS_PHASE_TIMESTAMPS   // measurement builds: in-situ timeline of one forward pass (lemas_k_timeline of the test library, include/lemas_hip_test.h)
static unsigned long long* g_tl = nullptr;
static int g_tl_slots = 0, g_tl_next = 0;
int lemas_internal_timeline(void* buf, int slots) { g_tl = static_cast<unsigned long long*>(buf); g_tl_slots = slots; return 0; }
#define TL_SLOT(P) do { (P).dbg = (g_tl && g_tl_next < g_tl_slots) ? g_tl + (size_t)(g_tl_next++) * 4096 : nullptr; } while (0)
#define TL_RESET() do { g_tl_next = 0; } while (0)
#else
int lemas_internal_timeline(void*, int) { set_error("lemas_k_timeline: not a measurement build"); return LEMAS_E_STATE; }
#define TL_SLOT(P) do { } while (0)
#define TL_RESET() do { } while (0)
#endif

int lemas_dit::enqueue_forward(hipStream_t s) {
  TL_RESET();
  const int d = cfg.dim, md = cfg.mel_dim, in = inner(), ffd = cfg.ff_mult * d;
  const int* step = d_step.as<int>();
  const float* tab = d_tab.as<float>();
  // input projection, x part (K = mel_dim) in fp32, broadcast onto both CFG branches  (dit.py:97)
  {
    RC_TRY(pbegin(PC_INPROJ, s));
    GemmF32Params g{};
    g.A = d_y.as<float>(); g.lda = md; g.W = ws.ptr(T("input_embed.proj.weight")); g.ldw = 2 * md + cfg.text_dim;
    g.out = d_xres.as<float>(); g.ldc = d; g.M = B * pitch; g.N = d; g.K = md;
    if (use_cfg) {
      g.add = d_pconst.as<float>();
      HIP_TRY(launch_gemm_f32(F32_BIAS_ADD2, g, s));
    } else {
      g.res = d_pconst.as<float>(); g.ldres = d;
      HIP_TRY(launch_gemm_f32(F32_BIAS_RES_SCALE, g, s));
    }
    RC_TRY(pend(s));
  }
  // From here on the two CFG branches (conditional rows, unconditional rows) are independent chains.  With "dual" on
  // they run as two concurrent lanes (second HIP stream / parallel hipGraph branch): each GEMM then has one round of
  // ~120 tiles, and one lane's epilogue / prologue / launch gap overlaps the other lane's K loop.
  const int lanes = lanes_for(s);
  // profile mode keeps the per-lane launch shapes but runs the lanes back to back on one stream, so the HIP events
  // around a launch time that kernel alone (two concurrent streams would add the other lane's queueing to it)
  const bool fork = lanes >= 2 && !profile;
  hipStream_t st[MAX_LANES];
  for (int ln = 0; ln < lanes; ++ln) {
    if (fork && ln >= 2 && !sx[ln - 2]) { set_error("lane streams are created by finalize(): call order"); return LEMAS_E_STATE; }
    st[ln] = !fork || ln == 0 ? s : ln == 1 ? s2 : sx[ln - 2];
  }
  const int bh = BB / lanes;                 // samples (branch-rows) per lane
  const int rows = bh * pitch;
  // per-sample length arrays ([B], shared by the two CFG branches) as a lane sees them: lane ln holds samples (ln % k) * bh .. of its branch
  const int len_batch = lanes == 1 ? B : bh;
  auto len_off = [&](int ln) { return lanes == 1 ? 0 : (ln % (lanes / 2)) * bh; };
  // LayerNorm tails inside the gate + residual GEMM launches: only when every workgroup of such a launch -- of BOTH lanes, which
  // run the same kind of launch at about the same time -- is resident at once (a tail waits for the other column tiles of its
  // row panel).  configs[1]: 2 x 120 workgroups of 96 KB LDS on 256 CUs.  Batched shapes keep the separate ln_mod launches.
  int ln_panels = 0;
  bool fuse_ln = false;
  const bool fold = fold_on();
  if (ln_fused && !fp8 && !fold && lanes <= 2) {      // (measurement builds; the counters are sized for at most two lanes)
    GemmParams t{};
    t.M = rows; t.N = d; t.K = in; t.n_valid = d; t.ldc = d; t.concurrency = lanes; t.tile = d == 1024 ? opt_tile_n1024 : 0;
    int per_cu = 1;
    const int wgs = gemm_bf16_ln_fusable(t, &ln_panels, &per_cu);
    fuse_ln = wgs > 0 && (long)lanes * wgs <= (long)n_cus * per_cu;
  }
  unsigned int* lncnt = nullptr;
  if (fuse_ln) {
    const size_t nb = (size_t)cfg.depth * 2 * lanes * ln_panels * sizeof(unsigned int);
    if (nb > d_lncnt.bytes) { set_error("lemas_dit: LayerNorm-tail counters were not sized by prepare()"); return LEMAS_E_STATE; }
    lncnt = d_lncnt.as<unsigned int>();
    HIP_TRY(hipMemsetAsync(lncnt, 0, nb, s));      // ahead of the fork: both lanes start from zeroed counters (a memset node in the graph)
  }
  auto ln_site = [&](int l, int site, int ln) { return lncnt + ((size_t)(l * 2 + site) * lanes + ln) * ln_panels; };
  // block_persist: 16 barrier words per (block, lane), zeroed by a memset node ahead of the fork (the LayerNorm tails' counter buffer)
  const bool persist = block_persist != 0 && !fp8 && !fold && !fuse_ln && !has_len && s != nullptr && cfg.dim == 1024 && lanes <= 2;
  unsigned int* psync = nullptr;
  if (persist) {
    const size_t nb = (size_t)cfg.depth * lanes * 16 * sizeof(unsigned int);
    if (nb > d_lncnt.bytes) { set_error("lemas_dit: barrier words were not sized by prepare()"); return LEMAS_E_STATE; }
    psync = d_lncnt.as<unsigned int>();
    HIP_TRY(hipMemsetAsync(psync, 0, nb, s));
  }
  if (fork) {
    HIP_TRY(hipEventRecord(ev_fork, s));
    for (int ln = 1; ln < lanes; ++ln) HIP_TRY(hipStreamWaitEvent(st[ln], ev_fork, 0));
  }

  auto convpos = [&](int ln) -> int {        // conv position embedding + residual (dit.py:98)
    hipStream_t q = st[ln];
    const size_t r0 = (size_t)ln * rows;
    RC_TRY(pbegin(PC_CONVPOS, q));
    ConvPosParams c{};
    c.b2 = bh; c.n = N; c.pitch = pitch; c.channels = d; c.groups = cfg.conv_pos_groups; c.taps = cfg.conv_pos_kernel;
    c.in_f32 = d_xres.as<float>() + r0 * d; c.w = wconv[0].as<bf16_t>(); c.bias = ws.ptr(T("input_embed.conv_pos_embed.conv1d.0.bias"));
    c.out_bf16 = d_cmid.as<bf16_t>() + r0 * d;
    HIP_TRY(launch_convpos(c, q));
    c.in_f32 = nullptr; c.in_bf16 = d_cmid.as<bf16_t>() + r0 * d; c.w = wconv[1].as<bf16_t>();
    c.bias = ws.ptr(T("input_embed.conv_pos_embed.conv1d.2.bias")); c.out_bf16 = nullptr;
    c.out_f32 = d_xres.as<float>() + r0 * d; c.residual = d_xres.as<float>() + r0 * d;
    HIP_TRY(launch_convpos(c, q));
    RC_TRY(pend(q));
    return 0;
  };

  // lane_skew 1: every stage; 10 + S: ONCE per ODE step -- lane 1's first launch of block 0 waits for lane 0's stage S of block 0 (0 QK+V, 1 attention,
  // 2 out-projection, 3 FF1, 4 FF2), after which the lanes run S + 1 stages apart without another edge
  const bool skew = fork && lane_skew == 1 && lanes == 2;
  const int skew_once = (fork && lane_skew >= 10 && lane_skew <= 14 && lanes == 2) ? lane_skew - 10 : -1;
  if (skew || skew_once >= 0)
    for (auto& e : ev_skew)
      if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  int skew_k = 0, skew_l = 0;
  // around every launch of a block: lane 0 records "stage k done", lane 1 waits for it before its own stage k
  auto skew_pre = [&](int ln, hipStream_t q) -> int {
    if (skew && ln == 1) HIP_TRY(hipStreamWaitEvent(q, ev_skew[skew_k & 7], 0));
    if (skew_once >= 0 && ln == 1 && skew_l == 0 && skew_k == 0) HIP_TRY(hipStreamWaitEvent(q, ev_skew[skew_once], 0));
    return 0;
  };
  auto skew_post = [&](int ln, hipStream_t q) -> int {
    if ((skew || (skew_once >= 0 && skew_l == 0 && skew_k == skew_once)) && ln == 0) HIP_TRY(hipEventRecord(ev_skew[skew_k & 7], q));
    ++skew_k;
    return 0;
  };
  // Ragged batch: the 128-row blocks that lie wholly in a sample's padding are not computed by the block chain (GemmParams::live_len; the
  // reference computes them and trims each sample to its duration afterwards, utils_infer.py:579-585).  Their x rows stay what the input
  // embedding made them, so the head below still emits finite (and unused) rows there.  bf16 chain without the LayerNorm options only.
  const bool can_skip = has_len && !fp8 && !fold && !fuse_ln;
  const int* live = (can_skip && skip_dead) ? (skip_dead == 2 ? d_live.as<int>() : d_len.as<int>()) : nullptr;
  // The ATTENTION HALF of a block contributes exactly nothing to rows past a sample's length -- the reference zeroes the out-projection's
  // output there (modules.py AttnProcessor: x.masked_fill(~mask, 0) behind to_out; EPI_GATE_RES "rows past kv_len contribute 0") -- and nothing
  // else reads what it computes for them (keys are masked, q rows are only their own).  So attn_norm, the QK / V projections, attention and
  // the out-projection skip the padding blocks ALWAYS (option skip_masked, default 1): bit-identical results, padding rows included.  The
  // FF half does update padding rows in the reference; it skips them only under skip_dead.
  const int* live_a = (can_skip && skip_masked) ? d_len.as<int>() : nullptr;
  auto block = [&](int l, int ln) -> int {   // one DiTBlock (modules.py:627-641) on one lane's rows
    hipStream_t q = st[ln];
    skew_k = 0;
    skew_l = l;
    const size_t r0 = (size_t)ln * rows;
    float* xres = d_xres.as<float>() + r0 * d;
    bf16_t* hbf = d_hbf.as<bf16_t>() + r0 * d;
    bf16_t* abf = d_abf.as<bf16_t>() + r0 * in;
    bf16_t* ffb = d_ff.as<bf16_t>() + r0 * ffd;
    GemmParams g{};
    g.M = rows; g.tab = tab; g.tab_stride = tab_stride; g.step_idx = step; g.seq_pitch = pitch; g.seq_valid = N; g.batch = len_batch;
    const int lo = len_off(ln);
    const int* live_a_l = live_a ? live_a + lo : nullptr;
    const int* live_l = live ? live + lo : nullptr;
    const int* len_l = has_len ? d_len.as<int>() + lo : nullptr;
    g.live_len = live_a_l;      // attention half; the FF half switches to `live_l` below
    g.heads = cfg.heads; g.npad = npad; g.rope_cos = d_rope_cos.as<float>(); g.rope_sin = d_rope_sin.as<float>();
    // attention variants with bit 16 take q already multiplied by softmax_scale * log2(e): the QK epilogue does it before rounding
    g.q_scale = (attn_variant & 16) ? (1.0f / sqrtf((float)cfg.dim_head)) * 1.4426950408889634f : 0.f;
    g.q = d_q.as<bf16_t>() + r0 * in; g.k = d_k.as<bf16_t>() + r0 * in; g.vt = d_vt.as<bf16_t>() + r0 * in;
    AttnParams at{};
    at.q = g.q; at.k = g.k; at.vt = g.vt; at.out = abf;
    at.kv_len = len_l; at.b2 = bh; at.batch = len_batch; at.heads = cfg.heads; at.n = N; at.npad = npad; at.pitch = pitch;
    at.scale = 1.0f / sqrtf((float)cfg.dim_head); at.variant = attn_variant; at.live_len = live_a_l;
    const BlockW& w = blocks[l];
    const int base = l * 6 * d;  // [shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp] (modules.py:312)
    uint8_t* h8 = fp8 ? d_h8.as<uint8_t>() + r0 * d : nullptr;
    uint8_t* hmx = fp8 ? d_hmx.as<uint8_t>() + r0 * (d / 32) : nullptr;
    uint8_t* a8 = fp8 ? d_a8.as<uint8_t>() + r0 * in : nullptr;
    uint8_t* amx = fp8 ? d_amx.as<uint8_t>() + r0 * (in / 32) : nullptr;
    uint8_t* ff8 = fp8 ? d_ff8.as<uint8_t>() + r0 * ffd : nullptr;
    uint8_t* ffmx = fp8 ? d_ffmx.as<uint8_t>() + r0 * (ffd / 32) : nullptr;
    // which of the block's GEMMs take fp8 operands (fp8_sites(): every site the option mask names, or none once the outlier guard has tripped)
    const int sites = fp8_sites();
    const bool f8_qkv = sites & 1, f8_out = sites & 2, f8_ff1 = sites & 4, f8_ff2 = sites & 8;
    // attention's QK^T on the fp8 matrix path: q and k leave the QK epilogue as MXFP8 (mode 1) or are quantised by a side launch (mode 2)
    const int f8qk = f8qk_mode() == 1 && !f8_qkv ? 2 : f8qk_mode();      // (only the fp8 QK GEMM bodies carry the MXFP8 epilogue)
    if (f8qk != 0) {
      at.q8 = d_q8.as<uint8_t>() + r0 * in; at.k8 = d_k8.as<uint8_t>() + r0 * in;
      at.q8_mx = d_qs8.as<uint8_t>() + r0 * (in / 32); at.k8_mx = d_ks8.as<uint8_t>() + r0 * (in / 32);
      at.variant = attn_variant | ATTN_F8QK;
      if (f8qk == 1) {
        g.q8 = const_cast<uint8_t*>(at.q8); g.k8 = const_cast<uint8_t*>(at.k8);
        g.q8_mx = const_cast<uint8_t*>(at.q8_mx); g.k8_mx = const_cast<uint8_t*>(at.k8_mx);
      }
    }
    g.concurrency = lanes;
    g.xcd_gx = opt_xcd_gx;
    g.xcd_runs = opt_xcd_runs;
    auto tile_for = [&](int n) { return fp8 ? 0 : n == 1024 ? opt_tile_n1024 : n == 2048 ? opt_tile_n2048 : 0; };
    // A / W / their scales for one GEMM: bf16 operands, or (f8) MXFP8 activations x per-channel-scaled e4m3 weights
    auto operands = [&](bool f8, const bf16_t* abf16, const uint8_t* af8, const uint8_t* afmx, const DevBuf& wb, const DevBuf& w8, const DevBuf& wsc,
                        const DevBuf& wq, size_t row_off, int K) {
      g.f8 = f8 ? 1 : 0;
      if (f8) {
        g.A = reinterpret_cast<const bf16_t*>(af8); g.a_mx = afmx;
        g.W = reinterpret_cast<const bf16_t*>(w8.as<uint8_t>() + row_off * K); g.w_scale = wsc.as<float>() + row_off;
      } else {
        g.A = abf16; g.a_mx = nullptr; g.w_scale = nullptr; g.W = (fp8_wonly ? wq.as<bf16_t>() : wb.as<bf16_t>()) + row_off * K;
      }
    };
    float* lnpart = d_lnpart.as<float>() + r0 * (d / 32) * 2;
    const int fo = fold_off(l);
    if (fold) {                     // ln fold: only the chain's entry needs a launch; later blocks get xs + partial sums from FF2
      if (l == 0) {
        RC_TRY(pbegin(PC_LN, q));
        HIP_TRY(launch_ln_prep(xres, hbf, lnpart, rows, d, tab, tab_stride, base + d, step, q));
        RC_TRY(pend(q));
      }
      g.ln_part = lnpart; g.ln_np = d / 32;
    } else if (!(fuse_ln && l > 0) && !(ln_skip && l > 0)) {      // fused: block l's attn_norm rows were written by block l-1's FF2 launch
      RC_TRY(pbegin(PC_LN, q));
      if (f8_qkv) HIP_TRY(launch_ln_mod_f8(xres, h8, hmx, rows, d, tab, tab_stride, base + d, base, step, q));
      else HIP_TRY(launch_ln_mod(xres, hbf, rows, d, tab, tab_stride, base + d, base, step, q, live_a_l, pitch, len_batch));
      RC_TRY(pend(q));
    }
    // one launch for QK and V only while all of its workgroups fit the chip in one round (128 + 64 at configs[1]); beyond that two
    // separately tiled launches pack better (measured: -5 % at N = 2814 and at batch 8 when fused regardless)
    const long qkv_wgs = (long)((rows + 255) / 256) * (3 * in / 128);
    const bool fuse_qkv = qkv_fused && lanes >= 2 && qkv_wgs <= 250;
    if (fuse_qkv) {
      GemmParams gq = g, gv = g;
      RC_TRY(pkernel(PC_GEMM_QKV, &gq.ev_start, &gq.ev_stop));
      operands(f8_qkv, hbf, h8, hmx, w.wqkv, w.wqkv8, w.sqkv, w.wqkvq, 0, d);
      gq.A = g.A; gq.a_mx = g.a_mx; gq.W = g.W; gq.w_scale = g.w_scale; gq.f8 = g.f8;
      gq.bias = w.bqkv.as<float>(); gq.N = 2 * in; gq.K = d; gq.n_valid = 2 * in; gq.kv_len = nullptr;
      gq.lnc1_off = fo; gq.lnc2_off = fo + 3 * in; gv.lnc1_off = fo + 2 * in; gv.lnc2_off = fo + 5 * in;
      operands(f8_qkv, hbf, h8, hmx, w.wqkv, w.wqkv8, w.sqkv, w.wqkvq, (size_t)2 * in, d);
      gv.A = g.A; gv.a_mx = g.a_mx; gv.W = g.W; gv.w_scale = g.w_scale; gv.f8 = g.f8;
      gv.bias = w.bqkv.as<float>() + 2 * in; gv.N = in; gv.K = d; gv.n_valid = in; gv.kv_len = nullptr;
      gq.tile = gv.tile = opt_tile_qkv;
      TL_SLOT(gq);
#ifdef LEMAS_PHASE_TIMESTAMPS
      gv.dbg = gq.dbg;
#endif
      RC_TRY(skew_pre(ln, q));
      HIP_TRY(launch_gemm_qkv_fused(gq, gv, q));
      RC_TRY(skew_post(ln, q));
      g.K = d;
    } else {
      RC_TRY(pkernel(PC_GEMM_QK, &g.ev_start, &g.ev_stop));
      operands(f8_qkv, hbf, h8, hmx, w.wqkv, w.wqkv8, w.sqkv, w.wqkvq, 0, d);
      g.bias = w.bqkv.as<float>(); g.N = 2 * in; g.K = d; g.n_valid = 2 * in;
      g.kv_len = nullptr; g.tile = tile_for(g.N); g.lnc1_off = fo; g.lnc2_off = fo + 3 * in;
      TL_SLOT(g);
      HIP_TRY(launch_gemm_bf16(EPI_QK_ROPE, g, q));
      RC_TRY(pkernel(PC_GEMM_V, &g.ev_start, &g.ev_stop));
      operands(f8_qkv, hbf, h8, hmx, w.wqkv, w.wqkv8, w.sqkv, w.wqkvq, (size_t)2 * in, d);
      g.bias = w.bqkv.as<float>() + 2 * in; g.N = in; g.n_valid = in; g.tile = tile_for(g.N); g.lnc1_off = fo + 2 * in; g.lnc2_off = fo + 5 * in;
      TL_SLOT(g);
      HIP_TRY(launch_gemm_bf16(EPI_V_T, g, q));
    }
    g.ln_part = nullptr;
    RC_TRY(pkernel(PC_ATTN, &at.ev_start, &at.ev_stop));
    // fp8 on an outlier checkpoint (outlier_rows.hip): the producers of the two residual-writing projections' inputs write bf16, and ONE side
    // launch per site computes the flagged output channels from those bf16 rows and writes their MXFP8 image for the fp8 GEMM of the site
    const bool orows = outlier_rows_on();
    at.out8 = (f8_out && !orows) ? a8 : nullptr; at.out_mx = (f8_out && !orows) ? amx : nullptr;
    auto outlier_rows = [&](const bf16_t* A, const DevBuf& wside, const DevBuf& bside, int K, int gate_off, const int* kvl, uint8_t* q8, uint8_t* qmx) -> int {
      OutlierRowsParams o{};
      o.A = A; o.W = wside.as<bf16_t>(); o.bias = bside.as<float>(); o.chan = d_flagged.as<int>(); o.nf = fp8_outlier_channels;
      o.x = xres; o.ldx = d; o.M = rows; o.K = K; o.tab = tab; o.tab_stride = tab_stride; o.gate_off = gate_off; o.step_idx = step;
      o.kv_len = kvl; o.seq_pitch = pitch; o.seq_valid = N; o.batch = len_batch; o.a8 = q8; o.amx = qmx;
#ifdef LEMAS_MEASUREMENT_BUILD
      HIP_TRY(launch_outlier_rows(o, q));
      return 0;
#else
      (void)o;
      set_error("the fp8 outlier decomposition exists in measurement builds only");
      return LEMAS_E_STATE;
#endif
    };
    TL_SLOT(at);
    RC_TRY(skew_pre(ln, q));
    if (f8qk == 2) HIP_TRY(launch_qk_mx8(g.q, g.k, const_cast<uint8_t*>(at.q8), const_cast<uint8_t*>(at.k8), const_cast<uint8_t*>(at.q8_mx),
                                         const_cast<uint8_t*>(at.k8_mx), (size_t)rows * in / 64, q));
    HIP_TRY(launch_attention(at, q));
    RC_TRY(skew_post(ln, q));
    RC_TRY(pkernel(PC_GEMM_OUT, &g.ev_start, &g.ev_stop));
    operands(f8_out, abf, a8, amx, w.wo, w.wo8, w.so, w.woq, 0, in);
    g.bias = (f8_out && orows) ? w.bo_z.as<float>() : w.bo; g.N = d; g.K = in; g.n_valid = d;
    g.out_f32 = xres; g.ldc = d; g.gate_off = base + 2 * d; g.kv_len = len_l; g.tile = tile_for(g.N);
    if (fuse_ln) {   // ff_norm (modules.py:637) as the tail of the out-projection launch
      g.ln_out = hbf; g.ln_scale_off = base + 4 * d; g.ln_shift_off = base + 3 * d; g.ln_cnt = ln_site(l, 0, ln); g.ln_err = ln_err_dev;
    }
    if (fold) { g.xs_out = hbf; g.xs_scale_off = base + 4 * d; g.ln_part_out = lnpart; g.ln_np = d / 32; }     // ff_norm's scale (modules.py:637)
    TL_SLOT(g);
    RC_TRY(skew_pre(ln, q));
    GemmParams g_out = g;
    if (persist) { g_out.tile = 0; g_out.ev_start = g_out.ev_stop = nullptr; }
    else {
      // (the side launch first: it writes the MXFP8 rows the fp8 GEMM reads; the two update disjoint columns of x)
      if (f8_out && orows) RC_TRY(outlier_rows(abf, w.wo_side, w.bo_side, in, base + 2 * d, len_l, a8, amx));
      HIP_TRY(launch_gemm_bf16(EPI_GATE_RES, g, q));
    }
    RC_TRY(skew_post(ln, q));
    g.ln_out = nullptr; g.xs_out = nullptr;
    g.live_len = live_l;        // FF half
    if (!fuse_ln && !fold && !persist && !ln_skip) {
      RC_TRY(pbegin(PC_LN, q));
      if (f8_ff1) HIP_TRY(launch_ln_mod_f8(xres, h8, hmx, rows, d, tab, tab_stride, base + 4 * d, base + 3 * d, step, q));
      else HIP_TRY(launch_ln_mod(xres, hbf, rows, d, tab, tab_stride, base + 4 * d, base + 3 * d, step, q, live_l, pitch, len_batch));
      RC_TRY(pend(q));
    }
    RC_TRY(pkernel(PC_GEMM_FF1, &g.ev_start, &g.ev_stop));
    operands(f8_ff1, hbf, h8, hmx, w.w1, w.w18, w.s1, w.w1q, 0, d);
    g.bias = w.b1; g.N = ffd; g.K = d; g.n_valid = ffd;
    g.out_bf16 = ffb; g.out_f8 = ff8; g.out_mx = ffmx; g.ldc = ffd; g.kv_len = nullptr; g.tile = tile_for(g.N);
    // FF1 writes what FF2 reads: MXFP8 straight from its epilogue -- except on an outlier checkpoint, where it writes bf16 and FF2's side launch quantises
    const bool ff1_f8_out = f8_ff2 && !(orows && !f8_ff1);
    if (fold) { g.ln_part = lnpart; g.lnc1_off = fo + 6 * in; g.lnc2_off = fo + 6 * in + ffd; }
    TL_SLOT(g);
    RC_TRY(skew_pre(ln, q));
    GemmParams g_ff1 = g;
    if (!persist) HIP_TRY(launch_gemm_bf16(ff1_f8_out ? EPI_BIAS_GELU_F8 : EPI_BIAS_GELU_BF16, g, q));
    RC_TRY(skew_post(ln, q));
    g.ln_part = nullptr;
    if (f8_ff2 && orows) RC_TRY(outlier_rows(ffb, w.w2_side, w.b2_side, ffd, base + 5 * d, nullptr, ff8, ffmx));
    RC_TRY(pkernel(PC_GEMM_FF2, &g.ev_start, &g.ev_stop));
    operands(f8_ff2, ffb, ff8, ffmx, w.w2, w.w28, w.s2, w.w2q, 0, ffd);
    g.bias = (f8_ff2 && orows) ? w.b2_z.as<float>() : w.b2; g.N = d; g.K = ffd; g.n_valid = d;
    g.out_f32 = xres; g.ldc = d; g.gate_off = base + 5 * d; g.tile = tile_for(g.N);
    if (fuse_ln) {   // the next block's attn_norm (modules.py:314: shift, scale first), or the final norm (:335: scale, shift) after the last
      const int nb = (l + 1) * 6 * d, fb = cfg.depth * 6 * d;
      g.ln_out = hbf; g.ln_cnt = ln_site(l, 1, ln); g.ln_err = ln_err_dev;
      if (l + 1 < cfg.depth) { g.ln_scale_off = nb + d; g.ln_shift_off = nb; }
      else { g.ln_scale_off = fb; g.ln_shift_off = fb + d; }
    }
    // ln fold: the next block's attn_norm scale (modules.py:314); the final norm after the last block stays a launch of its own
    if (fold && l + 1 < cfg.depth) { g.xs_out = hbf; g.xs_scale_off = (l + 1) * 6 * d + d; g.ln_part_out = lnpart; g.ln_np = d / 32; }
    TL_SLOT(g);
    RC_TRY(skew_pre(ln, q));
    if (persist) {      // out-projection -> ff_norm -> FF1 -> FF2 as one persistent launch (measurement build)
      GemmParams g_ff2 = g;
      g_ff1.ev_start = g_ff1.ev_stop = g_ff2.ev_start = g_ff2.ev_stop = nullptr;
      g_ff1.tile = g_ff2.tile = 0;
      // every workgroup of the launch waits for the others: both lanes' launches must fit the chip at one workgroup (96 KB of LDS) per CU
      const int wgs = xcd_grid((rows + 127) / 128, d / 128, opt_xcd_gx ? opt_xcd_gx : pick_xcd_gx(rows, d));
      if ((long)lanes * wgs > (long)n_cus) { set_error("lemas_dit: block_persist needs %d x %d workgroups co-resident on %d CUs", lanes, wgs, n_cus); return LEMAS_E_STATE; }
      HIP_TRY(launch_gemm_chain_ffhalf(g_out, g_ff1, g_ff2, base + 4 * d, base + 3 * d, psync + ((size_t)l * lanes + ln) * 16, ln_err_dev,
                                       block_persist == 2 ? 1 : 0, nullptr, q));
    } else {
      HIP_TRY(launch_gemm_bf16(EPI_GATE_RES, g, q));
    }
    RC_TRY(skew_post(ln, q));
    g.ln_out = nullptr; g.xs_out = nullptr;
    return 0;
  };

  auto head = [&](int ln) -> int {           // final AdaLN (order scale, shift: modules.py:333) + proj_out
    hipStream_t q = st[ln];
    const size_t r0 = (size_t)ln * rows;
    const int fb = cfg.depth * 6 * d;
    if (!fuse_ln) {                // fused: the last block's FF2 launch wrote the final norm's rows
      RC_TRY(pbegin(PC_LN, q));
      HIP_TRY(launch_ln_mod(d_xres.as<float>() + r0 * d, d_hbf.as<bf16_t>() + r0 * d, rows, d, tab, tab_stride, fb, fb + d, step, q));
      RC_TRY(pend(q));
    }
    RC_TRY(pbegin(PC_GEMM_FINAL, q));
    GemmParams g{};
    g.M = rows; g.seq_pitch = pitch; g.seq_valid = N; g.batch = B; g.heads = cfg.heads; g.npad = npad;
    g.A = d_hbf.as<bf16_t>() + r0 * d; g.W = wproj_out.as<bf16_t>(); g.bias = bproj_out.as<float>(); g.N = 128; g.K = d; g.n_valid = md;
    g.out_f32 = d_pred.as<float>() + r0 * md; g.ldc = md;
    HIP_TRY(launch_gemm_bf16(EPI_BIAS_F32, g, q));
    RC_TRY(pend(q));
    return 0;
  };

  for (int ln = 0; ln < lanes; ++ln) RC_TRY(convpos(ln));
  for (int l = 0; l < cfg.depth; ++l)
    for (int ln = 0; ln < lanes; ++ln) RC_TRY(block(l, ln));
  for (int ln = 0; ln < lanes; ++ln) RC_TRY(head(ln));
  if (fork) {
    for (int ln = 1; ln < lanes; ++ln) {
      hipEvent_t e = ln == 1 ? ev_join : ev_joinx[ln - 2];
      HIP_TRY(hipEventRecord(e, st[ln]));
      HIP_TRY(hipStreamWaitEvent(s, e, 0));
    }
  }
  return 0;
}

int lemas_dit::enqueue_update(float* traj, hipStream_t s) {
  RC_TRY(pbegin(PC_CFG_EULER, s));
  // traj: caller's dense [S+1, B, N, mel]; the step index is host-known only in the eager path (traj_k)
  HIP_TRY(launch_cfg_euler(d_y.as<float>(), d_pred.as<float>(), B * pitch, cfg.mel_dim, d_dt.as<float>(), d_cfg.as<float>(),
                           d_step.as<int>(), nullptr, use_cfg ? 1 : 0, s));
  RC_TRY(pend(s));
  if (traj) {
    const size_t w = (size_t)N * cfg.mel_dim * 4;
    HIP_TRY(hipMemcpy2DAsync(traj, w, d_y.p, (size_t)pitch * cfg.mel_dim * 4, w, B, hipMemcpyDeviceToDevice, s));
  }
  return 0;
}

// one ODE step (DiT forward of both CFG branches + CFG / clamp / Euler update) captured from stream s
int lemas_dit::capture_step(hipStream_t s, hipGraph_t* out) {
  hipGraph_t graph = nullptr;
  HIP_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  int rc = enqueue_forward(s);
  if (rc == 0) rc = enqueue_update(nullptr, s);
  hipError_t e = hipStreamEndCapture(s, &graph);
  if (rc != 0) { if (graph) (void)hipGraphDestroy(graph); return rc; }
  HIP_TRY(e);
  ++n_capture;
  *out = graph;
  return 0;
}

// the instantiated step graph for the prepared shape (see the cache's comment in the struct)
int lemas_dit::step_graph(hipStream_t s, hipGraphExec_t* exec, hipEvent_t* done) {
  if (graph_generation != moved) {  // one of THIS engine's buffers moved: every captured address is suspect
    drop_graphs();
    graph_generation = moved;
  }
  char key[112];
  snprintf(key, sizeof key, "B%d_P%d_cfg%d_len%d_dual%d_f8%d_ln%d_av%d", B, pitch, (int)use_cfg, (int)has_len, dual ? lanes_for(s) : 0, fp8 ? 16 + fp8_sites() : fp8_wonly ? 2 : 0,
           (int)ln_fused + 2 * (int)fold_on(), attn_variant);
  auto it = graphs.find(key);
  if (it == graphs.end()) {
    while ((int)graphs.size() >= (graph_cap > 0 ? graph_cap : 1)) {       // evict the least recently used bucket
      auto lru = graphs.begin();
      for (auto j = graphs.begin(); j != graphs.end(); ++j)
        if (j->second.used < lru->second.used) lru = j;
      drop_bucket(lru->second);
      graphs.erase(lru);
      ++n_evict;
    }
    it = graphs.emplace(key, GraphBucket{}).first;
  }
  GraphBucket& b = it->second;
  b.used = ++graph_tick;
  StepGraph* g = nullptr;
  for (auto& c : b.g)
    if (c.exec && c.N == N) g = &c;
  if (!g) {
    // an empty slot first (the bucket's second instantiated graph costs one instantiate, once), else the slot used longer ago
    g = !b.g[0].exec ? &b.g[0] : !b.g[1].exec ? &b.g[1] : (b.g[0].used <= b.g[1].used ? &b.g[0] : &b.g[1]);
    hipGraph_t graph = nullptr;
    RC_TRY(capture_step(s, &graph));
    bool patched = false;
    if (g->exec && graph_update) {
      HIP_TRY(hipEventSynchronize(g->done));          // its last launch (two lengths ago in this bucket) has long finished
      hipGraphNode_t bad = nullptr;
      hipGraphExecUpdateResult res = hipGraphExecUpdateError;
      patched = hipGraphExecUpdate(g->exec, graph, &bad, &res) == hipSuccess && res == hipGraphExecUpdateSuccess;
      if (patched) ++n_update;
      else { (void)hipGetLastError(); ++n_update_fail; }
    }
    if (!patched) {
      if (g->exec) { HIP_TRY(hipEventSynchronize(g->done)); (void)hipGraphExecDestroy(g->exec); g->exec = nullptr; }
      hipGraphExec_t ex = nullptr;
      hipError_t e = hipGraphInstantiate(&ex, graph, nullptr, nullptr, 0);
      if (e != hipSuccess) { (void)hipGraphDestroy(graph); HIP_TRY(e); }
      g->exec = ex;
      ++n_instantiate;
    }
    HIP_TRY(hipGraphDestroy(graph));
    if (!g->done) HIP_TRY(hipEventCreateWithFlags(&g->done, hipEventDisableTiming));
    g->N = N;
  }
  g->used = graph_tick;
  *exec = g->exec;
  *done = g->done;
  return 0;
}

int lemas_dit::solve(const lemas_sample_args* a, hipStream_t s) {
  if (!finalized || !prepared) { set_error("lemas_dit_solve: prepare() has not run on the finalized weights"); return LEMAS_E_STATE; }
  RC_TRY(health());
  if (a->struct_size != sizeof(lemas_sample_args)) { set_error("lemas_dit_solve: lemas_sample_args.struct_size mismatch (see lemas_dit_prepare)"); return LEMAS_E_ARG; }
  if (a->batch != B || a->frames != N || a->steps != S || !a->y) { set_error("lemas_dit_solve: arguments differ from prepare()"); return LEMAS_E_ARG; }
  const float* y_src = a->y_init ? a->y_init : a->y;
  const size_t ybytes = (size_t)B * N * cfg.mel_dim * 4;
  const size_t yw = (size_t)N * cfg.mel_dim * 4, ypitch = (size_t)pitch * cfg.mel_dim * 4;
  HIP_TRY(hipMemsetAsync(d_y.p, 0, (size_t)B * ypitch, s));   // padding rows restart from 0 every utterance
  HIP_TRY(hipMemcpy2DAsync(d_y.p, ypitch, y_src, yw, yw, B, hipMemcpyDeviceToDevice, s));
  if (a->trajectory) HIP_TRY(hipMemcpyAsync(a->trajectory, y_src, ybytes, hipMemcpyDeviceToDevice, s));
  HIP_TRY(launch_step_set(d_step.as<int>(), 0, s));

  const bool graph_ok = use_graph && !profile && !a->trajectory && s != nullptr;  // the legacy NULL stream cannot be captured
  if (graph_ok) {
    if (graph_generation != moved) {  // one of THIS engine's buffers moved: every captured address is suspect
      drop_graphs();
      graph_generation = moved;
    }
    hipGraphExec_t exec = nullptr;
    hipEvent_t done = nullptr;
    RC_TRY(step_graph(s, &exec, &done));
    // `done` fences this graph's launches for a later hipGraphExecUpdate / destroy: recorded also when a launch fails mid-loop, so that a
    // graph whose earlier launches are still running is never patched or destroyed under them
    hipError_t el = hipSuccess;
    for (int k = 0; k < S && el == hipSuccess; ++k) el = hipGraphLaunch(exec, s);
    const hipError_t er = hipEventRecord(done, s);
    HIP_TRY(el);
    HIP_TRY(er);
  } else {
    for (int k = 0; k < S; ++k) {
      RC_TRY(enqueue_forward(s));
      RC_TRY(enqueue_update(a->trajectory ? a->trajectory + (size_t)(k + 1) * B * N * cfg.mel_dim : nullptr, s));
    }
  }
  HIP_TRY(hipMemcpy2DAsync(a->y, yw, d_y.p, ypitch, yw, B, hipMemcpyDeviceToDevice, s));
  if (a->out)
    HIP_TRY(launch_select_rows(a->out, d_cond_eff.as<float>(), d_y.as<float>(), a->cond_mask, B, N, pitch, cfg.mel_dim, s));
  return 0;
}

// ------------------------------------------------------------------------------------------ C ABI
extern "C" {

int lemas_dit_create(const lemas_dit_config* cfg, lemas_dit** out) {
  if (!cfg || !out) { set_error("lemas_dit_create: null argument"); return LEMAS_E_ARG; }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    set_error("lemas_dit_create: no HIP device (this library has no CPU path)");
    return e != hipSuccess ? -(int)e : LEMAS_E_STATE;
  }
  RC_TRY(kernels_init());
  lemas_dit* m = new lemas_dit();
  m->cfg = *cfg;
  m->declare_schema();
  *out = m;
  return 0;
}
void lemas_dit_destroy(lemas_dit* m) { delete m; }

static int dit_load(lemas_dit* m, const char* name, const float* src, const int64_t* shape, int32_t ndim, bool on_device) {
  if (!m || !name || !src) { set_error("lemas_dit_load_weight: null argument"); return LEMAS_E_ARG; }
  // loaded-but-unused tensors of the reference checkpoint (cfm.py:171 accent classifier) are accepted and dropped
  if (strncmp(name, "accent_classifier.", 18) == 0) return 0;
  // a (re)loaded tensor lives at a new address: nothing prepared or captured on the old weights may be replayed
  if (m->finalized || !m->graphs.empty()) HIP_TRY(hipDeviceSynchronize());   // a reload while a solve() is in flight: wait before touching what it uses
  m->finalized = false;
  m->prepared = false;
  m->drop_graphs();
  return m->ws.load(name, src, shape, ndim, on_device);
}
int lemas_dit_load_weight(lemas_dit* m, const char* name, const float* host, const int64_t* shape, int32_t ndim) {
  return dit_load(m, name, host, shape, ndim, false);
}
int lemas_dit_load_weight_device(lemas_dit* m, const char* name, const float* dev, const int64_t* shape, int32_t ndim) {
  return dit_load(m, name, dev, shape, ndim, true);
}
int lemas_dit_finalize(lemas_dit* m) { return m ? m->finalize() : LEMAS_E_ARG; }

int lemas_dit_set_option(lemas_dit* m, const char* key, int64_t value) {
  if (!m || !key) return LEMAS_E_ARG;
  if (!strcmp(key, "graph")) { m->use_graph = value != 0; return 0; }
  if (!strcmp(key, "table_cache")) { m->table_cache = value != 0; return 0; }
  if (!strcmp(key, "dual")) {
    m->dual = value != 0;
    m->drop_graphs();
    return 0;
  }
  if (!strcmp(key, "skip_masked")) {
    if (m->skip_masked != (value != 0)) {
      m->skip_masked = value != 0;
      m->drop_graphs();
    }
    return 0;
  }
  if (!strcmp(key, "skip_dead")) {
    if (value < 0 || value > 2) { set_error("lemas_dit_set_option: skip_dead is 0 (off), 1 (padding blocks skipped) or 2 (all but the first padding block skipped)"); return LEMAS_E_ARG; }
    if (m->skip_dead != (int)value) {         // (a caller may set it before every batch: only a CHANGE invalidates the captured graphs)
      m->skip_dead = (int)value;
      m->drop_graphs();
    }
    return 0;
  }
  if (!strcmp(key, "lane_split")) {
    if (value != 0 && value != 1 && value != 2 && value != 4) { set_error("lemas_dit_set_option: lane_split is 0 (automatic), 1, 2 or 4 sample groups per CFG branch"); return LEMAS_E_ARG; }
    m->lane_split = (int)value;
    m->drop_graphs();
    return 0;
  }
  if (!strcmp(key, "attn_f8qk")) {
    if (value < 0 || value > 6 || (value & 3) == 3) { set_error("lemas_dit_set_option: attn_f8qk is 0 (off), 1 (QK epilogue writes MXFP8), 2 (side launch), + 4 = on the bf16 path too"); return LEMAS_E_ARG; }
    m->attn_f8qk = (int)value;
    m->drop_graphs();
    return 0;
  }
  if (!strcmp(key, "qkv_fused")) {
    m->qkv_fused = value != 0;
    m->drop_graphs();
    return 0;
  }
#ifdef LEMAS_MEASUREMENT_BUILD
  if (!strcmp(key, "ln_skip")) { m->ln_skip = value != 0; m->drop_graphs(); return 0; }
  if (!strcmp(key, "block_persist")) {
    if (value < 0 || value > 2) { set_error("lemas_dit_set_option: block_persist is 0, 1 or 2 (with the weight prefetch across the barrier)"); return LEMAS_E_ARG; }
    m->block_persist = (int)value;
    m->drop_graphs();
    return 0;
  }
#endif
#ifndef LEMAS_MEASUREMENT_BUILD
  if (!strcmp(key, "ln_fused") || !strcmp(key, "lane_skew") || !strcmp(key, "xcd_runs") || !strcmp(key, "block_persist") || !strcmp(key, "fp8_outlier_mode") ||
      !strcmp(key, "ln_skip")) {
    if (value == 0) return 0;       // "off" is what the product does anyway
    set_error("lemas_dit_set_option: '%s' is a measurement option -- its code exists only in builds of the library with -DLEMAS_MEASUREMENT_BUILD "
              "(LEMAS_EXTRA_HIPCC_FLAGS, lemas_tts_amd/build.py); the product library does not carry it", key);
    return LEMAS_E_ARG;
  }
#endif
  {
    int* slot = !strcmp(key, "tile_n1024") ? &m->opt_tile_n1024 : !strcmp(key, "tile_n2048") ? &m->opt_tile_n2048
              : !strcmp(key, "tile_qkv") ? &m->opt_tile_qkv : !strcmp(key, "xcd_gx") ? &m->opt_xcd_gx : !strcmp(key, "xcd_runs") ? &m->opt_xcd_runs
              : !strcmp(key, "attn_variant") ? &m->attn_variant : !strcmp(key, "lane_skew") ? &m->lane_skew : nullptr;
    if (slot) {
      if (slot == &m->attn_variant && !attention_variant_ok((int)value)) {
        set_error("lemas_dit_set_option: attn_variant %lld is not a product variant (csrc/attention.hip; measurement-only variants exist in "
                  "-DLEMAS_PHASE_TIMESTAMPS builds of the test library only)", (long long)value);
        return LEMAS_E_ARG;
      }
      *slot = (int)value;
      m->drop_graphs();     // a captured graph baked the kernels of the old choice
      return 0;
    }
  }
  if (!strcmp(key, "ln_fused")) {
    m->ln_fused = value != 0;
    m->drop_graphs();
    return 0;
  }
  if (!strcmp(key, "ln_fold")) {
    m->ln_fold = value != 0;
    m->tgrid_cached.clear();      // the c1 / c2 rows of the table exist only while the fold is on
    m->prepared = false;
    m->drop_graphs();
    return 0;
  }
  if (!strcmp(key, "fp8")) {      // block GEMMs on the MXFP8 path; takes effect at the next prepare()
    if (value < 0 || value > 2) { set_error("lemas_dit_set_option: fp8 is 0 (bf16), 1 (MXFP8 GEMMs) or 2 (weights-only fp8 accuracy point)"); return LEMAS_E_ARG; }
    if (m->fp8 != (value == 1) || m->fp8_wonly != (value == 2)) { m->prepared = false; m->tgrid_cached.clear(); }   // (ln-fold rows follow the weight image)
    m->fp8 = value == 1;
    m->fp8_wonly = value == 2;
    return 0;
  }
  if (!strcmp(key, "graph_cache")) {
    if (value < 1 || value > 1024) { set_error("lemas_dit_set_option: graph_cache is the number of step-graph buckets kept, 1 .. 1024"); return LEMAS_E_ARG; }
    m->graph_cap = (int)value;
    return 0;
  }
  if (!strcmp(key, "graph_update")) { m->graph_update = value != 0; return 0; }
  if (!strcmp(key, "fp8_outlier_guard")) {
    if (m->fp8_guard != (value != 0)) { m->fp8_guard = value != 0; m->fp8_ready = false; m->prepared = false; }     // (the e4m3 images differ: zeroed rows or not)
    m->drop_graphs();
    return 0;
  }
#ifdef LEMAS_MEASUREMENT_BUILD
  if (!strcmp(key, "fp8_outlier_mode")) {       // 0 (default) = every block GEMM on bf16 operands when the guard trips; 1 = the mixed-precision decomposition
    if (m->fp8_outlier_mode != (value != 0)) { m->fp8_outlier_mode = value != 0; m->fp8_ready = false; m->prepared = false; m->drop_graphs(); }
    return 0;
  }
#endif
  if (!strcmp(key, "fp8_sites")) {
    if (value < 0 || value > 15) { set_error("lemas_dit_set_option: fp8_sites is a mask of GEMM sites (1 QKV, 2 out-projection, 4 FF1, 8 FF2), 0 .. 15"); return LEMAS_E_ARG; }
    m->fp8_sites_opt = (int)value;
    m->drop_graphs();
    return 0;
  }
  if (!strcmp(key, "profile")) {
    m->profile = value != 0;
    for (auto& r : m->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    m->prof.clear();
    return 0;
  }
  set_error("lemas_dit_set_option: unknown option '%s'", key);
  return LEMAS_E_ARG;
}

int lemas_dit_get_stat(lemas_dit* m, const char* key, int64_t* value) {
  if (!m || !key || !value) return LEMAS_E_ARG;
  if (!strcmp(key, "graph_captures")) { *value = m->n_capture; return 0; }
  if (!strcmp(key, "graph_instantiates")) { *value = m->n_instantiate; return 0; }
  if (!strcmp(key, "graph_updates")) { *value = m->n_update; return 0; }
  if (!strcmp(key, "graph_update_failures")) { *value = m->n_update_fail; return 0; }
  if (!strcmp(key, "graph_evictions")) { *value = m->n_evict; return 0; }
  if (!strcmp(key, "graph_buckets")) { *value = (int64_t)m->graphs.size(); return 0; }
  if (!strcmp(key, "fp8_outlier_channels")) { *value = m->fp8_ready ? m->fp8_outlier_channels : -1; return 0; }      // -1: weights not quantised yet
  if (!strcmp(key, "fp8_gemms_kept_bf16")) { *value = m->fp8 ? 4 - __builtin_popcount(m->fp8_sites()) : 0; return 0; }       // per DiT block
  set_error("lemas_dit_get_stat: unknown counter '%s'", key);
  return LEMAS_E_ARG;
}

int lemas_dit_prepare(lemas_dit* m, const lemas_sample_args* a, void* stream) {
  if (!m || !a) return LEMAS_E_ARG;
  return m->prepare(a, (hipStream_t)stream);
}
int lemas_dit_solve(lemas_dit* m, const lemas_sample_args* a, void* stream) {
  if (!m || !a) return LEMAS_E_ARG;
  return m->solve(a, (hipStream_t)stream);
}
int lemas_dit_sample(lemas_dit* m, const lemas_sample_args* a, void* stream) {
  if (!m || !a) return LEMAS_E_ARG;
  RC_TRY(m->prepare(a, (hipStream_t)stream));
  return m->solve(a, (hipStream_t)stream);
}

int lemas_dit_forward(lemas_dit* m, const float* x, int32_t step_index, float* pred, void* stream) {
  if (!m || !x || !pred) return LEMAS_E_ARG;
  if (!m->finalized || !m->prepared) { set_error("lemas_dit_forward: prepare() has not run on the finalized weights"); return LEMAS_E_STATE; }
  if (step_index < 0 || step_index >= m->S) { set_error("lemas_dit_forward: step index out of range"); return LEMAS_E_ARG; }
  hipStream_t s = (hipStream_t)stream;
  const size_t w = (size_t)m->N * m->cfg.mel_dim * 4, wp = (size_t)m->pitch * m->cfg.mel_dim * 4;
  HIP_TRY(hipMemsetAsync(m->d_y.p, 0, (size_t)m->B * wp, s));
  HIP_TRY(hipMemcpy2DAsync(m->d_y.p, wp, x, w, w, m->B, hipMemcpyDeviceToDevice, s));
  HIP_TRY(launch_step_set(m->d_step.as<int>(), step_index, s));
  RC_TRY(m->enqueue_forward(s));
  HIP_TRY(hipMemcpy2DAsync(pred, w, m->d_pred.p, wp, w, m->BB, hipMemcpyDeviceToDevice, s));
  return 0;
}

int lemas_dit_health(lemas_dit* m) {
  if (!m) return LEMAS_E_ARG;
  HIP_TRY(hipDeviceSynchronize());
  return m->health();
}

int lemas_dit_profile_read(lemas_dit* m, char (*names)[32], double* total_ms, int64_t* launches, int32_t cap) {
  if (!m || !names || !total_ms || !launches) return LEMAS_E_ARG;
  HIP_TRY(hipDeviceSynchronize());
  double acc[PC_COUNT] = {0};
  int64_t cnt[PC_COUNT] = {0};
  for (auto& r : m->prof) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
    acc[r.cls] += ms;
    cnt[r.cls] += 1;
  }
  int n = 0;
  for (int c = 0; c < PC_COUNT && n < cap; ++c, ++n) {
    snprintf(names[n], 32, "%s", kProfNames[c]);
    total_ms[n] = acc[c];
    launches[n] = cnt[c];
  }
  return n;
}

}  // extern "C"
