// MEASUREMENT-ONLY device code of gemm_bf16.hip: the FF half of a DiT block as one persistent launch (engine option "block_persist").
// Included by gemm_bf16.hip ONLY in measurement builds (-DLEMAS_MEASUREMENT_BUILD): the product library carries neither this kernel nor its
// option.  Measured in round 5 (-17 ... -44 %, profiles/r05/r05_block_persist.txt), kept reproducible.  It reuses gemm_body and the row
// helpers of the including file, which is why it is an include and not a translation unit.
#pragma once
#ifndef LEMAS_MEASUREMENT_BUILD
#error "gemm_bf16_exp_chain.h is measurement-only"
#endif

// ================================================================================================================
// MEASUREMENT (round 5; engine option "block_persist", measurement builds only): the FF half of a DiT block -- out-projection -> ff_norm ->
// FF1 -> FF2 (modules.py:635-639) -- as ONE persistent launch per CFG lane.  The four stages are the very bodies of the separate launches
// (gemm_body 128 x 128 with eight waves, ln_core.h rows), so results are bit-identical; what changes is what sits BETWEEN the stages: a grid
// barrier (XCD-hierarchical arrival counters, price list row "barrier-xcd" of MI355X_MICROARCH.md) instead of a kernel boundary, no launch
// ramp / drain per stage, and -- `prefetch` -- the next stage's first WEIGHT tiles already streaming into LDS while the barrier is awaited
// (weights never depend on the previous stage).  The question it answers: does removing three dependent-launch boundaries and their ramps
// buy more than three grid barriers cost?  (profiles/r05/r05_block_persist.txt)
//
// Visibility follows the recipe of the LayerNorm tail above: every stage publishes with write-through (sc1) stores, each storing wave
// drains them (vmcnt(0)) before the workgroup's ONE arrival; a released workgroup executes one agent-scope acquire (invalidates this CU's
// vector L1) before it reads what other workgroups wrote.  Progress: all workgroups of the launch (<= 136) wait for each other, so the
// launch needs them co-resident -- the other lane's kernels never wait on anything and always drain, a second persistent launch (the other
// lane) brings <= 136 more workgroups of 96 KB: 272 > 256 CUs do NOT fit, so the engine runs the lanes' persistent launches on 120 + 8
// surplus workgroups each only when 2 x grid <= CUs.  Every wait is bounded and flags the engine instead of hanging the queue.
struct ChainParams {
  GemmParams out, ff1, ff2;        // out-projection (EPI_GATE_RES), FF1 (EPI_BIAS_GELU_BF16), FF2 (EPI_GATE_RES): as for the separate launches
  int ln_scale_off, ln_shift_off;  // ff_norm's modulation vectors inside the AdaLN table row (its input is out.out_f32, its output ff1.A)
  unsigned int* sync;              // [16] zeroed before the launch: [0..7] group arrivals, [8] top arrivals, [9] generation
  unsigned int* err;               // sticky error word (host-visible)
  int prefetch;                    // 1 = request the next stage's first two weight K-tiles before waiting at the barrier
};

// one arrival per workgroup; returns false when the wait gave up (the caller must not touch data of the incomplete stage)
template <typename PRE>
__device__ __forceinline__ bool chain_barrier(const ChainParams& c, int phase, int* lds_flag, PRE&& prefetch) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's write-through stores have reached memory
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned nwg = gridDim.x, g = blockIdx.x & 7;
    const unsigned members = (nwg >> 3) + (g < (nwg & 7) ? 1u : 0u), groups = nwg < 8 ? nwg : 8u;
    gu32* grp = (gu32*)(c.sync + g);
    gu32* top = (gu32*)(c.sync + 8);
    gu32* gen = (gu32*)(c.sync + 9);
    const unsigned want = (unsigned)phase + 1u;
    if (__hip_atomic_fetch_add(grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == members * want) {
      if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == groups * want)
        __hip_atomic_store(gen, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  prefetch();      // every wave: the next stage's first weight K-tiles, requested between this workgroup's arrival and its wait
  if (threadIdx.x == 0) {
    gu32* gen = (gu32*)(c.sync + 9);
    const unsigned want = (unsigned)phase + 1u;
    unsigned spins = 0;
    int fail = 0;
    while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 18)) {
        __hip_atomic_store((gu32*)c.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        fail = 1;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // buffer_inv sc1: this CU's L1 forgets what other workgroups have since rewritten
    *lds_flag = fail;
  }
  __syncthreads();
  return *lds_flag == 0;
}

__global__ __launch_bounds__(512) void gemm_chain_ffhalf_kernel(const ChainParams c) {
  using C = TileCfg<T128x128>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RING = body_lds_base<EPI_GATE_RES, C::BM, C::BN, C::ST, C::WM, C::WN, false>();
  float* rs = reinterpret_cast<float*>(smem + RING);
  int* flag = reinterpret_cast<int*>(smem + RING + C::BM * 8);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // first tile of this workgroup in a stage (false: none)
  auto first_tile = [&](const GemmParams& p, int& tm, int& tn) {
    return xcd_tile_coords(blockIdx.x, (p.M + C::BM - 1) / C::BM, p.N / C::BN, p.xcd_gx, tm, tn);
  };
  auto stage = [&](auto epi_tag, const GemmParams& p, bool wpre) {
    constexpr int EPI = decltype(epi_tag)::value;
    const int tiles_m = (p.M + C::BM - 1) / C::BM, tiles_n = p.N / C::BN;
    const int grid = xcd_grid(tiles_m, tiles_n, p.xcd_gx);
    for (int b = blockIdx.x; b < grid; b += gridDim.x) {
      int tm, tn;
      if (!xcd_tile_coords(b, tiles_m, tiles_n, p.xcd_gx, tm, tn)) continue;
      if (wpre && b == (int)blockIdx.x) gemm_body<EPI, C::BM, C::BN, C::ST, C::WM, C::WN, true, false, true>(p, smem, tm * C::BM, tn * C::BN, rs);
      else gemm_body<EPI, C::BM, C::BN, C::ST, C::WM, C::WN, true, false>(p, smem, tm * C::BM, tn * C::BN, rs);
      __syncthreads();      // the epilogue slabs are retired before the next tile's ring stages land
    }
  };
  const bool pre = c.prefetch != 0;
  auto none = []() {};
  auto pre_w = [&](const GemmParams& p) {
    int tm, tn;
    if (first_tile(p, tm, tn)) gemm_prefetch_w<C::BM, C::BN, C::ST, C::WM, C::WN>(p, smem, tn * C::BN);
  };
  // ---- stage 0: x += gate_msa * (attn . Wo^T + bo)
  stage(std::integral_constant<int, EPI_GATE_RES>{}, c.out, false);
  // (FF1's first weight tiles go out HERE, ahead of the LayerNorm stage, which does not touch LDS: they land under it)
  if (pre) { if (!chain_barrier(c, 0, flag, [&]() { pre_w(c.ff1); })) return; }
  else if (!chain_barrier(c, 0, flag, none)) return;
  // ---- stage 1: h = LayerNorm(x) (1 + scale_mlp) + shift_mlp, two rows per wave
  {
    const GemmParams& p = c.out;
    const float* base = p.tab + (size_t)p.step_idx[0] * p.tab_stride;
    float4 a[LN_PER][2], b[LN_PER][2];
    ln_load_vec(base + c.ln_scale_off, lane, a);
    ln_load_vec(base + c.ln_shift_off, lane, b);
    bf16_t* hout = const_cast<bf16_t*>(c.ff1.A);
    for (int r0 = (blockIdx.x * 8 + wave) * 2; r0 < p.M; r0 += gridDim.x * 16) {
      u32x4 raw[2][LN_PER][2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        int row = r0 + q;
        row = row < p.M ? row : p.M - 1;
        ln_load_row_sc1(p.out_f32 + (size_t)row * LN_D, lane, raw[q]);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) ln_wait_row(raw[q]);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float4 v[LN_PER][2];
#pragma unroll
        for (int i = 0; i < LN_PER; ++i)
#pragma unroll
          for (int h = 0; h < 2; ++h) v[i][h] = __builtin_bit_cast(float4, raw[q][i][h]);
        ln_row_store(v, a, b, hout + (size_t)(r0 + q) * LN_D, lane, r0 + q < p.M);
      }
    }
  }
  if (!chain_barrier(c, 1, flag, none)) return;
  // ---- stage 2: ff = gelu_tanh(h . W1^T + b1)
  stage(std::integral_constant<int, EPI_BIAS_GELU_BF16>{}, c.ff1, pre);
  if (pre) { if (!chain_barrier(c, 2, flag, [&]() { pre_w(c.ff2); })) return; }
  else if (!chain_barrier(c, 2, flag, none)) return;
  // ---- stage 3: x += gate_mlp * (ff . W2^T + b2)
  stage(std::integral_constant<int, EPI_GATE_RES>{}, c.ff2, pre);
}
