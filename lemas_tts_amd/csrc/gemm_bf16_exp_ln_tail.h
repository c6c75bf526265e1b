// MEASUREMENT-ONLY device code of gemm_bf16.hip: the LayerNorm tail inside the gate + residual GEMM launch (engine option "ln_fused").
// Included by gemm_bf16.hip ONLY in measurement builds (-DLEMAS_MEASUREMENT_BUILD, lemas_tts_amd/build.py): the product library carries
// neither this code nor its option.  Measured, lost (profiles/r02, r03), kept reproducible.  Not a translation unit of its own because it
// instantiates inside gemm_kernel's epilogue and shares its LDS carve-up.
#pragma once
#ifndef LEMAS_MEASUREMENT_BUILD
#error "gemm_bf16_exp_ln_tail.h is measurement-only"
#endif

// ---------------------------------------------------------------- LayerNorm-modulate tail of the gate + residual GEMM
// The AdaLN-modulated LayerNorm behind every gated residual update (modules.py:635-637, :639 -> the next block's :314, the final
// :335) needs whole rows of x_res; a 128-column tile holds an eighth of one.  As its own launch it costs a lane's chain 9-13 us per
// site (5 us of latency-bound kernel between two ~2 us dependent-launch boundaries: profiles/r02/r02_timeline_step.txt) for 0.4 % of the
// FLOPs.  Here the GEMM launch finishes the job itself: the TILES_N workgroups that share a row panel meet at the panel's arrival
// counter once their x_res tiles are out (write-through stores, drained by every wave before the one arrival per workgroup), and
// each then normalises R = BM / TILES_N rows of the panel.  Visibility follows the recipe of the hardware guide (producer: sc1
// payload stores -> s_waitcnt vmcnt(0) in every storing wave -> barrier -> relaxed agent-scope arrival; consumer: relaxed poll by
// one lane -> barrier -> sc1 loads, which bypass this CU's L1; no XCD's L2 holds a line of x_res that another workgroup wrote in
// this launch, because a workgroup only ever read the columns it then overwrote and sc1 stores drop the line).
// Progress: a waiting workgroup depends only on workgroups of the SAME launch; the caller fuses only when all of them (and the
// other lane's) fit the chip at once (gemm_bf16_ln_fusable), kernels that do not wait always drain, and the wait gives up after
// ~50 ms with the sticky error word set instead of hanging the queue.
typedef __attribute__((address_space(1))) unsigned int gu32;

__device__ __forceinline__ void ln_load_row_sc1(const float* row, int lane, u32x4 (&v)[LN_PER][2]) {
  static_assert(LN_PER == 2, "row image");
  const char* p0 = reinterpret_cast<const char*>(row) + lane * 32;   // float4 index (lane + 64 i) * 2 + h -> byte lane * 32 + i * 2048 + h * 16
  const char* p1 = p0 + 2048;
  asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
               "global_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
               "global_load_dwordx4 %2, %5, off sc1\n\t"
               "global_load_dwordx4 %3, %5, off offset:16 sc1"
               : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[1][0]), "=&v"(v[1][1]) : "v"(p0), "v"(p1) : "memory");
}
// retire the asm loads above: the wait names every destination, so no consumer can be scheduled ahead of it
__device__ __forceinline__ void ln_wait_row(u32x4 (&v)[LN_PER][2]) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[1][0]), "+v"(v[1][1]) : : "memory");
}

// The option needs the device to itself: the co-residency estimate that enables it (engine_dit.hip: lanes x workgroups <= CUs x workgroups
// per CU) knows nothing of other tenants (a second process sharing the GPU, CU masks, a side-stream kernel holding CUs).  It is off by default.
template <int TBM, int TBN, int NW>
__device__ __forceinline__ void ln_tail(const GemmParams& p, int m0, int n0, char* ln_lds) {
  constexpr int TILES_N = LN_D / TBN, R = TBM / TILES_N, RW = R / NW;
  static_assert(TILES_N * TBN == LN_D && R * TILES_N == TBM && RW * NW == R && RW >= 1, "rows of a panel must divide over its workgroups and waves");
  constexpr int ROWS = RW >= 2 ? 2 : 1;     // rows in flight per wave (as the stand-alone kernel)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tm = m0 / TBM, tn = n0 / TBN;
  // publish this workgroup's x_res tile
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  gu32* cnt = (gu32*)(p.ln_cnt + tm);
  if (tid == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // the modulation vectors do not depend on the panel: in flight while the other tiles arrive
  const float* base = p.tab + (size_t)p.step_idx[0] * p.tab_stride;
  float4 a[LN_PER][2], b[LN_PER][2];
  ln_load_vec(base + p.ln_scale_off, lane, a);
  ln_load_vec(base + p.ln_shift_off, lane, b);
  // `gave_up` travels through LDS (the ring is free by now): a workgroup whose wait timed out must NOT normalise rows of a panel that is
  // incomplete -- it flags the engine (sticky, host-visible) and leaves ln_out alone
  int* gave_up = reinterpret_cast<int*>(ln_lds);
  if (tid == 0) {
    unsigned spins = 0;
    int fail = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)TILES_N) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1u << 16)) {
        __hip_atomic_store((gu32*)p.ln_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        fail = 1;
        break;
      }
    }
    *gave_up = fail;
  }
  __syncthreads();
  if (*gave_up) return;
  const int row0 = m0 + tn * R + wave * RW;
#pragma unroll
  for (int r = 0; r < RW; r += ROWS) {
    u32x4 raw[ROWS][LN_PER][2];
#pragma unroll
    for (int q = 0; q < ROWS; ++q) {
      int row = row0 + r + q;
      row = row < p.M ? row : p.M - 1;
      ln_load_row_sc1(p.out_f32 + (size_t)row * LN_D, lane, raw[q]);
    }
#pragma unroll
    for (int q = 0; q < ROWS; ++q) ln_wait_row(raw[q]);
#pragma unroll
    for (int q = 0; q < ROWS; ++q) {
      float4 v[LN_PER][2];
#pragma unroll
      for (int i = 0; i < LN_PER; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) v[i][h] = __builtin_bit_cast(float4, raw[q][i][h]);
      const int row = row0 + r + q;
      ln_row_store(v, a, b, p.ln_out + (size_t)row * LN_D, lane, row < p.M);
    }
  }
}
