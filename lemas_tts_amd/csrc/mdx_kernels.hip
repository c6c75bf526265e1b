// Kernels of the UVR5 MDX-Net separation network (ConvTDFNet), exact fp32 on the f32-input matrix cores (gfx950).
//
// Activations are planar [b][c][t][f] fp32 with the frequency axis innermost -- the reference's own layout after its transpose
// (uvr5/lib_v5/mdxnet.py:107), so the TDF linears over f (modules.py:55-70) are plain row-major GEMMs (gemm_f32.hip) and nothing is ever
// re-laid out.  The convolutions are implicit GEMMs
//     D[co][position] = sum over (tap, ci) of  W[co][ci][tap] . x[ci][position + tap]
// with positions as the MFMA's 16 columns (16 consecutive f of one t: 64-B coalesced stores) and output channels as its 16 rows:
//   * one workgroup = 4 waves = 4 consecutive t rows x TF = 16 MT consecutive f x 48 output channels (MT x 3 accumulator tiles per wave);
//   * K runs over chunks of 4 input channels (8 selectable: launch_geom); per chunk the workgroup stages the INPUT HALO ((3 S + KH) rows x ((TF-1) S + KW) columns per
//     channel) and the 4 x taps x 48 weight slab in LDS once, and all KH x KW taps read their operands from that one halo: activations cross
//     L2 -> LDS once per chunk, not once per tap (a 3 x 3 im2col GEMM moves 9x the bytes; this kernel's L2 pull is ~4 B/clk/CU against the
//     ~11 B/clk a CU gets);
//   * operands are fetched with ds_read_b32 (32 banks, two 32-lane groups): the 16 lanes of a k-group read 16 consecutive words, the second
//     k-group of a 32-lane group sits one channel plane away, and planes / weight rows are padded to == 16 (mod 32) words, so reads are
//     conflict-free.  An f32 MFMA 16x16x4 occupies the pipe for 32 cycles: MT + 3 operand reads feed 3 MT of them, LDS is far from a limit;
//   * two LDS buffers: the global loads of chunk k + 1 are issued before chunk k is multiplied and parked after it, one barrier per chunk.
// Inference BatchNorm (mdxnet.py:52) is folded into the weights and bias when the engine finalizes; ReLU is the epilogue.  Three geometries
// share the body: 3 x 3 stride 1 pad 1 (TFC, modules.py:12), 2 x 2 stride 2 (the encoder's down-sampling, mdxnet.py:74) and the transposed
// 2 x 2 stride 2 (mdxnet.py:90) as a 1 x 1 "convolution" onto 4 Cout columns ordered (co, dt, df): a lane then owns the whole 2 x 2 output
// patch of one (co, position), multiplies it with the skip tensor (mdxnet.py:117) and stores two float2.
#include "common.h"

namespace {

constexpr int NTW = 48;        // output columns per workgroup (3 MFMA tiles)
constexpr int CK = 8;          // input channels per K chunk

template <int KH, int S, int PAD, int MT, int CKT = CK>
struct Geom {
  static constexpr int KW = KH, TAPS = KH * KW, TF = 16 * MT;
  static constexpr int ROWS = 3 * S + KH;                                  // 4 output rows
  static constexpr int LEAD = PAD > 0 ? 4 : 0;                             // the halo starts LEAD columns left of the tile: 16-B aligned loads
  static constexpr int ROWP = ((TF - 1) * S + KW - PAD + LEAD + 3) & ~3;
  static constexpr int COL0 = LEAD - PAD;
  static constexpr int PLANE0 = ROWS * ROWP;
  static constexpr int PLANE = PLANE0 + ((16 - PLANE0 % 32) + 32) % 32;    // == 16 (mod 32): the two k-groups of a lane group on disjoint banks
  static constexpr int CIW0 = TAPS * NTW;
  static constexpr int CIW = CIW0 + ((16 - CIW0 % 32) + 32) % 32;
  static constexpr int Q = ROWP / 4;
  static constexpr int A_F4 = CKT * ROWS * Q, W_F4 = CKT * CIW / 4;
  static_assert(PLANE % 4 == 0 && ROWP % 4 == 0 && CIW % 4 == 0 && (CKT == 4 || CKT == 8), "16-B LDS stores");
  static constexpr int A_IT = (A_F4 + 255) / 256, W_IT = (W_F4 + 255) / 256;
};

template <int KH, int S, int PAD, int MT, bool UP, bool VEC, int CKT>
__global__ __launch_bounds__(256, CKT == 4 ? 3 : 2) void mdx_conv_kernel(const MdxConvParams p) {
  using G = Geom<KH, S, PAD, MT, CKT>;
  const int nchunks = p.nchunks * (CK / CKT);            // the slab is [ntile][ci][CIW] linear: a chunk of 4 channels is half a chunk of 8
  __shared__ __attribute__((aligned(16))) float As[2][CKT * G::PLANE];
  __shared__ __attribute__((aligned(16))) float Ws[2][CKT * G::CIW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lk = lane >> 4;
  const int f0 = blockIdx.x * G::TF, t0 = blockIdx.y * 4;
  const int b = blockIdx.z / p.ntiles, nt = blockIdx.z % p.ntiles;
  const float* xb = p.x + (size_t)b * p.Cin * p.Ti * p.Fi;
  const float* wt = p.w + (size_t)nt * p.nchunks * (CK * G::CIW);        // (p.nchunks counts chunks of CK = 8)
  const size_t plane = (size_t)p.Ti * p.Fi;

  // Loader slots: a thread owns A_IT float4 slots of the halo (channel ci, halo row r, 4 columns from 4 q) and W_IT float4 of the weight slab;
  // everything but the channel base is constant over the K loop.  Loads are UNCONDITIONAL from clamped (always valid) addresses and the
  // halo's zero padding is applied when a slot is parked in LDS: a load under a branch made the compiler wait for every load in flight
  // before issuing the next one (s_waitcnt vmcnt(0) in front of each global_load), i.e. eight serial L2 round trips per chunk.
  const float* a_ptr[G::A_IT];   // address of the slot in channel `ci` of chunk 0 (or xb when the slot is outside the image)
  int a_lds[G::A_IT];            // LDS word offset, -1: no slot
  int a_lim[G::A_IT];            // the slot's channel is inside the tensor while chunk * CK < a_lim
  unsigned a_ok = 0;             // bit 4 it + e: element e of slot it is inside the image (VEC: all four or none)
#pragma unroll
  for (int it = 0; it < G::A_IT; ++it) {
    const int idx = tid + it * 256;
    const int ci = idx / (G::ROWS * G::Q), rem = idx % (G::ROWS * G::Q), r = rem / G::Q, q = rem % G::Q;
    const int tin = t0 * S - PAD + r, fin = f0 * S - G::LEAD + 4 * q;
    const bool slot = idx < G::A_F4, rowin = slot && tin >= 0 && tin < p.Ti;
    a_lds[it] = slot ? ci * G::PLANE + r * G::ROWP + 4 * q : -1;
    a_lim[it] = slot ? p.Cin - ci : 0;
    a_ptr[it] = xb;
    if (VEC) {
      if (rowin && fin >= 0 && fin < p.Fi) { a_ok |= 0xFu << (4 * it); a_ptr[it] = xb + (size_t)ci * plane + (size_t)tin * p.Fi + fin; }
    } else if (rowin) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (fin + e >= 0 && fin + e < p.Fi) a_ok |= 1u << (4 * it + e);
      a_ptr[it] = xb + (size_t)ci * plane + (size_t)tin * p.Fi + fin;      // element e is read only where its bit is set
    }
  }
  const f32x4 zv = {0.f, 0.f, 0.f, 0.f};
  f32x4 ra[G::A_IT], rw[G::W_IT];
  const f32x4* wsrc = reinterpret_cast<const f32x4*>(wt);

  auto gload = [&](int chunk) {
#pragma unroll
    for (int it = 0; it < G::A_IT; ++it) {
      const bool cok = chunk * CKT < a_lim[it];
      const float* src = cok ? a_ptr[it] + (size_t)chunk * CKT * plane : xb;
      if (VEC) {
        ra[it] = *reinterpret_cast<const f32x4*>(src);
      } else {
        f32x4 v = zv;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (cok && (a_ok >> (4 * it + e) & 1)) v[e] = src[e];
        ra[it] = v;
      }
    }
#pragma unroll
    for (int it = 0; it < G::W_IT; ++it) {
      const int idx = tid + it * 256;
      rw[it] = wsrc[(size_t)chunk * (CKT * G::CIW / 4) + (idx < G::W_F4 ? idx : 0)];
    }
  };
  auto park = [&](int buf, int chunk) {
#pragma unroll
    for (int it = 0; it < G::A_IT; ++it) {
      if (a_lds[it] < 0) continue;
      f32x4 v = ra[it];
      if (VEC && !((a_ok >> (4 * it) & 1) && chunk * CKT < a_lim[it])) v = zv;
      reinterpret_cast<f32x4*>(As[buf])[a_lds[it] >> 2] = v;       // (index in float4 units: the compiler cannot see that a_lds is a multiple of 4
                                                                   //  and splits a store through &As[..][a_lds] into ds_write2_b32 pairs: 4-way bank conflicts)
    }
#pragma unroll
    for (int it = 0; it < G::W_IT; ++it) {
      const int idx = tid + it * 256;
      if (idx < G::W_F4) *reinterpret_cast<f32x4*>(&Ws[buf][idx * 4]) = rw[it];
    }
  };

  f32x4 acc[MT][3];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = zv;

  const int a_base = lk * G::PLANE + (wave * S) * G::ROWP + l15 * S + G::COL0;
  const int w_base = lk * G::CIW + l15;

  gload(0);
  park(0, 0);
  __syncthreads();
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int cur = chunk & 1;
    if (chunk + 1 < nchunks) gload(chunk + 1);
    const float* A = &As[cur][a_base];
    const float* W = &Ws[cur][w_base];
#pragma unroll
    for (int tap = 0; tap < G::TAPS; ++tap) {
      const int dt = tap / G::KW, df = tap % G::KW;
#pragma unroll
      for (int kk = 0; kk < CKT / 4; ++kk) {
        float av[MT], bv[3];
#pragma unroll
        for (int i = 0; i < MT; ++i) av[i] = A[kk * 4 * G::PLANE + dt * G::ROWP + i * 16 * S + df];
#pragma unroll
        for (int j = 0; j < 3; ++j) bv[j] = W[kk * 4 * G::CIW + tap * NTW + j * 16];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[j], av[i], acc[i][j], 0, 0, 0);
      }
    }
    if (chunk + 1 < nchunks) park(cur ^ 1, chunk + 1);
    __syncthreads();
  }

  // D rows = output columns n = nt * 48 + j * 16 + lk * 4 + r, D columns = positions (lane & 15)
  const int t = t0 + wave;
  if (t >= p.Tg) return;
  if (!UP) {
    float* ob = p.out + (size_t)b * p.Cout * p.Tg * p.Fg;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = nt * NTW + j * 16 + lk * 4 + r;
        if (co >= p.Cout) continue;
        const float bias = p.bias[co];
        float* orow = ob + ((size_t)co * p.Tg + t) * p.Fg;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int f = f0 + i * 16 + l15;
          if (f >= p.Fg) continue;
          float v = acc[i][j][r] + bias;
          if (p.relu) v = fmaxf(v, 0.f);
          orow[f] = v;
        }
      }
  } else {
    const int To = 2 * p.Tg, Fo = 2 * p.Fg;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int co = nt * (NTW / 4) + j * 4 + lk;
      if (co >= p.Cout) continue;
      const float bias = p.bias[co];
      const size_t cb = ((size_t)b * p.Cout + co) * To;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int f = f0 + i * 16 + l15;
        if (f >= p.Fg) continue;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const size_t o = (cb + 2 * t + dt) * Fo + 2 * f;
          float v0 = acc[i][j][dt * 2] + bias, v1 = acc[i][j][dt * 2 + 1] + bias;
          if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
          if (p.skip) {
            const float2 s = *reinterpret_cast<const float2*>(p.skip + o);
            v0 *= s.x; v1 *= s.y;
          }
          *reinterpret_cast<float2*>(p.out + o) = make_float2(v0, v1);
        }
      }
    }
  }
}

// ---- split-bf16 ("bf16x3") form of the 3x3 convolution -------------------------------------------------------------------------------
// Engine option "bf16x3" (lemas_mdx_set_option, default 0 = the exact kernel above).  Every fp32 operand x is carried as TWO bf16 numbers,
// hi = bf16(x) and lo = bf16(x - hi) (x = hi + lo to 2^-17 relative), packed in one dword (hi << 16 | lo) -- the bytes of an fp32, so nothing
// grows in memory or LDS -- and each product is three bf16 MFMAs: W_hi A_hi + W_hi A_lo + W_lo A_hi (the dropped W_lo A_lo is 2^-16 of the
// product).  v_mfma_f32_16x16x32_bf16 does 16 x 16 x 32 in 16 cycles where the f32-input MFMA does 16 x 16 x 4 in 32: 16x the rate, 5.3x after
// the three-way split, at ~2^-16 relative precision per product (fp32 accumulation as before) -- between TF32 (2^-11, what the reference's
// onnxruntime CUDA provider uses for convolutions by default) and fp32 (2^-24).  Weights are split on the host at finalize(); activations stay
// fp32 in global memory and are split when a halo is parked in LDS, once per element, not once per use.
// K = 32 of one MFMA = 2 taps x 16 input channels: lane group g = lane >> 4 takes tap 2 tp + (g >> 1), channels 8 (g & 1) .. + 7 (its 8
// consecutive k); taps run in 5 pairs, the 10th tap has zero weights.  The halo stays planar ([ci][row][col], planes == 2 (mod 4) words so that
// the two channel blocks of a 32-lane read group sit on disjoint banks): a fragment is 8 ds_read_b32 + 8 v_perm_b32 (hi halves / lo halves of
// dword pairs); the weight slab is laid out on the host so that a lane's 8 dwords are contiguous (2 ds_read_b128).  One LDS buffer of 58 KB,
// two workgroups per CU: while one parks (converts) the other multiplies.
constexpr int CKB = 16;        // input channels per K chunk of the split-bf16 kernel
constexpr int BX_TAPS = 10;    // 9 taps + 1 of zero weights: 5 MFMA steps of 2 taps

template <int MT>
struct GeomBx {
  static constexpr int TF = 16 * MT, ROWS = 6, ROWP = TF + 8, COL0 = 3;
  static constexpr int PLANE0 = ROWS * ROWP;
  static constexpr int PLANE = PLANE0 + ((2 - PLANE0 % 4) + 4) % 4;          // == 2 (mod 4): 8 planes == 16 (mod 32)
  static constexpr int Q = ROWP / 4;
  static constexpr int A_F4 = CKB * ROWS * Q, W_WORDS = BX_TAPS * 2 * NTW * 8, W_F4 = W_WORDS / 4;
  static constexpr int A_IT = (A_F4 + 255) / 256, W_IT = (W_F4 + 255) / 256;
};

__device__ __forceinline__ void split_pack2(float x0, float x1, unsigned& p0, unsigned& p1) {
  unsigned h, l;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h) : "v"(x0), "v"(x1));          // round to nearest even: {bf16(x0), bf16(x1)}
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xFFFF0000u);   // exact
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(l) : "v"(r0), "v"(r1));
  p0 = (h << 16) | (l & 0xFFFFu);
  p1 = (h & 0xFFFF0000u) | (l >> 16);
}

template <int MT, bool VEC, int NPROD>
__global__ __launch_bounds__(256, 2) void mdx_conv3_bx_kernel(const MdxConvParams p) {
  using G = GeomBx<MT>;
  __shared__ __attribute__((aligned(16))) unsigned As[CKB * G::PLANE];
  __shared__ __attribute__((aligned(16))) unsigned Ws[G::W_WORDS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int f0 = blockIdx.x * G::TF, t0 = blockIdx.y * 4;
  const int b = blockIdx.z / p.ntiles, nt = blockIdx.z % p.ntiles;
  const float* xb = p.x + (size_t)b * p.Cin * p.Ti * p.Fi;
  const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.w) + (size_t)nt * p.nchunks * G::W_F4;
  const size_t plane = (size_t)p.Ti * p.Fi;

  const float* a_ptr[G::A_IT];
  int a_lds[G::A_IT], a_lim[G::A_IT];
  unsigned a_ok = 0;
#pragma unroll
  for (int it = 0; it < G::A_IT; ++it) {
    const int idx = tid + it * 256;
    const int ci = idx / (G::ROWS * G::Q), rem = idx % (G::ROWS * G::Q), r = rem / G::Q, q = rem % G::Q;
    const int tin = t0 - 1 + r, fin = f0 - 4 + 4 * q;
    const bool slot = idx < G::A_F4, rowin = slot && tin >= 0 && tin < p.Ti;
    a_lds[it] = slot ? ci * G::PLANE + r * G::ROWP + 4 * q : -1;
    a_lim[it] = slot ? p.Cin - ci : 0;
    a_ptr[it] = xb;
    if (VEC) {
      if (rowin && fin >= 0 && fin < p.Fi) { a_ok |= 0xFu << (4 * it); a_ptr[it] = xb + (size_t)ci * plane + (size_t)tin * p.Fi + fin; }
    } else if (rowin) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (fin + e >= 0 && fin + e < p.Fi) a_ok |= 1u << (4 * it + e);
      a_ptr[it] = xb + (size_t)ci * plane + (size_t)tin * p.Fi + fin;
    }
  }
  const f32x4 zv = {0.f, 0.f, 0.f, 0.f};
  f32x4 ra[G::A_IT];
  u32x4 rw[G::W_IT];

  auto gload = [&](int chunk) {
#pragma unroll
    for (int it = 0; it < G::A_IT; ++it) {
      const bool cok = chunk * CKB < a_lim[it];
      const float* src = cok ? a_ptr[it] + (size_t)chunk * CKB * plane : xb;
      if (VEC) {
        ra[it] = *reinterpret_cast<const f32x4*>(src);
      } else {
        f32x4 v = zv;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (cok && (a_ok >> (4 * it + e) & 1)) v[e] = src[e];
        ra[it] = v;
      }
    }
#pragma unroll
    for (int it = 0; it < G::W_IT; ++it) {
      const int idx = tid + it * 256;
      rw[it] = wsrc[(size_t)chunk * G::W_F4 + (idx < G::W_F4 ? idx : 0)];
    }
  };
  auto park = [&](int chunk) {
#pragma unroll
    for (int it = 0; it < G::A_IT; ++it) {
      if (a_lds[it] < 0) continue;
      f32x4 v = ra[it];
      if (VEC && !((a_ok >> (4 * it) & 1) && chunk * CKB < a_lim[it])) v = zv;
      unsigned p0, p1, p2, p3;
      split_pack2(v[0], v[1], p0, p1);
      split_pack2(v[2], v[3], p2, p3);
      uint2* dst = reinterpret_cast<uint2*>(&As[a_lds[it]]);                   // planes are == 2 (mod 4) words: 8-B aligned, not 16
      dst[0] = make_uint2(p0, p1);
      dst[1] = make_uint2(p2, p3);
    }
#pragma unroll
    for (int it = 0; it < G::W_IT; ++it) {
      const int idx = tid + it * 256;
      if (idx < G::W_F4) *reinterpret_cast<u32x4*>(&Ws[idx * 4]) = rw[it];
    }
  };

  f32x4 acc[MT][3];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = zv;

  // this lane's operand bases: channel block g & 1, tap 2 tp + (g >> 1) (tap 9 has zero weights: it reads tap 8's, valid, halo)
  const int a_base = (g & 1) * 8 * G::PLANE + wave * G::ROWP + l15 + G::COL0;
  int toff[5];
#pragma unroll
  for (int tp = 0; tp < 5; ++tp) {
    const int tap = min(2 * tp + (g >> 1), 8);
    toff[tp] = (tap / 3) * G::ROWP + tap % 3;
  }
  const int w_base = ((g >> 1) * 2 + (g & 1)) * NTW * 8 + l15 * 8;

  gload(0);
  for (int chunk = 0; chunk < p.nchunks; ++chunk) {
    park(chunk);
    __syncthreads();
    if (chunk + 1 < p.nchunks) gload(chunk + 1);
#pragma unroll
    for (int tp = 0; tp < 5; ++tp) {
      bf16x8 bh[3], bl[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const u32x4 w0 = *reinterpret_cast<const u32x4*>(&Ws[w_base + tp * 4 * NTW * 8 + j * 16 * 8]);
        const u32x4 w1 = *reinterpret_cast<const u32x4*>(&Ws[w_base + tp * 4 * NTW * 8 + j * 16 * 8 + 4]);
        u32x4 h, l;
        h[0] = __builtin_amdgcn_perm(w0[1], w0[0], 0x07060302u); l[0] = __builtin_amdgcn_perm(w0[1], w0[0], 0x05040100u);
        h[1] = __builtin_amdgcn_perm(w0[3], w0[2], 0x07060302u); l[1] = __builtin_amdgcn_perm(w0[3], w0[2], 0x05040100u);
        h[2] = __builtin_amdgcn_perm(w1[1], w1[0], 0x07060302u); l[2] = __builtin_amdgcn_perm(w1[1], w1[0], 0x05040100u);
        h[3] = __builtin_amdgcn_perm(w1[3], w1[2], 0x07060302u); l[3] = __builtin_amdgcn_perm(w1[3], w1[2], 0x05040100u);
        bh[j] = __builtin_bit_cast(bf16x8, h); bl[j] = __builtin_bit_cast(bf16x8, l);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const unsigned* A = &As[a_base + toff[tp] + i * 16];
        unsigned d[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = A[k * G::PLANE];
        u32x4 h, l;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          h[k] = __builtin_amdgcn_perm(d[2 * k + 1], d[2 * k], 0x07060302u);
          l[k] = __builtin_amdgcn_perm(d[2 * k + 1], d[2 * k], 0x05040100u);
        }
        const bf16x8 ah = __builtin_bit_cast(bf16x8, h), al = __builtin_bit_cast(bf16x8, l);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if (NPROD >= 4) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], al, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah, acc[i][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  const int t = t0 + wave;
  if (t >= p.Tg) return;
  float* ob = p.out + (size_t)b * p.Cout * p.Tg * p.Fg;
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = nt * NTW + j * 16 + (lane >> 4) * 4 + r;
      if (co >= p.Cout) continue;
      const float bias = p.bias[co];
      float* orow = ob + ((size_t)co * p.Tg + t) * p.Fg;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int f = f0 + i * 16 + l15;
        if (f >= p.Fg) continue;
        float v = acc[i][j][r] + bias;
        if (p.relu) v = fmaxf(v, 0.f);
        orow[f] = v;
      }
    }
}

static int g_bx_nprod = 3;      // bf16 MFMAs per product of the split-bf16 kernel: 3 (W_lo A_lo dropped) or 4
template <bool VEC, int NPROD>
hipError_t launch_bx(const MdxConvParams& p, hipStream_t s) {
  auto wgs = [&](int mt) { return (long)((p.Fg + 16 * mt - 1) / (16 * mt)) * ((p.Tg + 3) / 4) * p.B * p.ntiles; };
  int mt = 4;
  while (mt > 1 && wgs(mt) < 512) mt >>= 1;
  const dim3 grid((p.Fg + 16 * mt - 1) / (16 * mt), (p.Tg + 3) / 4, p.B * p.ntiles);
  if (grid.y > 65535 || grid.z > 65535) return hipErrorInvalidValue;
  if (mt == 4) hipLaunchKernelGGL((mdx_conv3_bx_kernel<4, VEC, NPROD>), grid, dim3(256), 0, s, p);
  else if (mt == 2) hipLaunchKernelGGL((mdx_conv3_bx_kernel<2, VEC, NPROD>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((mdx_conv3_bx_kernel<1, VEC, NPROD>), grid, dim3(256), 0, s, p);
  return hipGetLastError();
}

// chunk of 4 channels: half the LDS (28 KB at MT = 4) -> three workgroups per CU instead of two, twice the barriers per flop.  Chosen per launch
// by mdx_conv_ck() (the 3x3 convolution only; measured: DESIGN.md section 10)
static int g_ck3 = 4;      // measured at the Kim_Vocal_1 shape: 13.58 vs 14.25 ms (batch 2), 7.66 vs 8.00 ms (batch 1): profiles/r06/r06i_mdx_modes.txt
template <int KH, int S, int PAD, bool UP, bool VEC>
hipError_t launch_geom(const MdxConvParams& p, hipStream_t s) {
  // wider position tiles while they still give the chip ~2 workgroups per CU; the narrow ones for the deep, small levels
  auto wgs = [&](int mt) { return (long)((p.Fg + 16 * mt - 1) / (16 * mt)) * ((p.Tg + 3) / 4) * p.B * p.ntiles; };
  constexpr int MAXMT = (S == 2) ? 2 : 4;        // the stride-2 halo of a 64-wide tile would not fit the static 64 KB of LDS
  int mt = MAXMT;
  while (mt > 1 && wgs(mt) < 512) mt >>= 1;
  const dim3 grid((p.Fg + 16 * mt - 1) / (16 * mt), (p.Tg + 3) / 4, p.B * p.ntiles);
  if (grid.y > 65535 || grid.z > 65535) return hipErrorInvalidValue;
  if (KH == 3 && g_ck3 == 4) {
    if constexpr (KH == 3) {
      if (mt == 4) hipLaunchKernelGGL((mdx_conv_kernel<KH, S, PAD, 4, UP, VEC, 4>), grid, dim3(256), 0, s, p);
      else if (mt == 2) hipLaunchKernelGGL((mdx_conv_kernel<KH, S, PAD, 2, UP, VEC, 4>), grid, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((mdx_conv_kernel<KH, S, PAD, 1, UP, VEC, 4>), grid, dim3(256), 0, s, p);
    }
    return hipGetLastError();
  }
  if (mt == 4) {
    if constexpr (MAXMT >= 4) hipLaunchKernelGGL((mdx_conv_kernel<KH, S, PAD, 4, UP, VEC, 8>), grid, dim3(256), 0, s, p);
  } else if (mt == 2) {
    hipLaunchKernelGGL((mdx_conv_kernel<KH, S, PAD, 2, UP, VEC, 8>), grid, dim3(256), 0, s, p);
  } else {
    hipLaunchKernelGGL((mdx_conv_kernel<KH, S, PAD, 1, UP, VEC, 8>), grid, dim3(256), 0, s, p);
  }
  return hipGetLastError();
}

template <int KH, int S, int PAD, bool UP>
hipError_t launch_geom(const MdxConvParams& p, hipStream_t s) {
  // 16-B loads need whole float4 groups inside or outside the image: a row length that is a multiple of 4 (tiles start at multiples of 16)
  const bool vec = (p.Fi & 3) == 0 && (reinterpret_cast<size_t>(p.x) & 15) == 0;
  return vec ? launch_geom<KH, S, PAD, UP, true>(p, s) : launch_geom<KH, S, PAD, UP, false>(p, s);
}

// ---- first 1x1 convolution (+ folded norm, ReLU) with the transpose of mdxnet.py:105-107: [b][ci][f][t] -> [b][co][t][f] ----------------
constexpr int XT = 32;     // 32 x 32 (f, t) positions per workgroup

__global__ __launch_bounds__(256) void mdx_first_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ out, int Cin, int Cout, int F, int T, int relu) {
  extern __shared__ float sm[];                    // [Cin][XT f][XT + 1 t] then the weights [Cout][Cin] and bias [Cout]
  float* tile = sm;
  float* wl = sm + Cin * XT * (XT + 1);
  float* bl = wl + Cout * Cin;
  const int f0 = blockIdx.x * XT, t0 = blockIdx.y * XT, b = blockIdx.z;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = threadIdx.x; i < Cout * Cin; i += 256) wl[i] = w[i];
  for (int i = threadIdx.x; i < Cout; i += 256) bl[i] = bias[i];
  const float* xb = x + (size_t)b * Cin * F * T;
  for (int c = 0; c < Cin; ++c)
    for (int r = ty; r < XT; r += 8) {             // row = f, lanes along t (contiguous in the input)
      const int f = f0 + r, t = t0 + tx;
      tile[(c * XT + r) * (XT + 1) + tx] = (f < F && t < T) ? xb[((size_t)c * F + f) * T + t] : 0.f;
    }
  __syncthreads();
  float* ob = out + (size_t)b * Cout * T * F;
  for (int r = ty; r < XT; r += 8) {               // row = t, lanes along f (contiguous in the output)
    const int t = t0 + r, f = f0 + tx;
    if (t >= T || f >= F) continue;
    float xv[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) xv[c] = c < Cin ? tile[(c * XT + tx) * (XT + 1) + r] : 0.f;
    for (int co = 0; co < Cout; ++co) {
      float v = bl[co];
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (c < Cin) v = fmaf(wl[co * Cin + c], xv[c], v);
      if (relu) v = fmaxf(v, 0.f);
      ob[((size_t)co * T + t) * F + f] = v;
    }
  }
}

// ---- last 1x1 convolution with the transpose back (mdxnet.py:121-125): [b][ci][t][f] -> [b][co][f][t], co <= 8 ---------------------------
__global__ __launch_bounds__(256) void mdx_final_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ out, int Cin, int Cout, int F, int T) {
  extern __shared__ float sm[];                    // [Cout][XT t][XT + 1 f] then the weights [Cout][Cin]
  float* tile = sm;
  float* wl = sm + Cout * XT * (XT + 1);
  const int f0 = blockIdx.x * XT, t0 = blockIdx.y * XT, b = blockIdx.z;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = threadIdx.x; i < Cout * Cin; i += 256) wl[i] = w[i];
  __syncthreads();
  const float* xb = x + (size_t)b * Cin * T * F;
  for (int r = ty; r < XT; r += 8) {               // row = t, lanes along f
    const int t = t0 + r, f = f0 + tx;
    float v[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) v[o] = o < Cout ? bias[o] : 0.f;
    if (t < T && f < F)
#pragma unroll 8
      for (int c = 0; c < Cin; ++c) {
        const float xv = xb[((size_t)c * T + t) * F + f];
#pragma unroll
        for (int o = 0; o < 8; ++o)
          if (o < Cout) v[o] = fmaf(wl[o * Cin + c], xv, v[o]);
      }
#pragma unroll
    for (int o = 0; o < 8; ++o)
      if (o < Cout) tile[(o * XT + r) * (XT + 1) + tx] = v[o];
  }
  __syncthreads();
  float* ob = out + (size_t)b * Cout * F * T;
  for (int o = 0; o < Cout; ++o)
    for (int r = ty; r < XT; r += 8) {             // row = f, lanes along t
      const int f = f0 + r, t = t0 + tx;
      if (f < F && t < T) ob[((size_t)o * F + f) * T + t] = tile[(o * XT + tx) * (XT + 1) + r];
    }
}

// ---- GroupNorm(2, c) (the 'adamw' variant, mdxnet.py:54-55): statistics, then normalise + ReLU (+ skip product / residual sum) -----------
// x is a row matrix [b * c * t rows][ld] with `cols` valid columns; the rows of one (sample, group) are contiguous.
constexpr int GN_SPLIT = 64;

__global__ __launch_bounds__(256) void mdx_gn_partial_kernel(const float* __restrict__ x, int ld, int cols, long rows_per_group, double* __restrict__ part) {
  const int grp = blockIdx.y, sp = blockIdx.x;
  const long r0 = rows_per_group * sp / GN_SPLIT, r1 = rows_per_group * (sp + 1) / GN_SPLIT;
  const float* base = x + (size_t)grp * rows_per_group * ld;
  double s = 0.0, q = 0.0;
  for (long r = r0 + (threadIdx.x >> 6); r < r1; r += 4) {
    float fs = 0.f, fq = 0.f;                      // one row segment per wave pass in fp32, rows combined in double
    for (int c = threadIdx.x & 63; c < cols; c += 64) {
      const float v = base[(size_t)r * ld + c];
      fs += v; fq = fmaf(v, v, fq);
    }
    s += fs; q += fq;
  }
  __shared__ double ss[256], sq[256];
  ss[threadIdx.x] = s; sq[threadIdx.x] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { ss[threadIdx.x] += ss[threadIdx.x + o]; sq[threadIdx.x] += sq[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[((size_t)grp * GN_SPLIT + sp) * 2] = ss[0]; part[((size_t)grp * GN_SPLIT + sp) * 2 + 1] = sq[0]; }
}

__global__ void mdx_gn_final_kernel(const double* __restrict__ part, int ngroups, double count, float eps, float* __restrict__ stats) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ngroups) return;
  double s = 0.0, q = 0.0;
  for (int i = 0; i < GN_SPLIT; ++i) { s += part[((size_t)g * GN_SPLIT + i) * 2]; q += part[((size_t)g * GN_SPLIT + i) * 2 + 1]; }
  const double mean = s / count, var = fmax(q / count - mean * mean, 0.0);
  stats[2 * g] = (float)mean;
  stats[2 * g + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// mode 0: y = relu(norm(x)); 1: y = relu(norm(x)) * other; 2: y = relu(norm(x)) + other.  `other` / `out` share x's [rows][ld] shape.
__global__ __launch_bounds__(256) void mdx_gn_apply_kernel(const float* __restrict__ x, int ld, int cols, long rows, int C, int T,
                                                           const float* __restrict__ stats, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ other, int mode,
                                                           float* __restrict__ out) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long chan_row = row / T;                   // (b * C + c)
  const int c = (int)(chan_row % C), bi = (int)(chan_row / C);
  const int g = bi * 2 + (c >= C / 2 ? 1 : 0);
  const float mean = stats[2 * g], rstd = stats[2 * g + 1], ga = gamma[c], be = beta[c];
  for (int col = threadIdx.x & 63; col < cols; col += 64) {
    const size_t o = (size_t)row * ld + col;
    float v = fmaxf((x[o] - mean) * rstd * ga + be, 0.f);
    if (mode == 1) v *= other[o];
    if (mode == 2) v += other[o];
    out[o] = v;
  }
}

}  // namespace

hipError_t launch_mdx_conv(int kind, const MdxConvParams& p, hipStream_t s) {
  if (p.B <= 0 || p.Cin <= 0 || p.Cout <= 0 || p.Tg <= 0 || p.Fg <= 0 || p.nchunks <= 0 || p.ntiles <= 0) return hipErrorInvalidValue;
  if (kind == MDX_CONV3_BX) {
    const bool vec = (p.Fi & 3) == 0 && (reinterpret_cast<size_t>(p.x) & 15) == 0;
    if (g_bx_nprod == 4) return vec ? launch_bx<true, 4>(p, s) : launch_bx<false, 4>(p, s);
    return vec ? launch_bx<true, 3>(p, s) : launch_bx<false, 3>(p, s);
  }
  switch (kind) {
    case MDX_CONV3: return launch_geom<3, 1, 1, false>(p, s);
    case MDX_DOWN2: return launch_geom<2, 2, 0, false>(p, s);
    case MDX_UP2: return launch_geom<1, 1, 0, true>(p, s);
  }
  return hipErrorInvalidValue;
}

void mdx_conv_set_ck(int ck) { g_ck3 = ck == 8 ? 8 : 4; }
void mdx_conv_set_bx_products(int n) { g_bx_nprod = n == 4 ? 4 : 3; }

// padded K-slab row length of the re-laid weights (floats per input channel): the engine builds [ntile][chunk][8][mdx_conv_ciw(kind)]
int mdx_conv_ciw(int kind) {
  switch (kind) {
    case MDX_CONV3_BX: return BX_TAPS * 2 * NTW * 8 / CKB;     // words per input channel of a 16-channel chunk slab (layout: see the kernel)
    case MDX_CONV3: return Geom<3, 1, 1, 1>::CIW;
    case MDX_DOWN2: return Geom<2, 2, 0, 1>::CIW;
    case MDX_UP2: return Geom<1, 1, 0, 1>::CIW;
  }
  return 0;
}

hipError_t launch_mdx_first(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout, int F, int T, int relu,
                            hipStream_t s) {
  if (Cin > 8 || Cin <= 0 || Cout <= 0) return hipErrorInvalidValue;
  const size_t sm = ((size_t)Cin * XT * (XT + 1) + (size_t)Cout * Cin + Cout) * sizeof(float);
  if (sm > 64 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(mdx_first_kernel, dim3((F + XT - 1) / XT, (T + XT - 1) / XT, B), dim3(256), sm, s, x, w, bias, out, Cin, Cout, F, T, relu);
  return hipGetLastError();
}

hipError_t launch_mdx_final(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout, int F, int T, hipStream_t s) {
  if (Cout > 8 || Cout <= 0 || Cin <= 0) return hipErrorInvalidValue;
  const size_t sm = ((size_t)Cout * XT * (XT + 1) + (size_t)Cout * Cin) * sizeof(float);
  if (sm > 64 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(mdx_final_kernel, dim3((F + XT - 1) / XT, (T + XT - 1) / XT, B), dim3(256), sm, s, x, w, bias, out, Cin, Cout, F, T);
  return hipGetLastError();
}

hipError_t launch_mdx_groupnorm(const float* x, int ld, int cols, int B, int C, int T, const float* gamma, const float* beta, float eps,
                                const float* other, int mode, float* out, double* part /* [B * 2 * 64 * 2] */, float* stats /* [B * 2 * 2] */,
                                hipStream_t s) {
  if ((C & 1) || B <= 0 || cols <= 0) return hipErrorInvalidValue;
  const long rpg = (long)(C / 2) * T, rows = (long)B * C * T;
  hipLaunchKernelGGL(mdx_gn_partial_kernel, dim3(GN_SPLIT, B * 2), dim3(256), 0, s, x, ld, cols, rpg, part);
  hipLaunchKernelGGL(mdx_gn_final_kernel, dim3((B * 2 + 63) / 64), dim3(64), 0, s, part, B * 2, (double)rpg * cols, eps, stats);
  hipLaunchKernelGGL(mdx_gn_apply_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, ld, cols, rows, C, T, stats, gamma, beta, other, mode, out);
  return hipGetLastError();
}
