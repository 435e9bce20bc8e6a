// MLP2 / MLP3 with `mfma = "bf16"` as a bf16-STORAGE pipeline (reference: docs/tutorials/mlp.py:33-108 -- fc -> BatchNorm1d ->
// ReLU -> Dropout blocks with a residual; the "bf16 MFMA MLP head" of the mixed-corpus configuration).  Layer outputs and
// everything saved for the backward are [N, 128] bf16 matrices (256-byte rows, columns beyond the layer's width zero); master
// weights, BatchNorm statistics, gradients of the parameters and every accumulation are fp32.  These layers are bound by HBM
// at widths <= 170 (the bf16 matrix cores are three orders of magnitude faster than the rows arrive), so halving the bytes of
// every activation pass is what the mode is for -- with fp32 tensors in memory the bf16 GEMMs were SLOWER than the fp32 ones.
//
// Building blocks (all deterministic: per-workgroup partial sums, fixed-order second stages):
//   layer_fwd     Y = X W^T + b                 X fp32 [N, I <= 192] or bf16 [N,128]; Y bf16 [N,128] or fp32 [N,O]; W1-style
//                                                fragment image in LDS (mlp_head.hip's forward); also the data gradient
//                                                gX = dY W (+ add) with the image built from W^T
//   colsum<0>     sum y, sum y^2                 BatchNorm batch statistics
//   act           z = drop(relu(y s + t)) (+ r)  BatchNorm affine + ReLU + dropout + residual, bf16 -> bf16, one pass
//   colsum<1>     sum gu, sum gu xhat            gu = g o mask(y): the two sums of the BatchNorm backward
//   bwd_apply     dy = gs (gu - k1 - xhat k2)    gradient at the layer's GEMM output, bf16
//   layer_wgrad   gW = dY^T X, gb = sum dY       mlp_head.hip's backward with dY loaded instead of formed from a gate
//   rowdot        out = h w^T + b ; gh = g w ; gw = g^T h ; gb = sum g     the final O <= 4 outputs
// Dropout masks are counter-based (common.hpp dropout_keep keyed by (seed + counter, row * 128 + column)): the backward
// recomputes them, nothing is stored.
#include <type_traits>

#include "common.hpp"

namespace mlqem {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kLW = MLQEM_MLP1_HIDDEN_PAD;      // columns of every bf16 activation matrix
constexpr int kLayerThreads = 256;

__device__ __forceinline__ unsigned lpack(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float blo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bhi(unsigned u) { return __builtin_bit_cast(float, u & 0xFFFF0000u); }
__device__ __forceinline__ void unpack8(const u32x4 v, float (&f)[8]) {
  f[0] = blo(v.x); f[1] = bhi(v.x); f[2] = blo(v.y); f[3] = bhi(v.y);
  f[4] = blo(v.z); f[5] = bhi(v.z); f[6] = blo(v.w); f[7] = bhi(v.w);
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
  return u32x4{lpack(f[0], f[1]), lpack(f[2], f[3]), lpack(f[4], f[5]), lpack(f[6], f[7])};
}
// component c of a 4-vector, for c a constant after unrolling
__device__ __forceinline__ unsigned vget(const u32x4 v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }
__device__ __forceinline__ int layer_unit(int ob, int m) { return 32 * (ob >> 1) + 8 * (m >> 2) + 4 * (ob & 1) + (m & 3); }

// ------------------------------------------------------------------------------------------------ GEMM
struct LayerArgs {
  const void* x; int64_t ldx; int x_bf16;       // input rows: fp32 (ldx floats, padded to 4) or bf16 [N,128]
  const float* w; const float* b; int transposed;   // W [O,I] (Y = X W^T + b) or, transposed, W [K,U] used as its transpose (gX = dY W)
  const unsigned short* add;                    // optional bf16 [N,128] added to the result
  void* y; int y_f32; int64_t ldy;              // output: bf16 [N,128], or fp32 [N, ldy] (the first U columns)
  int64_t N; int K, U;                          // K input columns, U output units (<= 128)
  const void* image;
  int relu; float drop_p; uint64_t seed; const uint64_t* seed_counter;   // epilogue of a block WITHOUT BatchNorm: y = drop(relu(.))
};

__host__ __device__ inline int layer_image_u32x4(int G2) { return 8 * G2 * kWave + kLW / 4; }

// fragment image: A[m = unit][k], 8 bf16 per lane and 32-column group, then the bias
__global__ __launch_bounds__(256) void layer_image_kernel(const LayerArgs a, int G2, u32x4* __restrict__ image) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int n_frag = 8 * G2 * kWave;
  if (idx < n_frag) {
    const int l = idx & 63, g = (idx >> 6) % G2, ob = idx / (64 * G2);
    const int u = layer_unit(ob, l & 15), lq = l >> 4;
    float w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 32 * g + 8 * lq + j;
      float v = 0.f;
      if (u < a.U && k < a.K) v = a.transposed ? a.w[(int64_t)k * a.U + u] : a.w[(int64_t)u * a.K + k];
      w[j] = v;
    }
    image[idx] = u32x4{lpack(w[0], w[1]), lpack(w[2], w[3]), lpack(w[4], w[5]), lpack(w[6], w[7])};
    return;
  }
  const int t = idx - n_frag;
  if (t < kLW) reinterpret_cast<float*>(image + n_frag)[t] = (a.b && t < a.U) ? a.b[t] : 0.f;
}

template <int G2, bool IN_BF16>
__global__ __launch_bounds__(kLayerThreads) void layer_fwd_kernel(const LayerArgs a) {
  extern __shared__ u32x4 s_raw[];
  for (int idx = threadIdx.x; idx < layer_image_u32x4(G2); idx += kLayerThreads) s_raw[idx] = static_cast<const u32x4*>(a.image)[idx];
  __syncthreads();
  const u32x4* s_w = s_raw;
  const float* s_b = reinterpret_cast<const float*>(s_raw + 8 * G2 * kWave);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int64_t n_tiles = ceil_div(a.N, 16);
  const int64_t n_waves = (int64_t)gridDim.x * (kLayerThreads / kWave), wave = (int64_t)blockIdx.x * (kLayerThreads / kWave) + wid;
  const int kpad = (a.K + 3) / 4 * 4;
  const uint64_t dseed = a.seed + (a.seed_counter ? *a.seed_counter * 0xD1B54A32D192ED03ull : 0ull);
  const unsigned dthr = (unsigned)(a.drop_p * 65536.f);
  const float dinv = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  // unconditional, clamped loads; fix-ups where the values are consumed (see mlp_head.hip)
  // fp32 rows are loaded as float4 (HIP's struct of floats), bf16 rows as 4 dwords: a 16-byte load of float memory through an
  // unsigned-int vector type came out of the compiler as ONE global_load_dword with the other three components undefined
  constexpr int NL = IN_BF16 ? G2 : 2 * G2;
  using XT = std::conditional_t<IN_BF16, u32x4, float4>;
  struct Raw { XT v[NL]; };
  auto load_tile = [&](int64_t t, Raw& r) {
    const int64_t row = min(t * 16 + lr, a.N - 1);
    if constexpr (IN_BF16) {
      const unsigned short* xr = static_cast<const unsigned short*>(a.x) + row * kLW + 8 * lq;
#pragma unroll
      for (int g = 0; g < G2; ++g) r.v[g] = *reinterpret_cast<const u32x4*>(xr + 32 * g);       // 32 g + 8 lq < 128: inside the row
    } else {
      const float* xr = static_cast<const float*>(a.x) + row * a.ldx;
#pragma unroll
      for (int c = 0; c < 2 * G2; ++c)
        r.v[c] = *reinterpret_cast<const float4*>(xr + min(32 * (c >> 1) + 8 * lq + 4 * (c & 1), kpad - 4));
    }
  };
  auto to_frag = [&](const Raw& r, int g) {
    if constexpr (IN_BF16) {
      return r.v[g];                              // the loaded bytes ARE the fragment; pad columns of an activation matrix are zero
    } else {
      const float4 lo = r.v[2 * g], hi = r.v[2 * g + 1];
      float p[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      const int k0 = 32 * g + 8 * lq;
      if (32 * g + 32 > a.K) {                    // uniform: the boundary group; pad columns must not reach the MFMA
#pragma unroll
        for (int j = 0; j < 8; ++j) if (k0 + j >= a.K) p[j] = 0.f;
      }
      return pack8(p);
    }
  };
  Raw r0;
  u32x4 xc[G2];
  load_tile(wave, r0);
#pragma unroll
  for (int g = 0; g < G2; ++g) xc[g] = to_frag(r0, g);
  for (int64_t t = wave; t < n_tiles; t += n_waves) {
    Raw rn;
    load_tile(t + n_waves, rn);
    __builtin_amdgcn_sched_barrier(0);            // the prefetch is issued HERE
    f32x4 acc[8];
#pragma unroll
    for (int ob = 0; ob < 8; ++ob) acc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G2; ++g) {
      const bf16x8 xf = __builtin_bit_cast(bf16x8, xc[g]);
#pragma unroll
      for (int ob = 0; ob < 8; ++ob)
        acc[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, s_w[(ob * G2 + g) * kWave + lane]), xf, acc[ob], 0, 0, 0);
      if (g % 2 == 1) __builtin_amdgcn_sched_barrier(0);
    }
    const int64_t row = t * 16 + lr;
    // the epilogue's own load (the add operand) and the take-over of the next tile BEFORE the stores
    u32x4 addv[4];
    if (a.add) {
      const int64_t rc = min(row, a.N - 1);
#pragma unroll
      for (int p = 0; p < 4; ++p) addv[p] = *reinterpret_cast<const u32x4*>(a.add + rc * kLW + 32 * p + 8 * lq);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < G2; ++g) xc[g] = to_frag(rn, g);
    __builtin_amdgcn_sched_barrier(0);
    if (row < a.N) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc[2 * p + (e >> 2)][e & 3] + s_b[32 * p + 8 * lq + e];
        if (a.add) {
          float r[8];
          unpack8(addv[p], r);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
        if (a.relu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (a.drop_p > 0.f) {       // layer_fwd_f32_kernel's mask: one splitmix64 round per four units, keyed by (row, first unit)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            uint64_t z = dseed + ((uint64_t)(row * kLW + 32 * p + 8 * lq + 4 * h) + 1) * 0x9E3779B97F4A7C15ull;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z = z ^ (z >> 31);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * h + e] = (unsigned)((z >> (16 * e)) & 0xFFFFu) >= dthr ? v[4 * h + e] * dinv : 0.f;
          }
        }
        if (a.y_f32) {
          float* dst = static_cast<float*>(a.y) + row * a.ldy;
#pragma unroll
          for (int e = 0; e < 8; ++e) if (32 * p + 8 * lq + e < a.U) dst[32 * p + 8 * lq + e] = v[e];
        } else {
          *reinterpret_cast<u32x4*>(static_cast<unsigned short*>(a.y) + row * kLW + 32 * p + 8 * lq) = pack8(v);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ element-wise and column sums
// A thread owns 8 columns of rows (tid >> 4) + 16 k.  bf16 storage: the columns 8 t .. 8 t + 7, t = tid & 15 (one 16-byte access
// per matrix and row).  fp32 storage: 4 t .. 4 t + 3 and 64 + 4 t .. 64 + 4 t + 3 -- two 16-byte accesses, each of them contiguous
// across the 16 lanes of a row (eight adjacent floats per lane would make every load instruction touch every other 16 bytes).
template <typename ST> __device__ __forceinline__ int col_of(int t, int e) {
  if constexpr (std::is_same_v<ST, float>) return e < 4 ? 4 * t + e : 60 + 4 * t + e; else return 8 * t + e;
}
template <typename ST>
__device__ __forceinline__ void load8(const ST* __restrict__ row, int t, float (&f)[8]) {
  if constexpr (std::is_same_v<ST, float>) {
    const float4 lo = *reinterpret_cast<const float4*>(row + 4 * t), hi = *reinterpret_cast<const float4*>(row + 64 + 4 * t);
    f[0] = lo.x; f[1] = lo.y; f[2] = lo.z; f[3] = lo.w; f[4] = hi.x; f[5] = hi.y; f[6] = hi.z; f[7] = hi.w;
  } else {
    unpack8(*reinterpret_cast<const u32x4*>(row + 8 * t), f);
  }
}
template <typename ST>
__device__ __forceinline__ void store8(ST* __restrict__ row, int t, const float (&f)[8]) {
  if constexpr (std::is_same_v<ST, float>) {
    *reinterpret_cast<float4*>(row + 4 * t) = make_float4(f[0], f[1], f[2], f[3]);
    *reinterpret_cast<float4*>(row + 64 + 4 * t) = make_float4(f[4], f[5], f[6], f[7]);
  } else {
    *reinterpret_cast<u32x4*>(row + 8 * t) = pack8(f);
  }
}
// a value as the storage holds it (what the next reader sees)
template <typename ST> __device__ __forceinline__ float stored(float v) {
  if constexpr (std::is_same_v<ST, float>) return v; else return (float)(__bf16)v;
}

struct ActArgs {
  const void* y; const void* g; const void* res;                                   // [N,128] in the pipeline's storage (bf16 or fp32)
  const float* g32; int64_t ldg32;                                                  // the incoming gradient as fp32 [N, ldg32] instead of g
  const float* scale; const float* shift; const float* mean; const float* invstd;  // per column [128]
  const float* gs; const float* k1; const float* k2;                                // backward apply
  void* out;
  float* partial;                                                                   // column sums: [blocks][2][128]
  int64_t N; int C; int relu;
  float drop_p; uint64_t seed; const uint64_t* seed_counter;
  int64_t rows_per_block;
};

// the 8 per-column constants of this thread: two 16-byte loads (every per-column vector of this file is [128] floats with zeros
// beyond the layer's width).  Eight predicated scalar loads per vector cost more than the rows the thread then processes.
template <typename ST>
__device__ __forceinline__ void col8(const float* __restrict__ v, int t, float (&f)[8]) {
  const float4 lo = *reinterpret_cast<const float4*>(v + col_of<ST>(t, 0)), hi = *reinterpret_cast<const float4*>(v + col_of<ST>(t, 4));
  f[0] = lo.x; f[1] = lo.y; f[2] = lo.z; f[3] = lo.w; f[4] = hi.x; f[5] = hi.y; f[6] = hi.z; f[7] = hi.w;
}

// keep / drop decisions of the thread's 8 columns of a row as a bit mask: two splitmix64 rounds, four 16-bit uniforms each, keyed
// by the first column of each group of four (the scheme of common.hpp's dropout_keep; a mask in a register -- bool arrays handed
// through references went to scratch memory, byte by byte, and made every element-wise kernel of this file six times slower than
// its bytes)
template <typename ST>
__device__ __forceinline__ unsigned keep_mask8(const ActArgs& a, uint64_t seed, int64_t row, int t) {
  if (!(a.drop_p > 0.f)) return 0xFFu;
  const unsigned thr = (unsigned)(a.drop_p * 65536.f);
  unsigned m = 0;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    uint64_t z = seed + ((uint64_t)(row * kLW + col_of<ST>(t, 4 * half)) + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
#pragma unroll
    for (int v = 0; v < 4; ++v) m |= ((unsigned)((z >> (16 * v)) & 0xFFFFu) >= thr ? 1u : 0u) << (4 * half + v);
  }
  return m;
}

// z = drop(relu(y s + t)) (+ res); columns >= C stay zero
template <typename ST>
__global__ __launch_bounds__(256) void layer_act_kernel(const ActArgs a) {
  const int t = threadIdx.x & 15;
  float sc[8], sh[8];
  col8<ST>(a.scale, t, sc);
  col8<ST>(a.shift, t, sh);
  const uint64_t seed = a.seed + (a.seed_counter ? *a.seed_counter * 0xD1B54A32D192ED03ull : 0ull);
  const float inv = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  for (int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); row < a.N; row += (int64_t)gridDim.x * 16) {
    float y[8], r[8];
    load8(static_cast<const ST*>(a.y) + row * kLW, t, y);
    if (a.res) load8(static_cast<const ST*>(a.res) + row * kLW, t, r);
    const unsigned keep = keep_mask8<ST>(a, seed, row, t);
    float z[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float u = fmaf(y[e], sc[e], sh[e]);
      if (a.relu) u = fmaxf(u, 0.f);
      u = ((keep >> e) & 1u) ? u * inv : 0.f;
      if (a.res) u = stored<ST>(u) + r[e];           // the residual adds to the STORED activation (what the next layer and the backward see)
      z[e] = (col_of<ST>(t, e) < a.C) ? u : 0.f;
    }
    store8(static_cast<ST*>(a.out) + row * kLW, t, z);
  }
}

// The gradient at a block's pre-activation u = y s + t: gu = g o (u > 0) o keep / (1 - p), from g (storage type or fp32) and y.
template <typename ST>
__device__ __forceinline__ void load_gu(const ActArgs& a, uint64_t seed, float inv, const float (&sc)[8], const float (&sh)[8], int64_t row,
                                        int t, float (&gu)[8], float (&y)[8]) {
  float g[8];
  load8(static_cast<const ST*>(a.y) + row * kLW, t, y);
  if (a.g32) {
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = (col_of<ST>(t, e) < a.C) ? a.g32[row * a.ldg32 + col_of<ST>(t, e)] : 0.f;
  } else {
    load8(static_cast<const ST*>(a.g) + row * kLW, t, g);
  }
  const unsigned keep = keep_mask8<ST>(a, seed, row, t);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float u = fmaf(y[e], sc[e], sh[e]);
    const bool on = (!a.relu || u > 0.f) && ((keep >> e) & 1u) && col_of<ST>(t, e) < a.C;
    gu[e] = on ? g[e] * inv : 0.f;
  }
}

// MODE 0: s1 = sum y, s2 = sum y^2.  MODE 1: s1 = sum gu, s2 = sum gu xhat, xhat = (y - mean) invstd.
template <int MODE, typename ST>
__global__ __launch_bounds__(256) void layer_colsum_kernel(const ActArgs a) {
  __shared__ float s_red[2][16][kLW];
  const int t = threadIdx.x & 15, rl = threadIdx.x >> 4;
  float sc[8], sh[8], mu[8], is[8], s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sc[e] = sh[e] = mu[e] = is[e] = s1[e] = s2[e] = 0.f;
  if (MODE == 1) { col8<ST>(a.scale, t, sc); col8<ST>(a.shift, t, sh); col8<ST>(a.mean, t, mu); col8<ST>(a.invstd, t, is); }
  const uint64_t seed = a.seed + (a.seed_counter ? *a.seed_counter * 0xD1B54A32D192ED03ull : 0ull);
  const float inv = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_block, r1 = min(r0 + a.rows_per_block, a.N);
  // MODE 0: the sums are taken of y - s with s = row 0 of the matrix (as bn.hip does): E[y^2] - E[y]^2 of columns whose mean is large
  // against their spread loses the variance to cancellation in fp32; the shifted sums do not.  Workgroup 0 files s behind the partials.
  float sh0[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sh0[e] = 0.f;
  if (MODE == 0) {
    load8(static_cast<const ST*>(a.y), t, sh0);
    if (blockIdx.x == 0 && rl == 0) {
      float* __restrict__ shift = a.partial + (int64_t)gridDim.x * 2 * kLW;
#pragma unroll
      for (int e = 0; e < 8; ++e) shift[col_of<ST>(t, e)] = sh0[e];
    }
  }
  for (int64_t row = r0 + rl; row < r1; row += 16) {
    if (MODE == 0) {
      float y[8];
      load8(static_cast<const ST*>(a.y) + row * kLW, t, y);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = y[e] - sh0[e]; s1[e] += d; s2[e] = fmaf(d, d, s2[e]); }
    } else {
      float gu[8], y[8];
      load_gu<ST>(a, seed, inv, sc, sh, row, t, gu, y);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[e] += gu[e]; s2[e] = fmaf(gu[e], (y[e] - mu[e]) * is[e], s2[e]); }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { s_red[0][rl][col_of<ST>(t, e)] = s1[e]; s_red[1][rl][col_of<ST>(t, e)] = s2[e]; }
  __syncthreads();
  if (threadIdx.x < kLW) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { t1 += s_red[0][j][threadIdx.x]; t2 += s_red[1][j][threadIdx.x]; }
    a.partial[((int64_t)blockIdx.x * 2 + 0) * kLW + threadIdx.x] = t1;
    a.partial[((int64_t)blockIdx.x * 2 + 1) * kLW + threadIdx.x] = t2;
  }
}

// One 64-lane workgroup per column: partials added in a fixed order in double.
// MODE 0 -> mean, biased var, invstd, scale = gamma invstd, shift = beta - mean scale.
// MODE 1 -> dbeta (s1), dgamma (s2), gs = gamma invstd, k1 = s1 / N, k2 = s2 / N.
// One workgroup per column of the 128-wide layout (columns >= C get zeros: the element-wise kernels load the vectors eight
// columns at a time).  256 threads walk the partials (with 64 a thread summed 32 partials one load after the other: 15 us a
// launch, four launches a step); the 256 thread sums are added by a fixed tree, in double.  MODE 0 also applies the
// BatchNorm running-statistics update (torch: running = (1 - momentum) running + momentum batch, the variance unbiased;
// num_batches_tracked += 1) when the buffers are given -- five element-wise launches of the host side otherwise.
constexpr int kFinishThreads = 256;
template <int MODE>
__global__ __launch_bounds__(kFinishThreads) void layer_finish_kernel(const float* __restrict__ partial, int nblocks, int64_t N, int C,
                                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                      const float* __restrict__ invstd_in, float eps, float* __restrict__ o1,
                                                                      float* __restrict__ o2, float* __restrict__ o3, float* __restrict__ o4,
                                                                      float* __restrict__ o5, float* __restrict__ run_mean,
                                                                      float* __restrict__ run_var, float momentum,
                                                                      long long* __restrict__ batches) {
  __shared__ double s_t[2][kFinishThreads];
  const int c = blockIdx.x, j = threadIdx.x;
  if (c >= C) {
    if (j == 0) { o1[c] = 0.f; o2[c] = 0.f; o3[c] = 0.f; o4[c] = 0.f; o5[c] = 0.f; }
    return;
  }
  double t1 = 0.0, t2 = 0.0;
  for (int b = j; b < nblocks; b += kFinishThreads) {
    t1 += (double)partial[((int64_t)b * 2 + 0) * kLW + c];
    t2 += (double)partial[((int64_t)b * 2 + 1) * kLW + c];
  }
  s_t[0][j] = t1; s_t[1][j] = t2;
  __syncthreads();
  // a fixed tree over the 256 thread sums (thread 0 adding them one after the other was most of the launch's 11 us)
#pragma unroll
  for (int half = kFinishThreads / 2; half >= 1; half >>= 1) {
    if (j < half) { s_t[0][j] += s_t[0][j + half]; s_t[1][j] += s_t[1][j + half]; }
    __syncthreads();
  }
  if (j != 0) return;
  t1 = s_t[0][0]; t2 = s_t[1][0];
  if (MODE == 0) {
    const double ms = t1 / (double)N;                   // mean of the shifted values (shift = row 0, filed behind the partials)
    const double m = (double)partial[(int64_t)nblocks * 2 * kLW + c] + ms;
    double var = t2 / (double)N - ms * ms;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * is;
    o1[c] = (float)m; o2[c] = (float)var; o3[c] = is; o4[c] = sc; o5[c] = beta[c] - (float)m * sc;
    if (run_mean) {
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)m;
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * ((float)var * ((float)N / (float)(N - 1)));
    }
    if (batches && c == 0) *batches += 1;
  } else {
    o1[c] = (float)t1; o2[c] = (float)t2; o3[c] = gamma[c] * invstd_in[c];
    o4[c] = (float)(t1 / (double)N); o5[c] = (float)(t2 / (double)N);
  }
}

// dy = gs (gu - k1 - xhat k2) in the pipeline's storage; with gs = 1, k1 = k2 = 0 (a block without BatchNorm) dy = gu
template <typename ST>
__global__ __launch_bounds__(256) void layer_bwd_apply_kernel(const ActArgs a) {
  const int t = threadIdx.x & 15;
  float sc[8], sh[8], mu[8], is[8], gs[8], k1[8], k2[8];
  col8<ST>(a.scale, t, sc); col8<ST>(a.shift, t, sh); col8<ST>(a.mean, t, mu); col8<ST>(a.invstd, t, is);
  col8<ST>(a.gs, t, gs); col8<ST>(a.k1, t, k1); col8<ST>(a.k2, t, k2);
#pragma unroll
  for (int e = 0; e < 8; ++e) if (col_of<ST>(t, e) >= a.C) gs[e] = 0.f;  // dy of the pad columns is zero whatever the vectors hold there
  const uint64_t seed = a.seed + (a.seed_counter ? *a.seed_counter * 0xD1B54A32D192ED03ull : 0ull);
  const float inv = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  for (int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); row < a.N; row += (int64_t)gridDim.x * 16) {
    float gu[8], y[8], d[8];
    load_gu<ST>(a, seed, inv, sc, sh, row, t, gu, y);
#pragma unroll
    for (int e = 0; e < 8; ++e) d[e] = gs[e] * (gu[e] - k1[e] - (y[e] - mu[e]) * is[e] * k2[e]);
    store8(static_cast<ST*>(a.out) + row * kLW, t, d);
  }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// gW[128, K + 1] = dY^T [X | 1]: mlp_head.hip's column-split backward with the A fragments loaded (dY is in memory).
// Partial layout per workgroup: [wave][tile t 4 + s4][register][lane] (lane order, 256-byte stores).
constexpr int kLayerW1Floats = 4 * 32 * 4 * kWave;
constexpr int kLayerMaxBlocks = 512;

struct WgradLayerArgs {
  const unsigned short* dy; const void* x; int64_t ldx; int x_bf16;
  int64_t N; int K; float* partial;
};

template <bool IN_BF16>
__global__ __launch_bounds__(kLayerThreads) void layer_wgrad_kernel(const WgradLayerArgs a, int cpw) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int kpad = IN_BF16 ? kLW : (a.K + 3) / 4 * 4;
  const int col0 = 4 * (cpw * wid + lr);
  const bool xlane = lr < cpw && col0 <= a.K;
  const int colc = min(col0, kpad - 4);
  const bool x_raw[4] = {col0 + 0 < a.K, col0 + 1 < a.K, col0 + 2 < a.K, col0 + 3 < a.K};
  const float x_fill[4] = {xlane && col0 + 0 == a.K ? 1.f : 0.f, xlane && col0 + 1 == a.K ? 1.f : 0.f,
                           xlane && col0 + 2 == a.K ? 1.f : 0.f, xlane && col0 + 3 == a.K ? 1.f : 0.f};
  f32x4 acc[8][4];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[t][s] = f32x4{0.f, 0.f, 0.f, 0.f};
  using XT = std::conditional_t<IN_BF16, uint2, float4>;     // per row: four bf16 or four fp32 columns (loaded through their own types)
  struct Slab { XT xv[8]; u32x4 dv[8]; };
  const int64_t n_slabs = a.N / 32;
  const unsigned xo = (unsigned)(8 * lq * (IN_BF16 ? kLW : a.ldx) + colc), dof = (unsigned)(8 * lq * kLW + 8 * lr);
  auto issue_row = [&](Slab& d, int64_t s, int j) {
    const int64_t r0 = 32 * min(s, n_slabs - 1) + j;
    if constexpr (IN_BF16) d.xv[j] = *reinterpret_cast<const uint2*>(static_cast<const unsigned short*>(a.x) + r0 * kLW + xo);
    else d.xv[j] = *reinterpret_cast<const float4*>(static_cast<const float*>(a.x) + r0 * a.ldx + xo);
    d.dv[j] = *reinterpret_cast<const u32x4*>(a.dy + r0 * kLW + dof);
  };
  auto consume = [&](Slab& cur, int64_t first_row) {
    float xf[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = first_row + 8 * lq + j < a.N;
      const XT xq = cur.xv[j];
      float v[4];
      if constexpr (IN_BF16) { v[0] = blo(xq.x); v[1] = bhi(xq.x); v[2] = blo(xq.y); v[3] = bhi(xq.y); }
      else { v[0] = xq.x; v[1] = xq.y; v[2] = xq.z; v[3] = xq.w; }
#pragma unroll
      for (int c = 0; c < 4; ++c) xf[j][c] = x_raw[c] ? v[c] : x_fill[c];
      if (!ok) cur.dv[j] = u32x4{0u, 0u, 0u, 0u};            // a clamped row of the ragged tail contributes nothing
    }
    bf16x8 bx[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const u32x4 v = {lpack(xf[0][c], xf[1][c]), lpack(xf[2][c], xf[3][c]), lpack(xf[4][c], xf[5][c]), lpack(xf[6][c], xf[7][c])};
      bx[c] = __builtin_bit_cast(bf16x8, v);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int c = t >> 1;
      unsigned ad[4];
#pragma unroll
      for (int d = 0; d < 4; ++d)
        ad[d] = (t & 1) ? __builtin_amdgcn_perm(vget(cur.dv[2 * d + 1], c), vget(cur.dv[2 * d], c), 0x07060302u)
                        : __builtin_amdgcn_perm(vget(cur.dv[2 * d + 1], c), vget(cur.dv[2 * d], c), 0x05040100u);
      const u32x4 av = {ad[0], ad[1], ad[2], ad[3]};
      const bf16x8 af = __builtin_bit_cast(bf16x8, av);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) acc[t][s4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bx[s4], acc[t][s4], 0, 0, 0);
    }
  };
  Slab cur, nxt;
  const int64_t G = gridDim.x;
  if (n_slabs > 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) issue_row(cur, blockIdx.x, j);
    for (int64_t sl = blockIdx.x; sl < n_slabs; sl += G) {
#pragma unroll
      for (int j = 0; j < 8; ++j) issue_row(nxt, sl + G, j);
      __builtin_amdgcn_sched_barrier(0);
      consume(cur, 32 * sl);
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
    }
  }
  if (blockIdx.x == 0 && (a.N & 31)) {
    Slab t;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t row = min(32 * n_slabs + 8 * lq + j, a.N - 1);
      if constexpr (IN_BF16) t.xv[j] = *reinterpret_cast<const uint2*>(static_cast<const unsigned short*>(a.x) + row * kLW + colc);
      else t.xv[j] = *reinterpret_cast<const float4*>(static_cast<const float*>(a.x) + row * a.ldx + colc);
      t.dv[j] = *reinterpret_cast<const u32x4*>(a.dy + row * kLW + 8 * lr);
    }
    consume(t, 32 * n_slabs);
  }
  float* __restrict__ dst = a.partial + (int64_t)blockIdx.x * kLayerW1Floats + (wid * 32 * 4) * kWave + lane;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[((t * 4 + s4) * 4 + r) * kWave] = acc[t][s4][r];
}

// The same product for bf16 [N,128] inputs with the operands brought in by LDS DMA (global_load_lds_dwordx4: 64 lanes x 16 bytes
// from per-lane global addresses into one contiguous KB of LDS, no registers in between).  The register-prefetch form above
// holds ONE 32-row slab in flight per workgroup at one workgroup per CU (272 registers): 16 KB per CU against the ~40 KB the
// memory system needs in flight per CU, 2.8 TB/s.  Here a workgroup keeps kDmaSlabs - 1 slabs (16 KB each: dY 8 KB | X 8 KB)
// on their way in LDS, registers hold only the slab being multiplied (two workgroups per CU), and every wave reads the dY
// slab the workgroup fetched ONCE instead of fetching it itself: 47 -> 35 us per launch (134 MB: 3.8 TB/s).
//   DMA instruction i of a slab, issued by wave i / 2: lane (lq, lr) fetches bytes [16 lr, 16 lr + 16) of row 8 lq + i -> the
//   consumer's load j = i of lane (lq, lr) is the linear read `region j + 16 lane`: no bank conflicts, no index arithmetic.
//   One s_barrier per slab: a wave waits for its own DMAs of slab t (vmcnt), the barrier makes everybody's visible, and having
//   passed it means everybody has finished READING slab t - 1, whose buffer the next DMA may overwrite.  The barrier is the
//   raw instruction: __syncthreads() carries a release fence that waits for ALL outstanding DMAs (vmcnt(0)) -- the prefetch.
constexpr int kDmaSlabBytes = 16384;

template <int kDmaSlabs>
__global__ __launch_bounds__(kLayerThreads, 2) void layer_wgrad_lds_kernel(const WgradLayerArgs a, int cpw) {
  extern __shared__ __attribute__((aligned(16))) char s_dma[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int col0 = 4 * (cpw * wid + lr);
  const bool xlane = lr < cpw && col0 <= a.K;
  const int colc = min(col0, kLW - 4);
  const bool x_raw[4] = {col0 + 0 < a.K, col0 + 1 < a.K, col0 + 2 < a.K, col0 + 3 < a.K};
  const float x_fill[4] = {xlane && col0 + 0 == a.K ? 1.f : 0.f, xlane && col0 + 1 == a.K ? 1.f : 0.f,
                           xlane && col0 + 2 == a.K ? 1.f : 0.f, xlane && col0 + 3 == a.K ? 1.f : 0.f};
  unsigned x_keep[4], x_or[4];                   // per column: keep the loaded pair of bf16 values, or replace it by 1.0 | 1.0 / by zeros
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    x_keep[c] = x_raw[c] ? 0xFFFFFFFFu : 0u;
    x_or[c] = x_fill[c] != 0.f ? 0x3F803F80u : 0u;
  }
  f32x4 acc[8][4];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[t][s] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t n_slabs = a.N / 32;
  const unsigned short* xb = static_cast<const unsigned short*>(a.x);
  using lds_ptr = __attribute__((address_space(3))) void*;
  auto issue = [&](int64_t s, int buf) {          // this wave's four DMA instructions of slab s (clamped: the counts stay uniform)
    const int64_t r0 = 32 * min(s, n_slabs - 1);
    char* base = s_dma + buf * kDmaSlabBytes;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int i = 2 * wid + q;
      const int64_t off = (r0 + 8 * lq + i) * kLW + 8 * lr;
      __builtin_amdgcn_global_load_lds(a.dy + off, (lds_ptr)(base + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(xb + off, (lds_ptr)(base + 8192 + i * 1024), 16, 0, 0);
    }
  };
  // this lane's 8 bytes of an X row inside the region the DMA wrote: chunk colc / 8 of row-lane lq, half (colc % 8) / 4
  const int x_off = 8192 + (lq * 16 + (colc >> 3)) * 16 + (colc & 7) * 2;
  const int64_t G = gridDim.x;
  if (n_slabs > 0) {
#pragma unroll
    for (int s = 0; s < kDmaSlabs - 1; ++s) issue(blockIdx.x + s * G, s);
    int t = 0;
    for (int64_t sl = blockIdx.x; sl < n_slabs; sl += G, ++t) {
      __builtin_amdgcn_s_waitcnt(0x0F70 | (4 * (kDmaSlabs - 2)));      // vmcnt(4 (kDmaSlabs - 2)): this wave's parts of slab t are in LDS
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      issue(sl + (kDmaSlabs - 1) * G, (t + kDmaSlabs - 1) % kDmaSlabs);
      // The slab is read with ds_read instructions the compiler does not see as LDS loads: for a load it sees, its wait-count
      // pass puts s_waitcnt vmcnt(0) in front (any LDS load may alias any pending LDS DMA), i.e. it waits for the slabs that
      // were only just requested -- the whole prefetch.
      const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(s_dma) + (t % kDmaSlabs) * kDmaSlabBytes;
      const unsigned ad_d = lds0 + lane * 16, ad_x = lds0 + x_off;
      u32x4 dv[8];
      uint2 xv[8];
#define MLQEM_DS_READ(j)                                                                                         \
      asm volatile("ds_read_b128 %0, %2 offset:" #j "*1024\n\tds_read_b64 %1, %3 offset:" #j "*1024"               \
                   : "=&v"(dv[j]), "=&v"(xv[j]) : "v"(ad_d), "v"(ad_x));
      MLQEM_DS_READ(0) MLQEM_DS_READ(1) MLQEM_DS_READ(2) MLQEM_DS_READ(3)
      MLQEM_DS_READ(4) MLQEM_DS_READ(5) MLQEM_DS_READ(6) MLQEM_DS_READ(7)
#undef MLQEM_DS_READ
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      // B fragments: column c of this lane's four, rows 8 lq .. 8 lq + 7 -- the 16-bit field c of eight registers, gathered two
      // rows at a time by v_perm, then (raw & keep) | fill: the ones column behind the last input (it makes the bias gradient)
      // and the zero columns beyond it cost one v_and_or each instead of a round trip through fp32.
      bf16x8 bx[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        unsigned w[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const unsigned hi = c < 2 ? xv[2 * d + 1].x : xv[2 * d + 1].y, lo = c < 2 ? xv[2 * d].x : xv[2 * d].y;
          const unsigned raw = __builtin_amdgcn_perm(hi, lo, (c & 1) ? 0x07060302u : 0x05040100u);
          w[d] = (raw & x_keep[c]) | x_or[c];
        }
        const u32x4 v = {w[0], w[1], w[2], w[3]};
        bx[c] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int tt = 0; tt < 8; ++tt) {
        const int c = tt >> 1;
        unsigned ad[4];
#pragma unroll
        for (int d = 0; d < 4; ++d)
          ad[d] = (tt & 1) ? __builtin_amdgcn_perm(vget(dv[2 * d + 1], c), vget(dv[2 * d], c), 0x07060302u)
                           : __builtin_amdgcn_perm(vget(dv[2 * d + 1], c), vget(dv[2 * d], c), 0x05040100u);
        const u32x4 av = {ad[0], ad[1], ad[2], ad[3]};
        const bf16x8 af = __builtin_bit_cast(bf16x8, av);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) acc[tt][s4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bx[s4], acc[tt][s4], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0): the clamped DMAs of the last iterations land before the LDS is given back
  }
  if (blockIdx.x == 0 && (a.N & 31)) {              // the ragged tail: straight from memory, rows beyond N contribute nothing
    u32x4 dv[8];
    float xf[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t row = 32 * n_slabs + 8 * lq + j;
      const int64_t rc = min(row, a.N - 1);
      const uint2 xq = *reinterpret_cast<const uint2*>(xb + rc * kLW + colc);
      dv[j] = *reinterpret_cast<const u32x4*>(a.dy + rc * kLW + 8 * lr);
      if (row >= a.N) dv[j] = u32x4{0u, 0u, 0u, 0u};
      const float v[4] = {blo(xq.x), bhi(xq.x), blo(xq.y), bhi(xq.y)};
#pragma unroll
      for (int c = 0; c < 4; ++c) xf[j][c] = x_raw[c] ? v[c] : x_fill[c];
    }
    bf16x8 bx[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const u32x4 v = {lpack(xf[0][c], xf[1][c]), lpack(xf[2][c], xf[3][c]), lpack(xf[4][c], xf[5][c]), lpack(xf[6][c], xf[7][c])};
      bx[c] = __builtin_bit_cast(bf16x8, v);
    }
#pragma unroll
    for (int tt = 0; tt < 8; ++tt) {
      const int c = tt >> 1;
      unsigned ad[4];
#pragma unroll
      for (int d = 0; d < 4; ++d)
        ad[d] = (tt & 1) ? __builtin_amdgcn_perm(vget(dv[2 * d + 1], c), vget(dv[2 * d], c), 0x07060302u)
                         : __builtin_amdgcn_perm(vget(dv[2 * d + 1], c), vget(dv[2 * d], c), 0x05040100u);
      const u32x4 av = {ad[0], ad[1], ad[2], ad[3]};
      const bf16x8 af = __builtin_bit_cast(bf16x8, av);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) acc[tt][s4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bx[s4], acc[tt][s4], 0, 0, 0);
    }
  }
  float* __restrict__ dst = a.partial + (int64_t)blockIdx.x * kLayerW1Floats + (wid * 32 * 4) * kWave + lane;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[((t * 4 + s4) * 4 + r) * kWave] = acc[t][s4][r];
}

// ---- the bf16 x bf16 weight gradient without a single transpose instruction: ds_read_b64_tr_b16 ------------------------------
// Both MFMA operands of gW = dY^T X want, per lane, eight consecutive ROWS of one column -- the operands are row-major, so the
// kernels above gather them with v_perm (48 per 32 rows and wave: the launch was bound by that instruction stream, not by
// memory).  gfx950's transposing LDS read does the gather in the LDS crossbar: per 16-lane group it reads a 4-row x 16-column
// block of 16-bit elements and hands lane i column i.  The slab image in LDS is what the DMA writes (lane-linear: region i = rows
// {8 lq + i}, 256 B per row), with the 16-byte chunks of row r stored at position chunk ^ 2 g(r), g(r) = (r & 3) | ((r >> 3 & 1) << 2):
// the eight rows one 32-lane half reads together (4 per 16-lane group) then sit on eight different 32-byte bank ranges and the
// transposed reads are conflict-free.  Since the image is filled by DMA, the swizzle costs nothing: a lane simply FETCHES the
// chunk that belongs at its position.
//   tiles: dY columns 16 t .. 16 t + 15 (t = 0..7, every wave) x X columns 16 u .. 16 u + 15 (u = 2 wave, 2 wave + 1);
//   D[4 lq + r][lr] = gW[16 t + 4 lq + r][16 u + lr].  The bias gradient is two more MFMAs per wave against a constant operand
//   that is 1.0 in column 0: D[.][0] = sum over the rows of dY (tiles t = 2 wave, 2 wave + 1).  X needs no fix-up: the pad
//   columns of a bf16 activation are zeros, and whatever they produced would land in columns the second stage drops.
// Partial layout per workgroup: [wave][18 tiles: (j, u_local) row-major with t = (j + 2 wave) & 7, then the two bias tiles][register][lane].
constexpr int kTrTiles = 18;
constexpr int kLayerTrFloats = 4 * kTrTiles * 4 * kWave;
constexpr int kTrSlabs = 4;                 // LDS ring (16 KB per slab)

#define MLQEM_TR_READ(dst, addr, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #imm : "=&v"(dst) : "v"(addr))

__global__ __launch_bounds__(kLayerThreads, 2) void layer_wgrad_tr_kernel(const WgradLayerArgs a) {
  extern __shared__ __attribute__((aligned(16))) char s_dma[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane & 15, lq = lane >> 4, q = lr >> 2, p = lr & 3;
  const int g = q | ((lq & 1) << 2);              // swizzle key of the rows this lane's reads address (8 lq + q and 8 lq + 4 + q)
  f32x4 acc[8][2], accb[2];
#pragma unroll
  for (int t = 0; t < 8; ++t) acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  accb[0] = accb[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned one2 = lr == 0 ? 0x3F803F80u : 0u;     // the constant B operand of the bias tiles: 1.0 in column 0
  const bf16x4 b_ones = __builtin_bit_cast(bf16x4, (uint2{one2, one2}));
  const int64_t n_slabs = a.N / 32;
  const unsigned short* xb = static_cast<const unsigned short*>(a.x);
  using lds_ptr = __attribute__((address_space(3))) void*;
  auto issue = [&](int64_t s, int buf) {          // this wave's four DMA instructions of slab s (clamped: the counts stay uniform)
    const int64_t r0 = 32 * min(s, n_slabs - 1);
    char* base = s_dma + buf * kDmaSlabBytes;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = 2 * wid + k;                  // region i: rows 8 lq + i; this lane fills position lr with chunk lr ^ 2 g(row)
      const int gi = (i & 3) | ((lq & 1) << 2);
      const int64_t off = (r0 + 8 * lq + i) * kLW + 8 * (lr ^ (2 * gi));
      __builtin_amdgcn_global_load_lds(a.dy + off, (lds_ptr)(base + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(xb + off, (lds_ptr)(base + 8192 + i * 1024), 16, 0, 0);
    }
  };
  // byte addresses of this lane's transposed reads inside a slab (second half of the rows: + 4096).  dY tiles in the order
  // j -> t = (j + 2 wave) & 7: every wave reads all eight, starting with the two whose column sums (the bias gradient) are its
  // own -- the bias MFMAs then sit at fixed positions of the instruction stream, no branch, no register indexing.
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(s_dma);
  const unsigned row_off = q * 1024 + lq * 256 + 8 * p;
  unsigned ad_a[8], ad_b[2];
#pragma unroll
  for (int j = 0; j < 8; ++j) ad_a[j] = lds0 + row_off + 32 * (((j + 2 * wid) & 7) ^ g);
#pragma unroll
  for (int u = 0; u < 2; ++u) ad_b[u] = lds0 + 8192 + row_off + 32 * ((2 * wid + u) ^ g);
  // One slab out of the LDS buffer at byte offset `off`.  A transposed read returns four rows of a column = exactly one operand
  // of the k = 16 MFMA, so rows 8 lq .. 8 lq + 3 and 8 lq + 4 .. 8 lq + 7 go into two v_mfma_f32_16x16x16_bf16 (the k index is
  // only a label A and B share) and no register is ever copied or repacked.
  auto consume = [&](unsigned off) {
    bf16x4 al[8], ah[8], bl[2], bh[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) { const unsigned ad = ad_b[u] + off; MLQEM_TR_READ(bl[u], ad, 0); MLQEM_TR_READ(bh[u], ad, 4096); }
#pragma unroll
    for (int j = 0; j < 8; ++j) { const unsigned ad = ad_a[j] + off; MLQEM_TR_READ(al[j], ad, 0); MLQEM_TR_READ(ah[j], ad, 4096); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        acc[j][u] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al[j], bl[u], acc[j][u], 0, 0, 0);
        acc[j][u] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[j], bh[u], acc[j][u], 0, 0, 0);
      }
      if (j < 2) {
        accb[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al[j], b_ones, accb[j], 0, 0, 0);
        accb[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[j], b_ones, accb[j], 0, 0, 0);
      }
    }
  };
  const int64_t G = gridDim.x;
  if (n_slabs > 0) {
#pragma unroll
    for (int s = 0; s < kTrSlabs - 1; ++s) issue(blockIdx.x + s * G, s);
    int t = 0;
    for (int64_t sl = blockIdx.x; sl < n_slabs; sl += G, ++t) {
      __builtin_amdgcn_s_waitcnt(0x0F70 | (4 * (kTrSlabs - 2)));       // vmcnt: this wave's parts of slab t are in LDS
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      issue(sl + (kTrSlabs - 1) * G, (t + kTrSlabs - 1) % kTrSlabs);
      consume((unsigned)(t % kTrSlabs) * kDmaSlabBytes);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0): the clamped DMAs of the last iterations land before the LDS is given back
  }
  if (blockIdx.x == 0 && (a.N & 31)) {              // the ragged tail: its rows go through the same image (rows beyond N as zeros)
    __builtin_amdgcn_s_barrier();                   // every wave is done with the ring
    for (int k = threadIdx.x; k < 2 * 32 * 16; k += kLayerThreads) {
      const int which = k >> 9, row = (k >> 4) & 31, pos = k & 15;
      const int gr = (row & 3) | (((row >> 3) & 1) << 2);
      const int64_t r = 32 * n_slabs + row;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (r < a.N) v = *reinterpret_cast<const u32x4*>((which ? xb : a.dy) + r * kLW + 8 * (pos ^ (2 * gr)));
      *reinterpret_cast<u32x4*>(s_dma + which * 8192 + (row & 7) * 1024 + ((row >> 3) * 16 + pos) * 16) = v;
    }
    __syncthreads();
    consume(0u);
  }
  float* __restrict__ dst = a.partial + (int64_t)blockIdx.x * kLayerTrFloats + (wid * kTrTiles * 4) * kWave + lane;
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[((j * 2 + u) * 4 + r) * kWave] = acc[j][u][r];
#pragma unroll
  for (int tb = 0; tb < 2; ++tb)
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[((16 + tb) * 4 + r) * kWave] = accb[tb][r];
}

// second stage of layer_wgrad_tr_kernel: as layer_wgrad_reduce_kernel, decoding the tile layout above
__global__ __launch_bounds__(16 * kWave) void layer_wgrad_tr_reduce_kernel(const float* __restrict__ partial, int G, int K, int U,
                                                                          float* __restrict__ gw, float* __restrict__ gb) {
  __shared__ float s[16][kWave];
  const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int e = blockIdx.x * kWave + el;
  const int lane = e & 63, r = (e >> 6) & 3, tile = (e >> 8) % kTrTiles, wid = (e >> 8) / kTrTiles;
  const int lr = lane & 15, lq = lane >> 4;
  int o, c;
  bool live = e < kLayerTrFloats;
  // tile j of wave w holds dY tile t = (j + 2 w) & 7 (the wave's reading order); the bias tiles are j = 0, 1
  if (tile < 16) { o = 16 * (((tile >> 1) + 2 * wid) & 7) + 4 * lq + r; c = 16 * (2 * wid + (tile & 1)) + lr; live = live && o < U && c < K; }
  else { o = 16 * ((tile - 16 + 2 * wid) & 7) + 4 * lq + r; c = K; live = live && lr == 0 && o < U; }
  float v = 0.f;
  if (live) {
    const int per = (G + 15) / 16;
    const int g1 = min(G, (sl + 1) * per);
    int g = sl * per;
    float v1 = 0.f, v2 = 0.f, v3 = 0.f;
    for (; g + 3 < g1; g += 4) {
      v += partial[(int64_t)g * kLayerTrFloats + e];
      v1 += partial[(int64_t)(g + 1) * kLayerTrFloats + e];
      v2 += partial[(int64_t)(g + 2) * kLayerTrFloats + e];
      v3 += partial[(int64_t)(g + 3) * kLayerTrFloats + e];
    }
    for (; g < g1; ++g) v += partial[(int64_t)g * kLayerTrFloats + e];
    v = (v + v1) + (v2 + v3);
  }
  s[sl][el] = v;
  __syncthreads();
  if (sl != 0 || !live) return;
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) tot += s[k][el];
  if (c < K) gw[(int64_t)o * K + c] = tot;
  else if (gb) gb[o] = tot;
}

constexpr int kLayerReduceSlices = 16;      // as mlp_head.hip's second stage: 16 slices of the G range per element
__global__ __launch_bounds__(kLayerReduceSlices * kWave) void layer_wgrad_reduce_kernel(const float* __restrict__ partial, int G, int K, int U, int cpw,
                                                                 float* __restrict__ gw, float* __restrict__ gb) {
  __shared__ float s[kLayerReduceSlices][kWave];
  const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int e = blockIdx.x * kWave + el;
  const int lane = e & 63, r = (e >> 6) & 3, tile = (e >> 8) & 31, wid = e >> 13;
  const int lr = lane & 15, lq = lane >> 4, s4 = tile & 3, t = tile >> 2;
  const int o = 8 * (4 * lq + r) + t, c = 4 * (cpw * wid + lr) + s4;
  const bool live = e < kLayerW1Floats && lr < cpw && c <= K && o < U;
  float v = 0.f;
  if (live) {
    const int per = (G + kLayerReduceSlices - 1) / kLayerReduceSlices;
    const int g1 = min(G, (sl + 1) * per);
    int g = sl * per;
    float v1 = 0.f, v2 = 0.f, v3 = 0.f;
    for (; g + 3 < g1; g += 4) {
      v += partial[(int64_t)g * kLayerW1Floats + e];
      v1 += partial[(int64_t)(g + 1) * kLayerW1Floats + e];
      v2 += partial[(int64_t)(g + 2) * kLayerW1Floats + e];
      v3 += partial[(int64_t)(g + 3) * kLayerW1Floats + e];
    }
    for (; g < g1; ++g) v += partial[(int64_t)g * kLayerW1Floats + e];
    v = (v + v1) + (v2 + v3);
  }
  s[sl][el] = v;
  __syncthreads();
  if (sl != 0 || !live) return;
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < kLayerReduceSlices; ++k) tot += s[k][el];
  if (c < K) gw[(int64_t)o * K + c] = tot;
  else if (gb) gb[o] = tot;
}

// ------------------------------------------------------------------------------------------------ the final O <= 4 outputs
// forward: out[n,q] = sum_c h[n,c] w[q,c] + b[q] (operands rounded to bf16 like every GEMM of the mode); 16 lanes per row.
struct DotArgs {
  const void* h; const float* w; const float* b; float* out; int64_t ldo;               // forward (h in the pipeline's storage)
  const float* g; int64_t ldg; void* gh; float* partial;                                   // backward
  float gate_scale;                 // backward, > 0: gh = (h > 0 ? gate_scale : 0) g w -- h = drop(relu(u)) of a block without BatchNorm, gh = du
  int64_t N; int C; int O; int64_t rows_per_block;
};

template <typename ST>
__global__ __launch_bounds__(256) void layer_rowdot_fwd_kernel(const DotArgs a) {
  const int t = threadIdx.x & 15;
  float w[MLQEM_MLP1_MAX_OUT][8];
#pragma unroll
  for (int q = 0; q < MLQEM_MLP1_MAX_OUT; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) w[q][e] = (q < a.O && col_of<ST>(t, e) < a.C) ? stored<ST>(a.w[(int64_t)q * a.C + col_of<ST>(t, e)]) : 0.f;
  const int64_t n_iter = ceil_div(a.N, (int64_t)gridDim.x * 16);
  for (int64_t it = 0; it < n_iter; ++it) {                    // every lane of a 16-lane group stays in the loop (DPP sums)
    const int64_t row = it * gridDim.x * 16 + (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool ok = row < a.N;
    float h[8];
    load8(static_cast<const ST*>(a.h) + min(row, a.N - 1) * kLW, t, h);
#pragma unroll
    for (int q = 0; q < MLQEM_MLP1_MAX_OUT; ++q) {
      if (q >= a.O) break;
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s = fmaf(h[e], w[q][e], s);
      s = group16_sum(s);
      if (ok && t == 0) a.out[row * a.ldo + q] = s + a.b[q];
    }
  }
}

// backward: gh[n,c] = sum_q g[n,q] w[q,c] (storage type), and per-workgroup partials of gw[q,c] = sum_n g[n,q] h[n,c], gb[q] = sum_n g[n,q]:
// partial[block][q][c] for c < 128, partial[block][q][128] = gb.
template <typename ST>
__global__ __launch_bounds__(256) void layer_rowdot_bwd_kernel(const DotArgs a) {
  __shared__ float s_red[16][MLQEM_MLP1_MAX_OUT][kLW + 1];
  const int t = threadIdx.x & 15, rl = threadIdx.x >> 4;
  float w[MLQEM_MLP1_MAX_OUT][8], gw[MLQEM_MLP1_MAX_OUT][8], gb[MLQEM_MLP1_MAX_OUT];
#pragma unroll
  for (int q = 0; q < MLQEM_MLP1_MAX_OUT; ++q) {
    gb[q] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      w[q][e] = (q < a.O && col_of<ST>(t, e) < a.C) ? stored<ST>(a.w[(int64_t)q * a.C + col_of<ST>(t, e)]) : 0.f;
      gw[q][e] = 0.f;
    }
  }
  const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_block, r1 = min(r0 + a.rows_per_block, a.N);
  for (int64_t row = r0 + rl; row < r1; row += 16) {
    float h[8], gh[8];
    load8(static_cast<const ST*>(a.h) + row * kLW, t, h);
#pragma unroll
    for (int e = 0; e < 8; ++e) gh[e] = 0.f;
#pragma unroll
    for (int q = 0; q < MLQEM_MLP1_MAX_OUT; ++q) {
      if (q >= a.O) break;
      const float g = a.g[row * a.ldg + q], gr = stored<ST>(g);
      if (t == 0) gb[q] += g;
#pragma unroll
      for (int e = 0; e < 8; ++e) { gh[e] = fmaf(gr, w[q][e], gh[e]); gw[q][e] = fmaf(gr, h[e], gw[q][e]); }
    }
    if (a.gate_scale > 0.f) {
#pragma unroll
      for (int e = 0; e < 8; ++e) gh[e] = h[e] > 0.f ? gh[e] * a.gate_scale : 0.f;
    }
    store8(static_cast<ST*>(a.gh) + row * kLW, t, gh);
  }
#pragma unroll
  for (int q = 0; q < MLQEM_MLP1_MAX_OUT; ++q) {
#pragma unroll
    for (int e = 0; e < 8; ++e) s_red[rl][q][col_of<ST>(t, e)] = gw[q][e];
    if (t == 0) s_red[rl][q][kLW] = gb[q];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < MLQEM_MLP1_MAX_OUT * (kLW + 1); idx += 256) {
    const int q = idx / (kLW + 1), c = idx % (kLW + 1);
    float tt = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) tt += s_red[j][q][c];
    a.partial[((int64_t)blockIdx.x * MLQEM_MLP1_MAX_OUT + q) * (kLW + 1) + c] = tt;
  }
}

__global__ __launch_bounds__(kFinishThreads) void layer_rowdot_finish_kernel(const float* __restrict__ partial, int nblocks, int C, int O,
                                                                    float* __restrict__ gw, float* __restrict__ gb) {
  __shared__ double s_t[kFinishThreads];
  const int q = blockIdx.x / (kLW + 1), c = blockIdx.x % (kLW + 1), j = threadIdx.x;
  if (q >= O || (c >= C && c != kLW)) return;       // nothing is filed from these sums
  double t = 0.0;
  for (int b = j; b < nblocks; b += kFinishThreads) t += (double)partial[((int64_t)b * MLQEM_MLP1_MAX_OUT + q) * (kLW + 1) + c];
  s_t[j] = t;
  __syncthreads();
  if (j != 0) return;
  t = 0.0;
  for (int k = 0; k < kFinishThreads; ++k) t += s_t[k];
  if (c < C) gw[(int64_t)q * C + c] = (float)t;
  else if (c == kLW) gb[q] = (float)t;
}

// ================================================================================================ fp32 storage: the two GEMMs
// mfma = "f32" (the reference's own arithmetic): activations [N,128] fp32, products on v_mfma_f32_16x16x4_f32 (no rounding of
// the operands).  At 262 144 rows a 128 x 128 layer is 8.6 GFLOP against 157 TFLOP/s of fp32 matrix rate (55 us) and 268 MB
// against ~5 TB/s (50 us): both GEMMs are written to keep the matrix pipes fed, the general kernels of dense.hip ran them at a third.
//
// Forward / data gradient: Y^T tile = W (A operand, from an LDS image) x X^T (B operand, from the rows as they lie in memory).
//   A 16-row tile of X: lane (lr, lq) loads X[row lr][16 j + 4 lq .. + 3] for j < NJ (16-byte loads, 64 contiguous bytes per row
//   and instruction); component c of load j is the B operand of k-step 4 j + c with k = 16 j + 4 lq + c -- k is only a label A
//   and B share, so the image holds W[16 ob + lr][16 j + 4 lq + c] as float4 over c at [(ob NJ + j) 64 + lane]: one conflict-free
//   ds_read_b128 feeds four MFMAs.  D[4 lq + i][lr] = Y[row lr][16 ob + 4 lq + i]: a lane stores 16 bytes per output tile.
//   512 threads: two waves per SIMD share one image (at 64-96 KB only one workgroup fits a CU), one wave's loads and stores
//   hide behind the other's 32-cycle MFMAs.
struct LayerF32Args {
  const float* x; int64_t ldx; const float* w; const float* b; int transposed;
  const float* add;              // optional [N,128] added to the result
  float* y; int64_t ldy; int y_act;   // y_act: an activation matrix [N,128], every tile stored; else the first U columns of [N, ldy]
  int64_t N; int K, U; const void* image;
  int relu; float drop_p; uint64_t seed; const uint64_t* seed_counter;   // epilogue of a block WITHOUT BatchNorm: y = drop(relu(.))
};
constexpr int kF32Threads = 512;
__host__ __device__ inline int layer_f32_image_float4(int NJ, int NOB) { return NOB * NJ * kWave + kLW / 4; }

__global__ __launch_bounds__(256) void layer_f32_image_kernel(const LayerF32Args a, int NJ, int NOB, float4* __restrict__ image) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int n_frag = NOB * NJ * kWave;
  if (idx < n_frag) {
    const int l = idx & 63, j = (idx >> 6) % NJ, ob = idx / (64 * NJ);
    const int u = 16 * ob + (l & 15), lq = l >> 4;
    float w[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k = 16 * j + 4 * lq + c;
      w[c] = (u < a.U && k < a.K) ? (a.transposed ? a.w[(int64_t)k * a.U + u] : a.w[(int64_t)u * a.K + k]) : 0.f;
    }
    image[idx] = make_float4(w[0], w[1], w[2], w[3]);
    return;
  }
  const int t = idx - n_frag;
  if (t < kLW) reinterpret_cast<float*>(image + n_frag)[t] = (a.b && t < a.U) ? a.b[t] : 0.f;
}

template <int NJ, int NOB>
__global__ __launch_bounds__(kF32Threads) void layer_fwd_f32_kernel(const LayerF32Args a) {
  extern __shared__ float4 s_img[];
  for (int idx = threadIdx.x; idx < layer_f32_image_float4(NJ, NOB); idx += kF32Threads) s_img[idx] = static_cast<const float4*>(a.image)[idx];
  __syncthreads();
  const float* s_b = reinterpret_cast<const float*>(s_img + NOB * NJ * kWave);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int64_t n_tiles = ceil_div(a.N, 16);
  const int64_t n_waves = (int64_t)gridDim.x * (kF32Threads / kWave), wave = (int64_t)blockIdx.x * (kF32Threads / kWave) + wid;
  const int kpad = (a.K + 3) / 4 * 4;
  const uint64_t dseed = a.seed + (a.seed_counter ? *a.seed_counter * 0xD1B54A32D192ED03ull : 0ull);
  const unsigned dthr = (unsigned)(a.drop_p * 65536.f);
  const float dinv = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  struct Raw { float4 v[NJ]; };
  auto load_tile = [&](int64_t t, Raw& r) {       // unconditional, clamped loads; the fix-up where the values are consumed
    const float* xr = a.x + min(t * 16 + lr, a.N - 1) * a.ldx;
#pragma unroll
    for (int j = 0; j < NJ; ++j) r.v[j] = *reinterpret_cast<const float4*>(xr + min(16 * j + 4 * lq, kpad - 4));
  };
  auto fix = [&](Raw& r) {                        // columns >= K must not reach the MFMA (x's pad columns are scratch: 0 x NaN)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (16 * j + 16 <= a.K) continue;           // uniform
      const int k0 = 16 * j + 4 * lq;
      if (k0 + 0 >= a.K) r.v[j].x = 0.f;
      if (k0 + 1 >= a.K) r.v[j].y = 0.f;
      if (k0 + 2 >= a.K) r.v[j].z = 0.f;
      if (k0 + 3 >= a.K) r.v[j].w = 0.f;
    }
  };
  Raw cur;
  load_tile(wave, cur);
  fix(cur);
  for (int64_t t = wave; t < n_tiles; t += n_waves) {
    Raw nxt;
    load_tile(t + n_waves, nxt);
    __builtin_amdgcn_sched_barrier(0);            // the prefetch is issued HERE
    f32x4 acc[NOB];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) acc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      // consecutive MFMAs go to DIFFERENT accumulators: four k-steps in a row on one output tile would each wait for the one before
      float4 wv[NOB];
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob) wv[ob] = s_img[(ob * NJ + j) * kWave + lane];
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob) acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ob].x, cur.v[j].x, acc[ob], 0, 0, 0);
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob) acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ob].y, cur.v[j].y, acc[ob], 0, 0, 0);
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob) acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ob].z, cur.v[j].z, acc[ob], 0, 0, 0);
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob) acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ob].w, cur.v[j].w, acc[ob], 0, 0, 0);
    }
    const int64_t row = t * 16 + lr;
    float4 addv[NOB];
    if (a.add) {
      const float* ar = a.add + min(row, a.N - 1) * kLW;
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob) addv[ob] = *reinterpret_cast<const float4*>(ar + 16 * ob + 4 * lq);
    }
    __builtin_amdgcn_sched_barrier(0);
    cur = nxt;
    fix(cur);
    __builtin_amdgcn_sched_barrier(0);
    if (row < a.N) {
      float* yr = a.y + row * a.ldy;
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob) {
        const int u0 = 16 * ob + 4 * lq;
        float4 v = make_float4(acc[ob][0] + s_b[u0], acc[ob][1] + s_b[u0 + 1], acc[ob][2] + s_b[u0 + 2], acc[ob][3] + s_b[u0 + 3]);
        if (a.add) { v.x += addv[ob].x; v.y += addv[ob].y; v.z += addv[ob].z; v.w += addv[ob].w; }
        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (a.drop_p > 0.f) {       // one splitmix64 round per four units, keyed by (row, first unit); the backward gates by y > 0
          uint64_t z = dseed + ((uint64_t)(row * kLW + u0) + 1) * 0x9E3779B97F4A7C15ull;
          z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
          z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
          z = z ^ (z >> 31);
          v.x = (unsigned)(z & 0xFFFFu) >= dthr ? v.x * dinv : 0.f;
          v.y = (unsigned)((z >> 16) & 0xFFFFu) >= dthr ? v.y * dinv : 0.f;
          v.z = (unsigned)((z >> 32) & 0xFFFFu) >= dthr ? v.z * dinv : 0.f;
          v.w = (unsigned)((z >> 48) & 0xFFFFu) >= dthr ? v.w * dinv : 0.f;
        }
        if (a.y_act) {
          *reinterpret_cast<float4*>(yr + u0) = v;          // units >= U: zero image rows, zero bias -> zeros
        } else {
          if (u0 + 0 < a.U) yr[u0 + 0] = v.x;
          if (u0 + 1 < a.U) yr[u0 + 1] = v.y;
          if (u0 + 2 < a.U) yr[u0 + 2] = v.z;
          if (u0 + 3 < a.U) yr[u0 + 3] = v.w;
        }
      }
      if (a.y_act && NOB < 8) {                             // a narrow layer: the rest of the 128 columns are zeros
#pragma unroll
        for (int ob = NOB; ob < 8; ++ob) *reinterpret_cast<float4*>(yr + 16 * ob + 4 * lq) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
}

// Weight gradient gW[U, K] = dY^T X, gb = sum dY (the ones column of X at index K), fp32 operands.  k of the MFMA = 4 ROWS:
//   lane (lr, lq) loads dY[r + lq][64 wu + 4 lr .. + 3] and X[r + lq][64 wc + 4 lr .. + 3] (16 bytes each, 256 contiguous bytes
//   per row and instruction); component cu of the first is the A operand of "unit tile cu" (m = lr <-> unit 64 wu + 4 lr + cu),
//   component cx of the second the B operand of "column tile cx" (n = lr <-> column 64 wc + 4 lr + cx): 16 MFMAs per two loads,
//   no transposes, no LDS.  D[4 lq + i][lr] of tile (cu, cx) = gW[64 wu + 4 (4 lq + i) + cu][64 wc + 4 lr + cx].
//   A wave owns a 64 x 64 block of gW (wu, wc): 64 accumulator registers; a workgroup = NU x NC waves covers [U <= 128] x [K + 1 <= 192].
//   Rows in slabs of 32 dealt to the workgroups round-robin, one slab prefetched in registers; per-workgroup partial sums, added in
//   a fixed order by layer_wgrad_f32_reduce_kernel.
struct WgradF32Args {
  const float* dy; const float* x; int64_t ldx; int64_t N; int K; int NC; float* partial;
};
constexpr int kWgF32WaveFloats = 16 * 4 * kWave;       // a wave's 16 tiles

template <int Q>      // 4-row steps per slab (a slab in flight, one being multiplied: 2 x 2 Q float4 registers)
__global__ __launch_bounds__(384) void layer_wgrad_f32_kernel(const WgradF32Args a) {
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int wu = wid / a.NC, wc = wid - wu * a.NC;
  const int kpad = (a.K + 3) / 4 * 4;
  const int col0 = 64 * wc + 4 * lr;                       // this lane's four X columns
  const int colc = min(col0, kpad - 4);
  // per column: the loaded value, 1.0 (the ones column, index K) or 0.0 (beyond)
  const bool x_raw[4] = {col0 + 0 < a.K, col0 + 1 < a.K, col0 + 2 < a.K, col0 + 3 < a.K};
  const float x_fill[4] = {col0 + 0 == a.K ? 1.f : 0.f, col0 + 1 == a.K ? 1.f : 0.f, col0 + 2 == a.K ? 1.f : 0.f, col0 + 3 == a.K ? 1.f : 0.f};
  f32x4 acc[4][4];
#pragma unroll
  for (int cu = 0; cu < 4; ++cu)
#pragma unroll
    for (int cx = 0; cx < 4; ++cx) acc[cu][cx] = f32x4{0.f, 0.f, 0.f, 0.f};
  struct Slab { float4 d[Q], x[Q]; };
  constexpr int SR = 4 * Q;                                  // rows of a slab
  const int64_t n_slabs = a.N / SR;
  const float* dyl = a.dy + (int64_t)lq * kLW + 64 * wu + 4 * lr;
  const float* xl = a.x + (int64_t)lq * a.ldx + colc;
  auto issue = [&](Slab& s, int64_t sl) {
    const int64_t r0 = SR * min(sl, n_slabs - 1);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      s.d[q] = *reinterpret_cast<const float4*>(dyl + (r0 + 4 * q) * kLW);
      s.x[q] = *reinterpret_cast<const float4*>(xl + (r0 + 4 * q) * a.ldx);
    }
  };
  auto consume = [&](const Slab& s, int64_t first_row) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const bool ok = first_row + 4 * q + lq < a.N;         // the ragged tail: a clamped row contributes nothing
      const float dv[4] = {ok ? s.d[q].x : 0.f, ok ? s.d[q].y : 0.f, ok ? s.d[q].z : 0.f, ok ? s.d[q].w : 0.f};
      const float xr[4] = {s.x[q].x, s.x[q].y, s.x[q].z, s.x[q].w};
      float xv[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) xv[c] = x_raw[c] ? xr[c] : x_fill[c];
#pragma unroll
      for (int cu = 0; cu < 4; ++cu)
#pragma unroll
        for (int cx = 0; cx < 4; ++cx) acc[cu][cx] = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[cu], xv[cx], acc[cu][cx], 0, 0, 0);
    }
  };
  Slab cur, nxt;
  const int64_t G = gridDim.x;
  if (n_slabs > 0) {
    issue(cur, blockIdx.x);
    for (int64_t sl = blockIdx.x; sl < n_slabs; sl += G) {
      issue(nxt, sl + G);
      __builtin_amdgcn_sched_barrier(0);
      consume(cur, SR * sl);
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
    }
  }
  if (blockIdx.x == 0 && (a.N % SR)) {
    Slab t;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int64_t row = min(SR * n_slabs + 4 * q + lq, a.N - 1);
      t.d[q] = *reinterpret_cast<const float4*>(a.dy + row * kLW + 64 * wu + 4 * lr);
      t.x[q] = *reinterpret_cast<const float4*>(a.x + row * a.ldx + colc);
    }
    consume(t, SR * n_slabs);
  }
  float* __restrict__ dst = a.partial + ((int64_t)blockIdx.x * (blockDim.x / kWave) + wid) * kWgF32WaveFloats + lane;
#pragma unroll
  for (int cu = 0; cu < 4; ++cu)
#pragma unroll
    for (int cx = 0; cx < 4; ++cx)
#pragma unroll
      for (int i = 0; i < 4; ++i) dst[((cu * 4 + cx) * 4 + i) * kWave] = acc[cu][cx][i];
}

// second stage: element e of a workgroup's partial = (wave, tile (cu, cx), register i, lane) -> (unit, column); 16 slices of the G range
__global__ __launch_bounds__(16 * kWave) void layer_wgrad_f32_reduce_kernel(const float* __restrict__ partial, int G, int K, int U, int NC, int n_waves,
                                                                           float* __restrict__ gw, float* __restrict__ gb) {
  __shared__ float s[16][kWave];
  const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int e = blockIdx.x * kWave + el;
  const int per_wg = n_waves * kWgF32WaveFloats;
  const int lane = e & 63, i = (e >> 6) & 3, cx = (e >> 8) & 3, cu = (e >> 10) & 3, wid = e >> 12;
  const int lr = lane & 15, lq = lane >> 4;
  const int wu = wid / NC, wc = wid - wu * NC;
  const int o = 64 * wu + 4 * (4 * lq + i) + cu, c = 64 * wc + 4 * lr + cx;
  const bool live = e < per_wg && o < U && c <= K;
  float v = 0.f;
  if (live) {
    const int per = (G + 15) / 16;
    const int g1 = min(G, (sl + 1) * per);
    int g = sl * per;
    float v1 = 0.f, v2 = 0.f, v3 = 0.f;
    for (; g + 3 < g1; g += 4) {
      v += partial[(int64_t)g * per_wg + e];
      v1 += partial[(int64_t)(g + 1) * per_wg + e];
      v2 += partial[(int64_t)(g + 2) * per_wg + e];
      v3 += partial[(int64_t)(g + 3) * per_wg + e];
    }
    for (; g < g1; ++g) v += partial[(int64_t)g * per_wg + e];
    v = (v + v1) + (v2 + v3);
  }
  s[sl][el] = v;
  __syncthreads();
  if (sl != 0 || !live) return;
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) tot += s[k][el];
  if (c < K) gw[(int64_t)o * K + c] = tot;
  else if (gb) gb[o] = tot;
}

template <typename K>
static int layer_resident(K kernel, int threads, size_t lds) {
  int per_cu = 0, dev = 0, cus = 256;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds) != hipSuccess || per_cu < 1) per_cu = 1;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  return per_cu * cus;
}

template <int G2, bool IN_BF16>
static int launch_layer_fwd(LayerArgs a, void* workspace, hipStream_t s) {
  const size_t lds = (size_t)layer_image_u32x4(G2) * sizeof(u32x4);
  auto kernel = layer_fwd_kernel<G2, IN_BF16>;
  if (!ensure_dynamic_lds(kernel, lds)) return MLQEM_ERR_LAUNCH;      // per device (common.hpp)
  static const int res = layer_resident(kernel, kLayerThreads, lds);
  a.image = workspace;
  hipLaunchKernelGGL(layer_image_kernel, dim3((unsigned)ceil_div(layer_image_u32x4(G2) * 4, 256)), dim3(256), 0, s, a, G2, static_cast<u32x4*>(workspace));
  const int64_t tiles = ceil_div(a.N, 16);
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(res, ceil_div(tiles, kLayerThreads / kWave)));
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(kLayerThreads), lds, s, a);
  return launch_status();
}

constexpr int kColsumMaxBlocks = 2048;
static int colsum_blocks(int64_t N) {
  const int64_t want = ceil_div(N, (int64_t)64);
  return (int)(want < 1 ? 1 : (want > kColsumMaxBlocks ? kColsumMaxBlocks : want));
}

}  // namespace mlqem

using namespace mlqem;

extern "C" size_t mlqem_layer_workspace_bytes(void) {
  const size_t image = (size_t)layer_image_u32x4(6) * sizeof(u32x4);
  const size_t wgrad = (size_t)kLayerMaxBlocks * kLayerW1Floats * sizeof(float);
  const size_t colsum = ((size_t)kColsumMaxBlocks * 2 + 1) * kLW * sizeof(float);      // per-workgroup partial sums | the shift (row 0)
  const size_t dot = (size_t)kColsumMaxBlocks * MLQEM_MLP1_MAX_OUT * (kLW + 1) * sizeof(float);
  return std::max(std::max(image, wgrad), std::max(colsum, dot));
}

extern "C" int mlqem_layer_gemm_bf16(const void* x, int x_is_bf16, int64_t ldx, const float* w, int transposed, const float* b,
                                     const void* add_bf16, void* y, int y_is_f32, int64_t ldy, int relu, float drop_p, uint64_t seed,
                                     const uint64_t* seed_counter, int64_t N, int K, int U, void* workspace, size_t workspace_bytes,
                                     mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || K < 1 || U < 1 || !w || drop_p < 0.f || drop_p >= 1.f) return MLQEM_ERR_BAD_ARG;
  if (U > kLW || K > (x_is_bf16 ? kLW : 192)) return MLQEM_ERR_UNSUPPORTED;
  if (!x_is_bf16 && (ldx < (K + 3) / 4 * 4 || ldx % 4)) return MLQEM_ERR_BAD_ARG;
  if (y_is_f32 && ldy < U) return MLQEM_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < mlqem_layer_workspace_bytes() || !aligned_to(workspace, 16)) return MLQEM_ERR_WORKSPACE;
  if (N == 0) return MLQEM_OK;
  if (!x || !y || !aligned_to(x, 16) || (!y_is_f32 && !aligned_to(y, 16)) || (add_bf16 && !aligned_to(add_bf16, 16))) return MLQEM_ERR_BAD_ARG;
  LayerArgs a{x, ldx, x_is_bf16, w, b, transposed, static_cast<const unsigned short*>(add_bf16), y, y_is_f32, ldy, N, K, U, nullptr,
              relu, drop_p, seed, seed_counter};
  hipStream_t s = as_stream(stream);
  const int g2 = (K + 31) / 32;
  if (x_is_bf16) {
    if (g2 <= 2) return launch_layer_fwd<2, true>(a, workspace, s);
    return launch_layer_fwd<4, true>(a, workspace, s);
  }
  if (g2 <= 2) return launch_layer_fwd<2, false>(a, workspace, s);
  if (g2 <= 4) return launch_layer_fwd<4, false>(a, workspace, s);
  return launch_layer_fwd<6, false>(a, workspace, s);
}

// mode 0: batch statistics of y -> mean, var, invstd, scale, shift (each [128]; gamma / beta [C]).
// mode 1: backward sums from (g, y) -> dbeta, dgamma, gs, k1, k2.  g: bf16 [N,128], or fp32 [N, ldg32] when g32 != NULL.
template <typename ST>
static int layer_colstats(int mode, const void* y, const void* g, const float* g32, int64_t ldg32, const float* scale,
                          const float* shift, const float* mean, const float* invstd, const float* gamma,
                          const float* beta, float eps, int relu, float drop_p, uint64_t seed,
                          const uint64_t* seed_counter, int64_t N, int C, float* o1, float* o2, float* o3, float* o4,
                          float* o5, float* running_mean, float* running_var, float momentum,
                          int64_t* num_batches_tracked, void* workspace, size_t workspace_bytes, mlqem_stream_t stream) {
  begin_launches();
  if (N <= 0 || C < 1 || C > kLW || !y || !gamma || !o1 || !o2 || !o3 || !o4 || !o5 || drop_p < 0.f || drop_p >= 1.f) return MLQEM_ERR_BAD_ARG;
  if (mode == 0 ? !beta : (!scale || !shift || !mean || !invstd || (!g && !g32))) return MLQEM_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < mlqem_layer_workspace_bytes()) return MLQEM_ERR_WORKSPACE;
  hipStream_t s = as_stream(stream);
  int nb = colsum_blocks(N);
  {   // a fixed partition of the rows over the workgroups: a whole number of resident rounds (the backward sums fit five workgroups
      // per CU: 2048 of them ran as one round and a second one at 60 % of the occupancy)
    static const int r0 = layer_resident(layer_colsum_kernel<0, ST>, 256, 0), r1 = layer_resident(layer_colsum_kernel<1, ST>, 256, 0);
    const int res = mode == 0 ? r0 : r1;
    if (nb > res) nb = nb / res * res;
  }
  ActArgs a{};
  a.y = y; a.g = g; a.g32 = g32; a.ldg32 = ldg32;
  a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.partial = static_cast<float*>(workspace);
  a.N = N; a.C = C; a.relu = relu; a.drop_p = drop_p; a.seed = seed; a.seed_counter = seed_counter;
  a.rows_per_block = ceil_div(N, (int64_t)nb);
  if (mode == 0) {
    hipLaunchKernelGGL((layer_colsum_kernel<0, ST>), dim3(nb), dim3(256), 0, s, a);
    if ((running_mean == nullptr) != (running_var == nullptr) || (running_mean && N < 2)) return MLQEM_ERR_BAD_ARG;
    hipLaunchKernelGGL(layer_finish_kernel<0>, dim3(kLW), dim3(kFinishThreads), 0, s, a.partial, nb, N, C, gamma, beta, (const float*)nullptr, eps,
                       o1, o2, o3, o4, o5, running_mean, running_var, momentum, reinterpret_cast<long long*>(num_batches_tracked));
  } else {
    hipLaunchKernelGGL((layer_colsum_kernel<1, ST>), dim3(nb), dim3(256), 0, s, a);
    hipLaunchKernelGGL(layer_finish_kernel<1>, dim3(kLW), dim3(kFinishThreads), 0, s, a.partial, nb, N, C, gamma, (const float*)nullptr, invstd, 0.f,
                       o1, o2, o3, o4, o5, (float*)nullptr, (float*)nullptr, 0.f, (long long*)nullptr);
  }
  return launch_status();
}

#define MLQEM_COLSTATS_ARGS                                                                                                       \
  int mode, const void* y, const void* g, const float* g32, int64_t ldg32, const float* scale, const float* shift, const float* mean, \
      const float* invstd, const float* gamma, const float* beta, float eps, int relu, float drop_p, uint64_t seed,                   \
      const uint64_t* seed_counter, int64_t N, int C, float* o1, float* o2, float* o3, float* o4, float* o5, float* running_mean,     \
      float* running_var, float momentum, int64_t* num_batches_tracked, void* workspace, size_t workspace_bytes, mlqem_stream_t stream
#define MLQEM_COLSTATS_PASS                                                                                                        \
  mode, y, g, g32, ldg32, scale, shift, mean, invstd, gamma, beta, eps, relu, drop_p, seed, seed_counter, N, C, o1, o2, o3, o4, o5,  \
      running_mean, running_var, momentum, num_batches_tracked, workspace, workspace_bytes, stream
extern "C" int mlqem_layer_colstats_bf16(MLQEM_COLSTATS_ARGS) { return layer_colstats<unsigned short>(MLQEM_COLSTATS_PASS); }
extern "C" int mlqem_layer_colstats_f32(MLQEM_COLSTATS_ARGS) { return layer_colstats<float>(MLQEM_COLSTATS_PASS); }
#undef MLQEM_COLSTATS_ARGS
#undef MLQEM_COLSTATS_PASS

// op 0: out = drop(relu?(y scale + shift)) (+ res).   op 1: out = gs (gu - k1 - xhat k2), gu from (g | g32, y).
template <typename ST>
static int layer_pointwise(int op, const void* y, const void* g, const float* g32, int64_t ldg32, const void* res,
                           const float* scale, const float* shift, const float* mean, const float* invstd,
                           const float* gs, const float* k1, const float* k2, int relu, float drop_p, uint64_t seed,
                           const uint64_t* seed_counter, void* out, int64_t N, int C, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || C < 1 || C > kLW || drop_p < 0.f || drop_p >= 1.f) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!y || !out || !scale || !shift) return MLQEM_ERR_BAD_ARG;
  if (op == 1 && (!mean || !invstd || !gs || !k1 || !k2 || (!g && !g32))) return MLQEM_ERR_BAD_ARG;
  ActArgs a{};
  a.y = y; a.g = g; a.g32 = g32; a.ldg32 = ldg32;
  a.res = res; a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd;
  a.gs = gs; a.k1 = k1; a.k2 = k2; a.out = out;
  a.N = N; a.C = C; a.relu = relu; a.drop_p = drop_p; a.seed = seed; a.seed_counter = seed_counter;
  const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(N, 16), 256 * 8);
  if (op == 0) hipLaunchKernelGGL(layer_act_kernel<ST>, dim3(grid), dim3(256), 0, as_stream(stream), a);
  else hipLaunchKernelGGL(layer_bwd_apply_kernel<ST>, dim3(grid), dim3(256), 0, as_stream(stream), a);
  return launch_status();
}
#define MLQEM_POINTWISE_ARGS                                                                                                          \
  int op, const void* y, const void* g, const float* g32, int64_t ldg32, const void* res, const float* scale, const float* shift,       \
      const float* mean, const float* invstd, const float* gs, const float* k1, const float* k2, int relu, float drop_p, uint64_t seed, \
      const uint64_t* seed_counter, void* out, int64_t N, int C, mlqem_stream_t stream
#define MLQEM_POINTWISE_PASS op, y, g, g32, ldg32, res, scale, shift, mean, invstd, gs, k1, k2, relu, drop_p, seed, seed_counter, out, N, C, stream
extern "C" int mlqem_layer_pointwise_bf16(MLQEM_POINTWISE_ARGS) { return layer_pointwise<unsigned short>(MLQEM_POINTWISE_PASS); }
extern "C" int mlqem_layer_pointwise_f32(MLQEM_POINTWISE_ARGS) { return layer_pointwise<float>(MLQEM_POINTWISE_PASS); }
#undef MLQEM_POINTWISE_ARGS
#undef MLQEM_POINTWISE_PASS

extern "C" int mlqem_layer_wgrad_bf16(const void* dy, const void* x, int x_is_bf16, int64_t ldx, float* gw, float* gb, int64_t N,
                                      int K, int U, void* workspace, size_t workspace_bytes, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || K < 1 || U < 1 || !gw) return MLQEM_ERR_BAD_ARG;
  if (U > kLW || K > (x_is_bf16 ? kLW : MLQEM_MLP1_MAX_IN)) return MLQEM_ERR_UNSUPPORTED;
  if (!x_is_bf16 && (ldx < (K + 3) / 4 * 4 || ldx % 4)) return MLQEM_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < mlqem_layer_workspace_bytes()) return MLQEM_ERR_WORKSPACE;
  if (N > 0 && (!dy || !x || !aligned_to(dy, 16) || !aligned_to(x, 8))) return MLQEM_ERR_BAD_ARG;
  hipStream_t s = as_stream(stream);
  WgradLayerArgs a{static_cast<const unsigned short*>(dy), x, ldx, x_is_bf16, N, K, static_cast<float*>(workspace)};
  const int chunks = (K + 1 + 3) / 4, cpw = (chunks + 3) / 4;
  int G = 0;
  if (N > 0) {
    static const int r0 = layer_resident(layer_wgrad_kernel<false>, kLayerThreads, 0), r1 = layer_resident(layer_wgrad_kernel<true>, kLayerThreads, 0);
    G = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(x_is_bf16 ? r1 : r0, kLayerMaxBlocks), std::max<int64_t>(N / 32, 1)));
    constexpr int lds_form = 2;      // (was the A/B switch MLQEM_LAYER_WGRAD_LDS: settled)    // 0: register prefetch, 1: LDS DMA + v_perm, 2: LDS DMA + transposing reads
    const bool dma_ok = aligned_to(dy, 16) && aligned_to(x, 16);      // the DMA forms move 16 bytes per lane
    if (x_is_bf16 && lds_form == 2 && dma_ok) {    // the transposing-read form (default): see layer_wgrad_tr_kernel
      const size_t lds = (size_t)kTrSlabs * kDmaSlabBytes;
      if (!ensure_dynamic_lds(layer_wgrad_tr_kernel, lds)) return MLQEM_ERR_LAUNCH;      // per device (common.hpp)
      static const int rl = layer_resident(layer_wgrad_tr_kernel, kLayerThreads, lds);
      G = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(rl, kLayerMaxBlocks), std::max<int64_t>(N / 32, 1)));
      hipLaunchKernelGGL(layer_wgrad_tr_kernel, dim3(G), dim3(kLayerThreads), lds, s, a);
      hipLaunchKernelGGL(layer_wgrad_tr_reduce_kernel, dim3((unsigned)ceil_div(kLayerTrFloats, kWave)), dim3(16 * kWave), 0, s, a.partial, G, K, U, gw, gb);
      return launch_status();
    }
    if (x_is_bf16 && lds_form && dma_ok) {
      auto go = [&](auto kernel, int slabs) {
        const size_t lds = (size_t)slabs * kDmaSlabBytes;
        if (!ensure_dynamic_lds(kernel, lds)) return false;
        const int rl = layer_resident(kernel, kLayerThreads, lds);
        G = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(rl, kLayerMaxBlocks), std::max<int64_t>(N / 32, 1)));
        hipLaunchKernelGGL(kernel, dim3(G), dim3(kLayerThreads), lds, s, a, cpw);
        return true;
      };
      // depth 2, 3, 4 and 5 measure the same (34-35 us at 262 144 rows): with the operands in LDS the launch is bound by its
      // instruction stream (32 MFMAs + ~100 vector instructions per 32 rows and wave: the 8 x 8 transposes of the 16-bit operands
      // by v_perm), not by latency any more; ds_read_b64_tr_b16 on a swizzled image would remove the transposes
      if (!go(layer_wgrad_lds_kernel<4>, 4)) return MLQEM_ERR_LAUNCH;
    } else if (x_is_bf16) hipLaunchKernelGGL(layer_wgrad_kernel<true>, dim3(G), dim3(kLayerThreads), 0, s, a, cpw);
    else hipLaunchKernelGGL(layer_wgrad_kernel<false>, dim3(G), dim3(kLayerThreads), 0, s, a, cpw);
  }
  hipLaunchKernelGGL(layer_wgrad_reduce_kernel, dim3((unsigned)ceil_div(kLayerW1Floats, kWave)), dim3(kLayerReduceSlices * kWave), 0, s, a.partial, G, K, U, cpw, gw, gb);
  return launch_status();
}

template <typename ST>
static int layer_rowdot(const void* h, const float* w, const float* b, float* out, int64_t ldo, int64_t N, int C, int O, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || C < 1 || C > kLW || O < 1 || O > MLQEM_MLP1_MAX_OUT || ldo < O) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!h || !w || !b || !out) return MLQEM_ERR_BAD_ARG;
  DotArgs a{};
  a.h = h; a.w = w; a.b = b; a.out = out; a.ldo = ldo; a.N = N; a.C = C; a.O = O;
  hipLaunchKernelGGL(layer_rowdot_fwd_kernel<ST>, dim3((unsigned)std::min<int64_t>(ceil_div(N, 16), 256 * 16)), dim3(256), 0, as_stream(stream), a);
  return launch_status();
}

extern "C" int mlqem_layer_rowdot_bf16(const void* h, const float* w, const float* b, float* out, int64_t ldo, int64_t N, int C, int O,
                                       mlqem_stream_t stream) {
  return layer_rowdot<unsigned short>(h, w, b, out, ldo, N, C, O, stream);
}
extern "C" int mlqem_layer_rowdot_f32(const void* h, const float* w, const float* b, float* out, int64_t ldo, int64_t N, int C, int O,
                                      mlqem_stream_t stream) {
  return layer_rowdot<float>(h, w, b, out, ldo, N, C, O, stream);
}

template <typename ST>
static int layer_rowdot_bwd(const float* g, int64_t ldg, const void* h, const float* w, void* gh, float gate_scale, float* gw, float* gb,
                            int64_t N, int C, int O, void* workspace, size_t workspace_bytes, mlqem_stream_t stream) {
  begin_launches();
  if (N <= 0 || C < 1 || C > kLW || O < 1 || O > MLQEM_MLP1_MAX_OUT || ldg < O || !g || !h || !w || !gh || !gw || !gb) return MLQEM_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < mlqem_layer_workspace_bytes()) return MLQEM_ERR_WORKSPACE;
  hipStream_t s = as_stream(stream);
  const int nb = colsum_blocks(N);
  DotArgs a{};
  a.h = h; a.w = w; a.g = g; a.ldg = ldg; a.gh = gh; a.gate_scale = gate_scale;
  a.partial = static_cast<float*>(workspace); a.N = N; a.C = C; a.O = O; a.rows_per_block = ceil_div(N, (int64_t)nb);
  hipLaunchKernelGGL(layer_rowdot_bwd_kernel<ST>, dim3(nb), dim3(256), 0, s, a);
  hipLaunchKernelGGL(layer_rowdot_finish_kernel, dim3(MLQEM_MLP1_MAX_OUT * (kLW + 1)), dim3(kFinishThreads), 0, s, a.partial, nb, C, O, gw, gb);
  return launch_status();
}
extern "C" int mlqem_layer_rowdot_bwd_bf16(const float* g, int64_t ldg, const void* h, const float* w, void* gh, float gate_scale, float* gw,
                                           float* gb, int64_t N, int C, int O, void* workspace, size_t workspace_bytes, mlqem_stream_t stream) {
  return layer_rowdot_bwd<unsigned short>(g, ldg, h, w, gh, gate_scale, gw, gb, N, C, O, workspace, workspace_bytes, stream);
}
extern "C" int mlqem_layer_rowdot_bwd_f32(const float* g, int64_t ldg, const void* h, const float* w, void* gh, float gate_scale, float* gw,
                                          float* gb, int64_t N, int C, int O, void* workspace, size_t workspace_bytes, mlqem_stream_t stream) {
  return layer_rowdot_bwd<float>(g, ldg, h, w, gh, gate_scale, gw, gb, N, C, O, workspace, workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------------ fp32 storage: entry points
template <int NJ, int NOB>
static int launch_layer_fwd_f32(LayerF32Args a, void* workspace, hipStream_t s) {
  const size_t lds = (size_t)layer_f32_image_float4(NJ, NOB) * sizeof(float4);
  auto kernel = layer_fwd_f32_kernel<NJ, NOB>;
  if (!ensure_dynamic_lds(kernel, lds)) return MLQEM_ERR_LAUNCH;      // per device (common.hpp)
  static const int res = layer_resident(kernel, kF32Threads, lds);
  a.image = workspace;
  hipLaunchKernelGGL(layer_f32_image_kernel, dim3((unsigned)ceil_div(layer_f32_image_float4(NJ, NOB) * 4, 256)), dim3(256), 0, s, a, NJ, NOB,
                     static_cast<float4*>(workspace));
  const int64_t tiles = ceil_div(a.N, 16);
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(res, ceil_div(tiles, kF32Threads / kWave)));
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(kF32Threads), lds, s, a);
  return launch_status();
}

// Y = X W^T + b (or, transposed, X W) with fp32 rows in and out: x [N, ldx] (K columns used; an activation matrix has ldx = 128),
// y an activation matrix [N,128] (y_is_act; every column written, zeros beyond U) or the first U columns of [N, ldy].
extern "C" int mlqem_layer_gemm_f32(const float* x, int64_t ldx, const float* w, int transposed, const float* b, const float* add,
                                    float* y, int y_is_act, int64_t ldy, int relu, float drop_p, uint64_t seed, const uint64_t* seed_counter,
                                    int64_t N, int K, int U, void* workspace, size_t workspace_bytes, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || K < 1 || U < 1 || !w || drop_p < 0.f || drop_p >= 1.f) return MLQEM_ERR_BAD_ARG;
  if (U > kLW || K > 192) return MLQEM_ERR_UNSUPPORTED;
  if (ldx < (K + 3) / 4 * 4 || ldx % 4 || (y_is_act ? ldy != kLW : ldy < U)) return MLQEM_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < mlqem_layer_workspace_bytes() || !aligned_to(workspace, 16)) return MLQEM_ERR_WORKSPACE;
  if (N == 0) return MLQEM_OK;
  if (!x || !y || !aligned_to(x, 16) || (y_is_act && !aligned_to(y, 16)) || (add && !aligned_to(add, 16))) return MLQEM_ERR_BAD_ARG;
  LayerF32Args a{x, ldx, w, b, transposed, add, y, ldy, y_is_act, N, K, U, nullptr, relu, drop_p, seed, seed_counter};
  hipStream_t s = as_stream(stream);
  const int nj = (K + 15) / 16, narrow = U <= 64;
  if (nj <= 4) return narrow ? launch_layer_fwd_f32<4, 4>(a, workspace, s) : launch_layer_fwd_f32<4, 8>(a, workspace, s);
  if (nj <= 8) return narrow ? launch_layer_fwd_f32<8, 4>(a, workspace, s) : launch_layer_fwd_f32<8, 8>(a, workspace, s);
  return narrow ? launch_layer_fwd_f32<12, 4>(a, workspace, s) : launch_layer_fwd_f32<12, 8>(a, workspace, s);
}

// gw [U, K] = dy^T x, gb [U] = column sums of dy: dy an fp32 activation matrix [N,128], x fp32 [N, ldx] (K <= 191 columns).
extern "C" int mlqem_layer_wgrad_f32(const float* dy, const float* x, int64_t ldx, float* gw, float* gb, int64_t N, int K, int U,
                                     void* workspace, size_t workspace_bytes, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || K < 1 || U < 1 || !gw) return MLQEM_ERR_BAD_ARG;
  if (U > kLW || K > 191) return MLQEM_ERR_UNSUPPORTED;
  if (ldx < (K + 3) / 4 * 4 || ldx % 4) return MLQEM_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < mlqem_layer_workspace_bytes()) return MLQEM_ERR_WORKSPACE;
  if (N > 0 && (!dy || !x || !aligned_to(dy, 16) || !aligned_to(x, 16))) return MLQEM_ERR_BAD_ARG;
  hipStream_t s = as_stream(stream);
  const int NU = (U + 63) / 64, NC = (K + 1 + 63) / 64, n_waves = NU * NC;       // <= 2 x 3 waves
  WgradF32Args a{dy, x, ldx, N, K, NC, static_cast<float*>(workspace)};
  int G = 0;
  if (N > 0) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      int v = 0;
      if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    // the partial sums of a workgroup are n_waves x 16 KB
    const int64_t cap = (int64_t)(mlqem_layer_workspace_bytes() / ((size_t)n_waves * kWgF32WaveFloats * sizeof(float)));
    // persistent workgroups: exactly as many as are resident at once (at 210 registers two waves share a SIMD: eight waves per CU)
    constexpr int q_env = 8;      // (was the A/B switch MLQEM_LAYER_WGRAD_Q: settled)
    const bool q4 = q_env == 4;
    static int per_cu_of[8] = {};                    // by workgroup size (1..6 waves): asked of the runtime once, not per call
    int per_cu = per_cu_of[n_waves & 7];
    if (per_cu == 0) {
      if ((q4 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, layer_wgrad_f32_kernel<4>, n_waves * kWave, 0)
              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, layer_wgrad_f32_kernel<8>, n_waves * kWave, 0)) != hipSuccess || per_cu < 1)
        per_cu = 1;
      per_cu_of[n_waves & 7] = per_cu;
    }
    const int64_t want = (int64_t)cus * per_cu;
    G = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(want, cap), std::max<int64_t>(N / 32, 1)));
    if (q4) hipLaunchKernelGGL(layer_wgrad_f32_kernel<4>, dim3(G), dim3(n_waves * kWave), 0, s, a);
    else hipLaunchKernelGGL(layer_wgrad_f32_kernel<8>, dim3(G), dim3(n_waves * kWave), 0, s, a);
  }
  hipLaunchKernelGGL(layer_wgrad_f32_reduce_kernel, dim3((unsigned)ceil_div(n_waves * kWgF32WaveFloats, kWave)), dim3(16 * kWave), 0, s,
                     a.partial, G, K, U, NC, n_waves, gw, gb);
  return launch_status();
}
