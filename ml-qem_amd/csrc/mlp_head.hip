// The MLP regressor head as ONE forward and ONE backward launch (reference: docs/tutorials/mlp.py:18-30, MLP1 =
// fc2(relu(fc1 x)); the demo feature sets are 58 / 169 / 170 columns wide, hidden 64 / 128, 1 or 4 outputs).
//
// Shapes: N rows (circuits, 1e5..1e6 at the mixed-corpus scale), I <= 175 inputs, H <= 128 hidden, O2 <= 4 outputs.
// What bounds it: 170 x 128 is 43.5 kFLOP per 688-byte row = 63 FLOP/B, three times the ridge of the fp32 matrix cores
// (157 TFLOP/s / 8 TB/s = 20 FLOP/B): in fp32 the head is MFMA-bound (2 x 11.4 GFLOP at 262 144 rows = 146 us at peak), on
// the bf16 matrix cores (16x the rate) it is HBM-bound.  So
//   * forward: fc1 on the matrix cores with the whole W1 in LDS as ready-made A fragments, bias + ReLU + fc2 (a dot product
//     per output) in the epilogue -- the hidden activation never goes to memory as an operand, only as the STASH the
//     backward needs: fp32 [N,128] in fp32 mode (exact), bf16 [N,128] in bf16 mode (half the bytes);
//   * backward: one pass over x and the stash gives gW1, gb1, gW2, gb2 -- the hidden gradient gh = (gout w2) o (h > 0) is
//     formed in registers (the input needs no gradient: docs/tutorials/mlp.py trains on a plain feature matrix).
// fp32 mode is exact fp32 (v_mfma_f32_16x16x4_f32 = a k-ordered fmaf chain); bf16 mode rounds every GEMM operand to bf16
// (round-to-nearest-even), accumulates in fp32 on v_mfma_f32_16x16x32_bf16 and keeps h as bf16.
//
// Lane maps (guide section 3).  f32 16x16x4: A: lane l holds A[l & 15][l >> 4]; B: lane l holds B[l >> 4][l & 15].
// bf16 16x16x32: A: lane l holds A[l & 15][8 (l >> 4) + j], j = 0..7; B: B[8 (l >> 4) + j][l & 15].  C/D (both): lane l,
// register r holds D[4 (l >> 4) + r][l & 15].  The MFMA sums over k in any order as long as A and B agree, and the m / n
// indices of a tile may stand for any 16 rows / columns -- both freedoms are used to make every global access 16 bytes wide.
#include "common.hpp"

namespace mlqem {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kHeadH = MLQEM_MLP1_HIDDEN_PAD;   // columns of the stash (hidden width padded to 128)
constexpr int kHeadMaxOut = MLQEM_MLP1_MAX_OUT;
constexpr int kFwdThreads = 256;                // the waves of a workgroup share one W1 image in LDS
constexpr int kBwdThreads = 256;
#ifndef MLQEM_HEAD_BWD_KU
#define MLQEM_HEAD_BWD_KU 8    // k-steps (of 4 rows) per iteration of the fp32 backward: one iteration = the prefetch distance; 8 spills (two register sets)
#endif

struct Mlp1Args {
  const float* x; int64_t ldx; int64_t N; int I, H, O2;
  const float* w1; const float* b1; const float* w2; const float* b2;   // [H,I], [H], [O2,H], [O2]
  void* h;                      // stash [N,128]: float (fp32 mode) or bf16 (bf16 mode); forward may pass nullptr (inference)
  float* out; int64_t ldo;      // forward: [N,O2]
  const float* gout; int64_t ldg;   // backward: d loss / d out, [N,O2]
  float* partial;               // backward: per-workgroup partial sums (kHeadPartialFloats each)
  const void* image;            // forward: the LDS image built by mlp1_image_kernel
  // forward with the loss folded in (mlqem_mlp1_forward with a target): gout_w = 2 (out - target) / (N O2) is written beside
  // out, every workgroup leaves its sum of (out - target)^2 in loss_part[blockIdx.x], workgroup 0 the workgroup count behind them
  const float* target; int64_t ldt; float* gout_w; int64_t ldgw; float* loss_part;
};

constexpr int kHeadLossSlots = 1024;      // >= the forward's workgroup count (resident workgroups: at most a few per CU)

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const bf16x2 v = {(__bf16)lo, (__bf16)hi};             // one v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf16_round(float f) { return (float)(__bf16)f; }
__device__ __forceinline__ float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xFFFF0000u); }

// Output tile ob (0..7), tile row m (0..15) <-> hidden unit: lane group q = m >> 2 ends up with EIGHT consecutive units of
// every tile pair (32 (ob >> 1) + 8 q + 0..7), i.e. one 16-byte (bf16) or two 16-byte (fp32) stores per pair.
__device__ __forceinline__ int head_unit(int ob, int m) { return 32 * (ob >> 1) + 8 * (m >> 2) + 4 * (ob & 1) + (m & 3); }

// ------------------------------------------------------------------------------------------------ forward
// The LDS image of a forward workgroup: W1 as ready-made A fragments (8 output tiles x G k-groups x 64 lanes x 16 bytes),
// then b1[128], w2[4][128].  It is built ONCE per call by a small kernel into the workspace and copied into LDS by every
// workgroup with coalesced 16-byte loads: building it in each of the 256 workgroups from the row-major weights (scalar
// loads of an unaligned 170-float row, 88 per thread) took 30 us of a 140 us launch.
__host__ __device__ inline int head_image_u32x4(int G) { return 8 * G * kWave + (kHeadH + kHeadMaxOut * kHeadH) / 4; }

template <bool BF16>
__global__ __launch_bounds__(256) void mlp1_image_kernel(const Mlp1Args a, int G, u32x4* __restrict__ image) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int n_frag = 8 * G * kWave;
  if (idx < n_frag) {
    const int l = idx & 63, g = (idx >> 6) % G, ob = idx / (64 * G);
    const int o = head_unit(ob, l & 15), lq = l >> 4;
    const float* wr = a.w1 + (int64_t)o * a.I;
    u32x4 v;
    if constexpr (BF16) {
      float w[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 32 * g + 8 * lq + j;
        w[j] = (o < a.H && k < a.I) ? wr[k] : 0.f;
      }
      v = u32x4{pack_bf16(w[0], w[1]), pack_bf16(w[2], w[3]), pack_bf16(w[4], w[5]), pack_bf16(w[6], w[7])};
    } else {
      float w[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int k = 16 * g + 4 * lq + s4;
        w[s4] = (o < a.H && k < a.I) ? wr[k] : 0.f;
      }
      v = u32x4{__builtin_bit_cast(unsigned, w[0]), __builtin_bit_cast(unsigned, w[1]), __builtin_bit_cast(unsigned, w[2]),
                __builtin_bit_cast(unsigned, w[3])};
    }
    image[idx] = v;
    return;
  }
  float* tail = reinterpret_cast<float*>(image + n_frag);
  const int t = idx - n_frag;
  if (t < kHeadH) {
    tail[t] = t < a.H ? a.b1[t] : 0.f;
  } else if (t < kHeadH + kHeadMaxOut * kHeadH) {
    const int q = (t - kHeadH) / kHeadH, o = (t - kHeadH) % kHeadH;
    float w = (q < a.O2 && o < a.H) ? a.w2[(int64_t)q * a.H + o] : 0.f;
    if (BF16) w = bf16_round(w);
    tail[t] = w;
  }
}

template <int G>
__device__ __forceinline__ void head_fill_lds(const u32x4* __restrict__ image, u32x4* s_raw) {
  for (int idx = threadIdx.x; idx < head_image_u32x4(G); idx += kFwdThreads) s_raw[idx] = image[idx];
}

// bias + ReLU, the stash, fc2: the lane holds 32 hidden units of row `row` (hv[p][e] = unit 32 p + 8 lq + e).
template <bool BF16>
__device__ __forceinline__ void head_epilogue(const Mlp1Args& a, const f32x4 (&acc)[8], const float* s_b1, const float* s_w2,
                                              const float (&b2r)[kHeadMaxOut], int64_t row, int lq, float& lsum,
                                              const float (&tgt)[kHeadMaxOut]) {
  float hv[4][8];
#pragma unroll
  for (int ob = 0; ob < 8; ++ob)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = ob >> 1, e = 4 * (ob & 1) + r;
      float v = fmaxf(acc[ob][r] + s_b1[32 * p + 8 * lq + e], 0.f);
      if (BF16) v = bf16_round(v);           // fc2 and the backward see the stashed value
      hv[p][e] = v;
    }
  const bool row_ok = row < a.N;
  if (a.h && row_ok) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if constexpr (BF16) {
        u32x4 v = {pack_bf16(hv[p][0], hv[p][1]), pack_bf16(hv[p][2], hv[p][3]), pack_bf16(hv[p][4], hv[p][5]),
                   pack_bf16(hv[p][6], hv[p][7])};
        *reinterpret_cast<u32x4*>(static_cast<unsigned short*>(a.h) + row * kHeadH + 32 * p + 8 * lq) = v;
      } else {
        float* dst = static_cast<float*>(a.h) + row * kHeadH + 32 * p + 8 * lq;
        *reinterpret_cast<float4*>(dst) = make_float4(hv[p][0], hv[p][1], hv[p][2], hv[p][3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(hv[p][4], hv[p][5], hv[p][6], hv[p][7]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < kHeadMaxOut; ++q) {
    if (q >= a.O2) break;
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int e = 0; e < 8; ++e) s = fmaf(hv[p][e], s_w2[q * kHeadH + 32 * p + 8 * lq + e], s);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (lq == 0 && row_ok) {
      const float o = s + b2r[q];
      a.out[row * a.ldo + q] = o;
      if (a.target) {               // MSE against the target, and the gradient loss.backward() would hand to this output
        const float d = o - tgt[q];          // fetched before the tile's MFMAs: a load here would queue behind the stash stores
        a.gout_w[row * a.ldgw + q] = d * (2.0f / (float)(a.N * a.O2));
        lsum = fmaf(d, d, lsum);
      }
    }
  }
}

// end of a forward workgroup: its sum of squared errors (fixed order: lanes by a shuffle tree, waves in order)
__device__ __forceinline__ void head_loss_partial(const Mlp1Args& a, float lsum, float* s_loss) {
  if (!a.target) return;            // workgroup-uniform
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) lsum += __shfl_xor(lsum, off);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) s_loss[wid] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < kFwdThreads / kWave; ++w) t += s_loss[w];
    a.loss_part[blockIdx.x] = t;
    if (blockIdx.x == 0) a.loss_part[kHeadLossSlots] = (float)gridDim.x;
  }
}

template <int G>
__global__ __launch_bounds__(kFwdThreads) void mlp1_fwd_f32_kernel(const Mlp1Args a) {
  extern __shared__ u32x4 s_raw[];
  u32x4* s_w = s_raw;
  float* s_b1 = reinterpret_cast<float*>(s_raw + 8 * G * kWave);
  float* s_w2 = s_b1 + kHeadH;
  head_fill_lds<G>(static_cast<const u32x4*>(a.image), s_raw);
  __syncthreads();
  float b2r[kHeadMaxOut];       // a global load in the epilogue would sit behind the stash stores (one in-order vmcnt)
#pragma unroll
  for (int q = 0; q < kHeadMaxOut; ++q) b2r[q] = q < a.O2 ? a.b2[q] : 0.f;
  __shared__ float s_loss[kFwdThreads / kWave];
  float lsum = 0.f;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int64_t n_tiles = ceil_div(a.N, 16);
  const int64_t n_waves = (int64_t)gridDim.x * (kFwdThreads / kWave), wave = (int64_t)blockIdx.x * (kFwdThreads / kWave) + wid;
  const int ipad = (a.I + 3) / 4 * 4;          // the row's allocated width: ldx >= ipad (host checked)
  // lane (row lr, quarter lq) loads the float4 x[row][16 g + 4 lq .. + 3] once per 16-column group and feeds its four
  // components to four consecutive k-steps (the W image is built in the same k order).  Loads are UNCONDITIONAL (row and
  // column clamped into the matrix) and whatever must be zero is zeroed where the value is consumed: a fix-up at the load
  // makes the compiler wait for the load right there (one memory round trip per load instead of one per tile), and a fix-up
  // that ends up behind the epilogue's stores waits for those too (vmcnt is one in-order counter).
  auto load_tile = [&](int64_t t, float4 (&xv)[G]) {
    const int64_t row = min(t * 16 + lr, a.N - 1);
    const float* xr = a.x + row * a.ldx;
#pragma unroll
    for (int g = 0; g < G; ++g) xv[g] = *reinterpret_cast<const float4*>(xr + min(16 * g + 4 * lq, ipad - 4));
  };
  auto fix_tile = [&](float4 (&xv)[G]) {      // pad columns may hold anything: they must not reach the MFMA
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (16 * g + 16 <= a.I) continue;        // uniform: only the boundary groups pay
      const int k0 = 16 * g + 4 * lq;
      if (k0 + 0 >= a.I) xv[g].x = 0.f;
      if (k0 + 1 >= a.I) xv[g].y = 0.f;
      if (k0 + 2 >= a.I) xv[g].z = 0.f;
      if (k0 + 3 >= a.I) xv[g].w = 0.f;
    }
  };
  float4 xc[G], xn[G];
  load_tile(wave, xc);
  fix_tile(xc);
  for (int64_t t = wave; t < n_tiles; t += n_waves) {
    load_tile(t + n_waves, xn);                // the next tile's operands are in flight while this one multiplies
    float tgt[kHeadMaxOut];                      // this tile's targets (lanes lq == 0 own a row's outputs), fetched with the prefetch
#pragma unroll
    for (int q = 0; q < kHeadMaxOut; ++q)
      tgt[q] = (a.target && lq == 0 && q < a.O2 && t * 16 + lr < a.N) ? a.target[(t * 16 + lr) * a.ldt + q] : 0.f;
    __builtin_amdgcn_sched_barrier(0);         // ... which they are only if the loads are ISSUED here: left alone, the compiler
                                               // loads xn into xc's registers after xc's last use (the copy below coalesces)
    f32x4 acc[8];
#pragma unroll
    for (int ob = 0; ob < 8; ++ob) acc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G; ++g) {
      f32x4 wf[8];
#pragma unroll
      for (int ob = 0; ob < 8; ++ob) wf[ob] = __builtin_bit_cast(f32x4, s_w[(ob * G + g) * kWave + lane]);
      const float comp[4] = {xc[g].x, xc[g].y, xc[g].z, xc[g].w};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int ob = 0; ob < 8; ++ob) acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ob][s4], comp[s4], acc[ob], 0, 0, 0);
      // keep the fragment reads of later groups behind this group's MFMAs: hoisted to the top, the 88 ds_read_b128 of a tile
      // need 352 registers and spill
      if (g % 2 == 1) __builtin_amdgcn_sched_barrier(0);
    }
    // the next tile's values are taken over BEFORE this tile's stores are issued (see load_tile), but behind the MFMAs
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < G; ++g) xc[g] = xn[g];
    fix_tile(xc);
    __builtin_amdgcn_sched_barrier(0);
    head_epilogue<false>(a, acc, s_b1, s_w2, b2r, t * 16 + lr, lq, lsum, tgt);
  }
  head_loss_partial(a, lsum, s_loss);
}

// bf16 matrix cores: G2 = 32-column groups; lane (row lr, quarter lq) loads x[row][32 g + 8 lq .. + 7] (two float4) and
// rounds the eight values to one B fragment.
template <int G2>
__global__ __launch_bounds__(kFwdThreads) void mlp1_fwd_bf16_kernel(const Mlp1Args a) {
  extern __shared__ u32x4 s_raw[];
  u32x4* s_w = s_raw;
  float* s_b1 = reinterpret_cast<float*>(s_raw + 8 * G2 * kWave);
  float* s_w2 = s_b1 + kHeadH;
  head_fill_lds<G2>(static_cast<const u32x4*>(a.image), s_raw);
  __syncthreads();
  float b2r[kHeadMaxOut];       // a global load in the epilogue would sit behind the stash stores (one in-order vmcnt)
#pragma unroll
  for (int q = 0; q < kHeadMaxOut; ++q) b2r[q] = q < a.O2 ? a.b2[q] : 0.f;
  __shared__ float s_loss[kFwdThreads / kWave];
  float lsum = 0.f;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int64_t n_tiles = ceil_div(a.N, 16);
  const int64_t n_waves = (int64_t)gridDim.x * (kFwdThreads / kWave), wave = (int64_t)blockIdx.x * (kFwdThreads / kWave) + wid;
  const int ipad = (a.I + 3) / 4 * 4;
  auto load_tile = [&](int64_t t, float4 (&xv)[2 * G2]) {      // unconditional, clamped: see mlp1_fwd_f32_kernel
    const int64_t row = min(t * 16 + lr, a.N - 1);
    const float* xr = a.x + row * a.ldx;
#pragma unroll
    for (int c = 0; c < 2 * G2; ++c) xv[c] = *reinterpret_cast<const float4*>(xr + min(32 * (c >> 1) + 8 * lq + 4 * (c & 1), ipad - 4));
  };
  auto to_frag = [&](const float4 (&xv)[2 * G2], int g) {
    float4 p = xv[2 * g], q = xv[2 * g + 1];
    const int k0 = 32 * g + 8 * lq;
    if (32 * g + 32 > a.I) {                   // uniform: the boundary group; pad columns must not reach the MFMA
      if (k0 + 0 >= a.I) p.x = 0.f;
      if (k0 + 1 >= a.I) p.y = 0.f;
      if (k0 + 2 >= a.I) p.z = 0.f;
      if (k0 + 3 >= a.I) p.w = 0.f;
      if (k0 + 4 >= a.I) q.x = 0.f;
      if (k0 + 5 >= a.I) q.y = 0.f;
      if (k0 + 6 >= a.I) q.z = 0.f;
      if (k0 + 7 >= a.I) q.w = 0.f;
    }
    const u32x4 v = {pack_bf16(p.x, p.y), pack_bf16(p.z, p.w), pack_bf16(q.x, q.y), pack_bf16(q.z, q.w)};
    return v;
  };
  float4 xr0[2 * G2];
  u32x4 xc[G2];                                // the current tile as ready B fragments (half the registers of the fp32 values)
  load_tile(wave, xr0);
#pragma unroll
  for (int g = 0; g < G2; ++g) xc[g] = to_frag(xr0, g);
  for (int64_t t = wave; t < n_tiles; t += n_waves) {
    float4 xn[2 * G2];
    load_tile(t + n_waves, xn);
    float tgt[kHeadMaxOut];                      // this tile's targets (lanes lq == 0 own a row's outputs), fetched with the prefetch
#pragma unroll
    for (int q = 0; q < kHeadMaxOut; ++q)
      tgt[q] = (a.target && lq == 0 && q < a.O2 && t * 16 + lr < a.N) ? a.target[(t * 16 + lr) * a.ldt + q] : 0.f;
    __builtin_amdgcn_sched_barrier(0);         // issue the prefetch HERE (see mlp1_fwd_f32_kernel)
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
    // take the next tile over (wait for its loads, round) BEFORE this tile's stores are issued, but behind the MFMAs
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < G2; ++g) xc[g] = to_frag(xn, g);
    __builtin_amdgcn_sched_barrier(0);
    head_epilogue<true>(a, acc, s_b1, s_w2, b2r, t * 16 + lr, lq, lsum, tgt);
  }
  head_loss_partial(a, lsum, s_loss);
}

// ------------------------------------------------------------------------------------------------ backward
// gW1[128, I + 1] = gh^T [x | 1] (column I = gb1) and gW2[O2, 128 + 1] = gout^T [h | 1] summed over the rows: K = the rows.
// The 128 x 176 fp32 result does not fit one wave's registers, so the four waves of a workgroup split the x COLUMNS: wave w
// owns the float4 chunks [cpw w, cpw (w + 1)) of a row (cpw = chunks per wave, <= 11: 44 columns) and all 128 hidden units
// (32 accumulator tiles); every wave reads the whole stash row (it is the smaller operand and comes from L1 / L2 after the
// first wave), its own slice of x, and forms gh in registers.  Tile column n of B fragment s stands for x column
// 4 (cpw w + n) + s, tile row m of A fragment t for hidden unit 8 m + t (bf16) / 64 p + 4 m + s (fp32): all global loads
// are 16 bytes per lane.  Workgroups take row slabs round-robin; per-workgroup partial sums, fixed-order second stage.
//
// Partial sums per workgroup in the LANES' OWN order -- [wave][tile][register][lane] for the 32 gW1 tiles of a wave, then
// [wave][k][register][lane] for its two gW2 tiles, then gb2[4] -- so that every store instruction of the main kernel and
// every load instruction of the second stage moves 256 contiguous bytes (an (o, column)-ordered layout is 64 scattered
// dwords per instruction: 22 MB of 4-byte writes).  The second stage decodes an element's (o, column) from its position.
constexpr int kHeadF32Tiles = 2 * 11;                 // per wave in the fp32 backward: 2 hidden fragments x at most 11 column fragments
constexpr int kHeadW1Floats = 4 * 32 * 4 * kWave, kHeadW2Floats = 4 * 2 * 4 * kWave;
constexpr int kHeadPartialFloats = kHeadW1Floats + kHeadW2Floats + kHeadMaxOut;

template <int O2T>   // 1, or 4 (covers 2..4: absent outputs carry zeros)
__global__ __launch_bounds__(kBwdThreads) void mlp1_bwd_bf16_kernel(const Mlp1Args a, int cpw) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int ipad = (a.I + 3) / 4 * 4;
  const int col0 = 4 * (cpw * wid + lr);                 // this lane's x columns col0 .. col0 + 3
  const bool xlane = lr < cpw && col0 <= a.I;            // column I is the ones column (bias gradient)
  float w2r[O2T][8];                                     // fc2 weights of this lane's hidden units 8 lr + t, rounded to bf16
#pragma unroll
  for (int q = 0; q < O2T; ++q)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int o = 8 * lr + t;
      w2r[q][t] = (q < a.O2 && o < a.H) ? bf16_round(a.w2[(int64_t)q * a.H + o]) : 0.f;
    }
  f32x4 acc[8][4], acc2[2];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[t][s] = f32x4{0.f, 0.f, 0.f, 0.f};
  acc2[0] = acc2[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  float gb2 = 0.f;

  struct Slab { float4 xv[8]; u32x4 hv[8]; float gv[8][O2T]; };
  const int64_t n_slabs = a.N / 32;                        // full slabs; the ragged tail (N % 32 rows) is taken after the loop
  const unsigned short* hb = static_cast<const unsigned short*>(a.h);
  const int colc = min(col0, ipad - 4);                  // lanes without columns of their own re-read the last chunk (dropped below)
  // this lane's x components: a column < I is the loaded value, column I is the ones column, anything else is zero
  const bool x_raw[4] = {col0 + 0 < a.I, col0 + 1 < a.I, col0 + 2 < a.I, col0 + 3 < a.I};
  const float x_fill[4] = {xlane && col0 + 0 == a.I ? 1.f : 0.f, xlane && col0 + 1 == a.I ? 1.f : 0.f,
                           xlane && col0 + 2 == a.I ? 1.f : 0.f, xlane && col0 + 3 == a.I ? 1.f : 0.f};
  // Row j of a slab: a SCALAR base (row 32 s + j) plus 32-bit lane offsets that never change; loads are unconditional and there
  // is no control flow around them (a slab index beyond the end re-reads the last full slab and is never consumed): a fix-up at
  // the load, or a branch around it, makes the compiler wait for memory on the spot (see mlp1_bwd_f32_kernel).
  const unsigned xo = (unsigned)(8 * lq * a.ldx + colc), ho = (unsigned)(8 * lq * kHeadH + 8 * lr), go = (unsigned)(8 * lq * a.ldg);
  auto issue_row = [&](Slab& d, int64_t s, int j) {
    const int64_t r0 = 32 * min(s, n_slabs - 1) + j;      // uniform
    d.xv[j] = *reinterpret_cast<const float4*>(a.x + r0 * a.ldx + xo);
    d.hv[j] = *reinterpret_cast<const u32x4*>(hb + r0 * kHeadH + ho);
#pragma unroll
    for (int q = 0; q < O2T; ++q) d.gv[j][q] = a.gout[r0 * a.ldg + go + min(q, a.O2 - 1)];
  };
  // One slab: B fragments of x and the per-row scalars first, then the eight A fragments of the gated hidden gradient, four
  // MFMAs each; `after(t)` runs behind fragment t's MFMAs (the main loop issues row t of a later slab there).
  auto consume = [&](Slab& cur, int64_t first_row, auto&& after) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = first_row + 8 * lq + j < a.N;
#pragma unroll
      for (int q = 0; q < O2T; ++q) cur.gv[j][q] = (ok && q < a.O2) ? cur.gv[j][q] : 0.f;
      cur.xv[j].x = x_raw[0] ? cur.xv[j].x : x_fill[0];
      cur.xv[j].y = x_raw[1] ? cur.xv[j].y : x_fill[1];
      cur.xv[j].z = x_raw[2] ? cur.xv[j].z : x_fill[2];
      cur.xv[j].w = x_raw[3] ? cur.xv[j].w : x_fill[3];
    }
    // B fragments: x, four tile-column sets (s4): element j = row 8 lq + j
    bf16x8 bx[4];
    {
      u32x4 v0, v1, v2, v3;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        v0[d] = pack_bf16(cur.xv[2 * d].x, cur.xv[2 * d + 1].x);
        v1[d] = pack_bf16(cur.xv[2 * d].y, cur.xv[2 * d + 1].y);
        v2[d] = pack_bf16(cur.xv[2 * d].z, cur.xv[2 * d + 1].z);
        v3[d] = pack_bf16(cur.xv[2 * d].w, cur.xv[2 * d + 1].w);
      }
      bx[0] = __builtin_bit_cast(bf16x8, v0); bx[1] = __builtin_bit_cast(bf16x8, v1);
      bx[2] = __builtin_bit_cast(bf16x8, v2); bx[3] = __builtin_bit_cast(bf16x8, v3);
    }
    // per-row scalars: bf16-rounded gout (the data-gradient GEMM's operand)
    float gr[8][O2T];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int q = 0; q < O2T; ++q) gr[j][q] = bf16_round(cur.gv[j][q]);
    // gW2 / gb2: this wave's two A fragments of the un-gated stash (t = 2 wid, 2 wid + 1) against gout as B ([row][q = lr])
    {
      float gq[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < O2T; ++q) v = lr == q ? cur.gv[j][q] : v;
        gq[j] = v;
        if (wid == 0) gb2 += v;
      }
      const u32x4 gb = {pack_bf16(gq[0], gq[1]), pack_bf16(gq[2], gq[3]), pack_bf16(gq[4], gq[5]), pack_bf16(gq[6], gq[7])};
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int t = 2 * wid + k, c = t >> 1;
        u32x4 av;
#pragma unroll
        for (int d = 0; d < 4; ++d)
          av[d] = (t & 1) ? __builtin_amdgcn_perm(cur.hv[2 * d + 1][c], cur.hv[2 * d][c], 0x07060302u)
                          : __builtin_amdgcn_perm(cur.hv[2 * d + 1][c], cur.hv[2 * d][c], 0x05040100u);
        acc2[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, gb), acc2[k], 0, 0, 0);
      }
    }
    // gW1 | gb1: A fragment t = the gated hidden gradient of units 8 lr + t, element j = row 8 lq + j
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int c = t >> 1;
      u32x4 av;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        float g2[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int j = 2 * d + k;
          const unsigned hw = cur.hv[j][c];
          const float hval = (t & 1) ? bf16_hi(hw) : bf16_lo(hw);
          float gs = 0.f;
#pragma unroll
          for (int q = 0; q < O2T; ++q) gs = fmaf(gr[j][q], w2r[q][t], gs);
          g2[k] = hval > 0.f ? gs : 0.f;
        }
        av[d] = pack_bf16(g2[0], g2[1]);
      }
      const bf16x8 af = __builtin_bit_cast(bf16x8, av);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) acc[t][s4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bx[s4], acc[t][s4], 0, 0, 0);
      after(t);
    }
  };
  // Two register sets: the next slab is requested in one block in front of this slab's work and taken over behind it; the two
  // scheduling barriers pin that order (see mlp1_bwd_f32_kernel).
  Slab cur, nxt;
  const int64_t G = gridDim.x;
  if (n_slabs > 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) issue_row(cur, blockIdx.x, j);
    for (int64_t sl = blockIdx.x; sl < n_slabs; sl += G) {
#pragma unroll
      for (int j = 0; j < 8; ++j) issue_row(nxt, sl + G, j);
      __builtin_amdgcn_sched_barrier(0);
      consume(cur, 32 * sl, [](int) {});
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
    }
  }
  if (blockIdx.x == 0 && (a.N & 31)) {                     // the ragged tail: rows 32 n_slabs .. N - 1, lane rows clamped
    Slab t;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t row = min(32 * n_slabs + 8 * lq + j, a.N - 1);
      t.xv[j] = *reinterpret_cast<const float4*>(a.x + row * a.ldx + colc);
      t.hv[j] = *reinterpret_cast<const u32x4*>(hb + row * kHeadH + 8 * lr);
#pragma unroll
      for (int q = 0; q < O2T; ++q) t.gv[j][q] = a.gout[row * a.ldg + min(q, a.O2 - 1)];
    }
    consume(t, 32 * n_slabs, [](int) {});
  }
  // partial sums of this workgroup (lane order): register r of tile (t, s4) is gW1[8 (4 lq + r) + t][4 (cpw wid + lr) + s4]
  float* __restrict__ dst = a.partial + (int64_t)blockIdx.x * kHeadPartialFloats + (wid * 32 * 4) * kWave + lane;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[((t * 4 + s4) * 4 + r) * kWave] = acc[t][s4][r];
  float* __restrict__ dst2 = a.partial + (int64_t)blockIdx.x * kHeadPartialFloats + kHeadW1Floats;
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int r = 0; r < 4; ++r) dst2[((wid * 2 + k) * 4 + r) * kWave + lane] = acc2[k][r];   // gW2[q = lr][8 (4 lq + r) + 2 wid + k]
  if (wid == 0) {
    gb2 += __shfl_xor(gb2, 16);
    gb2 += __shfl_xor(gb2, 32);
    if (lq == 0 && lr < kHeadMaxOut) dst2[kHeadW2Floats + lr] = gb2;
  }
}

// fp32: K-steps of 4 rows (row = step base + lq), KU steps per iteration with the next iteration's operands in flight.
// Here the waves split the HIDDEN UNITS (wave w owns units 32 w .. 32 w + 31: two A fragments, float2 loads of the stash,
// fragment s: tile row m <-> unit 32 w + 2 m + s) and every wave multiplies them with ALL columns of [x | 1]: NG groups of 64
// columns as float4 loads (fragment (g, s4): tile column n <-> column 64 g + 4 n + s4) and NS fragments of 16 columns as
// scalar loads (fragment f: tile column n <-> column 64 NG + 16 f + n).  170 inputs: 2 x (8 + 3) + 2 = 24 MFMAs per k-step and
// wave against 34 with the columns split four ways (44 columns fill 11 of a float4 fragment set's 16 lanes) -- this kernel is
// bound by the fp32 matrix cores, so fewer MFMAs is the lever; x comes from L1 / L2 for three of the four waves.
// Partial sums (lane order): [wave][tile s (4 NG + NS) + b][register][lane], then [wave][s][register][lane] for gW2, then gb2.
template <int O2T, int KU, int NG, int NS>
__global__ __launch_bounds__(kBwdThreads) void mlp1_bwd_f32_kernel(const Mlp1Args a) {
  constexpr int NB = 4 * NG + NS;                            // B fragments
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int ipad = (a.I + 3) / 4 * 4;
  float w2r[O2T][2];
#pragma unroll
  for (int q = 0; q < O2T; ++q)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int o = 32 * wid + 2 * lr + s;
      w2r[q][s] = (q < a.O2 && o < a.H) ? a.w2[(int64_t)q * a.H + o] : 0.f;
    }
  f32x4 acc[2][NB], acc2[2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[s][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  acc2[0] = acc2[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  float gb2 = 0.f;

  struct Step { float4 xg[NG]; float xs[NS > 0 ? NS : 1]; float2 hv; float gv[O2T]; };
  const float* hf = static_cast<const float*>(a.h);
  const int64_t n_steps = a.N / 4;                          // full k-steps; the ragged tail (N % 4 rows) is taken after the loop
  const int64_t n_iters = ceil_div(n_steps, (int64_t)KU);
  // this lane's columns: a column < I is the loaded value, column I is the ones column (bias gradient), anything else is zero;
  // loads are clamped into the row (pads may hold anything, a clamped column is overridden below)
  int gcol[NG], scol[NS > 0 ? NS : 1];
  bool g_raw[NG][4], s_raw[NS > 0 ? NS : 1];
  float g_fill[NG][4], s_fill[NS > 0 ? NS : 1];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int c0 = 64 * g + 4 * lr;
    gcol[g] = min(c0, ipad - 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) { g_raw[g][c] = c0 + c < a.I; g_fill[g][c] = c0 + c == a.I ? 1.f : 0.f; }
  }
#pragma unroll
  for (int f = 0; f < NS; ++f) {
    const int c = 64 * NG + 16 * f + lr;
    scol[f] = min(c, ipad - 1);
    s_raw[f] = c < a.I; s_fill[f] = c == a.I ? 1.f : 0.f;
  }
  // Loads: a SCALAR base per k-step plus 32-bit lane offsets that never change, unconditional, no control flow (a step
  // beyond the end re-reads the last full step; its gout is taken as zero where it is consumed).  A fix-up at the load or a
  // branch around it makes the compiler wait for memory on the spot (vmcnt is one in-order counter, and at a join the
  // compiler no longer knows how many loads are in flight).
  unsigned xo[NG], so[NS > 0 ? NS : 1];
#pragma unroll
  for (int g = 0; g < NG; ++g) xo[g] = (unsigned)(lq * a.ldx + gcol[g]);
#pragma unroll
  for (int f = 0; f < NS; ++f) so[f] = (unsigned)(lq * a.ldx + scol[f]);
  const unsigned ho = (unsigned)(lq * kHeadH + 32 * wid + 2 * lr), go = (unsigned)(lq * a.ldg);
  auto issue = [&](Step& d, int64_t it, int u) {
    const int64_t r0 = 4 * min(it * KU + u, n_steps - 1);   // uniform
    const float* xb = a.x + r0 * a.ldx; const float* hb = hf + r0 * kHeadH; const float* gb = a.gout + r0 * a.ldg;
#pragma unroll
    for (int g = 0; g < NG; ++g) d.xg[g] = *reinterpret_cast<const float4*>(xb + xo[g]);
#pragma unroll
    for (int f = 0; f < NS; ++f) d.xs[f] = xb[so[f]];
    d.hv = *reinterpret_cast<const float2*>(hb + ho);
#pragma unroll
    for (int q = 0; q < O2T; ++q) d.gv[q] = gb[go + min(q, a.O2 - 1)];
  };
  auto consume = [&](const Step& c, bool ok) {
    float gv[O2T];
#pragma unroll
    for (int q = 0; q < O2T; ++q) gv[q] = (ok && q < a.O2) ? c.gv[q] : 0.f;
    float xb[NB];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      xb[4 * g + 0] = g_raw[g][0] ? c.xg[g].x : g_fill[g][0];
      xb[4 * g + 1] = g_raw[g][1] ? c.xg[g].y : g_fill[g][1];
      xb[4 * g + 2] = g_raw[g][2] ? c.xg[g].z : g_fill[g][2];
      xb[4 * g + 3] = g_raw[g][3] ? c.xg[g].w : g_fill[g][3];
    }
#pragma unroll
    for (int f = 0; f < NS; ++f) xb[4 * NG + f] = s_raw[f] ? c.xs[f] : s_fill[f];
    const float hv[2] = {c.hv.x, c.hv.y};
    // gW2 / gb2: the un-gated stash fragments against gout ([row = lq][q = lr])
    float gq = 0.f;
#pragma unroll
    for (int q = 0; q < O2T; ++q) gq = lr == q ? gv[q] : gq;
    if (wid == 0) gb2 += gq;
#pragma unroll
    for (int s = 0; s < 2; ++s) acc2[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[s], gq, acc2[s], 0, 0, 0);
    float gh[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float gs = 0.f;
#pragma unroll
      for (int q = 0; q < O2T; ++q) gs = fmaf(gv[q], w2r[q][s], gs);
      gh[s] = hv[s] > 0.f ? gs : 0.f;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int s = 0; s < 2; ++s) acc[s][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(gh[s], xb[b], acc[s][b], 0, 0, 0);
  };
  // Two register sets: the whole next iteration is requested in one block in front of this iteration's MFMAs and taken over
  // behind them.  The two scheduling barriers pin that order: left alone, the compiler folds the second set into the first
  // (the copy coalesces) and loads every value just before its use.  Tried and measured slower or no faster (the loads and
  // their address arithmetic interleaved with the MFMAs instead of a block in front of them): two sets with the loop
  // unrolled twice -- at the loop head the compiler's wait-count analysis gives up and waits for every load in flight,
  // vmcnt(0), one k-step after the last one was issued (254 vs 228 us on the column-split form); one set used as a ring --
  // the register allocator rotates the ring through AGPRs with copies that wait for the newest loads.
  Step cur[KU], nxt[KU];
  if (n_steps > 0) {
#pragma unroll
    for (int u = 0; u < KU; ++u) issue(cur[u], blockIdx.x, u);
    for (int64_t it = blockIdx.x; it < n_iters; it += gridDim.x) {
#pragma unroll
      for (int u = 0; u < KU; ++u) issue(nxt[u], it + gridDim.x, u);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < KU; ++u) consume(cur[u], it * KU + u < n_steps);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < KU; ++u) cur[u] = nxt[u];
    }
  }
  if (blockIdx.x == 0 && (a.N & 3)) {                       // the ragged tail: rows 4 n_steps .. N - 1, lane rows clamped
    const int64_t row = min(4 * n_steps + lq, a.N - 1);
    Step t;
#pragma unroll
    for (int g = 0; g < NG; ++g) t.xg[g] = *reinterpret_cast<const float4*>(a.x + row * a.ldx + gcol[g]);
#pragma unroll
    for (int f = 0; f < NS; ++f) t.xs[f] = a.x[row * a.ldx + scol[f]];
    t.hv = *reinterpret_cast<const float2*>(hf + row * kHeadH + 32 * wid + 2 * lr);
#pragma unroll
    for (int q = 0; q < O2T; ++q) t.gv[q] = a.gout[row * a.ldg + min(q, a.O2 - 1)];
    consume(t, 4 * n_steps + lq < a.N);
  }
  // partial sums (lane order): register r of tile (s, b) is gW1[32 wid + 2 (4 lq + r) + s][column of fragment b, lane lr]
  float* __restrict__ dst = a.partial + (int64_t)blockIdx.x * kHeadPartialFloats + (wid * kHeadF32Tiles * 4) * kWave + lane;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[((s * NB + b) * 4 + r) * kWave] = acc[s][b][r];
  float* __restrict__ dst2 = a.partial + (int64_t)blockIdx.x * kHeadPartialFloats + kHeadW1Floats;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) dst2[((wid * 2 + s) * 4 + r) * kWave + lane] = acc2[s][r];   // gW2[q = lr][32 wid + 2 (4 lq + r) + s]
  if (wid == 0) {
    gb2 += __shfl_xor(gb2, 16);
    gb2 += __shfl_xor(gb2, 32);
    if (lq == 0 && lr < kHeadMaxOut) dst2[kHeadW2Floats + lr] = gb2;
  }
}

// Second stage: element e of the lane-ordered layout summed over the G workgroups in a fixed order (16 slices of the G range
// per element, slice sums added in slice order), decoded to its (o, column) and filed into gW1 [H,I] / gb1 [H] / gW2 [O2,H] /
// gb2 [O2].  64 consecutive elements per workgroup: every load instruction reads 256 contiguous bytes of one partial.  The
// pass reads G x 90 KB (46 MB at 512 workgroups): with 4 slices every thread walked 128 partials one dependent add after the
// other and the launch took 22 us; 16 slices keep four times as many loads in flight.
constexpr int kHeadReduceSlices = 16;
__global__ __launch_bounds__(kHeadReduceSlices * kWave) void mlp1_bwd_reduce_kernel(const float* __restrict__ partial, int G, int I, int H, int O2,
                                                              int cpw, int bf16, float* __restrict__ gw1, float* __restrict__ gb1,
                                                              float* __restrict__ gw2, float* __restrict__ gb2,
                                                              const float* __restrict__ loss_part, float loss_scale,
                                                              float* __restrict__ loss_out) {
  __shared__ float s[kHeadReduceSlices][kWave];
  if (loss_out && blockIdx.x == gridDim.x - 1 && threadIdx.x < kWave) {
    // the mean squared error the forward left as per-workgroup sums: added in index order by one wave (a block whose elements
    // mostly stand for nothing has the time)
    const int n = min((int)loss_part[kHeadLossSlots], kHeadLossSlots);
    float t = 0.f;
    for (int b = threadIdx.x; b < n; b += kWave) t += loss_part[b];
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) t += __shfl_xor(t, off);
    if (threadIdx.x == 0) *loss_out = t * loss_scale;
  }
  const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int e = blockIdx.x * kWave + el;
  // where the element goes (-1: a lane / row / column that stands for nothing)
  int kind = -1, o = 0, c = 0;
  if (e < kHeadW1Floats) {
    const int lane = e & 63, r = (e >> 6) & 3, lr = lane & 15, lq = lane >> 4;
    if (bf16) {                                   // columns split over the waves: [wave][tile t 4 + s4][r][lane]
      const int tile = (e >> 8) & 31, wid = e >> 13, s4 = tile & 3, t = tile >> 2;
      o = 8 * (4 * lq + r) + t;
      c = 4 * (cpw * wid + lr) + s4;
      if (lr < cpw && c <= I && o < H) kind = c < I ? 0 : 1;
    } else {                                      // hidden units split over the waves: [wave][tile s NB + b][r][lane]
      const int ng = cpw >> 8, nb = 4 * ng + (cpw & 255);           // cpw carries (NG << 8) | NS for this layout
      const int tile = (e >> 8) % kHeadF32Tiles, wid = (e >> 8) / kHeadF32Tiles;
      if (wid < 4 && tile < 2 * nb) {
        const int sfr = tile / nb, b = tile % nb;
        o = 32 * wid + 2 * (4 * lq + r) + sfr;
        c = b < 4 * ng ? 64 * (b >> 2) + 4 * lr + (b & 3) : 64 * ng + 16 * (b - 4 * ng) + lr;
        if (c <= I && o < H) kind = c < I ? 0 : 1;
      }
    }
  } else if (e < kHeadW1Floats + kHeadW2Floats) {
    const int f = e - kHeadW1Floats;
    const int lane = f & 63, r = (f >> 6) & 3, k = (f >> 8) & 1, wid = f >> 9;
    const int lr = lane & 15, lq = lane >> 4;
    o = bf16 ? 8 * (4 * lq + r) + 2 * wid + k : 32 * wid + 2 * (4 * lq + r) + k;
    c = lr;
    if (c < O2 && o < H) kind = 2;
  } else if (e < kHeadPartialFloats) {
    c = e - kHeadW1Floats - kHeadW2Floats;
    if (c < O2) kind = 3;
  }
  float t = 0.f;
  if (kind >= 0) {
    const int per = (G + kHeadReduceSlices - 1) / kHeadReduceSlices;
    const int g1 = min(G, (sl + 1) * per);
    int g = sl * per;
    float t1 = 0.f, t2 = 0.f, t3 = 0.f;
    for (; g + 3 < g1; g += 4) {              // four independent chains, joined in a fixed order below
      t += partial[(int64_t)g * kHeadPartialFloats + e];
      t1 += partial[(int64_t)(g + 1) * kHeadPartialFloats + e];
      t2 += partial[(int64_t)(g + 2) * kHeadPartialFloats + e];
      t3 += partial[(int64_t)(g + 3) * kHeadPartialFloats + e];
    }
    for (; g < g1; ++g) t += partial[(int64_t)g * kHeadPartialFloats + e];
    t = (t + t1) + (t2 + t3);
  }
  s[sl][el] = t;
  __syncthreads();
  if (sl != 0 || kind < 0) return;
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < kHeadReduceSlices; ++k) tot += s[k][el];
  if (kind == 0) gw1[(int64_t)o * I + c] = tot;
  else if (kind == 1) gb1[o] = tot;
  else if (kind == 2) gw2[(int64_t)c * H + o] = tot;
  else gb2[c] = tot;
}

template <typename K>
static int head_resident_workgroups(K kernel, int threads, size_t lds) {
  int per_cu = 0, dev = 0, cus = 256;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds) != hipSuccess || per_cu < 1) per_cu = 1;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  return per_cu * cus;
}

constexpr int kHeadMaxBwdBlocks = 512;    // partial-sum slots of the backward workspace

template <bool BF16, int G>
static int launch_head_fwd(Mlp1Args a, void* workspace, hipStream_t s) {
  const size_t lds = (size_t)head_image_u32x4(G) * sizeof(u32x4);
  void (*kernel)(Mlp1Args);
  if constexpr (BF16) kernel = mlp1_fwd_bf16_kernel<G>; else kernel = mlp1_fwd_f32_kernel<G>;
  if (!ensure_dynamic_lds(kernel, lds)) return MLQEM_ERR_LAUNCH;      // per device (common.hpp)
  static const int res = head_resident_workgroups(kernel, kFwdThreads, lds);
  a.image = workspace;
  hipLaunchKernelGGL(mlp1_image_kernel<BF16>, dim3((unsigned)ceil_div(head_image_u32x4(G) * 4, 256)), dim3(256), 0, s, a, G,
                     static_cast<u32x4*>(workspace));
  const int64_t tiles = ceil_div(a.N, 16);
  // a workgroup of the loss form files its sum in loss_part[blockIdx.x] (kHeadLossSlots of them, then the count): the grid may
  // never outgrow the slots, whatever occupancy x CU count says on another compiler or part
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(res, kHeadLossSlots), ceil_div(tiles, kFwdThreads / kWave)));
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(kFwdThreads), lds, s, a);
  return launch_status();
}

}  // namespace mlqem

using namespace mlqem;

static bool head_shapes_ok(int64_t N, int I, int H, int O2) {
  return N >= 0 && I >= 1 && I <= MLQEM_MLP1_MAX_IN && H >= 1 && H <= kHeadH && O2 >= 1 && O2 <= kHeadMaxOut;
}

// the forward's image and the backward's partial sums share the front of the workspace (never alive together); the loss sums
// of a forward-with-target live behind them, from the forward launch to the backward's second stage
static size_t head_loss_offset() {
  const size_t bwd = (size_t)kHeadMaxBwdBlocks * (size_t)kHeadPartialFloats * sizeof(float);
  const size_t fwd = (size_t)head_image_u32x4(11) * sizeof(u32x4);      // the widest forward image
  return ((bwd > fwd ? bwd : fwd) + 255) / 256 * 256;
}

extern "C" size_t mlqem_mlp1_workspace_bytes(int I, int O2) {
  if (I < 1 || O2 < 1 || O2 > kHeadMaxOut) return 0;
  return head_loss_offset() + (size_t)(kHeadLossSlots + 1) * sizeof(float);     // ... | the forward's loss sums (+ their count)
}

extern "C" int mlqem_mlp1_forward(const float* x, int64_t ldx, const float* w1, const float* b1, const float* w2,
                                  const float* b2, void* h_stash, float* out, int64_t ldo, int64_t N, int I, int H, int O2,
                                  int bf16, const float* target, int64_t ldt, float* gout, int64_t ldg, void* workspace,
                                  size_t workspace_bytes, mlqem_stream_t stream) {
  begin_launches();
  if (!head_shapes_ok(N, I, H, O2)) return (I > MLQEM_MLP1_MAX_IN || H > kHeadH || O2 > kHeadMaxOut) ? MLQEM_ERR_UNSUPPORTED : MLQEM_ERR_BAD_ARG;
  if (ldx < (I + 3) / 4 * 4 || ldx % 4 || ldo < O2) return MLQEM_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < mlqem_mlp1_workspace_bytes(I, O2) || !aligned_to(workspace, 16)) return MLQEM_ERR_WORKSPACE;
  if (N == 0) return MLQEM_OK;
  if (!x || !w1 || !b1 || !w2 || !b2 || !out || !aligned_to(x, 16) || (h_stash && !aligned_to(h_stash, 16))) return MLQEM_ERR_BAD_ARG;
  if (target && (!gout || ldt < O2 || ldg < O2)) return MLQEM_ERR_BAD_ARG;
  Mlp1Args a{x, ldx, N, I, H, O2, w1, b1, w2, b2, h_stash, out, ldo, nullptr, 0, nullptr, nullptr,
             target, ldt, gout, ldg, reinterpret_cast<float*>(static_cast<char*>(workspace) + head_loss_offset())};
  hipStream_t s = as_stream(stream);
  if (bf16) {
    const int g2 = (I + 31) / 32;
    if (g2 <= 2) return launch_head_fwd<true, 2>(a, workspace, s);
    if (g2 <= 4) return launch_head_fwd<true, 4>(a, workspace, s);
    return launch_head_fwd<true, 6>(a, workspace, s);
  }
  const int g = (I + 15) / 16;
  if (g <= 4) return launch_head_fwd<false, 4>(a, workspace, s);
  if (g <= 8) return launch_head_fwd<false, 8>(a, workspace, s);
  return launch_head_fwd<false, 11>(a, workspace, s);
}

extern "C" int mlqem_mlp1_backward(const float* gout, int64_t ldg, const float* x, int64_t ldx, const void* h_stash,
                                   const float* w2, float* gw1, float* gb1, float* gw2, float* gb2, int64_t N, int I, int H,
                                   int O2, int bf16, float* loss_out, void* workspace, size_t workspace_bytes,
                                   mlqem_stream_t stream) {
  begin_launches();
  if (!head_shapes_ok(N, I, H, O2)) return (I > MLQEM_MLP1_MAX_IN || H > kHeadH || O2 > kHeadMaxOut) ? MLQEM_ERR_UNSUPPORTED : MLQEM_ERR_BAD_ARG;
  if (ldx < (I + 3) / 4 * 4 || ldx % 4 || ldg < O2 || !gw1 || !gb1 || !gw2 || !gb2) return MLQEM_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < mlqem_mlp1_workspace_bytes(I, O2)) return MLQEM_ERR_WORKSPACE;
  if (N > 0 && (!gout || !x || !h_stash || !w2 || !aligned_to(x, 16) || !aligned_to(h_stash, 16))) return MLQEM_ERR_BAD_ARG;
  if (loss_out && N == 0) return MLQEM_ERR_BAD_ARG;       // there is no loss of nothing (and no forward that left its sums)
  Mlp1Args a{x, ldx, N, I, H, O2, nullptr, nullptr, w2, nullptr, const_cast<void*>(h_stash), nullptr, 0, gout, ldg,
             static_cast<float*>(workspace), nullptr, nullptr, 0, nullptr, 0, nullptr};
  hipStream_t s = as_stream(stream);
  const int chunks = (I + 1 + 3) / 4;             // float4 chunks of [x | 1]
  const int cpw = (chunks + 3) / 4;               // per wave: <= 11 for I <= 175
  int G, reduce_layout = cpw;
  if (N == 0) {
    G = 0;                                        // nothing to sum: the second stage writes zeros
  } else if (bf16) {
    static const int res1 = head_resident_workgroups(mlp1_bwd_bf16_kernel<1>, kBwdThreads, 0);
    static const int res4 = head_resident_workgroups(mlp1_bwd_bf16_kernel<4>, kBwdThreads, 0);
    G = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(O2 == 1 ? res1 : res4, kHeadMaxBwdBlocks), std::max<int64_t>(N / 32, 1)));
    if (O2 == 1) hipLaunchKernelGGL(mlp1_bwd_bf16_kernel<1>, dim3(G), dim3(kBwdThreads), 0, s, a, cpw);
    else hipLaunchKernelGGL(mlp1_bwd_bf16_kernel<4>, dim3(G), dim3(kBwdThreads), 0, s, a, cpw);
  } else {
    constexpr int KU = MLQEM_HEAD_BWD_KU;
    // columns of [x | 1] as NG float4 groups of 64 and NS scalar fragments of 16
    const int ci = I + 1;
    const int ng = ci <= 112 ? 1 : 2, ns = ci <= 64 * ng ? 0 : 3;
    reduce_layout = (ng << 8) | ns;
    // every instantiation decays to void(*)(Mlp1Args), so a `static` inside the lambda would be ONE value shared by all
    // eight: the resident-workgroup counts are kept per (O2, ng, ns) instead
    static int res_of[2][2][2] = {};
    auto launch = [&](auto kernel) {
      int& res = res_of[O2 == 1 ? 0 : 1][ng - 1][ns ? 1 : 0];
      if (res == 0) res = head_resident_workgroups(kernel, kBwdThreads, 0);
      G = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(res, kHeadMaxBwdBlocks), ceil_div(std::max<int64_t>(N / 4, 1), KU)));
      hipLaunchKernelGGL(kernel, dim3(G), dim3(kBwdThreads), 0, s, a);
    };
    if (O2 == 1) {
      if (ng == 1 && ns == 0) launch(mlp1_bwd_f32_kernel<1, KU, 1, 0>);
      else if (ng == 1) launch(mlp1_bwd_f32_kernel<1, KU, 1, 3>);
      else if (ns == 0) launch(mlp1_bwd_f32_kernel<1, KU, 2, 0>);
      else launch(mlp1_bwd_f32_kernel<1, KU, 2, 3>);
    } else {
      if (ng == 1 && ns == 0) launch(mlp1_bwd_f32_kernel<4, KU, 1, 0>);
      else if (ng == 1) launch(mlp1_bwd_f32_kernel<4, KU, 1, 3>);
      else if (ns == 0) launch(mlp1_bwd_f32_kernel<4, KU, 2, 0>);
      else launch(mlp1_bwd_f32_kernel<4, KU, 2, 3>);
    }
  }
  hipLaunchKernelGGL(mlp1_bwd_reduce_kernel, dim3((unsigned)ceil_div(kHeadPartialFloats, kWave)), dim3(kHeadReduceSlices * kWave), 0, s, a.partial, G, I, H, O2,
                     reduce_layout, bf16 ? 1 : 0, gw1, gb1, gw2, gb2,
                     reinterpret_cast<const float*>(static_cast<const char*>(workspace) + head_loss_offset()),
                     N > 0 ? 1.0f / ((float)N * (float)O2) : 0.f, loss_out);
  return launch_status();
}
