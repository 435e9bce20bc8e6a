// ASAPooling's cluster sums (docs/tutorials/gnn.py:85,92,104-112) over the dense blocks of a structure
// (dense_block.hpp): the same plans as TransformerConv's edge softmax, the same wave layout -- a workgroup per block, its four waves
// sharing the column blocks, cell (u = 16 cb + 4 g + i, row r) in register i of lane l = 16 g + r.
//
//   x'[i] = sum_j softmax_j(LeakyReLU(a_i + c_j)) x[j]   over the in-entries of i AND i itself (add_remaining_self_loops)
//
// The score of a cell is one add of two per-node scalars; the weighted sum of the source rows is x'^T = X_U^T W on the f32 matrix
// cores, one accumulator tile per 16 channels (rows of at most 32 channels: the reference's 30).  Rows outside the blocks stay with the
// per-edge kernels (attn.hip, family_b_bwd.hip), which skip the rows a plan flags.
#include <utility>

#include "dense_block.hpp"

namespace mlqem {

constexpr int kPoolTiles = 2;                               // channel tiles of 16: D <= 32

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

// the cell bits of a column block for ASAPooling: the structure's entries and, always, the row itself
__device__ __forceinline__ uint32_t pool_bits(const uint32_t* maskrow, int cb, int g, int selfs) {
  uint32_t nib = cell_bits(maskrow, cb, g);
  const int d = selfs - (16 * cb + 4 * g);
  if ((unsigned)d < 4u) nib |= 1u << d;
  return nib;
}

// --------------------------------------------------------------------------------------------------------------- forward
// out[row, :] = x'[row]; stat[row] = {m, 1 / (den + 1e-16)} (the maximum in natural units) for the backward kernels here.
__global__ __launch_bounds__(kBlock) void dense_softmax_aggregate_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ a_dst,
                                                                         const float* __restrict__ c_src, float slope, int D,
                                                                         float* __restrict__ out, int64_t ldo, float2* __restrict__ stat,
                                                                         const DensePlan p) {
  __shared__ __attribute__((aligned(16))) int lds[kDbLdsInts];
  __shared__ __attribute__((aligned(16))) f32x4 red_o[kPoolTiles][kDbWaves * kWave];
  __shared__ float red_m[kDbWaves][kDbRows], red_d[kDbWaves][kDbRows];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int nblocks = *p.counter / kDbRows;
  const bool has1 = 16 + r < D;                            // the lane's channel of the second tile exists
  for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {
    const int32_t* __restrict__ rec = p.records + (int64_t)b * kDbStride;
    const int4 hdr = *reinterpret_cast<const int4*>(rec);
    if (!hdr.z) continue;
    const int nrows = hdr.x, ncb = (hdr.y + 15) >> 4;
    const bool valid = r < nrows;
    const int row = valid ? rec[kDbRowsOff + r] : rec[kDbRowsOff];
    const int selfs = valid ? rec[kDbSelfOff + r] : -1;
    const float ai = a_dst[row];
    const BlockLds l = block_stage(rec, lds);
    const uint32_t* maskrow = l.mask + r * kDbMaskWords;
    float m2 = kNoMax, den = 0.f;
    f32x4 o[kPoolTiles];
#pragma unroll
    for (int t = 0; t < kPoolTiles; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int cb = wave; cb < ncb; cb += kDbWaves) {
      const int4 id4 = *reinterpret_cast<const int4*>(l.uni + 16 * cb + 4 * g);
      const int ids[4] = {id4.x, id4.y, id4.z, id4.w};
      float c4[4], xv[kPoolTiles][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        c4[i] = c_src[ids[i]];
        const float* __restrict__ xr = x + (int64_t)ids[i] * ldx + r;
        xv[0][i] = xr[0];
        xv[1][i] = has1 ? xr[16] : 0.f;
      }
      const uint32_t nib = pool_bits(maskrow, cb, g, selfs);
      float s[4], bm = kNoMax;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s[i] = leaky(ai + c4[i], slope) * kLog2e;
        if (nib >> i & 1u) bm = fmaxf(bm, s[i]);
      }
      bm = rows_max(bm);
      const float mn = fmaxf(m2, bm);
      const float rs = __builtin_amdgcn_exp2f(m2 - mn);
      m2 = mn;
      den *= rs;
      float w[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        w[i] = (nib >> i & 1u) ? __builtin_amdgcn_exp2f(s[i] - mn) : 0.f;
        den += w[i];
      }
#pragma unroll
      for (int t = 0; t < kPoolTiles; ++t) {
        o[t] *= rs;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[t] = mfma4(xv[t][i], w[i], o[t]);
      }
    }
    if (g == 0) red_m[wave][r] = m2;
    __syncthreads();
    float mt = red_m[0][r];
#pragma unroll
    for (int w = 1; w < kDbWaves; ++w) mt = fmaxf(mt, red_m[w][r]);
    const float rs = __builtin_amdgcn_exp2f(m2 - mt);
    const float dw = rows_sum(den) * rs;
    if (g == 0) red_d[wave][r] = dw;
#pragma unroll
    for (int t = 0; t < kPoolTiles; ++t) red_o[t][wave * kWave + lane] = o[t] * rs;
    __syncthreads();
    if (wave < kPoolTiles && valid) {                      // wave t finishes channel tile t: channels 16 t + 4 g .. + 3 of the row
      const int t = wave, c0 = 16 * t + 4 * g;
      float d = 1e-16f;
#pragma unroll
      for (int w = 0; w < kDbWaves; ++w) d += red_d[w][r];
      const float inv = 1.0f / d;
      const f32x4 acc = waves_sum(red_o[t], lane) * inv;
      float* __restrict__ orow = out + (int64_t)row * ldo + c0;
      if (c0 + 4 <= ldo) *reinterpret_cast<f4u*>(orow) = f4u{acc[0], acc[1], acc[2], acc[3]};      // pads of a padded row: zeros (x's pads are)
      else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (c0 + i < D) orow[i] = acc[i];
      }
      if (t == 0 && g == 0 && stat) stat[row] = make_float2(mt * kLn2, inv);
    }
  }
}

// ------------------------------------------------------------------------------------------- segment max, its ties, its backward
// Per-channel work over a block's cells: no matrix product.  Lane (r, g) owns channels 8 g .. 8 g + 7 of block row r (rows of exactly
// 32 floats: 29-32 channels) and, as a stager, brings the same channels of union row r of the column block (two 16-byte loads); every
// lane then walks the 16 union rows against its row's cell bits, each row's values handed over by a DPP broadcast inside the 16-lane
// row of its g (scan_step).
//   kMax   out[row, c]  = max over the row's cells of A[u, c]                                  (ASAPooling's query: the segment max of x)
//   kTies  out[row, c]  = number of the row's cells with A[u, c] == own[row, c]                (A = x, own = xmax: the ties of that max)
//   kShare out[row, c] += sum over the row's cells with own[row, c] == A[u, c] of B[u, c]      (the OUT structure: own = x, A = xmax of the
//                                                                                              destinations, B = their shares of the gradient)
enum PoolScan { kMax, kTies, kShare };
typedef float f8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void store8(float* __restrict__ p, const float* v, int c0, int D) {      // channels c0 .. c0 + 7, those below D
  if (c0 + 8 <= D) {
    *reinterpret_cast<f4a*>(p) = f4a{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f4a*>(p + 4) = f4a{v[4], v[5], v[6], v[7]};
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (c0 + k < D) p[k] = v[k];
  }
}

// One union row of the column block against the lane's row: the stager lanes of the same 16-lane row (same g) hold that union row's
// eight channels in registers, and a DPP row broadcast hands them over -- no LDS tile (the first form staged the 16 union rows in LDS
// and read them back with 32-64 ds_read_b128 per lane and column block: the three scans were bound by LDS bandwidth, 57 / 84 / 113 us).
template <int MODE, int U> __device__ __forceinline__ void scan_step(const float (&a)[8], const float (&bsh)[8], const float (&mine)[8],
                                                                     uint32_t bits, float (&acc)[8]) {
  const bool on = bits >> U & 1u;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float tv = group16_bcast<U>(a[k]);
    if (MODE == kMax) acc[k] = fmaxf(acc[k], on ? tv : -INFINITY);
    else if (MODE == kTies) acc[k] += (on && tv == mine[k]) ? 1.f : 0.f;
    else {
      const float sv = group16_bcast<U>(bsh[k]);
      acc[k] += (on && tv == mine[k]) ? sv : 0.f;
    }
  }
}
template <int MODE, int... U> __device__ __forceinline__ void scan_steps(std::integer_sequence<int, U...>, const float (&a)[8],
                                                                         const float (&bsh)[8], const float (&mine)[8], uint32_t bits,
                                                                         float (&acc)[8]) {
  (scan_step<MODE, U>(a, bsh, mine, bits, acc), ...);
}

template <int MODE> __global__ __launch_bounds__(kBlock) void dense_pool_scan_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                                                    const float* __restrict__ own, int D,
                                                                                    float* __restrict__ out, const DensePlan p) {
  constexpr int LD = 32;                                   // every matrix here: rows of 32 floats
  __shared__ __attribute__((aligned(16))) int lds[kDbLdsInts];
  __shared__ __attribute__((aligned(16))) float red[kDbWaves][kWave][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int nblocks = *p.counter / kDbRows;
  for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {
    const int32_t* __restrict__ rec = p.records + (int64_t)b * kDbStride;
    const int4 hdr = *reinterpret_cast<const int4*>(rec);
    if (!hdr.z) continue;
    const int nrows = hdr.x, ncb = (hdr.y + 15) >> 4;
    const bool valid = r < nrows;
    const int row = valid ? rec[kDbRowsOff + r] : rec[kDbRowsOff];
    const int selfs = valid ? rec[kDbSelfOff + r] : -1;
    float mine[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, acc[8];
    if (MODE != kMax) {
      const f4a o0 = *reinterpret_cast<const f4a*>(own + (int64_t)row * LD + 8 * g), o1 = *reinterpret_cast<const f4a*>(own + (int64_t)row * LD + 8 * g + 4);
      mine[0] = o0.x; mine[1] = o0.y; mine[2] = o0.z; mine[3] = o0.w; mine[4] = o1.x; mine[5] = o1.y; mine[6] = o1.z; mine[7] = o1.w;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = MODE == kMax ? -INFINITY : 0.f;
    const BlockLds l = block_stage(rec, lds);
    const uint32_t* maskrow = l.mask + r * kDbMaskWords;
    // as a stager the lane is (union row r of the column block, channels 8 g .. 8 g + 7); the rows of the wave's NEXT column block are
    // in flight while this one is walked
    f4a n0, n1, m0, m1;
    auto fetch = [&](int cb) {
      const int uid = l.uni[16 * cb + r];
      n0 = *reinterpret_cast<const f4a*>(A + (int64_t)uid * LD + 8 * g);
      n1 = *reinterpret_cast<const f4a*>(A + (int64_t)uid * LD + 8 * g + 4);
      if (MODE == kShare) {
        m0 = *reinterpret_cast<const f4a*>(B + (int64_t)uid * LD + 8 * g);
        m1 = *reinterpret_cast<const f4a*>(B + (int64_t)uid * LD + 8 * g + 4);
      }
    };
    if (wave < ncb) fetch(wave);
    for (int cb = wave; cb < ncb; cb += kDbWaves) {
      const float a[8] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w};
      float bsh[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (MODE == kShare) { bsh[0] = m0.x; bsh[1] = m0.y; bsh[2] = m0.z; bsh[3] = m0.w; bsh[4] = m1.x; bsh[5] = m1.y; bsh[6] = m1.z; bsh[7] = m1.w; }
      if (cb + kDbWaves < ncb) fetch(cb + kDbWaves);
      uint32_t bits = (maskrow[cb >> 1] >> ((cb & 1) * 16)) & 0xFFFFu;       // the row's cells of this column block
      const int d = selfs - 16 * cb;
      if ((unsigned)d < 16u) bits |= 1u << d;              // the row itself: always an entry here
      scan_steps<MODE>(std::make_integer_sequence<int, 16>{}, a, bsh, mine, bits, acc);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[wave][lane][k] = acc[k];
    __syncthreads();
    if (wave == 0 && valid) {
      float t[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        t[k] = red[0][lane][k];
#pragma unroll
        for (int w = 1; w < kDbWaves; ++w) t[k] = MODE == kMax ? fmaxf(t[k], red[w][lane][k]) : t[k] + red[w][lane][k];
      }
      float* __restrict__ o = out + (int64_t)row * LD + 8 * g;
      if (MODE == kShare) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (8 * g + k < D) t[k] += o[k];
      }
      store8(o, t, 8 * g, D);
    }
  }
}

// the segment max of the rows OUTSIDE the blocks (the per-edge form: a 16-lane group per row, lane l channels 2 l and 2 l + 1 -- rows of exactly
// 32 floats: every load is in bounds whatever D is, and a maximum does not mind an entry read twice, so four source rows fly together from
// clamped places with no mask at all)
__global__ __launch_bounds__(kBlock) void pool_segment_max_rest_kernel(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ ptr,
                                                                       const int32_t* __restrict__ idx, int64_t N, int D,
                                                                       float* __restrict__ out, int64_t ldo, const uint8_t* __restrict__ skip) {
  const int64_t row = ((int64_t)row_block() * kBlock + threadIdx.x) / kGroup;
  const int l = threadIdx.x % kGroup;
  if (row >= N || skip[row]) return;
  const float2* __restrict__ xr = reinterpret_cast<const float2*>(x) + l;       // ldx == 32 (the host checks): 16 pairs a row
  float2 m = xr[row * 16];                                                       // the row itself
  const int beg = ptr[row], end = ptr[row + 1];
  for (int e0 = beg; e0 < end; e0 += kGroup) {
    const int k = min(kGroup, end - e0);
    const int j = idx[e0 + min(l, k - 1)];
    for (int u = 0; u < k; u += 4) {
      const int j0 = __shfl(j, u, kGroup), j1 = __shfl(j, min(u + 1, k - 1), kGroup), j2 = __shfl(j, min(u + 2, k - 1), kGroup),
                j3 = __shfl(j, min(u + 3, k - 1), kGroup);
      const float2 v0 = xr[(int64_t)j0 * 16], v1 = xr[(int64_t)j1 * 16], v2 = xr[(int64_t)j2 * 16], v3 = xr[(int64_t)j3 * 16];
      m.x = fmaxf(fmaxf(m.x, v0.x), fmaxf(fmaxf(v1.x, v2.x), v3.x));
      m.y = fmaxf(fmaxf(m.y, v0.y), fmaxf(fmaxf(v1.y, v2.y), v3.y));
    }
  }
  float* __restrict__ o = out + row * ldo + 2 * l;
  if (2 * l + 1 < D) *reinterpret_cast<float2*>(o) = m;
  else if (2 * l < D) o[0] = m.x;
}

// ------------------------------------------------------------------------------------------------- destination-side backward
// Per block row i (softmax_aggregate_bwd_dst_kernel's formulas, family_b_bwd.hip): delta_i = gnew_i . x'_i, and per cell
//   al_ij = exp(LeakyReLU(a_i + c_j) - m_i) / den_i,   gp_ij = al_ij (gnew_i . x_j - delta_i) LeakyReLU'(a_i + c_j),   g_a[i] = sum_j gp_ij;
// the record {a_i, m_i, 1 / den_i, delta_i} is filed for the source side.  gnew_i . x_j for the block's cells is one product on the matrix
// cores (two k-tiles of 16 channels).  m_i, 1 / den_i: the forward's (dense_softmax_aggregate_kernel).
__device__ __forceinline__ f4u load4_below(const float* __restrict__ p, int c0, int D) {       // channels c0 .. c0 + 3, zeros from D on
  f4u v = *reinterpret_cast<const f4u*>(p);
  if (c0 + 3 >= D) v.w = 0.f;
  if (c0 + 2 >= D) v.z = 0.f;
  if (c0 + 1 >= D) v.y = 0.f;
  if (c0 >= D) v.x = 0.f;
  return v;
}

__global__ __launch_bounds__(kBlock) void dense_softmax_aggregate_bwd_dst_kernel(const float* __restrict__ x, const float* __restrict__ xnew,
                                                                                 const float* __restrict__ gnew, const float* __restrict__ a_dst,
                                                                                 const float* __restrict__ c_src, const float2* __restrict__ stat,
                                                                                 float slope, int D, float4* __restrict__ rec_out,
                                                                                 float* __restrict__ g_a, const DensePlan p) {
  constexpr int LD = 32;
  __shared__ __attribute__((aligned(16))) int lds[kDbLdsInts];
  __shared__ float red[kDbWaves][kDbRows];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int nblocks = *p.counter / kDbRows;
  for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {
    const int32_t* __restrict__ rec = p.records + (int64_t)b * kDbStride;
    const int4 hdr = *reinterpret_cast<const int4*>(rec);
    if (!hdr.z) continue;
    const int nrows = hdr.x, ncb = (hdr.y + 15) >> 4;
    const bool valid = r < nrows;
    const int row = valid ? rec[kDbRowsOff + r] : rec[kDbRowsOff];
    const int selfs = valid ? rec[kDbSelfOff + r] : -1;
    const float ai = a_dst[row];
    const float2 st = stat[row];
    const float m2 = st.x * kLog2e, inv = st.y;
    f4u gi[kPoolTiles];
    float dpart = 0.f;
#pragma unroll
    for (int t = 0; t < kPoolTiles; ++t) {
      gi[t] = load4_below(gnew + (int64_t)row * LD + 16 * t + 4 * g, 16 * t + 4 * g, D);
      dpart += dot4(gi[t], load4_below(xnew + (int64_t)row * LD + 16 * t + 4 * g, 16 * t + 4 * g, D));
    }
    const float delta = rows_sum(dpart);
    const BlockLds l = block_stage(rec, lds);
    const uint32_t* maskrow = l.mask + r * kDbMaskWords;
    float ga = 0.f;
    for (int cb = wave; cb < ncb; cb += kDbWaves) {
      const int uid = l.uni[16 * cb + r];
      f4u xa[kPoolTiles];
#pragma unroll
      for (int t = 0; t < kPoolTiles; ++t) xa[t] = load4_below(x + (int64_t)uid * LD + 16 * t + 4 * g, 16 * t + 4 * g, D);
      const int4 id4 = *reinterpret_cast<const int4*>(l.uni + 16 * cb + 4 * g);
      const float c4[4] = {c_src[id4.x], c_src[id4.y], c_src[id4.z], c_src[id4.w]};
      const uint32_t nib = pool_bits(maskrow, cb, g, selfs);
      f32x4 dots = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < kPoolTiles; ++t) dots = mfma16(xa[t], gi[t], dots);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float pre = ai + c4[i];
        const float al = (nib >> i & 1u) ? __builtin_amdgcn_exp2f(leaky(pre, slope) * kLog2e - m2) * inv : 0.f;
        ga += al * (dots[i] - delta) * (pre > 0.f ? 1.f : slope);
      }
    }
    ga = rows_sum(ga);
    if (g == 0) red[wave][r] = ga;
    __syncthreads();
    if (wave == 0 && g == 0 && valid) {
      g_a[row] = red[0][r] + red[1][r] + red[2][r] + red[3][r];
      rec_out[row] = make_float4(ai, st.x, inv, delta);
    }
  }
}

// ------------------------------------------------------------------------------------------------------ source-side backward
// The plan of the OUT structure: block rows are sources j, the union holds their destinations i (softmax_aggregate_bwd_src_rc_kernel's
// formulas): g_x[j] = sum_i al_ij gnew_i + g_c[j] rank1,  g_c[j] = sum_i gp_ij, every weight recomputed from destination i's record.
__global__ __launch_bounds__(kBlock) void dense_softmax_aggregate_bwd_src_kernel(const float* __restrict__ x, const float* __restrict__ gnew,
                                                                                 const float4* __restrict__ stat, const float* __restrict__ c_src,
                                                                                 float slope, int D, float* __restrict__ gx, float* __restrict__ g_c,
                                                                                 const float* __restrict__ rank1, const DensePlan p) {
  constexpr int LD = 32;
  __shared__ __attribute__((aligned(16))) int lds[kDbLdsInts];
  __shared__ __attribute__((aligned(16))) f32x4 red_o[kPoolTiles][kDbWaves * kWave];
  __shared__ float red[kDbWaves][kDbRows];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int nblocks = *p.counter / kDbRows;
  const bool has1 = 16 + r < D;
  for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {
    const int32_t* __restrict__ rec = p.records + (int64_t)b * kDbStride;
    const int4 hdr = *reinterpret_cast<const int4*>(rec);
    if (!hdr.z) continue;
    const int nrows = hdr.x, ncb = (hdr.y + 15) >> 4;
    const bool valid = r < nrows;
    const int row = valid ? rec[kDbRowsOff + r] : rec[kDbRowsOff];
    const int selfs = valid ? rec[kDbSelfOff + r] : -1;
    const float cj = c_src[row];
    f4u xown[kPoolTiles];
#pragma unroll
    for (int t = 0; t < kPoolTiles; ++t) xown[t] = load4_below(x + (int64_t)row * LD + 16 * t + 4 * g, 16 * t + 4 * g, D);
    const BlockLds l = block_stage(rec, lds);
    const uint32_t* maskrow = l.mask + r * kDbMaskWords;
    float gc = 0.f;
    f32x4 o[kPoolTiles];
#pragma unroll
    for (int t = 0; t < kPoolTiles; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int cb = wave; cb < ncb; cb += kDbWaves) {
      const int uid = l.uni[16 * cb + r];
      f4u ga4[kPoolTiles];
#pragma unroll
      for (int t = 0; t < kPoolTiles; ++t) ga4[t] = load4_below(gnew + (int64_t)uid * LD + 16 * t + 4 * g, 16 * t + 4 * g, D);
      const int4 id4 = *reinterpret_cast<const int4*>(l.uni + 16 * cb + 4 * g);
      const int ids[4] = {id4.x, id4.y, id4.z, id4.w};
      float4 st[4];
      float gv[kPoolTiles][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        st[i] = stat[ids[i]];                              // {a_i, m_i, 1 / den_i, delta_i}
        const float* __restrict__ gp = gnew + (int64_t)ids[i] * LD + r;
        gv[0][i] = gp[0];
        gv[1][i] = has1 ? gp[16] : 0.f;
      }
      const uint32_t nib = pool_bits(maskrow, cb, g, selfs);
      f32x4 dots = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < kPoolTiles; ++t) dots = mfma16(ga4[t], xown[t], dots);          // gnew_i . x_j
      float al[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float pre = st[i].x + cj;
        al[i] = (nib >> i & 1u) ? __builtin_amdgcn_exp2f((leaky(pre, slope) - st[i].y) * kLog2e) * st[i].z : 0.f;
        gc += al[i] * (dots[i] - st[i].w) * (pre > 0.f ? 1.f : slope);
      }
#pragma unroll
      for (int t = 0; t < kPoolTiles; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) o[t] = mfma4(gv[t][i], al[i], o[t]);
    }
    gc = rows_sum(gc);
    if (g == 0) red[wave][r] = gc;
#pragma unroll
    for (int t = 0; t < kPoolTiles; ++t) red_o[t][wave * kWave + lane] = o[t];
    __syncthreads();
    if (wave < kPoolTiles && valid) {
      const int t = wave, c0 = 16 * t + 4 * g;
      const float gct = red[0][r] + red[1][r] + red[2][r] + red[3][r];
      const f32x4 acc = waves_sum(red_o[t], lane);
      float* __restrict__ orow = gx + (int64_t)row * LD + c0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (c0 + i < D) orow[i] = acc[i] + (rank1 ? gct * rank1[c0 + i] : 0.f);
      if (t == 0 && g == 0) g_c[row] = gct;
    }
  }
}

static int pool_grid(int64_t max_blocks) { return (int)std::max<int64_t>(1, std::min<int64_t>(max_blocks, 16384)); }

static bool plan_ok(const int32_t* records, const int32_t* counter, const uint8_t* row_flag, int64_t max_blocks) {
  return records && counter && row_flag && max_blocks > 0 && aligned_to(records, 16);
}

}  // namespace mlqem

using namespace mlqem;

extern "C" int mlqem_dense_pool_supported(int D) { return D > 0 && D <= 16 * kPoolTiles; }

extern "C" int mlqem_dense_softmax_aggregate_f32(const float* x, int64_t ldx, const int32_t* in_ptr, const int32_t* in_src, const float* a_dst,
                                                 const float* c_src, float negative_slope, int64_t N, int D, const int32_t* records,
                                                 const int32_t* counter, const uint8_t* row_flag, int64_t max_blocks, float* out, int64_t ldo,
                                                 float* stat, mlqem_stream_t stream) {
  begin_launches();
  if (!mlqem_dense_pool_supported(D)) return MLQEM_ERR_UNSUPPORTED;
  if (N < 0 || ldx < D || ldo < D) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!x || !in_ptr || !a_dst || !c_src || !out || !plan_ok(records, counter, row_flag, max_blocks) || (stat && !aligned_to(stat, 8)))
    return MLQEM_ERR_BAD_ARG;
  if (N > INT32_MAX) return MLQEM_ERR_UNSUPPORTED;
  const hipStream_t s = as_stream(stream);
  launch_softmax_aggregate(x, ldx, in_ptr, in_src, a_dst, c_src, negative_slope, N, D, out, ldo, row_flag, s);
  const DensePlan p{records, counter, row_flag, max_blocks};
  hipLaunchKernelGGL(dense_softmax_aggregate_kernel, dim3((unsigned)pool_grid(max_blocks)), dim3(kBlock), 0, s, x, ldx, a_dst, c_src,
                     negative_slope, D, out, ldo, reinterpret_cast<float2*>(stat), p);
  return launch_status();
}

// every matrix of the scans and of the backward kernels: rows of exactly 32 floats, 16-byte aligned
static bool rows32(const void* p, int64_t ld) { return p && ld == 32 && aligned_to(p, 16); }

extern "C" int mlqem_dense_segment_max_f32(const float* x, int64_t ldx, const int32_t* in_ptr, const int32_t* in_src, int64_t N, int D,
                                           const int32_t* records, const int32_t* counter, const uint8_t* row_flag, int64_t max_blocks,
                                           float* out, int64_t ldo, mlqem_stream_t stream) {
  begin_launches();
  if (!mlqem_dense_pool_supported(D) || D <= 28) return MLQEM_ERR_UNSUPPORTED;
  if (N < 0) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!rows32(x, ldx) || !rows32(out, ldo) || !in_ptr || !in_src || !plan_ok(records, counter, row_flag, max_blocks)) return MLQEM_ERR_BAD_ARG;
  if (N > INT32_MAX) return MLQEM_ERR_UNSUPPORTED;
  const hipStream_t s = as_stream(stream);
  const DensePlan p{records, counter, row_flag, max_blocks};
  hipLaunchKernelGGL(pool_segment_max_rest_kernel, dim3((unsigned)ceil_div(N * kGroup, kBlock)), dim3(kBlock), 0, s, x, ldx, in_ptr, in_src, N, D,
                     out, ldo, row_flag);
  hipLaunchKernelGGL(dense_pool_scan_kernel<kMax>, dim3((unsigned)pool_grid(max_blocks)), dim3(kBlock), 0, s, x, nullptr, nullptr, D, out, p);
  return launch_status();
}

extern "C" int mlqem_dense_softmax_aggregate_bwd_f32(const float* x, int64_t ldx, const float* xnew, int64_t ldn, const float* gnew, int64_t ldg,
                                                     const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                                                     const float* a_dst, const float* c_src, float negative_slope, int64_t N, int64_t E, int D,
                                                     const float* stat, const int32_t* in_records, const int32_t* in_counter,
                                                     const uint8_t* in_flag, int64_t in_max_blocks, const int32_t* out_records,
                                                     const int32_t* out_counter, const uint8_t* out_flag, int64_t out_max_blocks, float* gx,
                                                     int64_t ldgx, float* g_a, float* g_c, float* edge_al, const float* xmax, int64_t ldm,
                                                     float* tie_count, int64_t ldt, const float* gx_rank1, mlqem_stream_t stream) {
  begin_launches();
  if (!mlqem_dense_pool_supported(D) || D <= 28) return MLQEM_ERR_UNSUPPORTED;
  if (N < 0 || E < 0) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!rows32(x, ldx) || !rows32(xnew, ldn) || !rows32(gnew, ldg) || !rows32(gx, ldgx) || !rows32(xmax, ldm) || !rows32(tie_count, ldt) ||
      !stat || !aligned_to(stat, 8) || !a_dst || !c_src || !g_a || !g_c || !edge_al || !aligned_to(edge_al, 16) ||
      !plan_ok(in_records, in_counter, in_flag, in_max_blocks) || !plan_ok(out_records, out_counter, out_flag, out_max_blocks))
    return MLQEM_ERR_BAD_ARG;
  if (N > INT32_MAX) return MLQEM_ERR_UNSUPPORTED;
  const hipStream_t s = as_stream(stream);
  const DensePlan pin{in_records, in_counter, in_flag, in_max_blocks}, pout{out_records, out_counter, out_flag, out_max_blocks};
  // destination side: the per-edge kernel (with its tie counts) over the rows outside the blocks, then the blocks
  int code = softmax_aggregate_bwd_launches(x, ldx, xnew, ldn, gnew, ldg, in_ptr, in_src, out_ptr, out_dst, nullptr, a_dst, c_src, negative_slope,
                                            N, E, D, 0, gx, ldgx, g_a, g_c, edge_al, nullptr, xmax, ldm, tie_count, ldt, gx_rank1, in_flag,
                                            out_flag, 1, stream);
  if (code != MLQEM_OK) return code;
  const dim3 gin((unsigned)pool_grid(in_max_blocks)), gout((unsigned)pool_grid(out_max_blocks));
  hipLaunchKernelGGL(dense_softmax_aggregate_bwd_dst_kernel, gin, dim3(kBlock), 0, s, x, xnew, gnew, a_dst, c_src,
                     reinterpret_cast<const float2*>(stat), negative_slope, D, reinterpret_cast<float4*>(edge_al), g_a, pin);
  hipLaunchKernelGGL(dense_pool_scan_kernel<kTies>, gin, dim3(kBlock), 0, s, x, nullptr, xmax, D, tie_count, pin);
  // source side (reads the records of ALL destinations)
  code = softmax_aggregate_bwd_launches(x, ldx, xnew, ldn, gnew, ldg, in_ptr, in_src, out_ptr, out_dst, nullptr, a_dst, c_src, negative_slope, N,
                                        E, D, 0, gx, ldgx, g_a, g_c, edge_al, nullptr, xmax, ldm, tie_count, ldt, gx_rank1, in_flag, out_flag, 2,
                                        stream);
  if (code != MLQEM_OK) return code;
  hipLaunchKernelGGL(dense_softmax_aggregate_bwd_src_kernel, gout, dim3(kBlock), 0, s, x, gnew, reinterpret_cast<const float4*>(edge_al), c_src,
                     negative_slope, D, gx, g_c, gx_rank1, pout);
  return launch_status();
}

extern "C" int mlqem_dense_segment_max_bwd_f32(const float* x, int64_t ldx, const float* xmax, int64_t ldm, const int32_t* in_ptr,
                                               const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst, int64_t N, int D, float* gx,
                                               int64_t ldgx, float* gshare, int64_t lds, const float* tie_count, int64_t ldt,
                                               const float* gmax_row, const float* gmax_col, const int32_t* out_records,
                                               const int32_t* out_counter, const uint8_t* out_flag, int64_t out_max_blocks,
                                               mlqem_stream_t stream) {
  begin_launches();
  if (!mlqem_dense_pool_supported(D) || D <= 28) return MLQEM_ERR_UNSUPPORTED;
  if (N < 0) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!rows32(x, ldx) || !rows32(xmax, ldm) || !rows32(gx, ldgx) || !rows32(gshare, lds) || !rows32(tie_count, ldt) || !gmax_row || !gmax_col ||
      !plan_ok(out_records, out_counter, out_flag, out_max_blocks))
    return MLQEM_ERR_BAD_ARG;
  if (N > INT32_MAX) return MLQEM_ERR_UNSUPPORTED;
  const int code = segment_max_bwd_launches(x, ldx, xmax, ldm, nullptr, 0, in_ptr, in_src, out_ptr, out_dst, N, D, gx, ldgx, gshare, lds, tie_count,
                                            ldt, gmax_row, gmax_col, out_flag, stream);
  if (code != MLQEM_OK) return code;
  const DensePlan pout{out_records, out_counter, out_flag, out_max_blocks};
  hipLaunchKernelGGL(dense_pool_scan_kernel<kShare>, dim3((unsigned)pool_grid(out_max_blocks)), dim3(kBlock), 0, as_stream(stream), xmax, gshare, x,
                     D, gx, pout);
  return launch_status();
}

