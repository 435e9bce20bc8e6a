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

constexpr int kPoolMaxTiles = 3;                            // channel tiles of 16: the kernels are built for T = 2 (D <= 32, rows of 32 floats)
                                                            // and T = 3 (D <= 48, rows of 48: the heads-5/3 variants pool 45 channels, gnn.py:178-276)

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
template <int T> __global__ __launch_bounds__(kBlock) void dense_softmax_aggregate_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ a_dst,
                                                                         const float* __restrict__ c_src, float slope, int D,
                                                                         float* __restrict__ out, int64_t ldo, float2* __restrict__ stat,
                                                                         const DensePlan p) {
  __shared__ __attribute__((aligned(16))) int lds[kDbLdsInts];
  __shared__ __attribute__((aligned(16))) f32x4 red_o[T][kDbWaves * kWave];
  __shared__ float red_m[kDbWaves][kDbRows], red_d[kDbWaves][kDbRows];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int nblocks = *p.counter / kDbRows;
  bool hast[T];                                            // the lane's channel of tile t exists
#pragma unroll
  for (int t = 0; t < T; ++t) hast[t] = 16 * t + r < D;
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
    f32x4 o[T];
#pragma unroll
    for (int t = 0; t < T; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int cb = wave; cb < ncb; cb += kDbWaves) {
      const int4 id4 = *reinterpret_cast<const int4*>(l.uni + 16 * cb + 4 * g);
      const int ids[4] = {id4.x, id4.y, id4.z, id4.w};
      float c4[4], xv[T][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        c4[i] = c_src[ids[i]];
        const float* __restrict__ xr = x + (int64_t)ids[i] * ldx + r;
#pragma unroll
        for (int t = 0; t < T; ++t) xv[t][i] = hast[t] ? xr[16 * t] : 0.f;
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
      for (int t = 0; t < T; ++t) {
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
    for (int t = 0; t < T; ++t) red_o[t][wave * kWave + lane] = o[t] * rs;
    __syncthreads();
    if (wave < T && valid) {                      // wave t finishes channel tile t: channels 16 t + 4 g .. + 3 of the row
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

__device__ __forceinline__ void store4(float* __restrict__ p, const float* v, int c0, int D) {      // channels c0 .. c0 + 3, those below D
  if (c0 + 4 <= D) {
    *reinterpret_cast<f4a*>(p) = f4a{v[0], v[1], v[2], v[3]};
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (c0 + k < D) p[k] = v[k];
  }
}

// One union row of the column block against the lane's row: the stager lanes of the same 16-lane row (same g) hold that union row's
// eight channels in registers, and a DPP row broadcast hands them over -- no LDS tile (the first form staged the 16 union rows in LDS
// and read them back with 32-64 ds_read_b128 per lane and column block: the three scans were bound by LDS bandwidth, 57 / 84 / 113 us).
template <int MODE, int U, int CPL> __device__ __forceinline__ void scan_step(const float (&a)[CPL], const float (&bsh)[CPL], const float (&mine)[CPL],
                                                                              uint32_t bits, float (&acc)[CPL]) {
  const bool on = bits >> U & 1u;
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const float tv = group16_bcast<U>(a[k]);
    if (MODE == kMax) acc[k] = fmaxf(acc[k], on ? tv : -INFINITY);
    else if (MODE == kTies) acc[k] += (on && tv == mine[k]) ? 1.f : 0.f;
    else {
      const float sv = group16_bcast<U>(bsh[k]);
      acc[k] += (on && tv == mine[k]) ? sv : 0.f;
    }
  }
}
template <int MODE, int CPL, int... U> __device__ __forceinline__ void scan_steps(std::integer_sequence<int, U...>, const float (&a)[CPL],
                                                                                  const float (&bsh)[CPL], const float (&mine)[CPL], uint32_t bits,
                                                                                  float (&acc)[CPL]) {
  (scan_step<MODE, U, CPL>(a, bsh, mine, bits, acc), ...);
}

template <int MODE, int T> __global__ __launch_bounds__(kBlock) void dense_pool_scan_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                                                    const float* __restrict__ own, int D,
                                                                                    float* __restrict__ out, const DensePlan p) {
  constexpr int LD = 16 * T, CPL = 4 * T;                  // every matrix here: rows of LD floats; a lane owns CPL channels of its row
  __shared__ __attribute__((aligned(16))) int lds[kDbLdsInts];
  __shared__ __attribute__((aligned(16))) float red[kDbWaves][kWave][CPL];
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
    float mine[CPL], acc[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) mine[k] = 0.f;
    if (MODE != kMax) {
#pragma unroll
      for (int q = 0; q < T; ++q) {
        const f4a o0 = *reinterpret_cast<const f4a*>(own + (int64_t)row * LD + CPL * g + 4 * q);
        mine[4 * q] = o0.x; mine[4 * q + 1] = o0.y; mine[4 * q + 2] = o0.z; mine[4 * q + 3] = o0.w;
      }
    }
#pragma unroll
    for (int k = 0; k < CPL; ++k) acc[k] = MODE == kMax ? -INFINITY : 0.f;
    const BlockLds l = block_stage(rec, lds);
    const uint32_t* maskrow = l.mask + r * kDbMaskWords;
    // as a stager the lane is (union row r of the column block, channels 8 g .. 8 g + 7); the rows of the wave's NEXT column block are
    // in flight while this one is walked
    f4a nn[T], mm[T];
    auto fetch = [&](int cb) {
      const int uid = l.uni[16 * cb + r];
#pragma unroll
      for (int q = 0; q < T; ++q) nn[q] = *reinterpret_cast<const f4a*>(A + (int64_t)uid * LD + CPL * g + 4 * q);
      if (MODE == kShare) {
#pragma unroll
        for (int q = 0; q < T; ++q) mm[q] = *reinterpret_cast<const f4a*>(B + (int64_t)uid * LD + CPL * g + 4 * q);
      }
    };
    if (wave < ncb) fetch(wave);
    for (int cb = wave; cb < ncb; cb += kDbWaves) {
      float a[CPL], bsh[CPL];
#pragma unroll
      for (int q = 0; q < T; ++q) {
        a[4 * q] = nn[q].x; a[4 * q + 1] = nn[q].y; a[4 * q + 2] = nn[q].z; a[4 * q + 3] = nn[q].w;
        bsh[4 * q] = bsh[4 * q + 1] = bsh[4 * q + 2] = bsh[4 * q + 3] = 0.f;
        if (MODE == kShare) { bsh[4 * q] = mm[q].x; bsh[4 * q + 1] = mm[q].y; bsh[4 * q + 2] = mm[q].z; bsh[4 * q + 3] = mm[q].w; }
      }
      if (cb + kDbWaves < ncb) fetch(cb + kDbWaves);
      uint32_t bits = (maskrow[cb >> 1] >> ((cb & 1) * 16)) & 0xFFFFu;       // the row's cells of this column block
      const int d = selfs - 16 * cb;
      if ((unsigned)d < 16u) bits |= 1u << d;              // the row itself: always an entry here
      scan_steps<MODE, CPL>(std::make_integer_sequence<int, 16>{}, a, bsh, mine, bits, acc);
    }
#pragma unroll
    for (int k = 0; k < CPL; ++k) red[wave][lane][k] = acc[k];
    __syncthreads();
    if (wave == 0 && valid) {
      float t[CPL];
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        t[k] = red[0][lane][k];
#pragma unroll
        for (int w = 1; w < kDbWaves; ++w) t[k] = MODE == kMax ? fmaxf(t[k], red[w][lane][k]) : t[k] + red[w][lane][k];
      }
      float* __restrict__ o = out + (int64_t)row * LD + CPL * g;
      if (MODE == kShare) {
#pragma unroll
        for (int k = 0; k < CPL; ++k)
          if (CPL * g + k < D) t[k] += o[k];
      }
#pragma unroll
      for (int q = 0; q < T; ++q) store4(o + 4 * q, t + 4 * q, CPL * g + 4 * q, D);
    }
  }
}

// the segment max of the rows OUTSIDE the blocks (the per-edge form: a 16-lane group per row, lane l channels l + 16 t -- rows of exactly
// LD = 16 T floats: every load is in bounds whatever D is, and a maximum does not mind an entry read twice, so four source rows fly
// together from clamped places with no mask at all)
template <int T>
__global__ __launch_bounds__(kBlock) void pool_segment_max_rest_kernel(const float* __restrict__ x, const int32_t* __restrict__ ptr,
                                                                       const int32_t* __restrict__ idx, int64_t N, int D,
                                                                       float* __restrict__ out, const uint8_t* __restrict__ skip) {
  constexpr int LD = 16 * T;
  const int64_t row = ((int64_t)row_block() * kBlock + threadIdx.x) / kGroup;
  const int l = threadIdx.x % kGroup;
  if (row >= N || skip[row]) return;
  const float* __restrict__ xr = x + l;
  float m[T];
#pragma unroll
  for (int t = 0; t < T; ++t) m[t] = xr[row * LD + 16 * t];                      // the row itself
  const int beg = ptr[row], end = ptr[row + 1];
  for (int e0 = beg; e0 < end; e0 += kGroup) {
    const int k = min(kGroup, end - e0);
    const int j = idx[e0 + min(l, k - 1)];
    for (int u = 0; u < k; u += 4) {
      const int j0 = __shfl(j, u, kGroup), j1 = __shfl(j, min(u + 1, k - 1), kGroup), j2 = __shfl(j, min(u + 2, k - 1), kGroup),
                j3 = __shfl(j, min(u + 3, k - 1), kGroup);
      float v[4][T];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        v[0][t] = xr[(int64_t)j0 * LD + 16 * t]; v[1][t] = xr[(int64_t)j1 * LD + 16 * t];
        v[2][t] = xr[(int64_t)j2 * LD + 16 * t]; v[3][t] = xr[(int64_t)j3 * LD + 16 * t];
      }
#pragma unroll
      for (int t = 0; t < T; ++t) m[t] = fmaxf(fmaxf(m[t], v[0][t]), fmaxf(fmaxf(v[1][t], v[2][t]), v[3][t]));
    }
  }
#pragma unroll
  for (int t = 0; t < T; ++t)
    if (l + 16 * t < D) out[row * LD + l + 16 * t] = m[t];
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

template <int T> __global__ __launch_bounds__(kBlock) void dense_softmax_aggregate_bwd_dst_kernel(const float* __restrict__ x, const float* __restrict__ xnew,
                                                                                 const float* __restrict__ gnew, const float* __restrict__ a_dst,
                                                                                 const float* __restrict__ c_src, const float2* __restrict__ stat,
                                                                                 float slope, int D, float4* __restrict__ rec_out,
                                                                                 float* __restrict__ g_a, const DensePlan p) {
  constexpr int LD = 16 * T;
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
    f4u gi[T];
    float dpart = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      gi[t] = load4_below(gnew + (int64_t)row * LD + 16 * t + 4 * g, 16 * t + 4 * g, D);
      dpart += dot4(gi[t], load4_below(xnew + (int64_t)row * LD + 16 * t + 4 * g, 16 * t + 4 * g, D));
    }
    const float delta = rows_sum(dpart);
    const BlockLds l = block_stage(rec, lds);
    const uint32_t* maskrow = l.mask + r * kDbMaskWords;
    float ga = 0.f;
    for (int cb = wave; cb < ncb; cb += kDbWaves) {
      const int uid = l.uni[16 * cb + r];
      f4u xa[T];
#pragma unroll
      for (int t = 0; t < T; ++t) xa[t] = load4_below(x + (int64_t)uid * LD + 16 * t + 4 * g, 16 * t + 4 * g, D);
      const int4 id4 = *reinterpret_cast<const int4*>(l.uni + 16 * cb + 4 * g);
      const float c4[4] = {c_src[id4.x], c_src[id4.y], c_src[id4.z], c_src[id4.w]};
      const uint32_t nib = pool_bits(maskrow, cb, g, selfs);
      f32x4 dots = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < T; ++t) dots = mfma16(xa[t], gi[t], dots);
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
template <int T> __global__ __launch_bounds__(kBlock) void dense_softmax_aggregate_bwd_src_kernel(const float* __restrict__ x, const float* __restrict__ gnew,
                                                                                 const float4* __restrict__ stat, const float* __restrict__ c_src,
                                                                                 float slope, int D, float* __restrict__ gx, float* __restrict__ g_c,
                                                                                 const float* __restrict__ rank1, const DensePlan p) {
  constexpr int LD = 16 * T;
  __shared__ __attribute__((aligned(16))) int lds[kDbLdsInts];
  __shared__ __attribute__((aligned(16))) f32x4 red_o[T][kDbWaves * kWave];
  __shared__ float red[kDbWaves][kDbRows];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int nblocks = *p.counter / kDbRows;
  bool hast[T];
#pragma unroll
  for (int t = 0; t < T; ++t) hast[t] = 16 * t + r < D;
  for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {
    const int32_t* __restrict__ rec = p.records + (int64_t)b * kDbStride;
    const int4 hdr = *reinterpret_cast<const int4*>(rec);
    if (!hdr.z) continue;
    const int nrows = hdr.x, ncb = (hdr.y + 15) >> 4;
    const bool valid = r < nrows;
    const int row = valid ? rec[kDbRowsOff + r] : rec[kDbRowsOff];
    const int selfs = valid ? rec[kDbSelfOff + r] : -1;
    const float cj = c_src[row];
    f4u xown[T];
#pragma unroll
    for (int t = 0; t < T; ++t) xown[t] = load4_below(x + (int64_t)row * LD + 16 * t + 4 * g, 16 * t + 4 * g, D);
    const BlockLds l = block_stage(rec, lds);
    const uint32_t* maskrow = l.mask + r * kDbMaskWords;
    float gc = 0.f;
    f32x4 o[T];
#pragma unroll
    for (int t = 0; t < T; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int cb = wave; cb < ncb; cb += kDbWaves) {
      const int uid = l.uni[16 * cb + r];
      f4u ga4[T];
#pragma unroll
      for (int t = 0; t < T; ++t) ga4[t] = load4_below(gnew + (int64_t)uid * LD + 16 * t + 4 * g, 16 * t + 4 * g, D);
      const int4 id4 = *reinterpret_cast<const int4*>(l.uni + 16 * cb + 4 * g);
      const int ids[4] = {id4.x, id4.y, id4.z, id4.w};
      float4 st[4];
      float gv[T][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        st[i] = stat[ids[i]];                              // {a_i, m_i, 1 / den_i, delta_i}
        const float* __restrict__ gp = gnew + (int64_t)ids[i] * LD + r;
#pragma unroll
        for (int t = 0; t < T; ++t) gv[t][i] = hast[t] ? gp[16 * t] : 0.f;
      }
      const uint32_t nib = pool_bits(maskrow, cb, g, selfs);
      f32x4 dots = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < T; ++t) dots = mfma16(ga4[t], xown[t], dots);          // gnew_i . x_j
      float al[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float pre = st[i].x + cj;
        al[i] = (nib >> i & 1u) ? __builtin_amdgcn_exp2f((leaky(pre, slope) - st[i].y) * kLog2e) * st[i].z : 0.f;
        gc += al[i] * (dots[i] - st[i].w) * (pre > 0.f ? 1.f : slope);
      }
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) o[t] = mfma4(gv[t][i], al[i], o[t]);
    }
    gc = rows_sum(gc);
    if (g == 0) red[wave][r] = gc;
#pragma unroll
    for (int t = 0; t < T; ++t) red_o[t][wave * kWave + lane] = o[t];
    __syncthreads();
    if (wave < T && valid) {
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

extern "C" int mlqem_dense_pool_supported(int D) { return D > 0 && D <= 16 * kPoolMaxTiles; }
static int pool_tiles(int D) { return D <= 32 ? 2 : 3; }
// the scans and the backward kernels address rows of exactly 16 T floats: D = 29..32 (T = 2) or 45..48 (T = 3) in the padded row layout
static bool pool_full_rows(int D) { return (D > 28 && D <= 32) || (D > 44 && D <= 48); }

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
  if (pool_tiles(D) == 2)
    hipLaunchKernelGGL(dense_softmax_aggregate_kernel<2>, dim3((unsigned)pool_grid(max_blocks)), dim3(kBlock), 0, s, x, ldx, a_dst, c_src,
                       negative_slope, D, out, ldo, reinterpret_cast<float2*>(stat), p);
  else
    hipLaunchKernelGGL(dense_softmax_aggregate_kernel<3>, dim3((unsigned)pool_grid(max_blocks)), dim3(kBlock), 0, s, x, ldx, a_dst, c_src,
                       negative_slope, D, out, ldo, reinterpret_cast<float2*>(stat), p);
  return launch_status();
}

// every matrix of the scans and of the backward kernels: rows of exactly 16 T floats, 16-byte aligned
static bool rows_ld(const void* p, int64_t ld, int D) { return p && ld == 16 * pool_tiles(D) && aligned_to(p, 16); }

extern "C" int mlqem_dense_segment_max_f32(const float* x, int64_t ldx, const int32_t* in_ptr, const int32_t* in_src, int64_t N, int D,
                                           const int32_t* records, const int32_t* counter, const uint8_t* row_flag, int64_t max_blocks,
                                           float* out, int64_t ldo, mlqem_stream_t stream) {
  begin_launches();
  if (!mlqem_dense_pool_supported(D) || !pool_full_rows(D)) return MLQEM_ERR_UNSUPPORTED;
  if (N < 0) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!rows_ld(x, ldx, D) || !rows_ld(out, ldo, D) || !in_ptr || !in_src || !plan_ok(records, counter, row_flag, max_blocks)) return MLQEM_ERR_BAD_ARG;
  if (N > INT32_MAX) return MLQEM_ERR_UNSUPPORTED;
  const hipStream_t s = as_stream(stream);
  const DensePlan p{records, counter, row_flag, max_blocks};
  const dim3 rest((unsigned)ceil_div(N * kGroup, kBlock)), grid((unsigned)pool_grid(max_blocks));
  if (pool_tiles(D) == 2) {
    hipLaunchKernelGGL(pool_segment_max_rest_kernel<2>, rest, dim3(kBlock), 0, s, x, in_ptr, in_src, N, D, out, row_flag);
    hipLaunchKernelGGL((dense_pool_scan_kernel<kMax, 2>), grid, dim3(kBlock), 0, s, x, nullptr, nullptr, D, out, p);
  } else {
    hipLaunchKernelGGL(pool_segment_max_rest_kernel<3>, rest, dim3(kBlock), 0, s, x, in_ptr, in_src, N, D, out, row_flag);
    hipLaunchKernelGGL((dense_pool_scan_kernel<kMax, 3>), grid, dim3(kBlock), 0, s, x, nullptr, nullptr, D, out, p);
  }
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
  if (!mlqem_dense_pool_supported(D) || !pool_full_rows(D)) return MLQEM_ERR_UNSUPPORTED;
  if (N < 0 || E < 0) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!rows_ld(x, ldx, D) || !rows_ld(xnew, ldn, D) || !rows_ld(gnew, ldg, D) || !rows_ld(gx, ldgx, D) || !rows_ld(xmax, ldm, D) || !rows_ld(tie_count, ldt, D) ||
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
  const bool t2 = pool_tiles(D) == 2;
  if (t2) {
    hipLaunchKernelGGL(dense_softmax_aggregate_bwd_dst_kernel<2>, gin, dim3(kBlock), 0, s, x, xnew, gnew, a_dst, c_src,
                       reinterpret_cast<const float2*>(stat), negative_slope, D, reinterpret_cast<float4*>(edge_al), g_a, pin);
    hipLaunchKernelGGL((dense_pool_scan_kernel<kTies, 2>), gin, dim3(kBlock), 0, s, x, nullptr, xmax, D, tie_count, pin);
  } else {
    hipLaunchKernelGGL(dense_softmax_aggregate_bwd_dst_kernel<3>, gin, dim3(kBlock), 0, s, x, xnew, gnew, a_dst, c_src,
                       reinterpret_cast<const float2*>(stat), negative_slope, D, reinterpret_cast<float4*>(edge_al), g_a, pin);
    hipLaunchKernelGGL((dense_pool_scan_kernel<kTies, 3>), gin, dim3(kBlock), 0, s, x, nullptr, xmax, D, tie_count, pin);
  }
  // source side (reads the records of ALL destinations)
  code = softmax_aggregate_bwd_launches(x, ldx, xnew, ldn, gnew, ldg, in_ptr, in_src, out_ptr, out_dst, nullptr, a_dst, c_src, negative_slope, N,
                                        E, D, 0, gx, ldgx, g_a, g_c, edge_al, nullptr, xmax, ldm, tie_count, ldt, gx_rank1, in_flag, out_flag, 2,
                                        stream);
  if (code != MLQEM_OK) return code;
  if (t2)
    hipLaunchKernelGGL(dense_softmax_aggregate_bwd_src_kernel<2>, gout, dim3(kBlock), 0, s, x, gnew, reinterpret_cast<const float4*>(edge_al), c_src,
                       negative_slope, D, gx, g_c, gx_rank1, pout);
  else
    hipLaunchKernelGGL(dense_softmax_aggregate_bwd_src_kernel<3>, gout, dim3(kBlock), 0, s, x, gnew, reinterpret_cast<const float4*>(edge_al), c_src,
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
  if (!mlqem_dense_pool_supported(D) || !pool_full_rows(D)) return MLQEM_ERR_UNSUPPORTED;
  if (N < 0) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!rows_ld(x, ldx, D) || !rows_ld(xmax, ldm, D) || !rows_ld(gx, ldgx, D) || !rows_ld(gshare, lds, D) || !rows_ld(tie_count, ldt, D) || !gmax_row || !gmax_col ||
      !plan_ok(out_records, out_counter, out_flag, out_max_blocks))
    return MLQEM_ERR_BAD_ARG;
  if (N > INT32_MAX) return MLQEM_ERR_UNSUPPORTED;
  const int code = segment_max_bwd_launches(x, ldx, xmax, ldm, nullptr, 0, in_ptr, in_src, out_ptr, out_dst, N, D, gx, ldgx, gshare, lds, tie_count,
                                            ldt, gmax_row, gmax_col, out_flag, stream);
  if (code != MLQEM_OK) return code;
  const DensePlan pout{out_records, out_counter, out_flag, out_max_blocks};
  if (pool_tiles(D) == 2)
    hipLaunchKernelGGL((dense_pool_scan_kernel<kShare, 2>), dim3((unsigned)pool_grid(out_max_blocks)), dim3(kBlock), 0, as_stream(stream), xmax,
                       gshare, x, D, gx, pout);
  else
    hipLaunchKernelGGL((dense_pool_scan_kernel<kShare, 3>), dim3((unsigned)pool_grid(out_max_blocks)), dim3(kBlock), 0, as_stream(stream), xmax,
                       gshare, x, D, gx, pout);
  return launch_status();
}

