// ASAPooling's cluster sums (docs/tutorials/gnn.py:85,92,104-112) over the dense blocks of a structure
// (dense_block.hpp): the same plans as TransformerConv's edge softmax, the same wave layout -- a workgroup per block, its four waves
// sharing the column blocks, cell (u = 16 cb + 4 g + i, row r) in register i of lane l = 16 g + r.
//
//   x'[i] = sum_j softmax_j(LeakyReLU(a_i + c_j)) x[j]   over the in-entries of i AND i itself (add_remaining_self_loops)
//
// The score of a cell is one add of two per-node scalars; the weighted sum of the source rows is x'^T = X_U^T W on the f32 matrix
// cores, one accumulator tile per 16 channels (rows of at most 32 channels: the reference's 30).  Rows outside the blocks stay with the
// per-edge kernels (attn.hip, family_b_bwd.hip), which skip the rows a plan flags.
#include "dense_block.hpp"

namespace mlqem {

namespace {

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

int pool_grid(int64_t max_blocks) { return (int)std::max<int64_t>(1, std::min<int64_t>(max_blocks, 16384)); }

bool plan_ok(const int32_t* records, const int32_t* counter, const uint8_t* row_flag, int64_t max_blocks) {
  return records && counter && row_flag && max_blocks > 0 && aligned_to(records, 16);
}

}  // namespace
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
