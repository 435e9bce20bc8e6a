// CSR aggregation kernels: the "scatter-add" of the GNN layers, written as a gather over destination rows
// so that no atomics are needed and every output row is written exactly once (deterministic).
//
// Work decomposition.  The (row, channel-vector) space is flattened: thread t handles row = t / CV and the
// VEC-wide channel slice (t % CV) * VEC, CV = C / VEC.  Consecutive lanes therefore cover consecutive
// addresses of out[] (fully coalesced 4/8/16-B stores), the lanes of one row read one contiguous source row
// per edge, and no lane is idle whatever C is (C = 1 ... 125 on this path).  The CSR arrays are read with the
// same address by all lanes of a row (one request per row after coalescing).  Blocks are remapped so that
// every XCD walks one contiguous slice of the rows: op nodes are numbered in program order and an edge joins
// an op to the next op on the same wire, so the source rows of a tile lie within a few hundred rows of it and
// are served from that XCD's L2 after the first touch.
#include "common.hpp"

namespace mlqem {

struct AggArgs {
  const float* x; int64_t ldx;
  const int32_t* ptr; const int32_t* idx;
  const float* cscale; const float* rscale; const float* dself;
  float alpha, beta;
  const float* z; int64_t ldz;
  const float* bias;
  int act; float drop_p; uint64_t seed;
  float* out; int64_t ldo;
  int64_t N; int C; int CV;
};

template <int VEC, bool IS_MAX>
__global__ __launch_bounds__(kBlock) void csr_aggregate_kernel(const AggArgs a) {
  const unsigned blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int64_t t = (int64_t)blk * kBlock + threadIdx.x;
  const int64_t row = t / a.CV;
  if (row >= a.N) return;
  const int ch = (int)(t - row * a.CV) * VEC;

  const int beg = a.ptr[row], end = a.ptr[row + 1];
  const float* __restrict__ xc = a.x + ch;
  float acc[VEC];
  float self[VEC];
  const bool need_self = IS_MAX || a.dself != nullptr;  // Cheb-style calls have no self term: skip the row read
  if (need_self) vload<VEC>(xc + row * a.ldx, self);
#pragma unroll
  for (int v = 0; v < VEC; ++v) { if (!need_self) self[v] = 0.f; acc[v] = IS_MAX ? self[v] : 0.f; }

  int e = beg;
  // two edges per trip: both index loads, then both row loads, are in flight together
  for (; e + 1 < end; e += 2) {
    const int j0 = a.idx[e], j1 = a.idx[e + 1];
    float s0 = 1.f, s1 = 1.f;
    if (!IS_MAX && a.cscale) { s0 = a.cscale[j0]; s1 = a.cscale[j1]; }
    float r0[VEC], r1[VEC];
    vload<VEC>(xc + (int64_t)j0 * a.ldx, r0);
    vload<VEC>(xc + (int64_t)j1 * a.ldx, r1);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      if (IS_MAX) acc[v] = fmaxf(acc[v], fmaxf(r0[v], r1[v]));
      else { acc[v] = fmaf(s0, r0[v], acc[v]); acc[v] = fmaf(s1, r1[v], acc[v]); }
    }
  }
  if (e < end) {
    const int j0 = a.idx[e];
    const float s0 = (!IS_MAX && a.cscale) ? a.cscale[j0] : 1.f;
    float r0[VEC];
    vload<VEC>(xc + (int64_t)j0 * a.ldx, r0);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      if (IS_MAX) acc[v] = fmaxf(acc[v], r0[v]);
      else acc[v] = fmaf(s0, r0[v], acc[v]);
    }
  }

  float res[VEC];
  if (IS_MAX) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) res[v] = acc[v];
  } else {
    const float rs = a.rscale ? a.rscale[row] : 1.f;
    const float ds = a.dself ? a.dself[row] : 0.f;
    float zz[VEC];
    if (a.z) vload<VEC>(a.z + row * a.ldz + ch, zz);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float r = a.alpha * fmaf(ds, self[v], rs * acc[v]);
      if (a.z) r = fmaf(a.beta, zz[v], r);
      if (a.bias) r += a.bias[ch + v];
      if (a.act & 1) r = fmaxf(r, 0.f);
      if (a.drop_p > 0.f) {
        const float u = uniform01(a.seed, (uint64_t)(row * a.C + ch + v));
        r = u < a.drop_p ? 0.f : r * (1.f / (1.f - a.drop_p));
      }
      res[v] = r;
    }
  }
  vstore<VEC>(a.out + row * a.ldo + ch, res);
}

template <bool IS_MAX>
static int launch_aggregate(AggArgs a, hipStream_t stream) {
  if (a.N < 0 || a.C <= 0 || !a.x || !a.ptr || !a.out || a.ldx < a.C || a.ldo < a.C) return MLQEM_ERR_BAD_ARG;
  if (a.z && a.ldz < a.C) return MLQEM_ERR_BAD_ARG;
  if (a.N == 0) return MLQEM_OK;
  if (!a.idx) return MLQEM_ERR_BAD_ARG;
  // widest vector the shapes and base addresses allow
  int vec = 1;
  auto ok = [&](int v) {
    if (a.C % v || a.ldx % v || a.ldo % v) return false;
    if (!aligned_to(a.x, 4 * v) || !aligned_to(a.out, 4 * v)) return false;
    if (a.z && (a.ldz % v || !aligned_to(a.z, 4 * v))) return false;
    return true;
  };
  if (ok(4)) vec = 4; else if (ok(2)) vec = 2;
  a.CV = a.C / vec;
  const int64_t threads = a.N * a.CV;
  const int64_t blocks = ceil_div(threads, kBlock);
  if (blocks > 0x7fffffffLL) return MLQEM_ERR_UNSUPPORTED;
  dim3 grid((unsigned)blocks), block(kBlock);
  switch (vec) {
    case 4: hipLaunchKernelGGL((csr_aggregate_kernel<4, IS_MAX>), grid, block, 0, stream, a); break;
    case 2: hipLaunchKernelGGL((csr_aggregate_kernel<2, IS_MAX>), grid, block, 0, stream, a); break;
    default: hipLaunchKernelGGL((csr_aggregate_kernel<1, IS_MAX>), grid, block, 0, stream, a); break;
  }
  return launch_status();
}

__global__ __launch_bounds__(kBlock) void relu_dropout_bwd_kernel(const float* __restrict__ g,
                                                                  const float* __restrict__ y, float scale,
                                                                  float* __restrict__ gx, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) gx[i] = y[i] > 0.f ? g[i] * scale : 0.f;
}

}  // namespace mlqem

using namespace mlqem;

extern "C" int mlqem_csr_aggregate_f32(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* idx,
                                       const float* cscale, const float* rscale, const float* dself, float alpha,
                                       float beta, const float* z, int64_t ldz, const float* bias, int act,
                                       float drop_p, uint64_t seed, float* out, int64_t ldo, int64_t N, int C,
                                       mlqem_stream_t stream) {
  begin_launches();
  if (drop_p < 0.f || drop_p >= 1.f) return MLQEM_ERR_BAD_ARG;
  AggArgs a{x, ldx, ptr, idx, cscale, rscale, dself, alpha, beta, z, ldz, bias, act, drop_p, seed, out, ldo, N, C, 0};
  return launch_aggregate<false>(a, as_stream(stream));
}

extern "C" int mlqem_csr_segment_max_f32(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* idx,
                                         float* out, int64_t ldo, int64_t N, int C, mlqem_stream_t stream) {
  begin_launches();
  AggArgs a{x, ldx, ptr, idx, nullptr, nullptr, nullptr, 1.f, 0.f, nullptr, 0, nullptr, 0, 0.f, 0, out, ldo, N, C, 0};
  return launch_aggregate<true>(a, as_stream(stream));
}

extern "C" int mlqem_relu_dropout_bwd_f32(const float* g, const float* y, float scale, float* gx, int64_t n,
                                          mlqem_stream_t stream) {
  begin_launches();
  if (n < 0 || (n > 0 && (!g || !y || !gx))) return MLQEM_ERR_BAD_ARG;
  if (n == 0) return MLQEM_OK;
  hipLaunchKernelGGL(relu_dropout_bwd_kernel, dim3((unsigned)ceil_div(n, kBlock)), dim3(kBlock), 0, as_stream(stream),
                     g, y, scale, gx, n);
  return launch_status();
}
