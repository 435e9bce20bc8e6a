// Edge-softmax kernels of Family B (docs/tutorials/gnn.py:70-276): TransformerConv's multi-head attention and
// ASAPooling's attention-weighted cluster sum.  Both are "softmax over the in-edges of a row, then a weighted sum
// of source rows"; the softmax statistics of a row are recomputed by each thread that needs them (rows have a
// handful of in-edges), which keeps the kernels free of any [E]-sized intermediate.
#include "attn_fwd.hpp"
#include "common.hpp"

namespace mlqem {

// TransformerConv (heads=H, concat, root_weight, no edge features), inference form: no dropout, no
// statistics.  The schedule (short rows from registers, longer ones in one chunked pass) is in attn_fwd.hpp.
template <bool WIDE> __global__ __launch_bounds__(kBlock) void transformer_attn_kernel(const AttnFwdArgs a) {
  attn_forward<false, WIDE>(a);
}

// ASAPooling steps 3-4: score_e = LeakyReLU(a[dst] + c[src]), softmax over the in-edges of
// dst PLUS its own self-loop (add_remaining_self_loops), out[dst] = sum_e score_e * x[src_e].
// a[i] = att_w[:D] . lin(xq)[i] + att_b and c[j] = att_w[D:] . x[j] are per-node scalars made by the dense kernel.
// Thread = (row, channel) flattened.
__global__ __launch_bounds__(kBlock) void softmax_aggregate_kernel(const float* __restrict__ x, int64_t ldx,
                                                                   const int32_t* __restrict__ ptr,
                                                                   const int32_t* __restrict__ idx,
                                                                   const float* __restrict__ a_dst,
                                                                   const float* __restrict__ c_src, float slope,
                                                                   int64_t N, int C, float* __restrict__ out,
                                                                   int64_t ldo) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= N * C) return;
  const int64_t row = t / C;
  const int ch = (int)(t - row * C);
  const int beg = ptr[row], end = ptr[row + 1];
  const float ai = a_dst[row];
  auto leaky = [&](float v) { return v > 0.f ? v : v * slope; };
  float m = leaky(ai + c_src[row]);  // the self-loop is always there
  for_edge_chunks(beg, end, [&](int e, auto kc) {        // chunks of edges fetched together, used in edge order (common.hpp)
    constexpr int K = decltype(kc)::value;
    int jj[K];
    float cj[K];
#pragma unroll
    for (int u = 0; u < K; ++u) jj[u] = idx[e + u];
#pragma unroll
    for (int u = 0; u < K; ++u) cj[u] = c_src[jj[u]];
#pragma unroll
    for (int u = 0; u < K; ++u) m = fmaxf(m, leaky(ai + cj[u]));
  });
  float denom = 0.f, acc = 0.f;
  for_edge_chunks(beg, end, [&](int e, auto kc) {
    constexpr int K = decltype(kc)::value;
    int jj[K];
    float cj[K], xj[K];
#pragma unroll
    for (int u = 0; u < K; ++u) jj[u] = idx[e + u];
#pragma unroll
    for (int u = 0; u < K; ++u) {
      cj[u] = c_src[jj[u]];
      xj[u] = x[(int64_t)jj[u] * ldx + ch];
    }
#pragma unroll
    for (int u = 0; u < K; ++u) {
      const float p = expf(leaky(ai + cj[u]) - m);
      denom += p;
      acc = fmaf(p, xj[u], acc);
    }
  });
  {
    const float p = expf(leaky(ai + c_src[row]) - m);  // self-loop last, as appended by add_remaining_self_loops
    denom += p;
    acc = fmaf(p, x[row * ldx + ch], acc);
  }
  // PyG normalises every edge score first (p / (denom + 1e-16)) and then sums the messages
  out[row * ldo + ch] = acc / (denom + 1e-16f);
}

// LEConv(D -> 1) fitness of ASAPooling step 5 on per-node scalars pqr[N,3] = (lin1(x') , lin2(x'), lin3(x')):
// f[i] = sigmoid( sum_{e in in(i)} p[src_e] + p[i] - (deg_i + 1) * q[i] + r[i] ), edges incl. the added self-loop.
__global__ __launch_bounds__(kBlock) void leconv_fitness_kernel(const float* __restrict__ pqr,
                                                                const int32_t* __restrict__ ptr,
                                                                const int32_t* __restrict__ idx, int64_t N,
                                                                float* __restrict__ fitness) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const int beg = ptr[i], end = ptr[i + 1];
  const float qi = pqr[i * 3 + 1];
  float s = 0.f;
  for_edge_chunks(beg, end, [&](int e, auto kc) {                        // message a_j - b_i, summed in edge order
    constexpr int K = decltype(kc)::value;
    float pj[K];
#pragma unroll
    for (int u = 0; u < K; ++u) pj[u] = pqr[(int64_t)idx[e + u] * 3];
#pragma unroll
    for (int u = 0; u < K; ++u) s += pj[u] - qi;
  });
  s += pqr[i * 3] - qi;                                                 // the self-loop
  s += pqr[i * 3 + 2];
  fitness[i] = 1.0f / (1.0f + expf(-s));
}

// x_out[p,:] = x[perm[p],:] * scale[perm[p]]
__global__ __launch_bounds__(kBlock) void gather_scale_rows_kernel(const float* __restrict__ x, int64_t ldx,
                                                                   const int32_t* __restrict__ perm,
                                                                   const float* __restrict__ scale, int64_t K, int C,
                                                                   float* __restrict__ out, int64_t ldo) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= K * C) return;
  const int64_t p = t / C;
  const int c = (int)(t - p * C);
  const int64_t j = perm[p];
  out[p * ldo + c] = x[j * ldx + c] * (scale ? scale[j] : 1.f);
}

}  // namespace mlqem

using namespace mlqem;

extern "C" int mlqem_transformer_attention_f32(const float* qkvs, int64_t ld, const int32_t* in_ptr,
                                               const int32_t* in_src, const int32_t* loops, int64_t N, int H, int C,
                                               float* out, int64_t ldo, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || H <= 0 || C <= 0 || ld < 4 * H * C || ldo < H * C) return MLQEM_ERR_BAD_ARG;
  if (C > kAttnMaxC) return MLQEM_ERR_UNSUPPORTED;
  if (N == 0) return MLQEM_OK;
  if (!qkvs || !in_ptr || !out) return MLQEM_ERR_BAD_ARG;
  if (N > INT32_MAX) return MLQEM_ERR_UNSUPPORTED;
  const AttnFwdArgs a{qkvs, ld, in_ptr, in_src, loops, N, 0, H, C, 0.f, 0, out, ldo, nullptr, 0, nullptr, nullptr};
  const dim3 grid((unsigned)ceil_div(N * H * kGroup, kBlock));
  if (C > kGroup) hipLaunchKernelGGL(transformer_attn_kernel<true>, grid, dim3(kBlock), 0, as_stream(stream), a);
  else hipLaunchKernelGGL(transformer_attn_kernel<false>, grid, dim3(kBlock), 0, as_stream(stream), a);
  return launch_status();
}

extern "C" int mlqem_csr_softmax_aggregate_f32(const float* x, int64_t ldx, const int32_t* in_ptr,
                                               const int32_t* in_src, const float* a_dst, const float* c_src,
                                               float negative_slope, int64_t N, int C, float* out, int64_t ldo,
                                               mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || C <= 0 || ldx < C || ldo < C) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!x || !in_ptr || !a_dst || !c_src || !out) return MLQEM_ERR_BAD_ARG;
  hipLaunchKernelGGL(softmax_aggregate_kernel, dim3((unsigned)ceil_div(N * C, kBlock)), dim3(kBlock), 0,
                     as_stream(stream), x, ldx, in_ptr, in_src, a_dst, c_src, negative_slope, N, C, out, ldo);
  return launch_status();
}

extern "C" int mlqem_leconv_fitness_f32(const float* pqr, const int32_t* in_ptr, const int32_t* in_src, int64_t N,
                                        float* fitness, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!pqr || !in_ptr || !fitness) return MLQEM_ERR_BAD_ARG;
  hipLaunchKernelGGL(leconv_fitness_kernel, dim3((unsigned)ceil_div(N, kBlock)), dim3(kBlock), 0, as_stream(stream),
                     pqr, in_ptr, in_src, N, fitness);
  return launch_status();
}

extern "C" int mlqem_gather_scale_rows_f32(const float* x, int64_t ldx, const int32_t* perm, const float* scale,
                                           int64_t K, int C, float* out, int64_t ldo, mlqem_stream_t stream) {
  begin_launches();
  if (K < 0 || C <= 0 || ldx < C || ldo < C) return MLQEM_ERR_BAD_ARG;
  if (K == 0) return MLQEM_OK;
  if (!x || !perm || !out) return MLQEM_ERR_BAD_ARG;
  hipLaunchKernelGGL(gather_scale_rows_kernel, dim3((unsigned)ceil_div(K * C, kBlock)), dim3(kBlock), 0,
                     as_stream(stream), x, ldx, perm, scale, K, C, out, ldo);
  return launch_status();
}
