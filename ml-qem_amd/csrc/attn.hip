// Edge-softmax kernels of Family B (docs/tutorials/gnn.py:70-276): TransformerConv's multi-head attention and
// ASAPooling's attention-weighted cluster sum.  Both are "softmax over the in-edges of a row, then a weighted sum
// of source rows"; the softmax statistics of a row are recomputed by each thread that needs them (rows have a
// handful of in-edges), which keeps the kernels free of any [E]-sized intermediate.
#include "common.hpp"

namespace mlqem {

constexpr int kAttnMaxC = 32;  // channels per head held in registers (reference models: 15 and 25)
constexpr int kAttnShortRow = 6;   // in-edges (+ self) of a row whose scores are kept in registers

// TransformerConv (heads=H, concat, root_weight, no edge features).
// qkvs: [N, 4*H*C] = [query | key | value | skip] as produced by one fused projection.
// One 16-lane group = (row, head); lane l holds channels l and l + 16 (C <= 32), so a key/value row segment is one
// coalesced 64-byte read and q.k is a cross-lane sum.  Edge order: the row's CSR entries, then its self-loop(s) -- the
// order PyG's scatter sees.
__global__ __launch_bounds__(kBlock) void transformer_attn_kernel(const float* __restrict__ qkvs, int64_t ld,
                                                                  const int32_t* __restrict__ ptr,
                                                                  const int32_t* __restrict__ idx,
                                                                  const int32_t* __restrict__ loops, int64_t N, int H,
                                                                  int C, float* __restrict__ out, int64_t ldo) {
  const int64_t t = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / kGroup;
  const int l = threadIdx.x % kGroup;
  if (t >= N * H) return;                       // a whole group leaves together
  const int64_t row = t / H;
  const int h = (int)(t - row * H);
  const int HC = H * C;
  const float scale = 1.0f / sqrtf((float)C);
  const bool c0 = l < C, c1 = l + kGroup < C;
  const float* __restrict__ qi = qkvs + row * ld + h * C;
  const float q0 = c0 ? qi[l] : 0.f, q1 = c1 ? qi[l + kGroup] : 0.f;

  const int beg = ptr[row], end = ptr[row + 1];
  const int n_self = loops ? loops[row] : 0;
  auto score = [&](int64_t j) {
    const float* __restrict__ kj = qkvs + j * ld + HC + h * C;
    float s = q0 * (c0 ? kj[l] : 0.f);
    if (c1) s = fmaf(q1, kj[l + kGroup], s);
    return group16_sum(s) * scale;
  };
  float m = -INFINITY;
  float a0 = 0.f, a1 = 0.f, denom = 0.f;
  auto add_p = [&](int64_t j, float p) {
    denom += p;
    const float* __restrict__ vj = qkvs + j * ld + 2 * HC + h * C;
    if (c0) a0 = fmaf(p, vj[l], a0);
    if (c1) a1 = fmaf(p, vj[l + kGroup], a1);
  };
  const int deg = end - beg;
  const int cnt = deg + (n_self > 0 ? 1 : 0);
  if (cnt <= kAttnShortRow) {
    // short rows: source ids, then all key rows, then all value rows fetched together; every score computed once and
    // kept in registers (same expressions, same order as the general path: bit-identical)
    int64_t jj[kAttnShortRow];
    float sc[kAttnShortRow];
#pragma unroll
    for (int e = 0; e < kAttnShortRow; ++e) jj[e] = e < deg ? (int64_t)idx[beg + e] : row;
#pragma unroll
    for (int e = 0; e < kAttnShortRow; ++e) sc[e] = e < cnt ? score(jj[e]) : -INFINITY;
#pragma unroll
    for (int e = 0; e < kAttnShortRow; ++e) if (e < cnt) m = fmaxf(m, sc[e]);
#pragma unroll
    for (int e = 0; e < kAttnShortRow; ++e) if (e < deg) add_p(jj[e], expf(sc[e] - m) * 1.f);
#pragma unroll
    for (int e = 0; e < kAttnShortRow; ++e) if (e == deg && n_self > 0) add_p(row, expf(sc[e] - m) * (float)n_self);
  } else {
    // long rows in chunks (for_edge_chunks, common.hpp): same expressions in the same order as one edge at a time
    // pass 1: segment max
    for_edge_chunks(beg, end, [&](int e, auto kc) {
      constexpr int K = decltype(kc)::value;
      int64_t jj[K];
      float k0[K], k1[K];
#pragma unroll
      for (int u = 0; u < K; ++u) jj[u] = idx[e + u];
#pragma unroll
      for (int u = 0; u < K; ++u) {
        const float* __restrict__ kj = qkvs + jj[u] * ld + HC + h * C;
        k0[u] = c0 ? kj[l] : 0.f;
        k1[u] = c1 ? kj[l + kGroup] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < K; ++u) {
        float s = q0 * k0[u];
        if (c1) s = fmaf(q1, k1[u], s);
        m = fmaxf(m, group16_sum(s) * scale);
      }
    });
    if (n_self > 0) m = fmaxf(m, score(row));
    // pass 2: exp, sum, weighted value sum
    for_edge_chunks(beg, end, [&](int e, auto kc) {
      constexpr int K = decltype(kc)::value;
      int64_t jj[K];
      float k0[K], k1[K], v0[K], v1[K];
#pragma unroll
      for (int u = 0; u < K; ++u) jj[u] = idx[e + u];
#pragma unroll
      for (int u = 0; u < K; ++u) {
        const float* __restrict__ kj = qkvs + jj[u] * ld + HC + h * C;
        const float* __restrict__ vj = qkvs + jj[u] * ld + 2 * HC + h * C;
        k0[u] = c0 ? kj[l] : 0.f;
        k1[u] = c1 ? kj[l + kGroup] : 0.f;
        v0[u] = c0 ? vj[l] : 0.f;
        v1[u] = c1 ? vj[l + kGroup] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < K; ++u) {
        float s = q0 * k0[u];
        if (c1) s = fmaf(q1, k1[u], s);
        const float p = expf(group16_sum(s) * scale - m) * 1.f;
        denom += p;
        if (c0) a0 = fmaf(p, v0[u], a0);
        if (c1) a1 = fmaf(p, v1[u], a1);
      }
    });
    if (n_self > 0) add_p(row, expf(score(row) - m) * (float)n_self);
  }
  const float inv = 1.0f / (denom + 1e-16f);
  const float* __restrict__ skip = qkvs + row * ld + 3 * HC + h * C;
  float* __restrict__ o = out + row * ldo + h * C;
  if (c0) o[l] = a0 * inv + skip[l];
  if (c1) o[l + kGroup] = a1 * inv + skip[l + kGroup];
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
  hipLaunchKernelGGL(transformer_attn_kernel, dim3((unsigned)ceil_div(N * H * kGroup, kBlock)), dim3(kBlock), 0,
                     as_stream(stream), qkvs, ld, in_ptr, in_src, loops, N, H, C, out, ldo);
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
