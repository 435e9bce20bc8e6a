// Edge-softmax kernels of Family B (docs/tutorials/gnn.py:70-276): TransformerConv's multi-head attention and
// ASAPooling's attention-weighted cluster sum.  Both are "softmax over the in-edges of a row, then a weighted sum
// of source rows"; the softmax statistics of a row are recomputed by each thread that needs them (rows have a
// handful of in-edges), which keeps the kernels free of any [E]-sized intermediate.
#include "attn_q4.hpp"
#include "common.hpp"

namespace mlqem {

// TransformerConv (heads=H, concat, root_weight, no edge features), inference form: no dropout, no
// statistics.  The schedule (short rows from registers, longer ones in one chunked pass) is in attn_fwd.hpp.
template <int LPH, bool FAST> __global__ __launch_bounds__(kBlock) void transformer_attn_q4_kernel(const AttnFwdArgs a) {
  attn_forward_q4<false, LPH, FAST>(a);
}

template <bool WIDE> __global__ __launch_bounds__(kBlock) void transformer_attn_kernel(const AttnFwdArgs a) {
  attn_forward<false, WIDE>(a);
}

// ASAPooling steps 3-4: score_e = LeakyReLU(a[dst] + c[src]), softmax over the in-edges of
// dst PLUS its own self-loop (add_remaining_self_loops), out[dst] = sum_e score_e * x[src_e].
// a[i] = att_w[:D] . lin(xq)[i] + att_b and c[j] = att_w[D:] . x[j] are per-node scalars made by the dense kernel.
//
// One 16-lane group per row; lane l holds channels l, l + 16, ... (NV per lane).  The scalar work of an edge -- its
// source id, c[src], LeakyReLU, exp -- is done by ONE lane: a chunk is 16 edges, lane u owns edge u, and the weight and
// the source id reach the other lanes by a DPP row broadcast when the rows are accumulated.  (The thread-per-(row,
// channel) form repeated that scalar work in every one of a row's 30-45 threads, twice: it spent its time on exp.)
// One pass: a running maximum that grows rescales what was summed under the old one.
template <int NV> __global__ __launch_bounds__(kBlock) void softmax_aggregate_kernel(
    const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
    const float* __restrict__ a_dst, const float* __restrict__ c_src, float slope, int64_t N, int C,
    float* __restrict__ out, int64_t ldo, const uint8_t* __restrict__ skip) {
  const int64_t row = ((int64_t)row_block() * kBlock + threadIdx.x) / kGroup;
  const int l = threadIdx.x % kGroup;
  if (row >= N) return;                              // a whole group leaves together
  if (skip && skip[row]) return;                     // a row a dense block serves (dense_pool.hip)
  const int beg = ptr[row], end = ptr[row + 1];
  const float ai = a_dst[row];
  auto leaky = [&](float v) { return v > 0.f ? v : v * slope; };
  bool has[NV];
  int col[NV];                                       // (every load unconditional: common.hpp slice_columns)
  slice_columns<NV>(l, C, has, col);
  float own[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) own[v] = x[row * ldx + col[v]];
  const float c_own = c_src[row];
  float m = -INFINITY, den = 0.f, acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = 0.f;
  auto grow = [&](float cm) {
    if (cm > m) {
      const float r = expf(m - cm);                  // exp(-inf) = 0 the first time
      den *= r;
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[v] *= r;
      m = cm;
    }
  };
  for (int e0 = beg; e0 < end; e0 += kGroup) {
    const int k = min(kGroup, end - e0);              // group-uniform
    const int j = idx[e0 + min(l, k - 1)];            // lane u: edge e0 + u (lanes past k repeat the last edge, weight 0)
    const float cj = c_src[j];
    const float s = l < k ? leaky(ai + cj) : -INFINITY;
    grow(group16_max(s));
    const float p = l < k ? expf(s - m) : 0.f;
    den += group16_sum(p);
    // CNT source rows in flight, then CNT multiply-adds in edge order.  CNT = 2 serves the rows of a circuit DAG (one or two
    // in-edges: nine of ten rows of the first pooling) without issuing the other six row loads of an eight-chunk -- these kernels are
    // bound by instruction issue there, and a skipped slot is a skipped instruction only when the whole chunk form is smaller
    auto rows = [&](auto first, auto count) {
      constexpr int U0 = decltype(first)::value, CNT = decltype(count)::value;
      int ju[CNT];
      float pu[CNT], xv[CNT][NV];
      ju[0] = group16_bcast<U0 + 0>(j); ju[1] = group16_bcast<U0 + 1>(j);
      pu[0] = group16_bcast<U0 + 0>(p); pu[1] = group16_bcast<U0 + 1>(p);
      if constexpr (CNT == 8) {
        ju[2] = group16_bcast<U0 + 2>(j); ju[3] = group16_bcast<U0 + 3>(j); ju[4] = group16_bcast<U0 + 4>(j);
        ju[5] = group16_bcast<U0 + 5>(j); ju[6] = group16_bcast<U0 + 6>(j); ju[7] = group16_bcast<U0 + 7>(j);
        pu[2] = group16_bcast<U0 + 2>(p); pu[3] = group16_bcast<U0 + 3>(p); pu[4] = group16_bcast<U0 + 4>(p);
        pu[5] = group16_bcast<U0 + 5>(p); pu[6] = group16_bcast<U0 + 6>(p); pu[7] = group16_bcast<U0 + 7>(p);
      }
#pragma unroll
      for (int u = 0; u < CNT; ++u) {
        const float* __restrict__ xj = x + (int64_t)ju[u] * ldx;
#pragma unroll
        for (int v = 0; v < NV; ++v) xv[u][v] = xj[col[v]];
      }
#pragma unroll
      for (int u = 0; u < CNT; ++u)
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = fmaf(pu[u], xv[u][v], acc[v]);      // (a lane without the channel: never stored)
    };
    if (k <= 2) rows(EdgeChunk<0>{}, EdgeChunk<2>{});
    else {
      rows(EdgeChunk<0>{}, EdgeChunk<8>{});
      if (k > 8) rows(EdgeChunk<8>{}, EdgeChunk<8>{});
    }
  }
  {  // the self-loop last, as appended by add_remaining_self_loops
    const float s = leaky(ai + c_own);
    grow(s);
    const float p = expf(s - m);
    den += p;
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = fmaf(p, own[v], acc[v]);
  }
  // PyG normalises every edge score first (p / (denom + 1e-16)) and then sums the messages: the same value, one division
  const float inv = 1.0f / (den + 1e-16f);
  float* __restrict__ o = out + row * ldo + l;
#pragma unroll
  for (int v = 0; v < NV; ++v) if (has[v]) o[v * kGroup] = acc[v] * inv;
}

// Rows wider than 128 channels (no reference model has them): thread = (row, channel) flattened, every thread walks its row.
__global__ __launch_bounds__(kBlock) void softmax_aggregate_any_width_kernel(const float* __restrict__ x, int64_t ldx,
                                                                   const int32_t* __restrict__ ptr,
                                                                   const int32_t* __restrict__ idx,
                                                                   const float* __restrict__ a_dst,
                                                                   const float* __restrict__ c_src, float slope,
                                                                   int64_t N, int C, float* __restrict__ out,
                                                                   int64_t ldo) {
  const int64_t t = (int64_t)row_block() * kBlock + threadIdx.x;
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
  const int64_t i = (int64_t)row_block() * kBlock + threadIdx.x;
  if (i >= N) return;
  const int beg = ptr[i], end = ptr[i + 1];
  const float qi = pqr[i * 3 + 1];
  float s = 0.f;
  // (round 6: rows of more than 16 entries handed to their wave, 64 entries per round trip, one row after the other: 124 against 90 us
  // on the two poolings of 64 100-qubit circuits -- in the coarsened graph most rows are such rows, and a wave then walks dozens of them)
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

// The same for graphs of LONG rows (a coarsened graph: 28 entries per row on average, hundreds around a circuit's barriers): a 16-lane
// group per row, 32 entries per round trip read as two coalesced 64-byte pieces of the index list, the sum by the group's DPP adds.
// (A thread walking such a row alone reads 8 entries per dependent round trip from addresses of its own: 85 us for the 353 k rows of 64
// pooled 100-qubit circuits.  Handing only the rows of > 16 entries to their WAVE, one after the other, was slower: 124 us.)
__global__ __launch_bounds__(kBlock) void leconv_fitness_rows_kernel(const float* __restrict__ pqr, const int32_t* __restrict__ ptr,
                                                                     const int32_t* __restrict__ idx, int64_t N, float* __restrict__ fitness) {
  const int64_t i = ((int64_t)row_block() * kBlock + threadIdx.x) / kGroup;
  const int l = threadIdx.x % kGroup;
  if (i >= N) return;
  const int beg = ptr[i], end = ptr[i + 1];
  const float pi = pqr[i * 3], qi = pqr[i * 3 + 1], ri = pqr[i * 3 + 2];
  float s = 0.f;
  for (int e0 = beg; e0 < end; e0 += 2 * kGroup) {
    const int ea = e0 + l, eb = ea + kGroup;
    const int j0 = idx[min(ea, end - 1)], j1 = idx[min(eb, end - 1)];
    const float p0 = pqr[(int64_t)j0 * 3], p1 = pqr[(int64_t)j1 * 3];
    s += (ea < end ? p0 : 0.f) + (eb < end ? p1 : 0.f);
  }
  s = group16_sum(s) - (float)(end - beg) * qi;
  s += pi - qi;                                                         // the self-loop
  s += ri;
  if (l == 0) fitness[i] = 1.0f / (1.0f + expf(-s));
}

// x_out[p,:] = x[perm[p],:] * scale[perm[p]]
__global__ __launch_bounds__(kBlock) void gather_scale_rows_kernel(const float* __restrict__ x, int64_t ldx,
                                                                   const int32_t* __restrict__ perm,
                                                                   const float* __restrict__ scale, int64_t K, int C,
                                                                   float* __restrict__ out, int64_t ldo) {
  const int64_t t = (int64_t)row_block() * kBlock + threadIdx.x;
  if (t >= K * C) return;
  int c;
  const int64_t p = split_index(t, C, c);
  const int64_t j = perm[p];
  out[p * ldo + c] = x[j * ldx + c] * (scale ? scale[j] : 1.f);
}

// ASAPooling's forward up to the fitness projections in ONE pass over the rows (round 5), for graphs of SHORT rows (circuit DAGs: one
// or two in-edges per node but for barriers): per row i the segment max over N(i) + {i}, the composed score a_i = w_comp . max + b_comp,
// c_j = att_x . x_j of every entry (and c_i, stored for the backward), the softmax of LeakyReLU(a_i + c_j) over the entries, the
// cluster sum x'_i and pqr_i = W3 x'_i + b3.  Was: csr_aggregate<IS_MAX>, a [N,D]x[D,1] GEMM, another, softmax_aggregate_kernel, a
// [N,D]x[D,3] GEMM -- five passes that read x, the maxima or x' from memory (294 us on the circuit DAGs of 64 100-qubit circuits).
// A 16-lane group per row, lane l channels l, l + 16, ... (NV per lane); rows of at most two in-edges keep the gathered rows in
// registers between the maximum and the sum, longer rows gather twice (the second time from cache).
// (Round 6: every load unconditional -- a lane without a channel in slice v reads the row's last channel and masks the value:
// `has[v] ? p[..] : 0` compiles to a branch around each load; see softmax_aggregate_bwd_src_kernel, family_b_bwd.hip.)
template <int NV> __global__ __launch_bounds__(kBlock) void asap_scores_fused_kernel(
    const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
    const float* __restrict__ w_comp, const float* __restrict__ b_comp, const float* __restrict__ att_x, const float* __restrict__ w3,
    const float* __restrict__ b3, float slope, int64_t N, int C, float* __restrict__ xmax, int64_t ldm, float* __restrict__ a_dst,
    float* __restrict__ c_src, float* __restrict__ xnew, int64_t ldn, float* __restrict__ pqr) {
  const int64_t row = ((int64_t)row_block() * kBlock + threadIdx.x) / kGroup;
  const int l = threadIdx.x % kGroup;
  if (row >= N) return;
  auto leaky = [&](float v) { return v > 0.f ? v : v * slope; };
  bool has[NV];
  int col[NV];                                             // the lane's channel of slice v, or the last channel (loaded, then masked)
  slice_columns<NV>(l, C, has, col);
  float wc[NV], ax[NV], own[NV], w3v[3][NV];
  const int beg = ptr[row], deg = ptr[row + 1] - beg;
  auto masked = [&](float t, int v) { return has[v] ? t : 0.f; };
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    wc[v] = masked(w_comp[col[v]], v);
    ax[v] = masked(att_x[col[v]], v);
    own[v] = masked(x[row * ldx + col[v]], v);
#pragma unroll
    for (int t = 0; t < 3; ++t) w3v[t][v] = masked(w3[t * C + col[v]], v);
  }
  auto dot = [&](const float (&a)[NV], const float (&b)[NV]) {
    float d = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) d = fmaf(a[v], b[v], d);
    return group16_sum(d);
  };
  const float c_own = dot(ax, own);
  float mx[NV], acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) mx[v] = own[v];
  float a_i, m, den;
  if (deg <= 2) {                                          // (group-uniform) the gathered rows stay in registers
    float xs[2][NV], cs[2] = {0.f, 0.f};
    const int j0 = idx[deg > 0 ? beg : 0], j1 = idx[deg > 1 ? beg + 1 : 0];      // (entry 0 exists in every index array: allocated with >= 1)
    const int jj[2] = {deg > 0 ? j0 : (int)row, deg > 1 ? j1 : (int)row};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int v = 0; v < NV; ++v) xs[u][v] = masked(x[(int64_t)jj[u] * ldx + col[v]], v);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int v = 0; v < NV; ++v)
        if (u < deg) mx[v] = fmaxf(mx[v], xs[u][v]);
      cs[u] = dot(ax, xs[u]);
    }
    a_i = dot(wc, mx) + b_comp[0];
    const float s_own = leaky(a_i + c_own), s0 = deg > 0 ? leaky(a_i + cs[0]) : -INFINITY, s1 = deg > 1 ? leaky(a_i + cs[1]) : -INFINITY;
    m = fmaxf(s_own, fmaxf(s0, s1));
    const float p_own = expf(s_own - m), p0 = deg > 0 ? expf(s0 - m) : 0.f, p1 = deg > 1 ? expf(s1 - m) : 0.f;
    den = p0 + p1 + p_own;                                 // entries first, the self-loop last (add_remaining_self_loops)
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = fmaf(p_own, own[v], fmaf(p1, xs[1][v], p0 * xs[0][v]));
  } else {
    const int end = beg + deg;
    // eight (four: wide rows) gathered rows in flight at a time (one row per trip was a dependent round trip per entry: a barrier's
    // 100 entries took 300 us, the whole launch's duration)
    constexpr int kFly = NV >= 3 ? 4 : 8;
    for (int e0 = beg; e0 < end; e0 += kGroup) {           // the maximum
      const int k = min(kGroup, end - e0);
      const int j = idx[e0 + min(l, k - 1)];
      for (int u0 = 0; u0 < k; u0 += kFly) {
        float xj[kFly][NV];
#pragma unroll
        for (int u = 0; u < kFly; ++u) {
          const int ju = __shfl(j, min(u0 + u, k - 1), kGroup);
#pragma unroll
          for (int v = 0; v < NV; ++v) xj[u][v] = masked(x[(int64_t)ju * ldx + col[v]], v);
        }
#pragma unroll
        for (int u = 0; u < kFly; ++u)
#pragma unroll
          for (int v = 0; v < NV; ++v) mx[v] = fmaxf(mx[v], xj[u][v]);       // (past the end: the last entry again)
      }
    }
    a_i = dot(wc, mx) + b_comp[0];
    m = -INFINITY; den = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0.f;
    auto grow = [&](float cm) {
      if (cm > m) {
        const float r = expf(m - cm);
        den *= r;
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] *= r;
        m = cm;
      }
    };
    for (int e0 = beg; e0 < end; e0 += kGroup) {           // scores, weights, the sum (rows from cache), in entry order
      const int k = min(kGroup, end - e0);
      const int j = idx[e0 + min(l, k - 1)];
      for (int u0 = 0; u0 < k; u0 += kFly) {
        float xj[kFly][NV];
#pragma unroll
        for (int u = 0; u < kFly; ++u) {
          const int ju = __shfl(j, min(u0 + u, k - 1), kGroup);
#pragma unroll
          for (int v = 0; v < NV; ++v) xj[u][v] = masked(x[(int64_t)ju * ldx + col[v]], v);
        }
#pragma unroll
        for (int u = 0; u < kFly; ++u) {
          if (u0 + u < k) {                                // (group-uniform)
            const float sj = leaky(a_i + dot(ax, xj[u]));
            grow(sj);
            const float pj = expf(sj - m);
            den += pj;
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[v] = fmaf(pj, xj[u][v], acc[v]);
          }
        }
      }
    }
    const float s_own = leaky(a_i + c_own);
    grow(s_own);
    const float p_own = expf(s_own - m);
    den += p_own;
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = fmaf(p_own, own[v], acc[v]);
  }
  const float inv = 1.0f / (den + 1e-16f);
  float xn[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    xn[v] = acc[v] * inv;
    if (has[v]) {
      xmax[row * ldm + l + v * kGroup] = mx[v];
      xnew[row * ldn + l + v * kGroup] = xn[v];
    }
  }
  float out3[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) out3[t] = dot(w3v[t], xn) + b3[t];
  if (l == 0) {
    a_dst[row] = a_i;
    c_src[row] = c_own;
    pqr[row * 3 + 0] = out3[0]; pqr[row * 3 + 1] = out3[1]; pqr[row * 3 + 2] = out3[2];
  }
}

// A projection's weight [G C, I] and bias [G C] with every group of C rows spread to a pitch of CP rows (zero rows between), and back
// (the gradients' real rows): what gives q / k / v / skip a head pitch of 16 floats.  One launch each way (was four / two
// element-wise launches of the host framework per TransformerConv and step).
__global__ __launch_bounds__(kBlock) void pad_head_rows_kernel(const float* __restrict__ w, const float* __restrict__ b, int G, int C, int CP,
                                                               int I, float* __restrict__ wp, float* __restrict__ bp) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t total = (int64_t)G * CP * (I + 1);       // column I of a row: its bias
  if (t >= total) return;
  const int col = (int)(t % (I + 1));
  const int64_t row = t / (I + 1);
  const int g = (int)(row / CP), c = (int)(row % CP);
  const bool real = c < C;
  const int64_t src = (int64_t)g * C + c;
  if (col < I) wp[row * I + col] = real ? w[src * I + col] : 0.f;
  else if (bp) bp[row] = (real && b) ? b[src] : 0.f;
}
// The same from up to four SEPARATE weight matrices (query, key, value, skip: the parameters of a TransformerConv as the reference's
// state dict keeps them), part k covering the groups [k GP, (k + 1) GP): the two torch.cat launches in front of the padding go, and
// with CP == C this is the concatenation itself.
struct PadParts { const float* w[4]; const float* b[4]; };
__global__ __launch_bounds__(kBlock) void pad_head_rows_parts_kernel(const PadParts q, int parts, int GP, int C, int CP, int I,
                                                                     float* __restrict__ wp, float* __restrict__ bp) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t total = (int64_t)parts * GP * CP * (I + 1);       // column I of a row: its bias
  if (t >= total) return;
  const int col = (int)(t % (I + 1));
  const int64_t row = t / (I + 1);
  const int g = (int)(row / CP), c = (int)(row % CP);
  const int part = g / GP, gl = g - part * GP;
  const bool real = c < C;
  const int64_t src = (int64_t)gl * C + c;
  const float* __restrict__ w = part == 0 ? q.w[0] : part == 1 ? q.w[1] : part == 2 ? q.w[2] : q.w[3];
  const float* __restrict__ b = part == 0 ? q.b[0] : part == 1 ? q.b[1] : part == 2 ? q.b[2] : q.b[3];
  if (col < I) wp[row * I + col] = real ? w[src * I + col] : 0.f;
  else if (bp) bp[row] = (real && b) ? b[src] : 0.f;
}
__global__ __launch_bounds__(kBlock) void unpad_head_rows_kernel(const float* __restrict__ gwp, const float* __restrict__ gbp, int G, int C,
                                                                 int CP, int I, float* __restrict__ gw, float* __restrict__ gb) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t total = (int64_t)G * C * (I + 1);
  if (t >= total) return;
  const int col = (int)(t % (I + 1));
  const int64_t row = t / (I + 1);
  const int g = (int)(row / C), c = (int)(row % C);
  const int64_t src = (int64_t)g * CP + c;
  if (col < I) gw[row * I + col] = gwp[src * I + col];
  else if (gb && gbp) gb[row] = gbp[src];
}

}  // namespace mlqem

using namespace mlqem;

extern "C" int mlqem_asap_scores_fused_f32(const float* x, int64_t ldx, const int32_t* in_ptr, const int32_t* in_src, const float* w_comp,
                                           const float* b_comp, const float* att_x, const float* w3, const float* b3, float negative_slope,
                                           int64_t N, int C, float* xmax, int64_t ldm, float* a_dst, float* c_src, float* xnew, int64_t ldn,
                                           float* pqr, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || C <= 0 || ldx < C || ldm < C || ldn < C) return MLQEM_ERR_BAD_ARG;
  if (C > 64) return MLQEM_ERR_UNSUPPORTED;
  if (N == 0) return MLQEM_OK;
  if (!x || !in_ptr || !in_src || !w_comp || !b_comp || !att_x || !w3 || !b3 || !xmax || !a_dst || !c_src || !xnew || !pqr) return MLQEM_ERR_BAD_ARG;
  const dim3 grid((unsigned)ceil_div(N * kGroup, kBlock));
#define MLQEM_AF(NV) hipLaunchKernelGGL(asap_scores_fused_kernel<NV>, grid, dim3(kBlock), 0, as_stream(stream), x, ldx, in_ptr, in_src, w_comp, \
                                        b_comp, att_x, w3, b3, negative_slope, N, C, xmax, ldm, a_dst, c_src, xnew, ldn, pqr)
  if (C <= 16) MLQEM_AF(1);
  else if (C <= 32) MLQEM_AF(2);
  else if (C <= 48) MLQEM_AF(3);
  else MLQEM_AF(4);
#undef MLQEM_AF
  return launch_status();
}

extern "C" int mlqem_pad_head_rows_f32(const float* w, const float* b, int groups, int channels, int pitch, int cols, float* w_padded,
                                       float* b_padded, mlqem_stream_t stream) {
  begin_launches();
  if (groups <= 0 || channels <= 0 || pitch < channels || cols <= 0) return MLQEM_ERR_BAD_ARG;
  if (!w || !w_padded) return MLQEM_ERR_BAD_ARG;
  const int64_t total = (int64_t)groups * pitch * (cols + 1);
  hipLaunchKernelGGL(pad_head_rows_kernel, dim3((unsigned)ceil_div(total, kBlock)), dim3(kBlock), 0, as_stream(stream), w, b, groups, channels,
                     pitch, cols, w_padded, b_padded);
  return launch_status();
}

extern "C" int mlqem_pad_head_rows_parts_f32(const float* const* w, const float* const* b, int parts, int groups_per_part, int channels,
                                             int pitch, int cols, float* w_padded, float* b_padded, mlqem_stream_t stream) {
  begin_launches();
  if (parts < 1 || parts > 4 || groups_per_part <= 0 || channels <= 0 || pitch < channels || cols <= 0 || !w || !w_padded) return MLQEM_ERR_BAD_ARG;
  PadParts q{};
  for (int k = 0; k < parts; ++k) {
    if (!w[k]) return MLQEM_ERR_BAD_ARG;
    q.w[k] = w[k];
    q.b[k] = b ? b[k] : nullptr;
  }
  const int64_t total = (int64_t)parts * groups_per_part * pitch * (cols + 1);
  hipLaunchKernelGGL(pad_head_rows_parts_kernel, dim3((unsigned)ceil_div(total, kBlock)), dim3(kBlock), 0, as_stream(stream), q, parts,
                     groups_per_part, channels, pitch, cols, w_padded, b_padded);
  return launch_status();
}

extern "C" int mlqem_unpad_head_rows_f32(const float* gw_padded, const float* gb_padded, int groups, int channels, int pitch, int cols,
                                         float* gw, float* gb, mlqem_stream_t stream) {
  begin_launches();
  if (groups <= 0 || channels <= 0 || pitch < channels || cols <= 0) return MLQEM_ERR_BAD_ARG;
  if (!gw_padded || !gw) return MLQEM_ERR_BAD_ARG;
  const int64_t total = (int64_t)groups * channels * (cols + 1);
  hipLaunchKernelGGL(unpad_head_rows_kernel, dim3((unsigned)ceil_div(total, kBlock)), dim3(kBlock), 0, as_stream(stream), gw_padded, gb_padded,
                     groups, channels, pitch, cols, gw, gb);
  return launch_status();
}

extern "C" int mlqem_transformer_attention_f32(const float* qkvs, int64_t ld, const int32_t* in_ptr,
                                               const int32_t* in_src, const int32_t* loops, int64_t N, int H, int C,
                                               float* out, int64_t ldo, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || H <= 0 || C <= 0 || ld < 4 * H * C || ldo < H * C) return MLQEM_ERR_BAD_ARG;
  if (C > kAttnMaxC) return MLQEM_ERR_UNSUPPORTED;
  if (N == 0) return MLQEM_OK;
  if (!qkvs || !in_ptr || !out) return MLQEM_ERR_BAD_ARG;
  if (N > INT32_MAX) return MLQEM_ERR_UNSUPPORTED;
  const AttnFwdArgs a{qkvs, ld, in_ptr, in_src, loops, N, 0, H, C, 0.f, 0, nullptr, out, ldo, nullptr, 0, nullptr, nullptr, 0, nullptr, 0};
  if (attn_q4_enabled()) {
    const int lph = C > 16 ? 8 : 4;
    const dim3 grid4((unsigned)ceil_div(N * H * lph, kBlock));
    const bool fast = attn_q4_fast(a.H, a.C, a.CP > 0 ? a.CP : a.C, INT64_MAX, a.idx);
    if (lph == 8 && fast) hipLaunchKernelGGL((transformer_attn_q4_kernel<8, true>), grid4, dim3(kBlock), 0, as_stream(stream), a);
    else if (lph == 8) hipLaunchKernelGGL((transformer_attn_q4_kernel<8, false>), grid4, dim3(kBlock), 0, as_stream(stream), a);
    else if (fast) hipLaunchKernelGGL((transformer_attn_q4_kernel<4, true>), grid4, dim3(kBlock), 0, as_stream(stream), a);
    else hipLaunchKernelGGL((transformer_attn_q4_kernel<4, false>), grid4, dim3(kBlock), 0, as_stream(stream), a);
    return launch_status();
  }
  const dim3 grid((unsigned)ceil_div(N * H * kGroup, kBlock));
  if (C > kGroup) hipLaunchKernelGGL(transformer_attn_kernel<true>, grid, dim3(kBlock), 0, as_stream(stream), a);
  else hipLaunchKernelGGL(transformer_attn_kernel<false>, grid, dim3(kBlock), 0, as_stream(stream), a);
  return launch_status();
}

namespace mlqem {
// (also for dense_pool.hip: the rows its blocks do not serve)
void launch_softmax_aggregate(const float* x, int64_t ldx, const int32_t* in_ptr, const int32_t* in_src, const float* a_dst, const float* c_src,
                              float negative_slope, int64_t N, int C, float* out, int64_t ldo, const uint8_t* skip, hipStream_t stream) {
  const dim3 grid((unsigned)ceil_div(N * kGroup, kBlock));
#define MLQEM_SA(NV) hipLaunchKernelGGL(softmax_aggregate_kernel<NV>, grid, dim3(kBlock), 0, stream, x, ldx, in_ptr, in_src, a_dst, c_src, \
                                        negative_slope, N, C, out, ldo, skip)
  if (C <= 16) MLQEM_SA(1);
  else if (C <= 32) MLQEM_SA(2);
  else if (C <= 48) MLQEM_SA(3);
  else if (C <= 64) MLQEM_SA(4);
  else if (C <= 128) MLQEM_SA(8);
  else
    hipLaunchKernelGGL(softmax_aggregate_any_width_kernel, dim3((unsigned)ceil_div(N * C, kBlock)), dim3(kBlock), 0, stream, x, ldx, in_ptr,
                       in_src, a_dst, c_src, negative_slope, N, C, out, ldo);
#undef MLQEM_SA
}
}  // namespace mlqem

extern "C" int mlqem_csr_softmax_aggregate_f32(const float* x, int64_t ldx, const int32_t* in_ptr,
                                               const int32_t* in_src, const float* a_dst, const float* c_src,
                                               float negative_slope, int64_t N, int C, float* out, int64_t ldo,
                                               mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || C <= 0 || ldx < C || ldo < C) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!x || !in_ptr || !a_dst || !c_src || !out) return MLQEM_ERR_BAD_ARG;
  launch_softmax_aggregate(x, ldx, in_ptr, in_src, a_dst, c_src, negative_slope, N, C, out, ldo, nullptr, as_stream(stream));
  return launch_status();
}

extern "C" int mlqem_leconv_fitness_f32(const float* pqr, const int32_t* in_ptr, const int32_t* in_src, int64_t N,
                                        float* fitness, int long_rows, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!pqr || !in_ptr || !fitness) return MLQEM_ERR_BAD_ARG;
  if (long_rows && in_src) {
    hipLaunchKernelGGL(leconv_fitness_rows_kernel, dim3((unsigned)ceil_div(N * kGroup, kBlock)), dim3(kBlock), 0, as_stream(stream), pqr, in_ptr,
                       in_src, N, fitness);
    return launch_status();
  }
  hipLaunchKernelGGL(leconv_fitness_kernel, dim3((unsigned)ceil_div(N, kBlock)), dim3(kBlock), 0, as_stream(stream),
                     pqr, in_ptr, in_src, N, fitness);
  return launch_status();
}

namespace mlqem {
// ... rows in the padded layout: a thread per (row, 16-byte slice) -- one index, one scale and one vector load per four outputs
// (the per-element form above: an index, a scale, a division of the flat index and a 4-byte load per output)
__global__ __launch_bounds__(kBlock) void gather_scale_rows_v4_kernel(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ perm,
                                                                      const float* __restrict__ scale, int64_t K, int CV,
                                                                      float* __restrict__ out, int64_t ldo) {
  const int64_t t = (int64_t)row_block() * kBlock + threadIdx.x;
  if (t >= K * CV) return;
  int sl;
  const int64_t p = split_index(t, CV, sl);
  const int64_t j = perm[p];
  const float f = scale ? scale[j] : 1.f;
  const f4u v = *reinterpret_cast<const f4u*>(x + j * ldx + 4 * sl);
  *reinterpret_cast<f4u*>(out + p * ldo + 4 * sl) = v * f;      // (pads: scratch in, scratch out)
}
}  // namespace mlqem

extern "C" int mlqem_gather_scale_rows_f32(const float* x, int64_t ldx, const int32_t* perm, const float* scale,
                                           int64_t K, int C, float* out, int64_t ldo, mlqem_stream_t stream) {
  begin_launches();
  if (K < 0 || C <= 0 || ldx < C || ldo < C) return MLQEM_ERR_BAD_ARG;
  if (K == 0) return MLQEM_OK;
  if (!x || !perm || !out) return MLQEM_ERR_BAD_ARG;
  const int c4 = (C + 3) / 4 * 4;
  if (ldx % 4 == 0 && ldo % 4 == 0 && ldx >= c4 && ldo >= c4 && aligned_to(x, 16) && aligned_to(out, 16)) {
    hipLaunchKernelGGL(mlqem::gather_scale_rows_v4_kernel, dim3((unsigned)ceil_div(K * (c4 / 4), kBlock)), dim3(kBlock), 0, as_stream(stream), x,
                       ldx, perm, scale, K, c4 / 4, out, ldo);
    return launch_status();
  }
  hipLaunchKernelGGL(gather_scale_rows_kernel, dim3((unsigned)ceil_div(K * C, kBlock)), dim3(kBlock), 0,
                     as_stream(stream), x, ldx, perm, scale, K, C, out, ldo);
  return launch_status();
}
