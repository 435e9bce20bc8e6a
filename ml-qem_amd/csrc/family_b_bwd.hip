// Backward kernels of Family B (TransformerConv attention, ASAPooling).  Each edge-softmax backward is split
// into a destination-side pass (per-edge softmax weight alpha_e and score gradient g_e, written once to [E]-sized
// buffers in in-CSR order, plus the gradients that are sums over a destination's in-edges) and a source-side pass
// (gradients that are sums over a source's out-edges, read through `out_eid`, the in-CSR position of each
// out-CSR entry).  No atomics: every gradient row is written once, in a fixed order.
#include <type_traits>

#include "attn_q4.hpp"
#include "dense_block.hpp"
#include "common.hpp"

namespace mlqem {


// ---------------------------------------------------------------------------------------- TransformerConv
// All three kernels give one 16-lane group to a (row, head): lane l holds channels l and l + 16 (C <= 32); row segments
// are read with coalesced 64-byte loads and dot products over the channels are cross-lane sums (common.hpp).
//
// Forward with statistics (attn_fwd.hpp): the inference kernel's schedule plus m[N,H] (segment max) and den[N,H] (sum of
// exp + 1e-16) for the backward, attn_out (the sum before the skip term) and optional dropout on the attention weights
// (mask keyed by (seed, in-CSR position, head); self-loop entries use position E + row).
template <bool WIDE> __global__ __launch_bounds__(kBlock) void transformer_attn_train_kernel(const AttnFwdArgs a) {
  attn_forward<true, WIDE>(a);
}

// Destination side: g_q, g_skip, and per edge (in-CSR order; self entries at E + row): al = effective attention weight
// (after dropout), gs = d loss / d score * (1/sqrt(C)).  WIDE = channels 16..31 present (C > 16); rows are walked in
// chunks of edges whose key and value rows are fetched together (for_edge_chunks, common.hpp).
// (AttnBwdArgs: attn_fwd.hpp)

template <bool WIDE> __global__ __launch_bounds__(kBlock) void transformer_attn_bwd_dst_kernel(const AttnBwdArgs a) {
  const int64_t t = ((int64_t)row_block() * kBlock + threadIdx.x) / kGroup;
  const int l = threadIdx.x % kGroup;
  const int H = a.H, C = a.C, HC = H * C;
  if (t >= a.N * H) return;
  const int row = (int)(t / H);
  const int h = (int)(t - (int64_t)row * H);
  const float scale = 1.0f / sqrtf((float)C);
  const float keep = 1.f / (1.f - a.drop_p);
  const uint64_t seed = a.seed + (a.seed_counter ? *a.seed_counter * 0xD1B54A32D192ED03ull : 0ull);   // as the forward
  const bool c0 = l < C, c1 = WIDE && l + kGroup < C;
  const int o0 = h * C + l, o1 = c1 ? o0 + kGroup : o0;
  const float* __restrict__ qkvs = a.qkvs;
  const int64_t ld = a.ld;
  const int32_t* __restrict__ idx = a.idx;
  float* __restrict__ edge_al = a.edge_al;
  float* __restrict__ edge_gs = a.edge_gs;
  const float q0 = c0 ? qkvs[(int64_t)row * ld + o0] : 0.f, q1 = c1 ? qkvs[(int64_t)row * ld + o1] : 0.f;
  const float gi0 = c0 ? a.g[(int64_t)row * a.ldg + o0] : 0.f, gi1 = c1 ? a.g[(int64_t)row * a.ldg + o1] : 0.f;
  float d = gi0 * (c0 ? a.attn_out[(int64_t)row * a.lda + o0] : 0.f);
  if (WIDE) d = fmaf(gi1, c1 ? a.attn_out[(int64_t)row * a.lda + o1] : 0.f, d);
  const float delta = group16_sum(d);
  float gq0 = 0.f, gq1 = 0.f;
  const float m = a.stat_m[(int64_t)row * H + h];
  const float inv_den = 1.0f / a.stat_den[(int64_t)row * H + h];     // one division per row, as in the forward
  const int beg = a.ptr[row], end = a.ptr[row + 1];
  const int n_self = a.loops ? a.loops[row] : 0;
  auto use = [&](float k0, float k1, float v0, float v1, float mult, int64_t pos) {
    float s = q0 * k0, gv = gi0 * v0;
    if (WIDE) { s = fmaf(q1, k1, s); gv = fmaf(gi1, v1, gv); }
    s = group16_sum(s);
    gv = group16_sum(gv);
    const float alpha = expf(s * scale - m) * inv_den * mult;  // softmax weight (all copies of a repeated self-loop)
    float dmask = 1.f;
    if (a.drop_p > 0.f) dmask = uniform01_edge(seed, (uint64_t)(pos * H + h)) < a.drop_p ? 0.f : keep;
    const float gs = alpha * (gv * dmask - delta) * scale;
    if (l == 0) {
      edge_al[pos * H + h] = alpha * dmask;
      edge_gs[pos * H + h] = gs;
    }
    gq0 = fmaf(gs, k0, gq0);
    if (WIDE) gq1 = fmaf(gs, k1, gq1);
  };
  // Rows of eight or more in-edges, eight at a time: the scalar work of an edge -- softmax weight, dropout draw, score
  // gradient -- is done by ONE lane (lanes u and u + 8 own edge u) and lanes 0..7 store the chunk's al / gs entries
  // together; only the two dot products of an edge involve the whole group, and gs reaches the lanes by DPP broadcast.
  auto chunk8 = [&](int e, int k) {
    const int lu = l & 7;
    const int j = idx[e + min(lu, k - 1)];              // past the end: the last edge again (gs 0, nothing stored)
    int ju[8];
    float k0[8], k1[WIDE ? 8 : 1], v0[8], v1[WIDE ? 8 : 1];
    ju[0] = group16_bcast<0>(j); ju[1] = group16_bcast<1>(j); ju[2] = group16_bcast<2>(j); ju[3] = group16_bcast<3>(j);
    ju[4] = group16_bcast<4>(j); ju[5] = group16_bcast<5>(j); ju[6] = group16_bcast<6>(j); ju[7] = group16_bcast<7>(j);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float* __restrict__ kj = qkvs + (int64_t)ju[u] * ld + HC;
      k0[u] = c0 ? kj[o0] : 0.f;
      v0[u] = c0 ? kj[HC + o0] : 0.f;
      if (WIDE) {
        k1[u] = c1 ? kj[o1] : 0.f;
        v1[u] = c1 ? kj[HC + o1] : 0.f;
      }
    }
    float mys = 0.f, mygv = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float sd = q0 * k0[u], gd = gi0 * v0[u];
      if (WIDE) { sd = fmaf(q1, k1[u], sd); gd = fmaf(gi1, v1[u], gd); }
      sd = group16_sum(sd);
      gd = group16_sum(gd);
      if (lu == u) { mys = sd; mygv = gd; }
    }
    const int64_t pos = (int64_t)e + lu;
    const float alpha = expf(mys * scale - m) * inv_den;
    float dmask = 1.f;
    if (a.drop_p > 0.f) dmask = uniform01_edge(seed, (uint64_t)(pos * H + h)) < a.drop_p ? 0.f : keep;
    float gs = alpha * (mygv * dmask - delta) * scale;
    if (lu >= k) gs = 0.f;
    if (l < 8 && lu < k) {
      edge_al[pos * H + h] = alpha * dmask;
      edge_gs[pos * H + h] = gs;
    }
    float gu[8];
    gu[0] = group16_bcast<0>(gs); gu[1] = group16_bcast<1>(gs); gu[2] = group16_bcast<2>(gs); gu[3] = group16_bcast<3>(gs);
    gu[4] = group16_bcast<4>(gs); gu[5] = group16_bcast<5>(gs); gu[6] = group16_bcast<6>(gs); gu[7] = group16_bcast<7>(gs);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      gq0 = fmaf(gu[u], k0[u], gq0);
      if (WIDE) gq1 = fmaf(gu[u], k1[u], gq1);
    }
  };
  if (end - beg >= 8) {
    for (int e = beg; e < end; e += 8) chunk8(e, min(8, end - e));
  } else {
    for_edge_chunks(beg, end, [&](int e, auto kc) {
      constexpr int K = decltype(kc)::value;
      int jj[K];
      float k0[K], k1[WIDE ? K : 1], v0[K], v1[WIDE ? K : 1];
#pragma unroll
      for (int u = 0; u < K; ++u) jj[u] = idx[e + u];
#pragma unroll
      for (int u = 0; u < K; ++u) {
        const float* __restrict__ kj = qkvs + (int64_t)jj[u] * ld + HC;
        k0[u] = c0 ? kj[o0] : 0.f;
        v0[u] = c0 ? kj[HC + o0] : 0.f;
        if (WIDE) {
          k1[u] = c1 ? kj[o1] : 0.f;
          v1[u] = c1 ? kj[HC + o1] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < K; ++u) use(k0[u], WIDE ? k1[u] : 0.f, v0[u], WIDE ? v1[u] : 0.f, 1.f, e + u);
    });
  }
  if (n_self > 0) {
    const float* __restrict__ kj = qkvs + (int64_t)row * ld + HC;
    use(c0 ? kj[o0] : 0.f, c1 ? kj[o1] : 0.f, c0 ? kj[HC + o0] : 0.f, c1 ? kj[HC + o1] : 0.f, (float)n_self, a.E + row);
  } else if (l == 0) {
    edge_al[(a.E + row) * H + h] = 0.f;
    edge_gs[(a.E + row) * H + h] = 0.f;
  }
  float* __restrict__ gq = a.gqkvs + (int64_t)row * a.ldq;
  if (c0) { gq[o0] = gq0; gq[3 * HC + o0] = gi0; }
  if (c1) { gq[o1] = gq1; gq[3 * HC + o1] = gi1; }
}

// Source side: g_k[j] = sum_{e: j->i} gs_e q_i ; g_v[j] = sum_e al_e g_i  (self entry included).
template <bool WIDE> __global__ __launch_bounds__(kBlock) void transformer_attn_bwd_src_kernel(const AttnBwdArgs a) {
  const int64_t t = ((int64_t)row_block() * kBlock + threadIdx.x) / kGroup;
  const int l = threadIdx.x % kGroup;
  const int H = a.H, C = a.C, HC = H * C;
  if (t >= a.N * H) return;
  const int row = (int)(t / H);
  const int h = (int)(t - (int64_t)row * H);
  const bool c0 = l < C, c1 = WIDE && l + kGroup < C;
  const int o0 = h * C + l, o1 = c1 ? o0 + kGroup : o0;
  const float* __restrict__ qkvs = a.qkvs;
  const float* __restrict__ g = a.g;
  const float* __restrict__ edge_al = a.edge_al;
  const float* __restrict__ edge_gs = a.edge_gs;
  const int64_t ld = a.ld, ldg = a.ldg;
  float gk0 = 0.f, gk1 = 0.f, gv0 = 0.f, gv1 = 0.f;
  const int obeg = a.optr[row], oend = a.optr[row + 1];
  if (oend - obeg >= 8) {
    // long rows, eight edges at a time: lane u (and u + 8) fetches the index entries and the two per-edge scalars of edge u,
    // the others get them by DPP broadcast when the rows are fetched
    for (int e = obeg; e < oend; e += 8) {
      const int k = min(8, oend - e);
      const int lu = l & 7;
      const int ee = e + min(lu, k - 1);
      const int i = a.odst[ee];
      const int64_t pos = a.oeid[ee];
      const float gs = lu < k ? edge_gs[pos * H + h] : 0.f;    // past the end: the last edge again with weight 0
      const float al = lu < k ? edge_al[pos * H + h] : 0.f;
      int iu[8];
      float gsu[8], alu[8], qa[8], qb[WIDE ? 8 : 1], ga[8], gb[WIDE ? 8 : 1];
      group16_bcast8<0>(i, iu);
      group16_bcast8<0>(gs, gsu);
      group16_bcast8<0>(al, alu);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        qa[u] = c0 ? qkvs[(int64_t)iu[u] * ld + o0] : 0.f;
        ga[u] = c0 ? g[(int64_t)iu[u] * ldg + o0] : 0.f;
        if (WIDE) {
          qb[u] = c1 ? qkvs[(int64_t)iu[u] * ld + o1] : 0.f;
          gb[u] = c1 ? g[(int64_t)iu[u] * ldg + o1] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        gk0 = fmaf(gsu[u], qa[u], gk0);
        gv0 = fmaf(alu[u], ga[u], gv0);
        if (WIDE) { gk1 = fmaf(gsu[u], qb[u], gk1); gv1 = fmaf(alu[u], gb[u], gv1); }
      }
    }
  } else {
    for_edge_chunks(obeg, oend, [&](int e, auto kc) {                 // chunks fetched together, used in edge order
      constexpr int K = decltype(kc)::value;
      int ii[K], pp[K];
      float gs[K], al[K], qa[K], qb[WIDE ? K : 1], ga[K], gb[WIDE ? K : 1];
#pragma unroll
      for (int u = 0; u < K; ++u) {
        ii[u] = a.odst[e + u];
        pp[u] = a.oeid[e + u];
      }
#pragma unroll
      for (int u = 0; u < K; ++u) {
        gs[u] = edge_gs[(int64_t)pp[u] * H + h];
        al[u] = edge_al[(int64_t)pp[u] * H + h];
        qa[u] = c0 ? qkvs[(int64_t)ii[u] * ld + o0] : 0.f;
        ga[u] = c0 ? g[(int64_t)ii[u] * ldg + o0] : 0.f;
        if (WIDE) {
          qb[u] = c1 ? qkvs[(int64_t)ii[u] * ld + o1] : 0.f;
          gb[u] = c1 ? g[(int64_t)ii[u] * ldg + o1] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < K; ++u) {
        gk0 = fmaf(gs[u], qa[u], gk0);
        gv0 = fmaf(al[u], ga[u], gv0);
        if (WIDE) { gk1 = fmaf(gs[u], qb[u], gk1); gv1 = fmaf(al[u], gb[u], gv1); }
      }
    });
  }
  {   // the self entry, last
    const float gs = edge_gs[(a.E + row) * H + h], al = edge_al[(a.E + row) * H + h];
    if (c0) { gk0 = fmaf(gs, qkvs[(int64_t)row * ld + o0], gk0); gv0 = fmaf(al, g[(int64_t)row * ldg + o0], gv0); }
    if (c1) { gk1 = fmaf(gs, qkvs[(int64_t)row * ld + o1], gk1); gv1 = fmaf(al, g[(int64_t)row * ldg + o1], gv1); }
  }
  float* __restrict__ gq = a.gqkvs + (int64_t)row * a.ldq;
  if (c0) { gq[HC + o0] = gk0; gq[2 * HC + o0] = gv0; }
  if (c1) { gq[HC + o1] = gk1; gq[2 * HC + o1] = gv1; }
}

// The same three kernels with four channels per lane (attn_q4.hpp): LPH = 4 (C <= 16) or 8 lanes per (row, head), 16-byte row
// segments, entries (in-edges, then the self entry) four at a time with lane u of every quad owning entry u.
// (76 registers: six waves per SIMD.  Compiled for seven -- 72 registers, 20 bytes of scratch -- the circuit-DAG forward took 224 us
// instead of 200, for eight -- 64 registers, 56 bytes -- 343: measured, left to the compiler)
template <int LPH, bool FAST> __global__ __launch_bounds__(kBlock) void transformer_attn_train_q4_kernel(const AttnFwdArgs a) {
  attn_forward_q4<true, LPH, FAST>(a);
}

// (FAST: attn_q4.hpp -- no load under a lane-varying condition)
template <int LPH, bool FAST> __global__ __launch_bounds__(kBlock) void transformer_attn_bwd_dst_q4_kernel(const AttnBwdArgs a) {
  const int64_t t = ((int64_t)row_block() * kBlock + threadIdx.x) / LPH;
  const int lq = threadIdx.x % LPH, lu = lq & 3;
  const int H = a.H, C = a.C, HC = H * C;
  if (t >= a.N * H) return;
  const int row = (int)(t / H);
  if (a.skip_dst && a.skip_dst[row]) return;
  const int h = (int)(t - (int64_t)row * H);
  const int nv = min(4, max(0, C - 4 * lq));
  const int off = h * C + 4 * lq;
  const int CP = a.CP > 0 ? a.CP : C, HP = H * CP, offp = h * CP + 4 * lq;      // head pitch / part stride / lane offset inside qkvs, gqkvs
  const int nvs = min(4, max(0, CP - 4 * lq));           // floats of the lane's slot: what it stores (pads come out zero)
  const float scale = 1.0f / sqrtf((float)C);
  const float keep = 1.f / (1.f - a.drop_p);
  const uint64_t seed = a.seed + (a.seed_counter ? *a.seed_counter * 0xD1B54A32D192ED03ull : 0ull);   // as the forward
  const float* __restrict__ qkvs = a.qkvs;
  const int64_t ld = a.ld;
  const int32_t* __restrict__ idx = a.idx;
  float* __restrict__ edge_al = a.edge_al;
  float* __restrict__ edge_gs = a.edge_gs;
  const bool recompute = a.oeid == nullptr;              // the source side keeps no per-edge values (see AttnBwdArgs)
  f4u q, gi;
  if constexpr (FAST) {
    q = *reinterpret_cast<const f4u*>(qkvs + (int64_t)row * ld + offp);
    gi = mask4(*reinterpret_cast<const f4u*>(a.g + (int64_t)row * a.ldg + off), nv);
  } else {
    q = load_channels(qkvs + (int64_t)row * ld + offp, nv, true);
    gi = load_channels(a.g + (int64_t)row * a.ldg + off, nv, off + 4 <= a.ldg);
  }
  const float m = a.stat_m[(int64_t)row * H + h];
  const float inv_den = 1.0f / a.stat_den[(int64_t)row * H + h];     // one division per row, as in the forward
  // (round 6: the in-edge side table of the forward here as well -- one dependent round trip less for rows of at most two in-edges --
  // left the kernel where it was, 268 against 262 us: it is not bound by that latency)
  const int beg = a.ptr[row];
  const int deg = a.ptr[row + 1] - beg;
  const int n_self = a.loops ? a.loops[row] : 0;
  const int cnt = deg + (n_self > 0 ? 1 : 0);
  // delta = g . attn_out.  A row of more than four entries reads the attn_out the forward stored; a shorter one (ONE chunk) forms
  // it below as sum_u alpha_u mask_u (g . v_u) -- the forward stores nothing for it (attn_q4.hpp)
  float delta = 0.f;
  if (cnt > 4) {                                         // ((row, head)-uniform; rare on the circuit DAGs)
    const float* __restrict__ ao = a.attn_out + (int64_t)row * a.lda + off;
    delta = head_sum<LPH>(dot4(gi, FAST ? mask4(*reinterpret_cast<const f4u*>(ao), nv) : load_channels(ao, nv, off + 4 <= a.lda)));
  }
  f4u gq = {0.f, 0.f, 0.f, 0.f};
  for (int x0 = 0; x0 < cnt; x0 += 4) {
    const int k = min(4, cnt - x0);
    const int x = x0 + min(lu, k - 1);                   // past the end: the last entry again (gs 0, nothing stored)
    const bool is_self = x >= deg;
    int j;
    if constexpr (FAST) {
      const int jx = idx[max(beg + min(x, deg - 1), 0)];   // (unconditional: entry 0 exists in every index array)
      j = is_self ? row : jx;
    } else {
      j = is_self ? row : idx[beg + x];
    }
    int ju[4];
    ju[0] = quad_bcast<0>(j); ju[1] = quad_bcast<1>(j); ju[2] = quad_bcast<2>(j); ju[3] = quad_bcast<3>(j);
    f4u kk[4], vv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (FAST || (u < k && nv > 0)) {
        const float* __restrict__ kj = qkvs + (int64_t)ju[u] * ld + HP + offp;
        kk[u] = *reinterpret_cast<const f4u*>(kj);
        vv[u] = *reinterpret_cast<const f4u*>(kj + HP);
      } else {
        kk[u] = f4u{0.f, 0.f, 0.f, 0.f};
        vv[u] = f4u{0.f, 0.f, 0.f, 0.f};
      }
    }
    float mys = 0.f, mygv = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float sd = head_sum<LPH>(dot4(q, kk[u])), gd = head_sum<LPH>(dot4(gi, vv[u]));
      if (lu == u) { mys = sd; mygv = gd; }
    }
    const int64_t pos = is_self ? a.E + row : (int64_t)beg + x;
    const float alpha = expf(mys * scale - m) * inv_den * (is_self ? (float)n_self : 1.f);
    float dmask = 1.f;
    if (a.drop_p > 0.f) dmask = attn_dropped(seed, a.pair_key != 0, pos, H, h, row, j, a.drop_p) ? 0.f : keep;
    if (cnt <= 4) delta = quad_sum(lu < k ? alpha * dmask * mygv : 0.f);
    float gs = alpha * (mygv * dmask - delta) * scale;
    if (lu >= k) gs = 0.f;
    if (!recompute && lq < 4 && lu < k) {                // LPH = 8: both quads of the head hold the chunk, the first one stores
      edge_al[pos * H + h] = alpha * dmask;
      edge_gs[pos * H + h] = gs;
    }
    const float gu[4] = {quad_bcast<0>(gs), quad_bcast<1>(gs), quad_bcast<2>(gs), quad_bcast<3>(gs)};
#pragma unroll
    for (int u = 0; u < 4; ++u) gq += gu[u] * kk[u];
  }
  if (recompute) {
    // what the source side needs of this row, as ONE 16-byte record per (row, head): three 4-byte gathers from three arrays per
    // out-entry were three of its seven cache-line requests
    if (lq == 0) reinterpret_cast<float4*>(edge_al)[(int64_t)row * H + h] = make_float4(m, inv_den, delta, 0.f);
  } else if (n_self == 0 && lq == 0) {
    edge_al[(a.E + row) * H + h] = 0.f;
    edge_gs[(a.E + row) * H + h] = 0.f;
  }
  float* __restrict__ go = a.gqkvs + (int64_t)row * a.ldq;
  store_channels(go + offp, gq, nvs);
  store_channels(go + 3 * HP + offp, gi, nvs);
}

// Source side WITHOUT per-edge buffers and without out_eid: every weight is recomputed from what the forward kept per (row, head)
// -- alpha = exp(q_i . k_j / sqrt(C) - m_i) / den_i, delta_i = g_i . attn_out_i (filed by the destination side) -- the way attention
// backward passes are usually written:  g_k[j] = sum_i gs_ij q_i,  g_v[j] = sum_i alpha_ij mask_ij g_i,
// gs_ij = alpha_ij (g_i . v_j mask_ij - delta_i) / sqrt(C).  Per out-entry it gathers the same two rows (query, gradient) as the
// stored form and three scalars from [N, H] arrays (a few MB: cache-resident) instead of a position and two values from
// [E, H] arrays; the coarsened graphs of ASAPooling come without out_eid (linking 9.7 M entries cost 0.55 ms per step).
template <int LPH, bool FAST> __global__ __launch_bounds__(kBlock) void transformer_attn_bwd_src_rc_q4_kernel(const AttnBwdArgs a) {
  const int64_t t = ((int64_t)row_block() * kBlock + threadIdx.x) / LPH;
  const int lq = threadIdx.x % LPH, lu = lq & 3;
  const int H = a.H, C = a.C, HC = H * C;
  if (t >= a.N * H) return;
  const int row = (int)(t / H);
  if (a.skip_src && a.skip_src[row]) return;
  const int h = (int)(t - (int64_t)row * H);
  const int nv = min(4, max(0, C - 4 * lq));
  const int off = h * C + 4 * lq;
  const int CP = a.CP > 0 ? a.CP : C, HP = H * CP, offp = h * CP + 4 * lq;      // head pitch / part stride / lane offset inside qkvs, gqkvs
  const int nvs = min(4, max(0, CP - 4 * lq));           // floats of the lane's slot: what it stores (pads come out zero)
  const float scale = 1.0f / sqrtf((float)C);
  const float keep = 1.f / (1.f - a.drop_p);
  const uint64_t seed = a.seed + (a.seed_counter ? *a.seed_counter * 0xD1B54A32D192ED03ull : 0ull);   // as the forward
  const float* __restrict__ qkvs = a.qkvs;
  const float* __restrict__ g = a.g;
  const float* __restrict__ stat_delta = a.edge_al;
  const int64_t ld = a.ld, ldg = a.ldg;
  const bool g_fits = off + 4 <= ldg;
  const float* __restrict__ rj = qkvs + (int64_t)row * ld;
  f4u kown, vown;
  if constexpr (FAST) {
    kown = *reinterpret_cast<const f4u*>(rj + HP + offp);
    vown = *reinterpret_cast<const f4u*>(rj + 2 * HP + offp);
  } else {
    kown = load_channels(rj + HP + offp, nv, true);                  // runs over into the value part at most
    vown = load_channels(rj + 2 * HP + offp, nv, true);              // ... into the skip part
  }
  const int n_self = a.loops ? a.loops[row] : 0;
  f4u gk = {0.f, 0.f, 0.f, 0.f}, gv = {0.f, 0.f, 0.f, 0.f};
  const int obeg = a.optr[row];
  const int odeg = a.optr[row + 1] - obeg;
  const int cnt = odeg + (n_self > 0 ? 1 : 0);
  for (int x0 = 0; x0 < cnt; x0 += 4) {
    const int k = min(4, cnt - x0);
    const int x = x0 + min(lu, k - 1);
    const bool is_self = x >= odeg;
    int i;
    if constexpr (FAST) {
      const int ix = a.odst[max(obeg + min(x, odeg - 1), 0)];
      i = is_self ? row : ix;
    } else {
      i = is_self ? row : a.odst[obeg + x];
    }
    const int64_t sh = (int64_t)i * H + h;
    const float4 rec = reinterpret_cast<const float4*>(stat_delta)[sh];     // {m, 1 / den, delta} filed by the destination side
    const float m_i = rec.x, inv_den = rec.y, delta_i = rec.z;
    int iu[4];
    iu[0] = quad_bcast<0>(i); iu[1] = quad_bcast<1>(i); iu[2] = quad_bcast<2>(i); iu[3] = quad_bcast<3>(i);
    f4u qa[4], ga[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if constexpr (FAST) {                              // (a padded slot's pads are zeros: only the compact gradient row is masked)
        qa[u] = *reinterpret_cast<const f4u*>(qkvs + (int64_t)iu[u] * ld + offp);
        ga[u] = mask4(*reinterpret_cast<const f4u*>(g + (int64_t)iu[u] * ldg + off), nv);
      } else if (u < k && nv > 0) {
        qa[u] = *reinterpret_cast<const f4u*>(qkvs + (int64_t)iu[u] * ld + offp);
        ga[u] = g_fits ? *reinterpret_cast<const f4u*>(g + (int64_t)iu[u] * ldg + off) : load_channels(g + (int64_t)iu[u] * ldg + off, nv, false);
        if (nv < 4) qa[u].w = 0.f;                       // the lane's fourth component belongs to the next head / part
        if (nv < 3) qa[u].z = 0.f;
        if (nv < 2) qa[u].y = 0.f;
        if (g_fits) { if (nv < 4) ga[u].w = 0.f; if (nv < 3) ga[u].z = 0.f; if (nv < 2) ga[u].y = 0.f; }
      } else {
        qa[u] = f4u{0.f, 0.f, 0.f, 0.f};
        ga[u] = f4u{0.f, 0.f, 0.f, 0.f};
      }
    }
    float mys = 0.f, mygv = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float sd = head_sum<LPH>(dot4(qa[u], kown)), gd = head_sum<LPH>(dot4(ga[u], vown));
      if (lu == u) { mys = sd; mygv = gd; }
    }
    const float alpha = expf(mys * scale - m_i) * inv_den * (is_self ? (float)n_self : 1.f);
    float dmask = 1.f;
    if (a.drop_p > 0.f) dmask = attn_dropped(seed, true, 0, H, h, i, row, a.drop_p) ? 0.f : keep;
    float gs = alpha * (mygv * dmask - delta_i) * scale, al = alpha * dmask;
    if (lu >= k) { gs = 0.f; al = 0.f; }
    const float gsu[4] = {quad_bcast<0>(gs), quad_bcast<1>(gs), quad_bcast<2>(gs), quad_bcast<3>(gs)};
    const float alu[4] = {quad_bcast<0>(al), quad_bcast<1>(al), quad_bcast<2>(al), quad_bcast<3>(al)};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      gk += gsu[u] * qa[u];
      gv += alu[u] * ga[u];
    }
  }
  float* __restrict__ go = a.gqkvs + (int64_t)row * a.ldq;
  store_channels(go + HP + offp, gk, nvs);
  store_channels(go + 2 * HP + offp, gv, nvs);
}

template <int LPH, bool FAST> __global__ __launch_bounds__(kBlock) void transformer_attn_bwd_src_q4_kernel(const AttnBwdArgs a) {
  const int64_t t = ((int64_t)row_block() * kBlock + threadIdx.x) / LPH;
  const int lq = threadIdx.x % LPH, lu = lq & 3;
  const int H = a.H, C = a.C, HC = H * C;
  if (t >= a.N * H) return;
  const int row = (int)(t / H);
  const int h = (int)(t - (int64_t)row * H);
  const int nv = min(4, max(0, C - 4 * lq));
  const int off = h * C + 4 * lq;
  const int CP = a.CP > 0 ? a.CP : C, HP = H * CP, offp = h * CP + 4 * lq;      // head pitch / part stride / lane offset inside qkvs, gqkvs
  const int nvs = min(4, max(0, CP - 4 * lq));           // floats of the lane's slot: what it stores (pads come out zero)
  const float* __restrict__ qkvs = a.qkvs;
  const float* __restrict__ g = a.g;
  const float* __restrict__ edge_al = a.edge_al;
  const float* __restrict__ edge_gs = a.edge_gs;
  const int64_t ld = a.ld, ldg = a.ldg;
  const bool g_fits = off + 4 <= ldg;                    // a gradient row is compact [H C] (+ padding to 16 bytes)
  f4u gk = {0.f, 0.f, 0.f, 0.f}, gv = {0.f, 0.f, 0.f, 0.f};
  const int obeg = a.optr[row];
  const int odeg = a.optr[row + 1] - obeg;
  const int cnt = odeg + 1;                              // the self entry, last (its weights are zero when the row has no self-loop)
  for (int x0 = 0; x0 < cnt; x0 += 4) {
    const int k = min(4, cnt - x0);
    const int x = x0 + min(lu, k - 1);
    const bool is_self = x >= odeg;
    int i;
    int64_t pos;
    float gs, al;
    if constexpr (FAST) {
      const int ex = max(obeg + min(x, odeg - 1), 0);
      const int ix = a.odst[ex], px = a.oeid[ex];
      i = is_self ? row : ix;
      pos = is_self ? a.E + row : (int64_t)px;
      const float gsv = edge_gs[pos * H + h], alv = edge_al[pos * H + h];
      gs = lu < k ? gsv : 0.f;                               // past the end: the last entry again with weight 0
      al = lu < k ? alv : 0.f;
    } else {
      i = is_self ? row : a.odst[obeg + x];
      pos = is_self ? a.E + row : (int64_t)a.oeid[obeg + x];
      gs = lu < k ? edge_gs[pos * H + h] : 0.f;              // past the end: the last entry again with weight 0
      al = lu < k ? edge_al[pos * H + h] : 0.f;
    }
    int iu[4];
    iu[0] = quad_bcast<0>(i); iu[1] = quad_bcast<1>(i); iu[2] = quad_bcast<2>(i); iu[3] = quad_bcast<3>(i);
    const float gsu[4] = {quad_bcast<0>(gs), quad_bcast<1>(gs), quad_bcast<2>(gs), quad_bcast<3>(gs)};
    const float alu[4] = {quad_bcast<0>(al), quad_bcast<1>(al), quad_bcast<2>(al), quad_bcast<3>(al)};
    f4u qa[4], ga[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if constexpr (FAST) {                              // (gs = al = 0 past the end; a padded slot's pads are zeros)
        qa[u] = *reinterpret_cast<const f4u*>(qkvs + (int64_t)iu[u] * ld + offp);
        ga[u] = mask4(*reinterpret_cast<const f4u*>(g + (int64_t)iu[u] * ldg + off), nv);
      } else if (u < k && nv > 0) {
        qa[u] = *reinterpret_cast<const f4u*>(qkvs + (int64_t)iu[u] * ld + offp);      // a query segment runs over into the key part at most
        ga[u] = g_fits ? *reinterpret_cast<const f4u*>(g + (int64_t)iu[u] * ldg + off) : load_channels(g + (int64_t)iu[u] * ldg + off, nv, false);
        if (nvs > nv && g_fits) {                        // a padded slot is stored whole: its pads must come out zero, not the next head's
          if (nv < 4) ga[u].w = 0.f;
          if (nv < 3) ga[u].z = 0.f;
          if (nv < 2) ga[u].y = 0.f;
        }
      } else {
        qa[u] = f4u{0.f, 0.f, 0.f, 0.f};
        ga[u] = f4u{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      gk += gsu[u] * qa[u];
      gv += alu[u] * ga[u];
    }
  }
  float* __restrict__ go = a.gqkvs + (int64_t)row * a.ldq;
  store_channels(go + HP + offp, gk, nvs);
  store_channels(go + 2 * HP + offp, gv, nvs);
}

// --------------------------------------------------------------------------------------------- ASAPooling
// Destination side of x'[i] = sum_e softmax_e(LeakyReLU(a_i + c_src)) x[src] (in-edges + own self-loop):
// per edge al_e (softmax weight) and gp_e (gradient at the pre-activation a_i + c_src); g_a[i] = sum_e gp_e.
// One 16-lane group per destination row; lane l holds channels l, l + 16, ... (NV per lane).  As in the forward
// (attn.hip) the scalar work of an edge is done by ONE lane: a chunk is 16 edges, lane u owns edge u -- its c[src], its
// softmax weight, its gp -- and writes al / gp for it, so a chunk's results leave as two coalesced 64-byte stores; only the
// dot product gnew[i] . x[src] of an edge involves the whole group.  Statistics (max, denominator) are recomputed first
// by the same lane-per-edge walk.
// TIES: the walk also counts, per channel, the entries of the row (sources and the row itself) that attain xmax[row, :] -- what
// the backward of ASAPooling's segment max (same x, same entries) needs from a pass of its own otherwise (segment_max_share_kernel:
// one more gather of every source row); tie_count[row, c] as a float.
// (Round 6: every load unconditional -- a lane without a channel in slice v reads the row's last channel and masks the value; see
// softmax_aggregate_bwd_src_kernel.  `has[v] ? p[..] : 0` compiles to a branch around each load: ~100 branches per row here.)
template <int NV, bool TIES> __global__ __launch_bounds__(kBlock) void softmax_aggregate_bwd_dst_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ xnew, int64_t ldn,
    const float* __restrict__ gnew, int64_t ldg, const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
    const float* __restrict__ a_dst, const float* __restrict__ c_src, float slope, int64_t N, int64_t E, int C,
    float* __restrict__ edge_al, float* __restrict__ edge_gp, float* __restrict__ g_a,
    const float* __restrict__ xmax, int64_t ldm, float* __restrict__ tie_count, int64_t ldt, const uint8_t* __restrict__ skip) {
  const int64_t row = ((int64_t)row_block() * kBlock + threadIdx.x) / kGroup;
  const int l = threadIdx.x % kGroup;
  if (row >= N) return;
  if (skip && skip[row]) return;                       // a row a dense block serves (dense_pool.hip)
  // edge_gp == nullptr: the source side recomputes its weights (softmax_aggregate_bwd_src_rc_kernel); nothing is written per
  // edge, and edge_al takes one record {a_i, m_i, 1 / den_i, delta_i} per row
  const bool recompute = edge_gp == nullptr;
  const int beg = ptr[row], end = ptr[row + 1];
  const float ai = a_dst[row], c_own = c_src[row];
  auto leaky = [&](float v) { return v > 0.f ? v : v * slope; };
  bool has[NV];
  int col[NV];                                         // the lane's channel of slice v, or the last channel (loaded, then masked)
  slice_columns<NV>(l, C, has, col);
  float gi[NV], xs[NV];
  float mx[TIES ? NV : 1];
  int ties[TIES ? NV : 1];
  float d = 0.f;
  {
    float g0[NV], xn[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      g0[v] = gnew[row * ldg + col[v]];
      xn[v] = xnew[row * ldn + col[v]];
      xs[v] = x[row * ldx + col[v]];
      if (TIES) { mx[v] = xmax[row * ldm + col[v]]; ties[v] = 0; }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      gi[v] = has[v] ? g0[v] : 0.f;
      d = fmaf(gi[v], xn[v], d);
    }
  }
  const float delta = group16_sum(d);
  // statistics: running maximum and denominator over the edges, then the self-loop
  const float s_self = leaky(ai + c_own);
  float m = -INFINITY, den = 0.f;
  float ga = 0.f;                                      // lane u: the sum of its edges' gp
  const int deg = end - beg;
  const bool short_row = deg <= 2;                     // (group-uniform) nine of ten rows of a circuit DAG
  if (short_row) {
    // ONE round trip for the entries' ids and one for their scores and rows: the general walk below reads the index list twice
    // (statistics, then weights) and waits for each -- four dependent round trips
    const int j0 = idx[deg > 0 ? beg : 0], j1 = idx[deg > 1 ? beg + 1 : 0];      // (entry 0 exists in every index array)
    const float cj0 = c_src[j0], cj1 = c_src[j1];
    float x0[NV], x1[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      x0[v] = x[(int64_t)j0 * ldx + col[v]];
      x1[v] = x[(int64_t)j1 * ldx + col[v]];
    }
    const float pre0 = ai + cj0, pre1 = ai + cj1;
    const float s0 = deg > 0 ? leaky(pre0) : -INFINITY, s1 = deg > 1 ? leaky(pre1) : -INFINITY;
    if (deg > 0) {                                     // the general walk's sequence: the entries' chunk, then the self-loop
      m = fmaxf(s0, s1);
      den = expf(s0 - m) + (deg > 1 ? expf(s1 - m) : 0.f);
    }
    if (s_self > m) { den *= expf(m - s_self); m = s_self; }
    den += expf(s_self - m);
    const float inv = 1.0f / (den + 1e-16f);
    float d0 = 0.f, d1 = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      d0 = fmaf(gi[v], x0[v], d0);
      d1 = fmaf(gi[v], x1[v], d1);
      if (TIES) ties[v] += ((deg > 0 && has[v] && x0[v] == mx[v]) ? 1 : 0) + ((deg > 1 && has[v] && x1[v] == mx[v]) ? 1 : 0);
    }
    d0 = group16_sum(d0); d1 = group16_sum(d1);
    const float al0 = expf(s0 - m) * inv, al1 = expf(s1 - m) * inv;
    const float gp0 = deg > 0 ? al0 * (d0 - delta) * (pre0 > 0.f ? 1.f : slope) : 0.f;
    const float gp1 = deg > 1 ? al1 * (d1 - delta) * (pre1 > 0.f ? 1.f : slope) : 0.f;
    if (!recompute && l < deg) {
      edge_al[beg + l] = l == 0 ? al0 : al1;
      edge_gp[beg + l] = l == 0 ? gp0 : gp1;
    }
    ga = gp0 + gp1;
  } else {
  for (int e0 = beg; e0 < end; e0 += kGroup) {
    const int k = min(kGroup, end - e0);
    const float cj = c_src[idx[e0 + min(l, k - 1)]];
    const float s = l < k ? leaky(ai + cj) : -INFINITY;
    const float cm = group16_max(s);
    if (cm > m) { den *= expf(m - cm); m = cm; }
    den += group16_sum(l < k ? expf(s - m) : 0.f);
  }
  if (s_self > m) { den *= expf(m - s_self); m = s_self; }
  den += expf(s_self - m);
  }
  const float inv = 1.0f / (den + 1e-16f);
  if (!short_row)
  for (int e0 = beg; e0 < end; e0 += kGroup) {
    const int k = min(kGroup, end - e0);
    const int j = idx[e0 + min(l, k - 1)];
    const float pre = ai + c_src[j];
    float mydot = 0.f;
    auto dots = [&](auto first, auto count) {          // CNT source rows in flight; lane u keeps the dot product of edge u
      constexpr int U0 = decltype(first)::value, CNT = decltype(count)::value;   // CNT = 2: the short rows of a circuit DAG (see attn.hip)
      int ju[CNT];
      float xv[CNT][NV];
      ju[0] = group16_bcast<U0 + 0>(j); ju[1] = group16_bcast<U0 + 1>(j);
      if constexpr (CNT == 8) {
        ju[2] = group16_bcast<U0 + 2>(j); ju[3] = group16_bcast<U0 + 3>(j); ju[4] = group16_bcast<U0 + 4>(j);
        ju[5] = group16_bcast<U0 + 5>(j); ju[6] = group16_bcast<U0 + 6>(j); ju[7] = group16_bcast<U0 + 7>(j);
      }
#pragma unroll
      for (int u = 0; u < CNT; ++u) {
        const float* __restrict__ xj = x + (int64_t)ju[u] * ldx;
#pragma unroll
        for (int v = 0; v < NV; ++v) xv[u][v] = xj[col[v]];
      }
#pragma unroll
      for (int u = 0; u < CNT; ++u) {
        float dd = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) dd = fmaf(gi[v], xv[u][v], dd);
        dd = group16_sum(dd);
        if (l == U0 + u) mydot = dd;
        if (TIES) {
#pragma unroll
          for (int v = 0; v < NV; ++v) ties[v] += (U0 + u < k && has[v] && xv[u][v] == mx[v]) ? 1 : 0;
        }
      }
    };
    if (k <= 2) dots(EdgeChunk<0>{}, EdgeChunk<2>{});
    else {
      dots(EdgeChunk<0>{}, EdgeChunk<8>{});
      if (k > 8) dots(EdgeChunk<8>{}, EdgeChunk<8>{});
    }
    if (l < k) {
      const float al = expf(leaky(pre) - m) * inv;
      const float gp = al * (mydot - delta) * (pre > 0.f ? 1.f : slope);
      if (!recompute) {
        edge_al[e0 + l] = al;
        edge_gp[e0 + l] = gp;
      }
      ga += gp;
    }
  }
  if (!short_row) ga = group16_sum(ga);
  {  // the self-loop entry (position E + row)
    const float pre = ai + c_own;
    const float al = expf(leaky(pre) - m) * inv;
    float dd = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      dd = fmaf(gi[v], xs[v], dd);
      if (TIES) {
        ties[v] += (has[v] && xs[v] == mx[v]) ? 1 : 0;
        if (has[v]) tie_count[row * ldt + l + v * kGroup] = (float)ties[v];
      }
    }
    dd = group16_sum(dd);
    const float gp = al * (dd - delta) * (pre > 0.f ? 1.f : slope);
    ga += gp;
    if (l == 0) {
      if (recompute) {
        reinterpret_cast<float4*>(edge_al)[row] = make_float4(ai, m, inv, delta);
      } else {
        edge_al[E + row] = al;
        edge_gp[E + row] = gp;
      }
      g_a[row] = ga;
    }
  }
}

// The same for rows wider than 128 channels (no reference model has them): every lane repeats the per-edge scalars.
// Destination side of x'[i] = sum_e softmax_e(LeakyReLU(a_i + c_src)) x[src] (in-edges + own self-loop):
// per edge al_e (softmax weight) and gp_e (gradient at the pre-activation a_i + c_src); g_a[i] = sum_e gp_e.
// One 16-lane group per destination row; lane l holds channels l, l + 16, ...
__global__ __launch_bounds__(kBlock) void softmax_aggregate_bwd_dst_any_width_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ xnew, int64_t ldn,
    const float* __restrict__ gnew, int64_t ldg, const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
    const float* __restrict__ a_dst, const float* __restrict__ c_src, float slope, int64_t N, int64_t E, int C,
    float* __restrict__ edge_al, float* __restrict__ edge_gp, float* __restrict__ g_a) {
  const int64_t row = ((int64_t)row_block() * kBlock + threadIdx.x) / kGroup;
  const int l = threadIdx.x % kGroup;
  if (row >= N) return;
  const int beg = ptr[row], end = ptr[row + 1];
  const float ai = a_dst[row];
  auto leaky = [&](float v) { return v > 0.f ? v : v * slope; };
  // rows are walked in chunks of edges fetched together (for_edge_chunks, common.hpp); every sum runs in edge order
  auto walk_c = [&](auto&& use) {
    for_edge_chunks(beg, end, [&](int e, auto kc) {
      constexpr int K = decltype(kc)::value;
      int jj[K];
      float cj[K];
#pragma unroll
      for (int u = 0; u < K; ++u) jj[u] = idx[e + u];
#pragma unroll
      for (int u = 0; u < K; ++u) cj[u] = c_src[jj[u]];
#pragma unroll
      for (int u = 0; u < K; ++u) use(cj[u]);
    });
  };
  float m = leaky(ai + c_src[row]);
  walk_c([&](float cj) { m = fmaxf(m, leaky(ai + cj)); });
  float den = expf(leaky(ai + c_src[row]) - m);
  walk_c([&](float cj) { den += expf(leaky(ai + cj) - m); });
  den += 1e-16f;
  const float* __restrict__ gi = gnew + row * ldg;
  float d = 0.f;
  for (int c = l; c < C; c += kGroup) d = fmaf(gi[c], xnew[row * ldn + c], d);
  const float delta = group16_sum(d);
  float ga = 0.f;
  auto use = [&](float cjv, float dot, int64_t pos) {
    const float pre = ai + cjv;
    const float al = expf(leaky(pre) - m) / den;
    dot = group16_sum(dot);
    const float gp = al * (dot - delta) * (pre > 0.f ? 1.f : slope);
    if (l == 0) {
      edge_al[pos] = al;
      edge_gp[pos] = gp;
    }
    ga += gp;
  };
  auto visit = [&](int64_t j, int64_t pos) {
    const float* __restrict__ xj = x + j * ldx;
    float dot = 0.f;
    for (int c = l; c < C; c += kGroup) dot = fmaf(gi[c], xj[c], dot);
    use(c_src[j], dot, pos);
  };
  if (C <= 2 * kGroup) {
    const bool c0 = l < C, c1 = l + kGroup < C;
    const float g0 = c0 ? gi[l] : 0.f, g1 = c1 ? gi[l + kGroup] : 0.f;
    for_edge_chunks(beg, end, [&](int e, auto kc) {
      constexpr int K = decltype(kc)::value;
      int64_t jj[K];
      float cj[K], x0[K], x1[K];
#pragma unroll
      for (int u = 0; u < K; ++u) jj[u] = idx[e + u];
#pragma unroll
      for (int u = 0; u < K; ++u) {
        const float* __restrict__ xj = x + jj[u] * ldx;
        cj[u] = c_src[jj[u]];
        x0[u] = c0 ? xj[l] : 0.f;
        x1[u] = c1 ? xj[l + kGroup] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < K; ++u) {
        float dot = 0.f;                 // the same fmaf chain as visit(): channels l, l + 16
        if (c0) dot = fmaf(g0, x0[u], dot);
        if (c1) dot = fmaf(g1, x1[u], dot);
        use(cj[u], dot, e + u);
      }
    });
  } else {
    for (int e = beg; e < end; ++e) visit(idx[e], e);
  }
  visit(row, E + row);
  if (l == 0) g_a[row] = ga;
}

// The three source-/destination-side passes below give a 16-lane group to a row; lane l holds channels l, l + 16, ... (NV
// per lane).  A chunk is 16 edges: lane u fetches the index entries (and per-edge scalars) of edge u, the others get them by
// DPP broadcast when the rows are fetched, eight rows in flight.  (A thread per (row, channel) repeats the index loads and
// the address arithmetic in each of a row's 30-45 threads: these kernels are bound by instruction issue, see attn_fwd.hpp.)
// The *_any_width kernels after them are the thread-per-(row, channel) forms, kept for rows wider than 128 channels.

// Source side: g_x[j,:] (+)= sum_{e: j->i} al_e gnew[i,:] (self included); g_c[j] = sum_e gp_e.
// FUSE_MAX (round 5): the walk also carries the backward of ASAPooling's segment max over the same entries -- g_x[j, c] += sum over
// the destinations i of j (and j itself) with x[j, c] == xmax[i, c] of g_a[i] w_comp[c] / ties[i, c] -- which was a share pass over
// [N, C] (segment_max_share_from_counts_kernel) and a second walk that re-read g_x (segment_max_bwd_kernel): 0.28 ms of the level-0
// backward of a 64-circuit step.  MaxFuse: x, xmax, the tie counts, g_a and w_comp.
struct MaxFuse {
  const float* x; int64_t ldx; const float* xmax; int64_t ldm; const float* ties; int64_t ldt; const float* g_row; const float* g_col;
};
// (Round 6: every load of this kernel is UNCONDITIONAL -- a lane without a channel in slice v reads the row's last channel and masks
// the value -- and every load the maximum's part needs is issued in front of the comparison `x == xmax`.  `has[v] ? p[..] : 0` compiles to
// a branch around the load and a load whose only use sits inside `if (x == xmax)` is sunk into that branch: the first form of this
// kernel waited for memory six times at the top of every row, one slice after the other, where one round trip was due.)
// 1 / (the number of entries that attain a maximum): v_rcp_f32 (1 ulp; exact for 1, 2, 4, ...).  A division here is a dozen
// instructions, and -- worse -- the compiler turns `cond ? a / b : 0` into a branch and sinks the loads behind it into that branch.
__device__ __forceinline__ float share_of(float ties) { return __builtin_amdgcn_rcpf(ties > 0.f ? ties : 1.f); }
template <int NV, bool FUSE_MAX> __global__ __launch_bounds__(kBlock) void softmax_aggregate_bwd_src_kernel(
    const float* __restrict__ gnew, int64_t ldg, const int32_t* __restrict__ optr, const int32_t* __restrict__ odst,
    const int32_t* __restrict__ oeid, const float* __restrict__ edge_al, const float* __restrict__ edge_gp, int64_t N,
    int64_t E, int C, int accumulate, float* __restrict__ gx, int64_t ldgx, float* __restrict__ g_c, const float* __restrict__ rank1,
    const MaxFuse mf) {
  const int64_t row = ((int64_t)row_block() * kBlock + threadIdx.x) / kGroup;
  const int l = threadIdx.x % kGroup;
  if (row >= N) return;
  bool has[NV];
  int col[NV];                                         // the lane's channel of slice v, or the last channel (loaded, then masked)
  slice_columns<NV>(l, C, has, col);
  float acc[NV], xv[FUSE_MAX ? NV : 1], wc[FUSE_MAX ? NV : 1];
  const float al_self = edge_al[E + row];
  const int beg = optr[row], end = optr[row + 1];
  {
    float g0[NV], xm[FUSE_MAX ? NV : 1], n[FUSE_MAX ? NV : 1];
    const float g_own = FUSE_MAX ? mf.g_row[row] : 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      g0[v] = gnew[row * ldg + col[v]];
      if (FUSE_MAX) {
        xv[v] = mf.x[row * mf.ldx + col[v]];
        wc[v] = mf.g_col[col[v]];
        xm[v] = mf.xmax[row * mf.ldm + col[v]];
        n[v] = mf.ties[row * mf.ldt + col[v]];
      }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      acc[v] = has[v] ? al_self * g0[v] : 0.f;
      if (FUSE_MAX) {                                  // the row itself is an entry of its own maximum
        const float share = g_own * wc[v] * share_of(n[v]);
        acc[v] += (has[v] && xv[v] == xm[v]) ? share : 0.f;
      }
    }
  }
  float gc = 0.f;                                      // lane u: the gp of its edges
  for (int e0 = beg; e0 < end; e0 += kGroup) {
    const int k = min(kGroup, end - e0);
    const int ee = e0 + min(l, k - 1);
    const int i = odst[ee], pos = oeid[ee];
    const float alv = edge_al[pos], gpv = edge_gp[pos];
    const float al = l < k ? alv : 0.f;                  // lanes past the end: the last edge again with weight 0
    gc += l < k ? gpv : 0.f;
    auto rows = [&](auto first, auto count) {
      constexpr int U0 = decltype(first)::value, CNT = decltype(count)::value;   // CNT = 2: the short rows of a circuit DAG (see attn.hip)
      int iu[CNT];
      float au[CNT], gn[CNT][NV];
      iu[0] = group16_bcast<U0 + 0>(i); iu[1] = group16_bcast<U0 + 1>(i);
      au[0] = group16_bcast<U0 + 0>(al); au[1] = group16_bcast<U0 + 1>(al);
      if constexpr (CNT >= 4) {
        iu[2] = group16_bcast<U0 + 2>(i); iu[3] = group16_bcast<U0 + 3>(i);
        au[2] = group16_bcast<U0 + 2>(al); au[3] = group16_bcast<U0 + 3>(al);
      }
      if constexpr (CNT == 8) {
        iu[4] = group16_bcast<U0 + 4>(i); iu[5] = group16_bcast<U0 + 5>(i); iu[6] = group16_bcast<U0 + 6>(i); iu[7] = group16_bcast<U0 + 7>(i);
        au[4] = group16_bcast<U0 + 4>(al); au[5] = group16_bcast<U0 + 5>(al); au[6] = group16_bcast<U0 + 6>(al); au[7] = group16_bcast<U0 + 7>(al);
      }
      // the maximum's part rides in the two-entry form (the rows of a circuit DAG); longer rows take it four entries at a time
      // below -- with eight rows of three matrices in flight the kernel needed 130 registers (three waves per SIMD: 520 us)
      constexpr bool MAX_HERE = FUSE_MAX && CNT == 2;
      float xm[MAX_HERE ? CNT : 1][MAX_HERE ? NV : 1], tc[MAX_HERE ? CNT : 1][MAX_HERE ? NV : 1], gr[MAX_HERE ? CNT : 1];
#pragma unroll
      for (int u = 0; u < CNT; ++u) {
        const float* __restrict__ gi = gnew + (int64_t)iu[u] * ldg;
#pragma unroll
        for (int v = 0; v < NV; ++v) gn[u][v] = gi[col[v]];
        if (MAX_HERE) {
          gr[u] = mf.g_row[iu[u]];
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            xm[u][v] = mf.xmax[(int64_t)iu[u] * mf.ldm + col[v]];
            tc[u][v] = mf.ties[(int64_t)iu[u] * mf.ldt + col[v]];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < CNT; ++u)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          acc[v] = fmaf(au[u], has[v] ? gn[u][v] : 0.f, acc[v]);
          if (MAX_HERE) {                                  // (U0 + u < k: lanes past the end repeat the last entry)
            const float share = gr[u] * wc[v] * share_of(tc[u][v]);
            acc[v] += (U0 + u < k && has[v] && xv[v] == xm[u][v]) ? share : 0.f;
          }
        }
    };
    auto max_rows = [&](auto first) {                      // four destinations' maxima, tie counts and g_a in flight
      constexpr int U0 = decltype(first)::value;
      const int iu[4] = {group16_bcast<U0 + 0>(i), group16_bcast<U0 + 1>(i), group16_bcast<U0 + 2>(i), group16_bcast<U0 + 3>(i)};
      float xm[4][NV], tc[4][NV], gr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        gr[u] = mf.g_row[iu[u]];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          xm[u][v] = mf.xmax[(int64_t)iu[u] * mf.ldm + col[v]];
          tc[u][v] = mf.ties[(int64_t)iu[u] * mf.ldt + col[v]];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const float share = gr[u] * wc[FUSE_MAX ? v : 0] * share_of(tc[u][v]);
          acc[v] += (U0 + u < k && has[v] && xv[FUSE_MAX ? v : 0] == xm[u][v]) ? share : 0.f;
        }
    };
    if (k <= 2) rows(EdgeChunk<0>{}, EdgeChunk<2>{});
    else if (FUSE_MAX) {                                   // (four rows of one matrix in flight: the registers of the two-entry form)
      rows(EdgeChunk<0>{}, EdgeChunk<4>{}); max_rows(EdgeChunk<0>{});
      if (k > 4) { rows(EdgeChunk<4>{}, EdgeChunk<4>{}); max_rows(EdgeChunk<4>{}); }
      if (k > 8) { rows(EdgeChunk<8>{}, EdgeChunk<4>{}); max_rows(EdgeChunk<8>{}); }
      if (k > 12) { rows(EdgeChunk<12>{}, EdgeChunk<4>{}); max_rows(EdgeChunk<12>{}); }
    } else {
      rows(EdgeChunk<0>{}, EdgeChunk<8>{});
      if (k > 8) rows(EdgeChunk<8>{}, EdgeChunk<8>{});
    }
  }
  gc = group16_sum(gc) + edge_gp[E + row];
  float* __restrict__ d = gx + row * ldgx;
  float r1[NV], prev[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    r1[v] = rank1 ? rank1[col[v]] : 0.f;                  // + g_c[row] * att_x: the gradient through c = x . att_x
    prev[v] = accumulate ? d[col[v]] : 0.f;
  }
#pragma unroll
  for (int v = 0; v < NV; ++v)
    if (has[v]) d[l + v * kGroup] = prev[v] + (acc[v] + gc * r1[v]);
  if (l == 0) g_c[row] = gc;
}

// Source side WITHOUT per-edge buffers and without out_eid (see transformer_attn_bwd_src_rc_q4_kernel): for every out-entry j -> i
//   al = exp(LeakyReLU(a_i + c_j) - m_i) / den_i,   gp = al (gnew_i . x_j - delta_i) LeakyReLU'(a_i + c_j)
// from the row's own x_j, c_j and the destination's record {a_i, m_i, 1 / den_i, delta_i} (one 16-byte gather instead of a
// position and two values from [E]-sized arrays); the gathered gnew rows are the ones g_x needs anyway.
template <int NV> __global__ __launch_bounds__(kBlock) void softmax_aggregate_bwd_src_rc_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ gnew, int64_t ldg, const int32_t* __restrict__ optr,
    const int32_t* __restrict__ odst, const float4* __restrict__ stat, const float* __restrict__ c_src, float slope, int64_t N, int C,
    int accumulate, float* __restrict__ gx, int64_t ldgx, float* __restrict__ g_c, const float* __restrict__ rank1,
    const uint8_t* __restrict__ skip) {
  const int64_t row = ((int64_t)row_block() * kBlock + threadIdx.x) / kGroup;
  const int l = threadIdx.x % kGroup;
  if (row >= N) return;
  if (skip && skip[row]) return;
  auto leaky = [&](float v) { return v > 0.f ? v : v * slope; };
  const float cj = c_src[row];
  bool has[NV];
  int col[NV];
  slice_columns<NV>(l, C, has, col);
  float acc[NV], xr[NV];
  float gc;
  {  // the self entry
    const float4 st = stat[row];
    const float pre = st.x + cj;
    const float al = expf(leaky(pre) - st.y) * st.z;
    float dd = 0.f, g0[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      xr[v] = x[row * ldx + col[v]];
      g0[v] = gnew[row * ldg + col[v]];
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      xr[v] = has[v] ? xr[v] : 0.f;
      const float gn = has[v] ? g0[v] : 0.f;
      acc[v] = al * gn;
      dd = fmaf(gn, xr[v], dd);
    }
    dd = group16_sum(dd);
    gc = l == 0 ? al * (dd - st.w) * (pre > 0.f ? 1.f : slope) : 0.f;      // lane u: the gp of its entries (the self entry is lane 0's)
  }
  const int beg = optr[row], end = optr[row + 1];
  for (int e0 = beg; e0 < end; e0 += kGroup) {
    const int k = min(kGroup, end - e0);
    const int i = odst[e0 + min(l, k - 1)];
    const float4 st = stat[i];
    const float pre = st.x + cj;
    const float al = l < k ? expf(leaky(pre) - st.y) * st.z : 0.f;        // lanes past the end: the last entry again with weight 0
    float mydot = 0.f;
    auto rows = [&](auto first, auto count) {
      constexpr int U0 = decltype(first)::value, CNT = decltype(count)::value;
      int iu[CNT];
      float au[CNT], gn[CNT][NV];
      iu[0] = group16_bcast<U0 + 0>(i); iu[1] = group16_bcast<U0 + 1>(i);
      au[0] = group16_bcast<U0 + 0>(al); au[1] = group16_bcast<U0 + 1>(al);
      if constexpr (CNT == 8) {
        iu[2] = group16_bcast<U0 + 2>(i); iu[3] = group16_bcast<U0 + 3>(i); iu[4] = group16_bcast<U0 + 4>(i);
        iu[5] = group16_bcast<U0 + 5>(i); iu[6] = group16_bcast<U0 + 6>(i); iu[7] = group16_bcast<U0 + 7>(i);
        au[2] = group16_bcast<U0 + 2>(al); au[3] = group16_bcast<U0 + 3>(al); au[4] = group16_bcast<U0 + 4>(al);
        au[5] = group16_bcast<U0 + 5>(al); au[6] = group16_bcast<U0 + 6>(al); au[7] = group16_bcast<U0 + 7>(al);
      }
#pragma unroll
      for (int u = 0; u < CNT; ++u) {
        const float* __restrict__ gi = gnew + (int64_t)iu[u] * ldg;
#pragma unroll
        for (int v = 0; v < NV; ++v) gn[u][v] = gi[col[v]];
      }
#pragma unroll
      for (int u = 0; u < CNT; ++u) {
        float dd = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const float g = has[v] ? gn[u][v] : 0.f;
          acc[v] = fmaf(au[u], g, acc[v]);
          dd = fmaf(g, xr[v], dd);
        }
        dd = group16_sum(dd);
        if (l == U0 + u) mydot = dd;
      }
    };
    if (k <= 2) rows(EdgeChunk<0>{}, EdgeChunk<2>{});
    else {
      rows(EdgeChunk<0>{}, EdgeChunk<8>{});
      if (k > 8) rows(EdgeChunk<8>{}, EdgeChunk<8>{});
    }
    if (l < k) gc += al * (mydot - st.w) * (pre > 0.f ? 1.f : slope);
  }
  gc = group16_sum(gc);
  float* __restrict__ d = gx + row * ldgx;
  float r1[NV], prev[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    r1[v] = rank1 ? rank1[col[v]] : 0.f;                  // + g_c[row] * att_x: the gradient through c = x . att_x
    prev[v] = accumulate ? d[col[v]] : 0.f;
  }
#pragma unroll
  for (int v = 0; v < NV; ++v)
    if (has[v]) {
      const float r = acc[v] + (rank1 ? gc * r1[v] : 0.f);
      d[l + v * kGroup] = accumulate ? prev[v] + r : r;
    }
  if (l == 0) g_c[row] = gc;
}

// Segment-max backward.  Ties: the gradient of a row's maximum is split EVENLY among the entries that attain it
// (torch's scatter_reduce(amax) rule, which PyG >= 2.3 uses without torch_scatter): circuit graphs do contain exact
// ties, because identical gates on one qubit have identical feature rows.
// Pass 1 (destination side): gshare[i,c] = gmax[i,c] / #{entries of row i (sources and i itself) equal to xmax[i,c]}.
template <int NV> __global__ __launch_bounds__(kBlock) void segment_max_share_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ xmax, int64_t ldm,
    const float* __restrict__ gmax, int64_t ldg, const int32_t* __restrict__ iptr, const int32_t* __restrict__ isrc,
    int64_t N, int C, float* __restrict__ gshare, int64_t lds) {
  const int64_t row = ((int64_t)row_block() * kBlock + threadIdx.x) / kGroup;
  const int l = threadIdx.x % kGroup;
  if (row >= N) return;
  bool has[NV];
  int col[NV];
  slice_columns<NV>(l, C, has, col);
  float m[NV], gm[NV];
  int cnt[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    m[v] = xmax[row * ldm + col[v]];
    gm[v] = gmax[row * ldg + col[v]];
    cnt[v] = (has[v] && x[row * ldx + col[v]] == m[v]) ? 1 : 0;
  }
  const int beg = iptr[row], end = iptr[row + 1];
  for (int e0 = beg; e0 < end; e0 += kGroup) {
    const int k = min(kGroup, end - e0);
    const int j = isrc[e0 + min(l, k - 1)];
    auto rows = [&](auto first, int kk) {               // kk: edges of this half that exist
      constexpr int U0 = decltype(first)::value;
      int ju[8];
      float xs[8][NV];
      group16_bcast8<U0>(j, ju);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float* __restrict__ xj = x + (int64_t)ju[u] * ldx;
#pragma unroll
        for (int v = 0; v < NV; ++v) xs[u][v] = xj[col[v]];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int v = 0; v < NV; ++v) cnt[v] += (u < kk && has[v] && xs[u][v] == m[v]) ? 1 : 0;
    };
    rows(EdgeChunk<0>{}, k);
    if (k > 8) rows(EdgeChunk<8>{}, k - 8);
  }
#pragma unroll
  for (int v = 0; v < NV; ++v)
    if (has[v]) gshare[row * lds + l + v * kGroup] = gm[v] / (float)(cnt[v] > 0 ? cnt[v] : 1);
}

// Pass 1 when the tie counts are already known (softmax_aggregate_bwd_dst_kernel<., true>): an elementwise division, a thread per
// 16-byte slice of a row (VEC = 4: all three matrices in the padded row layout) or per element
template <int VEC>
__global__ __launch_bounds__(kBlock) void segment_max_share_from_counts_kernel(const float* __restrict__ gmax, int64_t ldg,
                                                                               const float* __restrict__ cnt, int64_t ldc, int64_t N, int C,
                                                                               float* __restrict__ gshare, int64_t lds,
                                                                               const float* __restrict__ g_row, const float* __restrict__ g_col) {
  const int cv = (C + VEC - 1) / VEC;
  const int64_t t = (int64_t)row_block() * kBlock + threadIdx.x;
  if (t >= N * cv) return;
  int cs;
  const int64_t row = split_index(t, cv, cs);
  const int c = cs * VEC;
  float g[VEC], n[VEC];
  if (g_row) {                       // gmax = g_row (x) g_col, never formed (ASAPooling's composed score projection: one-wide gradient)
    const float gr = g_row[row];
#pragma unroll
    for (int v = 0; v < VEC; ++v) g[v] = c + v < C ? gr * g_col[c + v] : 0.f;
  } else vload<VEC>(gmax + row * ldg + c, g);
  vload<VEC>(cnt + row * ldc + c, n);
#pragma unroll
  for (int v = 0; v < VEC; ++v) g[v] = g[v] / (n[v] > 0.f ? n[v] : 1.f);       // pad columns: scratch in, scratch out
  vstore<VEC>(gshare + row * lds + c, g);
}

// Pass 2 (source side): g_x[j,c] += sum over destinations i of j (and j itself) whose maximum equals x[j,c].
template <int NV> __global__ __launch_bounds__(kBlock) void segment_max_bwd_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ xmax, int64_t ldm,
    const float* __restrict__ gmax, int64_t ldg, const int32_t* __restrict__ optr, const int32_t* __restrict__ odst,
    int64_t N, int C, float* __restrict__ gx, int64_t ldgx, const uint8_t* __restrict__ skip) {
  const int64_t row = ((int64_t)row_block() * kBlock + threadIdx.x) / kGroup;
  const int l = threadIdx.x % kGroup;
  if (row >= N) return;
  if (skip && skip[row]) return;
  bool has[NV];
  int col[NV];
  slice_columns<NV>(l, C, has, col);
  float xv[NV], acc[NV], prev[NV];
  {
    float xm0[NV], gm0[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      xv[v] = x[row * ldx + col[v]];
      xm0[v] = xmax[row * ldm + col[v]];
      gm0[v] = gmax[row * ldg + col[v]];
      prev[v] = gx[row * ldgx + col[v]];
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = (has[v] && xv[v] == xm0[v]) ? gm0[v] : 0.f;
  }
  const int beg = optr[row], end = optr[row + 1];
  for (int e0 = beg; e0 < end; e0 += kGroup) {
    const int k = min(kGroup, end - e0);
    const int i = odst[e0 + min(l, k - 1)];
    auto rows = [&](auto first, auto count, int kk) {
      constexpr int U0 = decltype(first)::value, CNT = decltype(count)::value;   // CNT = 2: the short rows of a circuit DAG (see attn.hip)
      int iu[CNT];
      float xm[CNT][NV], gm[CNT][NV];
      iu[0] = group16_bcast<U0 + 0>(i); iu[1] = group16_bcast<U0 + 1>(i);
      if constexpr (CNT == 8) {
        iu[2] = group16_bcast<U0 + 2>(i); iu[3] = group16_bcast<U0 + 3>(i); iu[4] = group16_bcast<U0 + 4>(i);
        iu[5] = group16_bcast<U0 + 5>(i); iu[6] = group16_bcast<U0 + 6>(i); iu[7] = group16_bcast<U0 + 7>(i);
      }
#pragma unroll
      for (int u = 0; u < CNT; ++u) {
        const float* __restrict__ xi = xmax + (int64_t)iu[u] * ldm;
        const float* __restrict__ gi = gmax + (int64_t)iu[u] * ldg;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          xm[u][v] = xi[col[v]];
          gm[u][v] = gi[col[v]];
        }
      }
#pragma unroll
      for (int u = 0; u < CNT; ++u)
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] += (u < kk && has[v] && xv[v] == xm[u][v]) ? gm[u][v] : 0.f;
    };
    if (k <= 2) rows(EdgeChunk<0>{}, EdgeChunk<2>{}, k);
    else {
      rows(EdgeChunk<0>{}, EdgeChunk<8>{}, k);
      if (k > 8) rows(EdgeChunk<8>{}, EdgeChunk<8>{}, k - 8);
    }
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) if (has[v]) gx[row * ldgx + l + v * kGroup] = prev[v] + acc[v];
}

// Source side: g_x[j,:] (+)= sum_{e: j->i} al_e gnew[i,:] (self included); g_c[j] = sum_e gp_e.
__global__ __launch_bounds__(kBlock) void softmax_aggregate_bwd_src_any_width_kernel(
    const float* __restrict__ gnew, int64_t ldg, const int32_t* __restrict__ optr, const int32_t* __restrict__ odst,
    const int32_t* __restrict__ oeid, const float* __restrict__ edge_al, const float* __restrict__ edge_gp, int64_t N,
    int64_t E, int C, int accumulate, float* __restrict__ gx, int64_t ldgx, float* __restrict__ g_c) {
  const int64_t t = (int64_t)row_block() * kBlock + threadIdx.x;
  if (t >= N * C) return;
  const int64_t row = t / C;
  const int ch = (int)(t - row * C);
  float acc = edge_al[E + row] * gnew[row * ldg + ch];
  float gc = edge_gp[E + row];
  for_edge_chunks(optr[row], optr[row + 1], [&](int e, auto kc) {     // chunks fetched together, used in edge order
    constexpr int K = decltype(kc)::value;
    int pos[K];
    int64_t ii[K];
    float al[K], gp[K], gn[K];
#pragma unroll
    for (int u = 0; u < K; ++u) {
      pos[u] = oeid[e + u];
      ii[u] = odst[e + u];
    }
#pragma unroll
    for (int u = 0; u < K; ++u) {
      al[u] = edge_al[pos[u]];
      gp[u] = edge_gp[pos[u]];
      gn[u] = gnew[ii[u] * ldg + ch];
    }
#pragma unroll
    for (int u = 0; u < K; ++u) {
      acc = fmaf(al[u], gn[u], acc);
      gc += gp[u];
    }
  });
  float* d = gx + row * ldgx + ch;
  *d = accumulate ? *d + acc : acc;
  if (ch == 0) g_c[row] = gc;
}

// Segment-max backward.  Ties: the gradient of a row's maximum is split EVENLY among the entries that attain it
// (torch's scatter_reduce(amax) rule, which PyG >= 2.3 uses without torch_scatter): circuit graphs do contain exact
// ties, because identical gates on one qubit have identical feature rows.
// Pass 1 (destination side): gshare[i,c] = gmax[i,c] / #{entries of row i (sources and i itself) equal to xmax[i,c]}.
__global__ __launch_bounds__(kBlock) void segment_max_share_any_width_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ xmax, int64_t ldm,
    const float* __restrict__ gmax, int64_t ldg, const int32_t* __restrict__ iptr, const int32_t* __restrict__ isrc,
    int64_t N, int C, float* __restrict__ gshare, int64_t lds) {
  const int64_t t = (int64_t)row_block() * kBlock + threadIdx.x;
  if (t >= N * C) return;
  const int64_t row = t / C;
  const int ch = (int)(t - row * C);
  const float m = xmax[row * ldm + ch];
  int cnt = x[row * ldx + ch] == m ? 1 : 0;
  for_edge_chunks(iptr[row], iptr[row + 1], [&](int e, auto kc) {     // chunks of edges fetched together (common.hpp)
    constexpr int K = decltype(kc)::value;
    float xs[K];
#pragma unroll
    for (int u = 0; u < K; ++u) xs[u] = x[(int64_t)isrc[e + u] * ldx + ch];
#pragma unroll
    for (int u = 0; u < K; ++u) cnt += xs[u] == m ? 1 : 0;
  });
  gshare[row * lds + ch] = gmax[row * ldg + ch] / (float)(cnt > 0 ? cnt : 1);
}

// Pass 2 (source side): g_x[j,c] += sum over destinations i of j (and j itself) whose maximum equals x[j,c].
__global__ __launch_bounds__(kBlock) void segment_max_bwd_any_width_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ xmax, int64_t ldm,
    const float* __restrict__ gmax, int64_t ldg, const int32_t* __restrict__ optr, const int32_t* __restrict__ odst,
    int64_t N, int C, float* __restrict__ gx, int64_t ldgx) {
  const int64_t t = (int64_t)row_block() * kBlock + threadIdx.x;
  if (t >= N * C) return;
  const int64_t row = t / C;
  const int ch = (int)(t - row * C);
  const float v = x[row * ldx + ch];
  float acc = (v == xmax[row * ldm + ch]) ? gmax[row * ldg + ch] : 0.f;
  for_edge_chunks(optr[row], optr[row + 1], [&](int e, auto kc) {     // chunks fetched together, used in edge order
    constexpr int K = decltype(kc)::value;
    float xm[K], gm[K];
#pragma unroll
    for (int u = 0; u < K; ++u) {
      const int64_t i = odst[e + u];
      xm[u] = xmax[i * ldm + ch];
      gm[u] = gmax[i * ldg + ch];
    }
#pragma unroll
    for (int u = 0; u < K; ++u) if (v == xm[u]) acc += gm[u];
  });
  gx[row * ldgx + ch] += acc;
}

// x_out[p] = x'[perm[p]] * f[perm[p]]:  g_x'[perm[p],:] = g_out[p,:] * f ; g_f[perm[p]] = g_out[p,:] . x'[perm[p],:].
// Rows that were not kept get zero (buffers are zero-filled by the caller's memset: here by the kernel over N first).
__global__ __launch_bounds__(kBlock) void gather_scale_rows_bwd_kernel(
    const float* __restrict__ gout, int64_t ldgo, const float* __restrict__ xnew, int64_t ldn,
    const float* __restrict__ fitness, const int32_t* __restrict__ slot, int64_t N, int C, float* __restrict__ gxnew,
    int64_t ldgn, float* __restrict__ gfit) {
  const int64_t row = ((int64_t)row_block() * kBlock + threadIdx.x) / kGroup;   // one 16-lane group per row
  const int l = threadIdx.x % kGroup;
  if (row >= N) return;
  const int p = slot[row];
  const float f = fitness[row];
  float dot = 0.f;
  for (int c = l; c < C; c += kGroup) {
    float gv = 0.f;
    if (p >= 0) {
      const float go = gout[(int64_t)p * ldgo + c];
      gv = go * f;
      dot = fmaf(go, xnew[row * ldn + c], dot);
    }
    gxnew[row * ldgn + c] = gv;
  }
  dot = group16_sum(dot);
  if (l == 0) gfit[row] = dot;
}

// ... rows of at most 64 channels in the padded layout: a lane's four channels as one 16-byte access, no loop over the channels and no
// load under `if (kept)` (the form above waits for slot[row] before it issues the row loads, one channel slice after the other)
__global__ __launch_bounds__(kBlock) void gather_scale_rows_bwd_v4_kernel(
    const float* __restrict__ gout, int64_t ldgo, const float* __restrict__ xnew, int64_t ldn, const float* __restrict__ fitness,
    const int32_t* __restrict__ slot, int64_t N, int C, float* __restrict__ gxnew, int64_t ldgn, float* __restrict__ gfit) {
  const int64_t row = ((int64_t)row_block() * kBlock + threadIdx.x) / kGroup;   // one 16-lane group per row
  const int l = threadIdx.x % kGroup;
  if (row >= N) return;
  const int p = slot[row];
  const float f = fitness[row];
  const f4u xn = row4(xnew + row * ldn, l, C);
  const f4u go = row4(gout + (int64_t)max(p, 0) * ldgo, l, C);
  const bool kept = p >= 0;
  const float dot = group16_sum(dot4(go, xn));
  if (4 * l < ldgn) *reinterpret_cast<f4u*>(gxnew + row * ldgn + 4 * l) = kept ? go * f : f4u{0.f, 0.f, 0.f, 0.f};      // (pads: zeros)
  if (l == 0) gfit[row] = kept ? dot : 0.f;
}

// The two halves of gather_scale_rows_bwd_v4_kernel as two launches AROUND the fitness backward (round 6).  x_out = x'[perm] f[perm] gives
// g_f = g_out . x' (needed first: LEConv's backward turns it into g_pqr) and g_x' = g_out f, to which g_pqr W3 is then ADDED
// (pqr = x' W3^T + b3).  Written by the first launch and updated by a [N,3]x[3,D] GEMM, g_x' was stored, read and stored again:
// 0.61 GB per pooling of 706 k rows; here the first launch stores g_f only and the second forms g_x' once: 0.41 GB.
__global__ __launch_bounds__(kBlock) void gather_rows_dot_kernel(const float* __restrict__ gout, int64_t ldgo, const float* __restrict__ xnew,
                                                                 int64_t ldn, const int32_t* __restrict__ slot, int64_t N, int C,
                                                                 float* __restrict__ gfit) {
  const int64_t row = ((int64_t)row_block() * kBlock + threadIdx.x) / kGroup;
  const int l = threadIdx.x % kGroup;
  if (row >= N) return;
  const int p = slot[row];
  const f4u xn = row4(xnew + row * ldn, l, C);
  const f4u go = row4(gout + (int64_t)max(p, 0) * ldgo, l, C);
  const float dot = group16_sum(dot4(go, xn));
  if (l == 0) gfit[row] = p >= 0 ? dot : 0.f;
}
// g_x'[row] = (kept ? g_out[slot[row]] f[row] : 0) + sum_t g3[row, t] w3[t, :]   (K <= 3 terms).  Persistent groups: a 16-lane group
// keeps its slice of w3 in registers and walks rows group, group + G, ... two at a time (a group per row loaded the twelve weights
// again for every row: 164 us for the two poolings of 64 100-qubit circuits where the bytes take 60).
__global__ __launch_bounds__(kBlock) void scatter_scale_rank_kernel(const float* __restrict__ gout, int64_t ldgo, const float* __restrict__ fitness,
                                                                    const int32_t* __restrict__ slot, const float* __restrict__ g3, int64_t ldg3,
                                                                    const float* __restrict__ w3, int K, int64_t N, int C,
                                                                    float* __restrict__ gxnew, int64_t ldgn) {
  const int64_t group = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / kGroup, groups = (int64_t)gridDim.x * (kBlock / kGroup);
  const int l = threadIdx.x % kGroup;
  const int last = (C - 1) >> 2, lc = min(l, last);
  const int nv = min(4, max(0, C - 4 * l));
  const bool writes = 4 * l < ldgn;
  f4u wt[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    wt[t] = f4u{0.f, 0.f, 0.f, 0.f};
    if (t < K) {                                         // (uniform) compact [K, C]: a slice one channel at a time
      const float* __restrict__ wr = w3 + (int64_t)t * C;
      const int c0 = 4 * lc;
      wt[t] = f4u{nv > 0 ? wr[min(c0, C - 1)] : 0.f, nv > 1 ? wr[min(c0 + 1, C - 1)] : 0.f, nv > 2 ? wr[min(c0 + 2, C - 1)] : 0.f,
                  nv > 3 ? wr[min(c0 + 3, C - 1)] : 0.f};
    }
  }
  for (int64_t r0 = group; r0 < N; r0 += 2 * groups) {
    int p[2];
    float f[2], gt[2][3];
    f4u go[2];
    bool live[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t row = r0 + u * groups;
      live[u] = row < N;
      const int64_t rc = min(row, N - 1);
      p[u] = slot[rc];
      f[u] = fitness[rc];
#pragma unroll
      for (int t = 0; t < 3; ++t) gt[u][t] = t < K ? g3[rc * ldg3 + t] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) go[u] = row4(gout + (int64_t)max(p[u], 0) * ldgo, l, C);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f4u v = p[u] >= 0 ? go[u] * f[u] : f4u{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 3; ++t) v += gt[u][t] * wt[t];      // (wt is zero past the row's channels: the pads come out zero)
      if (live[u] && writes) *reinterpret_cast<f4u*>(gxnew + (r0 + u * groups) * ldgn + 4 * l) = v;
    }
  }
}

// LEConv + sigmoid backward on scalars: from g_f and f build the gradient of pqr[N,3] = (p, q, r):
//   g_raw = g_f f (1 - f);  g_p[j] = g_raw[j] + sum_{e: j->i} g_raw[i];  g_q[i] = -(indeg_i + 1) g_raw[i];  g_r = g_raw.
__global__ __launch_bounds__(kBlock) void leconv_fitness_bwd_kernel(
    const float* __restrict__ gfit, const float* __restrict__ fitness, const int32_t* __restrict__ iptr,
    const int32_t* __restrict__ optr, const int32_t* __restrict__ odst, int64_t N, float* __restrict__ gpqr,
    const uint8_t* __restrict__ skip) {
  const int64_t j = (int64_t)row_block() * kBlock + threadIdx.x;
  if (j >= N) return;
  if (skip && skip[j]) return;                         // a long row: leconv_fitness_bwd_long_kernel
  auto graw = [&](int64_t i) { const float f = fitness[i]; return gfit[i] * f * (1.f - f); };
  const float gj = graw(j);
  float gp = gj;
  for_edge_chunks(optr[j], optr[j + 1], [&](int e, auto kc) {         // chunks fetched together, used in edge order
    constexpr int K = decltype(kc)::value;
    float gr[K];
#pragma unroll
    for (int u = 0; u < K; ++u) gr[u] = graw(odst[e + u]);
#pragma unroll
    for (int u = 0; u < K; ++u) gp += gr[u];
  });
  gpqr[j * 3] = gp;
  gpqr[j * 3 + 1] = -(float)(iptr[j + 1] - iptr[j] + 1) * gj;
  gpqr[j * 3 + 2] = gj;
}

// The same for the LONG rows of a coarsened graph (a thread per row walked a row of 150 out-entries alone: 100 us on the level-1
// graph of 64 100-qubit circuits): a wave per row of the OUT structure's dense-block plan (lrows: its rows of 32+ entries; row_flag:
// those a usable block holds -- exactly the rows the kernel above skips), lanes over the entries, one wave sum.
__global__ __launch_bounds__(kBlock) void leconv_fitness_bwd_long_kernel(
    const float* __restrict__ gfit, const float* __restrict__ fitness, const int32_t* __restrict__ iptr,
    const int32_t* __restrict__ optr, const int32_t* __restrict__ odst, const int32_t* __restrict__ lrows,
    const int32_t* __restrict__ counter, const uint8_t* __restrict__ row_flag, float* __restrict__ gpqr) {
  const int lane = threadIdx.x & 63;
  const int total = *counter, nwaves = gridDim.x * (kBlock / kWave);
  auto graw = [&](int64_t i) { const float f = fitness[i]; return gfit[i] * f * (1.f - f); };
  for (int k = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6); k < total; k += nwaves) {
    const int j = lrows[k];
    if (j < 0 || !row_flag[j]) continue;                 // (wave-uniform)
    const int beg = optr[j], end = optr[j + 1];
    float gp = 0.f;
    for (int e = beg + lane; e < end; e += kWave) gp += graw(odst[e]);
    for (int d = 32; d > 0; d >>= 1) gp += __shfl_xor(gp, d, 64);
    if (lane == 0) {
      const float gj = graw(j);
      gpqr[(int64_t)j * 3] = gp + gj;
      gpqr[(int64_t)j * 3 + 1] = -(float)(iptr[j + 1] - iptr[j] + 1) * gj;
      gpqr[(int64_t)j * 3 + 2] = gj;
    }
  }
}

// ------------------------------------------------------------------------------------- rank-k weight gradients of a pooling
// ASAPooling's backward ends in three tiny weight gradients over the SAME rows: gw3 [3, D] = gpqr^T x', g_att_x [1, D] = g_c^T x and
// g_w_comp [1, D] = g_a^T segmax(x) (+ their bias sums) -- weighted column sums of three [N, D] matrices.  As three calls of the
// general MFMA weight-gradient kernel they were six launches per pooling (a first stage and a fixed-order second stage each, 12 of
// a Family B step's 134) at a few per cent of the matrix rate.  Here ONE pass reads the three matrices once: thread (row group,
// 16-byte column group) adds its rows' products in registers, the row groups of a workgroup meet through LDS in a fixed order, one
// partial per workgroup; a second launch adds the partials in workgroup order (deterministic).
constexpr int kRankTermsMax = 3, kRankAccMax = 5, kRankGroupsMax = 1024;
struct RankGradArgs {
  const float* x[kRankTermsMax]; int64_t ldx[kRankTermsMax];
  const float* g[kRankTermsMax]; int64_t ldg[kRankTermsMax]; int k[kRankTermsMax];
  int terms; int64_t N; int CV;
  float* partial;            // [groups][acc][4 CV + 4]: the sums, then (in the first of the last four floats) the sum of the weights
};

__global__ __launch_bounds__(kBlock) void rank_grad_partial_kernel(const RankGradArgs a) {
  __shared__ float4 s_acc[kRankAccMax][kBlock];
  __shared__ float s_gsum[kRankAccMax][kBlock];
  const int tid = threadIdx.x;
  const int R = kBlock / a.CV;                        // rows of a trip
  const int rg = tid / a.CV, cg = tid - rg * a.CV;
  const bool on = rg < R;
  // (accumulators indexed by COMPILE-TIME (term, weight column): a running index that depends on k[] put the array in scratch memory
  // and the kernel at 0.9 ms; the running index is used where it addresses LDS)
  float4 acc[kRankTermsMax][3];
  float gs[kRankTermsMax][3];
#pragma unroll
  for (int t = 0; t < kRankTermsMax; ++t)
#pragma unroll
    for (int c = 0; c < 3; ++c) { acc[t][c] = make_float4(0.f, 0.f, 0.f, 0.f); gs[t][c] = 0.f; }
  // (round 6: two rows per thread and trip, all loads before the first product -- 140 against 132 us: 114 registers instead of 60)
  for (int64_t base = (int64_t)blockIdx.x * R; base < a.N; base += (int64_t)gridDim.x * R) {
    const int64_t row = base + rg;
    if (!on || row >= a.N) continue;
#pragma unroll
    for (int t = 0; t < kRankTermsMax; ++t) {
      if (t < a.terms) {
        const float4 xv = *reinterpret_cast<const float4*>(a.x[t] + row * a.ldx[t] + 4 * cg);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if (c < a.k[t]) {
            const float w = a.g[t][row * a.ldg[t] + c];
            acc[t][c].x = fmaf(w, xv.x, acc[t][c].x); acc[t][c].y = fmaf(w, xv.y, acc[t][c].y);
            acc[t][c].z = fmaf(w, xv.z, acc[t][c].z); acc[t][c].w = fmaf(w, xv.w, acc[t][c].w);
            if (cg == 0) gs[t][c] += w;
          }
        }
      }
    }
  }
  {
    int j = 0;
#pragma unroll
    for (int t = 0; t < kRankTermsMax; ++t)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        if (t < a.terms && c < a.k[t]) { s_acc[j][tid] = acc[t][c]; s_gsum[j][tid] = gs[t][c]; ++j; }
  }
  __syncthreads();
  int n_acc = 0;
  for (int t = 0; t < a.terms; ++t) n_acc += a.k[t];
  const int pitch = 4 * a.CV + 4;
  for (int o = tid; o < n_acc * (a.CV + 1); o += kBlock) {       // (accumulator, column group | the weight sum): the row groups in order
    const int j = o / (a.CV + 1), c = o - j * (a.CV + 1);
    float* dst = a.partial + ((int64_t)blockIdx.x * n_acc + j) * pitch;
    if (c < a.CV) {
      float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int r = 0; r < R; ++r) {
        const float4 v = s_acc[j][r * a.CV + c];
        tot.x += v.x; tot.y += v.y; tot.z += v.z; tot.w += v.w;
      }
      *reinterpret_cast<float4*>(dst + 4 * c) = tot;
    } else {
      float tot = 0.f;
      for (int r = 0; r < R; ++r) tot += s_gsum[j][r * a.CV];
      dst[4 * a.CV] = tot;
    }
  }
}

// out[j, 0 .. D) = the sum over the groups of the partials; bias[j] = the summed weights.  A workgroup per (accumulator j, chunk of 16
// partial columns): thread (slice of 16, column) adds the groups slice, slice + 16, ... on four independent chains (a thread per
// output walking all 1024 groups alone took 235 us), then the sixteen slices are added in order: deterministic.
__global__ __launch_bounds__(kBlock) void rank_grad_finish_kernel(const float* __restrict__ partial, int groups, int n_acc, int CV, int D,
                                                                  float* __restrict__ out, float* __restrict__ bias) {
  __shared__ float s_part[16][16];
  const int j = blockIdx.x, chunk = blockIdx.y;
  const int c = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int pitch = 4 * CV + 4;
  const int col = chunk * 16 + c;                     // of a partial row: [0, 4 CV) the sums, 4 CV the weight sum
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
  if (col < pitch) {
    const float* __restrict__ p = partial + (int64_t)j * pitch + col;
    const int64_t stride = (int64_t)n_acc * pitch;
    int g = sl;
    for (; g + 48 < groups; g += 64) {
      t0 += p[(int64_t)g * stride]; t1 += p[(int64_t)(g + 16) * stride];
      t2 += p[(int64_t)(g + 32) * stride]; t3 += p[(int64_t)(g + 48) * stride];
    }
    for (; g < groups; g += 16) t0 += p[(int64_t)g * stride];
  }
  s_part[sl][c] = (t0 + t1) + (t2 + t3);
  __syncthreads();
  if (sl == 0 && col < pitch) {
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += s_part[q][c];
    if (col < D) out[j * D + col] = tot;
    else if (col == 4 * CV) bias[j] = tot;
  }
}

}  // namespace mlqem

using namespace mlqem;

#define MLQEM_GRID(n) dim3((unsigned)ceil_div((n), kBlock)), dim3(kBlock), 0, as_stream(stream)

namespace mlqem {
// the four-channels-per-lane kernels for callers in other translation units (dense_block.hip: the rows its blocks do not serve)
// (FAST forms whenever the layout allows: attn_q4_fast)
#define MLQEM_Q4(KERNEL, FASTCOND)                                                                                                     \
  do {                                                                                                                                 \
    const bool fast_ = (FASTCOND);                                                                                                     \
    const dim3 grid_((unsigned)ceil_div(a.N * a.H * (a.C > 16 ? 8 : 4), kBlock));                                                      \
    if (a.C > 16 && fast_) hipLaunchKernelGGL((KERNEL<8, true>), grid_, dim3(kBlock), 0, stream, a);                                   \
    else if (a.C > 16) hipLaunchKernelGGL((KERNEL<8, false>), grid_, dim3(kBlock), 0, stream, a);                                      \
    else if (fast_) hipLaunchKernelGGL((KERNEL<4, true>), grid_, dim3(kBlock), 0, stream, a);                                          \
    else hipLaunchKernelGGL((KERNEL<4, false>), grid_, dim3(kBlock), 0, stream, a);                                                    \
  } while (0)
void launch_attn_train_q4(const AttnFwdArgs& a, hipStream_t stream) {
  MLQEM_Q4(transformer_attn_train_q4_kernel, attn_q4_fast(a.H, a.C, a.CP > 0 ? a.CP : a.C, INT64_MAX, a.idx));
}
void launch_attn_bwd_dst_q4(const AttnBwdArgs& a, hipStream_t stream) {
  MLQEM_Q4(transformer_attn_bwd_dst_q4_kernel, attn_q4_fast(a.H, a.C, a.CP > 0 ? a.CP : a.C, std::min(a.ldg, a.lda), a.idx));
}
void launch_attn_bwd_src_rc_q4(const AttnBwdArgs& a, hipStream_t stream) {
  MLQEM_Q4(transformer_attn_bwd_src_rc_q4_kernel, attn_q4_fast(a.H, a.C, a.CP > 0 ? a.CP : a.C, a.ldg, a.odst));
}
void launch_attn_bwd_src_q4(const AttnBwdArgs& a, hipStream_t stream) {
  MLQEM_Q4(transformer_attn_bwd_src_q4_kernel, attn_q4_fast(a.H, a.C, a.CP > 0 ? a.CP : a.C, a.ldg, a.odst) && a.oeid != nullptr);
}
}  // namespace mlqem

extern "C" int mlqem_transformer_attention_train_f32(const float* qkvs, int64_t ld, const int32_t* in_ptr,
                                                     const int32_t* in_src, const int32_t* loops, int64_t N, int64_t E,
                                                     int H, int C, float drop_p, uint64_t seed,
                                                     const uint64_t* seed_counter, int pair_key, const int32_t* in_ell, int head_pitch,
                                                     float* out, int64_t ldo, float* attn_out, int64_t lda, float* stat_m, float* stat_den,
                                                     mlqem_stream_t stream) {
  begin_launches();
  const int CP = head_pitch > 0 ? head_pitch : C;
  if (N < 0 || E < 0 || H <= 0 || C <= 0 || CP < C || ld < 4 * H * CP || ldo < H * C || lda < H * C || drop_p < 0.f || drop_p >= 1.f)
    return MLQEM_ERR_BAD_ARG;
  if (CP != C && !attn_q4_enabled()) return MLQEM_ERR_UNSUPPORTED;         // the one-channel-per-lane forms read compact parts only
  if (C > kAttnMaxC) return MLQEM_ERR_UNSUPPORTED;
  if (N == 0) return MLQEM_OK;
  if (!qkvs || !in_ptr || !out || !attn_out || !stat_m || !stat_den) return MLQEM_ERR_BAD_ARG;
  if (N > INT32_MAX) return MLQEM_ERR_UNSUPPORTED;
  if (pair_key && !attn_q4_enabled()) return MLQEM_ERR_UNSUPPORTED;      // the one-channel-per-lane forms key by position only
  if (in_ell && !aligned_to(in_ell, 8)) return MLQEM_ERR_BAD_ARG;
  const AttnFwdArgs a{qkvs, ld, in_ptr, in_src, loops, N, E, H, C, drop_p, seed, seed_counter, out, ldo, attn_out, lda, stat_m, stat_den,
                      pair_key ? 1 : 0, in_ell, CP};
  if (attn_q4_enabled()) launch_attn_train_q4(a, as_stream(stream));
  else if (C > kGroup) hipLaunchKernelGGL(transformer_attn_train_kernel<true>, MLQEM_GRID(N * H * kGroup), a);
  else hipLaunchKernelGGL(transformer_attn_train_kernel<false>, MLQEM_GRID(N * H * kGroup), a);
  return launch_status();
}

extern "C" int mlqem_transformer_attention_bwd_f32(const float* qkvs, int64_t ld, const float* g, int64_t ldg,
                                                   const float* attn_out, int64_t lda, const float* stat_m,
                                                   const float* stat_den, const int32_t* in_ptr, const int32_t* in_src,
                                                   const int32_t* out_ptr, const int32_t* out_dst,
                                                   const int32_t* out_eid, const int32_t* loops, int64_t N, int64_t E,
                                                   int H, int C, float drop_p, uint64_t seed, const uint64_t* seed_counter,
                                                   int pair_key, int head_pitch, float* gqkvs, int64_t ldq, float* edge_al,
                                                   float* edge_gs, mlqem_stream_t stream) {
  begin_launches();
  const int CP = head_pitch > 0 ? head_pitch : C;
  if (N < 0 || E < 0 || H <= 0 || C <= 0 || CP < C || ld < 4 * H * CP || ldq < 4 * H * CP || ldg < H * C || lda < H * C)
    return MLQEM_ERR_BAD_ARG;
  if (CP != C && !attn_q4_enabled()) return MLQEM_ERR_UNSUPPORTED;
  if (C > kAttnMaxC) return MLQEM_ERR_UNSUPPORTED;
  if (N == 0) return MLQEM_OK;
  const bool recompute = out_eid == nullptr;             // no out_eid: the source side recomputes its weights (edge_al: [N H] floats)
  if (recompute && !aligned_to(edge_al, 16)) return MLQEM_ERR_BAD_ARG;       // the per-(row, head) records are 16-byte stores
  if (!qkvs || !g || !attn_out || !stat_m || !stat_den || !in_ptr || !out_ptr || !gqkvs || !edge_al || (!recompute && !edge_gs))
    return MLQEM_ERR_BAD_ARG;
  if (E > 0 && (!in_src || !out_dst)) return MLQEM_ERR_BAD_ARG;
  if (recompute && drop_p > 0.f && !pair_key) return MLQEM_ERR_BAD_ARG;   // a position-keyed draw cannot be found from the source side
  if ((recompute || pair_key) && !attn_q4_enabled()) return MLQEM_ERR_UNSUPPORTED;
  if (N > INT32_MAX) return MLQEM_ERR_UNSUPPORTED;
  const AttnBwdArgs a{qkvs, ld, g, ldg, attn_out, lda, stat_m, stat_den, in_ptr, in_src, out_ptr, out_dst, out_eid, loops,
                      N, E, H, C, drop_p, seed, seed_counter, gqkvs, ldq, edge_al, edge_gs, pair_key ? 1 : 0, CP};
  if (recompute) {
    launch_attn_bwd_dst_q4(a, as_stream(stream));
    launch_attn_bwd_src_rc_q4(a, as_stream(stream));
  } else if (attn_q4_enabled()) {
    launch_attn_bwd_dst_q4(a, as_stream(stream));
    launch_attn_bwd_src_q4(a, as_stream(stream));
  } else if (C > kGroup) {
    hipLaunchKernelGGL(transformer_attn_bwd_dst_kernel<true>, MLQEM_GRID(N * H * kGroup), a);
    hipLaunchKernelGGL(transformer_attn_bwd_src_kernel<true>, MLQEM_GRID(N * H * kGroup), a);
  } else {
    hipLaunchKernelGGL(transformer_attn_bwd_dst_kernel<false>, MLQEM_GRID(N * H * kGroup), a);
    hipLaunchKernelGGL(transformer_attn_bwd_src_kernel<false>, MLQEM_GRID(N * H * kGroup), a);
  }
  return launch_status();
}

namespace mlqem {
// mlqem_csr_softmax_aggregate_bwd_f32 for callers that serve some rows themselves (dense_pool.hip): skip_in / skip_out flag the rows
// the destination-side / source-side launch leaves alone; parts: 1 = the destination-side launch, 2 = the source-side launch
int softmax_aggregate_bwd_launches(const float* x, int64_t ldx, const float* xnew, int64_t ldn, const float* gnew, int64_t ldg,
                                   const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                                   const int32_t* out_eid, const float* a_dst, const float* c_src, float negative_slope, int64_t N, int64_t E,
                                   int C, int accumulate, float* gx, int64_t ldgx, float* g_a, float* g_c, float* edge_al, float* edge_gp,
                                   const float* xmax, int64_t ldm, float* tie_count, int64_t ldt, const float* gx_rank1,
                                   const uint8_t* skip_in, const uint8_t* skip_out, int parts, mlqem_stream_t stream,
                                   const float* fuse_max_col) {
  if (N < 0 || E < 0 || C <= 0 || ldx < C || ldn < C || ldg < C || ldgx < C) return MLQEM_ERR_BAD_ARG;
  if (tie_count && (!xmax || ldm < C || ldt < C)) return MLQEM_ERR_BAD_ARG;
  if (tie_count && C > 128) return MLQEM_ERR_UNSUPPORTED;      // the any-width form does not count
  if (N == 0) return MLQEM_OK;
  // no out_eid: the source side recomputes its weights; edge_al then holds one 16-byte record per row ([4 N] floats, 16-byte
  // aligned) and edge_gp is not used
  const bool recompute = out_eid == nullptr;
  if (!x || !xnew || !gnew || !in_ptr || !out_ptr || !a_dst || !c_src || !gx || !g_a || !g_c || !edge_al || (!recompute && !edge_gp))
    return MLQEM_ERR_BAD_ARG;
  if (E > 0 && (!in_src || !out_dst)) return MLQEM_ERR_BAD_ARG;
  if (recompute) {
    if (C > 128) return MLQEM_ERR_UNSUPPORTED;
    if (!aligned_to(edge_al, 16)) return MLQEM_ERR_BAD_ARG;
    edge_gp = nullptr;
  }
  if (gx_rank1 && C > 128) return MLQEM_ERR_UNSUPPORTED;      // the any-width source side does not add it
  if ((skip_in || skip_out) && C > 128) return MLQEM_ERR_UNSUPPORTED;
  // the segment max's backward inside the source-side walk: the stored form only, with the tie counts of this call's destination side
  if (fuse_max_col && (recompute || !tie_count || !xmax || C > 128)) return MLQEM_ERR_BAD_ARG;
  if (parts & 1) {
#define MLQEM_SAB(NV)                                                                                                                          \
  do {                                                                                                                                         \
    if (tie_count)                                                                                                                             \
      hipLaunchKernelGGL((softmax_aggregate_bwd_dst_kernel<NV, true>), MLQEM_GRID(N * kGroup), x, ldx, xnew, ldn, gnew, ldg, in_ptr, in_src,   \
                         a_dst, c_src, negative_slope, N, E, C, edge_al, edge_gp, g_a, xmax, ldm, tie_count, ldt, skip_in);                   \
    else                                                                                                                                       \
      hipLaunchKernelGGL((softmax_aggregate_bwd_dst_kernel<NV, false>), MLQEM_GRID(N * kGroup), x, ldx, xnew, ldn, gnew, ldg, in_ptr, in_src,  \
                         a_dst, c_src, negative_slope, N, E, C, edge_al, edge_gp, g_a, nullptr, 0, nullptr, 0, skip_in);                      \
  } while (0)
  if (C <= 16) MLQEM_SAB(1);
  else if (C <= 32) MLQEM_SAB(2);
  else if (C <= 48) MLQEM_SAB(3);
  else if (C <= 64) MLQEM_SAB(4);
  else if (C <= 128) MLQEM_SAB(8);
  else
    hipLaunchKernelGGL(softmax_aggregate_bwd_dst_any_width_kernel, MLQEM_GRID(N * kGroup), x, ldx, xnew, ldn, gnew, ldg, in_ptr,
                       in_src, a_dst, c_src, negative_slope, N, E, C, edge_al, edge_gp, g_a);
#undef MLQEM_SAB
  }
  if (!(parts & 2)) return launch_status();
#define MLQEM_SAS(NV)                                                                                                                    \
  do {                                                                                                                                   \
    if (recompute)                                                                                                                       \
      hipLaunchKernelGGL(softmax_aggregate_bwd_src_rc_kernel<NV>, MLQEM_GRID(N * kGroup), x, ldx, gnew, ldg, out_ptr, out_dst,           \
                         reinterpret_cast<const float4*>(edge_al), c_src, negative_slope, N, C, accumulate, gx, ldgx, g_c, gx_rank1,     \
                         skip_out);                                                                                                       \
    else                                                                                                                                 \
      if (fuse_max_col)                                                                                                                  \
        hipLaunchKernelGGL((softmax_aggregate_bwd_src_kernel<NV, true>), MLQEM_GRID(N * kGroup), gnew, ldg, out_ptr, out_dst, out_eid,   \
                           edge_al, edge_gp, N, E, C, accumulate, gx, ldgx, g_c, gx_rank1,                                               \
                           MaxFuse{x, ldx, xmax, ldm, tie_count, ldt, g_a, fuse_max_col});                                               \
      else                                                                                                                               \
        hipLaunchKernelGGL((softmax_aggregate_bwd_src_kernel<NV, false>), MLQEM_GRID(N * kGroup), gnew, ldg, out_ptr, out_dst, out_eid,  \
                           edge_al, edge_gp, N, E, C, accumulate, gx, ldgx, g_c, gx_rank1, MaxFuse{});                                   \
  } while (0)
  if (C <= 16) MLQEM_SAS(1);
  else if (C <= 32) MLQEM_SAS(2);
  else if (C <= 48) MLQEM_SAS(3);
  else if (C <= 64) MLQEM_SAS(4);
  else if (C <= 128) MLQEM_SAS(8);
  else
    hipLaunchKernelGGL(softmax_aggregate_bwd_src_any_width_kernel, MLQEM_GRID(N * C), gnew, ldg, out_ptr, out_dst, out_eid,
                       edge_al, edge_gp, N, E, C, accumulate, gx, ldgx, g_c);
#undef MLQEM_SAS
  return launch_status();
}
}  // namespace mlqem

extern "C" int mlqem_csr_softmax_aggregate_bwd_f32(const float* x, int64_t ldx, const float* xnew, int64_t ldn,
                                                   const float* gnew, int64_t ldg, const int32_t* in_ptr,
                                                   const int32_t* in_src, const int32_t* out_ptr,
                                                   const int32_t* out_dst, const int32_t* out_eid, const float* a_dst,
                                                   const float* c_src, float negative_slope, int64_t N, int64_t E,
                                                   int C, int accumulate, float* gx, int64_t ldgx, float* g_a,
                                                   float* g_c, float* edge_al, float* edge_gp, const float* xmax, int64_t ldm,
                                                   float* tie_count, int64_t ldt, const float* gx_rank1, const float* fuse_max_col,
                                                   mlqem_stream_t stream) {
  begin_launches();
  return softmax_aggregate_bwd_launches(x, ldx, xnew, ldn, gnew, ldg, in_ptr, in_src, out_ptr, out_dst, out_eid, a_dst, c_src, negative_slope,
                                        N, E, C, accumulate, gx, ldgx, g_a, g_c, edge_al, edge_gp, xmax, ldm, tie_count, ldt, gx_rank1,
                                        nullptr, nullptr, 3, stream, fuse_max_col);
}

static void launch_share_from_counts(const float* gmax, int64_t ldg, const float* cnt, int64_t ldc, int64_t N, int C, float* gshare,
                                     int64_t lds, mlqem_stream_t stream, const float* g_row = nullptr, const float* g_col = nullptr) {
  const int c4 = (C + 3) / 4 * 4;
  const bool vec = (g_row || (ldg % 4 == 0 && ldg >= c4 && aligned_to(gmax, 16))) && ldc % 4 == 0 && lds % 4 == 0 && ldc >= c4 && lds >= c4 &&
                   aligned_to(cnt, 16) && aligned_to(gshare, 16);
  if (vec) hipLaunchKernelGGL(segment_max_share_from_counts_kernel<4>, MLQEM_GRID(N * (c4 / 4)), gmax, ldg, cnt, ldc, N, C, gshare, lds, g_row, g_col);
  else hipLaunchKernelGGL(segment_max_share_from_counts_kernel<1>, MLQEM_GRID(N * C), gmax, ldg, cnt, ldc, N, C, gshare, lds, g_row, g_col);
}

namespace mlqem {
int segment_max_bwd_launches(const float* x, int64_t ldx, const float* xmax, int64_t ldm, const float* gmax, int64_t ldg, const int32_t* in_ptr,
                             const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst, int64_t N, int C, float* gx, int64_t ldgx,
                             float* gshare, int64_t lds, const float* tie_count, int64_t ldt, const float* gmax_row, const float* gmax_col,
                             const uint8_t* skip_out, mlqem_stream_t stream);
}
extern "C" int mlqem_csr_segment_max_bwd_f32(const float* x, int64_t ldx, const float* xmax, int64_t ldm,
                                             const float* gmax, int64_t ldg, const int32_t* in_ptr,
                                             const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                                             int64_t N, int C, float* gx, int64_t ldgx, float* gshare, int64_t lds,
                                             const float* tie_count, int64_t ldt, const float* gmax_row, const float* gmax_col,
                                             mlqem_stream_t stream) {
  begin_launches();
  return segment_max_bwd_launches(x, ldx, xmax, ldm, gmax, ldg, in_ptr, in_src, out_ptr, out_dst, N, C, gx, ldgx, gshare, lds, tie_count, ldt,
                                  gmax_row, gmax_col, nullptr, stream);
}
int mlqem::segment_max_bwd_launches(const float* x, int64_t ldx, const float* xmax, int64_t ldm, const float* gmax, int64_t ldg,
                                    const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst, int64_t N,
                                    int C, float* gx, int64_t ldgx, float* gshare, int64_t lds, const float* tie_count, int64_t ldt,
                                    const float* gmax_row, const float* gmax_col, const uint8_t* skip_out, mlqem_stream_t stream) {
  const bool rank1 = gmax_row != nullptr;             // gmax = gmax_row (x) gmax_col (needs tie_count: the counted form); gmax unused
  if (N < 0 || C <= 0 || ldx < C || ldm < C || (!rank1 && ldg < C) || ldgx < C || lds < C || (tie_count && ldt < C)) return MLQEM_ERR_BAD_ARG;
  if (rank1 && (!gmax_col || !tie_count)) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!x || !xmax || (!rank1 && !gmax) || !in_ptr || !out_ptr || !gx || !gshare) return MLQEM_ERR_BAD_ARG;
  if (tie_count && C > 128) return MLQEM_ERR_UNSUPPORTED;
#define MLQEM_SMB(NV)                                                                                                        \
  do {                                                                                                                       \
    if (tie_count)                                                                                                           \
      launch_share_from_counts(gmax, ldg, tie_count, ldt, N, C, gshare, lds, stream, gmax_row, gmax_col);                     \
    else                                                                                                                     \
    hipLaunchKernelGGL(segment_max_share_kernel<NV>, MLQEM_GRID(N * kGroup), x, ldx, xmax, ldm, gmax, ldg, in_ptr, in_src, N, \
                       C, gshare, lds);                                                                                      \
    hipLaunchKernelGGL(segment_max_bwd_kernel<NV>, MLQEM_GRID(N * kGroup), x, ldx, xmax, ldm, gshare, lds, out_ptr, out_dst,  \
                       N, C, gx, ldgx, skip_out);                                                                            \
  } while (0)
  if (C <= 16) MLQEM_SMB(1);
  else if (C <= 32) MLQEM_SMB(2);
  else if (C <= 48) MLQEM_SMB(3);
  else if (C <= 64) MLQEM_SMB(4);
  else if (C <= 128) MLQEM_SMB(8);
  else {
    hipLaunchKernelGGL(segment_max_share_any_width_kernel, MLQEM_GRID(N * C), x, ldx, xmax, ldm, gmax, ldg, in_ptr, in_src, N,
                       C, gshare, lds);
    hipLaunchKernelGGL(segment_max_bwd_any_width_kernel, MLQEM_GRID(N * C), x, ldx, xmax, ldm, gshare, lds, out_ptr, out_dst,
                       N, C, gx, ldgx);
  }
#undef MLQEM_SMB
  return launch_status();
}

extern "C" int mlqem_gather_scale_rows_bwd_f32(const float* gout, int64_t ldgo, const float* xnew, int64_t ldn,
                                               const float* fitness, const int32_t* slot, int64_t N, int C,
                                               float* gxnew, int64_t ldgn, float* gfit, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || C <= 0 || ldgo < C || ldn < C || ldgn < C) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!xnew || !fitness || !slot || !gxnew || !gfit) return MLQEM_ERR_BAD_ARG;
  const int c4 = (C + 3) / 4 * 4;
  auto padded = [&](const float* q, int64_t ld) { return q && ld % 4 == 0 && ld >= c4 && aligned_to(q, 16); };
  if (C <= 64 && padded(gout, ldgo) && padded(xnew, ldn) && padded(gxnew, ldgn)) {
    hipLaunchKernelGGL(gather_scale_rows_bwd_v4_kernel, MLQEM_GRID(N * kGroup), gout, ldgo, xnew, ldn, fitness, slot, N, C, gxnew, ldgn, gfit);
    return launch_status();
  }
  hipLaunchKernelGGL(gather_scale_rows_bwd_kernel, MLQEM_GRID(N * kGroup), gout, ldgo, xnew, ldn, fitness, slot, N, C, gxnew,
                     ldgn, gfit);
  return launch_status();
}

extern "C" int mlqem_gather_rows_dot_f32(const float* gout, int64_t ldgo, const float* xnew, int64_t ldn, const int32_t* slot, int64_t N, int C,
                                         float* gfit, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || C <= 0 || ldgo < C || ldn < C) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!gout || !xnew || !slot || !gfit) return MLQEM_ERR_BAD_ARG;
  const int c4 = (C + 3) / 4 * 4;
  auto padded = [&](const float* q, int64_t ld) { return ld % 4 == 0 && ld >= c4 && aligned_to(q, 16); };
  if (C > 64 || !padded(gout, ldgo) || !padded(xnew, ldn)) return MLQEM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(gather_rows_dot_kernel, MLQEM_GRID(N * kGroup), gout, ldgo, xnew, ldn, slot, N, C, gfit);
  return launch_status();
}

extern "C" int mlqem_scatter_scale_rank_f32(const float* gout, int64_t ldgo, const float* fitness, const int32_t* slot, const float* g3,
                                            int64_t ldg3, const float* w3, int K, int64_t N, int C, float* gxnew, int64_t ldgn,
                                            mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || C <= 0 || K < 0 || K > 3 || ldgo < C || ldgn < C || ldg3 < K) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!gout || !fitness || !slot || !gxnew || (K > 0 && (!g3 || !w3))) return MLQEM_ERR_BAD_ARG;
  const int c4 = (C + 3) / 4 * 4;
  auto padded = [&](const float* q, int64_t ld) { return ld % 4 == 0 && ld >= c4 && aligned_to(q, 16); };
  if (C > 64 || !padded(gout, ldgo) || !padded(gxnew, ldgn)) return MLQEM_ERR_UNSUPPORTED;
  const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(N * kGroup, kBlock), 256 * 16);      // persistent 16-lane groups
  hipLaunchKernelGGL(scatter_scale_rank_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), gout, ldgo, fitness, slot, g3, ldg3, w3, K, N,
                     C, gxnew, ldgn);
  return launch_status();
}

extern "C" int mlqem_leconv_fitness_bwd_f32(const float* gfit, const float* fitness, const int32_t* in_ptr,
                                            const int32_t* out_ptr, const int32_t* out_dst, int64_t N, float* gpqr,
                                            mlqem_stream_t stream) {
  begin_launches();
  if (N < 0) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!gfit || !fitness || !in_ptr || !out_ptr || !gpqr) return MLQEM_ERR_BAD_ARG;
  hipLaunchKernelGGL(leconv_fitness_bwd_kernel, MLQEM_GRID(N), gfit, fitness, in_ptr, out_ptr, out_dst, N, gpqr, nullptr);
  return launch_status();
}

extern "C" int mlqem_dense_leconv_fitness_bwd_f32(const float* gfit, const float* fitness, const int32_t* in_ptr, const int32_t* out_ptr,
                                                  const int32_t* out_dst, int64_t N, const int32_t* lrows, const int32_t* counter,
                                                  const uint8_t* row_flag, int64_t max_blocks, float* gpqr, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || max_blocks <= 0) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!gfit || !fitness || !in_ptr || !out_ptr || !out_dst || !lrows || !counter || !row_flag || !gpqr) return MLQEM_ERR_BAD_ARG;
  hipLaunchKernelGGL(leconv_fitness_bwd_kernel, MLQEM_GRID(N), gfit, fitness, in_ptr, out_ptr, out_dst, N, gpqr, row_flag);
  const int64_t waves = max_blocks * 16;                 // an upper bound on the plan's rows
  hipLaunchKernelGGL(leconv_fitness_bwd_long_kernel, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(waves, kBlock / kWave), 8192))),
                     dim3(kBlock), 0, as_stream(stream), gfit, fitness, in_ptr, out_ptr, out_dst, lrows, counter, row_flag, gpqr);
  return launch_status();
}

// Up to three weighted column sums over the same N rows in one pass (see rank_grad_partial_kernel): term t has k[t] <= 3 weight columns
// g[t] [N, ldg[t]] and a matrix x[t] [N, ldx[t]] of D columns in 16-byte rows; out [sum k, D] row-major in term order, bias [sum k] the
// sums of the weight columns.  workspace: mlqem_rank_grad_workspace_bytes(D) bytes.
extern "C" size_t mlqem_rank_grad_workspace_bytes(int D) {
  if (D <= 0) return 0;
  return (size_t)kRankGroupsMax * kRankAccMax * (size_t)(((D + 3) / 4) * 4 + 4) * sizeof(float);
}

extern "C" int mlqem_rank_grad_f32(int terms, const float* const* x, const int64_t* ldx, const float* const* g, const int64_t* ldg, const int* k,
                                   int64_t N, int D, float* out, float* bias, void* workspace, size_t workspace_bytes, mlqem_stream_t stream) {
  begin_launches();
  if (terms < 1 || terms > kRankTermsMax || N < 0 || D <= 0 || !x || !ldx || !g || !ldg || !k || !out || !bias) return MLQEM_ERR_BAD_ARG;
  const int cv = (D + 3) / 4;
  if (cv > 64) return MLQEM_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < mlqem_rank_grad_workspace_bytes(D)) return MLQEM_ERR_WORKSPACE;
  RankGradArgs a{};
  int n_acc = 0;
  for (int t = 0; t < terms; ++t) {
    if (k[t] < 1 || k[t] > 3 || ldx[t] < 4 * cv || ldx[t] % 4 || ldg[t] < k[t] || (N > 0 && (!x[t] || !g[t] || !aligned_to(x[t], 16)))) return MLQEM_ERR_BAD_ARG;
    a.x[t] = x[t]; a.ldx[t] = ldx[t]; a.g[t] = g[t]; a.ldg[t] = ldg[t]; a.k[t] = k[t];
    n_acc += k[t];
  }
  if (n_acc > kRankAccMax) return MLQEM_ERR_UNSUPPORTED;
  a.terms = terms; a.N = N; a.CV = cv; a.partial = static_cast<float*>(workspace);
  const int rows = kBlock / cv;
  // a workgroup per `rows` rows up to the cap: a small batch's rows (32 four-qubit circuits: 2-3 k) in ONE trip per workgroup -- eight
  // trips each, a dependent round trip apiece, were 12.6 us of a 0.4 ms step, twice
  const int groups = (int)std::max<int64_t>(1, std::min<int64_t>(kRankGroupsMax, ceil_div(std::max<int64_t>(N, 1), (int64_t)rows)));
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(rank_grad_partial_kernel, dim3((unsigned)groups), dim3(kBlock), 0, s, a);
  hipLaunchKernelGGL(rank_grad_finish_kernel, dim3((unsigned)n_acc, (unsigned)ceil_div((int64_t)(4 * cv + 4), (int64_t)16)), dim3(kBlock), 0, s,
                     a.partial, groups, n_acc, cv, D, out, bias);
  return launch_status();
}
