// TransformerConv's edge softmax, forward (docs/tutorials/gnn.py:80-91; PyG semantics in SURVEY appendix B.1), shared by
// the inference kernel (attn.hip) and the training kernel (family_b_bwd.hip).
//
//   out[i, h] = sum_e alpha_e v[src_e, h] + skip[i, h],  alpha = dropout(softmax_e(q[i,h] . k[src_e,h] / sqrt(C)))
//
// over the CSR in-edges of row i followed by its self-loop entry (multiplicity loops[i]); denominator + 1e-16 as in PyG.
//
// One 16-lane group owns a (row, head): lane l holds channel l (and l + 16 in the WIDE instantiation, C > 16), a key /
// value row segment is one coalesced 64-byte read, q.k is a cross-lane sum.  Two row shapes:
//
//  * SHORT rows (<= kAttnShort entries: every node of a circuit DAG except its barriers): all key AND value rows are
//    fetched together, scores stay in registers.
//  * LONGER rows (barrier nodes; the coarsened graphs ASAPooling makes of 100-qubit circuits, whose rows have 100-500
//    in-edges and hold 95 % of a batch's edges): the group walks the row eight edges at a time -- index entries, then eight
//    key and eight value rows in flight together -- in ONE pass: a running maximum that grows rescales what was summed
//    before it (the three-pass form gathered every key row three times).  The kernel is bound by the vector ALU, not by
//    memory, so the scalar work of an edge (exp, dropout draw) is done by one lane per edge and broadcast (absorb8).
//
// The un-normalised weights exp(s - m) are summed and the sum divided once (PyG normalises every weight first: the same
// value in another rounding).  The dropout mask is keyed by (seed, in-CSR position, head) as in the backward kernels and
// drawn by a 32-bit hash (common.hpp: uniform01_edge).
//
// Measured and dropped (scripts/attn_micro.py; cfg2 level 0 / 100-qubit level 1, forward): the four groups of a wave
// walking one long row together, one row after the other (209 / 921 us against 145 / 696 us for this form: rows of a wave
// then run serially, and keeping the wave together to the end costs the short rows their early exit); source ids from the
// ELL side table (no gain: the kernel does not wait on that round trip); value rows fetched after the scores (244 us).
#pragma once

#include "common.hpp"

namespace mlqem {

constexpr int kAttnMaxC = 32;   // channels per head held in registers (reference models: 15 and 25)
constexpr int kAttnShort = 6;   // entries (in-edges + self) of a row handled from registers

struct AttnFwdArgs {
  const float* qkvs; int64_t ld;            // [N, 4*H*C] = [query | key | value | skip]
  const int32_t* ptr; const int32_t* idx;   // CSR by destination
  const int32_t* loops;                     // optional [N]: multiplicity of the self-loop entry
  int64_t N, E; int H, C; float drop_p; uint64_t seed;
  const uint64_t* seed_counter;             // optional device-resident step counter mixed into the seed (hipGraph replay)
  float* out; int64_t ldo;
  float* attn_out; int64_t lda; float* stat_m; float* stat_den;   // training only
  int pair_key;                             // dropout draws keyed by (destination, head, source) instead of the in-CSR position
  const int32_t* ell;                       // optional [N,2] side table (mlqem_ell_from_csr): the first two in-edges of every row, so
                                            // that a row of at most two (a circuit DAG's, but for barriers) goes ell -> key / value
                                            // rows instead of ptr -> idx -> key / value rows (four-channels-per-lane kernels only)
  int CP;                                   // channel pitch of a head inside the parts of qkvs (0 = C: compact).  16 for the 15
                                            // channels of the reference's models: every key / value / query segment is then an aligned
                                            // 16-byte-per-lane, 64-byte-per-head piece (pads are zeros the projection's padded weights
                                            // produce); out / attn_out stay compact [N, H C] (four-channels-per-lane kernels only)
  const uint8_t* skip = nullptr;            // optional [N]: rows served by a dense block (dense_block.hpp) are left alone
};

// Arguments of the backward kernels (family_b_bwd.hip, tile_attn.hip).
struct AttnBwdArgs {
  const float* qkvs; int64_t ld; const float* g; int64_t ldg; const float* attn_out; int64_t lda;
  const float* stat_m; const float* stat_den;
  const int32_t* ptr; const int32_t* idx; const int32_t* optr; const int32_t* odst; const int32_t* oeid; const int32_t* loops;
  int64_t N, E; int H, C; float drop_p; uint64_t seed; const uint64_t* seed_counter;
  float* gqkvs; int64_t ldq; float* edge_al; float* edge_gs;
  int pair_key;                   // as the forward's (attn_fwd.hpp)
  int CP;                         // as the forward's: channel pitch of a head inside the parts of qkvs AND gqkvs (0 = C)
  const uint8_t* skip_dst = nullptr;   // optional [N]: rows whose destination side / source side a dense block serves (dense_block.hpp)
  const uint8_t* skip_src = nullptr;
};
// oeid == nullptr selects the RECOMPUTING source side (transformer_attn_bwd_src_rc_q4_kernel): the destination side then files
// delta[N, H] = g . attn_out per (row, head) in edge_al (its first N H floats) and writes nothing per edge.

// The key of an attention weight's dropout draw.  By position (the default): (in-CSR position, head), self entries at E + row.
// By pair: (destination row, head, source row) -- the same number from either end of an edge, so a source-side pass needs no map
// from its out-entries to in-CSR positions (graphs without parallel edges only: parallel edges would share a draw).
__device__ __forceinline__ uint64_t attn_drop_key(bool pair_key, int64_t pos, int H, int h, int dst, int src) {
  return pair_key ? ((uint64_t)((int64_t)dst * H + h) << 32) | (uint32_t)src : (uint64_t)(pos * H + h);
}

// Draws under the pair key: ONE 32-bit hash per (destination, source) serves two heads (its 16-bit halves; heads 2k and 2k + 1 share
// hash k), so a kernel that handles an entry for all heads draws once -- the draw had been a quarter of the per-entry vector
// instructions (five quarter-rate 32-bit multiplies per head).  A weight is dropped when its 16-bit field is below floor(p * 65536)
// (|P(drop) - p| < 1.6e-5, as common.hpp's dropout_keep).
__device__ __forceinline__ uint32_t attn_pair_hash(uint64_t seed, int dst, int src, int hpair) {
  uint32_t x = avalanche32((uint32_t)src ^ (uint32_t)seed);
  return avalanche32(x ^ (uint32_t)(seed >> 32) ^ ((uint32_t)dst * 0x9E3779B1u) ^ ((uint32_t)hpair * 0x85EBCA6Bu));
}
__device__ __forceinline__ bool attn_pair_dropped(uint32_t hash, int h, uint32_t thr16) { return ((hash >> (16 * (h & 1))) & 0xFFFFu) < thr16; }
__device__ __forceinline__ uint32_t attn_drop_threshold(float p) { return (uint32_t)(p * 65536.f); }
// whether the weight of (destination dst, head h, source src) -- in-CSR position pos -- is dropped, under either key
__device__ __forceinline__ bool attn_dropped(uint64_t seed, bool pair_key, int64_t pos, int H, int h, int dst, int src, float p) {
  if (pair_key) return attn_pair_dropped(attn_pair_hash(seed, dst, src, h >> 1), h, attn_drop_threshold(p));
  return uniform01_edge(seed, (uint64_t)(pos * H + h)) < p;
}

template <bool TRAIN, bool WIDE> __device__ __forceinline__ void attn_forward(const AttnFwdArgs& a) {
  const int64_t t = ((int64_t)row_block() * kBlock + threadIdx.x) / kGroup;
  const int l = threadIdx.x % kGroup;
  const int H = a.H, C = a.C, HC = H * C;
  if (t >= a.N * H) return;                             // a whole group leaves together
  const int row = (int)(t / H);
  const int h = (int)(t - (int64_t)row * H);
  const float scale = 1.0f / sqrtf((float)C);
  const float keep = 1.f / (1.f - a.drop_p);
  const uint64_t seed = a.seed + ((TRAIN && a.seed_counter) ? *a.seed_counter * 0xD1B54A32D192ED03ull : 0ull);
  const bool c0 = l < C, c1 = WIDE && l + kGroup < C;
  const int l1 = c1 ? l + kGroup : l;                   // a valid channel for the second load of the narrow case (unused)
  const float* __restrict__ qkvs = a.qkvs;
  const int64_t ld = a.ld;
  const int32_t* __restrict__ idx = a.idx;

  // everything that depends on the row number only, fetched together
  const float* __restrict__ qi = qkvs + (int64_t)row * ld + h * C;
  const float q0 = c0 ? qi[l] : 0.f, q1 = c1 ? qi[l1] : 0.f;
  const int beg = a.ptr[row];
  const int end = a.ptr[row + 1];
  const int n_self = a.loops ? a.loops[row] : 0;
  const int deg = end - beg;
  const int cnt = deg + (n_self > 0 ? 1 : 0);

  float m = -INFINITY, denom = 0.f, a0 = 0.f, a1 = 0.f;
  auto dot = [&](float qa, float qb, float k0, float k1) {
    float s = qa * k0;
    if (WIDE) s = fmaf(qb, k1, s);
    return group16_sum(s) * scale;
  };
  auto weight = [&](float p, int64_t pos, int hh) {     // the weight a value row gets: dropout on the attention weight
    if (TRAIN && a.drop_p > 0.f) return uniform01_edge(seed, (uint64_t)(pos * H + hh)) < a.drop_p ? 0.f : p * keep;
    return p;
  };
  // One chunk of up to eight edges e .. e + k - 1 of the row, absorbed into the running (max, denominator, sums).  The
  // scalar work of an edge -- its source id, exp, dropout draw -- is done by ONE lane (lanes u and u + 8 own edge u); the
  // ids and the weights reach the other lanes by DPP row broadcasts when the rows are fetched / accumulated, and only the
  // dot product q.k of an edge involves the whole group.
  auto absorb8 = [&](int e, int k) {
    const int lu = l & 7;
    const int j = idx[e + min(lu, k - 1)];              // past the end: the last edge again (weight 0)
    int ju[8];
    float k0[8], k1[WIDE ? 8 : 1], v0[8], v1[WIDE ? 8 : 1];
    ju[0] = group16_bcast<0>(j); ju[1] = group16_bcast<1>(j); ju[2] = group16_bcast<2>(j); ju[3] = group16_bcast<3>(j);
    ju[4] = group16_bcast<4>(j); ju[5] = group16_bcast<5>(j); ju[6] = group16_bcast<6>(j); ju[7] = group16_bcast<7>(j);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float* __restrict__ kj = qkvs + (int64_t)ju[u] * ld + HC + h * C;
      k0[u] = c0 ? kj[l] : 0.f;
      v0[u] = c0 ? kj[HC + l] : 0.f;
      if (WIDE) {
        k1[u] = c1 ? kj[l1] : 0.f;
        v1[u] = c1 ? kj[HC + l1] : 0.f;
      }
    }
    float mys = -INFINITY;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float s = dot(q0, q1, k0[u], WIDE ? k1[u] : 0.f);
      if (lu == u) mys = s;
    }
    if (lu >= k) mys = -INFINITY;
    const float cm = group16_max(mys);
    if (cm > m) {                                        // the running maximum grows: rescale what was summed under the old one
      const float r = expf(m - cm);                      // exp(-inf) = 0 the first time
      denom *= r; a0 *= r; a1 *= r;
      m = cm;
    }
    const float p = lu < k ? expf(mys - m) : 0.f;
    denom += group16_sum(l < 8 ? p : 0.f);
    const float w = weight(p, (int64_t)e + lu, h);
    float wu[8];
    wu[0] = group16_bcast<0>(w); wu[1] = group16_bcast<1>(w); wu[2] = group16_bcast<2>(w); wu[3] = group16_bcast<3>(w);
    wu[4] = group16_bcast<4>(w); wu[5] = group16_bcast<5>(w); wu[6] = group16_bcast<6>(w); wu[7] = group16_bcast<7>(w);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a0 = fmaf(wu[u], v0[u], a0);
      if (WIDE) a1 = fmaf(wu[u], v1[u], a1);
    }
  };
  // the self-loop entry of the group's own row, absorbed last (PyG appends it after the edges)
  auto absorb_self = [&]() {
    const float* __restrict__ kj = qkvs + (int64_t)row * ld + HC + h * C;
    const float ks0 = c0 ? kj[l] : 0.f, vs0 = c0 ? kj[HC + l] : 0.f;
    const float ks1 = c1 ? kj[l1] : 0.f, vs1 = c1 ? kj[HC + l1] : 0.f;
    const float s = dot(q0, q1, ks0, ks1);
    if (s > m) {
      const float rr = expf(m - s);
      denom *= rr; a0 *= rr; a1 *= rr;
      m = s;
    }
    const float p = expf(s - m) * (float)n_self;
    denom += p;
    const float w = weight(p, a.E + row, h);
    a0 = fmaf(w, vs0, a0);
    if (WIDE) a1 = fmaf(w, vs1, a1);
  };

  if (cnt <= kAttnShort) {
    int jj[kAttnShort];
    float k0[kAttnShort], k1[WIDE ? kAttnShort : 1], v0[kAttnShort], v1[WIDE ? kAttnShort : 1], sc[kAttnShort];
#pragma unroll
    for (int e = 0; e < kAttnShort; ++e) {
      int j = row;                                       // entry `deg` is the self-loop; entries past cnt are not used
      if (e < deg) j = idx[beg + e];
      jj[e] = j;
    }
#pragma unroll
    for (int e = 0; e < kAttnShort; ++e) {
      if (e < cnt) {                                     // group-uniform
        const float* __restrict__ kj = qkvs + (int64_t)jj[e] * ld + HC + h * C;
        k0[e] = c0 ? kj[l] : 0.f;
        if (WIDE) k1[e] = c1 ? kj[l1] : 0.f;
        v0[e] = c0 ? kj[HC + l] : 0.f;
        if (WIDE) v1[e] = c1 ? kj[HC + l1] : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < kAttnShort; ++e) sc[e] = e < cnt ? dot(q0, q1, k0[e], WIDE ? k1[e] : 0.f) : -INFINITY;
#pragma unroll
    for (int e = 0; e < kAttnShort; ++e) m = fmaxf(m, sc[e]);
#pragma unroll
    for (int e = 0; e < kAttnShort; ++e) {
      if (e < cnt) {
        const bool self = e == deg;
        const float p = expf(sc[e] - m) * (self ? (float)n_self : 1.f);
        denom += p;
        const float w = weight(p, self ? a.E + row : (int64_t)beg + e, h);
        a0 = fmaf(w, v0[e], a0);
        if (WIDE) a1 = fmaf(w, v1[e], a1);
      }
    }
  } else {
    for (int e = beg; e < end; e += 8) absorb8(e, min(8, end - e));
    if (n_self > 0) absorb_self();
  }

  denom += 1e-16f;
  const float inv = 1.0f / denom;
  a0 *= inv;
  a1 *= inv;
  const float* __restrict__ skip = qkvs + (int64_t)row * ld + 3 * HC + h * C;
  float* __restrict__ o = a.out + (int64_t)row * a.ldo + h * C;
  if (c0) o[l] = a0 + skip[l];
  if (c1) o[l1] = a1 + skip[l1];
  if (TRAIN) {
    float* __restrict__ ao = a.attn_out + (int64_t)row * a.lda + h * C;
    if (c0) ao[l] = a0;
    if (c1) ao[l1] = a1;
    if (l == 0) {
      a.stat_m[(int64_t)row * H + h] = m;
      a.stat_den[(int64_t)row * H + h] = denom;
    }
  }
}

}  // namespace mlqem
