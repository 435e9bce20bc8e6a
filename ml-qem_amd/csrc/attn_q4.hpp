// TransformerConv's edge softmax with FOUR CHANNELS PER LANE (docs/tutorials/gnn.py:80-91; PyG semantics in SURVEY appendix
// B.1): forward, destination-side and source-side backward.  Same formulas, arguments and outputs as attn_fwd.hpp /
// family_b_bwd.hip's 16-lane forms (one lane per channel), another mapping of the work onto the wave:
//
//   * a (row, head) belongs to LPH = 4 (C <= 16) or 8 (C <= 32) lanes; lane lq holds channels 4 lq .. 4 lq + 3 as one float4.
//     A key / value / query / gradient row segment is ONE 16-byte load per lane -- the heads of a row sit 4 C bytes apart
//     (60 for the reference's 15 channels), so the loads are 4-byte aligned dwordx4 accesses, which global memory takes -- instead
//     of one 4-byte load per channel: a wave issues a quarter of the load instructions per edge, serves 16 (row, head) pairs
//     instead of 4, and a dot product is four FMAs and two (three) DPP adds instead of one multiply and four.  The 16-lane forms
//     were bound by instruction issue (attn_fwd.hpp), not by memory: 0.29 of the HBM peak on the circuit DAGs.
//   * the entries of a row -- its in-edges, then its self-loop entry -- are walked FOUR at a time: lane u of every quad owns entry
//     u of the chunk (its source id, score, exp, dropout draw, the [E]-sized outputs) and the ids / weights reach the other
//     lanes by quad_perm broadcasts; rows of the circuit DAGs (1-3 entries) are one chunk.
//   * the last lane of a head may hold fewer than four channels (15 = 3 x 4 + 3): its float4 then covers the first channel(s)
//     of the next head or part, which are multiplied by a zeroed query / never stored; where such a load would leave the row's
//     allocation (the skip part of the last head, compact [N, H C] rows) the lane loads its valid channels one by one.
#pragma once

#include <cstdlib>

#include "attn_fwd.hpp"

namespace mlqem {

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // a 16-byte access that needs 4-byte alignment only

template <int U> __device__ __forceinline__ float quad_bcast(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), U * 0x55, 0xF, 0xF, true));
}
template <int U> __device__ __forceinline__ int quad_bcast(int v) { return __builtin_amdgcn_update_dpp(0, v, U * 0x55, 0xF, 0xF, true); }
__device__ __forceinline__ float quad_sum(float v) {
  v += dpp_row<0xB1>(v);
  v += dpp_row<0x4E>(v);
  return v;
}
__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, dpp_row<0xB1>(v));
  v = fmaxf(v, dpp_row<0x4E>(v));
  return v;
}
// sum over the LPH lanes of a (row, head): every lane ends with the same value
template <int LPH> __device__ __forceinline__ float head_sum(float v) {
  v = quad_sum(v);
  if (LPH == 8) v += dpp_row<0x141>(v);   // row_half_mirror: lane i of an 8-lane half row meets lane 7 - i, which is in the other quad
  return v;
}
__device__ __forceinline__ float dot4(const f4u& a, const f4u& b) { return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x))); }

// the lane's channels of a row segment that starts at p: `nv` of the four are real; `fits`: the 16 bytes lie inside the row's allocation
__device__ __forceinline__ f4u load_channels(const float* __restrict__ p, int nv, bool fits) {
  f4u v = {0.f, 0.f, 0.f, 0.f};
  if (nv == 4 || (fits && nv > 0)) {
    // (masked by selects on a COPY: `if (nv < 4) v.w = 0` writes a register the load is still filling, and the wave waits for the load
    // right there, in front of every load that could have been in flight beside it)
    const f4u t = *reinterpret_cast<const f4u*>(p);
    v = f4u{t.x, nv > 1 ? t.y : 0.f, nv > 2 ? t.z : 0.f, nv > 3 ? t.w : 0.f};
  } else {
    if (nv > 0) v.x = p[0];
    if (nv > 1) v.y = p[1];
    if (nv > 2) v.z = p[2];
  }
  return v;
}
typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));
// MLQEM_ATTN_VSTORE=0 (compile time): one dword store per channel (A/B builds)
#ifndef MLQEM_ATTN_VSTORE
#define MLQEM_ATTN_VSTORE 1
#endif
// A lane's channels as ONE store: 16 bytes, or 12 from the last lane of a 15-channel head (the fourth float is the next head's
// first channel, another lane's) -- two store instructions per wave and matrix instead of four that each touch every line.
__device__ __forceinline__ void store_channels(float* __restrict__ p, const f4u& v, int nv) {
#if MLQEM_ATTN_VSTORE
  if (nv == 4) { *reinterpret_cast<f4u*>(p) = v; return; }
  if (nv == 3) { *reinterpret_cast<f3u*>(p) = f3u{v.x, v.y, v.z}; return; }
#endif
  if (nv > 0) p[0] = v.x;
  if (nv > 1) p[1] = v.y;
  if (nv > 2) p[2] = v.z;
  if (nv > 3) p[3] = v.w;
}

// ------------------------------------------------------------------------------------------------------ forward
// FAST (host: attn_q4_fast): every lane of a (row, head) group has a channel, every head sits in a slot of 4 LPH floats whose pads are
// zeros, every 16-byte access lies inside its row and there is an index array -- no load of the kernel then needs a LANE-VARYING
// condition.  A conditional load compiles to a branch; masking a loaded vector in place (`if (nv < 4) v.w = 0`) writes a register the
// load is still filling and waits for it on the spot; a load under `u < k` waits for k.  The general forms waited for memory six to
// eight times per (row, head) where three round trips are due (round 6).  Entries past the end of a chunk repeat its last entry (their
// weights are zero): FAST loads their rows again instead of branching around them.
__device__ __forceinline__ f4u mask4(const f4u& t, int nv) { return f4u{t.x, nv > 1 ? t.y : 0.f, nv > 2 ? t.z : 0.f, nv > 3 ? t.w : 0.f}; }
// a lane's four channels 4 l .. 4 l + 3 of a row in the PADDED layout (16-byte aligned rows of at least round_up(C, 4) floats, pads are
// scratch), UNCONDITIONALLY: a lane past the row's last slice reads that slice and masks everything
__device__ __forceinline__ f4u row4(const float* __restrict__ row, int l, int C) {
  const int last = (C - 1) >> 2;
  const int nv = min(4, max(0, C - 4 * l));
  const f4u t = *reinterpret_cast<const f4u*>(row + 4 * min(l, last));
  return f4u{nv > 0 ? t.x : 0.f, nv > 1 ? t.y : 0.f, nv > 2 ? t.z : 0.f, nv > 3 ? t.w : 0.f};
}
inline bool attn_q4_fast(int H, int C, int CP, int64_t ld_compact, const void* idx) {
  const int lph = C > 16 ? 8 : 4;
  return 4 * (lph - 1) < C && CP >= 4 * lph && (int64_t)(H - 1) * C + 4 * lph <= ld_compact && idx != nullptr;
}

template <bool TRAIN, int LPH, bool FAST = false> __device__ __forceinline__ void attn_forward_q4(const AttnFwdArgs& a) {
  const int64_t t = ((int64_t)row_block() * kBlock + threadIdx.x) / LPH;
  const int lq = threadIdx.x % LPH, lu = lq & 3;
  const int H = a.H, C = a.C, HC = H * C;
  if (t >= a.N * H) return;                             // a whole (row, head) leaves together
  const int row = (int)(t / H);
  if (a.skip && a.skip[row]) return;
  const int h = (int)(t - (int64_t)row * H);
  const int nv = min(4, max(0, C - 4 * lq));            // real channels of this lane
  const int off = h * C + 4 * lq;                       // the lane's first channel inside a compact [H C] row (out, attn_out)
  const int CP = a.CP > 0 ? a.CP : C, HP = H * CP;      // head pitch and part stride inside a qkvs row
  const int offp = h * CP + 4 * lq;                     // the lane's first channel inside a part of qkvs
  const bool roomy = CP - 4 * lq >= 4;                  // the lane's 16 bytes lie inside its head's slot (pads are zeros)
  const float scale = 1.0f / sqrtf((float)C);
  const float keep = 1.f / (1.f - a.drop_p);
  const uint64_t seed = a.seed + ((TRAIN && a.seed_counter) ? *a.seed_counter * 0xD1B54A32D192ED03ull : 0ull);
  const float* __restrict__ qkvs = a.qkvs;
  const int64_t ld = a.ld;
  const int32_t* __restrict__ idx = a.idx;

  const float* __restrict__ ri = qkvs + (int64_t)row * ld;
  f4u q, skip;
  if constexpr (FAST) {
    q = *reinterpret_cast<const f4u*>(ri + offp);
    skip = *reinterpret_cast<const f4u*>(ri + 3 * HP + offp);
  } else {
    q = load_channels(ri + offp, nv, true);                                   // runs over into the key part at most
    skip = load_channels(ri + 3 * HP + offp, nv, roomy || 3 * HP + offp + 4 <= 4 * HP);
  }
  // the row's in-edges: from the ELL side table when the row has at most two (one dependent round trip less), else from the CSR arrays
  int s0 = -1, s1 = -1;
  bool fast = false;
  if (a.ell) {
    const int2 e2 = reinterpret_cast<const int2*>(a.ell)[row];
    fast = e2.x == -1 || (e2.x & kEllMore) == 0;
    s0 = e2.x; s1 = e2.y;
  }
  const int beg = a.ptr[row];                           // fast rows: only the dropout key by position reads it
  const int deg = fast ? (s0 >= 0 ? 1 : 0) + (s1 >= 0 ? 1 : 0) : a.ptr[row + 1] - beg;
  const int n_self = a.loops ? a.loops[row] : 0;
  const int cnt = deg + (n_self > 0 ? 1 : 0);           // the self-loop entry comes last (PyG appends it after the edges)

  float m = -INFINITY, denom = 0.f;
  f4u acc = {0.f, 0.f, 0.f, 0.f};
  for (int x0 = 0; x0 < cnt; x0 += 4) {
    const int k = min(4, cnt - x0);
    const int x = x0 + min(lu, k - 1);                   // past the end: the last entry again (weight 0)
    const bool is_self = x >= deg;
    int j;
    if constexpr (FAST) {
      const int jx = idx[max(beg + min(x, deg - 1), 0)];   // (unconditional: entry 0 exists in every index array)
      j = is_self ? row : (fast ? (x == 0 ? s0 : s1) : jx);
    } else {
      j = is_self ? row : (fast ? (x == 0 ? s0 : s1) : idx[beg + x]);
    }
    int ju[4];
    ju[0] = quad_bcast<0>(j); ju[1] = quad_bcast<1>(j); ju[2] = quad_bcast<2>(j); ju[3] = quad_bcast<3>(j);
    f4u kk[4], vv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (FAST || (u < k && nv > 0)) {                   // uniform over the (row, head)'s lanes but for the channel-less ones
        const float* __restrict__ kj = qkvs + (int64_t)ju[u] * ld + HP + offp;
        kk[u] = *reinterpret_cast<const f4u*>(kj);       // inside the row: a key segment runs over into the value part at most,
        vv[u] = *reinterpret_cast<const f4u*>(kj + HP);  // a value segment into the skip part
      } else {
        kk[u] = f4u{0.f, 0.f, 0.f, 0.f};
        vv[u] = f4u{0.f, 0.f, 0.f, 0.f};
      }
    }
    float mys = -INFINITY;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float s = head_sum<LPH>(dot4(q, kk[u])) * scale;
      if (lu == u) mys = s;
    }
    if (lu >= k) mys = -INFINITY;
    const float cm = quad_max(mys);
    if (cm > m) {                                        // the running maximum grows: rescale what was summed under the old one
      const float r = expf(m - cm);                      // exp(-inf) = 0 the first time
      denom *= r; acc *= r;
      m = cm;
    }
    float p = lu < k ? expf(mys - m) : 0.f;
    if (is_self) p *= (float)n_self;
    denom += quad_sum(p);
    float w = p;
    if (TRAIN && a.drop_p > 0.f) {
      const int64_t pos = is_self ? a.E + row : (int64_t)beg + x;
      w = attn_dropped(seed, a.pair_key != 0, pos, H, h, row, j, a.drop_p) ? 0.f : p * keep;
    }
    const float wu[4] = {quad_bcast<0>(w), quad_bcast<1>(w), quad_bcast<2>(w), quad_bcast<3>(w)};
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += wu[u] * vv[u];
  }
  denom += 1e-16f;
  acc *= 1.0f / denom;
  store_channels(a.out + (int64_t)row * a.ldo + off, acc + skip, nv);
  if (TRAIN) {
    // the backward needs attn_out only as delta = g . attn_out, and a row of at most four entries (one chunk: every row of a
    // circuit DAG but the barriers') yields delta from what its destination-side pass holds anyway: such rows store nothing
    if (cnt > 4) store_channels(a.attn_out + (int64_t)row * a.lda + off, acc, nv);
    if (lq == 0) {
      a.stat_m[(int64_t)row * H + h] = m;
      a.stat_den[(int64_t)row * H + h] = denom;
    }
  }
}

// the four-channels-per-lane kernels serve every shape the one-channel-per-lane forms do (C <= 32); the latter stay as the reference the
// q4 forms were first checked against (attn_fwd.hpp, family_b_bwd.hip) and are no longer launched
inline bool attn_q4_enabled() { return true; }

}  // namespace mlqem
