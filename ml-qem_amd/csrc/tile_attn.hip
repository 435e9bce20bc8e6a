// TransformerConv's edge softmax on TILES (docs/tutorials/gnn.py:80-91, the second TransformerConv of every reference GNN: it runs
// on the graph ASAPooling coarsened; PyG semantics in SURVEY appendix B.1).  Same formulas, arguments and outputs as the per-edge
// kernels of attn_q4.hpp / family_b_bwd.hip (forward with statistics, destination-side and recomputing source-side backward); what
// changes is where a gathered row comes from: a workgroup owns a tile of rows (tile_common.hpp), copies the key | value segments of
// the tile's source union into LDS once -- coalesced 16-byte loads, one 128-byte piece per head and part -- and every entry of every
// row of the tile is then two `ds_read_b128` at a 16-bit slot instead of two 64-byte gathers through L1.
//
// Layout of the work inside a workgroup (256 threads):
//   * a (row, head) belongs to a QUAD of lanes, lane lq holds channels 4 lq .. 4 lq + 3 (C <= 16 at a head pitch of 16: the training
//     layout of q | k | v | skip, functional._TransformerConv); entries are walked four at a time, lane u of the quad owns entry u's
//     scalar work (slot, score, exp, dropout draw), as in attn_q4.hpp;
//   * rows of at least kTileLongDeg entries ("long": the plan lists them first) are walked by a whole WAVE: the 16 quads are dealt
//     to the heads, the quads of a head take the row's four-entry chunks in turn, and their partial (max, denominator, sums) meet
//     through 2 KB of LDS in a fixed order; the others ("short") take one quad per (row, head).  Rows of 2 and of 198 entries are
//     both common in these graphs (median 2, mean 58): a wave of the per-edge kernels ran as long as its longest row.
//   * an entry whose source did not fit the tile's union (slot kTileNoSlot) is read from global memory in a wave-uniform side branch.
#include "attn_q4.hpp"
#include "tile_common.hpp"

namespace mlqem {

constexpr int kQuadScratch = 8;             // floats a lane files for its quad's partial result (two 16-byte records)

struct TileAttnLds {
  float* rows;      // [cap][pitch]
  float* own;       // [tile_rows][own_pitch]: operands of the tile's own rows
  float* scratch;   // [4 waves][64 lanes][kQuadScratch]
  TileLds c;        // the tile's row records, union ids and slots (tile_common.hpp)
};

__host__ __device__ inline size_t tile_attn_lds_bytes(int cap, int pitch, int tile_rows, int own_pitch, bool scratch) {
  return (size_t)cap * pitch * 4 + (size_t)tile_rows * own_pitch * 4 + (scratch ? (size_t)kBlock * kQuadScratch * 4 : 0) +
         tile_lds_common_bytes(cap, tile_rows);
}

__device__ __forceinline__ TileAttnLds tile_attn_carve(char* smem, int cap, int pitch, int tile_rows, int own_pitch, bool scratch) {
  TileAttnLds l;
  l.rows = reinterpret_cast<float*>(smem);
  l.own = l.rows + (size_t)cap * pitch;                           // every pitch is a multiple of 4 floats: 16-byte aligned parts
  l.scratch = l.own + (size_t)tile_rows * own_pitch;
  l.c = tile_lds_carve(reinterpret_cast<char*>(l.scratch + (scratch ? kBlock * kQuadScratch : 0)), cap, tile_rows);
  return l;
}

// `pieces` 16-byte pieces per slot copied from row uid[slot] of `src` (row pitch ld, first float `off`).  A slot belongs to a
// power-of-two group of lanes (no division per piece) and a thread has EIGHT loads in flight before its first LDS store: the staging
// is one round trip for unions of up to 8 * 256 / group slots (128 at 16 pieces per slot).
__device__ __forceinline__ void tile_stage_rows(const int* __restrict__ uid, int ucnt, const float* __restrict__ src, int64_t ld,
                                                int off, int pieces, float* __restrict__ rows, int pitch) {
  const int shift = pieces <= 8 ? 3 : pieces <= 16 ? 4 : pieces <= 32 ? 5 : 6;
  const int pc = threadIdx.x & ((1 << shift) - 1), s_in = threadIdx.x >> shift, per = kBlock >> shift;
  if (pc >= pieces) return;
  for (int s0 = 0; s0 < ucnt; s0 += 8 * per) {
    f4a v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int sl = s0 + u * per + s_in;
      if (sl < ucnt) v[u] = *reinterpret_cast<const f4a*>(src + (int64_t)uid[sl] * ld + off + 4 * pc);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int sl = s0 + u * per + s_in;
      if (sl < ucnt) *reinterpret_cast<f4a*>(rows + sl * pitch + 4 * pc) = v[u];
    }
  }
}
// ---------------------------------------------------------------------------------------------------------------- forward
// LDS slot: [ key h0 .. h(H-1) | value h0 .. h(H-1) ] at 16 floats per head (+ 4 floats so that consecutive slots start on
// different banks)
template <bool TRAIN> __global__ __launch_bounds__(kBlock) void tile_attn_fwd_kernel(const AttnFwdArgs a, const TilePlan p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int H = a.H, C = a.C, HP = H * 16, pitch = 2 * HP + 4;
  const TileAttnLds L = tile_attn_carve(smem, p.cap, pitch, p.tile_rows, 2 * HP, false);       // own rows: query | skip
  const int t = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int4 ti = tile_prologue(p, t, L.c);
  const int cnt = ti.x, ucnt = ti.z;
  int nlong = ti.y;
  const float* __restrict__ qkvs = a.qkvs;
  const int64_t ld = a.ld;
  tile_stage(p, ti, L.c, qkvs, ld, HP, 2 * H * 4, L.rows, pitch, TileOwn{qkvs, ld, 0, H * 4, 0}, TileOwn{qkvs, ld, 3 * HP, H * 4, HP}, L.own, 2 * HP);
  __syncthreads();
  const int nlong_run = nlong;

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, quad = lane >> 2, lq = lane & 3;
  const int nv = min(4, max(0, C - 4 * lq));
  const float scale = 1.0f / sqrtf((float)C);
  const float keep = 1.f / (1.f - a.drop_p);
  const bool drop = TRAIN && a.drop_p > 0.f;
  const uint64_t seed = a.seed + ((TRAIN && a.seed_counter) ? *a.seed_counter * 0xD1B54A32D192ED03ull : 0ull);
  const int32_t* __restrict__ idx = a.idx;

  // the entries 4 c + lu, c = c0, c0 + cstep, ... of (row, head h) folded into a running (max, denominator, sums)
  auto walk = [&](const int4& ri, int h, int n_self, bool with_self, int c0, int cstep, const f4u& q, float& m, float& den, f4u& acc) {
    const int row = ri.x, beg = ri.y, deg = ri.z;
    const int hoff = h * 16 + 4 * lq;
    const int nch = (deg + 3) >> 2;
    for (int c = c0; c < nch; c += cstep) {
      const int x = 4 * c + lq;
      const int k = min(4, deg - 4 * c);
      const bool valid = lq < k;
      const uint32_t lc = valid ? tile_slot(p, L.c, ri, x) : 0u;
      const bool ovf = valid && lc == kTileNoSlot;
      const int slot = ovf ? 0 : (int)lc;
      const int su[4] = {quad_bcast<0>(slot), quad_bcast<1>(slot), quad_bcast<2>(slot), quad_bcast<3>(slot)};
      f4u kk[4], vv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* kp = L.rows + su[u] * pitch + hoff;
        kk[u] = *reinterpret_cast<const f4a*>(kp);
        vv[u] = *reinterpret_cast<const f4a*>(kp + HP);
      }
      int j = 0;
      if (ovf) j = idx[beg + x];
      else if (drop && a.pair_key) j = L.c.uid[slot];
      if (__ballot(ovf) != 0ull) {                        // wave-uniform, rare: sources outside the staged union
        const int o = ovf ? 1 : 0;
        const int ou[4] = {quad_bcast<0>(o), quad_bcast<1>(o), quad_bcast<2>(o), quad_bcast<3>(o)};
        const int ju[4] = {quad_bcast<0>(j), quad_bcast<1>(j), quad_bcast<2>(j), quad_bcast<3>(j)};
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (ou[u]) {
            const float* __restrict__ kj = qkvs + (int64_t)ju[u] * ld + HP + hoff;
            kk[u] = *reinterpret_cast<const f4u*>(kj);
            vv[u] = *reinterpret_cast<const f4u*>(kj + HP);
          }
      }
      float mys = -INFINITY;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float s = quad_sum(dot4(q, kk[u])) * scale;
        if (lq == u) mys = s;
      }
      if (!valid) mys = -INFINITY;
      const float cm = quad_max(mys);
      if (cm > m) {
        const float r = expf(m - cm);
        den *= r; acc *= r;
        m = cm;
      }
      const float pe = valid ? expf(mys - m) : 0.f;
      den += quad_sum(pe);
      float w = pe;
      if (drop) w = attn_dropped(seed, a.pair_key != 0, (int64_t)beg + x, H, h, row, j, a.drop_p) ? 0.f : pe * keep;
      const float wu[4] = {quad_bcast<0>(w), quad_bcast<1>(w), quad_bcast<2>(w), quad_bcast<3>(w)};
#pragma unroll
      for (int u = 0; u < 4; ++u) acc += wu[u] * vv[u];
    }
    if (with_self && n_self > 0) {                         // the self-loop entry last (PyG appends it after the edges)
      const float* __restrict__ kj = qkvs + (int64_t)row * ld + HP + hoff;
      const f4u ks = *reinterpret_cast<const f4u*>(kj), vs = *reinterpret_cast<const f4u*>(kj + HP);
      const float s = quad_sum(dot4(q, ks)) * scale;
      if (s > m) {
        const float r = expf(m - s);
        den *= r; acc *= r;
        m = s;
      }
      const float pe = expf(s - m) * (float)n_self;
      den += pe;
      float w = pe;
      if (drop) w = attn_dropped(seed, a.pair_key != 0, a.E + row, H, h, row, row, a.drop_p) ? 0.f : pe * keep;
      acc += w * vs;
    }
  };
  auto finish = [&](int row, int h, float m, float den, f4u acc, const f4u& skip) {
    const int off = h * C + 4 * lq;
    den += 1e-16f;
    acc *= 1.0f / den;
    store_channels(a.out + (int64_t)row * a.ldo + off, acc + skip, nv);
    if (TRAIN) {
      store_channels(a.attn_out + (int64_t)row * a.lda + off, acc, nv);      // every row (the tiled backward reads every row's)
      if (lq == 0) {
        a.stat_m[(int64_t)row * H + h] = m;
        a.stat_den[(int64_t)row * H + h] = den;
      }
    }
  };

  // long rows: a wave per row, ONE LANE PER ENTRY, head after head.  A score is sixteen FMAs against the row's query (wave-uniform);
  // the softmax is two passes over up to 256 entries at a time whose scores stay in registers (the wave's maximum between them; a
  // longer row rescales what it has summed once per 256 entries); a lane sums the value rows of its own entries and the sixteen
  // per-lane sums meet in one reduce-scatter per (row, head).  Against a quad per (entry, head): no cross-lane sum per score, the
  // scalar work of an entry once per lane instead of once per quad -- a third of the vector instructions.
  const uint32_t thr16 = attn_drop_threshold(a.drop_p);
  const int mych = wave_channel16(lane);
  for (int r = wave; r < nlong_run; r += 4) {
    const int4 ri4 = L.c.rinfo[r];
    const int row = __builtin_amdgcn_readfirstlane(ri4.x), beg = __builtin_amdgcn_readfirstlane(ri4.y);
    const int deg = __builtin_amdgcn_readfirstlane(ri4.z), loff = __builtin_amdgcn_readfirstlane(ri4.w);
    const bool loc_in_lds = loff + deg <= kTileLocEntries;
    const int n_self = a.loops ? a.loops[row] : 0;
    const float* __restrict__ rrow = qkvs + (int64_t)row * ld;
    uint32_t hs[4] = {0u, 0u, 0u, 0u};                       // the draws of the first 256 entries, kept from an even head to the next
    for (int h = 0; h < H; ++h) {
      float q[16];
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        const f4u t = *reinterpret_cast<const f4a*>(L.own + r * 2 * HP + h * 16 + 4 * c4);
        q[4 * c4] = t.x * scale; q[4 * c4 + 1] = t.y * scale; q[4 * c4 + 2] = t.z * scale; q[4 * c4 + 3] = t.w * scale;
      }
      const float skipv = L.own[r * 2 * HP + HP + h * 16 + mych];
      float M = -INFINITY, den = 0.f, acc[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) acc[c] = 0.f;
      for (int e0 = 0; e0 < deg; e0 += 256) {
        float sc[4];
        int sl[4];                                            // the entry's slot, or -(source id + 1) when it has none
        float lmax = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          sc[i] = -INFINITY; sl[i] = 0;
          if (e0 + 64 * i < deg) {                            // wave-uniform
            const int e = e0 + 64 * i + lane;
            const bool valid = e < deg;
            const uint32_t lc = valid ? (loc_in_lds ? (uint32_t)L.c.loc[loff + e] : (uint32_t)p.loc[beg + e]) : 0u;
            const bool ovf = lc == kTileNoSlot;
            const int slot = ovf ? 0 : (int)lc;
            const float* kp = L.rows + slot * pitch + h * 16;
            f4u k0 = *reinterpret_cast<const f4a*>(kp), k1 = *reinterpret_cast<const f4a*>(kp + 4);
            f4u k2 = *reinterpret_cast<const f4a*>(kp + 8), k3 = *reinterpret_cast<const f4a*>(kp + 12);
            sl[i] = slot;
            if (__ballot(ovf) != 0ull) {                      // rare: sources outside the staged union
              if (ovf) {
                const int j = idx[beg + e];
                const float* __restrict__ kj = qkvs + (int64_t)j * ld + HP + h * 16;
                k0 = *reinterpret_cast<const f4u*>(kj); k1 = *reinterpret_cast<const f4u*>(kj + 4);
                k2 = *reinterpret_cast<const f4u*>(kj + 8); k3 = *reinterpret_cast<const f4u*>(kj + 12);
                sl[i] = -(j + 1);
              }
            }
            float sdot = q[0] * k0.x;
            sdot = fmaf(q[1], k0.y, sdot); sdot = fmaf(q[2], k0.z, sdot); sdot = fmaf(q[3], k0.w, sdot);
            sdot = fmaf(q[4], k1.x, sdot); sdot = fmaf(q[5], k1.y, sdot); sdot = fmaf(q[6], k1.z, sdot); sdot = fmaf(q[7], k1.w, sdot);
            sdot = fmaf(q[8], k2.x, sdot); sdot = fmaf(q[9], k2.y, sdot); sdot = fmaf(q[10], k2.z, sdot); sdot = fmaf(q[11], k2.w, sdot);
            sdot = fmaf(q[12], k3.x, sdot); sdot = fmaf(q[13], k3.y, sdot); sdot = fmaf(q[14], k3.z, sdot); sdot = fmaf(q[15], k3.w, sdot);
            if (valid) sc[i] = sdot;
            lmax = fmaxf(lmax, sc[i]);
          }
        }
        const float cm = wave_max_all(lmax);
        if (cm > M) {                                         // wave-uniform: once per 256 entries at most
          const float rs = expf(M - cm);
          den *= rs;
#pragma unroll
          for (int c = 0; c < 16; ++c) acc[c] *= rs;
          M = cm;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (e0 + 64 * i < deg) {
            const int e = e0 + 64 * i + lane;
            const float pe = expf(sc[i] - M);                 // an entry past the row's end: exp(-inf) = 0
            den += pe;
            float w = pe;
            if (drop) {
              uint32_t hh = hs[i];
              if ((h & 1) == 0 || e0 > 0) {
                const int j = sl[i] < 0 ? -sl[i] - 1 : L.c.uid[sl[i]];
                hh = attn_pair_hash(seed, row, j, h >> 1);
                if (e0 == 0) hs[i] = hh;
              }
              w = attn_pair_dropped(hh, h, thr16) ? 0.f : pe * keep;
            }
            f4u v0, v1, v2, v3;
            if (sl[i] >= 0) {
              const float* vp = L.rows + sl[i] * pitch + HP + h * 16;
              v0 = *reinterpret_cast<const f4a*>(vp); v1 = *reinterpret_cast<const f4a*>(vp + 4);
              v2 = *reinterpret_cast<const f4a*>(vp + 8); v3 = *reinterpret_cast<const f4a*>(vp + 12);
            } else {
              const float* __restrict__ vj = qkvs + (int64_t)(-sl[i] - 1) * ld + 2 * HP + h * 16;
              v0 = *reinterpret_cast<const f4u*>(vj); v1 = *reinterpret_cast<const f4u*>(vj + 4);
              v2 = *reinterpret_cast<const f4u*>(vj + 8); v3 = *reinterpret_cast<const f4u*>(vj + 12);
            }
            acc[0] = fmaf(w, v0.x, acc[0]); acc[1] = fmaf(w, v0.y, acc[1]); acc[2] = fmaf(w, v0.z, acc[2]); acc[3] = fmaf(w, v0.w, acc[3]);
            acc[4] = fmaf(w, v1.x, acc[4]); acc[5] = fmaf(w, v1.y, acc[5]); acc[6] = fmaf(w, v1.z, acc[6]); acc[7] = fmaf(w, v1.w, acc[7]);
            acc[8] = fmaf(w, v2.x, acc[8]); acc[9] = fmaf(w, v2.y, acc[9]); acc[10] = fmaf(w, v2.z, acc[10]); acc[11] = fmaf(w, v2.w, acc[11]);
            acc[12] = fmaf(w, v3.x, acc[12]); acc[13] = fmaf(w, v3.y, acc[13]); acc[14] = fmaf(w, v3.z, acc[14]); acc[15] = fmaf(w, v3.w, acc[15]);
          }
        }
      }
      float dsum = wave_sum_all(den);
      float mine = wave_reduce16(acc, lane);                  // channel mych of sum_e w_e v_e
      if (n_self > 0) {                                       // the self-loop entry last (PyG appends it after the edges)
        const float ss = group16_sum(rrow[h * 16 + mych] * scale * rrow[HP + h * 16 + mych]);     // a pad channel: q = 0
        if (ss > M) {
          const float rs = expf(M - ss);
          dsum *= rs; mine *= rs;
          M = ss;
        }
        const float pe = expf(ss - M) * (float)n_self;
        dsum += pe;
        float w = pe;
        if (drop) w = attn_pair_dropped(attn_pair_hash(seed, row, row, h >> 1), h, thr16) ? 0.f : pe * keep;
        mine = fmaf(w, rrow[2 * HP + h * 16 + mych], mine);
      }
      dsum += 1e-16f;
      mine *= 1.0f / dsum;
      if (lane < 16 && mych < C) {
        a.out[(int64_t)row * a.ldo + h * C + mych] = mine + skipv;
        if (TRAIN) a.attn_out[(int64_t)row * a.lda + h * C + mych] = mine;
      }
      if (TRAIN && lane == 0) {
        a.stat_m[(int64_t)row * H + h] = M;
        a.stat_den[(int64_t)row * H + h] = dsum;
      }
    }
  }
  // short rows: a quad per (row, head)
  const int npairs = (cnt - nlong) * H;
  for (int p0 = wave * 16; p0 < npairs; p0 += 64) {
    const int pr = p0 + quad;
    if (pr < npairs) {
      const int rix = pr / H, h = pr - rix * H;
      const int4 ri = L.c.rinfo[nlong + rix];
      const int row = ri.x;
      const int n_self = a.loops ? a.loops[row] : 0;
      const f4u q = *reinterpret_cast<const f4a*>(L.own + (nlong + rix) * 2 * HP + h * 16 + 4 * lq);
      const f4u skip = *reinterpret_cast<const f4a*>(L.own + (nlong + rix) * 2 * HP + HP + h * 16 + 4 * lq);
      float m = -INFINITY, den = 0.f;
      f4u acc = {0.f, 0.f, 0.f, 0.f};
      walk(ri, h, n_self, true, 0, 1, q, m, den, acc);
      finish(row, h, m, den, acc, skip);
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward, destination side
// g_q, g_skip and the record {m, 1 / den, delta = g . attn_out} per (row, head) the source side recomputes its weights from
// (AttnBwdArgs.edge_al as float4[N H]; nothing is written per edge).  Same LDS slot as the forward.
__global__ __launch_bounds__(kBlock) void tile_attn_bwd_dst_kernel(const AttnBwdArgs a, const TilePlan p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int H = a.H, C = a.C, HP = H * 16, pitch = 2 * HP + 4;
  const TileAttnLds L = tile_attn_carve(smem, p.cap, pitch, p.tile_rows, 0, true);
  const int t = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int4 ti = tile_prologue(p, t, L.c);
  const int cnt = ti.x, nlong = ti.y, ucnt = ti.z;
  const float* __restrict__ qkvs = a.qkvs;
  const int64_t ld = a.ld;
  tile_stage_rows(L.c.uid, ucnt, qkvs, ld, HP, 2 * H * 4, L.rows, pitch);
  tile_stage_loc(p, ti, L.c);
  __syncthreads();

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, quad = lane >> 2, lq = lane & 3;
  const int nv = min(4, max(0, C - 4 * lq));
  const float scale = 1.0f / sqrtf((float)C);
  const float keep = 1.f / (1.f - a.drop_p);
  const bool drop = a.drop_p > 0.f;
  const uint64_t seed = a.seed + (a.seed_counter ? *a.seed_counter * 0xD1B54A32D192ED03ull : 0ull);
  const int32_t* __restrict__ idx = a.idx;
  float* __restrict__ my_scratch = L.scratch + (wave * kWave + lane) * kQuadScratch;

  struct RowHead { f4u q, gi; float m, inv_den, delta; };
  auto load_row = [&](int row, int h) {
    RowHead r;
    const int off = h * C + 4 * lq;
    r.q = *reinterpret_cast<const f4u*>(qkvs + (int64_t)row * ld + h * 16 + 4 * lq);
    r.gi = load_channels(a.g + (int64_t)row * a.ldg + off, nv, off + 4 <= a.ldg);
    r.m = a.stat_m[(int64_t)row * H + h];
    r.inv_den = 1.0f / a.stat_den[(int64_t)row * H + h];
    r.delta = quad_sum(dot4(r.gi, load_channels(a.attn_out + (int64_t)row * a.lda + off, nv, off + 4 <= a.lda)));
    return r;
  };
  auto walk = [&](const int4& ri, int h, int n_self, bool with_self, int c0, int cstep, const RowHead& rh, f4u& gq) {
    const int row = ri.x, beg = ri.y, deg = ri.z;
    const int hoff = h * 16 + 4 * lq;
    const int nch = (deg + 3) >> 2;
    for (int c = c0; c < nch; c += cstep) {
      const int x = 4 * c + lq;
      const int k = min(4, deg - 4 * c);
      const bool valid = lq < k;
      const uint32_t lc = valid ? tile_slot(p, L.c, ri, x) : 0u;
      const bool ovf = valid && lc == kTileNoSlot;
      const int slot = ovf ? 0 : (int)lc;
      const int su[4] = {quad_bcast<0>(slot), quad_bcast<1>(slot), quad_bcast<2>(slot), quad_bcast<3>(slot)};
      f4u kk[4], vv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* kp = L.rows + su[u] * pitch + hoff;
        kk[u] = *reinterpret_cast<const f4a*>(kp);
        vv[u] = *reinterpret_cast<const f4a*>(kp + HP);
      }
      int j = 0;
      if (ovf) j = idx[beg + x];
      else if (drop && a.pair_key) j = L.c.uid[slot];
      if (__ballot(ovf) != 0ull) {
        const int o = ovf ? 1 : 0;
        const int ou[4] = {quad_bcast<0>(o), quad_bcast<1>(o), quad_bcast<2>(o), quad_bcast<3>(o)};
        const int ju[4] = {quad_bcast<0>(j), quad_bcast<1>(j), quad_bcast<2>(j), quad_bcast<3>(j)};
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (ou[u]) {
            const float* __restrict__ kj = qkvs + (int64_t)ju[u] * ld + HP + hoff;
            kk[u] = *reinterpret_cast<const f4u*>(kj);
            vv[u] = *reinterpret_cast<const f4u*>(kj + HP);
          }
      }
      float mys = 0.f, mygv = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float sd = quad_sum(dot4(rh.q, kk[u])), gd = quad_sum(dot4(rh.gi, vv[u]));
        if (lq == u) { mys = sd; mygv = gd; }
      }
      const float alpha = expf(mys * scale - rh.m) * rh.inv_den;
      float dmask = 1.f;
      if (drop) dmask = attn_dropped(seed, a.pair_key != 0, (int64_t)beg + x, H, h, row, j, a.drop_p) ? 0.f : keep;
      float gs = alpha * (mygv * dmask - rh.delta) * scale;
      if (!valid) gs = 0.f;
      const float gu[4] = {quad_bcast<0>(gs), quad_bcast<1>(gs), quad_bcast<2>(gs), quad_bcast<3>(gs)};
#pragma unroll
      for (int u = 0; u < 4; ++u) gq += gu[u] * kk[u];
    }
    if (with_self && n_self > 0) {
      const float* __restrict__ kj = qkvs + (int64_t)row * ld + HP + hoff;
      const f4u ks = *reinterpret_cast<const f4u*>(kj), vs = *reinterpret_cast<const f4u*>(kj + HP);
      const float sd = quad_sum(dot4(rh.q, ks)), gd = quad_sum(dot4(rh.gi, vs));
      const float alpha = expf(sd * scale - rh.m) * rh.inv_den * (float)n_self;
      float dmask = 1.f;
      if (drop) dmask = attn_dropped(seed, a.pair_key != 0, a.E + row, H, h, row, row, a.drop_p) ? 0.f : keep;
      gq += (alpha * (gd * dmask - rh.delta) * scale) * ks;
    }
  };
  auto finish = [&](int row, int h, const RowHead& rh, const f4u& gq) {
    const int offp = h * 16 + 4 * lq;
    float* __restrict__ go = a.gqkvs + (int64_t)row * a.ldq;
    *reinterpret_cast<f4u*>(go + offp) = gq;                           // pads: sums of gs * (zero key pad) = 0
    *reinterpret_cast<f4u*>(go + 3 * HP + offp) = rh.gi;               // pads: masked to zero by load_channels
    if (lq == 0) reinterpret_cast<float4*>(a.edge_al)[(int64_t)row * H + h] = make_float4(rh.m, rh.inv_den, rh.delta, 0.f);
  };

  const int QPH = 16 / H;
  const int lh = min(quad / QPH, H - 1), sub = quad - lh * QPH;
  const bool lactive = quad < QPH * H;
  for (int r = wave; r < nlong; r += 4) {
    const int4 ri = L.c.rinfo[r];
    const int row = ri.x;
    const int n_self = a.loops ? a.loops[row] : 0;
    f4u gq = {0.f, 0.f, 0.f, 0.f};
    RowHead rh = {};
    if (lactive) {
      rh = load_row(row, lh);
      walk(ri, lh, n_self, sub == 0, sub, QPH, rh, gq);
    }
    *reinterpret_cast<f4a*>(my_scratch) = gq;
    wave_sync();
    if (lactive && sub == 0) {
      const float* __restrict__ part = L.scratch + (wave * kWave + lh * QPH * 4 + lq) * kQuadScratch;
      f4u sum = {0.f, 0.f, 0.f, 0.f};
      for (int s = 0; s < QPH; ++s) sum += *reinterpret_cast<const f4a*>(part + s * 4 * kQuadScratch);
      finish(row, lh, rh, sum);
    }
    wave_sync();
  }
  const int npairs = (cnt - nlong) * H;
  for (int p0 = wave * 16; p0 < npairs; p0 += 64) {
    const int pr = p0 + quad;
    if (pr < npairs) {
      const int rix = pr / H, h = pr - rix * H;
      const int4 ri = L.c.rinfo[nlong + rix];
      const int row = ri.x;
      const int n_self = a.loops ? a.loops[row] : 0;
      const RowHead rh = load_row(row, h);
      f4u gq = {0.f, 0.f, 0.f, 0.f};
      walk(ri, h, n_self, true, 0, 1, rh, gq);
      finish(row, h, rh, gq);
    }
  }
}

// ------------------------------------------------------------------------------------------------------ backward, source side
// g_k[j] = sum_i gs_ij q_i, g_v[j] = sum_i alpha_ij mask_ij g_i over the out-entries j -> i (transformer_attn_bwd_src_rc_q4_kernel's
// formulas: every weight recomputed from the destination's record), on a plan of the OUT-CSR: the union holds DESTINATION rows.
// LDS slot, per head: [ query (16) | gradient (16, compact [H C] rows repacked to the pitch, pad zero) | m, 1 / den, delta, - ].
constexpr int kSrcHead = 36;
__global__ __launch_bounds__(kBlock) void tile_attn_bwd_src_kernel(const AttnBwdArgs a, const TilePlan p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int H = a.H, C = a.C, HP = H * 16, pitch = H * kSrcHead;
  const TileAttnLds L = tile_attn_carve(smem, p.cap, pitch, p.tile_rows, 0, true);
  const int t = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int4 ti = tile_prologue(p, t, L.c);
  const int cnt = ti.x, nlong = ti.y, ucnt = ti.z;
  const float* __restrict__ qkvs = a.qkvs;
  const float* __restrict__ g = a.g;
  const float4* __restrict__ rec = reinterpret_cast<const float4*>(a.edge_al);
  const int64_t ld = a.ld, ldg = a.ldg;
  {
    const int* __restrict__ un = L.c.uid;
    const int pieces = 9 * H, total = ucnt * pieces;
    for (int i = threadIdx.x; i < total; i += kBlock) {
      const int s = i / pieces, pc = i - s * pieces;
      const int h = pc / 9, k = pc - h * 9;
      const int64_t id = un[s];
      f4u v;
      if (k < 4) v = *reinterpret_cast<const f4u*>(qkvs + id * ld + h * 16 + 4 * k);
      else if (k < 8) {
        const int c0 = 4 * (k - 4), off = h * C + c0;
        v = load_channels(g + id * ldg + off, min(4, max(0, C - c0)), off + 4 <= ldg);
      } else {
        const float4 r = rec[id * H + h];
        v = f4u{r.x, r.y, r.z, 0.f};
      }
      *reinterpret_cast<f4a*>(L.rows + s * pitch + h * kSrcHead + 4 * k) = v;
    }
    tile_stage_loc(p, ti, L.c);
  }
  __syncthreads();

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, quad = lane >> 2, lq = lane & 3;
  const int nv = min(4, max(0, C - 4 * lq));
  const float scale = 1.0f / sqrtf((float)C);
  const float keep = 1.f / (1.f - a.drop_p);
  const bool drop = a.drop_p > 0.f;
  const uint64_t seed = a.seed + (a.seed_counter ? *a.seed_counter * 0xD1B54A32D192ED03ull : 0ull);
  const int32_t* __restrict__ odst = a.odst;
  float* __restrict__ my_scratch = L.scratch + (wave * kWave + lane) * kQuadScratch;

  auto walk = [&](const int4& ri, int h, int n_self, bool with_self, int c0, int cstep, const f4u& kown, const f4u& vown,
                  f4u& gk, f4u& gv) {
    const int row = ri.x, beg = ri.y, deg = ri.z;
    const int hoff = h * kSrcHead + 4 * lq;
    const int off = h * C + 4 * lq;
    const int nch = (deg + 3) >> 2;
    for (int c = c0; c < nch; c += cstep) {
      const int x = 4 * c + lq;
      const int k = min(4, deg - 4 * c);
      const bool valid = lq < k;
      const uint32_t lc = valid ? tile_slot(p, L.c, ri, x) : 0u;
      const bool ovf = valid && lc == kTileNoSlot;
      const int slot = ovf ? 0 : (int)lc;
      const int su[4] = {quad_bcast<0>(slot), quad_bcast<1>(slot), quad_bcast<2>(slot), quad_bcast<3>(slot)};
      f4u qa[4], ga[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* qp = L.rows + su[u] * pitch + hoff;
        qa[u] = *reinterpret_cast<const f4a*>(qp);
        ga[u] = *reinterpret_cast<const f4a*>(qp + 16);
      }
      f4a st = *reinterpret_cast<const f4a*>(L.rows + slot * pitch + h * kSrcHead + 32);     // the lane's own entry's record
      int i = 0;
      if (ovf) i = odst[beg + x];
      else if (drop) i = L.c.uid[slot];
      if (__ballot(ovf) != 0ull) {
        const int o = ovf ? 1 : 0;
        const int ou[4] = {quad_bcast<0>(o), quad_bcast<1>(o), quad_bcast<2>(o), quad_bcast<3>(o)};
        const int iu[4] = {quad_bcast<0>(i), quad_bcast<1>(i), quad_bcast<2>(i), quad_bcast<3>(i)};
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (ou[u]) {
            qa[u] = *reinterpret_cast<const f4u*>(qkvs + (int64_t)iu[u] * ld + h * 16 + 4 * lq);
            ga[u] = load_channels(g + (int64_t)iu[u] * ldg + off, nv, off + 4 <= ldg);
          }
        if (ovf) {
          const float4 r = rec[(int64_t)i * H + h];
          st = f4a{r.x, r.y, r.z, 0.f};
        }
      }
      float mys = 0.f, mygv = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float sd = quad_sum(dot4(qa[u], kown)), gd = quad_sum(dot4(ga[u], vown));
        if (lq == u) { mys = sd; mygv = gd; }
      }
      const float alpha = expf(mys * scale - st.x) * st.y;
      float dmask = 1.f;
      if (drop) dmask = attn_dropped(seed, true, 0, H, h, i, row, a.drop_p) ? 0.f : keep;
      float gs = alpha * (mygv * dmask - st.z) * scale, al = alpha * dmask;
      if (!valid) { gs = 0.f; al = 0.f; }
      const float gsu[4] = {quad_bcast<0>(gs), quad_bcast<1>(gs), quad_bcast<2>(gs), quad_bcast<3>(gs)};
      const float alu[4] = {quad_bcast<0>(al), quad_bcast<1>(al), quad_bcast<2>(al), quad_bcast<3>(al)};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        gk += gsu[u] * qa[u];
        gv += alu[u] * ga[u];
      }
    }
    if (with_self && n_self > 0) {
      const f4u qs = *reinterpret_cast<const f4u*>(qkvs + (int64_t)row * ld + h * 16 + 4 * lq);
      const f4u gsf = load_channels(g + (int64_t)row * ldg + off, nv, off + 4 <= ldg);
      const float4 r = rec[(int64_t)row * H + h];
      const float sd = quad_sum(dot4(qs, kown)), gd = quad_sum(dot4(gsf, vown));
      const float alpha = expf(sd * scale - r.x) * r.y * (float)n_self;
      float dmask = 1.f;
      if (drop) dmask = attn_dropped(seed, true, 0, H, h, row, row, a.drop_p) ? 0.f : keep;
      gk += (alpha * (gd * dmask - r.z) * scale) * qs;
      gv += (alpha * dmask) * gsf;
    }
  };
  auto finish = [&](int row, int h, const f4u& gk, const f4u& gv) {
    float* __restrict__ go = a.gqkvs + (int64_t)row * a.ldq + h * 16 + 4 * lq;
    *reinterpret_cast<f4u*>(go + HP) = gk;
    *reinterpret_cast<f4u*>(go + 2 * HP) = gv;
  };

  const int QPH = 16 / H;
  const int lh = min(quad / QPH, H - 1), sub = quad - lh * QPH;
  const bool lactive = quad < QPH * H;
  for (int r = wave; r < nlong; r += 4) {
    const int4 ri = L.c.rinfo[r];
    const int row = ri.x;
    const int n_self = a.loops ? a.loops[row] : 0;
    f4u gk = {0.f, 0.f, 0.f, 0.f}, gv = {0.f, 0.f, 0.f, 0.f};
    if (lactive) {
      const float* __restrict__ rj = qkvs + (int64_t)row * ld + lh * 16 + 4 * lq;
      const f4u kown = *reinterpret_cast<const f4u*>(rj + HP), vown = *reinterpret_cast<const f4u*>(rj + 2 * HP);
      walk(ri, lh, n_self, sub == 0, sub, QPH, kown, vown, gk, gv);
    }
    *reinterpret_cast<f4a*>(my_scratch) = gk;
    *reinterpret_cast<f4a*>(my_scratch + 4) = gv;
    wave_sync();
    if (lactive && sub == 0) {
      const float* __restrict__ part = L.scratch + (wave * kWave + lh * QPH * 4 + lq) * kQuadScratch;
      f4u sk = {0.f, 0.f, 0.f, 0.f}, sv = {0.f, 0.f, 0.f, 0.f};
      for (int s = 0; s < QPH; ++s) {
        sk += *reinterpret_cast<const f4a*>(part + s * 4 * kQuadScratch);
        sv += *reinterpret_cast<const f4a*>(part + s * 4 * kQuadScratch + 4);
      }
      finish(row, lh, sk, sv);
    }
    wave_sync();
  }
  const int npairs = (cnt - nlong) * H;
  for (int p0 = wave * 16; p0 < npairs; p0 += 64) {
    const int pr = p0 + quad;
    if (pr < npairs) {
      const int rix = pr / H, h = pr - rix * H;
      const int4 ri = L.c.rinfo[nlong + rix];
      const int row = ri.x;
      const int n_self = a.loops ? a.loops[row] : 0;
      const float* __restrict__ rj = qkvs + (int64_t)row * ld + h * 16 + 4 * lq;
      const f4u kown = *reinterpret_cast<const f4u*>(rj + HP), vown = *reinterpret_cast<const f4u*>(rj + 2 * HP);
      f4u gk = {0.f, 0.f, 0.f, 0.f}, gv = {0.f, 0.f, 0.f, 0.f};
      walk(ri, h, n_self, true, 0, 1, kown, vown, gk, gv);
      finish(row, h, gk, gv);
    }
  }
}

}  // namespace mlqem

using namespace mlqem;

static bool plan_ok(const int32_t* tinfo, const int32_t* rinfo, const int32_t* uni, const uint16_t* loc, int64_t nt, int cap, int tile_rows) {
  return nt >= 0 && nt <= INT32_MAX && cap > 0 && cap < (int)kTileNoSlot && tile_rows > 0 && tile_rows <= kTileMaxRows &&
         (nt == 0 || (tinfo && rinfo && uni && loc && aligned_to(tinfo, 16) && aligned_to(rinfo, 16)));
}

// Largest slot count `cap` of a plan whose attention kernels (forward / destination side: key | value; source side: query | gradient |
// record) fit `lds_bytes` of LDS per workgroup.
extern "C" int mlqem_tile_attention_cap(int H, int tile_rows, int lds_bytes) {
  if (H <= 0 || tile_rows <= 0) return 0;
  const int pitch = std::max(2 * H * 16 + 4, H * kSrcHead);
  const int64_t room = (int64_t)lds_bytes - (int64_t)kBlock * kQuadScratch * 4 - (int64_t)tile_lds_common_bytes(0, tile_rows) - 16;
  return (int)std::max<int64_t>(0, room / ((int64_t)pitch * 4 + 4));
}

extern "C" int mlqem_tile_attention_train_f32(const float* qkvs, int64_t ld, const int32_t* in_ptr, const int32_t* in_src,
                                              const int32_t* loops, int64_t N, int64_t E, int H, int C, float drop_p, uint64_t seed,
                                              const uint64_t* seed_counter, int pair_key, int head_pitch, const int32_t* tinfo,
                                              const int32_t* rinfo, const int32_t* uni, const uint16_t* loc, int64_t num_tiles, int cap, int tile_rows,
                                              float* out, int64_t ldo, float* attn_out, int64_t lda, float* stat_m, float* stat_den,
                                              mlqem_stream_t stream) {
  begin_launches();
  const bool train = attn_out != nullptr;
  if (N < 0 || E < 0 || H <= 0 || C <= 0 || ld < 4 * H * 16 || ldo < H * C || (train && lda < H * C) || drop_p < 0.f || drop_p >= 1.f)
    return MLQEM_ERR_BAD_ARG;
  if (C > 16 || C < 13 || H > 16 || (head_pitch > 0 ? head_pitch : C) != 16) return MLQEM_ERR_UNSUPPORTED;   // the training layout: head pitch 16
  if (!plan_ok(tinfo, rinfo, uni, loc, num_tiles, cap, tile_rows)) return MLQEM_ERR_BAD_ARG;
  if (N == 0 || num_tiles == 0) return MLQEM_OK;
  if (!qkvs || !in_ptr || !out || (train && (!stat_m || !stat_den)) || (E > 0 && !in_src)) return MLQEM_ERR_BAD_ARG;
  if (!train && drop_p > 0.f) return MLQEM_ERR_BAD_ARG;
  if (N > INT32_MAX || !aligned_to(qkvs, 16) || ld % 4 != 0) return MLQEM_ERR_UNSUPPORTED;
  const AttnFwdArgs a{qkvs, ld, in_ptr, in_src, loops, N, E, H, C, drop_p, seed, seed_counter, out, ldo, attn_out, lda, stat_m, stat_den,
                      pair_key ? 1 : 0, nullptr, 16};
  const TilePlan p{reinterpret_cast<const int4*>(tinfo), reinterpret_cast<const int4*>(rinfo), uni, loc, num_tiles, cap, tile_rows};
  const size_t lds = tile_attn_lds_bytes(cap, 2 * H * 16 + 4, tile_rows, 2 * H * 16, false);
  if (train) {
    if (!ensure_dynamic_lds(tile_attn_fwd_kernel<true>, lds)) return MLQEM_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(tile_attn_fwd_kernel<true>, dim3((unsigned)num_tiles), dim3(kBlock), lds, as_stream(stream), a, p);
  } else {
    if (!ensure_dynamic_lds(tile_attn_fwd_kernel<false>, lds)) return MLQEM_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(tile_attn_fwd_kernel<false>, dim3((unsigned)num_tiles), dim3(kBlock), lds, as_stream(stream), a, p);
  }
  return launch_status();
}

extern "C" int mlqem_tile_attention_bwd_f32(const float* qkvs, int64_t ld, const float* g, int64_t ldg, const float* attn_out, int64_t lda,
                                            const float* stat_m, const float* stat_den, const int32_t* in_ptr, const int32_t* in_src,
                                            const int32_t* out_ptr, const int32_t* out_dst, const int32_t* loops, int64_t N, int64_t E,
                                            int H, int C, float drop_p, uint64_t seed, const uint64_t* seed_counter, int pair_key,
                                            int head_pitch, const int32_t* in_tinfo, const int32_t* in_rinfo, const int32_t* in_uni,
                                            const uint16_t* in_loc, int64_t in_tiles, int in_cap, int in_tile_rows, const int32_t* out_tinfo,
                                            const int32_t* out_rinfo, const int32_t* out_uni, const uint16_t* out_loc, int64_t out_tiles,
                                            int out_cap, int out_tile_rows, float* gqkvs, int64_t ldq, float* rec, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || E < 0 || H <= 0 || C <= 0 || ld < 4 * H * 16 || ldq < 4 * H * 16 || ldg < H * C || lda < H * C || drop_p < 0.f || drop_p >= 1.f)
    return MLQEM_ERR_BAD_ARG;
  if (C > 16 || C < 13 || H > 16 || (head_pitch > 0 ? head_pitch : C) != 16) return MLQEM_ERR_UNSUPPORTED;
  if (!plan_ok(in_tinfo, in_rinfo, in_uni, in_loc, in_tiles, in_cap, in_tile_rows) ||
      !plan_ok(out_tinfo, out_rinfo, out_uni, out_loc, out_tiles, out_cap, out_tile_rows))
    return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!qkvs || !g || !attn_out || !stat_m || !stat_den || !in_ptr || !out_ptr || !gqkvs || !rec || !aligned_to(rec, 16)) return MLQEM_ERR_BAD_ARG;
  if (E > 0 && (!in_src || !out_dst)) return MLQEM_ERR_BAD_ARG;
  if (drop_p > 0.f && !pair_key) return MLQEM_ERR_BAD_ARG;        // a position-keyed draw cannot be found from the source side
  if (N > INT32_MAX || !aligned_to(qkvs, 16) || ld % 4 != 0) return MLQEM_ERR_UNSUPPORTED;
  const AttnBwdArgs a{qkvs, ld, g, ldg, attn_out, lda, stat_m, stat_den, in_ptr, in_src, out_ptr, out_dst, nullptr, loops,
                      N, E, H, C, drop_p, seed, seed_counter, gqkvs, ldq, rec, nullptr, 1, 16};
  const TilePlan pin{reinterpret_cast<const int4*>(in_tinfo), reinterpret_cast<const int4*>(in_rinfo), in_uni, in_loc, in_tiles, in_cap, in_tile_rows};
  const TilePlan pout{reinterpret_cast<const int4*>(out_tinfo), reinterpret_cast<const int4*>(out_rinfo), out_uni, out_loc, out_tiles, out_cap,
                      out_tile_rows};
  const size_t lds_d = tile_attn_lds_bytes(in_cap, 2 * H * 16 + 4, in_tile_rows, 0, true), lds_s = tile_attn_lds_bytes(out_cap, H * kSrcHead, out_tile_rows, 0, true);
  if (!ensure_dynamic_lds(tile_attn_bwd_dst_kernel, lds_d) || !ensure_dynamic_lds(tile_attn_bwd_src_kernel, lds_s)) return MLQEM_ERR_UNSUPPORTED;
  if (in_tiles > 0) hipLaunchKernelGGL(tile_attn_bwd_dst_kernel, dim3((unsigned)in_tiles), dim3(kBlock), lds_d, as_stream(stream), a, pin);
  if (out_tiles > 0) hipLaunchKernelGGL(tile_attn_bwd_src_kernel, dim3((unsigned)out_tiles), dim3(kBlock), lds_s, as_stream(stream), a, pout);
  return launch_status();
}
