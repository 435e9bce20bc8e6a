// ASAPooling's edge walks on TILES (docs/tutorials/gnn.py:85,92; PyG semantics in SURVEY appendix B.2, steps 2-4): the second
// pooling of every reference GNN runs on the coarsened graph, whose rows share their sources (tile_common.hpp).  Four kernels
// replace seven per-edge passes over that graph's 9.7 M entries (64 100-qubit circuits):
//
//   tile_pool_fwd_kernel      segment max  +  a_i = w_comp . max_i + b_comp  +  score softmax and cluster sum  +  the three LEConv
//                             projections of x'  -- ONE load of the tile's source union, two walks out of LDS
//                             (was: csr_aggregate<IS_MAX>, a [N,D]x[D,1] GEMM, softmax_aggregate_kernel, a [N,D]x[D,3] GEMM)
//   tile_pool_bwd_dst_kernel  destination side of the softmax-sum's backward, the tie counts of the segment max and its per-row
//                             gradient share  (was: softmax_aggregate_bwd_dst_kernel<.,true>, segment_max_share_from_counts_kernel)
//   tile_pool_bwd_src_kernel  source side of the softmax-sum's backward, weights recomputed from the destination's record
//                             (was: softmax_aggregate_bwd_src_rc_kernel)
//   tile_pool_max_bwd_kernel  source side of the segment max's backward (was: segment_max_bwd_kernel)
//
// Mapping as in the per-edge kernels (attn.hip, family_b_bwd.hip): a row belongs to a 16-lane group, lane l holds channels l, l + 16,
// ... (NV per lane), a chunk is 16 entries, lane u owns entry u's scalar work.  New: rows of at least kTileLongDeg entries are walked
// by the four groups of a WAVE together (chunk c goes to group c mod 4; the partial results meet through LDS in group order), and a
// gathered row is `ds_read`s at the entry's 16-bit slot.  Entries whose source did not fit the union (slot kTileNoSlot) are read from
// global memory in a wave-uniform side branch.
#include <type_traits>

#include "tile_common.hpp"

namespace mlqem {

constexpr int kGroupScratch = 8;            // floats a lane files for its group's partial result

__host__ __device__ inline size_t tile_pool_lds_bytes(int cap, int pitch, int tile_rows) {
  return (size_t)cap * pitch * 4 + (size_t)kBlock * kGroupScratch * 4 + tile_lds_common_bytes(cap, tile_rows);
}

struct TilePoolArgs {
  const float* x; int64_t ldx;              // [N, D] rows the walks gather (forward / dst side: sources; see each kernel)
  const int32_t* ptr; const int32_t* idx;   // the CSR the plan was built for
  const float* c_src;                       // [N]  c_j = x_j . att_x
  float slope; int64_t N; int D;
  // forward
  const float* w_comp; const float* b_comp; // [D], [1]: a_i = w_comp . segmax_i + b_comp
  const float* w3; const float* b3;         // [3, D] row-major, [3]: LEConv's three one-wide projections of x'
  float* xnew; int64_t ldn; float* xmax; int64_t ldm; float4* stat; float* pqr;
  // backward
  const float* gnew; int64_t ldg;           // [N, D] gradient of x'
  float* g_a; float* share; int64_t lds;    // dst side: g_a[N], share[N, D] = g_a[i] w_comp[c] / ties[i, c]
  float* gx; int64_t ldgx; float* g_c; const float* rank1;   // src side
};

// slot -> (pointer into the staged rows); entries past the chunk's end repeat a valid slot (their weight is zero / their compare masked)
struct ChunkSlots {
  int slot;        // the lane's own entry
  bool valid, ovf;
  int j;           // global id of the lane's entry when it overflowed
};
__device__ __forceinline__ ChunkSlots chunk_slots(const TilePlan& p, const TileLds& lc_, const int4& ri, const int32_t* __restrict__ idx, int e0,
                                                  int k, int l) {
  ChunkSlots c;
  c.valid = l < k;
  const int x = e0 + min(l, k - 1);
  const uint32_t lc = tile_slot(p, lc_, ri, x);
  c.ovf = lc == kTileNoSlot;
  c.slot = c.ovf ? 0 : (int)lc;
  c.j = c.ovf ? idx[ri.y + x] : 0;
  return c;
}

// `pieces` 16-byte pieces per slot from row `id` of `src`, then (optionally) one scalar and one float4 per slot
__device__ __forceinline__ void pool_stage(const int* __restrict__ un, int ucnt, const float* __restrict__ src, int64_t ld, int pieces,
                                           float* __restrict__ rows, int pitch, int at) {
  const int total = ucnt * pieces;
  for (int i0 = threadIdx.x; i0 < total; i0 += 4 * kBlock) {
    f4a v[4];
    int dst[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * kBlock;
      dst[u] = -1;
      if (i < total) {
        const int s = i / pieces, pc = i - s * pieces;
        v[u] = *reinterpret_cast<const f4a*>(src + (int64_t)un[s] * ld + 4 * pc);
        dst[u] = s * pitch + at + 4 * pc;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (dst[u] >= 0) *reinterpret_cast<f4a*>(rows + dst[u]) = v[u];
  }
}

// ------------------------------------------------------------------------------------------------------------------ forward
// LDS slot: [ x_j (16 NV floats) | c_j, -, -, - ]
template <int NV> __global__ __launch_bounds__(kBlock) void tile_pool_fwd_kernel(const TilePoolArgs a, const TilePlan p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int XW = 16 * NV, pitch = XW + 4;
  float* rows = reinterpret_cast<float*>(smem);
  float* scratch = rows + (size_t)p.cap * pitch;
  const TileLds LC = tile_lds_carve(reinterpret_cast<char*>(scratch + kBlock * kGroupScratch), p.cap, p.tile_rows);
  const int t = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int4 ti = tile_prologue(p, t, LC);
  const int cnt = ti.x, nlong = ti.y, ucnt = ti.z;
  const int D = a.D;
  const float* __restrict__ x = a.x;
  const int64_t ldx = a.ldx;
  const int* __restrict__ un = LC.uid;
  pool_stage(un, ucnt, x, ldx, (D + 3) / 4, rows, pitch, 0);
  for (int s = threadIdx.x; s < ucnt; s += kBlock) rows[s * pitch + XW] = a.c_src[un[s]];
  tile_stage_loc(p, ti, LC);
  __syncthreads();

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane >> 4, l = lane & 15;
  const float slope = a.slope;
  auto leaky = [&](float v) { return v > 0.f ? v : v * slope; };
  bool has[NV];
  float wc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    has[v] = l + 16 * v < D;
    wc[v] = has[v] ? a.w_comp[l + 16 * v] : 0.f;
  }
  const float bc = a.b_comp[0];
  const int32_t* __restrict__ idx = a.idx;
  float* __restrict__ my_scratch = scratch + (wave * kWave + lane) * kGroupScratch;
  const float* __restrict__ wave_scratch = scratch + (wave * kWave + l) * kGroupScratch;      // + g * 16 * kGroupScratch: group g's lane l

  // the x values of entries U0 .. U0 + CNT - 1 of the chunk
  auto fetch = [&](const ChunkSlots& cs, auto first, auto count, float (&xv)[decltype(count)::value][NV]) {
    constexpr int U0 = decltype(first)::value, CNT = decltype(count)::value;
    int su[CNT];
    if constexpr (CNT == 8) group16_bcast8<U0>(cs.slot, su);
    else { su[0] = group16_bcast<U0>(cs.slot); su[1] = group16_bcast<U0 + 1>(cs.slot); }
#pragma unroll
    for (int u = 0; u < CNT; ++u) {
      const float* xr = rows + su[u] * pitch + l;
#pragma unroll
      for (int v = 0; v < NV; ++v) xv[u][v] = has[v] ? xr[16 * v] : 0.f;
    }
    if (__ballot(cs.ovf) != 0ull) {                      // wave-uniform, rare (lanes past the chunk's end repeat its last entry, overflowed or not)
      for (int u = 0; u < CNT; ++u) {
        const int src = (lane & ~15) + U0 + u;
        const int o = __shfl(cs.ovf ? 1 : 0, src, 64), j = __shfl(cs.j, src, 64);
        if (o) {
#pragma unroll
          for (int v = 0; v < NV; ++v) xv[u][v] = has[v] ? x[(int64_t)j * ldx + l + 16 * v] : 0.f;
        }
      }
    }
  };

  // segment max over the chunks c0, c0 + cstep, ... of the row
  auto walk_max = [&](const int4& ri, int c0, int cstep, float (&mx)[NV]) {
    const int deg = ri.z;
    const int nch = (deg + 15) >> 4;
    for (int c = c0; c < nch; c += cstep) {
      const int e0 = 16 * c, k = min(16, deg - e0);
      const ChunkSlots cs = chunk_slots(p, LC, ri, idx, e0, k, l);
      auto part = [&](auto first, auto count) {          // entries past k repeat entry k - 1: the maximum does not change
        constexpr int CNT = decltype(count)::value;
        float xv[CNT][NV];
        fetch(cs, first, count, xv);
#pragma unroll
        for (int u = 0; u < CNT; ++u)
#pragma unroll
          for (int v = 0; v < NV; ++v) mx[v] = fmaxf(mx[v], xv[u][v]);
      };
      if (k <= 2) part(EdgeChunk<0>{}, EdgeChunk<2>{});
      else {
        part(EdgeChunk<0>{}, EdgeChunk<8>{});
        if (k > 8) part(EdgeChunk<8>{}, EdgeChunk<8>{});
      }
    }
  };
  // score softmax + weighted sum over the same chunks
  auto walk_sum = [&](const int4& ri, int c0, int cstep, float ai, float& m, float& den, float (&acc)[NV]) {
    const int deg = ri.z;
    const int nch = (deg + 15) >> 4;
    for (int c = c0; c < nch; c += cstep) {
      const int e0 = 16 * c, k = min(16, deg - e0);
      const ChunkSlots cs = chunk_slots(p, LC, ri, idx, e0, k, l);
      const float cj = cs.ovf ? a.c_src[cs.j] : rows[cs.slot * pitch + XW];
      const float s = cs.valid ? leaky(ai + cj) : -INFINITY;
      const float cm = group16_max(s);
      if (cm > m) {
        const float r = expf(m - cm);
        den *= r;
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] *= r;
        m = cm;
      }
      const float pe = cs.valid ? expf(s - m) : 0.f;
      den += group16_sum(pe);
      auto part = [&](auto first, auto count) {
        constexpr int U0 = decltype(first)::value, CNT = decltype(count)::value;
        float xv[CNT][NV], pu[CNT];
        fetch(cs, first, count, xv);
        if constexpr (CNT == 8) group16_bcast8<U0>(pe, pu);
        else { pu[0] = group16_bcast<U0>(pe); pu[1] = group16_bcast<U0 + 1>(pe); }
#pragma unroll
        for (int u = 0; u < CNT; ++u)
#pragma unroll
          for (int v = 0; v < NV; ++v) acc[v] = fmaf(pu[u], xv[u][v], acc[v]);
      };
      if (k <= 2) part(EdgeChunk<0>{}, EdgeChunk<2>{});
      else {
        part(EdgeChunk<0>{}, EdgeChunk<8>{});
        if (k > 8) part(EdgeChunk<8>{}, EdgeChunk<8>{});
      }
    }
  };
  auto self_entry = [&](int row, float ai, const float (&xs)[NV], float& m, float& den, float (&acc)[NV]) {
    const float s = leaky(ai + a.c_src[row]);
    if (s > m) {
      const float r = expf(m - s);
      den *= r;
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[v] *= r;
      m = s;
    }
    const float pe = expf(s - m);
    den += pe;
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = fmaf(pe, xs[v], acc[v]);
  };
  auto finish = [&](int row, float ai, const float (&mx)[NV], float m, float den, float (&acc)[NV]) {
    const float inv = 1.0f / (den + 1e-16f);
    float dots[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      acc[v] *= inv;
      if (has[v]) {
        const int c = l + 16 * v;
        a.xnew[(int64_t)row * a.ldn + c] = acc[v];
        a.xmax[(int64_t)row * a.ldm + c] = mx[v];
#pragma unroll
        for (int k3 = 0; k3 < 3; ++k3) dots[k3] = fmaf(acc[v], a.w3[k3 * D + c], dots[k3]);
      }
    }
#pragma unroll
    for (int k3 = 0; k3 < 3; ++k3) dots[k3] = group16_sum(dots[k3]);
    if (l == 0) {
      a.stat[row] = make_float4(ai, m, inv, 0.f);
#pragma unroll
      for (int k3 = 0; k3 < 3; ++k3) a.pqr[(int64_t)row * 3 + k3] = dots[k3] + a.b3[k3];
    }
  };
  auto own_row = [&](int row, float (&xs)[NV]) {
#pragma unroll
    for (int v = 0; v < NV; ++v) xs[v] = has[v] ? x[(int64_t)row * ldx + l + 16 * v] : 0.f;
  };
  auto score_of = [&](const float (&mx)[NV]) {
    float d = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) d = fmaf(wc[v], has[v] ? mx[v] : 0.f, d);
    return group16_sum(d) + bc;
  };

  // long rows: the wave's four groups walk one row together
  for (int r = wave; r < nlong; r += 4) {
    const int4 ri = LC.rinfo[r];
    const int row = ri.x;
    float xs[NV], mx[NV];
    own_row(row, xs);
#pragma unroll
    for (int v = 0; v < NV; ++v) mx[v] = xs[v];          // the row itself takes part in its maximum (add_remaining_self_loops)
    walk_max(ri, grp, 4, mx);
#pragma unroll
    for (int v = 0; v < NV; ++v) my_scratch[v] = mx[v];
    wave_sync();
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int v = 0; v < NV; ++v) mx[v] = fmaxf(mx[v], wave_scratch[g * 16 * kGroupScratch + v]);
    wave_sync();
    const float ai = score_of(mx);
    float m = -INFINITY, den = 0.f, acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0.f;
    walk_sum(ri, grp, 4, ai, m, den, acc);
    my_scratch[0] = m; my_scratch[1] = den;
#pragma unroll
    for (int v = 0; v < NV; ++v) my_scratch[2 + v] = acc[v];
    wave_sync();
    if (grp == 0) {
      float mm = -INFINITY;
#pragma unroll
      for (int g = 0; g < 4; ++g) mm = fmaxf(mm, wave_scratch[g * 16 * kGroupScratch]);
      float dd = 0.f, aa[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) aa[v] = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float* __restrict__ pg = wave_scratch + g * 16 * kGroupScratch;
        const float rr = pg[0] > -INFINITY ? expf(pg[0] - mm) : 0.f;
        dd = fmaf(pg[1], rr, dd);
#pragma unroll
        for (int v = 0; v < NV; ++v) aa[v] = fmaf(pg[2 + v], rr, aa[v]);
      }
      self_entry(row, ai, xs, mm, dd, aa);
      finish(row, ai, mx, mm, dd, aa);
    }
    wave_sync();
  }
  // short rows: a group per row
  const int nshort = cnt - nlong;
  for (int r0 = wave * 4; r0 < nshort; r0 += 16) {
    const int rix = r0 + grp;
    if (rix < nshort) {
      const int4 ri = LC.rinfo[nlong + rix];
      const int row = ri.x;
      float xs[NV], mx[NV];
      own_row(row, xs);
#pragma unroll
      for (int v = 0; v < NV; ++v) mx[v] = xs[v];
      walk_max(ri, 0, 1, mx);
      const float ai = score_of(mx);
      float m = -INFINITY, den = 0.f, acc[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[v] = 0.f;
      walk_sum(ri, 0, 1, ai, m, den, acc);
      self_entry(row, ai, xs, m, den, acc);
      finish(row, ai, mx, m, den, acc);
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward, destination side
// Per row i (softmax_aggregate_bwd_dst_kernel's formulas; statistics from the forward's record instead of a walk of their own):
//   al_ij = exp(LeakyReLU(a_i + c_j) - m_i) / den_i,  gp_ij = al_ij (gnew_i . x_j - delta_i) LeakyReLU'(a_i + c_j),  g_a[i] = sum_j gp_ij,
//   ties[i, c] = #{entries j of row i, i itself included, with x_j[c] == xmax[i, c]},  share[i, c] = g_a[i] w_comp[c] / ties[i, c]
// (the gradient of the segment max is the rank-one g_a (x) w_comp of the composed score projection, functional._ASAPool).
// Same LDS slot as the forward.
template <int NV> __global__ __launch_bounds__(kBlock) void tile_pool_bwd_dst_kernel(const TilePoolArgs a, const TilePlan p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int XW = 16 * NV, pitch = XW + 4;
  float* rows = reinterpret_cast<float*>(smem);
  float* scratch = rows + (size_t)p.cap * pitch;
  const TileLds LC = tile_lds_carve(reinterpret_cast<char*>(scratch + kBlock * kGroupScratch), p.cap, p.tile_rows);
  const int t = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int4 ti = tile_prologue(p, t, LC);
  const int cnt = ti.x, nlong = ti.y, ucnt = ti.z;
  const int D = a.D;
  const float* __restrict__ x = a.x;
  const int64_t ldx = a.ldx;
  const int* __restrict__ un = LC.uid;
  pool_stage(un, ucnt, x, ldx, (D + 3) / 4, rows, pitch, 0);
  for (int s = threadIdx.x; s < ucnt; s += kBlock) rows[s * pitch + XW] = a.c_src[un[s]];
  tile_stage_loc(p, ti, LC);
  __syncthreads();

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane >> 4, l = lane & 15;
  const float slope = a.slope;
  auto leaky = [&](float v) { return v > 0.f ? v : v * slope; };
  bool has[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) has[v] = l + 16 * v < D;
  const int32_t* __restrict__ idx = a.idx;
  float* __restrict__ my_scratch = scratch + (wave * kWave + lane) * kGroupScratch;
  const float* __restrict__ wave_scratch = scratch + (wave * kWave + l) * kGroupScratch;

  struct Row { float gi[NV], mx[NV], ai, m, inv, delta; };
  auto load_row = [&](int row) {
    Row r;
    float d = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = l + 16 * v;
      r.gi[v] = has[v] ? a.gnew[(int64_t)row * a.ldg + c] : 0.f;
      r.mx[v] = has[v] ? a.xmax[(int64_t)row * a.ldm + c] : 0.f;
      d = fmaf(r.gi[v], has[v] ? a.xnew[(int64_t)row * a.ldn + c] : 0.f, d);
    }
    r.delta = group16_sum(d);
    const float4 st = a.stat[row];
    r.ai = st.x; r.m = st.y; r.inv = st.z;
    return r;
  };
  // ga: lane u sums the gp of its entries; ties: per lane channel
  auto walk = [&](const int4& ri, int c0, int cstep, const Row& r, float& ga, int (&ties)[NV]) {
    const int deg = ri.z;
    const int nch = (deg + 15) >> 4;
    for (int c = c0; c < nch; c += cstep) {
      const int e0 = 16 * c, k = min(16, deg - e0);
      const ChunkSlots cs = chunk_slots(p, LC, ri, idx, e0, k, l);
      const bool o = cs.ovf;
      const float cj = o ? a.c_src[cs.j] : rows[cs.slot * pitch + XW];
      float mydot = 0.f;
      auto part = [&](auto first, auto count) {
        constexpr int U0 = decltype(first)::value, CNT = decltype(count)::value;
        int su[CNT];
        float xv[CNT][NV];
        if constexpr (CNT == 8) group16_bcast8<U0>(cs.slot, su);
        else { su[0] = group16_bcast<U0>(cs.slot); su[1] = group16_bcast<U0 + 1>(cs.slot); }
#pragma unroll
        for (int u = 0; u < CNT; ++u) {
          const float* xr = rows + su[u] * pitch + l;
#pragma unroll
          for (int v = 0; v < NV; ++v) xv[u][v] = has[v] ? xr[16 * v] : 0.f;
        }
        if (__ballot(o) != 0ull) {
          for (int u = 0; u < CNT; ++u) {
            const int src = (lane & ~15) + U0 + u;
            const int oo = __shfl(o ? 1 : 0, src, 64), j = __shfl(cs.j, src, 64);
            if (oo) {
#pragma unroll
              for (int v = 0; v < NV; ++v) xv[u][v] = has[v] ? x[(int64_t)j * ldx + l + 16 * v] : 0.f;
            }
          }
        }
#pragma unroll
        for (int u = 0; u < CNT; ++u) {
          float dd = 0.f;
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            dd = fmaf(r.gi[v], xv[u][v], dd);
            ties[v] += (U0 + u < k && has[v] && xv[u][v] == r.mx[v]) ? 1 : 0;
          }
          dd = group16_sum(dd);
          if (l == U0 + u) mydot = dd;
        }
      };
      if (k <= 2) part(EdgeChunk<0>{}, EdgeChunk<2>{});
      else {
        part(EdgeChunk<0>{}, EdgeChunk<8>{});
        if (k > 8) part(EdgeChunk<8>{}, EdgeChunk<8>{});
      }
      if (cs.valid) {
        const float pre = r.ai + cj;
        const float al = expf(leaky(pre) - r.m) * r.inv;
        ga += al * (mydot - r.delta) * (pre > 0.f ? 1.f : slope);
      }
    }
  };
  auto finish = [&](int row, const Row& r, float ga_edges, int (&ties)[NV]) {     // ga_edges: already summed over the group
    const float pre = r.ai + a.c_src[row];
    const float al = expf(leaky(pre) - r.m) * r.inv;
    float dd = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const float xs = has[v] ? x[(int64_t)row * ldx + l + 16 * v] : 0.f;
      dd = fmaf(r.gi[v], xs, dd);
      ties[v] += (has[v] && xs == r.mx[v]) ? 1 : 0;
    }
    dd = group16_sum(dd);
    const float ga = ga_edges + al * (dd - r.delta) * (pre > 0.f ? 1.f : slope);
#pragma unroll
    for (int v = 0; v < NV; ++v)
      if (has[v]) a.share[(int64_t)row * a.lds + l + 16 * v] = ga * a.w_comp[l + 16 * v] / (float)(ties[v] > 0 ? ties[v] : 1);
    if (l == 0) {
      a.g_a[row] = ga;
      a.stat[row] = make_float4(r.ai, r.m, r.inv, r.delta);
    }
  };

  for (int r0 = wave; r0 < nlong; r0 += 4) {
    const int4 ri = LC.rinfo[r0];
    const int row = ri.x;
    const Row r = load_row(row);
    float ga = 0.f;
    int ties[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) ties[v] = 0;
    walk(ri, grp, 4, r, ga, ties);
    my_scratch[0] = group16_sum(ga);
#pragma unroll
    for (int v = 0; v < NV; ++v) my_scratch[1 + v] = (float)ties[v];
    wave_sync();
    if (grp == 0) {
      float gsum = 0.f;
      int tt[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) tt[v] = 0;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float* __restrict__ pg = wave_scratch + g * 16 * kGroupScratch;
        gsum += pg[0];
#pragma unroll
        for (int v = 0; v < NV; ++v) tt[v] += (int)pg[1 + v];
      }
      finish(row, r, gsum, tt);
    }
    wave_sync();
  }
  const int nshort = cnt - nlong;
  for (int r0 = wave * 4; r0 < nshort; r0 += 16) {
    const int rix = r0 + grp;
    if (rix < nshort) {
      const int4 ri = LC.rinfo[nlong + rix];
      const int row = ri.x;
      const Row r = load_row(row);
      float ga = 0.f;
      int ties[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) ties[v] = 0;
      walk(ri, 0, 1, r, ga, ties);
      finish(row, r, group16_sum(ga), ties);
    }
  }
}

// ------------------------------------------------------------------------------------------------------ backward, source side
// On a plan of the OUT-CSR (the union holds DESTINATION rows i of the tile's source rows j):
//   g_x[j] = sum_i al_ij gnew_i + g_c[j] rank1,   g_c[j] = sum_i gp_ij   (self entry included; softmax_aggregate_bwd_src_rc_kernel)
// LDS slot: [ gnew_i (16 NV) | a_i, m_i, 1 / den_i, delta_i ]
template <int NV> __global__ __launch_bounds__(kBlock) void tile_pool_bwd_src_kernel(const TilePoolArgs a, const TilePlan p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int XW = 16 * NV, pitch = XW + 4;
  float* rows = reinterpret_cast<float*>(smem);
  float* scratch = rows + (size_t)p.cap * pitch;
  const TileLds LC = tile_lds_carve(reinterpret_cast<char*>(scratch + kBlock * kGroupScratch), p.cap, p.tile_rows);
  const int t = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int4 ti = tile_prologue(p, t, LC);
  const int cnt = ti.x, nlong = ti.y, ucnt = ti.z;
  const int D = a.D;
  const float* __restrict__ gnew = a.gnew;
  const int64_t ldg = a.ldg;
  const int* __restrict__ un = LC.uid;
  pool_stage(un, ucnt, gnew, ldg, (D + 3) / 4, rows, pitch, 0);
  for (int s = threadIdx.x; s < ucnt; s += kBlock) {
    const float4 st = a.stat[un[s]];
    *reinterpret_cast<f4a*>(rows + s * pitch + XW) = f4a{st.x, st.y, st.z, st.w};
  }
  tile_stage_loc(p, ti, LC);
  __syncthreads();

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane >> 4, l = lane & 15;
  const float slope = a.slope;
  auto leaky = [&](float v) { return v > 0.f ? v : v * slope; };
  bool has[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) has[v] = l + 16 * v < D;
  const int32_t* __restrict__ idx = a.idx;
  float* __restrict__ my_scratch = scratch + (wave * kWave + lane) * kGroupScratch;
  const float* __restrict__ wave_scratch = scratch + (wave * kWave + l) * kGroupScratch;

  auto walk = [&](const int4& ri, int c0, int cstep, const float (&xr)[NV], float cj, float (&acc)[NV], float& gc) {
    const int deg = ri.z;
    const int nch = (deg + 15) >> 4;
    for (int c = c0; c < nch; c += cstep) {
      const int e0 = 16 * c, k = min(16, deg - e0);
      const ChunkSlots cs = chunk_slots(p, LC, ri, idx, e0, k, l);
      const bool o = cs.ovf;
      f4a st = *reinterpret_cast<const f4a*>(rows + cs.slot * pitch + XW);
      if (o) {
        const float4 g4 = a.stat[cs.j];
        st = f4a{g4.x, g4.y, g4.z, g4.w};
      }
      const float pre = st.x + cj;
      const float al = cs.valid ? expf(leaky(pre) - st.y) * st.z : 0.f;
      float mydot = 0.f;
      auto part = [&](auto first, auto count) {
        constexpr int U0 = decltype(first)::value, CNT = decltype(count)::value;
        int su[CNT];
        float gn[CNT][NV], au[CNT];
        if constexpr (CNT == 8) { group16_bcast8<U0>(cs.slot, su); group16_bcast8<U0>(al, au); }
        else {
          su[0] = group16_bcast<U0>(cs.slot); su[1] = group16_bcast<U0 + 1>(cs.slot);
          au[0] = group16_bcast<U0>(al); au[1] = group16_bcast<U0 + 1>(al);
        }
#pragma unroll
        for (int u = 0; u < CNT; ++u) {
          const float* gr = rows + su[u] * pitch + l;
#pragma unroll
          for (int v = 0; v < NV; ++v) gn[u][v] = has[v] ? gr[16 * v] : 0.f;
        }
        if (__ballot(o) != 0ull) {
          for (int u = 0; u < CNT; ++u) {
            const int src = (lane & ~15) + U0 + u;
            const int oo = __shfl(o ? 1 : 0, src, 64), j = __shfl(cs.j, src, 64);
            if (oo) {
#pragma unroll
              for (int v = 0; v < NV; ++v) gn[u][v] = has[v] ? gnew[(int64_t)j * ldg + l + 16 * v] : 0.f;
            }
          }
        }
#pragma unroll
        for (int u = 0; u < CNT; ++u) {
          float dd = 0.f;
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            acc[v] = fmaf(au[u], gn[u][v], acc[v]);
            dd = fmaf(gn[u][v], xr[v], dd);
          }
          dd = group16_sum(dd);
          if (l == U0 + u) mydot = dd;
        }
      };
      if (k <= 2) part(EdgeChunk<0>{}, EdgeChunk<2>{});
      else {
        part(EdgeChunk<0>{}, EdgeChunk<8>{});
        if (k > 8) part(EdgeChunk<8>{}, EdgeChunk<8>{});
      }
      if (cs.valid) gc += al * (mydot - st.w) * (pre > 0.f ? 1.f : slope);
    }
  };
  auto own = [&](int row, float (&xr)[NV], float& cj) {
    cj = a.c_src[row];
#pragma unroll
    for (int v = 0; v < NV; ++v) xr[v] = has[v] ? a.x[(int64_t)row * a.ldx + l + 16 * v] : 0.f;
  };
  auto finish = [&](int row, const float (&xr)[NV], float cj, float (&acc)[NV], float gc_edges) {    // gc_edges: summed over the group
    const float4 st = a.stat[row];
    const float pre = st.x + cj;
    const float al = expf(leaky(pre) - st.y) * st.z;
    float dd = 0.f, gn[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      gn[v] = has[v] ? gnew[(int64_t)row * ldg + l + 16 * v] : 0.f;
      dd = fmaf(gn[v], xr[v], dd);
    }
    dd = group16_sum(dd);
    const float gc = gc_edges + al * (dd - st.w) * (pre > 0.f ? 1.f : slope);
#pragma unroll
    for (int v = 0; v < NV; ++v)
      if (has[v]) a.gx[(int64_t)row * a.ldgx + l + 16 * v] = fmaf(al, gn[v], acc[v]) + (a.rank1 ? gc * a.rank1[l + 16 * v] : 0.f);
    if (l == 0) a.g_c[row] = gc;
  };

  for (int r0 = wave; r0 < nlong; r0 += 4) {
    const int4 ri = LC.rinfo[r0];
    const int row = ri.x;
    float xr[NV], cj, acc[NV], gc = 0.f;
    own(row, xr, cj);
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0.f;
    walk(ri, grp, 4, xr, cj, acc, gc);
    my_scratch[0] = group16_sum(gc);
#pragma unroll
    for (int v = 0; v < NV; ++v) my_scratch[1 + v] = acc[v];
    wave_sync();
    if (grp == 0) {
      float gsum = 0.f, aa[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) aa[v] = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float* __restrict__ pg = wave_scratch + g * 16 * kGroupScratch;
        gsum += pg[0];
#pragma unroll
        for (int v = 0; v < NV; ++v) aa[v] += pg[1 + v];
      }
      finish(row, xr, cj, aa, gsum);
    }
    wave_sync();
  }
  const int nshort = cnt - nlong;
  for (int r0 = wave * 4; r0 < nshort; r0 += 16) {
    const int rix = r0 + grp;
    if (rix < nshort) {
      const int4 ri = LC.rinfo[nlong + rix];
      const int row = ri.x;
      float xr[NV], cj, acc[NV], gc = 0.f;
      own(row, xr, cj);
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[v] = 0.f;
      walk(ri, 0, 1, xr, cj, acc, gc);
      finish(row, xr, cj, acc, group16_sum(gc));
    }
  }
}

// Source side of the segment max's backward, same out-plan:  g_x[j, c] += sum over the destinations i of j, and j itself, whose
// maximum equals x[j, c], of share[i, c]  (segment_max_bwd_kernel).  LDS slot: [ xmax_i (16 NV) | share_i (16 NV) ]
template <int NV> __global__ __launch_bounds__(kBlock) void tile_pool_max_bwd_kernel(const TilePoolArgs a, const TilePlan p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int XW = 16 * NV, pitch = 2 * XW + 4;
  float* rows = reinterpret_cast<float*>(smem);
  float* scratch = rows + (size_t)p.cap * pitch;
  const TileLds LC = tile_lds_carve(reinterpret_cast<char*>(scratch + kBlock * kGroupScratch), p.cap, p.tile_rows);
  const int t = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int4 ti = tile_prologue(p, t, LC);
  const int cnt = ti.x, nlong = ti.y, ucnt = ti.z;
  const int D = a.D;
  const int* __restrict__ un = LC.uid;
  pool_stage(un, ucnt, a.xmax, a.ldm, (D + 3) / 4, rows, pitch, 0);
  pool_stage(un, ucnt, a.share, a.lds, (D + 3) / 4, rows, pitch, XW);
  tile_stage_loc(p, ti, LC);
  __syncthreads();

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane >> 4, l = lane & 15;
  bool has[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) has[v] = l + 16 * v < D;
  const int32_t* __restrict__ idx = a.idx;
  float* __restrict__ my_scratch = scratch + (wave * kWave + lane) * kGroupScratch;
  const float* __restrict__ wave_scratch = scratch + (wave * kWave + l) * kGroupScratch;

  auto walk = [&](const int4& ri, int c0, int cstep, const float (&xv)[NV], float (&acc)[NV]) {
    const int deg = ri.z;
    const int nch = (deg + 15) >> 4;
    for (int c = c0; c < nch; c += cstep) {
      const int e0 = 16 * c, k = min(16, deg - e0);
      const ChunkSlots cs = chunk_slots(p, LC, ri, idx, e0, k, l);
      const bool o = cs.ovf;
      auto part = [&](auto first, auto count) {
        constexpr int U0 = decltype(first)::value, CNT = decltype(count)::value;
        int su[CNT];
        float xm[CNT][NV], gm[CNT][NV];
        if constexpr (CNT == 8) group16_bcast8<U0>(cs.slot, su);
        else { su[0] = group16_bcast<U0>(cs.slot); su[1] = group16_bcast<U0 + 1>(cs.slot); }
#pragma unroll
        for (int u = 0; u < CNT; ++u) {
          const float* rr = rows + su[u] * pitch + l;
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            xm[u][v] = has[v] ? rr[16 * v] : 0.f;
            gm[u][v] = has[v] ? rr[XW + 16 * v] : 0.f;
          }
        }
        if (__ballot(o) != 0ull) {
          for (int u = 0; u < CNT; ++u) {
            const int src = (lane & ~15) + U0 + u;
            const int oo = __shfl(o ? 1 : 0, src, 64), j = __shfl(cs.j, src, 64);
            if (oo) {
#pragma unroll
              for (int v = 0; v < NV; ++v) {
                xm[u][v] = has[v] ? a.xmax[(int64_t)j * a.ldm + l + 16 * v] : 0.f;
                gm[u][v] = has[v] ? a.share[(int64_t)j * a.lds + l + 16 * v] : 0.f;
              }
            }
          }
        }
#pragma unroll
        for (int u = 0; u < CNT; ++u)
#pragma unroll
          for (int v = 0; v < NV; ++v)
            if (U0 + u < k && has[v] && xv[v] == xm[u][v]) acc[v] += gm[u][v];
      };
      if (k <= 2) part(EdgeChunk<0>{}, EdgeChunk<2>{});
      else {
        part(EdgeChunk<0>{}, EdgeChunk<8>{});
        if (k > 8) part(EdgeChunk<8>{}, EdgeChunk<8>{});
      }
    }
  };
  auto own = [&](int row, float (&xv)[NV]) {
#pragma unroll
    for (int v = 0; v < NV; ++v) xv[v] = has[v] ? a.x[(int64_t)row * a.ldx + l + 16 * v] : 0.f;
  };
  auto finish = [&](int row, const float (&xv)[NV], const float (&acc)[NV]) {
#pragma unroll
    for (int v = 0; v < NV; ++v)
      if (has[v]) {
        const int c = l + 16 * v;
        const float self = xv[v] == a.xmax[(int64_t)row * a.ldm + c] ? a.share[(int64_t)row * a.lds + c] : 0.f;
        a.gx[(int64_t)row * a.ldgx + c] += self + acc[v];
      }
  };

  for (int r0 = wave; r0 < nlong; r0 += 4) {
    const int4 ri = LC.rinfo[r0];
    const int row = ri.x;
    float xv[NV], acc[NV];
    own(row, xv);
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0.f;
    walk(ri, grp, 4, xv, acc);
#pragma unroll
    for (int v = 0; v < NV; ++v) my_scratch[v] = acc[v];
    wave_sync();
    if (grp == 0) {
      float aa[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) aa[v] = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int v = 0; v < NV; ++v) aa[v] += wave_scratch[g * 16 * kGroupScratch + v];
      finish(row, xv, aa);
    }
    wave_sync();
  }
  const int nshort = cnt - nlong;
  for (int r0 = wave * 4; r0 < nshort; r0 += 16) {
    const int rix = r0 + grp;
    if (rix < nshort) {
      const int4 ri = LC.rinfo[nlong + rix];
      const int row = ri.x;
      float xv[NV], acc[NV];
      own(row, xv);
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[v] = 0.f;
      walk(ri, 0, 1, xv, acc);
      finish(row, xv, acc);
    }
  }
}

}  // namespace mlqem

using namespace mlqem;

static bool plan_ok(const int32_t* tinfo, const int32_t* rinfo, const int32_t* uni, const uint16_t* loc, int64_t nt, int cap, int tile_rows) {
  return nt >= 0 && nt <= INT32_MAX && cap > 0 && cap < (int)kTileNoSlot && tile_rows > 0 && tile_rows <= kTileMaxRows &&
         (nt == 0 || (tinfo && rinfo && uni && loc && aligned_to(tinfo, 16) && aligned_to(rinfo, 16)));
}
static bool rows16(const float* p, int64_t ld, int D) { return p && aligned_to(p, 16) && ld % 4 == 0 && ld >= (D + 3) / 4 * 4; }

// Largest slot count of a plan whose pooling kernels fit `lds_bytes` per workgroup (the widest slot: xmax | share of the max backward).
extern "C" int mlqem_tile_pool_cap(int D, int tile_rows, int lds_bytes) {
  if (D <= 0 || D > 48 || tile_rows <= 0) return 0;
  const int NV = (D + 15) / 16, pitch = 2 * 16 * NV + 4;
  const int64_t room = (int64_t)lds_bytes - (int64_t)kBlock * kGroupScratch * 4 - (int64_t)tile_lds_common_bytes(0, tile_rows) - 16;
  return (int)std::max<int64_t>(0, room / ((int64_t)pitch * 4 + 4));
}

#define MLQEM_TILE_POOL(KERNEL, PITCH_OF_NV, PLAN, TILES, CAP)                                                                    \
  do {                                                                                                                            \
    const int NV = (D + 15) / 16;                                                                                                 \
    const size_t lds = tile_pool_lds_bytes(CAP, PITCH_OF_NV, (PLAN).tile_rows);                                                                     \
    if (NV == 1) {                                                                                                                \
      if (!ensure_dynamic_lds(KERNEL<1>, lds)) return MLQEM_ERR_UNSUPPORTED;                                                      \
      hipLaunchKernelGGL(KERNEL<1>, dim3((unsigned)(TILES)), dim3(kBlock), lds, as_stream(stream), a, PLAN);                       \
    } else if (NV == 2) {                                                                                                         \
      if (!ensure_dynamic_lds(KERNEL<2>, lds)) return MLQEM_ERR_UNSUPPORTED;                                                      \
      hipLaunchKernelGGL(KERNEL<2>, dim3((unsigned)(TILES)), dim3(kBlock), lds, as_stream(stream), a, PLAN);                       \
    } else {                                                                                                                      \
      if (!ensure_dynamic_lds(KERNEL<3>, lds)) return MLQEM_ERR_UNSUPPORTED;                                                      \
      hipLaunchKernelGGL(KERNEL<3>, dim3((unsigned)(TILES)), dim3(kBlock), lds, as_stream(stream), a, PLAN);                       \
    }                                                                                                                             \
  } while (0)

// ASAPooling steps 2-4 and LEConv's projections in one pass over the in-CSR plan.
extern "C" int mlqem_tile_asap_scores_f32(const float* x, int64_t ldx, const int32_t* in_ptr, const int32_t* in_src, const float* c_src,
                                          const float* w_comp, const float* b_comp, const float* w3, const float* b3, float negative_slope,
                                          int64_t N, int D, const int32_t* tinfo, const int32_t* rinfo, const int32_t* uni, const uint16_t* loc,
                                          int64_t num_tiles, int cap, int tile_rows, float* xnew, int64_t ldn, float* xmax, int64_t ldm, float* stat,
                                          float* pqr, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || D <= 0 || ldn < D || ldm < D) return MLQEM_ERR_BAD_ARG;
  if (D > 48) return MLQEM_ERR_UNSUPPORTED;
  if (!plan_ok(tinfo, rinfo, uni, loc, num_tiles, cap, tile_rows)) return MLQEM_ERR_BAD_ARG;
  if (N == 0 || num_tiles == 0) return MLQEM_OK;
  if (!in_ptr || !c_src || !w_comp || !b_comp || !w3 || !b3 || !xnew || !xmax || !stat || !pqr || !aligned_to(stat, 16)) return MLQEM_ERR_BAD_ARG;
  if (!rows16(x, ldx, D)) return MLQEM_ERR_BAD_ARG;
  if (N > INT32_MAX) return MLQEM_ERR_UNSUPPORTED;
  TilePoolArgs a = {};
  a.x = x; a.ldx = ldx; a.ptr = in_ptr; a.idx = in_src; a.c_src = c_src; a.slope = negative_slope; a.N = N; a.D = D;
  a.w_comp = w_comp; a.b_comp = b_comp; a.w3 = w3; a.b3 = b3;
  a.xnew = xnew; a.ldn = ldn; a.xmax = xmax; a.ldm = ldm; a.stat = reinterpret_cast<float4*>(stat); a.pqr = pqr;
  const TilePlan p{reinterpret_cast<const int4*>(tinfo), reinterpret_cast<const int4*>(rinfo), uni, loc, num_tiles, cap, tile_rows};
  MLQEM_TILE_POOL(tile_pool_fwd_kernel, 16 * NV + 4, p, num_tiles, cap);
  return launch_status();
}

// The backward of the same: destination side on the in-CSR plan, the two source sides on the out-CSR plan.
extern "C" int mlqem_tile_asap_scores_bwd_f32(const float* x, int64_t ldx, const float* xnew, int64_t ldn, const float* gnew, int64_t ldg,
                                              const float* xmax, int64_t ldm, const int32_t* in_ptr, const int32_t* in_src,
                                              const int32_t* out_ptr, const int32_t* out_dst, const float* c_src, const float* w_comp,
                                              const float* rank1, float negative_slope, int64_t N, int D, const int32_t* in_tinfo,
                                              const int32_t* in_rinfo, const int32_t* in_uni, const uint16_t* in_loc, int64_t in_tiles, int in_cap,
                                              int in_tile_rows, const int32_t* out_tinfo, const int32_t* out_rinfo, const int32_t* out_uni,
                                              const uint16_t* out_loc, int64_t out_tiles, int out_cap, int out_tile_rows, float* stat, float* g_a, float* share, int64_t lds_,
                                              float* gx, int64_t ldgx, float* g_c, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || D <= 0 || ldn < D) return MLQEM_ERR_BAD_ARG;
  if (D > 48) return MLQEM_ERR_UNSUPPORTED;
  if (!plan_ok(in_tinfo, in_rinfo, in_uni, in_loc, in_tiles, in_cap, in_tile_rows) ||
      !plan_ok(out_tinfo, out_rinfo, out_uni, out_loc, out_tiles, out_cap, out_tile_rows))
    return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!xnew || !in_ptr || !out_ptr || !c_src || !w_comp || !stat || !aligned_to(stat, 16) || !g_a || !gx || !g_c || ldgx < D) return MLQEM_ERR_BAD_ARG;
  if (!rows16(x, ldx, D) || !rows16(gnew, ldg, D) || !rows16(xmax, ldm, D) || !rows16(share, lds_, D)) return MLQEM_ERR_BAD_ARG;
  if (N > INT32_MAX) return MLQEM_ERR_UNSUPPORTED;
  TilePoolArgs a = {};
  a.x = x; a.ldx = ldx; a.c_src = c_src; a.slope = negative_slope; a.N = N; a.D = D; a.w_comp = w_comp;
  a.xnew = const_cast<float*>(xnew); a.ldn = ldn; a.xmax = const_cast<float*>(xmax); a.ldm = ldm; a.stat = reinterpret_cast<float4*>(stat);
  a.gnew = gnew; a.ldg = ldg; a.g_a = g_a; a.share = share; a.lds = lds_; a.gx = gx; a.ldgx = ldgx; a.g_c = g_c; a.rank1 = rank1;
  const TilePlan pin{reinterpret_cast<const int4*>(in_tinfo), reinterpret_cast<const int4*>(in_rinfo), in_uni, in_loc, in_tiles, in_cap, in_tile_rows};
  const TilePlan pout{reinterpret_cast<const int4*>(out_tinfo), reinterpret_cast<const int4*>(out_rinfo), out_uni, out_loc, out_tiles, out_cap,
                      out_tile_rows};
  a.ptr = in_ptr; a.idx = in_src;
  if (in_tiles > 0) MLQEM_TILE_POOL(tile_pool_bwd_dst_kernel, 16 * NV + 4, pin, in_tiles, in_cap);
  a.ptr = out_ptr; a.idx = out_dst;
  if (out_tiles > 0) {
    MLQEM_TILE_POOL(tile_pool_bwd_src_kernel, 16 * NV + 4, pout, out_tiles, out_cap);
    MLQEM_TILE_POOL(tile_pool_max_bwd_kernel, 32 * NV + 4, pout, out_tiles, out_cap);
  }
  return launch_status();
}
