// CSR aggregation kernels: the "scatter-add" of the GNN layers, written as a gather over destination rows
// so that no atomics are needed and every output row is written exactly once (deterministic).
//
// Work decomposition (csr_aggregate_kernel).  A 256-thread workgroup owns R consecutive destination rows and
// R*CV "items", an item being one row x one VEC-wide channel slice (CV = C / VEC; R = 1024 / CV, so every
// thread has 4 items whatever C is -- C runs from 1 to 125 on this path).  Items are numbered row-major, so
// consecutive lanes write consecutive addresses of out[] and the lanes of one row read one contiguous source
// row per edge.
//   phase 1  the block's slice of rowptr / rscale / dself goes to LDS with coalesced loads
//   phase 2  every thread reads, for ALL of its items at once, the first two col[] entries of the row (op nodes
//            have in-degree <= 2 except barriers; consecutive rows own consecutive col[] entries, so a wave reads
//            one or two cache lines), then issues the loads of the self row and of those two source rows together:
//            12 independent vector loads per thread are in flight instead of a ptr -> col -> row chain per row
//   phase 3  rows with more in-edges finish in a per-item loop; rows above kHeavyDegree (barrier nodes: one
//            in-edge per qubit) are left to phase 4
//   phase 4  the whole workgroup splits each heavy row's edges and reduces through LDS
// Blocks are remapped so that every XCD walks one contiguous slice of the rows: op nodes are numbered in
// program order and an edge joins an op to the next op on the same wire, so the source rows of a tile lie within
// a few hundred rows of it and are served from that XCD's L2 after the first touch.
#include <cstdlib>

#include "common.hpp"

namespace mlqem {

struct AggArgs {
  const float* x; int64_t ldx;
  const int32_t* ptr; const int32_t* idx;
  const int32_t* ell;   // optional [N,2]: the first two col[] entries of every row (see csr_aggregate_ell_kernel)
  const float* cscale; const float* rscale; const float* dself;
  float alpha, beta;
  const float* z; int64_t ldz;
  const float* bias;
  int act; float drop_p; uint64_t seed; const uint64_t* seed_counter;
  float* out; int64_t ldo;
  int64_t N; int C; int CV; int R; int nt;
};

// The pooled means of the activation a launch produces, from the same launch (mlqem_csr_aggregate_pool_f32): per-(tile, graph)
// partial sums of out and of wts * out in pool.hip's layout, tile = the workgroup's whole rows.
struct PoolFuse {
  const float* wts;        // optional [N]
  const int32_t* gptr;     // [B+1]
  int B;
  float* partial;          // [(tiles + B)][2][CV * VEC]
  const int2* tile_graph;  // [tiles]: (graph of the tile's first row, first row of the NEXT graph) (tile_graph_kernel)
  unsigned long long* mask;  // optional [tiles][kPoolMaskWords]: the sign bits of the tile's items as per-wave BALLOTS -- word
                             // ((k 4 + wave) 4 + v) of a tile, bit `lane`: out[item k 256 + 64 wave + lane][channel v of its slice] > 0
                             // (items = (row, 16-byte slice) pairs of the tile, row-major).  What the pooled activation's only
                             // reader in the backward needs of it (the ReLU / dropout gate); with it the caller may pass
                             // out == NULL and the activation never reaches memory.  r03 stored a byte per item from every lane
                             // (24 us of a 381 us launch); the ballots go through LDS and leave as ONE 256-byte store per tile
  int4* tile_info;           // with mask: [tiles] (graph of the tile's first row, that graph's first row, the next graph's, 0) for
                             // the backward pass that reads the ballots (pool.hip pool_bwd_tiles_kernel)
  unsigned* node_gate;       // optional (CV <= 4), [ceil(N / 2)] words: the same sign bits PER NODE, 16 bits each -- bit 4 slice + v of
                             // node i = out[i, 4 slice + v] > 0.  A gather can read a SOURCE's gate (2 bytes) where the ballots only
                             // serve a launch that walks the tiles themselves (pooled_grad.hip)
};

// graph of the first row of every tile of `rows` rows: one binary search per tile, outside the kernel that needs it (inside,
// its ten dependent scalar loads sat in front of every later scalar or LDS wait of the workgroup)
// (a 16-lane group per GRAPH files the tiles whose first row lies in it: two loads and a handful of stores -- a thread per tile searched
// the boundaries, ten dependent loads each: 24 us on every branch's stream for the 66 k tiles of the headline batch)
__global__ __launch_bounds__(kBlock) void tile_graph_kernel(const int32_t* __restrict__ gptr, int B, int rows, int64_t tiles,
                                                            int2* __restrict__ out, int4* __restrict__ info) {
  const int64_t g = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / kGroup;
  const int l = threadIdx.x & (kGroup - 1);
  if (g >= B) return;
  const int64_t beg = gptr[g], end = gptr[g + 1];
  const int64_t t0 = (beg + rows - 1) / rows, t1 = min(tiles, (end + rows - 1) / rows);      // tiles t with beg <= t rows < end
  for (int64_t t = t0 + l; t < t1; t += kGroup) {
    out[t] = make_int2((int)g, (int)end);   // with the boundary behind it: a tile inside one graph needs nothing else at its end
    if (info) info[t] = make_int4((int)g, (int)beg, (int)end, 0);
  }
}

constexpr int kPoolMaskWords = 2 * 4 * 4;     // items per thread x waves x channels of a slice (the pooled instantiation: 2, 4, 4)
constexpr int kRowsMax = 512;        // rows per block (LDS slices of ptr / rscale / dself)
constexpr int kHeavyDegree = 32;     // rows above this are reduced by the whole block
constexpr int kHeavyCap = 64;

// The bias vector is staged in LDS once per workgroup (zero beyond C): read from global memory inside finish_row it was
// one more dependent memory round trip at the very end of every item (+110 us on the 11 M-row GCN layers), and held in
// registers per item it pushed the 64-VGPR kernels into scratch.
constexpr int kBiasLds = kBlock * 4;   // CV <= kBlock slices of up to 4 channels

__device__ __forceinline__ void stage_bias(const AggArgs& a, float* s_bias, int cols) {
  if (!a.bias) return;                  // workgroup-uniform
  for (int i = threadIdx.x; i < cols; i += kBlock) s_bias[i] = i < a.C ? a.bias[i] : 0.f;
  __syncthreads();
}

// EPI = false: the plain form out = alpha * (rs * acc + ds * self) -- every backward aggregation and the roofline launch.
// Compiled apart from the full epilogue (z, bias, ReLU, dropout) because the kernels are pinned at 64 VGPRs for eight waves
// per SIMD and the epilogue's operands (about a dozen more SGPRs, 64-bit hash temporaries) tipped the hot path into
// scratch: +20 % write traffic and +15 % time on the plain launches, measured with the PMC passes of round 2.
template <int VEC, bool IS_MAX, bool EPI = true, bool POOL = false>
__device__ __forceinline__ unsigned finish_row(const AggArgs& a, int64_t row, int ch, const float (&acc)[VEC],
                                           const float (&self)[VEC], float rs, float ds, const float* s_bias,
                                           float* s_tile = nullptr, int64_t r0 = 0, const float* zpre = nullptr,
                                           unsigned long long* s_mask = nullptr, int item = 0, unsigned* s_gate = nullptr) {
  float res[VEC];
  if (IS_MAX) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) res[v] = acc[v];
  } else if (!EPI) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) res[v] = a.alpha * fmaf(ds, self[v], rs * acc[v]);
  } else {
    float zz[VEC];
    if (zpre) {                       // fetched with the row's other operands (the ELL kernel's items)
#pragma unroll
      for (int v = 0; v < VEC; ++v) zz[v] = zpre[v];
    } else if (a.z) vload<VEC>(a.z + row * a.ldz + ch, zz);
    bool keep[VEC];
    if (a.drop_p > 0.f)
      dropout_keep<VEC>(a.seed + (a.seed_counter ? *a.seed_counter * 0xD1B54A32D192ED03ull : 0ull), (uint64_t)(row * a.C + ch),
                        a.drop_p, keep);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float r = a.alpha * fmaf(ds, self[v], rs * acc[v]);
      if (a.z) r = fmaf(a.beta, zz[v], r);
      if (a.bias) r += s_bias[ch + v];
      if (a.act & 1) r = fmaxf(r, 0.f);
      if (a.drop_p > 0.f) r = keep[v] ? r * (1.f / (1.f - a.drop_p)) : 0.f;
      res[v] = r;
    }
  }
  if (!POOL || a.out) {
    if (a.nt) vstore_nt<VEC>(a.out + row * a.ldo + ch, res);
    else vstore<VEC>(a.out + row * a.ldo + ch, res);
  }
  unsigned signs = 0;       // POOL: returned -- a lane finishing its OWN item keeps them in a register and the kernel ballots them after
                            // its item loop (ballots in here kept the compiler from interleaving the items: 384 -> 449 us; handed
                            // back through a pointer to an array element they went through scratch memory: 413 us)
  if (POOL) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) signs |= (res[v] > 0.f ? 1u : 0u) << v;
  }
  if (POOL && s_mask) {
    // a hub row's items are finished by the lanes of the wave that walked the row: item `item` of the tile (k 256 + 64 wave +
    // lane) is bit `lane` of word (k 4 + wave) VEC + v -- OR-ed in bit by bit (rare), into the words the ballots are OR-ed into
    unsigned long long* w = s_mask + ((item >> 8) * 4 + ((item >> 6) & 3)) * VEC;
#pragma unroll
    for (int v = 0; v < VEC; ++v) if (res[v] > 0.f) atomicOr(w + v, 1ull << (item & 63));
    if (s_gate && signs) atomicOr(s_gate + ((int)(row - r0) >> 1), signs << ((((int)(row - r0) & 1) << 4) + ch));      // ch = 4 slice
  }
  if (POOL) {               // the workgroup's tile of the output, row-major, for the pooled partial sums at the end of the kernel
    float* t = s_tile + (int)(row - r0) * (a.CV * VEC) + ch;
#pragma unroll
    for (int v = 0; v < VEC; ++v) t[v] = res[v];
  }
  return signs;
}

template <int VEC, bool IS_MAX, int kItemsPerThread>
__global__ __launch_bounds__(kBlock) void csr_aggregate_kernel(const AggArgs a) {
  __shared__ int s_ptr[kRowsMax + 1];
  __shared__ float s_rs[kRowsMax];
  __shared__ float s_ds[kRowsMax];
  __shared__ int s_heavy[kHeavyCap];
  __shared__ int s_nheavy;
  __shared__ float s_red[kBlock * VEC];
  __shared__ float s_bias[kBiasLds];
  stage_bias(a, s_bias, a.CV * VEC);

  const unsigned blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int64_t r0 = (int64_t)blk * a.R;
  const int nrows = (int)min((int64_t)a.R, a.N - r0);
  const int tid = threadIdx.x;
  const bool use_self = IS_MAX || a.dself != nullptr;

  // ---- phase 1
  for (int i = tid; i <= nrows; i += kBlock) s_ptr[i] = a.ptr[r0 + i];
  for (int i = tid; i < nrows; i += kBlock) {
    s_rs[i] = a.rscale ? a.rscale[r0 + i] : 1.f;
    s_ds[i] = a.dself ? a.dself[r0 + i] : 0.f;
  }
  if (tid == 0) s_nheavy = 0;
  __syncthreads();
  const int e0 = 0;  // s_ptr holds absolute positions in col[]
  auto edge_src = [&](int e) { return a.idx[e]; };
  auto edge_w = [&](int, int j) { return (IS_MAX || !a.cscale) ? 1.f : a.cscale[j]; };

  // ---- phase 2: first two edges + self row of every item, all loads issued together
  const int n_items = nrows * a.CV;
  const unsigned magic = ((1u << 20) + a.CV - 1) / a.CV;
  int rl[kItemsPerThread], ch[kItemsPerThread], beg[kItemsPerThread], deg[kItemsPerThread];
  float acc[kItemsPerThread][VEC], self[kItemsPerThread][VEC];
  float v0[kItemsPerThread][VEC], v1[kItemsPerThread][VEC], w0[kItemsPerThread], w1[kItemsPerThread];
  int j0[kItemsPerThread], j1[kItemsPerThread];
#pragma unroll
  for (int k = 0; k < kItemsPerThread; ++k) {
    const int t = k * kBlock + tid;
    const bool live = t < n_items;
    rl[k] = live ? (int)(((unsigned)t * magic) >> 20) : 0;  // t / CV, exact for t * CV < 2^20
    ch[k] = live ? (t - rl[k] * a.CV) * VEC : 0;
    beg[k] = s_ptr[rl[k]];
    deg[k] = live ? s_ptr[rl[k] + 1] - beg[k] : -1;
    // a missing edge reads the row itself (a valid address) with weight 0 / as a no-op for max
    const bool fast = deg[k] <= kHeavyDegree;
    const int row = (int)(r0 + rl[k]);
    j0[k] = (fast && deg[k] > 0) ? a.idx[beg[k]] : row;
    j1[k] = (fast && deg[k] > 1) ? a.idx[beg[k] + 1] : row;
  }
#pragma unroll
  for (int k = 0; k < kItemsPerThread; ++k) {
    const bool fast = deg[k] <= kHeavyDegree;
    const float* __restrict__ xc = a.x + ch[k];
    w0[k] = (fast && deg[k] > 0) ? edge_w(0, j0[k]) : 0.f;
    w1[k] = (fast && deg[k] > 1) ? edge_w(0, j1[k]) : 0.f;
    vload<VEC>(xc + (int64_t)j0[k] * a.ldx, v0[k]);
    vload<VEC>(xc + (int64_t)j1[k] * a.ldx, v1[k]);
    if (use_self) vload<VEC>(xc + (r0 + rl[k]) * a.ldx, self[k]);
  }
#pragma unroll
  for (int k = 0; k < kItemsPerThread; ++k) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      if (!use_self) self[k][v] = 0.f;
      if (IS_MAX) {
        float m = self[k][v];
        if (deg[k] > 0) m = fmaxf(m, v0[k][v]);
        if (deg[k] > 1) m = fmaxf(m, v1[k][v]);
        acc[k][v] = m;
      } else {
        acc[k][v] = fmaf(w1[k], v1[k][v], w0[k] * v0[k][v]);
      }
    }
  }
  // ---- phase 3: remaining edges of medium rows, then the epilogue
#pragma unroll
  for (int k = 0; k < kItemsPerThread; ++k) {
    if (deg[k] < 0) continue;
    const int64_t row = r0 + rl[k];
    if (deg[k] > kHeavyDegree) {
      if (ch[k] == 0) {
        const int slot = atomicAdd(&s_nheavy, 1);
        if (slot < kHeavyCap) s_heavy[slot] = rl[k];
      }
      continue;
    }
    const float* __restrict__ xc = a.x + ch[k];
    for (int e = beg[k] + 2; e < beg[k] + deg[k]; ++e) {
      const int j = edge_src(e);
      const float w = edge_w(e, j);
      float r[VEC];
      vload<VEC>(xc + (int64_t)j * a.ldx, r);
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[k][v] = IS_MAX ? fmaxf(acc[k][v], r[v]) : fmaf(w, r[v], acc[k][v]);
    }
    finish_row<VEC, IS_MAX>(a, row, ch[k], acc[k], self[k], s_rs[rl[k]], s_ds[rl[k]], s_bias);
  }
  // ---- phase 4: heavy rows, one at a time, edges split over kBlock / CV slots
  __syncthreads();
  const int n_heavy = s_nheavy;
  if (n_heavy == 0) return;
  const int slots = kBlock / a.CV;  // >= 1 because CV <= kBlock is checked on the host
  const int slot = tid / a.CV, hch = (tid - slot * a.CV) * VEC;
  auto reduce_heavy_row = [&](int r) {
    __syncthreads();  // s_red is reused from the previous heavy row
    float part[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) part[v] = IS_MAX ? -INFINITY : 0.f;
    if (slot < slots) {
      for (int e = s_ptr[r] - e0 + slot; e < s_ptr[r + 1] - e0; e += slots) {
        const int j = edge_src(e);
        const float w = edge_w(e, j);
        float q[VEC];
        vload<VEC>(a.x + hch + (int64_t)j * a.ldx, q);
#pragma unroll
        for (int v = 0; v < VEC; ++v) part[v] = IS_MAX ? fmaxf(part[v], q[v]) : fmaf(w, q[v], part[v]);
      }
#pragma unroll
      for (int v = 0; v < VEC; ++v) s_red[tid * VEC + v] = part[v];
    }
    __syncthreads();
    if (slot == 0) {  // the first CV threads own one channel slice each and add the slots in a fixed order
      float tot[VEC], sf[VEC];
      const int64_t row = r0 + r;
      if (use_self) vload<VEC>(a.x + hch + row * a.ldx, sf);
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        if (!use_self) sf[v] = 0.f;
        tot[v] = IS_MAX ? sf[v] : 0.f;
      }
      for (int sl = 0; sl < slots; ++sl)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const float q = s_red[(sl * a.CV + tid) * VEC + v];
          tot[v] = IS_MAX ? fmaxf(tot[v], q) : tot[v] + q;
        }
      finish_row<VEC, IS_MAX>(a, row, hch, tot, sf, s_rs[r], s_ds[r], s_bias);
    }
  };
  if (n_heavy <= kHeavyCap) {
    for (int h = 0; h < n_heavy; ++h) reduce_heavy_row(s_heavy[h]);
  } else {  // more heavy rows than the list holds: find them again by scanning the block's rows
    for (int r = 0; r < nrows; ++r)
      if (s_ptr[r + 1] - s_ptr[r] > kHeavyDegree) reduce_heavy_row(r);
  }
}

// ELL-assisted variant.  `ell[row] = (s0, s1)` repeats the first two col[] entries of the row (-1 = no such
// edge; bit 31 of s0 set = the row has more than two edges, finish it from the CSR arrays).  On circuit graphs
// 99.8 % of the rows are complete after (s0, s1), so the row pointer -> col -> source row chain of the CSR walk
// becomes ell -> source row: one dependent round trip less, no LDS staging and no barrier on the fast path.
// Items are (row, channel-slice) pairs numbered row-major exactly as above; a workgroup owns kBlock * IPT items.

template <int VEC, bool IS_MAX, int kItemsPerThread, bool EPI, int kMinWaves = 8, bool POOL = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(kMinWaves, 8))) void csr_aggregate_ell_kernel(const AggArgs a, const PoolFuse pf) {
  // A block owns a.R = (kBlock * IPT) / CV whole rows; the local item index li < kBlock * IPT <= 2048 is split into
  // (row, slice) with a multiply-shift (exact for li * CV < 2^20) instead of a division.
  __shared__ float s_bias[EPI ? kBiasLds : 1];
  __shared__ float s_tile[POOL ? kBlock * kItemsPerThread * VEC : 1];       // POOL: this workgroup's rows of the output
  __shared__ float s_wts[POOL ? kBlock * kItemsPerThread : 1];               // ... the rows' pooling weights
  __shared__ int s_done;                                                     // ... tickets of the waves that are done
  __shared__ unsigned long long s_mask[POOL ? kPoolMaskWords : 1];           // ... the sign bits of its items (PoolFuse::mask)
  __shared__ unsigned s_gate[POOL ? kBlock : 1];                             // ... and per node (PoolFuse::node_gate): two rows a word
  static_assert(!POOL || (kItemsPerThread == 2 && VEC == 4 && kBlock == 256), "the ballot layout of the pooled form");
  if (EPI) stage_bias(a, s_bias, a.CV * VEC);
  const unsigned blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int64_t r0 = (int64_t)blk * a.R;
  const int nrows = (int)min((int64_t)a.R, a.N - r0);
  const int n_local = nrows * a.CV;
  const unsigned magic = ((1u << 20) + a.CV - 1) / a.CV;
  const int tid = threadIdx.x;
  const bool use_self = IS_MAX || a.dself != nullptr;
  // POOL: the graph of the tile's first row, searched for now (ten dependent scalar loads) so that the latency hides behind
  // the gathers instead of standing at the end of the kernel
  // POOL: the ticket counter starts at zero (the one barrier, at the start where every wave is anyway and nothing is in
  // flight); the rows' pooling weights are fetched now and parked in LDS just before the ticket is taken
  float wreg[kItemsPerThread];
  int2 tinfo = make_int2(0, 0);
  if constexpr (POOL) {
    if (tid == 0) s_done = 0;
    if (tid < kPoolMaskWords) s_mask[tid] = 0ull;
    s_gate[tid] = 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kItemsPerThread; ++k) {
      const int r = k * kBlock + tid;
      wreg[k] = (pf.wts && r < nrows) ? pf.wts[r0 + r] : 1.f;
    }
    tinfo = pf.tile_graph[blk];                     // needed by the last wave at the very end: fetched now, not waited for then
  }

  int row[kItemsPerThread], ch[kItemsPerThread];
  int2 e2[kItemsPerThread];
  float rs[kItemsPerThread], ds[kItemsPerThread];
  bool live[kItemsPerThread];
#pragma unroll
  for (int k = 0; k < kItemsPerThread; ++k) {
    const int li = k * kBlock + tid;
    live[k] = li < n_local;
    const int lj = live[k] ? li : 0;
    const int lrow = (int)(((unsigned)lj * magic) >> 20);
    row[k] = (int)r0 + lrow;
    ch[k] = (lj - lrow * a.CV) * VEC;
    {
      e2[k] = reinterpret_cast<const int2*>(a.ell)[row[k]];
      rs[k] = a.rscale ? a.rscale[row[k]] : 1.f;
      ds[k] = a.dself ? a.dself[row[k]] : 0.f;
    }
  }
  float acc[kItemsPerThread][VEC], self[kItemsPerThread][VEC];
  float v0[kItemsPerThread][VEC], v1[kItemsPerThread][VEC], w0[kItemsPerThread], w1[kItemsPerThread];
  float zq[EPI ? kItemsPerThread : 1][VEC];
  bool more[kItemsPerThread];
#pragma unroll
  for (int k = 0; k < kItemsPerThread; ++k) {
    more[k] = (e2[k].x & kEllMore) != 0 && e2[k].x != -1;
    const int s0 = e2[k].x == -1 ? -1 : (e2[k].x & ~kEllMore), s1 = e2[k].y;
    const int j0 = s0 >= 0 ? s0 : row[k];  // a missing edge reads the row itself with weight 0 / as a no-op for max
    const int j1 = s1 >= 0 ? s1 : row[k];
    const float* __restrict__ xc = a.x + ch[k];
    // (the column scales loaded UNCONDITIONALLY -- j0 / j1 are valid rows either way -- and masked afterwards: `s0 >= 0 ? cscale[j0] : 0`
    // compiles to a branch around the load, and the path of the lanes without that edge waited there for every load in flight
    // (a register written on both paths) BEFORE the row loads below were issued: a round trip per batch of items for most waves)
    float c0 = 1.f, c1 = 1.f;
    if (!IS_MAX && a.cscale) { c0 = a.cscale[j0]; c1 = a.cscale[j1]; }      // (uniform condition)
    w0[k] = s0 >= 0 ? c0 : 0.f;
    w1[k] = s1 >= 0 ? c1 : 0.f;
    vload<VEC>(xc + (int64_t)j0 * a.ldx, v0[k]);
    vload<VEC>(xc + (int64_t)j1 * a.ldx, v1[k]);
    if (use_self) vload<VEC>(xc + (int64_t)row[k] * a.ldx, self[k]);
    if (EPI && a.z) vload<VEC>(a.z + (int64_t)row[k] * a.ldz + ch[k], zq[k]);      // with the gathers, not behind the arithmetic
  }
#pragma unroll
  for (int k = 0; k < kItemsPerThread; ++k) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      if (!use_self) self[k][v] = 0.f;
      if (IS_MAX) {
        float m = self[k][v];
        if (w0[k] != 0.f) m = fmaxf(m, v0[k][v]);
        if (w1[k] != 0.f) m = fmaxf(m, v1[k][v]);
        acc[k][v] = m;
      } else {
        acc[k][v] = fmaf(w1[k], v1[k][v], w0[k] * v0[k][v]);
      }
    }
  }
  bool heavy[kItemsPerThread];
  unsigned sign_bits[kItemsPerThread];      // POOL with PoolFuse::mask: signs of the items this lane finishes itself
#pragma unroll
  for (int k = 0; k < kItemsPerThread; ++k) { heavy[k] = false; sign_bits[k] = 0u; }
#pragma unroll
  for (int k = 0; k < kItemsPerThread; ++k) {
    if (!live[k]) continue;
    if (more[k]) {  // rare: rows with more than two in-edges walk the CSR arrays from the third edge on
      const int beg = a.ptr[row[k]], end = a.ptr[row[k] + 1];
      if (end - beg > kHeavyDegree) {   // a hub row (barrier node): left to the wave-cooperative pass below
        heavy[k] = ch[k] == 0;          // its slice-0 item speaks for the whole row; the other slices just skip it
        continue;
      }
      const float* __restrict__ xc = a.x + ch[k];
      for (int e = beg + 2; e < end; ++e) {
        const int j = a.idx[e];
        const float w = (IS_MAX || !a.cscale) ? 1.f : a.cscale[j];
        float r[VEC];
        vload<VEC>(xc + (int64_t)j * a.ldx, r);
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[k][v] = IS_MAX ? fmaxf(acc[k][v], r[v]) : fmaf(w, r[v], acc[k][v]);
      }
    }
    const unsigned sg = finish_row<VEC, IS_MAX, EPI, POOL>(a, row[k], ch[k], acc[k], self[k], rs[k], ds[k], s_bias, s_tile, r0,
                                                           (EPI && a.z) ? zq[k] : nullptr);
    if (POOL) sign_bits[k] = sg;
  }
  if constexpr (POOL) {
    if (pf.mask) {          // the items' signs as per-wave ballots into the tile's words (PoolFuse::mask), one LDS atomic per word
      const int wv = tid >> 6;
#pragma unroll
      for (int k = 0; k < kItemsPerThread; ++k)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const unsigned long long b = __ballot((sign_bits[k] >> v) & 1u);
          if ((tid & 63) == 0 && b) atomicOr(&s_mask[(k * 4 + wv) * VEC + v], b);
        }
    }
    if (pf.node_gate) {
#pragma unroll
      for (int k = 0; k < kItemsPerThread; ++k) {
        const int lr = row[k] - (int)r0;
        if (sign_bits[k]) atomicOr(&s_gate[lr >> 1], sign_bits[k] << (((lr & 1) << 4) + ch[k]));
      }
    }
  }
  // Hub rows (barrier nodes: one in-edge per qubit), one at a time, by the WAVE that owns the row's slice-0 item: its 64
  // lanes split the row's edges (a lane = one edge slot x one channel slice), then the slots are added up by a shuffle
  // tree in a fixed order.  No LDS, no workgroup barrier: the other waves of the workgroup are not held up, and nothing
  // is shared that would need initialising (47 % of the benchmark's workgroups contain such a row).
  const int lane = tid & (kWave - 1);
#pragma unroll
  for (int k = 0; k < kItemsPerThread; ++k) {
    unsigned long long todo = __ballot(heavy[k]);
    while (todo) {
      const int owner = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const int r = __shfl(row[k], owner);
      const float rs_r = __shfl(rs[k], owner), ds_r = __shfl(ds[k], owner);
      const int beg = a.ptr[r], end = a.ptr[r + 1];
      if (a.CV <= kWave) {
        const int nslots = kWave / a.CV;
        const int slot = lane / a.CV, hch = (lane - slot * a.CV) * VEC;
        const bool worker = slot < nslots;
        float part[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) part[v] = IS_MAX ? -INFINITY : 0.f;
        // 64 edges at a time: one coalesced read of their source ids, handed to the (slot, slice) lanes by shuffle (the
        // loops are wave-uniform: every lane takes part in every shuffle), so a round costs one memory round trip
        for (int base = beg; base < end; base += kWave) {
          const int my_src = base + lane < end ? a.idx[base + lane] : 0;
          const int in_chunk = min(kWave, end - base);
          const int rounds = (in_chunk + nslots - 1) / nslots;
          for (int rd = 0; rd < rounds; ++rd) {
            const int el = slot + rd * nslots;
            const int j = __shfl(my_src, el < kWave ? el : 0);
            if (worker && el < in_chunk) {
              const float w = (IS_MAX || !a.cscale) ? 1.f : a.cscale[j];
              float q[VEC];
              vload<VEC>(a.x + hch + (int64_t)j * a.ldx, q);
#pragma unroll
              for (int v = 0; v < VEC; ++v) part[v] = IS_MAX ? fmaxf(part[v], q[v]) : fmaf(w, q[v], part[v]);
            }
          }
        }
        for (int off = 32; off >= 1; off >>= 1) {      // slot s takes slot s + off: same tree for every row
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const float other = __shfl_down(part[v], off * a.CV);
            if (worker && slot + off < nslots) part[v] = IS_MAX ? fmaxf(part[v], other) : part[v] + other;
          }
        }
        if (worker && slot == 0) {
          float sf[VEC];
          if (use_self) vload<VEC>(a.x + hch + (int64_t)r * a.ldx, sf);
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            if (!use_self) sf[v] = 0.f;
            if (IS_MAX) part[v] = fmaxf(part[v], sf[v]);
          }
          finish_row<VEC, IS_MAX, EPI, POOL>(a, r, hch, part, sf, rs_r, ds_r, s_bias, s_tile, r0, nullptr, (POOL && pf.mask) ? s_mask : nullptr,
                                             (int)(r - r0) * a.CV + hch / VEC, (POOL && pf.node_gate) ? s_gate : nullptr);
        }
      } else {   // more slices than lanes: every lane walks all edges for its slices
        for (int sl = lane; sl < a.CV; sl += kWave) {
          const int hch = sl * VEC;
          float tot[VEC], sf[VEC];
          if (use_self) vload<VEC>(a.x + hch + (int64_t)r * a.ldx, sf);
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            if (!use_self) sf[v] = 0.f;
            tot[v] = IS_MAX ? sf[v] : 0.f;
          }
          for (int e = beg; e < end; ++e) {
            const int j = a.idx[e];
            const float w = (IS_MAX || !a.cscale) ? 1.f : a.cscale[j];
            float q[VEC];
            vload<VEC>(a.x + hch + (int64_t)j * a.ldx, q);
#pragma unroll
            for (int v = 0; v < VEC; ++v) tot[v] = IS_MAX ? fmaxf(tot[v], q[v]) : fmaf(w, q[v], tot[v]);
          }
          finish_row<VEC, IS_MAX, EPI, POOL>(a, r, hch, tot, sf, rs_r, ds_r, s_bias, s_tile, r0, nullptr, (POOL && pf.mask) ? s_mask : nullptr,
                                             (int)(r - r0) * a.CV + hch / VEC, (POOL && pf.node_gate) ? s_gate : nullptr);
        }
      }
    }
  }
  if constexpr (POOL) {
    // Pooled partial sums of the tile: for every graph with rows in it, the sums over its rows of out and of wts * out per
    // column.  No workgroup barrier: the waves of this kernel never wait for each other (a wave walking a hub row would hold
    // the other three), so every wave publishes its rows in LDS and takes a ticket, and the wave that draws the LAST ticket
    // reduces the whole tile alone while the others have already left: 64 / (CV VEC) row groups x (CV VEC) columns, a
    // group's rows added in order, the groups added in order (deterministic).  One partial per (tile, graph) at index
    // tile + graph, pool.hip's layout, summed over the tiles by its finish kernel.
#pragma unroll
    for (int k = 0; k < kItemsPerThread; ++k) {
      const int r = k * kBlock + tid;
      if (r < nrows) s_wts[r] = wreg[k];
    }
    __threadfence_block();                        // this wave's rows of s_tile and s_wts are visible before its ticket is
    int last = 0;
    if (lane == 0) last = atomicAdd(&s_done, 1) == kBlock / kWave - 1;
    last = __shfl(last, 0);
    if (!last) return;
    __threadfence_block();
    if (pf.mask && lane < kPoolMaskWords) pf.mask[(int64_t)blk * kPoolMaskWords + lane] = s_mask[lane];
    if (pf.node_gate)         // R is even: a tile's rows start at a word
      for (int i = lane; i < (nrows + 1) / 2; i += kWave) pf.node_gate[(r0 >> 1) + i] = s_gate[i];
    // lane = slice * groups + group: a lane adds every groups-th row of its 16-byte channel slice, then the groups of a slice
    // (consecutive lanes) are added by a fixed shuffle tree
    const int cvv = a.CV * VEC;
    const int groups = kWave / a.CV;              // CV <= 64: checked on the host
    const int sl = lane / groups, grp = lane - sl * groups;
    const bool active = sl < a.CV;
    // the graphs with rows in this tile, in order (wave-uniform loop): the first one and its end came with tinfo; a further
    // boundary is loaded only by the tiles that contain one
    int g = tinfo.x;
    int64_t gbeg = r0, gend = tinfo.y;
    for (;; ) {
      const int s0 = (int)(max(r0, gbeg) - r0), s1 = (int)(min(r0 + nrows, gend) - r0);
      const bool more = gend < r0 + nrows && g + 1 < pf.B;
      if (s1 <= s0) {                             // an empty graph
        if (!more) break;
        ++g; gbeg = gend; gend = pf.gptr[g + 1];
        continue;
      }
      float a0[VEC], a1[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) a0[v] = a1[v] = 0.f;
      if (active)
        for (int r = s0 + grp; r < s1; r += groups) {
          float xv[VEC];
          vload<VEC>(s_tile + r * cvv + sl * VEC, xv);
          const float w = s_wts[r];
#pragma unroll
          for (int v = 0; v < VEC; ++v) { a0[v] += xv[v]; a1[v] = fmaf(w, xv[v], a1[v]); }
        }
      for (int off = 32; off >= 1; off >>= 1) {
        if (off >= groups) continue;              // wave-uniform
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const float o0 = __shfl_down(a0[v], off), o1 = __shfl_down(a1[v], off);
          if (grp + off < groups) { a0[v] += o0; a1[v] += o1; }
        }
      }
      if (active && grp == 0) {
        float* dst = pf.partial + (((int64_t)blk + g) * 2) * cvv + sl * VEC;
#pragma unroll
        for (int v = 0; v < VEC; ++v) { dst[v] = a0[v]; dst[cvv + v] = a1[v]; }
      }
      if (!more) break;
      ++g; gbeg = gend; gend = pf.gptr[g + 1];
    }
  }
}

// ell[row] from the CSR arrays (used by mlqem_csr_build and after batch assembly)
__global__ __launch_bounds__(kBlock) void ell_from_csr_kernel(const int32_t* __restrict__ ptr,
                                                              const int32_t* __restrict__ idx, int64_t N,
                                                              int32_t* __restrict__ ell) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const int beg = ptr[i], deg = ptr[i + 1] - beg;
  int s0 = deg > 0 ? idx[beg] : -1;
  const int s1 = deg > 1 ? idx[beg + 1] : -1;
  if (deg > 2) s0 |= kEllMore;
  reinterpret_cast<int2*>(ell)[i] = make_int2(s0, s1);
}

template <bool IS_MAX>
static int launch_aggregate(AggArgs a, hipStream_t stream, const PoolFuse* pool = nullptr, int* rows_per_tile = nullptr) {
  const bool no_store = pool && pool->mask && !a.out;       // the pooled form with the gate bits: the activation itself is optional
  if (a.N < 0 || a.C <= 0 || !a.x || !a.ptr || (!a.out && !no_store) || a.ldx < a.C || (a.out && a.ldo < a.C)) return MLQEM_ERR_BAD_ARG;
  if (a.z && a.ldz < a.C) return MLQEM_ERR_BAD_ARG;
  if (a.N == 0) return MLQEM_OK;
  if (!a.idx) return MLQEM_ERR_BAD_ARG;
  // Widest vector the shapes and base addresses allow.  Callers lay activations out with the leading dimension
  // rounded up to a multiple of 4 floats: when every operand owns round_up(C, v) columns per row the pad columns are
  // simply processed along (columns never mix, so whatever the pads hold stays in the pads) and the kernel moves
  // 16 bytes per lane (C = 22 in 24-float rows: 147 us instead of 190 us on the 2.8M-node benchmark batch).
  int vec = 1;
  auto ok = [&](int v) {
    const int cp = (a.C + v - 1) / v * v;
    if (a.ldx < cp || a.ldx % v || !aligned_to(a.x, 4 * v)) return false;
    if (a.out && (a.ldo < cp || a.ldo % v || !aligned_to(a.out, 4 * v))) return false;
    if (a.z && (a.ldz < cp || a.ldz % v || !aligned_to(a.z, 4 * v))) return false;
    return true;
  };
  if (ok(4)) vec = 4; else if (ok(2)) vec = 2;
  a.CV = (a.C + vec - 1) / vec;
  if (a.CV > kBlock) return MLQEM_ERR_UNSUPPORTED;
  // items per thread: measured best on MI355X (C = 10 and 22, 2.8M-node batch): 4 for the CSR walk (more loads in
  // flight per thread outweigh 6 waves/SIMD), 2 for the ELL-assisted kernel (8 waves/SIMD).  MLQEM_AGG_IPT overrides.
  constexpr int nt_env = 1;      // (was the A/B switch MLQEM_AGG_NT: settled)  // streaming stores (-8 % measured)
  a.nt = nt_env;
  constexpr int ipt_env = 0;      // (was the A/B switch MLQEM_AGG_IPT: settled)
  const int ipt = ipt_env > 0 ? ipt_env : (a.ell ? 2 : 4);
  a.R = std::min(kRowsMax, kBlock * ipt / a.CV);
  const int64_t blocks = ceil_div(a.N, a.R);
  if (blocks > 0x7fffffffLL) return MLQEM_ERR_UNSUPPORTED;
  dim3 grid((unsigned)blocks), block(kBlock);
  if (a.ell) {  // whole rows per block, as many as fit kBlock * ipt items
    a.R = std::max(1, kBlock * ipt / a.CV);
    const int64_t eblocks = ceil_div(a.N, a.R);
    if (eblocks > 0x7fffffffLL) return MLQEM_ERR_UNSUPPORTED;
    grid = dim3((unsigned)eblocks);
  }
  const bool epi = !IS_MAX && (a.z || a.bias || a.act || a.drop_p > 0.f);
  const PoolFuse no_pool{nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (pool) {     // the pooled form exists for the shape the models launch it with: ELL side table, 16-byte rows, an epilogue
    if (IS_MAX || !a.ell || vec != 4 || ipt != 2 || a.CV * 4 > kWave) return MLQEM_ERR_UNSUPPORTED;      // C <= 64: one wave holds a row of the tile
    if (rows_per_tile) *rows_per_tile = a.R;
    hipLaunchKernelGGL(tile_graph_kernel, dim3((unsigned)ceil_div((int64_t)std::max(pool->B, 1) * kGroup, kBlock)), dim3(kBlock), 0, stream, pool->gptr, pool->B,
                       a.R, (int64_t)grid.x, const_cast<int2*>(pool->tile_graph), pool->mask ? pool->tile_info : nullptr);
    if constexpr (!IS_MAX) {
      constexpr int pool_waves = 7;      // (was the A/B switch MLQEM_AGG_POOL_WAVES: settled)
      if (pool_waves >= 7) hipLaunchKernelGGL((csr_aggregate_ell_kernel<4, false, 2, true, 7, true>), grid, block, 0, stream, a, *pool);
      else hipLaunchKernelGGL((csr_aggregate_ell_kernel<4, false, 2, true, 6, true>), grid, block, 0, stream, a, *pool);
    }
    return launch_status();
  }
  constexpr int epi_waves = 7;      // (was the A/B switch MLQEM_AGG_EPI_WAVES: settled)   // measured: 7 -> 304/351 us, 6 -> 341/385, 8 (spilling) -> 356/385 (GCN / Cheb forward)
#define MLQEM_LAUNCH(V, P)                                                                                  \
  do {                                                                                                      \
    if (a.ell && epi && epi_waves == 8) hipLaunchKernelGGL((csr_aggregate_ell_kernel<V, IS_MAX, P, true, 8>), grid, block, 0, stream, a, no_pool);   \
    else if (a.ell && epi && epi_waves == 7) hipLaunchKernelGGL((csr_aggregate_ell_kernel<V, IS_MAX, P, true, 7>), grid, block, 0, stream, a, no_pool);   \
    else if (a.ell && epi) hipLaunchKernelGGL((csr_aggregate_ell_kernel<V, IS_MAX, P, true, 6>), grid, block, 0, stream, a, no_pool);   \
    else if (a.ell) hipLaunchKernelGGL((csr_aggregate_ell_kernel<V, IS_MAX, P, false>), grid, block, 0, stream, a, no_pool);    \
    else hipLaunchKernelGGL((csr_aggregate_kernel<V, IS_MAX, P>), grid, block, 0, stream, a);               \
  } while (0)
#define MLQEM_BY_IPT(V) do { if (ipt == 1) MLQEM_LAUNCH(V, 1); else if (ipt == 2) MLQEM_LAUNCH(V, 2); else if (ipt == 8) MLQEM_LAUNCH(V, 8); else MLQEM_LAUNCH(V, 4); } while (0)
  switch (vec) {
    case 4: MLQEM_BY_IPT(4); break;
    case 2: MLQEM_BY_IPT(2); break;
    default: MLQEM_BY_IPT(1); break;
  }
#undef MLQEM_BY_IPT
#undef MLQEM_LAUNCH
  return launch_status();
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void relu_dropout_bwd_kernel(const float* __restrict__ g, int64_t ldg,
                                                                  const float* __restrict__ y, int64_t ldy, float scale,
                                                                  float* __restrict__ gx, int64_t ldgx, int64_t N,
                                                                  int CV) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= N * CV) return;
  const int64_t r = t / CV;
  const int c = (int)(t - r * CV) * VEC;
  float gv[VEC], yv[VEC], o[VEC];
  vload<VEC>(g + r * ldg + c, gv);
  vload<VEC>(y + r * ldy + c, yv);
#pragma unroll
  for (int v = 0; v < VEC; ++v) o[v] = yv[v] > 0.f ? gv[v] * scale : 0.f;
  vstore_nt<VEC>(gx + r * ldgx + c, o);
}

// y = dropout(relu(x)) and, with a residual, sum = y + residual in the same pass (the trunk of MLP2 / MLP3:
// x1 + drop(relu(bn2(...))), docs/tutorials/mlp.py:60-66): one launch instead of three element-wise ones.  Masks: one hash
// per 4 consecutive elements of a row (dropout_keep), keyed by (seed + counter, row * C + col).
template <int VEC>
__global__ __launch_bounds__(kBlock) void relu_dropout_kernel(const float* __restrict__ x, int64_t ldx, float p, uint64_t seed,
                                                              const uint64_t* __restrict__ seed_counter,
                                                              const float* __restrict__ residual, int64_t ldr,
                                                              float* __restrict__ y, int64_t ldy, float* __restrict__ sum,
                                                              int64_t lds, int64_t N, int C, int CV) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= N * CV) return;
  const int64_t r = t / CV;
  const int c = (int)(t - r * CV) * VEC;
  float xv[VEC], rv[VEC], o[VEC], sv[VEC];
  vload<VEC>(x + r * ldx + c, xv);
  if (residual) vload<VEC>(residual + r * ldr + c, rv);
  bool keep[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) keep[v] = true;
  if (p > 0.f) dropout_keep<VEC>(seed + (seed_counter ? *seed_counter * 0xD1B54A32D192ED03ull : 0ull), (uint64_t)(r * C + c), p, keep);
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    const float u = fmaxf(xv[v], 0.f);
    o[v] = keep[v] ? u * (1.f / (1.f - p)) : 0.f;
    sv[v] = residual ? o[v] + rv[v] : o[v];
  }
  vstore<VEC>(y + r * ldy + c, o);
  if (sum) vstore<VEC>(sum + r * lds + c, sv);
}

int aggregate_pool_rows_per_tile(int C);

// The aggregation launch of mlqem_csr_aggregate_pool_f32 (pool.hip owns the entry point and the finish kernel): the epilogue
// form writing per-(tile, graph) pooled partial sums; *rows_per_tile = the rows a workgroup owns.
int launch_aggregate_with_pool(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* idx, const int32_t* ell, const float* cscale,
                               const float* rscale, const float* dself, float alpha, float beta, const float* z, int64_t ldz,
                               const float* bias, int act, float drop_p, uint64_t seed, const uint64_t* seed_counter, float* out,
                               int64_t ldo, int64_t N, int C, const float* pool_weights, const int32_t* graph_ptr, int B,
                               float* partial, int2* tile_graph, uint8_t* gate_bits, int* rows_per_tile, hipStream_t stream) {
  AggArgs a{x, ldx, ptr, idx, ell, cscale, rscale, dself, alpha, beta, z, ldz, bias, act, drop_p, seed, seed_counter, out, ldo, N, C, 0, 0};
  // gate_bits: [tiles] 16-byte tile records, then [tiles][kPoolMaskWords] ballot words (mlqem_csr_aggregate_pool_gate_bytes)
  const int64_t tiles = ceil_div(std::max<int64_t>(N, 1), (int64_t)aggregate_pool_rows_per_tile(C));
  int4* info = reinterpret_cast<int4*>(gate_bits);
  unsigned long long* words = reinterpret_cast<unsigned long long*>(info + tiles);
  const bool per_node = gate_bits && (C + 3) / 4 <= 4 && aggregate_pool_rows_per_tile(C) % 2 == 0;
  const PoolFuse pf{pool_weights, graph_ptr, B, partial, tile_graph, gate_bits ? words : nullptr, info,
                    per_node ? reinterpret_cast<unsigned*>(words + tiles * kPoolMaskWords) : nullptr};
  return launch_aggregate<false>(a, stream, &pf, rows_per_tile);
}

// rows a workgroup of the pooled form owns for C channels (what sizes the partial-sum workspace and the gate buffer)
int aggregate_pool_rows_per_tile(int C) { return std::max(1, kBlock * 2 / ((C + 3) / 4)); }
int aggregate_pool_mask_words() { return kPoolMaskWords; }

}  // namespace mlqem

using namespace mlqem;

extern "C" int mlqem_csr_aggregate_f32(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* idx,
                                       const int32_t* ell, const float* cscale, const float* rscale, const float* dself, float alpha,
                                       float beta, const float* z, int64_t ldz, const float* bias, int act,
                                       float drop_p, uint64_t seed, const uint64_t* seed_counter, float* out, int64_t ldo,
                                       int64_t N, int C, mlqem_stream_t stream) {
  begin_launches();
  if (drop_p < 0.f || drop_p >= 1.f) return MLQEM_ERR_BAD_ARG;
  AggArgs a{x, ldx, ptr, idx, ell, cscale, rscale, dself, alpha, beta, z, ldz, bias, act, drop_p, seed, seed_counter, out, ldo, N, C, 0, 0};
  return launch_aggregate<false>(a, as_stream(stream));
}

extern "C" int mlqem_csr_segment_max_f32(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* idx,
                                         const int32_t* ell, float* out, int64_t ldo, int64_t N, int C, mlqem_stream_t stream) {
  begin_launches();
  AggArgs a{x, ldx, ptr, idx, ell, nullptr, nullptr, nullptr, 1.f, 0.f, nullptr, 0, nullptr, 0, 0.f, 0, nullptr, out, ldo, N, C, 0, 0};
  return launch_aggregate<true>(a, as_stream(stream));
}

extern "C" int mlqem_relu_dropout_bwd_f32(const float* g, int64_t ldg, const float* y, int64_t ldy, float scale,
                                          float* gx, int64_t ldgx, int64_t N, int C, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || C <= 0 || ldg < C || ldy < C || ldgx < C) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!g || !y || !gx) return MLQEM_ERR_BAD_ARG;
  // padded rows (see mlqem_csr_aggregate_f32): 16 bytes per lane, pad columns processed along
  const int c4 = (C + 3) / 4 * 4;
  const bool wide = ldg >= c4 && ldy >= c4 && ldgx >= c4 && ldg % 4 == 0 && ldy % 4 == 0 && ldgx % 4 == 0 &&
                    aligned_to(g, 16) && aligned_to(y, 16) && aligned_to(gx, 16);
  if (wide)
    hipLaunchKernelGGL(relu_dropout_bwd_kernel<4>, dim3((unsigned)ceil_div(N * (c4 / 4), kBlock)), dim3(kBlock), 0,
                       as_stream(stream), g, ldg, y, ldy, scale, gx, ldgx, N, c4 / 4);
  else
    hipLaunchKernelGGL(relu_dropout_bwd_kernel<1>, dim3((unsigned)ceil_div(N * C, kBlock)), dim3(kBlock), 0,
                       as_stream(stream), g, ldg, y, ldy, scale, gx, ldgx, N, C);
  return launch_status();
}

extern "C" int mlqem_ell_from_csr(const int32_t* ptr, const int32_t* idx, int64_t N, int32_t* ell,
                                  mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || (N > 0 && (!ptr || !ell))) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  hipLaunchKernelGGL(ell_from_csr_kernel, dim3((unsigned)ceil_div(N, kBlock)), dim3(kBlock), 0, as_stream(stream), ptr,
                     idx, N, ell);
  return launch_status();
}

extern "C" int mlqem_relu_dropout_f32(const float* x, int64_t ldx, float drop_p, uint64_t seed, const uint64_t* seed_counter,
                                      const float* residual, int64_t ldr, float* y, int64_t ldy, float* sum, int64_t lds,
                                      int64_t N, int C, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || C <= 0 || ldx < C || ldy < C || drop_p < 0.f || drop_p >= 1.f) return MLQEM_ERR_BAD_ARG;
  if ((residual && ldr < C) || (sum && lds < C) || (sum && !residual)) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!x || !y) return MLQEM_ERR_BAD_ARG;
  const int c4 = (C + 3) / 4 * 4;
  auto rows_ok = [&](const float* q, int64_t ld) { return !q || (ld >= c4 && ld % 4 == 0 && aligned_to(q, 16)); };
  const bool wide = rows_ok(x, ldx) && rows_ok(y, ldy) && rows_ok(residual, ldr) && rows_ok(sum, lds);
  if (wide)
    hipLaunchKernelGGL(relu_dropout_kernel<4>, dim3((unsigned)ceil_div(N * (c4 / 4), kBlock)), dim3(kBlock), 0, as_stream(stream), x,
                       ldx, drop_p, seed, seed_counter, residual, ldr, y, ldy, sum, lds, N, C, c4 / 4);
  else
    hipLaunchKernelGGL(relu_dropout_kernel<1>, dim3((unsigned)ceil_div(N * C, kBlock)), dim3(kBlock), 0, as_stream(stream), x, ldx,
                       drop_p, seed, seed_counter, residual, ldr, y, ldy, sum, lds, N, C, C);
  return launch_status();
}
