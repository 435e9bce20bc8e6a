// Tile plans for the LDS-staged row walks (tile_common.hpp): which rows form a tile, the tile's source union, and every entry's
// slot in it.  Structural work, done once per structure and direction (the in-CSR serves the forward and the destination-side
// backward passes, the out-CSR the source-side ones) and reused by every pass of a train step.
//
// Replaces nothing of the reference by itself: it is the index the tiled forms of TransformerConv's and ASAPooling's edge walks
// (docs/tutorials/gnn.py:80-92,104-112) read instead of `edge_index`.
#include <mutex>

#include "tile_common.hpp"

namespace mlqem {

bool ensure_dynamic_lds_impl(const void* kernel, size_t bytes) {
  constexpr int kMaxDev = 64, kMaxKernels = 64;
  struct Entry { const void* f; size_t have[kMaxDev]; };
  static Entry table[kMaxKernels] = {};
  static int used = 0;
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return false;
  std::lock_guard<std::mutex> lock(mu);
  Entry* e = nullptr;
  for (int i = 0; i < used; ++i)
    if (table[i].f == kernel) { e = &table[i]; break; }
  if (!e) {
    if (used == kMaxKernels) return false;
    e = &table[used++];
    e->f = kernel;
  }
  if (e->have[dev] >= bytes || bytes <= 48 * 1024) return true;      // (the default limit serves small requests)
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  e->have[dev] = bytes;
  return true;
}

// exclusive prefix of one int per thread over a 256-thread workgroup; `tot` receives the total.  `tmp`: 8 ints of LDS.
__device__ __forceinline__ int block_exclusive_scan(int v, int* tmp, int& tot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  __syncthreads();                       // tmp may still be read from an earlier call
  if (lane == 63) tmp[wave] = inc;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (w < wave) base += tmp[w];
  }
  tot = tmp[0] + tmp[1] + tmp[2] + tmp[3];
  return base + inc - v;
}

// order[new_ptr[g] + r] = the cluster whose centre is the r-th kept node of graph g in NODE order (= program order: the encoder
// numbers a circuit's operations in instruction order, blackwater/data/utils.py:198-389).  slot[v] = cluster id of a kept centre v,
// -1 elsewhere (mlqem_asap_slot_map).  One workgroup per graph.
__global__ __launch_bounds__(kBlock) void tile_order_kernel(const int32_t* __restrict__ slot, const int32_t* __restrict__ gptr,
                                                            const int32_t* __restrict__ new_gptr, int32_t* __restrict__ order) {
  __shared__ int tmp[8];
  constexpr int kPer = 8;                                   // nodes per thread and trip: neighbours, so that the scan keeps their order
  const int g = blockIdx.x;
  const int n0 = gptr[g], n1 = gptr[g + 1];
  const int base = new_gptr[g], lim = new_gptr[g + 1];
  int run = 0;
  for (int v0 = n0; v0 < n1; v0 += kBlock * kPer) {
    int s[kPer], mine = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int v = v0 + (int)threadIdx.x * kPer + k;
      s[k] = v < n1 ? slot[v] : -1;
      mine += s[k] >= 0 ? 1 : 0;
    }
    int tot;
    int pos = base + run + block_exclusive_scan(mine, tmp, tot);
#pragma unroll
    for (int k = 0; k < kPer; ++k)
      if (s[k] >= 0) {
        if (pos < lim) order[pos] = s[k];
        ++pos;
      }
    run += tot;
  }
}

// One workgroup per tile: the rows at positions [t T, (t + 1) T) of `order`.  LDS: bits[max_words] | pre[max_words] | six tables of
// kTileMaxRows ints | tmp[16].  The union's bitset covers the ids [lo, lo + span) the tile's entries fall into (span clamped to the
// table: ids beyond it get no slot).
__global__ __launch_bounds__(kBlock) void tile_plan_kernel(const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                           const int32_t* __restrict__ order, int N, int T, int cap, int max_words,
                                                           int4* __restrict__ rinfo, int4* __restrict__ tinfo, int32_t* __restrict__ uni,
                                                           uint16_t* __restrict__ loc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* bits = reinterpret_cast<uint32_t*>(smem);
  uint32_t* pre = bits + max_words;
  int* rbeg = reinterpret_cast<int*>(pre + max_words);
  int* rdeg = rbeg + kTileMaxRows;
  int* rid = rdeg + kTileMaxRows;
  int* sbeg = rid + kTileMaxRows;           // the same three in tile order (long rows first)
  int* sdeg = sbeg + kTileMaxRows;
  int* srow = sdeg + kTileMaxRows;
  int* tmp = srow + kTileMaxRows;
  const int t = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int rank0 = t * T, cnt = min(T, N - rank0);
  if (tid == 0) { tmp[8] = INT32_MAX; tmp[9] = -1; }
  if (tid < cnt) {
    const int r = order ? order[rank0 + tid] : rank0 + tid;
    const int b = ptr[r];
    rid[tid] = r; rbeg[tid] = b; rdeg[tid] = ptr[r + 1] - b;
  }
  __syncthreads();
  {                                                        // the id range of the tile's entries
    int lo_t = INT32_MAX, hi_t = -1;
    for (int i = wave; i < cnt; i += 4) {
      const int b = rbeg[i], d = rdeg[i];
      for (int x = lane; x < d; x += kWave) {
        const int j = idx[b + x];
        lo_t = min(lo_t, j); hi_t = max(hi_t, j);
      }
    }
    if (hi_t >= 0) { atomicMin(&tmp[8], lo_t); atomicMax(&tmp[9], hi_t); }
  }
  __syncthreads();
  const int lo = tmp[9] >= 0 ? tmp[8] : 0;
  const int span = tmp[9] >= 0 ? (int)min((int64_t)tmp[9] - lo + 1, (int64_t)max_words * 32) : 0;
  const int words = (span + 31) >> 5;
  for (int w = tid; w < words; w += kBlock) bits[w] = 0u;
  __syncthreads();
  for (int i = wave; i < cnt; i += 4) {                    // mark the sources of the tile's rows
    const int b = rbeg[i], d = rdeg[i];
    for (int x = lane; x < d; x += kWave) {
      const int j = idx[b + x] - lo;
      if ((unsigned)j < (unsigned)span) atomicOr(&bits[j >> 5], 1u << (j & 31));
    }
  }
  __syncthreads();
  const int wpt = (words + kBlock - 1) / kBlock;            // words per thread, contiguous: slots ascend with the source id
  const int w0 = min(tid * wpt, words), w1 = min(w0 + wpt, words);
  int mine = 0;
  for (int w = w0; w < w1; ++w) mine += __popc(bits[w]);
  int total;
  int run = block_exclusive_scan(mine, tmp, total);
  int32_t* __restrict__ un = uni + (int64_t)t * cap;
  for (int w = w0; w < w1; ++w) {
    uint32_t b = bits[w];
    pre[w] = (uint32_t)run;
    while (b) {
      const int bit = __ffs(b) - 1;
      b &= b - 1;
      if (run < cap) un[run] = lo + w * 32 + bit;
      ++run;
    }
  }
  for (int s = total + tid; s < cap; s += kBlock) un[s] = 0;       // the kernels copy all `cap` ids before they know the count
  __syncthreads();
  for (int i = wave; i < cnt; i += 4) {                    // every entry's slot
    const int b = rbeg[i], d = rdeg[i];
    for (int x = lane; x < d; x += kWave) {
      const int j = idx[b + x] - lo;
      uint32_t s = kTileNoSlot;
      if ((unsigned)j < (unsigned)span) {
        const uint32_t r = pre[j >> 5] + (uint32_t)__popc(bits[j >> 5] & ((1u << (j & 31)) - 1u));
        if (r < (uint32_t)cap) s = r;
      }
      loc[b + x] = (uint16_t)s;
    }
  }
  // rows of at least kTileLongDeg entries first (a wave walks each of them together), the others after them; stable
  const bool is_long = tid < cnt && rdeg[tid] >= kTileLongDeg;
  const unsigned long long bal = __ballot(is_long);
  if (lane == 0 && wave < 2) tmp[4 + wave] = __popcll(bal);
  __syncthreads();
  const int nlong = tmp[4] + tmp[5];
  if (tid < cnt) {
    const int before = (wave == 1 ? tmp[4] : 0) + __popcll(bal & ((1ull << lane) - 1ull));     // long rows in front of this one
    const int pos = is_long ? before : nlong + (tid - before);
    sbeg[pos] = rbeg[tid]; sdeg[pos] = rdeg[tid]; srow[pos] = rid[tid];
  }
  __syncthreads();
  int entries;
  const int off = block_exclusive_scan(tid < cnt ? sdeg[tid] : 0, tmp, entries);
  if (tid < T) rinfo[(int64_t)t * T + tid] = tid < cnt ? make_int4(srow[tid], sbeg[tid], sdeg[tid], off) : make_int4(0, 0, 0, 0);
  if (tid == 0) tinfo[t] = make_int4(cnt, nlong, min(total, cap), entries);
}

}  // namespace mlqem

using namespace mlqem;

extern "C" int mlqem_tile_order_by_position(const int32_t* slot, const int32_t* graph_ptr, const int32_t* new_graph_ptr,
                                            int64_t num_graphs, int32_t* order, mlqem_stream_t stream) {
  begin_launches();
  if (num_graphs < 0) return MLQEM_ERR_BAD_ARG;
  if (num_graphs == 0) return MLQEM_OK;
  if (!slot || !graph_ptr || !new_graph_ptr || !order) return MLQEM_ERR_BAD_ARG;
  if (num_graphs > INT32_MAX) return MLQEM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(tile_order_kernel, dim3((unsigned)num_graphs), dim3(kBlock), 0, as_stream(stream), slot, graph_ptr, new_graph_ptr,
                     order);
  return launch_status();
}

extern "C" int64_t mlqem_tile_plan_max_span(void) { return 8192 * 32; }      // ids a tile's union bitset covers (64 KB of LDS with its prefix table)

extern "C" int mlqem_tile_plan_build(const int32_t* ptr, const int32_t* idx, const int32_t* order, int64_t num_rows, int tile_rows, int cap,
                                     int64_t max_span, int32_t* rinfo, int32_t* tinfo, int32_t* uni, uint16_t* loc, mlqem_stream_t stream) {
  begin_launches();
  if (num_rows < 0 || tile_rows <= 0 || tile_rows > kTileMaxRows || cap <= 0 || cap >= (int)kTileNoSlot || max_span < 0) return MLQEM_ERR_BAD_ARG;
  if (num_rows > INT32_MAX) return MLQEM_ERR_UNSUPPORTED;
  if (num_rows == 0) return MLQEM_OK;
  if (!ptr || !rinfo || !tinfo || !uni || !loc) return MLQEM_ERR_BAD_ARG;
  if (!aligned_to(rinfo, 16) || !aligned_to(tinfo, 16)) return MLQEM_ERR_BAD_ARG;
  const int64_t num_tiles = ceil_div(num_rows, tile_rows);
  const int max_words = (int)std::max<int64_t>(1, (std::min<int64_t>(max_span, mlqem_tile_plan_max_span()) + 31) / 32);
  const size_t lds = (size_t)max_words * 8 + (size_t)kTileMaxRows * 24 + 64;
  if (!ensure_dynamic_lds(tile_plan_kernel, lds)) return MLQEM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(tile_plan_kernel, dim3((unsigned)num_tiles), dim3(kBlock), lds, as_stream(stream), ptr, idx, order, (int)num_rows,
                     tile_rows, cap, max_words, reinterpret_cast<int4*>(rinfo), reinterpret_cast<int4*>(tinfo), uni, loc);
  return launch_status();
}
