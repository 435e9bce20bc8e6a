// Tiled walks over a CSR structure whose rows share their sources (round 5): the coarsened graphs ASAPooling makes of 100-qubit
// circuits (docs/tutorials/gnn.py:85,92,104-112) are unions of dense blocks around the circuit's barriers -- 95 % of their entries
// sit in rows of 65-200 entries, and rows whose centres are close in program order share nearly all of their sources (measured:
// scripts/tile_union_probe.py; 32 consecutive rows touch 126 distinct source rows on average, 320 at most, for 1 860 entries).
// The per-edge kernels gather one 64-128-byte segment per entry through L1 / TA and are bound by that request rate (0.9-1.5 TB/s of
// HBM, 24 cache accesses per load instruction: profiles/r04_family_b_100q_pmc.json).  Here a workgroup owns a TILE of rows, loads
// the tile's source union ONCE into LDS with coalesced 16-byte loads, and every entry is then a `ds_read` at a 16-bit local slot.
//
// A plan (mlqem_tile_plan_build, tile_plan.hip) is built once per structure and direction and serves every pass over it.  Tile t
// holds the rows at positions [t T, (t + 1) T) of `order` (T = tile_rows; every row in exactly one tile):
//   tinfo[t]  = {rows, long rows, union slots, entries}   the first `long rows` of the tile have at least kTileLongDeg entries (a wave
//                                                          walks such a row together)
//   rinfo[t T + i] = {row, first CSR entry, entries, offset of the row's entries inside the tile's entry list}   (padded to whole tiles)
//   uni[t * cap + s]                                       global row id of slot s (ascending)
//   loc[e]                                                 for every CSR entry e: its slot in its row's tile, or kTileNoSlot when the union
//                                                          outgrew `cap` (the entry is then read from global memory: always correct,
//                                                          fast when rare)
// A kernel's prologue (tile_prologue) needs TWO dependent round trips: {tinfo, rinfo, uni} -- all at addresses that follow from the tile
// number -- then {the union's rows, the tile's loc entries}; the walks read nothing but LDS and the rows' own operands.  (The first
// form read `loc` from global memory chunk by chunk inside the walk: every four entries a dependent round trip at two waves per
// SIMD -- the tiled kernels took 1.3-2.6x the time of the per-edge ones.)
#pragma once

#include "common.hpp"

namespace mlqem {

constexpr int kTileLongDeg = 32;            // rows of at least this many entries are walked by a whole wave
constexpr uint32_t kTileNoSlot = 0xFFFFu;
constexpr int kTileMaxRows = 128;           // rows per tile
constexpr int kTileLocEntries = 4096;       // entries of a tile whose slots are staged in LDS (rows past that read `loc` from global memory)

struct TilePlan {
  const int4* tinfo;
  const int4* rinfo;
  const int32_t* uni;
  const uint16_t* loc;
  int64_t nt;
  int cap, tile_rows;
};

// what every tiled kernel keeps in LDS besides its staged rows
struct TileLds {
  int4* rinfo;       // [tile_rows]
  int* uid;          // [cap]
  uint16_t* loc;     // [kTileLocEntries]
};
__host__ __device__ inline size_t tile_lds_common_bytes(int cap, int tile_rows) {
  return (size_t)tile_rows * 16 + (((size_t)cap * 4 + 15) / 16) * 16 + (size_t)kTileLocEntries * 2;
}
__device__ __forceinline__ TileLds tile_lds_carve(char* base, int cap, int tile_rows) {      // base: 16-byte aligned
  TileLds l;
  l.rinfo = reinterpret_cast<int4*>(base);
  l.uid = reinterpret_cast<int*>(base + (size_t)tile_rows * 16);
  l.loc = reinterpret_cast<uint16_t*>(base + (size_t)tile_rows * 16 + (((size_t)cap * 4 + 15) / 16) * 16);
  return l;
}

// First round trip of a tile: its record, its rows' records and its union's ids (the whole `cap`: the count is not known yet).
__device__ __forceinline__ int4 tile_prologue(const TilePlan& p, int t, const TileLds& l) {
  const int4 ti = p.tinfo[t];
  const int tid = threadIdx.x;
  if (tid < p.tile_rows) l.rinfo[tid] = p.rinfo[(int64_t)t * p.tile_rows + tid];
  const int32_t* __restrict__ un = p.uni + (int64_t)t * p.cap;
  for (int s = tid; s < p.cap; s += kBlock) l.uid[s] = un[s];
  __syncthreads();
  return ti;
}
// Part of the second round trip: the tile's slots, row after row (a row's entries are contiguous in `loc`).  Call between
// tile_prologue and the barrier that ends the staging.
__device__ __forceinline__ void tile_stage_loc(const TilePlan& p, const int4& ti, const TileLds& l) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = wave; i < ti.x; i += kBlock / kWave) {
    const int4 ri = l.rinfo[i];
    if (ri.w + ri.z <= kTileLocEntries)
      for (int x = lane; x < ri.z; x += kWave) l.loc[ri.w + x] = p.loc[ri.y + x];
  }
}
// the slot of entry x of a row (ri: its record)
__device__ __forceinline__ uint32_t tile_slot(const TilePlan& p, const TileLds& l, const int4& ri, int x) {
  return ri.w + ri.z <= kTileLocEntries ? (uint32_t)l.loc[ri.w + x] : (uint32_t)p.loc[ri.y + x];
}

// memory operations of one wave in program order for its other lanes (LDS scratch written by some lanes, read by others of the SAME
// wave: the lanes run in lock step, the fences keep the compiler from moving the accesses across)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

typedef float f4a __attribute__((ext_vector_type(4)));   // a 16-byte aligned access (LDS: ds_read_b128 / ds_write_b128)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device and per kernel: remembered per (device, kernel) so that a process
// that drives several GPUs raises the limit on each of them and a launch path pays one table lookup
bool ensure_dynamic_lds_impl(const void* kernel, size_t bytes);      // tile_plan.hip
template <class K> inline bool ensure_dynamic_lds(K kernel, size_t bytes) {
  if (bytes <= 48 * 1024) return true;
  if (bytes > 160 * 1024) return false;
  return ensure_dynamic_lds_impl(reinterpret_cast<const void*>(kernel), bytes);
}

}  // namespace mlqem
