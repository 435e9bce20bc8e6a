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

#include <cstdlib>

#include "common.hpp"

namespace mlqem {

constexpr int kTileLongDeg = 32;            // rows of at least this many entries are walked by a whole wave
constexpr uint32_t kTileNoSlot = 0xFFFFu;
constexpr int kTileMaxRows = 128;           // rows per tile
constexpr int kTileLocEntries = 4096;       // entries of a tile whose slots are staged in LDS (rows past that read `loc` from global memory)

typedef float f4a __attribute__((ext_vector_type(4)));   // a 16-byte aligned access (LDS: ds_read_b128 / ds_write_b128)

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
// The second round trip of a tile, ALL of it in flight at once: the tile's slots (`loc`), up to two operand blocks of the tile's own
// rows, and the union's rows -- every global load is issued before the first LDS store.  (Three helpers that each loaded and then
// stored were three round trips one after the other: 130 us of a 430 us kernel; and a loop over the rows with one row's `loc` loads per
// iteration before that was a round trip per row and wave.)
//   loc:   a thread takes entries q = tid, tid + 256, ... of the tile's entry list and finds each one's row by bisection over the row
//          records' offsets (a row's entries are contiguous in `loc`, the tile's rows are not);
//   own:   `pieces` 16-byte pieces from row rinfo[i].x of `src` (+ off floats) to own[i * pitch + at ..]; at most 4 pieces per thread
//          and block (tile_rows * pieces <= 1024);
//   union: `pieces` 16-byte pieces per slot from row uid[slot] of `src`; a slot belongs to a power-of-two group of lanes (no division
//          per piece), eight loads per thread in flight; unions past 8 * 256 / group slots take further rounds.
struct TileOwn { const float* src; int64_t ld; int off, pieces, at; };
__device__ __forceinline__ void tile_stage(const TilePlan& p, const int4& ti, const TileLds& l, const float* __restrict__ usrc, int64_t uld,
                                           int uoff, int upieces, float* __restrict__ rows, int upitch, const TileOwn& o0, const TileOwn& o1,
                                           float* __restrict__ own, int opitch) {
  const int cnt = ti.x, ucnt = ti.z, tid = threadIdx.x;
  // own-row operands
  f4a ov[2][4];
  const TileOwn* os[2] = {&o0, &o1};
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const TileOwn& o = *os[b];
    if (o.src) {
      const int total = cnt * o.pieces;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = tid + u * kBlock;
        if (i < total) {
          const int r = i / o.pieces, pc = i - r * o.pieces;
          ov[b][u] = *reinterpret_cast<const f4a*>(o.src + (int64_t)l.rinfo[r].x * o.ld + o.off + 4 * pc);
        }
      }
    }
  }
  // slots of the tile's entries (first round)
  const int ltotal = min(ti.w, kTileLocEntries);
  uint16_t lv[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int q = u * kBlock + tid;
    if (q < ltotal) {
      int lo = 0, hi = cnt;                                 // the last row whose offset is <= q (offsets ascend; empty rows share one)
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (l.rinfo[mid].w <= q) lo = mid; else hi = mid;
      }
      const int4 ri = l.rinfo[lo];
      lv[u] = p.loc[ri.y + (q - ri.w)];
    }
  }
  // the union's rows (first round)
  const int shift = upieces <= 8 ? 3 : upieces <= 16 ? 4 : upieces <= 32 ? 5 : 6;
  const int pc = tid & ((1 << shift) - 1), s_in = tid >> shift, per = kBlock >> shift;
  f4a uv[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int sl = u * per + s_in;
    if (pc < upieces && sl < ucnt) uv[u] = *reinterpret_cast<const f4a*>(usrc + (int64_t)l.uid[sl] * uld + uoff + 4 * pc);
  }
  // ... and only now the stores
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const TileOwn& o = *os[b];
    if (o.src) {
      const int total = cnt * o.pieces;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = tid + u * kBlock;
        if (i < total) {
          const int r = i / o.pieces, pc2 = i - r * o.pieces;
          *reinterpret_cast<f4a*>(own + r * opitch + o.at + 4 * pc2) = ov[b][u];
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int q = u * kBlock + tid;
    if (q < ltotal) l.loc[q] = lv[u];
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int sl = u * per + s_in;
    if (pc < upieces && sl < ucnt) *reinterpret_cast<f4a*>(rows + sl * upitch + 4 * pc) = uv[u];
  }
  // further rounds (large tiles: more than 2048 entries / 8 * 256 / group slots)
  for (int q0 = 8 * kBlock; q0 < ltotal; q0 += kBlock) {
    const int q = q0 + tid;
    if (q < ltotal) {
      int lo = 0, hi = cnt;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (l.rinfo[mid].w <= q) lo = mid; else hi = mid;
      }
      const int4 ri = l.rinfo[lo];
      l.loc[q] = p.loc[ri.y + (q - ri.w)];
    }
  }
  if (pc < upieces)
    for (int s0 = 8 * per; s0 < ucnt; s0 += 8 * per) {
      f4a v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int sl = s0 + u * per + s_in;
        if (sl < ucnt) v[u] = *reinterpret_cast<const f4a*>(usrc + (int64_t)l.uid[sl] * uld + uoff + 4 * pc);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int sl = s0 + u * per + s_in;
        if (sl < ucnt) *reinterpret_cast<f4a*>(rows + sl * upitch + 4 * pc) = v[u];
      }
    }
}
// the slots alone (kernels whose other staging has its own shape)
__device__ __forceinline__ void tile_stage_loc(const TilePlan& p, const int4& ti, const TileLds& l) {
  const int cnt = ti.x, total = min(ti.w, kTileLocEntries);
  for (int q0 = 0; q0 < total; q0 += 8 * kBlock) {
    uint16_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = q0 + u * kBlock + (int)threadIdx.x;
      if (q < total) {
        int lo = 0, hi = cnt;
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (l.rinfo[mid].w <= q) lo = mid; else hi = mid;
        }
        const int4 ri = l.rinfo[lo];
        v[u] = p.loc[ri.y + (q - ri.w)];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = q0 + u * kBlock + (int)threadIdx.x;
      if (q < total) l.loc[q] = v[u];
    }
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

// ---- whole-wave reductions for the rows a wave walks with ONE LANE PER ENTRY (all 64 lanes active) ----
__device__ __forceinline__ float wave_max_all(float v) {
  v = fmaxf(v, dpp_row<0xB1>(v));
  v = fmaxf(v, dpp_row<0x4E>(v));
  v = fmaxf(v, dpp_row<0x141>(v));
  v = fmaxf(v, dpp_row<0x140>(v));
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float wave_sum_all(float v) {
  v += dpp_row<0xB1>(v);
  v += dpp_row<0x4E>(v);
  v += dpp_row<0x141>(v);
  v += dpp_row<0x140>(v);
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}
// Sixteen per-lane partial sums -> their totals over the wave, one channel per lane: lane l ends with the total of channel
// wave_channel16(l).  A reduce-scatter: at every step a lane keeps half of its channels and hands the other half to its partner
// (lane ^ 1, ^ 2: DPP quad permutes; ^ 4, ^ 8, ^ 16, ^ 32: the LDS crossbar), 8 + 4 + 2 + 1 + 1 + 1 adds instead of 16 x 6.
__device__ __forceinline__ int wave_channel16(int lane) { return ((lane & 1) << 3) | ((lane & 2) << 1) | ((lane & 4) >> 1) | ((lane & 8) >> 3); }
__device__ __forceinline__ float wave_reduce16(const float (&v)[16], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
  float w[8], u[4], t[2];
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = (b0 ? v[i + 8] : v[i]) + dpp_row<0xB1>(b0 ? v[i] : v[i + 8]);
#pragma unroll
  for (int i = 0; i < 4; ++i) u[i] = (b1 ? w[i + 4] : w[i]) + dpp_row<0x4E>(b1 ? w[i] : w[i + 4]);
#pragma unroll
  for (int i = 0; i < 2; ++i) t[i] = (b2 ? u[i + 2] : u[i]) + __shfl_xor(b2 ? u[i] : u[i + 2], 4, 64);
  float r = (b3 ? t[1] : t[0]) + __shfl_xor(b3 ? t[0] : t[1], 8, 64);
  r += __shfl_xor(r, 16, 64);
  return r + __shfl_xor(r, 32, 64);
}

}  // namespace mlqem
