// Shared device/host helpers for libmlqem_hip.so (gfx950 only: 64-lane wavefronts, 8 XCDs, 256 CUs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>

#include "../../include/mlqem_hip.h"

namespace mlqem {

constexpr int kWave = 64;
constexpr int kBlock = 256;   // 4 waves: one per SIMD of a CU
constexpr int kXcds = 8;

inline hipStream_t as_stream(mlqem_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// hipGetLastError() reports the last error of ANY runtime call on this thread, including benign ones the caller
// made (hipErrorNotReady from an event query of torch's allocator).  Entry points therefore clear the slot
// first (begin_launches) and read it after their own launches (launch_status).
inline void begin_launches() { (void)hipGetLastError(); }
inline int launch_status() {
  return hipGetLastError() == hipSuccess ? MLQEM_OK : MLQEM_ERR_LAUNCH;
}

__host__ __device__ inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// A/B switch of an experiment in flight (scripts/ab_env.sh): read once per call site.  Settled switches become constants.
inline int ab_env(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

// Workgroups are dealt round-robin over the 8 XCDs (block b and b+8 share an L2).  Kernels that walk a
// row range want each XCD to own one CONTIGUOUS slice of it, so that the window of source rows a tile
// gathers from is already in that XCD's L2.  This is the bijective remap for any grid size.
__device__ __forceinline__ unsigned xcd_contiguous_block(unsigned b, unsigned nb) {
  const unsigned q = nb / kXcds, r = nb % kXcds;
  const unsigned xcd = b % kXcds, k = b / kXcds;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}

// Block number of a ROW kernel (a block owns a run of consecutive rows): XCD-contiguous, so that the blocks resident on one XCD
// walk neighbouring rows -- rows of the same graph, whose source rows are then in that XCD's 4 MB L2 (a pooled 100-qubit graph's
// key + value rows are 1.4 MB; dealt round-robin every XCD sees every graph and the gathers go out to the Infinity Cache).
// -DMLQEM_XCD_ROWS=0 compiles the plain numbering (A/B builds only).
#ifndef MLQEM_XCD_ROWS
#define MLQEM_XCD_ROWS 1
#endif
__device__ __forceinline__ unsigned row_block() {
#if MLQEM_XCD_ROWS
  return xcd_contiguous_block(blockIdx.x, gridDim.x);
#else
  return blockIdx.x;
#endif
}

// Counter-based uniform in [0,1): one splitmix64 round over (seed, element index).
__device__ __forceinline__ float uniform01(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// The same for the per-edge kernels (attention dropout draws one number per (edge, head) in EVERY lane of a 16-lane group,
// forward and backward): two rounds of a 32-bit avalanche hash, 4 integer multiplies instead of the ~14 of splitmix64's
// 64-bit products (32-bit integer multiplies run at quarter rate on CDNA).
__device__ __forceinline__ uint32_t avalanche32(uint32_t x) {
  x ^= x >> 16; x *= 0x7FEB352Du;
  x ^= x >> 15; x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float uniform01_edge(uint64_t seed, uint64_t idx) {
  uint32_t x = avalanche32((uint32_t)idx ^ (uint32_t)seed);
  x = avalanche32(x ^ (uint32_t)(seed >> 32) ^ ((uint32_t)(idx >> 32) * 0x9E3779B1u));
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}

// Keep/drop decisions for V consecutive elements from ONE splitmix64 round keyed by the index of their first element:
// element v is dropped when the v-th 16-bit field of the hash is < floor(p * 65536) (|P(drop) - p| < 1.6e-5).  The
// aggregation epilogue draws a whole 16-byte slice at once: a quarter of the 64-bit multiplies of one hash per element.
template <int V> __device__ __forceinline__ void dropout_keep(uint64_t seed, uint64_t first_idx, float p, bool (&keep)[V]) {
  static_assert(V <= 4, "one 64-bit hash carries four 16-bit uniforms");
  uint64_t z = seed + (first_idx + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  const unsigned thr = (unsigned)(p * 65536.f);
#pragma unroll
  for (int v = 0; v < V; ++v) keep[v] = (unsigned)((z >> (16 * v)) & 0xFFFFu) >= thr;
}

template <int V> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<2> { using type = float2; };
template <> struct VecT<4> { using type = float4; };

template <int V> __device__ __forceinline__ void vload(const float* p, float (&r)[V]) {
  if constexpr (V == 1) {
    r[0] = *p;
  } else if constexpr (V == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    r[0] = t.x; r[1] = t.y;
  } else {
    const float4 t = *reinterpret_cast<const float4*>(p);
    r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
  }
}
template <int V> __device__ __forceinline__ void vstore(float* p, const float (&r)[V]) {
  if constexpr (V == 1) {
    *p = r[0];
  } else if constexpr (V == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(r[0], r[1]);
  } else {
    *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
  }
}

// Streaming (non-temporal) stores: the aggregation output is written once and read by a LATER kernel, so it should not
// displace the source rows in L2 (measured -5...-9 % on the aggregation kernel).
typedef float nt_f2 __attribute__((ext_vector_type(2)));
typedef float nt_f4 __attribute__((ext_vector_type(4)));
template <int V> __device__ __forceinline__ void vstore_nt(float* p, const float (&r)[V]) {
  if constexpr (V == 1) {
    __builtin_nontemporal_store(r[0], p);
  } else if constexpr (V == 2) {
    nt_f2 t = {r[0], r[1]};
    __builtin_nontemporal_store(t, reinterpret_cast<nt_f2*>(p));
  } else {
    nt_f4 t = {r[0], r[1], r[2], r[3]};
    __builtin_nontemporal_store(t, reinterpret_cast<nt_f4*>(p));
  }
}

inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device and per kernel: remembered per (device, kernel), so that a process
// that drives several GPUs raises the limit on each of them (a `static const int once = hipFuncSetAttribute(...)` did it for the first
// device only and the launch failed on the second: ADVICE r04) and a launch path pays one table lookup.  Defined in row_order.hip.
bool ensure_dynamic_lds_impl(const void* kernel, size_t bytes);
template <class K> inline bool ensure_dynamic_lds(K kernel, size_t bytes) {
  if (bytes > 160 * 1024) return false;
  return ensure_dynamic_lds_impl(reinterpret_cast<const void*>(kernel), bytes);
}
// Sum over the 16 lanes of an aligned lane group (all 16 must be active); every lane gets the total.  The attention
// and pooling kernels give one group to a (row, head) or a row: lane l holds channels l, l + 16, ... so that a source row
// is read with 64-byte coalesced loads and a dot product over the channels is four cross-lane adds in a fixed order.
constexpr int kGroup = 16;
// The four steps are DPP row operations (a DPP row IS 16 lanes): lane i adds lane i^1, i^2, then its mirror in the half
// row and in the row -- four v_add_f32_dpp, no LDS crossbar (ds_bpermute) and nothing to wait for.  Every lane ends with
// the same value (each step adds the same two partial sums in both partners).
template <int CTRL> __device__ __forceinline__ float dpp_row(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// value of lane U of the 16-lane group, in every lane of the group (row_newbcast: one v_mov_b32_dpp)
template <int U> __device__ __forceinline__ float group16_bcast(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + U, 0xF, 0xF, true));
}
template <int U> __device__ __forceinline__ int group16_bcast(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x150 + U, 0xF, 0xF, true);
}
// lanes U0 .. U0 + 7 of the group, each broadcast to all sixteen
template <int U0, class T> __device__ __forceinline__ void group16_bcast8(T v, T (&out)[8]) {
  out[0] = group16_bcast<U0 + 0>(v); out[1] = group16_bcast<U0 + 1>(v); out[2] = group16_bcast<U0 + 2>(v);
  out[3] = group16_bcast<U0 + 3>(v); out[4] = group16_bcast<U0 + 4>(v); out[5] = group16_bcast<U0 + 5>(v);
  out[6] = group16_bcast<U0 + 6>(v); out[7] = group16_bcast<U0 + 7>(v);
}
__device__ __forceinline__ float group16_max(float v) {
  v = fmaxf(v, dpp_row<0xB1>(v));
  v = fmaxf(v, dpp_row<0x4E>(v));
  v = fmaxf(v, dpp_row<0x141>(v));
  v = fmaxf(v, dpp_row<0x140>(v));
  return v;
}
__device__ __forceinline__ float group16_sum(float v) {
  v += dpp_row<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_row<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_row<0x141>(v);   // row_half_mirror
  v += dpp_row<0x140>(v);   // row_mirror
  return v;
}

// The channel slices of a row group (lane l holds channels l, l + 16, ...; NV = 1, 2, 3, 4 slices for C <= 16 NV and 8 for 64 < C <= 128,
// as every launcher of these kernels picks it): col[v] = the lane's channel of slice v, or -- a lane past the end of a partly filled
// slice -- the row's LAST channel, so that every load of a kernel is unconditional and the value is masked by has[v] afterwards.
// `has[v] ? p[l + 16 v] : 0` compiles to a branch around the load (~100 branches per row in the pooling kernels), and a load whose only
// use sits inside a data-dependent `if` is sunk into it, where it waits for the condition's operands first (round 6: ASAPooling's
// source-side backward waited six times at the top of every row; 325 -> 234 us once its loads were unconditional and up front).
template <int NV> __device__ __forceinline__ void slice_columns(int l, int C, bool (&has)[NV], int (&col)[NV]) {
  constexpr int kFull = NV <= 4 ? NV - 1 : 4;          // slices every lane has a channel of: 16 (NV - 1) < C <= 16 NV; NV = 8: 64 < C
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    has[v] = v < kFull || l + v * kGroup < C;
    col[v] = v < kFull ? l + v * kGroup : min(l + v * kGroup, C - 1);
  }
}

// ELL side table of a CSR structure (mlqem_ell_from_csr): ell[row] = (s0, s1), the first two col[] entries of the row, -1 for
// a missing one; the sign bit of s0 says that the row has more than two entries.
constexpr int kEllMore = (int)0x80000000;

// Rows are walked in chunks of 8 / 4 / 2 / 1 edges: a chunk's index entries, then all of its source rows, are fetched
// before the first one is used, so a row of the coarsened graph of a 100-qubit circuit (100-500 edges) costs deg / 8
// dependent round trips instead of deg, and a two-edge row of a circuit DAG one instead of two.  The callback gets the
// first edge of the chunk and its size as a type (EdgeChunk<K>); it must consume the fetched values in edge order, so that
// every sum is the sum a one-edge-at-a-time loop forms, bit for bit.
template <int K> struct EdgeChunk { static constexpr int value = K; };
template <class F> __device__ __forceinline__ void for_edge_chunks(int beg, int end, F&& f) {
  int e = beg;
  for (; e + 8 <= end; e += 8) f(e, EdgeChunk<8>{});
  if (e + 4 <= end) { f(e, EdgeChunk<4>{}); e += 4; }
  if (e + 2 <= end) { f(e, EdgeChunk<2>{}); e += 2; }
  if (e < end) f(e, EdgeChunk<1>{});
}

// t = row * C + col for 0 <= col < C: a flat thread index split into (row, column).  The 64-bit division the plain expression
// compiles to is ~80 vector instructions; indices below 2^32 (every launch of this path) take the 32-bit one (~20).
__device__ __forceinline__ int64_t split_index(int64_t t, int C, int& col) {
  if (t < (1ll << 32)) {
    const uint32_t q = (uint32_t)t / (uint32_t)C;
    col = (int)((uint32_t)t - q * (uint32_t)C);
    return (int64_t)q;
  }
  const int64_t q = t / C;
  col = (int)(t - q * C);
  return q;
}

__device__ __forceinline__ bool aligned_to_dev(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// largest g in [0, B) with gptr[g] <= r (gptr[0] = 0 <= r): the graph of row r, or of the empty graphs just before it
__device__ __forceinline__ int graph_at(const int32_t* __restrict__ gptr, int B, int64_t r) {
  int lo = 0, hi = B;   // invariant: gptr[lo] <= r < gptr[hi] (gptr[B] = N > r)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int64_t)gptr[mid] <= r) lo = mid; else hi = mid;
  }
  return lo;
}

}  // namespace mlqem
