// Dense blocks of a CSR structure (round 5).  In the coarsened graph ASAPooling makes of a 100-qubit circuit 17-20 % of the rows hold
// 92 % of the entries (rows of 32-200 entries around the circuit's barriers), and long rows that are neighbours in program order have
// nearly the SAME sources: 16 such rows touch ~150-250 distinct source rows for ~2 400 entries.  A per-edge walk pays ~40 vector
// instructions per (entry, head) -- address arithmetic, a gather, a quad reduction per dot product -- and is bound by instruction
// issue (profiles/r05_level1_pmc.json).  A DENSE BLOCK is 16 such rows x the union U of their sources, with a bit per cell saying
// whether the cell is an entry of the structure; TransformerConv's edge softmax over it (docs/tutorials/gnn.py:80-91) is the masked
// attention of 16 queries over |U| keys, and its dot products run on the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, a
// k-ordered fmaf chain):
//     S^T[u, row] = K_U Q^T           a key segment is loaded once per block, not once per entry
//     out^T[c, row] = V_U^T W         W = the cells' weights in S^T's accumulator layout = the B operand as it lies
// Rows that are not in a dense block (short rows; blocks whose union outgrew the capacity) stay with the per-edge kernels, which skip
// the rows a plan flags.
//
// A plan (mlqem_dense_plan_build, dense_block.hip), once per structure and direction:
//   counter[0]            16 x the number of blocks
//   lrows[16 b + i]       the rows of block b (program order inside a graph; -1 past a graph's last long row)
//   row_flag[row]         1 when the row is served by a dense block
//   records[b * kDbStride ..]:  {rows, union size, usable, entries} | rows[16] | slot of the row ITSELF in the union[16] |
//                               union ids[kDbCap] (ascending; padded with a valid id) | mask[16][kDbCap / 32]
//                         mask bit (i, s): source slot s is an entry of row i -- or s is the row itself and it has a self-loop
#pragma once

#include "attn_q4.hpp"
#include "common.hpp"

namespace mlqem {

typedef float f4a __attribute__((ext_vector_type(4)));   // a 16-byte aligned access (LDS: ds_read_b128 / ds_write_b128)

constexpr int kDbRows = 16;
constexpr int kDbCap = 512;                                  // union slots of a block (32 column blocks of 16)
constexpr int kDbMinDeg = 32;                                // rows of at least this many entries go into dense blocks
constexpr int kDbRowsOff = 4, kDbSelfOff = kDbRowsOff + kDbRows, kDbUniOff = kDbSelfOff + kDbRows, kDbMaskOff = kDbUniOff + kDbCap;
constexpr int kDbMaskWords = kDbCap / 32;                    // per row
constexpr int kDbStride = 832;                               // ints per record (3 328 bytes: a multiple of 64)
static_assert(kDbMaskOff + kDbRows * kDbMaskWords <= kDbStride, "record layout");

struct DensePlan {
  const int32_t* records;
  const int32_t* counter;
  const uint8_t* row_flag;
  int64_t max_blocks;
};

// ---- device helpers of the block kernels (dense_block.hip: attention; dense_pool.hip: ASAPooling's cluster sums)
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// D += A(16 x 16, a lane's four k-values in `a`) x B(16 x 16, in `b`).  (ONE accumulator: the counters have 31 % of the block kernels'
// wave cycles stalled on issue behind this chain of dependent MFMAs, but two accumulators and an add ran the backward 10 % slower --
// 290 -> 318 us -- and the forward no faster: registers.)
__device__ __forceinline__ f32x4 mfma16(const f4u& a, const f4u& b, f32x4 c) {
  c = mfma4(a.x, b.x, c);
  c = mfma4(a.y, b.y, c);
  c = mfma4(a.z, b.z, c);
  c = mfma4(a.w, b.w, c);
  return c;
}
// over the four lanes that hold a block row's cells (l, l ^ 16, l ^ 32, l ^ 48)
__device__ __forceinline__ float rows_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}
__device__ __forceinline__ float rows_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
// ---- what every kernel keeps in LDS: the block's union ids and its cell mask (one coalesced round trip at the block's start; the loop
// then reads ids and bits at LDS latency and only row segments from global memory), and the four waves' partial results.
// A WORKGROUP owns a block; its four waves take the column blocks w, w + 4, ... : a structure has a few thousand blocks of 4-30
// column blocks each, and with one wave per block the longest block's chain of dependent loads was the kernel's duration.
constexpr int kDbWaves = kBlock / kWave;
constexpr int kDbLdsInts = kDbCap + kDbRows * kDbMaskWords;
static_assert(kDbMaskOff == kDbUniOff + kDbCap && kDbLdsInts % 4 == 0 && kDbLdsInts <= 4 * kBlock, "one int4 per thread stages a block");
struct BlockLds { const int* uni; const uint32_t* mask; };
__device__ __forceinline__ BlockLds block_stage(const int32_t* __restrict__ rec, int* base) {
  const int k = threadIdx.x * 4;
  int4 v = make_int4(0, 0, 0, 0);
  if (k < kDbLdsInts) v = *reinterpret_cast<const int4*>(rec + kDbUniOff + k);
  __syncthreads();                                         // the previous block's reads are done
  if (k < kDbLdsInts) *reinterpret_cast<int4*>(base + k) = v;
  __syncthreads();
  return BlockLds{base, reinterpret_cast<const uint32_t*>(base + kDbCap)};
}
// the cell bits of column block cb for this lane's four cells (u = 16 cb + 4 g + i, row r)
__device__ __forceinline__ uint32_t cell_bits(const uint32_t* maskrow, int cb, int g) {
  return (maskrow[cb >> 1] >> ((cb & 1) * 16 + 4 * g)) & 0xFu;
}
constexpr float kNoMax = -1e30f;                           // "no entry yet" (finite: differences of it stay numbers)
// sum over the four waves of one accumulator tile per lane: red[w][lane] (16 bytes each); the caller syncs before and reads after
__device__ __forceinline__ f32x4 waves_sum(const f32x4* red, int lane) {
  f32x4 t = red[lane];
#pragma unroll
  for (int w = 1; w < kDbWaves; ++w) t += red[w * kWave + lane];
  return t;
}


// launches of the per-edge kernels (family_b_bwd.hip) for the rows a plan leaves to them
void launch_attn_train_q4(const AttnFwdArgs& a, hipStream_t stream);
void launch_attn_bwd_dst_q4(const AttnBwdArgs& a, hipStream_t stream);
void launch_attn_bwd_src_rc_q4(const AttnBwdArgs& a, hipStream_t stream);
// ... and of ASAPooling's per-edge kernels (attn.hip, family_b_bwd.hip); skip: the plan's row flags
int softmax_aggregate_bwd_launches(const float* x, int64_t ldx, const float* xnew, int64_t ldn, const float* gnew, int64_t ldg,
                                   const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                                   const int32_t* out_eid, const float* a_dst, const float* c_src, float negative_slope, int64_t N, int64_t E,
                                   int C, int accumulate, float* gx, int64_t ldgx, float* g_a, float* g_c, float* edge_al, float* edge_gp,
                                   const float* xmax, int64_t ldm, float* tie_count, int64_t ldt, const float* gx_rank1,
                                   const uint8_t* skip_in, const uint8_t* skip_out, int parts, mlqem_stream_t stream,
                                   const float* fuse_max_col = nullptr);
int segment_max_bwd_launches(const float* x, int64_t ldx, const float* xmax, int64_t ldm, const float* gmax, int64_t ldg, const int32_t* in_ptr,
                             const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst, int64_t N, int C, float* gx, int64_t ldgx,
                             float* gshare, int64_t lds, const float* tie_count, int64_t ldt, const float* gmax_row, const float* gmax_col,
                             const uint8_t* skip_out, mlqem_stream_t stream);
void launch_softmax_aggregate(const float* x, int64_t ldx, const int32_t* in_ptr, const int32_t* in_src, const float* a_dst, const float* c_src,
                              float negative_slope, int64_t N, int C, float* out, int64_t ldo, const uint8_t* skip, hipStream_t stream);

}  // namespace mlqem
