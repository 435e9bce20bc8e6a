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

#include "attn_fwd.hpp"

namespace mlqem {

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

// launches of the per-edge kernels (family_b_bwd.hip) for the rows a plan leaves to them
void launch_attn_train_q4(const AttnFwdArgs& a, hipStream_t stream);
void launch_attn_bwd_dst_q4(const AttnBwdArgs& a, hipStream_t stream);
void launch_attn_bwd_src_rc_q4(const AttnBwdArgs& a, hipStream_t stream);

}  // namespace mlqem
