// Dense blocks (dense_block.hpp): the plan of a structure's long rows and TransformerConv's edge softmax over it on the matrix cores --
// forward, destination-side backward, source-side backward (docs/tutorials/gnn.py:80-91; the formulas are attn_q4.hpp's and
// family_b_bwd.hip's, cell for cell).
//
// Wave layout of every kernel here: one wave per block; lane l has r = l & 15 and g = l >> 4.  An accumulator tile of
// v_mfma_f32_16x16x4_f32 puts cell (u = 16 cb + 4 g + i, row r) into register i of lane l (cb = the column block of 16 union slots);
// the A operand of a step is one float of the union row (l & 15) of the column block, the B operand one float of block row r.
// The k index of a product may be permuted as long as A and B agree: a lane's FOUR channels 4 g .. 4 g + 3 (one 16-byte load of a
// row segment) serve the four k-steps of a 16-channel dot product, and its four cells serve the four k-steps of a product that sums
// over the column block.
#include <type_traits>

#include "dense_block.hpp"


namespace mlqem {

// ---------------------------------------------------------------------------------------------------------------- plan
constexpr int kRowsThreads = 1024;       // dense_rows_kernel's workgroup (16 waves)
__device__ __forceinline__ int block_sum(int v, int* tmp) {     // tmp: 16 ints of LDS
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) tmp[threadIdx.x >> 6] = v;
  __syncthreads();
  int t = 0;
#pragma unroll
  for (int w = 0; w < kRowsThreads / 64; ++w) t += tmp[w];
  return t;
}
template <int WAVES>
__device__ __forceinline__ int block_scan(int v, int* tmp, int& tot) {      // exclusive, over a workgroup of WAVES waves; tmp: WAVES ints of LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  __syncthreads();
  if (lane == 63) tmp[wave] = inc;
  __syncthreads();
  int base = 0, all = 0;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) {
    const int t = tmp[w];
    if (w < wave) base += t;
    all += t;
  }
  tot = all;
  return base + inc - v;
}

// One workgroup per graph: the graph's rows of at least min_deg entries, in the order of `order` (program position; null = row
// order), as whole blocks of 16 -- a graph's last block is padded with -1 -- at a place of lrows reserved with one atomic add;
// bgraph[b] = the graph of block b.  1024 threads, six rows each: the 5.5 k rows of a 100-qubit circuit's coarsened graph are ONE trip of
// either pass, and the second pass keeps the degrees of the first (at 256 threads and four rows a trip the nine trips -- dependent loads
// order -> ptr, a scan's two barriers -- were 27 us a launch whatever the batch).
__global__ __launch_bounds__(kRowsThreads) void dense_rows_kernel(const int32_t* __restrict__ ptr, const int32_t* __restrict__ order,
                                                                  const int32_t* __restrict__ gptr, int min_deg, int32_t* __restrict__ lrows,
                                                                  int32_t* __restrict__ bgraph, int32_t* __restrict__ counter) {
  __shared__ int tmp[kRowsThreads / 64 + 1];
  constexpr int kPer = 6;
  const int g = blockIdx.x, tid = threadIdx.x;
  const int p0 = gptr[g], p1 = gptr[g + 1];
  const bool one_trip = p1 - p0 <= kRowsThreads * kPer;
  int rows[kPer];
  unsigned is_long = 0;
  auto fetch = [&](int q0) {                                 // a thread's rows are neighbours in the order: the scan keeps it
    int d0[kPer], d1[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int p = min(q0 + tid * kPer + k, p1 - 1);
      rows[k] = order ? order[p] : p;
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) { d0[k] = ptr[rows[k]]; d1[k] = ptr[rows[k] + 1]; }
    is_long = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k)
      if (q0 + tid * kPer + k < p1 && d1[k] - d0[k] >= min_deg) is_long |= 1u << k;
  };
  int mine = 0;
  if (p1 > p0)
    for (int q0 = p0; q0 < p1; q0 += kRowsThreads * kPer) {
      fetch(q0);
      mine += __popc(is_long);
    }
  const int cnt = block_sum(mine, tmp);
  const int padded = (cnt + kDbRows - 1) / kDbRows * kDbRows;
  if (tid == 0) tmp[kRowsThreads / 64] = padded ? atomicAdd(counter, padded) : 0;
  __syncthreads();
  const int start = tmp[kRowsThreads / 64];
  for (int k = tid; k < padded / kDbRows; k += kRowsThreads) bgraph[start / kDbRows + k] = g;
  int run = 0;
  if (p1 > p0)
    for (int q0 = p0; q0 < p1; q0 += kRowsThreads * kPer) {
      if (!one_trip) fetch(q0);                              // (one trip: what the counting pass fetched is still here)
      int tot;
      int pos = start + run + block_scan<kRowsThreads / 64>(__popc(is_long), tmp, tot);
#pragma unroll
      for (int k = 0; k < kPer; ++k)
        if (is_long >> k & 1u) lrows[pos++] = rows[k];
      run += tot;
    }
  for (int k = cnt + tid; k < padded; k += kRowsThreads) lrows[start + k] = -1;
}

// One workgroup per block: the union of the 16 rows' sources (and the rows themselves), every entry's slot in it, the cell mask.
// The union's bitset covers the ids of the block's graph (gptr: a graph's rows are a contiguous range of ids).
// LDS: bits[max_words] | pre[max_words] | mask[16 * kDbMaskWords] | rows, beg, deg [16 each] | offsets [20] | tmp[16].
__global__ __launch_bounds__(kBlock) void dense_plan_kernel(const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                            const int32_t* __restrict__ loops, const int32_t* __restrict__ gptr,
                                                            const int32_t* __restrict__ lrows, const int32_t* __restrict__ bgraph,
                                                            const int32_t* __restrict__ counter, int max_words, int32_t* __restrict__ records,
                                                            uint8_t* __restrict__ row_flag) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x;
  if (b * kDbRows >= *counter) return;
  uint32_t* bits = reinterpret_cast<uint32_t*>(smem);
  uint32_t* pre = bits + max_words;
  uint32_t* mask = pre + max_words;
  int* rid = reinterpret_cast<int*>(mask + kDbRows * kDbMaskWords);
  int* rbeg = rid + kDbRows;
  int* rdeg = rbeg + kDbRows;
  int* roff = rdeg + kDbRows;                                // [17]: the rows' offsets in the block's entry list
  int* tmp = roff + kDbRows + 4;
  const int tid = threadIdx.x;
  const int gr = bgraph[b];
  const int lo = gptr[gr];
  const int64_t width = (int64_t)gptr[gr + 1] - lo;
  const bool clamped = width > (int64_t)max_words * 32;
  const int span = clamped ? max_words * 32 : (int)width;
  const int words = (span + 31) >> 5;
  if (tid == 0) { tmp[10] = 0; tmp[11] = 0; }
  if (tid < kDbRows) {
    const int r = lrows[b * kDbRows + tid];
    rid[tid] = r;
    rbeg[tid] = r >= 0 ? ptr[r] : 0;
    rdeg[tid] = r >= 0 ? ptr[r + 1] - ptr[r] : 0;
  }
  for (int w = tid; w < kDbRows * kDbMaskWords; w += kBlock) mask[w] = 0u;
  for (int w = tid; w < words; w += kBlock) bits[w] = 0u;
  __syncthreads();
  int nrows = 0;
  for (int i = 0; i < kDbRows; ++i) nrows += rid[i] >= 0 ? 1 : 0;        // (pads sit at the end)
  if (tid == 0) {
    int o = 0;
    for (int i = 0; i < kDbRows; ++i) { roff[i] = o; o += rdeg[i]; }
    roff[kDbRows] = o;
  }
  __syncthreads();
  const int total_entries = roff[kDbRows];
  // The block's entries as ONE list, a thread taking entries tid, tid + 256, ...: four loads in flight per thread (a loop over the
  // rows with a wave per row was a chain of dependent round trips: 91 us for 4 000 blocks).  f(row index in the block, source id).
  // (The first kKeep trips' entries -- 3 072: most blocks' whole list -- stay in registers between the two passes over the list: the
  // second pass read every index again and searched its row again, three of a block's six dependent global round trips.)
  constexpr int kKeep = 3;
  int cj[kKeep][4], ci[kKeep][4];
  auto fetch4 = [&](int q0, int (&ri)[4], int (&jv)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int q = q0 + k * kBlock;
      ri[k] = -1;
      jv[k] = 0;
      if (q < total_entries) {
        int i = 0;                                           // the row of entry q: the last i with roff[i] <= q
#pragma unroll
        for (int step = 8; step > 0; step >>= 1)
          if (i + step < kDbRows && roff[i + step] <= q) i += step;
        ri[k] = i;
        jv[k] = idx[rbeg[i] + (q - roff[i])];
      }
    }
  };
  auto for_entries = [&](auto f, auto first_pass) {
    constexpr bool FIRST = decltype(first_pass)::value;
#pragma unroll
    for (int trip = 0; trip < kKeep; ++trip) {
      const int q0 = tid + trip * 4 * kBlock;
      if (FIRST) fetch4(q0, ci[trip], cj[trip]);             // (past the end: ri = -1)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (ci[trip][k] >= 0) f(ci[trip][k], cj[trip][k]);
    }
    for (int q0 = tid + kKeep * 4 * kBlock; q0 < total_entries; q0 += 4 * kBlock) {
      int ri[4], jv[4];
      fetch4(q0, ri, jv);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (ri[k] >= 0) f(ri[k], jv[k]);
    }
  };
  {
    int bad = 0;
    if (tid < nrows) {                                       // the rows themselves
      const int j = rid[tid] - lo;
      if ((unsigned)j < (unsigned)span) atomicOr(&bits[j >> 5], 1u << (j & 31));
      else bad = 1;
    }
    for_entries([&](int i, int jr) {                         // mark the sources of the block's rows
      const int j = jr - lo;
      if ((unsigned)j < (unsigned)span) atomicOr(&bits[j >> 5], 1u << (j & 31));
      else bad = 1;                                          // an id outside the graph's range (or past the bitset)
      bad |= (jr == rid[i]) ? 1 : 0;                         // a self-loop among the entries: its cell would be two entries
    }, std::true_type{});
    if (bad) atomicOr(&tmp[10], 1);
  }
  __syncthreads();
  const int wpt = (words + kBlock - 1) / kBlock;             // words per thread, contiguous: slots ascend with the id
  const int w0 = min(tid * wpt, words), w1 = min(w0 + wpt, words);
  int mine = 0;
  for (int w = w0; w < w1; ++w) mine += __popc(bits[w]);
  int total;
  int run = block_scan<4>(mine, tmp, total);
  int32_t* __restrict__ rec = records + (int64_t)b * kDbStride;
  int32_t* __restrict__ un = rec + kDbUniOff;
  for (int w = w0; w < w1; ++w) {
    uint32_t bb = bits[w];
    pre[w] = (uint32_t)run;
    while (bb) {
      const int bit = __ffs(bb) - 1;
      bb &= bb - 1;
      if (run < kDbCap) un[run] = lo + w * 32 + bit;
      ++run;
    }
  }
  for (int s = total + tid; s < kDbCap; s += kBlock) un[s] = rid[0];     // pads: a valid row (its cells are masked out)
  const bool ok = total <= kDbCap && tmp[10] == 0;
  __syncthreads();
  if (ok) {
    for_entries([&](int i, int jr) {
      const int j = jr - lo;
      const uint32_t s = pre[j >> 5] + (uint32_t)__popc(bits[j >> 5] & ((1u << (j & 31)) - 1u));
      atomicOr(&mask[i * kDbMaskWords + (s >> 5)], 1u << (s & 31));
    }, std::false_type{});
  }
  int selfs = 0;
  if (tid < nrows) {
    const int j = rid[tid] - lo;
    if (ok) {
      selfs = (int)(pre[j >> 5] + (uint32_t)__popc(bits[j >> 5] & ((1u << (j & 31)) - 1u)));
      if (loops && loops[rid[tid]] > 0) atomicOr(&mask[tid * kDbMaskWords + (selfs >> 5)], 1u << (selfs & 31));
    }
    atomicAdd(&tmp[11], rdeg[tid]);
    row_flag[rid[tid]] = ok ? 1 : 0;
  }
  __syncthreads();
  if (tid < kDbRows) {
    rec[kDbRowsOff + tid] = rid[tid];
    rec[kDbSelfOff + tid] = selfs;
  }
  for (int w = tid; w < kDbRows * kDbMaskWords; w += kBlock) rec[kDbMaskOff + w] = (int32_t)mask[w];
  if (tid == 0) *reinterpret_cast<int4*>(rec) = make_int4(nrows, min(total, kDbCap), ok ? 1 : 0, tmp[11]);
}

// --------------------------------------------------------------------------------------------------------------- forward
// qkvs rows: [query | key | value | skip], each part H heads of pitch 16 (pads zero); out / attn_out compact [N, H C].
// One pass over a wave's column blocks with a running maximum per (row, head) (what was summed under the old maximum is rescaled when
// it grows -- the per-edge kernels' own scheme, attn_q4.hpp); the key segments of the wave's next column block are in flight while this
// one's scores, weights and value products are made; the waves' partial (maximum, denominator, weighted sum) meet in LDS.
template <int H, bool TRAIN> __global__ __launch_bounds__(kBlock) void dense_attn_fwd_kernel(const AttnFwdArgs a, const DensePlan p) {
  constexpr int CP = 16, HP = H * CP;
  __shared__ __attribute__((aligned(16))) int lds[kDbLdsInts];
  __shared__ __attribute__((aligned(16))) f32x4 red_o[H][kDbWaves * kWave];
  __shared__ float red_m[H][kDbWaves][kDbRows], red_d[H][kDbWaves][kDbRows];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int nblocks = *p.counter / kDbRows;
  const int C = a.C;
  const float sc2 = kLog2e / sqrtf((float)C);
  const float keep = 1.f / (1.f - a.drop_p);
  const uint32_t thr = attn_drop_threshold(a.drop_p);
  const bool drop = TRAIN && a.drop_p > 0.f;
  const uint64_t seed = a.seed + ((TRAIN && a.seed_counter) ? *a.seed_counter * 0xD1B54A32D192ED03ull : 0ull);
  const float* __restrict__ qkvs = a.qkvs;
  const int64_t ld = a.ld;
  const int nv = min(4, max(0, C - 4 * g));
  for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {
    const int32_t* __restrict__ rec = p.records + (int64_t)b * kDbStride;
    const int4 hdr = *reinterpret_cast<const int4*>(rec);
    if (!hdr.z) continue;                                  // (the whole workgroup: no barrier is skipped by a part of it)
    const int nrows = hdr.x, ncb = (hdr.y + 15) >> 4;
    const bool valid = r < nrows;
    const int row = valid ? rec[kDbRowsOff + r] : rec[kDbRowsOff];
    const int selfs = rec[kDbSelfOff + r];
    const float nself = a.loops ? (float)a.loops[row] : 0.f;
    const float* __restrict__ ri = qkvs + (int64_t)row * ld;
    f4u q[H], kn[H];
#pragma unroll
    for (int h = 0; h < H; ++h) q[h] = *reinterpret_cast<const f4u*>(ri + h * CP + 4 * g) * sc2;      // scores in units of log 2
    const BlockLds l = block_stage(rec, lds);
    const uint32_t* maskrow = l.mask + r * kDbMaskWords;
    if (wave < ncb) {
      const float* __restrict__ kr = qkvs + (int64_t)l.uni[16 * wave + r] * ld + HP + 4 * g;
#pragma unroll
      for (int h = 0; h < H; ++h) kn[h] = *reinterpret_cast<const f4u*>(kr + h * CP);
    }
    float m2[H], den[H];
    f32x4 o[H];
#pragma unroll
    for (int h = 0; h < H; ++h) { m2[h] = kNoMax; den[h] = 0.f; o[h] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int cb = wave; cb < ncb; cb += kDbWaves) {
      f4u k4[H];
#pragma unroll
      for (int h = 0; h < H; ++h) k4[h] = kn[h];
      if (cb + kDbWaves < ncb) {
        const float* __restrict__ kr = qkvs + (int64_t)l.uni[16 * (cb + kDbWaves) + r] * ld + HP + 4 * g;
#pragma unroll
        for (int h = 0; h < H; ++h) kn[h] = *reinterpret_cast<const f4u*>(kr + h * CP);
      }
      const int4 id4 = *reinterpret_cast<const int4*>(l.uni + 16 * cb + 4 * g);
      const int ids[4] = {id4.x, id4.y, id4.z, id4.w};
      float vv[H][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float* __restrict__ vr = qkvs + (int64_t)ids[i] * ld + 2 * HP + r;
#pragma unroll
        for (int h = 0; h < H; ++h) vv[h][i] = vr[h * CP];
      }
      const uint32_t nib = cell_bits(maskrow, cb, g);
      constexpr int HP2 = (H + 1) / 2;                     // one 32-bit hash serves a PAIR of heads (attn_fwd.hpp)
      uint32_t hash[HP2][4] = {};
      if (drop) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int hp = 0; hp < HP2; ++hp) hash[hp][i] = attn_pair_hash(seed, row, ids[i], hp);
      }
      const int u0 = 16 * cb + 4 * g;
      f32x4 s[H];
      float bm[H];
#pragma unroll
      for (int h = 0; h < H; ++h) {
        s[h] = mfma16(k4[h], q[h], f32x4{0.f, 0.f, 0.f, 0.f});
        bm[h] = kNoMax;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (nib >> i & 1u) bm[h] = fmaxf(bm[h], s[h][i]);
      }
#pragma unroll
      for (int h = 0; h < H; ++h) bm[h] = rows_max(bm[h]);
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const float mn = fmaxf(m2[h], bm[h]);
        const float rs = __builtin_amdgcn_exp2f(m2[h] - mn);
        m2[h] = mn;
        den[h] *= rs;
        o[h] *= rs;
        float w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float pr = (nib >> i & 1u) ? __builtin_amdgcn_exp2f(s[h][i] - mn) : 0.f;
          if (u0 + i == selfs) pr *= nself;
          den[h] += pr;
          w[i] = (drop && attn_pair_dropped(hash[h >> 1][i], h, thr)) ? 0.f : (drop ? pr * keep : pr);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) o[h] = mfma4(vv[h][i], w[i], o[h]);
      }
    }
    // the four waves' partial results, brought under one maximum
#pragma unroll
    for (int h = 0; h < H; ++h)
      if (g == 0) red_m[h][wave][r] = m2[h];
    __syncthreads();
#pragma unroll
    for (int h = 0; h < H; ++h) {
      float mt = red_m[h][0][r];
#pragma unroll
      for (int w = 1; w < kDbWaves; ++w) mt = fmaxf(mt, red_m[h][w][r]);
      const float rs = __builtin_amdgcn_exp2f(m2[h] - mt);
      m2[h] = mt;
      const float dw = rows_sum(den[h]) * rs;
      if (g == 0) red_d[h][wave][r] = dw;
      red_o[h][wave * kWave + lane] = o[h] * rs;
    }
    __syncthreads();
    if (wave < H) {                                        // wave h finishes head h
      const int h = wave;
      float d = 1e-16f;
#pragma unroll
      for (int w = 0; w < kDbWaves; ++w) d += red_d[h][w][r];
      const f32x4 t = waves_sum(red_o[h], lane);
      const float inv = 1.0f / d;
      const f4u acc = {t[0] * inv, t[1] * inv, t[2] * inv, t[3] * inv};                  // channels 4 g .. 4 g + 3 of (row, h)
      if (valid) {
        const f4u skip = *reinterpret_cast<const f4u*>(ri + 3 * HP + h * CP + 4 * g);
        const int off = h * C + 4 * g;
        store_channels(a.out + (int64_t)row * a.ldo + off, acc + skip, nv);
        if (TRAIN) {
          store_channels(a.attn_out + (int64_t)row * a.lda + off, acc, nv);
          if (g == 0) {
            a.stat_m[(int64_t)row * H + h] = m2[h] * kLn2;
            a.stat_den[(int64_t)row * H + h] = d;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------- destination-side backward
// Per block row i: g_q[i] = sum_j gs_ij k_j with gs_ij = alpha_ij (g_i . v_j mask_ij - delta_i) / sqrt(C); the skip part's gradient
// is g_i itself; {m, 1 / den, delta} is filed per (row, head) for the source side (the recomputing form: family_b_bwd.hip).
template <int H> __global__ __launch_bounds__(kBlock) void dense_attn_bwd_dst_kernel(const AttnBwdArgs a, const DensePlan p) {
  constexpr int CP = 16, HP = H * CP;
  __shared__ __attribute__((aligned(16))) int lds[kDbLdsInts];
  __shared__ __attribute__((aligned(16))) f32x4 red_o[H][kDbWaves * kWave];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int nblocks = *p.counter / kDbRows;
  const int C = a.C;
  const float scale = 1.0f / sqrtf((float)C), sc2 = kLog2e * scale;
  const float keep = 1.f / (1.f - a.drop_p);
  const uint32_t thr = attn_drop_threshold(a.drop_p);
  const bool drop = a.drop_p > 0.f;
  const uint64_t seed = a.seed + (a.seed_counter ? *a.seed_counter * 0xD1B54A32D192ED03ull : 0ull);
  const float* __restrict__ qkvs = a.qkvs;
  const int64_t ld = a.ld;
  const int nv = min(4, max(0, C - 4 * g));
  for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {
    const int32_t* __restrict__ rec = p.records + (int64_t)b * kDbStride;
    const int4 hdr = *reinterpret_cast<const int4*>(rec);
    if (!hdr.z) continue;
    const int nrows = hdr.x, ncb = (hdr.y + 15) >> 4;
    const bool valid = r < nrows;
    const int row = valid ? rec[kDbRowsOff + r] : rec[kDbRowsOff];
    const int selfs = rec[kDbSelfOff + r];
    const float nself = a.loops ? (float)a.loops[row] : 0.f;
    const float* __restrict__ ri = qkvs + (int64_t)row * ld;
    f4u q[H], gi[H], kn[H], vn[H];
    float m2[H], mnat[H], inv_den[H], delta[H];
    f32x4 gq[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const int off = h * C + 4 * g;
      q[h] = *reinterpret_cast<const f4u*>(ri + h * CP + 4 * g) * sc2;
      gi[h] = load_channels(a.g + (int64_t)row * a.ldg + off, nv, off + 4 <= a.ldg);
      const f4u ao = load_channels(a.attn_out + (int64_t)row * a.lda + off, nv, off + 4 <= a.lda);
      delta[h] = rows_sum(dot4(gi[h], ao));
      mnat[h] = a.stat_m[(int64_t)row * H + h];
      m2[h] = mnat[h] * kLog2e;
      inv_den[h] = 1.0f / a.stat_den[(int64_t)row * H + h];
      gq[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const BlockLds l = block_stage(rec, lds);
    const uint32_t* maskrow = l.mask + r * kDbMaskWords;
    if (wave < ncb) {
      const float* __restrict__ kr = qkvs + (int64_t)l.uni[16 * wave + r] * ld + HP + 4 * g;
#pragma unroll
      for (int h = 0; h < H; ++h) {
        kn[h] = *reinterpret_cast<const f4u*>(kr + h * CP);
        vn[h] = *reinterpret_cast<const f4u*>(kr + HP + h * CP);
      }
    }
    for (int cb = wave; cb < ncb; cb += kDbWaves) {
      f4u k4[H], v4[H];
#pragma unroll
      for (int h = 0; h < H; ++h) { k4[h] = kn[h]; v4[h] = vn[h]; }
      if (cb + kDbWaves < ncb) {
        const float* __restrict__ kr = qkvs + (int64_t)l.uni[16 * (cb + kDbWaves) + r] * ld + HP + 4 * g;
#pragma unroll
        for (int h = 0; h < H; ++h) {
          kn[h] = *reinterpret_cast<const f4u*>(kr + h * CP);
          vn[h] = *reinterpret_cast<const f4u*>(kr + HP + h * CP);
        }
      }
      const int4 id4 = *reinterpret_cast<const int4*>(l.uni + 16 * cb + 4 * g);
      const int ids[4] = {id4.x, id4.y, id4.z, id4.w};
      float kc[H][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float* __restrict__ kp = qkvs + (int64_t)ids[i] * ld + HP + r;
#pragma unroll
        for (int h = 0; h < H; ++h) kc[h][i] = kp[h * CP];
      }
      const uint32_t nib = cell_bits(maskrow, cb, g);
      constexpr int HP2 = (H + 1) / 2;                     // one 32-bit hash serves a PAIR of heads (attn_fwd.hpp)
      uint32_t hash[HP2][4] = {};
      if (drop) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int hp = 0; hp < HP2; ++hp) hash[hp][i] = attn_pair_hash(seed, row, ids[i], hp);
      }
      const int u0 = 16 * cb + 4 * g;
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const f32x4 s = mfma16(k4[h], q[h], f32x4{0.f, 0.f, 0.f, 0.f});
        const f32x4 gv = mfma16(v4[h], gi[h], f32x4{0.f, 0.f, 0.f, 0.f});
        float gs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float alpha = (nib >> i & 1u) ? __builtin_amdgcn_exp2f(s[i] - m2[h]) * inv_den[h] : 0.f;
          if (u0 + i == selfs) alpha *= nself;
          const float dmask = drop ? (attn_pair_dropped(hash[h >> 1][i], h, thr) ? 0.f : keep) : 1.f;
          gs[i] = alpha * (gv[i] * dmask - delta[h]) * scale;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) gq[h] = mfma4(kc[h][i], gs[i], gq[h]);
      }
    }
#pragma unroll
    for (int h = 0; h < H; ++h) red_o[h][wave * kWave + lane] = gq[h];
    __syncthreads();
    if (wave < H && valid) {                               // wave h finishes head h
      const int h = wave;
      const f32x4 t = waves_sum(red_o[h], lane);
      float* __restrict__ go = a.gqkvs + (int64_t)row * a.ldq;
      *reinterpret_cast<f4u*>(go + h * CP + 4 * g) = f4u{t[0], t[1], t[2], t[3]};        // pad channels: zero keys, zero sums
      *reinterpret_cast<f4u*>(go + 3 * HP + h * CP + 4 * g) = gi[h];
      if (g == 0) reinterpret_cast<float4*>(a.edge_al)[(int64_t)row * H + h] = make_float4(mnat[h], inv_den[h], delta[h], 0.f);
    }
  }
}

// ------------------------------------------------------------------------------------------------------ source-side backward
// The plan of the OUT structure: block rows are sources j, the union holds their destinations i.
//   g_k[j] = sum_i gs_ij q_i,   g_v[j] = sum_i alpha_ij mask_ij g_i,   every weight recomputed from {m_i, 1 / den_i, delta_i}.
template <int H> __global__ __launch_bounds__(kBlock) void dense_attn_bwd_src_kernel(const AttnBwdArgs a, const DensePlan p) {
  constexpr int CP = 16, HP = H * CP;
  __shared__ __attribute__((aligned(16))) int lds[kDbLdsInts];
  __shared__ __attribute__((aligned(16))) f32x4 red_o[2 * H][kDbWaves * kWave];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int nblocks = *p.counter / kDbRows;
  const int C = a.C;
  const float scale = 1.0f / sqrtf((float)C), sc2 = kLog2e * scale;
  const float keep = 1.f / (1.f - a.drop_p);
  const uint32_t thr = attn_drop_threshold(a.drop_p);
  const bool drop = a.drop_p > 0.f;
  const uint64_t seed = a.seed + (a.seed_counter ? *a.seed_counter * 0xD1B54A32D192ED03ull : 0ull);
  const float* __restrict__ qkvs = a.qkvs;
  const float* __restrict__ gr = a.g;
  const float4* __restrict__ stat = reinterpret_cast<const float4*>(a.edge_al);
  const int64_t ld = a.ld, ldg = a.ldg;
  const int nv = min(4, max(0, C - 4 * g));
  for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {
    const int32_t* __restrict__ rec = p.records + (int64_t)b * kDbStride;
    const int4 hdr = *reinterpret_cast<const int4*>(rec);
    if (!hdr.z) continue;
    const int nrows = hdr.x, ncb = (hdr.y + 15) >> 4;
    const bool valid = r < nrows;
    const int row = valid ? rec[kDbRowsOff + r] : rec[kDbRowsOff];
    const int selfs = rec[kDbSelfOff + r];
    const float nself = a.loops ? (float)a.loops[row] : 0.f;
    const float* __restrict__ rj = qkvs + (int64_t)row * ld;
    f4u kown[H], vown[H], qn[H], gn[H];
    f32x4 gk[H], gvv[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      kown[h] = *reinterpret_cast<const f4u*>(rj + HP + h * CP + 4 * g) * sc2;
      vown[h] = *reinterpret_cast<const f4u*>(rj + 2 * HP + h * CP + 4 * g);
      gk[h] = f32x4{0.f, 0.f, 0.f, 0.f};
      gvv[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const BlockLds l = block_stage(rec, lds);
    const uint32_t* maskrow = l.mask + r * kDbMaskWords;
    if (wave < ncb) {
      const int uid = l.uni[16 * wave + r];
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const int off = h * C + 4 * g;
        qn[h] = *reinterpret_cast<const f4u*>(qkvs + (int64_t)uid * ld + h * CP + 4 * g);
        gn[h] = load_channels(gr + (int64_t)uid * ldg + off, nv, off + 4 <= ldg);
      }
    }
    for (int cb = wave; cb < ncb; cb += kDbWaves) {
      f4u q4[H], g4[H];
#pragma unroll
      for (int h = 0; h < H; ++h) { q4[h] = qn[h]; g4[h] = gn[h]; }
      if (cb + kDbWaves < ncb) {
        const int uid = l.uni[16 * (cb + kDbWaves) + r];
#pragma unroll
        for (int h = 0; h < H; ++h) {
          const int off = h * C + 4 * g;
          qn[h] = *reinterpret_cast<const f4u*>(qkvs + (int64_t)uid * ld + h * CP + 4 * g);
          gn[h] = load_channels(gr + (int64_t)uid * ldg + off, nv, off + 4 <= ldg);
        }
      }
      const int4 id4 = *reinterpret_cast<const int4*>(l.uni + 16 * cb + 4 * g);
      const int ids[4] = {id4.x, id4.y, id4.z, id4.w};
      float4 st[H][4];
      float qc[H][4], gc[H][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float* __restrict__ qp = qkvs + (int64_t)ids[i] * ld + r;
        const float* __restrict__ gp = gr + (int64_t)ids[i] * ldg + r;
#pragma unroll
        for (int h = 0; h < H; ++h) {
          st[h][i] = stat[(int64_t)ids[i] * H + h];        // {m, 1 / den, delta} of destination i
          qc[h][i] = qp[h * CP];
          gc[h][i] = r < C ? gp[h * C] : 0.f;
        }
      }
      const uint32_t nib = cell_bits(maskrow, cb, g);
      constexpr int HP2 = (H + 1) / 2;                     // one 32-bit hash serves a PAIR of heads (attn_fwd.hpp)
      uint32_t hash[HP2][4] = {};
      if (drop) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int hp = 0; hp < HP2; ++hp) hash[hp][i] = attn_pair_hash(seed, ids[i], row, hp);      // (destination, source)
      }
      const int u0 = 16 * cb + 4 * g;
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const f32x4 s = mfma16(q4[h], kown[h], f32x4{0.f, 0.f, 0.f, 0.f});
        const f32x4 gv = mfma16(g4[h], vown[h], f32x4{0.f, 0.f, 0.f, 0.f});
        float gs[4], al[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float alpha = (nib >> i & 1u) ? __builtin_amdgcn_exp2f(s[i] - st[h][i].x * kLog2e) * st[h][i].y : 0.f;
          if (u0 + i == selfs) alpha *= nself;
          const float dmask = drop ? (attn_pair_dropped(hash[h >> 1][i], h, thr) ? 0.f : keep) : 1.f;
          gs[i] = alpha * (gv[i] * dmask - st[h][i].z) * scale;
          al[i] = alpha * dmask;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          gk[h] = mfma4(qc[h][i], gs[i], gk[h]);
          gvv[h] = mfma4(gc[h][i], al[i], gvv[h]);
        }
      }
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
      red_o[2 * h][wave * kWave + lane] = gk[h];
      red_o[2 * h + 1][wave * kWave + lane] = gvv[h];
    }
    __syncthreads();
    for (int w = wave; w < 2 * H; w += kDbWaves) {         // sum 2 h finishes g_k of head h, sum 2 h + 1 its g_v (three heads: six sums, four waves)
      if (!valid) break;
      const int h = w >> 1, part = 1 + (w & 1);
      const f32x4 t = waves_sum(red_o[w], lane);
      *reinterpret_cast<f4u*>(a.gqkvs + (int64_t)row * a.ldq + part * HP + h * CP + 4 * g) = f4u{t[0], t[1], t[2], t[3]};
    }
  }
}

static int dense_grid(int64_t max_blocks) { return (int)std::max<int64_t>(1, std::min<int64_t>(max_blocks, 16384)); }

static bool plan_ok(const int32_t* records, const int32_t* counter, const uint8_t* row_flag, int64_t max_blocks) {
  return records && counter && row_flag && max_blocks > 0 && aligned_to(records, 16);
}

}  // namespace mlqem

using namespace mlqem;

extern "C" int mlqem_dense_plan_record_ints(void) { return kDbStride; }
extern "C" int mlqem_dense_plan_min_degree(void) { return kDbMinDeg; }
extern "C" int64_t mlqem_dense_plan_max_blocks(int64_t num_rows, int64_t num_graphs) {
  return (num_rows + (kDbRows - 1) * num_graphs + kDbRows - 1) / kDbRows;
}

extern "C" int mlqem_dense_plan_build(const int32_t* ptr, const int32_t* idx, const int32_t* loops, const int32_t* order,
                                      const int32_t* graph_ptr, int64_t num_graphs, int64_t num_rows, int64_t max_span, int32_t* counter,
                                      int32_t* lrows, int32_t* records, uint8_t* row_flag, mlqem_stream_t stream) {
  begin_launches();
  if (num_graphs < 0 || num_rows < 0 || max_span < 0) return MLQEM_ERR_BAD_ARG;
  if (num_rows > INT32_MAX / 2 || num_graphs > INT32_MAX / 32) return MLQEM_ERR_UNSUPPORTED;
  if (num_graphs == 0 || num_rows == 0) return MLQEM_OK;
  if (!ptr || !idx || !graph_ptr || !counter || !lrows || !records || !row_flag || !aligned_to(records, 16)) return MLQEM_ERR_BAD_ARG;
  const int max_words = (int)std::max<int64_t>(1, (std::min<int64_t>(max_span, 8192 * 32) + 31) / 32);
  const size_t lds = (size_t)max_words * 8 + (size_t)kDbRows * kDbMaskWords * 4 + (size_t)kDbRows * 16 + 16 + 64;
  if (!ensure_dynamic_lds(dense_plan_kernel, lds)) return MLQEM_ERR_UNSUPPORTED;
  const int64_t max_blocks = mlqem_dense_plan_max_blocks(num_rows, num_graphs);
  int32_t* bgraph = lrows + max_blocks * kDbRows;            // (the second part of lrows: one graph id per block)
  hipLaunchKernelGGL(dense_rows_kernel, dim3((unsigned)num_graphs), dim3(kRowsThreads), 0, as_stream(stream), ptr, order, graph_ptr, kDbMinDeg,
                     lrows, bgraph, counter);
  hipLaunchKernelGGL(dense_plan_kernel, dim3((unsigned)max_blocks), dim3(kBlock), lds, as_stream(stream), ptr, idx, loops, graph_ptr, lrows,
                     bgraph, counter, max_words, records, row_flag);
  return launch_status();
}

extern "C" int mlqem_dense_attention_supported(int H, int C, int head_pitch) { return H >= 1 && H <= 3 && C <= 16 && head_pitch == 16; }

extern "C" int mlqem_dense_attention_train_f32(const float* qkvs, int64_t ld, const int32_t* in_ptr, const int32_t* in_src,
                                               const int32_t* loops, int64_t N, int64_t E, int H, int C, float drop_p, uint64_t seed,
                                               const uint64_t* seed_counter, int head_pitch, const int32_t* records, const int32_t* counter,
                                               const uint8_t* row_flag, int64_t max_blocks, int parts, float* out, int64_t ldo,
                                               float* attn_out, int64_t lda, float* stat_m, float* stat_den, mlqem_stream_t stream) {
  begin_launches();
  if (!mlqem_dense_attention_supported(H, C, head_pitch)) return MLQEM_ERR_UNSUPPORTED;
  if (N < 0 || E < 0 || C <= 0 || ld < 4 * H * head_pitch || ldo < H * C || lda < H * C || drop_p < 0.f || drop_p >= 1.f) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!qkvs || !in_ptr || !in_src || !out || !attn_out || !stat_m || !stat_den || !plan_ok(records, counter, row_flag, max_blocks))
    return MLQEM_ERR_BAD_ARG;
  if (N > INT32_MAX) return MLQEM_ERR_UNSUPPORTED;
  AttnFwdArgs a{qkvs, ld, in_ptr, in_src, loops, N, E, H, C, drop_p, seed, seed_counter, out, ldo, attn_out, lda, stat_m, stat_den,
                1, nullptr, head_pitch};
  a.skip = row_flag;
  const DensePlan p{records, counter, row_flag, max_blocks};
  if (parts & 1) launch_attn_train_q4(a, as_stream(stream));
  const dim3 grid((unsigned)dense_grid(max_blocks));
  if (parts & 2) {
    if (H == 3) hipLaunchKernelGGL((dense_attn_fwd_kernel<3, true>), grid, dim3(kBlock), 0, as_stream(stream), a, p);
    else if (H == 2) hipLaunchKernelGGL((dense_attn_fwd_kernel<2, true>), grid, dim3(kBlock), 0, as_stream(stream), a, p);
    else hipLaunchKernelGGL((dense_attn_fwd_kernel<1, true>), grid, dim3(kBlock), 0, as_stream(stream), a, p);
  }
  return launch_status();
}

extern "C" int mlqem_dense_attention_bwd_f32(const float* qkvs, int64_t ld, const float* g, int64_t ldg, const float* attn_out, int64_t lda,
                                             const float* stat_m, const float* stat_den, const int32_t* in_ptr, const int32_t* in_src,
                                             const int32_t* out_ptr, const int32_t* out_dst, const int32_t* loops, int64_t N, int64_t E, int H,
                                             int C, float drop_p, uint64_t seed, const uint64_t* seed_counter, int head_pitch,
                                             const int32_t* in_records, const int32_t* in_counter, const uint8_t* in_flag, int64_t in_max_blocks,
                                             const int32_t* out_records, const int32_t* out_counter, const uint8_t* out_flag,
                                             int64_t out_max_blocks, int parts, float* gqkvs, int64_t ldq, float* edge_al,
                                             mlqem_stream_t stream) {
  begin_launches();
  if (!mlqem_dense_attention_supported(H, C, head_pitch)) return MLQEM_ERR_UNSUPPORTED;
  if (N < 0 || E < 0 || C <= 0 || ld < 4 * H * head_pitch || ldq < 4 * H * head_pitch || ldg < H * C || lda < H * C || drop_p < 0.f || drop_p >= 1.f)
    return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  if (!qkvs || !g || !attn_out || !stat_m || !stat_den || !in_ptr || !in_src || !out_ptr || !out_dst || !gqkvs || !edge_al ||
      !aligned_to(edge_al, 16) || !plan_ok(in_records, in_counter, in_flag, in_max_blocks) ||
      !plan_ok(out_records, out_counter, out_flag, out_max_blocks))
    return MLQEM_ERR_BAD_ARG;
  if (N > INT32_MAX) return MLQEM_ERR_UNSUPPORTED;
  AttnBwdArgs a{qkvs, ld, g, ldg, attn_out, lda, stat_m, stat_den, in_ptr, in_src, out_ptr, out_dst, nullptr, loops,
                N, E, H, C, drop_p, seed, seed_counter, gqkvs, ldq, edge_al, nullptr, 1, head_pitch};
  a.skip_dst = in_flag;
  a.skip_src = out_flag;
  const DensePlan pin{in_records, in_counter, in_flag, in_max_blocks}, pout{out_records, out_counter, out_flag, out_max_blocks};
  const hipStream_t s = as_stream(stream);
  // parts: 1 = destination side, per-edge rows; 2 = destination side, blocks; 4 = source side, per-edge rows; 8 = source side, blocks.
  // The source side reads the records BOTH destination-side kernels file: a caller that spreads the parts over streams joins between.
  if (parts & 1) launch_attn_bwd_dst_q4(a, s);
  const dim3 gin((unsigned)dense_grid(in_max_blocks)), gout((unsigned)dense_grid(out_max_blocks));
  if (parts & 2) {
    if (H == 3) hipLaunchKernelGGL(dense_attn_bwd_dst_kernel<3>, gin, dim3(kBlock), 0, s, a, pin);
    else if (H == 2) hipLaunchKernelGGL(dense_attn_bwd_dst_kernel<2>, gin, dim3(kBlock), 0, s, a, pin);
    else hipLaunchKernelGGL(dense_attn_bwd_dst_kernel<1>, gin, dim3(kBlock), 0, s, a, pin);
  }
  if (parts & 4) launch_attn_bwd_src_rc_q4(a, s);
  if (parts & 8) {
    if (H == 3) hipLaunchKernelGGL(dense_attn_bwd_src_kernel<3>, gout, dim3(kBlock), 0, s, a, pout);
    else if (H == 2) hipLaunchKernelGGL(dense_attn_bwd_src_kernel<2>, gout, dim3(kBlock), 0, s, a, pout);
    else hipLaunchKernelGGL(dense_attn_bwd_src_kernel<1>, gout, dim3(kBlock), 0, s, a, pout);
  }
  return launch_status();
}
