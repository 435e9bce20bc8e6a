// Dense layers of the path (per-node projections inside the convs, the MLP heads) and their weight gradients.
// All shapes here are "tall and skinny": N rows (nodes or graphs) by I, O <= a few hundred.
//
// Both GEMM forms run on the f32-input matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, a k-ordered fmaf chain):
//   forward      Y[N,O] = X[N,I] W^T     M = 16 rows per tile, N = outputs, K = I; W fragments stay in registers
//   weight grad  gW[O,I] = gY^T X        M = outputs, N = inputs (+1 column of ones for the bias), K = the N rows,
//                                        4 rows per MFMA, operands loaded straight from HBM (each byte is used once)
// Operand lane maps (guide section 3): A: lane l holds A[l & 15][l >> 4]; B: lane l holds B[l >> 4][l & 15];
// C/D: lane l, register r holds D[(l >> 4) * 4 + r][l & 15].
#include <cstdlib>

#include <mutex>
#include <unordered_map>

#include "common.hpp"

namespace mlqem {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16x16x4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

struct LinArgs {
  const float* x; int64_t ldx;
  const float* w; const float* b; const float* rowscale;
  float* y; int64_t ldy;
  int64_t N; int I; int O; int act; int accumulate;
  float drop_p; uint64_t seed;
  int rs_cols;    // rowscale applies to outputs o < rs_cols
  int act_from;   // ReLU / dropout apply to outputs o >= act_from
  const float* gate; int64_t ldgate; float gate_scale;   // last: y = gate[n,o] > 0 ? y * gate_scale : 0
  const int32_t* xrows;   // optional row map of x (linear_mfma_v4_kernel and the lean column-block kernel): row n of X is row xrows[n] of the buffer
};

// ---------------------------------------------------------------------------------------------- forward
// One wave owns row tiles t, t + W, ... (W = waves in the grid) and OBT blocks of 16 outputs.
// KS = ceil(I / 4) k-steps; the W fragments of all k-steps are loaded once per wave.
template <int OBT, int KS, bool TRANSPOSED>
__global__ __launch_bounds__(kBlock) void linear_mfma_kernel(const LinArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int n_waves = (gridDim.x * kBlock) >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int ob0 = blockIdx.y * OBT;

  float wf[OBT][KS];
#pragma unroll
  for (int ob = 0; ob < OBT; ++ob) {
    const int o = (ob0 + ob) * 16 + lr;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = ks * 4 + lq;
      float v = 0.f;
      if (o < a.O && k < a.I) v = TRANSPOSED ? a.w[(int64_t)k * a.O + o] : a.w[(int64_t)o * a.I + k];
      wf[ob][ks] = v;
    }
  }
  float bias[OBT];
#pragma unroll
  for (int ob = 0; ob < OBT; ++ob) {
    const int o = (ob0 + ob) * 16 + lr;
    bias[ob] = (a.b && o < a.O) ? a.b[o] : 0.f;
  }

  const int64_t n_tiles = ceil_div(a.N, 16);
  for (int64_t t = wave; t < n_tiles; t += n_waves) {
    const int64_t n0 = t * 16;
    const int64_t arow = n0 + lr;
    const bool row_ok = arow < a.N;
    const float* __restrict__ xr = a.x + arow * a.ldx + lq;
    float af[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) af[ks] = (row_ok && ks * 4 + lq < a.I) ? xr[ks * 4] : 0.f;
    f32x4 acc[OBT];
#pragma unroll
    for (int ob = 0; ob < OBT; ++ob) acc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int ob = 0; ob < OBT; ++ob) acc[ob] = mfma16x16x4(af[ks], wf[ob][ks], acc[ob]);
    // every load of the epilogue before its first store (vmcnt is one in-order counter for loads and stores on gfx9: a
    // load behind a store is waited for by draining that store)
    float prev[OBT][4], gt[OBT][4], rs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = n0 + lq * 4 + r;
      rs[r] = (a.rowscale && row < a.N) ? a.rowscale[row] : 1.f;
#pragma unroll
      for (int ob = 0; ob < OBT; ++ob) {
        const int o = (ob0 + ob) * 16 + lr;
        const bool ok = o < a.O && row < a.N;
        prev[ob][r] = (a.accumulate && ok) ? a.y[row * a.ldy + o] : 0.f;
        gt[ob][r] = (a.gate && ok) ? a.gate[row * a.ldgate + o] : 1.f;
      }
    }
#pragma unroll
    for (int ob = 0; ob < OBT; ++ob) {
      const int o = (ob0 + ob) * 16 + lr;
      if (o >= a.O) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = n0 + lq * 4 + r;
        if (row >= a.N) continue;
        float* dst = a.y + row * a.ldy + o;
        float v = acc[ob][r] + bias[ob];
        if (a.accumulate) v += prev[ob][r];
        if (a.rowscale && o < a.rs_cols) v *= rs[r];
        if (o >= a.act_from) {
          if (a.act & 1) v = fmaxf(v, 0.f);
          if (a.drop_p > 0.f)
            v = uniform01(a.seed, (uint64_t)(row * a.O + o)) < a.drop_p ? 0.f : v * (1.f / (1.f - a.drop_p));
        }
        if (a.gate) v = gt[ob][r] > 0.f ? v * a.gate_scale : 0.f;
        *dst = v;
      }
    }
  }
}

// 16-byte operand loads.  The MFMA consumes K in any order as long as A and B agree, so lane (row r = l & 15,
// group q = l >> 4) loads the contiguous float4 X[r][16g + 4q .. +3] ONCE per 16-column group g and feeds its four
// components to four consecutive k-steps (step s of group g multiplies k = 16g + 4q + s); the W fragment of that step
// holds W[o][16g + 4q + s].  One load instruction then covers 64 bytes of each of 16 rows instead of 16 bytes:
// 2 loads instead of 6 for I = 22.  Needs 16-byte aligned rows (the padded activation layout).
template <int OBT, int G, bool TRANSPOSED>
__global__ __launch_bounds__(kBlock) void linear_mfma_v4_kernel(const LinArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int n_waves = (gridDim.x * kBlock) >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int ob0 = blockIdx.y * OBT;

  // W is the MFMA's A operand (M = outputs) and the X tile its B operand (N = rows): the result tile is Y^T, so a lane
  // ends up with FOUR CONSECUTIVE OUTPUTS of ONE row (D[o = 4*(l>>4) + r][n = l & 15]) and stores them as one float4.
  float wf[OBT][G][4];
#pragma unroll
  for (int ob = 0; ob < OBT; ++ob) {
    const int o = (ob0 + ob) * 16 + lr;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int k = 16 * g + 4 * lq + s4;
        float v = 0.f;
        if (o < a.O && k < a.I) v = TRANSPOSED ? a.w[(int64_t)k * a.O + o] : a.w[(int64_t)o * a.I + k];
        wf[ob][g][s4] = v;
      }
  }
  float bias[OBT][4];
#pragma unroll
  for (int ob = 0; ob < OBT; ++ob)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = (ob0 + ob) * 16 + lq * 4 + r;
      bias[ob][r] = (a.b && o < a.O) ? a.b[o] : 0.f;
    }
  const bool gate_vec = a.gate != nullptr;   // padded like the output (host checked)
  const int gate_cols = (a.O + 3) / 4 * 4;

  const int64_t n_tiles = ceil_div(a.N, 16);
  // PREFETCH (the three-tile persistent forms): the NEXT tile's operand rows are loaded -- unconditionally,
  // from a clamped row, masked when they are consumed -- before this tile's MFMAs, so a wave's loads fly under its own matrix work (two
  // waves per SIMD at 200 registers: nothing else hides them; 79 us for 249 MB at I = 128, O = 45)
  constexpr bool PREFETCH = OBT == 3 && G >= 6;   // (the four-tile forms at G >= 7 would drop to one wave per SIMD)
  const int kq = (a.I + 3) / 4 * 4;
  float4 nx[PREFETCH ? G : 1];
  auto load_raw = [&](int64_t t, float4 (&r)[PREFETCH ? G : 1]) {
    const int64_t rw = min(t * 16 + lr, a.N - 1);
    const int64_t xrw = a.xrows ? (int64_t)a.xrows[rw] : rw;
    const float* __restrict__ p = a.x + xrw * a.ldx;
#pragma unroll
    for (int g = 0; g < (PREFETCH ? G : 1); ++g) r[g] = *reinterpret_cast<const float4*>(p + min(16 * g + 4 * lq, kq - 4));
  };
  if (PREFETCH && wave < n_tiles) load_raw(wave, nx);
  for (int64_t t = wave; t < n_tiles; t += n_waves) {
    const int64_t row = t * 16 + lr;          // the row this lane loads AND stores
    const bool row_ok = row < a.N;
    float4 av[G];
    if constexpr (PREFETCH) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int k0 = 16 * g + 4 * lq;
        const bool on = row_ok && k0 < a.I;
        av[g] = make_float4(on ? nx[g].x : 0.f, (on && k0 + 1 < a.I) ? nx[g].y : 0.f, (on && k0 + 2 < a.I) ? nx[g].z : 0.f,
                            (on && k0 + 3 < a.I) ? nx[g].w : 0.f);
      }
      if (t + n_waves < n_tiles) load_raw(t + n_waves, nx);
    } else {
    const int64_t xrow = (a.xrows && row_ok) ? (int64_t)a.xrows[row] : row;
    const float* __restrict__ xr = a.x + xrow * a.ldx + 4 * lq;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int k0 = 16 * g + 4 * lq;
      av[g] = make_float4(0.f, 0.f, 0.f, 0.f);
      // one access path only (the host launches this kernel for padded rows: ldx >= round_up(I, 4), so the whole
      // float4 lies inside the row's allocation); columns >= I are zeroed below
      if (row_ok && k0 < a.I) av[g] = *reinterpret_cast<const float4*>(xr + 16 * g);
      if (k0 + 1 >= a.I) av[g].y = 0.f;
      if (k0 + 2 >= a.I) av[g].z = 0.f;
      if (k0 + 3 >= a.I) av[g].w = 0.f;
    }
    }
    // the gate rows are independent of the product: fetch them with the operands, not after the MFMAs
    float4 gv[OBT];
    if (gate_vec) {
#pragma unroll
      for (int ob = 0; ob < OBT; ++ob) {
        const int o0 = (ob0 + ob) * 16 + lq * 4;
        gv[ob] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (row_ok && o0 + 4 <= gate_cols) gv[ob] = *reinterpret_cast<const float4*>(a.gate + row * a.ldgate + o0);
      }
    }
    f32x4 acc[OBT];
#pragma unroll
    for (int ob = 0; ob < OBT; ++ob) acc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float comp[4] = {av[g].x, av[g].y, av[g].z, av[g].w};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int ob = 0; ob < OBT; ++ob) acc[ob] = mfma16x16x4(wf[ob][g][s4], comp[s4], acc[ob]);
    }
    if (!row_ok) continue;
    const float rscale = a.rowscale ? a.rowscale[row] : 1.f;
    // the accumulate operands of ALL tiles before the first store (a load behind a store drains it: one in-order vmcnt)
    float4 pv[OBT];
#pragma unroll
    for (int ob = 0; ob < OBT; ++ob) {
      const int o0 = (ob0 + ob) * 16 + lq * 4;
      pv[ob] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.accumulate && o0 < a.O) pv[ob] = *reinterpret_cast<const float4*>(a.y + row * a.ldy + o0);
    }
#pragma unroll
    for (int ob = 0; ob < OBT; ++ob) {
      const int o0 = (ob0 + ob) * 16 + lq * 4;
      if (o0 >= a.O) continue;
      float* dst = a.y + row * a.ldy + o0;   // padded output rows (host checked): always one float4, pads are scratch
      const float prev[4] = {pv[ob].x, pv[ob].y, pv[ob].z, pv[ob].w};
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = o0 + r;
        float u = acc[ob][r] + bias[ob][r] + prev[r];
        if (a.rowscale && o < a.rs_cols) u *= rscale;
        if (o >= a.act_from) {
          if (a.act & 1) u = fmaxf(u, 0.f);
          if (a.drop_p > 0.f)
            u = uniform01(a.seed, (uint64_t)(row * a.O + o)) < a.drop_p ? 0.f : u * (1.f / (1.f - a.drop_p));
        }
        v[r] = u;
      }
      if (a.gate) {   // padded like the output (host checked)
        const float4 m = gv[ob];
        v[0] = m.x > 0.f ? v[0] * a.gate_scale : 0.f; v[1] = m.y > 0.f ? v[1] * a.gate_scale : 0.f;
        v[2] = m.z > 0.f ? v[2] * a.gate_scale : 0.f; v[3] = m.w > 0.f ? v[3] * a.gate_scale : 0.f;
      }
      if (a.act & 256) vstore<4>(dst, v); else vstore_nt<4>(dst, v);
    }
  }
}

// NARROW outputs (O <= 4) of padded rows with nothing in the epilogue but the bias: ASAPooling's one-wide score projections a = w . segmax
// and c = att_x . x, and LEConv's three one-wide projections as one [3, D] matrix (docs/tutorials/gnn.py:93-101 through PyG's ASAPooling).
// A 16-row MFMA tile uses one to three of its sixteen output columns and loads a row as 64-byte pieces; here LPR lanes own a row
// (a float4 each: the row is one contiguous read), multiply against the weights in registers and add up across the lanes; R rows per
// thread keep R loads in flight.  45 MB of rows at N = 353 k: 23 -> ~9 us.
template <int LPR, int OC, int R>
__global__ __launch_bounds__(kBlock) void linear_rowdot_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                                               const float* __restrict__ b, float* __restrict__ y, int64_t ldy, int64_t N,
                                                               int I) {
  constexpr int kRows = kBlock / LPR;                 // rows per block and pass
  const int sub = threadIdx.x % LPR, rl = threadIdx.x / LPR;
  const int k0 = 4 * sub;
  float wv[OC][4];
#pragma unroll
  for (int o = 0; o < OC; ++o)
#pragma unroll
    for (int j = 0; j < 4; ++j) wv[o][j] = k0 + j < I ? w[o * I + k0 + j] : 0.f;
  const int64_t row0 = (int64_t)blockIdx.x * (kRows * R) + rl;
  float4 v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t row = row0 + (int64_t)r * kRows;
    v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < N && k0 < I) v[r] = *reinterpret_cast<const float4*>(x + row * ldx + k0);
    if (k0 + 1 >= I) v[r].y = 0.f;       // the pad columns of a row are scratch
    if (k0 + 2 >= I) v[r].z = 0.f;
    if (k0 + 3 >= I) v[r].w = 0.f;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t row = row0 + (int64_t)r * kRows;
    float acc[OC];
#pragma unroll
    for (int o = 0; o < OC; ++o) {
      acc[o] = fmaf(v[r].w, wv[o][3], fmaf(v[r].z, wv[o][2], fmaf(v[r].y, wv[o][1], v[r].x * wv[o][0])));
#pragma unroll
      for (int m = LPR / 2; m >= 1; m >>= 1) acc[o] += __shfl_xor(acc[o], m, LPR);
    }
    if (sub == 0 && row < N) {
#pragma unroll
      for (int o = 0; o < OC; ++o) y[row * ldy + o] = acc[o] + (b ? b[o] : 0.f);
    }
  }
}

// WIDE outputs from narrow inputs with NOTHING in the epilogue but the bias: the q / k / v / skip projection of a TransformerConv
// (22 -> 180 on the circuit DAGs, 45 -> 120 on the pooled graph; docs/tutorials/gnn.py:80-91).  The kernel above gives a wave
// OBT = 4 output tiles, so three workgroups write three 256-byte pieces of every 720-byte row at different times, each store
// instruction 64 bytes of 16 different rows: 1.7 TB/s at 11 M rows.  Here a wave owns 16 WHOLE rows: all NT tiles of the product
// (W fragments in registers: NT x G x 4), the result transposed through LDS into the rows' memory order -- a tile of 16 padded rows
// IS one contiguous block when ldy = round_up(O, 4) -- and written as full 1 KB wave stores; the bias is added on the way out.
template <int NT, int G, bool K24 = false>     // K24: 17-24 input columns as 16 + 8 (see linear_fanout_lds_kernel)
__global__ __launch_bounds__(kBlock) void linear_rows_lds_kernel(const LinArgs a, int LS) {
  extern __shared__ float s_rows_lds[];                  // [4 waves][16 rows][LS] | bias [NT * 16]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int n_waves = (gridDim.x * kBlock) >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int c4o = (a.O + 3) / 4 * 4, q4 = c4o / 4;
  float* __restrict__ tile = s_rows_lds + (size_t)wid * 16 * LS;
  float* __restrict__ s_bias = s_rows_lds + (size_t)4 * 16 * LS;
  for (int i = threadIdx.x; i < NT * 16; i += kBlock) s_bias[i] = (a.b && i < a.O) ? a.b[i] : 0.f;
  __syncthreads();
  float wf[NT][G][4];
#pragma unroll
  for (int ob = 0; ob < NT; ++ob) {
    const int o = ob * 16 + lr;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int k = (K24 && g == 1) ? (s4 < 2 ? 16 + 2 * lq + s4 : a.I) : 16 * g + 4 * lq + s4;
        wf[ob][g][s4] = (o < a.O && k < a.I) ? a.w[(int64_t)o * a.I + k] : 0.f;
      }
  }
  // a lane's places in the copy-out: float4 number i = lane + 64 j of the tile (row i / q4, column group i % q4)
  const int r_first = lane / q4, c_first = lane - r_first * q4, r_step = 64 / q4, c_step = 64 - r_step * q4;
  const int64_t n_tiles = ceil_div(a.N, 16);
  // the rows of a tile as this lane's B operands (zeros beyond I and beyond N); the NEXT tile's are requested before this tile's
  // products are formed: at 125 + 72 registers only two waves share a SIMD, and a wave that loads, multiplies, transposes and stores
  // one thing after the other left the launch at 3.1 TB/s of stores
  auto load_rows = [&](int64_t t, float4 (&v)[G]) {
    const int64_t row = t * 16 + lr;
    const bool row_ok = t < n_tiles && row < a.N;
    const int64_t xrow = (a.xrows && row_ok) ? (int64_t)a.xrows[row] : row;
    const float* __restrict__ xr = a.x + xrow * a.ldx + 4 * lq;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      v[g] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (K24 && g == 1) {
        if (row_ok && 16 + 2 * lq < a.I) {
          const float2 h = *reinterpret_cast<const float2*>(a.x + xrow * a.ldx + 16 + 2 * lq);
          v[g].x = h.x; v[g].y = h.y;
        }
      } else {
        const int k0 = 16 * g + 4 * lq;
        if (row_ok && k0 < a.I) v[g] = *reinterpret_cast<const float4*>(xr + 16 * g);   // padded rows: the float4 lies inside the row
      }
    }
  };
  auto mask_cols = [&](float4 (&v)[G]) {             // where the values are taken over, not behind the load (that would wait for it)
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (K24 && g == 1) {
        if (16 + 2 * lq + 1 >= a.I) v[g].y = 0.f;
        continue;
      }
      const int k0 = 16 * g + 4 * lq;
      if (k0 + 1 >= a.I) v[g].y = 0.f;
      if (k0 + 2 >= a.I) v[g].z = 0.f;
      if (k0 + 3 >= a.I) v[g].w = 0.f;
    }
  };
  float4 av[G], an[G];
  load_rows(wave, av);
  mask_cols(av);
  for (int64_t t = wave; t < n_tiles; t += n_waves) {
    load_rows(t + n_waves, an);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[NT];
#pragma unroll
    for (int ob = 0; ob < NT; ++ob) acc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float comp[4] = {av[g].x, av[g].y, av[g].z, av[g].w};
#pragma unroll
      for (int s4 = 0; s4 < ((K24 && g == 1) ? 2 : 4); ++s4)
#pragma unroll
        for (int ob = 0; ob < NT; ++ob) acc[ob] = mfma16x16x4(wf[ob][g][s4], comp[s4], acc[ob]);
    }
    // lane (lr, lq) holds outputs ob 16 + 4 lq .. + 3 of row lr: one 16-byte LDS store per tile
#pragma unroll
    for (int ob = 0; ob < NT; ++ob) {
      const int c0 = ob * 16 + 4 * lq;
      if (c0 < c4o) *reinterpret_cast<float4*>(tile + lr * LS + c0) = make_float4(acc[ob][0], acc[ob][1], acc[ob][2], acc[ob][3]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int nrows = (int)min((int64_t)16, a.N - t * 16);
    float* __restrict__ dst = a.y + t * 16 * a.ldy;      // ldy == c4o: the tile's rows are one contiguous block
    int r = r_first, c = c_first;
    for (int i = lane; i < nrows * q4; i += 64) {
      const float4 v = *reinterpret_cast<const float4*>(tile + r * LS + 4 * c);
      const float4 bb = *reinterpret_cast<const float4*>(s_bias + 4 * c);
      const float out[4] = {v.x + bb.x, v.y + bb.y, v.z + bb.z, v.w + bb.w};
      vstore_nt<4>(dst + (int64_t)i * 4, out);
      c += c_step; r += r_step;
      if (c >= q4) { c -= q4; ++r; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the tile is read before the next product overwrites it
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int g = 0; g < G; ++g) av[g] = an[g];
    mask_cols(av);
  }
}

// ------------------------------------------------------------------------------------- bf16 matrix cores
// Y = act(X W^T + b) with the operands ROUNDED TO BF16 IN REGISTERS (round-to-nearest-even) and fp32 accumulation on
// v_mfma_f32_16x16x32_bf16 -- the "bf16 MFMA MLP head" of BASELINE.json's mixed-corpus configuration.  X, W, Y stay
// fp32 in memory (these layers are bound by reading X, not by the matrix cores), so this is an arithmetic option, not a
// bandwidth one: results differ from the fp32 path at the 1e-2 level and equal "round both operands to bf16, multiply
// exactly, add in fp32".  Lane maps (guide section 3): A: lane l holds A[l & 15][8 (l >> 4) + j], j = 0..7; B likewise;
// C/D as for the f32 MFMA, so the transposed formulation of linear_mfma_v4_kernel carries over: W is the A operand,
// the X tile the B operand, a lane ends with four consecutive outputs of one row.
typedef short bf16x8 __attribute__((ext_vector_type(8)));

// round-to-nearest-even, NaN stays NaN: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, two values per
// instruction); the integer form of it (add 0x7fff + lsb, shift, a NaN branch) cost ~6 VALU instructions per value and made
// the bf16 GEMMs slower than the fp32 ones -- a 16 x 170 tile converts 48 values per lane against 24 MFMAs.
__device__ __forceinline__ short to_bf16(float f) { return __builtin_bit_cast(short, (__bf16)f); }

template <int OBT, int G, bool TRANSPOSED>
__global__ __launch_bounds__(kBlock) void linear_bf16_kernel(const LinArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int n_waves = (gridDim.x * kBlock) >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int ob0 = blockIdx.y * OBT;

  bf16x8 wf[OBT][G];
#pragma unroll
  for (int ob = 0; ob < OBT; ++ob) {
    const int o = (ob0 + ob) * 16 + lr;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 32 * g + 8 * lq + j;
        wf[ob][g][j] = (o < a.O && k < a.I) ? to_bf16(TRANSPOSED ? a.w[(int64_t)k * a.O + o] : a.w[(int64_t)o * a.I + k]) : (short)0;
      }
  }
  float bias[OBT][4];
#pragma unroll
  for (int ob = 0; ob < OBT; ++ob)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = (ob0 + ob) * 16 + lq * 4 + r;
      bias[ob][r] = (a.b && o < a.O) ? a.b[o] : 0.f;
    }
  const bool vec_load = a.ldx % 4 == 0 && aligned_to_dev(a.x, 16);
  const bool vec_store = a.ldy % 4 == 0 && aligned_to_dev(a.y, 16);

  const int64_t n_tiles = ceil_div(a.N, 16);
  for (int64_t t = wave; t < n_tiles; t += n_waves) {
    const int64_t row = t * 16 + lr;
    const bool row_ok = row < a.N;
    const float* __restrict__ xr = a.x + row * a.ldx;
    bf16x8 xf[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int k0 = 32 * g + 8 * lq;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
      if (row_ok && vec_load && k0 + 8 <= a.ldx) {       // both float4s inside the row's allocation
        const float4 p = *reinterpret_cast<const float4*>(xr + k0), q = *reinterpret_cast<const float4*>(xr + k0 + 4);
        v[0] = p.x; v[1] = p.y; v[2] = p.z; v[3] = p.w; v[4] = q.x; v[5] = q.y; v[6] = q.z; v[7] = q.w;
      } else if (row_ok) {
#pragma unroll
        for (int j = 0; j < 8; ++j) if (k0 + j < a.I) v[j] = xr[k0 + j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) xf[g][j] = (k0 + j < a.I) ? to_bf16(v[j]) : (short)0;
    }
    f32x4 acc[OBT];
#pragma unroll
    for (int ob = 0; ob < OBT; ++ob) acc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int ob = 0; ob < OBT; ++ob)
        acc[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ob][g], xf[g], acc[ob], 0, 0, 0);
    if (!row_ok) continue;
#pragma unroll
    for (int ob = 0; ob < OBT; ++ob) {
      const int o0 = (ob0 + ob) * 16 + lq * 4;
      if (o0 >= a.O) continue;
      float* dst = a.y + row * a.ldy + o0;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float u = acc[ob][r] + bias[ob][r];
        if (a.act & 1) u = fmaxf(u, 0.f);
        v[r] = u;
      }
      if (vec_store && o0 + 4 <= a.O) vstore<4>(dst, v);
      else
#pragma unroll
        for (int r = 0; r < 4; ++r) if (o0 + r < a.O) dst[r] = v[r];
    }
  }
}

// ------------------------------------------------------------------------- column-partitioned operands
// Y = X W^T (+ b) where X and/or Y are CONCATENATIONS OF COLUMN BLOCKS THAT LIVE IN SEPARATE BUFFERS: block p of X is
// xp[p][N, xc] and occupies columns [p*xw, p*xw + xc) of the concatenated matrix (xw = multiple of 4; the columns
// between xc and xw are padding: read as 0, written as scratch).  One read of a shared input serves several
// projections (fan-out: y_p = x W_p^T) and one write serves a sum of projections (fan-in: y = sum_p x_p W_p^T)
// without ever materialising the concatenation -- whose rows would be too wide for the aggregation gathers.
// Same lane maps and summation order as linear_mfma_v4_kernel; every row access is one aligned float4.
constexpr int kMaxParts = MLQEM_MAX_COL_PARTS;

struct PartsArgs {
  const float* xp[kMaxParts]; int64_t ldx[kMaxParts]; int xn, xw, xc;
  float* yp[kMaxParts]; int64_t ldy[kMaxParts]; int yn, yw, yc;
  // Weights by block, unpadded, as the layers store them: block k is wk[k][rows, cols] row-major and belongs to output
  // block k (fan-out: rows = yc outputs, cols = xc inputs; needs xn == 1) or, TRANSPOSED, to input block k (fan-in:
  // rows = xc, cols = yc outputs, used as its transpose; needs yn == 1).  wm[k] (optional) is subtracted element-wise:
  // the Clenshaw form of ChebConv needs W_0 - W_2.  bk[k] (optional): bias of output block k.
  const float* wk[kMaxParts]; const float* wm[kMaxParts]; const float* bk[kMaxParts];
  const float* rsk[kMaxParts];   // optional per-output-block row scale (fan-out): Y_k[n,:] *= rsk[k][n]  (GCN's D^-1/2)
  int64_t N; int I, O;   // I = xn * xw, O = yn * yw: the concatenated (padded) column spaces
  const float* gate; int64_t ldgate; float gate_scale;   // single-block Y only: y = gate[n,o] > 0 ? y * gate_scale : 0
  int plain_stores;   // a block's rows are wider than one store instruction (16 columns): see mlqem_linear_f32
  int act; float drop_p; uint64_t seed;   // ACT kernels only: ReLU (bit 0) and inverted dropout keyed by (seed, n*yc + o)
  const int32_t* xrows;   // optional row map: row n of X is row xrows[n] of the buffers (X read straight from the arena)
};

template <int OBT, int G, bool TRANSPOSED, bool ACT, bool GATE>
__global__ __launch_bounds__(kBlock) void linear_parts_kernel(const PartsArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int n_waves = (gridDim.x * kBlock) >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int ob0 = blockIdx.y * OBT;

  // this lane's input column groups: k0 = 16g + 4*lq .. +3 lie inside ONE block (xw is a multiple of 4)
  const float* xcol[G]; int64_t xld[G]; int xlive[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int k0 = 16 * g + 4 * lq;
    const int part = k0 / a.xw, lc = k0 - part * a.xw;
    const bool ok = k0 < a.I && part < a.xn && lc < a.xc;
    xcol[g] = ok ? a.xp[part] + lc : nullptr;
    xld[g] = ok ? a.ldx[part] : 0;
    xlive[g] = ok ? min(4, a.xc - lc) : 0;
  }
  float wf[OBT][G][4];
#pragma unroll
  for (int ob = 0; ob < OBT; ++ob) {
    const int o = (ob0 + ob) * 16 + lr;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int k = 16 * g + 4 * lq + s4;
        float v = 0.f;
        if (o < a.O && k < a.I && s4 < xlive[g]) {
          const int xpart = k / a.xw, xl = k - xpart * a.xw;      // input block / column inside it
          const int ypart = o / a.yw, yl = o - ypart * a.yw;      // output block / column inside it
          if (yl < a.yc && ypart < a.yn) {
            const int blk = TRANSPOSED ? xpart : ypart;
            const int64_t at = TRANSPOSED ? (int64_t)xl * a.yc + yl : (int64_t)yl * a.xc + xl;
            v = a.wk[blk][at];
            if (a.wm[blk]) v -= a.wm[blk][at];
          }
        }
        wf[ob][g][s4] = v;
      }
  }
  float bias[OBT][4];
  float* ycol[OBT]; int64_t yld[OBT]; const float* yrs[OBT];
#pragma unroll
  for (int ob = 0; ob < OBT; ++ob) {
    const int o0 = (ob0 + ob) * 16 + lq * 4;
    const int part = o0 / a.yw, lc = o0 - part * a.yw;
    const bool ok = o0 < a.O && part < a.yn;
    ycol[ob] = ok ? a.yp[part] + lc : nullptr;
    yld[ob] = ok ? a.ldy[part] : 0;
    yrs[ob] = ok ? a.rsk[part] : nullptr;
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[ob][r] = (ok && a.bk[part] && lc + r < a.yc) ? a.bk[part][lc + r] : 0.f;
  }

  const int64_t n_tiles = ceil_div(a.N, 16);
  for (int64_t t = wave; t < n_tiles; t += n_waves) {
    const int64_t row = t * 16 + lr;
    const bool row_ok = row < a.N;
    const int64_t xrow = (a.xrows && row_ok) ? (int64_t)a.xrows[row] : row;
    float4 av[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      av[g] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row_ok && xcol[g]) av[g] = *reinterpret_cast<const float4*>(xcol[g] + xrow * xld[g]);
      if (xlive[g] < 2) av[g].y = 0.f;     // padding may hold anything (NaN included): it must not reach the MFMA
      if (xlive[g] < 3) av[g].z = 0.f;
      if (xlive[g] < 4) av[g].w = 0.f;
      if (xlive[g] < 1) av[g].x = 0.f;
    }
    float4 gv[GATE ? OBT : 1];
    if (GATE) {   // host checked: one output block, gate rows padded like it; fetched with the operands
#pragma unroll
      for (int ob = 0; ob < OBT; ++ob) {
        gv[GATE ? ob : 0] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (row_ok && ycol[ob]) gv[GATE ? ob : 0] = *reinterpret_cast<const float4*>(a.gate + row * a.ldgate + (ob0 + ob) * 16 + lq * 4);
      }
    }
    f32x4 acc[OBT];
#pragma unroll
    for (int ob = 0; ob < OBT; ++ob) acc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float comp[4] = {av[g].x, av[g].y, av[g].z, av[g].w};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int ob = 0; ob < OBT; ++ob) acc[ob] = mfma16x16x4(wf[ob][g][s4], comp[s4], acc[ob]);
    }
    if (!row_ok) continue;
#pragma unroll
    for (int ob = 0; ob < OBT; ++ob) {
      if (!ycol[ob]) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[ob][r] + bias[ob][r];
      if (yrs[ob]) {
        const float rs = yrs[ob][row];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= rs;
      }
      if (ACT) {   // single-block Y (host checked): the column inside the block is the output index
        const int o0 = (ob0 + ob) * 16 + lq * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (a.act & 1) v[r] = fmaxf(v[r], 0.f);
          if (a.drop_p > 0.f)
            v[r] = uniform01(a.seed, (uint64_t)(row * a.yc + o0 + r)) < a.drop_p ? 0.f : v[r] * (1.f / (1.f - a.drop_p));
        }
      }
      if (GATE) {
        const float4 m = gv[GATE ? ob : 0];
        v[0] = m.x > 0.f ? v[0] * a.gate_scale : 0.f; v[1] = m.y > 0.f ? v[1] * a.gate_scale : 0.f;
        v[2] = m.z > 0.f ? v[2] * a.gate_scale : 0.f; v[3] = m.w > 0.f ? v[3] * a.gate_scale : 0.f;
      }
      if (a.plain_stores) vstore<4>(ycol[ob] + row * yld[ob], v); else vstore_nt<4>(ycol[ob] + row * yld[ob], v);
    }
  }
}

// Fan-out with every output block narrower than one MFMA tile (<= 16 columns: the hidden widths of this path): block k
// IS output tile k.  The weight fragments of ALL blocks live in LDS (yn * G KB), not in registers -- the register-resident
// form above needs ~170 VGPRs for six blocks and runs two waves per SIMD at 3 TB/s; here a wave holds one X fragment and
// one accumulator tile at a time (~48 VGPRs, eight waves per SIMD), reads a tile's fragments with one ds_read_b128 per
// 16 input columns, and a tile's store is 16 rows x (block width) contiguous bytes of ONE buffer.  Same MFMA step order
// per output as linear_parts_kernel: bit-identical results.
// K24 (G = 2, 17-24 input columns: the 22 node features of the headline model): the second k-group holds EIGHT columns, two per
// lane quarter (k = 16 + 2 lq + s, s < 2, an 8-byte load) instead of sixteen of which the upper eight are padding -- six MFMAs per
// output block and 16-row tile instead of eight (the launch's MFMA time at 11.3 M rows: 441 -> 331 us of a 945 us launch).
template <int G, bool K24 = false>
__global__ __launch_bounds__(kBlock) void linear_fanout_lds_kernel(const PartsArgs a) {
  static_assert(!K24 || G == 2, "the 24-column form has two k-groups");
  __shared__ float4 s_w[kMaxParts][G][kWave];
  __shared__ float s_b[kMaxParts][16];
  const int tid = threadIdx.x;
  for (int idx = tid; idx < a.yn * G * kWave; idx += kBlock) {
    const int blk = idx / (G * kWave), g = (idx / kWave) % G, l = idx % kWave;
    const int o = l & 15, lq = l >> 4;
    float v[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int k = (K24 && g == 1) ? (s4 < 2 ? 16 + 2 * lq + s4 : a.xc) : 16 * g + 4 * lq + s4;
      float w = 0.f;
      if (o < a.yc && k < a.xc) {
        w = a.wk[blk][(int64_t)o * a.xc + k];
        if (a.wm[blk]) w -= a.wm[blk][(int64_t)o * a.xc + k];
      }
      v[s4] = w;
    }
    s_w[blk][g][l] = make_float4(v[0], v[1], v[2], v[3]);
  }
  for (int idx = tid; idx < a.yn * 16; idx += kBlock) {
    const int blk = idx >> 4, o = idx & 15;
    s_b[blk][o] = (a.bk[blk] && o < a.yc) ? a.bk[blk][o] : 0.f;
  }
  __syncthreads();
  const int lane = tid & 63;
  const int wave = (blockIdx.x * kBlock + tid) >> 6;
  const int n_waves = (gridDim.x * kBlock) >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const float* xcol[G]; int xlive[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int k0 = (K24 && g == 1) ? 16 + 2 * lq : 16 * g + 4 * lq;
    xlive[g] = k0 < a.xc ? min((K24 && g == 1) ? 2 : 4, a.xc - k0) : 0;
    xcol[g] = xlive[g] ? a.xp[0] + k0 : nullptr;
  }
  const bool store_lane = 4 * lq < a.yw;       // the block's padded width: lanes beyond it own no output columns
  const int64_t n_tiles = ceil_div(a.N, 16);
  // The x rows of the NEXT tile are requested before this tile's stores are issued (and its row-map entry one tile earlier
  // still): a load issued behind a store can only be waited for by draining that store (one in-order vmcnt), so a tile whose
  // operand loads sit behind the previous tile's six stores starts with a write round trip.  Loads are unconditional (rows
  // clamped into the matrix); pad columns are zeroed where the values are consumed.
  auto load_x = [&](int64_t t, int xmap, float4 (&v)[G]) {
    const int64_t r = min(t * 16 + lr, a.N - 1);
    const int64_t xr = a.xrows ? (int64_t)xmap : r;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (K24 && g == 1) {
        const float2 h = xcol[g] ? *reinterpret_cast<const float2*>(xcol[g] + xr * a.ldx[0]) : make_float2(0.f, 0.f);
        v[g] = make_float4(h.x, h.y, 0.f, 0.f);
      } else {
        v[g] = xcol[g] ? *reinterpret_cast<const float4*>(xcol[g] + xr * a.ldx[0]) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto map_at = [&](int64_t t) {
    const int64_t r = min(t * 16 + lr, a.N - 1);
    return (a.xrows && t < n_tiles) ? a.xrows[r] : 0;
  };
  float4 av[G], an[G];
  int xnext = map_at(wave + n_waves);
  load_x(wave, map_at(wave), av);
  for (int64_t t = wave; t < n_tiles; t += n_waves) {
    const int64_t row = t * 16 + lr;
    const bool row_ok = row < a.N;
    // every global LOAD of the tile is issued here, before its first store: vmcnt counts loads and stores in one in-order
    // queue on gfx9, so a load behind a store can only be waited for by draining that store -- a row-scale load inside the
    // block loop cost one write round trip per block (1307 vs 846 us on the six-block first layer).  The row scales (needed
    // by this tile) go first, the next tile's map entry and x rows behind them: waiting for the scales then leaves the
    // prefetch in flight.
    float rsv[kMaxParts];
#pragma unroll
    for (int blk = 0; blk < kMaxParts; ++blk) rsv[blk] = (blk < a.yn && a.rsk[blk] && row_ok) ? a.rsk[blk][row] : 1.f;
    const int xmap2 = map_at(t + 2 * n_waves);
    load_x(t + n_waves, xnext, an);        // a tile beyond the end re-reads row N - 1 and is never used
    xnext = xmap2;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (xlive[g] < 2) av[g].y = 0.f;     // padding may hold anything (NaN included): it must not reach the MFMA
      if (xlive[g] < 3) av[g].z = 0.f;
      if (xlive[g] < 4) av[g].w = 0.f;
      if (xlive[g] < 1) av[g].x = 0.f;
    }
#pragma unroll
    for (int blk = 0; blk < kMaxParts; ++blk) {
      if (blk >= a.yn) break;
      const float rs = rsv[blk];
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float4 w4 = s_w[blk][g][lane];
        acc = mfma16x16x4(w4.x, av[g].x, acc);
        acc = mfma16x16x4(w4.y, av[g].y, acc);
        if (K24 && g == 1) continue;       // the group's eight columns are done
        acc = mfma16x16x4(w4.z, av[g].z, acc);
        acc = mfma16x16x4(w4.w, av[g].w, acc);
      }
      if (!row_ok || !store_lane) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[r] + s_b[blk][4 * lq + r];
        v[r] *= rs;                        // 1.0f where the block has no row scale: exact
      }
      if (a.plain_stores) vstore<4>(a.yp[blk] + row * a.ldy[blk] + 4 * lq, v);
      else vstore_nt<4>(a.yp[blk] + row * a.ldy[blk] + 4 * lq, v);
    }
#pragma unroll
    for (int g = 0; g < G; ++g) av[g] = an[g];
  }
}

// Generic fallback (any I; used for I > 128): thread per (row, output); a block stages one chunk of `oc` outputs' weights
// in LDS (blockIdx.y selects the chunk).
template <bool TRANSPOSED>
__global__ __launch_bounds__(kBlock) void linear_scalar_kernel(const LinArgs a, int oc) {
  extern __shared__ float w_lds[];  // [oc][I|1]: odd row stride -> conflict-free across o
  const int stride = a.I | 1;
  const int o0 = blockIdx.y * oc;
  const int no = min(oc, a.O - o0);
  for (int p = threadIdx.x; p < no * a.I; p += kBlock) {
    const int o = p / a.I, k = p % a.I;
    w_lds[o * stride + k] = TRANSPOSED ? a.w[(int64_t)k * a.O + o0 + o] : a.w[(int64_t)(o0 + o) * a.I + k];
  }
  __syncthreads();
  const int64_t total = a.N * no;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
    const int64_t row = t / no;
    const int ol = (int)(t - row * no), o = o0 + ol;
    const float* __restrict__ xr = a.x + row * a.ldx;
    const float* wr = w_lds + ol * stride;
    float acc = 0.f;  // one k-ordered chain, the same summation order as the MFMA path
    for (int k = 0; k < a.I; ++k) acc = fmaf(xr[k], wr[k], acc);
    float* dst = a.y + row * a.ldy + o;
    float r = acc + (a.b ? a.b[o] : 0.f);
    if (a.accumulate) r += *dst;
    if (a.rowscale && o < a.rs_cols) r *= a.rowscale[row];
    if (o >= a.act_from) {
      if (a.act & 1) r = fmaxf(r, 0.f);
      if (a.drop_p > 0.f)
        r = uniform01(a.seed, (uint64_t)(row * a.O + o)) < a.drop_p ? 0.f : r * (1.f / (1.f - a.drop_p));
    }
    if (a.gate) r = a.gate[row * a.ldgate + o] > 0.f ? r * a.gate_scale : 0.f;
    *dst = r;
  }
}

// ------------------------------------------------------------------------------------------ weight gradient
struct WgradArgs {
  const float* gyp[MLQEM_MAX_COL_PARTS]; int64_t ldgy[MLQEM_MAX_COL_PARTS];   // gy = column blocks in separate buffers (see PartsArgs); one block: gn = 1
  int gn, gw, gc;                          // blocks, columns per block in the concatenation, meaningful columns per block
  const float* x; int64_t ldx;
  const int32_t* xrows;    // optional row map for x (see PartsArgs)
  float* partial;          // [gridDim.x][O * (I + 1)]
  int64_t N; int I; int O;
};

constexpr int kWgradUnroll = 4;  // MFMA k-steps (of 4 rows) whose loads are issued together (default)

// kU: k-steps per iteration.  The wide tile shapes (six output tiles: the seven first-layer gradient blocks in one pass)
// can take kU = 2 (136 instead of 180 VGPRs: three waves per SIMD instead of two) -- measured SLOWER (1680 vs 1201 us on the
// seven-block pass): the kernel lives on loads in flight per wave, so four stays the default (MLQEM_WGRAD_WIDE_U).
template <int OBT, int IBT, int kU = kWgradUnroll, bool PF = false>
__global__ __launch_bounds__(kBlock) void wgrad_mfma_kernel(const WgradArgs a) {
  __shared__ f32x4 s_acc[4][kWave];   // the four waves' copies of ONE accumulator tile at a time (see the end)
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int wave = blockIdx.x * 4 + wid;
  const int n_waves = gridDim.x * 4;
  const int lr = lane & 15, lq = lane >> 4;
  const int I1 = a.I + 1;
  const int n_ib = (int)ceil_div(ceil_div(I1, 16), IBT);
  const int ob0 = (blockIdx.y / n_ib) * OBT, ib0 = (blockIdx.y % n_ib) * IBT;

  f32x4 acc[OBT][IBT];
#pragma unroll
  for (int ob = 0; ob < OBT; ++ob)
#pragma unroll
    for (int ib = 0; ib < IBT; ++ib) acc[ob][ib] = f32x4{0.f, 0.f, 0.f, 0.f};

  // this lane's gradient columns o = 16*ob + lr: which block they live in is fixed for the whole kernel
  const float* gcol[OBT]; int64_t gld[OBT];
#pragma unroll
  for (int ob = 0; ob < OBT; ++ob) {
    const int o = (ob0 + ob) * 16 + lr;
    const int part = o / a.gw, lc = o - part * a.gw;
    const bool ok = o < a.O && part < a.gn && lc < a.gc;
    gcol[ob] = ok ? a.gyp[part] + lc : nullptr;
    gld[ob] = ok ? a.ldgy[part] : 0;
  }

  const int64_t rows_per_iter = 4 * kU;
  const int64_t n_iters = ceil_div(a.N, rows_per_iter);
  // with a row map the x row numbers of the NEXT iteration are fetched during the current one, so the map lookup
  // never sits in front of the operand loads (a dependent lookup per iteration cost 40 % on this kernel)
  int xmap[kU];
  auto fetch_map = [&](int64_t it) {
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t n = it * rows_per_iter + u * 4 + lq;
      xmap[u] = (a.xrows && it < n_iters && n < a.N) ? a.xrows[n] : 0;
    }
  };
  auto issue = [&](int64_t it, const int (&xr)[kU], float (&A)[kU][OBT], float (&B)[kU][IBT]) {
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t n = it * rows_per_iter + u * 4 + lq;
      const bool ok = n < a.N;
      const int64_t xn = a.xrows ? (int64_t)xr[u] : n;
#pragma unroll
      for (int ob = 0; ob < OBT; ++ob) A[u][ob] = (ok && gcol[ob]) ? gcol[ob][n * gld[ob]] : 0.f;
#pragma unroll
      for (int ib = 0; ib < IBT; ++ib) {
        const int i = (ib0 + ib) * 16 + lr;
        float v = 0.f;
        if (ok) v = i < a.I ? a.x[xn * a.ldx + i] : (i == a.I ? 1.f : 0.f);  // column I: ones -> bias gradient
        B[u][ib] = v;
      }
    }
  };
  auto multiply = [&](const float (&A)[kU][OBT], const float (&B)[kU][IBT]) {
#pragma unroll
    for (int u = 0; u < kU; ++u)
#pragma unroll
      for (int ob = 0; ob < OBT; ++ob)
#pragma unroll
        for (int ib = 0; ib < IBT; ++ib) acc[ob][ib] = mfma16x16x4(A[u][ob], B[u][ib], acc[ob][ib]);
  };
  fetch_map(wave);
  if constexpr (PF) {
    // Software pipeline for the wide tile shapes (two waves per SIMD: too few for the hardware to overlap one wave's loads
    // with another's 48 MFMAs).  The kernel is bound by bytes in flight (Little's law: one 6.9 KB slab per wave in flight
    // at ~3 us of loaded latency is ~4.5 TB/s chip-wide), so the operands of iterations t + 1 AND t + 2 are in flight
    // while iteration t multiplies: three register sets, 252 VGPRs, still two waves per SIMD.
    float a0[kU][OBT], b0[kU][IBT], a1[kU][OBT], b1[kU][IBT];
    int xr0[kU], xr1[kU], xr2[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) xr0[u] = xmap[u];
    if (a.xrows) fetch_map(wave + n_waves);
#pragma unroll
    for (int u = 0; u < kU; ++u) xr1[u] = xmap[u];
    if (a.xrows) fetch_map(wave + 2 * n_waves);
    issue(wave, xr0, a0, b0);
    issue(wave + n_waves, xr1, a1, b1);
    for (int64_t it = wave; it < n_iters; it += n_waves) {
      float a2[kU][OBT], b2[kU][IBT];
#pragma unroll
      for (int u = 0; u < kU; ++u) xr2[u] = xmap[u];
      if (a.xrows) fetch_map(it + 3 * n_waves);
      issue(it + 2 * n_waves, xr2, a2, b2);           // rows beyond N load nothing and contribute zeros
      multiply(a0, b0);
#pragma unroll
      for (int u = 0; u < kU; ++u) {
#pragma unroll
        for (int ob = 0; ob < OBT; ++ob) { a0[u][ob] = a1[u][ob]; a1[u][ob] = a2[u][ob]; }
#pragma unroll
        for (int ib = 0; ib < IBT; ++ib) { b0[u][ib] = b1[u][ib]; b1[u][ib] = b2[u][ib]; }
      }
    }
  } else {
    for (int64_t it = wave; it < n_iters; it += n_waves) {
      float af[kU][OBT], bf[kU][IBT];
      int xcur[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) xcur[u] = xmap[u];
      if (a.xrows) fetch_map(it + n_waves);
      issue(it, xcur, af, bf);
      multiply(af, bf);
    }
  }
  // the four waves of the block add up through LDS in a fixed order, one accumulator tile at a time (4 KB of LDS instead
  // of 4 KB per tile: twelve tiles would cap the kernel at three workgroups per CU)
  float* __restrict__ dst = a.partial + (int64_t)blockIdx.x * a.O * I1;
#pragma unroll
  for (int ob = 0; ob < OBT; ++ob)
#pragma unroll
    for (int ib = 0; ib < IBT; ++ib) {
      s_acc[wid][lane] = acc[ob][ib];
      __syncthreads();
      if (wid == 0) {
        f32x4 t = s_acc[0][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) t += s_acc[w][lane];
        const int i = (ib0 + ib) * 16 + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = (ob0 + ob) * 16 + lq * 4 + r;
          if (o < a.O && i < I1) dst[o * I1 + i] = t[r];
        }
      }
      __syncthreads();
    }
}

// The same product for the WIDE tile shapes (the seven first-layer gradient blocks against x in one pass), rebuilt around
// what bounds it.  48 fp32 MFMAs per 16 rows are 0.44 ms of matrix-pipe time against 0.83 ms of memory time for this
// launch (scripts/micro/mix_roofline.hip: the same loads without arithmetic), so the two have to overlap; with 250
// registers the kernel above runs two waves per SIMD and they do not (1.19 ms).  Here:
//   * KU k-steps per iteration, the loads of DEPTH later iterations in flight while one multiplies -- register sets are
//     [KU][OBT + IBT] floats, all blocks share one leading dimension (a scalar), 32-bit row offsets;
//   * vmcnt is ONE in-order counter for loads and stores on gfx9: waiting for a load waits for everything issued before
//     it.  A row-map entry fetched after iteration j's operand loads and needed at the top of iteration j + 1 would drain
//     those loads and flatten the pipeline, so the map entries of iteration j + 2*DEPTH - 1 are fetched at the top of
//     iteration j, in front of its operand loads;
//   * three to four waves per SIMD (about 160 registers).
template <int OBT, int IBT, int KU, int DEPTH>
__global__ __launch_bounds__(kBlock) void wgrad_pipe_kernel(const WgradArgs a) {
  __shared__ f32x4 s_acc[4][kWave];
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int I1 = a.I + 1;
  const int64_t n_waves = (int64_t)gridDim.x * 4, wave = (int64_t)blockIdx.x * 4 + wid;
  constexpr int kRows = 4 * KU;
  const int64_t n_iters = ceil_div(a.N, (int64_t)kRows);
  const int64_t ldg = a.ldgy[0];

  f32x4 acc[OBT][IBT];
#pragma unroll
  for (int ob = 0; ob < OBT; ++ob)
#pragma unroll
    for (int ib = 0; ib < IBT; ++ib) acc[ob][ib] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* gcol[OBT];
#pragma unroll
  for (int ob = 0; ob < OBT; ++ob) {
    const int o = ob * 16 + lr;
    const int part = o / a.gw, lc = o - part * a.gw;
    gcol[ob] = (o < a.O && part < a.gn && lc < a.gc) ? a.gyp[part] + lc : nullptr;
  }
  const float* xcol[IBT]; float xfill[IBT];
#pragma unroll
  for (int ib = 0; ib < IBT; ++ib) {
    const int i = ib * 16 + lr;
    xcol[ib] = i < a.I ? a.x + i : nullptr;
    xfill[ib] = i == a.I ? 1.f : 0.f;            // column I: ones -> the bias gradient
  }

  constexpr int kMapSets = 2 * DEPTH;            // map entries of iterations it .. it + 2*DEPTH - 1, oldest first
  int xm[kMapSets][KU];
  auto fetch_map = [&](int64_t it, int (&m)[KU]) {
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int64_t n = it * kRows + u * 4 + lq;
      m[u] = (a.xrows && n < a.N) ? a.xrows[n] : 0;
    }
  };
  float A[DEPTH + 1][KU][OBT], B[DEPTH + 1][KU][IBT];
  auto issue = [&](int64_t it, const int (&m)[KU], float (&ga)[KU][OBT], float (&xb)[KU][IBT]) {
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int64_t n = it * kRows + u * 4 + lq;
      const bool ok = n < a.N;
      const int64_t xn = a.xrows ? (int64_t)m[u] : n;
#pragma unroll
      for (int ob = 0; ob < OBT; ++ob) ga[u][ob] = (ok && gcol[ob]) ? gcol[ob][n * ldg] : 0.f;
#pragma unroll
      for (int ib = 0; ib < IBT; ++ib) xb[u][ib] = ok ? (xcol[ib] ? xcol[ib][xn * a.ldx] : xfill[ib]) : 0.f;
    }
  };
  // prologue: map entries of the first 2*DEPTH - 1 iterations, then the operands of the first DEPTH
#pragma unroll
  for (int d = 0; d < kMapSets - 1; ++d) fetch_map(wave + d * n_waves, xm[d]);
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue(wave + d * n_waves, xm[d], A[d], B[d]);
  for (int64_t it = wave; it < n_iters; it += n_waves) {
    fetch_map(it + (kMapSets - 1) * n_waves, xm[kMapSets - 1]);
    issue(it + DEPTH * n_waves, xm[DEPTH], A[DEPTH], B[DEPTH]);      // rows beyond N load nothing and contribute zeros
#pragma unroll
    for (int u = 0; u < KU; ++u)
#pragma unroll
      for (int ob = 0; ob < OBT; ++ob)
#pragma unroll
        for (int ib = 0; ib < IBT; ++ib) acc[ob][ib] = mfma16x16x4(A[0][u][ob], B[0][u][ib], acc[ob][ib]);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int u = 0; u < KU; ++u) {
#pragma unroll
        for (int ob = 0; ob < OBT; ++ob) A[d][u][ob] = A[d + 1][u][ob];
#pragma unroll
        for (int ib = 0; ib < IBT; ++ib) B[d][u][ib] = B[d + 1][u][ib];
      }
#pragma unroll
    for (int d = 0; d < kMapSets - 1; ++d)
#pragma unroll
      for (int u = 0; u < KU; ++u) xm[d][u] = xm[d + 1][u];
  }
  float* __restrict__ dst = a.partial + (int64_t)blockIdx.x * a.O * I1;
#pragma unroll
  for (int ob = 0; ob < OBT; ++ob)
#pragma unroll
    for (int ib = 0; ib < IBT; ++ib) {
      s_acc[wid][lane] = acc[ob][ib];
      __syncthreads();
      if (wid == 0) {
        f32x4 t = s_acc[0][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) t += s_acc[w][lane];
        const int i = ib * 16 + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = ob * 16 + lq * 4 + r;
          if (o < a.O && i < I1) dst[o * I1 + i] = t[r];
        }
      }
      __syncthreads();
    }
}

// ------------------------------------------------------------------------- fused backward of a narrow layer
// One pass for what the backward of a hidden layer of width <= 12 needs from its GEMM: the gated data gradient
//     gx = (x > 0 ? gate_scale : 0) * (gy W)                      (mask hand-over: x is the previous activation)
// and the weight / bias gradients
//     gW = gy^T x,      gb = sum_n gb_src[n, :]                   (gb_src = gy unless the bias sits behind another op)
// gy, x and gb_src are read ONCE (the separate data-gradient and weight-gradient kernels read gy and x twice: 288 instead
// of 192 bytes per row on GCN layer 2).  A wave takes 16-row tiles: three contiguous 768-byte loads, parked in its own LDS
// region; the data gradient is one MFMA tile (W^T as the A operand, as in linear_mfma_v4_kernel), the two gradient tiles
// ([gy | gb_src]^T against [x | 1]) accumulate over the wave's tiles with K = the rows, as in wgrad_mfma_kernel; partials
// per workgroup, fixed-order second stage (wgrad_reduce_kernel): deterministic.
constexpr int kBwdFusedRows = 25;   // rows of the fused backward's gradient image: [gy^T x | gb_src^T x | colsum(gx)]
struct BwdFusedArgs {
  const float* gy; int64_t ldgy; const float* gbs; int64_t ldgbs; const float* x; int64_t ldx; const float* w;
  float gate_scale; int gate;
  float* gx; int64_t ldgx; float* partial;
  int64_t N; int I; int O;
};

__global__ __launch_bounds__(kBlock) void linear_bwd_fused_kernel(const BwdFusedArgs a) {
  __shared__ float s_t[4][3][16][12];     // per wave: gy, x, gb_src tiles
  __shared__ f32x4 s_acc[4][kWave];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wave = blockIdx.x * 4 + wid, n_waves = gridDim.x * 4;
  const int lr = lane & 15, lq = lane >> 4;
  const int I1 = a.I + 1;
  float wf[4];                              // A operand of the data gradient: A[m = i][k = o] = W[o][i], o = 4 lq + s
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    const int o = 4 * lq + s4;
    wf[s4] = (lr < a.I && o < a.O) ? a.w[(int64_t)o * a.I + lr] : 0.f;
  }
  f32x4 acc_w = f32x4{0.f, 0.f, 0.f, 0.f}, acc_b = f32x4{0.f, 0.f, 0.f, 0.f};
  float cs[4] = {0.f, 0.f, 0.f, 0.f};       // this lane's share of sum_n gx[n, 4 lq .. + 3]
  const int tr = lane / 3, tc = (lane - tr * 3) * 4;      // this lane's float4 of a 16 x 12 tile (lanes 0..47)
  const bool loader = lane < 48;
  float (*t_gy)[12] = s_t[wid][0];
  float (*t_x)[12] = s_t[wid][1];
  float (*t_gb)[12] = s_t[wid][2];
  const int64_t n_tiles = ceil_div(a.N, 16);
  for (int64_t t = wave; t < n_tiles; t += n_waves) {
    const int64_t r0 = t * 16;
    if (loader) {
      const int64_t r = r0 + tr;
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0, v2 = v0;
      if (r < a.N) {   // 4-column chunks beyond a row's padded width do not exist in memory
        if (tc < (a.O + 3) / 4 * 4) {
          v0 = *reinterpret_cast<const float4*>(a.gy + r * a.ldgy + tc);
          v2 = *reinterpret_cast<const float4*>(a.gbs + r * a.ldgbs + tc);
        }
        if (tc < (a.I + 3) / 4 * 4) v1 = *reinterpret_cast<const float4*>(a.x + r * a.ldx + tc);
      }
      *reinterpret_cast<float4*>(&t_gy[tr][tc]) = v0;
      *reinterpret_cast<float4*>(&t_x[tr][tc]) = v1;
      *reinterpret_cast<float4*>(&t_gb[tr][tc]) = v2;
    }
    // data gradient: lane (row lr, quarter lq) feeds gy[lr][4 lq .. + 3] to four k-steps and ends with gx[lr][4 lq .. + 3]
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f), xv = b;
    if (lq < 3) {
      b = *reinterpret_cast<const float4*>(&t_gy[lr][4 * lq]);
      xv = *reinterpret_cast<const float4*>(&t_x[lr][4 * lq]);
      // pad columns may hold anything: they must not reach the MFMA
      if (4 * lq + 1 >= a.O) b.y = 0.f;
      if (4 * lq + 2 >= a.O) b.z = 0.f;
      if (4 * lq + 3 >= a.O) b.w = 0.f;
      if (4 * lq >= a.O) b.x = 0.f;
    }
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    acc = mfma16x16x4(wf[0], b.x, acc);
    acc = mfma16x16x4(wf[1], b.y, acc);
    acc = mfma16x16x4(wf[2], b.z, acc);
    acc = mfma16x16x4(wf[3], b.w, acc);
    if (lq < 3 && 4 * lq < a.I && r0 + lr < a.N) {
      float v[4] = {acc[0], acc[1], acc[2], acc[3]};
      if (a.gate) {
        v[0] = xv.x > 0.f ? v[0] * a.gate_scale : 0.f; v[1] = xv.y > 0.f ? v[1] * a.gate_scale : 0.f;
        v[2] = xv.z > 0.f ? v[2] * a.gate_scale : 0.f; v[3] = xv.w > 0.f ? v[3] * a.gate_scale : 0.f;
      }
      vstore_nt<4>(a.gx + (r0 + lr) * a.ldgx + 4 * lq, v);
#pragma unroll
      for (int r = 0; r < 4; ++r) cs[r] += v[r];
    }
    // weight / bias gradients: A[m = o][k = row], B[k = row][n = i]; column I of B is the ones column
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = 4 * u + lq;
      const bool live = r0 + row < a.N;
      const float a1 = (lr < a.O) ? t_gy[row][lr < 12 ? lr : 0] : 0.f;
      const float a2 = (lr < a.O) ? t_gb[row][lr < 12 ? lr : 0] : 0.f;
      const float bc = lr < a.I ? t_x[row][lr < 12 ? lr : 0] : ((lr == a.I && live) ? 1.f : 0.f);
      acc_w = mfma16x16x4(a1, bc, acc_w);
      acc_b = mfma16x16x4(a2, bc, acc_b);
    }
  }
  // partials: [workgroup][(block * 12 + o) * (I + 1) + i], block 0 = gy rows, block 1 = gb_src rows (the layout of
  // mlqem_linear_wgrad_parts_f32 with two 12-wide blocks)
  float* __restrict__ dst = a.partial + (int64_t)blockIdx.x * kBwdFusedRows * I1;
  // row 24: the column sums of gx (what the layer BELOW needs as its bias gradient when its own aggregation sits between the two:
  // the first-layer weight-gradient pass then has one block less to read)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) cs[r] += __shfl_xor(cs[r], o);
  }
  float* s_cs = reinterpret_cast<float*>(&s_t[0][0][0][0]);      // [4][12], the tiles are done with
  __syncthreads();
  if (lr == 0 && lq < 3) {
#pragma unroll
    for (int r = 0; r < 4; ++r) s_cs[wid * 12 + 4 * lq + r] = cs[r];
  }
  __syncthreads();
  if (threadIdx.x < I1)
    dst[24 * I1 + threadIdx.x] = threadIdx.x < a.I ? (s_cs[threadIdx.x] + s_cs[12 + threadIdx.x]) + (s_cs[24 + threadIdx.x] + s_cs[36 + threadIdx.x]) : 0.f;
#pragma unroll
  for (int tile = 0; tile < 2; ++tile) {
    s_acc[wid][lane] = tile == 0 ? acc_w : acc_b;
    __syncthreads();
    if (wid == 0) {
      f32x4 tt = s_acc[0][lane];
#pragma unroll
      for (int w = 1; w < 4; ++w) tt += s_acc[w][lane];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = lq * 4 + r;
        if (o < 12 && lr < I1) dst[(tile * 12 + o) * I1 + lr] = tt[r];
      }
    }
    __syncthreads();
  }
}

// bf16 matrix-core weight gradient (the "bf16 MFMA MLP head" trains on v_mfma_f32_16x16x32_bf16 too): gW = gY^T X with both
// operands rounded to bf16 in registers, fp32 accumulation, fp32 tensors in memory.  M = outputs, N = inputs (+ the ones
// column for the bias), K = 32 rows per MFMA: lane l holds rows 8 (l >> 4) .. + 7 of column (l & 15) of each operand tile.
// Same partial-sum layout and second stage as wgrad_mfma_kernel.
template <int OBT, int IBT>
__global__ __launch_bounds__(kBlock) void wgrad_bf16_kernel(const WgradArgs a) {
  __shared__ f32x4 s_acc[4][OBT * IBT][kWave];
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int wave = blockIdx.x * 4 + wid;
  const int n_waves = gridDim.x * 4;
  const int lr = lane & 15, lq = lane >> 4;
  const int I1 = a.I + 1;
  const int n_ib = (int)ceil_div(ceil_div(I1, 16), IBT);
  const int ob0 = (blockIdx.y / n_ib) * OBT, ib0 = (blockIdx.y % n_ib) * IBT;
  f32x4 acc[OBT][IBT];
#pragma unroll
  for (int ob = 0; ob < OBT; ++ob)
#pragma unroll
    for (int ib = 0; ib < IBT; ++ib) acc[ob][ib] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t n_iters = ceil_div(a.N, 32);
  for (int64_t it = wave; it < n_iters; it += n_waves) {
    bf16x8 af[OBT], bf[IBT];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t n = it * 32 + 8 * lq + j;
      const bool ok = n < a.N;
#pragma unroll
      for (int ob = 0; ob < OBT; ++ob) {
        const int o = (ob0 + ob) * 16 + lr;
        af[ob][j] = (ok && o < a.O) ? to_bf16(a.gyp[0][n * a.ldgy[0] + o]) : (short)0;
      }
#pragma unroll
      for (int ib = 0; ib < IBT; ++ib) {
        const int i = (ib0 + ib) * 16 + lr;
        float v = 0.f;
        if (ok) v = i < a.I ? a.x[n * a.ldx + i] : (i == a.I ? 1.f : 0.f);
        bf[ib][j] = to_bf16(v);
      }
    }
#pragma unroll
    for (int ob = 0; ob < OBT; ++ob)
#pragma unroll
      for (int ib = 0; ib < IBT; ++ib)
        acc[ob][ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ob], bf[ib], acc[ob][ib], 0, 0, 0);
  }
#pragma unroll
  for (int ob = 0; ob < OBT; ++ob)
#pragma unroll
    for (int ib = 0; ib < IBT; ++ib) s_acc[wid][ob * IBT + ib][lane] = acc[ob][ib];
  __syncthreads();
  if (wid == 0) {
    float* __restrict__ dst = a.partial + (int64_t)blockIdx.x * a.O * I1;
#pragma unroll
    for (int ob = 0; ob < OBT; ++ob)
#pragma unroll
      for (int ib = 0; ib < IBT; ++ib) {
        f32x4 t = s_acc[0][ob * IBT + ib][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) t += s_acc[w][ob * IBT + ib][lane];
        const int i = (ib0 + ib) * 16 + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = (ob0 + ob) * 16 + lq * 4 + r;
          if (o < a.O && i < I1) dst[o * I1 + i] = t[r];
        }
      }
  }
}

// Stage 2: fixed-order sum over the G partials.  Block = 64 consecutive (o,i) pairs x 16 slices of the G range: every load
// instruction of a wave reads 256 contiguous bytes of one partial (8 pairs x 32 slices read eight 32-byte pieces per
// instruction: 32 us for the 34 MB of a 180 x 46 gradient, twelve launches per Family B step); a thread adds its G/16
// partials on four independent chains, then the 16 slice sums of a pair are added in slice order.
constexpr int kReducePairs = 64, kReduceSlices = 16;
__global__ __launch_bounds__(kReducePairs * kReduceSlices) void wgrad_reduce_kernel(const float* __restrict__ partial, int G, int I, int O,
                                                                                   float* __restrict__ gw, float* __restrict__ gb,
                                                                                   int accumulate, int pack_w, int pack_c) {
  // pack_c > 0: the partials hold blocks of pack_c rows back to back (the first stage packed the real columns of pack_w-wide
  // blocks into fewer MFMA tiles); gw / gb keep pack_w rows per block, the pad rows written as zeros
  __shared__ float s[kReduceSlices][kReducePairs];
  const int I1 = I + 1, pairs = O * I1;
  const int pl = threadIdx.x % kReducePairs, sl = threadIdx.x / kReducePairs;
  const int p = blockIdx.x * kReducePairs + pl;
  float t = 0.f;
  if (p < pairs) {
    const int per = (G + kReduceSlices - 1) / kReduceSlices;
    const int g1 = min(G, (sl + 1) * per);
    int g = sl * per;
    float t1 = 0.f, t2 = 0.f, t3 = 0.f;
    for (; g + 3 < g1; g += 4) {
      t += partial[(int64_t)g * pairs + p];
      t1 += partial[(int64_t)(g + 1) * pairs + p];
      t2 += partial[(int64_t)(g + 2) * pairs + p];
      t3 += partial[(int64_t)(g + 3) * pairs + p];
    }
    for (; g < g1; ++g) t += partial[(int64_t)g * pairs + p];
    t = (t + t1) + (t2 + t3);
  }
  s[sl][pl] = t;
  __syncthreads();
  if (sl == 0 && p < pairs) {
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < kReduceSlices; ++k) tot += s[k][pl];
    int o = p / I1;
    const int k = p % I1;
    if (pack_c > 0) {
      const int part = o / pack_c, lc = o - part * pack_c;
      o = part * pack_w + lc;
      if (!accumulate && lc < pack_w - pack_c) {          // pack_w - pack_c <= pack_c (checked by the launcher)
        if (k < I) gw[(o + pack_c) * I + k] = 0.f;
        else if (gb) gb[o + pack_c] = 0.f;
      }
    }
    if (k < I) {
      float* d = gw + o * I + k;
      *d = accumulate ? *d + tot : tot;
    } else if (gb) {
      gb[o] = accumulate ? gb[o] + tot : tot;
    }
  }
}

// Workgroups of the first stage (= partial sums per (o, i) pair).  1024 = 4 per CU: measured best on MI355X for the
// 2.8M-row shapes (x[22]^T g[10]: 103 us at 512, 81 us at 1024, 94 us at 2048).
constexpr int kWgradBlocks = 1024;

// The grid of a kernel whose workgroups walk the row tiles with stride gridDim.x, cut to a WHOLE number of resident rounds (occupancy x
// compute units): 2048 workgroups of a kernel that fits 6 per CU ran as one full round and a second one at a third of the
// occupancy -- a third of the launch's time for a sixth of its work (the Family A launchers below have sized their grids this way
// since round 2; the general kernels behind Family B's dense layers did not until round 4).
static int resident_of(const void* kernel) {
  static std::mutex mu;
  static std::unordered_map<const void*, int> cache;
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(kernel);
  if (it != cache.end()) return it->second;
  int per_cu = 0, dev = 0, cus = 256;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0) != hipSuccess || per_cu < 1) per_cu = 2;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  return cache[kernel] = per_cu * cus;
}
template <typename K>
static dim3 whole_rounds(K kernel, dim3 grid) {
  constexpr int off = 0;      // (was the A/B switch MLQEM_WHOLE_ROUNDS: settled)
  if (off) return grid;
  const int64_t per_round = std::max<int64_t>(1, resident_of(reinterpret_cast<const void*>(kernel)) / std::max(1u, grid.y));
  if ((int64_t)grid.x > per_round) grid.x = (unsigned)((int64_t)grid.x / per_round * per_round);
  return grid;
}

template <int OBT, bool TRANSPOSED>
static void launch_linear_mfma(const LinArgs& a, int ks, dim3 grid, hipStream_t s) {
  switch (ks) {
#define MLQEM_CASE(K) case K: hipLaunchKernelGGL((linear_mfma_kernel<OBT, K, TRANSPOSED>), whole_rounds(linear_mfma_kernel<OBT, K, TRANSPOSED>, grid), dim3(kBlock), 0, s, a); break;
    MLQEM_CASE(1) MLQEM_CASE(2) MLQEM_CASE(3) MLQEM_CASE(4) MLQEM_CASE(6) MLQEM_CASE(8) MLQEM_CASE(12) MLQEM_CASE(16)
    MLQEM_CASE(20) MLQEM_CASE(24) MLQEM_CASE(32)
#undef MLQEM_CASE
  }
}

template <int OBT, bool TRANSPOSED>
static void launch_linear_v4(const LinArgs& a, int g, dim3 grid, hipStream_t s) {
  // a wave's prologue loads OBT x G x 4 weight registers by 4-byte gathers: with many of them (128 at I = 128, O = 45: the weights ARE the
  // kernel when a wave then serves three row tiles) the grid is ONE resident round of persistent waves
  const bool one_round = OBT * g >= 16;      // (101 -> 81 us at N = 353 k)
  switch (g) {
#define MLQEM_CASE(K) case K: { \
      dim3 gr = whole_rounds(linear_mfma_v4_kernel<OBT, K, TRANSPOSED>, grid); \
      if (one_round) gr.x = (unsigned)std::min<int64_t>(gr.x, std::max<int64_t>(1, resident_of(reinterpret_cast<const void*>(linear_mfma_v4_kernel<OBT, K, TRANSPOSED>)) / std::max(1u, grid.y))); \
      hipLaunchKernelGGL((linear_mfma_v4_kernel<OBT, K, TRANSPOSED>), gr, dim3(kBlock), 0, s, a); } break;
    MLQEM_CASE(1) MLQEM_CASE(2) MLQEM_CASE(3) MLQEM_CASE(4) MLQEM_CASE(5) MLQEM_CASE(6) MLQEM_CASE(7) MLQEM_CASE(8)
#undef MLQEM_CASE
  }
}

static int round_ks(int ks) {
  const int opts[] = {1, 2, 3, 4, 6, 8, 12, 16, 20, 24, 32};
  for (int o : opts) if (ks <= o) return o;
  return -1;
}

}  // namespace mlqem

using namespace mlqem;

static int run_linear_parts(PartsArgs& a, int transposed, hipStream_t s);   // defined with the column-block launchers below

extern "C" int mlqem_linear_f32(const float* x, int64_t ldx, const float* w, int transposed, const float* b,
                                const float* rowscale, float* y, int64_t ldy, int64_t N, int I, int O, int act,
                                int accumulate, float drop_p, uint64_t seed, int rs_cols, int act_from,
                                const float* gate, int64_t ldgate, float gate_scale, const int32_t* x_rows,
                                mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || I <= 0 || O <= 0 || ldx < I || ldy < O || drop_p < 0.f || drop_p >= 1.f) return MLQEM_ERR_BAD_ARG;
  if (gate && ldgate < O) return MLQEM_ERR_BAD_ARG;
  if (rs_cols < 0) rs_cols = O;
  if (act_from < 0) act_from = 0;
  if (N == 0) return MLQEM_OK;
  if (!x || !w || !y) return MLQEM_ERR_BAD_ARG;
  // Streaming (non-temporal) stores only when one store instruction writes a row's whole output (O <= 16): with wider
  // rows each instruction writes a 64-byte piece of the row and the pieces must meet in L2 to leave as full lines
  // (measured: O = 22: 279 -> 175 us, O = 180: 266 -> 174 us with plain stores; O <= 16 unchanged).  act bit 8 = plain.
  const bool plain = O > 16;
  LinArgs a{x, ldx, w, b, rowscale, y, ldy, N, I, O, act | (plain ? 256 : 0), accumulate, drop_p, seed, rs_cols, act_from,
            gate, ldgate, gate_scale};
  hipStream_t s = as_stream(stream);
  {
    // wide rows from narrow inputs, bias only (TransformerConv's q / k / v / skip projection): whole rows per wave through LDS
    const int c4 = (O + 3) / 4 * 4;
    constexpr int rows_env = 1;      // (was the A/B switch MLQEM_LINEAR_ROWS: settled)
    if (rows_env && !transposed && !accumulate && !gate && !rowscale && act == 0 && drop_p == 0.f && O >= 96 && O <= 192 && I <= 48 &&
        ldy == c4 && ldx % 4 == 0 && ldx >= (I + 3) / 4 * 4 && aligned_to(x, 16) && aligned_to(y, 16) && N >= 4096) {
      a.xrows = x_rows;
      const int nt = O <= 128 ? 8 : 12, g = I <= 32 ? 2 : 3;
      const int LS = c4 + (((c4 / 4) & 1) ? 0 : 4);          // an odd number of float4 per LDS row: 16-byte stores of 16 rows spread over the banks
      const size_t lds = ((size_t)4 * 16 * LS + (size_t)nt * 16) * sizeof(float);
      int cus = 256, dev = 0;
      if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
      }
      // persistent waves: exactly the workgroups that are resident at once (registers allow two per CU at twelve output tiles; three
      // were launched until round 4 -- the third ran alone after the others had finished: 166 us where 125 were due)
      auto go = [&](auto kernel) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, lds) != hipSuccess || per_cu < 1) per_cu = 2;
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((int64_t)cus * per_cu, ceil_div(ceil_div(N, 16), 4)));
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), lds, s, a, LS);
      };
      const bool k24 = g == 2 && I > 16 && I <= 24 && ldx % 2 == 0;
      if (nt == 8 && g == 2) { if (k24) go(linear_rows_lds_kernel<8, 2, true>); else go(linear_rows_lds_kernel<8, 2>); }
      else if (nt == 8) go(linear_rows_lds_kernel<8, 3>);
      else if (g == 2) { if (k24) go(linear_rows_lds_kernel<12, 2, true>); else go(linear_rows_lds_kernel<12, 2>); }
      else go(linear_rows_lds_kernel<12, 3>);
      return launch_status();
    }
  }
  {
    // one to four outputs, bias only: lanes along the row (linear_rowdot_kernel)
    if (!transposed && !accumulate && !gate && !rowscale && act == 0 && drop_p == 0.f && !x_rows && O <= 4 && I <= 64 &&
        ldx % 4 == 0 && ldx >= (I + 3) / 4 * 4 && aligned_to(x, 16) && N >= 4096) {
      const bool narrow = I <= 32;
      const int lpr = narrow ? 8 : 16;
      constexpr int R = 4;
      const unsigned grid = (unsigned)ceil_div(N, (kBlock / lpr) * R);
#define MLQEM_ROWDOT(L, OC) hipLaunchKernelGGL((linear_rowdot_kernel<L, OC, R>), dim3(grid), dim3(kBlock), 0, s, x, ldx, w, b, y, ldy, N, I)
      switch (O) {
        case 1: if (narrow) MLQEM_ROWDOT(8, 1); else MLQEM_ROWDOT(16, 1); break;
        case 2: if (narrow) MLQEM_ROWDOT(8, 2); else MLQEM_ROWDOT(16, 2); break;
        case 3: if (narrow) MLQEM_ROWDOT(8, 3); else MLQEM_ROWDOT(16, 3); break;
        default: if (narrow) MLQEM_ROWDOT(8, 4); else MLQEM_ROWDOT(16, 4); break;
      }
#undef MLQEM_ROWDOT
      return launch_status();
    }
  }
  // The column-block kernel with one block on each side is the lean form of this GEMM (bias, row scale, ReLU/dropout,
  // gate; no accumulate, no column ranges) and runs 1.4-1.6x faster than the general kernel below on the tall-skinny
  // shapes of this path (2.8M x 22 -> 10: 69 vs 108 us): use it whenever it can express the call (one output tile
  // row, i.e. O <= 64: wider outputs re-read x per tile row and the general kernel below wins, 174 vs 268 us at O = 180).
  const int c4i = (I + 3) / 4 * 4, c4o = (O + 3) / 4 * 4;
  const bool lean = !accumulate && rs_cols >= O && act_from == 0 && c4i <= 64 && c4o <= 64 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= c4i &&
                    ldy >= c4o && aligned_to(x, 16) && aligned_to(y, 16) && !(transposed && (act || drop_p > 0.f)) &&
                    (!gate || (ldgate % 4 == 0 && ldgate >= c4o && aligned_to(gate, 16))) && !(transposed && rowscale) &&
                    !(act & ~1);
  constexpr int lean_env = 1;      // (was the A/B switch MLQEM_LINEAR_LEAN: settled)
  if (lean && lean_env) {
    PartsArgs p{};
    p.xp[0] = x; p.ldx[0] = ldx; p.xn = 1; p.xw = c4i; p.xc = I;
    p.yp[0] = y; p.ldy[0] = ldy; p.yn = 1; p.yw = c4o; p.yc = O;
    p.wk[0] = w; p.bk[0] = transposed ? nullptr : b; p.rsk[0] = rowscale;
    p.N = N; p.I = c4i; p.O = c4o;
    p.gate = gate; p.ldgate = ldgate; p.gate_scale = gate_scale;
    p.plain_stores = O > 16;
    p.act = act & 1; p.drop_p = drop_p; p.seed = seed;
    p.xrows = x_rows;
    if (!(transposed && b)) {   // a bias on the data-gradient form is not something the lean kernel carries
      const int rc = run_linear_parts(p, transposed, s);
      if (rc != MLQEM_ERR_UNSUPPORTED) return rc;
    }
  }
  const int ks = round_ks((I + 3) / 4);
  if (ks > 0) {
    const int ob = (O + 15) / 16;
    const bool v4_form = ldx % 4 == 0 && ldx >= c4i && aligned_to(x, 16) && ldy % 4 == 0 && ldy >= c4o && aligned_to(y, 16) &&
                         (!gate || (ldgate % 4 == 0 && ldgate >= c4o && aligned_to(gate, 16)));
    // three output tiles (O = 33..48: the 45-wide pooled rows) are their own form on the 16-byte path: a quarter fewer weight registers
    // and MFMAs than the four-tile form, which is what leaves room for the operand prefetch at two waves per SIMD
    const int obt = ob == 1 ? 1 : (ob == 2 ? 2 : (ob == 3 && v4_form ? 3 : 4));
    const int64_t tiles = ceil_div(N, 16);
    const unsigned gx = (unsigned)std::min<int64_t>(ceil_div(tiles, 4), 256 * 8);  // 4 waves per block
    dim3 grid(gx, (unsigned)ceil_div(ob, obt));
    constexpr int v4_env = 1;      // (was the A/B switch MLQEM_LINEAR_V4: settled)
    const bool padded = v4_form;
    if (x_rows && !(v4_env && padded)) return MLQEM_ERR_UNSUPPORTED;   // the row map is carried by the lean and the 16-byte kernels only
    a.xrows = x_rows;
    if (v4_env && padded) {  // padded activation rows on every operand: one 16-byte access path, no scalar tails
      const int g = (I + 15) / 16;
      if (obt == 1) transposed ? launch_linear_v4<1, true>(a, g, grid, s) : launch_linear_v4<1, false>(a, g, grid, s);
      else if (obt == 2) transposed ? launch_linear_v4<2, true>(a, g, grid, s) : launch_linear_v4<2, false>(a, g, grid, s);
      else if (obt == 3) transposed ? launch_linear_v4<3, true>(a, g, grid, s) : launch_linear_v4<3, false>(a, g, grid, s);
      else transposed ? launch_linear_v4<4, true>(a, g, grid, s) : launch_linear_v4<4, false>(a, g, grid, s);
      return launch_status();
    }
    if (obt == 1) transposed ? launch_linear_mfma<1, true>(a, ks, grid, s) : launch_linear_mfma<1, false>(a, ks, grid, s);
    else if (obt == 2) transposed ? launch_linear_mfma<2, true>(a, ks, grid, s) : launch_linear_mfma<2, false>(a, ks, grid, s);
    else transposed ? launch_linear_mfma<4, true>(a, ks, grid, s) : launch_linear_mfma<4, false>(a, ks, grid, s);
    return launch_status();
  }
  if (x_rows) return MLQEM_ERR_UNSUPPORTED;
  const int oc = (int)std::min<int64_t>(O, (48 * 1024) / ((I | 1) * sizeof(float)));
  if (oc < 1) return MLQEM_ERR_UNSUPPORTED;
  const size_t lds = (size_t)oc * (I | 1) * sizeof(float);
  const int64_t blocks = std::min<int64_t>(ceil_div(N * oc, kBlock), 256 * 16);
  dim3 grid((unsigned)blocks, (unsigned)ceil_div(O, oc));
  if (transposed)
    hipLaunchKernelGGL(linear_scalar_kernel<true>, grid, dim3(kBlock), lds, s, a, oc);
  else
    hipLaunchKernelGGL(linear_scalar_kernel<false>, grid, dim3(kBlock), lds, s, a, oc);
  return launch_status();
}

template <bool TRANSPOSED, bool ACT, bool GATE>
static bool launch_linear_parts(const PartsArgs& a, int g, int obt, dim3 grid, hipStream_t s) {
#define MLQEM_PARTS(OB, K) hipLaunchKernelGGL((linear_parts_kernel<OB, K, TRANSPOSED, ACT, GATE>), whole_rounds(linear_parts_kernel<OB, K, TRANSPOSED, ACT, GATE>, grid), dim3(kBlock), 0, s, a); return true;
#define MLQEM_PARTS_G(OB) switch (g) { case 1: MLQEM_PARTS(OB, 1) case 2: MLQEM_PARTS(OB, 2) case 3: MLQEM_PARTS(OB, 3) case 4: MLQEM_PARTS(OB, 4) default: return false; }
  if (obt == 1) MLQEM_PARTS_G(1)
  if (obt == 2) MLQEM_PARTS_G(2)
  if (obt == 3) MLQEM_PARTS_G(3)
  if (obt == 6) {   // up to 96 output columns from ONE read of a narrow x (the first layers of all three branches)
    if (ACT || GATE) return false;
    switch (g) { case 1: MLQEM_PARTS(6, 1) case 2: MLQEM_PARTS(6, 2) default: return false; }
  }
  MLQEM_PARTS_G(4)
#undef MLQEM_PARTS_G
#undef MLQEM_PARTS
  return true;
}

// Workgroups of `kernel` that are resident at once on the whole device (occupancy x compute units).  The persistent
// kernels below split their rows statically over the grid (deterministic partial sums), so a grid that is not a whole
// number of resident rounds ends with a thin last round that runs at a fraction of the occupancy the kernel needs:
// measured on the seven-block weight gradient, 3 workgroups per CU (= its occupancy) 1.01 ms, 4 per CU 1.47 ms.
template <typename K>
static int resident_workgroups(K kernel) {
  int per_cu = 0, dev = 0, cus = 256;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0) != hipSuccess || per_cu < 1) per_cu = 2;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  return per_cu * cus;
}

// Picks the tile shape and launches; `a` is complete except for the grid-related choices.
static int run_linear_parts(PartsArgs& a, int transposed, hipStream_t s) {
  const int g = (a.I + 15) / 16, ob = (a.O + 15) / 16;
  if (g > 4 || ob > 16) return MLQEM_ERR_UNSUPPORTED;
  constexpr int lds_env = 1;      // (was the A/B switch MLQEM_FANOUT_LDS: settled)
  if (lds_env && !transposed && a.xn == 1 && a.yn >= 2 && a.yw <= 16 && !a.act && a.drop_p == 0.f && !a.gate) {
    // several narrow output blocks from one read of x: weight fragments in LDS, one MFMA tile per block
    const int64_t tiles = ceil_div(a.N, 16);
    constexpr int plain_env = 0;      // (was the A/B switch MLQEM_FANOUT_PLAIN: settled)
    constexpr int grid_env = 0;      // (was the A/B switch MLQEM_FANOUT_GRID: settled)    // workgroups per CU; 0 = whole resident rounds
    constexpr int rounds_env = 1;      // (was the A/B switch MLQEM_FANOUT_ROUNDS: settled)
    a.plain_stores = plain_env;
    constexpr int k24_env = 1;      // (was the A/B switch MLQEM_FANOUT_K24: settled)
    const bool k24 = k24_env && g == 2 && a.xc <= 24 && a.ldx[0] % 2 == 0 && aligned_to(a.xp[0], 8);
    int res = 0;
    switch (g) {
      case 1: { static const int r = resident_workgroups(linear_fanout_lds_kernel<1>); res = r; break; }
      case 2: { static const int r = resident_workgroups(linear_fanout_lds_kernel<2>), r24 = resident_workgroups(linear_fanout_lds_kernel<2, true>);
                res = k24 ? r24 : r; break; }
      case 3: { static const int r = resident_workgroups(linear_fanout_lds_kernel<3>); res = r; break; }
      default: { static const int r = resident_workgroups(linear_fanout_lds_kernel<4>); res = r; break; }
    }
    dim3 grid((unsigned)std::min<int64_t>(ceil_div(tiles, 4), grid_env > 0 ? 256 * grid_env : res * rounds_env));
    switch (g) {
      case 1: hipLaunchKernelGGL(linear_fanout_lds_kernel<1>, grid, dim3(kBlock), 0, s, a); break;
      case 2:
        if (k24) hipLaunchKernelGGL((linear_fanout_lds_kernel<2, true>), grid, dim3(kBlock), 0, s, a);
        else hipLaunchKernelGGL(linear_fanout_lds_kernel<2>, grid, dim3(kBlock), 0, s, a);
        break;
      case 3: hipLaunchKernelGGL(linear_fanout_lds_kernel<3>, grid, dim3(kBlock), 0, s, a); break;
      default: hipLaunchKernelGGL(linear_fanout_lds_kernel<4>, grid, dim3(kBlock), 0, s, a); break;
    }
    return launch_status();
  }
  const int obt = ob <= 3 ? ob : ((ob == 5 || ob == 6) && g <= 2 && !a.act && a.drop_p == 0.f && !a.gate ? 6 : 4);
  const int64_t tiles = ceil_div(a.N, 16);
  dim3 grid((unsigned)std::min<int64_t>(ceil_div(tiles, 4), 256 * 8), (unsigned)ceil_div(ob, obt));
  const bool act = a.act != 0 || a.drop_p > 0.f;
  const bool gate = a.gate != nullptr;
  bool ok;
  if (transposed) ok = act ? false : (gate ? launch_linear_parts<true, false, true>(a, g, obt, grid, s)
                                           : launch_linear_parts<true, false, false>(a, g, obt, grid, s));
  else if (act) ok = gate ? false : launch_linear_parts<false, true, false>(a, g, obt, grid, s);
  else ok = gate ? launch_linear_parts<false, false, true>(a, g, obt, grid, s) : launch_linear_parts<false, false, false>(a, g, obt, grid, s);
  return ok ? launch_status() : MLQEM_ERR_UNSUPPORTED;
}

static bool parts_ok(const mlqem_col_parts* p, bool vector_rows) {
  if (!p || p->count < 1 || p->count > MLQEM_MAX_COL_PARTS || p->width < 1 || p->cols < 1 || p->cols > p->width) return false;
  if (vector_rows && p->width % 4) return false;
  for (int i = 0; i < p->count; ++i) {
    if (!p->ptr[i] || p->ld[i] < (vector_rows ? p->width : p->cols)) return false;
    if (vector_rows && (p->ld[i] % 4 || !aligned_to(p->ptr[i], 16))) return false;
  }
  return true;
}

extern "C" int mlqem_linear_parts_f32(const mlqem_col_parts* x, const float* const* w_blocks,
                                      const float* const* w_minus_blocks, int transposed,
                                      const float* const* bias_blocks, const float* const* rowscale_blocks,
                                      const mlqem_col_parts* y, int64_t N, const float* gate, int64_t ldgate,
                                      float gate_scale, const int32_t* x_rows, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || !w_blocks || !parts_ok(x, true) || !parts_ok(y, true)) return MLQEM_ERR_BAD_ARG;
  if (transposed ? y->count != 1 : x->count != 1) return MLQEM_ERR_BAD_ARG;   // the blocks sit on ONE side
  if (gate && (y->count != 1 || ldgate < y->width || ldgate % 4 || !aligned_to(gate, 16))) return MLQEM_ERR_BAD_ARG;
  const int nblk = transposed ? x->count : y->count;
  for (int i = 0; i < nblk; ++i) if (!w_blocks[i]) return MLQEM_ERR_BAD_ARG;
  if (N == 0) return MLQEM_OK;
  PartsArgs a{};
  for (int i = 0; i < x->count; ++i) { a.xp[i] = static_cast<const float*>(x->ptr[i]); a.ldx[i] = x->ld[i]; }
  for (int i = 0; i < y->count; ++i) { a.yp[i] = static_cast<float*>(y->ptr[i]); a.ldy[i] = y->ld[i]; }
  for (int i = 0; i < nblk; ++i) {
    a.wk[i] = w_blocks[i];
    a.wm[i] = w_minus_blocks ? w_minus_blocks[i] : nullptr;
    a.bk[i] = (bias_blocks && !transposed) ? bias_blocks[i] : nullptr;
    a.rsk[i] = (rowscale_blocks && !transposed) ? rowscale_blocks[i] : nullptr;
  }
  a.xn = x->count; a.xw = x->width; a.xc = x->cols;
  a.yn = y->count; a.yw = y->width; a.yc = y->cols;
  a.N = N; a.I = x->count * x->width; a.O = y->count * y->width;
  a.gate = gate; a.ldgate = ldgate; a.gate_scale = gate_scale;
  a.xrows = x_rows;
  a.plain_stores = y->cols > 16;
  return run_linear_parts(a, transposed, as_stream(stream));
}

template <int OBT, bool TRANSPOSED>
static bool launch_linear_bf16(const LinArgs& a, int g, dim3 grid, hipStream_t s) {
  switch (g) {
#define MLQEM_CASE(K) case K: hipLaunchKernelGGL((linear_bf16_kernel<OBT, K, TRANSPOSED>), grid, dim3(kBlock), 0, s, a); return true;
    MLQEM_CASE(1) MLQEM_CASE(2) MLQEM_CASE(3) MLQEM_CASE(4) MLQEM_CASE(5) MLQEM_CASE(6) MLQEM_CASE(7) MLQEM_CASE(8)
#undef MLQEM_CASE
  }
  return false;
}

extern "C" int mlqem_linear_bf16_f32(const float* x, int64_t ldx, const float* w, int transposed, const float* b, float* y,
                                     int64_t ldy, int64_t N, int I, int O, int act, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || I <= 0 || O <= 0 || ldx < I || ldy < O) return MLQEM_ERR_BAD_ARG;
  if (I > 256) return MLQEM_ERR_UNSUPPORTED;
  if (N == 0) return MLQEM_OK;
  if (!x || !w || !y) return MLQEM_ERR_BAD_ARG;
  LinArgs a{x, ldx, w, b, nullptr, y, ldy, N, I, O, act, 0, 0.f, 0, O, 0, nullptr, 0, 1.f};
  const int g = (I + 31) / 32, ob = (O + 15) / 16;
  const int obt = (ob == 1) ? 1 : ((ob == 2 || g > 4) ? 2 : 4);    // keep OBT * G weight fragments <= 16 (64 VGPRs)
  const int64_t tiles = ceil_div(N, 16);
  dim3 grid((unsigned)std::min<int64_t>(ceil_div(tiles, 4), 256 * 8), (unsigned)ceil_div(ob, obt));
  hipStream_t s = as_stream(stream);
  bool ok;
  if (transposed)
    ok = obt == 1 ? launch_linear_bf16<1, true>(a, g, grid, s) : obt == 2 ? launch_linear_bf16<2, true>(a, g, grid, s)
                                                                          : launch_linear_bf16<4, true>(a, g, grid, s);
  else
    ok = obt == 1 ? launch_linear_bf16<1, false>(a, g, grid, s) : obt == 2 ? launch_linear_bf16<2, false>(a, g, grid, s)
                                                                           : launch_linear_bf16<4, false>(a, g, grid, s);
  return ok ? launch_status() : MLQEM_ERR_UNSUPPORTED;
}

extern "C" size_t mlqem_linear_wgrad_workspace_bytes(int I, int O) {
  return (size_t)kWgradBlocks * O * (I + 1) * sizeof(float);
}

static int launch_wgrad(WgradArgs a, float* gw, float* gb, int accumulate, hipStream_t s) {
  constexpr int wide_u = 4;      // (was the A/B switch MLQEM_WGRAD_WIDE_U: settled)   // measured on the 7-block pass: 4 -> 1201 us (180 VGPRs, 2 waves/SIMD), 2 -> 1680 us (136 VGPRs, 3 waves): loads in flight per wave matter more than occupancy
  constexpr int wide_pf = 1;      // (was the A/B switch MLQEM_WGRAD_PF: settled)
  const int ob_ = (a.O + 15) / 16, ib_ = (a.I + 1 + 15) / 16;
  const bool wide = ob_ >= 4 && ob_ <= 6 && ib_ <= 2;
  constexpr int pipe_env = 2;      // (was the A/B switch MLQEM_WGRAD_PIPE: settled)    // 0: wgrad_mfma_kernel, 1: KU = 4 (two waves per SIMD), 2: KU = 2 (three)
  constexpr int wide6_env = 1;      // (was the A/B switch MLQEM_WGRAD_WIDE6: settled)
  constexpr int pipe_grid = 0;      // (was the A/B switch MLQEM_WGRAD_GRID: settled)  // workgroups per CU; 0 = the resident count
  bool uniform_ld = true;
  for (int k = 1; k < a.gn; ++k) uniform_ld = uniform_ld && a.ldgy[k] == a.ldgy[0];
  const int64_t iters = ceil_div(std::max<int64_t>(a.N, 1), 4 * (wide && wide_u == 2 ? 2 : kWgradUnroll));
  int G = (int)std::max<int64_t>(1, std::min<int64_t>(kWgradBlocks, ceil_div(iters, 4)));
  // the workgroups walk the rows with stride gridDim.x and leave one partial each: a whole number of resident rounds (whole_rounds)
#define MLQEM_WG(KERNEL, GY) { const dim3 gr = whole_rounds(KERNEL, dim3((unsigned)G, (unsigned)(GY))); G = (int)gr.x; hipLaunchKernelGGL(KERNEL, gr, dim3(kBlock), 0, s, a); }
  const int ob = (a.O + 15) / 16, ib = (a.I + 1 + 15) / 16;
  if (ob == 1 && ib <= 2) {
    MLQEM_WG((wgrad_mfma_kernel<1, 2>), 1)
  } else if (ob == 2 && ib <= 2) {   // two or three column blocks of gy against a narrow x: still ONE pass over x
    MLQEM_WG((wgrad_mfma_kernel<2, 2>), 1)
  } else if (ob == 3 && ib <= 2) {
    MLQEM_WG((wgrad_mfma_kernel<3, 2>), 1)
  } else if (ob == 4 && ib <= 2) {   // up to eight blocks (the three first layers of Family A share x): one pass
    if (wide_u == 2) MLQEM_WG((wgrad_mfma_kernel<4, 2, 2>), 1)
    else MLQEM_WG((wgrad_mfma_kernel<4, 2>), 1)
  } else if (ob <= 6 && ib <= 2 && pipe_env && uniform_ld) {
    // the first-layer blocks: software-pipelined form (see wgrad_pipe_kernel).  Blocks are 12 floats wide with 10 real columns:
    // the MFMA rows take the REAL columns back to back (six blocks: 60 rows = four tiles instead of 72 = five; seven: five instead
    // of six) -- fewer load instructions and a third fewer MFMAs for the same bytes; the second stage spreads them out again.
    constexpr int pack_env = 1;      // (was the A/B switch MLQEM_WGRAD_PACK: settled)
    const int obp = (a.gn * a.gc + 15) / 16;
    const bool pack = pack_env && a.gn > 1 && a.gc < a.gw && a.gw - a.gc <= a.gc && obp < ob && obp >= 4 && pipe_env == 2;
    const int pw = a.gw, pc = a.gc;
    if (pack) { a.gw = a.gc; a.O = a.gn * a.gc; }
    static const int res4p = resident_workgroups(wgrad_pipe_kernel<4, 2, 2, 2>), res5p = resident_workgroups(wgrad_pipe_kernel<5, 2, 2, 2>);
    static const int res2 = resident_workgroups(wgrad_pipe_kernel<6, 2, 2, 2>), res4 = resident_workgroups(wgrad_pipe_kernel<6, 2, 4, 2>);
    const int want = pipe_grid > 0 ? 256 * pipe_grid : (pack ? (obp == 4 ? res4p : res5p) : pipe_env == 2 ? res2 : res4);      // one resident round exactly
    const int Gp = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(kWgradBlocks, want), ceil_div(ceil_div(a.N, 16), 4)));
    if (pack && obp == 4) hipLaunchKernelGGL((wgrad_pipe_kernel<4, 2, 2, 2>), dim3(Gp), dim3(kBlock), 0, s, a);
    else if (pack) hipLaunchKernelGGL((wgrad_pipe_kernel<5, 2, 2, 2>), dim3(Gp), dim3(kBlock), 0, s, a);
    else if (pipe_env == 2) hipLaunchKernelGGL((wgrad_pipe_kernel<6, 2, 2, 2>), dim3(Gp), dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL((wgrad_pipe_kernel<6, 2, 4, 2>), dim3(Gp), dim3(kBlock), 0, s, a);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ceil_div(a.O * (a.I + 1), kReducePairs)), dim3(kReducePairs * kReduceSlices), 0, s, a.partial,
                       Gp, a.I, a.O, gw, gb, accumulate, pack ? pw : 0, pack ? pc : 0);
    return launch_status();
  } else if (ob <= 6 && ib <= 2) {
    if (wide_u == 2) MLQEM_WG((wgrad_mfma_kernel<6, 2, 2>), 1)
    else if (wide_pf) MLQEM_WG((wgrad_mfma_kernel<6, 2, kWgradUnroll, true>), 1)
    else MLQEM_WG((wgrad_mfma_kernel<6, 2>), 1)
  } else if (ob == 1) {
    MLQEM_WG((wgrad_mfma_kernel<1, 4>), (unsigned)ceil_div(ib, 4))
  } else if (ob > 6 && ib <= 2 && wide6_env) {
    // many outputs against a narrow x (Family B's first projection: gy[180]^T x[22]): six output tiles per workgroup -- a row's
    // gy is read as 384-byte pieces by ceil(ob / 6) workgroup columns instead of 128-byte pieces by ceil(ob / 2), and x is
    // re-read two times instead of six
    // (the software-pipelined form of this shape, wgrad_mfma_kernel<6, 2, 4, true>: 186 against 180 us at N = 706 k, round 6)
    MLQEM_WG((wgrad_mfma_kernel<6, 2>), (unsigned)ceil_div(ob, 6))
  } else if (ob > 2 && ib == 3) {
    // a few output tiles against three input tiles (Family B's second projection: gy[128]^T x[45]): x is re-read once per workgroup
    // column, so fewer, taller columns than the general form below: 4 x 3 tiles, 79 us at N = 353 k (2 x 4: 100 us, x read four times;
    // 8 x 3, x read once: 109 us at 170 registers)
    MLQEM_WG((wgrad_mfma_kernel<4, 3>), (unsigned)ceil_div(ob, 4))
  } else {
    MLQEM_WG((wgrad_mfma_kernel<2, 4>), (unsigned)(ceil_div(ob, 2) * ceil_div(ib, 4)))
  }
#undef MLQEM_WG
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ceil_div(a.O * (a.I + 1), kReducePairs)), dim3(kReducePairs * kReduceSlices), 0, s, a.partial,
                     G, a.I, a.O, gw, gb, accumulate, 0, 0);
  return launch_status();
}

extern "C" int mlqem_linear_wgrad_f32(const float* gy, int64_t ldgy, const float* x, int64_t ldx, float* gw, float* gb,
                                      int64_t N, int I, int O, int accumulate, void* workspace, size_t workspace_bytes,
                                      const int32_t* x_rows, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || I <= 0 || O <= 0 || !gw || ldgy < O || ldx < I) return MLQEM_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < mlqem_linear_wgrad_workspace_bytes(I, O)) return MLQEM_ERR_WORKSPACE;
  if (N > 0 && (!gy || !x)) return MLQEM_ERR_BAD_ARG;
  WgradArgs a{};
  a.gyp[0] = gy; a.ldgy[0] = ldgy; a.gn = 1; a.gw = O; a.gc = O;
  a.x = x; a.ldx = ldx; a.xrows = x_rows; a.partial = static_cast<float*>(workspace); a.N = N; a.I = I; a.O = O;
  return launch_wgrad(a, gw, gb, accumulate, as_stream(stream));
}

extern "C" int mlqem_linear_wgrad_parts_f32(const mlqem_col_parts* gy, const float* x, int64_t ldx, float* gw, float* gb,
                                            int64_t N, int I, int accumulate, void* workspace, size_t workspace_bytes,
                                            const int32_t* x_rows, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || I <= 0 || !gw || ldx < I || !parts_ok(gy, false)) return MLQEM_ERR_BAD_ARG;
  const int O = gy->count * gy->width;
  if (!workspace || workspace_bytes < mlqem_linear_wgrad_workspace_bytes(I, O)) return MLQEM_ERR_WORKSPACE;
  if (N > 0 && !x) return MLQEM_ERR_BAD_ARG;
  WgradArgs a{};
  for (int i = 0; i < gy->count; ++i) { a.gyp[i] = static_cast<const float*>(gy->ptr[i]); a.ldgy[i] = gy->ld[i]; }
  a.gn = gy->count; a.gw = gy->width; a.gc = gy->cols;
  a.x = x; a.ldx = ldx; a.xrows = x_rows; a.partial = static_cast<float*>(workspace); a.N = N; a.I = I; a.O = O;
  return launch_wgrad(a, gw, gb, accumulate, as_stream(stream));
}

extern "C" int mlqem_linear_wgrad_bf16_f32(const float* gy, int64_t ldgy, const float* x, int64_t ldx, float* gw, float* gb,
                                           int64_t N, int I, int O, int accumulate, void* workspace, size_t workspace_bytes,
                                           mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || I <= 0 || O <= 0 || !gw || ldgy < O || ldx < I) return MLQEM_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < mlqem_linear_wgrad_workspace_bytes(I, O)) return MLQEM_ERR_WORKSPACE;
  if (N > 0 && (!gy || !x)) return MLQEM_ERR_BAD_ARG;
  WgradArgs a{};
  a.gyp[0] = gy; a.ldgy[0] = ldgy; a.gn = 1; a.gw = O; a.gc = O;
  a.x = x; a.ldx = ldx; a.xrows = nullptr; a.partial = static_cast<float*>(workspace); a.N = N; a.I = I; a.O = O;
  hipStream_t s = as_stream(stream);
  const int64_t iters = ceil_div(std::max<int64_t>(N, 1), 32);
  const int G = (int)std::max<int64_t>(1, std::min<int64_t>(kWgradBlocks, ceil_div(iters, 4)));
  const int ob = (O + 15) / 16, ib = (I + 1 + 15) / 16;
  hipLaunchKernelGGL((wgrad_bf16_kernel<2, 2>), dim3(G, (unsigned)(ceil_div(ob, 2) * ceil_div(ib, 2))), dim3(kBlock), 0, s, a);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ceil_div(O * (I + 1), kReducePairs)), dim3(kReducePairs * kReduceSlices), 0, s, a.partial, G, I, O,
                     gw, gb, accumulate, 0, 0);
  return launch_status();
}

extern "C" int mlqem_linear_bwd_fused_f32(const float* gy, int64_t ldgy, const float* gb_src, int64_t ldgbs, const float* x,
                                          int64_t ldx, const float* w, int gate, float gate_scale, float* gx, int64_t ldgx,
                                          float* gw2, float* gb2, int64_t N, int I, int O, void* workspace,
                                          size_t workspace_bytes, mlqem_stream_t stream) {
  begin_launches();
  if (N < 0 || I <= 0 || O <= 0 || !gw2 || !gb2) return MLQEM_ERR_BAD_ARG;
  if (I > 12 || O > 12) return MLQEM_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < mlqem_linear_wgrad_workspace_bytes(I, kBwdFusedRows)) return MLQEM_ERR_WORKSPACE;
  if (N > 0 && (!gy || !x || !w || !gx)) return MLQEM_ERR_BAD_ARG;
  if (!gb_src) { gb_src = gy; ldgbs = ldgy; }
  auto rows_ok = [](const float* p, int64_t ld, int cols) { return ld >= (cols + 3) / 4 * 4 && ld % 4 == 0 && aligned_to(p, 16); };
  if (!rows_ok(gy, ldgy, O) || !rows_ok(gb_src, ldgbs, O) || !rows_ok(x, ldx, I) || !rows_ok(gx, ldgx, I)) return MLQEM_ERR_UNSUPPORTED;
  hipStream_t s = as_stream(stream);
  const int64_t tiles = ceil_div(std::max<int64_t>(N, 1), 16);
  const int G = (int)std::max<int64_t>(1, std::min<int64_t>(kWgradBlocks * 2, ceil_div(tiles, 4)));
  BwdFusedArgs a{gy, ldgy, gb_src, ldgbs, x, ldx, w, gate_scale, gate, gx, ldgx, static_cast<float*>(workspace), N, I, O};
  // the partial buffer is sized for kWgradBlocks workgroups of 25 x (I + 1) floats: cap the grid accordingly
  const int Gc = std::min(G, kWgradBlocks);
  hipLaunchKernelGGL(linear_bwd_fused_kernel, dim3(Gc), dim3(kBlock), 0, s, a);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ceil_div(kBwdFusedRows * (I + 1), kReducePairs)), dim3(kReducePairs * kReduceSlices), 0, s, a.partial, Gc, I,
                     kBwdFusedRows, gw2, gb2, 0, 0, 0);
  return launch_status();
}
