/*
 * mlqem_hip.h -- C ABI of libmlqem_hip.so, the MI355X (gfx950) implementation of the ml-qem
 * expectation-value-regressor hot path.
 *
 * The reference (qiskit-community/ml-qem) has no FFI of its own: its per-batch arithmetic is delegated to
 * torch_geometric / torch-sparse (requirements.txt:1-2).  Each entry point below therefore cites the
 * reference call site whose third-party op it replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name starts with h_ (host) ; the caller owns all buffers
 *   - fp32 data, int32 indices (int64 only where the reference hands over torch.long edge_index)
 *   - row-major matrices with an explicit leading dimension (elements, not bytes)
 *   - asynchronous on `stream` (a hipStream_t passed as void*); no allocation, no host sync, no global state
 *   - returns 0 on success, a negative MLQEM_ERR_* code otherwise (bad shape, unsupported width, launch error)
 *   - re-entrant across streams and threads
 *
 * Graph layout.  A batch of graphs is the disjoint union of its members: N nodes, graph g owns the node range
 * [graph_ptr[g], graph_ptr[g+1]).  Connectivity is held as TWO CSR structures over the edges that are not
 * self-loops: `in_ptr/in_src` groups edges by destination (forward aggregation), `out_ptr/out_dst` by source
 * (the transpose, used by the backward pass).  Self-loops are not stored as edges: `loops[i]` is the number of
 * (i,i) entries the caller's edge list had, and each layer adds the self term it needs analytically.
 */
#ifndef MLQEM_HIP_H
#define MLQEM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLQEM_OK 0
#define MLQEM_ERR_BAD_ARG (-1)
#define MLQEM_ERR_UNSUPPORTED (-2)
#define MLQEM_ERR_LAUNCH (-3)
#define MLQEM_ERR_WORKSPACE (-4)

typedef void* mlqem_stream_t; /* hipStream_t */

#define MLQEM_ABI_VERSION 42 /* bumped whenever a signature below changes; bindings compare it at load time */
int mlqem_abi_version(void);
const char* mlqem_error_string(int code);

/* ------------------------------------------------------------------------------------------------------
 * Graph structure.  Replaces what PyG derives inside every conv from `edge_index`
 * (docs/tutorials/gnn.py:104-113 passes the raw [2,E] tensor to TransformerConv / ASAPooling;
 *  blackwater/data/loaders/exp_val.py:33 AddSelfLoops; PyG gcn_norm / get_laplacian / degree).
 * ---------------------------------------------------------------------------------------------------- */

/* Bytes of scratch mlqem_csr_build needs for a graph with N nodes and E edge-list entries. */
size_t mlqem_csr_build_workspace_bytes(int64_t N, int64_t E);

/* edge_index: [2,E] int64 row-major (row 0 = source, row 1 = destination), self-loops allowed.
 * Outputs: in_ptr[N+1], in_src[E] (only the first in_ptr[N] entries are meaningful), out_ptr[N+1], out_dst[E],
 * out_eid[E] (optional: position in in_src[] of the edge each out_dst[] entry stands for; the backward kernels of the
 * edge-softmax ops read per-edge buffers through it), loops[N].  Inside a row the original edge order is kept. */
int mlqem_csr_build(const int64_t* edge_index, int64_t E, int64_t N, int32_t* in_ptr, int32_t* in_src,
                    int32_t* out_ptr, int32_t* out_dst, int32_t* out_eid, int32_t* loops, void* workspace,
                    size_t workspace_bytes, mlqem_stream_t stream);

/* Per-node normalisation scalars of the three Family-A convolutions (docs/tutorials/01_ngem.ipynb cell [9]):
 *   gcn_dinv[i]  = (indeg[i] + 1)^-1/2                     GCNConv, add_remaining_self_loops, degree by destination
 *   sage_rinv[i] = 1 / max(indeg[i] + loops[i], 1)         SAGEConv mean over the raw in-edges (self-loops count)
 *   cheb_dinv[i] = outdeg[i]^-1/2, 0 when outdeg[i] == 0   ChebConv sym normalisation, self-loops removed, degree by source
 * Any output pointer may be NULL. */
int mlqem_graph_norms(const int32_t* in_ptr, const int32_t* out_ptr, const int32_t* loops, int64_t N,
                      float* gcn_dinv, float* sage_rinv, float* cheb_dinv, mlqem_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * CSR aggregation ("scatter-add") -- the roofline kernel.  Replaces the gather x[edge_index[0]] + scatter(...,
 * reduce='sum'|'mean') inside GCNConv / SAGEConv / ChebConv / LEConv propagate (01_ngem.ipynb cell [9];
 * docs/tutorials/gnn.py:85,92 via ASAPooling) and, on the transposed CSR, their backward.
 *
 *   agg[i,:] = rscale[i] * sum_{e in [ptr[i],ptr[i+1])} cscale[idx[e]] * x[idx[e],:]  +  dself[i] * x[i,:]
 *   out[i,:] = act( alpha * agg[i,:] + beta * z[i,:] + bias[:] )
 *
 * cscale / rscale / dself / z / bias may be NULL (meaning 1 / 1 / 0 / absent / absent).
 * ell (optional, [N,2] int32 from mlqem_ell_from_csr for the SAME ptr/idx): the first two col[] entries of every
 * row, (-1 = no such edge, bit 31 of the first = row continues in col[]).  With it the row pointer -> col -> source
 * row dependent chain of the CSR walk shortens to ell -> source row for the 99.8 % of circuit-graph rows that have
 * at most two in-edges; rows with more fall back to ptr/idx, so the result is identical with or without it.
 * Vector width: when x, out (and z) own round_up(C,4) columns per row (leading dimensions multiples of 4, bases
 * 16-byte aligned) the kernel moves 16 bytes per lane and processes the pad columns along; columns never mix, so
 * the pads may hold anything.  16-byte accesses run ~1.4x the rate of 8-byte ones here: C = 12 takes less time
 * than C = 10, so the host lays every activation out with a padded leading dimension.
 * CONTRACT: with 16-byte-aligned rows the columns C .. round_up(C,4)-1 of `out` ARE WRITTEN (scratch values).  A
 * caller whose `out` is a column slice of a wider matrix must pass an unaligned base / a leading dimension that is
 * not a multiple of 4, or use a padded buffer; the Python binding refuses such slices (ops._owns_pad_columns).
 * seed_counter (may be NULL): a device-resident uint64 added (times an odd constant) to `seed` when the dropout mask is
 * drawn -- a launch captured in a hipGraph then draws a fresh mask on every replay as the caller bumps the counter.
 * ---------------------------------------------------------------------------------------------------- */
int mlqem_csr_aggregate_f32(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* idx, const int32_t* ell,
                            const float* cscale, const float* rscale, const float* dself, float alpha, float beta,
                            const float* z, int64_t ldz, const float* bias, int act, float drop_p, uint64_t seed,
                            const uint64_t* seed_counter, float* out, int64_t ldo, int64_t N, int C, mlqem_stream_t stream);

/* mlqem_csr_aggregate_f32 AND the pooled means of its output (mlqem_segment_pool_f32 of `out` with `pool_weights`) from ONE
 * pass: the aggregation's workgroups keep their rows of the output in LDS and leave per-(tile, graph) partial sums, added in
 * tile order by the pool's finish kernel (deterministic).  Replaces the last hidden layer of a Family A branch followed by
 * global_mean_pool (01_ngem.ipynb cell [9]; DESIGN section 3: the last conv is folded into the pool) without reading the
 * [N, C] activation a second time.  out_mean / out_wmean: [B, C] (either may be NULL).  Needs the ELL side table and rows of
 * round_up(C, 4) floats (MLQEM_ERR_UNSUPPORTED otherwise: call the two entry points).  gate_bits (optional,
 * mlqem_csr_aggregate_pool_gate_bytes(N, C) bytes, 16-byte aligned): the signs of `out` -- the ReLU / dropout gate
 * mlqem_segment_pool_bwd_f32 applies -- in the launch's own tiling: with R = the rows of a workgroup's tile (512 / ceil(C / 4))
 * and T = ceil(N / R) tiles, first T 16-byte records int32 (graph of the tile's first row, that graph's first row, the next
 * graph's first row, 0), then T x 32 uint64 words: the items (row, 16-byte column slice) of a tile are numbered row-major,
 * item i = 256 k + 64 w + l sets bit l of word (4 k + w) 4 + v of its tile iff out[row, 4 slice + v] > 0 (per-wave ballots: ABI
 * 25; a byte per item until 24); then, for C <= 16 (ABI 36), ceil(N / 2) uint32 words holding the same signs PER NODE, 16 bits each:
 * bit 4 slice + v of node i = out[i, 4 slice + v] > 0 -- what mlqem_pooled_grad_aggregate_f32 gathers.  With gate_bits `out` may be NULL -- a pooled activation whose only other reader is that gate
 * (the last hidden layer of a Family A branch) is then never written to memory.  workspace:
 * mlqem_csr_aggregate_pool_workspace_bytes(N, B, C). */
size_t mlqem_csr_aggregate_pool_workspace_bytes(int64_t N, int64_t B, int C);
size_t mlqem_csr_aggregate_pool_gate_bytes(int64_t N, int C);
int mlqem_csr_aggregate_pool_f32(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* idx, const int32_t* ell,
                                 const float* cscale, const float* rscale, const float* dself, float alpha, float beta,
                                 const float* z, int64_t ldz, const float* bias, int act, float drop_p, uint64_t seed,
                                 const uint64_t* seed_counter, float* out, int64_t ldo, int64_t N, int C,
                                 const float* pool_weights, const int32_t* graph_ptr, int64_t B, float* out_mean,
                                 int64_t ld_mean, float* out_wmean, int64_t ld_wmean, uint8_t* gate_bits, void* workspace,
                                 size_t workspace_bytes, mlqem_stream_t stream);

/* Segment max with the node itself included: out[i,:] = max(x[i,:], max_e x[idx[e],:])
 * (ASAPooling's scatter(..., reduce='max') after add_remaining_self_loops; docs/tutorials/gnn.py:85,92). */
int mlqem_csr_segment_max_f32(const float* x, int64_t ldx, const int32_t* ptr, const int32_t* idx, const int32_t* ell,
                              float* out, int64_t ldo, int64_t N, int C, mlqem_stream_t stream);

/* ell[i] = (col[ptr[i]], col[ptr[i]+1]) with the conventions above; ell: [N,2] int32, 8-byte aligned. */
int mlqem_ell_from_csr(const int32_t* ptr, const int32_t* idx, int64_t N, int32_t* ell, mlqem_stream_t stream);

/* BatchNorm1d in training mode over [N, C] rows: the bn1 / bn2 of MLP2 / MLP3 (docs/tutorials/mlp.py:45-66,87-108) and its
 * autograd; replaces torch.nn.BatchNorm1d's forward / backward kernels for that call site (C <= 256).
 *   forward:  mean[C], var[C] (biased), invstd[C] = rsqrt(var + eps), y = (x - mean) * invstd * gamma + beta
 *   backward: dbeta = sum dy, dgamma = sum dy * xhat, dx = gamma * invstd * (dy - dbeta / N - xhat * dgamma / N)
 * gamma / beta may be NULL (no affine).  Deterministic (per-workgroup partials summed in a fixed order, in double).  The
 * running statistics (momentum, unbiased variance) are the caller's: they are [C]-sized host-side bookkeeping. */
size_t mlqem_batch_norm_workspace_bytes(int64_t N, int C);
int mlqem_batch_norm_train_f32(const float* x, int64_t ldx, int64_t N, int C, const float* gamma, const float* beta, float eps,
                               float* y, int64_t ldy, float* mean, float* var, float* invstd, void* workspace,
                               size_t workspace_bytes, mlqem_stream_t stream);
int mlqem_batch_norm_train_bwd_f32(const float* dy, int64_t ldg, const float* x, int64_t ldx, int64_t N, int C,
                                   const float* gamma, const float* mean, const float* invstd, float* dx, int64_t lddx,
                                   float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                                   mlqem_stream_t stream);

/* ----------------------------------------------------------------------------------------------------
 * The two ends of a train step that are not model layers (docs/tutorials/__ml_models.py:100-187:
 * `loss = criterion(out, y); loss.backward(); optimizer.step()` with torch.nn.MSELoss and torch.optim.Adam).
 *
 * mlqem_mse_loss_grad_f32: *loss = mean over the N x C elements of (out - y)^2 and, when g is given,
 *   g = 2 (out - y) / (N C) -- the gradient `loss.backward()` hands to the model's output -- from one pass, one launch; g has
 *   g_rows >= N rows, the rows beyond N (filler rows of a padded batch, which the loss does not see) come out zero;
 *   per-workgroup partial sums added in index order by the workgroup that finishes last (deterministic).
 *   workspace: mlqem_mse_loss_workspace_bytes(); ticket: one zero-initialised unsigned the call leaves at zero.
 * mlqem_adam_step_f32: torch.optim.Adam(betas, eps; amsgrad = False, weight_decay = 0, maximize = False) on ONE flat
 *   buffer of n floats: step <- step + 1; m <- lerp(m, g, 1 - beta1); v <- beta2 v + (1 - beta2) g^2;
 *   p <- p - lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)  (betas arrive as doubles: 1 - beta and the bias
 *   corrections are formed in double, as torch forms them, the update itself in fp32).  `lr` and `step` (a float, as torch keeps
 *   it) live on the device, so the launch can sit in a captured hipGraph and a scheduler can change the rate between
 *   replays; ticket as above; bump_counter (optional, ABI 42): a device counter the launch's last workgroup increments -- the
 *   trainers' dropout step counter, so that a step needs no launch of its own for it.  Replaces torch's multi-tensor kernel, which runs a buffer of this path's size (1.8 k-180 k
 *   floats) on one workgroup.
 * ---------------------------------------------------------------------------------------------------- */
size_t mlqem_mse_loss_workspace_bytes(void);
int mlqem_mse_loss_grad_f32(const float* out, int64_t ldo, const float* y, int64_t ldy, float* g, int64_t ldg, int64_t N, int C,
                            int64_t g_rows, float* loss, void* workspace, size_t workspace_bytes, unsigned* ticket, mlqem_stream_t stream);
int mlqem_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const float* lr,
                        float* step, double beta1, double beta2, double eps, unsigned* ticket, uint64_t* bump_counter,
                        mlqem_stream_t stream);

/* nn.Sequential(Linear(I, H), [Dropout(p)], Linear(H, O)) as ONE launch per direction: y = dropout(x W1^T + b1) W2^T + b2.
 * Replaces `self.obs_seq(observable)` and `self.body_seq(merged)` of the graph models (01_ngem.ipynb cell [9]; docs/tutorials/
 * gnn.py:94-98,121): heads that see one row per circuit, where every separate GEMM / bias / dropout / weight-gradient launch is
 * pure launch latency.  w1: [H, I], w2: [O, H] row-major (torch's Linear.weight); H <= 16, O <= 8 (MLQEM_ERR_UNSUPPORTED beyond).
 *   forward:  hidden [N, H] = the dropped, rescaled first-layer output (NULL at inference), mask [N] = bit j set when hidden unit
 *             j of the row was kept (required with drop_p > 0 and hidden), y [N, O]; mask keyed by (seed [+ seed_counter], n H + j).
 *   backward: gw1 [H, I], gb1 [H], gw2 [O, H], gb2 [O] (gb1 / gb2 may be NULL) and, when gx is given, gx [N, I] = the gradient of
 *             the input; the row slices of a workgroup meet in LDS and are added in slice order (deterministic, one launch);
 *             workspace / ticket: reserved for a two-stage form (mlqem_seq2_backward_workspace_bytes returns 0; both may be NULL). */
int mlqem_seq2_forward_f32(const float* x, int64_t ldx, int64_t N, int I, const float* w1, const float* b1, int H, const float* w2,
                           const float* b2, int O, float drop_p, uint64_t seed, const uint64_t* seed_counter, float* hidden,
                           uint32_t* mask, float* y, int64_t ldy, mlqem_stream_t stream);
size_t mlqem_seq2_backward_workspace_bytes(int64_t N, int I, int H, int O);
int mlqem_seq2_backward_f32(const float* gy, int64_t ldgy, const float* x, int64_t ldx, int64_t N, int I, const float* w1, int H,
                            const float* w2, int O, const float* hidden, const uint32_t* mask, float drop_p, float* gx, int64_t ldgx,
                            float* gw1, float* gb1, float* gw2, float* gb2, void* workspace, size_t workspace_bytes, unsigned* ticket,
                            mlqem_stream_t stream);

/* gx[n,c] = (y[n,c] > 0) ? g[n,c] * scale : 0  -- backward of ReLU followed by inverted dropout, recovered from the
 * output y (an element that was clamped OR dropped has y == 0 and no gradient either way). */
int mlqem_relu_dropout_bwd_f32(const float* g, int64_t ldg, const float* y, int64_t ldy, float scale, float* gx,
                               int64_t ldgx, int64_t N, int C, mlqem_stream_t stream);

/* y = dropout(relu(x)) (inverted dropout, mask keyed by (seed [+ seed_counter], n*C + c)) and, when residual is given,
 * sum = y + residual in the same pass -- the element-wise tail of an MLP2 / MLP3 trunk layer (docs/tutorials/mlp.py:60-66:
 * x1 = drop(relu(bn1(fc1 x))), x2 = drop(relu(bn2(fc2 x1))), x1 + x2) as ONE launch.  y is what the backward needs
 * (mlqem_relu_dropout_bwd_f32 recovers the mask from it); sum may be NULL; seed_counter as in mlqem_csr_aggregate_f32. */
int mlqem_relu_dropout_f32(const float* x, int64_t ldx, float drop_p, uint64_t seed, const uint64_t* seed_counter,
                           const float* residual, int64_t ldr, float* y, int64_t ldy, float* sum, int64_t lds, int64_t N, int C,
                           mlqem_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Dense layers.  Replace torch.nn.Linear / torch_geometric.nn.Linear (docs/tutorials/gnn.py:94-98 body_seq,
 * docs/tutorials/mlp.py:18-108 MLP1/2/3, the per-node projections inside every conv).
 * ---------------------------------------------------------------------------------------------------- */

/* y[n,:] = drop( act( (x[n,:] @ W^T + b [+ y[n,:] if accumulate]) * rowscale[n] ) ),  W: [O,I] row-major as torch
 * stores it (transposed = 0), or the same with x @ W, W: [I,O] (transposed = 1: the data-gradient form gx = gy @ W).
 * b and rowscale may be NULL.  accumulate chains several calls into one sum of products (ChebConv's sum_k lins[k](T_k),
 * SAGEConv's lin_l(mean) + lin_r(x)); the activation belongs on the last call.  act bit 0 = ReLU; drop_p > 0 applies
 * inverted dropout keyed by (seed, n*O + o).  Column ranges: rowscale applies to outputs o < rs_cols and ReLU/dropout to
 * outputs o >= act_from (-1, -1 = every column), so that one launch can serve several layers that read the same input
 * rows (the three first-layer projections of GCN | Cheb | SAGE).  gate (may be NULL), applied last:
 * y[n,o] = gate[n,o] > 0 ? y[n,o] * gate_scale : 0 -- the backward of a ReLU/dropout epilogue whose output `gate` is this
 * layer's input, folded into the data-gradient GEMM that produces the incoming gradient (no separate masking pass).
 * x_rows (may be NULL): row n of x is row x_rows[n] of the buffer -- the first layers read their input rows straight
 * from the device-resident dataset through the batch's row map (src_node of mlqem_batch_assemble) instead of from a
 * gathered copy; only for calls the lean kernel can express (padded operands, I, O <= 64, no accumulate), else
 * MLQEM_ERR_UNSUPPORTED.
 * Runs on the f32-input matrix cores (v_mfma_f32_16x16x4_f32) for I <= 128. */
int mlqem_linear_f32(const float* x, int64_t ldx, const float* w, int transposed, const float* b,
                     const float* rowscale, float* y, int64_t ldy, int64_t N, int I, int O, int act, int accumulate,
                     float drop_p, uint64_t seed, int rs_cols, int act_from, const float* gate, int64_t ldgate,
                     float gate_scale, const int32_t* x_rows, mlqem_stream_t stream);

/* y = act(x @ W^T + b) (transposed = 0, W: [O, I]) or y = x @ W (transposed = 1, W: [I, O]: the data-gradient form) with
 * both operands rounded to bf16 (nearest-even) in registers and fp32 accumulation on v_mfma_f32_16x16x32_bf16: the "bf16
 * MFMA MLP head" option of the MLP regressors (docs/tutorials/mlp.py:18-108) for BASELINE.json's mixed-corpus
 * configuration.  x, W, y are fp32 in memory; b may be NULL; act bit 0 = ReLU; I <= 256. */
int mlqem_linear_bf16_f32(const float* x, int64_t ldx, const float* w, int transposed, const float* b, float* y, int64_t ldy,
                          int64_t N, int I, int O, int act, mlqem_stream_t stream);

/* gw[o,i] (+)= sum_n bf16(gy[n,o]) * bf16(x[n,i]);  gb[o] (+)= sum_n bf16(gy[n,o])  on the same matrix cores (32 rows per
 * MFMA, fp32 accumulation, deterministic two-stage reduction): with the two entry points above the bf16 head trains
 * end to end on v_mfma_f32_16x16x32_bf16.  workspace: mlqem_linear_wgrad_workspace_bytes(I, O). */
int mlqem_linear_wgrad_bf16_f32(const float* gy, int64_t ldgy, const float* x, int64_t ldx, float* gw, float* gb, int64_t N,
                                int I, int O, int accumulate, void* workspace, size_t workspace_bytes, mlqem_stream_t stream);

#define MLQEM_MAX_COL_PARTS 8

/* A matrix given as up to MLQEM_MAX_COL_PARTS COLUMN BLOCKS in separate buffers: block p is ptr[p][N, cols] (row stride ld[p]) and
 * stands at columns [p*width, p*width + cols) of the concatenation; columns cols..width-1 of a block are padding (read
 * as 0, written as scratch).  For mlqem_linear_parts_f32 width and every ld must be multiples of 4 and every ptr 16-byte
 * aligned (the padded activation layout). */
typedef struct mlqem_col_parts {
  int32_t count, width, cols, reserved;
  void* ptr[MLQEM_MAX_COL_PARTS];
  int64_t ld[MLQEM_MAX_COL_PARTS];
} mlqem_col_parts;

/* Projections over column blocks, the blocks on ONE side:
 *   transposed = 0, fan-out (x->count == 1):  Y_k = X W_k^T + b_k        W_k = w_blocks[k]: [y->cols, x->cols]
 *   transposed = 1, fan-in  (y->count == 1):  Y   = sum_k X_k W_k        W_k = w_blocks[k]: [x->cols, y->cols]
 * Weight blocks are row-major and unpadded, exactly as the layers store them; w_minus_blocks[k] (array or entries may
 * be NULL) is subtracted element-wise from W_k (the Clenshaw form of ChebConv multiplies by W_0 - W_2);
 * bias_blocks[k] (fan-out only, may be NULL): [y->cols]; rowscale_blocks[k] (fan-out only, may be NULL): [N], block k's
 * rows are multiplied by it after the bias (GCNConv's D^-1/2 pre-scaling).  One launch replaces
 *   - the per-term projections of ChebConv / SAGEConv that read the same input rows (lins[k](x), lin_l(x), lin_r(x);
 *     01_ngem.ipynb cell [9]) and
 *   - the sum of per-term data gradients (gx = sum_k g_k W_k)
 * without building the concatenation, whose wide rows would slow the aggregation gathers.  Concatenated width of the
 * input side <= 64 columns.  gate / gate_scale as in mlqem_linear_f32 (fan-in; gate rows padded like Y's). */
int mlqem_linear_parts_f32(const mlqem_col_parts* x, const float* const* w_blocks, const float* const* w_minus_blocks,
                           int transposed, const float* const* bias_blocks, const float* const* rowscale_blocks,
                           const mlqem_col_parts* y, int64_t N, const float* gate, int64_t ldgate, float gate_scale,
                           const int32_t* x_rows, mlqem_stream_t stream);

size_t mlqem_linear_wgrad_workspace_bytes(int I, int O);

/* gw[o,i] (+)= sum_n gy[n,o] * x[n,i] ;  gb[o] (+)= sum_n gy[n,o]  (gb may be NULL).  Matrix-core partial sums per
 * workgroup, then a fixed-order reduction: deterministic.  x_rows (may be NULL): row map for x as in mlqem_linear_f32. */
int mlqem_linear_wgrad_f32(const float* gy, int64_t ldgy, const float* x, int64_t ldx, float* gw, float* gb,
                           int64_t N, int I, int O, int accumulate, void* workspace, size_t workspace_bytes,
                           const int32_t* x_rows, mlqem_stream_t stream);

/* The same with gy given as column blocks (O = gy->count * gy->width rows of gw / entries of gb, padding rows are 0):
 * the weight gradients of all terms that share the input x in one pass over x. */
int mlqem_linear_wgrad_parts_f32(const mlqem_col_parts* gy, const float* x, int64_t ldx, float* gw, float* gb, int64_t N,
                                 int I, int accumulate, void* workspace, size_t workspace_bytes, const int32_t* x_rows,
                                 mlqem_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * The MLP regressor as one forward and one backward launch.  Replaces MLP1.forward and its autograd
 * (docs/tutorials/mlp.py:18-30 == blackwater/library/learning/mlp.py: fc2(relu(fc1(x))); trained by h10_mlp.ipynb cells
 * [10]-[13] on a plain feature matrix, so x needs no gradient).  I <= MLQEM_MLP1_MAX_IN inputs, H <= 128 hidden units,
 * O2 <= MLQEM_MLP1_MAX_OUT outputs; x rows padded to a multiple of 4 floats and 16-byte aligned (ldx % 4 == 0).
 *   forward : h = relu(x W1^T + b1) ; out = h W2^T + b2.  h_stash (optional, [N, MLQEM_MLP1_HIDDEN_PAD], 16-byte aligned)
 *             receives h for the backward: fp32 when bf16 == 0, bf16 (2 bytes per element) when bf16 != 0.
 *   backward: gW1 = gh^T x, gb1 = sum gh, gW2 = gout^T h, gb2 = sum gout with gh = (gout W2) o (h > 0), from ONE pass
 *             over x and the stash; per-workgroup partial sums, fixed-order second stage (deterministic).
 * bf16 == 0: exact fp32 on v_mfma_f32_16x16x4_f32.  bf16 != 0: every GEMM operand rounded to bf16 (nearest-even), fp32
 * accumulation on v_mfma_f32_16x16x32_bf16, the stash kept in bf16 (the "bf16 MFMA MLP head" of the mixed-corpus
 * configuration); gb2 is summed from the unrounded gout.  workspace (both calls; 16-byte aligned): mlqem_mlp1_workspace_bytes(I, O2) -- the
 * forward keeps the W1 fragment image its workgroups copy into LDS there, the backward its partial sums; a forward and a
 * backward on one stream may share it.
 *   The training cell's `loss = MSELoss()(out, y); loss.backward()` (docs/tutorials/__ml_models.py:148-160) folded in: with
 *   `target` [N, O2] the forward also writes gout = 2 (out - target) / (N O2) -- the gradient the backward call takes -- and
 *   leaves its per-workgroup sums of (out - target)^2 in the workspace; the NEXT backward call on the same workspace then
 *   writes *loss_out = mean (out - target)^2 (fixed summation order).  target == NULL / loss_out == NULL: neither. */
#define MLQEM_MLP1_HIDDEN_PAD 128
#define MLQEM_MLP1_MAX_OUT 4
#define MLQEM_MLP1_MAX_IN 175
size_t mlqem_mlp1_workspace_bytes(int I, int O2);
int mlqem_mlp1_forward(const float* x, int64_t ldx, const float* w1, const float* b1, const float* w2, const float* b2,
                       void* h_stash, float* out, int64_t ldo, int64_t N, int I, int H, int O2, int bf16, const float* target,
                       int64_t ldt, float* gout, int64_t ldg, void* workspace, size_t workspace_bytes, mlqem_stream_t stream);
int mlqem_mlp1_backward(const float* gout, int64_t ldg, const float* x, int64_t ldx, const void* h_stash, const float* w2,
                        float* gw1, float* gb1, float* gw2, float* gb2, int64_t N, int I, int H, int O2, int bf16,
                        float* loss_out, void* workspace, size_t workspace_bytes, mlqem_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * MLP2 / MLP3 with bf16 STORAGE (docs/tutorials/mlp.py:33-108 == blackwater/library/learning/mlp.py: fc -> BatchNorm1d -> ReLU ->
 * Dropout blocks with a residual; `mfma = "bf16"`, the "bf16 MFMA MLP head" of the mixed-corpus configuration).  Every
 * activation matrix -- layer outputs and what the backward re-reads -- is [N, MLQEM_MLP1_HIDDEN_PAD] bf16 (256-byte rows,
 * 16-byte aligned, columns beyond the layer's width zero); weights, BatchNorm statistics, parameter gradients and all sums are
 * fp32; GEMM operands are rounded to bf16 (nearest-even) and accumulated in fp32 on v_mfma_f32_16x16x32_bf16.  Dropout masks
 * are counter-based, keyed by (seed + *seed_counter, row * 128 + column), and recomputed in the backward.  All entry points
 * take workspace = mlqem_layer_workspace_bytes() (16-byte aligned); partial sums per workgroup, fixed-order second stages.
 *   gemm      : Y = X W^T + b (transposed = 0, W [U,K]) or Y = X W (+ add) (transposed = 1, W [K,U]: the data gradient);
 *               X fp32 [N,K <= 192] (ldx floats, ldx % 4 == 0) or bf16 [N,128] (K <= 128); Y bf16 [N,128] or fp32 [N,ldy];
 *               relu / drop_p (ABI 42): Y = dropout(relu(.)) in the epilogue, mlqem_layer_gemm_f32's mask -- a block without
 *               BatchNorm (MLP3's fc3, mlp.py:96-104) in one launch; its backward gates by Y > 0 (rowdot_bwd's gate_scale).
 *   colstats 0: batch statistics of y -> mean, biased var, invstd, scale = gamma invstd, shift = beta - mean scale (outputs
 *               are [128]; columns >= C come out 0), and, when running_mean / running_var [C] are given, BatchNorm1d's
 *               update of them in the same launch (running = (1 - momentum) running + momentum batch, variance unbiased,
 *               *num_batches_tracked += 1 when given; N >= 2).
 *   colstats 1: with gu = g o (y scale + shift > 0) o keep / (1 - p): dbeta = sum gu, dgamma = sum gu xhat, gs = gamma invstd,
 *               k1 = dbeta / N, k2 = dgamma / N  (g: bf16 [N,128], or fp32 [N,ldg32] when g32 != NULL).
 *   pointwise 0: out = dropout(relu?(y scale + shift)) (+ res);   1: out = gs (gu - k1 - xhat k2)   (BatchNorm backward).
 *   wgrad     : gw [U,K] = dY^T X, gb [U] = sum dY  (X fp32 [N,K <= 175] or bf16 [N,128], K <= 128).
 *   rowdot    : the final O <= 4 outputs, out = h w^T + b; rowdot_bwd: gh = g w (bf16), gw = g^T h, gb = sum g; with gate_scale > 0
 *               gh = (h > 0 ? gate_scale : 0) g w -- h = dropout(relu(u)) of a block without BatchNorm is its own gate, gh is then
 *               the gradient at u (ABI 25). */
size_t mlqem_layer_workspace_bytes(void);
int mlqem_layer_gemm_bf16(const void* x, int x_is_bf16, int64_t ldx, const float* w, int transposed, const float* b,
                          const void* add_bf16, void* y, int y_is_f32, int64_t ldy, int relu, float drop_p, uint64_t seed,
                          const uint64_t* seed_counter, int64_t N, int K, int U, void* workspace, size_t workspace_bytes,
                          mlqem_stream_t stream);
int mlqem_layer_colstats_bf16(int mode, const void* y, const void* g, const float* g32, int64_t ldg32, const float* scale,
                              const float* shift, const float* mean, const float* invstd, const float* gamma, const float* beta,
                              float eps, int relu, float drop_p, uint64_t seed, const uint64_t* seed_counter, int64_t N, int C,
                              float* o1, float* o2, float* o3, float* o4, float* o5, float* running_mean, float* running_var,
                              float momentum, int64_t* num_batches_tracked, void* workspace, size_t workspace_bytes,
                              mlqem_stream_t stream);
int mlqem_layer_pointwise_bf16(int op, const void* y, const void* g, const float* g32, int64_t ldg32, const void* res,
                               const float* scale, const float* shift, const float* mean, const float* invstd, const float* gs,
                               const float* k1, const float* k2, int relu, float drop_p, uint64_t seed,
                               const uint64_t* seed_counter, void* out, int64_t N, int C, mlqem_stream_t stream);
int mlqem_layer_wgrad_bf16(const void* dy, const void* x, int x_is_bf16, int64_t ldx, float* gw, float* gb, int64_t N, int K, int U,
                           void* workspace, size_t workspace_bytes, mlqem_stream_t stream);
int mlqem_layer_rowdot_bf16(const void* h, const float* w, const float* b, float* out, int64_t ldo, int64_t N, int C, int O,
                            mlqem_stream_t stream);
int mlqem_layer_rowdot_bwd_bf16(const float* g, int64_t ldg, const void* h, const float* w, void* gh, float gate_scale, float* gw,
                                float* gb, int64_t N, int C, int O, void* workspace, size_t workspace_bytes, mlqem_stream_t stream);

/* The same pipeline with fp32 STORAGE (`mfma = "f32"`: the reference's own arithmetic, docs/tutorials/mlp.py:33-108 in torch's
 * default dtype): every activation matrix is [N, MLQEM_MLP1_HIDDEN_PAD] fp32 (512-byte rows, 16-byte aligned, columns beyond the
 * layer's width zero), GEMM operands are NOT rounded (v_mfma_f32_16x16x4_f32).  colstats / pointwise / rowdot / rowdot_bwd:
 * the bf16 entry points' signatures and meaning with fp32 activation matrices for y, g, res, out, h, gh (g32 stays the narrow
 * [N, ldg32] form of the incoming gradient).  Same workspace, same counter-based dropout (keyed by the first of every four columns
 * a lane owns: masks differ from the bf16 pipeline's, forward and backward of one pipeline agree).
 *   gemm_f32  : Y = X W^T + b (transposed = 0, W [U,K]) or Y = X W (+ add [N,128]) (transposed = 1, W [K,U]); X fp32 [N, ldx],
 *               K <= 192 columns used (ldx % 4 == 0; an activation matrix: ldx = 128); Y an activation matrix (y_is_act, ldy = 128,
 *               every column written) or the first U columns of [N, ldy]; U <= 128.  relu / drop_p: the epilogue of a block
 *               without BatchNorm, Y = dropout(relu(.)) (counter-based mask; its backward gates by Y > 0: rowdot_bwd's gate_scale).
 *   wgrad_f32 : gw [U,K] = dY^T X, gb [U] = sum dY; dY an activation matrix, X fp32 [N, ldx], K <= 191. */
int mlqem_layer_gemm_f32(const float* x, int64_t ldx, const float* w, int transposed, const float* b, const float* add, float* y,
                         int y_is_act, int64_t ldy, int relu, float drop_p, uint64_t seed, const uint64_t* seed_counter, int64_t N,
                         int K, int U, void* workspace, size_t workspace_bytes, mlqem_stream_t stream);
int mlqem_layer_colstats_f32(int mode, const void* y, const void* g, const float* g32, int64_t ldg32, const float* scale,
                             const float* shift, const float* mean, const float* invstd, const float* gamma, const float* beta,
                             float eps, int relu, float drop_p, uint64_t seed, const uint64_t* seed_counter, int64_t N, int C,
                             float* o1, float* o2, float* o3, float* o4, float* o5, float* running_mean, float* running_var,
                             float momentum, int64_t* num_batches_tracked, void* workspace, size_t workspace_bytes,
                             mlqem_stream_t stream);
int mlqem_layer_pointwise_f32(int op, const void* y, const void* g, const float* g32, int64_t ldg32, const void* res,
                              const float* scale, const float* shift, const float* mean, const float* invstd, const float* gs,
                              const float* k1, const float* k2, int relu, float drop_p, uint64_t seed,
                              const uint64_t* seed_counter, void* out, int64_t N, int C, mlqem_stream_t stream);
int mlqem_layer_wgrad_f32(const float* dy, const float* x, int64_t ldx, float* gw, float* gb, int64_t N, int K, int U,
                          void* workspace, size_t workspace_bytes, mlqem_stream_t stream);
int mlqem_layer_rowdot_f32(const void* h, const float* w, const float* b, float* out, int64_t ldo, int64_t N, int C, int O,
                           mlqem_stream_t stream);
int mlqem_layer_rowdot_bwd_f32(const float* g, int64_t ldg, const void* h, const float* w, void* gh, float gate_scale, float* gw,
                               float* gb, int64_t N, int C, int O, void* workspace, size_t workspace_bytes, mlqem_stream_t stream);

/* Backward of a narrow hidden layer (I, O <= 12) in ONE pass over its operands:
 *   gx[n,:] = (x[n,:] > 0 ? gate_scale : 0) * (gy[n,:] @ W)   (gate != 0; plain gy @ W otherwise)      W: [O, I]
 *   gw2[0:O, :] = gy^T x,   gb2[12:12+O] = sum_n gb_src[n,:]   (gw2: [25, I], gb2: [25] -- rows 0..23 in the layout of
 *   mlqem_linear_wgrad_parts_f32 over the two 12-wide blocks [gy | gb_src]; gb_src = NULL means gy)
 *   gw2[24, 0:I] = sum_n gx[n,:]   (the bias gradient of the layer BELOW when an aggregation sits between the two layers, as
 *   with GCNConv: conv1's bias gradient is the column sum of what conv2's backward writes, so the first-layer weight-gradient
 *   pass reads one block less; gb2[24] is written as 0)
 * Replaces the autograd of torch_geometric's GCNConv linear (01_ngem.ipynb cell [9], conv2: gy = the transposed
 * aggregation of the incoming gradient, gb_src = that gradient itself, x = the previous activation whose ReLU/dropout
 * mask is handed over) with one read of gy, x and gb_src instead of two.  Padded 16-byte rows required.
 * workspace: mlqem_linear_wgrad_workspace_bytes(I, 25). */
int mlqem_linear_bwd_fused_f32(const float* gy, int64_t ldgy, const float* gb_src, int64_t ldgbs, const float* x, int64_t ldx,
                               const float* w, int gate, float gate_scale, float* gx, int64_t ldgx, float* gw2, float* gb2,
                               int64_t N, int I, int O, void* workspace, size_t workspace_bytes, mlqem_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Pooling over graphs.  Replaces global_mean_pool (docs/tutorials/gnn.py:114; 01_ngem.ipynb cell [9]).
 * ---------------------------------------------------------------------------------------------------- */
/* out_mean[g,:] = (1/n_g) sum_{r in graph g} x[r,:] and/or out_wmean[g,:] = (1/n_g) sum_r weights[r] * x[r,:] over the
 * contiguous row range [graph_ptr[g], graph_ptr[g+1]) of every graph (either output may be NULL, not both; weights may be
 * NULL = 1; an empty graph gives 0).  The weighted form is the last conv layer of a Family A branch folded into its
 * pool: mean_pool(P (h W^T)) = wmean(h) W^T with weights = P^T 1, the column sums of the layer's propagation matrix
 * (01_ngem.ipynb cell [9] puts no non-linearity between conv3 / cheb_conv2 / sage_conv2 and global_mean_pool).
 * Rows are tiled over workgroups (1024 rows each, whatever graph they belong to), per-(tile, graph) partial sums are
 * added in tile order by a second kernel: balanced for graphs of 7 ... 20 000 nodes, deterministic, no atomics.
 * workspace: mlqem_segment_pool_workspace_bytes(N, B, C) bytes. */
size_t mlqem_segment_pool_workspace_bytes(int64_t N, int64_t B, int C);
int mlqem_segment_pool_f32(const float* x, int64_t ldx, const float* weights, const int32_t* graph_ptr, int64_t N, int64_t B,
                           int C, float* out_mean, int64_t ld_mean, float* out_wmean, int64_t ld_wmean, void* workspace,
                           size_t workspace_bytes, mlqem_stream_t stream);
/* Backward: gx[r,:] = (g_mean[g,:] + weights[r] * g_wmean[g,:]) / n_g for every row r of graph g (either gradient may be
 * NULL); gate (may be NULL), applied last: gx = gate[r,:] > 0 ? gx * gate_scale : 0 -- the ReLU/dropout mask of the
 * pooled activation, see mlqem_linear_f32.  16-byte accesses when gx, gate AND the [B,C] gradients own round_up(C,4)
 * columns per row.  gate_bits (instead of gate; needs the 16-byte form and C <= 64): the same mask as the sign bits
 * mlqem_csr_aggregate_pool_f32 leaves for the same N and C (tile records + per-wave ballots, see there) -- the gate then costs
 * half a byte per slice instead of sixteen, read at wave-uniform addresses, and the activation need not exist in memory at all. */
int mlqem_segment_pool_bwd_f32(const float* g_mean, int64_t ld_gmean, const float* g_wmean, int64_t ld_gwmean,
                               const float* weights, const int32_t* graph_ptr, int64_t N, int64_t B, int C,
                               const float* gate, int64_t ldgate, float gate_scale, const uint8_t* gate_bits, float* gx,
                               int64_t ldgx, mlqem_stream_t stream);

/* mlqem_segment_pool_bwd_f32(gate_bits=) followed by mlqem_csr_aggregate_f32 of its result over the transposed structure, in ONE
 * launch that never gathers the [N, C] gradient: the first backward aggregation of a Family A branch (01_ngem.ipynb cell [9]; the
 * reference's autograd through global_mean_pool and the branch's last conv).  A source's row is computed from what defines it -- the
 * 16 gate bits the pooled forward left for the node, its pool weight t_j = weights[j], the aggregation's column scale cscale[j], and the two
 * gradient rows of the row's own graph (edges stay inside a graph):
 *     g[j, :]   = gate(j, :) ? ((g_mean[b, :] + t_j g_wmean[b, :]) / n_b) * gate_scale : 0
 *     out[i, :] = alpha * (rscale[i] * sum_e cscale[idx[e]] * g[idx[e], :] + dself[i] * g[i, :])
 * Both are written (g, optional, for the dense consumers: weight and bias gradients); results equal the two-launch form bit for bit.
 * mlqem_pooled_grad_colsum_f32: sum_j g[j, :] without g -- partial[mlqem_pooled_grad_colsum_groups(N)][round_up(C, 4)], added over the
 * groups by the caller (the bias gradient of a layer whose aggregation did not write g).
 * gate_bits: the buffer mlqem_csr_aggregate_pool_f32 filled for the same N and C.  Needs C <= 16 (mlqem_pooled_grad_aggregate_supported),
 * the ELL side table of (ptr, idx), and 16-byte rows everywhere (ld % 4 == 0, ld >= round_up(C, 4)). */
int mlqem_pooled_grad_aggregate_supported(int C);
int mlqem_pooled_grad_colsum_groups(int64_t N);
int mlqem_pooled_grad_colsum_f32(const uint8_t* gate_bits, const float* weights, const float* g_mean, int64_t ld_gmean, const float* g_wmean,
                                 int64_t ld_gwmean, const int32_t* graph_ptr, int64_t B, float gate_scale, int64_t N, int C, float* partial,
                                 mlqem_stream_t stream);
int mlqem_pooled_grad_aggregate_f32(const uint8_t* gate_bits, const float* weights, const float* cscale, const float* g_mean, int64_t ld_gmean,
                                    const float* g_wmean, int64_t ld_gwmean, const int32_t* graph_ptr, int64_t B, float gate_scale,
                                    const int32_t* ptr, const int32_t* idx, const int32_t* ell, const float* rscale,
                                    const float* dself, float alpha, float* out, int64_t ldo, float* g, int64_t ldg, int64_t N, int C,
                                    mlqem_stream_t stream);

/* What remains of a branch's last conv once it is folded into its pool: out[b, col[t]] (+)= P_t[b, :] . W_t (+ bias[col]),
 * t < n_terms (Family A: GCN one term, Cheb and SAGE two each -> three columns), and its backward:
 * gP_t[b, :] = gout[b, col[t]] * W_t;  gW[t, :] = sum_b gout[b, col[t]] * P_t[b, :];  gb[k] = sum_b gout[b, k].
 * P_t: [B, C] device matrices (row stride ldp[t]); W_t: [C]; bias[k]: one float on the device or NULL.
 * gP, ldgp: HOST arrays of n_terms device pointers / row strides. */
#define MLQEM_HEAD_MAX_TERMS 8
typedef struct mlqem_head_desc {
  int32_t n_terms, n_cols;
  const void* P[MLQEM_HEAD_MAX_TERMS];
  int64_t ldp[MLQEM_HEAD_MAX_TERMS];
  const void* W[MLQEM_HEAD_MAX_TERMS];
  int32_t col[MLQEM_HEAD_MAX_TERMS];
  const void* bias[MLQEM_HEAD_MAX_TERMS];
} mlqem_head_desc;
int mlqem_pooled_head_f32(const mlqem_head_desc* desc, int64_t B, int C, float* out, int64_t ldo, mlqem_stream_t stream);
int mlqem_pooled_head_bwd_f32(const mlqem_head_desc* desc, const float* gout, int64_t ldg, int64_t B, int C, float* const* gP,
                              const int64_t* ldgp, float* gW, float* gb, mlqem_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Batch assembly from a device-resident dataset.  Replaces torch_geometric.loader.DataLoader's collate
 * (Batch.from_data_list; call sites docs/tutorials/gnn.py:293-307, docs/tutorials/__ml_models.py:105-119), which
 * concatenates ~9 attributes of B Data objects on the host every step.
 *
 * Arena (whole dataset, G graphs): x[Ntot,F], nscal[Ntot,K] (per-node scalars, e.g. the three norms), a_gptr[G+1],
 * and the CSR arrays of mlqem_csr_build run over the whole arena (global node ids).
 * Selection: sel[B] graph ids (repeats allowed); b_nptr[B+1] / b_eptr[B+1] = prefix sums of the selected graphs'
 * node / edge counts (host knows them without a sync); Nb = b_nptr[B], Eb = b_eptr[B].
 * Outputs: the batch's x (xb may be NULL: no copy of the feature rows is made and the caller's first layers read them
 * from the arena through src_node, the x_rows argument of the dense entry points), nscal_b -- PLANAR [K, Nb], scalar k of all nodes contiguous, so each is usable as a vector
 * without a strided copy --, src_node[Nb] (arena row of every batch node), both CSR structures (node ids rebased to
 * the batch), loops and, when the arena's ELL side tables a_in_ell / a_out_ell (mlqem_ell_from_csr over the arena)
 * are given, the batch's side tables in_ell_b / out_ell_b [Nb,2] rebased the same way (NULL = not wanted).
 * derived_b (may be NULL; needs K >= 3 with nscal = [gcn_dinv, sage_rinv, cheb_dinv, ...] and a_loops): planar [3, Nb] =
 * gcn_dinv^2 (the (i,i) weight of the GCN operator), loops * sage_rinv (the self share of SAGE's mean), -cheb_dinv -- the
 * per-node scalars the layers derive, written in the same pass instead of by three element-wise kernels per batch.
 * Fixed-shape launches (hipGraph replay over size buckets): Nb must equal b_nptr[B], but Eb may be a CAPACITY >= b_eptr[B];
 * the kernels take the real edge total from b_eptr[B] on the device, so one captured launch serves every selection whose
 * totals fit the bucket (the caller pads the node count with a slice of an edgeless filler graph of the arena).
 * ---------------------------------------------------------------------------------------------------- */
int mlqem_batch_assemble(const float* x, int64_t ldx, int F, const float* nscal, int K, const int32_t* a_gptr,
                         const int32_t* a_in_ptr, const int32_t* a_in_src, const int32_t* a_out_ptr,
                         const int32_t* a_out_dst, const int32_t* a_out_eid, const int32_t* a_loops,
                         const int32_t* a_in_ell, const int32_t* a_out_ell, const int32_t* sel, const int32_t* b_nptr,
                         const int32_t* b_eptr, int64_t B, int64_t Nb, int64_t Eb, float* xb, int64_t ldxb,
                         float* nscal_b, float* derived_b, int32_t* src_node, int32_t* in_ptr_b, int32_t* in_src_b, int32_t* out_ptr_b,
                         int32_t* out_dst_b, int32_t* out_eid_b, int32_t* loops_b, int32_t* in_ell_b, int32_t* out_ell_b,
                         mlqem_stream_t stream);

/* The per-graph inputs of a batch in ONE launch (ABI 40): dst[k][b, :] = src[k][sel[b], :] for up to four fp32 row-major matrices
 * src[k] [G, width[k]] (the dataset's labels y, noisy values, circuit depths and observables; reference collate:
 * torch_geometric DataLoader at docs/tutorials/__ml_models.py:105-119 concatenates them per batch on the host) -- five torch gather
 * launches per step before, which a captured step of 32 four-qubit circuits (~80 launches of ~5 us) feels. */
int mlqem_gather_rows_f32(int count, const float* const* src, const int64_t* width, const int32_t* sel, int64_t B, float* const* dst,
                          mlqem_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Family B (docs/tutorials/gnn.py:70-276): TransformerConv attention and ASAPooling.  Forward kernels.
 * ---------------------------------------------------------------------------------------------------- */

/* TransformerConv(heads=H, concat=True, beta=False, edge_dim=None, root_weight=True) message + aggregate + skip
 * (gnn.py:80-91,104,109).  qkvs[N, 4*H*C] = [query | key | value | skip] from ONE fused projection.
 * out[i, h*C:(h+1)*C] = sum_e softmax_e( q_i.k_src / sqrt(C) ) v_src + skip_i ; softmax over the CSR in-edges of i
 * followed by loops[i] copies of the self-loop (loops may be NULL); denominator + 1e-16 as in PyG.  C <= 32. */
int mlqem_transformer_attention_f32(const float* qkvs, int64_t ld, const int32_t* in_ptr, const int32_t* in_src,
                                    const int32_t* loops, int64_t N, int H, int C, float* out, int64_t ldo,
                                    mlqem_stream_t stream);

/* ASAPooling steps 3-4 (gnn.py:85,92): out[i,:] = sum_e softmax_e(LeakyReLU(a_dst[i] + c_src[src_e])) x[src_e,:]
 * over the CSR in-edges of i plus its own self-loop (add_remaining_self_loops). */
int mlqem_csr_softmax_aggregate_f32(const float* x, int64_t ldx, const int32_t* in_ptr, const int32_t* in_src,
                                    const float* a_dst, const float* c_src, float negative_slope, int64_t N, int C,
                                    float* out, int64_t ldo, mlqem_stream_t stream);

/* ASAPooling step 5, LEConv(D->1) + sigmoid on per-node scalars pqr[N,3] = (lin1 x', lin2 x', lin3 x'):
 * fitness[i] = sigmoid( sum_{e in in(i) + self} (p[src_e] - q[i]) + r[i] ).
 * long_rows (ABI 41) != 0: the graph's rows are long (a coarsened graph): a 16-lane group per row instead of a thread (the sum of a
 * row's p values is then a tree sum, not the entries in order). */
int mlqem_leconv_fitness_f32(const float* pqr, const int32_t* in_ptr, const int32_t* in_src, int64_t N,
                             float* fitness, int long_rows, mlqem_stream_t stream);

/* out[p,:] = x[perm[p],:] * scale[perm[p]]  (x_out = x'[perm] * fitness[perm]; scale may be NULL). */
int mlqem_gather_scale_rows_f32(const float* x, int64_t ldx, const int32_t* perm, const float* scale, int64_t K, int C,
                                float* out, int64_t ldo, mlqem_stream_t stream);

/* Boundaries of the pooled batch on the device (ABI 37): new_graph_ptr[0] = 0, new_graph_ptr[g + 1] = new_graph_ptr[g] +
 * ceil((float)(graph_ptr[g + 1] - graph_ptr[g]) * ratio) -- the k of PyG's topk(x, ratio, batch) (torch_geometric
 * nn/pool/topk_pool.py, called from ASAPooling, reference docs/tutorials/gnn.py:85,92), evaluated in float32 as there.  Replaces
 * the host-side cumsum over per-graph sizes for batches whose sizes the host does not look at (size-stable captured steps).
 * graph_ptr [B + 1], new_graph_ptr [B + 1] int32; 0 < ratio <= 1.  One small launch, no workspace, asynchronous on `stream`. */
int mlqem_pool_keep_ptr(const int32_t* graph_ptr, int64_t B, float ratio, int32_t* new_graph_ptr, mlqem_stream_t stream);

/* ASAPooling step 6 (PyG topk(fitness, ratio, batch)): for graph g keep its new_graph_ptr[g+1]-new_graph_ptr[g]
 * nodes of largest fitness, listed by descending fitness (ties: lower index first), graphs in order.
 * max_graph_nodes: an upper bound on a graph's node count when the caller has one (0: none, N is used) -- it sizes the
 * index field of the sort key, and lets batches of large graphs (>= 1024 nodes on average) go through ONE device-wide radix
 * sort with the graph index in the key's top bits instead of a segmented sort that gives each graph to one workgroup.
 * slot (ABI 41; may be NULL): [N], mlqem_asap_slot_map's result for perm (slot[perm[p]] = p, -1 elsewhere) written by the same
 * launches -- every node is one of the sorted keys -- where the caller made it with a launch of its own. */
size_t mlqem_segment_topk_workspace_bytes(int64_t N, int64_t B);
int mlqem_segment_topk(const float* fitness, const int32_t* graph_ptr, const int32_t* new_graph_ptr, int64_t N,
                       int64_t B, int64_t K, int64_t max_graph_nodes, int32_t* perm, int32_t* slot, void* workspace,
                       size_t workspace_bytes, mlqem_stream_t stream);

/* ASAPooling step 7 (torch-sparse S^T A S, remove_diag, coo): only the PATTERN is consumed by the models.
 * Two hops with a sort-unique in between (counting every 3-step path explodes around 100-wire barriers):
 *   hop1_count: slot[N] (cluster id of each kept node, -1 elsewhere), offsets[K+1] = exclusive scan of the (cluster p,
 *               node v) candidates, v reachable in one step from a member of p; offsets[K] = total, read back by the caller
 *   hop1_fill:  keys = p << 32 | v                 -> mlqem_sort_unique_u64 -> M distinct pairs
 *   hop2_count / hop2_fill over those pairs: keys = p << 32 | q for every kept w in N+[v], q = slot[w] != p
 *               -> mlqem_sort_unique_u64 -> the pooled edges in row-major (p, q) order, as SparseTensor.coo() lists them
 *   mlqem_keys_to_edge_index: [2,E] int64 (src = p, dst = q); feed it to mlqem_csr_build. */
size_t mlqem_asap_coarsen_workspace_bytes(int64_t K);   /* for a count call over K items (hop1: clusters, hop2: pairs) */
int mlqem_asap_hop1_count(const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                          const int32_t* perm, int64_t N, int64_t K, int32_t* slot, int64_t* offsets, void* workspace,
                          size_t workspace_bytes, mlqem_stream_t stream);
int mlqem_asap_hop1_fill(const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                         const int32_t* perm, const int64_t* offsets, int64_t K, uint64_t* keys, mlqem_stream_t stream);
int mlqem_asap_hop2_count(const uint64_t* pairs, int64_t M, const int32_t* out_ptr, const int32_t* out_dst,
                          const int32_t* slot, int64_t* offsets, void* workspace, size_t workspace_bytes,
                          mlqem_stream_t stream);
int mlqem_asap_hop2_fill(const uint64_t* pairs, int64_t M, const int32_t* out_ptr, const int32_t* out_dst,
                         const int32_t* slot, const int64_t* offsets, uint64_t* keys, mlqem_stream_t stream);
size_t mlqem_sort_unique_u64_workspace_bytes(int64_t T);
int mlqem_sort_unique_u64(const uint64_t* keys, int64_t T, uint64_t* out_keys, int64_t* out_count, void* workspace,
                          size_t workspace_bytes, mlqem_stream_t stream);
int mlqem_keys_to_edge_index(const uint64_t* keys, int64_t E, int64_t* edge_index, mlqem_stream_t stream);

/* Coarsened connectivity of LARGE graphs (100-qubit circuits pool to thousands of clusters; their second pooling runs on
 * graphs with hub clusters of hundreds of neighbours, where the two-hop path above sorts ~50 candidate keys per distinct
 * edge).  One wave per cluster p builds the row's reach as bitsets in LDS -- nodes v in N+[N-[c_p]], then clusters
 * q = slot[w], w in N+[v] -- and the transposed row the same way from N-[N-[c_p]], so duplicates collapse without a sort or a
 * global atomic and rows come out in ascending order; both are kept as bit matrices [K][ceil(kmax / 32)] in the workspace.  Two calls, ONE host read between them:
 *   rows_count: slot[N], new_in_ptr[K + 1], new_out_ptr[K + 1]; the caller reads new_out_ptr[K] (the edge total E);
 *   rows_fill:  new_in_src[E], new_out_dst[E], new_out_eid[E] -- the arrays mlqem_csr_build yields from the two-hop path's
 *               edge list -- from the SAME workspace, untouched in between.
 * nmax / kmax: largest graph / largest pooled graph of the batch; nmax + 2 kmax + 96 <= mlqem_asap_coarsen_rows_max_bits()
 * (131 072: 64 KB of LDS for the four waves of a workgroup), else MLQEM_ERR_UNSUPPORTED.  Replaces the same
 * ASAPooling.forward lines as the entry points above (gnn.py:105-107,110-112). */
size_t mlqem_asap_coarsen_rows_workspace_bytes(int64_t K, int kmax);

/* slot[N] alone: slot[perm[p]] = p for the K kept centres, -1 elsewhere (every coarsening entry point writes it too).
 * ASAPooling's backward needs it (x_out = x'[perm] * fitness[perm], gnn.py:105-107,110-112) even when nobody reads the
 * coarsened connectivity -- the second pooling of every reference model is followed by global_mean_pool (gnn.py:112-114)
 * -- so the host can skip the coarsening and call only this. */
int mlqem_asap_slot_map(const int32_t* perm, int64_t N, int64_t K, int32_t* slot, mlqem_stream_t stream);
/* The same in ONE launch given the graphs' node ranges before / after pooling (ABI 40; graph g's centres are its own nodes). */
int mlqem_asap_slot_map_graphs(const int32_t* perm, const int32_t* graph_ptr, const int32_t* new_graph_ptr, int64_t B, int64_t N,
                               int64_t K, int32_t* slot, mlqem_stream_t stream);
int mlqem_asap_coarsen_rows_max_bits(void);
int mlqem_asap_coarsen_rows_count(const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                                  const int32_t* graph_ptr, const int32_t* new_graph_ptr, const int32_t* perm, int64_t N,
                                  int64_t K, int64_t B, int nmax, int kmax, int32_t* slot, int32_t* new_in_ptr,
                                  int32_t* new_out_ptr, void* workspace, size_t workspace_bytes, mlqem_stream_t stream);
int mlqem_asap_coarsen_rows_fill(const int32_t* new_graph_ptr, int64_t K, int64_t B, int kmax, const int32_t* new_in_ptr,
                                 const int32_t* new_out_ptr, int32_t* new_in_src, int32_t* new_out_dst, int32_t* new_out_eid,
                                 const void* workspace, size_t workspace_bytes, mlqem_stream_t stream);

/* The same arrays from SORTED LISTS instead of bit matrices (round 4; the default for large graphs): nothing dense is written
 * to or read from global memory, and the three hops of a cluster's reach are split into per-NODE lists built once --
 * C(v): the kept centres among N+[v]; R(u) / R'(u): the C lists of N+[u] / N-[u] one after the other -- so that a hub's list
 * (a barrier: ~300 entries) is formed once and read, coalesced, by every cluster that contains the hub.  Persistent waves OR a
 * cluster's few lists into two LDS bitsets (row and transposed row), read them out in ascending order and clear them on the
 * way; a structural bound per row (sums of two-hop degree sums, clamped to k_g - 1) places every row in a scratch; degrees ->
 * scans -> the CSR pointers; the fill pass copies the lists to their place and finds every out-entry's twin in the in-row by
 * binary search, one thread per entry.
 *   lists_caps:  totals[4] (device int64) = the totals of the row bounds (out, in) and of the per-node list sizes (out, in):
 *                `capacity` must cover all four when the caller has no structural bound of its own (one 32-byte read);
 *                workspace of mlqem_asap_coarsen_lists_workspace_bytes(N, K, 0, 0);
 *   lists_count: slot[N], new_in_ptr[K + 1], new_out_ptr[K + 1]; E = stored edges of the input structure (or a bound);
 *   lists_fill:  new_in_src / new_out_dst / new_out_eid, each holding edge_capacity entries (new_out_ptr[K] <= edge_capacity
 *                <= capacity), from the SAME workspace; new_out_eid may be NULL (no link pass: the recomputed backward forms
 *                need no out_eid); *overflow (device, may be NULL) = 1 if `capacity` was too small.
 * capacity < 2^32 (places inside the per-node records are 32 bits), else MLQEM_ERR_UNSUPPORTED.
 * kmax <= mlqem_asap_coarsen_lists_max_k() (65 535 clusters per pooled graph), else MLQEM_ERR_UNSUPPORTED.
 * ABI 41: slot_ready != 0: slot[] already holds mlqem_asap_slot_map's result (the caller made it for the backward: no fill, no map
 * launch here); new_loops (may be NULL): [K], zeroed (the coarsened graph lists no self-loops).
 * Replaces the same ASAPooling.forward lines (gnn.py:105-107,110-112; PyG semantics: SURVEY appendix B.2 step 7). */
size_t mlqem_asap_coarsen_lists_workspace_bytes(int64_t N, int64_t K, int64_t E, int64_t capacity);
int mlqem_asap_coarsen_lists_max_k(void);
int mlqem_asap_coarsen_lists_caps(const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                                  const int32_t* new_graph_ptr, const int32_t* perm, int64_t N, int64_t K, int64_t B,
                                  int64_t* totals, void* workspace, size_t workspace_bytes, mlqem_stream_t stream);
int mlqem_asap_coarsen_lists_count(const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                                   const int32_t* graph_ptr, const int32_t* new_graph_ptr, const int32_t* perm, int64_t N,
                                   int64_t K, int64_t B, int64_t E, int kmax, int64_t capacity, int32_t* slot, int slot_ready,
                                   int32_t* new_in_ptr, int32_t* new_out_ptr, int32_t* new_loops, void* workspace, size_t workspace_bytes,
                                   mlqem_stream_t stream);
int mlqem_asap_coarsen_lists_fill(int64_t N, int64_t K, int64_t E, int64_t capacity, const int32_t* new_in_ptr,
                                  const int32_t* new_out_ptr, int32_t* new_in_src, int32_t* new_out_dst, int32_t* new_out_eid,
                                  int64_t edge_capacity, int32_t* overflow, void* workspace, size_t workspace_bytes,
                                  mlqem_stream_t stream);

/* Coarsened connectivity WITHOUT host read-backs, for batches whose graphs all pool to at most
 * mlqem_asap_coarsen_dense_max_k() clusters (512): the pooled adjacency of every graph is built as a k_g x k_g bit
 * matrix in LDS by one workgroup per graph (idempotent atomicOr: order-independent), device scans of the row / column
 * popcounts give new_in_ptr / new_out_ptr directly, and a second kernel lists rows and columns in ascending order --
 * the same arrays mlqem_csr_build yields from the two-hop path's sorted edge list.  graph_ptr / new_graph_ptr: node
 * ranges of the graphs before / after pooling; perm: the kept centres, graph by graph (mlqem_segment_topk); kmax >= the
 * largest k_g (host knows it: k_g = ceil(ratio * n_g); a bound will do).  new_in_src / new_out_dst / new_out_eid must hold
 * sum_g k_g (k_g - 1) entries; only the first new_in_ptr[K] are written.  new_loops[K] = 0 (no diagonal).
 * slot_ready (ABI 40): `slot` already holds mlqem_asap_slot_map(perm) -- the call then makes three launches (bit matrices, one
 * scan of both degree vectors, the fill) where ABI <= 39 made eleven. */
int mlqem_asap_coarsen_dense_max_k(void);
size_t mlqem_asap_coarsen_dense_workspace_bytes(int64_t B, int64_t K, int kmax);
int mlqem_asap_coarsen_dense(const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                             const int32_t* graph_ptr, const int32_t* new_graph_ptr, const int32_t* perm, int64_t N, int64_t K,
                             int64_t B, int kmax, int32_t* slot, int slot_ready, int32_t* new_in_ptr, int32_t* new_in_src,
                             int32_t* new_out_ptr, int32_t* new_out_dst, int32_t* new_out_eid, int32_t* new_loops,
                             void* workspace, size_t workspace_bytes, mlqem_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Family B backward.  Edge-softmax gradients are split into a destination-side pass and a source-side pass.  No atomics; every
 * gradient row is written once.  Two forms of the hand-over between the passes:
 *   stored      (out_eid given): the destination side writes per-edge buffers in in-CSR order (E entries, then one self-loop
 *               entry per node at E + row) and the source side reads them through out_eid;
 *   recomputed  (out_eid == NULL, round 4): nothing is kept per edge; the source side recomputes every weight from the
 *               per-row statistics of the forward (and one more per-row number the destination side files), the way
 *               attention backward passes are usually written.  It needs no map from out-entries to in-CSR positions --
 *               ASAPooling's coarsened graphs come without one (linking their 9.7 M entries cost 0.55 ms of an 8.7 ms step) --
 *               and reads [N]-sized arrays where the stored form reads [E]-sized ones.
 * ---------------------------------------------------------------------------------------------------- */

/* Training forward of mlqem_transformer_attention_f32: same result, plus attn_out (the sum before the skip term; since ABI 25
 * written only for rows of MORE THAN FOUR entries, in-edges + self entry: the backward forms what it needs of a shorter row --
 * g . attn_out -- from the row's own entries, so on circuit DAGs attn_out is neither written nor read but for barrier rows;
 * the buffer is opaque to callers, hand it to the backward as it is) and the softmax statistics stat_m / stat_den [N,H]; drop_p > 0 drops attention weights (TransformerConv(dropout=0.1),
 * gnn.py:83,90) with a mask keyed by (seed, in-CSR position, head) -- or, pair_key != 0, by (seed, destination, head, source):
 * the same draw from either end of an edge, for graphs WITHOUT parallel edges (they would share a draw); the recomputed
 * backward needs it.  seed_counter (may be NULL): a device-resident step counter mixed into the seed, so that a launch captured
 * in a hipGraph draws a fresh mask per replay (the backward entry point must be given the same seed, counter and key form).
 * in_ell (optional, ABI 25): the [N,2] side table of mlqem_ell_from_csr over the same in-CSR -- rows of at most two in-edges then
 * reach their key / value rows without the ptr -> idx round trip (same result bit for bit).
 * head_pitch (ABI 25; 0 = C): the channel pitch of a head inside the four parts of qkvs -- and of gqkvs in the backward --
 * i.e. qkvs is [N, 4 H head_pitch] with part p of head h at column (p H + h) head_pitch and zeros in the head_pitch - C pad
 * channels (a projection whose weight and bias rows are padded the same way produces exactly that).  16 for the reference's
 * 15 channels makes every gathered key / value / query segment an aligned 64-byte piece (forward -8 %, backward -7..-17 % on
 * the 100-qubit graphs for 7 % more bytes); out, attn_out and g stay compact [N, H C]; the backward writes zeros into gqkvs' pads. */
int mlqem_transformer_attention_train_f32(const float* qkvs, int64_t ld, const int32_t* in_ptr, const int32_t* in_src,
                                          const int32_t* loops, int64_t N, int64_t E, int H, int C, float drop_p,
                                          uint64_t seed, const uint64_t* seed_counter, int pair_key, const int32_t* in_ell,
                                          int head_pitch, float* out, int64_t ldo, float* attn_out, int64_t lda, float* stat_m,
                                          float* stat_den, mlqem_stream_t stream);

/* gqkvs[N, 4HC] = gradient of [query | key | value | skip] given g = dL/d out.  Stored form: edge_al / edge_gs: scratch
 * [(E+N)*H].  Recomputed form (out_eid == NULL): edge_al: scratch [4*N*H] (16-byte aligned), edge_gs unused (may be NULL); with drop_p > 0 it needs
 * pair_key != 0 (MLQEM_ERR_BAD_ARG otherwise). */
int mlqem_transformer_attention_bwd_f32(const float* qkvs, int64_t ld, const float* g, int64_t ldg,
                                        const float* attn_out, int64_t lda, const float* stat_m, const float* stat_den,
                                        const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr,
                                        const int32_t* out_dst, const int32_t* out_eid, const int32_t* loops, int64_t N,
                                        int64_t E, int H, int C, float drop_p, uint64_t seed, const uint64_t* seed_counter,
                                        int pair_key, int head_pitch, float* gqkvs, int64_t ldq, float* edge_al, float* edge_gs,
                                        mlqem_stream_t stream);

/* Backward of mlqem_csr_softmax_aggregate_f32: gx (+)= d/dx, g_a[N] = d/d a_dst, g_c[N] = d/d c_src.
 * xnew = the forward output, gnew its gradient.  Stored form: edge_al / edge_gp: scratch [E+N].  Recomputed form
 * (out_eid == NULL; C <= 128): edge_al: scratch [4 N], 16-byte aligned; edge_gp unused (may be NULL).
 * tie_count (may be NULL; C <= 128): with xmax = the segment max of x over the same entries (mlqem_csr_segment_max_f32, what
 * ASAPooling computes from the same x: gnn.py:105-107), tie_count[i, c] = the number of entries of row i (sources and i itself) whose
 * value equals xmax[i, c] -- the destination-side walk holds every source row anyway, and mlqem_csr_segment_max_bwd_f32 then
 * needs no walk of its own to split a maximum's gradient among ties.
 * gx_rank1 (may be NULL; C <= 128): a [C] vector r; gx additionally receives g_c[j] * r -- the gradient through the source score
 * c_j = x_j . r (ASAPooling's att on the source half, gnn.py:105-107), which otherwise is a read-modify-write pass over gx.
 * fuse_max_col (may be NULL; stored form with tie_count only; ABI 34): a [C] vector w; the source-side walk then also adds the
 * backward of the segment max whose gradient is g_a (x) w -- gx[j, c] += sum over the destinations i of j, and j itself, with
 * x[j, c] == xmax[i, c] of g_a[i] w[c] / max(tie_count[i, c], 1) -- and mlqem_csr_segment_max_bwd_f32 is not called at all. */
int mlqem_csr_softmax_aggregate_bwd_f32(const float* x, int64_t ldx, const float* xnew, int64_t ldn, const float* gnew,
                                        int64_t ldg, const int32_t* in_ptr, const int32_t* in_src,
                                        const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_eid,
                                        const float* a_dst, const float* c_src, float negative_slope, int64_t N,
                                        int64_t E, int C, int accumulate, float* gx, int64_t ldgx, float* g_a,
                                        float* g_c, float* edge_al, float* edge_gp, const float* xmax, int64_t ldm,
                                        float* tie_count, int64_t ldt, const float* gx_rank1, const float* fuse_max_col,
                                        mlqem_stream_t stream);

/* Backward of mlqem_csr_segment_max_f32, ACCUMULATING into gx: the gradient of a row's maximum goes to the entries
 * (sources or the row itself) whose value equals it, split evenly among ties (torch scatter_reduce(amax) rule).
 * gmax_row [N] / gmax_col [C] (may be NULL): gmax = gmax_row (x) gmax_col, never formed (needs tie_count; gmax is then unused).
 * gshare: scratch [N, lds >= C].  tie_count (may be NULL): the counts mlqem_csr_softmax_aggregate_bwd_f32 left; the
 * destination-side pass over the in-edges is then an elementwise division. */
int mlqem_csr_segment_max_bwd_f32(const float* x, int64_t ldx, const float* xmax, int64_t ldm, const float* gmax,
                                  int64_t ldg, const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr,
                                  const int32_t* out_dst, int64_t N, int C, float* gx, int64_t ldgx, float* gshare,
                                  int64_t lds, const float* tie_count, int64_t ldt, const float* gmax_row, const float* gmax_col,
                                  mlqem_stream_t stream);

/* Backward of x_out = x'[perm] * fitness[perm] over all N rows (slot[i] = cluster id or -1 from
 * mlqem_asap_hop1_count): gxnew[i,:] = gout[slot[i],:] * fitness[i] (0 for dropped rows), gfit[i] = gout[slot[i]].x'[i]. */
int mlqem_gather_scale_rows_bwd_f32(const float* gout, int64_t ldgo, const float* xnew, int64_t ldn,
                                    const float* fitness, const int32_t* slot, int64_t N, int C, float* gxnew,
                                    int64_t ldgn, float* gfit, mlqem_stream_t stream);

/* ABI 41: mlqem_gather_scale_rows_bwd_f32 as two launches around the fitness backward (ASAPooling.backward, gnn.py:85,92): the first
 * stores gfit[row] = g_out[slot[row]] . x'[row] (0 for rows that were not kept) only; the second forms
 * gxnew[row] = (kept ? g_out[slot[row]] fitness[row] : 0) + sum_{t < K} g3[row, t] w3[t, :]  (K <= 3; w3 compact [K, C]) in ONE store --
 * the gradient through x_out = x'[perm] f[perm] plus the gradient g_pqr W3 through pqr = x' W3^T + b3, which was a read-modify-write
 * GEMM over gxnew.  Rows of at most 64 channels in the padded layout (16-byte rows); MLQEM_ERR_UNSUPPORTED otherwise. */
int mlqem_gather_rows_dot_f32(const float* gout, int64_t ldgo, const float* xnew, int64_t ldn, const int32_t* slot, int64_t N, int C,
                              float* gfit, mlqem_stream_t stream);
int mlqem_scatter_scale_rank_f32(const float* gout, int64_t ldgo, const float* fitness, const int32_t* slot, const float* g3, int64_t ldg3,
                                 const float* w3, int K, int64_t N, int C, float* gxnew, int64_t ldgn, mlqem_stream_t stream);

/* The three tiny weight gradients at the end of ASAPooling's backward in ONE pass (ABI 40): up to three weighted column sums over the
 * same N rows.  Term t: weight columns g[t] [N, ldg[t]] (k[t] <= 3 of them), matrix x[t] [N, ldx[t]] with D columns in 16-byte rows;
 * out [sum k, D] row-major in term order = g[t]^T x[t], bias [sum k] = the column sums of the weights.  Replaces the autograd of
 * ASAPooling's `lin` / `att` / `gnn_score` projections (torch_geometric ASAPooling.forward, called at docs/tutorials/gnn.py:105-112):
 * three [N, k] x [N, D] weight-gradient GEMMs.  Deterministic (fixed-order partial sums).  workspace:
 * mlqem_rank_grad_workspace_bytes(D); D <= 256. */
size_t mlqem_rank_grad_workspace_bytes(int D);
int mlqem_rank_grad_f32(int terms, const float* const* x, const int64_t* ldx, const float* const* g, const int64_t* ldg, const int* k,
                        int64_t N, int D, float* out, float* bias, void* workspace, size_t workspace_bytes, mlqem_stream_t stream);

/* Backward of mlqem_leconv_fitness_f32 onto pqr[N,3]. */
int mlqem_leconv_fitness_bwd_f32(const float* gfit, const float* fitness, const int32_t* in_ptr, const int32_t* out_ptr,
                                 const int32_t* out_dst, int64_t N, float* gpqr, mlqem_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Row order of a coarsened graph (csrc/row_order.hip).  The graph ASAPooling coarsens out of a large circuit
 * (docs/tutorials/gnn.py:85,92,104-112: TransformerConv 2 and ASAPooling 2 of every reference GNN run on it) is a union of dense
 * blocks: rows whose centres are close in program order share nearly all their sources.  The dense-block plans below take their
 * rows in that order.  (ABI <= 38 also exported LDS-staged "tile" kernels built on the same order -- mlqem_tile_plan_build,
 * mlqem_tile_attention_*, mlqem_tile_asap_scores_*: measured slower than the per-edge kernels, superseded by the dense blocks,
 * removed with ABI 39.)
 * ---------------------------------------------------------------------------------------------------- */
/* order[new_graph_ptr[g] + r] = the cluster of graph g whose centre is the r-th kept node of the graph in node order
 * (= program order of a circuit); slot[] from mlqem_asap_slot_map.  One workgroup per graph. */
int mlqem_tile_order_by_position(const int32_t* slot, const int32_t* graph_ptr, const int32_t* new_graph_ptr, int64_t num_graphs,
                                 int32_t* order, mlqem_stream_t stream);

/* ASAPooling's forward up to the fitness projections in one pass, for graphs of short rows (C <= 64): per row the segment max over
 * its in-entries and itself (xmax), the composed score a_dst = w_comp . xmax + b_comp, c_src = att_x . x, the score softmax + cluster
 * sum (xnew) and pqr[N, 3] = xnew W3^T + b3.  Stands for mlqem_csr_segment_max_f32 + mlqem_linear_f32 x 3 +
 * mlqem_csr_softmax_aggregate_f32 of ASAPooling.forward (docs/tutorials/gnn.py:85,92); same outputs. */
int mlqem_asap_scores_fused_f32(const float* x, int64_t ldx, const int32_t* in_ptr, const int32_t* in_src, const float* w_comp,
                                const float* b_comp, const float* att_x, const float* w3, const float* b3, float negative_slope,
                                int64_t N, int C, float* xmax, int64_t ldm, float* a_dst, float* c_src, float* xnew, int64_t ldn,
                                float* pqr, mlqem_stream_t stream);

/* The q / k / v / skip projection of a TransformerConv (docs/tutorials/gnn.py:80-91) writes a head's C channels at a pitch of
 * `pitch` floats when its weight [groups * channels, cols] and bias [groups * channels] have every group of rows spread to that pitch
 * with zero rows between (groups = 4 heads); mlqem_unpad_head_rows_f32 takes the real rows of the padded gradients back.  b, b_padded,
 * gb_padded, gb may be NULL. */
int mlqem_pad_head_rows_f32(const float* w, const float* b, int groups, int channels, int pitch, int cols, float* w_padded,
                            float* b_padded, mlqem_stream_t stream);
/* The same from up to four separate [groups_per_part * channels, cols] matrices (and biases; b or b[k] may be NULL) laid one after the
 * other: the query / key / value / skip projections of a TransformerConv (reference construction docs/tutorials/gnn.py:80-91) as ONE
 * padded weight without a torch.cat in front (ABI 40); pitch == channels gives the plain concatenation. */
int mlqem_pad_head_rows_parts_f32(const float* const* w, const float* const* b, int parts, int groups_per_part, int channels, int pitch,
                                  int cols, float* w_padded, float* b_padded, mlqem_stream_t stream);
int mlqem_unpad_head_rows_f32(const float* gw_padded, const float* gb_padded, int groups, int channels, int pitch, int cols,
                              float* gw, float* gb, mlqem_stream_t stream);

/* The small-tensor algebra around ASAPooling's score and fitness projections (docs/tutorials/gnn.py:85,92: ASAPooling.lin, .att,
 * .gnn_score.lin1/2/3; D = the pooling's channels), one launch per direction instead of ten element-wise launches each:
 *   forward : w_comp[D] = att_q W_lin, b_comp[1] = att_q . b_lin + att_b (the query projection composed into the one-wide score
 *             projection that is its only consumer), att_q[D] / att_x[D] = the halves of att_w[2 D], w3[3 D] = (l1_w; l2_w; l3_w),
 *             b3[3] = (l1_b, 0, l3_b)
 *   backward: g_lin_w[D D], g_lin_b[D], g_att_w[2 D] from g_w_comp[D], g_att_b[1], g_att_x[D] by the chain rule. */
int mlqem_asap_compose_f32(const float* lin_w, const float* lin_b, const float* att_w, const float* att_b, const float* l1_w,
                           const float* l1_b, const float* l2_w, const float* l3_w, const float* l3_b, int D, float* w_comp,
                           float* b_comp, float* att_q, float* att_x, float* w3, float* b3, mlqem_stream_t stream);
int mlqem_asap_compose_bwd_f32(const float* g_w_comp, const float* g_att_b, const float* lin_w, const float* lin_b,
                               const float* att_w, const float* g_att_x, int D, float* g_lin_w, float* g_lin_b, float* g_att_w,
                               mlqem_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Dense blocks (csrc/dense_block.hpp): TransformerConv's edge softmax (docs/tutorials/gnn.py:80-91) over the LONG rows of a
 * structure on the f32 matrix cores.  A block = 16 rows of at least mlqem_dense_plan_min_degree() entries that follow each other
 * in `order` inside one graph x the union of their sources (at most 512 slots), with one bit per cell; the rows of a usable block
 * are flagged in row_flag and left alone by the per-edge kernels, which serve every other row in the same call.
 *
 * mlqem_dense_plan_build, once per structure and direction (in-CSR: forward and destination-side backward; out-CSR: source-side
 * backward):  counter[1] (zero on entry) receives 16 x the number of blocks; lrows[17 * max_blocks] the blocks' rows, then their graphs;
 * records[max_blocks * mlqem_dense_plan_record_ints()] (16-byte aligned) the blocks; row_flag[num_rows] (zero on entry) the flags.
 * max_blocks = mlqem_dense_plan_max_blocks(num_rows, num_graphs).  graph_ptr[num_graphs + 1]: the graphs' ranges of positions in
 * `order` AND of row ids (null order: the rows themselves); max_span: a bound on the rows of one graph (ids beyond 262 144 from a block's first are left to the per-edge kernels).
 * ---------------------------------------------------------------------------------------------------- */
int mlqem_dense_plan_record_ints(void);
int mlqem_dense_plan_min_degree(void);
int64_t mlqem_dense_plan_max_blocks(int64_t num_rows, int64_t num_graphs);
int mlqem_dense_plan_build(const int32_t* ptr, const int32_t* idx, const int32_t* loops, const int32_t* order,
                           const int32_t* graph_ptr, int64_t num_graphs, int64_t num_rows, int64_t max_span, int32_t* counter,
                           int32_t* lrows, int32_t* records, uint8_t* row_flag, mlqem_stream_t stream);
/* 1 when the dense kernels serve this shape: one or two heads of at most 16 channels at a head pitch of 16 */
int mlqem_dense_attention_supported(int H, int C, int head_pitch);
/* mlqem_transformer_attention_train_f32 (pair-keyed dropout draws, no side table) with the rows of the plan's blocks on the matrix
 * cores: same arguments, outputs and statistics.  `parts` selects the launches of this call -- 1: the per-edge kernel over the rows
 * outside the blocks, 2: the block kernel (3: both) -- so that a caller may put the two, which touch disjoint rows, on two streams. */
int mlqem_dense_attention_train_f32(const float* qkvs, int64_t ld, const int32_t* in_ptr, const int32_t* in_src,
                                    const int32_t* loops, int64_t N, int64_t E, int H, int C, float drop_p, uint64_t seed,
                                    const uint64_t* seed_counter, int head_pitch, const int32_t* records, const int32_t* counter,
                                    const uint8_t* row_flag, int64_t max_blocks, int parts, float* out, int64_t ldo,
                                    float* attn_out, int64_t lda, float* stat_m, float* stat_den, mlqem_stream_t stream);
/* mlqem_transformer_attention_bwd_f32 in its recomputing form (no out_eid, no per-edge buffers; edge_al: [4 N H] floats, 16-byte
 * aligned) with a plan of the in-CSR for the destination side and one of the out-CSR for the source side.  `parts`: 1 / 2 =
 * destination side per-edge rows / blocks, 4 / 8 = source side per-edge rows / blocks (15: all four, in that order).  The source
 * side reads what BOTH destination-side launches file in edge_al. */
int mlqem_dense_attention_bwd_f32(const float* qkvs, int64_t ld, const float* g, int64_t ldg, const float* attn_out, int64_t lda,
                                  const float* stat_m, const float* stat_den, const int32_t* in_ptr, const int32_t* in_src,
                                  const int32_t* out_ptr, const int32_t* out_dst, const int32_t* loops, int64_t N, int64_t E, int H,
                                  int C, float drop_p, uint64_t seed, const uint64_t* seed_counter, int head_pitch,
                                  const int32_t* in_records, const int32_t* in_counter, const uint8_t* in_flag, int64_t in_max_blocks,
                                  const int32_t* out_records, const int32_t* out_counter, const uint8_t* out_flag,
                                  int64_t out_max_blocks, int parts, float* gqkvs, int64_t ldq, float* edge_al,
                                  mlqem_stream_t stream);

/* ASAPooling's cluster sums over the same plans (csrc/dense_pool.hip; rows of at most 48 channels: mlqem_dense_pool_supported -- two or
 * three channel tiles of 16; the scans and the backward entry points take D = 29..32 in rows of exactly 32 floats or D = 45..48 in rows of
 * exactly 48, the second pooling of the heads-3/2 and of the heads-5/3 variants, gnn.py:70-276).  The dense attention entry points above
 * take one to three heads.
 * mlqem_csr_softmax_aggregate_f32 with the rows of the plan's blocks on the matrix cores; stat (optional, [N, 2] floats, 8-byte
 * aligned) receives {maximum, 1 / denominator} of the block rows for the backward kernels below. */
int mlqem_dense_pool_supported(int D);
/* The pooling's other walks over the plans, for matrices whose rows are exactly 32 floats (29-32 channels, 16-byte aligned; anything
 * else: MLQEM_ERR_UNSUPPORTED / BAD_ARG, the caller keeps the per-edge entry points):
 *   mlqem_dense_segment_max_f32           = mlqem_csr_segment_max_f32 (the row itself included)
 *   mlqem_dense_softmax_aggregate_bwd_f32 = mlqem_csr_softmax_aggregate_bwd_f32 in its recomputing form with tie counts (edge_al: [4 N]
 *                                           floats, 16-byte aligned); stat: what mlqem_dense_softmax_aggregate_f32 left
 *   mlqem_dense_segment_max_bwd_f32       = mlqem_csr_segment_max_bwd_f32 with tie counts and a rank-one gradient of the maximum */
/* mlqem_leconv_fitness_bwd_f32 with the rows of the OUT structure's plan (lrows / counter / row_flag of mlqem_dense_plan_build) summed
 * by a wave each instead of a thread. */
int mlqem_dense_leconv_fitness_bwd_f32(const float* gfit, const float* fitness, const int32_t* in_ptr, const int32_t* out_ptr,
                                       const int32_t* out_dst, int64_t N, const int32_t* lrows, const int32_t* counter,
                                       const uint8_t* row_flag, int64_t max_blocks, float* gpqr, mlqem_stream_t stream);
int mlqem_dense_segment_max_f32(const float* x, int64_t ldx, const int32_t* in_ptr, const int32_t* in_src, int64_t N, int D,
                                const int32_t* records, const int32_t* counter, const uint8_t* row_flag, int64_t max_blocks, float* out,
                                int64_t ldo, mlqem_stream_t stream);
int mlqem_dense_softmax_aggregate_bwd_f32(const float* x, int64_t ldx, const float* xnew, int64_t ldn, const float* gnew, int64_t ldg,
                                          const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                                          const float* a_dst, const float* c_src, float negative_slope, int64_t N, int64_t E, int D,
                                          const float* stat, const int32_t* in_records, const int32_t* in_counter, const uint8_t* in_flag,
                                          int64_t in_max_blocks, const int32_t* out_records, const int32_t* out_counter,
                                          const uint8_t* out_flag, int64_t out_max_blocks, float* gx, int64_t ldgx, float* g_a, float* g_c,
                                          float* edge_al, const float* xmax, int64_t ldm, float* tie_count, int64_t ldt,
                                          const float* gx_rank1, mlqem_stream_t stream);
int mlqem_dense_segment_max_bwd_f32(const float* x, int64_t ldx, const float* xmax, int64_t ldm, const int32_t* in_ptr,
                                    const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst, int64_t N, int D, float* gx,
                                    int64_t ldgx, float* gshare, int64_t lds, const float* tie_count, int64_t ldt, const float* gmax_row,
                                    const float* gmax_col, const int32_t* out_records, const int32_t* out_counter, const uint8_t* out_flag,
                                    int64_t out_max_blocks, mlqem_stream_t stream);
int mlqem_dense_softmax_aggregate_f32(const float* x, int64_t ldx, const int32_t* in_ptr, const int32_t* in_src, const float* a_dst,
                                      const float* c_src, float negative_slope, int64_t N, int D, const int32_t* records,
                                      const int32_t* counter, const uint8_t* row_flag, int64_t max_blocks, float* out, int64_t ldo,
                                      float* stat, mlqem_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Host-side native encoder (no GPU needed).  Replaces the Python loops of circuit_to_graph_data_json
 * (blackwater/data/utils.py:198-389) for the part ExpValueEntry.to_pyg_data consumes
 * (blackwater/data/generators/exp_val.py:63-70): the op-node feature matrix and the op->op qubit-wire edges, in the
 * reference's node and edge order, as float64 (the reference holds Python floats).
 * ---------------------------------------------------------------------------------------------------- */
typedef struct mlqem_backend_props {
  int num_qubits;                    /* length of t1 / t2 / readout */
  const double* t1;                  /* seconds, as get_backend_properties_v1 stores them */
  const double* t2;
  const double* readout;
  int num_gate_types;                /* len(properties["gates_set"]); "barrier" and "measure" are appended internally */
  const char* const* gate_names;     /* one-hot column order */
  int num_gate_props;                /* entries of properties["gate_props"] */
  const char* const* gate_keys;      /* "cx_0_1", "sx_3", ... */
  const double* gate_error;
  const double* gate_length;
} mlqem_backend_props;

/* Two-call pattern.  Size query: x == NULL -> *num_nodes, *num_edges, *num_features, *depth are filled.  Fill: pass
 * buffers x[N*F], edge_src[E], edge_dst[E], edge_attr[E*3] (edge_attr may be NULL) with *num_nodes / *num_edges set to
 * their capacities.  Errors: MLQEM_ERR_UNSUPPORTED for a well-formed circuit the encoding does not cover (a gate
 * outside gates_set, a non-barrier gate on more than 3 qubits, more than 3 parameters, a qubit beyond the calibration
 * table: where the reference raises); MLQEM_ERR_BAD_ARG for text that is not OpenQASM 2 (unbalanced or too deeply nested
 * parentheses -- angle expressions nest at most 64 levels --, bad or out-of-range indices, mismatched register sizes,
 * more than 2^20 register bits, truncated statements).  mlqem_encode_last_error() has the message (per thread).  The
 * parser is bounded: no input makes it recurse or allocate beyond these limits (csrc `make asan` runs it under
 * AddressSanitizer / UBSan over a malformed corpus, tests/test_encoder_fuzz.py). */
int mlqem_encode_qasm(const char* qasm, const mlqem_backend_props* props, int use_qubit_features, int use_gate_features,
                      int64_t* num_nodes, int64_t* num_edges, int* num_features, int* depth, double* x,
                      int32_t* edge_src, int32_t* edge_dst, double* edge_attr);
const char* mlqem_encode_last_error(void);

/* The same encoding for ALL circuits of one estimator run() (NgemJob.result loops over them one by one:
 * blackwater/library/ngem/estimator.py:49-84), as the COLLATED batch the models consume (what PyG's Batch.from_data_list makes
 * of the per-circuit Data objects, docs/tutorials/__ml_models.py:105-119), on `threads` host threads (0 = one per core, at
 * most 16).  Two calls, so that the caller owns every output buffer at its exact size:
 *   parse: scans and checks every text; node_ptr[count+1] / edge_ptr[count+1] are the prefix sums of the circuits' node and
 *          edge counts, depths[count] (optional) their depths, *num_features the row width; *handle keeps the parsed circuits.
 *          On an error nothing is kept, *failed (optional) is the index of the first bad circuit and the message names it.
 *   fill:  x[node_ptr[count], F] as float32 (the rounding the reference's torch.tensor(..., dtype=float) applies),
 *          edge_src / edge_dst[edge_ptr[count]] as int64 with each circuit's node offset added (the two rows of edge_index),
 *          batch[node_ptr[count]] (optional) = the circuit index of every node.  May be called more than once.
 *   free:  releases the handle (NULL is fine).
 * Same error codes as mlqem_encode_qasm; same arrays, circuit by circuit (tests/test_native_encoder.py). */
int mlqem_qasm_batch_parse(const char* const* qasm, int64_t count, const mlqem_backend_props* props, int use_qubit_features,
                           int use_gate_features, int threads, void** handle, int64_t* node_ptr, int64_t* edge_ptr,
                           int* depths, int* num_features, int64_t* failed);
int mlqem_qasm_batch_fill(void* handle, int threads, float* x, int64_t* edge_src, int64_t* edge_dst, int64_t* batch);
void mlqem_qasm_batch_free(void* handle);

/* The parsed batch as a COMPACT OP STREAM for device-side expansion (round 4; mlqem_encode_expand below): 16 bytes per op and two
 * bytes per qubit argument instead of the 112 bytes per node of rows and indices mlqem_qasm_batch_fill writes -- a 1024-circuit
 * run() of 100-qubit circuits uploads 0.2 GB instead of 1.3 GB and the host writes a seventh of the bytes.
 *   mlqem_op_rec:  p0 = the first parameter as float32 (what the row holds); q[] = calibration indices (Qubit.index) of up to
 *                  three qargs (unused for barriers); slot = one-hot column; meta = q_cnt (bits 0-1, 0 for a barrier) | barrier
 *                  (bit 2) | p_cnt (bits 4-5); winc = index of the op's first qubit argument in `wires` (the next op's winc, or
 *                  the circuit's wire_ptr end, closes the range).
 *   wires:         the circuit-local wire (flat qubit number) of EVERY qubit argument, op after op -- what the op -> op edges
 *                  are built from (utils.py:334-347: one edge per qubit wire from the previous op on it).
 *   mlqem_x_patch: x[node, col] = value for what the record has no room for (second / third parameters; the calibration entry
 *                  of a gate on three qubits); unused slots hold node = 0xFFFFFFFF.
 * stream_sizes: wire_ptr / patch_ptr [count + 1] = prefix sums of the circuits' wire and patch-slot counts, *max_wires the widest
 * circuit; stream_fill writes the three arrays (node_ptr as from mlqem_qasm_batch_parse).  MLQEM_ERR_UNSUPPORTED beyond 65 535
 * wires / calibration qubits or 2^32 ops. */
typedef struct mlqem_op_rec { float p0; uint16_t q[3]; uint8_t slot; uint8_t meta; uint32_t winc; } mlqem_op_rec;
typedef struct mlqem_x_patch { uint32_t node; uint32_t col; float value; } mlqem_x_patch;
int mlqem_qasm_batch_stream_sizes(void* handle, int64_t* wire_ptr, int64_t* patch_ptr, int* max_wires);
int mlqem_qasm_batch_stream_fill(void* handle, int threads, const int64_t* wire_ptr, const int64_t* patch_ptr, mlqem_op_rec* ops,
                                 uint16_t* wires, mlqem_x_patch* patches);
/* g1[(G + 2) * Q], g2[(G + 2) * Q * Q] (G = num_gate_types, Q = num_qubits): index into gate_error / gate_length of the
 * calibration entry of (one-hot slot, qubit) / (slot, qubit, qubit), -1 where there is none -- the lookups of utils.py:263-269 as
 * tables the device indexes.  Keys naming a qubit >= num_qubits have no place in the tables and are left out: the host string lookup
 * would still find such a key, but no circuit the encoder accepts addresses that qubit (ops on qubits >= num_qubits are refused),
 * so the two lookups agree on every op that can occur. */
int mlqem_props_gate_tables(const mlqem_backend_props* props, int32_t* g1, int32_t* g2);

/* Device-side expansion of the op stream (all pointers device memory): x[N, F] (row stride ldx >= F; F = 3 + num_slots + 9 (qubit
 * features) + 2 (gate features)), edge_src / edge_dst[E] as int64 with node offsets applied (the two rows of edge_index, in the
 * reference's edge order), batch[N] (optional): the arrays mlqem_qasm_batch_fill writes, bit for bit.  ops / wires / patches as
 * above; node_ptr[B + 1] as from mlqem_qasm_batch_parse; W = wire_ptr[B]; E = edge_ptr[B]; t1 / t2 / readout: the calibration
 * table as float32 (the rounding the rows get); g1 / g2: mlqem_props_gate_tables; gate_error / gate_length as float32;
 * num_slots = num_gate_types + 2.  Edges: a stable radix sort of the qubit arguments by (circuit, wire) puts every wire's ops in
 * program order; neighbours are the endpoints of an edge; a source's out-edges are listed latest-inserted first, as the
 * reference's DAG hands them back (blackwater/data/utils.py:334-347).  Asynchronous on `stream`;
 * replaces circuit_to_graph_data_json + ExpValueEntry.to_pyg_data + Batch.from_data_list for a run() of circuits
 * (blackwater/library/ngem/estimator.py:49-84). */
size_t mlqem_encode_expand_workspace_bytes(int64_t N, int64_t W);
int mlqem_encode_expand(const mlqem_op_rec* ops, const uint16_t* wires, const mlqem_x_patch* patches, int64_t num_patches,
                        const int64_t* node_ptr, int64_t N, int64_t W, int64_t E, int64_t B, int max_wires, const float* t1,
                        const float* t2, const float* readout, int num_cal_qubits, const int32_t* g1, const int32_t* g2,
                        const float* gate_error, const float* gate_length, int num_slots, int use_qubit_features,
                        int use_gate_features, float* x, int64_t ldx, int64_t* edge_src, int64_t* edge_dst, int64_t* batch,
                        void* workspace, size_t workspace_bytes, mlqem_stream_t stream);

/* Circuit-level features of the MLP regressors -- the per-circuit part of encode_data / encode_data_v2_ecr
 * (docs/tutorials/mlp.py:111-145 count_gates_by_rotation_angle, :148-252; == blackwater/library/learning/mlp.py) from
 * the same op scan as the graph encoder (host CPU):
 *   gate_counts[i] = number of ops named gate_names[i]                  (QuantumCircuit.count_ops lookups)
 *   angle_hist[b]  = number of one-qubit rx/ry/rz ops whose angle lies in bin b of bin_edges[num_edges]
 *                    (numpy.histogram rule: [e_b, e_b+1), last bin closed; values outside are dropped).
 * The caller passes the edges (numpy.arange(-2pi, 2pi + bin, bin) in the reference) so both sides bin identically. */
int mlqem_circuit_features_qasm(const char* qasm, const char* const* gate_names, int num_gates, const double* bin_edges,
                                int num_edges, int64_t* gate_counts, int64_t* angle_hist);
/* The same for every circuit of one run() (PostProcessedJob.result loops over them: blackwater/library/learning/estimator.py:
 * 220-245) on `threads` host threads (0 = one per core, at most 16): gate_counts[count, num_gates], angle_hist[count,
 * num_edges - 1].  On an error *failed (optional) is the index of the first bad circuit and the message names it. */
int mlqem_circuit_features_qasm_batch(const char* const* qasm, int64_t count, const char* const* gate_names, int num_gates,
                                      const double* bin_edges, int num_edges, int threads, int64_t* gate_counts,
                                      int64_t* angle_hist, int64_t* failed);

#ifdef __cplusplus
}
#endif
#endif /* MLQEM_HIP_H */
