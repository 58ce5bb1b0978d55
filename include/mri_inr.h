/*
 * mri_inr.h -- C ABI of the MI355X (gfx950) hot path for Benjamin-Fouquet/mri_interpolation.
 *
 * The reference is pure Python: its hot path sits behind Python class surfaces, not an FFI
 * (SURVEY.md section 8b).  Each entry point below names the reference op chain it replaces
 * (file:line into the reference repository); the Python classes in mri_interpolation_amd/ bind
 * these with ctypes (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (a torch tensor's data_ptr())
 *     unless the parameter is documented as host memory; the library never allocates,
 *     frees or retains memory;
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue work on it and never
 *     synchronise; the library keeps no global state besides the last-error string;
 *   - all matrices are float32, row-major, with an explicit leading dimension in elements;
 *   - return value 0 = success, negative = mri_status; mri_last_error() returns a
 *     thread-local message for the last failure.  No C++ exception crosses this boundary.
 */
#ifndef MRI_INR_H
#define MRI_INR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRI_MAX_LEVELS 32
#define MRI_MAX_DIM 7 /* len(PRIMES), reference encoding.py:40,90-92 */

typedef enum mri_status {
  MRI_OK = 0,
  MRI_ERR_INVALID_ARGUMENT = -1,
  MRI_ERR_UNSUPPORTED = -2,
  MRI_ERR_LAUNCH = -3
} mri_status;

typedef enum mri_activation {
  MRI_ACT_IDENTITY = 0,
  MRI_ACT_RELU = 1, /* nn.ReLU,  reference models.py:31,54 */
  MRI_ACT_SINE = 2, /* sin(w0*z), reference models.py:108-114 */
  MRI_ACT_GELU = 3  /* nn.GELU (erf form), reference models.py:672,736 */
} mri_activation;

/* How mri_linear_backward_data turns dL/d(activation output) into dL/d(pre-activation). */
typedef enum mri_deriv_mode {
  MRI_DERIV_NONE = 0,     /* no multiply (identity / network input)                     */
  MRI_DERIV_MUL = 1,      /* multiply by a stored derivative matrix (sine, gelu)        */
  MRI_DERIV_RELU_MASK = 2 /* multiply by (stored activation output > 0)                 */
} mri_deriv_mode;

/* Host-side description of a multiresolution hash grid (HOST memory, copied by value).
 * Mirrors what MultiResHashGrid / MultiResHashGridV2 compute in __init__
 * (reference encoding.py:168-185, 310-330): one resolution per level and axis, one table
 * size per level, tables concatenated in one flat (sum T_l, F) float32 array. */
typedef struct mri_grid_desc {
  int32_t dim;        /* D, 1..MRI_MAX_DIM                                   */
  int32_t n_levels;   /* L, 1..MRI_MAX_LEVELS                                */
  int32_t n_features; /* F, 1..8                                             */
  int32_t reserved;
  float resolution[MRI_MAX_LEVELS][MRI_MAX_DIM + 1]; /* res_l on axis d (float32, as the
                                                        reference multiplies in float32)  */
  uint32_t table_size[MRI_MAX_LEVELS];               /* T_l, any value >= 1 (not only 2^k) */
  uint64_t table_offset[MRI_MAX_LEVELS];             /* first row of level l in the flat table */
} mri_grid_desc;

/* Library / build identification: "mri_inr <version> gfx950". */
const char* mri_version(void);
const char* mri_last_error(void);
/* Speed-only tuning knobs ("xcd_affinity" 0/1, "bwd_lds_max_parts" n, "bwd_blocks_per_level" n,
 * "bwd_dense_max_parts" n, "bwd_dense_blocks" n, "bwd_fuse_dense" 0/1, "fwd_pair" 0/1,
 * "mlp_stagger" 0..8, "mlp_x3" 0/1/2: which kernel serves the decoder -- 1 (default) the bf16-pipe
 * kernel with exact three-term operands, 2 its four-wave form (128-wide), 0 the f32-MFMA kernels;
 * "siren_rows" 0/1: SirenNet of width 256 -- 1 (default) the chain kernels that keep a wave's rows in
 * registers across the layers (csrc/siren_rows.hip), 0 the LDS-image kernels of every other width);
 * results stay within fp32 summation-order noise.
 * ONE option trades accuracy, for grids with two features per level: "bwd_records" 0 (default) / 1,
 * the gradient records of mri_hashgrid_backward* (see there).  0: every contribution w * g is the f32
 * product the reference's autograd forms, their sum per table entry is exact.  1: each contribution is
 * rounded to 18-21 significant bits (8-byte packed records) before the exact sum -- 13 us faster per
 * step at BASELINE config 4 and NOT f32-equivalent (per table entry 17x the median error of the f32
 * records against float64, tests/test_gpu_round3.py). */
int mri_set_option(const char* name, int32_t value);

/* ---- hash-grid encoding --------------------------------------------------------------
 * Replaces, per level, the op chain of _HashGrid.forward / _HashGridV2.forward
 * (reference encoding.py:108-128, 232-270) incl. fast_hash (encoding.py:69-78), and the
 * torch.cat over levels (encoding.py:190-191, 335-336).
 *   x    (n, D) row-major coordinates.
 *   out  element (row i, level l, feature f) is written to
 *        out[l * out_level_stride + i * out_row_stride + f * out_feat_stride].
 *        Reference layout (n, L*F): level stride F, row stride L*F, feature stride 1.
 *        Feature-major layout (L*F, n) used inside the fused trainer (coalesced stores,
 *        read back by mri_linear_forward with x_row_stride 1): level stride F*n, row
 *        stride 1, feature stride n.
 */
int mri_hashgrid_forward(const mri_grid_desc* grid, const float* x, int64_t n,
                         const float* table, float* out, int64_t out_level_stride,
                         int64_t out_row_stride, int64_t out_feat_stride, void* stream);

/* Replaces the autograd of the same chain: aten::embedding_dense_backward per level plus
 * the mul/sum backward (reference encoding.py:127-128).
 * d_table (sum T_l, F) is ACCUMULATED into (caller zeroes it).  d_out uses the same
 * three-stride addressing as `out` above.
 *   method 0 = choose per level; 1 = global float atomics; 2 = LDS owner-computes scan with
 *   64-bit fixed-point accumulation (bitwise reproducible gradients).
 *   Accuracy of methods 0 / 2 (the binned path): the contributions of a table entry are added as
 *   integers in units of 2^-e, e chosen per level so that n * max|d_out| * 2^e < 2^61: the sum is exact
 *   but for what lies below that unit -- a contribution smaller than 2^-40 of the level's max |d_out|
 *   (n = 2^18; 2^-(58 - log2 n) in general) is truncated toward zero, so an entry whose only
 *   contributions are that small reads 0 where the reference holds a denormal-scale value.  With
 *   mri_set_option("bwd_records", 1) each contribution is first rounded to 18-21 significant bits and
 *   the cut-off is 2^-45 of the level's maximum (2^(E-45), E the exponent of max |d_out|).
 *   workspace: device scratch of at least mri_hashgrid_backward_workspace_bytes(grid, n)
 *   bytes, 16-byte aligned; no initialisation needed (the call clears what it needs), it
 *   may be shared by calls with different grids / n on the same stream.  May be NULL for
 *   method 1.
 */
int64_t mri_hashgrid_backward_workspace_bytes(const mri_grid_desc* grid, int64_t n);
/* Optional split: the first stage of the binned backward (counting the records per table
 * slice) needs only the coordinates, so a trainer can run it on a side stream beside the
 * forward pass and then call mri_hashgrid_backward with `method | MRI_BWD_PREPARED` on the same
 * workspace (same grid, n and method; nothing else may use the workspace in between). */
#define MRI_BWD_PREPARED 16
/* `method | MRI_BWD_OVERWRITE`: d_table is OVERWRITTEN with the gradient (every row of every
 * level is written) instead of accumulated into, which saves the caller's memset and the
 * read half of the accumulation. */
#define MRI_BWD_OVERWRITE 32
int mri_hashgrid_backward_prepare(const mri_grid_desc* grid, const float* x, int64_t n,
                                  int32_t method, void* workspace, int64_t workspace_bytes,
                                  void* stream);
int mri_hashgrid_backward(const mri_grid_desc* grid, const float* x, const float* d_out,
                          int64_t n, int64_t dout_level_stride, int64_t dout_row_stride,
                          int64_t dout_feat_stride, float* d_table, int32_t method,
                          void* workspace, int64_t workspace_bytes, void* stream);
/* Same, for the levels whose bit is set in `level_mask` only (bit l = level l): the gradient of
 * one level group at a time, so that a data-parallel caller can start reducing a finished group
 * (all-reduce over RCCL) while the next one is computed.  With MRI_BWD_PREPARED one
 * mri_hashgrid_backward_prepare call on the workspace serves every group; without it each call
 * recounts all levels.  Rows of levels outside the mask are not touched. */
int mri_hashgrid_backward_levels(const mri_grid_desc* grid, const float* x, const float* d_out,
                                 int64_t n, int64_t dout_level_stride, int64_t dout_row_stride,
                                 int64_t dout_feat_stride, float* d_table, int32_t method,
                                 uint32_t level_mask, void* workspace, int64_t workspace_bytes,
                                 void* stream);
/* Table gradient AND the table's optimiser step in one pass (one rank, no gradient accumulation):
 * embedding_dense_backward x L followed by torch.optim.Adam.step on the embedding weights (reference
 * models.py:68-70 configure_optimizers, :61-66 training_step).  Where a gradient entry is complete
 * the Adam update of mri_adam_step (same operations, same order: bit-identical results) is applied
 * to table / exp_avg / exp_avg_sq in place; the gradient itself is never written (8 bytes of HBM
 * traffic per table parameter and step less, and the optimiser launch shrinks to the decoder).
 * `step` is the 1-based step number, `grad_scale` multiplies the gradient first.  method: 0 / 2
 * (+ MRI_BWD_PREPARED); returns MRI_ERR_UNSUPPORTED if a level of the grid would take the atomic
 * kernel (then call mri_hashgrid_backward + mri_adam_step). */
int mri_hashgrid_backward_adam(const mri_grid_desc* grid, const float* x, const float* d_out,
                               int64_t n, int64_t dout_level_stride, int64_t dout_row_stride,
                               int64_t dout_feat_stride, float* table, float* exp_avg,
                               float* exp_avg_sq, double lr, double beta1, double beta2, double eps,
                               int32_t step, float grad_scale, int32_t method, void* workspace,
                               int64_t workspace_bytes, void* stream);

/* ---- fully connected layers (f32 MFMA) ------------------------------------------------
 * y = act(w0 * (x W^T + b))   with W (N, K) row-major as nn.Linear / SirenLayer store it.
 * Replaces F.linear + activation: reference models.py:46-56 (BaseMLP.layers),
 * models.py:153-156 (SirenLayer.forward), models.py:730-736 (HashMLP decoder Linear+act).
 *   x element (m, k) at x[m*x_row_stride + k*x_col_stride]  (x_col_stride 1 = row-major,
 *     x_row_stride 1 = feature-major (K, M) as written by the level-major encoder output).
 *   deriv (optional, (M, ldd)) receives d act / d z  (w0*cos(w0 z) for sine, gelu'(z));
 *     pass NULL for identity / relu (relu's mask is recovered from y).
 *   w0 is only used by MRI_ACT_SINE.
 */
int mri_linear_forward(const float* x, int64_t x_row_stride, int64_t x_col_stride,
                       const float* weight, const float* bias /* may be NULL */, int64_t m,
                       int32_t n, int32_t k, int32_t activation, float w0, float* y,
                       int64_t ldy, float* deriv /* may be NULL */, int64_t ldd, void* stream);

/* dx = (dy W) (.) g   where g is chosen by deriv_mode from `deriv` ((M, K), leading dim ldd):
 * the gradient w.r.t. the PREVIOUS layer's pre-activation (autograd of F.linear followed by
 * the previous activation's backward).  dx element (m, k) at dx[m*dx_row_stride + k*dx_col_stride]
 * (deriv_mode must be MRI_DERIV_NONE when dx_col_stride != 1).
 */
int mri_linear_backward_data(const float* dy, int64_t lddy, const float* weight, int64_t m,
                             int32_t n, int32_t k, int32_t deriv_mode, const float* deriv,
                             int64_t ldd, float* dx, int64_t dx_row_stride,
                             int64_t dx_col_stride, void* stream);

/* d_weight (N, K) += dy^T x ;  d_bias (N) += column sums of dy (d_bias may be NULL).
 * Accumulates with float atomics (caller zeroes the gradient buffers once per step).
 * x addressed with the same two strides as in mri_linear_forward. */
int mri_linear_backward_weight(const float* dy, int64_t lddy, const float* x,
                               int64_t x_row_stride, int64_t x_col_stride, int64_t m, int32_t n,
                               int32_t k, float* d_weight, float* d_bias, void* stream);

/* In-place dy *= g (same deriv modes as above) -- activation after the LAST Linear
 * (BaseMLP keeps one, reference models.py:54). */
int mri_apply_deriv(float* dy, int64_t lddy, int32_t deriv_mode, const float* deriv,
                    int64_t ldd, int64_t m, int32_t n, void* stream);

/* ---- frequency encoding ----------------------------------------------------------------------
 * `Frequency.forward` (reference encoding.py:43-66): for every row and input axis d,
 * out[d*2L + l] = sin(x_d * 2^l), out[d*2L + L + l] = cos(x_d * 2^l), l = 0..L-1.
 *   x (n, dim) row-major with leading dimension ldx; out (n, dim*2L) with leading dimension ldo.
 * mri_frequency_backward writes (does not accumulate) dx (n, dim) from d_out (n, dim*2L). */
int mri_frequency_forward(const float* x, int64_t ldx, int64_t n, int32_t dim, int32_t n_levels,
                          float* out, int64_t ldo, void* stream);
int mri_frequency_backward(const float* x, int64_t ldx, const float* d_out, int64_t ldg, int64_t n,
                           int32_t dim, int32_t n_levels, float* dx, int64_t lddx, void* stream);

/* ---- fused tiny MLP ------------------------------------------------------------------------
 * The decoder of BASELINE configs 2/4/5: k_in -> hidden -> hidden -> 1, ReLU on the hidden
 * layers, linear output (reference config/hash_config.json "network"; module form
 * models.py:46-56 / 730-744 with ReLU, no BatchNorm).  ONE persistent kernel per call:
 * activations never leave the CU; k_in <= 32: weights in registers, products on the bf16 matrix
 * pipe with every f32 operand split exactly into three bf16 terms (f32-accurate, csrc/bf16x3.h);
 * hidden 64 with 32 < k_in <= 64: weights resident in LDS, products on the f32 MFMA.
 *   x        (k_in, n) FEATURE-MAJOR input features (the layout mri_hashgrid_forward writes)
 *   w1 (hidden, k_in), w2 (hidden, hidden), w3 (1, hidden) row-major as nn.Linear stores them
 * mri_tiny_mlp_supported: 1 if (k_in, hidden, dim_out) has a fused kernel
 *   (hidden 128 with k_in <= 32, hidden 64 with k_in <= 64, dim_out 1), else 0.
 * mri_tiny_mlp_forward:  y[n] = MLP(x).
 * mri_tiny_mlp_train:    forward + F.mse_loss(target, y) + backward in one pass:
 *   d_w*, d_b* += parameter gradients, loss_out[0] += mean squared error,
 *   d_x (k_in, n) feature-major = dLoss/dx (optional, may be NULL), y (optional) predictions.
 *   Gradients are those of mean((y - target)^2) / grad_divisor.  Per-workgroup partial sums go
 *   through `workspace` (mri_tiny_mlp_workspace_bytes, no initialisation needed) and are added
 *   in a fixed order: results are bitwise reproducible. */
int mri_tiny_mlp_supported(int32_t k_in, int32_t hidden, int32_t dim_out);
int64_t mri_tiny_mlp_workspace_bytes(int32_t k_in, int32_t hidden, int64_t n);
int mri_tiny_mlp_forward(const float* x, int64_t n, int32_t k_in, int32_t hidden, const float* w1,
                         const float* b1, const float* w2, const float* b2, const float* w3,
                         const float* b3, float* y, void* stream);
int mri_tiny_mlp_train(const float* x, const float* target, int64_t n, int32_t k_in,
                       int32_t hidden, const float* w1, const float* b1, const float* w2,
                       const float* b2, const float* w3, const float* b3, float grad_divisor,
                       float* d_w1, float* d_b1, float* d_w2, float* d_b2, float* d_w3,
                       float* d_b3, float* d_x /* may be NULL */, float* loss_out,
                       float* y /* may be NULL */, void* workspace, int64_t workspace_bytes,
                       void* stream);
/* Same, but d_w*, d_b* and loss_out are overwritten instead of accumulated into. */
int mri_tiny_mlp_train_overwrite(const float* x, const float* target, int64_t n, int32_t k_in,
                                 int32_t hidden, const float* w1, const float* b1,
                                 const float* w2, const float* b2, const float* w3,
                                 const float* b3, float grad_divisor, float* d_w1, float* d_b1,
                                 float* d_w2, float* d_b2, float* d_w3, float* d_b3, float* d_x,
                                 float* loss_out, float* y, void* workspace,
                                 int64_t workspace_bytes, void* stream);

/* The same step for a SLICE of a batch: columns [0, n) of feature-major blocks x / d_x whose feature
 * rows are x_ld elements apart (x_ld >= n), `n` targets, out of a batch of n_total coordinates --
 * the mean of the loss and the gradient scale are taken over n_total, so the slices of a batch
 * add up to the whole-batch call (first slice with overwrite = 1, the others with 0).  Lets a
 * caller run the encoder of slice k+1 beside the decoder of slice k. */
int mri_tiny_mlp_train_slice(const float* x, int64_t x_ld, const float* target, int64_t n,
                             int64_t n_total, int32_t k_in, int32_t hidden, const float* w1,
                             const float* b1, const float* w2, const float* b2, const float* w3,
                             const float* b3, float grad_divisor, float* d_w1, float* d_b1,
                             float* d_w2, float* d_b2, float* d_w3, float* d_b3, float* d_x,
                             float* loss_out, float* y, int32_t overwrite, void* workspace,
                             int64_t workspace_bytes, void* stream);

/* mri_tiny_mlp_train[_overwrite] that also reports max |d_x| per PAIR of feature rows: dx_pair_absmax
 * (k_in / 2 non-negative floats) is raised with atomic maxima, so the caller zeroes it before the call.
 * With a hash-grid encoder of two features per level this is max |dLoss / d features| per level, the scale
 * mri_hashgrid_backward_scaled needs for its gradient records -- taken where d_x is produced instead of by
 * a pass over it.  Served by the bf16-pipe decoder kernel only (mri_tiny_mlp_dx_absmax_supported, else
 * MRI_ERR_UNSUPPORTED); d_x must be requested. */
int mri_tiny_mlp_dx_absmax_supported(int32_t k_in, int32_t hidden);
int mri_tiny_mlp_train_dx_absmax(const float* x, const float* target, int64_t n, int32_t k_in,
                                 int32_t hidden, const float* w1, const float* b1, const float* w2,
                                 const float* b2, const float* w3, const float* b3, float grad_divisor,
                                 float* d_w1, float* d_b1, float* d_w2, float* d_b2, float* d_w3,
                                 float* d_b3, float* d_x, float* loss_out, float* y, int32_t overwrite,
                                 float* dx_pair_absmax, void* workspace, int64_t workspace_bytes,
                                 void* stream);

/* mri_hashgrid_backward_levels with max |d_out| per level supplied by the caller (level_absmax, n_levels
 * non-negative floats, each >= the true maximum's binade is what matters: values above it would clip;
 * NULL = found by a pass over d_out, as the other entry points do).  Grids with two features per level
 * store their gradient records relative to that scale (csrc/hashgrid_bwd.hip, packed records). */
int mri_hashgrid_backward_scaled(const mri_grid_desc* grid, const float* x, const float* d_out, int64_t n,
                                 int64_t dout_level_stride, int64_t dout_row_stride,
                                 int64_t dout_feat_stride, float* d_table, int32_t flags,
                                 uint32_t level_mask, const float* level_absmax, void* workspace,
                                 int64_t workspace_bytes, void* stream);

/* Encoder + decoder in ONE kernel (reference models.py:739-754, HashMLP.forward = encoder -> ReLU MLP,
 * with F.mse_loss and the decoder's autograd): the decoder's workgroups look the features of their
 * next row tile up themselves while the matrix pipe works on the current one, so the lookup kernel
 * and the feature block (2 x 4 L F bytes per coordinate) disappear.  Same arithmetic as
 * mri_hashgrid_forward followed by mri_tiny_mlp_train[_overwrite]: bit-identical loss, gradients and
 * d_enc.  Grids with n_features = 2, dim 2..4, <= 16 levels, < 2^29 table rows; hidden 64 or 128
 * (mri_hash_tiny_mlp_supported, else MRI_ERR_UNSUPPORTED).  coords (n, dim) row-major; d_enc
 * (2 L, d_enc_ld) feature-major = dLoss / d features, the input of mri_hashgrid_backward; workspace as
 * mri_tiny_mlp_workspace_bytes(2 L, hidden, n). */
int mri_hash_tiny_mlp_supported(const mri_grid_desc* grid, int32_t hidden);
int mri_hash_tiny_mlp_train(const mri_grid_desc* grid, const float* table, const float* coords,
                            const float* target, int64_t n, int32_t hidden, const float* w1,
                            const float* b1, const float* w2, const float* b2, const float* w3,
                            const float* b3, float grad_divisor, float* d_w1, float* d_b1,
                            float* d_w2, float* d_b2, float* d_w3, float* d_b3, float* d_enc,
                            int64_t d_enc_ld, float* loss_out, float* y, int32_t overwrite,
                            void* workspace, int64_t workspace_bytes, void* stream);

/* Gradient w.r.t. the COORDINATES: the reference detaches only the integer part of x * res
 * (encoding.py:111-113), so autograd carries d out / d x through the interpolation weights:
 * d_x[i][d] = res_d * sum over levels and corners of (+-1) prod_{e != d} w_e * <d_out, row>.
 * Not on the training path (coordinates carry no gradient there); d_x is (n, D) row-major. */
int mri_hashgrid_backward_input(const mri_grid_desc* grid, const float* x, const float* d_out,
                                int64_t n, int64_t dout_level_stride, int64_t dout_row_stride,
                                int64_t dout_feat_stride, const float* table, float* d_x,
                                void* stream);

/* ---- lookup and decoder running side by side ----------------------------------------------
 * The decoder kernel is bound by the f32 matrix rate and leaves the texture path idle; the lookup
 * is bound by the texture path, needs no LDS and 27 registers.  Launched on two streams they share
 * the CUs: mri_hashgrid_forward_signal (queue it FIRST) produces the feature-major block
 * (L*2, out_ld) slice by slice -- a slice = slice_rows consecutive coordinates =
 * mri_tiny_mlp_round_rows(), what one round of the decoder's workgroups reads -- and adds
 * mri_hashgrid_forward_signal_blocks() to ready[slice] as its blocks finish (write-through stores,
 * agent-scope counter; the counters only ever grow: the caller keeps the running total and passes
 * it as ready_target).  mri_tiny_mlp_train_overlapped is mri_tiny_mlp_train[_overwrite] whose
 * workgroups wait for ready[r] >= ready_target before touching round r's rows (bounded wait: a
 * producer that never arrives sets *status = 1 instead of hanging the GPU).  Same results as the
 * two plain calls, bit for bit.  Needs n_features == 2, 2 <= dim <= 4, hidden == 128. */
int64_t mri_hashgrid_forward_signal_blocks(const mri_grid_desc* grid, int64_t slice_rows);
int mri_hashgrid_forward_signal(const mri_grid_desc* grid, const float* x, int64_t n,
                                const float* table, float* out, int64_t out_ld,
                                int64_t slice_rows, uint64_t* ready, void* stream);
int64_t mri_tiny_mlp_round_rows(int32_t k_in, int32_t hidden, int64_t n);
int mri_tiny_mlp_train_overlapped(const float* x, const float* target, int64_t n, int32_t k_in,
                                  int32_t hidden, const float* w1, const float* b1,
                                  const float* w2, const float* b2, const float* w3,
                                  const float* b3, float grad_divisor, float* d_w1, float* d_b1,
                                  float* d_w2, float* d_b2, float* d_w3, float* d_b3, float* d_x,
                                  float* loss_out, int32_t overwrite, const uint64_t* ready,
                                  uint64_t ready_target, int32_t* status, void* workspace,
                                  int64_t workspace_bytes, void* stream);

/* ---- fused SIREN chain -------------------------------------------------------------------
 * SirenNet.forward (reference models.py:230-233): n_sine_layers x [F.linear -> sin(w0 .)]
 * (SirenLayer.forward, models.py:153-156; the first layer with w0_first) and the linear head,
 * in ONE persistent kernel: a row tile's activations stay on chip across all layers -- in the
 * registers of the wave that owns the rows at hidden = 256 (csrc/siren_rows.hip), in an LDS image
 * at the other widths (csrc/siren_chain.hip) --, the hidden x hidden weights stream from L2.
 * Supported: hidden in
 * {32, 64, 128, 256}, dim_in <= 8, 1 <= n_sine_layers <= MRI_SIREN_MAX_LAYERS, one output, biases everywhere
 * (mri_siren_supported); other shapes go layer by layer through mri_linear_*.
 * weight / bias: HOST arrays of n_sine_layers + 1 device pointers -- [0] (hidden, dim_in),
 * [1 .. n-1] (hidden, hidden), [n] the head (1, hidden); all row-major, 16-byte aligned.
 * act / deriv: HOST arrays of n_sine_layers device pointers to (n, hidden) row-major buffers that
 * receive each sine layer's output and its derivative w0 cos(.) for the backward pass, or both
 * NULL (inference: nothing but y is written).  y: (n) predictions.
 * workspace: mri_siren_forward_workspace_bytes device bytes, 16-byte aligned (0 bytes / NULL with one
 * sine layer): each call first splits the hidden x hidden weights there into the three bf16 terms
 * the matrix pipe multiplies (f32-accurate, csrc/bf16x3.h), in the order the kernel streams them. */
#define MRI_SIREN_MAX_LAYERS 8
int mri_siren_supported(int32_t dim_in, int32_t hidden, int32_t n_sine_layers, int32_t dim_out);
int64_t mri_siren_forward_workspace_bytes(int32_t hidden, int32_t n_sine_layers);
int mri_siren_forward(const float* x, int64_t n, int32_t dim_in, int32_t hidden,
                      int32_t n_sine_layers, const float* const* weight,
                      const float* const* bias, float w0_first, float w0, float* const* act,
                      float* const* deriv, float* y, void* workspace, int64_t workspace_bytes,
                      void* stream);

/* Backward of the same network (autograd of models.py:230-233 + the loss gradient dy the caller
 * got from mri_mse_loss): dz walks the layers inside LDS (one persistent kernel), each hidden x hidden
 * weight gradient is one persistent kernel that keeps the whole result in MFMA
 * accumulators; partial sums meet in `workspace` and are added in a fixed order (bitwise
 * reproducible).  act / deriv: what mri_siren_forward stored; dz: HOST array of n_sine_layers
 * device pointers to (n, hidden) scratch ([0] unused, may be NULL); d_weight / d_bias: HOST arrays
 * of n_sine_layers + 1 device pointers, gradients are ADDED to them.  head_done = 1: the head's
 * backward was done by mri_siren_forward_loss (dy, act / deriv [n_sine_layers - 1] unused).
 * head_done = 1 continues THAT call: same n, same network, the SAME `workspace` bytes, untouched in
 * between, and the same "siren_rows" option -- at hidden = 256 the two calls share the work of the head
 * (what the forward call leaves in dz[n_sine_layers - 1] and in the workspace is private to the pair). */
int64_t mri_siren_backward_workspace_bytes(int64_t n, int32_t hidden, int32_t n_sine_layers);
int mri_siren_backward(const float* x, const float* dy, int64_t n, int32_t dim_in, int32_t hidden,
                       int32_t n_sine_layers, const float* const* weight,
                       const float* const* act, const float* const* deriv, float* const* dz,
                       float* const* d_weight, float* const* d_bias, int32_t head_done,
                       void* workspace, int64_t workspace_bytes, void* stream);
/* Training forward WITH the loss (models.py:61-66 training_step: y_pred = forward(x);
 * F.mse_loss(y, y_pred)): the kernel of mri_siren_forward also compares its predictions with
 * `target`, and runs the head's backward in the tile's tail, where the last sine layer's output
 * and derivative are still in registers (they are NOT written to act / deriv [n_sine_layers - 1],
 * which may be NULL): dz_last (n, hidden) = dLoss / d(pre-activation of the last sine layer), and,
 * ADDED to them, d_w_head (1, hidden), d_b_head (1), d_b_last (hidden), loss_out[0] += mean over
 * n_total (n <= n_total: a slice of a batch).  Gradients are scaled by 1 / grad_divisor.  Follow
 * with mri_siren_backward(..., head_done = 1, dy = NULL), which then starts from dz_last =
 * dz[n_sine_layers - 1].  Needs n_sine_layers >= 2; workspace: mri_siren_backward_workspace_bytes. */
int mri_siren_forward_loss(const float* x, const float* target, int64_t n, int64_t n_total,
                           int32_t dim_in, int32_t hidden, int32_t n_sine_layers,
                           const float* const* weight, const float* const* bias, float w0_first,
                           float w0, float grad_divisor, float* const* act, float* const* deriv,
                           float* dz_last, float* y, float* d_w_head, float* d_b_head,
                           float* d_b_last, float* loss_out, void* workspace,
                           int64_t workspace_bytes, void* stream);

/* ---- loss ------------------------------------------------------------------------------
 * F.mse_loss(y, y_pred) (reference models.py:64): loss_out[0] += mean((pred-target)^2)
 * (device scalar, caller zeroes), d_pred = 2 (pred - target) / (count * grad_divisor) if
 * d_pred != NULL.  pred/target/d_pred are contiguous with `count` elements.
 * grad_divisor > 1 pre-averages gradients over data-parallel ranks. */
int mri_mse_loss(const float* pred, const float* target, int64_t count, float grad_divisor,
                 float* loss_out, float* d_pred, void* stream);

/* ---- optimiser -------------------------------------------------------------------------
 * torch.optim.Adam, single step over one flat buffer (reference models.py:68-70; defaults
 * betas (0.9, 0.999), eps 1e-8, no weight decay).  `step` is the 1-based step count.
 * grad_scale multiplies the gradient first (1/world_size after an all-reduce sum).
 * The four pointers may be any 4-byte aligned range of their buffers as long as they share
 * one offset within a 16-byte line (the same slice of four 16-byte aligned buffers does):
 * data-parallel ranks step each level group as soon as its reduction has landed. */
int mri_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                  int64_t count, double lr, double beta1, double beta2, double eps,
                  int32_t step, float grad_scale, void* stream);

/* ---- one training step, queued by one call ----------------------------------------------------
 * The per-batch body of the reference's training loop for its HashMLP path with the tiny-MLP decoder
 * (reference launcher.py:156-165 pl.Trainer.fit -> models.py:61-70 training_step / backward / Adam.step, with
 * the DataLoader of datamodules.py:198-205 producing the next batch meanwhile) as ONE host call that
 * composes the entry points above:
 *   side stream: [count this batch's gradient records if `counted` = 0]; sample + gather the NEXT batch;
 *                zero its absmax buffer; count its records
 *   main stream: lookup -> decoder forward + MSE + backward (overwriting every gradient) -> table gradient
 *                (MRI_BWD_PREPARED | MRI_BWD_OVERWRITE) -> Adam over [param, param + n_params)
 * The side work of a call is awaited by the next call (`join_pending` = 1) or through `ev_join`.
 * All pointers are device pointers except `grid`, `shape`, `axis_offset` (host); streams and events are the
 * caller's.  Feature-major (2 L, n) blocks `enc`, `d_enc`.  Same launches on the same data as the separate
 * calls: results are bit-identical.  Why: queued op by op from an interpreter the step costs more host
 * time than GPU time on a slow host.  One launch fewer than the separate calls: with n_params > 0
 * the table gradient's last conversion (int64 sums -> f32, bin_finalize_kernel) is done by the Adam kernel as it
 * fetches the gradient (the same expression: the same bits) -- `d_table` then does NOT hold the gradient of those
 * levels after the call. */
typedef struct mri_fused_step_args {
  const mri_grid_desc* grid;
  float* table;                                  /* (sum T_l, F), inside the flat parameter buffer */
  const float *w1, *b1, *w2, *b2, *w3, *b3;      /* decoder k_in -> hidden -> hidden -> 1 */
  float *d_table, *d_w1, *d_b1, *d_w2, *d_b2, *d_w3, *d_b3, *loss;
  int32_t hidden, bwd_method, counted, join_pending;
  const float* coords;                           /* (n, D) this step's batch */
  const float* target;                           /* (n) */
  int64_t n;
  float *enc, *d_enc;                            /* (2 L, n) each */
  void* tiny_ws;                                 /* mri_tiny_mlp_workspace_bytes */
  int64_t tiny_ws_bytes;
  void* bwd_ws;                                  /* mri_hashgrid_backward_workspace_bytes, this batch's */
  int64_t bwd_ws_bytes;
  float* absmax;                                 /* 32 zeroed floats, or NULL (pass over d_enc instead) */
  float *param, *grad, *exp_avg, *exp_avg_sq;    /* Adam: one range of the flat buffers */
  int64_t n_params;
  double lr, beta1, beta2, eps;
  int32_t step;                                  /* 1-based */
  float grad_scale;
  int64_t* next_idx;                             /* NULL: no batch is produced */
  float *next_coords, *next_target;
  int64_t next_n;
  void* next_bwd_ws;                             /* NULL: the next batch is not counted ahead */
  int64_t next_bwd_ws_bytes;
  float* next_absmax;
  uint64_t seed;                                 /* mri_sample_indices(seed, first, lo, hi, next_n) */
  int64_t first, lo, hi;
  int32_t dim, reserved;
  int64_t shape[MRI_MAX_DIM], axis_offset[MRI_MAX_DIM];
  const float *axes, *volume;                    /* mri_gather_batch */
  void *stream, *stream_side, *ev_fork, *ev_join;
  void* ev_phase[5];                             /* all NULL, or five timing events recorded on `stream` around
                                                    lookup | decoder | table gradient | Adam (a measuring caller
                                                    brackets the phases without leaving this call) */
  float grad_divisor;                            /* gradients of mean((y - t)^2) / grad_divisor; 0 = 1.  Data
                                                    parallel: the world size, with n_params = 0 -- the caller
                                                    reduces `grad` over the ranks and steps (mri_adam_step) */
  float reserved2;
} mri_fused_step_args;
int mri_fused_step(const mri_fused_step_args* args);
int64_t mri_fused_step_args_bytes(void); /* sizeof(mri_fused_step_args): lets a binding check its layout */

/* ---- coordinate-batch producer ------------------------------------------------------------
 * Replaces MriImage.__getitem__ + DataLoader(shuffle=True) collate (reference
 * datamodules.py:140-172, 198-205) with on-device generation.
 * mri_sample_indices: idx_out[i] = lo + perm_epoch((first + i) mod (hi-lo)), a keyed
 *   bijection of [0, hi-lo) (without-replacement shuffle, one key per epoch/seed).
 * mri_gather_batch: flat C-order voxel index -> coordinates (row-major (n, D)) through the
 *   per-axis linspace tables (`axes` = concatenation of the D axis arrays built with
 *   torch.linspace on the host, axis_offset[d] = start of axis d) and targets from `volume`.
 *   shape / axis_offset are HOST arrays of length D. */
int mri_sample_indices(uint64_t seed, int64_t first, int64_t lo, int64_t hi, int64_t n,
                       int64_t* idx_out, void* stream);
int mri_gather_batch(const int64_t* idx, int64_t n, int32_t dim, const int64_t* shape,
                     const float* axes, const int64_t* axis_offset, const float* volume,
                     float* coords_out, float* target_out /* may be NULL */, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MRI_INR_H */
