// Declarations shared by the two families of fused SIREN chain kernels: siren_chain.hip (the activation image of a
// row tile in LDS, every width) and siren_rows.hip (H = 256: the activations of a wave's rows in registers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace mri {

constexpr int kKc = 16;        // contraction depth of a weight chunk: one bf16 MFMA step
constexpr int kMaxSine = MRI_SIREN_MAX_LAYERS;  // sine layers, the first one included
constexpr int kMaxIn = 8;

struct ChainArgs {
  const float* x;                   // (n, dim_in) row-major
  int64_t n;
  int dim_in, n_sine;
  const float* w[kMaxSine + 1];     // [0] (H, dim_in); [1 .. n_sine-1] (H, H); [n_sine] (1, H)
  const float* b[kMaxSine + 1];
  float w0_first, w0;
  float* act[kMaxSine];             // (n, H) per sine layer, or null (inference)
  float* deriv[kMaxSine];
  float* y;                         // (n)
  // loss mode (MODE 2): the head's backward runs in the forward kernel's tail, where the last sine
  // layer's output and derivative are still in registers
  const float* target;              // (n)
  float grad_scale, inv_n;          // 2 / (n_total divisor), 1 / n_total
  float* dz_last;                   // (n, H): dLoss / d(pre-activation of the last sine layer)
  float* partial;                   // [gridDim.x][fwd_slab_floats]
  const char* wsplit;               // split W of layers 1 .. n_sine-1 (split_matrix_bytes each)
  float* dy_ws;                     // siren_rows.hip, loss mode: dLoss/dy per row, in the workspace (rows_dy_offset)
};

// loss-mode slab: dW_head [H] | db_last [H] | db_head, loss (padded to 4)
__host__ __device__ inline int fwd_slab_floats(int hidden) { return 2 * hidden + 4; }

// Split weights (siren_split_weights_kernel): per H x H matrix, chunk kc = contraction indices
// [16 kc, 16 kc + 16), term planes h | m | l of [H rows][2 slots][8 bf16]; slot q of row n holds
// contraction indices 16 kc + 4 q + e and 16 kc + 8 + 4 q + e (e = 0..3): the eight positions lane
// half q of a 32x32x16 MFMA contracts when the other operand is read from the f32 image as two
// 16-byte fragments at k = 4 q and 8 + 4 q.  A chunk is contiguous: 3 x H x 32 bytes.
__host__ __device__ inline int64_t split_matrix_bytes(int H) { return (int64_t)(H / kKc) * 3 * H * 32; }

struct BwdArgs {
  const float* x;                  // (n, dim_in)
  const float* dy;                 // (n): dLoss / dy
  int64_t n;
  int dim_in, n_sine;
  const float* w[kMaxSine + 1];    // as ChainArgs
  const float* act_last;           // (n, H): output of the last sine layer
  const float* deriv[kMaxSine];    // (n, H) per sine layer: w0 cos(.)
  float* dz[kMaxSine];             // (n, H) for sine layers 1 .. n_sine-1 ([0] unused)
  float* partial;                  // [gridDim.x][bwd_slab_floats]
  const char* wtsplit;             // split W^T of layers 1 .. n_sine-1 (split_matrix_bytes each)
  int head_done;                   // dz[n_sine-1] is an INPUT (the forward kernel's loss mode wrote it)
  const float* dy_ws;              // siren_rows.hip: dLoss/dy per row, where the loss-mode forward kernel left it
};

// slab: dW_head [H] | db_head [1] (padded to 4) | db_l [n_sine][H] | dW_first [H][kMaxIn]
__host__ __device__ inline int bwd_slab_floats(int hidden, int n_sine) {
  return hidden + 4 + n_sine * hidden + hidden * kMaxIn;
}


struct WgradArgs {
  const float* dz;    // (n, H)
  const float* act;   // (n, H): the layer's input
  int64_t n;
  float* partial;     // [slabs][H * H]
};

// siren_rows.hip
bool rows_supported(int hidden, int n_sine);
int64_t rows_dy_offset(int64_t n, int hidden, int n_sine);  // bytes into the workspace's slab region
int rows_blocks(int64_t n);
int forward_rows(const ChainArgs& a, int mode, hipStream_t st);
bool rows_backward_supported(int hidden, int n_sine, int dim_in, int head_done);
int backward_rows(const BwdArgs& a, hipStream_t st);

}  // namespace mri
