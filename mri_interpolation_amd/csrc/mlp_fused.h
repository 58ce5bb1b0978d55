// Shared by the fused tiny-MLP kernels (mlp_fused.hip: f32 MFMA; mlp_x3.hip: bf16x3 MFMA).
#pragma once
#include "common.h"
#include "hashgrid_common.h"

namespace mri {

struct FusedArgs {
  const float* x;      // (k_in, n) feature-major
  const float* target; // (n)
  const float* w1; const float* b1;   // (H, k_in), (H)
  const float* w2; const float* b2;   // (H, H), (H)
  const float* w3; const float* b3;   // (1, H), (1)
  float* y;            // (n) predictions, optional
  float* dx;           // (k_in, n) feature-major, optional
  float* partial;      // [gridDim.x][slab] partial gradients + loss
  int64_t n;
  int64_t ld;          // leading dimension of x and dx (elements between feature rows), >= n
  int k_in;
  float grad_scale;    // 2 / (n * grad_divisor)
  int stagger;         // team kernel: segments team 1 runs behind team 0
  float inv_n;
  // team kernel, optional: x is being PRODUCED by a kernel running beside this one (the
  // signalling hash-grid lookup of hashgrid.hip); slice r = the rows round r of the workgroups
  // reads, complete when ready[r] >= ready_target
  const unsigned long long* ready;
  unsigned long long ready_target;
  int* status;         // set to 1 if a wait gave up (the producer never arrived)
  // mlp_x3.hip, optional: max |dx| per PAIR of feature rows (k_in / 2 non-negative floats, atomicMax'ed as
  // their bit patterns: the caller zeroes them) -- the scale the table-gradient records need per level
  unsigned int* dx_absmax;
};


// mlp_x3.hip, optional: the decoder looks its input features up in the hash grid itself (F = 2, so
// k_in = 2 n_levels <= 32): no lookup kernel, the features never reach HBM.  FusedArgs::x is unused then.
struct EncodeArgs {
  const float* coords = nullptr;  // (n, dim) row-major
  const float* table = nullptr;   // all levels, (rows, 2); rows < 2^29 (32-bit byte offsets)
  LevelTab tab;
  int n_levels = 0;
  int dim = 0;                    // 0: features come from FusedArgs::x
};

// slab layout (floats): dW1 [H*k_in] | db1 [H] | dW2 [H*H] | db2 [H] | dW3 [H] | db3 [1] | loss [1]
__host__ __device__ inline int slab_floats(int H, int k_in) { return H * k_in + H + H * H + H + H + 2; }

// mlp_x3.hip: k_in <= 32 -> 128 -> 128 -> 1 and -> 64 -> 64 -> 1 on the bf16 matrix pipe (three-term split operands), one
// 512-thread workgroup per CU, one slab per workgroup (train).  `blocks` <= x3_max_blocks(n).
bool x3_supported(int k_in, int hidden);
bool x3_addressable(const FusedArgs& a);  // 32-bit lane offsets inside x / dx
int x3_blocks(int64_t n);
int launch_tiny_mlp_x3(const FusedArgs& a, int hidden, bool train, int blocks, hipStream_t st);
bool x3_encode_supported(int dim, int n_features, int n_levels, int hidden);
int launch_tiny_mlp_x3_encoded(const FusedArgs& a, const EncodeArgs& e, int hidden, int blocks, hipStream_t st);

}  // namespace mri
