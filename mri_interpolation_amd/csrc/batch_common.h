// The on-device batch producer's arithmetic, shared by its own kernels (train_ops.hip: sample_kernel,
// gather_batch_kernel) and by the table gradient's fused produce-and-count kernel (hashgrid_bwd.hip).
// Replaces DataLoader(shuffle=True) + MriImage.__getitem__ (reference datamodules.py:140-172, 198-205).
#pragma once
#include "common.h"

namespace mri {

// Keyed bijection of [0, range): a 4-round Feistel network on the smallest even-width bit
// field that covers `range`, with cycle walking for values that fall outside.  One key =
// one shuffle of the data set (what DataLoader(shuffle=True) draws per epoch).
__device__ __forceinline__ uint32_t mix32(uint32_t x, uint32_t k) {
  x ^= k;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

__device__ __forceinline__ uint64_t feistel(uint64_t v, int half_bits, uint64_t key) {
  const uint64_t half_mask = (1ull << half_bits) - 1;
  uint32_t left = (uint32_t)(v >> half_bits), right = (uint32_t)(v & half_mask);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t k = (uint32_t)(key >> (16 * r)) ^ (0x9e3779b9u * (r + 1));
    const uint32_t f = mix32(right, k) & (uint32_t)half_mask;
    const uint32_t nl = right;
    right = left ^ f;
    left = nl;
  }
  return ((uint64_t)left << half_bits) | right;
}

// position i of the epoch's permutation of [lo, lo + range)
__device__ __forceinline__ int64_t sample_index(uint64_t key, int64_t first, int64_t lo, int64_t range, int half_bits,
                                                int64_t i) {
  uint64_t v = (uint64_t)((first + i) % range);
  do {
    v = feistel(v, half_bits, key);
  } while (v >= (uint64_t)range);  // cycle walking: the field is < 4x range, so ~1-4 rounds
  return lo + (int64_t)v;
}

struct ShapeTab {
  int64_t shape[MRI_MAX_DIM];
  int64_t axis_offset[MRI_MAX_DIM];
};

// scramble the user seed so that nearby seeds give unrelated keys
inline uint64_t sample_key(uint64_t seed) {
  uint64_t key = seed + 0x9E3779B97F4A7C15ull;
  key = (key ^ (key >> 30)) * 0xBF58476D1CE4E5B9ull;
  key = (key ^ (key >> 27)) * 0x94D049BB133111EBull;
  key ^= key >> 31;
  return key;
}

// even number of bits of the Feistel field for `range` (< 0: too large)
inline int sample_field_bits(int64_t range) {
  int bits = 2;
  while ((1ll << bits) < range) ++bits;
  if (bits & 1) ++bits;
  return bits <= 62 ? bits : -1;
}

}  // namespace mri
