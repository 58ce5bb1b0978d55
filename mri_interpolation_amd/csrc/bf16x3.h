// f32-accurate products on the bf16 matrix pipe of gfx950 ("bf16x3").
//
// An f32 value x is split EXACTLY into three bf16 terms, x = h + m + l (h = bf16(x), m = bf16(x - h),
// l = bf16(x - h - m): 3 x 8 significant bits cover f32's 24, the subtractions are exact).  A product
// a b is then the sum of nine bf16 x bf16 products, each exact in f32; the three smallest (m l, l m,
// l l <= 2^-25 |a b|) are dropped and the other six are accumulated in f32 by v_mfma_f32_*_bf16.
// Measured against float64 (tools/probes/bf16x3_probe.hip, K = 256): rms error 1.9e-8 of sum |a b|,
// the f32 MFMA's own is 2.4e-8 -- the same arithmetic quality as v_mfma_f32_32x32x2_f32, at 6/16 of
// its cycles, and unlike the f32 MFMA (which occupies the vector ALU for all of its 64 cycles) a
// bf16 MFMA holds the SIMD's issue port for 8 of its 16 / 32 cycles: epilogue VALU work of another
// wave runs beside it.
//
// Operand maps used below (cdna_hip_programming.md, fragment layout), li = lane & 15, g = lane >> 4:
//   v_mfma_f32_16x16x32_bf16:  A[row li][k = 8 g + j], B[k = 8 g + j][col li], j = 0..7 in the lane's
//   four operand registers;  C[row 4 g + r][col li] in accumulator register r.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mri {
namespace x3 {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// two floats -> one dword of two bf16 (round to nearest even; `a` in the low half)
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  f32x2 v = {a, b};
  bf16x2 r = __builtin_convertvector(v, bf16x2);
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ float lo_f32(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float hi_f32(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }

// exact three-way split of a pair
__device__ __forceinline__ void split2(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = pack_bf16(a, b);
  a -= lo_f32(h), b -= hi_f32(h);
  m = pack_bf16(a, b);
  a -= lo_f32(m), b -= hi_f32(m);
  l = pack_bf16(a, b);
}

// One operand fragment (8 contraction indices of one row / column) in its three terms.
struct Frag {
  u32x4 h, m, l;
};

__device__ __forceinline__ Frag split8(const float (&v)[8]) {
  Frag f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint32_t h, m, l;
    split2(v[2 * q], v[2 * q + 1], h, m, l);
    f.h[q] = h, f.m[q] = m, f.l[q] = l;
  }
  return f;
}

__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// c += a b over the fragment's 32 contraction indices: the six products, smallest first.
__device__ __forceinline__ f32x4 mma6(const Frag& a, const Frag& b, f32x4 c) {
  c = mfma16(a.l, b.h, c);
  c = mfma16(a.h, b.l, c);
  c = mfma16(a.m, b.m, c);
  c = mfma16(a.m, b.h, c);
  c = mfma16(a.h, b.m, c);
  c = mfma16(a.h, b.h, c);
  return c;
}

// The three largest of the nine products only: per-product error ~2^-16 |a b| (the 2^-16 terms h l, l h,
// m m are dropped).  NOT f32-accurate per product; measured as an option for the batch-contracting weight
// gradients only (mlp_x3.hip, MRI_DW_TERMS), where 2^18 such errors of random sign meet in one sum.
__device__ __forceinline__ f32x4 mma3(const Frag& a, const Frag& b, f32x4 c) {
  c = mfma16(a.m, b.h, c);
  c = mfma16(a.h, b.m, c);
  c = mfma16(a.h, b.h, c);
  return c;
}

// ---- LDS images ------------------------------------------------------------------------------------
// An activation image holds one term of a [rows][128] tile of bf16, rows of 256 bytes; 16-byte chunk
// `ch` (8 columns) of row `row` sits at chunk ch ^ sw(row) of the row.  With this XOR (found by
// search over the linear maps, tools/probes/x3_layout_probe.hip checks the reads) every access of
// the kernel is conflict-free on the 64-bank read path and 2-way (which costs nothing extra, the
// store's register transfer is longer) on the 32-bank write path:
//   row read   ds_read_b128        lane (row li (+16), chunk 4 s + g): B operand of the layer chains
//   transposed ds_read_b64_tr_b16  4 rows x 16 columns per 16-lane group: operands that contract rows
//   store      ds_write_b64        lane (row li (+16)) stores 4 consecutive columns
constexpr int kImgRowBytes = 256;
__device__ __forceinline__ int sw(int row) { return ((row & 3) << 1) ^ (((row >> 2) & 1) * 9); }
__device__ __forceinline__ int img_off(int row, int ch) {
  return kImgRowBytes * row + 16 * (ch ^ sw(row));
}
constexpr int kImgBytes = 32 * kImgRowBytes;  // one term of a 32-row tile

// A narrow image: [rows][32] bf16 (the decoder's input tile), rows of 64 bytes, four chunks.
__device__ __forceinline__ int img32_off(int row, int ch) {
  return 64 * row + 16 * (ch ^ ((4 - ((row >> 2) & 3)) & 3));
}
constexpr int kImg32Bytes = 32 * 64;

__device__ __forceinline__ u32x4 lds_read_b128(const char* p) {
  return *reinterpret_cast<const u32x4*>(p);
}

// the hardware transpose read: see tr_frag below for the addressing
__device__ __forceinline__ u32x2 lds_read_tr(const char* p) {
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  return __builtin_bit_cast(u32x2, v);
}

// Fragment of an operand whose CONTRACTION index is the image's row (batch) index and whose free
// index is 16 consecutive columns starting at chunk c0 (two chunks): lane (li, g) receives, for
// column 8 c0 + li, rows 4 g .. 4 g + 3 (elements 0..3) and 16 + 4 g .. 16 + 4 g + 3 (elements 4..7):
// contraction position 8 g + j <-> row 16 (j >> 2) + 4 g + (j & 3), the same for both operands of a
// product.  Per 16-lane group ds_read_b64_tr_b16 takes a 4-row x 16-column block: lane 4 q + p of
// the group supplies the address of row q, columns 4 p .. 4 p + 3.
template <class OffFn>
__device__ __forceinline__ u32x4 tr_frag(const char* img, int c0, int lane, OffFn off) {
  const int li = lane & 15, g = lane >> 4, q = li >> 2, p = li & 3;
  const int row = 4 * g + q;
  const u32x2 a = lds_read_tr(img + off(row, c0 + (p >> 1)) + 8 * (p & 1));
  const u32x2 b = lds_read_tr(img + off(16 + row, c0 + (p >> 1)) + 8 * (p & 1));
  u32x4 r = {a[0], a[1], b[0], b[1]};
  return r;
}

}  // namespace x3
}  // namespace mri
