// Fused tiny-MLP (ReLU) forward + MSE + backward on the bf16 matrix pipe, f32-accurate ("bf16x3",
// see bf16x3.h): the decoder of BASELINE configs 2 / 4 / 5 (k_in <= 32 -> H -> H -> 1, H = 128 or 64;
// reference models.py:46-66 training_step, 730-744 HashMLP decoder) in one persistent kernel per step.
// The description below is for H = 128; the kernel's head comment has the H = 64 mapping.
//
// Same contract as mlp_fused.hip's f32-MFMA kernels (FusedArgs, one slab of partial gradients per
// workgroup, summed in a fixed order by slab_reduce_kernel), different machine mapping.  Every f32
// operand is split exactly into three bf16 terms and each product runs as six bf16 MFMAs: 6/16 of
// the f32 MFMA's cycles at the same accuracy.  The split costs LDS (6 bytes per activation instead
// of 4, 96 KiB for W2's terms alone), so the roles of mlp_fused.hip are swapped:
//
//   * WEIGHTS LIVE IN REGISTERS.  The workgroup is 8 waves, wave w owns hidden units 16 w .. 16 w + 15
//     in both hidden layers and holds, as ready-made MFMA operand fragments, the two larger terms
//     of its rows of W2 (forward) and of its columns of W2 (backward): 64 registers, loaded and
//     split once per launch.  W2's third term (one MFMA in six) and all of W1 are LDS images read
//     row-wise and transposed.  Its slices of dW2 (16 x 128) and dW1 (16 x 32) are MFMA accumulators
//     that live across all tiles.
//   * ACTIVATIONS LIVE IN LDS as three-term bf16 images [32 rows][128 units] (x: [32][32]), written
//     once by the wave that produced the units and read by all eight: row-wise (ds_read_b128) as
//     the B operand of the layer chains, transposed (ds_read_b64_tr_b16) where the batch index is
//     contracted (dW2, dW1).  One XOR swizzle serves both kinds of read and the 8-byte stores.
//   * v_mfma_f32_16x16x32_bf16 with the weights in the A slot: the accumulator then holds, per lane,
//     four CONSECUTIVE units of one row -- one packed 8-byte LDS store per term.
//
// A 32-row tile takes three workgroup barriers (see the schedule in the kernel); two waves per SIMD
// cover each other's LDS latencies, and VALU epilogues (bias, ReLU, the exact split: 5.5
// instructions per element) of one wave run beside the other's MFMAs.  DESIGN.md 4.3-4.4 and EXPERIMENTS.md (Part II 4.9) have the
// measurements and what was tried.
#include <algorithm>

#include "bf16x3.h"
#include "mlp_fused.h"

namespace mri {
namespace {
using namespace x3;

#ifndef MRI_DW_TERMS  // A/B builds: bf16 products per weight-gradient product (6 = f32-accurate; 3: see bf16x3.h mma3)
#define MRI_DW_TERMS 6
#endif
__device__ __forceinline__ f32x4 mma_dw(const Frag& a, const Frag& b, f32x4 c) {
  return MRI_DW_TERMS == 3 ? mma3(a, b, c) : mma6(a, b, c);
}

constexpr int kX3Threads = 512;
constexpr int kX3Rows = 32;
constexpr int kX3H = 128;

// Segment timing for tools/x3_segments.py (a tools-only build with -DMRI_X3_PROFILE; the shipped
// library compiles these to nothing): shader-clock cycles per barrier-separated segment, split
// into work (mark -> barrier entry) and wait (barrier entry -> exit), summed over tiles.
#ifdef MRI_X3_PROFILE
__device__ long long* g_x3_profile = nullptr;
constexpr int kX3ProfSlots = 32;
#ifdef MRI_X3_CLOCK_ONLY
// tools/x3_clock.py: the clock the tile loop runs at = shader cycles (s_memtime) per 100 MHz tick (s_memrealtime),
// ONE pair of stamps around the whole loop (slots 24 / 25), none inside it (MI355X_MICROARCH.md, DVFS give-back 6)
#define X3P_START
#define X3P_BEGIN long long p_acc[kX3ProfSlots] = {}; const long long p_c0 = __builtin_amdgcn_s_memtime(), \
                                                                    p_r0 = __builtin_amdgcn_s_memrealtime();
#define X3P_MARK(i) if ((i) == 18) { p_acc[24] = __builtin_amdgcn_s_memtime() - p_c0; \
                                     p_acc[25] = __builtin_amdgcn_s_memrealtime() - p_r0; }
#define X3P_SYNC(i) __syncthreads();
#else
#define X3P_START const long long p_t0 = clock64();
#define X3P_BEGIN long long p_t = clock64(); long long p_acc[kX3ProfSlots] = {}; p_acc[20] = p_t - p_t0;
#define X3P_MARK(i) { const long long p_n = clock64(); p_acc[i] += p_n - p_t; p_t = p_n; }
#define X3P_SYNC(i) { X3P_MARK(2 * (i)) __syncthreads(); X3P_MARK(2 * (i) + 1) }
#endif
#define X3P_END                                                                               \
  if (g_x3_profile && (threadIdx.x & 63) == 0) {                                              \
    long long* dst = g_x3_profile + ((int64_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * kX3ProfSlots; \
    for (int q = 0; q < kX3ProfSlots; ++q) dst[q] = p_acc[q];                                 \
  }
#else
#define X3P_START
#define X3P_BEGIN
#define X3P_MARK(i)
#define X3P_SYNC(i) __syncthreads();
#define X3P_END
#endif

struct __attribute__((aligned(16))) X3Smem {
  char x[2][3][kImg32Bytes];  // input tile, double-buffered (the next tile is staged during S7)
  char h1[3][kImgBytes];
  char z2[3][kImgBytes];      // dLoss / d(pre-activation 2)
  char z1[3][kImgBytes];      // dLoss / d(pre-activation 1)
  char w1[3][4 * kImg32Bytes];  // W1: [hidden unit][input feature]: A operand of layer 1 (row reads) and of dx (transposed)
  char w2l[4 * kImgBytes];      // the smallest term of W2 [out][in]: read row-wise (layer 2) and transposed (dz1)
  float ypart[8][kX3Rows];
  float tgt[2][kX3Rows];
  float b1[kX3H], b2[kX3H], w3[kX3H];  // read per tile (registers are the scarce resource here)
  float gc[2][kX3Rows][4];             // gather mode: coordinates of the next two tiles
  float pa[kX3Threads][2];             // gather mode: a thread's partial sums, parked between two segments
  uint32_t dxmax[16];                  // max |dx| per feature pair (bit patterns), a.dx_absmax
};

// Three-term fragments at byte offset `off` of an image whose terms are TERM bytes apart.
template <int TERM>
__device__ __forceinline__ Frag ld_row(const char* img, int off) {
  Frag f;
  f.h = lds_read_b128(img + off), f.m = lds_read_b128(img + TERM + off), f.l = lds_read_b128(img + 2 * TERM + off);
  return f;
}
// transposed: rows 4 g .. + 3 at `off`, rows 16 + 4 g .. + 3 HALF bytes further (see tr_frag, bf16x3.h)
template <int TERM, int HALF>
__device__ __forceinline__ Frag ld_tr(const char* img, int off) {
  Frag f;
  u32x2 a0 = lds_read_tr(img + off), a1 = lds_read_tr(img + off + HALF);
  u32x2 b0 = lds_read_tr(img + TERM + off), b1 = lds_read_tr(img + TERM + off + HALF);
  u32x2 c0 = lds_read_tr(img + 2 * TERM + off), c1 = lds_read_tr(img + 2 * TERM + off + HALF);
  f.h = u32x4{a0[0], a0[1], a1[0], a1[1]};
  f.m = u32x4{b0[0], b0[1], b1[0], b1[1]};
  f.l = u32x4{c0[0], c0[1], c1[0], c1[1]};
  return f;
}
__device__ __forceinline__ void st4(char* img, int off, const float (&v)[4]) {
  uint32_t h0, m0, l0, h1, m1, l1;
  split2(v[0], v[1], h0, m0, l0);
  split2(v[2], v[3], h1, m1, l1);
  *reinterpret_cast<u32x2*>(img + off) = u32x2{h0, h1};
  *reinterpret_cast<u32x2*>(img + kImgBytes + off) = u32x2{m0, m1};
  *reinterpret_cast<u32x2*>(img + 2 * kImgBytes + off) = u32x2{l0, l1};
}

// v summed over the four 16-lane groups (lanes li, li + 16, li + 32, li + 48), in every lane: two
// row swaps (v_permlane16_swap / v_permlane32_swap: one VALU instruction each, no LDS round trip)
__device__ __forceinline__ float sum_groups(float v) {
  const uint32_t b = __float_as_uint(v);
  auto p = __builtin_amdgcn_permlane16_swap(b, b, false, false);
  v = __uint_as_float(p[0]) + __uint_as_float(p[1]);
  const uint32_t c = __float_as_uint(v);
  auto q = __builtin_amdgcn_permlane32_swap(c, c, false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

template <bool TRAIN, int U, int H, int GD = 0>
__global__ __launch_bounds__(kX3Threads / U) void tiny_mlp_x3_kernel(const FusedArgs a, const EncodeArgs e) {
  // GD > 0 (gather mode, training, U = 1): the input features are looked up in the GD-dimensional hash
  // grid `e` by the thread that stages them -- thread (row tid & 31, level tid >> 5) of the staging map
  // below owns exactly one (coordinate, level) pair and its two features -- see "gather mode" below.
  constexpr bool GATHER = GD > 0;
  static_assert(!GATHER || (TRAIN && U == 1), "gather mode: the 8-wave training kernel");
  // U = strips of 16 hidden units per wave: 1 -> 8 waves (two per SIMD, 256 registers each),
  // 2 -> 4 waves (one per SIMD, 512 registers; every activation fragment feeds two strips).
  // H = 128: wave w owns strip w (U = 1) for both 16-row halves of the tile.  H = 64 (U = 1): four
  // strips; wave w owns strip w & 3 for the tile's row half w >> 2 (TL = 1 half per wave), on the same
  // 256-byte image rows (units 64..127 unused), so every address and swizzle below is shared.  The
  // batch-contracting products (dW2, dW1) then run over the wave's OWN 16 rows with the other half of
  // the 32-deep step zeroed (its partner's rows are not behind a barrier); the two partial sums of a
  // strip meet in the epilogue.
  static_assert((H == 128) || (H == 64 && U == 1), "hidden width");
  constexpr int WAVES = 8 / U, THREADS = kX3Threads / U;
  constexpr int KS = H / 32, KT = H / 16;          // 32-deep contraction steps / 16-unit tiles over the hidden units
  constexpr int TL = H == 128 ? 2 : 1;             // row halves of the tile a wave works on
  constexpr int YS = H == 128 ? WAVES : 4;         // y shares per row
  constexpr int IMG = kImgBytes, IMG32 = kImg32Bytes;
  __shared__ X3Smem sm;
  X3P_START
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g = lane >> 4;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const int th = H == 128 ? 0 : w >> 2;            // first row half of this wave
  auto strip = [&](int u) { return H == 128 ? U * w + u : (w & 3); };
  auto unit0 = [&](int u) { return 16 * strip(u); };  // first hidden unit of the wave's strip u
  const int yslot = H == 128 ? w : (w & 3);
  const bool y_owner = yslot == 0;                 // the wave(s) that store y and sum the loss

  // ---- resident weight fragments (A operands: row = unit unit0(u) + li, 8 g + j = contraction)
  // W2's two larger terms stay in registers (64 per strip, both orientations); its smallest term, used
  // by one MFMA in six, is read from an LDS image: all three in registers left no room for the fragment
  // double buffers, and a spilled fragment's reload waits behind every load in flight (vmcnt)
  u32x4 w2f_h[U][KS], w2f_m[U][KS], w2t_h[U][KS], w2t_m[U][KS];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int n = unit0(u) + li;
    float v[8];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = a.w2[n * H + 32 * s + 8 * g + j];
      const Frag f = split8(v);
      w2f_h[u][s] = f.h, w2f_m[u][s] = f.m;
    }
    if (TRAIN) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {  // dz1 = dz2 W2: row = input unit n of layer 2, contraction = its output unit
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = a.w2[(32 * s + 8 * g + j) * H + n];
        const Frag f = split8(v);
        w2t_h[u][s] = f.h, w2t_m[u][s] = f.m;
      }
    }
  }
  for (int e = tid; e < H * (H / 4); e += THREADS) {  // W2's third term: row = output unit, 4 inputs per store
    const int n = e / (H / 4), c4 = e % (H / 4);
    const float* v = a.w2 + n * H + 4 * c4;  // 4-byte loads: parameter views need not be 16-byte aligned
    uint32_t h0, m0, l0, h1, m1, l1;
    split2(v[0], v[1], h0, m0, l0);
    split2(v[2], v[3], h1, m1, l1);
    *reinterpret_cast<u32x2*>(sm.w2l + img_off(n, c4 >> 1) + 8 * (c4 & 1)) = u32x2{l0, l1};
  }
  for (int e = tid; e < H * 16; e += THREADS) {  // W1 image: row = hidden unit, pairs of input features
    const int n = e >> 4, kp = e & 15;
    const float v0 = 2 * kp < a.k_in ? a.w1[n * a.k_in + 2 * kp] : 0.f;
    const float v1 = 2 * kp + 1 < a.k_in ? a.w1[n * a.k_in + 2 * kp + 1] : 0.f;
    uint32_t h, m, l;
    split2(v0, v1, h, m, l);
    const int off = img32_off(n, kp >> 2) + 4 * (kp & 3);
    *reinterpret_cast<uint32_t*>(sm.w1[0] + off) = h;
    *reinterpret_cast<uint32_t*>(sm.w1[1] + off) = m;
    *reinterpret_cast<uint32_t*>(sm.w1[2] + off) = l;
  }
  if (tid < H) sm.b1[tid] = a.b1[tid], sm.b2[tid] = a.b2[tid], sm.w3[tid] = a.w3[tid];
  if (tid < 16) sm.dxmax[tid] = 0u;
  const float b3 = a.b3[0];

  f32x4 g_w2[U][KT], g_w1[U][2];
  float g_b1[U][4], g_b2[U][4], g_w3[U][4], g_b3 = 0.f, loss = 0.f;
#pragma unroll
  for (int u = 0; u < U; ++u) {
#pragma unroll
    for (int q = 0; q < KT; ++q) g_w2[u][q] = zero4;
    g_w1[u][0] = g_w1[u][1] = zero4;
#pragma unroll
    for (int r = 0; r < 4; ++r) g_b1[u][r] = g_b2[u][r] = g_w3[u][r] = 0.f;
  }

  // ---- LDS addresses.  The XOR swizzles split into a tile-invariant lane part (five registers, below)
  // and a compile-time part that is XORed / added in at the use: chunk (4 s + g) ^ sw = 4 (s ^ sw>>2) +
  // (g ^ sw&3), and 64 (s ^ q) = 64 s ^ 64 q, so row-read address (s, t) = (a_row ^ 64 s) + 4096 t; the
  // transposed reads of 16 columns from chunk 2 k: (a_tr ^ 32 k), rows 16.. 4096 further.  Built
  // naively from (row, chunk) these addresses were 240 of the tile's 600 VALU instructions.
  const int q4 = li >> 2, p4 = li & 3, swl = sw(li), trow = 4 * g + q4, swr = sw(trow), swxg = (4 - g) & 3;
  int a_row = 256 * li + 16 * (g ^ (swl & 3)) + 64 * (swl >> 2);
  int a_tr = 256 * trow + 8 * (p4 & 1) + 16 * ((p4 >> 1) ^ (swr & 1)) + 16 * (swr & 14);
  int a_row32 = 64 * li + 16 * (g ^ ((4 - q4) & 3));                                   // + 1024 t
  int a_tr32 = 64 * trow + 8 * (p4 & 1) + 16 * ((p4 >> 1) ^ (swxg & 1)) + 16 * (swxg & 2);  // ^ 32 k, + 1024
  int a_out[U];                                                                        // + 4096 t
#pragma unroll
  for (int u = 0; u < U; ++u)
    a_out[u] = 256 * li + 8 * (g & 1) + 16 * ((g >> 1) ^ (swl & 1)) + ((32 * strip(u)) ^ (16 * (swl & 14)));
  // dz2 / dz1 columns of the wave's strip as the A operand of a batch-contracting product: every row
  // of the tile it wrote itself -- all 32 (H = 128), or its half with the other half of the step zero
  auto own_tr = [&](const char* img, int off) {
    if (H == 128) return ld_tr<IMG, 4096>(img, off);
    Frag f;
    const u32x2 x0 = lds_read_tr(img + off + 4096 * th), x1 = lds_read_tr(img + IMG + off + 4096 * th),
                x2 = lds_read_tr(img + 2 * IMG + off + 4096 * th);
    const uint32_t keep_lo = th ? 0u : 0xffffffffu, keep_hi = ~keep_lo;
    f.h = u32x4{x0[0] & keep_lo, x0[1] & keep_lo, x0[0] & keep_hi, x0[1] & keep_hi};
    f.m = u32x4{x1[0] & keep_lo, x1[1] & keep_lo, x1[0] & keep_hi, x1[1] & keep_hi};
    f.l = u32x4{x2[0] & keep_lo, x2[1] & keep_lo, x2[0] & keep_hi, x2[1] & keep_hi};
    return f;
  };
  char* const smb = reinterpret_cast<char*>(&sm);
  char* const i_h1 = smb + offsetof(X3Smem, h1);
  char* const i_z2 = smb + offsetof(X3Smem, z2);
  char* const i_z1 = smb + offsetof(X3Smem, z1);
  const char* const i_w2l = smb + offsetof(X3Smem, w2l);
  const char* const i_w1 = smb + offsetof(X3Smem, w1);

  // ---- input staging.  Global addresses are a wave-uniform base (SGPRs: the tile's first row) plus a
  // 32-bit lane offset rebuilt at each use: per-lane 64-bit pointers cost six registers here, were
  // spilled, and their reloads serialised the loop top behind the input loads' HBM latency.
  const uint32_t ld32 = (uint32_t)a.ld;  // k_in * ld < 2^30: checked by the launcher
  float xv[U][2], xt = 0.f;
  auto load_x = [&](int64_t m0) {
    int t_op = tid;
    asm volatile("" : "+v"(t_op));
    const int sb = t_op & 31, skp = t_op >> 5;
    const float* __restrict__ xs = a.x + m0;
    const bool live = sb < a.n - m0;
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int k = 2 * (skp + (16 / U) * q);
      xv[q][0] = (live && k < a.k_in) ? xs[(uint32_t)k * ld32 + sb] : 0.f;
      xv[q][1] = (live && k + 1 < a.k_in) ? xs[(uint32_t)(k + 1) * ld32 + sb] : 0.f;
    }
    if (TRAIN) xt = (skp == 0 && live) ? (a.target + m0)[sb] : 0.f;
  };
  auto store_x = [&](int buf) {
    int t_op = tid;
    asm volatile("" : "+v"(t_op));
    const int sb = t_op & 31, skp = t_op >> 5;
#pragma unroll
    for (int q = 0; q < U; ++q) {
      uint32_t h, m, l;
      split2(xv[q][0], xv[q][1], h, m, l);
      const int kp = skp + (16 / U) * q;
      const int off = img32_off(sb, kp >> 2) + 4 * (kp & 3);
      *reinterpret_cast<uint32_t*>(sm.x[buf][0] + off) = h;
      *reinterpret_cast<uint32_t*>(sm.x[buf][1] + off) = m;
      *reinterpret_cast<uint32_t*>(sm.x[buf][2] + off) = l;
    }
    if (TRAIN && skp == 0) sm.tgt[buf][sb] = xt;
  };

  // ---- gather mode.  The (coordinate, level) pair of this thread needs 2^GD table rows of 8 bytes.  They
  // are fetched in two batches, corners with the lower (xc = 0) and the upper (xc = 1) vertex on axis 0
  // -- the partial sums and their order are those of hashgrid_fwd_pair_kernel, so the features are
  // bit-identical to the lookup kernel's -- and each batch spends at least a segment in flight:
  //   S5(i): take batch 1 of tile i+1 -> features -> x image;      issue batch 0 of tile i+2
  //   S7(i): take batch 0 of tile i+2 -> partial sums, parked;     issue batch 1 of tile i+2
  // What stays in registers between the segments is the batch in flight; the coordinates of the two
  // upcoming tiles live in LDS (sm.gc, written a barrier before their first use), level constants
  // come from the kernel arguments through scalar loads at each use.
  constexpr int NB = GATHER ? 1 << (GD - 1) : 1;  // corners per batch
  float2 gv[NB];
  float gcr = 0.f;  // a coordinate of tile i+3 on its way to sm.gc (threads < 32 GD)
  struct Lvl {
    float res[GATHER ? GD : 1];
    uint32_t size, magic, rows;
    bool pow2, on;
  };
  auto level_consts = [&]() {
    Lvl L;
    int t_op = tid;
    asm volatile("" : "+v"(t_op));  // re-derived per use: nothing of this is kept across segments
    const int wu = __builtin_amdgcn_readfirstlane(t_op >> 6);
    const int la = min(2 * wu, e.n_levels - 1), lb = min(2 * wu + 1, e.n_levels - 1);
    const bool hi = (t_op & 32) != 0;
#pragma unroll
    for (int d = 0; d < (GATHER ? GD : 1); ++d) L.res[d] = hi ? e.tab.res[lb][d] : e.tab.res[la][d];
    L.size = hi ? e.tab.size[lb] : e.tab.size[la];
    L.magic = hi ? e.tab.magic[lb] : e.tab.magic[la];
    L.pow2 = (hi ? e.tab.pow2[lb] : e.tab.pow2[la]) != 0;
    L.rows = (uint32_t)(hi ? e.tab.offset[lb] : e.tab.offset[la]);
    L.on = (t_op >> 5) < e.n_levels;
    return L;
  };
  auto g_cell = [&](int cb, const Lvl& L) {
    int t_op = tid;
    asm volatile("" : "+v"(t_op));
    float p[GATHER ? GD : 1];
#pragma unroll
    for (int d = 0; d < (GATHER ? GD : 1); ++d) p[d] = sm.gc[cb][t_op & 31][d];
    return locate<(GATHER ? GD : 1)>(p, 0, L.res);
  };
  auto g_issue = [&](int cb, int xc) {
    if constexpr (GATHER) {
      const Lvl L = level_consts();
      const Cell<GD> c = g_cell(cb, L);
      const float2* __restrict__ rows2 = reinterpret_cast<const float2*>(e.table);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        uint32_t h;
        float wgt;
        corner<GD>(c, (nb << 1) | xc, h, wgt);
        gv[nb] = rows2[L.rows + slot_of(h, L.size, L.magic, L.pow2)];  // uniform base + 32-bit offset
      }
    }
  };
  auto g_take = [&](int cb, int xc, float& p0, float& p1) {
    if constexpr (GATHER) {
      const Lvl L = level_consts();
      const Cell<GD> c = g_cell(cb, L);
      p0 = 0.f, p1 = 0.f;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        uint32_t h;
        float wgt;
        corner<GD>(c, (nb << 1) | xc, h, wgt);
        p0 = p0 + gv[nb].x * wgt;
        p1 = p1 + gv[nb].y * wgt;
      }
    }
  };
  // coordinates of the tile at m0: global -> register (threads < 32 GD, one float each, contiguous)
  auto g_coords_load = [&](int64_t m0) {
    if constexpr (GATHER) {
      int t_op = tid;
      asm volatile("" : "+v"(t_op));
      const bool live = t_op < 32 * GD && t_op / GD < a.n - m0;
      gcr = live ? (e.coords + m0 * GD)[t_op] : 0.f;
    }
  };
  auto g_coords_store = [&](int cb) {
    if constexpr (GATHER) {
      int t_op = tid;
      asm volatile("" : "+v"(t_op));
      if (t_op < 32 * GD) sm.gc[cb][t_op / GD][t_op % GD] = gcr;
    }
  };
  // batch 1 of the tile staged next has landed: features = parked batch-0 sums + these -> x image
  auto g_finish = [&](int cb, int xbuf, int64_t m0, float pa0, float pa1) {
    if constexpr (GATHER) {
      float pb0, pb1;
      g_take(cb, 1, pb0, pb1);
      int t_op = tid;
      asm volatile("" : "+v"(t_op));
      const bool on = (t_op >> 5) < e.n_levels && (t_op & 31) < a.n - m0;
      xv[0][0] = on ? pa0 + pb0 : 0.f;  // as hashgrid_fwd_pair_kernel: own partial sum + the partner's
      xv[0][1] = on ? pb1 + pa1 : 0.f;
      store_x(xbuf);
    }
  };
  auto g_target = [&](int64_t m0) {
    int t_op = tid;
    asm volatile("" : "+v"(t_op));
    xt = (t_op < 32 && t_op < a.n - m0) ? (a.target + m0)[t_op] : 0.f;
  };

  // Fragments are fetched one group ahead of the MFMAs that use them, into the other half of a
  // two-entry buffer (pinned with sched_barriers: left alone, hipcc waits for each read right in
  // front of its MFMA and exposes the LDS latency two or three times per six MFMAs).
#define X3_PIN __builtin_amdgcn_sched_barrier(0);
  // ---- layer 1 of the tile staged in x[xbuf]: h1 = relu(x W1^T + b1) -> image, sign bits -> mask1 ----
  uint32_t mask1 = 0;
  auto layer1 = [&](int xbuf) {
    const char* const i_x = smb + offsetof(X3Smem, x) + xbuf * (3 * IMG32);
    Frag w1f[U], xb[TL];
#pragma unroll
    for (int u = 0; u < U; ++u) w1f[u] = ld_row<4 * IMG32>(i_w1, a_row32 + 1024 * strip(u));
#pragma unroll
    for (int tt = 0; tt < TL; ++tt) xb[tt] = ld_row<IMG32>(i_x, a_row32 + 1024 * (th + tt));
    f32x4 bias[U];
#pragma unroll
    for (int u = 0; u < U; ++u) bias[u] = *reinterpret_cast<const f32x4*>(&sm.b1[unit0(u) + 4 * g]);
    X3_PIN
    mask1 = 0;
#pragma unroll
    for (int tt = 0; tt < TL; ++tt) {
      f32x4 c[U];
#pragma unroll
      for (int u = 0; u < U; ++u) c[u] = mma6(w1f[u], xb[tt], zero4);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float h[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          h[r] = fmaxf(c[u][r] + bias[u][r], 0.f);
          mask1 |= (h[r] > 0.f ? 1u : 0u) << (8 * u + 4 * tt + r);
        }
        st4(i_h1, a_out[u] + 4096 * (th + tt), h);
      }
    }
  };

  // A tile takes three (H = 128) or four workgroup barriers.  A wave's dW2 / dW1 products need the other waves' h1 / x
  // but only ITS OWN columns of dz2 / dz1 (read back transposed from the image it has just written:
  // its own LDS writes, which the LDS executes in order: no barrier), so they run in the segment that produces them, beside the
  // latency-bound loss / ReLU-mask work of the other wave on the SIMD; the next tile's layer 1 runs
  // beside dx.
  //   H = 128 (EARLY_L1), three barriers:
  //   S7 dx of tile i-1; S2 layer 2 -> y shares | B2 | S5 y, loss, dz2 -> image; stage x(i+1); dW2 |
  //   B3 | S6 dz1 -> image; dW1; layer 1 of tile i+1 | B4 | S7 dx; S2 of tile i+1 ...
  //   H = 64 and inference, four: layer 1 of the next tile behind dx, a barrier B0 in front of layer 2
  //   (the H = 128 order until round 3: 11.55 k cycles per tile, S7 2.56 k of them at a third of the pipe)
  // EARLY_L1 (training, H = 128): three barriers, layer 1 of the next tile before B4 -- config 4: 11.55 k ->
  // 11.07 k cycles per tile, kernel 0.185 -> 0.179 ms, bit-identical results; the 64-wide decoder (a quarter
  // of the matrix work per row, latency-bound) measured 2 % slower that way and keeps the four-barrier order.
  constexpr bool EARLY_L1 = TRAIN && H == 128;
  const int64_t tiles = (a.n + kX3Rows - 1) / kX3Rows;
  const int64_t stride = gridDim.x;
  int buf = 0;
  if constexpr (GATHER) {
    const int64_t t0 = blockIdx.x, t1 = t0 + stride, t2 = t1 + stride;
    g_coords_load(t0 * kX3Rows);
    g_coords_store(0);
    if (t1 < tiles) {
      g_coords_load(t1 * kX3Rows);
      g_coords_store(1);
    }
    __syncthreads();
    float pa0, pa1;
    g_issue(0, 0);
    g_take(0, 0, pa0, pa1);
    g_issue(0, 1);
    g_target(t0 * kX3Rows);
    g_finish(0, 0, t0 * kX3Rows, pa0, pa1);
    if (t1 < tiles) {  // loop invariant: batch 1 of the next tile in flight, its batch-0 sums parked
      g_issue(1, 0);
      g_take(1, 0, pa0, pa1);
      *reinterpret_cast<float2*>(&sm.pa[tid][0]) = make_float2(pa0, pa1);
      g_issue(1, 1);
      g_target(t1 * kX3Rows);
    }
    if (t2 < tiles) g_coords_load(t2 * kX3Rows);  // -> sm.gc[0] in S2 of the first tile
  } else {
    load_x((int64_t)blockIdx.x * kX3Rows);
    store_x(0);
    if (blockIdx.x + stride < tiles) load_x((blockIdx.x + stride) * kX3Rows);  // stays in registers
  }
  __syncthreads();
  layer1(0);
  if (EARLY_L1) __syncthreads();  // h1 of the first tile complete (otherwise: B0 at the loop top)
  X3P_BEGIN
  for (int64_t tile = blockIdx.x; tile < tiles; tile += stride, buf ^= 1) {
    // opaque per tile: hipcc otherwise materialises every (a_row ^ 64 s) + ... of the tile body once,
    // outside the loop, and keeps ~30 address registers alive next to the weights
    asm volatile("" : "+v"(a_row), "+v"(a_tr), "+v"(a_row32), "+v"(a_tr32));
    const int64_t m0 = tile * kX3Rows;
    const bool has_next = tile + stride < tiles, has_next2 = tile + 2 * stride < tiles;
    const char* const i_x = smb + offsetof(X3Smem, x) + buf * (3 * IMG32);
    // the input of tile i+1 (in registers since a tile ago) goes to LDS, the loads of tile i+2 start.
    // Training does this in S5: loads and stores share vmcnt and hipcc waits for ALL of it before the
    // loads' destination registers are written, which here would mean the dx stores of S7
    auto stage_next = [&]() {
      if (has_next) store_x(buf ^ 1);
      if (has_next2) load_x((tile + 2 * stride) * kX3Rows);
    };
    if (!TRAIN) stage_next();
    if (!EARLY_L1) {
      X3P_SYNC(0)  // B0: h1 of this tile complete (and, inference, x of the next staged)
    }
    // (EARLY_L1: h1 of this tile was written before B4 of the previous tile -- or before the barrier in
    //  front of the loop -- so layer 2 starts without a barrier, beside dx of the previous tile)
    if (GATHER && has_next2) g_coords_store(buf);  // coordinates of tile i+2 (read from S5 on: behind B2)
    // ---- S2: h2 = relu(h1 W2^T + b2); this wave's share of y ---------------------------------------
    float h2[U][TL][4];
    {
      Frag hb[2];
      u32x4 wl[U][KS];  // W2's third term, rows of this wave's strips
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int s = 0; s < KS; ++s) wl[u][s] = lds_read_b128(i_w2l + (a_row ^ (64 * s)) + 4096 * strip(u));
      hb[0] = ld_row<IMG>(i_h1, a_row + 4096 * th);
      f32x4 bias[U], w3v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        bias[u] = *reinterpret_cast<const f32x4*>(&sm.b2[unit0(u) + 4 * g]);
        w3v[u] = *reinterpret_cast<const f32x4*>(&sm.w3[unit0(u) + 4 * g]);
      }
#pragma unroll
      for (int tt = 0; tt < TL; ++tt) {
        f32x4 c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) c[u] = zero4;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const int i = KS * tt + s;  // next fragment: step (s + 1) % KS of half (i + 1) / KS
          if (i + 1 < KS * TL)
            hb[(i + 1) & 1] = ld_row<IMG>(i_h1, (a_row ^ (64 * ((s + 1) % KS))) + 4096 * (th + (i + 1) / KS));
          X3_PIN
#pragma unroll
          for (int u = 0; u < U; ++u) c[u] = mma6(Frag{w2f_h[u][s], w2f_m[u][s], wl[u][s]}, hb[i & 1], c[u]);
          X3_PIN
        }
        float yp = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            h2[u][tt][r] = fmaxf(c[u][r] + bias[u][r], 0.f);
            yp += w3v[u][r] * h2[u][tt][r];
          }
        yp = sum_groups(yp);
        if (g == 0) sm.ypart[yslot][16 * (th + tt) + li] = yp;
      }
    }
    X3P_SYNC(2)  // B2
    // ---- S5: prediction, loss, dy (every lane for its own row), dz2 of this wave's units -----------
    {
      float yq[TL][YS], tg[TL];  // all LDS reads first: no read -> use -> read chains
      f32x4 w3v[U];
#pragma unroll
      for (int tt = 0; tt < TL; ++tt) {
#pragma unroll
        for (int q = 0; q < YS; ++q) yq[tt][q] = sm.ypart[q][16 * (th + tt) + li];
        tg[tt] = TRAIN ? sm.tgt[buf][16 * (th + tt) + li] : 0.f;
      }
      if (TRAIN) {
#pragma unroll
        for (int u = 0; u < U; ++u) w3v[u] = *reinterpret_cast<const f32x4*>(&sm.w3[unit0(u) + 4 * g]);
      }
#pragma unroll
      for (int tt = 0; tt < TL; ++tt) {
        const int t = th + tt;
        float y = b3;
#pragma unroll
        for (int q = 0; q < YS; ++q) y += yq[tt][q];
        const bool live = 16 * t + li < a.n - m0;
        if (y_owner && g == 0 && live && a.y) (a.y + m0)[16 * t + li] = y;
        if (TRAIN) {
          const float diff = live ? y - tg[tt] : 0.f;
          const float d = diff * a.grad_scale;
          if (y_owner && g == 0) {
            loss += diff * diff;
            g_b3 += d;
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            float z[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              g_w3[u][r] += d * h2[u][tt][r];
              const float dw = d * w3v[u][r];
              z[r] = h2[u][tt][r] > 0.f ? dw : 0.f;
              g_b2[u][r] += z[r];
            }
            st4(i_z2, a_out[u] + 4096 * t, z);
          }
        }
      }
    }
    if (!TRAIN) {
      if (has_next) layer1(buf ^ 1);  // x(i+1) was staged before B0
      continue;
    }
    if constexpr (GATHER) {
      if (has_next) {
        const float2 pa = *reinterpret_cast<const float2*>(&sm.pa[tid][0]);
        g_finish(buf ^ 1, buf ^ 1, (tile + stride) * kX3Rows, pa.x, pa.y);
      }
      if (has_next2) {
        g_issue(buf, 0);
        g_target((tile + 2 * stride) * kX3Rows);
      }
    } else {
      stage_next();
    }
    // ---- S5b: dW2[units][:] += dz2^T h1 (contracts the 32 rows: both operands transposed reads; the
    //      dz2 columns are this wave's own stores above, and a wave's LDS operations execute in order)
    {
      Frag za[U], hk[2];
      hk[0] = ld_tr<IMG, 4096>(i_h1, a_tr);
#pragma unroll
      for (int u = 0; u < U; ++u) za[u] = own_tr(i_z2, a_tr ^ (32 * strip(u)));
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        if (kt + 1 < KT) hk[(kt + 1) & 1] = ld_tr<IMG, 4096>(i_h1, a_tr ^ (32 * (kt + 1)));
        X3_PIN
#pragma unroll
        for (int u = 0; u < U; ++u) g_w2[u][kt] = mma_dw(za[u], hk[kt & 1], g_w2[u][kt]);
        X3_PIN
      }
    }
    X3P_SYNC(3)  // B3
    // ---- S6: dz1[:, units] = (dz2 W2) (.) (h1 > 0) -------------------------------------------------
    {
      Frag zb[2];
      // W2^T's third term: lane (li, g) = input unit unit0 + li, output units 32 s + 8 g + j: transposed
      // read of rows 32 s + 8 g + q (+ 4) of the image; their swizzles differ by sw(q + 4) = sw(q) ^ 9
      u32x4 wl[U][KS];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int a0 = 2048 * g + 256 * q4 + 8 * (p4 & 1) + 16 * ((p4 >> 1) ^ (sw(q4) & 1)) +
                       ((32 * strip(u)) ^ (16 * (sw(q4) & 14)));
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const u32x2 lo = lds_read_tr(i_w2l + a0 + 8192 * s), hi = lds_read_tr(i_w2l + (a0 ^ 144) + 8192 * s + 1024);
          wl[u][s] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
      }
      zb[0] = ld_row<IMG>(i_z2, a_row + 4096 * th);
#pragma unroll
      for (int tt = 0; tt < TL; ++tt) {
        f32x4 c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) c[u] = zero4;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const int i = KS * tt + s;
          if (i + 1 < KS * TL)
            zb[(i + 1) & 1] = ld_row<IMG>(i_z2, (a_row ^ (64 * ((s + 1) % KS))) + 4096 * (th + (i + 1) / KS));
          X3_PIN
#pragma unroll
          for (int u = 0; u < U; ++u) c[u] = mma6(Frag{w2t_h[u][s], w2t_m[u][s], wl[u][s]}, zb[i & 1], c[u]);
          X3_PIN
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          float z[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            z[r] = ((mask1 >> (8 * u + 4 * tt + r)) & 1u) ? c[u][r] : 0.f;
            g_b1[u][r] += z[r];
          }
          st4(i_z1, a_out[u] + 4096 * (th + tt), z);
        }
      }
    }
    // ---- S6b: dW1[units][:] += dz1^T x (this wave's own dz1 columns, as dW2 above) -------------------
    {
      const Frag xk = ld_tr<IMG32, 1024>(i_x, a_tr32), x1 = ld_tr<IMG32, 1024>(i_x, a_tr32 ^ 32);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const Frag za = own_tr(i_z1, a_tr ^ (32 * strip(u)));
        g_w1[u][0] = mma_dw(za, xk, g_w1[u][0]);
        g_w1[u][1] = mma_dw(za, x1, g_w1[u][1]);
      }
    }
    // ---- S6c: layer 1 of the NEXT tile.  h1 of this tile is dead since B3 (its last readers are layer 2
    //      and dW2), x of the next tile was staged before B3, and this tile's ReLU mask was consumed by
    //      the dz1 epilogue above.  In this segment the 12 MFMAs ride beside dz1 / dW1; behind B4, where
    //      they ran until round 3, they shared a segment with dx alone (24 MFMAs on four of the eight
    //      waves: 0.33 of the matrix pipe) and needed a barrier of their own in front of layer 2.
    if (EARLY_L1 && has_next) layer1(buf ^ 1);
    X3P_SYNC(4)  // B4: z1 of this tile (and, EARLY_L1, h1 of the next) complete
    if constexpr (GATHER) {
      if (has_next2) {
        float pa0, pa1;
        g_take(buf, 0, pa0, pa1);
        *reinterpret_cast<float2*>(&sm.pa[tid][0]) = make_float2(pa0, pa1);
        g_issue(buf, 1);
      }
      if (tile + 3 * stride < tiles) g_coords_load((tile + 3 * stride) * kX3Rows);
    }
    // ---- S7: dx^T = W1^T dz1^T (waves 0..3: 16 features x 16 rows each).  No barrier follows: waves 4..7
    //      go straight to layer 2 of the next tile, waves 0..3 after their 24 MFMAs -------------------------
    if (a.dx && w < 4) {
      const int kt = w & 1, bt = w >> 1;
      f32x4 c = zero4;
      Frag wa[2], zc[2];
      // W1^T fragment: lane (li, g) = feature 16 kt + li, units 32 s + 8 g + j: transposed read of rows
      // 32 s + 8 g + q (+ 4) of the W1 image (64-byte rows, swizzle by (row >> 2) & 3 = 2 g (+ 1))
      const int aw = 512 * g + 64 * q4 + 8 * (p4 & 1);
      const int c_lo = (2 * kt + (p4 >> 1)) ^ ((4 - ((2 * g) & 3)) & 3), c_hi = (2 * kt + (p4 >> 1)) ^ ((4 - ((2 * g + 1) & 3)) & 3);
      auto w1t_frag = [&](int s) {
        Frag f;
#pragma unroll
        for (int term = 0; term < 3; ++term) {
          const char* img = i_w1 + term * (4 * IMG32) + aw + 2048 * s;
          const u32x2 lo = lds_read_tr(img + 16 * c_lo), hi = lds_read_tr(img + 256 + 16 * c_hi);
          (term == 0 ? f.h : term == 1 ? f.m : f.l) = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
        return f;
      };
      wa[0] = w1t_frag(0), zc[0] = ld_row<IMG>(i_z1, a_row + 4096 * bt);
      const uint32_t seen0 = sm.dxmax[8 * kt + 2 * g], seen1 = sm.dxmax[8 * kt + 2 * g + 1];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (s + 1 < KS) {
          wa[(s + 1) & 1] = w1t_frag(s + 1);
          zc[(s + 1) & 1] = ld_row<IMG>(i_z1, (a_row ^ (64 * (s + 1))) + 4096 * bt);
        }
        X3_PIN
        c = mma6(wa[s & 1], zc[s & 1], c);
        X3_PIN
      }
      float* __restrict__ dxs = a.dx + m0;
      const bool live = 16 * bt + li < a.n - m0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 16 * kt + 4 * g + r;
        if (k < a.k_in && live) dxs[(uint32_t)k * ld32 + 16 * bt + li] = c[r];
      }
      if (TRAIN && a.dx_absmax) {
        // max |dx| per feature pair (rows beyond n have dz = 0, hence c = 0), kept in LDS: two more
        // registers across the tile loop cost this kernel 9 %
        const float d0 = fmaxf(fabsf(c[0]), fabsf(c[1])), d1 = fmaxf(fabsf(c[2]), fabsf(c[3]));
        // (the running maxima were read before the MFMAs: after a few tiles no lane raises them any more,
        // and a stale value only costs an atomic that changes nothing)
        if (__float_as_uint(d0) > seen0) atomicMax(&sm.dxmax[8 * kt + 2 * g], __float_as_uint(d0));
        if (__float_as_uint(d1) > seen1) atomicMax(&sm.dxmax[8 * kt + 2 * g + 1], __float_as_uint(d1));
      }
    }
    if (!EARLY_L1 && has_next) layer1(buf ^ 1);
  }
#undef X3_PIN
  if (TRAIN && a.dx && a.dx_absmax) {
    __syncthreads();
    if (tid < 16 && 2 * tid < a.k_in) atomicMax(a.dx_absmax + tid, sm.dxmax[tid]);
  }
  X3P_MARK(18)
  if (!TRAIN) {
    X3P_END
    return;
  }

  // per-unit sums: the 16 lanes li of a group hold the 16 (+16) rows' shares, in a fixed order
  auto rows_sum = [](float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
  };
  float s_b1[U][4], s_b2[U][4], s_w3[U][4];
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s_b1[u][r] = rows_sum(g_b1[u][r]), s_b2[u][r] = rows_sum(g_b2[u][r]), s_w3[u][r] = rows_sum(g_w3[u][r]);
    }
  float s_b3 = rows_sum(g_b3), s_loss = rows_sum(loss);  // only the y owners' lanes g == 0 hold shares
  if (H == 64) {
    // the two waves of a strip (row halves 0 and 1) each hold a partial sum of everything: wave w + 4
    // hands its values to wave w through LDS (the images are dead), which adds them in a fixed order
    constexpr int PER = 4 * KT + 8 + 12 + 2;  // floats per lane
    float* x = reinterpret_cast<float*>(smb);
    __syncthreads();
    float* mine = x + ((w & 3) * 64 + lane) * PER;
    if (w >= 4) {
      int o = 0;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[o++] = g_w2[0][kt][r];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[o++] = g_w1[0][kt][r];
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[o++] = s_b1[0][r], mine[o++] = s_b2[0][r], mine[o++] = s_w3[0][r];
      mine[o++] = s_b3, mine[o++] = s_loss;
    }
    __syncthreads();
    if (w >= 4) {
      X3P_MARK(21)
      X3P_END
      return;
    }
    int o = 0;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) g_w2[0][kt][r] += mine[o++];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) g_w1[0][kt][r] += mine[o++];
#pragma unroll
    for (int r = 0; r < 4; ++r) s_b1[0][r] += mine[o++], s_b2[0][r] += mine[o++], s_w3[0][r] += mine[o++];
    s_b3 += mine[o++], s_loss += mine[o++];
  }

  // ---- this workgroup's slab: every element of dW2 / dW1 has exactly one owner lane ----------------
  float* slab = a.partial + (int64_t)blockIdx.x * slab_floats(H, a.k_in);
  float* p_w1 = slab;
  float* p_b1 = p_w1 + H * a.k_in;
  float* p_w2 = p_b1 + H;
  float* p_b2 = p_w2 + H * H;
  float* p_w3 = p_b2 + H;
  float* p_b3 = p_w3 + H;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int n = unit0(u) + 4 * g;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) p_w2[(n + r) * H + 16 * kt + li] = g_w2[u][kt][r];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * kt + li < a.k_in) p_w1[(n + r) * a.k_in + 16 * kt + li] = g_w1[u][kt][r];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (li == 0) p_b1[n + r] = s_b1[u][r], p_b2[n + r] = s_b2[u][r], p_w3[n + r] = s_w3[u][r];
  }
  if (w == 0 && lane == 0) {
    p_b3[0] = s_b3;
    p_b3[1] = s_loss * a.inv_n;
  }
  X3P_MARK(21)
  X3P_END
}

// ---------------------------------------------------------------------------------------------------
// Inference form for H = 128 (dense-grid predict / interpolate passes, launcher.py:150-185): the same
// arithmetic per row as the training kernel's forward half (bit-identical predictions), on 64-row
// tiles with two barriers per tile -- no gradient images, so the h1 image can be twice as tall, and
// all three terms of the wave's W2 rows fit in registers.
struct __attribute__((aligned(16))) X3InferSmem {
  char x[2][3][64 * 64];     // input tile [64 rows][32 features], three terms, double-buffered
  char h1[3][64 * 256];
  char w1[3][4 * kImg32Bytes];
  float ypart[8][64];
  float b1[kX3H], b2[kX3H], w3[kX3H];
};

__global__ __launch_bounds__(kX3Threads) void tiny_mlp_x3_infer_kernel(const FusedArgs a) {
  constexpr int H = kX3H, ROWS = 64, NSUB = 4, XT = 64 * 64, HT = 64 * 256;
  __shared__ X3InferSmem sm;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g = lane >> 4;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  Frag w2f[4];
  {
    const int n = 16 * w + li;
    float v[8];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = a.w2[n * H + 32 * s + 8 * g + j];
      w2f[s] = split8(v);
    }
  }
  for (int e = tid; e < H * 16; e += kX3Threads) {
    const int n = e >> 4, kp = e & 15;
    const float v0 = 2 * kp < a.k_in ? a.w1[n * a.k_in + 2 * kp] : 0.f;
    const float v1 = 2 * kp + 1 < a.k_in ? a.w1[n * a.k_in + 2 * kp + 1] : 0.f;
    uint32_t h, m, l;
    split2(v0, v1, h, m, l);
    const int off = img32_off(n, kp >> 2) + 4 * (kp & 3);
    *reinterpret_cast<uint32_t*>(sm.w1[0] + off) = h;
    *reinterpret_cast<uint32_t*>(sm.w1[1] + off) = m;
    *reinterpret_cast<uint32_t*>(sm.w1[2] + off) = l;
  }
  if (tid < H) sm.b1[tid] = a.b1[tid], sm.b2[tid] = a.b2[tid], sm.w3[tid] = a.w3[tid];
  const float b3 = a.b3[0];

  const int swl = sw(li), q4 = li >> 2;
  int a_row = 256 * li + 16 * (g ^ (swl & 3)) + 64 * (swl >> 2);  // (a_row ^ 64 s) + 4096 t, see the training kernel
  int a_row32 = 64 * li + 16 * (g ^ ((4 - q4) & 3));               // + 1024 t
  const int a_out = 256 * li + 8 * (g & 1) + 16 * ((g >> 1) ^ (swl & 1)) + ((32 * w) ^ (16 * (swl & 14)));
  char* const smb = reinterpret_cast<char*>(&sm);
  char* const i_h1 = smb + offsetof(X3InferSmem, h1);
  const char* const i_w1 = smb + offsetof(X3InferSmem, w1);
  const uint32_t ld32 = (uint32_t)a.ld;
  float xv[2][2];
  auto load_x = [&](int64_t m0) {  // thread = (row sb of 64, feature pairs skp and skp + 8)
    int t_op = tid;
    asm volatile("" : "+v"(t_op));
    const int sb = t_op & 63, skp = t_op >> 6;
    const float* __restrict__ xs = a.x + m0;
    const bool live = sb < a.n - m0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int k = 2 * (skp + 8 * q);
      xv[q][0] = (live && k < a.k_in) ? xs[(uint32_t)k * ld32 + sb] : 0.f;
      xv[q][1] = (live && k + 1 < a.k_in) ? xs[(uint32_t)(k + 1) * ld32 + sb] : 0.f;
    }
  };
  auto store_x = [&](int buf) {
    int t_op = tid;
    asm volatile("" : "+v"(t_op));
    const int sb = t_op & 63, skp = t_op >> 6;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      uint32_t h, m, l;
      split2(xv[q][0], xv[q][1], h, m, l);
      const int kp = skp + 8 * q;
      const int off = img32_off(sb, kp >> 2) + 4 * (kp & 3);
      *reinterpret_cast<uint32_t*>(sm.x[buf][0] + off) = h;
      *reinterpret_cast<uint32_t*>(sm.x[buf][1] + off) = m;
      *reinterpret_cast<uint32_t*>(sm.x[buf][2] + off) = l;
    }
  };
  auto layer1 = [&](int xbuf) {
    const char* const i_x = smb + offsetof(X3InferSmem, x) + xbuf * (3 * XT);
    const Frag w1f = ld_row<4 * kImg32Bytes>(i_w1, a_row32 + 1024 * w);
    const f32x4 bias = *reinterpret_cast<const f32x4*>(&sm.b1[16 * w + 4 * g]);
    Frag xb[2];
    xb[0] = ld_row<XT>(i_x, a_row32);
#pragma unroll
    for (int t = 0; t < NSUB; ++t) {
      if (t + 1 < NSUB) xb[(t + 1) & 1] = ld_row<XT>(i_x, a_row32 + 1024 * (t + 1));
      __builtin_amdgcn_sched_barrier(0);
      const f32x4 c = mma6(w1f, xb[t & 1], zero4);
      __builtin_amdgcn_sched_barrier(0);
      float h[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) h[r] = fmaxf(c[r] + bias[r], 0.f);
      uint32_t h0, m0, l0, h1, m1, l1;
      split2(h[0], h[1], h0, m0, l0);
      split2(h[2], h[3], h1, m1, l1);
      const int off = a_out + 4096 * t;
      *reinterpret_cast<u32x2*>(i_h1 + off) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(i_h1 + HT + off) = u32x2{m0, m1};
      *reinterpret_cast<u32x2*>(i_h1 + 2 * HT + off) = u32x2{l0, l1};
    }
  };

  const int64_t tiles = (a.n + ROWS - 1) / ROWS, stride = gridDim.x;
  int buf = 0;
  load_x((int64_t)blockIdx.x * ROWS);
  store_x(0);
  if (blockIdx.x + stride < tiles) load_x((blockIdx.x + stride) * ROWS);
  __syncthreads();
  layer1(0);
  for (int64_t tile = blockIdx.x; tile < tiles; tile += stride, buf ^= 1) {
    asm volatile("" : "+v"(a_row), "+v"(a_row32));
    const int64_t m0 = tile * ROWS;
    const bool has_next = tile + stride < tiles, has_next2 = tile + 2 * stride < tiles;
    __syncthreads();  // h1 of this tile complete
    // ---- layer 2 and this strip's share of y; the next tile's input goes to LDS meanwhile ---------
    {
      const f32x4 bias = *reinterpret_cast<const f32x4*>(&sm.b2[16 * w + 4 * g]);
      const f32x4 w3v = *reinterpret_cast<const f32x4*>(&sm.w3[16 * w + 4 * g]);
      Frag hb[2];
      hb[0] = ld_row<HT>(i_h1, a_row);
#pragma unroll
      for (int t = 0; t < NSUB; ++t) {
        f32x4 c = zero4;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int i = 4 * t + s;
          if (i + 1 < 4 * NSUB)
            hb[(i + 1) & 1] = ld_row<HT>(i_h1, (a_row ^ (64 * ((s + 1) & 3))) + 4096 * ((i + 1) >> 2));
          __builtin_amdgcn_sched_barrier(0);
          c = mma6(w2f[s], hb[i & 1], c);
          __builtin_amdgcn_sched_barrier(0);
        }
        float yp = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) yp += w3v[r] * fmaxf(c[r] + bias[r], 0.f);
        yp = sum_groups(yp);
        if (g == 0) sm.ypart[w][16 * t + li] = yp;
      }
    }
    if (has_next) store_x(buf ^ 1);
    if (has_next2) load_x((tile + 2 * stride) * ROWS);
    __syncthreads();  // y shares complete, x of the next tile staged, every wave done with h1
    if (w == 0) {
      float y = b3;
#pragma unroll
      for (int q = 0; q < 8; ++q) y += sm.ypart[q][lane];
      if (lane < a.n - m0) (a.y + m0)[lane] = y;
    }
    if (has_next) layer1(buf ^ 1);
  }
}

}  // namespace

bool x3_supported(int k_in, int hidden) { return (hidden == 128 || hidden == 64) && k_in >= 1 && k_in <= 32; }

bool x3_addressable(const FusedArgs& a) { return (int64_t)a.k_in * a.ld < (1ll << 30); }

int x3_blocks(int64_t n) { return (int)std::min<int64_t>(ceil_div(n, kX3Rows), 256); }

bool x3_infer_wide(int hidden) { return hidden == 128 && options().mlp_x3 == 1; }

int launch_tiny_mlp_x3(const FusedArgs& a, int hidden, bool train, int blocks, hipStream_t st) {
  const bool wide = options().mlp_x3 == 2 && hidden == 128;  // 4 waves of two strips (tools: A/B against 8 x 1)
  if (!train && x3_infer_wide(hidden)) {
    const int ib = (int)std::min<int64_t>(ceil_div(a.n, 64), 256);
    hipLaunchKernelGGL(tiny_mlp_x3_infer_kernel, dim3(ib), dim3(kX3Threads), 0, st, a);
    return check_launch("tiny_mlp_x3_infer_kernel");
  }
  const EncodeArgs none{};
  if (hidden == 64 && train)
    hipLaunchKernelGGL((tiny_mlp_x3_kernel<true, 1, 64>), dim3(blocks), dim3(kX3Threads), 0, st, a, none);
  else if (hidden == 64)
    hipLaunchKernelGGL((tiny_mlp_x3_kernel<false, 1, 64>), dim3(blocks), dim3(kX3Threads), 0, st, a, none);
  else if (train && wide)
    hipLaunchKernelGGL((tiny_mlp_x3_kernel<true, 2, 128>), dim3(blocks), dim3(kX3Threads / 2), 0, st, a, none);
  else if (train)
    hipLaunchKernelGGL((tiny_mlp_x3_kernel<true, 1, 128>), dim3(blocks), dim3(kX3Threads), 0, st, a, none);
  else if (wide)
    hipLaunchKernelGGL((tiny_mlp_x3_kernel<false, 2, 128>), dim3(blocks), dim3(kX3Threads / 2), 0, st, a, none);
  else
    hipLaunchKernelGGL((tiny_mlp_x3_kernel<false, 1, 128>), dim3(blocks), dim3(kX3Threads), 0, st, a, none);
  return check_launch("tiny_mlp_x3_kernel");
}

bool x3_encode_supported(int dim, int n_features, int n_levels, int hidden) {
  return n_features == 2 && dim >= 2 && dim <= 4 && n_levels >= 1 && 2 * n_levels <= 32 &&
         (hidden == 128 || hidden == 64);
}

int launch_tiny_mlp_x3_encoded(const FusedArgs& a, const EncodeArgs& e, int hidden, int blocks, hipStream_t st) {
#define MRI_X3_ENC(HH, DD) \
  hipLaunchKernelGGL((tiny_mlp_x3_kernel<true, 1, HH, DD>), dim3(blocks), dim3(kX3Threads), 0, st, a, e)
  if (hidden == 128) {
    if (e.dim == 2) MRI_X3_ENC(128, 2);
    else if (e.dim == 3) MRI_X3_ENC(128, 3);
    else MRI_X3_ENC(128, 4);
  } else {
    if (e.dim == 2) MRI_X3_ENC(64, 2);
    else if (e.dim == 3) MRI_X3_ENC(64, 3);
    else MRI_X3_ENC(64, 4);
  }
#undef MRI_X3_ENC
  return check_launch("tiny_mlp_x3_kernel (encoded)");
}

}  // namespace mri

#ifdef MRI_X3_PROFILE
extern "C" int mri_debug_set_x3_profile(long long* device_buffer) {
  return hipMemcpyToSymbol(HIP_SYMBOL(mri::g_x3_profile), &device_buffer, sizeof(device_buffer)) ==
                 hipSuccess
             ? 0
             : -1;
}
#endif
