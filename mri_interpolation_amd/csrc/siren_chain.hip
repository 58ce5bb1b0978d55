// Fused sine-MLP (SIREN) chain kernels for gfx950: a row tile's activations stay ON CHIP across
// all layers.
//
// Replaces, for SirenNet(dim_in <= 8 -> 256 x n -> 1) (reference models.py:153-156 SirenLayer.forward,
// :230-233 SirenNet.forward: `F.linear` + `sin(w0 .)` per layer, then the linear head), the
// layer-by-layer GEMM launches of linear.hip, which re-read every (n, 256) activation from HBM
// as the next layer's operand and pay a prologue / epilogue bubble per 128 x 128 tile:
//
//   siren_forward_kernel   one persistent 512-thread workgroup per CU walks 64-row tiles; per tile
//     the first layer (K = dim_in) runs on the VALU straight into an LDS activation image, every
//     256 x 256 layer multiplies that image (A operand, ds_read_b128) with the layer's weights
//     streamed from L2 in 32-deep chunks by LDS-DMA (global_load_lds_dwordx4, double buffered,
//     XOR-swizzled on the source side so that the B-operand ds_read_b128 are conflict-free) on
//     v_mfma_f32_32x32x2_f32 (exact f32: the 1e-5 parity target rules out bf16), applies
//     bias / w0 / sincos in registers and overwrites the image; the 1-wide head is a wave
//     reduction over the image.  For training the activation a_l = sin(.) and its derivative
//     w0 cos(.) leave for HBM once (the backward kernels need them); they are dripped out of
//     registers beside the NEXT layer's MFMAs, a chunk ahead of the next wait.
//
// Data layout: activations / derivatives row-major (n, 256) per layer, as linear.hip writes them,
// so the layer-wise kernels and the chain kernels are interchangeable per layer.
#include <algorithm>

#include "common.h"
#include "device_math.h"

namespace mri {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kH = 256;        // hidden width of the chain kernels
constexpr int kRows = 64;      // rows of a tile
constexpr int kThreads = 512;  // 8 waves: 2 row blocks x 4 column blocks of 32 x 64
constexpr int kLd = kH + 4;    // image row stride: rows 4 banks apart -> conflict-free ds_read_b128
constexpr int kKc = 32;        // contraction depth of a weight chunk (32 KiB of LDS)
constexpr int kChunks = kH / kKc;
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
};

// Geometry of a chain workgroup: ROWS-row tiles, ROWS / 32 row blocks x 4 column blocks of
// 32 x 64 per wave (ROWS * 8 threads), weight chunks KC deep.  Two shapes are built:
//   <64, 32>  one 512-thread workgroup per CU (151 KiB of LDS)
//   <32, 16>  two 256-thread workgroups per CU (75 KiB each): the two waves of a SIMD then belong
//             to different workgroups, walk different tiles and do not meet at each other's
//             barriers, so one's waits (DMA landing, LDS latency, barriers: a third of a wave's
//             cycles) fall beside the other's MFMAs instead of beside its waits; the price is
//             twice the weight traffic from L2 (each 32-row tile streams all of W).
template <int ROWS, int KC>
struct Geo {
  static constexpr int rows = ROWS, kc = KC;
  static constexpr int waves = ROWS / 32 * 4, threads = waves * 64;
  static constexpr int chunks = kH / KC;             // per layer
  static constexpr int slots = KC / 4;               // 16-byte slots per chunk row
  static constexpr int piece_rows = 64 / slots;      // rows of W per 1-KiB DMA instruction (64 lanes x 16 B)
  static constexpr int pieces_per_wave = KC / waves; // kH * KC * 4 B / 1 KiB / waves
  static constexpr int swz_shift = slots == 8 ? 1 : 2;  // rows a bank row of 16 slots spans
  static constexpr int octets = KC / 8;
  static constexpr int drip_regs = 16 / chunks > 0 ? 16 / chunks : 1;  // per tile and chunk
  static_assert(KC % waves == 0 && (slots == 8 || slots == 4), "unsupported chain geometry");
  static_assert(pieces_per_wave * waves * piece_rows == kH, "the DMA pieces must tile the chunk");
};

template <class G>
struct Smem {
  float wbuf[2][kH * G::kc];        // weight chunks [n][KC] with 16-byte slots XOR-swizzled
  float img[G::rows * kLd];         // activation image of the tile
  float xs[G::rows * kMaxIn];
  float bias[kMaxSine][kH];
  float w_last[kH];
};

// row of register r of a 32x32 accumulator: (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
__device__ __forceinline__ int acc_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// Queue the LDS-DMA of chunk (weights w of one layer, columns [KC kc, KC kc + KC)) into `dst`:
// pieces of 1 KiB (piece_rows rows x KC floats), lane = (row in piece, 16-byte slot); the slot a
// lane FETCHES is its LDS slot XOR f(row), the involution the fragment reads undo, with
// f(n) = (n >> swz_shift) & (slots - 1): the 16 rows a ds_read_b128 lane group touches then fall
// on 16 different 16-byte slots of the 256-byte bank row.
template <class G>
__device__ __forceinline__ void issue_chunk(const float* __restrict__ w, int kc, float* dst,
                                            int wave, int lane) {
#pragma unroll
  for (int i = 0; i < G::pieces_per_wave; ++i) {
    const int piece = wave * G::pieces_per_wave + i;
    const int n = piece * G::piece_rows + lane / G::slots;
    const int q = (lane % G::slots) ^ ((n >> G::swz_shift) & (G::slots - 1));
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(w + n * kH + kc * G::kc + q * 4),
        (__attribute__((address_space(3))) void*)(dst + piece * G::piece_rows * G::kc), 16, 0, 0);
  }
}

// One KC-deep chunk of acc[t] += img[rows][k] * W[cols_t][k] for the wave's 32 x 64 tile.
// Lane half lh takes k = 8 j + 4 lh + e of octet j (element e of its 16-byte fragment): the two
// halves of an MFMA's 2-deep contraction are k and k + 4.
template <class G>
__device__ __forceinline__ void mma_chunk(f32x16 (&acc)[2], const float* __restrict__ a_row,
                                          const float* __restrict__ wb, int nb0, int nb1,
                                          int sw0, int sw1, int lh) {
  float4 av[2], bv[2][2];
  auto fetch = [&](int buf, int j) {
    av[buf] = *reinterpret_cast<const float4*>(a_row + 8 * j);
    bv[buf][0] = *reinterpret_cast<const float4*>(wb + nb0 + (((2 * j + lh) ^ sw0) << 2));
    bv[buf][1] = *reinterpret_cast<const float4*>(wb + nb1 + (((2 * j + lh) ^ sw1) << 2));
  };
  auto compute = [&](int buf) {
    const float ae[4] = {av[buf].x, av[buf].y, av[buf].z, av[buf].w};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float be[4] = {bv[buf][t].x, bv[buf][t].y, bv[buf][t].z, bv[buf][t].w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae[e], be[e], acc[t], 0, 0, 0);
    }
  };
  fetch(0, 0);
#pragma unroll
  for (int j = 0; j < G::octets; ++j) {
    if (j + 1 < G::octets) fetch((j + 1) & 1, j + 1);
    __builtin_amdgcn_sched_barrier(0);
    compute(j & 1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Phase timing for tools/siren_phases.py (a tools-only build with -DSIREN_PROFILE; the shipped
// library compiles these to nothing): shader-clock cycles per phase of the forward kernel, per wave.
#ifdef SIREN_PROFILE
__device__ long long* g_siren_profile = nullptr;
#define SP_BEGIN long long sp_t = clock64(); long long sp_acc[8] = {};
#define SP_MARK(i) { const long long sp_n = clock64(); sp_acc[i] += sp_n - sp_t; sp_t = sp_n; }
#define SP_END                                                                         \
  if (g_siren_profile && (threadIdx.x & 63) == 0) {                                    \
    long long* dst = g_siren_profile + ((int64_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8; \
    for (int q = 0; q < 8; ++q) dst[q] = sp_acc[q];                                    \
  }
#else
#define SP_BEGIN
#define SP_MARK(i)
#define SP_END
#endif

template <bool STORE, class G>
__global__ __launch_bounds__(G::threads, 2) void siren_forward_kernel(const ChainArgs a) {
  __shared__ Smem<G> sm;
  constexpr int kThreadsG = G::threads;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int rb = wave >> 2, cb = wave & 3;
  const int n_mm = a.n_sine - 1;  // 256 x 256 layers

  // ---- small resident parameters ------------------------------------------------------------
  for (int l = 0; l < a.n_sine; ++l)
    for (int e = tid; e < kH; e += kThreadsG) sm.bias[l][e] = a.b[l][e];
  for (int e = tid; e < kH; e += kThreadsG) sm.w_last[e] = a.w[a.n_sine][e];
  const float b_last = a.b[a.n_sine][0];

  const int64_t tiles = (a.n + G::rows - 1) / G::rows;
  // this lane's fragment addresses
  const float* a_row = sm.img + (rb * 32 + l31) * kLd + 4 * lh;
  const int n0 = cb * 64 + l31, n1 = n0 + 32;
  const int nb0 = n0 * G::kc, nb1 = n1 * G::kc;
  const int sw0 = (n0 >> G::swz_shift) & (G::slots - 1), sw1 = (n1 >> G::swz_shift) & (G::slots - 1);

  // chunk stream: chunk s (layer 1 + (s / chunks) % n_mm, columns KC (s % chunks)) lives in wbuf[s & 1]
  int s = 0;
  if (n_mm > 0 && (int64_t)blockIdx.x < tiles) issue_chunk<G>(a.w[1], 0, sm.wbuf[0], wave, lane);

  // activations waiting to leave for HBM (STORE): the outputs of MFMA layer `pend_l` of the tile
  // at `pend_m0`, dripped out a few registers per chunk beside the next layer's MFMAs
  float pa[2][16], pd[2][16];
  int pend_l = -1;
  int64_t pend_m0 = 0;
  // lane's element offset inside a tile's (ROWS, H) block for accumulator register 0 of tile 0
  const int lane_off = (rb * 32 + 4 * lh) * kH + cb * 64 + l31;
  auto drip = [&](int r_lo, int r_hi, bool tile_full) {
    if (!STORE || pend_l < 0) return;
    // uniform bases + a 32-bit lane offset; the offset is re-derived from an opaque copy per call,
    // or hipcc hoists all 32 store offsets out of the layer loop and spills
    float* __restrict__ ga = a.act[pend_l] + pend_m0 * kH;
    float* __restrict__ gd = a.deriv[pend_l] + pend_m0 * kH;
    int off = lane_off;
    asm volatile("" : "+v"(off));
    const int64_t rows_left = a.n - pend_m0 - rb * 32 - 4 * lh;  // rows below this one are live
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (r < r_lo || r >= r_hi) continue;
        const int dr = (r & 3) + 8 * (r >> 2);
        if (tile_full || dr < rows_left) {  // tile_full is wave-uniform: no per-store branches
          ga[off + dr * kH + t * 32] = pa[t][r];
          gd[off + dr * kH + t * 32] = pd[t][r];
        }
      }
  };

  SP_BEGIN
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t m0 = tile * G::rows;
    const bool full_tile = m0 + G::rows <= a.n;  // wave-uniform: stores need no per-lane row check
    // ---- x tile -> LDS -------------------------------------------------------------------------
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // previous tile is done with img / xs
    if (tid < G::rows * kMaxIn) {
      const int row = tid / kMaxIn, d = tid % kMaxIn;
      sm.xs[tid] = (d < a.dim_in && m0 + row < a.n) ? a.x[(m0 + row) * a.dim_in + d] : 0.f;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    SP_MARK(0)  // tile top: x tile
    // ---- first layer on the VALU: thread <-> (column, 32 of the rows) --------------------------
    {
      const int col = tid & (kH - 1), r0 = (tid >> 8) * 32;
      float wr[kMaxIn];
#pragma unroll
      for (int d = 0; d < kMaxIn; ++d) wr[d] = d < a.dim_in ? a.w[0][col * a.dim_in + d] : 0.f;
      const float bias = sm.bias[0][col];
      float* __restrict__ ga = STORE ? a.act[0] : nullptr;
      float* __restrict__ gd = STORE ? a.deriv[0] : nullptr;
#pragma unroll 8  // 8 independent sincos chains in flight: the phase is latency bound at 4
      for (int r = 0; r < 32; r += 2) {
        float z0 = 0.f, z1 = 0.f;
#pragma unroll
        for (int d = 0; d < kMaxIn; ++d) {
          if (d < a.dim_in) {
            z0 += sm.xs[(r0 + r) * kMaxIn + d] * wr[d];
            z1 += sm.xs[(r0 + r + 1) * kMaxIn + d] * wr[d];
          }
        }
        float s0, c0, s1, c1;
        sincos_fast2(a.w0_first * (z0 + bias), a.w0_first * (z1 + bias), &s0, &c0, &s1, &c1);
        sm.img[(r0 + r) * kLd + col] = s0;
        sm.img[(r0 + r + 1) * kLd + col] = s1;
        if (STORE) {
          const int64_t row = m0 + r0 + r;
          if (row < a.n) ga[row * kH + col] = s0, gd[row * kH + col] = a.w0_first * c0;
          if (row + 1 < a.n) ga[(row + 1) * kH + col] = s1, gd[(row + 1) * kH + col] = a.w0_first * c1;
        }
      }
    }
    SP_MARK(1)  // first layer
    // ---- 256 x 256 layers ----------------------------------------------------------------------
    for (int l = 1; l <= n_mm; ++l) {
      f32x16 acc[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] = 0.f, acc[1][r] = 0.f;
#pragma unroll
      for (int kc = 0; kc < G::chunks; ++kc, ++s) {
        // Chunk s has landed (this wave's own pieces; vmcnt counts loads, stores and LDS-DMA
        // together).  The stores dripped at the start of the previous chunk have had a whole
        // chunk to retire, so waiting for everything costs nothing: counted waits that left them
        // in flight measured 6.27 against 6.29 ms per 2^20-row pass.
#ifndef SIREN_EXPERIMENT_NO_DMA_WAIT  // timing experiment only (results are then wrong)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // ... for every wave, and wbuf[(s + 1) & 1] is free
        SP_MARK(2)  // chunk wait + barrier
        {
          const bool more_k = kc + 1 < G::chunks;
          const int nl = more_k ? l : (l < n_mm ? l + 1 : 1);
          if (more_k || l < n_mm || tile + gridDim.x < tiles)
            issue_chunk<G>(a.w[nl], more_k ? kc + 1 : 0, sm.wbuf[(s + 1) & 1], wave, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (G::chunks <= 16) drip(kc * 16 / G::chunks, (kc + 1) * 16 / G::chunks, full_tile);
        SP_MARK(3)  // DMA issue + dripped stores
        mma_chunk<G>(acc, a_row + kc * G::kc, sm.wbuf[s & 1], nb0, nb1, sw0, sw1, lh);
        SP_MARK(4)  // fragment reads + MFMAs
      }
      pend_l = -1;  // fully dripped
      // ---- epilogue: bias, w0, sincos; the image becomes this layer's output -------------------
      const float w0 = a.w0;
      const float bias0 = sm.bias[l][n0], bias1 = sm.bias[l][n1];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float bj = t ? bias1 : bias0;
          float s0, c0, s1, c1;
          sincos_fast2(w0 * (acc[t][r] + bj), w0 * (acc[t][r + 1] + bj), &s0, &c0, &s1, &c1);
          pa[t][r] = s0, pa[t][r + 1] = s1;
          pd[t][r] = w0 * c0, pd[t][r + 1] = w0 * c1;
        }
      SP_MARK(5)  // epilogue arithmetic
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // every wave has read the image for the last time
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          sm.img[(rb * 32 + acc_row(r, lh)) * kLd + cb * 64 + t * 32 + l31] = pa[t][r];
      if (STORE) {
        pend_l = l, pend_m0 = m0;
        if (l == n_mm) {  // no next MFMA layer in this tile: leave now
          drip(0, 16, full_tile);
          pend_l = -1;
        }
      }
      SP_MARK(6)  // epilogue barrier + image write
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // image complete
    // ---- head: y[row] = img[row] . w_last + b_last, one wave per row, 16 bytes per lane ------
    {
      const float4 wv = *reinterpret_cast<const float4*>(sm.w_last + 4 * lane);
#pragma unroll
      for (int i = 0; i < G::rows / G::waves; ++i) {
        const int row = wave * (G::rows / G::waves) + i;
        const float4 xv = *reinterpret_cast<const float4*>(sm.img + row * kLd + 4 * lane);
        float acc1 = xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc1 += __shfl_down(acc1, off, 64);
        if (lane == 0 && m0 + row < a.n) a.y[m0 + row] = acc1 + b_last;
      }
    }
    SP_MARK(7)  // head
  }
  SP_END
}

// ------------------------------------------------------------------------------------------
// Backward chain: dz of a 64-row tile walks the layers from the head down to the first layer
// inside LDS.  Per tile:
//   head     dz_L = dy w_head^T (.) d_L  on the VALU into the image; dW_head, db_head partials
//   layer l  da_{l-1} = dz_l W_l on the MFMAs: the image (dz_l) is the A operand, W_l streams
//            from L2 UNTRANSPOSED in 32-row chunks (row n of W_l = contraction index n, its 256
//            columns = output columns: B fragments are conflict-free ds_read_b32, no swizzle);
//            dz_{l-1} = da_{l-1} (.) d_{l-1} in the epilogue (d prefetched a layer ahead of its
//            use), bias-gradient column sums in registers, the image becomes dz_{l-1}
//   first    dW_first = dz_0^T x on the VALU (K = dim_in)
// dz_l (l >= 1) leaves for HBM once: siren_wgrad_kernel contracts it with a_{l-1} over the batch.
// Partial sums (biases, head, first layer) leave through one slab per workgroup, summed in a
// fixed order by slab_sum_kernel (bitwise reproducible, no float atomics).
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
};

// slab: dW_head [H] | db_head [1] (padded to 4) | db_l [n_sine][H] | dW_first [H][kMaxIn]
__host__ __device__ inline int bwd_slab_floats(int n_sine) { return kH + 4 + n_sine * kH + kH * kMaxIn; }

struct BwdSmem {
  float wbuf[2][kKc * kH];          // weight chunks [32 rows n][256]
  float img[kRows * kLd];
  float xs[kRows * kMaxIn];
  float w_last[kH];
  float dy[kRows];
  float gb[kMaxSine][2][kH];        // bias-gradient column sums per layer and row block (sole owners)
};

__device__ __forceinline__ void issue_rows(const float* __restrict__ w, int kc, float* dst,
                                           int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 4 + i;  // one 1-KiB row per wave instruction
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(w + (kc * kKc + row) * kH + lane * 4),
        (__attribute__((address_space(3))) void*)(dst + row * kH), 16, 0, 0);
  }
}

// acc[t] += img[rows][n] * W[n][cols_t] over the chunk's 32 contraction indices n
__device__ __forceinline__ void mma_chunk_rows(f32x16 (&acc)[2], const float* __restrict__ a_row,
                                               const float* __restrict__ wb_lane) {
  float4 av[2];
  float bv[2][2][4];
  auto fetch = [&](int buf, int j) {
    av[buf] = *reinterpret_cast<const float4*>(a_row + 8 * j);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bv[buf][0][e] = wb_lane[(8 * j + e) * kH];
      bv[buf][1][e] = wb_lane[(8 * j + e) * kH + 32];
    }
  };
  auto compute = [&](int buf) {
    const float ae[4] = {av[buf].x, av[buf].y, av[buf].z, av[buf].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae[e], bv[buf][0][e], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae[e], bv[buf][1][e], acc[1], 0, 0, 0);
    }
  };
  fetch(0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j + 1 < 4) fetch((j + 1) & 1, j + 1);
    __builtin_amdgcn_sched_barrier(0);
    compute(j & 1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

__global__ __launch_bounds__(kThreads) void siren_backward_kernel(const BwdArgs a) {
  __shared__ BwdSmem sm;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int rb = wave >> 2, cb = wave & 3;
  const int L = a.n_sine;  // sine layers 0 .. L-1, head = layer L
  for (int e = tid; e < kH; e += kThreads) sm.w_last[e] = a.w[L][e];

  const int64_t tiles = (a.n + kRows - 1) / kRows;
  const float* a_row = sm.img + (rb * 32 + l31) * kLd + 4 * lh;
  const int n0 = cb * 64 + l31;
  const int lane_off = (rb * 32 + 4 * lh) * kH + n0;  // element (row of register 0, column n0)
  const int col = tid & (kH - 1), r0 = (tid >> 8) * 32;   // VALU phases: (column, half of the rows)

  // running sums of this workgroup
  float g_whead = 0.f, g_bhead = 0.f, g_blast = 0.f;  // (col, half) mapping; g_bhead: tid < 64
  float g_wfirst[kMaxIn];                             // (col, half) mapping
  for (int e = tid; e < kMaxSine * 2 * kH; e += kThreads) (&sm.gb[0][0][0])[e] = 0.f;
#pragma unroll
  for (int d = 0; d < kMaxIn; ++d) g_wfirst[d] = 0.f;

  int s = 0;
  if (L > 1 && (int64_t)blockIdx.x < tiles) issue_rows(a.w[L - 1], 0, sm.wbuf[0], wave, lane);

  float pz[2][16];  // dz waiting to leave for HBM, dripped beside the next layer's MFMAs
  int pend_l = -1;
  int64_t pend_m0 = 0;
  auto drip = [&](int r_lo, int r_hi, bool tile_full) {
    if (pend_l < 0) return;
    float* __restrict__ gz = a.dz[pend_l] + pend_m0 * kH;
    int off = lane_off;
    asm volatile("" : "+v"(off));
    const int64_t rows_left = a.n - pend_m0 - rb * 32 - 4 * lh;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (r < r_lo || r >= r_hi) continue;
        const int dr = (r & 3) + 8 * (r >> 2);
        if (tile_full || dr < rows_left) gz[off + dr * kH + t * 32] = pz[t][r];
      }
  };
  // derivative of a layer's activation in the accumulator layout (zeros beyond n)
  auto load_deriv = [&](const float* __restrict__ d, int64_t m0, bool tile_full, float (&dv)[2][16]) {
    const float* __restrict__ g = d + m0 * kH;
    int off = lane_off;
    asm volatile("" : "+v"(off));
    const int64_t rows_left = a.n - m0 - rb * 32 - 4 * lh;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        dv[t][r] = (tile_full || dr < rows_left) ? g[off + dr * kH + t * 32] : 0.f;
      }
  };

  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t m0 = tile * kRows;
    const bool full_tile = m0 + kRows <= a.n;
    __builtin_amdgcn_s_barrier();  // previous tile is done with img / xs / dy
    if (tid < kRows * kMaxIn) {
      const int row = tid / kMaxIn, d = tid % kMaxIn;
      sm.xs[tid] = (d < a.dim_in && m0 + row < a.n) ? a.x[(m0 + row) * a.dim_in + d] : 0.f;
    }
    if (tid < kRows) {
      const float v = m0 + tid < a.n ? a.dy[m0 + tid] : 0.f;
      sm.dy[tid] = v;
      g_bhead += v;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- head: dz_{L-1} = dy w_head (.) d_{L-1}; dW_head += dy^T a_{L-1} ------------------------
    {
      const float wl = sm.w_last[col];
      const float* __restrict__ ga = a.act_last + m0 * kH + col;
      const float* __restrict__ gd = a.deriv[L - 1] + m0 * kH + col;
      float* __restrict__ gz = L > 1 ? a.dz[L - 1] + m0 * kH + col : nullptr;
#pragma unroll 2
      for (int rr = 0; rr < 32; rr += 8) {
        float av[8], dv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = r0 + rr + i;
          const bool live = full_tile || m0 + row < a.n;
          av[i] = live ? ga[row * kH] : 0.f;
          dv[i] = live ? gd[row * kH] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = r0 + rr + i;
          const float dyv = sm.dy[row];
          g_whead += dyv * av[i];
          const float dz = (dyv * wl) * dv[i];
          g_blast += dz;
          sm.img[row * kLd + col] = dz;
          if (gz && (full_tile || m0 + row < a.n)) gz[row * kH] = dz;
        }
      }
    }
    // ---- layers L-1 .. 1: da_{l-1} = dz_l W_l, dz_{l-1} = da_{l-1} (.) d_{l-1} ------------------
    for (int l = L - 1; l >= 1; --l) {
      f32x16 acc[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] = 0.f, acc[1][r] = 0.f;
      float dv[2][16];
#pragma unroll
      for (int kc = 0; kc < kChunks; ++kc, ++s) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // chunk s landed for every wave; the other buffer is free
        {
          const bool more_k = kc + 1 < kChunks;
          const int nl = more_k ? l : (l > 1 ? l - 1 : L - 1);
          if (more_k || l > 1 || tile + gridDim.x < tiles)
            issue_rows(a.w[nl], more_k ? kc + 1 : 0, sm.wbuf[(s + 1) & 1], wave, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        drip(2 * kc, 2 * kc + 2, full_tile);
        if (kc == 0) load_deriv(a.deriv[l - 1], m0, full_tile, dv);  // lands beside the MFMAs
        mma_chunk_rows(acc, a_row + kc * kKc, sm.wbuf[s & 1] + 4 * lh * kH + n0);
      }
      pend_l = -1;
      float sum0 = 0.f, sum1 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pz[0][r] = acc[0][r] * dv[0][r];
        pz[1][r] = acc[1][r] * dv[1][r];
        sum0 += pz[0][r];
        sum1 += pz[1][r];
      }
      sum0 += __shfl_xor(sum0, 32, 64);  // the two lane halves hold different rows of one column
      sum1 += __shfl_xor(sum1, 32, 64);
      if (lh == 0) {  // (layer, row block, column) has exactly one owner lane: plain read-add-write
        sm.gb[l - 1][rb][n0] += sum0;
        sm.gb[l - 1][rb][n0 + 32] += sum1;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // every wave has read the image for the last time
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          sm.img[(rb * 32 + acc_row(r, lh)) * kLd + n0 + t * 32] = pz[t][r];
      if (l - 1 >= 1) {  // dz_{l-1} feeds the weight gradient of layer l-1
        pend_l = l - 1, pend_m0 = m0;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // image = dz_0
    // ---- first layer: dW_first[col][d] += sum_rows dz_0[row][col] x[row][d] --------------------
#pragma unroll 4
    for (int r = 0; r < 32; ++r) {
      const float dz = sm.img[(r0 + r) * kLd + col];
#pragma unroll
      for (int d = 0; d < kMaxIn; ++d)
        if (d < a.dim_in) g_wfirst[d] += dz * sm.xs[(r0 + r) * kMaxIn + d];
    }
  }

  // ---- this workgroup's slab -------------------------------------------------------------------
  float* slab = a.partial + (int64_t)blockIdx.x * bwd_slab_floats(L);
  float* p_whead = slab;
  float* p_bhead = slab + kH;
  float* p_b = slab + kH + 4;
  float* p_wfirst = p_b + L * kH;
  __syncthreads();
  float* red = sm.img;  // scratch; (col, half) sums: two halves per column
  red[tid] = g_whead;
  red[kThreads + tid] = g_blast;
  __syncthreads();
  if (tid < kH) {
    p_whead[tid] = red[tid] + red[kH + tid];
    p_b[(L - 1) * kH + tid] = red[kThreads + tid] + red[kThreads + kH + tid];
    for (int l = 0; l + 1 < L; ++l) p_b[l * kH + tid] = sm.gb[l][0][tid] + sm.gb[l][1][tid];
  }
  __syncthreads();
#pragma unroll
  for (int d = 0; d < kMaxIn; ++d) {
    red[tid] = g_wfirst[d];
    __syncthreads();
    if (tid < kH) p_wfirst[tid * kMaxIn + d] = red[tid] + red[kH + tid];
    __syncthreads();
  }
  red[tid] = tid < kRows ? g_bhead : 0.f;
  __syncthreads();
  if (tid == 0) {
    float sb = 0.f;
    for (int c = 0; c < kRows; ++c) sb += red[c];
    p_bhead[0] = sb, p_bhead[1] = 0.f, p_bhead[2] = 0.f, p_bhead[3] = 0.f;
  }
}

// Sum the backward slabs in a fixed order and add them to the gradient tensors.
struct BwdReduceArgs {
  const float* partial;
  int slabs, n_sine, dim_in;
  float* d_w[kMaxSine + 1];
  float* d_b[kMaxSine + 1];
};

__global__ __launch_bounds__(256) void siren_bwd_reduce_kernel(const BwdReduceArgs r) {
  const int slab = bwd_slab_floats(r.n_sine);
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= slab) return;
  float sum = 0.f;
  const float* p = r.partial + e;
  int b = 0;
  for (; b + 8 <= r.slabs; b += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[(int64_t)(b + j) * slab];
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += v[j];
  }
  for (; b < r.slabs; ++b) sum += p[(int64_t)b * slab];
  const int L = r.n_sine;
  if (e < kH) {
    r.d_w[L][e] += sum;                                      // head weight (1, H)
  } else if (e < kH + 4) {
    if (e == kH) r.d_b[L][0] += sum;                         // head bias
  } else if (e < kH + 4 + L * kH) {
    const int q = e - kH - 4;
    r.d_b[q / kH][q % kH] += sum;                            // sine-layer biases
  } else {
    const int q = e - kH - 4 - L * kH, o = q / kMaxIn, d = q % kMaxIn;
    if (d < r.dim_in) r.d_w[0][o * r.dim_in + d] += sum;     // first layer (H, dim_in)
  }
}

// ------------------------------------------------------------------------------------------
// Weight gradient of a 256 x 256 layer: dW[n][k] = sum_rows dz[row][n] a[row][k].
// One persistent workgroup per CU holds the WHOLE 256 x 256 result in MFMA accumulators (8 waves x
// 2 x 4 tiles of 32 x 32 = 128 registers) and streams its share of the batch through LDS in
// 32-row chunks of dz and a (LDS-DMA, double buffered, rows as they lie in HBM: both operands are
// read along their contiguous axis, conflict-free ds_read_b32, no padding, no swizzle).  A chunk
// feeds 128 MFMAs per wave, so the chunk barrier costs ~2 %.  The per-workgroup results meet in a
// slab workspace and are summed in a fixed order (slab_sum_kernel).
struct WgradArgs {
  const float* dz;    // (n, H)
  const float* act;   // (n, H): the layer's input
  int64_t n;
  float* partial;     // [gridDim.x][H * H]
};

struct WgradSmem {
  float z[2][kKc * kH];
  float a[2][kKc * kH];
};

__global__ __launch_bounds__(kThreads) void siren_wgrad_kernel(const WgradArgs g) {
  __shared__ WgradSmem sm;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;  // 64 rows (n) x 128 columns (k) of dW per wave
  const int64_t chunks = (g.n + kKc - 1) / kKc;
  const int64_t per = (chunks + gridDim.x - 1) / gridDim.x;
  const int64_t c_lo = (int64_t)blockIdx.x * per, c_hi = c_lo + per < chunks ? c_lo + per : chunks;

  f32x16 acc[2][4];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;

  auto issue = [&](int64_t c, int buf) {  // 32 rows of dz and of a: 8 wave instructions per wave
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave * 4 + i;
      int64_t src = c * kKc + row;
      if (src >= g.n) src = g.n - 1;  // stays inside the buffers; such rows are zeroed below
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(g.dz + src * kH + lane * 4),
          (__attribute__((address_space(3))) void*)(sm.z[buf] + row * kH), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(g.act + src * kH + lane * 4),
          (__attribute__((address_space(3))) void*)(sm.a[buf] + row * kH), 16, 0, 0);
    }
  };
  if (c_lo < c_hi) issue(c_lo, 0);
  for (int64_t c = c_lo; c < c_hi; ++c) {
    const int buf = (int)(c - c_lo) & 1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // chunk c landed for every wave; the other buffer is free
    if (c + 1 < c_hi) issue(c + 1, buf ^ 1);
    if ((c + 1) * kKc > g.n) {  // the batch ends inside this chunk: rows beyond it contribute 0
      const int live = (int)(g.n - c * kKc);
      for (int e = tid; e < (kKc - live) * kH; e += kThreads) sm.z[buf][live * kH + e] = 0.f;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    const float* zp = sm.z[buf] + lh * kH + wr * 64 + l31;
    const float* ap = sm.a[buf] + lh * kH + wc * 128 + l31;
    float zv[2][2], av[2][4];
    auto fetch = [&](int b, int rp) {
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) zv[b][ti] = zp[2 * rp * kH + ti * 32];
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) av[b][tj] = ap[2 * rp * kH + tj * 32];
    };
    auto compute = [&](int b) {
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
          acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(zv[b][ti], av[b][tj], acc[ti][tj], 0, 0, 0);
    };
    fetch(0, 0);
#pragma unroll
    for (int rp = 0; rp < kKc / 2; ++rp) {
      if (rp + 1 < kKc / 2) fetch((rp + 1) & 1, rp + 1);
      __builtin_amdgcn_sched_barrier(0);
      compute(rp & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float* slab = g.partial + (int64_t)blockIdx.x * kH * kH;
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        slab[(wr * 64 + ti * 32 + acc_row(r, lh)) * kH + wc * 128 + tj * 32 + l31] = acc[ti][tj][r];
}

// dst[e] += sum over slabs of partial[slab][e], fixed order
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ partial, int slabs,
                                                       int count, float* __restrict__ dst) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= count) return;
  const float* p = partial + e;
  float sum = 0.f;
  int b = 0;
  for (; b + 8 <= slabs; b += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[(int64_t)(b + j) * count];
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += v[j];
  }
  for (; b < slabs; ++b) sum += p[(int64_t)b * count];
  dst[e] += sum;
}

int chain_blocks(int64_t n) { return (int)std::min<int64_t>(ceil_div(n, kRows), 256); }
int wgrad_blocks(int64_t n) { return (int)std::min<int64_t>(ceil_div(n, kKc), 256); }

bool chain_supported(int dim_in, int hidden, int n_sine, int dim_out) {
  return hidden == kH && dim_in >= 1 && dim_in <= kMaxIn && n_sine >= 1 && n_sine <= kMaxSine &&
         dim_out == 1;
}

}  // namespace
}  // namespace mri

using namespace mri;

extern "C" int mri_siren_supported(int32_t dim_in, int32_t hidden, int32_t n_sine_layers,
                                   int32_t dim_out) {
  return chain_supported(dim_in, hidden, n_sine_layers, dim_out) ? 1 : 0;
}

extern "C" int mri_siren_forward(const float* x, int64_t n, int32_t dim_in, int32_t hidden,
                                 int32_t n_sine_layers, const float* const* weight,
                                 const float* const* bias, float w0_first, float w0,
                                 float* const* act, float* const* deriv, float* y, void* stream) {
  MRI_REQUIRE(chain_supported(dim_in, hidden, n_sine_layers, 1),
              "fused SIREN chain: %d -> %d x %d -> 1 is not supported (hidden 256, dim_in <= 8, "
              "<= %d sine layers)", dim_in, hidden, n_sine_layers, kMaxSine);
  MRI_REQUIRE(n >= 0 && n < (1ll << 31), "n = %lld out of range", (long long)n);
  if (n == 0) return MRI_OK;
  MRI_REQUIRE(x && weight && bias && y, "NULL pointer");
  MRI_REQUIRE((act == nullptr) == (deriv == nullptr), "act and deriv go together");
  ChainArgs a{};
  a.x = x, a.n = n, a.dim_in = dim_in, a.n_sine = n_sine_layers;
  a.w0_first = w0_first, a.w0 = w0, a.y = y;
  for (int l = 0; l <= n_sine_layers; ++l) {
    MRI_REQUIRE(weight[l] && bias[l], "NULL parameter pointer (layer %d)", l);
    MRI_REQUIRE((reinterpret_cast<uintptr_t>(weight[l]) & 15) == 0,
                "weights must be 16-byte aligned (layer %d)", l);
    a.w[l] = weight[l], a.b[l] = bias[l];
  }
  if (act)
    for (int l = 0; l < n_sine_layers; ++l) {
      MRI_REQUIRE(act[l] && deriv[l], "NULL activation buffer (layer %d)", l);
      a.act[l] = act[l], a.deriv[l] = deriv[l];
    }
  hipStream_t st = (hipStream_t)stream;
  if (options().siren_two_per_cu) {  // two 256-thread workgroups per CU, 32-row tiles
    using G = Geo<32, 16>;
    const int blocks = (int)std::min<int64_t>(ceil_div(n, G::rows), 512);
    if (act)
      hipLaunchKernelGGL((siren_forward_kernel<true, G>), dim3(blocks), dim3(G::threads), 0, st, a);
    else
      hipLaunchKernelGGL((siren_forward_kernel<false, G>), dim3(blocks), dim3(G::threads), 0, st, a);
  } else {  // one 512-thread workgroup per CU, 64-row tiles
    using G = Geo<64, 32>;
    const int blocks = (int)std::min<int64_t>(ceil_div(n, G::rows), 256);
    if (act)
      hipLaunchKernelGGL((siren_forward_kernel<true, G>), dim3(blocks), dim3(G::threads), 0, st, a);
    else
      hipLaunchKernelGGL((siren_forward_kernel<false, G>), dim3(blocks), dim3(G::threads), 0, st, a);
  }
  return check_launch("siren_forward_kernel");
}

extern "C" int64_t mri_siren_backward_workspace_bytes(int64_t n, int32_t n_sine_layers) {
  if (n < 1 || n_sine_layers < 1 || n_sine_layers > kMaxSine) return -1;
  const int64_t chain = (int64_t)chain_blocks(n) * bwd_slab_floats(n_sine_layers);
  const int64_t wgrad = n_sine_layers > 1 ? (int64_t)wgrad_blocks(n) * kH * kH : 0;
  return std::max(chain, wgrad) * 4;
}

extern "C" int mri_siren_backward(const float* x, const float* dy, int64_t n, int32_t dim_in,
                                  int32_t hidden, int32_t n_sine_layers,
                                  const float* const* weight, const float* const* act,
                                  const float* const* deriv, float* const* dz,
                                  float* const* d_weight, float* const* d_bias, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
  MRI_REQUIRE(chain_supported(dim_in, hidden, n_sine_layers, 1),
              "fused SIREN chain: %d -> %d x %d -> 1 is not supported", dim_in, hidden,
              n_sine_layers);
  MRI_REQUIRE(n >= 0 && n < (1ll << 31), "n = %lld out of range", (long long)n);
  if (n == 0) return MRI_OK;
  MRI_REQUIRE(x && dy && weight && act && deriv && dz && d_weight && d_bias, "NULL pointer");
  const int L = n_sine_layers;
  const int64_t need = mri_siren_backward_workspace_bytes(n, L);
  MRI_REQUIRE(workspace && workspace_bytes >= need,
              "SIREN backward needs a workspace of %lld bytes (mri_siren_backward_workspace_bytes)",
              (long long)need);
  hipStream_t st = (hipStream_t)stream;
  BwdArgs a{};
  a.x = x, a.dy = dy, a.n = n, a.dim_in = dim_in, a.n_sine = L;
  a.partial = static_cast<float*>(workspace);
  for (int l = 0; l <= L; ++l) {
    MRI_REQUIRE(weight[l] && d_weight[l] && d_bias[l], "NULL parameter / gradient pointer (layer %d)", l);
    MRI_REQUIRE((reinterpret_cast<uintptr_t>(weight[l]) & 15) == 0, "weights must be 16-byte aligned");
    a.w[l] = weight[l];
  }
  for (int l = 0; l < L; ++l) {
    MRI_REQUIRE(act[l] && deriv[l] && (l == 0 || dz[l]), "NULL activation buffer (layer %d)", l);
    MRI_REQUIRE((reinterpret_cast<uintptr_t>(act[l]) & 15) == 0 &&
                    (l == 0 || (reinterpret_cast<uintptr_t>(dz[l]) & 15) == 0),
                "activation buffers must be 16-byte aligned");
    a.deriv[l] = deriv[l];
    a.dz[l] = dz[l];
  }
  a.act_last = act[L - 1];
  const int blocks = chain_blocks(n);
  hipLaunchKernelGGL(siren_backward_kernel, dim3(blocks), dim3(kThreads), 0, st, a);
  if (int rc = check_launch("siren_backward_kernel")) return rc;
  BwdReduceArgs r{};
  r.partial = a.partial, r.slabs = blocks, r.n_sine = L, r.dim_in = dim_in;
  for (int l = 0; l <= L; ++l) r.d_w[l] = d_weight[l], r.d_b[l] = d_bias[l];
  hipLaunchKernelGGL(siren_bwd_reduce_kernel, dim3((unsigned)ceil_div(bwd_slab_floats(L), 256)),
                     dim3(256), 0, st, r);
  if (int rc = check_launch("siren_bwd_reduce_kernel")) return rc;
  for (int l = L - 1; l >= 1; --l) {  // dW_l = dz_l^T a_{l-1}
    WgradArgs g{};
    g.dz = dz[l], g.act = act[l - 1], g.n = n, g.partial = static_cast<float*>(workspace);
    const int wb = wgrad_blocks(n);
    hipLaunchKernelGGL(siren_wgrad_kernel, dim3(wb), dim3(kThreads), 0, st, g);
    hipLaunchKernelGGL(slab_sum_kernel, dim3(kH * kH / 256), dim3(256), 0, st, g.partial, wb,
                       kH * kH, d_weight[l]);
    if (int rc = check_launch("siren_wgrad_kernel")) return rc;
  }
  return MRI_OK;
}

#ifdef SIREN_PROFILE
extern "C" int mri_debug_set_siren_profile(long long* device_buffer) {
  return hipMemcpyToSymbol(HIP_SYMBOL(mri::g_siren_profile), &device_buffer, sizeof(device_buffer)) ==
                 hipSuccess
             ? 0
             : -1;
}
#endif
