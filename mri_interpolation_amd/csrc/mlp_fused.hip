// Fully fused tiny-MLP (ReLU) forward + MSE + backward for gfx950: one kernel per training step.
//
// Replaces, for the decoder of BASELINE configs 2/4/5 (in -> H -> H -> 1, ReLU hidden, linear
// output; config/hash_config.json "network"), the op chain F.linear/ReLU x3 + F.mse_loss and its
// whole autograd (reference models.py:46-66, 730-744): 3 forward GEMMs, the loss, 5 backward
// GEMMs and the bias / last-layer reductions -- 12 launches and ~1 GB of activation traffic per
// step in the layer-wise path -- become ONE persistent kernel that never writes an activation
// to HBM.
//
// Layout of a workgroup (256 threads = 4 waves, one 64-coordinate tile at a time):
//   LDS  W2 (H x H), W1 (H x K_in), w3, b1, b2          loaded once, resident for the launch
//        x tile (64 x K_in), h1 (64 x H), h2 (64 x H)   h2 becomes dz2 in place, h1 becomes dz1
//   VGPR dW2 / dW1 accumulators (f32 MFMA C registers) live across ALL tiles of the workgroup;
//        bias and last-layer gradients accumulate in plain registers.
//   All seven products of a tile run on v_mfma_f32_32x32x2_f32 / 16x16x4_f32 (exact f32; the
//   128-wide decoder has since moved to the bf16 pipe with three-term operands, mlp_x3.hip: these
//   kernels serve H = 64, option mlp_x3 = 0 and the overlapped form), fragments read from LDS images
//   whose leading dimensions (H+1, K_in+1) make both orientations of every operand
//   conflict-free, so W2 and the activations are stored once and read as W2 and W2^T.
//   Input features arrive feature-major (K_in, n) straight from the hash-grid kernel and the
//   feature gradient leaves feature-major for the hash-grid backward.
// At the end every workgroup writes its partial parameter gradients and loss to a workspace
// slab; a second tiny kernel sums the slabs in a fixed order, so the result is bitwise
// reproducible (no float atomics anywhere on the training path).
#include <algorithm>

#include "common.h"
#include "mlp_fused.h"

namespace mri {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTile = 64;       // coordinates per workgroup iteration
constexpr int kThreads = 256;   // 4 waves

// Consumer side of the hand-off (MI355X_MICROARCH.md, inter-workgroup visibility): ONE lane polls
// with relaxed agent-scope loads, then ONE agent-scope acquire (invalidates this CU's L1) and
// its vmcnt(0); the caller's workgroup barrier follows, after which plain loads see the
// producer's write-through stores.  The poll is bounded (~1 s): a producer that never runs must
// not hang the GPU -- the kernel then finishes on garbage and reports through `status`.  The bound is
// spent ONCE per launch: after any workgroup has given up, `status` is set and every later wait
// (of every workgroup) returns at its first look instead of spinning out its own second.
__device__ __forceinline__ void wait_slice(const FusedArgs& a, int64_t slice) {
  int spins = 0;
  while (__hip_atomic_load(a.ready + slice, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <
         a.ready_target) {
    if (__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
    __builtin_amdgcn_s_sleep(8);
    if (++spins > (1 << 22)) {
      __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Segment timing for tools/mlp_segments.py (a tools-only build with -DMRI_MLP_PROFILE; the
// shipped library compiles these to nothing): shader-clock cycles per barrier-separated segment,
// split into work (mark -> barrier entry) and wait (barrier entry -> exit), summed over tiles.
#ifdef MRI_MLP_PROFILE
__device__ long long* g_mlp_profile = nullptr;
constexpr int kProfSlots = 32;
#define PROF_KERNEL_START const long long p_t0 = clock64();
#define PROF_BEGIN long long p_t = clock64(); long long p_acc[kProfSlots] = {}; p_acc[20] = p_t - p_t0;
#define PROF_MARK(i) { const long long p_n = clock64(); p_acc[i] += p_n - p_t; p_t = p_n; }
#define PROF_SYNC(i) { PROF_MARK(2 * (i)) __syncthreads(); PROF_MARK(2 * (i) + 1) }
#define PROF_END(wave_in_wg)                                                              \
  if (g_mlp_profile && (threadIdx.x & 63) == 0) {                                         \
    long long* dst = g_mlp_profile + ((int64_t)blockIdx.x * 8 + (wave_in_wg)) * kProfSlots; \
    for (int q = 0; q < kProfSlots; ++q) dst[q] = p_acc[q];                               \
  }
#else
#define PROF_KERNEL_START
#define PROF_BEGIN
#define PROF_MARK(i)
#define PROF_SYNC(i) __syncthreads();
#define PROF_END(w)
#endif

template <int H, int KP>
struct Smem {
  static constexpr int ldw2 = H + 1, ldw1 = KP + 1, lda = H + 1, ldx = KP + 1;
  float w2[H * ldw2];
  float w1[H * ldw1];
  float h1[kTile * lda];
  float h2[kTile * lda];
  float xs[kTile * ldx];
  float w3[H], b1[H], b2[H];
  float dy[kTile];
  float ypart[4 * kTile];
};


// acc[ti][tj] += A(i, c) * B(j, c) over `steps` pairs of contraction indices, 32x32 tiles.
// a / b point at this lane's element of tile (0,0) for contraction index 0; consecutive tiles are
// a_tile / b_tile floats apart, consecutive contraction PAIRS a_step / b_step floats apart.
template <int TI, int TJ, bool PIN = true>
__device__ __forceinline__ void mfma32(f32x16 (&acc)[TI][TJ], const float* __restrict__ a,
                                       int a_tile, int a_step, const float* __restrict__ b,
                                       int b_tile, int b_step, int steps) {
  // Software pipeline over groups of U contraction pairs: the LDS reads of group g+1 are
  // issued before the MFMAs of group g, so a wave that is alone on its matrix pipe does not
  // expose the ds_read latency once per group.  `steps` is a multiple of U at every call site.
  constexpr int U = TI * TJ >= 4 ? 2 : 4;  // >= 4 MFMAs (256 cycles) per group either way
  float av[2][U][TI], bv[2][U][TJ];
  auto fetch = [&](int buf, int s0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) av[buf][u][ti] = a[ti * a_tile + (s0 + u) * a_step];
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) bv[buf][u][tj] = b[tj * b_tile + (s0 + u) * b_step];
    }
  };
  auto compute = [&](int buf) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
          acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][u][ti], bv[buf][u][tj],
                                                             acc[ti][tj], 0, 0, 0);
  };
  // sched_barrier: hipcc otherwise sinks a group's reads below the previous group's MFMAs
  // (read, wait lgkmcnt(0), mfma, read, ... : one exposed LDS latency per pair)
  fetch(0, 0);
  int s = 0;
  for (; s + 2 * U <= steps; s += 2 * U) {
    fetch(1, s + U);
    if (PIN) __builtin_amdgcn_sched_barrier(0);
    compute(0);
    if (PIN) __builtin_amdgcn_sched_barrier(0);
    if (s + 2 * U < steps) fetch(0, s + 2 * U);
    if (PIN) __builtin_amdgcn_sched_barrier(0);
    compute(1);
    if (PIN) __builtin_amdgcn_sched_barrier(0);
  }
  if (s < steps) compute(0);  // steps = odd multiple of U
}

template <int TI, int TJ>
__device__ __forceinline__ void zero(f32x16 (&acc)[TI][TJ]) {
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;
}

// row of a 32x32 MFMA accumulator register: (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
__device__ __forceinline__ int acc_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

template <int H, int KP, bool TRAIN>
__global__ __launch_bounds__(kThreads, 2) void tiny_mlp_kernel(const FusedArgs a) {
  using S = Smem<H, KP>;
  __shared__ S sm;
  constexpr int NB = H / 32;        // 32-wide blocks of the hidden width
  constexpr int KB32 = KP / 32;     // 32-wide blocks of the (padded) input width
  // forward tiles: (64 x H) = 2 x NB tiles of 32x32 over 4 waves
  constexpr int FTI = NB == 4 ? 2 : 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int f_ti0 = NB == 4 ? 0 : (wave >> 1);   // first row block of this wave
  const int f_tj = NB == 4 ? wave : (wave & 1);  // column block of this wave
  // weight-gradient tiles: dW2 = NB x NB tiles; wave owns row block(s)
  constexpr int GTJ = NB == 4 ? 4 : 1;
  const int g_ob = NB == 4 ? wave : (wave >> 1);
  const int g_ib0 = NB == 4 ? 0 : (wave & 1);

  // ---- resident weights ---------------------------------------------------------------
  for (int e = tid; e < H * H; e += kThreads) sm.w2[(e / H) * S::ldw2 + (e % H)] = a.w2[e];
  for (int e = tid; e < H * KP; e += kThreads) {
    const int o = e / KP, k = e % KP;
    sm.w1[o * S::ldw1 + k] = k < a.k_in ? a.w1[o * a.k_in + k] : 0.f;
  }
  for (int e = tid; e < H; e += kThreads) {
    sm.w3[e] = a.w3[e];
    sm.b1[e] = a.b1[e];
    sm.b2[e] = a.b2[e];
  }
  const float b3 = a.b3[0];

  // persistent accumulators
  f32x16 g_w2[1][GTJ];
  f32x16 g_w1[1][1];
  zero(g_w2);
  zero(g_w1);
  float g_b1 = 0.f;                 // wave's forward column block: column f_tj*32 + l31 (both halves)
  float g_b2 = 0.f, g_w3 = 0.f;     // thread (o = tid % H, half = tid / H) partials
  float g_b3 = 0.f, loss = 0.f;     // threads < 64
  const bool dw1_owner = NB == 4 ? true : (KB32 == 2 ? true : wave < 2);
  const int w1_ob = NB == 4 ? wave : (KB32 == 2 ? (wave >> 1) : wave);
  const int w1_kb = NB == 4 ? 0 : (KB32 == 2 ? (wave & 1) : 0);

  // x tile: KP x 64 floats = KP * 16 float4 pieces, XV per thread, fetched one tile ahead into
  // registers (feature-major rows are contiguous along the batch)
  constexpr int XV = KP * (kTile / 4) / kThreads;
  static_assert(XV * kThreads == KP * (kTile / 4), "x tile must divide over the workgroup");
  auto load_x = [&](int64_t m0, float4 (&v)[XV]) {
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int e = tid + j * kThreads;
      const int k = e / (kTile / 4), c4 = (e % (kTile / 4)) * 4;
      v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < a.k_in && m0 + c4 < a.n) {
        const float* src = a.x + (int64_t)k * a.ld + m0 + c4;
        if (m0 + c4 + 3 < a.n && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
          v[j] = *reinterpret_cast<const float4*>(src);
        } else {
          v[j].x = src[0];
          if (m0 + c4 + 1 < a.n) v[j].y = src[1];
          if (m0 + c4 + 2 < a.n) v[j].z = src[2];
          if (m0 + c4 + 3 < a.n) v[j].w = src[3];
        }
      }
    }
  };
  const int64_t tiles = (a.n + kTile - 1) / kTile;
  // the tile's targets are prefetched with x (a load issued where the loss needs it exposes its
  // whole latency); dx of a tile is stored one tile late, right after the next tile's loads
  // were issued, so the wait for x_next never has a fresh store in front of it
  auto load_t = [&](int64_t m0) {
    return (TRAIN && tid < kTile && m0 + tid < a.n) ? a.target[m0 + tid] : 0.f;
  };
  constexpr int DXT = (KP / 16) * (kTile / 16) / 4;  // 16 x 16 dx tiles per wave
  f32x4 dx_pend[DXT];
  int64_t dx_m0 = -1;
  auto flush_dx = [&]() {
    if (dx_m0 < 0) return;
    const int l15 = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int u = 0; u < DXT; ++u) {
      const int tile = wave + 4 * u;
      const int kb = tile / (kTile / 16), cb = tile % (kTile / 16);
      const int64_t m = dx_m0 + cb * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = kb * 16 + lq * 4 + r;
        if (k < a.k_in && m < a.n) a.dx[(int64_t)k * a.ld + m] = dx_pend[u][r];
      }
    }
    dx_m0 = -1;
  };
  float4 x_next[XV];
  float t_next = 0.f;
  if ((int64_t)blockIdx.x < tiles) {
    load_x((int64_t)blockIdx.x * kTile, x_next);
    t_next = load_t((int64_t)blockIdx.x * kTile);
  }
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t m0 = t * kTile;
    __syncthreads();  // B0: previous tile is done with xs / h1 / h2
    // ---- stage x tile: xs[c][k] from the prefetched registers, then fetch the next tile -------
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int e = tid + j * kThreads;
      const int k = e / (kTile / 4), c4 = (e % (kTile / 4)) * 4;
      float* dst = sm.xs + c4 * S::ldx + k;
      dst[0] = x_next[j].x, dst[S::ldx] = x_next[j].y, dst[2 * S::ldx] = x_next[j].z,
      dst[3 * S::ldx] = x_next[j].w;
    }
    const float t_cur = t_next;
    if (t + gridDim.x < tiles) {
      load_x((t + gridDim.x) * kTile, x_next);
      t_next = load_t((t + gridDim.x) * kTile);
    }
    if (TRAIN) flush_dx();
    __syncthreads();  // B1

    // ---- layer 1: h1 = relu(x W1^T + b1) -----------------------------------------------------
    {
      f32x16 acc[FTI][1];
      zero(acc);
      mfma32<FTI, 1>(acc, sm.xs + (f_ti0 * 32 + l31) * S::ldx + lh, 32 * S::ldx, 2,
                     sm.w1 + (f_tj * 32 + l31) * S::ldw1 + lh, 0, 2, KP / 2);
      const int col = f_tj * 32 + l31;
      const float bias = sm.b1[col];
#pragma unroll
      for (int ti = 0; ti < FTI; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          sm.h1[((f_ti0 + ti) * 32 + acc_row(r, lh)) * S::lda + col] = fmaxf(acc[ti][0][r] + bias, 0.f);
    }
    __syncthreads();  // B2
    // ---- layer 2: h2 = relu(h1 W2^T + b2) ----------------------------------------------------
    {
      f32x16 acc[FTI][1];
      zero(acc);
      mfma32<FTI, 1>(acc, sm.h1 + (f_ti0 * 32 + l31) * S::lda + lh, 32 * S::lda, 2,
                     sm.w2 + (f_tj * 32 + l31) * S::ldw2 + lh, 0, 2, H / 2);
      const int col = f_tj * 32 + l31;
      const float bias = sm.b2[col];
#pragma unroll
      for (int ti = 0; ti < FTI; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          sm.h2[((f_ti0 + ti) * 32 + acc_row(r, lh)) * S::lda + col] = fmaxf(acc[ti][0][r] + bias, 0.f);
    }
    __syncthreads();  // B3
    // ---- layer 3 (one output): y[c] = h2[c] . w3 + b3, wave w sums its quarter of H ----------
    {
      constexpr int Q = H / 4;
      const float* row = sm.h2 + lane * S::lda + wave * Q;
      const float* w = sm.w3 + wave * Q;
      float s = 0.f;
#pragma unroll 8
      for (int o = 0; o < Q; ++o) s += row[o] * w[o];
      sm.ypart[wave * kTile + lane] = s;
    }
    __syncthreads();  // B4
    if (tid < kTile) {
      const int64_t m = m0 + tid;
      const float y = sm.ypart[tid] + sm.ypart[kTile + tid] + sm.ypart[2 * kTile + tid] +
                      sm.ypart[3 * kTile + tid] + b3;
      float d = 0.f;
      if (m < a.n) {
        if (a.y) a.y[m] = y;
        if (TRAIN) {
          const float diff = y - t_cur;
          loss += diff * diff;
          d = diff * a.grad_scale;
          g_b3 += d;
        }
      }
      sm.dy[tid] = d;
    }
    if (!TRAIN) continue;
    __syncthreads();  // B5
    // ---- dz2 = (dy w3^T) * (h2 > 0) in place; dW3, db2 partials -------------------------------
    {
      constexpr int HALVES = kThreads / H;          // 2 (H = 128) or 4 (H = 64)
      constexpr int ROWS = kTile / HALVES;
      const int o = tid % H, c0 = (tid / H) * ROWS;
      const float w3o = sm.w3[o];
#pragma unroll 8
      for (int c = c0; c < c0 + ROWS; ++c) {
        const float h = sm.h2[c * S::lda + o];
        const float d = sm.dy[c];
        g_w3 += d * h;
        const float dz = h > 0.f ? d * w3o : 0.f;
        g_b2 += dz;
        sm.h2[c * S::lda + o] = dz;
      }
    }
    __syncthreads();  // B6
    // ---- dW2 += dz2^T h1 ; dz1 = (dz2 W2) * (h1 > 0) ------------------------------------------
    mfma32<1, GTJ>(g_w2, sm.h2 + (g_ob * 32 + l31) + lh * S::lda, 0, 2 * S::lda,
                   sm.h1 + (g_ib0 * 32 + l31) + lh * S::lda, 32, 2 * S::lda, kTile / 2);
    f32x16 dz1[FTI][1];
    zero(dz1);
    mfma32<FTI, 1>(dz1, sm.h2 + (f_ti0 * 32 + l31) * S::lda + lh, 32 * S::lda, 2,
                   sm.w2 + (f_tj * 32 + l31) + lh * S::ldw2, 0, 2 * S::ldw2, H / 2);
    __syncthreads();  // B7: every wave is done reading h1 as an operand
    {
      const int col = f_tj * 32 + l31;
#pragma unroll
      for (int ti = 0; ti < FTI; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* p = sm.h1 + ((f_ti0 + ti) * 32 + acc_row(r, lh)) * S::lda + col;
          const float v = *p > 0.f ? dz1[ti][0][r] : 0.f;
          g_b1 += v;
          *p = v;
        }
    }
    __syncthreads();  // B8
    // ---- dW1 += dz1^T x -------------------------------------------------------------------
    if (dw1_owner)
      mfma32<1, 1>(g_w1, sm.h1 + (w1_ob * 32 + l31) + lh * S::lda, 0, 2 * S::lda,
                   sm.xs + (w1_kb * 32 + l31) + lh * S::ldx, 0, 2 * S::ldx, kTile / 2);
    // ---- dx^T = W1^T dz1^T (k_in x 64), 16x16 tiles so that all four waves take part ---------
    if (a.dx) {
      const int l15 = lane & 15, lq = lane >> 4;
#pragma unroll
      for (int u = 0; u < DXT; ++u) {
        const int tile = wave + 4 * u;
        const int kb = tile / (kTile / 16), cb = tile % (kTile / 16);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // lane group lq takes hidden units 8 lq + j + 32 m (step = 8 m + j): two lanes per LDS
        // bank instead of four (see the team kernel)
        const float* pa = sm.w1 + 8 * lq * S::ldw1 + kb * 16 + l15;      // A(i = k, kk = o)
        const float* pb = sm.h1 + (cb * 16 + l15) * S::lda + 8 * lq;     // B(kk = o, j = c)
#pragma unroll 8
        for (int s = 0; s < H / 4; ++s) {
          const int o = 32 * (s / 8) + s % 8;
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[o * S::ldw1], pb[o], acc, 0, 0, 0);
        }
        dx_pend[u] = acc;
      }
      dx_m0 = m0;
    }
  }
  if (!TRAIN) return;
  flush_dx();

  // ---- write this workgroup's partial gradients (plain stores, summed by reduce kernel) ----
  float* slab = a.partial + (int64_t)blockIdx.x * slab_floats(H, a.k_in);
  float* p_w1 = slab;
  float* p_b1 = p_w1 + H * a.k_in;
  float* p_w2 = p_b1 + H;
  float* p_b2 = p_w2 + H * H;
  float* p_w3 = p_b2 + H;
  float* p_b3 = p_w3 + H;
  __syncthreads();
  // dW2: wave's tiles
#pragma unroll
  for (int tj = 0; tj < GTJ; ++tj)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      p_w2[(g_ob * 32 + acc_row(r, lh)) * H + (g_ib0 + tj) * 32 + l31] = g_w2[0][tj][r];
  // dW1
  if (dw1_owner) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = w1_ob * 32 + acc_row(r, lh), k = w1_kb * 32 + l31;
      if (k < a.k_in) p_w1[o * a.k_in + k] = g_w1[0][0][r];
    }
  }
  // db1: column f_tj*32 + l31, two lane halves, and (H = 64) two waves per column block
  float* red = sm.h2;  // scratch
  red[tid] = g_b1;
  __syncthreads();
  if (tid < H) {
    float s = 0.f;
    for (int w = 0; w < 4; ++w) {
      const int tj = NB == 4 ? w : (w & 1);
      if (tj == tid / 32) s += red[w * 64 + (tid & 31)] + red[w * 64 + 32 + (tid & 31)];
    }
    p_b1[tid] = s;
  }
  __syncthreads();
  red[tid] = g_b2;
  red[kThreads + tid] = g_w3;
  __syncthreads();
  if (tid < H) {
    float s2 = 0.f, s3 = 0.f;
    for (int h = 0; h < kThreads / H; ++h) {
      s2 += red[h * H + tid];
      s3 += red[kThreads + h * H + tid];
    }
    p_b2[tid] = s2;
    p_w3[tid] = s3;
  }
  __syncthreads();
  if (tid < kTile) {
    red[tid] = g_b3;
    red[kTile + tid] = loss;
  }
  __syncthreads();
  if (tid == 0) {
    float s = 0.f, l = 0.f;
    for (int c = 0; c < kTile; ++c) {
      s += red[c];
      l += red[kTile + c];
    }
    p_b3[0] = s;
    p_b3[1] = l * a.inv_n;
  }
}

// ------------------------------------------------------------------------------------------
// H = 128: two teams of four waves per workgroup.
//
// With 156 KiB of weights + activations only ONE workgroup fits a CU, so with four waves every
// SIMD holds a single wave and nothing covers its VALU / LDS phases (epilogues, output layer,
// dz2): measured 55 % matrix-pipe busy.  Here a 512-thread workgroup runs two teams that share
// the resident weights; each team owns a 32-coordinate tile (its own x / h1 / h2 images) and
// walks the same nine barrier-separated segments.  Every SIMD then hosts one wave of each team,
// and the two cover each other's LDS and MFMA latencies.  All eight waves meet at every s_barrier;
// team 1 may start `stagger` segments late (option mlp_stagger: the same barrier count per team,
// extra barriers before team 1's first tile and after team 0's last).  Measured flat from 1 to
// 6 and 2 % faster at 0 (lockstep, no idle segments at either end of the launch): the default.
constexpr int kTeamTile = 32;
constexpr int kTeamThreads = 256;

// Where the fetch-before-compute order of mfma32 is pinned with sched_barriers (measured per
// segment): the short chains (layer 1, dW1: 16 MFMAs, where hipcc's own order exposes every
// group's LDS latency) gain, the long ones (layer 2, dW2 + dz1) are 1 % faster left to hipcc.
#ifndef PIN_S1
#define PIN_S1 true
#endif
#ifndef PIN_S2
#define PIN_S2 false
#endif
#ifndef PIN_S6
#define PIN_S6 false
#endif

template <int H, int KP>
struct TeamSmem {
  static constexpr int ldw2 = H + 1, ldw1 = KP + 1, lda = H + 1, ldx = KP + 1;
  float w2[H * ldw2];
  float w1[H * ldw1];
  float w3[H], b1[H], b2[H];
  struct Team {
    float h1[kTeamTile * lda];
    float h2[kTeamTile * lda];
    float xs[kTeamTile * ldx];
    float dy[kTeamTile];
    float ypart[8 * kTeamTile];
  } team[2];
};

template <int H, int KP, bool TRAIN>
__global__ __launch_bounds__(2 * kTeamThreads) void tiny_mlp_team_kernel(const FusedArgs a) {
  static_assert(H == 128 && KP == 32, "team kernel is laid out for 32 -> 128 -> 128 -> 1");
  using S = TeamSmem<H, KP>;
  __shared__ S sm;
  PROF_KERNEL_START
  // wave-uniform indices live in SGPRs (readfirstlane), so the addresses built from them do not
  // each take a vector register for the whole kernel
  const int team = __builtin_amdgcn_readfirstlane(threadIdx.x / kTeamThreads);
  const int tid = threadIdx.x % kTeamThreads;        // thread within the team
  const int tid_outer = tid;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave within the team: owns block w
  const int l31 = lane & 31, lh = lane >> 5;
  typename S::Team& tm = sm.team[team];

  {  // resident weights: W2 with 16-byte loads (all in flight before the first LDS store)
    constexpr int kW2Vec = H * H / 4 / (2 * kTeamThreads);
    float4 v[kW2Vec];
#pragma unroll
    for (int q = 0; q < kW2Vec; ++q)
      v[q] = reinterpret_cast<const float4*>(a.w2)[threadIdx.x + q * 2 * kTeamThreads];
#pragma unroll
    for (int q = 0; q < kW2Vec; ++q) {
      const int e = (threadIdx.x + q * 2 * kTeamThreads) * 4;
      float* dst = sm.w2 + (e / H) * S::ldw2 + (e % H);
      dst[0] = v[q].x, dst[1] = v[q].y, dst[2] = v[q].z, dst[3] = v[q].w;
    }
  }
  for (int e = threadIdx.x; e < H * KP; e += 2 * kTeamThreads) {
    const int o = e / KP, k = e % KP;
    sm.w1[o * S::ldw1 + k] = k < a.k_in ? a.w1[o * a.k_in + k] : 0.f;
  }
  for (int e = threadIdx.x; e < H; e += 2 * kTeamThreads) {
    sm.w3[e] = a.w3[e];
    sm.b1[e] = a.b1[e];
    sm.b2[e] = a.b2[e];
  }
  const float b3 = a.b3[0];

  // 512 threads leave 256 registers per lane.  Without this (empty) AGPR operand hipcc keeps all
  // of them architectural, puts the 96 accumulator registers there too and spills addresses;
  // with it the MFMA accumulators go to the 128 AccVGPRs and nothing spills.
  asm volatile("" ::"a"(0.f));
  f32x16 g_w2[1][4];
  f32x16 g_w1[1][1];
  zero(g_w2);
  zero(g_w1);
  float g_b1 = 0.f, g_b2 = 0.f, g_w3 = 0.f, g_b3 = 0.f, loss = 0.f;

  // x tile of a team: 32 features x 32 coordinates = 256 float4 pieces, one per thread
  // interior tiles of a full-width, 16-byte aligned input (wave-uniform test, the usual case) take
  // one unconditional load / four unconditional stores: the general path is ~100 branchy
  // instructions per tile in a segment that no MFMA covers
  const bool fast_io = a.k_in == KP && (a.ld % 4) == 0 &&
                       (reinterpret_cast<uintptr_t>(a.x) & 15) == 0;
  auto load_x = [&](int64_t m0, int tid) {
    const int k = tid / (kTeamTile / 4), c4 = (tid % (kTeamTile / 4)) * 4;
    if (fast_io && m0 + kTeamTile <= a.n)
      return *reinterpret_cast<const float4*>(a.x + (int64_t)k * a.ld + m0 + c4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < a.k_in && m0 + c4 < a.n) {
      const float* src = a.x + (int64_t)k * a.ld + m0 + c4;
      if (m0 + c4 + 3 < a.n && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
        v = *reinterpret_cast<const float4*>(src);
      } else {
        v.x = src[0];
        if (m0 + c4 + 1 < a.n) v.y = src[1];
        if (m0 + c4 + 2 < a.n) v.z = src[2];
        if (m0 + c4 + 3 < a.n) v.w = src[3];
      }
    }
    return v;
  };

  const int64_t tiles = (a.n + kTeamTile - 1) / kTeamTile;
  const int64_t rounds = (tiles + 2 * (int64_t)gridDim.x - 1) / (2 * (int64_t)gridDim.x);
  auto tile_of = [&](int64_t r) { return (r * gridDim.x + blockIdx.x) * 2 + team; };
  // target of the team's tile, one coordinate per lane of the first half-wave (prefetched with x:
  // a load issued where the loss needs it would expose its whole latency inside a segment)
  auto load_t = [&](int64_t m0, int tid) {
    return (TRAIN && tid < kTeamTile && m0 + tid < a.n) ? a.target[m0 + tid] : 0.f;
  };
  // dx of a tile leaves one tile late, right after the NEXT tile's loads were issued: the wait
  // for x_next at S0 then never has a freshly issued store in front of it
  const int dx_kb = w >> 1, dx_cb = w & 1;  // 2 x 2 tiles of 16 x 16: features x coordinates
  f32x4 dx_pend = {0.f, 0.f, 0.f, 0.f};
  int64_t dx_m0 = -1;
  auto flush_dx = [&](int lane) {
    if (dx_m0 < 0) return;
    const int l15 = lane & 15, lq = lane >> 4;
    const int64_t m = dx_m0 + dx_cb * 16 + l15;
    if (a.k_in == KP && dx_m0 + kTeamTile <= a.n) {
      float* dst = a.dx + (int64_t)(dx_kb * 16 + lq * 4) * a.ld + m;
#pragma unroll
      for (int q = 0; q < 4; ++q) dst[(int64_t)q * a.ld] = dx_pend[q];
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = dx_kb * 16 + lq * 4 + q;
        if (k < a.k_in && m < a.n) a.dx[(int64_t)k * a.ld + m] = dx_pend[q];
      }
    }
    dx_m0 = -1;
  };
  if (a.ready) {  // the first round's rows exist
    if (threadIdx.x == 0) wait_slice(a, 0);
    __syncthreads();
  }
  float4 x_next = load_x(tile_of(0) * kTeamTile, tid);  // beyond n -> zeros
  float t_next = load_t(tile_of(0) * kTeamTile, tid);

  __syncthreads();  // weights resident
  if (TRAIN && team == 1)  // stagger: team 1 runs `stagger` segments behind team 0
    for (int q = 0; q < a.stagger; ++q) __syncthreads();
  PROF_BEGIN
  for (int64_t r = 0; r < rounds; ++r) {
    // a team without a tile in the last round still walks the segments (rows >= n are inert)
    const int64_t m0 = tile_of(r) * kTeamTile;
    // thread indices re-derived per tile from an opaque copy: hipcc otherwise keeps ~30 address
    // registers (one per LDS access pattern below) alive across the whole loop
    int tid_opaque = tid_outer;
    asm volatile("" : "+v"(tid_opaque));
    const int tid = tid_opaque, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    // the rows prefetched below (round r + 1) must have been produced
    if (a.ready && r + 1 < rounds && threadIdx.x == 0) wait_slice(a, r + 1);
    PROF_SYNC(0)  // S0
    {
      const int k = tid / (kTeamTile / 4), c4 = (tid % (kTeamTile / 4)) * 4;
      float* dst = tm.xs + c4 * S::ldx + k;
      dst[0] = x_next.x, dst[S::ldx] = x_next.y, dst[2 * S::ldx] = x_next.z,
      dst[3 * S::ldx] = x_next.w;
    }
    const float t_cur = t_next;
    if (r + 1 < rounds) {
      x_next = load_x(tile_of(r + 1) * kTeamTile, tid);
      t_next = load_t(tile_of(r + 1) * kTeamTile, tid);
    }
    if (TRAIN) flush_dx(lane);
    PROF_SYNC(1)  // S1: layer 1
    {
      f32x16 acc[1][1];
      zero(acc);
      mfma32<1, 1, PIN_S1>(acc, tm.xs + l31 * S::ldx + lh, 0, 2, sm.w1 + (w * 32 + l31) * S::ldw1 + lh,
                   0, 2, KP / 2);
      const int col = w * 32 + l31;
      const float bias = sm.b1[col];
#pragma unroll
      for (int q = 0; q < 16; ++q)
        tm.h1[acc_row(q, lh) * S::lda + col] = fmaxf(acc[0][0][q] + bias, 0.f);
    }
    PROF_SYNC(2)  // S2: layer 2
    {
      f32x16 acc[1][1];
      zero(acc);
      mfma32<1, 1, PIN_S2>(acc, tm.h1 + l31 * S::lda + lh, 0, 2, sm.w2 + (w * 32 + l31) * S::ldw2 + lh,
                   0, 2, H / 2);
      const int col = w * 32 + l31;
      const float bias = sm.b2[col];
#pragma unroll
      for (int q = 0; q < 16; ++q)
        tm.h2[acc_row(q, lh) * S::lda + col] = fmaxf(acc[0][0][q] + bias, 0.f);
    }
    PROF_SYNC(3)  // S3: output layer, thread = (coordinate l31, eighth of H)
    {
      constexpr int Q = H / 8;
      const int part = w * 2 + lh;
      const float* row = tm.h2 + l31 * S::lda + part * Q;
      const float* wv = sm.w3 + part * Q;
      float sacc = 0.f;
#pragma unroll
      for (int o = 0; o < Q; ++o) sacc += row[o] * wv[o];
      tm.ypart[part * kTeamTile + l31] = sacc;
    }
    PROF_SYNC(4)  // S4: prediction, loss, dy
    if (tid < kTeamTile) {
      const int64_t m = m0 + tid;
      float y = b3;
#pragma unroll
      for (int q = 0; q < 8; ++q) y += tm.ypart[q * kTeamTile + tid];
      float d = 0.f;
      if (m < a.n) {
        if (a.y) a.y[m] = y;
        if (TRAIN) {
          const float diff = y - t_cur;
          loss += diff * diff;
          d = diff * a.grad_scale;
          g_b3 += d;
        }
      }
      tm.dy[tid] = d;
    }
    if (!TRAIN) continue;
    PROF_SYNC(5)  // S5: dz2 in place, dW3 / db2 partials
    {
      // all 16 + 16 reads first, then the arithmetic, then the 16 writes: a read -> write chain
      // per coordinate would pay the LDS latency 16 times
      constexpr int C = kTeamTile / 2;
      const int o = tid % H, c0 = (tid / H) * C;
      const float w3o = sm.w3[o];
      float hv[C], dv[C];
#pragma unroll
      for (int c = 0; c < C; ++c) hv[c] = tm.h2[(c0 + c) * S::lda + o];
#pragma unroll
      for (int c = 0; c < C; ++c) dv[c] = tm.dy[c0 + c];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        g_w3 += dv[c] * hv[c];
        const float dz = hv[c] > 0.f ? dv[c] * w3o : 0.f;
        g_b2 += dz;
        tm.h2[(c0 + c) * S::lda + o] = dz;
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the two running sums inside this segment
    }
    PROF_SYNC(6)  // S6: dW2 += dz2^T h1 ; dz1 = dz2 W2 (accumulators only)
    mfma32<1, 4, PIN_S6>(g_w2, tm.h2 + (w * 32 + l31) + lh * S::lda, 0, 2 * S::lda,
                 tm.h1 + l31 + lh * S::lda, 32, 2 * S::lda, kTeamTile / 2);
    f32x16 dz1[1][1];
    zero(dz1);
    mfma32<1, 1, PIN_S6>(dz1, tm.h2 + l31 * S::lda + lh, 0, 2, sm.w2 + (w * 32 + l31) + lh * S::ldw2, 0,
                 2 * S::ldw2, H / 2);
    PROF_SYNC(7)  // S7: dz1 = (.) * (h1 > 0) over h1
    {
      const int col = w * 32 + l31;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float* p = tm.h1 + acc_row(q, lh) * S::lda + col;
        const float v = *p > 0.f ? dz1[0][0][q] : 0.f;
        g_b1 += v;
        *p = v;
      }
    }
    PROF_SYNC(8)  // S8: dW1 += dz1^T x ; dx^T = W1^T dz1^T
    if (a.dx) {
      // One wave per SIMD works here (the other team sits in a short segment), so nothing but this
      // wave's own instruction stream hides latencies.  dW1 (16 steps of 32x32x2) and dx^T (32
      // steps of 16x16x4) are two independent accumulator chains: they are interleaved 1 : 2, in
      // groups of WU + 2 WU steps whose fragments are fetched one group ahead.
      constexpr int WU = 2, GROUPS = (kTeamTile / 2) / WU;
      static_assert(GROUPS * 2 * WU == H / 4 && GROUPS % 2 == 0, "dW1 : dx steps are 1 : 2");
      const int l15 = lane & 15, lq = lane >> 4;
      // dx^T: the four lane groups of a 16x16x4 step take hidden units 8 lq + j + 32 m (step =
      // 8 m + j) instead of 4 step + lq: with the odd leading dimensions the (l15, lq) lanes then
      // fall on banks l15 + 8 lq, two lanes per bank (the floor for 64 lanes) instead of four
      const float* pa = sm.w1 + 8 * lq * S::ldw1 + dx_kb * 16 + l15;
      const float* pb = tm.h1 + (dx_cb * 16 + l15) * S::lda + 8 * lq;
      const float* wa_p = tm.h1 + (w * 32 + l31) + lh * S::lda;   // dz1^T, step stride 2 lda
      const float* wb_p = tm.xs + l31 + lh * S::ldx;              // x^T,   step stride 2 ldx
      float wa[2][WU], wb[2][WU], xa[2][2 * WU], xb[2][2 * WU];
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      auto fetch = [&](int buf, int g) {
#pragma unroll
        for (int u = 0; u < WU; ++u) {
          wa[buf][u] = wa_p[(g * WU + u) * 2 * S::lda];
          wb[buf][u] = wb_p[(g * WU + u) * 2 * S::ldx];
        }
#pragma unroll
        for (int u = 0; u < 2 * WU; ++u) {
          const int st = g * 2 * WU + u, o = 32 * (st / 8) + st % 8;
          xa[buf][u] = pa[o * S::ldw1];
          xb[buf][u] = pb[o];
        }
      };
      auto compute = [&](int buf) {
#pragma unroll
        for (int u = 0; u < WU; ++u) {
          g_w1[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[buf][u], wb[buf][u], g_w1[0][0], 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[buf][2 * u], xb[buf][2 * u], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[buf][2 * u + 1], xb[buf][2 * u + 1], acc, 0, 0, 0);
        }
      };
      fetch(0, 0);
#pragma unroll
      for (int g = 0; g < GROUPS; g += 2) {
        fetch(1, g + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(0);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 2 < GROUPS) fetch(0, g + 2);
        __builtin_amdgcn_sched_barrier(0);
        compute(1);
        __builtin_amdgcn_sched_barrier(0);
      }
      dx_pend = acc;
      dx_m0 = m0;
    } else {
      mfma32<1, 1>(g_w1, tm.h1 + (w * 32 + l31) + lh * S::lda, 0, 2 * S::lda,
                   tm.xs + l31 + lh * S::ldx, 0, 2 * S::ldx, kTeamTile / 2);
    }
  }
  if (!TRAIN) return;
  PROF_MARK(18)
  flush_dx(lane);
  if (team == 0)
    for (int q = 0; q < a.stagger; ++q) __syncthreads();
  __syncthreads();  // both teams are done with their activation images

  // ---- one slab per workgroup: team 1 hands its sums to team 0 through LDS (the weight images
  //      are dead by now), team 0 adds its own on top and writes -- half the slab traffic of one
  //      slab per team, and the order of the additions is fixed
  float* x_w2 = sm.w2;  // [H][H] exchange images, plain leading dimensions
  float* x_w1 = sm.w1;  // [H][KP]
  if (team == 1) {
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int q = 0; q < 16; ++q)
        x_w2[(w * 32 + acc_row(q, lh)) * H + tj * 32 + l31] = g_w2[0][tj][q];
#pragma unroll
    for (int q = 0; q < 16; ++q) x_w1[(w * 32 + acc_row(q, lh)) * KP + l31] = g_w1[0][0][q];
  }
  float* red = sm.team[0].h2;   // scratch shared by both teams below: [2 teams][3][256] + tail
  float* red1 = sm.team[1].h2;
  (team == 0 ? red : red1)[tid] = g_b1;
  (team == 0 ? red : red1)[kTeamThreads + tid] = g_b2;
  (team == 0 ? red : red1)[2 * kTeamThreads + tid] = g_w3;
  if (tid < kTeamTile) {
    (team == 0 ? red : red1)[3 * kTeamThreads + tid] = g_b3;
    (team == 0 ? red : red1)[3 * kTeamThreads + kTeamTile + tid] = loss;
  }
  __syncthreads();
  if (team != 0) {
    PROF_MARK(21)
    PROF_END(threadIdx.x >> 6)
    return;
  }
  float* slab = a.partial + (int64_t)blockIdx.x * slab_floats(H, a.k_in);
  float* p_w1 = slab;
  float* p_b1 = p_w1 + H * a.k_in;
  float* p_w2 = p_b1 + H;
  float* p_b2 = p_w2 + H * H;
  float* p_w3 = p_b2 + H;
  float* p_b3 = p_w3 + H;
#pragma unroll
  for (int tj = 0; tj < 4; ++tj)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int e = (w * 32 + acc_row(q, lh)) * H + tj * 32 + l31;
      p_w2[e] = g_w2[0][tj][q] + x_w2[e];
    }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int r = w * 32 + acc_row(q, lh);
    if (l31 < a.k_in) p_w1[r * a.k_in + l31] = g_w1[0][0][q] + x_w1[r * KP + l31];
  }
  // per-column sums live in two threads per team (the two 16-coordinate halves of the dz2 / mask
  // passes for b2 / w3, the two lane halves of the accumulator layout for b1)
  if (tid < H) {
    const int i1 = (tid / 32) * 64 + (tid & 31);
    p_b1[tid] = (red[i1] + red[i1 + 32]) + (red1[i1] + red1[i1 + 32]);
    p_b2[tid] = (red[kTeamThreads + tid] + red[kTeamThreads + H + tid]) +
                (red1[kTeamThreads + tid] + red1[kTeamThreads + H + tid]);
    p_w3[tid] = (red[2 * kTeamThreads + tid] + red[2 * kTeamThreads + H + tid]) +
                (red1[2 * kTeamThreads + tid] + red1[2 * kTeamThreads + H + tid]);
  }
  if (tid == 0) {
    float sb = 0.f, sl = 0.f;
    for (int t = 0; t < 2; ++t) {
      const float* rr = t == 0 ? red : red1;
      for (int c = 0; c < kTeamTile; ++c) {
        sb += rr[3 * kTeamThreads + c];
        sl += rr[3 * kTeamThreads + kTeamTile + c];
      }
    }
    p_b3[0] = sb;
    p_b3[1] = sl * a.inv_n;
  }
  PROF_MARK(21)  // epilogue: team merge + slab
  PROF_END(threadIdx.x >> 6)
}

// Sum the per-workgroup slabs in a fixed order into the gradient tensors (accumulating).
struct ReduceArgs {
  const float* partial;
  int slabs, slab;
  int n_seg;
  int overwrite;
  int seg_begin[8];   // offsets inside a slab
  int seg_len[8];
  float* dst[8];
};

__global__ __launch_bounds__(256) void slab_reduce_kernel(const ReduceArgs r) {
  // 64 elements per workgroup; 4 thread groups each sum a quarter of the slabs (8 loads in
  // flight), then the quarters are added in a fixed order -> same bits every run.
  __shared__ float quarter[4][64];
  const int e = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
  const int per = (r.slabs + 3) / 4;
  const int b_lo = grp * per, b_hi = min(r.slabs, b_lo + per);
  float s = 0.f;
  if (e < r.slab) {
    const float* p = r.partial + e;
    int b = b_lo;
    for (; b + 8 <= b_hi; b += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = p[(int64_t)(b + j) * r.slab];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; b < b_hi; ++b) s += p[(int64_t)b * r.slab];
  }
  quarter[grp][threadIdx.x & 63] = s;
  __syncthreads();
  if (grp != 0 || e >= r.slab) return;
  s = ((quarter[0][threadIdx.x] + quarter[1][threadIdx.x]) + quarter[2][threadIdx.x]) +
      quarter[3][threadIdx.x];
#pragma unroll
  for (int g = 0; g < 8; ++g)
    if (g < r.n_seg && e >= r.seg_begin[g] && e < r.seg_begin[g] + r.seg_len[g])
      r.dst[g][e - r.seg_begin[g]] = r.overwrite ? s : r.dst[g][e - r.seg_begin[g]] + s;
}

template <int H, int KP>
int launch(const FusedArgs& a, bool train, int blocks, hipStream_t st) {
  if (train)
    hipLaunchKernelGGL((tiny_mlp_kernel<H, KP, true>), dim3(blocks), dim3(kThreads), 0, st, a);
  else
    hipLaunchKernelGGL((tiny_mlp_kernel<H, KP, false>), dim3(blocks), dim3(kThreads), 0, st, a);
  return check_launch("tiny_mlp_kernel");
}

int pick_blocks(int hidden, int64_t n) {
  if (hidden == 128)  // team kernel: one 512-thread workgroup per CU, two 32-row tiles per round
    return (int)std::min<int64_t>(ceil_div(n, 2 * kTeamTile), 256);
  return (int)std::min<int64_t>(ceil_div(n, kTile), 512);  // 52-75 KiB of LDS: two per CU
}

// one slab per workgroup; the workspace is sized for whichever kernel the options select later
int slab_count(int hidden, int k_in, int64_t n) {
  int blocks = pick_blocks(hidden, n);
  if (x3_supported(k_in, hidden)) blocks = std::max(blocks, x3_blocks(n));
  return blocks;
}

bool supported(int k_in, int hidden, int dim_out) {
  if (dim_out != 1 || k_in < 1) return false;
  if (hidden == 128) return k_in <= 32;
  if (hidden == 64) return k_in <= 64;
  return false;
}

// the bf16x3 kernel serves 128-wide decoders unless switched off (option mlp_x3) or the input is
// being produced beside the kernel (the ready counters are the team kernel's)
bool use_x3(const FusedArgs& a, int hidden) {
  return options().mlp_x3 && x3_supported(a.k_in, hidden) && x3_addressable(a) && a.ready == nullptr;
}

int dispatch(const FusedArgs& a, int hidden, bool train, int blocks, hipStream_t st) {
  if (use_x3(a, hidden)) return launch_tiny_mlp_x3(a, hidden, train, blocks, st);
  if (hidden == 128) {
    MRI_REQUIRE((reinterpret_cast<uintptr_t>(a.w2) & 15) == 0,
                "the f32-MFMA decoder kernel reads w2 in 16-byte pieces: w2 must be 16-byte aligned");  // tiny_mlp_kernel<128, ...> (one 4-wave team) is not instantiated: 7 % slower
    if (train)
      hipLaunchKernelGGL((tiny_mlp_team_kernel<128, 32, true>), dim3(blocks),
                         dim3(2 * kTeamThreads), 0, st, a);
    else
      hipLaunchKernelGGL((tiny_mlp_team_kernel<128, 32, false>), dim3(blocks),
                         dim3(2 * kTeamThreads), 0, st, a);
    return check_launch("tiny_mlp_team_kernel");
  }
  if (a.k_in <= 32) return launch<64, 32>(a, train, blocks, st);
  return launch<64, 64>(a, train, blocks, st);
}

}  // namespace
}  // namespace mri

using namespace mri;

extern "C" int mri_tiny_mlp_supported(int32_t k_in, int32_t hidden, int32_t dim_out) {
  return supported(k_in, hidden, dim_out) ? 1 : 0;
}

extern "C" int64_t mri_tiny_mlp_workspace_bytes(int32_t k_in, int32_t hidden, int64_t n) {
  if (!supported(k_in, hidden, 1)) return -1;
  return (int64_t)slab_count(hidden, k_in, std::max<int64_t>(n, 1)) * slab_floats(hidden, k_in) * 4;
}

extern "C" int mri_tiny_mlp_forward(const float* x, int64_t n, int32_t k_in, int32_t hidden,
                                    const float* w1, const float* b1, const float* w2,
                                    const float* b2, const float* w3, const float* b3, float* y,
                                    void* stream) {
  MRI_REQUIRE(supported(k_in, hidden, 1), "tiny MLP %d -> %d -> %d -> 1 is not supported", k_in,
              hidden, hidden);
  MRI_REQUIRE(n >= 0, "negative n");
  if (n == 0) return MRI_OK;
  MRI_REQUIRE(x && w1 && b1 && w2 && b2 && w3 && b3 && y, "NULL device pointer");
  FusedArgs a{};
  a.x = x, a.w1 = w1, a.b1 = b1, a.w2 = w2, a.b2 = b2, a.w3 = w3, a.b3 = b3, a.y = y;
  a.n = n, a.ld = n, a.k_in = k_in;
  return dispatch(a, hidden, false, use_x3(a, hidden) ? x3_blocks(n) : pick_blocks(hidden, n),
                  (hipStream_t)stream);
}

struct Overlap {
  const unsigned long long* ready = nullptr;
  unsigned long long target = 0;
  int* status = nullptr;
};

static int tiny_mlp_train_impl(int overwrite, int64_t ld, int64_t n_total, const float* x, const float* target, int64_t n, int32_t k_in,
                                  int32_t hidden, const float* w1, const float* b1,
                                  const float* w2, const float* b2, const float* w3,
                                  const float* b3, float grad_divisor, float* d_w1, float* d_b1,
                                  float* d_w2, float* d_b2, float* d_w3, float* d_b3, float* d_x,
                                  float* loss_out, float* y, void* workspace,
                                  int64_t workspace_bytes, void* stream,
                                  const Overlap& overlap = Overlap(), float* dx_absmax = nullptr) {
  MRI_REQUIRE(supported(k_in, hidden, 1), "tiny MLP %d -> %d -> %d -> 1 is not supported", k_in,
              hidden, hidden);
  MRI_REQUIRE(n >= 0 && grad_divisor > 0.f, "bad n / grad_divisor");
  MRI_REQUIRE(ld >= n && n_total >= n, "slice of %lld columns in a block of %lld, batch of %lld",
              (long long)n, (long long)ld, (long long)n_total);
  if (n == 0) return MRI_OK;
  MRI_REQUIRE(x && target && w1 && b1 && w2 && b2 && w3 && b3, "NULL device pointer");
  MRI_REQUIRE(d_w1 && d_b1 && d_w2 && d_b2 && d_w3 && d_b3 && loss_out, "NULL gradient pointer");
  const int slab = slab_floats(hidden, k_in);
  MRI_REQUIRE(workspace && workspace_bytes >= (int64_t)slab_count(hidden, k_in, n) * slab * 4,
              "tiny MLP needs a workspace of %lld bytes (mri_tiny_mlp_workspace_bytes)",
              (long long)slab_count(hidden, k_in, n) * slab * 4);
  FusedArgs a{};
  a.x = x, a.target = target;
  a.w1 = w1, a.b1 = b1, a.w2 = w2, a.b2 = b2, a.w3 = w3, a.b3 = b3;
  a.y = y, a.dx = d_x, a.partial = static_cast<float*>(workspace);
  a.n = n, a.ld = ld, a.k_in = k_in;
  a.grad_scale = (float)(2.0 / ((double)n_total * (double)grad_divisor));
  a.inv_n = (float)(1.0 / (double)n_total);
  a.stagger = std::min(std::max(options().mlp_stagger, 0), 8);
  a.ready = overlap.ready, a.ready_target = overlap.target, a.status = overlap.status;
  a.dx_absmax = reinterpret_cast<unsigned int*>(dx_absmax);
  if (dx_absmax && !(use_x3(a, hidden) && d_x))
    return fail(MRI_ERR_UNSUPPORTED, "max |d_x| per feature pair comes from the bf16-pipe decoder kernel only "
                                     "(mri_tiny_mlp_dx_absmax_supported), with d_x requested");
  const int blocks = use_x3(a, hidden) ? x3_blocks(n) : pick_blocks(hidden, n);
  const int slabs = blocks;
  if (int rc = dispatch(a, hidden, true, blocks, (hipStream_t)stream)) return rc;
  ReduceArgs r{};
  r.partial = a.partial, r.slabs = slabs, r.slab = slab, r.n_seg = 7, r.overwrite = overwrite;
  const int lens[7] = {hidden * k_in, hidden, hidden * hidden, hidden, hidden, 1, 1};
  float* dsts[7] = {d_w1, d_b1, d_w2, d_b2, d_w3, d_b3, loss_out};
  int off = 0;
  for (int g = 0; g < 7; ++g) {
    r.seg_begin[g] = off, r.seg_len[g] = lens[g], r.dst[g] = dsts[g];
    off += lens[g];
  }
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)ceil_div(slab, 64)), dim3(256), 0,
                     (hipStream_t)stream, r);
  return check_launch("slab_reduce_kernel");
}

extern "C" int mri_tiny_mlp_train(const float* x, const float* target, int64_t n, int32_t k_in,
                                  int32_t hidden, const float* w1, const float* b1,
                                  const float* w2, const float* b2, const float* w3,
                                  const float* b3, float grad_divisor, float* d_w1, float* d_b1,
                                  float* d_w2, float* d_b2, float* d_w3, float* d_b3, float* d_x,
                                  float* loss_out, float* y, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
  return tiny_mlp_train_impl(0, n, n, x, target, n, k_in, hidden, w1, b1, w2, b2, w3, b3, grad_divisor,
                             d_w1, d_b1, d_w2, d_b2, d_w3, d_b3, d_x, loss_out, y, workspace,
                             workspace_bytes, stream);
}

extern "C" int mri_tiny_mlp_train_overwrite(const float* x, const float* target, int64_t n,
                                            int32_t k_in, int32_t hidden, const float* w1,
                                            const float* b1, const float* w2, const float* b2,
                                            const float* w3, const float* b3, float grad_divisor,
                                            float* d_w1, float* d_b1, float* d_w2, float* d_b2,
                                            float* d_w3, float* d_b3, float* d_x, float* loss_out,
                                            float* y, void* workspace, int64_t workspace_bytes,
                                            void* stream) {
  return tiny_mlp_train_impl(1, n, n, x, target, n, k_in, hidden, w1, b1, w2, b2, w3, b3, grad_divisor,
                             d_w1, d_b1, d_w2, d_b2, d_w3, d_b3, d_x, loss_out, y, workspace,
                             workspace_bytes, stream);
}

extern "C" int mri_tiny_mlp_train_slice(const float* x, int64_t x_ld, const float* target, int64_t n,
                                        int64_t n_total, int32_t k_in, int32_t hidden,
                                        const float* w1, const float* b1, const float* w2,
                                        const float* b2, const float* w3, const float* b3,
                                        float grad_divisor, float* d_w1, float* d_b1, float* d_w2,
                                        float* d_b2, float* d_w3, float* d_b3, float* d_x,
                                        float* loss_out, float* y, int32_t overwrite,
                                        void* workspace, int64_t workspace_bytes, void* stream) {
  return tiny_mlp_train_impl(overwrite ? 1 : 0, x_ld, n_total, x, target, n, k_in, hidden, w1, b1,
                             w2, b2, w3, b3, grad_divisor, d_w1, d_b1, d_w2, d_b2, d_w3, d_b3, d_x,
                             loss_out, y, workspace, workspace_bytes, stream);
}

extern "C" int mri_hash_tiny_mlp_supported(const mri_grid_desc* grid, int32_t hidden) {
  if (!grid || !options().mlp_x3) return 0;
  int64_t rows = 0;
  for (int l = 0; l < grid->n_levels && l < MRI_MAX_LEVELS; ++l)
    rows = std::max<int64_t>(rows, (int64_t)grid->table_offset[l] + grid->table_size[l]);
  return x3_encode_supported(grid->dim, grid->n_features, grid->n_levels, hidden) && rows < (1ll << 29) ? 1 : 0;
}

extern "C" int mri_hash_tiny_mlp_train(const mri_grid_desc* grid, const float* table, const float* coords,
                                       const float* target, int64_t n, int32_t hidden, const float* w1,
                                       const float* b1, const float* w2, const float* b2, const float* w3,
                                       const float* b3, float grad_divisor, float* d_w1, float* d_b1,
                                       float* d_w2, float* d_b2, float* d_w3, float* d_b3, float* d_enc,
                                       int64_t d_enc_ld, float* loss_out, float* y, int32_t overwrite,
                                       void* workspace, int64_t workspace_bytes, void* stream) {
  MRI_REQUIRE(grid != nullptr, "grid descriptor is NULL");
  MRI_REQUIRE(mri_hash_tiny_mlp_supported(grid, hidden),
              "encoder (dim %d, %d levels x %d features) + decoder %d: no one-kernel form "
              "(mri_hash_tiny_mlp_supported)", grid->dim, grid->n_levels, grid->n_features, hidden);
  MRI_REQUIRE(n >= 0 && grad_divisor > 0.f, "bad n / grad_divisor");
  if (n == 0) return MRI_OK;
  const int k_in = 2 * grid->n_levels;
  MRI_REQUIRE(table && coords && target && w1 && b1 && w2 && b2 && w3 && b3, "NULL device pointer");
  MRI_REQUIRE(d_w1 && d_b1 && d_w2 && d_b2 && d_w3 && d_b3 && loss_out, "NULL gradient pointer");
  MRI_REQUIRE(!d_enc || d_enc_ld >= n, "d_enc rows are %lld apart, batch of %lld", (long long)d_enc_ld,
              (long long)n);
  const int slab = slab_floats(hidden, k_in);
  const int blocks = x3_blocks(n);
  MRI_REQUIRE(workspace && workspace_bytes >= (int64_t)blocks * slab * 4,
              "needs a workspace of %lld bytes (mri_tiny_mlp_workspace_bytes)", (long long)blocks * slab * 4);
  FusedArgs a{};
  a.target = target;
  a.w1 = w1, a.b1 = b1, a.w2 = w2, a.b2 = b2, a.w3 = w3, a.b3 = b3;
  a.y = y, a.dx = d_enc, a.partial = static_cast<float*>(workspace);
  a.n = n, a.ld = d_enc ? d_enc_ld : n, a.k_in = k_in;
  MRI_REQUIRE(x3_addressable(a), "d_enc block beyond 32-bit lane offsets");
  a.grad_scale = (float)(2.0 / ((double)n * (double)grad_divisor));
  a.inv_n = (float)(1.0 / (double)n);
  EncodeArgs e{};
  e.coords = coords, e.table = table, e.tab = make_tab(grid), e.n_levels = grid->n_levels, e.dim = grid->dim;
  if (int rc = launch_tiny_mlp_x3_encoded(a, e, hidden, blocks, (hipStream_t)stream)) return rc;
  ReduceArgs r{};
  r.partial = a.partial, r.slabs = blocks, r.slab = slab, r.n_seg = 7, r.overwrite = overwrite ? 1 : 0;
  const int lens[7] = {hidden * k_in, hidden, hidden * hidden, hidden, hidden, 1, 1};
  float* dsts[7] = {d_w1, d_b1, d_w2, d_b2, d_w3, d_b3, loss_out};
  int off = 0;
  for (int g = 0; g < 7; ++g) {
    r.seg_begin[g] = off, r.seg_len[g] = lens[g], r.dst[g] = dsts[g];
    off += lens[g];
  }
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)ceil_div(slab, 64)), dim3(256), 0,
                     (hipStream_t)stream, r);
  return check_launch("slab_reduce_kernel");
}

extern "C" int mri_tiny_mlp_dx_absmax_supported(int32_t k_in, int32_t hidden) {
  return options().mlp_x3 && x3_supported(k_in, hidden) ? 1 : 0;
}

extern "C" int mri_tiny_mlp_train_dx_absmax(const float* x, const float* target, int64_t n, int32_t k_in,
                                            int32_t hidden, const float* w1, const float* b1,
                                            const float* w2, const float* b2, const float* w3,
                                            const float* b3, float grad_divisor, float* d_w1, float* d_b1,
                                            float* d_w2, float* d_b2, float* d_w3, float* d_b3, float* d_x,
                                            float* loss_out, float* y, int32_t overwrite,
                                            float* dx_pair_absmax, void* workspace,
                                            int64_t workspace_bytes, void* stream) {
  MRI_REQUIRE(dx_pair_absmax != nullptr, "NULL dx_pair_absmax");
  return tiny_mlp_train_impl(overwrite ? 1 : 0, n, n, x, target, n, k_in, hidden, w1, b1, w2, b2, w3, b3,
                             grad_divisor, d_w1, d_b1, d_w2, d_b2, d_w3, d_b3, d_x, loss_out, y, workspace,
                             workspace_bytes, stream, Overlap(), dx_pair_absmax);
}

extern "C" int64_t mri_tiny_mlp_round_rows(int32_t k_in, int32_t hidden, int64_t n) {
  // rows one round of the training kernel's workgroups consumes (the team kernel only)
  if (!supported(k_in, hidden, 1) || hidden != 128 || n < 1) return 0;
  return (int64_t)pick_blocks(hidden, n) * 2 * kTeamTile;
}

extern "C" int mri_tiny_mlp_train_overlapped(const float* x, const float* target, int64_t n,
                                             int32_t k_in, int32_t hidden, const float* w1,
                                             const float* b1, const float* w2, const float* b2,
                                             const float* w3, const float* b3, float grad_divisor,
                                             float* d_w1, float* d_b1, float* d_w2, float* d_b2,
                                             float* d_w3, float* d_b3, float* d_x, float* loss_out,
                                             int32_t overwrite, const uint64_t* ready,
                                             uint64_t ready_target, int32_t* status,
                                             void* workspace, int64_t workspace_bytes,
                                             void* stream) {
  MRI_REQUIRE(mri_tiny_mlp_round_rows(k_in, hidden, n) > 0,
              "the overlapped decoder step needs the 128-wide kernel");
  MRI_REQUIRE(ready && status, "NULL ready / status pointer");
  Overlap o;
  o.ready = reinterpret_cast<const unsigned long long*>(ready);
  o.target = ready_target, o.status = status;
  return tiny_mlp_train_impl(overwrite ? 1 : 0, n, n, x, target, n, k_in, hidden, w1, b1, w2, b2,
                             w3, b3, grad_divisor, d_w1, d_b1, d_w2, d_b2, d_w3, d_b3, d_x,
                             loss_out, nullptr, workspace, workspace_bytes, stream, o);
}

#ifdef MRI_MLP_PROFILE
extern "C" int mri_debug_set_mlp_profile(long long* device_buffer) {
  return hipMemcpyToSymbol(HIP_SYMBOL(mri::g_mlp_profile), &device_buffer, sizeof(device_buffer)) ==
                 hipSuccess
             ? 0
             : -1;
}
#endif
