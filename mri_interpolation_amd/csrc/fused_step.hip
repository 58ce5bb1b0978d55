// One training step of the hash-grid + tiny-MLP path queued by ONE call: the host side of the hot loop.
//
// The step is ~12 kernel launches, a few memsets and a stream fork / join.  Queued from Python -- one
// ctypes call per op through torch's stream and event objects -- that costs 0.2 ... 0.55 ms of host time
// per step depending on the box (tools/host_profile.py: 57 % of it interpreter and wrapper overhead), against
// 0.52 ms of GPU time: on a slow host the loop is HOST-bound.  hipGraph replays do not help on this
// runtime (a replay of the same step cost the host 0.46 ms: EXPERIMENTS.md, round 3).  This entry
// point composes the library's own entry points in C: the host cost of a step is one call plus the HIP
// launches themselves.
//
// Replaces the per-batch body of pl.Trainer.fit for the reference's HashMLP path (reference
// launcher.py:156-165 -> models.py:61-70: training_step, backward, Adam.step; datamodules.py:198-205: the
// DataLoader producing the next batch meanwhile), in the steady state of FusedStep.train_step with
// count_ahead: same launches, same order, same data -- bit-identical parameters.
#include <algorithm>
#include <cstdint>

#include "common.h"
#include "hashgrid_common.h"  // FinTab: the table gradient's last conversion, done by the Adam kernel

// Host-side cost of each call of a step, for tools/step_trace.py (a tools-only build with -DMRI_STEP_TRACE;
// the shipped library compiles these to nothing): nanoseconds per slot, summed over calls.
#ifdef MRI_STEP_TRACE
#include <time.h>
static long long g_trace_ns[24];
static long long g_trace_calls;
static inline long long trace_now() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec * 1000000000ll + t.tv_nsec;
}
#define TRACE_BEGIN long long t_prev = trace_now(); ++g_trace_calls;
#define TRACE(i) { const long long t_now = trace_now(); g_trace_ns[i] += t_now - t_prev; t_prev = t_now; }
extern "C" void mri_debug_step_trace(long long* out25) {
  for (int i = 0; i < 24; ++i) out25[i] = g_trace_ns[i];
  out25[24] = g_trace_calls;
}
#else
#define TRACE_BEGIN
#define TRACE(i)
#endif

extern "C" int64_t mri_fused_step_args_bytes(void) { return (int64_t)sizeof(mri_fused_step_args); }

extern "C" int mri_fused_step(const mri_fused_step_args* a) {
  using namespace mri;
  MRI_REQUIRE(a != nullptr && a->grid != nullptr, "NULL arguments");
  MRI_REQUIRE(a->n >= 1 && a->stream_side && a->ev_fork && a->ev_join, "fused step: n >= 1, a side stream and two events");
  for (int i = 1; i < 5; ++i)
    MRI_REQUIRE((a->ev_phase[i] != nullptr) == (a->ev_phase[0] != nullptr), "fused step: five phase events or none");
  hipStream_t main = (hipStream_t)a->stream, side = (hipStream_t)a->stream_side;
  hipEvent_t fork = (hipEvent_t)a->ev_fork, join = (hipEvent_t)a->ev_join;
  TRACE_BEGIN
  // join: the previous call's side work (this batch produced, its records counted) is done
  if (a->join_pending && hipStreamWaitEvent(main, join, 0) != hipSuccess)
    return fail(MRI_ERR_LAUNCH, "fused step: hipStreamWaitEvent(join)");
  // fork: everything the side stream touches (the other batch buffer, record workspace and absmax buffer)
  // was last used by work queued on `main` before this point
  if (hipEventRecord(fork, main) != hipSuccess || hipStreamWaitEvent(side, fork, 0) != hipSuccess)
    return fail(MRI_ERR_LAUNCH, "fused step: fork");
  TRACE(0)
  const mri_grid_desc* g = a->grid;
  const int64_t n = a->n, ld = n;  // feature-major blocks (2 L, n)
  const int32_t k_in = g->n_levels * g->n_features;
  const float divisor = a->grad_divisor > 0.f ? a->grad_divisor : 1.0f;
  int rc;
  if (!a->counted) {  // this batch was not counted ahead: count it now, on the side stream (as the eager step does)
    if ((rc = mri_hashgrid_backward_prepare(g, a->coords, n, a->bwd_method, a->bwd_ws, a->bwd_ws_bytes, side)))
      return rc;
  }
  if (a->next_idx) {  // next batch: sample + gather, zero its absmax buffer, count its records
    if ((rc = mri_sample_indices(a->seed, a->first, a->lo, a->hi, a->next_n, a->next_idx, side))) return rc;
    TRACE(1)
    if ((rc = mri_gather_batch(a->next_idx, a->next_n, a->dim, a->shape, a->axes, a->axis_offset, a->volume,
                               a->next_coords, a->next_target, side)))
      return rc;
    TRACE(2)
    if (a->next_absmax && hipMemsetAsync(a->next_absmax, 0, 32 * sizeof(float), side) != hipSuccess)
      return fail(MRI_ERR_LAUNCH, "fused step: memset");
    TRACE(3)
    if (a->next_bwd_ws &&
        (rc = mri_hashgrid_backward_prepare(g, a->next_coords, a->next_n, a->bwd_method, a->next_bwd_ws,
                                            a->next_bwd_ws_bytes, side)))
      return rc;
    TRACE(4)
  }
  auto phase = [&](int i) {  // a measuring caller's timing events (all or none)
    return !a->ev_phase[0] || hipEventRecord((hipEvent_t)a->ev_phase[i], main) == hipSuccess;
  };
  if (!phase(0)) return fail(MRI_ERR_LAUNCH, "fused step: hipEventRecord(phase)");
  if ((rc = mri_hashgrid_forward(g, a->coords, n, a->table, a->enc, (int64_t)g->n_features * ld, 1, ld, main)))
    return rc;
  if (!phase(1)) return fail(MRI_ERR_LAUNCH, "fused step: hipEventRecord(phase)");
  TRACE(5)
  if (a->absmax)
    rc = mri_tiny_mlp_train_dx_absmax(a->enc, a->target, n, k_in, a->hidden, a->w1, a->b1, a->w2, a->b2, a->w3,
                                      a->b3, divisor, a->d_w1, a->d_b1, a->d_w2, a->d_b2, a->d_w3, a->d_b3, a->d_enc,
                                      a->loss, nullptr, 1, a->absmax, a->tiny_ws, a->tiny_ws_bytes, main);
  else
    rc = mri_tiny_mlp_train_overwrite(a->enc, a->target, n, k_in, a->hidden, a->w1, a->b1, a->w2, a->b2, a->w3,
                                      a->b3, divisor, a->d_w1, a->d_b1, a->d_w2, a->d_b2, a->d_w3, a->d_b3, a->d_enc,
                                      a->loss, nullptr, a->tiny_ws, a->tiny_ws_bytes, main);
  if (rc) return rc;
  if (!phase(2)) return fail(MRI_ERR_LAUNCH, "fused step: hipEventRecord(phase)");
  TRACE(6)
  if (!a->counted) {  // the scatter needs this batch's count (the first step of a loop only: in the steady
                      // state the count is a step old and was joined at the top)
    if (hipEventRecord(join, side) != hipSuccess || hipStreamWaitEvent(main, join, 0) != hipSuccess)
      return fail(MRI_ERR_LAUNCH, "fused step: join (count)");
  }
  const int32_t flags = a->bwd_method | MRI_BWD_PREPARED | MRI_BWD_OVERWRITE;
  // The levels whose sums meet in the workspace's int64 area end with a conversion launch (bin_finalize_kernel, 8 us
  // between the accumulation and Adam): when Adam follows at once over a range that holds the table's gradient, its
  // kernel converts as it fetches (same expression, same bits) and the launch never happens.
  FinTab fin{};
  const bool fold = a->n_params > 0 && a->d_table >= a->grad && a->d_table < a->grad + a->n_params &&
                    ((reinterpret_cast<uintptr_t>(a->param) | reinterpret_cast<uintptr_t>(a->grad) |
                      reinterpret_cast<uintptr_t>(a->exp_avg) | reinterpret_cast<uintptr_t>(a->exp_avg_sq)) & 15) == 0;
  if (fold) set_finalize_export(&fin);
  rc = mri_hashgrid_backward_scaled(g, a->coords, a->d_enc, n, (int64_t)g->n_features * ld, 1, ld, a->d_table, flags,
                                    0xffffffffu, a->absmax, a->bwd_ws, a->bwd_ws_bytes, main);
  set_finalize_export(nullptr);  // (`fin` dies with this call)
  if (rc) return rc;
  if (!phase(3)) return fail(MRI_ERR_LAUNCH, "fused step: hipEventRecord(phase)");
  TRACE(7)
  if (fin.n > 0) {  // (taken: Adam owes the conversion)
    const int64_t off = a->d_table - a->grad;
    fin.lo = INT64_MAX, fin.hi = 0;
    for (int s = 0; s < fin.n; ++s) {
      fin.seg[s].begin += off;
      fin.lo = std::min(fin.lo, fin.seg[s].begin);
      fin.hi = std::max(fin.hi, fin.seg[s].begin + fin.seg[s].words);
    }
    MRI_REQUIRE(fin.hi <= a->n_params, "fused step: table gradient beyond the optimizer's range");
    rc = adam_step_fin(a->param, a->grad, a->exp_avg, a->exp_avg_sq, a->n_params, a->lr, a->beta1, a->beta2, a->eps,
                       a->step, a->grad_scale, fin, main);
  } else if (a->n_params > 0) {  // (0: a data-parallel caller reduces the gradient over the ranks first, then steps)
    rc = mri_adam_step(a->param, a->grad, a->exp_avg, a->exp_avg_sq, a->n_params, a->lr, a->beta1, a->beta2, a->eps,
                       a->step, a->grad_scale, main);
  }
  if (rc) return rc;
  if (!phase(4)) return fail(MRI_ERR_LAUNCH, "fused step: hipEventRecord(phase)");
  TRACE(8)
  // the side work is joined by the NEXT call (join_pending = 1), or by the caller through `ev_join`
  if (hipEventRecord(join, side) != hipSuccess) return fail(MRI_ERR_LAUNCH, "fused step: hipEventRecord(join)");
  TRACE(9)
  return MRI_OK;
}
