// Streaming eval metrics on gfx950: tf.metrics.auc (200 thresholds) + tf.metrics.accuracy(labels, tf.round(pred))
// + the running sum of the per-batch losses, accumulated in device memory and read back ONCE per evaluate().
// Reference call sites: fm/fm.py:150-153 (eval_metric_ops of all five scripts), Estimator.evaluate(steps=200)
// fm/fm.py:221.  Semantics restated from TF 1.13 metrics_impl.py:
//   thresholds t_0 = -1e-7, t_i = i/199 (i = 1..198, computed in double, stored as fp32), t_199 = 1 + 1e-7;
//   tp[i] = #(label & pred > t_i), fp[i] = #(!label & pred > t_i)  -- a strict fp32 comparison against the fp32
//   threshold VALUE, not a rounded bucket index.
// Since the thresholds ascend, pred > t_i  <=>  i < k(pred) with k(pred) = #{i : t_i < pred}: the kernel finds k by
// binary search over the real threshold values (LDS) and counts examples per (label, k); tp / fp are suffix sums of
// that histogram, taken on the host when the counters are read back.  A NaN prediction compares false everywhere
// (k = 0), as in TF.  Integer atomics only: the result does not depend on the order of arrival.
//
// state layout (uint64, zeroed by the caller before the first batch):
//   [0 .. T]          hist_pos[k], k = 0..T       (T = number of thresholds = 200)
//   [T+1 .. 2T+1]     hist_neg[k]
//   [2T+2]            number of examples with round_half_even(pred) == label
//   [2T+3]            number of examples
//   [2T+4]            number of batches
//   [2T+5]            sum of the batch losses, as the bit pattern of a double
#include "rsx_common.h"

#define MET_T 256
#define MET_MAX_TH 256

__global__ __launch_bounds__(MET_T) void eval_metrics_k(const float* __restrict__ prob, const float* __restrict__ labels,
                                                        const float* __restrict__ thresholds, int T,
                                                        const float* __restrict__ batch_loss, unsigned long long* state,
                                                        int B) {
  __shared__ float th[MET_MAX_TH];
  __shared__ unsigned int hp[MET_MAX_TH + 1], hn[MET_MAX_TH + 1];
  __shared__ unsigned int correct;
  for (int i = threadIdx.x; i <= T; i += MET_T) {
    if (i < T) th[i] = thresholds[i];
    hp[i] = 0u;
    hn[i] = 0u;
  }
  if (threadIdx.x == 0) correct = 0u;
  __syncthreads();
  unsigned int my_correct = 0u;
  for (int b = blockIdx.x * MET_T + threadIdx.x; b < B; b += gridDim.x * MET_T) {
    const float p = prob[b];
    const bool y = labels[b] > 0.5f;                      // tf.cast(labels, bool) of a 0/1 label
    // k = #{i : th[i] < p} = first index whose threshold is NOT below p (ascending thresholds)
    int lo = 0, hi = T;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (p > th[mid]) lo = mid + 1; else hi = mid;
    }
    atomicAdd(y ? &hp[lo] : &hn[lo], 1u);
    my_correct += (rintf(p) == labels[b]) ? 1u : 0u;       // tf.round = round half to even; tf.equal on floats
  }
  if (my_correct) atomicAdd(&correct, my_correct);
  __syncthreads();
  for (int i = threadIdx.x; i <= T; i += MET_T) {
    if (hp[i]) atomicAdd(&state[i], (unsigned long long)hp[i]);
    if (hn[i]) atomicAdd(&state[T + 1 + i], (unsigned long long)hn[i]);
  }
  if (threadIdx.x == 0) {
    if (correct) atomicAdd(&state[2 * T + 2], (unsigned long long)correct);
    if (blockIdx.x == 0) {
      atomicAdd(&state[2 * T + 3], (unsigned long long)B);
      atomicAdd(&state[2 * T + 4], 1ull);
      if (batch_loss) {       // one adder per launch and launches of one stream are ordered: a fixed summation order
        double* ls = reinterpret_cast<double*>(&state[2 * T + 5]);
        *ls = *ls + (double)batch_loss[0];
      }
    }
  }
}

extern "C" int rsx_eval_metrics_state_words(int num_thresholds) { return 2 * num_thresholds + 6; }

extern "C" int rsx_eval_metrics_update(const float* prob, const float* labels, const float* thresholds, int num_thresholds,
                                       const float* batch_loss, uint64_t* state, int B, rsx_stream_t stream) {
  if (!prob || !labels || !thresholds || !state || B < 0 || num_thresholds < 2 || num_thresholds > MET_MAX_TH)
    return RSX_EINVAL;
  if (B == 0) return RSX_OK;
  int blocks = (B + MET_T - 1) / MET_T;
  if (blocks > 256) blocks = 256;
  RSX_LAUNCH(eval_metrics_k, dim3(blocks), dim3(MET_T), 0, rsx_s(stream), prob, labels, thresholds,
                     num_thresholds, batch_loss, reinterpret_cast<unsigned long long*>(state), B);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}
