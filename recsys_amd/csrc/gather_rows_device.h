// Several plain row gathers as one unit of work (rsx_gather_rows_multi): its own launch (gather_rows_multi_k, din.hip) or extra
// workgroups of din.py's two prepare launches (rsx_din_prepare2_gather: the lookups depend on the batch's ids only, like the
// prepare kernels, so the 11 us bandwidth-bound gather runs beside those two latency-bound launches instead of after them).
#pragma once
#include "rsx_common.h"

struct GatherJobs {
  rsx_gather_job j[RSX_GATHER_MAX_JOBS];
  uint32_t blk_end[RSX_GATHER_MAX_JOBS];      // 256-thread workgroups of jobs 0 .. k
};

// workgroup BLK (256 threads, wave-uniform) of the job list G.  A MACRO on purpose: G must be the kernel's own by-value parameter,
// indexed with compile-time indices in the kernel's body -- handed to an inline function by reference, the whole struct was
// copied to scratch memory (392 bytes per lane; the two prepare launches of din.py went from 6 to 60 us: DESIGN.md 4c-6).
#define RSX_GATHER_ROWS_BLOCK(G, BLK)                                                                                     \
  do {                                                                                                                    \
    rsx_gather_job jb_ = (G).j[0];                                                                                        \
    uint32_t b0_ = 0;                                                                                                     \
    _Pragma("unroll") for (int k_ = 1; k_ < RSX_GATHER_MAX_JOBS; ++k_) {                                                  \
      if ((BLK) >= (G).blk_end[k_ - 1]) {                                                                                 \
        jb_ = (G).j[k_];                                                                                                  \
        b0_ = (G).blk_end[k_ - 1];                                                                                        \
      }                                                                                                                   \
    }                                                                                                                     \
    if (jb_.K == 1) { /* scalar rows (tf.gather of a 1-D variable stored with a row stride: DIN's item bias) */           \
      const long long e_ = (long long)((BLK) - b0_) * 256 + threadIdx.x;                                                  \
      if (e_ < jb_.n) jb_.out[e_ * jb_.ld_out] = jb_.table[((long long)jb_.row_base + jb_.ids[e_]) * jb_.ld_table];       \
    } else {                                                                                                              \
      const int lpr_ = jb_.K >> 2;                                                                                        \
      const long long t_ = (long long)((BLK) - b0_) * 256 + threadIdx.x;                                                  \
      const long long e_ = t_ / lpr_;                                                                                     \
      if (e_ < jb_.n) {                                                                                                   \
        const int q_ = (int)(t_ - e_ * lpr_);                                                                             \
        const long long row_ = (long long)jb_.row_base + jb_.ids[e_];                                                     \
        *reinterpret_cast<float4*>(jb_.out + e_ * jb_.ld_out + 4 * q_) =                                                  \
            reinterpret_cast<const float4*>(jb_.table)[row_ * lpr_ + q_];                                                 \
      }                                                                                                                   \
    }                                                                                                                     \
  } while (0)

// host: validates jobs [first, first + count) of jobs_h and packs them; *blocks = workgroups they need.  rsx_status.
static inline int gather_jobs_pack(const rsx_gather_job* jobs_h, int first, int count, GatherJobs& g, uint32_t* blocks) {
  uint32_t end = 0;
  for (int k = 0; k < RSX_GATHER_MAX_JOBS; ++k) {
    const rsx_gather_job& j = jobs_h[first + (k < count ? k : (count > 0 ? count - 1 : 0))];
    if (k < count) {
      if (!j.table || !j.ids || !j.out || j.n < 0 || j.K <= 0 || j.ld_out < j.K) return RSX_EINVAL;
      if (j.K == 1 ? j.ld_table < 1 : ((j.K & 3) || (j.ld_out & 3))) return RSX_EINVAL;
      end += (uint32_t)((j.n * (j.K == 1 ? 1 : (j.K >> 2)) + 255) / 256);
    }
    g.j[k] = j;
    g.blk_end[k] = end;
  }
  *blocks = end;
  return RSX_OK;
}
