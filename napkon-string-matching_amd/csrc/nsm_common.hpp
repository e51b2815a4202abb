// Shared device/host helpers of libnsm_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nsm_hip.h"

namespace nsm {

constexpr int kWave = 64;            // CDNA wavefront
constexpr int kBlock = 256;          // 4 waves, one per SIMD
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr uint8_t kNever = 255;      // "no count can reach the threshold"

void set_error(const char* fmt, ...);
int hip_status(hipError_t err, const char* what);

// Append one hit.  The counter keeps counting past `cap` so the host can size a retry.
__device__ __forceinline__ void emit_hit(nsm_hit* __restrict__ hits, unsigned long long cap,
                                         unsigned long long* __restrict__ count, double score, int i,
                                         int j) {
  const unsigned long long pos = atomicAdd(count, 1ull);
  if (pos < cap) {
    nsm_hit h;
    h.score = score;
    h.i = i;
    h.j = j;
    hits[pos] = h;
  }
}

__device__ __forceinline__ int wave_first(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ bool category_match(uint64_t cl, uint64_t cr, int mode) {
  // types/comparable_data.py:467-476; the predicate kind was chosen by the host from row 0.
  const bool inter = (cl & cr) != 0;
  return mode == NSM_CAT_INTERSECT_OR_BOTH_EMPTY ? (inter || ((cl | cr) == 0)) : inter;
}

}  // namespace nsm
