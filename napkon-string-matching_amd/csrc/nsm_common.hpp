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

// The posting-entry format a table declares (nsm_hip.h: post_format / post_row_bits) must be one the kernels can decode for
// this many rows of this width: a struct filled for an older ABI, or by hand, would otherwise send them out of bounds.
inline int check_post_format(const nsm_set_table* t, const char* who) {
  int width_log = 0;
  while ((1 << width_log) < t->width) ++width_log;
  const int f = t->post_format, b = t->post_row_bits;
  const bool ok = f >= 0 && f <= 2 && (f == 0 ? b == 0 : (b > 0 && b + 2 * width_log <= 32 && static_cast<long long>(t->n) <= (1ll << b)));
  if (!ok) {
    set_error("%s: post_format %d with post_row_bits %d does not describe a table of %d rows of width %d", who, f, b, t->n, t->width);
    return NSM_E_BADARG;
  }
  return 0;
}

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

// Append the hits of a whole wavefront with ONE atomic on the counter: every lane calls this (all 64 enabled),
// `hit` says whether the lane has a record.  At low thresholds millions of pairs hit (the reference's default
// configuration keeps 2.6 % of them at cache_threshold 0.5) and one same-address atomic per record is then what
// the kernel waits for.
__device__ __forceinline__ void emit_hits_wave(nsm_hit* __restrict__ hits, unsigned long long cap,
                                               unsigned long long* __restrict__ count, bool hit, double score, int i,
                                               int j) {
  const unsigned long long who = __ballot(hit);
  if (who == 0ull) return;
  const int leader = __builtin_ctzll(who);
  const int lane = static_cast<int>(threadIdx.x & (kWave - 1));
  unsigned long long base = 0;
  if (lane == leader) base = atomicAdd(count, static_cast<unsigned long long>(__popcll(who)));
  const uint32_t lo = __builtin_amdgcn_readlane(static_cast<uint32_t>(base), leader);
  const uint32_t hi = __builtin_amdgcn_readlane(static_cast<uint32_t>(base >> 32), leader);
  base = (static_cast<unsigned long long>(hi) << 32) | lo;
  if (hit) {
    const unsigned long long pos = base + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(who >> 32),
                                                                     __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(who), 0u));
    if (pos < cap) {
      nsm_hit h;
      h.score = score;
      h.i = i;
      h.j = j;
      hits[pos] = h;
    }
  }
}

// Wave-wide max / or of a per-lane value, returned as a SCALAR (loops and branches on it stay on the
// SALU).  Four DPP steps leave every 16-lane row with its own result, four v_readlane + SALU ops
// combine the rows -- no LDS traffic, unlike the ds_bpermute chain __shfl_xor expands to.  Must be
// called with all 64 lanes enabled.
template <typename Op>
__device__ __forceinline__ uint32_t wave_reduce_u32(uint32_t v, Op op) {
  const int x = static_cast<int>(v);
  int r = x;
  r = static_cast<int>(op(static_cast<uint32_t>(r), static_cast<uint32_t>(__builtin_amdgcn_update_dpp(r, r, 0xB1, 0xf, 0xf, false))));   // quad_perm [1,0,3,2]
  r = static_cast<int>(op(static_cast<uint32_t>(r), static_cast<uint32_t>(__builtin_amdgcn_update_dpp(r, r, 0x4E, 0xf, 0xf, false))));   // quad_perm [2,3,0,1]
  r = static_cast<int>(op(static_cast<uint32_t>(r), static_cast<uint32_t>(__builtin_amdgcn_update_dpp(r, r, 0x141, 0xf, 0xf, false))));  // row_half_mirror
  r = static_cast<int>(op(static_cast<uint32_t>(r), static_cast<uint32_t>(__builtin_amdgcn_update_dpp(r, r, 0x140, 0xf, 0xf, false))));  // row_mirror
  const uint32_t a = static_cast<uint32_t>(__builtin_amdgcn_readlane(r, 0));
  const uint32_t b = static_cast<uint32_t>(__builtin_amdgcn_readlane(r, 16));
  const uint32_t c = static_cast<uint32_t>(__builtin_amdgcn_readlane(r, 32));
  const uint32_t d = static_cast<uint32_t>(__builtin_amdgcn_readlane(r, 48));
  return op(op(a, b), op(c, d));
}

// values must be >= 0
__device__ __forceinline__ int wave_max_i32(int v) {
  return static_cast<int>(wave_reduce_u32(static_cast<uint32_t>(v), [](uint32_t x, uint32_t y) { return x > y ? x : y; }));
}

__device__ __forceinline__ unsigned long long wave_or_u64(unsigned long long v) {
  auto bit_or = [](uint32_t x, uint32_t y) { return x | y; };
  const uint32_t lo = wave_reduce_u32(static_cast<uint32_t>(v), bit_or);
  const uint32_t hi = wave_reduce_u32(static_cast<uint32_t>(v >> 32), bit_or);
  return (static_cast<unsigned long long>(hi) << 32) | lo;
}

// One code unit of Hyyro's LCS recurrence, V' = (V + (V & M)) | (V ^ (V & M)), spelled as e32 instructions:
// left to itself hipcc fuses the expression into three v_bitop3_b32 and an add per code unit, and VOP3 ops
// issue at half the rate of e32 ops with VGPR operands (profiles/r01_valu_issue_rates_gfx950.txt).
__device__ __forceinline__ uint32_t lcs_step32(uint32_t v, uint32_t m) {
  uint32_t u, t, x, r;
  asm("v_and_b32 %0, %1, %2" : "=v"(u) : "v"(v), "v"(m));
  asm("v_add_u32 %0, %1, %2" : "=v"(t) : "v"(v), "v"(u));
  asm("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(v), "v"(u));
  asm("v_or_b32 %0, %1, %2" : "=v"(r) : "v"(t), "v"(x));
  return r;
}

__device__ __forceinline__ unsigned long long lcs_step64(unsigned long long v, unsigned long long m) {
  const uint32_t vl = static_cast<uint32_t>(v), vh = static_cast<uint32_t>(v >> 32);
  const uint32_t ml = static_cast<uint32_t>(m), mh = static_cast<uint32_t>(m >> 32);
  uint32_t ul, uh, xl, xh, rl, rh;
  asm("v_and_b32 %0, %1, %2" : "=v"(ul) : "v"(vl), "v"(ml));
  asm("v_and_b32 %0, %1, %2" : "=v"(uh) : "v"(vh), "v"(mh));
  unsigned long long t;  // one VALU op for the 64-bit add (v_add_co + v_addc_co issue in 9.3 cycles, this in 5.3)
  asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(t) : "v"(v), "v"((static_cast<unsigned long long>(uh) << 32) | ul));
  asm("v_xor_b32 %0, %1, %2" : "=v"(xl) : "v"(vl), "v"(ul));
  asm("v_xor_b32 %0, %1, %2" : "=v"(xh) : "v"(vh), "v"(uh));
  asm("v_or_b32 %0, %1, %2" : "=v"(rl) : "v"(static_cast<uint32_t>(t)), "v"(xl));
  asm("v_or_b32 %0, %1, %2" : "=v"(rh) : "v"(static_cast<uint32_t>(t >> 32)), "v"(xh));
  return (static_cast<unsigned long long>(rh) << 32) | rl;
}


__device__ __forceinline__ bool category_match(uint64_t cl, uint64_t cr, int mode) {
  // types/comparable_data.py:467-476; the predicate kind was chosen by the host from row 0.
  const bool inter = (cl & cr) != 0;
  return mode == NSM_CAT_INTERSECT_OR_BOTH_EMPTY ? (inter || ((cl | cr) == 0)) : inter;
}

}  // namespace nsm
