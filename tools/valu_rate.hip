// Micro-benchmark: sustained wave64 issue rate of VALU ops on gfx950 (8 waves per SIMD, independent
// chains).  Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define ITER 2048
template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t s0, uint32_t s1) {
  uint32_t v[8], w[8];
  uint64_t pk[4], pw[4];
  for (int q = 0; q < 8; ++q) { v[q] = (threadIdx.x * 2654435761u + q) & 0xffff; w[q] = (v[q] ^ s1) & 0x3fff; }
  for (int q = 0; q < 4; ++q) { pk[q] = v[q]; pw[q] = w[q]; }
  if (KIND >= 1000) asm volatile("s_nop 0" ::: "s20");
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (KIND == 0) asm volatile("v_xor_b32 %0, %2, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 1) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 2) asm volatile("v_and_b32 %0, %1, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 3) asm volatile("v_and_b32 %0, %2, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 4) asm volatile("v_or_b32 %0, %1, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 5) asm volatile("v_xnor_b32 %0, %1, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 6) asm volatile("v_not_b32 %0, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 7) asm volatile("v_add_u32 %0, %1, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 8) asm volatile("v_add_u32 %0, %2, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 9) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 10) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 11) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 12) asm volatile("v_mov_b32 %0, %1" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 13) asm volatile("v_mov_b32 %0, %2" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 14) asm volatile("v_cmp_eq_u32 vcc, %1, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 15) asm volatile("v_cmp_eq_u32 vcc, %2, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 16) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 17) asm volatile("v_min_u32 %0, %1, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 18) asm volatile("v_min3_u32 %0, %0, %1, %3" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 19) asm volatile("v_min_f32 %0, %1, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 20) asm volatile("v_max_f32 %0, %1, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 21) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 22) asm volatile("v_add_f32 %0, %1, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 23) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 24) asm volatile("v_fma_f32 %0, %0, %1, %3" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 25) asm volatile("v_fmac_f32 %0, %1, %3" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 26) asm volatile("v_fma_f32 %0, %0, %2, %3" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 27) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 28) asm volatile("v_mad_u32_u24 %0, %0, %1, %3" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 29) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 30) asm volatile("v_add3_u32 %0, %0, %1, %3" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 31) asm volatile("v_or3_b32 %0, %0, %1, %3" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 32) asm volatile("v_xad_u32 %0, %0, %1, %3" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 33) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 34) asm volatile("v_and_or_b32 %0, %0, %1, %3" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 35) asm volatile("v_bfi_b32 %0, %0, %1, %3" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 36) asm volatile("v_perm_b32 %0, %0, %1, %3" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 37) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 38) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 39) asm volatile("v_sad_u8 %0, %0, %1, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 40) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 41) asm volatile("v_cmp_eq_f32 vcc, %1, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 42) asm volatile("v_med3_f32 %0, %0, %1, %3" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 43) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 44) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 45) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pk[q&3]) : "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 46) asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 47) asm volatile("v_add_co_u32 %0, vcc, %1, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 48) asm volatile("v_dot4_u32_u8 %0, %0, %1, %3" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 49) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 50) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 51) asm volatile("v_readlane_b32 s20, %0, 3" : "+v"(v[q]) : "v"(w[q]), "s"(s0), "v"(w[(q + 1) & 7]), "v"(pk[q&3]), "v"(pw[q&3]) : "vcc", "s20");
        if (KIND == 52) asm volatile("ds_read_b64 %0, %1" : "=v"(pk[q&3]) : "v"(w[q] & 0xff8));
      }
    }
  }
  uint32_t acc = 0;
  for (int q = 0; q < 8; ++q) acc ^= v[q] ^ w[q];
  for (int q = 0; q < 4; ++q) acc ^= (uint32_t)pk[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int KIND>
void run(const char* name, int insts_per_slot) {
  const int blocks = 256 * 8;
  uint32_t* out;
  (void)hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 12345u, 777u);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 12345u, 777u);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 3;
  const double winst = (double)blocks * 4 * ITER * 32 * insts_per_slot;
  const double rate = winst / (ms * 1e-3);
  printf("%-28s %8.3f ms  %.3e wave-inst/s  %.2f cyc/inst/SIMD @2.4GHz\n", name, ms, rate, 1024.0 * 2.4e9 / rate);
  (void)hipFree(out);
}
int main() {
  run<0>("v_xor_b32 sgpr", 1);
  run<1>("v_xor_b32 vgpr", 1);
  run<2>("v_and_b32 vgpr", 1);
  run<3>("v_and_b32 sgpr", 1);
  run<4>("v_or_b32 vgpr", 1);
  run<5>("v_xnor_b32", 1);
  run<6>("v_not_b32", 1);
  run<7>("v_add_u32", 1);
  run<8>("v_add_u32 sgpr", 1);
  run<9>("v_sub_u32", 1);
  run<10>("v_lshlrev_b32 imm", 1);
  run<11>("v_lshrrev_b32 imm", 1);
  run<12>("v_mov_b32 vgpr", 1);
  run<13>("v_mov_b32 sgpr", 1);
  run<14>("v_cmp_eq_u32 (vcc)", 1);
  run<15>("v_cmp_eq_u32 sgpr", 1);
  run<16>("v_cndmask_b32", 1);
  run<17>("v_min_u32", 1);
  run<18>("v_min3_u32", 1);
  run<19>("v_min_f32", 1);
  run<20>("v_max_f32", 1);
  run<21>("v_mul_f32", 1);
  run<22>("v_add_f32", 1);
  run<23>("v_sub_f32", 1);
  run<24>("v_fma_f32", 1);
  run<25>("v_fmac_f32", 1);
  run<26>("v_fma_f32 sgpr", 1);
  run<27>("v_mul_u32_u24", 1);
  run<28>("v_mad_u32_u24", 1);
  run<29>("v_mul_lo_u32", 1);
  run<30>("v_add3_u32", 1);
  run<31>("v_or3_b32", 1);
  run<32>("v_xad_u32", 1);
  run<33>("v_lshl_or_b32", 1);
  run<34>("v_and_or_b32", 1);
  run<35>("v_bfi_b32", 1);
  run<36>("v_perm_b32", 1);
  run<37>("v_bcnt_u32_b32", 1);
  run<38>("v_bfe_u32", 1);
  run<39>("v_sad_u8", 1);
  run<40>("v_cvt_f32_u32", 1);
  run<41>("v_cmp_eq_f32", 1);
  run<42>("v_med3_f32", 1);
  run<43>("v_pk_add_u16", 1);
  run<44>("v_pk_min_u16", 1);
  run<45>("v_pk_mul_f32", 1);
  run<46>("v_addc_co_u32", 1);
  run<47>("v_add_co_u32", 1);
  run<48>("v_dot4_u32_u8", 1);
  run<49>("v_mbcnt_lo", 1);
  run<50>("v_mov_b32 dpp row_shr", 1);
  run<51>("v_readlane (salu dst)", 1);
  run<52>("ds_read_b64 (same addr)", 1);
  return 0;
}
