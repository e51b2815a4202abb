// gfx950: issue rate of 64-bit add forms (8 waves/SIMD, independent chains).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define ITER 2048
template <int KIND>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint64_t s0) {
  uint64_t v[8], w[8];
  for (int q = 0; q < 8; ++q) { v[q] = threadIdx.x * 2654435761ull + q; w[q] = v[q] ^ s0; }
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (KIND == 0) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(v[q]) : "v"(w[q]));
        if (KIND == 1) { uint32_t lo = (uint32_t)v[q], hi = (uint32_t)(v[q] >> 32); asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"((uint32_t)w[q]), "v"((uint32_t)(w[q] >> 32)) : "vcc"); v[q] = ((uint64_t)hi << 32) | lo; }
        if (KIND == 2) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(v[q]) : "v"((uint32_t)w[q]), "v"((uint32_t)(w[q] >> 32)) : "vcc");
        if (KIND == 3) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(v[q]));
      }
    }
  }
  uint64_t acc = 0;
  for (int q = 0; q < 8; ++q) acc ^= v[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int KIND>
void run(const char* name) {
  const int blocks = 256 * 8;
  uint64_t* out;
  (void)hipMalloc(&out, blocks * 256 * 8);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 12345ull);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 12345ull);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double slots = (double)blocks * 4 * ITER * 32;
  printf("%-36s %8.3f ms  %.2f cycles per 64-bit add per SIMD @2.4GHz\n", name, ms, 1024.0 * 2.4e9 / (slots / (ms * 1e-3)));
  (void)hipFree(out);
}
int main() {
  run<0>("v_lshl_add_u64 (shift 0)");
  run<1>("v_add_co_u32 + v_addc_co_u32");
  run<2>("v_mad_u64_u32");
  run<3>("v_lshlrev_b64");
  return 0;
}
