// VALU throughput of the whole chip against waves per SIMD (wall clock): does one wave per SIMD reach the SIMD's issue rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP4(...) __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__
#define REP16(...) REP4(__VA_ARGS__) REP4(__VA_ARGS__) REP4(__VA_ARGS__) REP4(__VA_ARGS__)
#define REP64(...) REP16(__VA_ARGS__) REP16(__VA_ARGS__) REP16(__VA_ARGS__) REP16(__VA_ARGS__)
__global__ void k_and(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a = seed + threadIdx.x, b = seed * 3 + threadIdx.x, c = seed * 5 + 1, d = seed * 7 + 2, m = 0x7FFFF;
  for (int it = 0; it < iters; ++it) {
    REP64(asm volatile("v_and_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_bfrev_b32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}
__global__ void k_dep(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a = seed + threadIdx.x, m = 0x7FFFF;
  for (int it = 0; it < iters; ++it) {
    REP64(asm volatile("v_and_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_bfrev_b32 %0, %0" : "+v"(a) : "v"(m));)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
template <class K> void run(const char* name, K k, int blocks, int threads) {
  uint32_t* out; (void)hipMalloc(&out, (size_t)blocks * threads * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2000;
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, iters, 1u);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, iters, 1u);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)blocks * threads / 64, inst = waves * iters * 256.0;
  printf("%-6s %5d blocks x %4d threads (%.1f waves/SIMD): %.3f ms, %.2f ns per wave-instruction per SIMD -> %.2f G wave-inst/s\n", name, blocks, threads,
         waves / 1024, ms, ms * 1e6 / (inst / 1024), inst / ms / 1e6);
  (void)hipFree(out);
}
int main() {
  run("indep", k_and, 256, 256); run("indep", k_and, 256, 512); run("indep", k_and, 256, 1024); run("indep", k_and, 512, 1024);
  run("dep", k_dep, 256, 256); run("dep", k_dep, 256, 512); run("dep", k_dep, 256, 1024); run("dep", k_dep, 512, 1024);
  return 0;
}
