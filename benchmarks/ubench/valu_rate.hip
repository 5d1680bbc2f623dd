// Issue cost (cycles per wave instruction, one wave on its SIMD) of the integer VALU operations the Go playout kernel
// is made of.  hipcc --offload-arch=gfx950 -O2 valu_rate.hip -o valu_rate ; ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP4(...) __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__
#define REP16(...) REP4(__VA_ARGS__) REP4(__VA_ARGS__) REP4(__VA_ARGS__) REP4(__VA_ARGS__)
#define REP64(...) REP16(__VA_ARGS__) REP16(__VA_ARGS__) REP16(__VA_ARGS__) REP16(__VA_ARGS__)

#define KERNEL(name, ...)                                                                          \
  __global__ void name(uint32_t* out, unsigned long long* cyc, uint32_t seed) {                    \
    uint32_t a = seed + threadIdx.x, b = seed * 3 + threadIdx.x, c = seed * 5 + 1, d = seed * 7 + 2; \
    uint32_t m = 0x7FFFF;                                                                          \
    unsigned long long t0 = clock64();                                                             \
    for (int it = 0; it < 16; ++it) { REP64(__VA_ARGS__) }                                                \
    unsigned long long t1 = clock64();                                                             \
    out[threadIdx.x] = a ^ b ^ c ^ d;                                                              \
    if (threadIdx.x == 0) *cyc = t1 - t0;                                                          \
  }

// four independent chains per body = 4 instructions
KERNEL(k_and,   asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));)
KERNEL(k_add,   asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));)
KERNEL(k_brev,  asm volatile("v_bfrev_b32 %0, %0\n v_bfrev_b32 %1, %1\n v_bfrev_b32 %2, %2\n v_bfrev_b32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));)
KERNEL(k_bcnt,  asm volatile("v_bcnt_u32_b32 %0, %4, %0\n v_bcnt_u32_b32 %1, %4, %1\n v_bcnt_u32_b32 %2, %4, %2\n v_bcnt_u32_b32 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));)
KERNEL(k_lshlor, asm volatile("v_lshl_or_b32 %0, %0, 1, %4\n v_lshl_or_b32 %1, %1, 1, %4\n v_lshl_or_b32 %2, %2, 1, %4\n v_lshl_or_b32 %3, %3, 1, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));)
KERNEL(k_andor, asm volatile("v_and_or_b32 %0, %0, %4, %1\n v_and_or_b32 %1, %1, %4, %2\n v_and_or_b32 %2, %2, %4, %3\n v_and_or_b32 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));)
KERNEL(k_or3,   asm volatile("v_or3_b32 %0, %0, %4, %1\n v_or3_b32 %1, %1, %4, %2\n v_or3_b32 %2, %2, %4, %3\n v_or3_b32 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));)
KERNEL(k_bitop3, asm volatile("v_bitop3_b32 %0, %0, %4, %1 bitop3:0x6c\n v_bitop3_b32 %1, %1, %4, %2 bitop3:0x6c\n v_bitop3_b32 %2, %2, %4, %3 bitop3:0x6c\n v_bitop3_b32 %3, %3, %4, %0 bitop3:0x6c" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));)
KERNEL(k_xor,   asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));)
KERNEL(k_lshr,  asm volatile("v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));)
KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc");)
KERNEL(k_cmp,   asm volatile("v_cmp_ne_u32 vcc, %0, %4\n v_cmp_ne_u32 vcc, %1, %4\n v_cmp_ne_u32 vcc, %2, %4\n v_cmp_ne_u32 vcc, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc");)
KERNEL(k_mulhi, asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));)
KERNEL(k_lshl64, { uint64_t x = ((uint64_t)a << 32) | b, y = ((uint64_t)c << 32) | d; asm volatile("v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1" : "+v"(x), "+v"(y)); a = x >> 32; b = (uint32_t)x; c = y >> 32; d = (uint32_t)y; })
KERNEL(k_ffbl,  asm volatile("v_ffbl_b32 %0, %0\n v_ffbl_b32 %1, %1\n v_ffbl_b32 %2, %2\n v_ffbl_b32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));)
// dependent chain (each instruction reads the previous result)
KERNEL(k_and_dep, asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %0, %0, %4\n v_and_b32 %0, %0, %4\n v_and_b32 %0, %0, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));)
KERNEL(k_brev_dep, asm volatile("v_bfrev_b32 %0, %0\n v_bfrev_b32 %0, %0\n v_bfrev_b32 %0, %0\n v_bfrev_b32 %0, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));)
KERNEL(k_addco_dep, asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc");)
KERNEL(k_cmp_cnd, asm volatile("v_cmp_ne_u32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_ne_u32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc");)

template <class K> void run(const char* name, K k, int waves_per_block) {
  uint32_t* out; unsigned long long* cyc;
  hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 8);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves_per_block), 0, 0, out, cyc, 12345u);
  hipDeviceSynchronize();
  unsigned long long h = 0; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-12s waves/block %d: %.2f cycles per instruction\n", name, waves_per_block, (double)h / (16.0 * 64 * 4));
  hipFree(out); hipFree(cyc);
}
#define RUN(k) run(#k, k, 1); run(#k, k, 4); run(#k, k, 8); run(#k, k, 16);
int main() {
  RUN(k_and) RUN(k_brev) RUN(k_bcnt) RUN(k_mulhi) RUN(k_and_dep)
  return 0;
}
