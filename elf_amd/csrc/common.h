// Shared host/device helpers for the elfhip library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/elfhip.h"

void elfhip_set_error(const char* fmt, ...);

#define HIPCHECK(x)                                                                         \
  do {                                                                                      \
    hipError_t e_ = (x);                                                                    \
    if (e_ != hipSuccess) {                                                                 \
      elfhip_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_));    \
      return ELFHIP_EHIP;                                                                   \
    }                                                                                       \
  } while (0)

#define ELF_GOLDEN 0x9E3779B97F4A7C15ull

// splitmix64 counter RNG of SURVEY.md 8(d)
__host__ __device__ inline uint64_t elf_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// r mod n for 1 <= n < 2^16, without the 64-bit division routine:
// r = hi*2^32 + lo  ->  ((hi mod n) * (2^32 mod n) + (lo mod n)) mod n, all in 32 bits.
__device__ inline uint32_t elf_mod_u64_small(uint64_t r, uint32_t n) {
  uint32_t hi = (uint32_t)(r >> 32), lo = (uint32_t)r;
  uint32_t p32 = ((0xFFFFFFFFu % n) + 1u) % n;  // 2^32 mod n in 32-bit arithmetic
  uint32_t a = hi % n, b = lo % n;
  return (a * p32 + b) % n;  // a*p32 < 2^32 because a,p32 < 2^16
}

#ifdef __HIPCC__
// The same r mod n through per-n reciprocals kept in LDS (the playout kernels draw one move per board and ply):
// tab[0][n] = floor((2^32 - 1) / n), tab[1][n] = 2^32 mod n for n = 1..NMAX; three multiply-high reductions
// instead of five 32-bit divisions.  elf_modtab_fill is called by every thread of the block (it synchronises).
template <int NMAX>
__device__ inline void elf_modtab_fill(uint32_t (&tab)[2][NMAX + 1]) {
  for (int n = threadIdx.x + 1; n <= NMAX; n += blockDim.x) {
    tab[0][n] = 0xFFFFFFFFu / (uint32_t)n;
    tab[1][n] = ((0xFFFFFFFFu % (uint32_t)n) + 1u) % (uint32_t)n;
  }
  __syncthreads();
}
// x mod n with rc = floor((2^32 - 1) / n): the estimate mulhi(x, rc) is the quotient or one less
__device__ inline uint32_t elf_mod_small(uint32_t x, uint32_t n, uint32_t rc) {
  const uint32_t r = x - __umulhi(x, rc) * n;
  return r >= n ? r - n : r;
}
template <int NMAX>
__device__ inline uint32_t elf_mod_u64_tab(uint64_t r, uint32_t n, const uint32_t (&tab)[2][NMAX + 1]) {
  const uint32_t rc = tab[0][n], p32 = tab[1][n];
  const uint32_t hi = elf_mod_small((uint32_t)(r >> 32), n, rc), lo = elf_mod_small((uint32_t)r, n, rc);
  return elf_mod_small(hi * p32 + lo, n, rc);   // hi * p32 + lo < 2^32: both factors are below n <= NMAX < 2^15
}
// position of the k-th (from 0) set bit, k < popcount(x): halving steps, no loop
__device__ inline uint32_t elf_kth_bit32(uint32_t x, int k) {
  uint32_t pos = 0;
#pragma unroll
  for (int w = 16; w >= 1; w >>= 1) {
    const int c = __popc(x & ((1u << w) - 1u));
    const bool hi = k >= c;
    k -= hi ? c : 0; x = hi ? x >> w : x; pos += hi ? (uint32_t)w : 0u;
  }
  return pos;
}
__device__ inline uint32_t elf_kth_bit64(uint64_t x, int k) {
  const uint32_t lo = (uint32_t)x;
  const int c = __popc(lo);
  const bool hi = k >= c;
  return elf_kth_bit32(hi ? (uint32_t)(x >> 32) : lo, hi ? k - c : k) + (hi ? 32u : 0u);
}
#endif

// Abstract slab of boards; one subclass per game (american.hip, go.hip, ...).
struct BoardSlab {
  int game = 0, n = 0, device = 0;
  hipStream_t stream = nullptr;
  uint64_t step_counter = 0;  // plies driven by playout() since reset (RNG counter)
  // raw device arrays other components read (american: [0]=words[6][n], [1]=hist[6][4][n], [2]=hmeta[n])
  void* dev_ptr[4] = {nullptr, nullptr, nullptr, nullptr};
  virtual ~BoardSlab() {}
  virtual int alloc() = 0;
  virtual int reset() = 0;
  virtual int reset_mask(const uint8_t* d_which) = 0;  // reset only boards with d_which[i] != 0
  virtual int export_states(uint64_t* h) = 0;
  // the export kernel alone: array-of-structs states into a DEVICE buffer of n*state_words u64 (enqueued, no sync)
  virtual int export_states_to(uint64_t* d_aos) = 0;
  virtual int import_states(const uint64_t* h) = 0;
  virtual int legal_mask(uint64_t* d_mask, uint8_t* d_terminal) = 0;
  virtual int step(const int16_t* d_actions, uint8_t* d_ok) = 0;
  virtual int features(void* d_out, int dtype) = 0;
  virtual int evaluate(float* d_value) = 0;
  virtual int playout(int nply, uint64_t seed, uint64_t first_id, unsigned long long* d_moves) = 0;
};

struct elfhip_boards { BoardSlab* slab; };

BoardSlab* elfhip_make_american();
BoardSlab* elfhip_make_go19();
BoardSlab* elfhip_make_russian();
BoardSlab* elfhip_make_ugolki();
