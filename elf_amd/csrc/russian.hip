// Russian checkers (flying kings, forced capture, forced continuation) on gfx950.
// One lane owns one board; the state is 5 x u64 in registers (white pawns, white kings, black
// pawns, black kings, meta), struct-of-arrays in HBM.
//
// Behaviour follows /root/reference/src_cpp/elfgames/russian_checkers/game/CheckersBoard.cc
// (cited per function) but the formulation is different: the reference walks an int[8][8] array
// square by square and resolves every move through std::to_string + std::map (CheckersBoard.cc:85-101).
// Here every diagonal ray is a Kogge-Stone occluded fill on a 64-bit board (3 shift/and/or rounds
// cover the 7-square diagonal), the "a further capture is possible from this landing square" test
// of the flying king is four reverse fills from the capturable enemy men, and an action index is
// base[dir][dist] + popcount(valid from-squares above the origin) - no tables, no strings.
#include "common.h"

#include "russian_dev.h"

namespace {
using namespace rudev;

// ---- kernels ------------------------------------------------------------------------------------
__global__ void ru_reset_kernel(uint64_t* w, int n, const uint8_t* __restrict__ which) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n || (which && !which[b])) return;
  Ru s; ru_clear(s);
  ru_store(w, n, b, s);
}

__global__ void ru_import_kernel(uint64_t* w, const uint64_t* aos, int n) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  Ru s{aos[b * 5 + 0], aos[b * 5 + 1], aos[b * 5 + 2], aos[b * 5 + 3], aos[b * 5 + 4]};
  ru_store(w, n, b, s);
}

__global__ void ru_export_kernel(const uint64_t* w, uint64_t* aos, int n) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  Ru s; ru_load(w, n, b, s);
  aos[b * 5 + 0] = s.wp; aos[b * 5 + 1] = s.wk; aos[b * 5 + 2] = s.bp; aos[b * 5 + 3] = s.bk; aos[b * 5 + 4] = s.w4;
}

// GetValidMovesBinary :85-101 + terminated() CheckersState.h:51-53
__global__ void __launch_bounds__(256)
ru_mask_kernel(const uint64_t* __restrict__ w, uint64_t* __restrict__ mask, uint8_t* __restrict__ term, int n) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  Ru s; ru_load(w, n, b, s);
  uint64_t m[5];
  ru_moves(s, m);
#pragma unroll
  for (int i = 0; i < 5; ++i) mask[(size_t)b * 5 + i] = m[i];
  if (term) term[b] = (ru_count(m) == 0 || ru_ply(s.w4) >= RU_MAX_PLY) ? 1 : 0;
}

// CheckersState::forward CheckersState.cc:4-19
__global__ void __launch_bounds__(256)
ru_step_kernel(uint64_t* __restrict__ w, const int16_t* __restrict__ actions, uint8_t* __restrict__ ok, int n) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  int action = actions[b];
  bool good = action >= 0 && action < RU_ACTIONS;
  if (good) {
    Ru s; ru_load(w, n, b, s);
    uint64_t m[5];
    ru_moves(s, m);
    uint64_t word = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) word = (i == (action >> 6)) ? m[i] : word;
    good = ru_ply(s.w4) < RU_MAX_PLY && ((word >> (action & 63)) & 1);
    if (good) { ru_play(s, action); ru_store(w, n, b, s); }
  }
  if (ok) ok[b] = good ? 1 : 0;
}

// evaluateGame CheckersState.h:52-64
__global__ void __launch_bounds__(256)
ru_evaluate_kernel(const uint64_t* __restrict__ w, float* __restrict__ v, int n) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  Ru s; ru_load(w, n, b, s);
  uint64_t m[5];
  ru_moves(s, m);
  float r = 0.f;
  if (ru_ply(s.w4) >= RU_MAX_PLY) r = -1.f;
  else if (ru_count(m) == 0) r = ru_white(s.w4) ? 1.f : -1.f;
  v[b] = r;
}

__global__ void __launch_bounds__(256)
ru_playout_kernel(uint64_t* __restrict__ w, int n, int nply, uint64_t seed, uint64_t first_id, uint64_t step_base,
                  unsigned long long* __restrict__ moves_out) {
  __shared__ uint32_t s_mod[2][128 + 1];
  __shared__ uint2 s_act[RU_ACTIONS];
  for (int i = threadIdx.x; i < RU_ACTIONS; i += blockDim.x) s_act[i] = c_ru_act.e[i];
  elf_modtab_fill<128>(s_mod);
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long moves = 0;
#ifdef RU_PROF
  for (int k = 0; k < 16; ++k) s_ru_prof[k][threadIdx.x] = 0;
#endif
  if (b < n) {
    Ru s64; ru_load(w, n, b, s64);
    R32 s = ru_to32(s64);   // the rules run on the 32 dark squares; the stored layout stays the reference-facing one
    const uint64_t key = seed + (first_id + (uint64_t)b) * ELF_GOLDEN + step_base;
    for (int step = 0; step < nply; ++step) {
      uint64_t m[5];
      RPROF_T0;
      ru_moves(s, m);
      RPROF_ADD(5);
      int total = ru_count(m);
      if (total == 0 || ru_ply(s.w4) >= RU_MAX_PLY) { ru_clear(s); continue; }
      uint64_t r = elf_splitmix64(key + (uint64_t)step);
      int kk = (int)(total <= 128 ? elf_mod_u64_tab<128>(r, (uint32_t)total, s_mod) : elf_mod_u64_small(r, (uint32_t)total));
      const int a = ru_kth(m, kk);
      RPROF_ADD(6);
      ru_play(s, a, s_act);
      RPROF_ADD(7);
      ++moves;
    }
    ru_store(w, n, b, ru_to64(s));
  }
#ifdef RU_PROF
  for (int k = 0; k < 16; ++k) {
    unsigned long long v = s_ru_prof[k][threadIdx.x];
    for (int o = 32; o > 0; o >>= 1) { unsigned long long u = __shfl_xor(v, o); v = u > v ? u : v; }
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_ru_prof[k], v);
  }
#endif
  if (moves_out) {
    for (int o = 32; o > 0; o >>= 1) moves += __shfl_down(moves, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(moves_out, moves);
  }
}

// CheckersFeature::extract CheckersFeature.cc:57-82 over GetObservation CheckersBoard.cc:123-136.
// One wave per board, lane = square.  The observation of the white player is the board rotated by
// 180 degrees and negated; the black player's is the raw board - so with black to move the "own"
// planes 0/1 hold WHITE pieces.  That is what the reference's nets are trained on; reproduced.
template <int DTYPE>
__global__ void __launch_bounds__(256)
ru_features_kernel(const uint64_t* __restrict__ w, int n, void* __restrict__ out) {
  int b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (b >= n) return;
  Ru s; ru_load(w, n, b, s);
  const bool white = ru_white(s.w4);
  const int rot = 63 - lane;
  bool pl[6];
  // active player's observation == 1 / == 3
  pl[0] = white ? (s.bp >> rot) & 1 : (s.wp >> lane) & 1;
  pl[1] = white ? (s.bk >> rot) & 1 : (s.wk >> lane) & 1;
  // passive player's observation == 1 / == 3
  pl[2] = white ? (s.wp >> lane) & 1 : (s.bp >> rot) & 1;
  pl[3] = white ? (s.wk >> lane) & 1 : (s.bk >> rot) & 1;
  pl[4] = !white;
  pl[5] = white;
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    if (DTYPE == ELFHIP_F32) ((float*)out)[((size_t)b * 6 + q) * 64 + lane] = pl[q] ? 1.f : 0.f;
    else if (DTYPE == ELFHIP_BF16) ((uint16_t*)out)[((size_t)b * 6 + q) * 64 + lane] = pl[q] ? 0x3F80 : 0;
    else {
      uint64_t bits = __ballot(pl[q]);
      if (lane == 0) ((uint64_t*)out)[(size_t)b * 6 + q] = bits;
    }
  }
}

struct RussianSlab : BoardSlab {
  uint64_t* w = nullptr;    // [5][n]
  uint64_t* aos = nullptr;  // [n][5] staging for export/import
  ~RussianSlab() override { (void)hipFree(w); (void)hipFree(aos); }
  int grid() const { return (n + 255) / 256; }
  int alloc() override {
    if (hipMalloc(&w, sizeof(uint64_t) * 5 * (size_t)n) != hipSuccess ||
        hipMalloc(&aos, sizeof(uint64_t) * 5 * (size_t)n) != hipSuccess) {
      elfhip_set_error("russian: device allocation of %d boards failed", n);
      return ELFHIP_ENOMEM;
    }
    dev_ptr[0] = w;
    return reset();
  }
  int reset() override {
    step_counter = 0;
    hipLaunchKernelGGL(ru_reset_kernel, dim3(grid()), dim3(256), 0, stream, w, n, (const uint8_t*)nullptr);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int reset_mask(const uint8_t* d_which) override {
    hipLaunchKernelGGL(ru_reset_kernel, dim3(grid()), dim3(256), 0, stream, w, n, d_which);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int export_states_to(uint64_t* d_aos) override {
    hipLaunchKernelGGL(ru_export_kernel, dim3(grid()), dim3(256), 0, stream, w, d_aos, n);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int export_states(uint64_t* h) override {
    if (int rc = export_states_to(aos)) return rc;
    HIPCHECK(hipMemcpyAsync(h, aos, sizeof(uint64_t) * 5 * (size_t)n, hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipStreamSynchronize(stream));
    return ELFHIP_OK;
  }
  int import_states(const uint64_t* h) override {
    HIPCHECK(hipMemcpyAsync(aos, h, sizeof(uint64_t) * 5 * (size_t)n, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(ru_import_kernel, dim3(grid()), dim3(256), 0, stream, w, aos, n);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(stream));
    return ELFHIP_OK;
  }
  int legal_mask(uint64_t* d_mask, uint8_t* d_terminal) override {
    hipLaunchKernelGGL(ru_mask_kernel, dim3(grid()), dim3(256), 0, stream, w, d_mask, d_terminal, n);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int step(const int16_t* d_actions, uint8_t* d_ok) override {
    hipLaunchKernelGGL(ru_step_kernel, dim3(grid()), dim3(256), 0, stream, w, d_actions, d_ok, n);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int features(void* d_out, int dtype) override {
    int blocks = (n + 3) / 4;
    if (dtype == ELFHIP_F32)
      hipLaunchKernelGGL(ru_features_kernel<ELFHIP_F32>, dim3(blocks), dim3(256), 0, stream, w, n, d_out);
    else if (dtype == ELFHIP_BF16)
      hipLaunchKernelGGL(ru_features_kernel<ELFHIP_BF16>, dim3(blocks), dim3(256), 0, stream, w, n, d_out);
    else if (dtype == ELFHIP_BITS)
      hipLaunchKernelGGL(ru_features_kernel<ELFHIP_BITS>, dim3(blocks), dim3(256), 0, stream, w, n, d_out);
    else { elfhip_set_error("russian features: unknown dtype %d", dtype); return ELFHIP_EINVAL; }
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int evaluate(float* d_value) override {
    hipLaunchKernelGGL(ru_evaluate_kernel, dim3(grid()), dim3(256), 0, stream, w, d_value, n);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int playout(int nply, uint64_t seed, uint64_t first_id, unsigned long long* d_moves) override {
    hipLaunchKernelGGL(ru_playout_kernel, dim3(grid()), dim3(256), 0, stream, w, n, nply, seed, first_id,
                       step_counter, d_moves);
    HIPCHECK(hipGetLastError());
    step_counter += (uint64_t)nply;
    return ELFHIP_OK;
  }
};

}  // namespace

BoardSlab* elfhip_make_russian() { return new RussianSlab(); }

#ifdef RU_PROF
extern "C" int elfhip_debug_ru_prof(unsigned long long* h_out16, int reset) {
  if (h_out16 && hipMemcpyFromSymbol(h_out16, HIP_SYMBOL(rudev::g_ru_prof), 16 * 8) != hipSuccess) return -2;
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rudev::g_ru_prof), z, 16 * 8) != hipSuccess) return -2; }
  return 0;
}
#endif
