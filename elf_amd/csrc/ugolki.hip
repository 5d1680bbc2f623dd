// Ugolki ("corners": 9 men a side, orthogonal steps and chained orthogonal jumps) on gfx950.
// One lane owns one board; the state is 4 x u64 in registers (black men, white men, jump_action,
// meta), struct-of-arrays in HBM.
//
// Behaviour follows /root/reference/src_cpp/elfgames/ugolki/game/GameBoard.cc (cited per function).
// The reference builds a std::vector of {bits, flags} moves and resolves each through a
// std::stringstream + std::map<string,int> (GameBoard.cc:113-129).  Here the eight move families
// stay bitboards indexed by the LOW square of the two-square pattern; the reference's action table
// (HashAllMoves.h) is exactly "family-major, low square ascending", so the 417-bit legal mask is
// eight row-compressed bitboards packed end to end plus the pass bit - no loops over moves at all.
#include "common.h"

#include "ugolki_dev.h"

namespace {
using namespace ugdev;

__global__ void ug_reset_kernel(uint64_t* w, int n, const uint8_t* __restrict__ which) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n || (which && !which[b])) return;
  Ug s; ug_clear(s);
  ug_store(w, n, b, s);
}

__global__ void ug_import_kernel(uint64_t* w, const uint64_t* aos, int n) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  Ug s; s.p[0] = aos[b * 4 + 0]; s.p[1] = aos[b * 4 + 1]; s.ja = aos[b * 4 + 2]; s.meta = aos[b * 4 + 3];
  ug_store(w, n, b, s);
}

__global__ void ug_export_kernel(const uint64_t* w, uint64_t* aos, int n) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  Ug s; ug_load(w, n, b, s);
  aos[b * 4 + 0] = s.p[0]; aos[b * 4 + 1] = s.p[1]; aos[b * 4 + 2] = s.ja; aos[b * 4 + 3] = s.meta;
}

// GetValidMovesBinary :113-129 + terminated() GameState.h:49-51
__global__ void __launch_bounds__(256)
ug_mask_kernel(const uint64_t* __restrict__ w, uint64_t* __restrict__ mask, uint8_t* __restrict__ term, int n) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  Ug s; ug_load(w, n, b, s);
  uint64_t fam[8], m[7];
  bool pass = ug_families(s, fam);
  ug_mask_words(fam, pass, m);
#pragma unroll
  for (int i = 0; i < 7; ++i) mask[(size_t)b * 7 + i] = m[i];
  if (term) term[b] = (ug_over(s) || ug_ply(s) >= UG_MAX_PLY) ? 1 : 0;
}

// GameState::forward GameState.cc:4-18.  TryPlay (:132-213) accepts exactly the members of the legal
// mask on reachable states (tests/test_oracle_ugolki.py pins that) plus "pass whenever a chain runs".
__global__ void __launch_bounds__(256)
ug_step_kernel(uint64_t* __restrict__ w, const int16_t* __restrict__ actions, uint8_t* __restrict__ ok, int n) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  int action = actions[b];
  bool good = action >= 0 && action <= UG_PASS;
  if (good) {
    Ug s; ug_load(w, n, b, s);
    good = !ug_over(s) && ug_ply(s) < UG_MAX_PLY;
    if (good) {
      int g = -1, i = 0;
      if (action == UG_PASS) good = s.ja != 0;
      else {
        uint64_t fam[8];
        ug_families(s, fam);
        ug_decode(action, g, i);
        uint64_t sel = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) sel = (q == g) ? fam[q] : sel;
        good = (sel >> i) & 1;
      }
      if (good) { ug_play(s, g, i); ug_store(w, n, b, s); }
    }
  }
  if (ok) ok[b] = good ? 1 : 0;
}

// evaluateGame GameState.cc:56-70
__global__ void ug_evaluate_kernel(const uint64_t* __restrict__ w, float* __restrict__ v, int n) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  Ug s; ug_load(w, n, b, s);
  float r = 0.f;
  if (ug_ply(s) >= UG_MAX_PLY) r = -1.f;
  else if (ug_over(s)) r = (ug_bwin(s) > 0 && ug_wwin(s) > 0) ? -1.f : (ug_bwin(s) == 2 ? 1.f : -1.f);
  v[b] = r;
}

__global__ void __launch_bounds__(256)
ug_playout_kernel(uint64_t* __restrict__ w, int n, int nply, uint64_t seed, uint64_t first_id, uint64_t step_base,
                  unsigned long long* __restrict__ moves_out) {
  __shared__ uint32_t s_mod[2][128 + 1];
  elf_modtab_fill<128>(s_mod);
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long moves = 0;
  if (b < n) {
    Ug s; ug_load(w, n, b, s);
    const uint64_t key = seed + (first_id + (uint64_t)b) * ELF_GOLDEN + step_base;
    for (int step = 0; step < nply; ++step) {
      uint64_t fam[8];
      bool pass = ug_families(s, fam);
      int total = ug_count(fam, pass);
      if (total == 0 || ug_over(s) || ug_ply(s) >= UG_MAX_PLY) { ug_clear(s); continue; }
      uint64_t r = elf_splitmix64(key + (uint64_t)step);
      int kk = (int)(total <= 128 ? elf_mod_u64_tab<128>(r, (uint32_t)total, s_mod) : elf_mod_u64_small(r, (uint32_t)total));
      // k-th legal move in ascending action index == k-th set bit over fam[0..7], then pass
      int g = -1; uint64_t sel = 0; bool found = false;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        int c = __popcll(fam[q]);
        bool here = !found && kk < c;
        if (here) { g = q; sel = fam[q]; found = true; }
        if (!found) kk -= c;
      }
      int i = 0;
      if (found) {
        i = (int)elf_kth_bit64(sel, kk);
      }
      ug_play(s, g, i);
      ++moves;
    }
    ug_store(w, n, b, s);
  }
  if (moves_out) {
    for (int o = 32; o > 0; o >>= 1) moves += __shfl_down(moves, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(moves_out, moves);
  }
}

// BoardFeature::extract BoardFeature.cc:45-63: plane 0 = the mover's men in the mover's view, plane 1 =
// the other side's men in THEIR view (white's view is rotated by 180 degrees, GetObservation :229-269),
// planes 2/3 = black/white to move, planes 4..15 stay zero (NUM_FEATURES = 16).
template <int DTYPE>
__global__ void __launch_bounds__(256)
ug_features_kernel(const uint64_t* __restrict__ w, int n, void* __restrict__ out) {
  int b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (b >= n) return;
  Ug s; ug_load(w, n, b, s);
  const bool white = ug_active(s) == 1;
  const bool blk = (s.p[0] >> lane) & 1, wht = (s.p[1] >> (63 - lane)) & 1;
#pragma unroll
  for (int q = 0; q < UG_PLANES; ++q) {
    bool v = q == 0 ? (white ? wht : blk) : q == 1 ? (white ? blk : wht) : q == 2 ? !white : q == 3 ? white : false;
    if (DTYPE == ELFHIP_F32) ((float*)out)[((size_t)b * UG_PLANES + q) * 64 + lane] = v ? 1.f : 0.f;
    else if (DTYPE == ELFHIP_BF16) ((uint16_t*)out)[((size_t)b * UG_PLANES + q) * 64 + lane] = v ? 0x3F80 : 0;
    else {
      uint64_t bits = __ballot(v);
      if (lane == 0) ((uint64_t*)out)[(size_t)b * UG_PLANES + q] = bits;
    }
  }
}

struct UgolkiSlab : BoardSlab {
  uint64_t* w = nullptr;    // [4][n]
  uint64_t* aos = nullptr;  // [n][4] staging for export/import
  ~UgolkiSlab() override { (void)hipFree(w); (void)hipFree(aos); }
  int grid() const { return (n + 255) / 256; }
  int alloc() override {
    if (hipMalloc(&w, sizeof(uint64_t) * 4 * (size_t)n) != hipSuccess ||
        hipMalloc(&aos, sizeof(uint64_t) * 4 * (size_t)n) != hipSuccess) {
      elfhip_set_error("ugolki: device allocation of %d boards failed", n);
      return ELFHIP_ENOMEM;
    }
    dev_ptr[0] = w;
    return reset();
  }
  int reset() override {
    step_counter = 0;
    hipLaunchKernelGGL(ug_reset_kernel, dim3(grid()), dim3(256), 0, stream, w, n, (const uint8_t*)nullptr);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int reset_mask(const uint8_t* d_which) override {
    hipLaunchKernelGGL(ug_reset_kernel, dim3(grid()), dim3(256), 0, stream, w, n, d_which);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int export_states_to(uint64_t* d_aos) override {
    hipLaunchKernelGGL(ug_export_kernel, dim3(grid()), dim3(256), 0, stream, w, d_aos, n);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int export_states(uint64_t* h) override {
    if (int rc = export_states_to(aos)) return rc;
    HIPCHECK(hipMemcpyAsync(h, aos, sizeof(uint64_t) * 4 * (size_t)n, hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipStreamSynchronize(stream));
    return ELFHIP_OK;
  }
  int import_states(const uint64_t* h) override {
    HIPCHECK(hipMemcpyAsync(aos, h, sizeof(uint64_t) * 4 * (size_t)n, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(ug_import_kernel, dim3(grid()), dim3(256), 0, stream, w, aos, n);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(stream));
    return ELFHIP_OK;
  }
  int legal_mask(uint64_t* d_mask, uint8_t* d_terminal) override {
    hipLaunchKernelGGL(ug_mask_kernel, dim3(grid()), dim3(256), 0, stream, w, d_mask, d_terminal, n);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int step(const int16_t* d_actions, uint8_t* d_ok) override {
    hipLaunchKernelGGL(ug_step_kernel, dim3(grid()), dim3(256), 0, stream, w, d_actions, d_ok, n);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int features(void* d_out, int dtype) override {
    int blocks = (n + 3) / 4;
    if (dtype == ELFHIP_F32)
      hipLaunchKernelGGL(ug_features_kernel<ELFHIP_F32>, dim3(blocks), dim3(256), 0, stream, w, n, d_out);
    else if (dtype == ELFHIP_BF16)
      hipLaunchKernelGGL(ug_features_kernel<ELFHIP_BF16>, dim3(blocks), dim3(256), 0, stream, w, n, d_out);
    else if (dtype == ELFHIP_BITS)
      hipLaunchKernelGGL(ug_features_kernel<ELFHIP_BITS>, dim3(blocks), dim3(256), 0, stream, w, n, d_out);
    else { elfhip_set_error("ugolki features: unknown dtype %d", dtype); return ELFHIP_EINVAL; }
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int evaluate(float* d_value) override {
    hipLaunchKernelGGL(ug_evaluate_kernel, dim3(grid()), dim3(256), 0, stream, w, d_value, n);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int playout(int nply, uint64_t seed, uint64_t first_id, unsigned long long* d_moves) override {
    hipLaunchKernelGGL(ug_playout_kernel, dim3(grid()), dim3(256), 0, stream, w, n, nply, seed, first_id,
                       step_counter, d_moves);
    HIPCHECK(hipGetLastError());
    step_counter += (uint64_t)nply;
    return ELFHIP_OK;
  }
};

}  // namespace

BoardSlab* elfhip_make_ugolki() { return new UgolkiSlab(); }
