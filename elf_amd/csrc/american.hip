// American checkers on gfx950: one lane owns one board, the whole state lives in registers
// (6 x u64), boards are struct-of-arrays in HBM so every load/store is a coalesced 8 B/lane access.
//
// Behaviour follows the reference's GameBoard.cc / GameState.cc / BoardFeature.cc
// (/root/reference/src_cpp/elfgames/american_checkers/game, cited per function) but the
// formulation is different: the reference emits std::vector<int64_t> move words and resolves each
// through std::to_string + std::map<string,int> (GameBoard.cc:143-146).  Here the eight move
// families stay bitboards; an action index is base[family] + popcount(valid_shifts below bit), so
// mask building, k-th-legal selection and play are branch-free bit arithmetic.
#include "common.h"
#include "american_dev.h"

namespace {
using namespace amdev;

// ---- SoA access ---------------------------------------------------------------------------------
__device__ inline void am_load(const uint64_t* __restrict__ w, int n, int b, Am& s) {
  s.f[0] = w[0 * (size_t)n + b]; s.f[1] = w[1 * (size_t)n + b];
  s.k[0] = w[2 * (size_t)n + b]; s.k[1] = w[3 * (size_t)n + b];
  s.w4 = w[4 * (size_t)n + b]; s.w5 = w[5 * (size_t)n + b];
}
__device__ inline void am_store(uint64_t* __restrict__ w, int n, int b, const Am& s) {
  w[0 * (size_t)n + b] = s.f[0]; w[1 * (size_t)n + b] = s.f[1];
  w[2 * (size_t)n + b] = s.k[0]; w[3 * (size_t)n + b] = s.k[1];
  w[4 * (size_t)n + b] = s.w4; w[5 * (size_t)n + b] = s.w5;
}

// history ring (GameState::_history, GameState.cc:14-16): 6 slots x {f0,f1,k0,k1} SoA,
// hmeta = len | head<<3 | active-bit of slot j at bit 8+j
__device__ inline void am_hist_set_single(uint64_t* hist, uint32_t* hmeta, int n, int b, const Am& s) {
  hist[(0 * 4 + 0) * (size_t)n + b] = s.f[0]; hist[(0 * 4 + 1) * (size_t)n + b] = s.f[1];
  hist[(0 * 4 + 2) * (size_t)n + b] = s.k[0]; hist[(0 * 4 + 3) * (size_t)n + b] = s.k[1];
  hmeta[b] = 1u | (0u << 3) | ((uint32_t)am_active(s) << 8);
}
__device__ inline void am_hist_push(uint64_t* hist, uint32_t* hmeta, int n, int b, const Am& s) {
  uint32_t m = hmeta[b];
  int len = m & 7, head = (m >> 3) & 7;
  int slot;
  if (len < 6) { slot = (head + len) % 6; ++len; }
  else { slot = head; head = (head + 1) % 6; }
  hist[(slot * 4 + 0) * (size_t)n + b] = s.f[0]; hist[(slot * 4 + 1) * (size_t)n + b] = s.f[1];
  hist[(slot * 4 + 2) * (size_t)n + b] = s.k[0]; hist[(slot * 4 + 3) * (size_t)n + b] = s.k[1];
  uint32_t abits = (m >> 8) & 0x3F;
  abits = (abits & ~(1u << slot)) | ((uint32_t)am_active(s) << slot);
  hmeta[b] = (uint32_t)len | ((uint32_t)head << 3) | (abits << 8);
}

// ---- kernels ------------------------------------------------------------------------------------
__global__ void am_reset_kernel(uint64_t* w, uint64_t* hist, uint32_t* hmeta, int n, const uint8_t* __restrict__ which) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n || (which && !which[b])) return;
  Am s; am_clear(s);
  am_store(w, n, b, s);
  am_hist_set_single(hist, hmeta, n, b, s);
}

__global__ void am_import_kernel(uint64_t* w, uint64_t* hist, uint32_t* hmeta, const uint64_t* aos, int n) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  Am s;
  s.f[0] = aos[b * 6 + 0]; s.f[1] = aos[b * 6 + 1]; s.k[0] = aos[b * 6 + 2]; s.k[1] = aos[b * 6 + 3];
  s.w4 = aos[b * 6 + 4]; s.w5 = aos[b * 6 + 5];
  am_store(w, n, b, s);
  am_hist_set_single(hist, hmeta, n, b, s);
}

__global__ void am_export_kernel(const uint64_t* w, uint64_t* aos, int n) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  Am s; am_load(w, n, b, s);
  aos[b * 6 + 0] = s.f[0]; aos[b * 6 + 1] = s.f[1]; aos[b * 6 + 2] = s.k[0]; aos[b * 6 + 3] = s.k[1];
  aos[b * 6 + 4] = s.w4; aos[b * 6 + 5] = s.w5;
}

__global__ void am_mask_kernel(const uint64_t* __restrict__ w, uint64_t* __restrict__ mask,
                               uint8_t* __restrict__ term, int n) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  Am s; am_load(w, n, b, s);
  uint64_t fam[8];
  am_families(s, fam);
  int total = am_count(fam);
  bool over = total == 0;
  am_apply_repeat_rule(s, fam, total);
  uint64_t m[3];
  am_mask_words(fam, m);
  mask[(size_t)b * 3 + 0] = m[0]; mask[(size_t)b * 3 + 1] = m[1]; mask[(size_t)b * 3 + 2] = m[2];
  if (term) term[b] = (over || am_ply(s) >= AM_MAX_PLY) ? 1 : 0;
}

// GameState::forward GameState.cc:4-18
__global__ void am_step_kernel(uint64_t* __restrict__ w, uint64_t* __restrict__ hist, uint32_t* __restrict__ hmeta,
                               const int16_t* __restrict__ actions, uint8_t* __restrict__ ok, int n) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  int action = actions[b];
  if (action < 0) { if (ok) ok[b] = 0; return; }
  Am s; am_load(w, n, b, s);
  uint64_t fam[8];
  am_families(s, fam);
  int total = am_count(fam);
  bool good = total != 0 && am_ply(s) < AM_MAX_PLY && action < AM_ACTIONS;
  int t = 0, i = 0;
  if (good) {
    am_apply_repeat_rule(s, fam, total);
    t = c_acttab.fam[action]; i = c_acttab.shift[action];
    uint64_t sel = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) sel = (q == t) ? fam[q] : sel;
    good = (sel >> i) & 1;
  }
  if (good) {
    am_play(s, action, t, i);
    am_store(w, n, b, s);
    am_hist_push(hist, hmeta, n, b, s);
  }
  if (ok) ok[b] = good ? 1 : 0;
}

__global__ void am_evaluate_kernel(const uint64_t* __restrict__ w, float* __restrict__ v, int n) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  Am s; am_load(w, n, b, s);
  uint64_t fam[8];
  am_families(s, fam);
  bool over = am_count(fam) == 0;
  float r = 0.f;  // GameState.cc:65-77
  if (am_ply(s) >= AM_MAX_PLY) r = -1.f;
  else if (over) r = am_active(s) == 0 ? -1.f : 1.f;
  v[b] = r;
}

// Random-legal playouts, `nply` plies per launch, state in registers.
__global__ void __launch_bounds__(256)
am_playout_kernel(uint64_t* __restrict__ w, uint64_t* __restrict__ hist, uint32_t* __restrict__ hmeta, int n,
                  int nply, uint64_t seed, uint64_t first_id, uint64_t step_base,
                  unsigned long long* __restrict__ moves_out) {
  __shared__ uint32_t s_mod[2][64 + 1];   // legal moves of a position: at most 12 pieces x 4
  elf_modtab_fill<64>(s_mod);
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long moves = 0;
  if (b < n) {
    Am s; am_load(w, n, b, s);
    const uint64_t key = seed + (first_id + (uint64_t)b) * ELF_GOLDEN + step_base;
    for (int step = 0; step < nply; ++step) {
      uint64_t fam[8];
      am_families(s, fam);
      int total = am_count(fam);
      if (total == 0 || am_ply(s) >= AM_MAX_PLY) { am_clear(s); continue; }
      total = am_apply_repeat_rule(s, fam, total);
      uint64_t r = elf_splitmix64(key + (uint64_t)step);
      int kk = (int)(total <= 64 ? elf_mod_u64_tab<64>(r, (uint32_t)total, s_mod) : elf_mod_u64_small(r, (uint32_t)total));
      // k-th legal move in ascending action index == k-th set bit over fam[0..7]
      int t = 0; uint64_t sel = 0; bool found = false;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        int c = __popcll(fam[q]);
        bool here = !found && kk < c;
        if (here) { t = q; sel = fam[q]; found = true; }
        if (!found) kk -= c;
      }
      int i = (int)elf_kth_bit64(sel, kk);
      uint64_t valid = (t & 1) ? ((t >= 4) ? V10 : V5) : ((t >= 4) ? V8 : V4);
      int base = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) base = (q == t) ? fam_info(q).base : base;
      int action = base + __popcll(valid & ((1ull << i) - 1));
      am_play(s, action, t, i);
      ++moves;
    }
    am_store(w, n, b, s);
    am_hist_set_single(hist, hmeta, n, b, s);  // playouts do not keep the 6-deep history
  }
  if (moves_out) {
    // one atomic per wave
    for (int o = 32; o > 0; o >>= 1) moves += __shfl_down(moves, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(moves_out, moves);
  }
}

// BoardFeature::extract BoardFeature.cc:47-77 + GetObservation GameBoard.cc:210-277.
// One wave per board, lane = square (y*8+x); each plane is one ballot / one coalesced 256 B store.
template <int DTYPE>
__global__ void __launch_bounds__(256)
am_features_kernel(const uint64_t* __restrict__ hist, const uint32_t* __restrict__ hmeta, int n, void* __restrict__ out) {
  int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (wave >= n) return;
  int b = wave;
  uint32_t m = hmeta[b];
  int len = m & 7, head = (m >> 3) & 7;
  uint32_t abits = (m >> 8) & 0x3F;
  for (int slot = 0; slot < 6; ++slot) {
    int k = slot - (6 - len);  // history index, oldest first; negative -> zero padding
    float pl[6] = {0, 0, 0, 0, 0, 0};
    if (k >= 0) {
      int hs = (head + k) % 6;
      uint64_t f0 = hist[(hs * 4 + 0) * (size_t)n + b], f1 = hist[(hs * 4 + 1) * (size_t)n + b];
      uint64_t k0 = hist[(hs * 4 + 2) * (size_t)n + b], k1 = hist[(hs * 4 + 3) * (size_t)n + b];
      int act = (abits >> hs) & 1;
      am_slot_planes(f0, f1, k0, k1, act, lane, pl);
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      int plane = slot * 6 + q;
      if (DTYPE == ELFHIP_F32) {
        ((float*)out)[((size_t)b * 36 + plane) * 64 + lane] = pl[q];
      } else if (DTYPE == ELFHIP_BF16) {
        ((uint16_t*)out)[((size_t)b * 36 + plane) * 64 + lane] = pl[q] != 0.f ? 0x3F80 : 0;
      } else {
        uint64_t bits = __ballot(pl[q] != 0.f);
        if (lane == 0) ((uint64_t*)out)[(size_t)b * 36 + plane] = bits;
      }
    }
  }
}

struct AmericanSlab : BoardSlab {
  uint64_t* w = nullptr;      // [6][n]
  uint64_t* hist = nullptr;   // [6 slots][4][n]
  uint32_t* hmeta = nullptr;  // [n]
  uint64_t* aos = nullptr;    // [n][6] staging for export/import
  ~AmericanSlab() override {
    (void)hipFree(w); (void)hipFree(hist); (void)hipFree(hmeta); (void)hipFree(aos);
  }
  int grid() const { return (n + 255) / 256; }
  int alloc() override {
    if (hipMalloc(&w, sizeof(uint64_t) * 6 * (size_t)n) != hipSuccess ||
        hipMalloc(&hist, sizeof(uint64_t) * 24 * (size_t)n) != hipSuccess ||
        hipMalloc(&hmeta, sizeof(uint32_t) * (size_t)n) != hipSuccess ||
        hipMalloc(&aos, sizeof(uint64_t) * 6 * (size_t)n) != hipSuccess) {
      elfhip_set_error("american: device allocation of %d boards failed", n);
      return ELFHIP_ENOMEM;
    }
    dev_ptr[0] = w; dev_ptr[1] = hist; dev_ptr[2] = hmeta;
    return reset();
  }
  int reset() override {
    step_counter = 0;
    hipLaunchKernelGGL(am_reset_kernel, dim3(grid()), dim3(256), 0, stream, w, hist, hmeta, n, (const uint8_t*)nullptr);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int reset_mask(const uint8_t* d_which) override {
    hipLaunchKernelGGL(am_reset_kernel, dim3(grid()), dim3(256), 0, stream, w, hist, hmeta, n, d_which);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int export_states_to(uint64_t* d_aos) override {
    hipLaunchKernelGGL(am_export_kernel, dim3(grid()), dim3(256), 0, stream, w, d_aos, n);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int export_states(uint64_t* h) override {
    if (int rc = export_states_to(aos)) return rc;
    HIPCHECK(hipMemcpyAsync(h, aos, sizeof(uint64_t) * 6 * (size_t)n, hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipStreamSynchronize(stream));
    return ELFHIP_OK;
  }
  int import_states(const uint64_t* h) override {
    HIPCHECK(hipMemcpyAsync(aos, h, sizeof(uint64_t) * 6 * (size_t)n, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(am_import_kernel, dim3(grid()), dim3(256), 0, stream, w, hist, hmeta, aos, n);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(stream));
    return ELFHIP_OK;
  }
  int legal_mask(uint64_t* d_mask, uint8_t* d_terminal) override {
    hipLaunchKernelGGL(am_mask_kernel, dim3(grid()), dim3(256), 0, stream, w, d_mask, d_terminal, n);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int step(const int16_t* d_actions, uint8_t* d_ok) override {
    hipLaunchKernelGGL(am_step_kernel, dim3(grid()), dim3(256), 0, stream, w, hist, hmeta, d_actions, d_ok, n);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int features(void* d_out, int dtype) override {
    int blocks = (n + 3) / 4;  // 4 waves (boards) per 256-thread block
    if (dtype == ELFHIP_F32)
      hipLaunchKernelGGL(am_features_kernel<ELFHIP_F32>, dim3(blocks), dim3(256), 0, stream, hist, hmeta, n, d_out);
    else if (dtype == ELFHIP_BF16)
      hipLaunchKernelGGL(am_features_kernel<ELFHIP_BF16>, dim3(blocks), dim3(256), 0, stream, hist, hmeta, n, d_out);
    else if (dtype == ELFHIP_BITS)
      hipLaunchKernelGGL(am_features_kernel<ELFHIP_BITS>, dim3(blocks), dim3(256), 0, stream, hist, hmeta, n, d_out);
    else { elfhip_set_error("american features: unknown dtype %d", dtype); return ELFHIP_EINVAL; }
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int evaluate(float* d_value) override {
    hipLaunchKernelGGL(am_evaluate_kernel, dim3(grid()), dim3(256), 0, stream, w, d_value, n);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int playout(int nply, uint64_t seed, uint64_t first_id, unsigned long long* d_moves) override {
    hipLaunchKernelGGL(am_playout_kernel, dim3(grid()), dim3(256), 0, stream, w, hist, hmeta, n, nply, seed,
                       first_id, step_counter, d_moves);
    HIPCHECK(hipGetLastError());
    step_counter += (uint64_t)nply;
    return ELFHIP_OK;
  }
};

}  // namespace

BoardSlab* elfhip_make_american() { return new AmericanSlab(); }
