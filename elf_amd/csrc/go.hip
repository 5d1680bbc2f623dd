// Go 19x19 on gfx950.  BUILD-DEFINED rules (the reference fork has no Go source);
// semantics are those of oracle/go.c, which this file is held bit-exact to.
//
// Formulation: one lane owns one board.  A board is 19 row bitmasks per colour held in VGPRs
// (every row loop is fully unrolled so indices are static and nothing spills to scratch), plus a
// third bitboard AT = "stones of groups that have exactly one liberty".  AT is maintained
// incrementally, which makes the whole-board legal mask pure bit arithmetic:
//     T = empty | (own & ~AT) | (opp & AT)          legal = empty & neighbours(T) & ~ko
// (a point is playable iff it touches an empty point, an own group that keeps a liberty, or an
// opponent group in atari).  Groups are grown by alternating row sweeps whose exit is a wave-wide
// ballot (__any), so all 64 boards of a wavefront iterate together; after a move a group is grown
// only until its liberty status is decided (go_dev.h: go_group_status).  Boards are struct-of-arrays
// in HBM (u32 word w of board b at [w*n + b]) so loads/stores are coalesced 4 B/lane.
// Positional superko uses a per-board open-addressing table of 512 buckets x 4 {hash, game
// generation} in HBM (32 KB/board, sized for 288 GB parts) instead of scanning the move history.
#include "common.h"
#include "go_dev.h"

namespace {
using namespace godev;

// ---- kernels ------------------------------------------------------------------------------------
__global__ void go_zobrist_init_kernel(uint64_t* gz) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 2 * GO_POINTS) gz[i] = elf_splitmix64(GO_ZSEED + (uint64_t)i);
}

__global__ void __launch_bounds__(64) go_reset_kernel(GoDev d, const uint8_t* __restrict__ which) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= d.n || (which && !which[b])) return;
  GoS s; go_clear(s);
  go_store(d, b, s);
  uint64_t gen = d.gen[b] + 1;
  d.gen[b] = gen;
  go_table_test_and_insert(d.table + (size_t)b * TAB_SLOTS, gen, 0);  // the empty start position
  go_hist_set_single(d, b, s);
}

__global__ void __launch_bounds__(64) go_import_kernel(GoDev d, const uint64_t* __restrict__ aos) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = b < d.n;
  GoS s; go_clear(s);
  if (valid) {
    const uint64_t* w = aos + (size_t)b * GO_WORDS;
    uint64_t m = w[20];
    s.tm = m & 1; s.passes = (m >> 1) & 3; s.superko = (m >> 3) & 1; s.ply = (m >> 16) & 0xFFFF;
    s.ko = (m >> 32) & 0xFFFF; s.last = (m >> 48) & 0xFFFF;
    uint32_t cb = (uint32_t)w[21], cw = (uint32_t)(w[21] >> 32);
    s.capP = s.tm ? cw : cb; s.capQ = s.tm ? cb : cw;
    s.hash = w[19];
    ROWS(i) {
      uint32_t bl = (uint32_t)w[i] & FULL, wh = (uint32_t)(w[i] >> 32) & FULL;
      s.P[i] = s.tm ? wh : bl; s.Q[i] = s.tm ? bl : wh;
    }
  }
  go_recompute_at(s);
  if (valid) {
    go_store(d, b, s);
    uint64_t gen = d.gen[b] + 1;
    d.gen[b] = gen;
    go_table_test_and_insert(d.table + (size_t)b * TAB_SLOTS, gen, s.hash);
    go_hist_set_single(d, b, s);
  }
}

__global__ void __launch_bounds__(64) go_export_kernel(GoDev d, uint64_t* __restrict__ aos) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= d.n) return;
  GoS s; go_load(d, b, s);
  uint64_t* w = aos + (size_t)b * GO_WORDS;
  ROWS(i) {
    uint32_t bl = s.tm ? s.Q[i] : s.P[i], wh = s.tm ? s.P[i] : s.Q[i];
    w[i] = (uint64_t)bl | ((uint64_t)wh << 32);
  }
  w[19] = s.hash;
  w[20] = (uint64_t)s.tm | ((uint64_t)s.passes << 1) | ((uint64_t)s.superko << 3) | ((uint64_t)s.ply << 16) |
          ((uint64_t)(s.ko & 0xFFFF) << 32) | ((uint64_t)(s.last & 0xFFFF) << 48);
  uint32_t cb = s.tm ? s.capQ : s.capP, cw = s.tm ? s.capP : s.capQ;
  w[21] = (uint64_t)cb | ((uint64_t)cw << 32);
}

__global__ void __launch_bounds__(64) go_mask_kernel(GoDev d, uint64_t* __restrict__ mask, uint8_t* __restrict__ term) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= d.n) return;
  GoS s; go_load(d, b, s);
  uint32_t L[GN];
  go_legal_rows(s, L);
  uint64_t m[6];
  const bool alive = !go_terminated(s);
  go_pack_mask(L, alive, m);
#pragma unroll
  for (int k = 0; k < 6; ++k) mask[(size_t)b * 6 + k] = m[k];
  if (term) term[b] = alive ? 0 : 1;
}

__global__ void __launch_bounds__(64) go_step_kernel(GoDev d, const uint64_t* __restrict__ gz,
                                                     const int16_t* __restrict__ actions, uint8_t* __restrict__ ok) {
  __shared__ uint64_t z[GO_ZWORDS];
  go_load_zobrist(z, gz);
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = b < d.n;
  GoS s; go_clear(s);
  int a = -1;
  if (valid) { go_load(d, b, s); a = actions[b]; }
  bool good = valid && a >= 0 && a <= GO_PASS && !go_terminated(s);
  uint32_t L[GN];
  go_legal_rows(s, L);
  if (good && a < GO_POINTS) {
    uint32_t r = (uint32_t)a / 19u, c = (uint32_t)a % 19u, row = 0;
    ROWS(i) row = (uint32_t)i == r ? L[i] : row;
    good = (row >> c) & 1;
  }
  uint64_t gen = valid ? d.gen[b] : 0;
  go_forward(s, good, (uint32_t)(a < 0 ? 0 : a), d.table + (size_t)(valid ? b : 0) * TAB_SLOTS, gen, z);
  if (good) { go_store(d, b, s); go_hist_push(d, b, s); }
  if (valid && ok) ok[b] = good ? 1 : 0;
}

// x mod n for n <= 361 with rc = floor((2^32 - 1) / n): the estimate q = mulhi(x, rc) is the quotient or one less
__device__ inline uint32_t mod_small(uint32_t x, uint32_t n, uint32_t rc) {
  const uint32_t r = x - __umulhi(x, rc) * n;
  return r >= n ? r - n : r;
}
// position of the k-th (from 0) set bit of a 19-bit row, k < popcount(x): five halving steps, no loop
__device__ inline uint32_t kth_bit(uint32_t x, int k) {
  uint32_t pos = 0;
#pragma unroll
  for (int w = 16; w >= 1; w >>= 1) {
    const int c = __popc(x & ((1u << w) - 1u));
    const bool hi = k >= c;
    k -= hi ? c : 0; x = hi ? x >> w : x; pos += hi ? (uint32_t)w : 0u;
  }
  return pos;
}

// Random-legal playouts: nply steps per launch with the board in registers.
// Two kernels share the body.  The plain one (64-thread blocks, no register limit: 294 registers, nothing spilled)
// is the one for up to one wave per SIMD - BASELINE.json configs[1]'s 65,536 boards on the MI355X's 1,024 SIMDs.
// A lone wave issues an integer instruction only every ~5 cycles, a SIMD takes one every ~2.7 (benchmarks/ubench/):
// for larger slabs the dense kernel (256-thread blocks sharing the LDS tables, at most 256 registers so that two
// waves fit a SIMD, a few registers spilled) runs the same boards 30-38 % faster than queueing them behind the
// first 65,536.
__device__ __forceinline__ void
go_playout_body(GoDev d, const uint64_t* __restrict__ gz, int nply, uint64_t seed, uint64_t first_id,
                uint64_t step_base, unsigned long long* __restrict__ moves_out) {
  __shared__ uint64_t z[GO_ZWORDS];
  __shared__ uint32_t s_mod[2][GO_POINTS + 1];   // per legal-move count n: floor((2^32 - 1) / n), 2^32 mod n
  for (int n = threadIdx.x + 1; n <= GO_POINTS; n += blockDim.x) {
    s_mod[0][n] = 0xFFFFFFFFu / (uint32_t)n;
    s_mod[1][n] = ((0xFFFFFFFFu % (uint32_t)n) + 1u) % (uint32_t)n;
  }
  go_load_zobrist(z, gz);
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = b < d.n;
  GoS s; go_clear(s);
  uint64_t gen = 0;
  ulonglong2* tab = d.table + (size_t)(valid ? b : 0) * TAB_SLOTS;
  if (valid) { go_load(d, b, s); gen = d.gen[b]; }
  const uint64_t key = seed + (first_id + (uint64_t)b) * ELF_GOLDEN + step_base;
  unsigned long long moves = 0;
#ifdef GO_PROF
  if (threadIdx.x < 32) s_go_prof[threadIdx.x] = 0;
  __syncthreads();
#endif
#pragma unroll 1
  for (int step = 0; step < nply; ++step) {
    PROF_T0;
    const bool term = go_terminated(s);
    if (valid && term) {  // reset in place, 0 moves
      go_clear(s);
      gen += 1;
      go_table_test_and_insert(tab, gen, 0);
    }
    const bool act = valid && !term;
    uint32_t L[GN];
    int inc[GN];   // legal points up to and including each row
    int n = go_legal_rows(s, L, inc);
    uint32_t a = GO_PASS;
    if (act && n > 0) {
      const uint64_t r = elf_splitmix64(key + (uint64_t)step);
      // r mod n (= elf_mod_u64_small) through the per-n reciprocals in LDS: three multiply-high reductions
      const uint32_t rc = s_mod[0][n], p32 = s_mod[1][n];
      const uint32_t hi = mod_small((uint32_t)(r >> 32), (uint32_t)n, rc), lo = mod_small((uint32_t)r, (uint32_t)n, rc);
      int k = (int)mod_small(hi * p32 + lo, (uint32_t)n, rc);
#ifdef GO_SELECT_SCAN   // A/B only: the row by a running scan
      uint32_t row = 0, rowbits = 0;
      bool found = false;
      ROWS(i) {
        int c = __popc(L[i]);
        bool here = !found && k < c;
        if (here) { row = i; rowbits = L[i]; found = true; }
        if (!found) k -= c;
      }
#else
      // the row that holds the k-th legal point = the number of rows whose running count is still <= k
      // (a compare and an add-with-carry per row), then that row's mask and the count before it by row index
      uint32_t row = 0;
#pragma unroll
      for (int i = 0; i < GN - 1; ++i) row += inc[i] <= k ? 1u : 0u;
      uint32_t rowbits = L[0]; int before = 0;
#pragma unroll
      for (int i = 1; i < GN; ++i) { const bool here = row == (uint32_t)i; rowbits = here ? L[i] : rowbits; before = here ? inc[i - 1] : before; }
      k -= before;
#endif
      a = row * 19 + kth_bit(rowbits, k);
    }
    PROF_ADD(4);
    go_forward(s, act, a, tab, gen, z);
    PROF_ADD(5);
    moves += act ? 1 : 0;
  }
#ifdef GO_PROF
  __syncthreads();
  if (threadIdx.x < 32) atomicAdd(&g_go_prof[threadIdx.x], s_go_prof[threadIdx.x]);
#endif
  if (valid) { go_store(d, b, s); d.gen[b] = gen; go_hist_set_single(d, b, s); }  // playouts keep no plane history
  if (moves_out) {
    for (int o = 32; o > 0; o >>= 1) moves += __shfl_down(moves, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(moves_out, moves);
  }
}

__global__ void __launch_bounds__(64)
go_playout_kernel(GoDev d, const uint64_t* __restrict__ gz, int nply, uint64_t seed, uint64_t first_id,
                  uint64_t step_base, unsigned long long* __restrict__ moves_out) {
  go_playout_body(d, gz, nply, seed, first_id, step_base, moves_out);
}
__attribute__((amdgpu_waves_per_eu(2, 2))) __global__ void __launch_bounds__(256)
go_playout_dense_kernel(GoDev d, const uint64_t* __restrict__ gz, int nply, uint64_t seed, uint64_t first_id,
                        uint64_t step_base, unsigned long long* __restrict__ moves_out) {
  go_playout_body(d, gz, nply, seed, first_id, step_base, moves_out);
}

// Tromp-Taylor area score, komi 7.5 (oracle/go.c go_tt_score); value = 0 unless terminated
__global__ void __launch_bounds__(64) go_evaluate_kernel(GoDev d, float* __restrict__ value, float* __restrict__ score) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = b < d.n;
  GoS s; go_clear(s);
  if (valid) go_load(d, b, s);
  uint32_t E[GN], RP[GN], RQ[GN];
  ROWS(i) E[i] = ~(s.P[i] | s.Q[i]) & FULL;
  ROWS(i) { RP[i] = NBR(s.P, i) & E[i]; RQ[i] = NBR(s.Q, i) & E[i]; }
  go_fill(RP, E);
  go_fill(RQ, E);
  int ap = 0, aq = 0;
  ROWS(i) { ap += __popc(s.P[i]) + __popc(RP[i] & ~RQ[i]); aq += __popc(s.Q[i]) + __popc(RQ[i] & ~RP[i]); }
  int black = s.tm ? aq : ap, white = s.tm ? ap : aq;
  float sc = (float)black - (float)white - 7.5f;
  if (valid) {
    if (score) score[b] = sc;
    if (value) value[b] = go_terminated(s) ? (sc > 0 ? 1.f : -1.f) : 0.f;
  }
}

// 18 AGZ-style planes (oracle/go.c go_features): 2i = side-to-move's stones i plies ago, 2i+1 = the
// opponent's (i = 0..7, zero when the game is younger), 16 / 17 = all ones if black / white is to move.
// One workgroup per board, thread = point.  BITS form: u64 [18][6], bit r*19+c.
template <int DTYPE>
__global__ void __launch_bounds__(384) go_features_kernel(GoDev d, void* __restrict__ out) {
  __shared__ unsigned long long sbits[18 * 6];
  const int b = blockIdx.x, p = threadIdx.x;
  const size_t n = d.n;
  const uint32_t tm = d.meta[b] & 1;
  const uint32_t hm = d.hmeta[b];
  const int len = hm & 15, head = (hm >> 4) & 7;
  const int r = p / 19, c = p % 19;
  if (DTYPE == ELFHIP_BITS) { if (p < 18 * 6) sbits[p] = 0; __syncthreads(); }
  for (int i = 0; i < 8; ++i) {
    float own = 0.f, opp = 0.f;
    if (p < GO_POINTS && i < len) {
      const int slot = (head + len - 1 - i) & 7;
      const uint32_t bl = d.hist[((size_t)slot * 38 + r) * n + b], wh = d.hist[((size_t)slot * 38 + 19 + r) * n + b];
      const uint32_t o = tm ? wh : bl, q = tm ? bl : wh;
      own = (float)((o >> c) & 1); opp = (float)((q >> c) & 1);
    }
    if (p < GO_POINTS) {
      if (DTYPE == ELFHIP_F32) {
        ((float*)out)[((size_t)b * 18 + 2 * i) * GO_POINTS + p] = own;
        ((float*)out)[((size_t)b * 18 + 2 * i + 1) * GO_POINTS + p] = opp;
      } else if (DTYPE == ELFHIP_BF16) {
        ((uint16_t*)out)[((size_t)b * 18 + 2 * i) * GO_POINTS + p] = own != 0.f ? 0x3F80 : 0;
        ((uint16_t*)out)[((size_t)b * 18 + 2 * i + 1) * GO_POINTS + p] = opp != 0.f ? 0x3F80 : 0;
      } else {
        if (own != 0.f) atomicOr(&sbits[(2 * i) * 6 + (p >> 6)], 1ull << (p & 63));
        if (opp != 0.f) atomicOr(&sbits[(2 * i + 1) * 6 + (p >> 6)], 1ull << (p & 63));
      }
    }
  }
  if (p < GO_POINTS) {
    const float bk = tm == 0 ? 1.f : 0.f, wk = tm == 1 ? 1.f : 0.f;
    if (DTYPE == ELFHIP_F32) {
      ((float*)out)[((size_t)b * 18 + 16) * GO_POINTS + p] = bk;
      ((float*)out)[((size_t)b * 18 + 17) * GO_POINTS + p] = wk;
    } else if (DTYPE == ELFHIP_BF16) {
      ((uint16_t*)out)[((size_t)b * 18 + 16) * GO_POINTS + p] = bk != 0.f ? 0x3F80 : 0;
      ((uint16_t*)out)[((size_t)b * 18 + 17) * GO_POINTS + p] = wk != 0.f ? 0x3F80 : 0;
    } else {
      atomicOr(&sbits[(16 + tm) * 6 + (p >> 6)], 1ull << (p & 63));
    }
  }
  if (DTYPE == ELFHIP_BITS) {
    __syncthreads();
    if (p < 18 * 6) ((unsigned long long*)out)[(size_t)b * 18 * 6 + p] = sbits[p];
  }
}

struct GoSlab : BoardSlab {
  GoDev d{};
  uint64_t* gz = nullptr;   // Zobrist table [2*361]
  uint64_t* aos = nullptr;  // [n][22] staging
  ~GoSlab() override {
    (void)hipFree(d.rows); (void)hipFree(d.meta); (void)hipFree(d.hash); (void)hipFree(d.gen);
    (void)hipFree(d.table); (void)hipFree(gz); (void)hipFree(aos); (void)hipFree(d.hist); (void)hipFree(d.hmeta);
  }
  int grid() const { return (n + 63) / 64; }
  int one_wave_per_simd = 65536;   // lanes of one wave on every SIMD of the device (set in alloc)
  int alloc() override {
    d.n = n;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) one_wave_per_simd = cus * 4 * 64;
    size_t N = (size_t)n;
    if (hipMalloc(&d.rows, sizeof(uint32_t) * NROWW * N) != hipSuccess ||
        hipMalloc(&d.meta, sizeof(uint32_t) * 4 * N) != hipSuccess ||
        hipMalloc(&d.hash, sizeof(uint64_t) * N) != hipSuccess ||
        hipMalloc(&d.gen, sizeof(uint64_t) * N) != hipSuccess ||
        hipMalloc(&d.table, sizeof(ulonglong2) * TAB_SLOTS * N) != hipSuccess ||
        hipMalloc(&gz, sizeof(uint64_t) * 2 * GO_POINTS) != hipSuccess ||
        hipMalloc(&d.hist, sizeof(uint32_t) * 8 * 38 * N) != hipSuccess ||
        hipMalloc(&d.hmeta, sizeof(uint32_t) * N) != hipSuccess ||
        hipMalloc(&aos, sizeof(uint64_t) * GO_WORDS * N) != hipSuccess) {
      elfhip_set_error("go19: device allocation of %d boards failed (needs %.1f MB)", n,
                       (double)N * (sizeof(ulonglong2) * TAB_SLOTS + 600) / 1e6);
      return ELFHIP_ENOMEM;
    }
    dev_ptr[0] = &d; dev_ptr[1] = gz;  // host pointer to the GoDev descriptor + Zobrist table (mcts_go.hip)
    HIPCHECK(hipMemsetAsync(d.gen, 0, sizeof(uint64_t) * N, stream));
    HIPCHECK(hipMemsetAsync(d.table, 0, sizeof(ulonglong2) * TAB_SLOTS * N, stream));
    hipLaunchKernelGGL(go_zobrist_init_kernel, dim3(3), dim3(256), 0, stream, gz);
    HIPCHECK(hipGetLastError());
    return reset();
  }
  int reset() override {
    step_counter = 0;
    hipLaunchKernelGGL(go_reset_kernel, dim3(grid()), dim3(64), 0, stream, d, (const uint8_t*)nullptr);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int reset_mask(const uint8_t* d_which) override {
    hipLaunchKernelGGL(go_reset_kernel, dim3(grid()), dim3(64), 0, stream, d, d_which);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int export_states_to(uint64_t* d_aos) override {
    hipLaunchKernelGGL(go_export_kernel, dim3(grid()), dim3(64), 0, stream, d, d_aos);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int export_states(uint64_t* h) override {
    if (int rc = export_states_to(aos)) return rc;
    HIPCHECK(hipMemcpyAsync(h, aos, sizeof(uint64_t) * GO_WORDS * (size_t)n, hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipStreamSynchronize(stream));
    return ELFHIP_OK;
  }
  int import_states(const uint64_t* h) override {
    HIPCHECK(hipMemcpyAsync(aos, h, sizeof(uint64_t) * GO_WORDS * (size_t)n, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(go_import_kernel, dim3(grid()), dim3(64), 0, stream, d, aos);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(stream));
    return ELFHIP_OK;
  }
  int legal_mask(uint64_t* d_mask, uint8_t* d_terminal) override {
    hipLaunchKernelGGL(go_mask_kernel, dim3(grid()), dim3(64), 0, stream, d, d_mask, d_terminal);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int step(const int16_t* d_actions, uint8_t* d_ok) override {
    hipLaunchKernelGGL(go_step_kernel, dim3(grid()), dim3(64), 0, stream, d, gz, d_actions, d_ok);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int features(void* d_out, int dtype) override {
    if (dtype == ELFHIP_F32) hipLaunchKernelGGL(go_features_kernel<ELFHIP_F32>, dim3(n), dim3(384), 0, stream, d, d_out);
    else if (dtype == ELFHIP_BF16) hipLaunchKernelGGL(go_features_kernel<ELFHIP_BF16>, dim3(n), dim3(384), 0, stream, d, d_out);
    else if (dtype == ELFHIP_BITS) hipLaunchKernelGGL(go_features_kernel<ELFHIP_BITS>, dim3(n), dim3(384), 0, stream, d, d_out);
    else { elfhip_set_error("go19 features: unknown dtype %d", dtype); return ELFHIP_EINVAL; }
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int evaluate(float* d_value) override {
    hipLaunchKernelGGL(go_evaluate_kernel, dim3(grid()), dim3(64), 0, stream, d, d_value, (float*)nullptr);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int playout(int nply, uint64_t seed, uint64_t first_id, unsigned long long* d_moves) override {
    if (n > one_wave_per_simd)   // more boards than one wave per SIMD can hold: two waves per SIMD, see go_playout_body
      hipLaunchKernelGGL(go_playout_dense_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d, gz, nply, seed, first_id,
                         step_counter, d_moves);
    else
      hipLaunchKernelGGL(go_playout_kernel, dim3(grid()), dim3(64), 0, stream, d, gz, nply, seed, first_id,
                         step_counter, d_moves);
    HIPCHECK(hipGetLastError());
    step_counter += (uint64_t)nply;
    return ELFHIP_OK;
  }
};

}  // namespace

BoardSlab* elfhip_make_go19() { return new GoSlab(); }

#ifdef GO_PROF
extern "C" int elfhip_debug_go_prof(unsigned long long* h_out32, int reset) {
  if (h_out32 && hipMemcpyFromSymbol(h_out32, HIP_SYMBOL(godev::g_go_prof), 32 * 8) != hipSuccess) return -2;
  if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(godev::g_go_prof), z, 32 * 8) != hipSuccess) return -2; }
  return 0;
}
#endif
