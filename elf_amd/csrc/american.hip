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

namespace {

constexpr int AM_ACTIONS = 170;
constexpr int AM_MAX_PLY = 250;       // GameBoard.h:46
constexpr int AM_REPEAT = 4;          // GameBoard.h:48
constexpr uint64_t AM_UNUSED = (1ull << 8) | (1ull << 17) | (1ull << 26) | (1ull << 35);  // GameBoard.h:60
constexpr uint64_t AM_PLAYABLE = ((1ull << 36) - 1) ^ AM_UNUSED;

// move families in action-index order (HashAllMoves.h:193-363):
//  0: 0x11<<i dir1 (right forward)   1: 0x21<<i dir1 (left forward)
//  2: 0x11<<i dir0 (right backward)  3: 0x21<<i dir0 (left backward)
//  4: -(0x101<<i) dir1  5: -(0x401<<i) dir1  6: -(0x101<<i) dir0  7: -(0x401<<i) dir0
constexpr uint64_t valid_shifts(int span, bool jump) {
  uint64_t v = 0;
  for (int i = 0; i < 36; ++i) {
    bool ok = ((AM_PLAYABLE >> i) & 1) && (i + span < 36) && ((AM_PLAYABLE >> (i + span)) & 1);
    if (jump) ok = ok && ((AM_PLAYABLE >> (i + span / 2)) & 1);
    if (ok) v |= 1ull << i;
  }
  return v;
}
constexpr uint64_t V4 = valid_shifts(4, false), V5 = valid_shifts(5, false);
constexpr uint64_t V8 = valid_shifts(8, true), V10 = valid_shifts(10, true);
constexpr int popc_c(uint64_t v) { int c = 0; for (; v; v &= v - 1) ++c; return c; }
constexpr int N4 = popc_c(V4), N5 = popc_c(V5), N8 = popc_c(V8), N10 = popc_c(V10);
static_assert(N4 == 25 && N5 == 24 && N8 == 18 && N10 == 18, "action table layout");
static_assert(2 * (N4 + N5 + N8 + N10) == AM_ACTIONS, "170 actions");

struct FamInfo { uint64_t valid; int span; int base; int dir; };
__device__ __host__ constexpr FamInfo fam_info(int t) {
  switch (t) {
    case 0: return {V4, 4, 0, 1};
    case 1: return {V5, 5, N4, 1};
    case 2: return {V4, 4, N4 + N5, 0};
    case 3: return {V5, 5, 2 * N4 + N5, 0};
    case 4: return {V8, 8, 2 * (N4 + N5), 1};
    case 5: return {V10, 10, 2 * (N4 + N5) + N8, 1};
    case 6: return {V8, 8, 2 * (N4 + N5) + N8 + N10, 0};
    default: return {V10, 10, 2 * (N4 + N5) + 2 * N8 + N10, 0};
  }
}

// action -> (family, shift) table, built at compile time, resident in constant memory
struct ActTab { uint8_t fam[AM_ACTIONS + 2]; uint8_t shift[AM_ACTIONS + 2]; };
constexpr ActTab make_acttab() {
  ActTab t{};
  for (int f = 0; f < 8; ++f) {
    FamInfo fi = fam_info(f);
    int r = 0;
    for (int i = 0; i < 36; ++i)
      if ((fi.valid >> i) & 1) { t.fam[fi.base + r] = (uint8_t)f; t.shift[fi.base + r] = (uint8_t)i; ++r; }
  }
  return t;
}
__constant__ ActTab c_acttab = make_acttab();

struct Am {
  uint64_t f[2], k[2];  // forward / backward per colour
  uint64_t w4, w5;      // packed scalars, layout in oracle/american.h
};

__device__ inline int am_active(const Am& s) { return (int)(s.w4 & 1); }
__device__ inline int am_ply(const Am& s) { return (int)((s.w4 >> 8) & 0xFFFF); }

__device__ inline void am_clear(Am& s) {  // GameBoard.cc:10-35
  s.f[0] = 0x1eff; s.k[0] = 0; s.f[1] = 0; s.k[1] = 0x7fbc00000ull;
  s.w4 = (1ull << 8) | (171ull << 24);
  s.w5 = 0xFFFFFFFFull;
}

// The eight family bitboards, normalised so that bit i of fam[t] <=> move word W_t << i.
// GameBoard.cc:321-351 (shift rules), :359-463 (jumps are mandatory).
__device__ inline void am_families(const Am& s, uint64_t fam[8]) {
  int a = am_active(s);
  uint64_t pa = s.f[a] | s.k[a], pp = s.f[a ^ 1] | s.k[a ^ 1];
  uint64_t e = AM_PLAYABLE ^ (pa | pp);
  uint64_t f = s.f[a], k = s.k[a];
  uint64_t rfj = (e >> 8) & (pp >> 4) & f, lfj = (e >> 10) & (pp >> 5) & f;
  uint64_t rbj = (e << 8) & (pp << 4) & k, lbj = (e << 10) & (pp << 5) & k;
  bool anyj = (rfj | lfj | rbj | lbj) != 0;
  fam[4] = rfj; fam[5] = lfj; fam[6] = rbj >> 8; fam[7] = lbj >> 10;
  fam[0] = anyj ? 0 : ((e >> 4) & f);
  fam[1] = anyj ? 0 : ((e >> 5) & f);
  fam[2] = anyj ? 0 : (((e << 4) & k) >> 4);
  fam[3] = anyj ? 0 : (((e << 5) & k) >> 5);
}

__device__ inline int am_count(const uint64_t fam[8]) {
  int c = 0;
#pragma unroll
  for (int t = 0; t < 8; ++t) c += __popcll(fam[t]);
  return c;
}

// repeat rule GameBoard.cc:151-159: with more than one legal move, a side that repeated >= 4
// times may not play the move it played two turns ago.
__device__ inline int am_apply_repeat_rule(const Am& s, uint64_t fam[8], int total) {
  int a = am_active(s);
  int rep = (int)((s.w4 >> (32 + 8 * a)) & 0xFF);
  if (total > 1 && rep >= AM_REPEAT) {
    int banned = (int)((s.w5 >> (16 * a + 8)) & 0xFF);  // last_move_<side>[1]
    if (banned < AM_ACTIONS) {
      int bt = c_acttab.fam[banned], bi = c_acttab.shift[banned];
      int removed = 0;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        uint64_t bit = (t == bt) ? (1ull << bi) : 0;
        removed += (fam[t] & bit) != 0;
        fam[t] &= ~bit;
      }
      total -= removed;
    }
  }
  return total;
}

// compress the bits of x selected by the compile-time mask V to the low end (software pext over
// the mask's contiguous runs; V has 7-8 runs)
template <uint64_t V>
__device__ inline uint64_t compress_bits(uint64_t x) {
  uint64_t out = 0;
  int pos = 0;
#pragma unroll
  for (int s = 0; s < 36; ++s) {
    bool start = ((V >> s) & 1) && (s == 0 || !((V >> (s - 1)) & 1));
    if (start) {
      int len = 0;
      while (s + len < 36 && ((V >> (s + len)) & 1)) ++len;
      out |= ((x >> s) & ((1ull << len) - 1)) << pos;
      pos += len;
    }
  }
  return out;
}

// 170-bit action mask from the family bitboards
__device__ inline void am_mask_words(const uint64_t fam[8], uint64_t m[3]) {
  uint64_t c[8];
  c[0] = compress_bits<V4>(fam[0]); c[1] = compress_bits<V5>(fam[1]);
  c[2] = compress_bits<V4>(fam[2]); c[3] = compress_bits<V5>(fam[3]);
  c[4] = compress_bits<V8>(fam[4]); c[5] = compress_bits<V10>(fam[5]);
  c[6] = compress_bits<V8>(fam[6]); c[7] = compress_bits<V10>(fam[7]);
  m[0] = m[1] = m[2] = 0;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    int base = fam_info(t).base;
    int w = base >> 6, o = base & 63;
    m[w] |= c[t] << o;
    if (o != 0 && w + 1 < 3) m[w + 1] |= c[t] >> (64 - o);
  }
}

__device__ inline bool am_has_jump_from(const Am& s, int active, uint64_t piece) {  // GameBoard.cc:465-528
  uint64_t pa = s.f[active] | s.k[active], pp = s.f[active ^ 1] | s.k[active ^ 1];
  uint64_t e = AM_PLAYABLE ^ (pa | pp);
  uint64_t fj = ((e >> 8) & (pp >> 4) & piece) | ((e >> 10) & (pp >> 5) & piece);
  uint64_t bj = ((e << 8) & (pp << 4) & piece) | ((e << 10) & (pp << 5) & piece);
  // black always jumps forward, backward only as a king (piece in backward[BLACK]); white mirrored
  bool king = active == 0 ? (piece & s.k[0]) != 0 : (piece & s.f[1]) != 0;
  uint64_t own = active == 0 ? fj : bj, other = active == 0 ? bj : fj;
  return (own | (king ? other : 0)) != 0;
}

// CheckersPlay GameBoard.cc:37-129 for the move (family t, shift i) whose action index is `action`
__device__ inline void am_play(Am& s, int action, int t, int i) {
  const int a = am_active(s), p = a ^ 1;
  const bool jump_move = t >= 4;
  const int span = (t & 1) ? (jump_move ? 10 : 5) : (jump_move ? 8 : 4);
  const uint64_t wbits = (1ull | (1ull << span)) << i;  // the two-bit move word
  const int dir = (t & 2) ? 0 : 1;
  // repeat bookkeeping :59-79
  int lm0 = (int)((s.w5 >> (16 * a)) & 0xFF), lm1 = (int)((s.w5 >> (16 * a + 8)) & 0xFF);
  int rep = (int)((s.w4 >> (32 + 8 * a)) & 0xFF);
  rep = (lm1 == action) ? rep + 1 : 0;
  lm1 = lm0; lm0 = action;
  s.w5 = (s.w5 & ~(0xFFFFull << (16 * a))) | ((uint64_t)lm0 << (16 * a)) | ((uint64_t)lm1 << (16 * a + 8));
  int ply = am_ply(s) + 1;
  int jumpflag = (int)((s.w4 >> 1) & 1) | (jump_move ? 1 : 0);
  if (jump_move) {  // :85-100
    uint64_t taken = 1ull << (i + span / 2);
    s.f[p] &= ~taken;
    s.k[p] &= ~taken;
  }
  if (s.f[a] & wbits) s.f[a] ^= wbits;  // :102-106
  if (s.k[a] & wbits) s.k[a] ^= wbits;
  const uint64_t dest = dir ? (1ull << (i + span)) : (1ull << i);
  uint64_t repw = (s.w4 & ~(0xFFull << (32 + 8 * a))) | ((uint64_t)(rep & 0xFF) << (32 + 8 * a));
  repw &= 0x0000FFFF00000000ull;
  bool cont = jumpflag && am_has_jump_from(s, a, dest);  // :111-116 (Q1, Q3)
  int nactive = a;
  if (!cont) {
    if (a == 0 && (dest & 0x780000000ull)) s.k[0] |= dest;  // crown :118-121
    else if (a == 1 && (dest & 0xfull)) s.f[1] |= dest;
    jumpflag = 0;
    nactive = p;
  }
  s.w4 = (uint64_t)nactive | ((uint64_t)jumpflag << 1) | ((uint64_t)ply << 8) | ((uint64_t)action << 24) | repw;
}

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
__global__ void am_reset_kernel(uint64_t* w, uint64_t* hist, uint32_t* hmeta, int n) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
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
      int kk = (int)elf_mod_u64_small(r, (uint32_t)total);
      // k-th legal move in ascending action index == k-th set bit over fam[0..7]
      int t = 0; uint64_t sel = 0; bool found = false;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        int c = __popcll(fam[q]);
        bool here = !found && kk < c;
        if (here) { t = q; sel = fam[q]; found = true; }
        if (!found) kk -= c;
      }
      for (int j = 0; j < kk; ++j) sel &= sel - 1;
      int i = __ffsll((long long)sel) - 1;
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
  int y = lane >> 3, x = lane & 7;
  // square -> padded bit index, black orientation (inverse of GameBoard.cc:230-232)
  auto sq2bit = [](int yy, int xx) -> int {
    int rr = 7 - yy;
    int t = 6 + (rr & 1) - xx;
    if (t < 0 || t > 6 || (t & 1)) return -1;
    int buff = rr * 4 + (t >> 1);
    return buff + (buff >> 3);
  };
  int bit_b = sq2bit(y, x);          // black's view
  int bit_w = sq2bit(7 - y, 7 - x);  // white's view (180 degree rotation)
  for (int slot = 0; slot < 6; ++slot) {
    int k = slot - (6 - len);  // history index, oldest first; negative -> zero padding
    float pl[6] = {0, 0, 0, 0, 0, 0};
    if (k >= 0) {
      int hs = (head + k) % 6;
      uint64_t f0 = hist[(hs * 4 + 0) * (size_t)n + b], f1 = hist[(hs * 4 + 1) * (size_t)n + b];
      uint64_t k0 = hist[(hs * 4 + 2) * (size_t)n + b], k1 = hist[(hs * 4 + 3) * (size_t)n + b];
      int act = (abits >> hs) & 1;
      // pawns / kings per colour (a king sits in both words; GetObservation tests kings first)
      uint64_t pawn[2] = {f0 & ~k0, k1 & ~f1}, king[2] = {k0, f1};
      int ba = act == 0 ? bit_b : bit_w;  // mover's planes in the mover's orientation
      int bp = act == 0 ? bit_w : bit_b;  // opponent's planes in the opponent's orientation (Q7)
      if (ba >= 0) { pl[0] = (pawn[act] >> ba) & 1; pl[1] = (king[act] >> ba) & 1; }
      if (bp >= 0) { pl[2] = (pawn[act ^ 1] >> bp) & 1; pl[3] = (king[act ^ 1] >> bp) & 1; }
      pl[4] = act == 0 ? 1.f : 0.f;
      pl[5] = act == 1 ? 1.f : 0.f;
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
    return reset();
  }
  int reset() override {
    step_counter = 0;
    hipLaunchKernelGGL(am_reset_kernel, dim3(grid()), dim3(256), 0, stream, w, hist, hmeta, n);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int export_states(uint64_t* h) override {
    hipLaunchKernelGGL(am_export_kernel, dim3(grid()), dim3(256), 0, stream, w, aos, n);
    HIPCHECK(hipGetLastError());
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
