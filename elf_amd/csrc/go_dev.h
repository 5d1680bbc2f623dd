// Device-side Go 19x19 board code shared by the board slab (go.hip) and the Go tree search
// (mcts_go.hip).  See go.hip for the formulation notes.
#pragma once
#include "common.h"

namespace godev {

#ifdef GO_PROF
// timing-only instrumentation (never in the product build): per-wave cycle counts by section
__device__ unsigned long long g_go_prof[32];
static __shared__ unsigned long long s_go_prof[32];
#define PROF_T0 unsigned long long _t0 = clock64()
#define PROF_ADD(k) do { unsigned long long _t1 = clock64(); if (threadIdx.x == 0) s_go_prof[k] += _t1 - _t0; _t0 = clock64(); } while (0)
#define PROF_INC(k, v) do { if (threadIdx.x == 0) s_go_prof[k] += (v); } while (0)
#else
#define PROF_T0
#define PROF_ADD(k)
#define PROF_INC(k, v)
#endif

constexpr int GN = 19;
constexpr uint32_t FULL = 0x7FFFFu;
constexpr int GO_POINTS = 361;
constexpr int GO_PASS = 361;
constexpr int GO_MAX_PLY = 722;
constexpr uint32_t NO_POINT = 0xFFFFu;
constexpr uint64_t GO_ZSEED = 0x60BA7D5EEDull;
constexpr int TAB_SLOTS = 2048;                  // superko set entries per board (a game has <= 723 positions)
constexpr int TAB_BUCKETS = TAB_SLOTS / 4;        // buckets of 4 x {hash, generation} = one 64-byte line
constexpr int GO_WORDS = 22;
constexpr int NROWW = 57;  // B[19], W[19], AT[19]
// LDS Zobrist block: [0, 722) the per-point keys z[colour*361 + point]; then the row-chunk table
// zc[colour][row][chunk 0..4][16]: XOR of the keys of the stones in a 4-column nibble of a row, so the
// keys of a whole captured row come out of 5 lookups instead of one dependent lookup per stone
constexpr int GO_ZC_BASE = 2 * GO_POINTS, GO_ZC_ROW = 5 * 16;
constexpr int GO_ZWORDS = GO_ZC_BASE + 2 * GN * GO_ZC_ROW;

#define ROWS(i) _Pragma("unroll") for (int i = 0; i < GN; ++i)
#define ROWS_REV(i) _Pragma("unroll") for (int i = GN - 1; i >= 0; --i)

struct GoS {
  uint32_t P[GN], Q[GN], AT[GN];  // P = stones of the side to move, Q = the other side
  uint32_t tm, passes, superko, ply, ko, last, capP, capQ;
  uint64_t hash;
};

__device__ inline bool go_terminated(const GoS& s) { return s.passes >= 2 || s.ply >= GO_MAX_PLY || s.superko; }

__device__ inline void go_clear(GoS& s) {
  ROWS(i) { s.P[i] = 0; s.Q[i] = 0; s.AT[i] = 0; }
  s.tm = 0; s.passes = 0; s.superko = 0; s.ply = 0; s.ko = NO_POINT; s.last = NO_POINT; s.capP = 0; s.capQ = 0;
  s.hash = 0;
}

// 4-neighbourhood of a row-bitboard set
#define NBR(X, i) ((((X)[i] << 1) | ((X)[i] >> 1) | ((i) > 0 ? (X)[(i) > 0 ? (i)-1 : 0] : 0u) | ((i) < GN - 1 ? (X)[(i) < GN - 1 ? (i) + 1 : 0] : 0u)) & FULL)

// Flood fill: grow F (subset of M) to the union of M-connected components it touches.
// Alternating down/up sweeps, in place; exits when no lane of the wave changed.
__device__ inline void go_fill(uint32_t (&F)[GN], const uint32_t (&M)[GN]) {
  // A complete sweep that changes nothing proves a fixpoint, so test after every sweep.
  PROF_INC(20, 1);
  for (;;) {
    uint32_t chg = 0;
    PROF_INC(21, 1);
    ROWS(i) {
      uint32_t up = i > 0 ? F[i > 0 ? i - 1 : 0] : 0u, dn = i < GN - 1 ? F[i < GN - 1 ? i + 1 : 0] : 0u;
      uint32_t t = (F[i] | up | dn) & M[i];
      t = (t | (t << 1) | (t >> 1)) & M[i];
      t = (t | (t << 1) | (t >> 1)) & M[i];
      chg |= t ^ F[i];
      F[i] = t;
    }
    if (!__any(chg != 0)) break;
    chg = 0;
    PROF_INC(21, 1);
    ROWS_REV(i) {
      uint32_t up = i > 0 ? F[i > 0 ? i - 1 : 0] : 0u, dn = i < GN - 1 ? F[i < GN - 1 ? i + 1 : 0] : 0u;
      uint32_t t = (F[i] | up | dn) & M[i];
      t = (t | (t << 1) | (t >> 1)) & M[i];
      t = (t | (t << 1) | (t >> 1)) & M[i];
      chg |= t ^ F[i];
      F[i] = t;
    }
    if (!__any(chg != 0)) break;
  }
}

__device__ inline bool go_any_rows(const uint32_t (&X)[GN]) {
  uint32_t o = 0;
  ROWS(i) o |= X[i];
  return __any(o != 0);
}

// number of liberties of the stone set G (empty points adjacent to it)
__device__ inline int go_libs(const uint32_t (&G)[GN], const GoS& s) {
  int n = 0;
  ROWS(i) {
    uint32_t e = ~(s.P[i] | s.Q[i]) & FULL;
    n += __popc(NBR(G, i) & e);
  }
  return n;
}

__device__ inline void go_set_at(GoS& s, const uint32_t (&G)[GN], bool in_atari) {
  ROWS(i) s.AT[i] = (s.AT[i] & ~G[i]) | (in_atari ? G[i] : 0u);
}

// whole-board legal point mask for the side to move
__device__ inline int go_legal_rows(const GoS& s, uint32_t (&L)[GN]) {
  uint32_t T[GN], E[GN];
  ROWS(i) {
    E[i] = ~(s.P[i] | s.Q[i]) & FULL;
    T[i] = E[i] | (s.P[i] & ~s.AT[i]) | (s.Q[i] & s.AT[i]);
  }
  const uint32_t kr = s.ko == NO_POINT ? 99u : s.ko / 19u, kb = s.ko == NO_POINT ? 0u : 1u << (s.ko % 19u);
  int n = 0;
  ROWS(i) {
    uint32_t l = E[i] & NBR(T, i);
    l &= ~((uint32_t)i == kr ? kb : 0u);
    L[i] = l;
    n += __popc(l);
  }
  return n;
}

// the same, also giving the running count of legal points after every row (for picking the k-th one)
__device__ inline int go_legal_rows(const GoS& s, uint32_t (&L)[GN], int (&inc)[GN]) {
  uint32_t T[GN], E[GN];
  ROWS(i) {
    E[i] = ~(s.P[i] | s.Q[i]) & FULL;
    T[i] = E[i] | (s.P[i] & ~s.AT[i]) | (s.Q[i] & s.AT[i]);
  }
  uint32_t kr = s.ko == NO_POINT ? 99u : s.ko / 19u;
  const uint32_t kb = s.ko == NO_POINT ? 0u : 1u << (s.ko - kr * 19u);
  // kr is used as a row number: left to itself the compiler turns every "row == kr" into a range test on the point
  // (add, compare, select per row) to save this one division
  asm volatile("" : "+v"(kr));
  int n = 0;
  ROWS(i) {
    uint32_t l = E[i] & NBR(T, i);
    l &= ~((uint32_t)i == kr ? kb : 0u);
    L[i] = l;
    n += __popc(l);
    inc[i] = n;
  }
  return n;
}

// One growth pass over the whole board, in place: rows in ascending (DOWN) or descending order.  A row takes the
// final value of its predecessor in sweep order and is then flooded EXACTLY along the row: for a seed set x inside a
// row mask m, (m + x) carries through every run of m upwards from its lowest seed, so ((m + x) ^ m) & m are the run
// bits above the seeds; the same on the bit-reversed row gives the bits below.  After the pass every row is closed
// sideways and towards the sweep direction; what can be left open is only the direction the sweep came from:
//   pend = bits whose neighbour in the row before (in sweep order) is an M point the set does not hold.
// pend == 0 <=> F is a union of complete M-components: no extra pass is needed to prove the fixpoint.
//   nlib = distinct empty points (E) adjacent to F as the pass leaves it (each row is counted one step later,
//          when both of its vertical neighbours are final): a lower bound of the group's liberties, exact when pend == 0.
// ~16 VALU instructions per row.
// acc += popcount(x) as ONE instruction (v_bcnt_u32_b32 adds its third operand); left to itself the compiler counts
// two rows separately and joins them with a three-way add - three instructions for two rows
#ifdef GO_POPC_PLAIN   // A/B only
#define GO_POPC_ACC(acc, x) (acc) += __popc(x)
#else
#define GO_POPC_ACC(acc, x) do { const uint32_t _x = (x); asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(_x)); } while (0)
#endif
template <bool EVAL, bool DOWN>
__device__ inline void go_grow(uint32_t (&F)[GN], const uint32_t (&M)[GN], const uint32_t (&E)[GN], uint32_t& pend, int& nlib) {
  pend = 0; nlib = 0;
  uint32_t prev = 0, prev2 = 0;  // the rows one and two steps back in sweep order, final
#pragma unroll
  for (int k = 0; k < GN; ++k) {
    const int i = DOWN ? k : GN - 1 - k;
    const int ip = k > 0 ? (DOWN ? i - 1 : i + 1) : i;       // the row before this one in sweep order
    const uint32_t m = M[i];
    const uint32_t x = F[i] | (prev & m);
    const uint32_t up = ((m + x) ^ m) & m;
    const uint32_t mr = __brev(m), xr = __brev(x);
    const uint32_t dn = ((mr + xr) ^ mr) & mr;
    const uint32_t g = x | up | __brev(dn);
    F[i] = g;
    if (k > 0) {
      pend |= g & M[ip] & ~prev;
      if (EVAL) GO_POPC_ACC(nlib, ((prev << 1) | (prev >> 1) | prev2 | g) & E[ip]);
    }
    prev2 = prev; prev = g;
  }
  if (EVAL) GO_POPC_ACC(nlib, ((prev << 1) | (prev >> 1) | prev2) & E[DOWN ? GN - 1 : 0]);
}

// grow F (subset of M) to the complete M-components it touches; every lane of the wave takes part
// The seeds run straight up their columns first (one instruction per row): the first pass sweeps DOWN, so whatever a
// seed reaches by going up a column and then sideways or down is found in that pass already instead of in the next
// (richer pre-passes - a sideways step per row, or the exact row flood on the way up - cost more than they save).
__device__ inline void go_seed_up(uint32_t (&F)[GN], const uint32_t (&M)[GN]) {
#pragma unroll
  for (int i = GN - 2; i >= 0; --i) F[i] |= F[i + 1] & M[i];
}

__device__ inline void go_fill_complete(uint32_t (&F)[GN], const uint32_t (&M)[GN]) {
  PROF_INC(20, 1);
  go_seed_up(F, M);
  for (;;) {
    uint32_t pend; int nl;
    PROF_INC(21, 1);
    go_grow<false, true>(F, M, M, pend, nl);
    if (!__any(pend != 0)) break;
    PROF_INC(21, 1);
    go_grow<false, false>(F, M, M, pend, nl);
    if (!__any(pend != 0)) break;
  }
}

// Liberty status of the group that contains the seed F (subset of M), after a move next to it:
// AT must end up set on the whole group iff it has exactly one liberty.  The group is only grown as
// far as the answer needs: as soon as the part found so far already touches two empty points the
// group cannot be in atari, and if none of it carried an AT bit before (need_full == false) there is
// nothing to rewrite, so the lane is done without ever seeing the rest of a large chain.  Lanes that
// must set or clear bits (atari, or a group leaving atari) grow to the complete group.
// size / libs are exact when the group was completed; libs = 2 (meaning "at least two") otherwise.
__device__ inline void go_group_status(GoS& s, uint32_t (&F)[GN], const uint32_t (&M)[GN], const uint32_t (&E)[GN],
                                       bool need_full, int& size, int& libs) {
  bool fin = false, noop = false;
  int nlib = 0;
  PROF_INC(20, 1);
  go_seed_up(F, M);
  for (;;) {
    uint32_t pend;
    PROF_INC(21, 1);
    go_grow<true, true>(F, M, E, pend, nlib);
    if (!fin) {
      if (pend == 0) fin = true;
      else if (nlib >= 2 && !need_full) { fin = true; noop = true; }
    }
    if (!__any(!fin)) break;
    PROF_INC(21, 1);
    go_grow<true, false>(F, M, E, pend, nlib);
    if (!fin) {
      if (pend == 0) fin = true;
      else if (nlib >= 2 && !need_full) { fin = true; noop = true; }
    }
    if (!__any(!fin)) break;
  }
  // a lane that finished by completion re-evaluates the same complete F on later passes: nlib stays exact
  size = 0;
  if (!noop) {
    const bool in_atari = nlib == 1;
    ROWS(i) { s.AT[i] = (s.AT[i] & ~F[i]) | (in_atari ? F[i] : 0u); size += __popc(F[i]); }
    libs = nlib;
  } else {
    libs = 2;
  }
}

// Play a stone for the side to move at row r / column bit `bit` (must be legal) when act is true;
// lanes with act == false run the same instructions with empty seeds and change nothing.
// Updates stones, AT, hash, ko, capP.  z = Zobrist table in LDS ([colour*361 + point]).
// hash_ready(hash) is called once the new position's hash is final (captures and the stone are in),
// before the liberty bookkeeping: the caller may start loading whatever the superko test will need.
template <class HR>
__device__ inline void go_place(GoS& s, bool act, uint32_t r, uint32_t bit, const uint64_t* z, HR hash_ready) {
  if (!act) bit = 0;
  uint32_t ncap = 0, cap_pt = 0;
  PROF_T0;
  uint32_t F[GN];
  // the stone as a one-hot row set, built once (one compare per row); the loops below shift it by a row or a
  // column instead of comparing the row index again (one wave per SIMD: registers are not the constraint)
  uint32_t ST[GN];
  ROWS(i) ST[i] = (uint32_t)i == r ? bit : 0u;
  // ---- captures: opponent groups in atari next to the stone (their one liberty is this point) ----
  uint32_t own_at = 0;      // own stones in atari next to the point
  ROWS(i) {
    const uint32_t s0 = ST[i], su = i + 1 < GN ? ST[i + 1 < GN ? i + 1 : 0] : 0u, sd = i > 0 ? ST[i > 0 ? i - 1 : 0] : 0u;
    const uint32_t nbr = ((s0 << 1) | (s0 >> 1) | su | sd) & FULL;
    F[i] = nbr & s.Q[i] & s.AT[i];
    own_at |= nbr & s.P[i] & s.AT[i];
  }
  const bool own_at_nbr = own_at != 0;  // an own group in atari touches the point: it merges and must be rewritten
  const bool any_cap = go_any_rows(F);
  PROF_ADD(0);
  if (any_cap) {
    PROF_INC(10, 1);
    go_fill_complete(F, s.Q);
    PROF_ADD(1);
    uint32_t R[GN];
    ROWS(i) {
      s.Q[i] &= ~F[i];
      s.AT[i] &= ~F[i];
      ncap += __popc(F[i]);
      if (F[i]) cap_pt = i * 19 + (__ffs(F[i]) - 1);
      if (__any(F[i] != 0)) {
        const uint64_t* t = z + GO_ZC_BASE + ((s.tm ^ 1) * GN + i) * GO_ZC_ROW;
        const uint32_t f = F[i];
        s.hash ^= t[f & 15] ^ t[16 + ((f >> 4) & 15)] ^ t[32 + ((f >> 8) & 15)] ^ t[48 + ((f >> 12) & 15)] ^ t[64 + (f >> 16)];
      }
    }
    // own groups in atari next to the captured stones gain a liberty: they leave AT
    ROWS(i) R[i] = NBR(F, i) & s.P[i] & s.AT[i];
    PROF_ADD(2);
    if (go_any_rows(R)) {
      PROF_INC(11, 1);
      go_fill_complete(R, s.P);
      ROWS(i) s.AT[i] &= ~R[i];
      PROF_ADD(1);
    }
  }
  // ---- the stone itself ----
  uint32_t E[GN];
  ROWS(i) {
    const uint32_t s0 = ST[i];
    s.P[i] |= s0;
    F[i] = s0;
    E[i] = ~(s.P[i] | s.Q[i]) & FULL;
  }
  if (act) { s.hash ^= z[s.tm * GO_POINTS + r * 19 + (__ffs(bit) - 1)]; hash_ready(s.hash); }
  int own_size = 0, own_libs = 0;
  PROF_ADD(0);
  PROF_INC(12, 1);
  go_group_status(s, F, s.P, E, own_at_nbr, own_size, own_libs);
  PROF_ADD(3);
  // ---- the (up to four) opponent groups that lost this liberty.  None of them carries an AT bit:
  //      a neighbour in atari was just captured. ----
#ifdef GO_FIXED_DIRECTIONS   // A/B only: one job per direction, whatever the lanes hold there
  const bool crowded = !__any(act && s.ply < 48);
#pragma unroll 1
  for (int d = 0; d < 4; ++d) {  // 0 up, 1 down, 2 left, 3 right
    if (d == 0) { ROWS(i) F[i] = (i + 1 < GN ? ST[i + 1 < GN ? i + 1 : 0] : 0u) & s.Q[i]; }
    else if (d == 1) { ROWS(i) F[i] = (i > 0 ? ST[i > 0 ? i - 1 : 0] : 0u) & s.Q[i]; }
    else if (d == 2) { ROWS(i) F[i] = (ST[i] >> 1) & s.Q[i]; }
    else { ROWS(i) F[i] = (ST[i] << 1) & s.Q[i]; }   // Q has no bits above column 18
    const bool any = crowded || go_any_rows(F);
    PROF_ADD(0);
    if (!any) continue;
    PROF_INC(13 + d, 1);
    int sz, lb;
    go_group_status(s, F, s.Q, E, false, sz, lb);
    PROF_ADD(3);
  }
#else
  // Every lane walks ITS OWN neighbours (up, down, left, right, the ones that hold an opponent stone, in that
  // order): the wave runs as many jobs as its busiest lane has opponent neighbours - one or two early in a game,
  // three late (a point with four is hardly ever legal) - instead of one per direction.  A direction without a
  // stone was a job that changed nothing, so the boards come out the same.
  uint32_t q_up = 0, q_dn = 0, q_0 = 0;
  ROWS(i) {
    q_up |= (i + 1 < GN ? ST[i + 1 < GN ? i + 1 : 0] : 0u) & s.Q[i];   // row r - 1, the stone's column
    q_dn |= (i > 0 ? ST[i > 0 ? i - 1 : 0] : 0u) & s.Q[i];             // row r + 1
    q_0 = (uint32_t)i == r ? s.Q[i] : q_0;
  }
  const bool e0 = q_up != 0, e1 = q_dn != 0, e2 = (q_0 & (bit >> 1)) != 0, e3 = (q_0 & (bit << 1)) != 0;
  const int c1 = e0 ? 1 : 0, c2 = c1 + (e1 ? 1 : 0), c3 = c2 + (e2 ? 1 : 0), cnt = c3 + (e3 ? 1 : 0);
  PROF_ADD(0);
#pragma unroll 1
  for (int j = 0; j < 4; ++j) {
    if (!__any(cnt > j)) break;
    const bool t0 = e0 && j == 0, t1 = e1 && c1 == j, t2 = e2 && c2 == j, t3 = e3 && c3 == j;
    const uint32_t rs = t0 ? r - 1u : (t1 ? r + 1u : r);
    const uint32_t bs = (t0 || t1) ? bit : (t2 ? bit >> 1 : (t3 ? bit << 1 : 0u));
    ROWS(i) F[i] = (uint32_t)i == rs ? bs : 0u;
    PROF_ADD(0);
    PROF_INC(13 + j, 1);
    int sz, lb;
    go_group_status(s, F, s.Q, E, false, sz, lb);
    PROF_ADD(3);
  }
#endif
  if (act) {
    s.capP += ncap;
    s.ko = (ncap == 1 && own_size == 1 && own_libs == 1) ? cap_pt : NO_POINT;
  }
}

// Positional superko set of one game: open addressing over 64-byte buckets of four {hash, game
// generation} entries; an entry of another generation is free (reset = generation + 1, O(1)).  A key
// lives in the first bucket of its probe sequence that had a free entry when it was inserted, so a
// lookup ends at the first bucket that holds the key or still has a free entry.  One probe = one
// memory round trip for four entries; the first bucket can be loaded as soon as the hash is known
// (go_place's prefetch hook) and checked after the liberty bookkeeping has hidden its latency.
struct GoBucket { ulonglong2 e[4]; };
__device__ inline uint32_t go_bucket_of(uint64_t h) { return (uint32_t)(h >> 40) & (TAB_BUCKETS - 1); }
__device__ inline void go_bucket_load(const ulonglong2* tab, uint32_t b, GoBucket& k) {
#pragma unroll
  for (int j = 0; j < 4; ++j) k.e[j] = tab[(size_t)b * 4 + j];
}
// 1: h present; 0: absent and entry `free` of the bucket is free; -1: bucket full of other keys
__device__ inline int go_bucket_probe(const GoBucket& k, uint64_t gen, uint64_t h, int& free) {
  int r = -1; free = 0;
#pragma unroll
  for (int j = 3; j >= 0; --j) {
    const bool live = k.e[j].y == gen;
    if (!live) { r = r == 1 ? 1 : 0; free = j; }
    if (live && k.e[j].x == h) r = 1;
  }
  return r;
}
// returns 1 if h was already present, else inserts it.  first = the prefetched bucket go_bucket_of(h), or null
__device__ inline uint32_t go_table_test_and_insert(ulonglong2* tab, uint64_t gen, uint64_t h, const GoBucket* first = nullptr) {
  uint32_t b = go_bucket_of(h);
  for (int probe = 0; probe < TAB_BUCKETS; ++probe) {
    GoBucket k;
    if (probe == 0 && first) k = *first; else go_bucket_load(tab, b, k);
    int free;
    const int r = go_bucket_probe(k, gen, h, free);
    if (r == 1) return 1;
    if (r == 0) { tab[(size_t)b * 4 + free] = make_ulonglong2(h, gen); return 0; }
    b = (b + 1) & (TAB_BUCKETS - 1);
  }
  return 0;
}

// One full forward() for this lane's board: a = point index or GO_PASS; act = lane takes part.
// The caller has verified legality.  Swaps P/Q so that P is again the side to move.
// superko(hash) -> 0/1 decides whether the new position repeats an earlier one of this game.
template <class HR, class SK>
__device__ inline void go_forward_t(GoS& s, bool act, uint32_t a, const uint64_t* z, HR hash_ready, SK superko) {
  const bool is_pass = a >= GO_POINTS;
  const bool pt_move = act && !is_pass;
  const uint32_t r = pt_move ? a / 19u : 0u, bit = pt_move ? 1u << (a % 19u) : 0u;
  go_place(s, pt_move, r, bit, z, hash_ready);
  PROF_T0;
  if (act) {
    if (is_pass) { s.passes += 1; s.ko = NO_POINT; }
    else { s.passes = 0; s.superko = superko(s.hash); }
    s.ply += 1; s.last = a; s.tm ^= 1;
    ROWS(i) { uint32_t t = s.P[i]; s.P[i] = s.Q[i]; s.Q[i] = t; }
    uint32_t t = s.capP; s.capP = s.capQ; s.capQ = t;
  }
  PROF_ADD(6);
}
// the board slab's version: the game's own superko table, test and insert
__device__ inline void go_forward(GoS& s, bool act, uint32_t a, ulonglong2* tab, uint64_t gen, const uint64_t* z) {
  GoBucket first;
  go_forward_t(s, act, a, z, [&](uint64_t h) { go_bucket_load(tab, go_bucket_of(h), first); },
               [&](uint64_t h) { return go_table_test_and_insert(tab, gen, h, &first); });
}

// ---- SoA storage: rows [57][n] (B, W, AT), meta [4][n], hash [n], gen [n], table [n][1024] -------
struct GoDev {
  uint32_t* rows; uint32_t* meta; uint64_t* hash; uint64_t* gen; ulonglong2* table; int n;
  uint32_t* hist;   // [8 slots][38 rows (19 black, 19 white)][n]: stone snapshots for the 18 planes
  uint32_t* hmeta;  // [n]: len (1..8) | head << 4   (ring; newest = (head + len - 1) % 8)
};

// stone history of GoState (8 most recent positions, oracle/go.c push_stones)
__device__ inline void go_hist_write(const GoDev& d, int b, int slot, const GoS& s) {
  ROWS(i) {
    d.hist[((size_t)slot * 38 + i) * d.n + b] = s.tm ? s.Q[i] : s.P[i];
    d.hist[((size_t)slot * 38 + 19 + i) * d.n + b] = s.tm ? s.P[i] : s.Q[i];
  }
}
__device__ inline void go_hist_set_single(const GoDev& d, int b, const GoS& s) {
  go_hist_write(d, b, 0, s);
  d.hmeta[b] = 1u;
}
__device__ inline void go_hist_push(const GoDev& d, int b, const GoS& s) {
  uint32_t m = d.hmeta[b];
  int len = m & 15, head = (m >> 4) & 7, slot;
  if (len < 8) { slot = (head + len) & 7; ++len; } else { slot = head; head = (head + 1) & 7; }
  go_hist_write(d, b, slot, s);
  d.hmeta[b] = (uint32_t)len | ((uint32_t)head << 4);
}

__device__ inline void go_load(const GoDev& d, int b, GoS& s) {
  const size_t n = d.n;
  uint32_t m0 = d.meta[0 * n + b], m1 = d.meta[1 * n + b];
  s.tm = m0 & 1; s.passes = (m0 >> 1) & 3; s.superko = (m0 >> 3) & 1; s.ply = m0 >> 16;
  s.ko = m1 & 0xFFFF; s.last = m1 >> 16;
  uint32_t cb = d.meta[2 * n + b], cw = d.meta[3 * n + b];
  s.capP = s.tm ? cw : cb; s.capQ = s.tm ? cb : cw;
  s.hash = d.hash[b];
  ROWS(i) {
    uint32_t bl = d.rows[(size_t)i * n + b], wh = d.rows[(size_t)(19 + i) * n + b];
    s.P[i] = s.tm ? wh : bl; s.Q[i] = s.tm ? bl : wh;
    s.AT[i] = d.rows[(size_t)(38 + i) * n + b];
  }
}

__device__ inline void go_store(const GoDev& d, int b, const GoS& s) {
  const size_t n = d.n;
  d.meta[0 * n + b] = s.tm | (s.passes << 1) | (s.superko << 3) | (s.ply << 16);
  d.meta[1 * n + b] = (s.ko & 0xFFFF) | (s.last << 16);
  d.meta[2 * n + b] = s.tm ? s.capQ : s.capP;
  d.meta[3 * n + b] = s.tm ? s.capP : s.capQ;
  d.hash[b] = s.hash;
  ROWS(i) {
    d.rows[(size_t)i * n + b] = s.tm ? s.Q[i] : s.P[i];
    d.rows[(size_t)(19 + i) * n + b] = s.tm ? s.P[i] : s.Q[i];
    d.rows[(size_t)(38 + i) * n + b] = s.AT[i];
  }
}

__device__ inline void go_load_zobrist(uint64_t* z, const uint64_t* gz) {
  for (int i = threadIdx.x; i < 2 * GO_POINTS; i += blockDim.x) z[i] = gz[i];
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * GN * GO_ZC_ROW; e += blockDim.x) {
    const int v = e & 15, chunk = (e >> 4) % 5, cr = e / GO_ZC_ROW;  // cr = colour * 19 + row
    uint64_t x = 0;
    for (int b = 0; b < 4; ++b) {
      const int col = chunk * 4 + b;
      if (((v >> b) & 1) && col < GN) x ^= z[(cr / GN) * GO_POINTS + (cr % GN) * GN + col];
    }
    z[GO_ZC_BASE + e] = x;
  }
  __syncthreads();
}

// AT from scratch (import): label every group of both colours
__device__ inline void go_recompute_at(GoS& s) {
  ROWS(i) s.AT[i] = 0;
#pragma unroll 1
  for (int col = 0; col < 2; ++col) {
    uint32_t rem[GN];
    ROWS(i) rem[i] = col ? s.Q[i] : s.P[i];
    while (go_any_rows(rem)) {
      uint32_t F[GN];
      bool found = false;
      ROWS(i) {
        uint32_t lb = rem[i] & (0u - rem[i]);
        F[i] = found ? 0u : lb;
        found = found || rem[i] != 0;
      }
      if (col) go_fill(F, s.Q); else go_fill(F, s.P);
      int libs = go_libs(F, s);
      ROWS(i) { s.AT[i] |= (libs == 1) ? F[i] : 0u; rem[i] &= ~F[i]; }
    }
  }
}

// rows -> 362-bit action mask (bit r*19+c; bit 361 = pass)
__device__ inline void go_pack_mask(const uint32_t (&L)[GN], bool alive, uint64_t (&m)[6]) {
#pragma unroll
  for (int k = 0; k < 6; ++k) m[k] = 0;
  ROWS(i) {
    const int base = i * 19, w = base >> 6, o = base & 63;
    uint64_t v = alive ? (uint64_t)L[i] : 0ull;
    m[w] |= v << o;
    if (o + 19 > 64) m[w + 1] |= v >> (64 - o);
  }
  if (alive) m[GO_PASS >> 6] |= 1ull << (GO_PASS & 63);
}


// read-only membership test of the positional-superko set (tree search: a node must not insert)
__device__ inline uint32_t go_table_contains(const ulonglong2* tab, uint64_t gen, uint64_t h) {
  uint32_t b = go_bucket_of(h);
  for (int probe = 0; probe < TAB_BUCKETS; ++probe) {
    GoBucket k;
    go_bucket_load(tab, b, k);
    int free;
    const int r = go_bucket_probe(k, gen, h, free);
    if (r >= 0) return (uint32_t)r;
    b = (b + 1) & (TAB_BUCKETS - 1);
  }
  return 0;
}

}  // namespace godev
