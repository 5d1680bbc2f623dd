/* TEST INFRASTRUCTURE ONLY (oracle) - see ugolki.h.  Move-list restatement (one {bits, flags} entry per
 * move, looked up in a table), deliberately unlike the packed-field mask assembly of
 * elf_amd/csrc/ugolki.hip. */
#include "ugolki.h"
#include <string.h>

uint64_t elf_splitmix64(uint64_t x);

#define WHITE_BASE 0xE0E0E00000000000ull
#define BLACK_BASE 0x0000000000070707ull
#define BLACK 0
#define WHITE 1

static uint64_t g_bits[UG_NUM_ACTIONS], g_flags[UG_NUM_ACTIONS];
static int g_n = 0;

/* table rule (verified against the reference's i_to_m): eight groups in this order, each ascending in
 * the low bit i of the two-bit pattern: 0x3<<i (x<7) flags 1, 0; 0x101<<i (i<56) flags 1, 0;
 * 0x5<<i (x<6) flags 3, 2; 0x10001<<i (i<48) flags 3, 2; then {0, 0} = pass */
static void build(void) {
  if (g_n) return;
  static const uint64_t pat[4] = {0x3, 0x101, 0x5, 0x10001};
  int n = 0;
  for (int g = 0; g < 8; ++g) {
    int p = g >> 1, jump = g >= 4, dir = !(g & 1);
    for (int i = 0; i < 64; ++i) {
      int ok = p == 0 ? (i % 8 < 7) : p == 1 ? (i < 56) : p == 2 ? (i % 8 < 6) : (i < 48);
      if (!ok) continue;
      g_bits[n] = pat[p] << i; g_flags[n] = (uint64_t)(dir | (jump ? 2 : 0)); ++n;
    }
  }
  g_bits[n] = 0; g_flags[n] = 0; ++n;
  g_n = n;  /* 417 */
}

int ug_action(int a, uint64_t* bits, uint64_t* flags) {
  build();
  if (a < 0 || a >= g_n) return 0;
  *bits = g_bits[a]; *flags = g_flags[a];
  return 1;
}

static int find_action(uint64_t bits, uint64_t flags) {
  for (int a = 0; a < g_n; ++a) if (g_bits[a] == bits && g_flags[a] == flags) return a;
  return -1;
}

void ug_clear(ug_state* s) {
  s->w[0] = WHITE_BASE; s->w[1] = BLACK_BASE; s->w[2] = 0;
  s->w[3] = (uint64_t)BLACK | (1ull << 8);
}

static int active_of(const ug_state* s) { return (int)(s->w[3] & 1); }
static int ply_of(const ug_state* s) { return (int)((s->w[3] >> 8) & 0xFFFF); }
static int bwin_of(const ug_state* s) { return (int)((s->w[3] >> 1) & 7); }
static int wwin_of(const ug_state* s) { return (int)((s->w[3] >> 4) & 7); }

typedef struct { int n; uint64_t bits[256], flags[256]; } ML;
static void push(ML* l, uint64_t b, uint64_t f) { l->bits[l->n] = b; l->flags[l->n] = f; l->n++; }
static uint64_t dirn(uint64_t move, uint64_t pieces) { return pieces < (pieces ^ move); }  /* :357-362 */

/* one square at a time; a piece on (y,x) steps/jumps along the four axes */
static void gen(const ug_state* s, uint64_t pieces, uint64_t invalid_move, int steps, int jumps, ML* l) {
  uint64_t all = s->w[0] | s->w[1];
  uint64_t all2 = all | invalid_move;
  /* order of the reference: right, left, forward, backward moves, then the same for jumps (:421-494) */
  static const int dx[4] = {1, -1, 0, 0}, dy[4] = {0, 0, 1, -1};
  static const uint64_t pat[4] = {0x3, 0x3, 0x101, 0x101}, jpat[4] = {0x5, 0x5, 0x10001, 0x10001};
  if (steps)
    for (int k = 0; k < 4; ++k)
      for (int i = 0; i < 64; ++i) {
        if (!((pieces >> i) & 1)) continue;
        int y = i / 8 + dy[k], x = i % 8 + dx[k];
        if (x < 0 || x > 7 || y < 0 || y > 7) continue;
        int t = y * 8 + x;
        if ((all >> t) & 1) continue;
        uint64_t mv = pat[k] << (i < t ? i : t);
        push(l, mv, dirn(mv, pieces));
      }
  if (jumps)
    for (int k = 0; k < 4; ++k)
      for (int i = 0; i < 64; ++i) {
        if (!((pieces >> i) & 1)) continue;
        int y = i / 8 + 2 * dy[k], x = i % 8 + 2 * dx[k];
        if (x < 0 || x > 7 || y < 0 || y > 7) continue;
        int t = y * 8 + x, mid = (i + t) / 2;
        if ((all2 >> t) & 1) continue;
        if (!((all2 >> mid) & 1)) continue;
        uint64_t mv = jpat[k] << (i < t ? i : t);
        push(l, mv, dirn(mv, pieces) | 2u);
      }
}

/* _jumps_from :499-567: direction is computed against ALL active pieces, jumps start from the pawn only */
static void jumps_from(const ug_state* s, uint64_t pieces, uint64_t jump_action, ML* l) {
  uint64_t pawn = pieces & jump_action, invalid = jump_action ^ pawn;
  ML t; t.n = 0;
  gen(s, pawn, invalid, 0, 1, &t);
  for (int i = 0; i < t.n; ++i) push(l, t.bits[i], dirn(t.bits[i], pieces) | 2u);
  if (t.n) push(l, 0, 0);
}

static uint64_t restricted(const ug_state* s) {   /* base-leaving rule :374-383 / :167-177 */
  int a = active_of(s);
  if (a == BLACK && !(s->w[WHITE] & BLACK_BASE) && (s->w[BLACK] & WHITE_BASE)) return s->w[BLACK] & WHITE_BASE;
  if (a == WHITE && !(s->w[BLACK] & WHITE_BASE) && (s->w[WHITE] & BLACK_BASE)) return s->w[WHITE] & BLACK_BASE;
  return s->w[a];
}

static void legal_moves(const ug_state* s, ML* l) {   /* get_legal_moves :367-384 */
  l->n = 0;
  if (s->w[2] != 0) { jumps_from(s, s->w[active_of(s)], s->w[2], l); return; }
  gen(s, restricted(s), 0, 1, 1, l);
}

void ug_legal_mask(const ug_state* s, uint64_t m[7]) {
  build();
  ML l; legal_moves(s, &l);
  memset(m, 0, 7 * sizeof(uint64_t));
  for (int i = 0; i < l.n; ++i) { int a = find_action(l.bits[i], l.flags[i]); m[a >> 6] |= 1ull << (a & 63); }
}

static int ctz64(uint64_t v) { return __builtin_ctzll(v); }   /* log(x)/log(2) of :198-199; exact for 2^k, k<64 (checked) */

int ug_try_play(const ug_state* s, int c) {
  build();
  if (c < 0 || c >= g_n) return 0;   /* the reference dereferences end() here: undefined, never legal */
  uint64_t action = g_bits[c], direction = g_flags[c] & 1, jump_move = g_flags[c] & 2;
  uint64_t all = s->w[0] | s->w[1], empty = ~all, active_pieces;
  uint64_t ja = s->w[2];
  if (ja != 0) {
    if (c == UG_PASS) return 1;
    uint64_t old_place = ja ^ (all & ja);
    if ((ja & action) == 0) return 0;
    if ((old_place & (ja & action)) != 0) return 0;
    active_pieces = s->w[active_of(s)];
    all = all | old_place; empty = ~all;
  } else active_pieces = restricted(s);
  uint64_t pawn = active_pieces & action;
  if (!pawn) return 0;
  if ((action ^ (pawn & action)) == 0) return 0;
  if (dirn(action, active_pieces) != direction) return 0;
  if (!jump_move) {
    if (ja != 0) return 0;
    return (empty & (action ^ pawn)) != 0;
  }
  int start = ctz64(pawn), dest = ctz64(action ^ pawn);
  if (!(empty & (1ull << dest))) return 0;
  if (direction) return (all & (1ull << (start + (dest - start) / 2))) != 0;
  return (all & (1ull << (dest + (start - dest) / 2))) != 0;
}

void ug_play(ug_state* s, int a) {
  build();
  uint64_t action = g_bits[a], jump = g_flags[a] & 2;
  int active = active_of(s), bw = bwin_of(s), ww = wwin_of(s), ply = ply_of(s) + 1;
  if (jump) s->w[2] = action;
  s->w[active] ^= action;
  if ((s->w[BLACK] & BLACK_BASE) == BLACK_BASE) bw++;
  else if ((s->w[WHITE] & WHITE_BASE) == WHITE_BASE) ww++;
  else { bw = 0; ww = 0; }
  int stay = 0;
  if (jump) { ML l; l.n = 0; jumps_from(s, s->w[active], s->w[2], &l); stay = l.n != 0; }
  if (!stay) { s->w[2] = 0; active = 1 - active; }
  s->w[3] = (uint64_t)active | ((uint64_t)(bw & 7) << 1) | ((uint64_t)(ww & 7) << 4) | ((uint64_t)(ply & 0xFFFF) << 8);
}

int ug_is_over(const ug_state* s) {
  int bw = bwin_of(s), ww = wwin_of(s);
  return (bw > 0 && ww > 0) || bw == 2 || ww == 2;
}

void ug_observation(const ug_state* s, int player, int8_t out[64]) {
  memset(out, 0, 64);
  for (int i = 0; i < 64; ++i) {
    int y = i / 8, x = i % 8, v = ((s->w[BLACK] >> i) & 1) ? 1 : ((s->w[WHITE] >> i) & 1) ? -1 : 0;
    if (!v) continue;
    if (player == BLACK) out[y * 8 + x] = (int8_t)v; else out[(7 - y) * 8 + (7 - x)] = (int8_t)-v;
  }
}

int ug_terminated(const ug_state* s) { return ug_is_over(s) || ply_of(s) >= UG_MAX_PLY; }

int ug_forward(ug_state* s, int a) {
  if (ug_terminated(s) || a < 0 || a > UG_PASS) return 0;
  if (!ug_try_play(s, a)) return 0;
  ug_play(s, a);
  return 1;
}

float ug_evaluate(const ug_state* s) {
  if (!ug_terminated(s)) return 0.f;
  if (ply_of(s) >= UG_MAX_PLY) return -1.f;
  if (bwin_of(s) > 0 && wwin_of(s) > 0) return -1.f;
  return bwin_of(s) == 2 ? 1.f : -1.f;
}

void ug_features(const ug_state* s, float out[16 * 64]) {
  memset(out, 0, sizeof(float) * 16 * 64);
  int a = active_of(s);
  int8_t o[64];
  ug_observation(s, a, o);
  for (int j = 0; j < 64; ++j) if (o[j] == 1) out[j] = 1.f;
  ug_observation(s, 1 - a, o);
  for (int j = 0; j < 64; ++j) if (o[j] == 1) out[64 + j] = 1.f;
  for (int j = 0; j < 64; ++j) out[(a == BLACK ? 2 : 3) * 64 + j] = 1.f;
}

uint64_t ug_playout(ug_state* boards, int n, uint64_t board_id0, uint64_t step_base, int nply, uint64_t seed) {
  uint64_t moves = 0;
  for (int b = 0; b < n; ++b) {
    ug_state* s = &boards[b];
    for (int t = 0; t < nply; ++t) {
      uint64_t m[7]; ug_legal_mask(s, m);
      int cnt = 0;
      for (int i = 0; i < 7; ++i) cnt += __builtin_popcountll(m[i]);
      if (cnt == 0 || ug_terminated(s)) { ug_clear(s); continue; }
      uint64_t r = elf_splitmix64(seed + (board_id0 + (uint64_t)b) * 0x9E3779B97F4A7C15ull + step_base + (uint64_t)t);
      int k = (int)(r % (uint64_t)cnt), a;
      for (a = 0; a < UG_NUM_ACTIONS; ++a) if ((m[a >> 6] >> (a & 63)) & 1) { if (k == 0) break; --k; }
      ug_play(s, a);
      ++moves;
    }
  }
  return moves;
}
