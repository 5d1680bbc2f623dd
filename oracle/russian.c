/* TEST INFRASTRUCTURE ONLY (oracle) - see russian.h.  Straight 8x8-array restatement (like the
 * reference), deliberately unlike the bitboard ray formulation of elf_amd/csrc/russian.hip. */
#include "russian.h"
#include <string.h>

uint64_t elf_splitmix64(uint64_t x);

#define UP 1
#define DOWN (-1)
#define LEFT (-1)
#define RIGHT 1
#define WP 1
#define WK 3
#define BP (-1)
#define BK (-3)

typedef struct { int b[8][8]; int white; int ny, nx; int ply, last; } B;

static int g_from[RU_NUM_ACTIONS], g_to[RU_NUM_ACTIONS], g_idx[64][64], g_ready = 0;

/* table rule (verified against the reference's i_to_m): directions (-1,+1), (-1,-1), (+1,-1), (+1,+1);
 * distance 1..7; from-square descending over the dark squares */
static void build(void) {
  if (g_ready) return;
  static const int dy[4] = {-1, -1, 1, 1}, dx[4] = {1, -1, -1, 1};
  memset(g_idx, -1, sizeof(g_idx));
  int n = 0;
  for (int k = 0; k < 4; ++k)
    for (int d = 1; d < 8; ++d)
      for (int s = 63; s >= 0; --s) {
        int y = s / 8, x = s % 8;
        if ((y + x) % 2 == 0) continue;
        int ty = y + dy[k] * d, tx = x + dx[k] * d;
        if (ty < 0 || ty > 7 || tx < 0 || tx > 7) continue;
        g_from[n] = s; g_to[n] = ty * 8 + tx; g_idx[s][ty * 8 + tx] = n; ++n;
      }
  g_ready = 1;
}

void ru_action(int a, int* from, int* to) { build(); *from = g_from[a]; *to = g_to[a]; }

static void load(const ru_state* s, B* b) {
  for (int y = 0; y < 8; ++y)
    for (int x = 0; x < 8; ++x) {
      uint64_t bit = 1ull << (y * 8 + x);
      b->b[y][x] = (s->w[0] & bit) ? WP : (s->w[1] & bit) ? WK : (s->w[2] & bit) ? BP : (s->w[3] & bit) ? BK : 0;
    }
  b->white = (int)(s->w[4] & 1);
  int has = (int)((s->w[4] >> 1) & 1);
  b->ny = has ? (int)((s->w[4] >> 2) & 7) : -1;
  b->nx = has ? (int)((s->w[4] >> 5) & 7) : -1;
  b->ply = (int)((s->w[4] >> 8) & 0xFFFF);
  b->last = (int)((s->w[4] >> 24) & 0xFFFF);
}

static void store(const B* b, ru_state* s) {
  s->w[0] = s->w[1] = s->w[2] = s->w[3] = 0;
  for (int y = 0; y < 8; ++y)
    for (int x = 0; x < 8; ++x) {
      uint64_t bit = 1ull << (y * 8 + x);
      int v = b->b[y][x];
      if (v == WP) s->w[0] |= bit; else if (v == WK) s->w[1] |= bit; else if (v == BP) s->w[2] |= bit; else if (v == BK) s->w[3] |= bit;
    }
  uint64_t has = b->ny != -1;
  s->w[4] = (uint64_t)(b->white & 1) | (has << 1) | ((uint64_t)(has ? b->ny : 0) << 2) | ((uint64_t)(has ? b->nx : 0) << 5) |
            ((uint64_t)(b->ply & 0xFFFF) << 8) | ((uint64_t)(b->last & 0xFFFF) << 24);
}

void ru_clear(ru_state* s) {
  B b; memset(&b, 0, sizeof(b));
  b.white = 0; b.ply = 1; b.last = RU_M_INVALID; b.ny = b.nx = -1;
  for (int y = 0; y < 8; ++y)
    for (int x = 0; x < 8; ++x)
      b.b[y][x] = ((y + x) % 2 == 0) ? 0 : (y < 3 ? WP : (y > 4 ? BP : 0));
  store(&b, s);
}

static int ovf(int v) { return v < 0 || v > 7; }
typedef struct { int n; int from[64], to[64]; } ML;
static void push(ML* l, int f, int t) { l->from[l->n] = f; l->to[l->n] = t; l->n++; }

static int is_enemy(const B* b, int v) { return b->white ? (v == BK || v == BP) : (v == WK || v == WP); }

/* _pawnJumpInDirection :593-628 */
static void pawn_jumps(const B* b, int y, int x, ML* l) {
  static const int dy[4] = {UP, UP, DOWN, DOWN}, dx[4] = {LEFT, RIGHT, LEFT, RIGHT};
  for (int k = 0; k < 4; ++k) {
    int ey = y + dy[k], ex = x + dx[k];
    if (ovf(ey) || ovf(ex) || !is_enemy(b, b->b[ey][ex])) continue;
    int ty = ey + dy[k], tx = ex + dx[k];
    if (!ovf(ty) && !ovf(tx) && b->b[ty][tx] == 0) push(l, y * 8 + x, ty * 8 + tx);
  }
}

/* _kingJumpCheck :412-453 */
static int king_jump_check(const B* b, int y, int x, int dy, int dx) {
  int ty = y + dy, tx = x + dx;
  for (;;) {
    if (ovf(ty) || ovf(tx)) return 0;
    if (b->b[ty][tx] != 0) break;
    ty += dy; tx += dx;
  }
  if (!is_enemy(b, b->b[ty][tx])) return 0;
  ty += dy; tx += dx;
  return !ovf(ty) && !ovf(tx) && b->b[ty][tx] == 0;
}

/* _kingJumpInDirection :457-530 (board copied by value: the captured piece is lifted locally) */
static void king_jump_dir(const B* b0, int y, int x, int dy, int dx, ML* l) {
  B b = *b0;
  int ty = y, tx = x;
  for (;;) {
    ty += dy; tx += dx;
    if (ovf(ty) || ovf(tx)) return;
    if (b.b[ty][tx] != 0) break;
  }
  if (!is_enemy(&b, b.b[ty][tx])) return;
  b.b[ty][tx] = 0;
  ty += dy; tx += dx;
  int first = l->n;
  for (;;) {
    if (ovf(ty) || ovf(tx) || b.b[ty][tx] != 0) break;
    if (king_jump_check(&b, ty, tx, UP, LEFT) || king_jump_check(&b, ty, tx, UP, RIGHT) ||
        king_jump_check(&b, ty, tx, DOWN, LEFT) || king_jump_check(&b, ty, tx, DOWN, RIGHT)) {
      l->n = first;                      /* jumps.clear(): only this landing square survives */
      push(l, y * 8 + x, ty * 8 + tx);
      break;
    }
    push(l, y * 8 + x, ty * 8 + tx);
    ty += dy; tx += dx;
  }
}

static void king_jumps(const B* b, int y, int x, ML* l) {
  king_jump_dir(b, y, x, UP, LEFT, l); king_jump_dir(b, y, x, UP, RIGHT, l);
  king_jump_dir(b, y, x, DOWN, LEFT, l); king_jump_dir(b, y, x, DOWN, RIGHT, l);
}

static void pawn_moves(const B* b, int y, int x, ML* l) {  /* :317-346 */
  int d = b->white ? UP : DOWN;
  if (!ovf(y + d) && !ovf(x + LEFT) && b->b[y + d][x + LEFT] == 0) push(l, y * 8 + x, (y + d) * 8 + x + LEFT);
  if (!ovf(y + d) && !ovf(x + RIGHT) && b->b[y + d][x + RIGHT] == 0) push(l, y * 8 + x, (y + d) * 8 + x + RIGHT);
}

static void king_moves(const B* b, int y, int x, ML* l) {  /* :348-404 */
  static const int dy[4] = {UP, UP, DOWN, DOWN}, dx[4] = {LEFT, RIGHT, LEFT, RIGHT};
  for (int k = 0; k < 4; ++k) {
    int ty = y + dy[k], tx = x + dx[k];
    while (!ovf(ty) && !ovf(tx) && b->b[ty][tx] == 0) { push(l, y * 8 + x, ty * 8 + tx); ty += dy[k]; tx += dx[k]; }
  }
}

/* getAllMoves :259-273 */
static void all_moves(const B* b, ML* l) {
  l->n = 0;
  if (b->ny != -1) {
    int v = b->b[b->ny][b->nx];
    if (v > 1 || v < -1) king_jumps(b, b->ny, b->nx, l); else pawn_jumps(b, b->ny, b->nx, l);
    return;
  }
  int pawn = b->white ? WP : BP, king = b->white ? WK : BK;
  for (int y = 0; y < 8; ++y)
    for (int x = 0; x < 8; ++x) {
      if (b->b[y][x] == pawn) pawn_jumps(b, y, x, l); else if (b->b[y][x] == king) king_jumps(b, y, x, l);
    }
  if (l->n > 0) return;
  for (int y = 0; y < 8; ++y)
    for (int x = 0; x < 8; ++x) {
      if (b->b[y][x] == pawn) pawn_moves(b, y, x, l); else if (b->b[y][x] == king) king_moves(b, y, x, l);
    }
}

void ru_legal_mask(const ru_state* s, uint64_t m[5]) {
  build();
  B b; load(s, &b);
  ML l; all_moves(&b, &l);
  memset(m, 0, 5 * sizeof(uint64_t));
  for (int i = 0; i < l.n; ++i) { int a = g_idx[l.from[i]][l.to[i]]; m[a >> 6] |= 1ull << (a & 63); }
}

int ru_is_over(const ru_state* s) { B b; load(s, &b); ML l; all_moves(&b, &l); return l.n == 0; }

/* CheckersPlay :36-82 */
void ru_play(ru_state* s, int action) {
  build();
  B b; load(s, &b);
  int ys = g_from[action] / 8, xs = g_from[action] % 8, yd = g_to[action] / 8, xd = g_to[action] % 8;
  int buff = b.b[ys][xs], y = ys, x = xs;
  int dy = ys - yd > 0 ? DOWN : UP, dx = xs - xd < 0 ? RIGHT : LEFT;
  while (y != yd && x != xd) {
    y += dy; x += dx;
    if (b.b[y][x] != 0) {
      b.b[y][x] = 0;
      ML l; l.n = 0;
      /* Q9: continuation is evaluated from the destination BEFORE the piece is moved there */
      if (buff > 1 || buff < -1) king_jumps(&b, yd, xd, &l); else pawn_jumps(&b, yd, xd, &l);
      if (l.n > 0) { b.ny = yd; b.nx = xd; } else { b.ny = -1; b.nx = -1; }
    }
  }
  b.b[yd][xd] = buff;
  b.b[ys][xs] = 0;
  if (b.ny == -1) b.white = !b.white;
  if (b.b[yd][xd] == WP && yd == 7) b.b[yd][xd] = WK;   /* _tryConvertIntoKing :275-280 */
  if (b.b[yd][xd] == BP && yd == 0) b.b[yd][xd] = BK;
  b.ply++; b.last = action;
  store(&b, s);
}

void ru_observation(const ru_state* s, int white, int8_t out[64]) {
  B b; load(s, &b);
  for (int y = 0; y < 8; ++y)
    for (int x = 0; x < 8; ++x) out[y * 8 + x] = (int8_t)(white ? -b.b[7 - y][7 - x] : b.b[y][x]);
}

int ru_terminated(const ru_state* s) { return ru_is_over(s) || (int)((s->w[4] >> 8) & 0xFFFF) >= RU_MAX_PLY; }

int ru_forward(ru_state* s, int a) {
  if (ru_terminated(s) || a < 0 || a >= RU_NUM_ACTIONS) return 0;
  uint64_t m[5]; ru_legal_mask(s, m);
  if (!((m[a >> 6] >> (a & 63)) & 1)) return 0;
  ru_play(s, a);
  return 1;
}

float ru_evaluate(const ru_state* s) {
  if (!ru_terminated(s)) return 0.f;
  if ((int)((s->w[4] >> 8) & 0xFFFF) >= RU_MAX_PLY) return -1.f;
  return (s->w[4] & 1) ? 1.f : -1.f;   /* black to move and stuck: -1; white stuck: +1 */
}

/* CheckersFeature::extract: planes 0/1 = GetObservation(active) == 1 / == 3, 2/3 the same for the
 * passive player, 4/5 = black / white to move (Q8: for black the observation is the raw board, so
 * "own" planes hold the opponent's pieces - reproduced, not fixed) */
void ru_features(const ru_state* s, float out[6 * 64]) {
  memset(out, 0, sizeof(float) * 6 * 64);
  int white = (int)(s->w[4] & 1);
  int8_t oa[64], op[64];
  ru_observation(s, white, oa);
  ru_observation(s, !white, op);
  for (int j = 0; j < 64; ++j) {
    if (oa[j] == 1) out[0 * 64 + j] = 1.f;
    if (oa[j] == 3) out[1 * 64 + j] = 1.f;
    if (op[j] == 1) out[2 * 64 + j] = 1.f;
    if (op[j] == 3) out[3 * 64 + j] = 1.f;
    out[(white ? 5 : 4) * 64 + j] = 1.f;
  }
}

uint64_t ru_playout(ru_state* boards, int n, uint64_t board_id0, uint64_t step_base, int nply, uint64_t seed) {
  uint64_t moves = 0;
  for (int b = 0; b < n; ++b) {
    ru_state* s = &boards[b];
    for (int t = 0; t < nply; ++t) {
      uint64_t m[5]; ru_legal_mask(s, m);
      int cnt = 0;
      for (int i = 0; i < 5; ++i) cnt += __builtin_popcountll(m[i]);
      if (cnt == 0 || (int)((s->w[4] >> 8) & 0xFFFF) >= RU_MAX_PLY) { ru_clear(s); continue; }
      uint64_t r = elf_splitmix64(seed + (board_id0 + (uint64_t)b) * 0x9E3779B97F4A7C15ull + step_base + (uint64_t)t);
      int k = (int)(r % (uint64_t)cnt), a;
      for (a = 0; a < RU_NUM_ACTIONS; ++a) if ((m[a >> 6] >> (a & 63)) & 1) { if (k == 0) break; --k; }
      ru_play(s, a);
      ++moves;
    }
  }
  return moves;
}
