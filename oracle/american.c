/* TEST INFRASTRUCTURE ONLY (oracle) - see american.h.
 * Plain-C restatement of the reference's American checkers board, written from the behaviour of
 * /root/reference/src_cpp/elfgames/american_checkers/game/GameBoard.cc (cited per function).
 * Deliberately straightforward (per-move loops, like the reference) so that it is an independent
 * check of the branch-free HIP kernel in elf_amd/csrc/american.hip.
 */
#include "american.h"
#include <string.h>

#define BLACK 0
#define WHITE 1
/* GameBoard.h:60-61: 36-bit padded layout, bits 8,17,26,35 unused */
static const uint64_t UNUSED = (1ull << 8) | (1ull << 17) | (1ull << 26) | (1ull << 35);
static const uint64_t MASK36 = (1ull << 36) - 1;

/* ---- action table -------------------------------------------------------------------------
 * HashAllMoves.h:193-363 lists, in this order, every two-bit word that fits the padded board:
 *   0x11<<i (25), 0x21<<i (24) with dir=1, the same 49 with dir=0,
 *   -(0x101<<i) (18), -(0x401<<i) (18) with dir=1, the same 36 with dir=0.
 * A word is valid when its two end bits, and for jumps the middle bit, are playable squares. */
static int64_t g_move[AM_NUM_ACTIONS];
static int g_dir[AM_NUM_ACTIONS];
static int g_ready = 0;

static int playable(int bit) { return bit >= 0 && bit < 35 && !((UNUSED >> bit) & 1); }

static void build_table(void) {
  if (g_ready) return;
  int n = 0;
  for (int dir = 1; dir >= 0; --dir) {
    for (int i = 0; i < 36; ++i)
      if (playable(i) && playable(i + 4)) { g_move[n] = (int64_t)(0x11ull << i); g_dir[n++] = dir; }
    for (int i = 0; i < 36; ++i)
      if (playable(i) && playable(i + 5)) { g_move[n] = (int64_t)(0x21ull << i); g_dir[n++] = dir; }
  }
  for (int dir = 1; dir >= 0; --dir) {
    for (int i = 0; i < 36; ++i)
      if (playable(i) && playable(i + 4) && playable(i + 8)) { g_move[n] = -(int64_t)(0x101ull << i); g_dir[n++] = dir; }
    for (int i = 0; i < 36; ++i)
      if (playable(i) && playable(i + 5) && playable(i + 10)) { g_move[n] = -(int64_t)(0x401ull << i); g_dir[n++] = dir; }
  }
  g_ready = (n == AM_NUM_ACTIONS) ? 1 : -1;
}

int64_t am_action_move(int a) { build_table(); return g_move[a]; }
int am_action_dir(int a) { build_table(); return g_dir[a]; }

static int find_action(int64_t move, int dir) {
  for (int a = 0; a < AM_NUM_ACTIONS; ++a)
    if (g_move[a] == move && g_dir[a] == dir) return a;
  return -1;
}

/* ---- field access ------------------------------------------------------------------------- */
typedef struct {
  uint64_t fwd[2], bwd[2], pieces[2], empty;
  int active, passive, jump, ply, last_move, rep[2];
  int lm[2][2]; /* [player][0|1], -1 = none */
} B;

static int sx8(uint64_t v) { v &= 0xFF; return v == 0xFF ? -1 : (int)v; }

static void load(const am_state* s, B* b) {
  b->fwd[0] = s->w[0]; b->fwd[1] = s->w[1]; b->bwd[0] = s->w[2]; b->bwd[1] = s->w[3];
  b->pieces[0] = b->fwd[0] | b->bwd[0];
  b->pieces[1] = b->fwd[1] | b->bwd[1];
  b->empty = UNUSED ^ MASK36 ^ (b->pieces[0] | b->pieces[1]);
  b->active = (int)(s->w[4] & 1); b->passive = 1 - b->active;
  b->jump = (int)((s->w[4] >> 1) & 1);
  b->ply = (int)((s->w[4] >> 8) & 0xFFFF);
  b->last_move = (int)((s->w[4] >> 24) & 0xFF);
  b->rep[0] = (int)((s->w[4] >> 32) & 0xFF);
  b->rep[1] = (int)((s->w[4] >> 40) & 0xFF);
  b->lm[0][0] = sx8(s->w[5]); b->lm[0][1] = sx8(s->w[5] >> 8);
  b->lm[1][0] = sx8(s->w[5] >> 16); b->lm[1][1] = sx8(s->w[5] >> 24);
}

static void store(const B* b, am_state* s) {
  s->w[0] = b->fwd[0]; s->w[1] = b->fwd[1]; s->w[2] = b->bwd[0]; s->w[3] = b->bwd[1];
  s->w[4] = (uint64_t)(b->active & 1) | ((uint64_t)(b->jump & 1) << 1) | ((uint64_t)(b->ply & 0xFFFF) << 8) |
            ((uint64_t)(b->last_move & 0xFF) << 24) | ((uint64_t)(b->rep[0] & 0xFF) << 32) |
            ((uint64_t)(b->rep[1] & 0xFF) << 40);
  s->w[5] = (uint64_t)(b->lm[0][0] & 0xFF) | ((uint64_t)(b->lm[0][1] & 0xFF) << 8) |
            ((uint64_t)(b->lm[1][0] & 0xFF) << 16) | ((uint64_t)(b->lm[1][1] & 0xFF) << 24);
}

/* GameBoard.cc:10-35 */
void am_clear(am_state* s) {
  B b;
  memset(&b, 0, sizeof(b));
  b.active = BLACK; b.passive = WHITE; b.ply = 1;
  b.fwd[BLACK] = 0x1eff; b.bwd[BLACK] = 0;
  b.fwd[WHITE] = 0; b.bwd[WHITE] = 0x7fbc00000ull;
  b.last_move = AM_M_INVALID; b.jump = 0;
  b.lm[0][0] = b.lm[0][1] = b.lm[1][0] = b.lm[1][1] = -1;
  store(&b, s);
}

/* GameBoard.cc:416-463 (_get_jumps) then :359-414 (_get_moves).  Emission order: rf, lf, rb, lb,
 * each ascending source-bit index.  Returns count; moves[] receives the signed two-bit words. */
static int gen_moves(const B* b, int64_t* moves) {
  int n = 0;
  uint64_t e = b->empty, opp = b->pieces[b->passive];
  uint64_t f = b->fwd[b->active], k = b->bwd[b->active];
  uint64_t rfj = (e >> 8) & (opp >> 4) & f, lfj = (e >> 10) & (opp >> 5) & f;
  uint64_t rbj = (e << 8) & (opp << 4) & k, lbj = (e << 10) & (opp << 5) & k;
  if (rfj | lfj | rbj | lbj) {
    for (int i = 0; i < 64; ++i) if ((rfj >> i) & 1) moves[n++] = -(int64_t)(0x101ull << i);
    for (int i = 0; i < 64; ++i) if ((lfj >> i) & 1) moves[n++] = -(int64_t)(0x401ull << i);
    for (int i = 0; i < 64; ++i) if ((rbj >> i) & 1) moves[n++] = -(int64_t)(0x101ull << (i - 8));
    for (int i = 0; i < 64; ++i) if ((lbj >> i) & 1) moves[n++] = -(int64_t)(0x401ull << (i - 10));
    return n;
  }
  uint64_t rf = (e >> 4) & f, lf = (e >> 5) & f, rb = (e << 4) & k, lb = (e << 5) & k;
  for (int i = 0; i < 64; ++i) if ((rf >> i) & 1) moves[n++] = (int64_t)(0x11ull << i);
  for (int i = 0; i < 64; ++i) if ((lf >> i) & 1) moves[n++] = (int64_t)(0x21ull << i);
  for (int i = 0; i < 64; ++i) if ((rb >> i) & 1) moves[n++] = (int64_t)(0x11ull << (i - 4));
  for (int i = 0; i < 64; ++i) if ((lb >> i) & 1) moves[n++] = (int64_t)(0x21ull << (i - 5));
  return n;
}

/* GameBoard.cc:353-357 */
static int move_dir(const B* b, int64_t move, int player) {
  if (move < 0) move = -move;
  return (int64_t)b->pieces[player] < (int64_t)(b->pieces[player] ^ (uint64_t)move);
}

/* GameBoard.cc:131-162 */
void am_legal_mask(const am_state* s, uint64_t m[3]) {
  build_table();
  B b; load(s, &b);
  int64_t mv[64];
  int n = gen_moves(&b, mv);
  m[0] = m[1] = m[2] = 0;
  for (int i = 0; i < n; ++i) {
    int a = find_action(mv[i], move_dir(&b, mv[i], b.active));
    m[a >> 6] |= 1ull << (a & 63);
  }
  /* repeat rule :151-159 */
  if (n > 1 && b.rep[b.active] >= AM_REPEAT_MOVE) {
    int a = b.lm[b.active][1];
    m[a >> 6] &= ~(1ull << (a & 63));
  }
}

/* _jumps_from GameBoard.cc:465-528: only emptiness of the result matters to CheckersPlay */
static int has_jump_from(const B* b, uint64_t piece) {
  uint64_t e = b->empty, opp = b->pieces[b->passive];
  uint64_t rfj = 0, lfj = 0, rbj = 0, lbj = 0;
  if (b->active == BLACK) {
    rfj = (e >> 8) & (opp >> 4) & piece; lfj = (e >> 10) & (opp >> 5) & piece;
    if (piece & b->bwd[BLACK]) { rbj = (e << 8) & (opp << 4) & piece; lbj = (e << 10) & (opp << 5) & piece; }
  } else {
    rbj = (e << 8) & (opp << 4) & piece; lbj = (e << 10) & (opp << 5) & piece;
    if (piece & b->fwd[WHITE]) { rfj = (e >> 8) & (opp >> 4) & piece; lfj = (e >> 10) & (opp >> 5) & piece; }
  }
  return (rfj | lfj | rbj | lbj) != 0;
}

/* GameBoard.cc:37-129 */
int am_play(am_state* s, int action) {
  build_table();
  B b; load(s, &b);
  int64_t move = g_move[action];
  int active = b.active, passive = b.passive;
  b.last_move = action;
  /* repeat bookkeeping :59-79 */
  if (b.lm[active][1] == action) b.rep[active] += 1; else b.rep[active] = 0;
  b.lm[active][1] = b.lm[active][0];
  b.lm[active][0] = action;
  b.ply += 1;
  if (move < 0) {
    move = -move;
    int sum = 0;
    for (int i = 0; i < 64; ++i) if (((uint64_t)move >> i) & 1) sum += i;
    uint64_t taken = 1ull << (sum / 2);   /* Q5: the reference shifts a 32-bit 1; index <= 30 always */
    b.pieces[passive] ^= taken;
    if (b.fwd[passive] & taken) b.fwd[passive] ^= taken;
    if (b.bwd[passive] & taken) b.bwd[passive] ^= taken;
    b.jump = 1;
  }
  b.pieces[active] ^= (uint64_t)move;
  if (b.fwd[active] & (uint64_t)move) b.fwd[active] ^= (uint64_t)move;
  if (b.bwd[active] & (uint64_t)move) b.bwd[active] ^= (uint64_t)move;
  uint64_t dest = (uint64_t)move & b.pieces[active];
  b.empty = UNUSED ^ MASK36 ^ (b.pieces[0] | b.pieces[1]);
  if (b.jump && has_jump_from(&b, dest)) { store(&b, s); return 1; }  /* Q1,Q3 */
  if (active == BLACK && (dest & 0x780000000ull)) b.bwd[BLACK] |= dest;
  else if (active == WHITE && (dest & 0xfull)) b.fwd[WHITE] |= dest;
  b.jump = 0;
  b.active = passive; b.passive = active;
  store(&b, s);
  return 0;
}

int am_is_over(const am_state* s) {
  B b; load(s, &b);
  int64_t mv[64];
  return gen_moves(&b, mv) == 0;
}

/* GameBoard.cc:210-277 */
void am_observation(const am_state* s, int player, int8_t out[64]) {
  B b; load(s, &b);
  memset(out, 0, 64);
  uint64_t bp = b.fwd[BLACK], bk = b.bwd[BLACK], wp = b.bwd[WHITE], wk = b.fwd[WHITE];
  for (int i = 0; i < 35; ++i) {
    int v;
    if ((bk >> i) & 1) v = 3; else if ((wk >> i) & 1) v = -3;
    else if ((bp >> i) & 1) v = 1; else if ((wp >> i) & 1) v = -1; else continue;
    int buff = i - i / 9;
    int x = 6 - (buff % 4) * 2 + (buff / 4) % 2;
    int y = 7 - buff / 4;
    if (player == BLACK) out[y * 8 + x] = (int8_t)v;
    else out[(7 - y) * 8 + (7 - x)] = (int8_t)(-v);
  }
}

/* ---- GameState wrapper -------------------------------------------------------------------- */
void am_game_reset(am_game* g) {
  am_clear(&g->board);
  g->hist[0] = g->board;
  g->hist_len = 1;
}

int am_game_terminated(const am_game* g) {
  return am_is_over(&g->board) || (int)((g->board.w[4] >> 8) & 0xFFFF) >= AM_MAX_PLY;
}

int am_game_forward(am_game* g, int action) {
  if (am_game_terminated(g) || action > AM_NUM_ACTIONS) return 0;   /* Q6: '>' not '>=' */
  if (action >= AM_NUM_ACTIONS) return 0; /* reference would index out of range; we reject */
  uint64_t m[3];
  am_legal_mask(&g->board, m);
  if (!((m[action >> 6] >> (action & 63)) & 1)) return 0;
  am_play(&g->board, action);
  if (g->hist_len == AM_HISTORY) {
    memmove(&g->hist[0], &g->hist[1], sizeof(am_state) * (AM_HISTORY - 1));
    g->hist_len--;
  }
  g->hist[g->hist_len++] = g->board;
  return 1;
}

float am_game_evaluate(const am_game* g) {
  if (!am_game_terminated(g)) return 0.0f;
  if ((int)((g->board.w[4] >> 8) & 0xFFFF) >= AM_MAX_PLY) return -1.0f;
  return (g->board.w[4] & 1) == BLACK ? -1.0f : 1.0f;
}

/* BoardFeature.cc:47-77; planes are left-padded with zero slots when history < 6 */
void am_game_features(const am_game* g, float out[36 * 64]) {
  memset(out, 0, sizeof(float) * 36 * 64);
  int i = AM_HISTORY - g->hist_len;
  for (int k = 0; i < AM_HISTORY; ++i, ++k) {
    const am_state* b = &g->hist[k];
    int active = (int)(b->w[4] & 1), passive = 1 - active;
    int8_t oa[64], op[64];
    am_observation(b, active, oa);
    am_observation(b, passive, op);   /* Q7: opponent planes in the opponent's own orientation */
    float* base = out + (6 * i) * 64;
    for (int j = 0; j < 64; ++j) {
      if (oa[j] == 1) base[0 * 64 + j] = 1.0f;
      if (oa[j] == 3) base[1 * 64 + j] = 1.0f;
      if (op[j] == 1) base[2 * 64 + j] = 1.0f;
      if (op[j] == 3) base[3 * 64 + j] = 1.0f;
      base[(active == BLACK ? 4 : 5) * 64 + j] = 1.0f;
    }
  }
}

/* ---- synthetic playout driver --------------------------------------------- */
uint64_t elf_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

uint64_t am_playout(am_state* boards, int n, uint64_t board_id0, uint64_t step_base, int nply, uint64_t seed) {
  uint64_t moves = 0;
  for (int b = 0; b < n; ++b) {
    am_state* s = &boards[b];
    for (int t = 0; t < nply; ++t) {
      uint64_t m[3];
      am_legal_mask(s, m);
      int cnt = __builtin_popcountll(m[0]) + __builtin_popcountll(m[1]) + __builtin_popcountll(m[2]);
      int ply = (int)((s->w[4] >> 8) & 0xFFFF);
      if (cnt == 0 || ply >= AM_MAX_PLY) { am_clear(s); continue; }
      uint64_t r = elf_splitmix64(seed + (board_id0 + (uint64_t)b) * 0x9E3779B97F4A7C15ull + step_base + (uint64_t)t);
      int k = (int)(r % (uint64_t)cnt);
      int a = 0;
      for (a = 0; a < AM_NUM_ACTIONS; ++a)
        if ((m[a >> 6] >> (a & 63)) & 1) { if (k == 0) break; --k; }
      am_play(s, a);
      ++moves;
    }
  }
  return moves;
}
