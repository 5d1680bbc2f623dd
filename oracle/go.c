/* TEST INFRASTRUCTURE ONLY (oracle) - see go.h.  PARITY UNPINNED vs ELF (no Go in the reference).
 * Classic formulation: point array, groups labelled by flood fill, liberties counted per group.
 */
#include "go.h"
#include <string.h>

uint64_t elf_splitmix64(uint64_t x); /* american.c */

uint64_t go_zobrist(int colour, int point) {
  return elf_splitmix64(GO_ZSEED + (uint64_t)(colour * GO_POINTS + point));
}

static int nbrs(int p, int out[4]) {
  int r = p / GO_N, c = p % GO_N, n = 0;
  if (r > 0) out[n++] = p - GO_N;
  if (r < GO_N - 1) out[n++] = p + GO_N;
  if (c > 0) out[n++] = p - 1;
  if (c < GO_N - 1) out[n++] = p + 1;
  return n;
}

/* Label every group: gid[p] = group index (or -1), libs[g] = number of distinct liberties */
typedef struct { int16_t gid[GO_POINTS]; int16_t libs[GO_POINTS]; int16_t size[GO_POINTS]; int ngroups; } groups_t;

static void label_groups(const int8_t* pt, groups_t* G) {
  int stack[GO_POINTS];
  int16_t libmark[GO_POINTS];
  memset(libmark, -1, sizeof(libmark));
  for (int p = 0; p < GO_POINTS; ++p) G->gid[p] = -1;
  G->ngroups = 0;
  for (int p0 = 0; p0 < GO_POINTS; ++p0) {
    if (pt[p0] == 0 || G->gid[p0] >= 0) continue;
    int g = G->ngroups++, sp = 0, col = pt[p0], nl = 0, sz = 0;
    stack[sp++] = p0; G->gid[p0] = (int16_t)g;
    while (sp) {
      int p = stack[--sp]; ++sz;
      int nb[4], k = nbrs(p, nb);
      for (int j = 0; j < k; ++j) {
        int q = nb[j];
        if (pt[q] == 0) { if (libmark[q] != g) { libmark[q] = (int16_t)g; ++nl; } }
        else if (pt[q] == col && G->gid[q] < 0) { G->gid[q] = (int16_t)g; stack[sp++] = q; }
      }
    }
    G->libs[g] = (int16_t)nl; G->size[g] = (int16_t)sz;
  }
}

static int legal_point(const go_state* s, const groups_t* G, int p) {
  if (s->pt[p] != 0 || p == s->ko) return 0;
  int own = s->to_move + 1, nb[4], k = nbrs(p, nb);
  for (int j = 0; j < k; ++j) {
    int q = nb[j];
    if (s->pt[q] == 0) return 1;                         /* an empty neighbour: a liberty */
    int g = G->gid[q];
    if (s->pt[q] == own) { if (G->libs[g] >= 2) return 1; }  /* joins a group that keeps a liberty */
    else if (G->libs[g] == 1) return 1;                  /* captures a group in atari */
  }
  return 0;                                              /* suicide */
}

static void push_stones(go_state* s) {
  int keep = s->nstones_hist < GO_HISTORY ? s->nstones_hist : GO_HISTORY - 1;
  memmove(s->stones_hist[1], s->stones_hist[0], (size_t)keep * GO_POINTS);
  memcpy(s->stones_hist[0], s->pt, GO_POINTS);
  s->nstones_hist = keep + 1;
}

void go_reset(go_state* s) {
  memset(s, 0, sizeof(*s));
  s->ko = GO_NO_POINT; s->last = GO_NO_POINT;
  s->hist[0] = 0; s->nhist = 1;
  s->nstones_hist = 0;
  push_stones(s);
}

int go_terminated(const go_state* s) { return s->passes >= 2 || s->ply >= GO_MAX_PLY || s->superko; }

void go_legal_mask(const go_state* s, uint64_t m[6]) {
  memset(m, 0, 6 * sizeof(uint64_t));
  if (go_terminated(s)) return;
  groups_t G; label_groups(s->pt, &G);
  for (int p = 0; p < GO_POINTS; ++p)
    if (legal_point(s, &G, p)) m[p >> 6] |= 1ull << (p & 63);
  m[GO_PASS >> 6] |= 1ull << (GO_PASS & 63);
}

int go_forward(go_state* s, int a) {
  if (a < 0 || a >= GO_ACTIONS || go_terminated(s)) return 0;
  if (a == GO_PASS) {
    s->passes++; s->ply++; s->ko = GO_NO_POINT; s->last = a; s->to_move ^= 1;
    push_stones(s);
    return 1;
  }
  groups_t G; label_groups(s->pt, &G);
  if (!legal_point(s, &G, a)) return 0;
  int own = s->to_move + 1, opp = 2 - s->to_move;
  int nb[4], k = nbrs(a, nb), ncap = 0, cap_pt = -1;
  s->pt[a] = (int8_t)own;
  s->hash ^= go_zobrist(s->to_move, a);
  for (int j = 0; j < k; ++j) {
    int q = nb[j];
    if (s->pt[q] != opp) continue;          /* may already have been removed with an earlier neighbour */
    int g = G.gid[q];
    if (G.libs[g] != 1) continue;           /* its only liberty was `a` */
    for (int p = 0; p < GO_POINTS; ++p)
      if (G.gid[p] == g && s->pt[p] == opp) {
        s->pt[p] = 0; s->hash ^= go_zobrist(s->to_move ^ 1, p); ++ncap; cap_pt = p;
      }
  }
  s->cap[s->to_move] += ncap;
  /* ko: exactly one stone captured by a lone stone that now has exactly one liberty */
  s->ko = GO_NO_POINT;
  if (ncap == 1) {
    groups_t H; label_groups(s->pt, &H);
    int g = H.gid[a];
    if (H.size[g] == 1 && H.libs[g] == 1) s->ko = cap_pt;
  }
  s->passes = 0; s->ply++; s->last = a; s->to_move ^= 1;
  /* positional superko: the new position occurred before in this game -> terminal flag */
  for (int i = 0; i < s->nhist; ++i) if (s->hist[i] == s->hash) { s->superko = 1; break; }
  s->hist[s->nhist++] = s->hash;
  push_stones(s);
  return 1;
}

float go_tt_score(const go_state* s) {
  /* Tromp-Taylor: a point scores for a colour if it holds that colour or is empty and reaches
   * only that colour through empty points */
  int reach[GO_POINTS]; memset(reach, 0, sizeof(reach));
  int stack[GO_POINTS], black = 0, white = 0;
  int8_t seen[GO_POINTS]; memset(seen, 0, sizeof(seen));
  for (int p0 = 0; p0 < GO_POINTS; ++p0) {
    if (s->pt[p0] == 1) { ++black; continue; }
    if (s->pt[p0] == 2) { ++white; continue; }
    if (seen[p0]) continue;
    int sp = 0, sz = 0, touch = 0;
    stack[sp++] = p0; seen[p0] = 1;
    while (sp) {
      int p = stack[--sp]; ++sz;
      int nb[4], k = nbrs(p, nb);
      for (int j = 0; j < k; ++j) {
        int q = nb[j];
        if (s->pt[q] == 0) { if (!seen[q]) { seen[q] = 1; stack[sp++] = q; } }
        else touch |= s->pt[q];
      }
    }
    if (touch == 1) black += sz; else if (touch == 2) white += sz;
  }
  return (float)black - (float)white - 7.5f;
}

float go_evaluate(const go_state* s) {
  if (!go_terminated(s)) return 0.0f;
  return go_tt_score(s) > 0 ? 1.0f : -1.0f;
}

/* 18 planes: 2i = to-move player's stones i plies ago, 2i+1 = opponent's, i=0..7 (latest first);
 * 16 = all ones if black to move, 17 = all ones if white to move */
void go_features(const go_state* s, float out[GO_PLANES * GO_POINTS]) {
  memset(out, 0, sizeof(float) * GO_PLANES * GO_POINTS);
  int own = s->to_move + 1;
  for (int i = 0; i < s->nstones_hist && i < GO_HISTORY; ++i)
    for (int p = 0; p < GO_POINTS; ++p) {
      int v = s->stones_hist[i][p];
      if (v == own) out[(2 * i) * GO_POINTS + p] = 1.0f;
      else if (v != 0) out[(2 * i + 1) * GO_POINTS + p] = 1.0f;
    }
  for (int p = 0; p < GO_POINTS; ++p) out[(16 + s->to_move) * GO_POINTS + p] = 1.0f;
}

void go_export(const go_state* s, uint64_t w[GO_STATE_WORDS]) {
  memset(w, 0, sizeof(uint64_t) * GO_STATE_WORDS);
  for (int r = 0; r < GO_N; ++r)
    for (int c = 0; c < GO_N; ++c) {
      int v = s->pt[r * GO_N + c];
      if (v == 1) w[r] |= 1ull << c;
      else if (v == 2) w[r] |= 1ull << (32 + c);
    }
  w[19] = s->hash;
  w[20] = (uint64_t)(s->to_move & 1) | ((uint64_t)(s->passes & 3) << 1) | ((uint64_t)(s->superko & 1) << 3) |
          ((uint64_t)(s->ply & 0xFFFF) << 16) | ((uint64_t)(s->ko & 0xFFFF) << 32) | ((uint64_t)(s->last & 0xFFFF) << 48);
  w[21] = (uint64_t)(uint32_t)s->cap[0] | ((uint64_t)(uint32_t)s->cap[1] << 32);
}

void go_import(go_state* s, const uint64_t w[GO_STATE_WORDS]) {
  memset(s, 0, sizeof(*s));
  for (int r = 0; r < GO_N; ++r)
    for (int c = 0; c < GO_N; ++c) {
      if ((w[r] >> c) & 1) s->pt[r * GO_N + c] = 1;
      else if ((w[r] >> (32 + c)) & 1) s->pt[r * GO_N + c] = 2;
    }
  s->hash = 0;
  for (int p = 0; p < GO_POINTS; ++p) if (s->pt[p]) s->hash ^= go_zobrist(s->pt[p] - 1, p);
  s->to_move = (int)(w[20] & 1); s->passes = (int)((w[20] >> 1) & 3); s->superko = (int)((w[20] >> 3) & 1);
  s->ply = (int)((w[20] >> 16) & 0xFFFF); s->ko = (int)((w[20] >> 32) & 0xFFFF); s->last = (int)((w[20] >> 48) & 0xFFFF);
  s->cap[0] = (int)(uint32_t)w[21]; s->cap[1] = (int)(w[21] >> 32);
  s->hist[0] = s->hash; s->nhist = 1;
  s->nstones_hist = 0;
  push_stones(s);
}

int go_sizeof_state(void) { return (int)sizeof(go_state); }

uint64_t go_playout(go_state* boards, int n, uint64_t board_id0, uint64_t step_base, int nply, uint64_t seed) {
  uint64_t moves = 0;
  for (int b = 0; b < n; ++b) {
    go_state* s = &boards[b];
    for (int t = 0; t < nply; ++t) {
      if (go_terminated(s)) { go_reset(s); continue; }
      uint64_t m[6];
      go_legal_mask(s, m);
      m[GO_PASS >> 6] &= ~(1ull << (GO_PASS & 63));
      int cnt = 0;
      for (int i = 0; i < 6; ++i) cnt += __builtin_popcountll(m[i]);
      if (cnt == 0) { go_forward(s, GO_PASS); ++moves; continue; }
      uint64_t r = elf_splitmix64(seed + (board_id0 + (uint64_t)b) * 0x9E3779B97F4A7C15ull + step_base + (uint64_t)t);
      int k = (int)(r % (uint64_t)cnt), a;
      for (a = 0; a < GO_POINTS; ++a)
        if ((m[a >> 6] >> (a & 63)) & 1) { if (k == 0) break; --k; }
      go_forward(s, a);
      ++moves;
    }
  }
  return moves;
}
