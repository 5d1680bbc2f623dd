// Go 19x19 tree search on gfx950 (BUILD-DEFINED rules, see go.hip; the search itself follows the
// reference's tree_search.h exactly as the checkers search does - mcts_core.h).
//
// A Go move costs ~10^4 VALU operations and only vectorises ACROSS boards (go_dev.h: one lane owns
// one board, the flood fills iterate wave-wide), whereas the PUCT descent vectorises across the
// up-to-362 edges of ONE node.  So a batch step is split by the shape of the work:
//   1. go_tree_select_kernel   wave per tree : B descents with virtual loss; a new child only gets
//                                              an id and (parent, action) - its position is pending
//   2. go_tree_expand_kernel   lane per leaf : clone parent, play the move (captures, ko, Zobrist,
//                                              superko against game history + tree path), legal
//                                              mask, terminal flag and Tromp-Taylor value
//   3. go_tree_features_kernel block per leaf: 18 planes from the leaf, its ancestors and the
//                                              game's stone history
//      (the network runs here, outside this library's search calls)
//   4. go_tree_backup_kernel   wave per tree : pi2response + setEvaluation per new leaf, then the
//                                              batch's updateEdgeStats
// Node positions are 64 x u32 per node (black / white / atari rows, meta, captures, hash).
#include <stdlib.h>
#include "go_dev.h"
#include "mcts_core.h"

namespace {
using namespace godev;
using namespace mcts;

constexpr int NA = 362;
constexpr int GO_SW = 64;  // u32 words per node: rows[57] | meta0 | meta1 | capB | capW | hash lo | hash hi | spare
using GoScratch = ExpandScratch<NA, NA>;

struct GoX {  // per-slot scratch between the kernels of one batch step  [n_trees * B]
  uint64_t* leaf_mask;   // [slots][6]
  uint8_t* leaf_term;    // terminated()
  uint8_t* leaf_flip;    // to move == white
  float* leaf_tval;      // +-1 from the Tromp-Taylor score when terminated
};

struct GoGame {
  // allocateState is deferred: the child keeps NOT_VISITED without META_STATE_SET until the expand kernel
  __device__ static inline void on_new_child(const TreeDev&, TV&, int, int, int) {}
};

__device__ inline void node_load(const uint32_t* w, GoS& s) {
  const uint32_t m0 = w[57], m1 = w[58];
  s.tm = m0 & 1; s.passes = (m0 >> 1) & 3; s.superko = (m0 >> 3) & 1; s.ply = m0 >> 16;
  s.ko = m1 & 0xFFFF; s.last = m1 >> 16;
  const uint32_t cb = w[59], cw = w[60];
  s.capP = s.tm ? cw : cb; s.capQ = s.tm ? cb : cw;
  s.hash = (uint64_t)w[61] | ((uint64_t)w[62] << 32);
  ROWS(i) {
    const uint32_t bl = w[i], wh = w[19 + i];
    s.P[i] = s.tm ? wh : bl; s.Q[i] = s.tm ? bl : wh;
    s.AT[i] = w[38 + i];
  }
}
__device__ inline void node_store(uint32_t* w, const GoS& s) {
  ROWS(i) {
    w[i] = s.tm ? s.Q[i] : s.P[i];
    w[19 + i] = s.tm ? s.P[i] : s.Q[i];
    w[38 + i] = s.AT[i];
  }
  w[57] = s.tm | (s.passes << 1) | (s.superko << 3) | (s.ply << 16);
  w[58] = (s.ko & 0xFFFF) | (s.last << 16);
  w[59] = s.tm ? s.capQ : s.capP;
  w[60] = s.tm ? s.capP : s.capQ;
  w[61] = (uint32_t)s.hash; w[62] = (uint32_t)(s.hash >> 32); w[63] = 0;
}

// TreeSearchT::setRootNodeState tree_search.h:570-585 (+ align_state's clear when not persistent)
__global__ void __launch_bounds__(64)
go_tree_set_roots_kernel(TreeDev d, GoDev g, int persistent, uint8_t* __restrict__ ok) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= d.n_trees) return;
  if (d.t_flags[t] & FLAG_INACTIVE) { if (ok) ok[t] = 1; return; }
  if (!persistent) { reset_tree(d, t); d.t_flags[t] &= ~FLAG_ROOT_MISMATCH; }
  TV tv = tree_view(d, t);
  GoS s; go_load(g, t, s);
  uint32_t w[GO_SW];
  node_store(w, s);
  bool good = true;
  if (tv.meta[0] & META_STATE_SET) {
    for (int k = 0; k < 38; ++k) good = good && tv.state[k] == w[k];
    for (int k = 57; k < 63; ++k) good = good && tv.state[k] == w[k];
    if (!good) d.t_flags[t] |= FLAG_ROOT_MISMATCH;
  } else {
    for (int k = 0; k < GO_SW; ++k) tv.state[k] = w[k];
    tv.meta[0] |= META_STATE_SET;
  }
  if (ok) ok[t] = good ? 1 : 0;
}

// step 1: the descents of one batch
__global__ void __launch_bounds__(64) go_tree_select_kernel(TreeDev d, Opt o) {
  const int t = blockIdx.x, lane = threadIdx.x;
  int flags = d.t_flags[t];
  if (flags) {
    if (lane < o.B) { d.leaf[(size_t)t * o.B + lane] = 0; d.locked[(size_t)t * o.B + lane] = 0; }
    return;  // overflowed / mismatched trees are parked until cleared
  }
  TV tv = tree_view(d, t);
  int nn = d.t_nnodes[t];
  unsigned long long depth_sum = 0;
  for (int j = 0; j < o.B; ++j) {
    int depth = 0;
    const int leaf = select_one<GoGame>(d, tv, o, t, j, lane, nn, flags, depth);
    depth_sum += (unsigned long long)depth;
    // requestEvaluation tree_search_node.h:169-179
    const uint32_t meta = tv.meta[leaf];
    const int lock = (meta & 0xFF) == NOT_VISITED;
    if (lock) tv.meta[leaf] = (meta & ~0xFFu) | EVAL_REQUESTED;
    __threadfence_block();
    if (lane == 0) { d.leaf[(size_t)t * o.B + j] = leaf; d.locked[(size_t)t * o.B + j] = (uint8_t)lock; }
  }
  if (lane == 0) {
    d.t_nnodes[t] = nn; d.t_flags[t] = flags;
    atomicAdd(&d.stats[0], (unsigned long long)o.B);
    if (depth_sum) atomicAdd(&d.stats[2], depth_sum);
  }
}

// step 2: positions of the new leaves (lane = batch slot)
__global__ void __launch_bounds__(64)
go_tree_expand_kernel(TreeDev d, Opt o, GoDev g, const uint64_t* __restrict__ gz, GoX x, uint64_t* __restrict__ leaf_states,
                      uint8_t* __restrict__ need_eval) {
  __shared__ uint64_t z[GO_ZWORDS];
  go_load_zobrist(z, gz);
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  const int nslots = d.n_trees * o.B;
  const bool valid = slot < nslots;
  const int t = valid ? slot / o.B : 0;
  const bool live = valid && d.t_flags[t] == 0 && d.locked[slot] != 0;
  const int leaf = live ? d.leaf[slot] : 0;
  TV tv = tree_view(d, t);
  GoS s; go_clear(s);
  const bool pending = live && !(tv.meta[leaf] & META_STATE_SET);
  uint32_t a = 0;
  int parent = -1;
  if (pending) {
    parent = tv.parent[leaf];
    a = tv.paction[leaf];
    node_load(tv.state + (size_t)parent * GO_SW, s);
  } else if (live) {
    node_load(tv.state + (size_t)leaf * GO_SW, s);
  }
  // allocateState tree_search.h:209-224: clone + forward.  Positional superko = the new position is
  // in the game's history (the slab's table, read only) or on the path from the root to the parent.
  const ulonglong2* tab = g.table + (size_t)t * TAB_SLOTS;
  const uint64_t gen = g.gen[t];
  go_forward_t(s, pending, a, z, [](uint64_t) {}, [&](uint64_t h) -> uint32_t {
    if (go_table_contains(tab, gen, h)) return 1u;
    for (int p = parent; p >= 0; p = tv.parent[p]) {
      const uint32_t* w = tv.state + (size_t)p * GO_SW;
      if (w[61] == (uint32_t)h && w[62] == (uint32_t)(h >> 32)) return 1u;
    }
    return 0u;
  });
  if (pending) {
    node_store(tv.state + (size_t)leaf * GO_SW, s);
    tv.meta[leaf] = tv.meta[leaf] | META_STATE_SET;
  }
  uint32_t L[GN];
  go_legal_rows(s, L);
  const bool alive = !go_terminated(s);
  uint64_t m[6];
  go_pack_mask(L, alive, m);
  // Tromp-Taylor value of terminated leaves (GameState::evaluateGame's role); fills run wave-wide
  float tval = 0.f;
  if (__any(live && !alive)) {
    uint32_t E[GN], RP[GN], RQ[GN];
    ROWS(i) E[i] = ~(s.P[i] | s.Q[i]) & FULL;
    ROWS(i) { RP[i] = NBR(s.P, i) & E[i]; RQ[i] = NBR(s.Q, i) & E[i]; }
    go_fill(RP, E);
    go_fill(RQ, E);
    int ap = 0, aq = 0;
    ROWS(i) { ap += __popc(s.P[i]) + __popc(RP[i] & ~RQ[i]); aq += __popc(s.Q[i]) + __popc(RQ[i] & ~RP[i]); }
    const int black = s.tm ? aq : ap, white = s.tm ? ap : aq;
    tval = ((float)black - (float)white - 7.5f) > 0 ? 1.f : -1.f;
  }
  if (valid) {
#pragma unroll
    for (int k = 0; k < 6; ++k) x.leaf_mask[(size_t)slot * 6 + k] = live ? m[k] : 0ull;
    x.leaf_term[slot] = (live && !alive) ? 1 : 0;
    x.leaf_flip[slot] = (uint8_t)s.tm;
    x.leaf_tval[slot] = tval;
    if (need_eval) need_eval[slot] = (live && alive) ? 1 : 0;
    if (leaf_states) {
      uint64_t* w = leaf_states + (size_t)slot * GO_WORDS;
      ROWS(i) {
        const uint32_t bl = s.tm ? s.Q[i] : s.P[i], wh = s.tm ? s.P[i] : s.Q[i];
        w[i] = live ? ((uint64_t)bl | ((uint64_t)wh << 32)) : 0ull;
      }
      w[19] = live ? s.hash : 0ull;
      w[20] = live ? ((uint64_t)s.tm | ((uint64_t)s.passes << 1) | ((uint64_t)s.superko << 3) | ((uint64_t)s.ply << 16) |
                      ((uint64_t)(s.ko & 0xFFFF) << 32) | ((uint64_t)(s.last & 0xFFFF) << 48)) : 0ull;
      const uint32_t cb = s.tm ? s.capQ : s.capP, cw = s.tm ? s.capP : s.capQ;
      w[21] = live ? ((uint64_t)cb | ((uint64_t)cw << 32)) : 0ull;
    }
  }
}

// step 3: the 18 planes of each leaf that goes to the network (zeros for the other slots):
// 2i / 2i+1 = stones of the leaf's side to move / of the other side i plies earlier (leaf, its
// ancestors up to the root, then the game's own stone history from the board slab), 16/17 = colour.
__global__ void __launch_bounds__(384)
go_tree_features_kernel(TreeDev d, Opt o, GoDev g, const uint8_t* __restrict__ need_eval, float* __restrict__ out) {
  const int slot = blockIdx.x, p = threadIdx.x;
  if (p >= GO_POINTS) return;
  float* dst = out + (size_t)slot * 18 * GO_POINTS;
  if (!need_eval[slot]) {
    for (int q = 0; q < 18; ++q) dst[(size_t)q * GO_POINTS + p] = 0.f;
    return;
  }
  const int t = slot / o.B;
  TV tv = tree_view(d, t);
  const int leaf = d.leaf[slot];
  const uint32_t tm = tv.state[(size_t)leaf * GO_SW + 57] & 1;
  const int r = p / 19, c = p % 19;
  const size_t n = g.n;
  const uint32_t hm = g.hmeta[t];
  const int len = hm & 15, head = (hm >> 4) & 7;
  int node = leaf, k = 0;  // k: plies below the root once the chain is exhausted
  for (int i = 0; i < 8; ++i) {
    uint32_t bl = 0, wh = 0;
    if (node >= 0) {
      const uint32_t* w = tv.state + (size_t)node * GO_SW;
      bl = w[r]; wh = w[19 + r];
      node = tv.parent[node];
    } else {
      ++k;  // the root itself is the slab's newest entry (index len-1)
      if (len - 1 - k >= 0) {
        const int hs = (head + len - 1 - k) & 7;
        bl = g.hist[((size_t)hs * 38 + r) * n + t]; wh = g.hist[((size_t)hs * 38 + 19 + r) * n + t];
      }
    }
    const uint32_t own = tm ? wh : bl, opp = tm ? bl : wh;
    dst[(size_t)(2 * i) * GO_POINTS + p] = (float)((own >> c) & 1);
    dst[(size_t)(2 * i + 1) * GO_POINTS + p] = (float)((opp >> c) & 1);
  }
  dst[(size_t)16 * GO_POINTS + p] = tm == 0 ? 1.f : 0.f;
  dst[(size_t)17 * GO_POINTS + p] = tm == 1 ? 1.f : 0.f;
}

// step 4: setEvaluation of the locked leaves + backup of the batch.  One workgroup per tree, one wave per
// leaf (up to BACKUP_WAVES at a time): the waves prepare their leaves' edge lists side by side, each in its own
// LDS scratch, then take their edge ranges in slot order (the order the reference's threads-of-one would have
// expanded them in) and write them; wave 0 then runs the batch's updateEdgeStats.
constexpr int BACKUP_WAVES = 8;
__global__ void __launch_bounds__(64 * BACKUP_WAVES)
go_tree_backup_kernel(TreeDev d, Opt o, GoX x, const float* __restrict__ pi, const float* __restrict__ vin) {
  extern __shared__ __attribute__((aligned(16))) char go_backup_smem[];
  __shared__ int s_leaf[MAXB];
  __shared__ int s_locked[MAXB];
  __shared__ int s_m[MAXB];
  const int t = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
  GoScratch& S = ((GoScratch*)go_backup_smem)[wave];
  int flags = d.t_flags[t];
  if (flags) return;
  TV tv = tree_view(d, t);
  int ne = d.t_nedges[t];
  unsigned long long evals = 0, edge_sum = 0;
  if ((int)threadIdx.x < o.B) { s_leaf[threadIdx.x] = d.leaf[(size_t)t * o.B + threadIdx.x]; s_locked[threadIdx.x] = d.locked[(size_t)t * o.B + threadIdx.x]; }
  __syncthreads();
  for (int j0 = 0; j0 < o.B; j0 += W) {
    const int j = j0 + wave;
    const bool have = j < o.B && s_locked[j < o.B ? j : 0];
    const size_t slot = (size_t)t * o.B + (have ? j : 0);
    bool terminal = true;
    float value = 0.0f;
    int m = 0;
    if (have) {
      terminal = x.leaf_term[slot] != 0;
      value = terminal ? x.leaf_tval[slot] : (vin ? vin[slot] : 0.0f);
      if (!terminal) {
        evals += 1;
        m = expand_prepare<NA, NA>(d, x.leaf_mask + slot * 6, pi ? pi + slot * NA : nullptr, lane, S);
      }
    }
    if (lane == 0 && j < o.B) s_m[j] = m;
    __syncthreads();
    // edge ranges in slot order; every wave walks the same counts, so `ne` and `flags` agree in all of them
    int my_eb = ne, my_m = 0;
    const int jend = j0 + W < o.B ? j0 + W : o.B;
    for (int k = j0; k < jend; ++k) {
      int mk = s_m[k];
      if (mk < 0) { flags |= FLAG_LEGAL_OVERFLOW; mk = 0; }
      if (ne + mk > d.max_edges) { flags |= FLAG_EDGE_OVERFLOW; mk = 0; }
      if (k == j) { my_eb = ne; my_m = mk; }
      ne += mk;
    }
    if (have) {
      expand_commit<NA, NA>(tv, s_leaf[j], x.leaf_flip[slot] != 0, value, my_eb, my_m, lane, S);
      edge_sum += (unsigned long long)my_m;
    }
    __syncthreads();
  }
  __threadfence_block();
  __syncthreads();
  if (wave == 0) backup_batch(d, tv, o, t, s_leaf);
  if (lane == 0) {
    if (wave == 0) { d.t_nedges[t] = ne; d.t_flags[t] = flags; }
    if (evals) atomicAdd(&d.stats[1], evals);
    if (edge_sum) atomicAdd(&d.stats[3], edge_sum);
  }
}

struct GoMcts : MctsBase {
  GoX x{};
  GoDev g{};
  const uint64_t* gz = nullptr;
  uint8_t* own_need = nullptr;  // need_eval scratch when the caller passes none
  bool have_roots = false;

  int alloc(int n_trees, const elfhip_ts_options* opt) {
    const size_t S = (size_t)n_trees * opt->num_rollouts_per_batch;
    MCTS_ALLOC(x.leaf_mask, S * 6 * 8); MCTS_ALLOC(x.leaf_term, S); MCTS_ALLOC(x.leaf_flip, S); MCTS_ALLOC(x.leaf_tval, S * 4);
    MCTS_ALLOC(own_need, S);
    static_assert(sizeof(GoScratch) * BACKUP_WAVES <= 150 * 1024, "backup scratch must fit the CU's LDS");
    HIPCHECK(hipFuncSetAttribute((const void*)go_tree_backup_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(GoScratch) * BACKUP_WAVES)));
    return alloc_common(n_trees, opt, NA, GO_SW, 256);
  }
  int set_roots(BoardSlab* slab, uint8_t* d_ok) override {
    g = *(const GoDev*)slab->dev_ptr[0];
    gz = (const uint64_t*)slab->dev_ptr[1];
    have_roots = true;
    hipLaunchKernelGGL(go_tree_set_roots_kernel, dim3((d.n_trees + 63) / 64), dim3(64), 0, stream, d, g, o.persistent, d_ok);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int select(uint64_t* d_leaf_states, float* d_leaf_features, uint8_t* d_need_eval) override {
    if (!have_roots) { elfhip_set_error("mcts_select(go19): set_roots has not been called"); return ELFHIP_ESTATE; }
    const int nslots = d.n_trees * o.B;
    uint8_t* need = d_need_eval ? d_need_eval : own_need;
    hipLaunchKernelGGL(go_tree_select_kernel, dim3(d.n_trees), dim3(64), 0, stream, d, o);
    hipLaunchKernelGGL(go_tree_expand_kernel, dim3((nslots + 63) / 64), dim3(64), 0, stream, d, o, g, gz, x, d_leaf_states, need);
    if (d_leaf_features)
      hipLaunchKernelGGL(go_tree_features_kernel, dim3(nslots), dim3(384), 0, stream, d, o, g, (const uint8_t*)need, d_leaf_features);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int backup(const float* d_pi, const float* d_value) override {
    // few trees: a wave per leaf (latency: 2.3 -> 0.3 ms per batch at 64 trees); many trees: the chip is full with
    // one wave per tree and twelve trees per CU fit the LDS (4,096 trees: 2.7 ms against 5.1 ms with 8 waves)
    int waves = d.n_trees > 1024 ? 1 : (o.B < BACKUP_WAVES ? o.B : BACKUP_WAVES);
    if (const char* e = getenv("ELFHIP_GO_BACKUP_WAVES")) {   // tests: both shapes on the same small search
      const int w = atoi(e);
      if (w >= 1 && w <= BACKUP_WAVES) waves = w;
    }
    hipLaunchKernelGGL(go_tree_backup_kernel, dim3(d.n_trees), dim3(64 * waves), sizeof(GoScratch) * waves, stream, d, o, x, d_pi, d_value);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  // TreeSearchT::run with the uniform evaluator (pi = 1/362 before filtering, V = 0)
  int run_uniform() override {
    const int nbatch = (o.R + o.B - 1) / o.B;
    for (int b = 0; b < nbatch; ++b) {
      int rc = select(nullptr, nullptr, nullptr);
      if (rc != ELFHIP_OK) return rc;
      rc = backup(nullptr, nullptr);
      if (rc != ELFHIP_OK) return rc;
    }
    return ELFHIP_OK;
  }
};

}  // namespace

mcts::MctsBase* elfhip_make_mcts_go(int n_trees, const elfhip_ts_options* opt, int* rc) {
  GoMcts* m = new (std::nothrow) GoMcts();
  if (!m) { elfhip_set_error("mcts_create: host allocation failed"); *rc = ELFHIP_ENOMEM; return nullptr; }
  *rc = m->alloc(n_trees, opt);
  if (*rc != ELFHIP_OK) { delete m; return nullptr; }
  return m;
}
