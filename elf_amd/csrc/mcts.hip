// Batched PUCT tree search on gfx950: one wavefront owns one tree; the 64 lanes work on the edges
// of the node being visited (UCT scoring + wave argmax, edge init, subtree compaction), the
// per-rollout control flow is wave-uniform.
//
// Semantics follow the reference's single-thread search operation for operation
//   elf/ai/tree_search/tree_search.h:241-378 (batch_rollouts / single_rollout)
//   tree_search_node.h:188-317,385-421 (setEvaluation, findMove/UCT, addVirtualLoss,
//       updateEdgeStats, followEdge), :458-507 (treeAdvance)
//   tree_search_base.h:175-201 (EdgeInfo::getScore), :310-367 (MCTSResultT::addActions)
//   elfgames/american_checkers/mcts/MCTSGameActor.h:198-335 (pre_evaluate, pi2response)
// including the float/double mix in getScore and the libstdc++ orders (stdorder.h) that decide
// every tie.  What is different is the machinery: no mutexes, no heap nodes, no hash maps.
// A tree is two ping-pong arenas in HBM (nodes + a struct-of-arrays edge pool; a node's edges are
// contiguous and already stored in unordered_map iteration order, so the UCT scan is one coalesced
// read); treeAdvance compacts the kept subtree into the other arena with a breadth-first copy.
#include <cmath>
#include <new>
#include <vector>
#include "american_dev.h"
#include "common.h"
#include "stdorder.h"

namespace {
using namespace amdev;

constexpr int NA = 170;          // actions
constexpr int MAXD = 256;        // longest trajectory (a game has < 250 plies)
constexpr int MAXB = 16;         // num_rollouts_per_batch limit
constexpr int MAXLEGAL = 64;     // legal moves per position limit (12 pieces x 4)
enum { NOT_VISITED = 0, EVAL_REQUESTED = 1, VISITED = 2 };
constexpr uint32_t META_STATE_SET = 1u << 9, META_FLIP = 1u << 8;
enum { FLAG_NODE_OVERFLOW = 1, FLAG_EDGE_OVERFLOW = 2, FLAG_DEPTH_OVERFLOW = 4, FLAG_LEGAL_OVERFLOW = 8, FLAG_ROOT_MISMATCH = 16 };

struct Opt {
  int R, B, vl, use_prior; float c_puct; int uqz, ruqz, persistent, pick;
};

struct TreeDev {
  int n_trees, max_nodes, max_edges;
  // nodes [2][n_trees][max_nodes]
  uint32_t* n_meta; uint32_t* n_ebase; int32_t* n_visits; float* n_V; float* n_meanQ; float* n_parentQ;
  int32_t* n_parent; uint64_t* n_board;  // board: [..][6]
  // edges [2][n_trees][max_edges]
  uint16_t* e_action; float* e_prior; int32_t* e_child; float* e_reward; int32_t* e_visits; float* e_vloss;
  // per tree
  int32_t* t_arena; int32_t* t_nnodes; int32_t* t_nedges; int32_t* t_flags;
  uint64_t* t_hist;   // [n_trees][5][4] boards that preceded the root, most recent first
  uint32_t* t_hmeta;  // count (0..5) | active bit of entry i at bit 8+i
  // batch scratch
  uint16_t* traj_node; uint16_t* traj_edge;  // [n_trees][B][MAXD]
  int32_t* traj_len; int32_t* leaf; uint8_t* locked;  // [n_trees][B]
  const uint16_t* uniform_order;  // std::sort order of 170 equal keys
  unsigned long long* stats;      // rollouts, evals, depth_sum, edge_sum
};

struct TV {  // one tree, one arena
  uint32_t* meta; uint32_t* ebase; int32_t* visits; float* V; float* meanQ; float* parentQ; int32_t* parent; uint64_t* board;
  uint16_t* ea; float* ep; int32_t* ec; float* er; int32_t* ev; float* el;
};

__device__ inline TV tree_view(const TreeDev& d, int t, int arena) {
  size_t nb = ((size_t)arena * d.n_trees + t) * d.max_nodes, eb = ((size_t)arena * d.n_trees + t) * d.max_edges;
  TV v;
  v.meta = d.n_meta + nb; v.ebase = d.n_ebase + nb; v.visits = d.n_visits + nb; v.V = d.n_V + nb;
  v.meanQ = d.n_meanQ + nb; v.parentQ = d.n_parentQ + nb; v.parent = d.n_parent + nb; v.board = d.n_board + nb * 6;
  v.ea = d.e_action + eb; v.ep = d.e_prior + eb; v.ec = d.e_child + eb; v.er = d.e_reward + eb;
  v.ev = d.e_visits + eb; v.el = d.e_vloss + eb;
  return v;
}

__device__ inline void load_board(const uint64_t* p, Am& s) {
  s.f[0] = p[0]; s.f[1] = p[1]; s.k[0] = p[2]; s.k[1] = p[3]; s.w4 = p[4]; s.w5 = p[5];
}
__device__ inline void store_board(uint64_t* p, const Am& s) {
  p[0] = s.f[0]; p[1] = s.f[1]; p[2] = s.k[0]; p[3] = s.k[1]; p[4] = s.w4; p[5] = s.w5;
}

__device__ inline void init_node(TV& tv, int id, float pq, int parent) {  // NodeT ctor tree_search_node.h:110-115
  tv.meta[id] = NOT_VISITED; tv.ebase[id] = 0; tv.visits[id] = 0; tv.V[id] = 0.f;
  tv.meanQ[id] = pq; tv.parentQ[id] = pq; tv.parent[id] = parent;
}

// One descent from the root (single_rollout tree_search.h:319-378).  Wave-uniform; returns the leaf.
// All lanes perform the scalar stores redundantly (same value, same address) so that each lane's
// later loads are ordered after its own stores.
__device__ inline int select_one(const TreeDev& d, TV& tv, const Opt& o, int t, int slot, int lane, int& nn, int& flags,
                                 int& depth_out) {
  uint16_t* tn = d.traj_node + ((size_t)t * o.B + slot) * MAXD;
  uint16_t* te = d.traj_edge + ((size_t)t * o.B + slot) * MAXD;
  int node = 0, depth = 0;
  for (;;) {
    const uint32_t meta = tv.meta[node];
    if ((meta & 0xFF) != VISITED) break;
    const int ne = (int)(meta >> 16);
    if (ne == 0) break;  // findMove: no edges (terminal position)
    if (depth >= MAXD) { flags |= FLAG_DEPTH_OVERFLOW; break; }
    const bool flip = (meta & META_FLIP) != 0;
    const int eb = (int)tv.ebase[node];
    float mq = tv.meanQ[node];
    if (o.uqz || (o.ruqz && depth == 0)) mq = 0.0f;
    const float pq = tv.parentQ[node];
    const int nv = tv.visits[node];
    const double sq = sqrt((double)(nv + 1));
    float best = -INFINITY; int bestj = 0x7FFFFFFF;
    float total_uq = 0.f; int total_cnt = 0;
    for (int base = 0; base < ne; base += 64) {
      const int j = base + lane;
      const bool ok = j < ne;
      float score = -INFINITY, uq = 0.f; int counted = 0;
      if (ok) {  // EdgeInfo::getScore tree_search_base.h:175-201
        const float prior = tv.ep[eb + j], reward = tv.er[eb + j], vloss = tv.el[eb + j];
        const int visits = tv.ev[eb + j];
        float r = flip ? -reward : reward;
        r -= vloss;
        const int nvl = (int)((float)visits + vloss);
        const float q = nvl > 0 ? r / (float)nvl : (flip ? -mq : mq);
        uq = visits > 0 ? reward / (float)visits : mq;
        const float u = (float)((double)(prior / (float)(1 + visits)) * sq);
        score = o.use_prior ? (u * o.c_puct + q) : q;
        counted = nvl != 0;
      }
      // first strict maximum in edge order == max score, lowest index on ties
      float s = score; int sj = ok ? j : 0x7FFFFFFF;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        float s2 = __shfl_xor(s, off); int j2 = __shfl_xor(sj, off);
        if (s2 > s || (s2 == s && j2 < sj)) { s = s2; sj = j2; }
      }
      if (s > best || (s == best && sj < bestj)) { best = s; bestj = sj; }
      // total_unsigned_q is a float sum in edge order: keep the order
      const int lim = min(64, ne - base);
      for (int k = 0; k < lim; ++k) {
        float v = __shfl(uq, k); int c = __shfl(counted, k);
        if (c) { total_uq += v; total_cnt++; }
      }
    }
    mq = (pq + total_uq) / (float)(total_cnt + 1);  // tree_search_node.h:240-241
    tv.meanQ[node] = mq;
    const int e = eb + bestj;
    if (o.vl > 0) tv.el[e] = tv.el[e] + (float)o.vl;  // addVirtualLoss :246-264
    tn[depth] = (uint16_t)node; te[depth] = (uint16_t)bestj;
    int child = tv.ec[e];
    if (child < 0) {  // followEdge :295-317 + allocateState tree_search.h:209-224
      if (nn >= d.max_nodes) { flags |= FLAG_NODE_OVERFLOW; ++depth; break; }
      child = nn++;
      init_node(tv, child, mq, node);
      Am s; load_board(tv.board + (size_t)node * 6, s);
      const int action = tv.ea[e];
      am_play(s, action, c_acttab.fam[action], c_acttab.shift[action]);
      store_board(tv.board + (size_t)child * 6, s);
      tv.meta[child] = NOT_VISITED | META_STATE_SET;
      tv.ec[e] = child;
    }
    __threadfence_block();
    node = child;
    ++depth;
  }
  d.traj_len[(size_t)t * o.B + slot] = depth;
  depth_out = depth;
  return node;
}

struct ExpandScratch {
  stdorder::Item items[NA + 2];
  uint16_t keys[MAXLEGAL]; float pri[MAXLEGAL]; int16_t nxt[MAXLEGAL]; int16_t buckets[128]; uint16_t pos[MAXLEGAL];
  int m;
};

// setEvaluation (tree_search_node.h:188-215) of a locked leaf from the actor's response
// (MCTSGameActor.h:198-335).  pi == nullptr selects the uniform evaluator (pi = 1/170, V = 0).
__device__ inline void expand_leaf(const TreeDev& d, TV& tv, int leaf, const float* __restrict__ pi, float vin, int lane,
                                   int& ne_cnt, int& flags, ExpandScratch& S, unsigned long long& evals,
                                   unsigned long long& edge_sum) {
  Am s; load_board(tv.board + (size_t)leaf * 6, s);
  uint64_t fam[8];
  am_families(s, fam);
  const int total = am_count(fam);
  const bool flip = am_active(s) == 1;                       // q_flip = currentPlayer == WHITE
  const bool terminal = total == 0 || am_ply(s) >= 250;      // GameState.h:51-53
  float value; int m = 0;
  const int eb = ne_cnt;
  if (terminal) {
    // evaluateGame GameState.cc:65-77: -1 at the move cap, else -1 if black is stuck, +1 if white is
    value = (am_ply(s) < 250 && am_active(s) == 1) ? 1.0f : -1.0f;
  } else {
    value = vin;
    evals += 1;
    am_apply_repeat_rule(s, fam, total);
    uint64_t mk[3];
    am_mask_words(fam, mk);
    if (pi) {
      for (int i = lane; i < NA; i += 64) { S.items[i].key = pi[i]; S.items[i].idx = (uint16_t)i; }
    } else {
      for (int i = lane; i < NA; i += 64) { S.items[i].key = 1.0f / 170.0f; S.items[i].idx = d.uniform_order[i]; }
    }
    __syncthreads();
    if (lane == 0) {
      if (pi) stdorder::sort_desc(S.items, NA);              // std::sort, MCTSGameActor.h:295-300
      int mm = 0; bool over = false;
      for (int i = 0; i < NA; ++i) {                          // checkMove filter :305-331
        const int a = S.items[i].idx;
        if ((mk[a >> 6] >> (a & 63)) & 1) {
          if (mm < MAXLEGAL) { S.keys[mm] = (uint16_t)a; S.pri[mm] = S.items[i].key; ++mm; } else over = true;
        }
      }
      float tot = 1e-10f;                                     // normalize :240-251
      for (int k = 0; k < mm; ++k) tot += S.pri[k];
      for (int k = 0; k < mm; ++k) S.pri[k] /= tot;
      stdorder::umap_iteration_order(S.keys, mm, S.nxt, S.buckets, S.pos);
      S.m = over ? -1 : mm;
    }
    __syncthreads();
    m = S.m;
    if (m < 0) { flags |= FLAG_LEGAL_OVERFLOW; m = 0; }
    if (eb + m > d.max_edges) { flags |= FLAG_EDGE_OVERFLOW; m = 0; }
    if (lane < m) {
      const int p = S.pos[lane];
      tv.ea[eb + lane] = S.keys[p]; tv.ep[eb + lane] = S.pri[p];
      tv.ec[eb + lane] = -1; tv.er[eb + lane] = 0.f; tv.ev[eb + lane] = 0; tv.el[eb + lane] = 0.f;
    }
    __syncthreads();
    ne_cnt += m;
    edge_sum += (unsigned long long)m;
  }
  tv.V[leaf] = value;
  tv.ebase[leaf] = (uint32_t)eb;
  tv.meta[leaf] = VISITED | (flip ? META_FLIP : 0u) | META_STATE_SET | ((uint32_t)m << 16);
  __threadfence_block();
}

// backup of one batch (tree_search.h:291-305 + updateEdgeStats tree_search_node.h:267-292):
// per DISTINCT leaf, along the first trajectory that reached it; duplicates only raise `count`.
__device__ inline void backup_batch(const TreeDev& d, TV& tv, const Opt& o, int t, const int* leafs) {
  for (int j = 0; j < o.B; ++j) {
    const int leaf = leafs[j];
    bool first = true; int count = 0;
    for (int k = 0; k < o.B; ++k) if (leafs[k] == leaf) { if (k < j) first = false; ++count; }
    if (!first) continue;
    const float reward = tv.V[leaf];
    const int len = d.traj_len[(size_t)t * o.B + j];
    const uint16_t* tn = d.traj_node + ((size_t)t * o.B + j) * MAXD;
    const uint16_t* te = d.traj_edge + ((size_t)t * o.B + j) * MAXD;
    for (int i = 0; i < len; ++i) {
      const int node = tn[i];
      const int e = (int)tv.ebase[node] + te[i];
      tv.visits[node] = tv.visits[node] + 1;
      tv.er[e] = tv.er[e] + reward;
      tv.ev[e] = tv.ev[e] + 1;
      tv.el[e] = tv.el[e] - (float)(o.vl * count);
    }
  }
  __threadfence_block();
}

// ---- kernels (grid = n_trees blocks of one wave) ------------------------------------------------

__global__ void __launch_bounds__(64) mcts_clear_kernel(TreeDev d, const uint8_t* __restrict__ which) {
  const int t = blockIdx.x;
  if (which && !which[t]) return;
  TV tv = tree_view(d, t, 0);
  if (threadIdx.x == 0) {
    init_node(tv, 0, 0.f, -1);
    d.t_arena[t] = 0; d.t_nnodes[t] = 1; d.t_nedges[t] = 0; d.t_flags[t] = 0; d.t_hmeta[t] = 0;
  }
}

// TreeSearchT::setRootNodeState tree_search.h:570-585 (+ align_state's clear when not persistent)
__global__ void __launch_bounds__(64)
mcts_set_roots_kernel(TreeDev d, int persistent, const uint64_t* __restrict__ w, const uint64_t* __restrict__ hist,
                      const uint32_t* __restrict__ hmeta, int n, uint8_t* __restrict__ ok) {
  const int t = blockIdx.x;
  if (threadIdx.x != 0) return;
  if (!persistent) {
    TV c = tree_view(d, t, 0);
    init_node(c, 0, 0.f, -1);
    d.t_arena[t] = 0; d.t_nnodes[t] = 1; d.t_nedges[t] = 0; d.t_flags[t] &= ~FLAG_ROOT_MISMATCH;
  }
  TV tv = tree_view(d, t, d.t_arena[t]);
  uint64_t b[6];
  for (int k = 0; k < 6; ++k) b[k] = w[(size_t)k * n + t];
  bool good = true;
  if (tv.meta[0] & META_STATE_SET) {
    for (int k = 0; k < 6; ++k) good = good && tv.board[k] == b[k];
    if (!good) d.t_flags[t] |= FLAG_ROOT_MISMATCH;
  } else {
    for (int k = 0; k < 6; ++k) tv.board[k] = b[k];
    tv.meta[0] |= META_STATE_SET;
  }
  // the (up to) five boards that preceded the root, most recent first, for the feature planes
  const uint32_t hm = hmeta[t];
  const int len = hm & 7, head = (hm >> 3) & 7;
  const uint32_t abits = (hm >> 8) & 0x3F;
  uint32_t cnt = 0, ab = 0;
  for (int i = 0; i < 5; ++i) {
    const int j = len - 2 - i;
    if (j < 0) break;
    const int hs = (head + j) % 6;
    for (int k = 0; k < 4; ++k) d.t_hist[((size_t)t * 5 + i) * 4 + k] = hist[((size_t)hs * 4 + k) * n + t];
    ab |= ((abits >> hs) & 1u) << i;
    ++cnt;
  }
  d.t_hmeta[t] = cnt | (ab << 8);
  if (ok) ok[t] = good ? 1 : 0;
}

// BoardFeature planes of a leaf: its own board and up to five ancestors (then the root's history)
__device__ inline void emit_leaf_features(const TreeDev& d, const TV& tv, int t, int leaf, int lane, float* __restrict__ out) {
  int node = leaf, hidx = 0;
  const uint32_t hm = d.t_hmeta[t];
  const int hcnt = hm & 7;
  for (int slot = 5; slot >= 0; --slot) {
    float pl[6] = {0, 0, 0, 0, 0, 0};
    if (node >= 0) {
      const uint64_t* b = tv.board + (size_t)node * 6;
      am_slot_planes(b[0], b[1], b[2], b[3], (int)(b[4] & 1), lane, pl);
      node = tv.parent[node];
    } else if (hidx < hcnt) {
      const uint64_t* b = d.t_hist + ((size_t)t * 5 + hidx) * 4;
      am_slot_planes(b[0], b[1], b[2], b[3], (int)((hm >> (8 + hidx)) & 1), lane, pl);
      ++hidx;
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) out[(size_t)(slot * 6 + q) * 64 + lane] = pl[q];
  }
}

// MODE 0: fused uniform-policy search (all batches of one run() in one launch)
// MODE 1: select one batch, emit leaves for an external evaluator
// MODE 2: expand locked leaves from the evaluator's (pi, V) and back up the batch
template <int MODE>
__global__ void __launch_bounds__(64)
mcts_search_kernel(TreeDev d, Opt o, uint64_t* __restrict__ leaf_states, float* __restrict__ leaf_feat,
                   uint8_t* __restrict__ need_eval, const float* __restrict__ pi, const float* __restrict__ vin) {
  __shared__ ExpandScratch S;
  __shared__ int s_leaf[MAXB];
  __shared__ int s_locked[MAXB];
  const int t = blockIdx.x, lane = threadIdx.x;
  int flags = d.t_flags[t];
  if (flags) return;  // overflowed / mismatched trees are parked until cleared
  TV tv = tree_view(d, t, d.t_arena[t]);
  int nn = d.t_nnodes[t], ne = d.t_nedges[t];
  unsigned long long rollouts = 0, evals = 0, depth_sum = 0, edge_sum = 0;
  const int nbatch = MODE == 0 ? (o.R + o.B - 1) / o.B : 1;
  for (int batch = 0; batch < nbatch && !flags; ++batch) {
    if (MODE != 2) {
      for (int j = 0; j < o.B; ++j) {
        int depth = 0;
        const int leaf = select_one(d, tv, o, t, j, lane, nn, flags, depth);
        rollouts += 1; depth_sum += (unsigned long long)depth;
        // requestEvaluation tree_search_node.h:169-179
        const uint32_t meta = tv.meta[leaf];
        const int lock = (meta & 0xFF) == NOT_VISITED;
        if (lock) tv.meta[leaf] = (meta & ~0xFFu) | EVAL_REQUESTED;
        __threadfence_block();
        if (lane == 0) { s_leaf[j] = leaf; s_locked[j] = lock; }
        if (MODE == 1) {
          if (lane == 0) { d.leaf[(size_t)t * o.B + j] = leaf; d.locked[(size_t)t * o.B + j] = (uint8_t)lock; }
          const size_t slot = (size_t)t * o.B + j;
          if (leaf_states && lane < 6) leaf_states[slot * 6 + lane] = tv.board[(size_t)leaf * 6 + lane];
          if (leaf_feat) emit_leaf_features(d, tv, t, leaf, lane, leaf_feat + slot * 36 * 64);
          if (need_eval && lane == 0) {
            Am s; load_board(tv.board + (size_t)leaf * 6, s);
            uint64_t fam[8]; am_families(s, fam);
            const bool terminal = am_count(fam) == 0 || am_ply(s) >= 250;
            need_eval[slot] = (lock && !terminal) ? 1 : 0;
          }
        }
      }
      __syncthreads();
    } else {
      if (lane < o.B) { s_leaf[lane] = d.leaf[(size_t)t * o.B + lane]; s_locked[lane] = d.locked[(size_t)t * o.B + lane]; }
      __syncthreads();
    }
    if (MODE != 1) {
      for (int j = 0; j < o.B; ++j) {
        if (!s_locked[j]) continue;
        const size_t slot = (size_t)t * o.B + j;
        expand_leaf(d, tv, s_leaf[j], MODE == 2 ? pi + slot * NA : nullptr, MODE == 2 ? vin[slot] : 0.0f, lane, ne, flags,
                    S, evals, edge_sum);
      }
      backup_batch(d, tv, o, t, s_leaf);
      __syncthreads();
    }
  }
  if (lane == 0) {
    d.t_nnodes[t] = nn; d.t_nedges[t] = ne; d.t_flags[t] = flags;
    if (rollouts) atomicAdd(&d.stats[0], rollouts);
    if (evals) atomicAdd(&d.stats[1], evals);
    if (depth_sum) atomicAdd(&d.stats[2], depth_sum);
    if (edge_sum) atomicAdd(&d.stats[3], edge_sum);
  }
}

// chooseAction + MCTSResultT::addActions (tree_search.h:588-621, tree_search_base.h:310-367) and
// MCTSGameAI::getValue (MCTSGameAI.h:40-46).  Edges are reported in iteration order.
__global__ void __launch_bounds__(64)
mcts_result_kernel(TreeDev d, int pick, int16_t* __restrict__ best_out, int32_t* __restrict__ total_out,
                   float* __restrict__ bestq_out, float* __restrict__ rootv_out, int32_t* __restrict__ nedges_out,
                   int16_t* __restrict__ ea_out, int32_t* __restrict__ ev_out, float* __restrict__ er_out,
                   float* __restrict__ ep_out, int cap, int32_t* __restrict__ policy_out) {
  const int t = blockIdx.x, lane = threadIdx.x;
  TV tv = tree_view(d, t, d.t_arena[t]);
  const uint32_t meta = tv.meta[0];
  const int ne = (meta & 0xFF) == VISITED ? (int)(meta >> 16) : 0;
  const int eb = (int)tv.ebase[0];
  float best = -INFINITY; int bestj = 0x7FFFFFFF; int total = 0;
  if (policy_out) for (int a = lane; a < NA; a += 64) policy_out[(size_t)t * NA + a] = 0;
  __syncthreads();
  for (int base = 0; base < ne; base += 64) {
    const int j = base + lane; const bool ok = j < ne;
    int visits = 0; float score = -INFINITY;
    if (ok) {
      visits = tv.ev[eb + j];
      score = pick == 0 ? (float)visits : tv.ep[eb + j];
      if (j < cap) {
        if (ea_out) ea_out[(size_t)t * cap + j] = (int16_t)tv.ea[eb + j];
        if (ev_out) ev_out[(size_t)t * cap + j] = visits;
        if (er_out) er_out[(size_t)t * cap + j] = tv.er[eb + j];
        if (ep_out) ep_out[(size_t)t * cap + j] = tv.ep[eb + j];
      }
      if (policy_out) policy_out[(size_t)t * NA + tv.ea[eb + j]] = visits;
    }
    float s = score; int sj = ok ? j : 0x7FFFFFFF; int tv_sum = visits;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      float s2 = __shfl_xor(s, off); int j2 = __shfl_xor(sj, off);
      if (s2 > s || (s2 == s && j2 < sj)) { s = s2; sj = j2; }
      tv_sum += __shfl_xor(tv_sum, off);
    }
    if (s > best || (s == best && sj < bestj)) { best = s; bestj = sj; }
    total += tv_sum;
  }
  if (lane == 0) {
    const float rootv = tv.V[0];
    if (ne == 0) {
      if (best_out) best_out[t] = -1;
      if (bestq_out) bestq_out[t] = rootv;
    } else {
      if (best_out) best_out[t] = (int16_t)tv.ea[eb + bestj];
      if (bestq_out) bestq_out[t] = total == 0 ? rootv : tv.er[eb + bestj] / (float)tv.ev[eb + bestj];
    }
    if (total_out) total_out[t] = total;
    if (rootv_out) rootv_out[t] = rootv;
    if (nedges_out) nedges_out[t] = ne;
  }
}

// mcts_make_diverse_move / mcts_update_info (ClientGameSelfPlay.cc:94-122) + addMCTSPolicy
// (GameStateExt.h:79-100) + MCTSPolicy::normalize (tree_search_base.h:251-262)
__global__ void __launch_bounds__(64)
mcts_pick_kernel(TreeDev d, const int32_t* __restrict__ ply, int cutoff, uint64_t seed, uint64_t counter,
                 int16_t* __restrict__ move, uint8_t* __restrict__ pol, float* __restrict__ value) {
  const int t = blockIdx.x, lane = threadIdx.x;
  TV tv = tree_view(d, t, d.t_arena[t]);
  const uint32_t meta = tv.meta[0];
  const int ne = (meta & 0xFF) == VISITED ? (int)(meta >> 16) : 0;
  const int eb = (int)tv.ebase[0];
  if (pol) for (int a = lane; a < NA; a += 64) pol[(size_t)t * NA + a] = 0;
  __syncthreads();
  // pass 1: totals, most-visited edge (first strict max in edge order)
  int total = 0, bestv = -1, bestj = 0x7FFFFFFF;
  for (int base = 0; base < ne; base += 64) {
    const int j = base + lane;
    int v = j < ne ? tv.ev[eb + j] : -1, vj = j < ne ? j : 0x7FFFFFFF, sum = j < ne ? v : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      int v2 = __shfl_xor(v, off), j2 = __shfl_xor(vj, off);
      if (v2 > v || (v2 == v && j2 < vj)) { v = v2; vj = j2; }
      sum += __shfl_xor(sum, off);
    }
    if (v > bestv || (v == bestv && vj < bestj)) { bestv = v; bestj = vj; }
    total += sum;
  }
  const bool diverse = ply && ply[t] <= cutoff && total > 0;
  const uint64_t r = diverse ? elf_splitmix64(seed + (uint64_t)t * ELF_GOLDEN + counter) % (uint64_t)total : 0;
  const float exp_sum = (float)total, maxp = total > 0 ? (float)bestv / exp_sum : 0.f;
  int chosen = ne > 0 ? bestj : -1, cum = 0;
  bool picked = false;
  for (int base = 0; base < ne; base += 64) {
    const int j = base + lane;
    const int v = j < ne ? tv.ev[eb + j] : 0;
    if (pol && j < ne) {
      const float p = (float)v / exp_sum;
      const float q = p / maxp * 255.0f;
      pol[(size_t)t * NA + tv.ea[eb + j]] = total > 0 ? (uint8_t)q : 0;
    }
    if (diverse && !picked) {
      int incl = v;  // inclusive prefix sum over the wave
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { int x = __shfl_up(incl, off); if (lane >= off) incl += x; }
      const unsigned long long hit = __ballot(j < ne && (uint64_t)(cum + incl) > r);
      if (hit) { chosen = base + (__ffsll((long long)hit) - 1); picked = true; }
      cum += __shfl(incl, 63);
    }
  }
  if (lane == 0) {
    move[t] = chosen < 0 ? (int16_t)-1 : (int16_t)tv.ea[eb + chosen];
    if (value) value[t] = (ne == 0 || total == 0) ? tv.V[0] : tv.er[eb + bestj] / (float)tv.ev[eb + bestj];
  }
}

// TreeT::treeAdvance tree_search_node.h:458-474: keep the chosen child's subtree, drop the rest.
// The kept subtree is copied breadth-first into the other arena (new indices, same contents).
__global__ void __launch_bounds__(64) mcts_advance_kernel(TreeDev d, const int16_t* __restrict__ actions) {
  extern __shared__ uint16_t s_map[];  // new index -> old index  [max_nodes]
  __shared__ int s_child;
  const int t = blockIdx.x, lane = threadIdx.x;
  const int a = actions[t];
  if (a < 0) return;
  const int arena = d.t_arena[t];
  TV o = tree_view(d, t, arena), n = tree_view(d, t, arena ^ 1);
  const uint32_t meta = o.meta[0];
  const int ne0 = (meta & 0xFF) == VISITED ? (int)(meta >> 16) : 0;
  const int eb0 = (int)o.ebase[0];
  if (lane == 0) s_child = -1;
  __syncthreads();
  for (int j = lane; j < ne0; j += 64) if (o.ea[eb0 + j] == (uint16_t)a) s_child = o.ec[eb0 + j];
  __syncthreads();
  const int c = s_child;
  if (c < 0) {  // never expanded: fresh root (allocateRoot -> addNode(0.0))
    if (lane == 0) { init_node(n, 0, 0.f, -1); d.t_arena[t] = arena ^ 1; d.t_nnodes[t] = 1; d.t_nedges[t] = 0; }
    return;
  }
  auto copy_node = [&](int dst, int src, int parent) {
    n.meta[dst] = o.meta[src]; n.visits[dst] = o.visits[src]; n.V[dst] = o.V[src];
    n.meanQ[dst] = o.meanQ[src]; n.parentQ[dst] = o.parentQ[src]; n.parent[dst] = parent;
    for (int k = 0; k < 6; ++k) n.board[(size_t)dst * 6 + k] = o.board[(size_t)src * 6 + k];
  };
  if (lane == 0) { copy_node(0, c, -1); s_map[0] = (uint16_t)c; }
  __syncthreads();
  int count = 1, ecount = 0;
  for (int i = 0; i < count; ++i) {
    const int old = s_map[i];
    const uint32_t m = o.meta[old];
    const int ne = (m & 0xFF) == VISITED ? (int)(m >> 16) : 0;
    const int eb = (int)o.ebase[old];
    if (lane == 0) n.ebase[i] = (uint32_t)ecount;
    for (int base = 0; base < ne; base += 64) {
      const int j = base + lane; const bool ok = j < ne;
      int oc = -1;
      if (ok) {
        oc = o.ec[eb + j];
        n.ea[ecount + j] = o.ea[eb + j]; n.ep[ecount + j] = o.ep[eb + j]; n.er[ecount + j] = o.er[eb + j];
        n.ev[ecount + j] = o.ev[eb + j]; n.el[ecount + j] = o.el[eb + j];
      }
      const unsigned long long has = __ballot(ok && oc >= 0);
      if (ok) {
        int ni = -1;
        if (oc >= 0) {
          ni = count + __popcll(has & ((1ull << lane) - 1));
          s_map[ni] = (uint16_t)oc;
          copy_node(ni, oc, i);
        }
        n.ec[ecount + j] = ni;
      }
      count += __popcll(has);
    }
    ecount += ne;
    __syncthreads();
  }
  if (lane == 0) { d.t_arena[t] = arena ^ 1; d.t_nnodes[t] = count; d.t_nedges[t] = ecount; }
}

#define ALLOC(ptr, bytes)                                                                       \
  do {                                                                                          \
    if (hipMalloc((void**)&(ptr), (bytes)) != hipSuccess) {                                     \
      elfhip_set_error("mcts: device allocation of %zu bytes failed", (size_t)(bytes));          \
      return ELFHIP_ENOMEM;                                                                     \
    }                                                                                           \
    allocs.push_back((void*)(ptr));                                                             \
  } while (0)

struct MctsImpl {
  TreeDev d{};
  Opt o{};
  int device = 0;
  hipStream_t stream = nullptr;
  std::vector<void*> allocs;
  uint16_t* uniform_order = nullptr;
  ~MctsImpl() { for (void* p : allocs) (void)hipFree(p); }

  int alloc(int n_trees, const elfhip_ts_options* opt) {
    o.R = opt->num_rollouts_per_thread; o.B = opt->num_rollouts_per_batch; o.vl = opt->virtual_loss;
    o.use_prior = opt->use_prior; o.c_puct = opt->c_puct; o.uqz = opt->unexplored_q_zero;
    o.ruqz = opt->root_unexplored_q_zero; o.persistent = opt->persistent_tree; o.pick = opt->pick_method;
    d.n_trees = n_trees;
    d.max_nodes = opt->max_nodes > 0 ? opt->max_nodes : 8 * o.R + 64;
    if (d.max_nodes > 30000) d.max_nodes = 30000;  // u16 node ids; the advance kernel maps them in LDS
    d.max_edges = opt->max_edges > 0 ? opt->max_edges : 12 * d.max_nodes;
    const size_t NN = (size_t)2 * n_trees * d.max_nodes, NE = (size_t)2 * n_trees * d.max_edges, T = (size_t)n_trees;
    ALLOC(d.n_meta, NN * 4); ALLOC(d.n_ebase, NN * 4); ALLOC(d.n_visits, NN * 4); ALLOC(d.n_V, NN * 4);
    ALLOC(d.n_meanQ, NN * 4); ALLOC(d.n_parentQ, NN * 4); ALLOC(d.n_parent, NN * 4); ALLOC(d.n_board, NN * 48);
    ALLOC(d.e_action, NE * 2); ALLOC(d.e_prior, NE * 4); ALLOC(d.e_child, NE * 4); ALLOC(d.e_reward, NE * 4);
    ALLOC(d.e_visits, NE * 4); ALLOC(d.e_vloss, NE * 4);
    ALLOC(d.t_arena, T * 4); ALLOC(d.t_nnodes, T * 4); ALLOC(d.t_nedges, T * 4); ALLOC(d.t_flags, T * 4);
    ALLOC(d.t_hist, T * 5 * 4 * 8); ALLOC(d.t_hmeta, T * 4);
    ALLOC(d.traj_node, T * o.B * MAXD * 2); ALLOC(d.traj_edge, T * o.B * MAXD * 2);
    ALLOC(d.traj_len, T * o.B * 4); ALLOC(d.leaf, T * o.B * 4); ALLOC(d.locked, T * o.B);
    ALLOC(uniform_order, NA * 2); ALLOC(d.stats, 4 * 8);
    d.uniform_order = uniform_order;
    // std::sort order of 170 equal keys, computed once with the same code the kernels run
    stdorder::Item it[NA];
    for (int i = 0; i < NA; ++i) { it[i].key = 1.0f / 170.0f; it[i].idx = (uint16_t)i; }
    stdorder::sort_desc(it, NA);
    uint16_t ord[NA];
    for (int i = 0; i < NA; ++i) ord[i] = it[i].idx;
    HIPCHECK(hipMemcpy(uniform_order, ord, sizeof(ord), hipMemcpyHostToDevice));
    HIPCHECK(hipMemsetAsync(d.stats, 0, 32, stream));
    HIPCHECK(hipMemsetAsync(d.t_flags, 0, T * 4, stream));
    return clear(nullptr);
  }
  int clear(const uint8_t* d_which) {
    hipLaunchKernelGGL(mcts_clear_kernel, dim3(d.n_trees), dim3(64), 0, stream, d, d_which);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
};

}  // namespace

struct elfhip_mcts { MctsImpl* m; };

#define CHECK_M(h)                                                                                  \
  if (!(h) || !(h)->m) { elfhip_set_error("%s: null handle", __func__); return ELFHIP_EINVAL; }       \
  HIPCHECK(hipSetDevice((h)->m->device));

extern "C" {

int elfhip_mcts_create(int game, int n_trees, const elfhip_ts_options* opt, int device, elfhip_mcts** out) {
  if (!out || !opt) { elfhip_set_error("mcts_create: null argument"); return ELFHIP_EINVAL; }
  *out = nullptr;
  if (game != ELFHIP_AMERICAN) { elfhip_set_error("mcts_create: game %d has no search kernels yet (american only)", game); return ELFHIP_EINVAL; }
  if (n_trees <= 0 || n_trees > (1 << 22)) { elfhip_set_error("mcts_create: n_trees=%d out of range", n_trees); return ELFHIP_EINVAL; }
  if (opt->num_rollouts_per_batch < 1 || opt->num_rollouts_per_batch > MAXB) {
    elfhip_set_error("mcts_create: num_rollouts_per_batch=%d not in [1,%d]", opt->num_rollouts_per_batch, MAXB);
    return ELFHIP_EINVAL;
  }
  if (opt->num_rollouts_per_thread < 1 || opt->virtual_loss < 0 || opt->pick_method < 0 || opt->pick_method > 1) {
    elfhip_set_error("mcts_create: bad rollouts / virtual_loss / pick_method");
    return ELFHIP_EINVAL;
  }
  int ndev = elfhip_device_count();
  if (ndev < 0) return ndev;
  if (device < 0 || device >= ndev) { elfhip_set_error("mcts_create: device %d not in [0,%d)", device, ndev); return ELFHIP_EINVAL; }
  HIPCHECK(hipSetDevice(device));
  MctsImpl* m = new (std::nothrow) MctsImpl();
  if (!m) { elfhip_set_error("mcts_create: host allocation failed"); return ELFHIP_ENOMEM; }
  m->device = device;
  int rc = m->alloc(n_trees, opt);
  if (rc != ELFHIP_OK) { delete m; return rc; }
  *out = new elfhip_mcts{m};
  return ELFHIP_OK;
}

int elfhip_mcts_destroy(elfhip_mcts* h) {
  if (!h) return ELFHIP_OK;
  if (h->m) { (void)hipSetDevice(h->m->device); (void)hipStreamSynchronize(h->m->stream); delete h->m; }
  delete h;
  return ELFHIP_OK;
}

int elfhip_mcts_set_stream(elfhip_mcts* h, void* s) { CHECK_M(h); h->m->stream = (hipStream_t)s; return ELFHIP_OK; }

int elfhip_mcts_clear(elfhip_mcts* h, const uint8_t* d_which) { CHECK_M(h); return h->m->clear(d_which); }

int elfhip_mcts_set_roots(elfhip_mcts* h, elfhip_boards* b, uint8_t* d_ok) {
  CHECK_M(h);
  if (!b || !b->slab || b->slab->game != ELFHIP_AMERICAN || b->slab->n != h->m->d.n_trees || b->slab->device != h->m->device) {
    elfhip_set_error("mcts_set_roots: boards handle must be american checkers with n == n_trees on the same device");
    return ELFHIP_EINVAL;
  }
  MctsImpl* m = h->m;
  hipLaunchKernelGGL(mcts_set_roots_kernel, dim3(m->d.n_trees), dim3(64), 0, m->stream, m->d, m->o.persistent,
                     (const uint64_t*)b->slab->dev_ptr[0], (const uint64_t*)b->slab->dev_ptr[1],
                     (const uint32_t*)b->slab->dev_ptr[2], b->slab->n, d_ok);
  HIPCHECK(hipGetLastError());
  return ELFHIP_OK;
}

int elfhip_mcts_run_uniform(elfhip_mcts* h) {
  CHECK_M(h);
  MctsImpl* m = h->m;
  hipLaunchKernelGGL(mcts_search_kernel<0>, dim3(m->d.n_trees), dim3(64), 0, m->stream, m->d, m->o, (uint64_t*)nullptr,
                     (float*)nullptr, (uint8_t*)nullptr, (const float*)nullptr, (const float*)nullptr);
  HIPCHECK(hipGetLastError());
  return ELFHIP_OK;
}

int elfhip_mcts_select(elfhip_mcts* h, uint64_t* d_leaf_states, float* d_leaf_features, uint8_t* d_need_eval) {
  CHECK_M(h);
  MctsImpl* m = h->m;
  hipLaunchKernelGGL(mcts_search_kernel<1>, dim3(m->d.n_trees), dim3(64), 0, m->stream, m->d, m->o, d_leaf_states,
                     d_leaf_features, d_need_eval, (const float*)nullptr, (const float*)nullptr);
  HIPCHECK(hipGetLastError());
  return ELFHIP_OK;
}

int elfhip_mcts_backup(elfhip_mcts* h, const float* d_pi, const float* d_value) {
  CHECK_M(h);
  if (!d_pi || !d_value) { elfhip_set_error("mcts_backup: null pi / value"); return ELFHIP_EINVAL; }
  MctsImpl* m = h->m;
  hipLaunchKernelGGL(mcts_search_kernel<2>, dim3(m->d.n_trees), dim3(64), 0, m->stream, m->d, m->o, (uint64_t*)nullptr,
                     (float*)nullptr, (uint8_t*)nullptr, d_pi, d_value);
  HIPCHECK(hipGetLastError());
  return ELFHIP_OK;
}

int elfhip_mcts_result(elfhip_mcts* h, int16_t* d_best, int32_t* d_total_visits, float* d_best_q, float* d_root_value,
                       int32_t* d_n_edges, int16_t* d_edge_action, int32_t* d_edge_visits, float* d_edge_reward,
                       float* d_edge_prior, int edge_cap, int32_t* d_policy_visits) {
  CHECK_M(h);
  if (edge_cap < 0) { elfhip_set_error("mcts_result: negative edge_cap"); return ELFHIP_EINVAL; }
  MctsImpl* m = h->m;
  hipLaunchKernelGGL(mcts_result_kernel, dim3(m->d.n_trees), dim3(64), 0, m->stream, m->d, m->o.pick, d_best, d_total_visits,
                     d_best_q, d_root_value, d_n_edges, d_edge_action, d_edge_visits, d_edge_reward, d_edge_prior, edge_cap,
                     d_policy_visits);
  HIPCHECK(hipGetLastError());
  return ELFHIP_OK;
}

int elfhip_mcts_pick_moves(elfhip_mcts* h, const int32_t* d_ply, int policy_distri_cutoff, uint64_t seed, uint64_t counter,
                           int16_t* d_move, uint8_t* d_policy_u8, float* d_value) {
  CHECK_M(h);
  if (!d_move) { elfhip_set_error("mcts_pick_moves: null d_move"); return ELFHIP_EINVAL; }
  MctsImpl* m = h->m;
  hipLaunchKernelGGL(mcts_pick_kernel, dim3(m->d.n_trees), dim3(64), 0, m->stream, m->d, d_ply, policy_distri_cutoff, seed,
                     counter, d_move, d_policy_u8, d_value);
  HIPCHECK(hipGetLastError());
  return ELFHIP_OK;
}

int elfhip_mcts_advance(elfhip_mcts* h, const int16_t* d_actions) {
  CHECK_M(h);
  if (!d_actions) { elfhip_set_error("mcts_advance: null actions"); return ELFHIP_EINVAL; }
  MctsImpl* m = h->m;
  hipLaunchKernelGGL(mcts_advance_kernel, dim3(m->d.n_trees), dim3(64), (size_t)m->d.max_nodes * 2, m->stream, m->d, d_actions);
  HIPCHECK(hipGetLastError());
  return ELFHIP_OK;
}

int elfhip_mcts_stats(elfhip_mcts* h, unsigned long long* h_stats4, int32_t* d_flags) {
  CHECK_M(h);
  MctsImpl* m = h->m;
  if (d_flags) HIPCHECK(hipMemcpyAsync(d_flags, m->d.t_flags, (size_t)m->d.n_trees * 4, hipMemcpyDeviceToDevice, m->stream));
  if (h_stats4) {
    HIPCHECK(hipMemcpyAsync(h_stats4, m->d.stats, 32, hipMemcpyDeviceToHost, m->stream));
  }
  HIPCHECK(hipStreamSynchronize(m->stream));
  return ELFHIP_OK;
}

}  // extern "C"
