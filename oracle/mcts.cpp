// TEST INFRASTRUCTURE ONLY (oracle).  CPU restatement of the reference's tree search, num_threads = 1,
// as a template over the game (American checkers: pinned as below; Go 19x19: the same search over the
// BUILD-DEFINED Go rules of oracle/go.c - there is no Go in the reference, parity vs ELF unpinned):
//   elf/ai/tree_search/tree_search.h:112-152,241-378 (run / batch_rollouts / single_rollout),
//   tree_search_node.h:98-422,436-609 (NodeT, TreeT), tree_search_base.h:137-216,284-367
//   (EdgeInfo::getScore, MCTSResultT::addActions), mcts.h:81-105,167-202 (act / align_state),
//   elfgames/american_checkers/mcts/MCTSGameActor.h:198-335 (pre_evaluate, pi2response).
// The reference headers cannot be compiled here (spdlog / nlohmann-json / tbb / concurrentqueue
// are un-vendored), so this is a restatement, pinned by the known answers the survey captured
// from the compiled reference (tests/golden/mcts_appendix_c.json).
// It deliberately uses the SAME libstdc++ containers as the reference (std::unordered_map for the
// edges, std::sort for pi) so iteration order and tie-breaking are the library's own; the HIP
// path re-implements both orders explicitly and is checked against this.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <random>
#include <unordered_map>
#include <utility>
#include <vector>

#include "american.h"
#include "go.h"
#include "russian.h"
#include "ugolki.h"

namespace {

using Action = unsigned short;  // Coord, GameBoard.h:51

struct TsOptions {  // TSOptions / SearchAlgoOptions, tree_search_options.h:24-29,91-110
  int num_rollouts;            // num_rollouts_per_thread (num_threads == 1)
  int num_rollouts_per_batch;
  int virtual_loss;
  int use_prior;
  float c_puct;
  int unexplored_q_zero, root_unexplored_q_zero;
  int persistent_tree;
  int pick_method;             // 0 most_visited, 1 strongest_prior
};

struct Edge {  // EdgeInfo tree_search_base.h:137-159
  float prior; int child; float reward; int num_visits; float virtual_loss;
  explicit Edge(float p) : prior(p), child(-1), reward(0), num_visits(0), virtual_loss(0) {}
};

template <class State>
struct NodeT {  // NodeT tree_search_node.h:98-115,323-334
  int status = 0;  // 0 NOT_VISITED, 1 EVAL_REQUESTED, 2 VISITED
  int numVisits = 0;
  float V = 0.f, meanQ = 0.f, parentQ = 0.f;
  bool flip = false;
  std::unordered_map<Action, Edge> edges;
  std::unique_ptr<State> state;
  explicit NodeT(float pq) : meanQ(pq), parentQ(pq) {}
};

struct Response { std::vector<std::pair<Action, float>> pi; float value = 0; bool q_flip = false; };

uint64_t words_hash(const uint64_t* w, int n) {
  uint64_t h = 0x1234567ull;
  for (int i = 0; i < n; ++i) h = elf_splitmix64(h ^ w[i]);
  return h;
}

void scripted(uint64_t h, int na, float* pi, float* v) {
  for (int i = 0; i < na; ++i) {
    uint64_t r = elf_splitmix64(h + (uint64_t)i);
    pi[i] = (float)((r & 1023) + 1) / 1024.0f;
  }
  uint64_t r = elf_splitmix64(h + 1000);
  *v = ((float)(r & 2047) - 1024.0f) / 1024.0f;
}

}  // namespace

extern "C" {

// Scripted network stand-in: pi on a 2^-10 grid (ties do occur), V on a 2^-10 grid in (-1,1), so
// every float sum in the search is exact and the result is independent of backup order.
void omcts_scripted_policy(const uint64_t* board6, uint64_t seed, float* pi170, float* v) {
  scripted(words_hash(board6, 6) ^ seed, AM_NUM_ACTIONS, pi170, v);
}
// the same for Go, keyed on the 22 exported state words (go_export)
void ogomcts_scripted_policy(const uint64_t* state22, uint64_t seed, float* pi362, float* v) {
  scripted(words_hash(state22, GO_STATE_WORDS) ^ seed, GO_ACTIONS, pi362, v);
}

void orumcts_scripted_policy(const uint64_t* w5, uint64_t seed, float* pi280, float* v) {
  scripted(words_hash(w5, 5) ^ seed, RU_NUM_ACTIONS, pi280, v);
}
void ougmcts_scripted_policy(const uint64_t* w4, uint64_t seed, float* pi418, float* v) {
  scripted(words_hash(w4, 4) ^ seed, UG_NUM_ACTIONS, pi418, v);
}

}  // extern "C"

namespace {

// russian_checkers/mcts/CheckersMCTSActor.h:79-335 and ugolki/mcts/MCTSGameActor.h:79-335 are the
// American actor with other type names; the state is the bare board (no history in the planes)
struct RuTraits {
  using State = ru_state;
  static constexpr int NA = RU_NUM_ACTIONS, MW = 5;
  static bool flip(const State& s) { return (s.w[4] & 1) == 1; }
  static bool terminated(const State& s) { return ru_terminated(&s); }
  static float evaluate(const State& s) { return ru_evaluate(&s); }
  static void legal(const State& s, uint64_t* m) { ru_legal_mask(&s, m); }
  static bool forward(State& s, int a) { return ru_forward(&s, a); }
  static void policy(const State& s, uint64_t seed, float* pi, float* v) { orumcts_scripted_policy(s.w, seed, pi, v); }
  static bool same(const State& a, const State& b) { return memcmp(a.w, b.w, sizeof(a.w)) == 0; }
  static void words(const State& s, uint64_t* w) { memcpy(w, s.w, sizeof(s.w)); }
};
struct UgTraits {
  using State = ug_state;
  static constexpr int NA = UG_NUM_ACTIONS, MW = 7;
  static bool flip(const State& s) { return (s.w[3] & 1) == 1; }
  static bool terminated(const State& s) { return ug_terminated(&s); }
  static float evaluate(const State& s) { return ug_evaluate(&s); }
  static void legal(const State& s, uint64_t* m) {   // checkMove = TryPlay (ugolki/game/GameState.cc:20-22)
    for (int i = 0; i < MW; ++i) m[i] = 0;
    for (int a = 0; a < NA; ++a) if (ug_try_play(&s, a)) m[a >> 6] |= 1ull << (a & 63);
  }
  static bool forward(State& s, int a) { return ug_forward(&s, a); }
  static void policy(const State& s, uint64_t seed, float* pi, float* v) { ougmcts_scripted_policy(s.w, seed, pi, v); }
  static bool same(const State& a, const State& b) { return memcmp(a.w, b.w, sizeof(a.w)) == 0; }
  static void words(const State& s, uint64_t* w) { memcpy(w, s.w, sizeof(s.w)); }
};

struct AmTraits {
  using State = am_game;
  static constexpr int NA = AM_NUM_ACTIONS, MW = 3;
  static bool flip(const State& s) { return (s.board.w[4] & 1) == 1; }          // currentPlayer == WHITE
  static bool terminated(const State& s) { return am_game_terminated(&s); }
  static float evaluate(const State& s) { return am_game_evaluate(&s); }
  static void legal(const State& s, uint64_t* m) { am_legal_mask(&s.board, m); }
  static bool forward(State& s, int a) { return am_game_forward(&s, a); }
  static void policy(const State& s, uint64_t seed, float* pi, float* v) { omcts_scripted_policy(s.board.w, seed, pi, v); }
  static bool same(const State& a, const State& b) { return memcmp(a.board.w, b.board.w, sizeof(a.board.w)) == 0; }
  static void words(const State& s, uint64_t* w) { memcpy(w, s.board.w, sizeof(s.board.w)); }
};

struct GoTraits {
  using State = go_state;
  static constexpr int NA = GO_ACTIONS, MW = 6;
  static bool flip(const State& s) { return s.to_move == 1; }
  static bool terminated(const State& s) { return go_terminated(&s); }
  static float evaluate(const State& s) { return go_evaluate(&s); }
  static void legal(const State& s, uint64_t* m) { go_legal_mask(&s, m); }
  static bool forward(State& s, int a) { return go_forward(&s, a); }
  static void policy(const State& s, uint64_t seed, float* pi, float* v) {
    uint64_t w[GO_STATE_WORDS]; go_export(&s, w); ogomcts_scripted_policy(w, seed, pi, v);
  }
  static bool same(const State& a, const State& b) {
    uint64_t wa[GO_STATE_WORDS], wb[GO_STATE_WORDS]; go_export(&a, wa); go_export(&b, wb);
    return memcmp(wa, wb, sizeof(wa)) == 0;
  }
  static void words(const State& s, uint64_t* w) { go_export(&s, w); }
};

template <class G>
struct Mcts {
  using State = typename G::State;
  using Node = NodeT<State>;
  TsOptions opt;
  int policy_kind;  // 0 uniform (pi = 1/num_actions, V = 0), 1 scripted, 2 caller's callback (recorded network outputs)
  void (*policy_cb)(const uint64_t* state_words, float* pi, float* v) = nullptr;
  uint64_t policy_seed;
  std::unordered_map<int, std::unique_ptr<Node>> nodes;  // TreeT::allocatedNodes_
  int next_id = 0, root = -1;
  long evals = 0, rollouts = 0, depth_sum = 0, edge_sum = 0;
  float root_epsilon = 0.f, root_alpha = 0.f;
  std::mt19937 rng;  // actors_[0]->rng()
  bool policy_only = false;  // next run() = MCTSAI_T::actPolicyOnly -> TreeSearchT::runPolicyOnly (mcts.h:107-113)

  void enhance_exploration(Node* n) {  // tree_search_node.h:144-167
    if (root_epsilon == 0.0) return;
    std::gamma_distribution<> dis(root_alpha);
    std::vector<float> etas(n->edges.size());
    float Z = 1e-10;
    for (size_t i = 0; i < n->edges.size(); ++i) { etas[i] = dis(rng); Z += etas[i]; }
    int i = 0;
    for (auto& p : n->edges) { p.second.prior = (1 - root_epsilon) * p.second.prior + root_epsilon * etas[i] / Z; i++; }
  }

  Mcts() { clear(); }
  int add_node(float pq) { nodes[next_id].reset(new Node(pq)); return next_id++; }
  void clear() { nodes.clear(); next_id = 0; root = -1; root = add_node(0.f); }  // TreeT::clear
  Node* get(int id) { auto it = nodes.find(id); return it == nodes.end() ? nullptr : it->second.get(); }

  void recursive_free(int id) {  // tree_search_node.h:497-507
    if (id < 0) return;
    Node* n = get(id);
    for (auto& p : n->edges) recursive_free(p.second.child);
    nodes.erase(id);
  }
  void tree_advance(Action a) {  // :458-474
    int next = -1;
    Node* r = get(root);
    for (auto& p : r->edges) { if (p.first == a) next = p.second.child; else recursive_free(p.second.child); }
    nodes.erase(root);
    root = next;
    if (root < 0) root = add_node(0.f);
  }

  // MCTSGameActor.h:198-218 + :256-335
  void evaluate(const State& s, Response* resp) {
    resp->q_flip = G::flip(s);
    resp->pi.clear();
    if (G::terminated(s)) { resp->value = G::evaluate(s) > 0 ? 1.0f : -1.0f; return; }
    ++evals;
    float pi[G::NA], v = 0.f;
    if (policy_kind == 0) { for (int i = 0; i < G::NA; ++i) pi[i] = 1.0f / (float)G::NA; }
    else if (policy_kind == 2) { uint64_t w[32]; G::words(s, w); policy_cb(w, pi, &v); }
    else G::policy(s, policy_seed, pi, &v);
    resp->value = v;
    std::vector<std::pair<Action, float>> all;
    for (int i = 0; i < G::NA; ++i) all.push_back(std::make_pair((Action)i, pi[i]));
    using dt = std::pair<Action, float>;
    std::sort(all.begin(), all.end(), [](const dt& a, const dt& b) { return a.second > b.second; });
    uint64_t m[G::MW]; G::legal(s, m);
    for (auto& p : all) if ((m[p.first >> 6] >> (p.first & 63)) & 1) resp->pi.push_back(p);
    float total = 1e-10;
    for (auto& p : resp->pi) total += p.second;
    for (auto& p : resp->pi) p.second /= total;
  }

  // tree_search_node.h:218-244 + UCT :385-421 + getScore tree_search_base.h:175-201
  bool find_move(Node* n, int depth, Action* out) {
    if (n->status != 2 || n->edges.empty()) return false;
    if (opt.unexplored_q_zero || (opt.root_unexplored_q_zero && depth == 0)) n->meanQ = 0.0f;
    float best = std::numeric_limits<float>::lowest(), total_uq = 0.f;
    Action besta = 0; int total_visits = 0;
    const int all_visits = n->numVisits + 1;
    for (const auto& ap : n->edges) {
      const Edge& e = ap.second;
      float r = e.reward;
      if (n->flip) r = -r;
      r -= e.virtual_loss;
      const int nvl = e.num_visits + e.virtual_loss;   // float add, truncation to int
      float q = nvl > 0 ? r / nvl : (n->flip ? -n->meanQ : n->meanQ);
      float uq = e.num_visits > 0 ? e.reward / e.num_visits : n->meanQ;
      float pp = e.prior / (1 + e.num_visits) * std::sqrt(all_visits);  // double sqrt, one rounding
      float score = opt.use_prior ? (pp * opt.c_puct + q) : q;
      if (score > best) { best = score; besta = ap.first; }
      if (nvl != 0) { total_uq += uq; total_visits++; }
    }
    *out = besta;
    n->meanQ = (n->parentQ + total_uq) / (total_visits + 1);
    return true;
  }

  struct Traj { std::vector<std::pair<Node*, Action>> traj; Node* leaf; };

  Traj single_rollout(Node* rootn) {  // tree_search.h:319-378
    Node* node = rootn; Traj t; int depth = 0;
    while (node->status == 2) {
      Action a;
      if (!find_move(node, depth, &a)) break;
      Edge& e = node->edges.find(a)->second;
      if (opt.virtual_loss > 0) e.virtual_loss += (float)opt.virtual_loss;
      t.traj.push_back(std::make_pair(node, a));
      if (e.child < 0) e.child = add_node(node->meanQ);  // followEdge :295-317
      Node* next = get(e.child);
      if (!next) break;
      if (!next->state) {  // allocateState tree_search.h:209-224
        std::unique_ptr<State> st(new State(*node->state));
        if (!G::forward(*st, a)) break;  // NODE_STATE_INVALID (cannot happen: edges are legal)
        next->state = std::move(st);
      }
      node = next; ++depth;
    }
    t.leaf = node;
    depth_sum += depth; ++rollouts;
    return t;
  }

  void batch_rollouts(Node* rootn) {  // tree_search.h:241-308
    std::vector<Traj> trajs;
    for (int j = 0; j < opt.num_rollouts_per_batch; ++j) trajs.push_back(single_rollout(rootn));
    std::vector<Node*> locked;
    // first-trajectory + count per distinct leaf, kept in first-seen order (the reference walks an
    // unordered_map keyed by Node*; order only matters for float sums, exact in our parity configs)
    std::vector<std::pair<Node*, std::pair<Traj*, int>>> counts;
    for (Traj& t : trajs) {
      if (t.leaf->status == 0) { t.leaf->status = 1; locked.push_back(t.leaf); }
      bool seen = false;
      for (auto& c : counts) if (c.first == t.leaf) { c.second.second++; seen = true; break; }
      if (!seen) counts.push_back(std::make_pair(t.leaf, std::make_pair(&t, 1)));
    }
    for (Node* leaf : locked) {
      Response r; evaluate(*leaf->state, &r);
      for (auto& ap : r.pi) leaf->edges.insert(std::make_pair(ap.first, Edge(ap.second)));  // setEvaluation :188-215
      edge_sum += (long)r.pi.size();
      leaf->V = r.value; leaf->flip = r.q_flip; leaf->status = 2;
    }
    for (auto& c : counts) {
      float reward = c.first->V;
      int count = c.second.second;
      for (auto& p : c.second.first->traj) {  // updateEdgeStats :267-292
        Edge& e = p.first->edges.find(p.second)->second;
        p.first->numVisits++;
        e.reward += reward; e.num_visits++;
        e.virtual_loss -= (float)(opt.virtual_loss * count);
      }
    }
  }
};

template <class G>
void* create(const int* iopt, float c_puct, int policy_kind, uint64_t policy_seed) {
  Mcts<G>* m = new Mcts<G>();
  m->opt = TsOptions{iopt[0], iopt[1], iopt[2], iopt[3], c_puct, iopt[4], iopt[5], iopt[6], iopt[7]};
  m->policy_kind = policy_kind; m->policy_seed = policy_seed;
  return m;
}

// MCTSAI_T::act -> TreeSearchT::run + chooseAction.  Edges are reported in the unordered_map's
// iteration order.  Returns best action (most_visited: first strict max in that order).
template <class G>
int run(Mcts<G>* m, const typename G::State* g, int max_edges, int* n_edges, uint16_t* actions, int* visits,
        float* rewards, float* priors, int* total_visits, float* root_value, float* best_q) {
  using State = typename G::State;
  if (!m->opt.persistent_tree) m->clear();
  auto* rootn = m->get(m->root);
  if (!rootn->state) rootn->state.reset(new State(*g));   // setRootNodeState
  if (!G::same(*rootn->state, *g)) return -2;              // "Root state is not the same"
  if (m->policy_only) {
    // TreeSearchT::runPolicyOnly tree_search.h:475-496: evaluate the root once if it is not visited
    // yet, no noise, no rollouts; the result ranks the root's edges by prior (MCTSResult::PRIOR)
    if (rootn->status != 2) {
      Response r; m->evaluate(*rootn->state, &r);
      for (auto& ap : r.pi) rootn->edges.insert(std::make_pair(ap.first, Edge(ap.second)));
      m->edge_sum += (long)r.pi.size();
      rootn->V = r.value; rootn->flip = r.q_flip; rootn->status = 2;
    }
  } else {
    if (m->root_epsilon > 0.0) m->enhance_exploration(rootn); // TreeSearchT::run tree_search.h:502-506
    for (int idx = 0; idx < m->opt.num_rollouts; idx += m->opt.num_rollouts_per_batch) m->batch_rollouts(rootn);
  }
  const int pick_method = m->policy_only ? 1 : m->opt.pick_method;
  // MCTSResultT::addActions tree_search_base.h:310-367
  float max_score = std::numeric_limits<float>::lowest();
  int best = 0, tv = 0, k = 0; const Edge* beste = nullptr;
  for (const auto& ap : rootn->edges) {
    float score = pick_method == 0 ? (float)ap.second.num_visits : ap.second.prior;
    if (k < max_edges) { actions[k] = ap.first; visits[k] = ap.second.num_visits; rewards[k] = ap.second.reward; priors[k] = ap.second.prior; }
    ++k;
    tv += ap.second.num_visits;
    if (score > max_score) { max_score = score; best = ap.first; beste = &ap.second; }
  }
  *n_edges = k; *total_visits = tv; *root_value = rootn->V;
  // MCTSGameAI::getValue MCTSGameAI.h:40-46
  *best_q = (tv == 0 || !beste) ? rootn->V : beste->reward / beste->num_visits;
  return k == 0 ? -1 : best;
}

}  // namespace

extern "C" {

using AmMcts = Mcts<AmTraits>;
using GoMcts = Mcts<GoTraits>;

void* omcts_create(const int* iopt, float c_puct, int policy_kind, uint64_t policy_seed) { return create<AmTraits>(iopt, c_puct, policy_kind, policy_seed); }
void omcts_set_callback(void* h, void (*cb)(const uint64_t*, float*, float*)) { ((AmMcts*)h)->policy_cb = cb; }
void omcts_set_noise(void* h, float eps, float alpha, unsigned seed) { AmMcts* m = (AmMcts*)h; m->root_epsilon = eps; m->root_alpha = alpha; m->rng.seed(seed); }
void omcts_set_policy_only(void* h, int on) { ((AmMcts*)h)->policy_only = on != 0; }
void omcts_destroy(void* h) { delete (AmMcts*)h; }
void omcts_clear(void* h) { ((AmMcts*)h)->clear(); }                       // MCTSAI_T::resetTree / endGame
void omcts_advance(void* h, int action) { ((AmMcts*)h)->tree_advance((Action)action); }  // advanceMoves
void omcts_stats(void* h, long* out4) { AmMcts* m = (AmMcts*)h; out4[0] = m->rollouts; out4[1] = m->evals; out4[2] = m->depth_sum; out4[3] = m->edge_sum; }

void* ogomcts_create(const int* iopt, float c_puct, int policy_kind, uint64_t policy_seed) { return create<GoTraits>(iopt, c_puct, policy_kind, policy_seed); }
void ogomcts_set_callback(void* h, void (*cb)(const uint64_t*, float*, float*)) { ((GoMcts*)h)->policy_cb = cb; }
void ogomcts_set_noise(void* h, float eps, float alpha, unsigned seed) { GoMcts* m = (GoMcts*)h; m->root_epsilon = eps; m->root_alpha = alpha; m->rng.seed(seed); }
void ogomcts_set_policy_only(void* h, int on) { ((GoMcts*)h)->policy_only = on != 0; }
void ogomcts_destroy(void* h) { delete (GoMcts*)h; }
void ogomcts_clear(void* h) { ((GoMcts*)h)->clear(); }
void ogomcts_advance(void* h, int action) { ((GoMcts*)h)->tree_advance((Action)action); }
void ogomcts_stats(void* h, long* out4) { GoMcts* m = (GoMcts*)h; out4[0] = m->rollouts; out4[1] = m->evals; out4[2] = m->depth_sum; out4[3] = m->edge_sum; }
int ogomcts_run(void* h, const go_state* g, int max_edges, int* n_edges, uint16_t* actions, int* visits, float* rewards,
                float* priors, int* total_visits, float* root_value, float* best_q) {
  return run<GoTraits>((GoMcts*)h, g, max_edges, n_edges, actions, visits, rewards, priors, total_visits, root_value, best_q);
}

// The game's own rng and its uses in the self-play loop, with the library types the reference uses:
// GameBase::_rng (common/GameBase.h:43-49,76), params.seed = _rng() (ClientGameSelfPlay.cc:59) and
// mcts_make_diverse_move (:94-111) = MCTSGameAI::getMCTSPolicy (mcts/MCTSGameAI.h:48-54: visit counts, normalize())
// + MCTSPolicy::sampleAction (tree_search_base.h:251-269) + elf_utils::sample_multinomial (elf/utils/utils.h:162-185).
void* orng_create(unsigned seed) { return new std::mt19937(seed); }
void orng_destroy(void* h) { delete (std::mt19937*)h; }
unsigned orng_next_u32(void* h) { return (unsigned)(*(std::mt19937*)h)(); }
// visits in edge (iteration) order -> index of the sampled edge
int orng_sample_policy(void* h, int n, const int* visits) {
  std::vector<std::pair<Action, float>> policy;
  for (int i = 0; i < n; ++i) policy.push_back(std::make_pair((Action)i, (float)visits[i]));   // MCTSResultT::addActions: score = num_visits
  float exp_sum = 0;                                                                           // normalize(t = 1)
  for (auto& e : policy) { float x = std::pow(e.second, 1.0 / 1); e.second = x; exp_sum += x; }
  for (auto& e : policy) e.second /= exp_sum;
  std::vector<float> accu(policy.size() + 1);                                                  // sample_multinomial
  float Z = 0.0;
  for (const auto& vv : policy) Z += vv.second;
  std::uniform_real_distribution<> dis(0, Z);
  float rd = dis(*(std::mt19937*)h);
  accu[0] = 0;
  for (size_t i = 1; i < accu.size(); i++) {
    accu[i] = policy[i - 1].second + accu[i - 1];
    if (rd < accu[i]) return (int)(i - 1);
  }
  return (int)policy.size() - 1;
}

// One findMove on a hand-built node (tests/golden/uct_known_answers.json): edges are inserted with the given keys
// in the given order; returns the chosen action and the node's new unsignedMeanQ_.
void omcts_test_find_move(int n, const uint16_t* keys, const float* prior, const float* reward, const int* visits,
                          const float* vloss, int flip, float mean_q, float parent_q, int num_visits, int depth,
                          const int* iopt4, float c_puct, int* best_action, float* new_mean_q) {
  AmMcts m;
  m.opt = TsOptions{1, 1, 0, iopt4[0], c_puct, iopt4[1], iopt4[2], 1, 0};
  NodeT<am_game> node(parent_q);
  node.status = 2; node.flip = flip != 0; node.meanQ = mean_q; node.parentQ = parent_q; node.numVisits = num_visits;
  for (int i = 0; i < n; ++i) {
    Edge e(prior[i]); e.reward = reward[i]; e.num_visits = visits[i]; e.virtual_loss = vloss[i];
    node.edges.insert(std::make_pair((Action)keys[i], e));
  }
  Action a = 0;
  m.find_move(&node, depth, &a);
  *best_action = a; *new_mean_q = node.meanQ;
}

int omcts_run(void* h, const am_game* g, int max_edges, int* n_edges, uint16_t* actions, int* visits,
              float* rewards, float* priors, int* total_visits, float* root_value, float* best_q) {
  return run<AmTraits>((AmMcts*)h, g, max_edges, n_edges, actions, visits, rewards, priors, total_visits, root_value, best_q);
}

// libstdc++ behaviours exposed for checking the HIP path's explicit emulations (tests only)
#define ORACLE_MCTS_API(PRE, TRAITS)                                                                              \
  void* PRE##create(const int* iopt, float c_puct, int policy_kind, uint64_t policy_seed) { return create<TRAITS>(iopt, c_puct, policy_kind, policy_seed); } \
  void PRE##set_callback(void* h, void (*cb)(const uint64_t*, float*, float*)) { ((Mcts<TRAITS>*)h)->policy_cb = cb; }  \
  void PRE##set_noise(void* h, float eps, float alpha, unsigned seed) { auto* m = (Mcts<TRAITS>*)h; m->root_epsilon = eps; m->root_alpha = alpha; m->rng.seed(seed); } \
  void PRE##set_policy_only(void* h, int on) { ((Mcts<TRAITS>*)h)->policy_only = on != 0; }                        \
  void PRE##destroy(void* h) { delete (Mcts<TRAITS>*)h; }                                                          \
  void PRE##clear(void* h) { ((Mcts<TRAITS>*)h)->clear(); }                                                        \
  void PRE##advance(void* h, int action) { ((Mcts<TRAITS>*)h)->tree_advance((Action)action); }                      \
  void PRE##stats(void* h, long* out4) { auto* m = (Mcts<TRAITS>*)h; out4[0] = m->rollouts; out4[1] = m->evals; out4[2] = m->depth_sum; out4[3] = m->edge_sum; } \
  int PRE##run(void* h, const TRAITS::State* g, int max_edges, int* n_edges, uint16_t* actions, int* visits, float* rewards, \
               float* priors, int* total_visits, float* root_value, float* best_q) {                               \
    return run<TRAITS>((Mcts<TRAITS>*)h, g, max_edges, n_edges, actions, visits, rewards, priors, total_visits, root_value, best_q); \
  }
ORACLE_MCTS_API(orumcts_, RuTraits)
ORACLE_MCTS_API(ougmcts_, UgTraits)

void omcts_std_sort_desc(int n, const float* keys, uint16_t* order_out) {
  std::vector<std::pair<Action, float>> v;
  for (int i = 0; i < n; ++i) v.push_back(std::make_pair((Action)i, keys[i]));
  using dt = std::pair<Action, float>;
  std::sort(v.begin(), v.end(), [](const dt& a, const dt& b) { return a.second > b.second; });
  for (int i = 0; i < n; ++i) order_out[i] = v[i].first;
}
void omcts_std_umap_order(int n, const uint16_t* keys_in_insert_order, uint16_t* iter_order_out) {
  std::unordered_map<Action, int> m;
  for (int i = 0; i < n; ++i) m.insert(std::make_pair(keys_in_insert_order[i], i));
  int k = 0;
  for (auto& p : m) iter_order_out[k++] = p.first;
}

}  // extern "C"
