// tree_search.h -- the tree search of the reference (elf/tree_search*.h, elf/mcts.h, go/mcts.h)
// for games whose boards live on the GPU: the tree stays on the host, every leaf is evaluated
// in a device batch.
//
// The reference runs `num_threads` OS threads per game; each does `num_rollout_per_thread`
// rollouts, blocking on a condition variable while Python evaluates its leaf.  Here the threads
// of a game are state machines that advance in lockstep: each descends with the reference's UCT
// until it needs a leaf evaluated (or meets a node another thread is having evaluated -- the
// reference's node lock) and parks; the GameContext gathers the parked leaves of ALL games into
// one batch -- child positions are made on the device (goenv_copy_boards + goenv_ply_sel), planes
// written straight into the batch tensor -- and feeds the replies back.  With num_threads = 1
// the order of evaluations, the statistics and the chosen moves are those of the reference
// (tests/golden/mcts_stream.json); with more threads the reference itself is not deterministic.
//
// Kept as in the reference: UCT with its float/double mix (tree_search_alg.h:20-46), edge
// initialisation from the per-thread mt19937(thread_id) (tree_search.h:187-193), the reward of a
// rollout = value of the node it stops at (:124,158-162), AccumulateStats (tree_search_node.h:
// 113-127), MostVisited / StrongestPrior / UniformRandom picks (tree_search_alg.h:48-98), edges
// in a std::unordered_map keyed by the reference's Coord so that ties fall the same way.
#pragma once

#include <cmath>
#include <limits>
#include <random>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace gogame {

// elf/tree_search_options.h:21-55
struct TSOptions {
  int max_num_moves = 0;
  int num_threads = 16;
  int num_rollout_per_thread = 100;
  bool verbose = false;
  bool verbose_time = false;
  std::string save_tree_filename;
  bool persistent_tree = false;
  std::string pick_method = "most_visited";
  bool use_prior = false;
  int pseudo_games = 0;
};

// go/board.h:146-150: Coord = (y+1)*21 + (x+1); 0 = M_PASS.  Actions are p = x*19+y, -1 = pass.
inline unsigned short coord_of_action(int a) {
  if (a < 0) return 0;
  const int x = a / 19, y = a % 19;
  return (unsigned short)((y + 1) * 21 + (x + 1));
}
inline int action_of_coord(unsigned short c) {
  if (c == 0) return -1;
  const int x = c % 21 - 1, y = c / 21 - 1;
  return x * 19 + y;
}

struct EdgeInfo {  // elf/tree_search_base.h:31-47
  float prior = 0.f;
  int next = -1;
  float acc_reward = 0.f;
  int n = 0;
};

struct TreeNode {  // elf/tree_search_node.h:73-170
  bool visited = false;
  bool pending = false;   // a thread is having it evaluated (the reference's lock_node_)
  int slot = -1;          // device board that holds this node's position, -1 = not made yet
  int depth = 0;          // plies below the root
  int count = 0;
  float V = 0.f;
  std::unordered_map<unsigned short, EdgeInfo> sa;
};

// A leaf that wants evaluating.  If parent_slot >= 0 the position does not exist yet: it is
// parent_slot's position after `action`, to be made in `slot`.
struct LeafRequest {
  int thread = 0;
  int node = 0;
  int slot = 0;
  int parent_slot = -1;
  int action = -1;
  int depth = 0;
};

class TreeSearch {
 public:
  TreeSearch(const TSOptions& opt, int pool_base, int pool_size) : opt_(opt), pool_base_(pool_base), pool_size_(pool_size) {
    if (opt_.num_threads < 1) throw std::range_error("mcts_options.num_threads must be >= 1");
    if (opt_.num_rollout_per_thread < 0) throw std::range_error("mcts_options.num_rollout_per_thread must be >= 0");
    threads_.resize(opt_.num_threads);
    for (int t = 0; t < opt_.num_threads; ++t) threads_[t].rng.seed(t);  // TSOneThreadT, tree_search.h:54
  }

  int num_threads() const { return (int)threads_.size(); }
  int seq_of(int thread) const { return threads_[thread].seq; }
  void bump_seq(int thread) { threads_[thread].seq += 1; }

  // MCTSAI_T::Act (elf/mcts.h:40-46): reset_tree (:90-93) unless persistent_tree, in which case the tree
  // is advanced along the moves played since the last search (advance_moves :105-110 -> TreeAdvance,
  // elf/tree_search_node.h:188-200): the subtree under the played move becomes the tree, with its
  // statistics; everything else is freed and its pool slots are recycled.  Then Run.
  //
  // NOTE on parity: in the reference's Go path GoState::_moves is never written (go/go_state.h:62-74,105;
  // GoGame keeps its own list, go/game.cc:92), so moves_since() is always empty, TreeAdvance never runs and
  // SetStateIfNull (elf/tree_search.h:245) keeps the root at the first position the search ever saw: with
  // persistent_tree the reference searches the opening position for ever.  That is not reproduced; this is
  // the behaviour the option is written for (what the reference's MiniRTS states, which do implement
  // moves_since, get).
  void begin(int root_slot, const std::vector<int>& moves_since = std::vector<int>()) {
    bool keep = opt_.persistent_tree && !nodes_.empty();
    if (keep) {
      for (int a : moves_since) tree_advance(a);
      TreeNode& root = nodes_[0];
      if (root.slot >= pool_base_) free_slots_.push_back(root.slot - pool_base_);
      root.slot = root_slot;   // the game's own board holds the root position
      // room for a whole search on top of what is kept, or the tree is dropped
      if ((int)free_slots_.size() < opt_.num_threads * opt_.num_rollout_per_thread) keep = false;
    }
    if (!keep) fresh(root_slot);
    for (auto& th : threads_) {
      th.left = opt_.num_rollout_per_thread;
      th.node = 0;
      th.traj.clear();
      th.waiting = false;
      th.fresh = true;
    }
  }

  // MCTSAI_T::GameEnd -> reset_tree: the game restarted, nothing of the tree applies
  void reset() { nodes_.clear(); }
  // pool slots a context must give every game: one search's worth; three with persistent_tree (a kept
  // subtree may hold up to two searches' worth of positions; a larger one is dropped, see begin())
  static int pool_slots(const TSOptions& o) { return o.num_threads * o.num_rollout_per_thread * (o.persistent_tree ? 3 : 1); }

  // Runs every thread until it parks; leaves that need evaluating are appended to `out`.
  void advance(std::vector<LeafRequest>* out) {
    for (int t = 0; t < (int)threads_.size(); ++t) run_thread(t, out);
  }

  // The reply for thread `thread`'s outstanding leaf: up to k (action, prior) pairs in descending
  // prior (actions < 0 are padding) and the value (NodeT::ExpandIfNecessary,
  // tree_search_node.h:87-111; the edge init of tree_search.h:187-193).
  void expand(int thread, const int32_t* actions, const float* priors, int k, float value) {
    VThread& th = threads_[thread];
    if (!th.waiting) throw std::range_error("TreeSearch::expand: thread has no leaf outstanding");
    const int id = th.node;
    for (int i = 0; i < k; ++i) {
      if (actions[i] < 0) continue;
      const unsigned short c = coord_of_action(actions[i]);
      if (nodes_[id].sa.count(c)) continue;
      EdgeInfo e;
      e.prior = priors[i];
      e.next = (int)nodes_.size();
      nodes_.emplace_back();
      nodes_.back().depth = nodes_[id].depth + 1;
      e.acc_reward = (float)(th.rng() % (unsigned long)(opt_.pseudo_games + 1));
      e.n = opt_.pseudo_games;
      nodes_[id].sa.emplace(c, e);
    }
    nodes_[id].V = value;
    nodes_[id].visited = true;
    nodes_[id].pending = false;
    th.waiting = false;
    finish_rollout(th, id);  // NODE_JUST_VISITED ends the descent, tree_search.h:97
  }

  bool done() const {
    for (const auto& th : threads_)
      if (th.left > 0 || th.waiting) return false;
    return true;
  }

  // TreeSearchT::Run's pick (tree_search.h:262-270); -1 = pass (an empty root: A() = M_PASS)
  int best_action(std::mt19937* uniform_rng) const {
    const auto& sa = nodes_[0].sa;
    unsigned short best = 0;
    float max_score = std::numeric_limits<float>::lowest();
    if (opt_.pick_method == "uniform_random") {
      if (sa.empty()) return -1;
      int idx = (int)((*uniform_rng)() % sa.size());
      auto it = sa.begin();
      while (--idx >= 0) ++it;
      return action_of_coord(it->first);
    }
    const bool prior = opt_.pick_method == "strongest_prior";
    if (!prior && opt_.pick_method != "most_visited")
      throw std::range_error("MCTS Pick method unknown! " + opt_.pick_method);
    for (const auto& kv : sa) {
      const float score = prior ? kv.second.prior : (float)kv.second.n;
      if (score > max_score) { max_score = score; best = kv.first; }
    }
    return action_of_coord(best);
  }

  // (visits, accumulated reward, prior) of the root's edges, for inspection
  std::vector<std::vector<float>> root_edges() const {
    std::vector<std::vector<float>> out;
    for (const auto& kv : nodes_[0].sa)
      out.push_back({(float)action_of_coord(kv.first), (float)kv.second.n, kv.second.acc_reward, kv.second.prior});
    return out;
  }
  // the same for the root's child under `action` (empty: no such child, or not expanded yet)
  std::vector<std::vector<float>> child_edges(int action) const {
    std::vector<std::vector<float>> out;
    auto it = nodes_[0].sa.find(coord_of_action(action));
    if (it == nodes_[0].sa.end()) return out;
    for (const auto& kv : nodes_[it->second.next].sa)
      out.push_back({(float)action_of_coord(kv.first), (float)kv.second.n, kv.second.acc_reward, kv.second.prior});
    return out;
  }
  int root_count() const { return nodes_.empty() ? 0 : nodes_[0].count; }
  size_t num_nodes() const { return nodes_.size(); }
  size_t free_slots() const { return free_slots_.size(); }

 private:
  struct VThread {
    int left = 0;
    int node = 0;
    bool waiting = false;   // a leaf of this thread is out for evaluation
    bool fresh = true;      // the next step starts a new rollout at the root
    int seq = 0;            // the spawned AIComm's SeqInfo::seq (elf/ai_comm.h:58-61,63-68)
    std::vector<std::pair<int, unsigned short>> traj;
    std::mt19937 rng;
  };

  TSOptions opt_;
  int pool_base_, pool_size_;
  std::vector<int> free_slots_;   // pool slots not in use, as offsets from pool_base_; handed out from the back
  std::vector<TreeNode> nodes_;
  std::vector<VThread> threads_;

  void fresh(int root_slot) {
    nodes_.clear();
    nodes_.emplace_back();
    nodes_[0].slot = root_slot;
    free_slots_.resize(pool_size_);
    for (int i = 0; i < pool_size_; ++i) free_slots_[i] = pool_size_ - 1 - i;   // slot 0 first
  }

  // NodeAllocT::TreeAdvance (elf/tree_search_node.h:188-200): the child under `action` becomes the root
  // (a fresh root if there is none); the other subtrees and the old root are freed.
  void tree_advance(int action) {
    const unsigned short c = coord_of_action(action);
    auto it = nodes_[0].sa.find(c);
    if (it == nodes_[0].sa.end()) {
      fresh(-1);   // (begin() gives the new root the game's board)
      return;
    }
    std::vector<TreeNode> kept;
    std::vector<int> order{it->second.next};   // old ids in breadth-first order = new ids
    for (size_t i = 0; i < order.size(); ++i)
      for (auto& kv : nodes_[order[i]].sa) order.push_back(kv.second.next);
    std::vector<int> remap(nodes_.size(), -1);
    for (size_t i = 0; i < order.size(); ++i) remap[order[i]] = (int)i;
    kept.reserve(order.size());
    std::vector<char> used(pool_size_, 0);
    for (int old : order) {
      kept.push_back(std::move(nodes_[old]));
      TreeNode& nd = kept.back();
      nd.depth -= 1;
      for (auto& kv : nd.sa) kv.second.next = remap[kv.second.next];
      if (nd.slot >= pool_base_) used[nd.slot - pool_base_] = 1;
    }
    nodes_.swap(kept);
    free_slots_.clear();
    for (int i = pool_size_ - 1; i >= 0; --i)
      if (!used[i]) free_slots_.push_back(i);
  }

  // UCT, elf/tree_search_alg.h:20-46 (same arithmetic types)
  unsigned short uct(const TreeNode& nd) const {
    unsigned short best_a = 0;
    float max_score = std::numeric_limits<float>::lowest();
    const float c_puct = 0.5;
    const float sqrt_count1 = std::sqrt((float)nd.count + 1);
    for (const auto& kv : nd.sa) {
      const EdgeInfo& info = kv.second;
      float score = (info.acc_reward + 0.5) / (info.n + 1);
      if (opt_.use_prior) score += c_puct * info.prior / (1 + info.n) * sqrt_count1;
      if (score > max_score) { max_score = score; best_a = kv.first; }
    }
    return best_a;
  }

  // backpropagation with the value of the node the rollout stopped at (tree_search.h:122-131)
  void finish_rollout(VThread& th, int stop_node) {
    const float reward = nodes_[stop_node].V;
    for (const auto& step : th.traj) {
      TreeNode& nd = nodes_[step.first];
      auto it = nd.sa.find(step.second);
      if (it == nd.sa.end()) continue;  // AccumulateStats: "Not found, skip"
      nd.count += 1;
      it->second.acc_reward += reward;
      it->second.n += 1;
    }
    th.left -= 1;
    th.fresh = true;
  }

  void run_thread(int t, std::vector<LeafRequest>* out) {
    VThread& th = threads_[t];
    while (!th.waiting && th.left > 0) {
      if (th.fresh) { th.traj.clear(); th.node = 0; th.fresh = false; }
      TreeNode& nd = nodes_[th.node];
      if (!nd.visited) {
        if (nd.pending) return;  // another thread holds the node: wait for its expansion
        nd.pending = true;
        th.waiting = true;
        LeafRequest rq;
        rq.thread = t; rq.node = th.node; rq.depth = nd.depth;
        if (nd.slot < 0) {       // TSOneThreadT::_forward: copy of the parent's state + forward(a)
          if (free_slots_.empty()) throw std::runtime_error("TreeSearch: node pool exhausted");
          nd.slot = pool_base_ + free_slots_.back();
          free_slots_.pop_back();
          rq.parent_slot = nodes_[th.traj.back().first].slot;
          rq.action = action_of_coord(th.traj.back().second);
        }
        rq.slot = nd.slot;
        out->push_back(rq);
        return;
      }
      const unsigned short a = uct(nd);
      th.traj.emplace_back(th.node, a);
      auto it = nd.sa.find(a);
      if (it == nd.sa.end()) { finish_rollout(th, th.node); continue; }  // no edges: Descent fails, break
      th.node = it->second.next;
    }
  }
};

}  // namespace gogame
