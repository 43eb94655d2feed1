// go_game.cc -- the `go_game` Python module of the MI355X build: same classes, methods, batch
// keys, shapes and dtypes as the reference's pybind module (go/python_wrapper.cc:21-34,
// elf/pybind_interface.h:18-95, go/game_context.h:22-112), with the games resident on the GPU.
//
// What is different underneath (on purpose): the reference runs one OS thread per game and one
// per collector and hands samples over with condition variables and per-key memcpy
// (elf/comm_template.h, elf/state_collector.h, elf/hist.h).  Here a GameContext owns ONE batch of
// boards in HBM (libgoenv.so, include/goenv.h); Wait() forms the next batch of parked games,
// lets the device write planes / masks / sequencing fields straight into the registered batch
// tensors (device pointers are used in place, host pointers go through a staging copy), and
// Steps() applies the replies to exactly those games with one launch.  No game threads exist.
//
// Host language: C++ + pybind11, as the reference.  GPU access only through the C-ABI.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <algorithm>
#include <chrono>
#include <ctime>
#include <deque>
#include <iostream>
#include <map>
#include <memory>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "device_env.h"
#include "offline_loader.h"
#include "sgf_record.h"
#include "tree_search.h"
#include "worker_pool.h"

namespace py = pybind11;
using namespace gogame;

// ------------------------------------------------------------------------------------------
// option / descriptor PODs (field names as in the reference)
// ------------------------------------------------------------------------------------------

// elf/python_options_utils_cpp.h:20-55
struct ContextOptions {
  int num_games = 1;
  int max_num_threads = 0;
  int T = 1;
  bool verbose_comm = false;
  bool verbose_collector = false;
  bool wait_per_group = false;
  int num_collectors = 1;
  TSOptions mcts_options;
  void print() const {
    std::cout << "#Game: " << num_games << std::endl;
    std::cout << "#Max_thread: " << max_num_threads << std::endl;
    std::cout << "#Collectors: " << num_collectors << std::endl;
    std::cout << "T: " << T << std::endl;
    if (verbose_comm) std::cout << "Comm Verbose On" << std::endl;
    if (verbose_collector) std::cout << "Comm Collector On" << std::endl;
    std::cout << "Wait per group: " << (wait_per_group ? "True" : "False") << std::endl;
  }
};

// go/go_game_specific.h:16-57, plus the fields this build adds (all defaulted so that code
// written for the reference runs unchanged).
struct GameOptions {
  unsigned int seed = 0;
  int num_planes = GOENV_NUM_PLANES;
  int num_future_actions = 3;
  int num_games_per_thread = 300;
  std::string mode;
  bool use_mcts = false;
  int data_aug = -1;
  float start_ratio_pre_moves = 0.5f;
  float ratio_pre_moves = 0.0f;
  int move_cutoff = -1;
  std::string list_filename;
  bool verbose = false;
  // --- extensions ---
  int gpu = 0;                 // HIP device that holds the boards
  float komi = 7.5f;           // for the win statistics only (go/sgf.h:92)
  int ply_cap = 0;             // > 0: a selfplay game also ends (and restarts) beyond this ply
  bool end_on_two_passes = false;  // selfplay: score + restart after IsGameEnd (the reference
                                   // restarts only on an illegal move, go/game.cc:81-90)
  bool signal_restart = false;     // selfplay: restart -> game_counter++, seq=0, last_terminal=1
                                   // (the reference's Go loop never calls AIComm::Restart there)
  int64_t board_id_base = 0;       // global id of game 0 (rank * num_games in a sharded job): the on-device
                                   // RNG is keyed by the global id, so ranks play different games
  bool preset_zero_planes = false; // the caller's `s` tensors were zero-filled once and stay the caller's: the 11 constant
                                   // planes (12, 13, 16..24) are never written (goenv_set_constant_planes_preset)
  bool fuse_selfplay = true;       // selfplay with device tensors: Steps() = ONE launch that also writes the
                                   // games' next samples (false: Wait() extracts them, as in round 1)
};

// elf/state_collector.h:112-143
struct EntryInfo {
  std::string key;
  std::string type;
  std::vector<int> sz;
  uint64_t p = 0;
  std::size_t byte_size = 0;
};

// elf/comm_template.h:32-44
struct GroupStat {
  int gid = -1;
  int hist_len = 1;
  std::string name;
  std::string info() const {
    return "[gid=" + std::to_string(gid) + "][T=" + std::to_string(hist_len) + "][name=\"" + name + "\"]";
  }
};

// elf/state_collector.h:30-40.  `s` holds the ids of the batch's games (the reference holds
// pointers to their host-side history objects, which do not exist here).
struct Infos {
  int gid = -1;
  std::vector<int> s;
  int batchsize() const { return (int)s.size(); }
};

// ------------------------------------------------------------------------------------------

namespace {

struct KeySpec {
  const char* key;
  const char* type;  // type strings of elf/copier.hh:44-52
  int elem;          // bytes per element
  int inner;         // elements per sample (-1: num_future_actions)
};

// keys of GameState (go/go_game_specific.h:112) as EntryFunc reports them
// (go/game_context.h:67-83) + the extension key `legal`.
const KeySpec kKeys[] = {
    {"s", "float", 4, GOENV_FEAT_FLOATS},      {"offline_a", "int64_t", 8, -1},
    {"id", "int32_t", 4, 1},                   {"seq", "int32_t", 4, 1},
    {"game_counter", "int32_t", 4, 1},         {"last_terminal", "char", 1, 1},
    {"move_idx", "int32_t", 4, 1},             {"winner", "int32_t", 4, 1},
    {"a", "int64_t", 8, 1},                    {"V", "float", 4, 1},
    {"pi", "float", 4, GOENV_NUM_POINTS},      {"aug_code", "int32_t", 4, 1},
    {"game_record_idx", "int32_t", 4, 1},      {"legal", "unsigned char", 1, GOENV_NUM_POINTS},
};

const KeySpec* find_key(const std::string& key) {
  for (const KeySpec& k : kKeys)
    if (key == k.key) return &k;
  return nullptr;
}

struct Slot {
  void* p = nullptr;
  size_t bytes = 0;
  bool device = false;
  int T = 1;            // history entries the tensor holds ([T, B, ...], elf/state_collector.h:125-131)
  size_t row = 0;       // bytes of one sample of this key
};

// CondPerGroupT (elf/comm_template.h:56-92): per game and exclusive group, when the group last took
// a sample of the game.
struct CondState {
  int last_used_seq = 0, last_seq = 0, game_counter = 0;
};

// The host-kept part of one entry of a game's history (HistT<GameState>, elf/hist.h:19-66); planes,
// mask and replies of the entry live in the device rings.
struct HistEntry {
  int32_t meta[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // kMetaKeys order
  std::vector<int64_t> offline_a;
};

struct Collector {
  int gid = 0;
  int batchsize = 0;
  int exclusive_id = 0;
  int timeout_usec = 0;
  GroupStat gstat;
  std::map<std::string, Slot> input, reply;
  bool busy = false;
  std::vector<int> games;      // games parked in the outstanding batch
  std::vector<int> boards;     // their boards (row order)
  bool contiguous = false;     // boards are first, first+1, ...
  bool idx_pinned = false;     // ... or listed in pin_idx (a tree-search leaf batch) instead of idx
  long batches = 0, samples = 0;
  // selfplay with device tensors: Steps() already wrote the NEXT samples of `games` into the input
  // tensors (one fused launch); the next Wait() only hands the batch out
  bool prefilled = false;
  uint64_t prefill_stamp = 0;
  // device staging
  DevBuf idx, aug, actions, status, stage_s, stage_legal, stage_meta[8], stage_pi;
  // tree search: the index arrays of a leaf batch and its top-k / value results, read and written by the kernels in place
  // (every leaf batch ends with a stream synchronise in Steps(), so the collector's next batch may overwrite them)
  PinBuf pin_idx, pin_src, pin_dst, pin_act, pin_topa, pin_topp, pin_v;
  void release_buffers() {
    for (PinBuf* b : {&pin_idx, &pin_src, &pin_dst, &pin_act, &pin_topa, &pin_topp, &pin_v}) b->release();
    idx.release(); aug.release(); actions.release(); status.release(); stage_s.release(); stage_legal.release();
    for (auto& m : stage_meta) m.release();
    stage_pi.release();
  }
};

struct SeqState {  // SeqInfo, elf/common.h:11-19
  int seq = 0, game_counter = 0;
  char last_terminal = 0;
};

struct PendingSample {  // what extract() decided for the game's current sample
  int loader = 0;
  int aug = 0;
  int seq = 0, game_counter = 0;
  char last_terminal = 0;
};

const char* kMetaKeys[8] = {"id", "seq", "game_counter", "last_terminal", "move_idx", "winner", "aug_code",
                            "game_record_idx"};

}  // namespace

class GameContext {
 public:
  GameContext(const ContextOptions& co, const GameOptions& opt) : co_(co), opt_(opt) {
    if (co_.num_games <= 0) throw std::range_error("ContextOptions.num_games must be positive");
    offline_ = !(opt_.mode == "online" || opt_.mode == "selfplay");
    mcts_ = opt_.use_mcts && !offline_;  // GoGame::Init builds an MCTSGoAI only for online / selfplay, go/game.cc:38-41
    if (mcts_) {
      const TSOptions& ts = co_.mcts_options;
      if (ts.num_threads < 1 || ts.num_rollout_per_thread < 1)
        throw std::range_error("mcts_options: num_threads and num_rollout_per_thread must be >= 1");
      if (ts.pick_method != "most_visited" && ts.pick_method != "strongest_prior" && ts.pick_method != "uniform_random")
        throw std::range_error("MCTS Pick method unknown! " + ts.pick_method);
      tick_ = ts.num_threads + 1;
    }
    if (offline_ && (opt_.num_future_actions < 0 || opt_.num_future_actions > GOENV_MAX_FUTURE))
      throw std::range_error("GameOptions.num_future_actions must be in 0.." + std::to_string(GOENV_MAX_FUTURE));
    if (offline_ && !opt_.list_filename.empty()) {
      if (!corpus_.load_list(opt_.list_filename))
        std::cout << "Loading " << opt_.list_filename << " failed!" << std::endl;
      else {
        std::cout << "Loading list_file: " << opt_.list_filename << std::endl;
        std::cout << "Loaded: #Game: " << corpus_.size() << std::endl;
      }
    }
  }
  ~GameContext() { Stop(); }

  // --- collectors (CommT::AddCollectors, elf/comm_template.h:170-181) ------------------------
  GroupStat CreateGroupStat() const { return GroupStat(); }

  int AddCollectors(int batchsize, int exclusive_id, int timeout_usec, const GroupStat& gstat) {
    if (started_) throw std::range_error("AddCollectors after Start()");
    if (batchsize <= 0) throw std::range_error("AddCollectors: batchsize must be positive");
    if (gstat.hist_len < 1 || gstat.hist_len > std::max(1, co_.T))
      throw std::range_error("AddCollectors: hist_len " + std::to_string(gstat.hist_len) + " outside 1..ContextOptions.T = " +
                             std::to_string(std::max(1, co_.T)) + " (a game keeps T entries: the group would never get a sample)");
    if (exclusive_id < 0 || exclusive_id > (int)groups_.size())
      throw std::range_error("AddCollectors: exclusive_id out of order");
    std::unique_ptr<Collector> c(new Collector());
    c->gid = (int)collectors_.size();
    c->batchsize = batchsize;
    c->exclusive_id = exclusive_id;
    c->timeout_usec = timeout_usec;
    c->gstat = gstat;
    c->gstat.gid = c->gid;
    if (exclusive_id == (int)groups_.size()) groups_.emplace_back();
    groups_[exclusive_id].members.push_back(c->gid);
    if (groups_[exclusive_id].members.size() == 1) groups_[exclusive_id].name = gstat.name;
    collectors_.push_back(std::move(c));
    return (int)collectors_.size() - 1;
  }

  std::string GetCollectorInfos() const {
    std::stringstream ss;
    for (size_t i = 0; i < groups_.size(); ++i) {
      ss << "Group " << i << ": " << std::endl;
      for (int gid : groups_[i].members) {
        const Collector& c = *collectors_[gid];
        ss << "  Collector[" << gid << "] Batchsize: " << c.batchsize << " Info: " << c.gstat.info() << std::endl;
      }
    }
    return ss.str();
  }

  // CollectorGroupT::GetEntry (elf/state_collector.h:200-206) + EntryFunc (go/game_context.h:67-83)
  EntryInfo GetTensorSpec(int gid, const std::string& key, int T) {
    const Collector& c = collector(gid);
    EntryInfo e;
    if (key.empty()) return e;
    const KeySpec* k = find_key(key);
    if (k) {
      e.key = key;
      e.type = k->type;
    }
    e.sz.push_back(T);
    e.sz.push_back(c.batchsize);
    if (k) {
      if (key == "s") { e.sz.push_back(GOENV_NUM_PLANES); e.sz.push_back(19); e.sz.push_back(19); }
      else if (k->inner == -1) e.sz.push_back(opt_.num_future_actions);
      else if (k->inner > 1) e.sz.push_back(k->inner);
    }
    return e;
  }

  // CollectorGroupT::AddEntry (elf/state_collector.h:208-218)
  void AddTensor(int gid, const std::string& input_reply, const EntryInfo& e) {
    Collector& c = collector(gid);
    std::map<std::string, Slot>* dst = nullptr;
    if (input_reply == "input") dst = &c.input;
    else if (input_reply == "reply") dst = &c.reply;
    else throw std::range_error("Unknown input_reply " + input_reply);
    const KeySpec* k = find_key(e.key);
    if (!k) throw std::range_error("AddTensor: unknown key \"" + e.key + "\"");
    const size_t inner = k->inner == -1 ? (size_t)opt_.num_future_actions : (size_t)k->inner;
    const size_t row = inner * (size_t)k->elem;
    const size_t want = (size_t)c.batchsize * row;  // per history entry
    const size_t T = want ? e.byte_size / want : 0;
    if (e.p == 0 || T < 1 || T * want != e.byte_size || (int)T > std::max(1, co_.T))
      throw std::range_error("AddTensor: \"" + e.key + "\" needs a buffer of T x " + std::to_string(want) +
                             " bytes with T in 1..ContextOptions.T, got " + std::to_string(e.byte_size));
    // a group is offered a sample once the game's history holds hist_len entries (CondPerGroupT::Check,
    // elf/comm_template.h:88): a tensor with more entries than that would be padded (elf/hist.h:90-98) with
    // rows this build does not reproduce -- ask for hist_len >= T instead
    if ((int)T > std::max(1, c.gstat.hist_len))
      throw std::range_error("AddTensor: \"" + e.key + "\" holds T = " + std::to_string(T) + " entries but the group's hist_len is " +
                             std::to_string(c.gstat.hist_len) + "; set GroupStat.hist_len >= T");
    Slot s;
    s.p = reinterpret_cast<void*>(e.p);
    s.bytes = e.byte_size;
    s.device = goenv_pointer_is_device(s.p) != 0;
    s.T = (int)T;
    s.row = row;
    (*dst)[e.key] = s;
  }

  std::map<std::string, int> GetParams() const {  // go/game_context.h:56-65
    return {{"num_action", GOENV_NUM_POINTS},  {"board_size", GOENV_BOARD_SIZE},
            {"num_planes", GOENV_NUM_PLANES},  {"num_future_actions", opt_.num_future_actions},
            {"our_stone_plane", 7},            {"opponent_stone_plane", 8}};
  }

  int size() const { return co_.num_games; }
  std::string Version() const { return "elf_amd.go_game/goenv-abi-" + std::to_string(goenv_version()); }
  void SetStream(uint64_t stream) { stream_ = reinterpret_cast<void*>(stream); if (env_) env_->set_stream(stream_); }

  // --- life cycle -------------------------------------------------------------------------------
  void Start() {
    if (started_) return;
    if (collectors_.empty()) throw std::range_error("Start(): no collectors were added");
    const int npt = offline_ ? std::max(1, opt_.num_games_per_thread) : 1;
    if (offline_ && corpus_.size() == 0) throw std::range_error("Start(): offline mode needs a readable list_filename");
    // tree search: every game owns a pool of node positions behind the games' own boards
    const int pool = mcts_ ? TreeSearch::pool_slots(co_.mcts_options) : 0;
    const long long n_boards = (long long)co_.num_games * (mcts_ ? 1 + pool : npt);
    if (n_boards > (1ll << 30)) throw std::range_error("Start(): num_games x node pool exceeds 2^30 boards");
    env_.reset(new DeviceEnv(opt_.gpu, (int)n_boards, opt_.seed ? opt_.seed : 1, opt_.board_id_base, opt_.komi,
                             opt_.ply_cap, offline_ ? 1 : 0));
    env_->set_stream(stream_);
    if (opt_.preset_zero_planes) {
      // device tensors only: the planes of a host tensor travel through staging buffers of this context,
      // whose constant planes nobody would have cleared
      for (const auto& c : collectors_)
        for (const auto& kv : c->input)
          if (kv.first == "s" && !kv.second.device)
            throw std::range_error("GameOptions.preset_zero_planes needs the \"s\" tensors in device memory");
      if (co_.T > 1) throw std::range_error("GameOptions.preset_zero_planes is not supported with history length T > 1");
      chk(goenv_set_constant_planes_preset(env_->ctx(), 1), "goenv_set_constant_planes_preset");
    }
    games_.clear();
    games_.resize(co_.num_games);
    searches_.clear();
    finished_.clear();
    pick_rng_.seed(opt_.seed ? opt_.seed : 1);
    if (mcts_) {
      if (group_of("actor") < 0) throw std::range_error("Start(): use_mcts needs a collector that takes \"actor\" samples");
      for (int g = 0; g < co_.num_games; ++g) {
        searches_.emplace_back(new TreeSearch(co_.mcts_options, co_.num_games + g * pool, pool));
        games_[g].requests.resize(co_.mcts_options.num_threads);
      }
    }
    for (auto& g : groups_) { g.ready.clear(); g.stamp.clear(); }
    setup_general();
    const uint64_t base_seed = opt_.seed ? (uint64_t)opt_.seed : clock_seed();
    for (int g = 0; g < co_.num_games; ++g) {
      GameSlot& gs = games_[g];
      // GoGame::GoGame, go/game.cc:20-34 (seed 0: clock-derived per game)
      const uint64_t seed = opt_.seed ? base_seed : (base_seed + (uint64_t)g * 2341479ull) % 100000000ull;
      gs.rng.seed((std::mt19937::result_type)seed);
      if (offline_) {
        OfflineOptions oo;
        oo.num_future_actions = opt_.num_future_actions; oo.data_aug = opt_.data_aug;
        oo.start_ratio_pre_moves = opt_.start_ratio_pre_moves; oo.ratio_pre_moves = opt_.ratio_pre_moves;
        oo.move_cutoff = opt_.move_cutoff; oo.verbose = opt_.verbose;
        for (int i = 0; i < npt; ++i) {
          // go/game.cc:51: seed + game_idx * i * 997 + i * 13773 + 7, narrowed to int
          const uint64_t ls = seed + (uint64_t)(int64_t)(g * i * 997 + i * 13773 + 7);
          gs.loaders.emplace_back(new OfflineLoader(oo, &corpus_, env_.get(), g * npt + i, (uint32_t)ls));
        }
      }
      gs.phase = (opt_.mode == "online" && group_of("human_actor") >= 0) ? 0 : 1;
      if (general_) {
        gs.cond.assign(groups_.size(), CondState());
        gs.hist.assign(hist_T_, HistEntry());
      } else {
        produce_sample(g);
      }
    }
    if (offline_) offline_flush_all();   // the first load's pre-moves of every replay, a few large launches
    started_ = true;
    if (general_) {
      std::vector<int> all(co_.num_games);
      for (int g = 0; g < co_.num_games; ++g) all[g] = g;
      produce_general(all);
    }
    finalize_searches();
  }

  void Stop() {
    if (!started_) return;
    if (env_) {
      try {
        env_->sync();
        if (offline_) {
          int32_t bad = 0;
          if (goenv_offline_bad_moves(env_->ctx(), &bad, env_->stream()) == 0 && bad > 0)
            std::cout << "[go_game] WARNING: the device rejected " << bad << " replayed record move(s) that the host took for legal" << std::endl;
        }
      } catch (...) {}
    }
    for (auto& c : collectors_) { c->busy = false; c->prefilled = false; c->games.clear(); c->release_buffers(); }
    for (DevBuf* b : {&ts_src_, &ts_dst_, &ts_act_, &ts_status_, &ts_topa_, &ts_topp_, &ts_v_, &g_rows_, &g_stage_s_,
                      &g_stage_l_, &g_stage_, &g_idx_, &g_act_, &g_pi_, &g_status_})
      b->release();
    for (auto& kv : ring_) kv.second.release();
    ring_.clear();
    games_.clear();
    searches_.clear();
    finished_.clear();
    env_.reset();
    started_ = false;
  }

  // --- the batch loop (ContextT::Wait / Steps, elf/comm_template.h:363-365) ---------------------
  Infos Wait(int timeout_usec) {
    (void)timeout_usec;
    if (co_.wait_per_group)
      throw std::range_error("wait_batch error. group_id = -1 while #queue_per_group = " +
                             std::to_string(collectors_.size()));
    HostTimer t(&host_wait_us_, &host_waits_);
    return wait_impl(-1);
  }

  Infos WaitGroup(int group_id, int timeout_usec) {
    (void)timeout_usec;
    if (!co_.wait_per_group || group_id < 0 || group_id >= (int)collectors_.size())
      throw std::range_error("wait_batch error. group_id = " + std::to_string(group_id) +
                             " while #queue_per_group = " + std::to_string(co_.wait_per_group ? collectors_.size() : 0));
    HostTimer t(&host_wait_us_, &host_waits_);
    return wait_impl(group_id);
  }

  void Steps(const Infos& infos) {
    if (infos.gid < 0) return;  // timed-out Wait: nothing to release (comm_template.h:270)
    HostTimer t(&host_steps_us_, &host_steps_);
    Collector& c = collector(infos.gid);
    if (!c.busy) throw std::range_error("Steps(): collector " + std::to_string(infos.gid) + " has no batch outstanding");
    if (general_) {
      steps_general(c);
      return;
    }
    bool fused = false;
    if (offline_) steps_offline(c);
    else if (leaf_batch(c)) steps_mcts(c);
    else fused = steps_play(c);
    c.busy = false;
    if (fused) {  // the games stay with the collector: their next samples are in its tensors already
      c.prefilled = true;
      c.prefill_stamp = ++clock_;
      return;
    }
    if (!leaf_batch(c))
      for (int ticket : c.games) produce_sample(ticket / tick_);
    c.games.clear();
    c.boards.clear();
    finalize_searches();
  }

  void PrintSummary() const {
    for (const auto& c : collectors_)
      std::cout << "Collector[" << c->gid << "]: #batches " << c->batches << ", #samples " << c->samples << std::endl;
  }

  // --- extensions -------------------------------------------------------------------------------
  // End-of-episode statistics vector (include/goenv.h GOENV_ST_*): finished-and-uncounted boards
  // are scored first.  This is the buffer a multi-GPU job all-reduces.
  std::vector<int64_t> Stats() {
    need_env();
    // (tree search: the node positions behind the games are not games; only restarts count)
    if (!mcts_) chk(goenv_finish(env_->ctx(), env_->stream()), "goenv_finish");
    std::vector<int64_t> out(GOENV_NUM_STATS);
    chk(goenv_stats_host(env_->ctx(), out.data(), env_->stream()), "goenv_stats_host");
    return out;
  }
  // tree search: what the last finished search of `game_idx` chose, and the search in progress
  py::dict SearchInfo(int game_idx) {
    if (!mcts_) throw std::range_error("SearchInfo: the context does not run a tree search (use_mcts)");
    if (game_idx < 0 || game_idx >= co_.num_games || !started_) throw std::range_error("SearchInfo: no such game");
    py::dict d;
    d["last_move"] = games_[game_idx].last_move;
    d["moves_played"] = games_[game_idx].moves_played;
    d["ply"] = games_[game_idx].ply;
    d["nodes"] = searches_[game_idx]->num_nodes();
    d["root_edges"] = searches_[game_idx]->root_edges();
    return d;
  }

  // host time spent inside Wait / Steps since Start (or the last reset): what the GPU is NOT doing
  py::dict HostTimes(bool reset) {
    py::dict d;
    d["wait_us"] = host_wait_us_; d["waits"] = host_waits_;
    d["steps_us"] = host_steps_us_; d["steps"] = host_steps_;
    if (reset) { host_wait_us_ = host_steps_us_ = 0; host_waits_ = host_steps_ = 0; }
    return d;
  }

  int FuseMax() { need_env(); return goenv_get_fuse_max(env_->ctx()); }
  // offline mode: rows whose owed moves the device rejected although the record's legal prefix said otherwise
  // (0 unless the host's bookkeeping and the boards have diverged); synchronises the stream
  int OfflineBadMoves() {
    need_env();
    int32_t n = 0;
    chk(goenv_offline_bad_moves(env_->ctx(), &n, env_->stream()), "goenv_offline_bad_moves");
    return n;
  }

  uint64_t StatsPtr() { need_env(); return reinterpret_cast<uint64_t>(goenv_stats_ptr(env_->ctx())); }

  std::string ShowBoard(int game_idx) {  // go/game_context.h:89-92
    if (game_idx < 0 || game_idx >= co_.num_games) return "Invalid game_idx [" + std::to_string(game_idx) + "]";
    need_env();
    const int board = offline_ ? games_[game_idx].loaders[games_[game_idx].sample.loader]->board() : game_idx;
    if (offline_) {   // what the replay's board still owes (the last next()), so that the print is the host's position
      std::vector<OfflineLoader*> one{games_[game_idx].loaders[games_[game_idx].sample.loader].get()};
      offline_launch(one, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    }
    DevBuf col, sc;
    col.ensure(env_->ctx(), GOENV_NUM_POINTS);
    sc.ensure(env_->ctx(), 64);
    chk(goenv_get_state_sel(env_->ctx(), nullptr, board, 1, col.as<uint8_t>(), nullptr, nullptr, nullptr,
                            sc.as<int32_t>(), env_->stream()), "goenv_get_state_sel");
    uint8_t color[GOENV_NUM_POINTS];
    int32_t scal[16];
    env_->d2h(color, col.get(), sizeof(color));
    env_->d2h(scal, sc.get(), sizeof(scal));
    env_->sync();
    std::stringstream ss;
    static const char* cols = "ABCDEFGHJKLMNOPQRST";
    ss << "   ";
    for (int x = 0; x < 19; ++x) ss << cols[x] << ' ';
    ss << "\n";
    for (int y = 18; y >= 0; --y) {
      ss << (y + 1 < 10 ? " " : "") << y + 1 << ' ';
      for (int x = 0; x < 19; ++x) {
        const int p = x * 19 + y;
        const char ch = color[p] == 1 ? 'X' : (color[p] == 2 ? 'O' : '.');
        ss << ch << (scal[GOENV_S_LAST1] == p ? ')' : ' ');
      }
      ss << "\n";
    }
    ss << "ply " << scal[GOENV_S_PLY] << ", next " << (scal[GOENV_S_NEXT] == 1 ? "B" : "W") << ", captured by B/W "
       << scal[GOENV_S_BCAP] << "/" << scal[GOENV_S_WCAP] << "\n";
    return ss.str();
  }

 private:
  struct GameSlot {
    std::mt19937 rng;
    std::vector<std::unique_ptr<OfflineLoader>> loaders;
    SeqState seq;
    PendingSample sample;
    int phase = 1;      // online mode: 0 = human to move, 1 = actor to move
    int ply = 1;        // host mirror of _ply for play modes that keep host sequencing
    std::vector<LeafRequest> requests;  // tree search: the outstanding leaf of every search thread
    int last_move = -3;                 // tree search: the move the last finished search played
    std::vector<int> moves_since;       // tree search: moves accepted on the game's board since the last search began
    long moves_played = 0;
    // general path (histories, several groups per sample)
    std::vector<HistEntry> hist;        // ring of ContextOptions.T entries
    int hist_head = -1, hist_size = 0;
    std::vector<CondState> cond;        // per exclusive group
    int pending = 0;                    // groups that still hold the game's newest sample
  };
  struct Group {
    std::string name;
    std::vector<int> members;
    std::deque<int> ready;
    std::deque<uint64_t> stamp;   // when each ready ticket was queued (order among prefilled batches)
    size_t rr = 0;
  };

  struct HostTimer {
    double* acc; long* cnt;
    std::chrono::steady_clock::time_point t0;
    HostTimer(double* a, long* c) : acc(a), cnt(c), t0(std::chrono::steady_clock::now()) {}
    ~HostTimer() { *acc += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); ++*cnt; }
  };
  double host_wait_us_ = 0, host_steps_us_ = 0;
  long host_waits_ = 0, host_steps_ = 0;

  ContextOptions co_;
  GameOptions opt_;
  bool offline_ = false;
  bool started_ = false;
  void* stream_ = nullptr;
  SgfCorpus corpus_;
  std::unique_ptr<DeviceEnv> env_;
  std::vector<std::unique_ptr<Collector>> collectors_;
  std::vector<Group> groups_;
  std::vector<GameSlot> games_;
  size_t rr_group_ = 0;
  uint64_t clock_ = 0;          // orders queued games and prefilled batches: the oldest is served first
  // tree search (use_mcts): a batch row is a ticket = game * tick_ + (0: the game's own sample,
  // 1 + t: the leaf of search thread t); without tree search tick_ = 1 and tickets are game ids
  bool mcts_ = false;
  int tick_ = 1;
  std::vector<std::unique_ptr<TreeSearch>> searches_;
  std::vector<int> finished_;   // games whose search has spent its rollouts
  std::mt19937 pick_rng_;
  DevBuf ts_src_, ts_dst_, ts_act_, ts_status_, ts_topa_, ts_topp_, ts_v_;
  WorkerPool pool_;             // host threads for per-game work between two device batches
  // general path: a game keeps a history of hist_T_ samples (planes, mask, replies in device rings of
  // [num_games x hist_T_] rows, labels on the host) and a sample may go to several exclusive groups
  bool general_ = false;
  bool reply_a_ = false, reply_pi_ = false;   // some collector REPLIES the key (an input-only `a` is history, not a move)
  int hist_T_ = 1;
  std::map<std::string, DevBuf> ring_;   // key -> ring (s, legal, pi, a, V)
  DevBuf g_rows_, g_stage_s_, g_stage_l_, g_stage_, g_idx_, g_act_, g_pi_, g_status_;
  std::vector<OfflineLoader*> off_lds_;   // scratch of fill_inputs_offline

  static uint64_t clock_seed() {
    using namespace std::chrono;
    const long ms = (long)duration_cast<milliseconds>(system_clock::now().time_since_epoch()).count();
    return (uint64_t)((long)time(nullptr) * 1000 + ms);
  }

  void need_env() const {
    if (!env_) throw std::range_error("the context is not started");
  }

  Collector& collector(int gid) {
    if (gid < 0 || gid >= (int)collectors_.size()) throw std::range_error("no collector with gid " + std::to_string(gid));
    return *collectors_[gid];
  }

  // exclusive group that takes samples named `name` (CondPerGroupT::check_name,
  // elf/comm_template.h:46-54: an unnamed group takes everything)
  int group_of(const std::string& name) const {
    for (size_t i = 0; i < groups_.size(); ++i)
      if (groups_[i].name.empty() || groups_[i].name == name) return (int)i;
    return -1;
  }

  std::string sample_name(const GameSlot& gs) const {
    if (offline_) return std::string();  // OfflineLoader::extract leaves GameState::name empty
    return gs.phase == 0 ? "human_actor" : "actor";  // go/game.cc:46,60
  }

  // The game is ready to be sampled again: AI::Act's before_act + Prepare + the host part of
  // extract (elf/ai.h:69-80,157-168); it is queued for the exclusive group that takes its name.
  void produce_sample(int g) {
    GameSlot& gs = games_[g];
    if (offline_) {
      gs.sample.loader = (int)(gs.rng() % gs.loaders.size());  // go/game.cc:96
      OfflineLoader& ld = *gs.loaders[gs.sample.loader];
      if (ld.just_started()) {  // before_act -> AIComm::Restart, go/offpolicy_loader.h:49-51
        gs.seq.seq = 0; gs.seq.game_counter += 1; gs.seq.last_terminal = 1;
      }
      gs.sample.aug = 0;  // drawn in wait_impl, at extract time, after Prepare
      ld.prefetch();      // 76,800 replays do not fit the caches: ask for this one's lines now, use them in Wait
    }
    if (mcts_ && gs.phase == 1) {  // MCTSAI_T::Act: a fresh tree rooted at the game's position
      searches_[g]->begin(g, gs.moves_since);   // (persistent_tree: advanced along the moves since the last search)
      gs.moves_since.clear();
      pump(g);
      return;
    }
    const int e = group_of(sample_name(gs));
    if (e < 0) throw std::range_error("no collector takes samples named \"" + sample_name(gs) + "\"");
    enqueue(e, g * tick_);
  }

  void enqueue(int e, int ticket) {
    groups_[e].ready.push_back(ticket);
    groups_[e].stamp.push_back(++clock_);
  }

  // lets the search threads of game g run until each waits for a leaf; queues those leaves
  void pump(int g) {
    std::vector<LeafRequest> rq;
    searches_[g]->advance(&rq);
    queue_leaves(g, rq);
  }

  void queue_leaves(int g, const std::vector<LeafRequest>& rq) {
    const int e = group_of("actor");
    for (const LeafRequest& r : rq) {
      games_[g].requests[r.thread] = r;
      enqueue(e, g * tick_ + 1 + r.thread);
    }
    if (searches_[g]->done()) finished_.push_back(g);
  }

  bool leaf_batch(const Collector& c) const {
    if (!mcts_ || c.games.empty()) return false;
    int leaves = 0;
    for (int ticket : c.games) leaves += ticket % tick_ != 0;
    if (leaves != 0 && leaves != (int)c.games.size())
      throw std::range_error("a batch mixes search leaves and game samples: give \"actor\" and \"human_actor\" their own collectors");
    return leaves != 0;
  }

  // MCTSAI_T::Act's tail for every game whose rollouts are spent: pick the move, play it on the
  // game's board (GoGame::Act, go/game.cc:80-93), start the next sample
  void finalize_searches() {
    while (!finished_.empty()) {
      std::vector<int> games;
      games.swap(finished_);
      const int m = (int)games.size();
      std::vector<int32_t> boards(games.begin(), games.end());
      std::vector<int64_t> acts(m);
      for (int i = 0; i < m; ++i) acts[i] = searches_[games[i]]->best_action(&pick_rng_);
      ts_dst_.ensure(env_->ctx(), (size_t)m * 4);
      ts_act_.ensure(env_->ctx(), (size_t)m * 8);
      ts_status_.ensure(env_->ctx(), (size_t)m * 4);
      env_->h2d(ts_dst_.get(), boards.data(), (size_t)m * 4);
      env_->h2d(ts_act_.get(), acts.data(), (size_t)m * 8);
      uint32_t flags = GOENV_F_RESET_ILLEGAL;
      if (opt_.end_on_two_passes || opt_.ply_cap > 0) flags |= GOENV_F_RESTART_AFTER_END;
      chk(goenv_ply_sel(env_->ctx(), ts_dst_.as<int32_t>(), 0, m, flags, ts_act_.as<int64_t>(), nullptr, nullptr,
                        nullptr, nullptr, nullptr, ts_status_.as<int32_t>(), env_->stream()), "goenv_ply_sel");
      std::vector<int32_t> st(m);
      env_->d2h(st.data(), ts_status_.get(), (size_t)m * 4);
      env_->sync();
      const bool has_human = opt_.mode == "online" && group_of("human_actor") >= 0 &&
                             groups_[group_of("human_actor")].name == "human_actor";
      for (int i = 0; i < m; ++i) {
        GameSlot& gs = games_[games[i]];
        gs.last_move = (int)acts[i];
        gs.moves_played += 1;
        if (st[i] == 1) {
          gs.ply += 1;
          gs.moves_since.push_back((int)acts[i]);
        } else {          // illegal / ended: the board was reset, nothing of the tree applies (MCTSAI_T::GameEnd)
          gs.ply = 1;
          gs.moves_since.clear();
          searches_[games[i]]->reset();
        }
        if (has_human) gs.phase = 0;
        produce_sample(games[i]);
      }
    }
  }

  Infos wait_impl(int only_gid) {
    if (!started_) throw std::range_error("Wait() before Start()");
    // the oldest batch that Steps() has already filled (fused selfplay step), if any
    Collector* pre = nullptr;
    for (auto& cp : collectors_) {
      Collector& c = *cp;
      if (!c.prefilled || (only_gid >= 0 && c.gid != only_gid)) continue;
      if (!pre || c.prefill_stamp < pre->prefill_stamp) pre = &c;
    }
    // a free collector whose group has parked games; full batches first, a short batch only when
    // nothing else can progress
    Collector* pick = nullptr;
    for (int pass = 0; pass < 2 && !pick; ++pass) {
      if (pass == 1 && pre) break;
      for (size_t k = 0; k < groups_.size() && !pick; ++k) {
        Group& grp = groups_[(rr_group_ + k) % groups_.size()];
        if (grp.ready.empty()) continue;
        for (size_t j = 0; j < grp.members.size(); ++j) {
          Collector& c = *collectors_[grp.members[(grp.rr + j) % grp.members.size()]];
          if (c.busy || c.prefilled || (only_gid >= 0 && c.gid != only_gid)) continue;
          if (pass == 0 && (int)grp.ready.size() < c.batchsize) continue;
          pick = &c;
          grp.rr = (grp.rr + j + 1) % grp.members.size();
          break;
        }
      }
    }
    Infos infos;
    if (pre && (!pick || pre->prefill_stamp < groups_[pick->exclusive_id].stamp.front())) {
      Collector& c = *pre;
      c.prefilled = false;
      c.busy = true;
      c.batches += 1;
      c.samples += (long)c.games.size();
      infos.gid = c.gid;
      infos.s = c.games;
      return infos;
    }
    if (!pick) return infos;  // gid = -1, as a timed-out wait (elf/state_collector.h:91)
    rr_group_ = (rr_group_ + 1) % groups_.size();
    Collector& c = *pick;
    Group& grp = groups_[c.exclusive_id];
    const int n = std::min<int>(c.batchsize, (int)grp.ready.size());
    c.games.assign(grp.ready.begin(), grp.ready.begin() + n);
    grp.ready.erase(grp.ready.begin(), grp.ready.begin() + n);
    grp.stamp.erase(grp.stamp.begin(), grp.stamp.begin() + n);
    if (general_) fill_inputs_general(c);
    else fill_inputs(c);
    c.busy = true;
    c.batches += 1;
    c.samples += n;
    infos.gid = c.gid;
    infos.s = c.games;
    return infos;
  }

  // device pointer to write `key` rows into: the registered tensor itself, or staging
  template <typename T>
  T* out_ptr(Collector& c, const std::string& key, DevBuf& stage, size_t bytes, bool* staged) {
    auto it = c.input.find(key);
    if (it == c.input.end()) return nullptr;
    if (it->second.device) return reinterpret_cast<T*>(it->second.p);
    stage.ensure(env_->ctx(), bytes);
    *staged = true;
    return stage.as<T>();
  }

  void fill_inputs(Collector& c) {
    if (offline_) {
      fill_inputs_offline(c);
      return;
    }
    const int n = (int)c.games.size();
    c.boards.resize(n);
    std::vector<int32_t> aug(n, 0);
    std::vector<int32_t> meta[8];
    const bool host_meta = opt_.mode == "online" || mcts_;
    if (host_meta)
      for (auto& m : meta) m.assign(n, 0);
    std::vector<int32_t> mk_src, mk_dst;   // tree search: leaf positions still to be made
    std::vector<int64_t> mk_act;
    for (int r = 0; r < n; ++r) {
      const int g = c.games[r] / tick_, leaf = c.games[r] % tick_;
      GameSlot& gs = games_[g];
      if (leaf) {
        // MCTSActor::evaluate -> DirectPredictAI::Act on the leaf's state (go/mcts.h:29-34,
        // go/go_ai.h:79-87) through the search thread's spawned AIComm (elf/mcts.h:172-178)
        const LeafRequest& rq = gs.requests[leaf - 1];
        c.boards[r] = rq.slot;
        if (rq.parent_slot >= 0) { mk_src.push_back(rq.parent_slot); mk_dst.push_back(rq.slot); mk_act.push_back(rq.action); }
        TreeSearch& ts = *searches_[g];
        meta[0][r] = g; meta[1][r] = ts.seq_of(leaf - 1); meta[2][r] = 0; meta[3][r] = 0;
        ts.bump_seq(leaf - 1);
        meta[4][r] = gs.ply + rq.depth; meta[5][r] = 0; meta[6][r] = 0; meta[7][r] = -1;
      } else {
        c.boards[r] = g;
        if (host_meta) {  // online: DirectPredictAI / HumanPlayer extract, go/go_ai.h:79-87,106-113
          meta[0][r] = g; meta[1][r] = gs.seq.seq; meta[2][r] = gs.seq.game_counter;
          meta[3][r] = gs.seq.last_terminal;
          gs.seq.seq += 1; gs.seq.last_terminal = 0;
          meta[4][r] = gs.ply; meta[5][r] = 0; meta[6][r] = 0; meta[7][r] = -1;
        }
      }
    }
    if (!mk_src.empty()) {  // TSOneThreadT::_forward for every new leaf: copy of the parent, forward(a)
      const size_t m = mk_src.size();
      // the three arrays in pinned blocks of the collector, read by the kernels in place: no copy call (a pageable
      // hipMemcpyAsync of a few KB costs the host 10 us and more, and a leaf batch had seven of them)
      c.pin_src.ensure(m * 4); c.pin_dst.ensure(m * 4); c.pin_act.ensure(m * 8);
      std::memcpy(c.pin_src.as<int32_t>(), mk_src.data(), m * 4);
      std::memcpy(c.pin_dst.as<int32_t>(), mk_dst.data(), m * 4);
      std::memcpy(c.pin_act.as<int64_t>(), mk_act.data(), m * 8);
      chk(goenv_copy_boards(env_->ctx(), c.pin_src.as<int32_t>(), c.pin_dst.as<int32_t>(), (int)m, env_->stream()),
          "goenv_copy_boards");
      chk(goenv_ply_sel(env_->ctx(), c.pin_dst.as<int32_t>(), 0, (int)m, 0u, c.pin_act.as<int64_t>(), nullptr, nullptr,
                        nullptr, nullptr, nullptr, nullptr, env_->stream()), "goenv_ply_sel");
    }
    c.contiguous = true;
    for (int r = 1; r < n; ++r)
      if (c.boards[r] != c.boards[0] + r) { c.contiguous = false; break; }
    const int32_t* d_idx = nullptr;
    c.idx_pinned = false;
    if (!c.contiguous) {
      if (mcts_ && leaf_batch(c)) {   // (ends with a synchronise in steps_mcts: the block is free again by the next batch)
        c.pin_idx.ensure((size_t)n * 4);
        std::memcpy(c.pin_idx.as<int32_t>(), c.boards.data(), (size_t)n * 4);
        d_idx = c.pin_idx.as<int32_t>();
        c.idx_pinned = true;
      } else {
        c.idx.ensure(env_->ctx(), (size_t)n * 4);
        env_->h2d(c.idx.get(), c.boards.data(), (size_t)n * 4);
        d_idx = c.idx.as<int32_t>();
      }
    }
    const int32_t* d_aug = nullptr;
    bool st_s = false, st_l = false, st_m[8] = {false, false, false, false, false, false, false, false};
    float* d_s = out_ptr<float>(c, "s", c.stage_s, (size_t)n * GOENV_FEAT_FLOATS * 4, &st_s);
    uint8_t* d_l = out_ptr<uint8_t>(c, "legal", c.stage_legal, (size_t)n * GOENV_NUM_POINTS, &st_l);
    goenv_sample_out so;
    std::memset(&so, 0, sizeof(so));
    so.mode = opt_.signal_restart ? GOENV_SEQ_EPISODES : GOENV_SEQ_RUN_ON;
    bool want_dev_meta = false;
    if (!host_meta) {  // selfplay: the device keeps seq / game_counter per board
      void** fields[8] = {(void**)&so.id, (void**)&so.seq, (void**)&so.game_counter, (void**)&so.last_terminal,
                          (void**)&so.move_idx, (void**)&so.winner, (void**)&so.aug_code, (void**)&so.game_record_idx};
      for (int k = 0; k < 8; ++k) {
        const size_t eb = k == 3 ? 1 : 4;
        *fields[k] = out_ptr<char>(c, kMetaKeys[k], c.stage_meta[k], (size_t)n * eb, &st_m[k]);
        want_dev_meta |= *fields[k] != nullptr;
      }
    }
    // the device always advances seq in selfplay mode, registered or not
    const goenv_sample_out* meta_arg = host_meta ? nullptr : &so;
    (void)want_dev_meta;
    chk(goenv_collect_sel(env_->ctx(), d_idx, c.contiguous ? c.boards[0] : 0, n, d_aug, d_s, d_l, meta_arg,
                          env_->stream()), "goenv_collect_sel");
    bool need_sync = false;
    auto flush = [&](const std::string& key, DevBuf& stage, bool staged, size_t bytes) {
      if (!staged) return;
      env_->d2h(c.input[key].p, stage.get(), bytes);
      need_sync = true;
    };
    flush("s", c.stage_s, st_s, (size_t)n * GOENV_FEAT_FLOATS * 4);
    flush("legal", c.stage_legal, st_l, (size_t)n * GOENV_NUM_POINTS);
    for (int k = 0; k < 8; ++k) flush(kMetaKeys[k], c.stage_meta[k], st_m[k], (size_t)n * (k == 3 ? 1 : 4));
    if (host_meta) {
      for (int k = 0; k < 8; ++k) {
        auto it = c.input.find(kMetaKeys[k]);
        if (it == c.input.end()) continue;
        if (k == 3) {
          std::vector<int8_t> lt(n);
          for (int r = 0; r < n; ++r) lt[r] = (int8_t)meta[3][r];
          put(it->second, lt.data(), (size_t)n);
        } else {
          put(it->second, meta[k].data(), (size_t)n * 4);
        }
      }
    }
    // host-resident tensors must be complete when Wait returns; device tensors are stream-ordered
    if (need_sync) env_->sync();
  }

  // Offline batches: AI::Act up to the batch hand-over for every replay of the batch -- Prepare
  // (elf/ai_comm.h:62-67) and OfflineLoader::extract (go/offpolicy_loader.cc:128-145) -- packed in ONE
  // pass straight into a block of the library's pinned staging ring, then ONE launch
  // (goenv_offline_batch): what every replay's board owes (the move of its last next(), or a reload's
  // reset + handicap + pre-moves), the planes under the sample's D4 code, and the labels into the
  // registered device tensors.  No allocation, no copy engine, no wait.
  void fill_inputs_offline(Collector& c) {
    const int n = (int)c.games.size();
    const int k = opt_.num_future_actions;
    const int row_w = 12 + 2 * k;
    c.boards.resize(n);
    off_lds_.resize(n);
    size_t total_moves = 0;
    for (int r = 0; r < n; ++r) {
      GameSlot& gs = games_[c.games[r]];
      OfflineLoader* ld = gs.loaders[gs.sample.loader].get();
      off_lds_[r] = ld;
      total_moves += (size_t)ld->pending().count;
    }
    const size_t moves_off = (size_t)n * row_w * 4;
    void* block = nullptr;
    chk(goenv_stage_acquire(env_->ctx(), moves_off + total_moves * 2 + 16, &block), "goenv_stage_acquire");
    int32_t* rows = static_cast<int32_t*>(block);
    int16_t* mv = reinterpret_cast<int16_t*>(static_cast<char*>(block) + moves_off);
    size_t at = 0;
    for (int r = 0; r < n; ++r) {
      const int g = c.games[r];
      GameSlot& gs = games_[g];
      OfflineLoader& ld = *off_lds_[r];
      int32_t* row = rows + (size_t)r * row_w;
      c.boards[r] = ld.board();
      const int aug = ld.draw_aug_code();
      gs.sample.aug = aug;
      const PendingOp& op = ld.pending();
      row[0] = ld.board();
      row[1] = (int32_t)((uint32_t)aug | (op.reset ? 8u : 0u) | ((uint32_t)(op.handi & 15) << 4) | ((uint32_t)op.count << 8));
      row[2] = (int32_t)at;
      row[3] = 0;
      row[4] = g; row[5] = gs.seq.seq; row[6] = gs.seq.game_counter; row[7] = gs.seq.last_terminal;
      gs.seq.seq += 1; gs.seq.last_terminal = 0;
      row[8] = ld.ply(); row[9] = ld.winner_label(); row[10] = aug; row[11] = ld.curr_game();
      ld.future_actions(aug, reinterpret_cast<int64_t*>(row + 12));
      if (op.count > 0) {
        std::memcpy(mv + at, op.rec->moves.data() + op.from, (size_t)op.count * 2);
        at += (size_t)op.count;
      }
      ld.clear_pending();
    }
    bool st_s = false, st_l = false;
    float* d_s = out_ptr<float>(c, "s", c.stage_s, (size_t)n * GOENV_FEAT_FLOATS * 4, &st_s);
    uint8_t* d_l = out_ptr<uint8_t>(c, "legal", c.stage_legal, (size_t)n * GOENV_NUM_POINTS, &st_l);
    goenv_sample_out lab;
    std::memset(&lab, 0, sizeof(lab));
    void** fields[8] = {(void**)&lab.id, (void**)&lab.seq, (void**)&lab.game_counter, (void**)&lab.last_terminal,
                        (void**)&lab.move_idx, (void**)&lab.winner, (void**)&lab.aug_code, (void**)&lab.game_record_idx};
    const Slot* host_slot[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int f = 0; f < 8; ++f) {
      auto it = c.input.find(kMetaKeys[f]);
      if (it == c.input.end()) continue;
      if (it->second.device) *fields[f] = it->second.p;
      else host_slot[f] = &it->second;
    }
    int64_t* d_offa = nullptr;
    auto ita = c.input.find("offline_a");
    if (ita != c.input.end()) {
      if (ita->second.device) d_offa = (int64_t*)ita->second.p;
      else host_slot[8] = &ita->second;
    }
    // host-resident label tensors (the reference's pinned buffers): plain copies out of the block, before
    // the launch hands the block to the device
    for (int f = 0; f < 8; ++f) {
      if (!host_slot[f]) continue;
      if (f == 3) {
        int8_t* dst = static_cast<int8_t*>(host_slot[f]->p);
        for (int r = 0; r < n; ++r) dst[r] = (int8_t)rows[(size_t)r * row_w + 4 + f];
      } else {
        int32_t* dst = static_cast<int32_t*>(host_slot[f]->p);
        for (int r = 0; r < n; ++r) dst[r] = rows[(size_t)r * row_w + 4 + f];
      }
    }
    if (host_slot[8]) {
      int64_t* dst = static_cast<int64_t*>(host_slot[8]->p);
      for (int r = 0; r < n; ++r) std::memcpy(dst + (size_t)r * k, rows + (size_t)r * row_w + 12, (size_t)k * 8);
    }
    chk(goenv_offline_batch(env_->ctx(), block, n, k, moves_off, d_s, d_l, &lab, d_offa, env_->stream()),
        "goenv_offline_batch");
    if (st_s) env_->d2h(c.input["s"].p, c.stage_s.get(), (size_t)n * GOENV_FEAT_FLOATS * 4);
    if (st_l) env_->d2h(c.input["legal"].p, c.stage_legal.get(), (size_t)n * GOENV_NUM_POINTS);
    if (st_s || st_l) env_->sync();   // host-resident tensors must be complete when Wait returns
  }

  // small host-produced field -> registered tensor (host: plain copy; device: staged H2D)
  void put(const Slot& s, const void* src, size_t bytes) {
    if (s.device) env_->h2d(s.p, src, bytes);
    else std::memcpy(s.p, src, bytes);
  }

  // reply tensor `key` as a device pointer (staging a host tensor in); nullptr if not registered
  const void* reply_ptr(Collector& c, const std::string& key, DevBuf& stage, size_t bytes) {
    auto it = c.reply.find(key);
    if (it == c.reply.end()) return nullptr;
    if (it->second.device) return it->second.p;
    stage.ensure(env_->ctx(), bytes);
    env_->h2d(stage.get(), it->second.p, bytes);
    return stage.get();
  }

  // selfplay / online: DirectPredictAI::handle_response + GoState::forward, restart on an
  // illegal move (go/go_ai.h:89-93, go/game.cc:80-93)
  // Returns true when the step was FUSED: one launch sampled / took the moves, applied them and wrote
  // the games' next samples into this collector's input tensors (the batch is then `prefilled`).
  bool steps_play(Collector& c) {
    const int n = (int)c.games.size();
    const int32_t* d_idx = c.contiguous ? nullptr : c.idx.as<int32_t>();
    const int first = c.contiguous ? c.boards[0] : 0;
    const bool online = opt_.mode == "online";
    uint32_t flags = 0;
    int human = 0;
    for (int ticket : c.games) human += games_[ticket / tick_].phase == 0;
    if (!(online && human)) flags |= GOENV_F_RESET_ILLEGAL;  // a human's illegal move is asked again
    if (opt_.end_on_two_passes || opt_.ply_cap > 0) flags |= GOENV_F_RESTART_AFTER_END;
    const int64_t* d_a = (const int64_t*)reply_ptr(c, "a", c.actions, (size_t)n * 8);
    const float* d_pi = nullptr;
    if (!d_a) {
      // the reference reads an unwritten GameState::a here (SURVEY 3.1 note ii); this build
      // samples the move from the policy head over the legal points instead
      d_pi = (const float*)reply_ptr(c, "pi", c.stage_pi, (size_t)n * GOENV_NUM_POINTS * 4);
      if (!d_pi) throw std::range_error("Steps(): the reply of collector " + std::to_string(c.gid) +
                                        " carries neither \"a\" nor \"pi\"");
    }
    if (can_fuse(c)) {
      goenv_sample_out so;
      std::memset(&so, 0, sizeof(so));
      so.mode = opt_.signal_restart ? GOENV_SEQ_EPISODES : GOENV_SEQ_RUN_ON;
      void** fields[8] = {(void**)&so.id, (void**)&so.seq, (void**)&so.game_counter, (void**)&so.last_terminal,
                          (void**)&so.move_idx, (void**)&so.winner, (void**)&so.aug_code, (void**)&so.game_record_idx};
      for (int k = 0; k < 8; ++k) {
        auto it = c.input.find(kMetaKeys[k]);
        *fields[k] = it == c.input.end() ? nullptr : it->second.p;
      }
      auto dev_in = [&](const char* key) -> void* {
        auto it = c.input.find(key);
        return it == c.input.end() ? nullptr : it->second.p;
      };
      float* d_s = (float*)dev_in("s");
      uint8_t* d_l = (uint8_t*)dev_in("legal");
      if (d_a)
        chk(goenv_ply_sel(env_->ctx(), d_idx, first, n, flags | GOENV_F_COLLECT, d_a, nullptr, d_s, d_l, &so, nullptr,
                          nullptr, env_->stream()), "goenv_ply_sel");
      else
        chk(goenv_ply_policy_sel(env_->ctx(), d_idx, first, n, flags | GOENV_F_COLLECT, d_pi, nullptr, nullptr, d_s, d_l,
                                 &so, nullptr, nullptr, env_->stream()), "goenv_ply_policy_sel");
      return true;
    }
    if (!d_a) {
      c.actions.ensure(env_->ctx(), (size_t)n * 8);
      chk(goenv_sample_policy(env_->ctx(), d_idx, first, n, d_pi, nullptr, c.actions.as<int64_t>(), env_->stream()),
          "goenv_sample_policy");
      d_a = c.actions.as<int64_t>();
    }
    int32_t* d_status = nullptr;
    if (online) {
      c.status.ensure(env_->ctx(), (size_t)n * 4);
      d_status = c.status.as<int32_t>();
    }
    chk(goenv_ply_sel(env_->ctx(), d_idx, first, n, flags, d_a, nullptr, nullptr, nullptr, nullptr, nullptr, d_status,
                      env_->stream()), "goenv_ply_sel");
    if (online) {  // host sequencing needs the legality of each move
      std::vector<int32_t> st(n);
      env_->d2h(st.data(), d_status, (size_t)n * 4);
      // a persistent search tree follows the human's moves too (moves_since, elf/mcts.h:105-110)
      const bool track = mcts_ && co_.mcts_options.persistent_tree;
      std::vector<int64_t> played(track ? n : 0);
      if (track) env_->d2h(played.data(), d_a, (size_t)n * 8);
      env_->sync();
      const bool has_human = group_of("human_actor") >= 0 && groups_[group_of("human_actor")].name == "human_actor";
      for (int r = 0; r < n; ++r) {
        GameSlot& gs = games_[c.games[r] / tick_];
        if (gs.phase == 0) {
          if (st[r] == 1) {
            gs.ply += 1; gs.phase = 1;
            if (track) gs.moves_since.push_back((int)played[r]);
          }
        } else {
          if (st[r] == 1) gs.ply += 1;
          else gs.ply = 1;  // reset
          if (has_human) gs.phase = 0;
        }
      }
    }
    return false;
  }

  // A selfplay step may be fused when the device keeps the sequencing (plain selfplay), every
  // registered input tensor is device memory (the kernel writes them in place), and no parked game
  // could starve: the group's queue is empty, or another collector of the group stays free for it.
  bool can_fuse(const Collector& c) const {
    if (!opt_.fuse_selfplay || offline_ || mcts_ || opt_.mode != "selfplay") return false;
    for (const auto& kv : c.input)
      if (!kv.second.device) return false;
    const Group& grp = groups_[c.exclusive_id];
    if (grp.ready.empty()) return true;
    for (int gid : grp.members) {
      const Collector& o = *collectors_[gid];
      if (o.gid != c.gid && !o.busy && !o.prefilled) return true;
    }
    return false;
  }

  // tree search: the replies (pi, V) of a batch of leaves.  DirectPredictAI::get_last_pi keeps the
  // five legal points of largest pi (go/go_ai.h:24-68) -- chosen on the device against each leaf's
  // legal ballots -- and the node is expanded with them (tree_search_node.h:87-111).
  void steps_mcts(Collector& c) {
    const int n = (int)c.games.size();
    constexpr int K = 5;
    const int32_t* d_idx = c.contiguous ? nullptr : (c.idx_pinned ? c.pin_idx.as<int32_t>() : c.idx.as<int32_t>());
    const int first = c.contiguous ? c.boards[0] : 0;
    const float* d_pi = (const float*)reply_ptr(c, "pi", c.stage_pi, (size_t)n * GOENV_NUM_POINTS * 4);
    auto itv = c.reply.find("V");
    if (!d_pi || itv == c.reply.end())
      throw std::range_error("Steps(): tree search needs \"pi\" and \"V\" in the reply of collector " + std::to_string(c.gid));
    // the kernel writes the top-k lists straight into pinned host memory, V is copied into it: one synchronise, no staging
    c.pin_topa.ensure((size_t)n * K * 4);
    c.pin_topp.ensure((size_t)n * K * 4);
    c.pin_v.ensure((size_t)n * 4);
    chk(goenv_topk_policy(env_->ctx(), d_idx, first, n, d_pi, nullptr, K, c.pin_topa.as<int32_t>(), c.pin_topp.as<float>(),
                          env_->stream()), "goenv_topk_policy");
    if (itv->second.device) env_->d2h(c.pin_v.as<float>(), itv->second.p, (size_t)n * 4);
    env_->sync();
    if (!itv->second.device) std::memcpy(c.pin_v.as<float>(), itv->second.p, (size_t)n * 4);
    const int32_t* top_a = c.pin_topa.as<int32_t>();
    const float *top_p = c.pin_topp.as<float>(), *value = c.pin_v.as<float>();
    // The trees of different games share nothing: the games of the batch are expanded and run on
    // to their next leaves in parallel on the host pool (rows of one game stay in row order), then
    // their leaves are queued in game order, so the batches are the same for any number of threads.
    std::vector<int> touched;
    for (int r = 0; r < n; ++r) touched.push_back(c.games[r] / tick_);
    std::sort(touched.begin(), touched.end());
    touched.erase(std::unique(touched.begin(), touched.end()), touched.end());
    std::vector<std::vector<int>> rows_of(touched.size());
    for (int r = 0; r < n; ++r) {
      const int g = c.games[r] / tick_;
      rows_of[std::lower_bound(touched.begin(), touched.end(), g) - touched.begin()].push_back(r);
    }
    std::vector<std::vector<LeafRequest>> next(touched.size());
    pool_.run((int)touched.size(), [&](int i) {
      TreeSearch& ts = *searches_[touched[i]];
      for (int r : rows_of[i])
        ts.expand(c.games[r] % tick_ - 1, &top_a[(size_t)r * K], &top_p[(size_t)r * K], K, value[r]);
      ts.advance(&next[i]);
    });
    for (size_t i = 0; i < touched.size(); ++i) queue_leaves(touched[i], next[i]);
  }

  // ------------------------------------------------------------------------------------------
  // General path: history length T > 1 and samples that several exclusive groups take
  // (HistT elf/hist.h:19-66, CondPerGroupT::Check and SendDataWaitReply elf/comm_template.h:68-91,
  // 215-261, CopyToMem / CopyFromMem elf/hist.h:71-121).  The plain Go topologies (every group T = 1,
  // one group per sample name) never come here.
  // ------------------------------------------------------------------------------------------
  static bool ring_key(const std::string& k) { return k == "s" || k == "legal" || k == "pi" || k == "a" || k == "V"; }

  std::vector<std::string> possible_names() const {
    if (offline_) return {""};
    if (opt_.mode == "online") return {"human_actor", "actor"};
    return {"actor"};
  }

  void setup_general() {
    general_ = false;
    reply_a_ = reply_pi_ = false;
    hist_T_ = std::max(1, co_.T);
    for (const auto& c : collectors_) {
      reply_a_ |= c->reply.count("a") != 0;
      reply_pi_ |= c->reply.count("pi") != 0;
      if (c->gstat.hist_len > 1) general_ = true;
      for (const auto* m : {&c->input, &c->reply})
        for (const auto& kv : *m)
          if (kv.second.T > 1) general_ = true;
    }
    for (const std::string& name : possible_names()) {
      int takers = 0;
      for (const Group& g : groups_) takers += g.name.empty() || g.name == name;
      if (takers > 1) general_ = true;
    }
    if (!general_) return;
    if (mcts_) throw std::range_error("Start(): use_mcts with hist_len > 1 or several groups per sample is not supported");
    const size_t rows = (size_t)co_.num_games * hist_T_;
    for (const auto& c : collectors_)
      for (const auto* m : {&c->input, &c->reply})
        for (const auto& kv : *m)
          if (ring_key(kv.first) && !ring_.count(kv.first)) {
            // cleared once: a sample that no group takes (history gate) acts on the newest entry's reply,
            // which nobody has written -- the reference's GameState starts zeroed too (`a` = 0, `pi` = 0)
            DevBuf& rb = ring_[kv.first];
            rb.ensure(env_->ctx(), rows * kv.second.row);
            chk(goenv_memset(env_->ctx(), rb.get(), 0, rows * kv.second.row, env_->stream()), "goenv_memset");
          }
  }

  int ring_row(const GameSlot& gs, int g, int back) const {
    if (back >= gs.hist_size) back = gs.hist_size - 1;   // "fill them with the oldest hist", elf/hist.h:90-98
    return g * hist_T_ + (gs.hist_head - back + 2 * hist_T_) % hist_T_;
  }

  int board_of_game(const GameSlot& gs, int g) const {
    return offline_ ? gs.loaders[gs.sample.loader]->board() : g;
  }

  // AI::Act up to SendDataWaitReply for every game of the list: Prepare pushes a history entry, extract
  // fills it (labels on the host, planes and mask into the device rings in ONE launch for the whole
  // list), then the sample is offered to every exclusive group.  A sample that no group takes (the
  // history gate) lets the game act at once, as the reference does -- hence the loop.
  void produce_general(std::vector<int> games) {
    const int k = opt_.num_future_actions;
    while (!games.empty()) {
      const int m = (int)games.size();
      std::vector<int32_t> boards(m), augs(m, 0), rows(m);
      for (int i = 0; i < m; ++i) {
        const int g = games[i];
        GameSlot& gs = games_[g];
        if (offline_) {
          gs.sample.loader = (int)(gs.rng() % gs.loaders.size());  // go/game.cc:96
          if (gs.loaders[gs.sample.loader]->just_started()) {      // before_act -> AIComm::Restart
            gs.seq.seq = 0; gs.seq.game_counter += 1; gs.seq.last_terminal = 1;
          }
        }
        gs.hist_head = (gs.hist_head + 1) % hist_T_;               // HistT::Prepare -> GetRoom
        gs.hist_size = std::min(gs.hist_size + 1, hist_T_);
        HistEntry& e = gs.hist[gs.hist_head];
        e.meta[0] = g; e.meta[1] = gs.seq.seq; e.meta[2] = gs.seq.game_counter; e.meta[3] = gs.seq.last_terminal;
        gs.seq.seq += 1; gs.seq.last_terminal = 0;                 // SeqInfo::Inc
        e.offline_a.assign(k, 0);
        if (offline_) {                                            // offpolicy_loader.cc:128-145
          OfflineLoader& ld = *gs.loaders[gs.sample.loader];
          e.meta[7] = ld.curr_game(); e.meta[4] = ld.ply(); e.meta[5] = ld.winner_label();
          gs.sample.aug = ld.draw_aug_code();
          e.meta[6] = gs.sample.aug;
          ld.future_actions(gs.sample.aug, e.offline_a.data());
        } else {                                                   // go/go_ai.h:79-87,106-113
          gs.sample.aug = 0;
          e.meta[4] = gs.ply; e.meta[5] = 0; e.meta[6] = 0; e.meta[7] = -1;
        }
        boards[i] = board_of_game(gs, g);
        augs[i] = gs.sample.aug;
        rows[i] = g * hist_T_ + gs.hist_head;
      }
      // extract: planes and mask of all listed boards -> staging rows -> their ring rows
      const bool want_s = ring_.count("s") != 0, want_l = ring_.count("legal") != 0;
      if (want_s || want_l) {
        g_idx_.ensure(env_->ctx(), (size_t)m * 4); g_rows_.ensure(env_->ctx(), (size_t)m * 4); g_act_.ensure(env_->ctx(), (size_t)m * 8);
        env_->h2d(g_idx_.get(), boards.data(), (size_t)m * 4);
        env_->h2d(g_rows_.get(), rows.data(), (size_t)m * 4);
        env_->h2d(g_act_.get(), augs.data(), (size_t)m * 4);
        if (want_s) g_stage_s_.ensure(env_->ctx(), (size_t)m * GOENV_FEAT_FLOATS * 4);
        if (want_l) g_stage_l_.ensure(env_->ctx(), (size_t)m * GOENV_NUM_POINTS);
        if (offline_) {   // what the boards owe + the planes, one launch (labels stay with the host's histories)
          std::vector<OfflineLoader*> lds(m);
          for (int i = 0; i < m; ++i) lds[i] = games_[games[i]].loaders[games_[games[i]].sample.loader].get();
          offline_launch(lds, augs.data(), nullptr, nullptr, want_s ? g_stage_s_.as<float>() : nullptr,
                         want_l ? g_stage_l_.as<uint8_t>() : nullptr, nullptr, nullptr);
        } else {
          chk(goenv_collect_sel(env_->ctx(), g_idx_.as<int32_t>(), 0, m, g_act_.as<int32_t>(),
                                want_s ? g_stage_s_.as<float>() : nullptr, want_l ? g_stage_l_.as<uint8_t>() : nullptr, nullptr,
                                env_->stream()), "goenv_collect_sel");
        }
        if (want_s)
          chk(goenv_copy_rows(env_->ctx(), g_stage_s_.get(), nullptr, ring_["s"].get(), g_rows_.as<int32_t>(), m,
                              (size_t)GOENV_FEAT_FLOATS * 4, env_->stream()), "goenv_copy_rows");
        if (want_l)
          chk(goenv_copy_rows(env_->ctx(), g_stage_l_.get(), nullptr, ring_["legal"].get(), g_rows_.as<int32_t>(), m,
                              (size_t)GOENV_NUM_POINTS, env_->stream()), "goenv_copy_rows");
      }
      // SendDataWaitReply: one taker per exclusive group whose name and history gate pass
      std::vector<int> free_to_act;
      for (int g : games)
        if (route(g) == 0) free_to_act.push_back(g);
      if (!free_to_act.empty()) act_general(free_to_act);
      games.swap(free_to_act);
    }
  }

  int route(int g) {
    GameSlot& gs = games_[g];
    const std::string name = sample_name(gs);
    const HistEntry& e = gs.hist[gs.hist_head];
    int sent = 0;
    for (size_t i = 0; i < groups_.size(); ++i) {
      const GroupStat& st = collectors_[groups_[i].members[0]]->gstat;
      if (!st.name.empty() && st.name != name) continue;           // check_name, comm_template.h:46-54
      CondState& cs = gs.cond[i];
      if (e.meta[2] > cs.game_counter) {                           // a new game: no frame is missed, :78-83
        cs.game_counter = e.meta[2];
        cs.last_used_seq -= cs.last_seq + 1;
      }
      cs.last_seq = e.meta[1];
      if (gs.hist_size < st.hist_len || e.meta[1] - cs.last_used_seq < st.hist_len - 1) continue;   // :88, overlap 1
      cs.last_used_seq = e.meta[1];
      enqueue((int)i, g * tick_);
      ++sent;
    }
    gs.pending = sent;
    return sent;
  }

  // handle_response + forward for games whose every taker is done (or that had none)
  void act_general(const std::vector<int>& games) {
    const int m = (int)games.size();
    if (offline_) {   // OfflineLoader::handle_response -> next(): host bookkeeping (see steps_offline)
      for (int g : games) {
        GameSlot& gs = games_[g];
        gs.loaders[gs.sample.loader]->next();
      }
      return;
    }
    // play modes: the move is the newest history entry's reply -- `a` if a collector replies it, else
    // sampled from `pi` over the legal points
    const bool have_a = reply_a_, have_pi = reply_pi_;
    if (!have_a && !have_pi)
      throw std::range_error("no collector replies \"a\" or \"pi\": the games cannot move");
    std::vector<int32_t> boards(m), rows(m);
    int human = 0;
    for (int i = 0; i < m; ++i) {
      const GameSlot& gs = games_[games[i]];
      if (gs.pending != 0 || gs.hist_size == 0) throw std::range_error("internal: a game acts while its sample is still out");
      boards[i] = games[i];
      rows[i] = games[i] * hist_T_ + gs.hist_head;
      human += gs.phase == 0;
    }
    g_idx_.ensure(env_->ctx(), (size_t)m * 4); g_rows_.ensure(env_->ctx(), (size_t)m * 4);
    g_act_.ensure(env_->ctx(), (size_t)m * 8); g_status_.ensure(env_->ctx(), (size_t)m * 4);
    env_->h2d(g_idx_.get(), boards.data(), (size_t)m * 4);
    env_->h2d(g_rows_.get(), rows.data(), (size_t)m * 4);
    if (have_a) {
      chk(goenv_copy_rows(env_->ctx(), ring_["a"].get(), g_rows_.as<int32_t>(), g_act_.get(), nullptr, m, 8, env_->stream()),
          "goenv_copy_rows");
    } else {
      g_pi_.ensure(env_->ctx(), (size_t)m * GOENV_NUM_POINTS * 4);
      chk(goenv_copy_rows(env_->ctx(), ring_["pi"].get(), g_rows_.as<int32_t>(), g_pi_.get(), nullptr, m,
                          (size_t)GOENV_NUM_POINTS * 4, env_->stream()), "goenv_copy_rows");
      chk(goenv_sample_policy(env_->ctx(), g_idx_.as<int32_t>(), 0, m, g_pi_.as<float>(), nullptr, g_act_.as<int64_t>(),
                              env_->stream()), "goenv_sample_policy");
    }
    const bool online = opt_.mode == "online";
    uint32_t flags = 0;
    if (!(online && human)) flags |= GOENV_F_RESET_ILLEGAL;
    if (opt_.end_on_two_passes || opt_.ply_cap > 0) flags |= GOENV_F_RESTART_AFTER_END;
    chk(goenv_ply_sel(env_->ctx(), g_idx_.as<int32_t>(), 0, m, flags, g_act_.as<int64_t>(), nullptr, nullptr, nullptr, nullptr,
                      nullptr, g_status_.as<int32_t>(), env_->stream()), "goenv_ply_sel");
    std::vector<int32_t> st(m);
    env_->d2h(st.data(), g_status_.get(), (size_t)m * 4);
    env_->sync();
    const bool has_human = online && group_of("human_actor") >= 0 && groups_[group_of("human_actor")].name == "human_actor";
    for (int i = 0; i < m; ++i) {
      GameSlot& gs = games_[games[i]];
      const bool restarted = st[i] == 2 || (st[i] == 0 && (flags & GOENV_F_RESET_ILLEGAL));
      if (gs.phase == 0) {                    // a human's move: asked again when illegal
        if (st[i] == 1) { gs.ply += 1; gs.phase = 1; }
        else if (restarted) gs.ply = 1;
        continue;
      }
      if (st[i] == 1) gs.ply += 1;
      else gs.ply = 1;
      if (restarted && opt_.signal_restart) {  // AICommT::Restart, elf/ai_comm.h:91-95
        gs.seq.seq = 0; gs.seq.game_counter += 1; gs.seq.last_terminal = 1;
      }
      if (has_human) gs.phase = 0;
    }
  }

  // row lists of one tensor: entry t of row r <- history entry (T - 1 - t) back from the newest
  void window_rows(const Collector& c, int T, std::vector<int32_t>* hist_rows, std::vector<int32_t>* tensor_rows) const {
    const int n = (int)c.games.size();
    hist_rows->resize((size_t)T * n);
    tensor_rows->resize((size_t)T * n);
    for (int t = 0; t < T; ++t)
      for (int r = 0; r < n; ++r) {
        const int g = c.games[r] / tick_;
        (*hist_rows)[(size_t)t * n + r] = ring_row(games_[g], g, T - 1 - t);
        (*tensor_rows)[(size_t)t * n + r] = t * c.batchsize + r;
      }
  }

  // CopyToMem (elf/hist.h:71-100): every registered input key, time-major, oldest entry first
  void fill_inputs_general(Collector& c) {
    const int n = (int)c.games.size();
    c.boards.assign(n, 0);
    bool need_sync = false;
    std::vector<int32_t> hrows, trows;
    std::vector<std::vector<char>> bufs;
    for (auto& kv : c.input) {
      const std::string& key = kv.first;
      Slot& sl = kv.second;
      const int T = sl.T;
      window_rows(c, T, &hrows, &trows);
      const size_t cnt = hrows.size();
      if (ring_key(key)) {
        g_rows_.ensure(env_->ctx(), cnt * 8);
        env_->h2d(g_rows_.get(), hrows.data(), cnt * 4);
        env_->h2d(g_rows_.as<int32_t>() + cnt, trows.data(), cnt * 4);
        void* dst = sl.p;
        if (!sl.device) {
          g_stage_.ensure(env_->ctx(), sl.bytes);
          dst = g_stage_.get();
        }
        chk(goenv_copy_rows(env_->ctx(), ring_[key].get(), g_rows_.as<int32_t>(), dst, g_rows_.as<int32_t>() + cnt, (int)cnt,
                            sl.row, env_->stream()), "goenv_copy_rows");
        if (!sl.device) {   // only the rows of this batch reach the caller's buffer
          for (int t = 0; t < T; ++t)
            env_->d2h((char*)sl.p + (size_t)t * c.batchsize * sl.row, (char*)dst + (size_t)t * c.batchsize * sl.row, (size_t)n * sl.row);
          env_->sync();     // the staging buffer is reused by the next key
          need_sync = false;
        }
        continue;
      }
      // host-kept keys: the eight labels and offline_a
      int mk = -1;
      for (int i = 0; i < 8; ++i)
        if (key == kMetaKeys[i]) mk = i;
      bufs.emplace_back((size_t)T * n * sl.row);       // kept until the copies below have been waited for
      std::vector<char>& buf = bufs.back();
      for (int t = 0; t < T; ++t) {
        char* base = buf.data() + (size_t)t * n * sl.row;
        for (int r = 0; r < n; ++r) {
          const int g = c.games[r] / tick_;
          const GameSlot& gs = games_[g];
          const HistEntry& e = gs.hist[ring_row(gs, g, T - 1 - t) - g * hist_T_];
          char* out = base + (size_t)r * sl.row;
          if (mk == 3) { const int8_t v = (int8_t)e.meta[3]; std::memcpy(out, &v, 1); }
          else if (mk >= 0) std::memcpy(out, &e.meta[mk], 4);
          else if (key == "offline_a") std::memcpy(out, e.offline_a.data(), sl.row);
          else std::memset(out, 0, sl.row);
        }
        char* dst = (char*)sl.p + (size_t)t * c.batchsize * sl.row;
        if (sl.device) { env_->h2d(dst, base, (size_t)n * sl.row); need_sync = true; }
        else std::memcpy(dst, base, (size_t)n * sl.row);
      }
    }
    if (need_sync) env_->sync();   // `bufs` is pageable host memory: be done with it before it goes
  }

  // CopyFromMem (elf/hist.h:102-121) of the reply keys into the games' histories, then every game whose
  // takers are all done acts and offers its next sample
  void steps_general(Collector& c) {
    std::vector<int32_t> hrows, trows;
    for (auto& kv : c.reply) {
      if (!ring_key(kv.first)) continue;
      Slot& sl = kv.second;
      window_rows(c, sl.T, &hrows, &trows);
      const size_t cnt = hrows.size();
      g_rows_.ensure(env_->ctx(), cnt * 8);
      env_->h2d(g_rows_.get(), hrows.data(), cnt * 4);
      env_->h2d(g_rows_.as<int32_t>() + cnt, trows.data(), cnt * 4);
      const void* src = sl.p;
      if (!sl.device) {
        g_stage_.ensure(env_->ctx(), sl.bytes);
        env_->h2d(g_stage_.get(), sl.p, sl.bytes);
        src = g_stage_.get();
      }
      chk(goenv_copy_rows(env_->ctx(), src, g_rows_.as<int32_t>() + cnt, ring_[kv.first].get(), g_rows_.as<int32_t>(), (int)cnt,
                          sl.row, env_->stream()), "goenv_copy_rows");
      if (!sl.device) env_->sync();
    }
    std::vector<int> released;
    for (int ticket : c.games) {
      GameSlot& gs = games_[ticket / tick_];
      if (--gs.pending == 0) released.push_back(ticket / tick_);
    }
    c.busy = false;
    c.games.clear();
    c.boards.clear();
    if (!released.empty()) {
      act_general(released);
      produce_general(released);
    }
  }

  // offline: OfflineLoader::handle_response -> next() for every sample of the batch
  // (go/offpolicy_loader.cc:147-152,120-126); the common case is one batched forward
  void steps_offline(Collector& c) {
    for (int ticket : c.games) {
      GameSlot& gs = games_[ticket];
      gs.loaders[gs.sample.loader]->next();   // host bookkeeping; the board is brought up to date when it is next sampled
    }
  }

  // Packs what the replays' boards owe (+ the sample each is about to give, if any) into a block of the
  // library's pinned staging ring and issues the one launch that does it all (goenv_offline_batch).
  // aug / meta / off_a describe the samples (NULL: no sample, boards only); d_s / d_l / lab / d_offa are the
  // device destinations (any may be NULL).
  void offline_launch(const std::vector<OfflineLoader*>& lds, const int32_t* aug, const int32_t* const* meta,
                      const int64_t* off_a, float* d_s, uint8_t* d_l, const goenv_sample_out* lab, int64_t* d_offa) {
    const int n = (int)lds.size();
    if (n == 0) return;
    const int k = std::min(opt_.num_future_actions, GOENV_MAX_FUTURE);
    const int row_w = 12 + 2 * k;
    size_t total_moves = 0;
    for (OfflineLoader* ld : lds) total_moves += (size_t)ld->pending().count;
    const size_t moves_off = (size_t)n * row_w * 4;
    void* block = nullptr;
    chk(goenv_stage_acquire(env_->ctx(), moves_off + total_moves * 2 + 16, &block), "goenv_stage_acquire");
    int32_t* rows = static_cast<int32_t*>(block);
    int16_t* mv = reinterpret_cast<int16_t*>(static_cast<char*>(block) + moves_off);
    size_t at = 0;
    for (int r = 0; r < n; ++r) {
      OfflineLoader& ld = *lds[r];
      const PendingOp& op = ld.pending();
      int32_t* row = rows + (size_t)r * row_w;
      row[0] = ld.board();
      row[1] = (int32_t)((uint32_t)(aug ? aug[r] & 7 : 0) | (op.reset ? 8u : 0u) | ((uint32_t)(op.handi & 15) << 4) |
                         ((uint32_t)op.count << 8));
      row[2] = (int32_t)at;
      row[3] = 0;
      for (int f = 0; f < 8; ++f) row[4 + f] = meta ? meta[f][r] : 0;
      if (off_a && k > 0) std::memcpy(row + 12, off_a + (size_t)r * opt_.num_future_actions, (size_t)k * 8);
      else std::memset(row + 12, 0, (size_t)k * 8);
      if (op.count > 0) {
        std::memcpy(mv + at, op.rec->moves.data() + op.from, (size_t)op.count * 2);
        at += (size_t)op.count;
      }
      ld.clear_pending();
    }
    chk(goenv_offline_batch(env_->ctx(), block, n, k, moves_off, d_s, d_l, lab, d_offa, env_->stream()),
        "goenv_offline_batch");
  }

  // every replay whose board owes something, in launches of at most 16,384 rows (Start(): the first
  // load of each replay fast-forwards up to half a record)
  void offline_flush_all() {
    std::vector<OfflineLoader*> lds;
    auto go = [&]() {
      offline_launch(lds, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
      lds.clear();
    };
    for (GameSlot& gs : games_)
      for (auto& ld : gs.loaders)
        if (!ld->pending().empty()) {
          lds.push_back(ld.get());
          if (lds.size() == 16384) go();
        }
    go();
  }
};

// ------------------------------------------------------------------------------------------

PYBIND11_MODULE(go_game, m) {
  m.doc() = "MI355X-native drop-in for ELF's go_game module (DarkForest Go environment)";

  py::class_<TSOptions>(m, "TSOptions")
      .def(py::init<>())
      .def_readwrite("max_num_moves", &TSOptions::max_num_moves)
      .def_readwrite("num_threads", &TSOptions::num_threads)
      .def_readwrite("num_rollout_per_thread", &TSOptions::num_rollout_per_thread)
      .def_readwrite("verbose", &TSOptions::verbose)
      .def_readwrite("persistent_tree", &TSOptions::persistent_tree)
      .def_readwrite("pick_method", &TSOptions::pick_method)
      .def_readwrite("use_prior", &TSOptions::use_prior)
      .def_readwrite("pseudo_games", &TSOptions::pseudo_games)
      .def_readwrite("verbose_time", &TSOptions::verbose_time)
      .def_readwrite("save_tree_filename", &TSOptions::save_tree_filename);

  py::class_<ContextOptions>(m, "ContextOptions")
      .def(py::init<>())
      .def("print", &ContextOptions::print)
      .def_readwrite("num_games", &ContextOptions::num_games)
      .def_readwrite("max_num_threads", &ContextOptions::max_num_threads)
      .def_readwrite("T", &ContextOptions::T)
      .def_readwrite("verbose_comm", &ContextOptions::verbose_comm)
      .def_readwrite("verbose_collector", &ContextOptions::verbose_collector)
      .def_readwrite("wait_per_group", &ContextOptions::wait_per_group)
      .def_readwrite("mcts_options", &ContextOptions::mcts_options)
      .def_readwrite("num_collectors", &ContextOptions::num_collectors);

  py::class_<GameOptions>(m, "GameOptions")
      .def(py::init<>())
      .def_readwrite("seed", &GameOptions::seed)
      .def_readwrite("mode", &GameOptions::mode)
      .def_readwrite("data_aug", &GameOptions::data_aug)
      .def_readwrite("start_ratio_pre_moves", &GameOptions::start_ratio_pre_moves)
      .def_readwrite("ratio_pre_moves", &GameOptions::ratio_pre_moves)
      .def_readwrite("move_cutoff", &GameOptions::move_cutoff)
      .def_readwrite("num_planes", &GameOptions::num_planes)
      .def_readwrite("num_future_actions", &GameOptions::num_future_actions)
      .def_readwrite("list_filename", &GameOptions::list_filename)
      .def_readwrite("verbose", &GameOptions::verbose)
      .def_readwrite("num_games_per_thread", &GameOptions::num_games_per_thread)
      .def_readwrite("use_mcts", &GameOptions::use_mcts)
      .def_readwrite("gpu", &GameOptions::gpu)
      .def_readwrite("komi", &GameOptions::komi)
      .def_readwrite("ply_cap", &GameOptions::ply_cap)
      .def_readwrite("end_on_two_passes", &GameOptions::end_on_two_passes)
      .def_readwrite("signal_restart", &GameOptions::signal_restart)
      .def_readwrite("board_id_base", &GameOptions::board_id_base)
      .def_readwrite("preset_zero_planes", &GameOptions::preset_zero_planes)
      .def_readwrite("fuse_selfplay", &GameOptions::fuse_selfplay);

  py::class_<EntryInfo>(m, "EntryInfo")
      .def(py::init<>())
      .def_readwrite("key", &EntryInfo::key)
      .def_readwrite("type", &EntryInfo::type)
      .def_readwrite("sz", &EntryInfo::sz)
      .def_readwrite("p", &EntryInfo::p)
      .def_readwrite("byte_size", &EntryInfo::byte_size);

  py::class_<GroupStat>(m, "GroupStat")
      .def(py::init<>())
      .def("info", &GroupStat::info)
      .def_readwrite("hist_len", &GroupStat::hist_len)
      .def_readwrite("name", &GroupStat::name);

  py::class_<Infos>(m, "Infos")
      .def(py::init<>())
      .def("batchsize", &Infos::batchsize)
      .def_readwrite("gid", &Infos::gid)
      .def_readwrite("s", &Infos::s);

  py::class_<GameContext>(m, "GameContext")
      .def(py::init<const ContextOptions&, const GameOptions&>())
      .def("Wait", &GameContext::Wait, py::call_guard<py::gil_scoped_release>())
      .def("WaitGroup", &GameContext::WaitGroup, py::call_guard<py::gil_scoped_release>())
      .def("Steps", &GameContext::Steps, py::call_guard<py::gil_scoped_release>())
      .def("Version", &GameContext::Version)
      .def("PrintSummary", &GameContext::PrintSummary)
      .def("CreateGroupStat", &GameContext::CreateGroupStat, py::return_value_policy::copy)
      .def("AddCollectors", &GameContext::AddCollectors)
      .def("Start", &GameContext::Start)
      .def("Stop", &GameContext::Stop)
      .def("__len__", &GameContext::size)
      .def("AddTensor", &GameContext::AddTensor)
      .def("GetTensorSpec", &GameContext::GetTensorSpec, py::return_value_policy::copy)
      .def("GetCollectorInfos", &GameContext::GetCollectorInfos)
      .def("GetParams", &GameContext::GetParams)
      .def("ShowBoard", &GameContext::ShowBoard)
      .def("SetStream", &GameContext::SetStream)
      .def("Stats", &GameContext::Stats)
      .def("StatsPtr", &GameContext::StatsPtr)
      .def("FuseMax", &GameContext::FuseMax)
      .def("OfflineBadMoves", &GameContext::OfflineBadMoves)
      .def("HostTimes", &GameContext::HostTimes, py::arg("reset") = false)
      .def("SearchInfo", &GameContext::SearchInfo);

  // the host-side tree search by itself (CPU tests drive it against a board engine of their own)
  py::class_<TreeSearch>(m, "TreeSearch")
      .def(py::init<const TSOptions&, int, int>())
      .def("begin", &TreeSearch::begin, py::arg("root_slot"), py::arg("moves_since") = std::vector<int>())
      .def("reset", &TreeSearch::reset)
      .def_static("pool_slots", &TreeSearch::pool_slots)
      .def("child_edges", &TreeSearch::child_edges)
      .def("root_count", &TreeSearch::root_count)
      .def("free_slots", &TreeSearch::free_slots)
      .def("advance", [](TreeSearch& ts) {
        std::vector<LeafRequest> rq;
        ts.advance(&rq);
        py::list out;
        for (const LeafRequest& r : rq) {
          py::dict d;
          d["thread"] = r.thread; d["node"] = r.node; d["slot"] = r.slot; d["parent_slot"] = r.parent_slot;
          d["action"] = r.action; d["depth"] = r.depth;
          out.append(d);
        }
        return out;
      })
      .def("expand", [](TreeSearch& ts, int thread, const std::vector<int32_t>& actions, const std::vector<float>& priors,
                        float value) {
        if (actions.size() != priors.size()) throw std::range_error("expand: actions and priors differ in length");
        ts.expand(thread, actions.data(), priors.data(), (int)actions.size(), value);
      })
      .def("done", &TreeSearch::done)
      .def("best_action", [](const TreeSearch& ts, unsigned seed) { std::mt19937 r(seed); return ts.best_action(&r); },
           py::arg("seed") = 1u)
      .def("root_edges", &TreeSearch::root_edges)
      .def("num_nodes", &TreeSearch::num_nodes)
      .def("seq_of", &TreeSearch::seq_of)
      .def("bump_seq", &TreeSearch::bump_seq);

  // host-side helpers exposed for tests and tools
  m.def("parse_sgf", [](const std::string& text) {
    const SgfRecord r = parse_sgf(text);
    py::dict d;
    d["ok"] = r.ok; d["board_size"] = r.board_size; d["handi"] = r.handi; d["komi"] = r.komi;
    d["winner"] = r.winner; d["num_moves"] = r.num_moves;
    d["moves"] = std::vector<int>(r.moves.begin(), r.moves.end());
    d["players"] = std::vector<int>(r.players.begin(), r.players.end());
    return d;
  });
  m.def("corpus_names", [](const std::string& list_or_tar) {   // what OfflineLoader::InitSharedBuffer lists
    SgfCorpus c;
    c.load_list(list_or_tar);
    std::vector<std::string> v;
    for (size_t i = 0; i < c.size(); ++i) v.push_back(c.name(i));
    return v;
  });
  m.def("corpus_record", [](const std::string& list_or_tar, size_t i) {
    SgfCorpus c;
    py::dict d;
    if (!c.load_list(list_or_tar) || i >= c.size()) return d;
    const SgfRecord& r = c.get(i);
    d["ok"] = r.ok; d["num_moves"] = r.num_moves; d["handi"] = r.handi; d["winner"] = r.winner;
    d["board_size"] = r.board_size;
    d["moves"] = std::vector<int>(r.moves.begin(), r.moves.end());
    return d;
  });
  // the offline loader's own Mersenne twister against the standard engine (same seeds as GoGame uses)
  m.def("mt19937_matches_std", [](uint32_t seed, int draws) {
    Mt19937 a(seed);
    std::mt19937 b(seed);
    for (int i = 0; i < draws; ++i)
      if (a() != (uint32_t)b()) return false;
    return true;
  });
  m.def("aug_action", &goenv_aug_action);        // BoardFeature::Coord2Action, go/board_feature.h:62-65
  m.def("aug_action_inv", &goenv_aug_action_inv);  // BoardFeature::Action2Coord, :67-71
}
