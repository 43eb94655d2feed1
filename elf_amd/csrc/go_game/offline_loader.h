// offline_loader.h -- SGF replay feeding the batch (the reference's OfflineLoader,
// go/offpolicy_loader.{h,cc} on top of elf/replay_loader.h), with the board living on the GPU.
//
// What stays on the host: the SGF records, the position of each replay inside its record, the
// std::mt19937 streams (same engine, same seeds, same call order as the reference, so that a
// fixed GameOptions.seed gives the reference's samples) and the per-game sequencing.
// What runs on the device: every rules operation -- Reset/ApplyHandicap, the pre-move fast-forward of
// after_reload, the per-sample forward() and Extract.
//
// The host never waits for the device.  A replay's board is always "its record, replayed from the
// start, up to move idx", and whether forward() accepts move i of a record does not depend on who asks:
// it is found once per record (SgfRecord::legal_prefix, one device replay when the record is first
// used).  So next() / reload() only RECORD what the board owes -- a reset with the record's handicap
// and a run of consecutive record moves (PendingOp) -- and the GameContext sends that, for all
// replays of a batch, inside the one launch that extracts the batch (goenv_offline_batch).
#pragma once

#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "device_env.h"
#include "sgf_record.h"

namespace gogame {

struct OfflineOptions {
  int num_future_actions = 3;
  int data_aug = -1;
  float start_ratio_pre_moves = 0.5f;
  float ratio_pre_moves = 0.0f;
  int move_cutoff = -1;
  bool verbose = false;
};

// The shared, lazily filled record store (OfflineLoader::InitSharedBuffer +
// elf/shared_replay_buffer.h): list file -> names; records parsed on first use.
class SgfCorpus {
 public:
  // elf::tar::file_is_tar, elf/tar_loader.cc:16-18
  static bool is_tar(const std::string& filename) {
    return filename.substr(filename.find_last_of(".") + 1) == "tar";
  }

  bool load_list(const std::string& list_filename) {
    names_.clear();
    cache_.clear();
    tar_.clear();
    if (is_tar(list_filename)) return load_tar(list_filename);
    std::ifstream f(list_filename);
    if (!f.is_open()) return false;
    names_.clear();
    for (std::string line; std::getline(f, line);) names_.push_back(line);
    while (!names_.empty() && names_.back().empty()) names_.pop_back();
    const size_t slash = list_filename.rfind('/');
    dir_ = slash == std::string::npos ? std::string() : list_filename.substr(0, slash + 1);
    return !names_.empty();
  }
  size_t size() const { return names_.size(); }
  const std::string& name(size_t i) const { return names_[i]; }
  const SgfRecord& get(size_t i) {
    auto it = cache_.find(i);
    if (it == cache_.end()) {
      SgfRecord rec = tar_.empty() ? load_sgf_file(dir_ + names_[i]) : parse_sgf(tar_member(names_[i]));
      it = cache_.emplace(i, std::unique_ptr<SgfRecord>(new SgfRecord(std::move(rec)))).first;
    }
    return *it->second;
  }

  // TarLoader::Load (elf/tar_loader.cc:32-41): the first member of that name, cut at its first NUL
  std::string tar_member(const std::string& name) const {
    for (const TarEntry& e : entries_)
      if (e.name == name) {
        std::string text = tar_.substr(e.offset, e.size);
        const size_t nul = text.find('\0');
        return nul == std::string::npos ? text : text.substr(0, nul);
      }
    return std::string();
  }

 private:
  struct TarEntry {
    std::string name;
    size_t offset, size;
  };
  std::vector<std::string> names_;
  std::string dir_;
  std::map<size_t, std::unique_ptr<SgfRecord>> cache_;
  std::string tar_;               // whole archive (SGF corpora are small text)
  std::vector<TarEntry> entries_;

  // TarLoader::List (elf/tar_loader.cc:24-31): every 512-byte header in archive order, whatever
  // its type, until the null record (vendor microtar reads name at 0, octal size at 124 and stops
  // at a header whose checksum field starts with NUL).
  bool load_tar(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) return false;
    std::stringstream ss;
    ss << f.rdbuf();
    tar_ = ss.str();
    entries_.clear();
    size_t pos = 0;
    while (pos + 512 <= tar_.size()) {
      const char* h = tar_.data() + pos;
      if (h[148] == '\0') break;
      TarEntry e;
      e.name = std::string(h, strnlen(h, 100));
      e.size = (size_t)std::strtoul(std::string(h + 124, 12).c_str(), nullptr, 8);
      e.offset = pos + 512;
      if (e.offset + e.size > tar_.size()) break;
      entries_.push_back(e);
      names_.push_back(e.name);
      pos = e.offset + ((e.size + 511) / 512) * 512;
    }
    if (tar_.empty()) tar_.assign(1, '\0');  // marks "tar mode" even for an empty archive
    return !names_.empty();
  }
};

// std::mt19937 restated (same recurrence, tempering and seeding, so the same stream: checked against the
// standard engine by go_game.mt19937_matches_std) with its position index in front of the state words.
// A context holds one engine per replay -- 76,800 of them, 190 MB, at 256 games x 300 replays -- and a
// batch draws from 256 random ones: the index and the word it points at are the two cache lines a draw
// touches, and prefetch() can name both (the standard engine keeps its index behind the 2.5-KB state).
class Mt19937 {
 public:
  explicit Mt19937(uint32_t seed = 5489u) { this->seed(seed); }
  void seed(uint32_t s) {
    x_[0] = s;
    for (uint32_t i = 1; i < N; ++i) x_[i] = 1812433253u * (x_[i - 1] ^ (x_[i - 1] >> 30)) + i;
    p_ = N;
  }
  uint32_t operator()() {
    if (p_ >= N) refill();
    uint32_t y = x_[p_++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
  void prefetch() const {
    __builtin_prefetch(&p_);
    __builtin_prefetch(&x_[p_ < N ? p_ : 0]);
  }

 private:
  static constexpr uint32_t N = 624, M = 397;
  uint32_t p_ = N;
  uint32_t x_[N];
  void refill() {
    for (uint32_t i = 0; i < N; ++i) {
      const uint32_t y = (x_[i] & 0x80000000u) | (x_[(i + 1) % N] & 0x7fffffffu);
      x_[i] = x_[(i + M) % N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    p_ = 0;
  }
};

// What a replay's board on the device still owes: reset + handicap, then `count` consecutive moves of
// `rec` from index `from`.
struct PendingOp {
  bool reset = false;
  int handi = 0;
  const SgfRecord* rec = nullptr;
  int from = 0, count = 0;
  bool empty() const { return !reset && count == 0; }
};

// One replay: a record, a position in it and a board on the device.
class OfflineLoader {
 public:
  OfflineLoader(const OfflineOptions& opt, SgfCorpus* corpus, DeviceEnv* env, int board, uint32_t seed)
      : opt_(opt), corpus_(corpus), env_(env), board_(board), rng_(seed) {
    reload();  // the reference reloads in the constructor (offpolicy_loader.cc:18-21)
  }

  int board() const { return board_; }
  int ply() const { return ply_; }
  bool just_started() const { return ply_ == 1; }  // GoState::JustStarted, go/go_state.h:55
  int curr_game() const { return curr_game_; }
  int winner_label() const {                       // offpolicy_loader.cc:132-133
    const int w = rec_->winner;
    return w == 1 ? 1 : (w == 2 ? -1 : 0);
  }
  // aug code of the next sample (offpolicy_loader.cc:135-137); draws from the replay's RNG
  int draw_aug_code() {
    int code = opt_.data_aug;
    if (code == -1 || code >= 8) code = (int)(rng_() % 8);
    return code;
  }
  // the next k record moves as actions in the augmented frame (save_forward_moves, :154-173)
  void future_actions(int aug, int64_t* out) const {
    for (int i = 0; i < opt_.num_future_actions; ++i) {
      const int at = idx_ + i;
      const int mv = at < (int)rec_->moves.size() ? rec_->moves[at] : -3;
      int a = mv >= 0 ? d4_table()[(aug & 7) * GOENV_NUM_POINTS + mv] : -1;
      if (a < 0 || a >= GOENV_NUM_POINTS) a = 0;  // "invalid action!" clamp, :162-169
      out[i] = a;
    }
  }

  // Coord2Action of every (D4 code, point) (go/board_feature.h:62-65), tabulated once
  static const int16_t* d4_table() {
    static const std::vector<int16_t> t = [] {
      std::vector<int16_t> v(8 * GOENV_NUM_POINTS);
      for (int c = 0; c < 8; ++c)
        for (int p = 0; p < GOENV_NUM_POINTS; ++p) v[c * GOENV_NUM_POINTS + p] = (int16_t)goenv_aug_action(c, p);
      return v;
    }();
    return t.data();
  }

  // OfflineLoader::next (offpolicy_loader.cc:120-126): host bookkeeping only
  void next() {
    if (need_reload()) reload();
    finish_forward(forward());
  }

  // the lines the next sample of this replay will touch (called when the replay is chosen, a few
  // microseconds before it is used)
  void prefetch() const {
    __builtin_prefetch(this);
    __builtin_prefetch(reinterpret_cast<const char*>(this) + 64);
    rng_.prefetch();
  }

  // what the board owes since it was last brought up to date; the caller sends it and clears it
  const PendingOp& pending() const { return pend_; }
  void clear_pending() { pend_ = PendingOp(); }

 private:
  OfflineOptions opt_;
  SgfCorpus* corpus_;
  DeviceEnv* env_;
  int board_;
  const SgfRecord* rec_ = nullptr;
  int idx_ = 0;          // Sgf::iterator::_move_idx
  int curr_game_ = 0;
  int game_loaded_ = 0;
  int ply_ = 1;
  PendingOp pend_;
  Mt19937 rng_;          // last: the fields above share two cache lines (prefetch())

  // s().forward(curr().GetCoord()): legality is the record's (legal_prefix); an accepted move is owed
  bool forward() {
    if (idx_ >= (int)rec_->moves.size()) {   // past the last node the iterator yields a pass: always legal
      throw std::range_error("offline loader: forward past the end of a record (num_future_actions = 0?)");
    }
    if (idx_ >= rec_->legal_prefix) return false;
    owe(1);
    return true;
  }
  void owe(int n) {
    if (n <= 0) return;
    if (pend_.count == 0) { pend_.rec = rec_; pend_.from = idx_; }
    pend_.count += n;
  }

  bool need_reload() const {  // offpolicy_loader.cc:115-118
    const bool done = idx_ >= (int)rec_->moves.size();
    const int step_left = rec_->num_moves - idx_ - 1;
    return done || step_left < opt_.num_future_actions || (opt_.move_cutoff >= 0 && idx_ >= opt_.move_cutoff);
  }

  void finish_forward(bool legal) {
    if (!legal) reload();
    else { ++idx_; ++ply_; }
  }

  // `count` calls of next() in a row (the pre-move loop of after_reload, :101).  Runs of moves
  // that need no reload check are sent to the device as one replay launch.
  void advance(int count) {
    while (count > 0) {
      if (need_reload()) {
        reload();
        finish_forward(forward());
        --count;
        continue;
      }
      int run = 0;
      while (run < count) {  // how many next() calls pass the reload test on host data alone
        const int at = idx_ + run;
        const bool done = at >= (int)rec_->moves.size();
        const int left = rec_->num_moves - at - 1;
        if (done || left < opt_.num_future_actions || (opt_.move_cutoff >= 0 && at >= opt_.move_cutoff)) break;
        ++run;
      }
      const int applied = std::max(0, std::min(run, rec_->legal_prefix - idx_));
      owe(applied);
      idx_ += applied;
      ply_ += applied;
      count -= applied;
      if (applied < run) {  // that forward() failed: reload, the call is spent
        reload();
        --count;
      }
    }
  }

  // ReplayLoaderT::Reload + after_reload (elf/replay_loader.h:28-35, offpolicy_loader.cc:70-108)
  void reload() {
    while (true) {
      curr_game_ = (int)(rng_() % corpus_->size());  // get_key, :110-113
      rec_ = &corpus_->get((size_t)curr_game_);
      idx_ = 0;
      if (rec_->num_moves < 10 || rec_->board_size != GOENV_BOARD_SIZE) continue;
      if (need_reload()) continue;
      if (rec_->legal_prefix < 0) rec_->legal_prefix = env_->count_legal_prefix(rec_->handi, rec_->moves);
      pend_ = PendingOp();          // whatever the board owed is void: it starts over
      pend_.reset = true;
      pend_.handi = rec_->handi;
      pend_.rec = rec_;
      ply_ = 1;
      ++game_loaded_;
      const float ratio = game_loaded_ == 1 ? opt_.start_ratio_pre_moves : opt_.ratio_pre_moves;
      const float scaled = (float)rec_->num_moves * ratio;   // int * float, as in :97
      int base = (int)((double)scaled + 0.5);
      if (base == 0) base = 1;
      const int pre_moves = (int)(rng_() % (unsigned long)base);
      advance(pre_moves);
      return;
    }
  }
};

}  // namespace gogame
