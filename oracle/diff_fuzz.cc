// TEST INFRASTRUCTURE -- not part of the product path.
//
// Native differential fuzz: the UNMODIFIED reference rules engine (through the goref_* harness,
// oracle/ref_harness*.cc over /root/reference/go/{board,board_feature,go_state}.cc) against the
// plain-C restatement (goor_*, oracle/go_oracle.c), move by move inside one C loop so that millions
// of plies take seconds.  Linked by oracle/Makefile into oracle/_ref/libgodiff.so (reference
// objects + harness + go_oracle.c); driven by tests/test_oracle_vs_ref.py only.
//
// What one ply compares (everything the HIP path is later checked against on the oracle side):
//   GoState::forward's legality + Play's game-end flag   go/go_state.cc:73-78, go/board.cc:1083-1179
//   canonical view: colours, per-stone liberties / stones / last_placed, 11 scalars
//   FindAllValidMoves mask and the candidate mask (legal minus own true eyes)   go/board.cc:763-779
// and on a share of the plies: all 9,025 plane floats under a random D4 code
// (go/board_feature.cc:175-195), GetTrompTaylorScore, GetFastScore CN/JP (go/board.cc:1560-1688),
// TryPlay2 of every point (check_move), the heuristic move filters, the cfg-2 policy move.
//
// Moves: candidate / any-legal / uniformly random point (mostly illegal) / pass / resign, with
// handicaps 0-9 on a share of the games.  A game is dropped (and counted) when the reference nears
// its MAX_GROUP limit (go/board.h:67: ids are u8, >= 172 groups corrupt it).

#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

extern "C" {
// reference side (oracle/ref_harness.cc, ref_harness_sgf.cc)
int goref_sizeof_board();
void goref_clear(void*);
int goref_forward(void*, int, int*);
int goref_check_move(const void*, int);
void goref_apply_handicap(void*, int);
void goref_canonical(const void*, uint8_t*, int16_t*, int16_t*, uint16_t*, int32_t*);
int goref_num_groups(const void*);
int goref_legal(const void*, uint8_t*);
int goref_candidates(const void*, uint8_t*);
void goref_extract(const void*, int, float*);
float goref_tt_score(const void*);
float goref_fast_score(const void*, int);
void goref_tactics(const void*, int, int16_t*, uint8_t*, uint8_t*, int16_t*, int16_t*);
int goref_policy_move(const void*, uint64_t, uint64_t, uint64_t);
// restatement side (oracle/go_oracle.c)
int goor_sizeof_board(void);
void goor_clear(void*);
int goor_forward(void*, int, int*);
int goor_check_move(const void*, int);
void goor_apply_handicap(void*, int);
void goor_canonical(const void*, uint8_t*, int16_t*, int16_t*, uint16_t*, int32_t*);
int goor_legal(const void*, uint8_t*);
int goor_candidates(const void*, uint8_t*);
void goor_extract(const void*, int, float*);
float goor_tt_score(const void*);
float goor_fast_score(const void*, int);
void goor_tactics(const void*, int, int16_t*, uint8_t*, uint8_t*, int16_t*, int16_t*);
int goor_policy_move(const void*, uint64_t, uint64_t, uint64_t);
}

namespace {

enum { NP = 361, NF = 25 * 361 };
// counts[] slots
enum { C_PLIES, C_GAMES, C_APPLIED, C_ILLEGAL, C_PASSES, C_RESIGNS, C_KO_SET, C_CAPTURES, C_EXTRACTS, C_SCORES,
       C_TACTICS, C_HANDICAP_GAMES, C_DROPPED_MAXGROUP, C_GAME_ENDS, C_CHECKMOVE_SWEEPS, C_MAX_GROUPS, C_N };

struct Rng {
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
};

struct View {
  uint8_t color[NP];
  int16_t libs[NP], stones[NP];
  uint16_t lastp[NP];
  int32_t sc[16];
};

struct Fail {
  std::mutex mu;
  bool failed = false;
  char msg[512];
  void set(const char* what, uint64_t game_seed, int ply, int action, int extra) {
    std::lock_guard<std::mutex> g(mu);
    if (failed) return;
    failed = true;
    snprintf(msg, sizeof msg, "%s differs: game seed %llu ply %d action %d (%d)", what,
             (unsigned long long)game_seed, ply, action, extra);
  }
};

int pick(const uint8_t* mask, int n, Rng& rng) {
  if (n == 0) return -1;
  int k = (int)rng.below((uint32_t)n);
  for (int p = 0; p < NP; ++p)
    if (mask[p] && k-- == 0) return p;
  return -1;
}

void run_games(uint64_t seed, int tid, int64_t target, std::atomic<int64_t>* done_plies, int64_t* counts, Fail* fail) {
  std::vector<uint8_t> rb(goref_sizeof_board() + 64), ob(goor_sizeof_board() + 64);
  void* r = rb.data();
  void* o = ob.data();
  std::vector<float> fr(NF), fo(NF);
  View vr, vo;
  uint8_t lr[NP], lo[NP], cr[NP], co[NP];
  int16_t sa_r[NP], sa_o[NP], lad_r[NP], lad_o[NP], lib_r[2 * NP], lib_o[2 * NP];
  uint8_t ko_r[NP], ko_o[NP], cd_r[NP], cd_o[NP];
  Rng top{seed * 0x9E3779B97F4A7C15ull + (uint64_t)tid * 0xD1B54A32D192ED03ull + 1};
  while (!fail->failed && done_plies->load(std::memory_order_relaxed) < target) {
    const uint64_t gseed = top.next();
    Rng rng{gseed};
    goref_clear(r);
    goor_clear(o);
    ++counts[C_GAMES];
    memset(&vr, 0, sizeof vr);
    if (rng.below(4) == 0) {
      int h = (int)rng.below(10);
      goref_apply_handicap(r, h);
      goor_apply_handicap(o, h);
      ++counts[C_HANDICAP_GAMES];
    }
    // per-game style: share of eye-filling moves, passes, random probes
    const uint32_t fill_eyes = rng.below(3);      // 0: candidates only, 1: mixed, 2: any legal move
    const uint32_t pass_pm = rng.below(40);       // passes per 1000 plies
    const uint32_t probe_pm = 20 + rng.below(80); // uniformly random points per 1000 plies
    const int cap = 500 + (int)rng.below(400);
    int64_t local = 0;
    for (int ply = 0; ply < cap; ++ply) {
      int nl = goref_legal(r, lr), nc = goref_candidates(r, cr);
      if (goor_legal(o, lo) != nl || memcmp(lr, lo, NP)) { fail->set("legal mask", gseed, ply, 0, nl); break; }
      if (goor_candidates(o, co) != nc || memcmp(cr, co, NP)) { fail->set("candidate mask", gseed, ply, 0, nc); break; }
      uint32_t u = rng.below(1000);
      int a;
      if (u < pass_pm) a = -1;
      else if (u < pass_pm + probe_pm) a = (int)rng.below(NP);
      else if (u == 999 && ply > 30 && rng.below(8) == 0) a = -2;
      else {
        bool any = fill_eyes == 2 || (fill_eyes == 1 && rng.below(4) == 0);
        a = any ? pick(lr, nl, rng) : pick(cr, nc, rng);
      }
      int er = 0, eo = 0;
      const int32_t cap_before = vr.sc[2] + vr.sc[3];
      int okr = goref_forward(r, a, &er), oko = goor_forward(o, a, &eo);
      if (okr != oko || (okr && er != eo)) { fail->set("forward()", gseed, ply, a, okr * 2 + er); break; }
      ++local;
      if (!okr) ++counts[C_ILLEGAL];
      else {
        ++counts[C_APPLIED];
        if (a == -1) ++counts[C_PASSES];
        if (a == -2) ++counts[C_RESIGNS];
      }
      goref_canonical(r, vr.color, vr.libs, vr.stones, vr.lastp, vr.sc);
      goor_canonical(o, vo.color, vo.libs, vo.stones, vo.lastp, vo.sc);
      if (memcmp(vr.color, vo.color, NP)) { fail->set("colours", gseed, ply, a, 0); break; }
      if (memcmp(vr.libs, vo.libs, sizeof vr.libs)) { fail->set("liberties", gseed, ply, a, 0); break; }
      if (memcmp(vr.stones, vo.stones, sizeof vr.stones)) { fail->set("group sizes", gseed, ply, a, 0); break; }
      if (memcmp(vr.lastp, vo.lastp, sizeof vr.lastp)) { fail->set("last_placed", gseed, ply, a, 0); break; }
      if (memcmp(vr.sc, vo.sc, 11 * sizeof(int32_t))) { fail->set("scalars", gseed, ply, a, 0); break; }
      if (okr && a >= 0) {
        if (vr.sc[8] == 0) ++counts[C_KO_SET];
        if (ply > 0 && vr.sc[2] + vr.sc[3] != cap_before) ++counts[C_CAPTURES];
      }
      const int ng = goref_num_groups(r);
      if (ng > counts[C_MAX_GROUPS]) counts[C_MAX_GROUPS] = ng;
      uint32_t v = rng.below(64);
      if (v < 16) {  // planes under a random D4 code
        int aug = (int)rng.below(8);
        goref_extract(r, aug, fr.data());
        goor_extract(o, aug, fo.data());
        if (memcmp(fr.data(), fo.data(), NF * sizeof(float))) { fail->set("planes", gseed, ply, a, aug); break; }
        ++counts[C_EXTRACTS];
      }
      if (v >= 16 && v < 24) {
        if (goref_tt_score(r) != goor_tt_score(o)) { fail->set("Tromp-Taylor", gseed, ply, a, 0); break; }
        if (goref_fast_score(r, 0) != goor_fast_score(o, 0)) { fail->set("FastScore CN", gseed, ply, a, 0); break; }
        if (goref_fast_score(r, 1) != goor_fast_score(o, 1)) { fail->set("FastScore JP", gseed, ply, a, 0); break; }
        ++counts[C_SCORES];
      }
      if (v == 24) {  // TryPlay2 of every point, pass and resign
        bool bad = false;
        for (int p = -2; p < NP && !bad; ++p) bad = goref_check_move(r, p) != goor_check_move(o, p);
        if (bad) { fail->set("check_move sweep", gseed, ply, a, 0); break; }
        uint64_t ps = rng.next();
        if (goref_policy_move(r, ps, (uint64_t)ply, 3) != goor_policy_move(o, ps, (uint64_t)ply, 3)) {
          fail->set("policy move", gseed, ply, a, 0); break;
        }
        ++counts[C_CHECKMOVE_SWEEPS];
      }
      if (v == 25 && rng.below(4) == 0) {  // heuristic filters (ladders make them the slow part)
        int thres = 1 + (int)rng.below(5);
        goref_tactics(r, thres, sa_r, ko_r, cd_r, lib_r, lad_r);
        goor_tactics(o, thres, sa_o, ko_o, cd_o, lib_o, lad_o);
        if (memcmp(sa_r, sa_o, sizeof sa_r) || memcmp(ko_r, ko_o, NP) || memcmp(cd_r, cd_o, NP) ||
            memcmp(lib_r, lib_o, sizeof lib_r) || memcmp(lad_r, lad_o, sizeof lad_r)) {
          fail->set("move filters", gseed, ply, a, thres); break;
        }
        ++counts[C_TACTICS];
      }
      if (okr && er && ply > 5) { ++counts[C_GAME_ENDS]; break; }
      if (ng >= 165) { ++counts[C_DROPPED_MAXGROUP]; break; }
    }
    counts[C_PLIES] += local;
    done_plies->fetch_add(local, std::memory_order_relaxed);
  }
}

}  // namespace

extern "C" {

int godiff_num_counts() { return C_N; }

// Runs games on `threads` host threads until at least `target_plies` forward() calls have been
// compared (or a difference is found).  counts_out[C_N] receives the totals (C_MAX_GROUPS is a
// maximum).  Returns 0 when every comparison was equal, 1 otherwise with the first difference in err.
int godiff_run(uint64_t seed, int64_t target_plies, int threads, int64_t* counts_out, char* err, int errlen) {
  if (threads < 1) threads = 1;
  std::vector<std::vector<int64_t>> counts(threads, std::vector<int64_t>(C_N, 0));
  std::atomic<int64_t> done(0);
  Fail fail;
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back(run_games, seed, t, target_plies, &done, counts[t].data(), &fail);
  for (auto& th : pool) th.join();
  for (int i = 0; i < C_N; ++i) {
    int64_t v = 0;
    for (int t = 0; t < threads; ++t) v = i == C_MAX_GROUPS ? (counts[t][i] > v ? counts[t][i] : v) : v + counts[t][i];
    counts_out[i] = v;
  }
  if (fail.failed && err && errlen > 0) snprintf(err, (size_t)errlen, "%s", fail.msg);
  return fail.failed ? 1 : 0;
}

}  // extern "C"
