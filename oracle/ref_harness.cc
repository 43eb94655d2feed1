// TEST INFRASTRUCTURE -- not part of the product path.
//
// Thin extern "C" harness around the UNMODIFIED reference rules engine.  It is
// compiled by oracle/Makefile together with /root/reference/go/{board,common,
// board_feature}.cc *where they lie* (no reference source is copied into this
// repo) into oracle/_ref/libgoref.so.  It only exists where /root/reference is
// mounted; the prebuilt .so travels to the GPU box with the tree.
//
// Users: tests/ (validates oracle/go_oracle.c and the HIP path against the real
// reference), tests/golden/make_golden.py (fixture generation) and bench.py's
// cpu_baseline leg (kind "reference").  Nothing under elf_amd/ may load it.
//
// Everything here is our own code; it calls the reference's public C API
// declared in go/board.h:241-370 and the BoardFeature class of
// go/board_feature.h:30-102.

#include <stdint.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "board.h"          // -I/root/reference/go
#include "board_feature.h"  // -I/root/reference/go

namespace {

inline int coord_to_action(Coord c) {
  if (c == M_PASS) return -1;
  if (c == M_RESIGN) return -2;
  return EXPORT_OFFSET(c);
}

inline Coord action_to_coord(int a) {
  if (a == -1) return M_PASS;
  if (a == -2) return M_RESIGN;
  return OFFSETXY(EXPORT_X(a), EXPORT_Y(a));
}

// counter-based RNG shared by the CPU oracle and the HIP path (SURVEY 8d cfg 2)
inline uint64_t mix64(uint64_t z) {
  z ^= z >> 30;
  z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27;
  z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return z;
}
inline uint64_t policy_rand(uint64_t seed, uint64_t board_id, uint64_t game, uint64_t ply) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (board_id + 1);
  z = mix64(z ^ (game * 0xD1B54A32D192ED03ull));
  return mix64(z + ply * 0x9E3779B97F4A7C15ull);
}

}  // namespace

extern "C" {

int goref_sizeof_board() { return (int)sizeof(Board); }

void goref_clear(void* b) { ClearBoard((Board*)b); }

void goref_copy(void* dst, const void* src) { CopyBoard((Board*)dst, (const Board*)src); }

// GoState::forward (go/go_state.cc:73-78).  Returns 1 if legal (applied), 0 if not.
// *game_end receives Play()'s return value (IsGameEnd), which GoState drops.
int goref_forward(void* b, int action, int* game_end) {
  Board* board = (Board*)b;
  GroupId4 ids;
  if (!TryPlay2(board, action_to_coord(action), &ids)) return 0;
  bool end = Play(board, &ids);
  if (game_end) *game_end = end ? 1 : 0;
  return 1;
}

int goref_check_move(const void* b, int action) {
  GroupId4 ids;
  return TryPlay2((const Board*)b, action_to_coord(action), &ids) ? 1 : 0;
}

int goref_place_handicap(void* b, int x, int y, int player) {
  return PlaceHandicap((Board*)b, x, y, (Stone)player) ? 1 : 0;
}

int goref_is_game_end(const void* b) { return IsGameEnd((const Board*)b) ? 1 : 0; }

// Canonical view.
void goref_canonical(const void* b, uint8_t* color, int16_t* libs, int16_t* stones,
                     uint16_t* last_placed, int32_t* sc) {
  const Board* board = (const Board*)b;
  for (int x = 0; x < BOARD_SIZE; ++x) {
    for (int y = 0; y < BOARD_SIZE; ++y) {
      Coord c = OFFSETXY(x, y);
      int p = EXPORT_OFFSET_XY(x, y);
      const Info& info = board->_infos[c];
      bool has = HAS_STONE(info.color);
      if (color) color[p] = info.color;
      if (libs) libs[p] = has ? board->_groups[info.id].liberties : 0;
      if (stones) stones[p] = has ? board->_groups[info.id].stones : 0;
      if (last_placed) last_placed[p] = has ? info.last_placed : 0;
    }
  }
  if (sc) {
    memset(sc, 0, 16 * sizeof(int32_t));
    sc[0] = board->_ply;
    sc[1] = board->_next_player;
    sc[2] = board->_b_cap;
    sc[3] = board->_w_cap;
    sc[4] = coord_to_action(board->_last_move);
    sc[5] = coord_to_action(board->_last_move2);
    sc[6] = coord_to_action(board->_last_move3);
    sc[7] = coord_to_action(board->_last_move4);
    sc[8] = board->_ko_age;
    sc[9] = board->_simple_ko == 0 ? -1 : coord_to_action(board->_simple_ko);
    sc[10] = board->_simple_ko_color;
  }
}

int goref_num_groups(const void* b) { return ((const Board*)b)->_num_groups; }

// FindAllValidMoves for the side to move -> u8[361] mask in export order.
int goref_legal(const void* b, uint8_t* mask) {
  const Board* board = (const Board*)b;
  AllMoves am;
  FindAllValidMoves(board, board->_next_player, &am);
  memset(mask, 0, NUM_INTERSECTION);
  for (int i = 0; i < am.num_moves; ++i) mask[EXPORT_OFFSET(am.moves[i])] = 1;
  return am.num_moves;
}

// cfg-2 candidate set: legal moves that do not fill an own true eye
// (FindAllValidMoves minus IsTrueEye, go/board.cc:763-779,1550-1552).
int goref_candidates(const void* b, uint8_t* mask) {
  const Board* board = (const Board*)b;
  AllMoves am;
  FindAllValidMoves(board, board->_next_player, &am);
  memset(mask, 0, NUM_INTERSECTION);
  int n = 0;
  for (int i = 0; i < am.num_moves; ++i) {
    if (IsTrueEye(board, am.moves[i], board->_next_player)) continue;
    mask[EXPORT_OFFSET(am.moves[i])] = 1;
    ++n;
  }
  return n;
}

// BoardFeature::Extract with a D4 code decoded as GoState::extractor(code) does
// (go/go_state.h:81-86).
void goref_extract(const void* b, int aug_code, float* out) {
  const Board* board = (const Board*)b;
  BoardFeature bf(*board, (BoardFeature::Rot)(aug_code % 4), (aug_code >> 2) == 1);
  std::vector<float> f;
  bf.Extract(&f);
  memcpy(out, f.data(), sizeof(float) * MAX_NUM_FEATURE * NUM_INTERSECTION);
}

int goref_coord2action(int aug_code, int action) {
  Board board;
  ClearBoard(&board);
  BoardFeature bf(board, (BoardFeature::Rot)(aug_code % 4), (aug_code >> 2) == 1);
  if (action < 0) return (int)bf.Coord2Action(action_to_coord(action));
  return (int)bf.Coord2Action(action_to_coord(action));
}

int goref_action2coord(int aug_code, int action) {
  Board board;
  ClearBoard(&board);
  BoardFeature bf(board, (BoardFeature::Rot)(aug_code % 4), (aug_code >> 2) == 1);
  return coord_to_action(bf.Action2Coord(action));
}

float goref_tt_score(const void* b) { return GetTrompTaylorScore((const Board*)b, NULL, NULL); }
float goref_fast_score(const void* b, int rule) { return GetFastScore((const Board*)b, rule); }

// Heuristic move filters of go/board.cc for the side to move, one value per point in export
// order.  All five come straight from the reference's functions:
//   self_atari[p] = num_stones of IsSelfAtari(board, NULL, c, player, &n) (:168-200), 0 if not
//   gives_ko[p]   = IsMoveGivingSimpleKo (:320-335) of a legal move (ids from TryPlay2)
//   cand[p]       = FindAllCandidateMoves(board, player, thres) (:685-715)
//   libs[p][2]    = find_only_liberty (:1042-1058) / find_two_liberties (:1060-1081) of the group
//                   of the stone at p, sorted ascending (the reference's order follows its stone
//                   list), -1 = none
//   ladder[p]     = CheckLadder (:348-391) of a legal move
void goref_tactics(const void* b, int thres, int16_t* self_atari, uint8_t* gives_ko, uint8_t* cand,
                   int16_t* libs, int16_t* ladder) {
  const Board* board = (const Board*)b;
  const Stone player = board->_next_player;
  memset(cand, 0, NUM_INTERSECTION);
  AllMoves am;
  FindAllCandidateMoves(board, player, thres, &am);
  for (int i = 0; i < am.num_moves; ++i) cand[EXPORT_OFFSET(am.moves[i])] = 1;
  for (int x = 0; x < BOARD_SIZE; ++x) {
    for (int y = 0; y < BOARD_SIZE; ++y) {
      const Coord c = OFFSETXY(x, y);
      const int p = EXPORT_OFFSET_XY(x, y);
      self_atari[p] = 0; gives_ko[p] = 0; ladder[p] = 0;
      libs[2 * p] = libs[2 * p + 1] = -1;
      if (HAS_STONE(board->_infos[c].color)) {
        const short id = board->_infos[c].id;
        Coord m[2];
        if (board->_groups[id].liberties == 1 && find_only_liberty(board, id, &m[0])) {
          libs[2 * p] = (int16_t)EXPORT_OFFSET(m[0]);
        } else if (board->_groups[id].liberties == 2 && find_two_liberties(board, id, m)) {
          int a0 = EXPORT_OFFSET(m[0]), a1 = EXPORT_OFFSET(m[1]);
          libs[2 * p] = (int16_t)(a0 < a1 ? a0 : a1);
          libs[2 * p + 1] = (int16_t)(a0 < a1 ? a1 : a0);
        }
        continue;
      }
      GroupId4 ids;
      if (!TryPlay2(board, c, &ids)) continue;
      int n = 0;
      if (IsSelfAtari(board, NULL, c, player, &n)) self_atari[p] = (int16_t)n;
      gives_ko[p] = IsMoveGivingSimpleKo(board, &ids, player) ? 1 : 0;
      ladder[p] = (int16_t)CheckLadder(board, &ids, player);
    }
  }
}

// One ply of the cfg-2 policy on the reference board.  Returns the action played.
int goref_policy_move(const void* b, uint64_t seed, uint64_t board_id, uint64_t game) {
  const Board* board = (const Board*)b;
  uint8_t mask[NUM_INTERSECTION];
  int n = goref_candidates(b, mask);
  if (n == 0) return -1;
  uint64_t u = policy_rand(seed, board_id, game, (uint64_t)board->_ply);
  uint32_t idx = (uint32_t)(((u >> 32) * (uint64_t)n) >> 32);
  for (int p = 0; p < NUM_INTERSECTION; ++p) {
    if (mask[p]) {
      if (idx == 0) return p;
      --idx;
    }
  }
  return -1;
}

// CPU baseline: the reference's own TryPlay/Play/FindAllValidMoves/Extract driven
// by the cfg-2 policy, `threads` worker threads over `n_boards` boards (the
// reference runs one thread per game, elf/comm_template.h:336; a pool of
// hardware threads is the favourable version of that).  Runs until
// `budget_sec` of wall time has elapsed (checked between games) or every board
// has finished `games_per_board` games.  Returns plies applied; *seconds = wall.
int64_t goref_bench_playouts(int n_boards, int threads, uint64_t seed, int max_ply,
                             int with_features, int games_per_board, double budget_sec,
                             double* seconds, double* checksum) {
  std::atomic<int64_t> total_plies(0);
  std::atomic<int> next_board(0);
  std::vector<double> sums(threads, 0.0);
  auto t0 = std::chrono::steady_clock::now();
  auto worker = [&](int tid) {
    std::vector<float> feat;
    Board board;
    double local = 0.0;
    int64_t plies = 0;
    for (;;) {
      int bid = next_board.fetch_add(1);
      if (bid >= n_boards) break;
      for (int g = 0; g < games_per_board; ++g) {
        double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (budget_sec > 0 && el > budget_sec) break;
        ClearBoard(&board);
        BoardFeature bf(board);
        for (;;) {
          if (with_features) {
            bf.Extract(&feat);
            local += feat[7 * NUM_INTERSECTION + (plies % NUM_INTERSECTION)];
          }
          int a = goref_policy_move(&board, seed, (uint64_t)bid, (uint64_t)g);
          GroupId4 ids;
          if (!TryPlay2(&board, action_to_coord(a), &ids)) break;  // cannot happen
          bool end = Play(&board, &ids);
          ++plies;
          if (end) break;
          if (max_ply > 0 && board._ply > max_ply) break;
        }
      }
    }
    sums[tid] = local;
    total_plies += plies;
  };
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) pool.emplace_back(worker, t);
  for (auto& th : pool) th.join();
  double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (seconds) *seconds = el;
  double cs = 0;
  for (double s : sums) cs += s;
  if (checksum) *checksum = cs;
  return total_plies.load();
}

// Batch checker for the GPU-vs-reference tests (tests/test_gpu_vs_reference.py): `n` reference boards in
// one array (`stride` bytes apart).  For board b: if actions[b] != -3 play it through TryPlay2 + Play
// (GoState::forward) and require legality == want_legal[b] (or legal when want_legal is NULL); then compare the
// reference's canonical view, FindAllValidMoves mask and all 25 planes under aug[b] with the arrays
// the GPU produced (any of them may be NULL).  Returns the number of boards with a difference;
// first_bad[0] = lowest such board, first_bad[1] = what differed (1 legality, 2 colours, 3 liberties,
// 4 group sizes, 5 last_placed, 6 scalars, 7 legal mask, 8 planes), first_bad[2] = index in that array.
int goref_batch_check(void* boards, int stride, int n, const int32_t* actions, const int32_t* want_legal,
                      const uint8_t* color, const int16_t* libs, const int16_t* stones,
                      const uint16_t* lastp, const int32_t* sc, const uint8_t* legal, const float* planes,
                      const int32_t* aug, int threads, int32_t* first_bad) {
  if (threads < 1) threads = 1;
  std::atomic<int> next(0), bad(0);
  std::atomic<int> lowest(n);
  std::vector<int32_t> why(n, 0), where(n, 0);
  auto worker = [&]() {
    std::vector<float> f;
    uint8_t c[NUM_INTERSECTION], m[NUM_INTERSECTION];
    int16_t l[NUM_INTERSECTION], st[NUM_INTERSECTION];
    uint16_t lp[NUM_INTERSECTION];
    int32_t s[16];
    for (;;) {
      int b = next.fetch_add(1);
      if (b >= n) break;
      Board* board = (Board*)((char*)boards + (size_t)b * stride);
      int w = 0, at = 0;
      if (actions && actions[b] != -3) {
        GroupId4 ids;
        bool ok = TryPlay2(board, action_to_coord(actions[b]), &ids);
        if (ok) Play(board, &ids);
        if ((int)ok != (want_legal ? want_legal[b] : 1)) w = 1;
      }
      if (!w) {
        goref_canonical(board, c, l, st, lp, s);
        auto diff = [&](const void* a, const void* g, size_t elem, size_t count) -> int {
          if (!g || !memcmp(a, g, elem * count)) return -1;
          for (size_t i = 0; i < count; ++i)
            if (memcmp((const char*)a + i * elem, (const char*)g + i * elem, elem)) return (int)i;
          return 0;
        };
        int d;
        if ((d = diff(c, color ? color + (size_t)b * NUM_INTERSECTION : 0, 1, NUM_INTERSECTION)) >= 0) { w = 2; at = d; }
        else if ((d = diff(l, libs ? libs + (size_t)b * NUM_INTERSECTION : 0, 2, NUM_INTERSECTION)) >= 0) { w = 3; at = d; }
        else if ((d = diff(st, stones ? stones + (size_t)b * NUM_INTERSECTION : 0, 2, NUM_INTERSECTION)) >= 0) { w = 4; at = d; }
        else if ((d = diff(lp, lastp ? lastp + (size_t)b * NUM_INTERSECTION : 0, 2, NUM_INTERSECTION)) >= 0) { w = 5; at = d; }
        else if ((d = diff(s, sc ? sc + (size_t)b * 16 : 0, 4, 11)) >= 0) { w = 6; at = d; }
      }
      if (!w && legal) {
        goref_legal(board, m);
        for (int p = 0; p < NUM_INTERSECTION && !w; ++p)
          if (m[p] != legal[(size_t)b * NUM_INTERSECTION + p]) { w = 7; at = p; }
      }
      if (!w && planes) {
        int code = aug ? aug[b] : 0;
        BoardFeature bf(*board, (BoardFeature::Rot)(code % 4), (code >> 2) == 1);
        bf.Extract(&f);
        const float* g = planes + (size_t)b * MAX_NUM_FEATURE * NUM_INTERSECTION;
        if (memcmp(f.data(), g, sizeof(float) * MAX_NUM_FEATURE * NUM_INTERSECTION)) {
          w = 8;
          for (int i = 0; i < MAX_NUM_FEATURE * NUM_INTERSECTION; ++i)
            if (memcmp(&f[i], &g[i], 4)) { at = i; break; }
        }
      }
      if (w) {
        why[b] = w; where[b] = at;
        bad.fetch_add(1);
        int cur = lowest.load();
        while (b < cur && !lowest.compare_exchange_weak(cur, b)) {}
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) pool.emplace_back(worker);
  for (auto& th : pool) th.join();
  if (first_bad) {
    int b = lowest.load();
    first_bad[0] = b < n ? b : -1;
    first_bad[1] = b < n ? why[b] : 0;
    first_bad[2] = b < n ? where[b] : 0;
  }
  return bad.load();
}

// Batch helpers of the same tests.  goref_batch_policy: the cfg-2 policy move of every board
// (game counter per board).  goref_batch_ended: 1 where IsGameEnd or _ply > max_ply (what the device
// restarts before the next ply under AUTO_RESTART).
void goref_batch_policy(const void* boards, int stride, int n, uint64_t seed, uint64_t id_base,
                        const int32_t* games, int32_t* out, int threads) {
  if (threads < 1) threads = 1;
  std::atomic<int> next(0);
  auto worker = [&]() {
    for (;;) {
      int b = next.fetch_add(1);
      if (b >= n) break;
      out[b] = goref_policy_move((const char*)boards + (size_t)b * stride, seed, id_base + (uint64_t)b,
                                 games ? (uint64_t)games[b] : 0);
    }
  };
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) pool.emplace_back(worker);
  for (auto& th : pool) th.join();
}

int goref_batch_ended(const void* boards, int stride, int n, int max_ply, uint8_t* out) {
  int k = 0;
  for (int b = 0; b < n; ++b) {
    const Board* board = (const Board*)((const char*)boards + (size_t)b * stride);
    out[b] = (IsGameEnd(board) || (max_ply > 0 && board->_ply > max_ply)) ? 1 : 0;
    k += out[b];
  }
  return k;
}

}  // extern "C"
