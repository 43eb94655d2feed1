/*
 * TEST INFRASTRUCTURE -- not part of the product path.
 *
 * go_oracle.c: plain-C CPU restatement of the DarkForest Go hot path of
 * facebookresearch/ELF (rules update, legal moves, 25 feature planes, scoring).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * it, and only as the checker -- nothing under elf_amd/ links or loads it.
 *
 * It restates WHAT go/board.cc and go/board_feature.cc compute, not HOW: the
 * reference keeps incremental group tables and linked lists, this file
 * recomputes groups and liberties by flood fill from the stone colours.  That
 * is legal because the reference's stored liberties/stones always equal the
 * from-scratch values (its own VerifyBoard invariant, go/board.cc:1455-1462).
 *
 * Parity pinning: checked against (1) the real reference compiled from
 * /root/reference (oracle/_ref/libgoref.so, differential fuzz in
 * tests/test_oracle_vs_ref.py), (2) the committed golden vectors in
 * tests/golden/ that were generated from that build (sample1.sgf replay --
 * the reference's only Go fixture, go/sample1.sgf -- random playouts, D4
 * codes, handicaps, hand-built edge cases).
 *
 * Point index p = x*19 + y everywhere (the reference's export order,
 * go/board.h:155); actions: p, -1 = pass (M_PASS), -2 = resign (M_RESIGN).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define BS 19
#define NP 361
#define S_EMPTY 0
#define S_BLACK 1
#define S_WHITE 2
#define OPP(c) (3 - (c))

typedef struct {
  uint8_t color[NP];
  uint16_t last_placed[NP]; /* go/board.h:46; never cleared on capture (board.cc:406-408) */
  int32_t ply;              /* starts at 1, go/board.cc:44 */
  int32_t next;             /* go/board.cc:37 */
  int32_t b_cap, w_cap;     /* stones captured BY black / white, board.cc:1132-1133 */
  int32_t last[4];          /* _last_move.._last_move4 as actions */
  int32_t ko_age;           /* u16 arithmetic, go/board.h:113 */
  int32_t ko;               /* _simple_ko as action, -1 = never set */
  int32_t ko_color;
} GoBoard;

/* groups of one position, recomputed from colours */
typedef struct {
  int16_t gid[NP];    /* group index per point, -1 for empty */
  int16_t glib[NP];   /* liberties per group index */
  int16_t gstones[NP];
  int ngroups;
} Groups;

static inline int nbrs(int p, int out[4]) {
  /* order L,T,R,B of go/board.h:183 (x-1, y-1, x+1, y+1); only valid ones */
  int x = p / BS, y = p % BS, n = 0;
  if (x > 0) out[n++] = p - BS;
  if (y > 0) out[n++] = p - 1;
  if (x < BS - 1) out[n++] = p + BS;
  if (y < BS - 1) out[n++] = p + 1;
  return n;
}

static void analyze(const GoBoard* b, Groups* g) {
  int stack[NP];
  int16_t libmark[NP];
  memset(g->gid, 0xff, sizeof(g->gid));
  memset(libmark, 0xff, sizeof(libmark));
  g->ngroups = 0;
  for (int s = 0; s < NP; ++s) {
    if (b->color[s] == S_EMPTY || g->gid[s] >= 0) continue;
    int id = g->ngroups++;
    int sp = 0, stones = 0, libs = 0;
    stack[sp++] = s;
    g->gid[s] = (int16_t)id;
    while (sp) {
      int p = stack[--sp];
      ++stones;
      int nb[4], n = nbrs(p, nb);
      for (int i = 0; i < n; ++i) {
        int q = nb[i];
        if (b->color[q] == S_EMPTY) {
          if (libmark[q] != id) { libmark[q] = (int16_t)id; ++libs; }
        } else if (b->color[q] == b->color[s] && g->gid[q] < 0) {
          g->gid[q] = (int16_t)id;
          stack[sp++] = q;
        }
      }
    }
    g->glib[id] = (int16_t)libs;
    g->gstones[id] = (int16_t)stones;
  }
}

void goor_clear(GoBoard* b) {
  /* ClearBoard, go/board.cc:30-46 */
  memset(b, 0, sizeof(*b));
  b->ply = 1;
  b->next = S_BLACK;
  for (int i = 0; i < 4; ++i) b->last[i] = -1;
  b->ko = -1;
}

int goor_sizeof_board(void) { return (int)sizeof(GoBoard); }

/* IsSimpleKoViolation go/board.cc:156-161 + IsSuicideMove :131-154 given groups */
static int legal_at(const GoBoard* b, const Groups* g, int p, int player) {
  if (b->color[p] != S_EMPTY) return 0;
  if (b->ko == p && b->ko_age == 0 && b->ko_color == player) return 0;
  int nb[4], n = nbrs(p, nb);
  for (int i = 0; i < n; ++i) {
    int q = nb[i];
    if (b->color[q] == S_EMPTY) return 1;
    int lib = g->glib[g->gid[q]];
    if (b->color[q] == player) { if (lib > 1) return 1; }
    else { if (lib == 1) return 1; }
  }
  return 0;
}

static void shift_last(GoBoard* b, int action, int player) {
  /* update_next_move go/board.cc:1014-1027 */
  b->next = OPP(player);
  b->last[3] = b->last[2];
  b->last[2] = b->last[1];
  b->last[1] = b->last[0];
  b->last[0] = action;
  b->ply++;
}

int goor_is_game_end(const GoBoard* b) {
  /* IsGameEnd go/board.cc:1690-1692 */
  return b->ply > 1 && ((b->last[0] == -1 && b->last[1] == -1) || b->last[0] == -2);
}

/* TryPlay (go/board.cc:626-662) + Play (:1083-1179) for an explicit player. */
static int play_as(GoBoard* b, int action, int player, int* game_end) {
  if (action == -1 || action == -2) {
    /* pass / resign: Play returns early, board.cc:1092-1095 -- the ko is NOT aged */
    shift_last(b, action, player);
    if (game_end) *game_end = goor_is_game_end(b);
    return 1;
  }
  if (action < 0 || action >= NP) return 0;
  Groups g;
  analyze(b, &g);
  if (!legal_at(b, &g, action, player)) return 0;

  int nb[4], n = nbrs(action, nb);
  int merged = 0, total_capture = 0, capture_p = -1;
  for (int i = 0; i < n; ++i)
    if (b->color[nb[i]] == player) merged = 1;
  /* enemy neighbour groups whose only liberty is `action` die, board.cc:1129-1148 */
  for (int i = 0; i < n; ++i) {
    int q = nb[i];
    if (b->color[q] != OPP(player)) continue; /* also skips stones removed by an earlier i */
    int id = g.gid[q];
    if (g.glib[id] != 1) continue;
    for (int s = 0; s < NP; ++s)
      if (g.gid[s] == id && b->color[s] != S_EMPTY) {
        b->color[s] = S_EMPTY;
        ++total_capture;
        capture_p = s; /* only used when total_capture == 1 */
      }
  }
  if (player == S_BLACK) b->b_cap += total_capture; else b->w_cap += total_capture;
  b->color[action] = (uint8_t)player;
  b->last_placed[action] = (uint16_t)b->ply; /* board.cc:532,1157 */

  /* simple ko, board.cc:1163-1171: new group = 1 stone, 1 liberty, exactly 1 stone captured */
  int newlib = 0;
  if (!merged) {
    n = nbrs(action, nb);
    for (int i = 0; i < n; ++i) if (b->color[nb[i]] == S_EMPTY) ++newlib;
  }
  if (!merged && newlib == 1 && total_capture == 1) {
    b->ko = capture_p;
    b->ko_color = OPP(player);
    b->ko_age = 0;
  } else {
    b->ko_age = (b->ko_age + 1) & 0xffff;
  }
  shift_last(b, action, player);
  if (game_end) *game_end = 0; /* Play returns false for board moves, board.cc:1178 */
  return 1;
}

/* GoState::forward go/go_state.cc:73-78 (TryPlay2 always plays _next_player, board.cc:622-624) */
int goor_forward(GoBoard* b, int action, int* game_end) { return play_as(b, action, b->next, game_end); }

int goor_check_move(const GoBoard* b, int action) {
  if (action == -1 || action == -2) return 1;
  if (action < 0 || action >= NP) return 0;
  Groups g;
  analyze(b, &g);
  return legal_at(b, &g, action, b->next);
}

/* PlaceHandicap go/board.cc:48-64 */
int goor_place_handicap(GoBoard* b, int x, int y, int player) {
  if (b->ply > 1) return 0;
  if (x < 0 || x >= BS || y < 0 || y >= BS) return 0;
  if (!play_as(b, x * BS + y, player, NULL)) return 0;
  b->ply = 1;
  for (int i = 0; i < 4; ++i) b->last[i] = -1;
  return 1;
}

/* HandicapTable go/go_state.cc:32-70.  Effective stones per count ("*n" copies the
 * n-stone table; the "*n" token itself decodes off-board and is rejected by
 * TryPlay, go/go_state.cc:49-56 + board.cc:641). */
static const int8_t HANDI_XY[9][2] = {
    {3, 3}, {15, 15}, {3, 15}, {15, 3}, /* D4 Q16 D16 Q4 */
    {3, 9}, {15, 9},                    /* D10 Q10 */
    {9, 15}, {9, 3},                    /* K16 K4 */
    {9, 9}};                            /* K10 */
static const int8_t HANDI_SETS[10][10] = {
    {-1}, {-1},
    {0, 1, -1},                     /* 2: D4 Q16 */
    {0, 1, 3, -1},                  /* 3: D4 Q16 Q4 */
    {0, 1, 2, 3, -1},               /* 4: D4 Q16 D16 Q4 */
    {0, 1, 2, 3, 8, -1},            /* 5: *4 K10 */
    {0, 1, 2, 3, 4, 5, -1},         /* 6: *4 D10 Q10 */
    {0, 1, 2, 3, 4, 5, 8, -1},      /* 7: *4 D10 Q10 K10 */
    {0, 1, 2, 3, 4, 5, 6, 7, -1},   /* 8: *4 D10 Q10 K16 K4 */
    {0, 1, 2, 3, 4, 5, 6, 7, 8, -1} /* 9: *8 K10 */
};

void goor_apply_handicap(GoBoard* b, int handi) {
  if (handi < 2 || handi > 9) return; /* absent from the table: nothing placed */
  for (int i = 0; HANDI_SETS[handi][i] >= 0; ++i) {
    const int8_t* xy = HANDI_XY[HANDI_SETS[handi][i]];
    goor_place_handicap(b, xy[0], xy[1], S_BLACK);
  }
}

/* FindAllValidMoves go/board.cc:763-779 for the side to move */
int goor_legal(const GoBoard* b, uint8_t* mask) {
  Groups g;
  analyze(b, &g);
  int n = 0;
  for (int p = 0; p < NP; ++p) { mask[p] = (uint8_t)legal_at(b, &g, p, b->next); n += mask[p]; }
  return n;
}

/* IsEye / IsFakeEye / IsTrueEye go/board.cc:1501-1552 */
static int true_eye(const GoBoard* b, int p, int player) {
  if (b->color[p] != S_EMPTY) return 0;
  int x = p / BS, y = p % BS;
  int nb[4], n = nbrs(p, nb);
  for (int i = 0; i < n; ++i) if (b->color[nb[i]] != player) return 0;
  int n_opp = 0, n_bound = 0;
  for (int dx = -1; dx <= 1; dx += 2)
    for (int dy = -1; dy <= 1; dy += 2) {
      int xx = x + dx, yy = y + dy;
      if (xx < 0 || xx >= BS || yy < 0 || yy >= BS) ++n_bound;
      else if (b->color[xx * BS + yy] == OPP(player)) ++n_opp;
    }
  int fake = (n_bound > 0 && n_opp >= 1) || (n_bound == 0 && n_opp >= 2);
  return !fake;
}

/* cfg-2 candidate set: legal and not an own true eye */
int goor_candidates(const GoBoard* b, uint8_t* mask) {
  Groups g;
  analyze(b, &g);
  int n = 0;
  for (int p = 0; p < NP; ++p) {
    mask[p] = (uint8_t)(legal_at(b, &g, p, b->next) && !true_eye(b, p, b->next));
    n += mask[p];
  }
  return n;
}

void goor_canonical(const GoBoard* b, uint8_t* color, int16_t* libs, int16_t* stones,
                    uint16_t* last_placed, int32_t* sc) {
  Groups g;
  analyze(b, &g);
  for (int p = 0; p < NP; ++p) {
    int has = b->color[p] != S_EMPTY;
    if (color) color[p] = b->color[p];
    if (libs) libs[p] = has ? g.glib[g.gid[p]] : 0;
    if (stones) stones[p] = has ? g.gstones[g.gid[p]] : 0;
    if (last_placed) last_placed[p] = has ? b->last_placed[p] : 0;
  }
  if (sc) {
    memset(sc, 0, 16 * sizeof(int32_t));
    sc[0] = b->ply; sc[1] = b->next; sc[2] = b->b_cap; sc[3] = b->w_cap;
    sc[4] = b->last[0]; sc[5] = b->last[1]; sc[6] = b->last[2]; sc[7] = b->last[3];
    sc[8] = b->ko_age; sc[9] = b->ko; sc[10] = b->ko_color;
  }
}

/* ---- heuristic move filters, by copying the board and trying the move -------- */

/* IsSelfAtari(board, NULL, c, player, &n), go/board.cc:168-200: legal, and the group the stone
 * ends up in has exactly one liberty after the move (its early-outs are consequences of that). */
static int self_atari_at(const GoBoard* b, int p, int* num_stones) {
  GoBoard t = *b;
  if (!play_as(&t, p, b->next, NULL)) return 0;
  Groups g;
  analyze(&t, &g);
  if (g.glib[g.gid[p]] != 1) return 0;
  *num_stones = g.gstones[g.gid[p]];
  return 1;
}

/* IsMoveGivingSimpleKo, go/board.cc:320-335 */
static int gives_ko_at(const GoBoard* b, const Groups* g, int p, int player) {
  int nb[4], n = nbrs(p, nb), hits = 0;
  for (int i = 0; i < n; ++i) {
    int q = nb[i];
    if (b->color[q] == S_EMPTY) return 0;
    if (b->color[q] == player) return 0;
    if (g->glib[g->gid[q]] == 1 && g->gstones[g->gid[q]] == 1) ++hits;
  }
  return hits == 1;
}

static int empty_nbrs(const GoBoard* b, int p, int out[4]) {
  int nb[4], n = nbrs(p, nb), k = 0;
  for (int i = 0; i < n; ++i) if (b->color[nb[i]] == S_EMPTY) out[k++] = nb[i];
  return k;
}

#define MAX_LADDER_SEARCH 1024
/* CheckLadderUseSearch, go/board.cc:203-317 */
static int ladder_search(GoBoard* b, int victim, int* num_call, int depth) {
  ++*num_call;
  const int c = b->last[0], c2 = b->last[1];
  Groups g;
  analyze(b, &g);
  const int lib = g.glib[g.gid[c]];
  if (victim == OPP(b->next)) { /* capturer to play */
    if (lib == 1) return depth;
    if (lib >= 3) return 0;
    int esc[4], n_esc = empty_nbrs(b, c, esc);
    if (n_esc <= 1) return 0;
    int must_block = -1;
    for (int i = 0; i < 2; ++i) {
      int tmp[4];
      if (empty_nbrs(b, esc[i], tmp) == 3) { must_block = esc[i]; break; }
    }
    if (must_block < 0 && *num_call >= MAX_LADDER_SEARCH) must_block = esc[0];
    if (must_block >= 0) {
      if (play_as(b, must_block, b->next, NULL)) {
        int d = ladder_search(b, victim, num_call, depth + 1);
        if (d > 0) return d;
      }
    } else {
      GoBoard t = *b;
      if (play_as(&t, esc[0], t.next, NULL)) {
        int d = ladder_search(&t, victim, num_call, depth + 1);
        if (d > 0) return d;
      }
      if (play_as(b, esc[1], b->next, NULL)) {
        int d = ladder_search(b, victim, num_call, depth + 1);
        if (d > 0) return d;
      }
    }
  } else { /* victim to play */
    if (lib == 1) return 0;
    int esc[4];
    if (c2 < 0 || empty_nbrs(b, c2, esc) == 0) return 0; /* the reference calls error() here */
    const int flee = esc[0];
    if (play_as(b, flee, b->next, NULL)) {
      analyze(b, &g);
      const int l = g.glib[g.gid[flee]];
      if (l >= 3) return 0;
      if (l == 2) {
        int nb[4], n = nbrs(flee, nb);
        for (int i = 0; i < n; ++i)
          if (b->color[nb[i]] == OPP(victim) && g.glib[g.gid[nb[i]]] == 1) return 0;
      }
      int d = ladder_search(b, victim, num_call, depth + 1);
      if (d > 0) return d;
    }
  }
  return 0;
}

/* CheckLadder, go/board.cc:348-391, of a legal move p of the side to move */
static int check_ladder_at(const GoBoard* b, const Groups* g, int p, int player) {
  int nb[4], n = nbrs(p, nb), lib = 0;
  int seen[4], n_seen = 0, n_enemy = 0, n_self = 0, enemy_three = 0, in_atari = 0;
  for (int i = 0; i < n; ++i) {
    int q = nb[i];
    if (b->color[q] == S_EMPTY) { ++lib; continue; }
    int id = g->gid[q], dup = 0;
    for (int j = 0; j < n_seen; ++j) dup |= seen[j] == id;
    if (dup) continue;
    seen[n_seen++] = id;
    if (b->color[q] == OPP(player)) {
      if (n_enemy >= 1) enemy_three = 0;
      else if (g->glib[id] >= 3) enemy_three = 1;
      ++n_enemy;
    } else {
      if (n_self >= 1) in_atari = 0;
      else if (g->glib[id] == 1) in_atari = 1;
      ++n_self;
    }
  }
  if (lib != 2 || !enemy_three || !in_atari) return 0;
  GoBoard t = *b;
  play_as(&t, p, player, NULL);
  int num_call = 0;
  return ladder_search(&t, player, &num_call, 1);
}

/* same contract as goref_tactics (oracle/ref_harness.cc) */
void goor_tactics(const GoBoard* b, int thres, int16_t* self_atari, uint8_t* gives_ko, uint8_t* cand,
                  int16_t* libs, int16_t* ladder) {
  Groups g;
  analyze(b, &g);
  const int player = b->next;
  for (int p = 0; p < NP; ++p) {
    self_atari[p] = 0; gives_ko[p] = 0; cand[p] = 0; ladder[p] = 0;
    libs[2 * p] = libs[2 * p + 1] = -1;
    if (b->color[p] != S_EMPTY) {
      const int id = g.gid[p];
      if (g.glib[id] <= 2) { /* find_only_liberty / find_two_liberties, board.cc:1042-1081 */
        int k = 0;
        for (int q = 0; q < NP && k < 2; ++q) {
          if (b->color[q] != S_EMPTY) continue;
          int nb[4], n = nbrs(q, nb), adj = 0;
          for (int i = 0; i < n; ++i) adj |= b->color[nb[i]] != S_EMPTY && g.gid[nb[i]] == id;
          if (adj) libs[2 * p + k++] = (int16_t)q;
        }
      }
      continue;
    }
    if (!legal_at(b, &g, p, player)) continue;
    int n = 0;
    const int sa = self_atari_at(b, p, &n);
    if (sa) self_atari[p] = (int16_t)n;
    gives_ko[p] = (uint8_t)gives_ko_at(b, &g, p, player);
    ladder[p] = (int16_t)check_ladder_at(b, &g, p, player);
    /* FindAllCandidateMoves, board.cc:685-715 */
    cand[p] = (uint8_t)(!true_eye(b, p, player) && !(sa && n >= thres));
  }
}

/* stage an arbitrary position (same contract as goenv_set_state) */
void goor_set_state(GoBoard* b, const uint8_t* color, const uint16_t* last_placed, const int32_t* sc) {
  memcpy(b->color, color, NP);
  memcpy(b->last_placed, last_placed, NP * sizeof(uint16_t));
  b->ply = sc[0]; b->next = sc[1]; b->b_cap = sc[2]; b->w_cap = sc[3];
  for (int i = 0; i < 4; ++i) b->last[i] = sc[4 + i];
  b->ko_age = sc[8]; b->ko = sc[9]; b->ko_color = sc[10];
}

/* D4: BoardFeature::Transform go/board_feature.h:38-48 */
static inline int d4(int code, int x, int y) {
  int rot = code % 4, flip = (code >> 2) == 1, ox, oy;
  if (rot == 1) { ox = y; oy = BS - x - 1; }
  else if (rot == 2) { ox = BS - x - 1; oy = BS - y - 1; }
  else if (rot == 3) { ox = BS - y - 1; oy = x; }
  else { ox = x; oy = y; }
  if (flip) { int t = ox; ox = oy; oy = t; }
  return ox * BS + oy;
}

/* Coord2Action / Action2Coord go/board_feature.h:62-71 on export actions */
int goor_aug_action(int code, int action) {
  if (action < 0) return action;
  return d4(code, action / BS, action % BS);
}
int goor_aug_action_inv(int code, int action) {
  if (action < 0) return action;
  for (int p = 0; p < NP; ++p) if (d4(code, p / BS, p % BS) == action) return p;
  return -1;
}

/* DistanceTransform go/board_feature.cc:17-38 on the (already transformed) plane */
static void distance_transform(float* a) {
  for (int j = 0; j < BS; ++j) {
    for (int i = 1; i < BS; ++i) { float v = a[(i - 1) * BS + j] + 1; if (v < a[i * BS + j]) a[i * BS + j] = v; }
    for (int i = BS - 2; i >= 0; --i) { float v = a[(i + 1) * BS + j] + 1; if (v < a[i * BS + j]) a[i * BS + j] = v; }
  }
  for (int i = 0; i < BS; ++i) {
    for (int j = 1; j < BS; ++j) { float v = a[i * BS + j - 1] + 1; if (v < a[i * BS + j]) a[i * BS + j] = v; }
    for (int j = BS - 2; j >= 0; --j) { float v = a[i * BS + j + 1] + 1; if (v < a[i * BS + j]) a[i * BS + j] = v; }
  }
}

/* BoardFeature::Extract go/board_feature.cc:175-195 */
void goor_extract(const GoBoard* b, int code, float* out) {
  Groups g;
  analyze(b, &g);
  memset(out, 0, sizeof(float) * 25 * NP);
  int player = b->next, opp = OPP(player);
  for (int p = 0; p < NP; ++p) {
    int t = d4(code, p / BS, p % BS);
    int c = b->color[p];
    if (c != S_EMPTY) {
      int lib = g.glib[g.gid[p]];
      int cls = lib == 1 ? 0 : (lib == 2 ? 1 : 2);            /* GetLibertyMap3binary :83-99 */
      int base = (c == player) ? 0 : 3;
      out[(base + cls) * NP + t] = 1.0f;
      out[((c == player) ? 7 : 8) * NP + t] = 1.0f;            /* GetStones :102-111 */
      /* GetHistoryExp :135-146 -- double exp, narrowed to float */
      out[((c == player) ? 10 : 11) * NP + t] = (float)exp(((int)b->last_placed[p] - b->ply) / 10.0);
    } else {
      out[9 * NP + t] = 1.0f;
    }
    out[14 * NP + t] = (c == player) ? 0.0f : 10000.0f;        /* GetDistanceMap :148-160 */
    out[15 * NP + t] = (c == opp) ? 0.0f : 10000.0f;
  }
  /* GetSimpleKo :113-121 with GetSimpleKoLocation go/board.cc:337-344 */
  if (b->ko_age == 0 && b->ko >= 0) out[6 * NP + d4(code, b->ko / BS, b->ko % BS)] = 1.0f;
  distance_transform(out + 14 * NP);
  distance_transform(out + 15 * NP);
}

/* GetTrompTaylorScore(board, NULL, NULL) go/board.cc:1584-1688 */
float goor_tt_score(const GoBoard* b) {
  int seen[NP], queue[NP];
  int terr[3] = {0, 0, 0};
  memset(seen, 0, sizeof(seen));
  int any_stone = 0;
  for (int p = 0; p < NP; ++p) if (b->color[p] != S_EMPTY) { terr[b->color[p]]++; any_stone = 1; }
  if (!any_stone) return 0.0f; /* board.cc:1663-1666 */
  for (int s = 0; s < NP; ++s) {
    if (b->color[s] != S_EMPTY || seen[s]) continue;
    int qs = 0, qe = 0, touch = 0;
    queue[qe++] = s; seen[s] = 1;
    while (qs < qe) {
      int p = queue[qs++];
      int nb[4], n = nbrs(p, nb);
      for (int i = 0; i < n; ++i) {
        int q = nb[i];
        if (b->color[q] == S_EMPTY) { if (!seen[q]) { seen[q] = 1; queue[qe++] = q; } }
        else touch |= b->color[q];
      }
    }
    if (touch == S_BLACK || touch == S_WHITE) terr[touch] += qe;
  }
  return (float)(terr[S_BLACK] - terr[S_WHITE]);
}

/* GetFastScore go/board.cc:1560-1582 (rule 0 = Chinese, 1 = Japanese; _rollout_passes is
 * never written by the hot path and stays 0) */
float goor_fast_score(const GoBoard* b, int rule) {
  int sb = 0, sw = 0, eb = 0, ew = 0;
  for (int p = 0; p < NP; ++p) {
    if (b->color[p] == S_BLACK) ++sb;
    else if (b->color[p] == S_WHITE) ++sw;
    else if (true_eye(b, p, S_WHITE)) ++ew; /* GetEyeColor tests white first, :1554-1558 */
    else if (true_eye(b, p, S_BLACK)) ++eb;
  }
  if (rule == 1) return (float)(int16_t)(eb - ew + b->b_cap - b->w_cap);
  return (float)(int16_t)(eb + sb - ew - sw);
}

/* counter-based policy RNG shared with the HIP path (SURVEY 8d cfg 2) */
static inline uint64_t mix64(uint64_t z) {
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27; z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return z;
}
uint64_t goor_policy_rand(uint64_t seed, uint64_t board_id, uint64_t game, uint64_t ply) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (board_id + 1);
  z = mix64(z ^ (game * 0xD1B54A32D192ED03ull));
  return mix64(z + ply * 0x9E3779B97F4A7C15ull);
}

int goor_policy_move(const GoBoard* b, uint64_t seed, uint64_t board_id, uint64_t game) {
  uint8_t mask[NP];
  int n = goor_candidates(b, mask);
  if (n == 0) return -1;
  uint64_t u = goor_policy_rand(seed, board_id, game, (uint64_t)b->ply);
  uint32_t idx = (uint32_t)(((u >> 32) * (uint64_t)n) >> 32);
  for (int p = 0; p < NP; ++p) if (mask[p]) { if (idx == 0) return p; --idx; }
  return -1;
}

/* Restatement of goenv_sample_policy (an extension with no counterpart in the reference, which
 * leaves move selection to MCTS / Python): integer weights floor(pi*2^24) over the legal points,
 * r = rand % total, first point whose running sum exceeds r; all-zero weights -> cfg-2 policy.
 * pi is indexed in the frame of `aug`. */
int goor_sample_policy(const GoBoard* b, uint64_t seed, uint64_t board_id, uint64_t game, const float* pi, int aug) {
  uint8_t mask[NP];
  goor_legal(b, mask);
  uint32_t w[NP];
  uint64_t total = 0;
  for (int p = 0; p < NP; ++p) {
    w[p] = 0;
    if (mask[p]) {
      float f = pi[aug ? d4(aug, p / BS, p % BS) : p];
      if (f > 0.f) {
        float sc = f * 16777216.0f;
        w[p] = sc >= 16777216.0f ? 16777216u : (uint32_t)sc;
      }
    }
    total += w[p];
  }
  if (total == 0) return goor_policy_move(b, seed, board_id, game);
  /* r = floor(u * total / 2^64): a multiply-high instead of a 64-bit modulo (no divide on the device) */
  uint64_t u = goor_policy_rand(seed ^ 0xA5A5A5A55A5A5A5Aull, board_id, game, (uint64_t)b->ply);
  uint64_t r = (uint64_t)(((unsigned __int128)u * total) >> 64);
  uint64_t run = 0;
  for (int p = 0; p < NP; ++p) { run += w[p]; if (run > r) return p; }
  return -1;
}

/* One full random playout; returns plies applied, records moves (may be NULL). */
int goor_playout(GoBoard* b, uint64_t seed, uint64_t board_id, uint64_t game, int max_ply,
                 int16_t* moves, int max_record) {
  int n = 0;
  for (;;) {
    int a = goor_policy_move(b, seed, board_id, game);
    int end = 0;
    if (!goor_forward(b, a, &end)) return -1;
    if (moves && n < max_record) moves[n] = (int16_t)a;
    ++n;
    if (end) break;
    if (max_ply > 0 && b->ply > max_ply) break;
  }
  return n;
}

/* threaded CPU baseline ("port" kind): cfg-2 playouts, optional Extract per ply */
typedef struct {
  int tid, threads, n_boards, max_ply, with_features, games;
  uint64_t seed;
  double budget;
  struct timespec t0;
  int64_t plies;
  double sum;
} Work;

static double since(const struct timespec* t0) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)(t.tv_sec - t0->tv_sec) + 1e-9 * (double)(t.tv_nsec - t0->tv_nsec);
}

static void* bench_worker(void* arg) {
  Work* w = (Work*)arg;
  float* feat = (float*)malloc(sizeof(float) * 25 * NP);
  GoBoard b;
  for (int bid = w->tid; bid < w->n_boards; bid += w->threads) {
    for (int g = 0; g < w->games; ++g) {
      if (w->budget > 0 && since(&w->t0) > w->budget) goto out;
      goor_clear(&b);
      for (;;) {
        if (w->with_features) { goor_extract(&b, 0, feat); w->sum += feat[7 * NP + (w->plies % NP)]; }
        int a = goor_policy_move(&b, w->seed, (uint64_t)bid, (uint64_t)g);
        int end = 0;
        if (!goor_forward(&b, a, &end)) break;
        w->plies++;
        if (end) break;
        if (w->max_ply > 0 && b.ply > w->max_ply) break;
      }
    }
  }
out:
  free(feat);
  return NULL;
}

int64_t goor_bench_playouts(int n_boards, int threads, uint64_t seed, int max_ply, int with_features,
                            int games_per_board, double budget_sec, double* seconds, double* checksum) {
  if (threads < 1) threads = 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
  Work* w = (Work*)calloc(threads, sizeof(Work));
  struct timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int t = 0; t < threads; ++t) {
    w[t].tid = t; w[t].threads = threads; w[t].n_boards = n_boards; w[t].max_ply = max_ply;
    w[t].with_features = with_features; w[t].games = games_per_board; w[t].seed = seed;
    w[t].budget = budget_sec; w[t].t0 = t0;
    pthread_create(&th[t], NULL, bench_worker, &w[t]);
  }
  int64_t plies = 0;
  double sum = 0;
  for (int t = 0; t < threads; ++t) { pthread_join(th[t], NULL); plies += w[t].plies; sum += w[t].sum; }
  if (seconds) *seconds = since(&t0);
  if (checksum) *checksum = sum;
  free(th); free(w);
  return plies;
}
