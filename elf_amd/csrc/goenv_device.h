// goenv_device.h -- wave-per-board device code of the gfx950 Go environment.
//
// One 64-lane wavefront owns one 19x19 board.  Lane l handles the points
// p = l + 64*k, k = 0..5 (p = x*19 + y, the reference's export order,
// go/board.h:155), so every batch-buffer store is lane-contiguous.  Neighbour
// lookups go through a 21x21 padded label image in LDS (index c = (x+1)*21 +
// (y+1)) whose rim is a sentinel colour -- the same trick as the reference's
// 21x21 `_infos` (go/board.h:80) -- so no edge tests are needed.
//
// State model (differs from the reference on purpose): every stone carries the
// label (colour<<9 | root) where root is the point index of the most recently
// added stone of its group.  Merging groups and removing captured groups are
// therefore single lane-parallel relabel passes (replacing the linked-list
// walks of MergeGroups/EmptyGroup, go/board.cc:559-593,412-434), and the
// liberty table lib[root] is rebuilt from scratch each ply by letting every
// empty point add 1 to each distinct neighbouring root with LDS atomics
// (replacing the incremental bookkeeping of go/board.cc:393-409,529-555,
// 595-620).  The reference's stored liberties always equal this from-scratch
// count (VerifyBoard, go/board.cc:1455-1462), which is what makes the two
// models observably identical.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace goenv {

constexpr int BS = 19;
constexpr int NPTS = 361;
constexpr int PW = 21;       // padded width
constexpr int KCH = 6;       // 64-lane chunks covering 361 points
constexpr int LAB_N = 448;   // padded label image entries (441 used)
constexpr int TAB_N = 368;   // per-point tables, 361 rounded up to 16 B of u16
constexpr int HIST_LUT_N = 2048;

constexpr int S_EMPTY = 0, S_BLACK = 1, S_WHITE = 2;
constexpr uint32_t BORDER = 3u << 9;
constexpr int DIST_INF = 64;

// scalar slots inside the record (same order as the canonical view, include/goenv.h)
enum { SC_PLY = 0, SC_NEXT, SC_BCAP, SC_WCAP, SC_L1, SC_L2, SC_L3, SC_L4, SC_KOAGE, SC_KO,
       SC_KOCOLOR, SC_DONE, SC_GAME, SC_COUNTED, SC_R14, SC_R15 };

// Packed per-board record in HBM: 2272 B, 16-B aligned, one contiguous read per wave.
struct __attribute__((aligned(16))) BoardRec {
  uint16_t lab[TAB_N];    // colour<<9 | root, 0 = empty            (export order)
  uint16_t lastp[TAB_N];  // ply of placement (go/board.h:46); stale under empties
  uint16_t lib[TAB_N];    // liberties of the group whose root is this point
  int32_t sc[16];
};
static_assert(sizeof(BoardRec) == 2272, "record layout");

// Per-wave LDS working set (4.7 KB -> 32 waves/CU fit in 160 KB).
struct WaveLds {
  uint32_t lib[TAB_N];      // liberties by root
  uint32_t desc[TAB_N];     // per-point feature descriptor; also flood-fill scratch
  uint16_t lab[LAB_N];      // padded label image
  uint8_t dist[2][TAB_N];   // L1 distance to nearest black / white stone
};

struct Scalars {
  int ply, next, bcap, wcap, l1, l2, l3, l4, koage, ko, kocolor, done, game, counted;
};

struct Geo {
  int p[KCH];  // point index (may be >= 361 in the last chunk)
  int c[KCH];  // padded index, 0 (a rim cell) for invalid lanes
};

#define GO_UNROLL _Pragma("unroll")
#define WSYNC() __syncthreads()  // block == one wave: lowers to a wave-level fence

__device__ __forceinline__ int popc64(unsigned long long v) { return __popcll(v); }

__device__ __forceinline__ void geo_init(Geo& g, int lane) {
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    int p = lane + 64 * k;
    int x = p / BS, y = p - x * BS;
    g.p[k] = p;
    g.c[k] = (p < NPTS) ? (x + 1) * PW + (y + 1) : 0;
  }
}

__device__ __forceinline__ int pad_of(int p) {
  int x = p / BS, y = p - x * BS;
  return (x + 1) * PW + (y + 1);
}

__device__ __forceinline__ void load_scalars(const BoardRec* rec, Scalars& s) {
  const int32_t* sc = rec->sc;
  s.ply = sc[SC_PLY]; s.next = sc[SC_NEXT]; s.bcap = sc[SC_BCAP]; s.wcap = sc[SC_WCAP];
  s.l1 = sc[SC_L1]; s.l2 = sc[SC_L2]; s.l3 = sc[SC_L3]; s.l4 = sc[SC_L4];
  s.koage = sc[SC_KOAGE]; s.ko = sc[SC_KO]; s.kocolor = sc[SC_KOCOLOR];
  s.done = sc[SC_DONE]; s.game = sc[SC_GAME]; s.counted = sc[SC_COUNTED];
}

__device__ __forceinline__ void store_scalars(BoardRec* rec, const Scalars& s, int lane) {
  // one 64-B line, written by 16 lanes
  int v = 0;
  switch (lane) {
    case SC_PLY: v = s.ply; break;       case SC_NEXT: v = s.next; break;
    case SC_BCAP: v = s.bcap; break;     case SC_WCAP: v = s.wcap; break;
    case SC_L1: v = s.l1; break;         case SC_L2: v = s.l2; break;
    case SC_L3: v = s.l3; break;         case SC_L4: v = s.l4; break;
    case SC_KOAGE: v = s.koage; break;   case SC_KO: v = s.ko; break;
    case SC_KOCOLOR: v = s.kocolor; break; case SC_DONE: v = s.done; break;
    case SC_GAME: v = s.game; break;     case SC_COUNTED: v = s.counted; break;
    default: v = 0;
  }
  if (lane < 16) rec->sc[lane] = v;
}

// ClearBoard, go/board.cc:30-46 (scalars only; the caller zeroes the tables)
__device__ __forceinline__ void clear_scalars(Scalars& s) {
  s.ply = 1; s.next = S_BLACK; s.bcap = 0; s.wcap = 0;
  s.l1 = s.l2 = s.l3 = s.l4 = -1;
  s.koage = 0; s.ko = -1; s.kocolor = 0; s.done = 0; s.counted = 0;
}

__device__ __forceinline__ void fill_rim(WaveLds& L, int lane) {
  for (int i = lane; i < LAB_N; i += 64) L.lab[i] = (uint16_t)BORDER;
}

// HBM record -> LDS image + registers.
__device__ __forceinline__ void load_board(const BoardRec* rec, WaveLds& L, const Geo& g, int lane,
                                           uint32_t lab[KCH], uint32_t lastp[KCH]) {
  uint32_t lb[KCH];
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    bool v = g.p[k] < NPTS;
    lab[k] = v ? rec->lab[g.p[k]] : 0u;
    lastp[k] = v ? rec->lastp[g.p[k]] : 0u;
    lb[k] = v ? rec->lib[g.p[k]] : 0u;
  }
  fill_rim(L, lane);
  WSYNC();
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    if (g.p[k] < NPTS) {
      L.lab[g.c[k]] = (uint16_t)lab[k];
      L.lib[g.p[k]] = lb[k];
    }
  }
  WSYNC();
}

__device__ __forceinline__ void clear_board_lds(WaveLds& L, const Geo& g, int lane, uint32_t lab[KCH],
                                                uint32_t lastp[KCH]) {
  fill_rim(L, lane);
  WSYNC();
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    lab[k] = 0; lastp[k] = 0;
    if (g.p[k] < NPTS) { L.lab[g.c[k]] = 0; L.lib[g.p[k]] = 0; }
  }
  WSYNC();
}

__device__ __forceinline__ void store_board(BoardRec* rec, const WaveLds& L, const Geo& g, int lane,
                                            const uint32_t lab[KCH], const uint32_t lastp[KCH],
                                            const Scalars& s) {
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    if (g.p[k] < NPTS) {
      rec->lab[g.p[k]] = (uint16_t)lab[k];
      rec->lastp[g.p[k]] = (uint16_t)lastp[k];
      rec->lib[g.p[k]] = (uint16_t)L.lib[g.p[k]];
    }
  }
  store_scalars(rec, s, lane);
}

// 4 neighbours of padded index c in the reference's L,T,R,B order (go/board.h:183):
// x-1, y-1, x+1, y+1.
__device__ __forceinline__ void nbr4(const WaveLds& L, int c, uint32_t n[4]) {
  n[0] = L.lab[c - PW];
  n[1] = L.lab[c - 1];
  n[2] = L.lab[c + PW];
  n[3] = L.lab[c + 1];
}

__device__ __forceinline__ bool is_stone(uint32_t v) { return v != 0 && v < BORDER; }

// Liberty table from scratch: every empty point adds 1 to each DISTINCT neighbouring root
// (== the distinct-liberty count RecomputeGroupLiberties produces, go/board.cc:595-620).
__device__ __forceinline__ void rebuild_lib(WaveLds& L, const Geo& g, const uint32_t lab[KCH]) {
  GO_UNROLL
  for (int k = 0; k < KCH; ++k)
    if (g.p[k] < NPTS) L.lib[g.p[k]] = 0;
  WSYNC();
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    if (g.p[k] < NPTS && lab[k] == 0) {
      uint32_t n[4];
      nbr4(L, g.c[k], n);
      int r0 = is_stone(n[0]) ? (int)(n[0] & 511) : -1;
      int r1 = is_stone(n[1]) ? (int)(n[1] & 511) : -1;
      int r2 = is_stone(n[2]) ? (int)(n[2] & 511) : -1;
      int r3 = is_stone(n[3]) ? (int)(n[3] & 511) : -1;
      if (r0 >= 0) atomicAdd(&L.lib[r0], 1u);
      if (r1 >= 0 && r1 != r0) atomicAdd(&L.lib[r1], 1u);
      if (r2 >= 0 && r2 != r0 && r2 != r1) atomicAdd(&L.lib[r2], 1u);
      if (r3 >= 0 && r3 != r0 && r3 != r1 && r3 != r2) atomicAdd(&L.lib[r3], 1u);
    }
  }
  WSYNC();
}

// Legal mask (FindAllValidMoves, go/board.cc:763-779: empty, not a simple-ko violation
// :156-161, not suicide :131-154) and own-true-eye mask (IsTrueEye :1501-1552) for `player`.
template <bool WANT_EYE>
__device__ __forceinline__ void move_masks(const WaveLds& L, const Geo& g, const uint32_t lab[KCH],
                                           const Scalars& s, int player, bool legal[KCH], bool eye[KCH]) {
  const uint32_t opp = 3u - (uint32_t)player;
  const int ko_p = (s.koage == 0 && s.kocolor == player) ? s.ko : -1;
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    bool ok = false, e = false;
    if (g.p[k] < NPTS && lab[k] == 0) {
      uint32_t n[4];
      nbr4(L, g.c[k], n);
      bool all_own = true;
      GO_UNROLL
      for (int i = 0; i < 4; ++i) {
        uint32_t col = n[i] >> 9;
        if (n[i] == 0) { ok = true; all_own = false; }
        else if (col == (uint32_t)player) { if (L.lib[n[i] & 511] > 1) ok = true; }
        else if (col == opp) { all_own = false; if (L.lib[n[i] & 511] == 1) ok = true; }
      }
      if (g.p[k] == ko_p) ok = false;
      if (WANT_EYE && all_own) {
        int c = g.c[k];
        uint32_t d0 = L.lab[c - PW - 1] >> 9, d1 = L.lab[c - PW + 1] >> 9;
        uint32_t d2 = L.lab[c + PW - 1] >> 9, d3 = L.lab[c + PW + 1] >> 9;
        int n_opp = (d0 == opp) + (d1 == opp) + (d2 == opp) + (d3 == opp);
        int n_rim = (d0 == 3) + (d1 == 3) + (d2 == 3) + (d3 == 3);
        bool fake = (n_rim > 0 && n_opp >= 1) || (n_rim == 0 && n_opp >= 2);
        e = !fake;
      }
    }
    legal[k] = ok;
    eye[k] = e;
  }
}

// Legality of one move for `player` (TryPlay, go/board.cc:626-662).  Wave-uniform.
__device__ __forceinline__ bool move_is_legal(const WaveLds& L, const Scalars& s, int p, int player) {
  const int c = pad_of(p);
  if (L.lab[c] != 0) return false;
  if (s.ko == p && s.koage == 0 && s.kocolor == player) return false;
  uint32_t n[4];
  nbr4(L, c, n);
  bool ok = false;
  GO_UNROLL
  for (int i = 0; i < 4; ++i) {
    uint32_t col = n[i] >> 9;
    if (n[i] == 0) ok = true;
    else if (col == (uint32_t)player) { if (L.lib[n[i] & 511] > 1) ok = true; }
    else if (col == 3u - (uint32_t)player) { if (L.lib[n[i] & 511] == 1) ok = true; }
  }
  return ok;
}

// update_next_move, go/board.cc:1014-1027
__device__ __forceinline__ void advance(Scalars& s, int action, int player) {
  s.next = 3 - player;
  s.l4 = s.l3; s.l3 = s.l2; s.l2 = s.l1; s.l1 = action;
  s.ply += 1;
}

// IsGameEnd, go/board.cc:1690-1692
__device__ __forceinline__ bool game_end(const Scalars& s) {
  return s.ply > 1 && ((s.l1 == -1 && s.l2 == -1) || s.l1 == -2);
}

// Play (go/board.cc:1083-1179) of a LEGAL board move p for `player`.
__device__ __forceinline__ void apply_move(WaveLds& L, const Geo& g, int lane, uint32_t lab[KCH],
                                           uint32_t lastp[KCH], Scalars& s, int p, int player) {
  const int c = pad_of(p);
  const uint32_t opp = 3u - (uint32_t)player;
  uint32_t n[4];
  nbr4(L, c, n);
  // own neighbour roots merge into the new stone; enemy neighbour groups whose last
  // liberty is p are captured (board.cc:1113-1148).
  int own[4], dead[4];
  bool merged = false;
  GO_UNROLL
  for (int i = 0; i < 4; ++i) {
    uint32_t col = n[i] >> 9;
    int r = (int)(n[i] & 511);
    own[i] = (col == (uint32_t)player) ? r : -1;
    dead[i] = (col == opp && L.lib[r] == 1) ? r : -1;
    own[i] = __builtin_amdgcn_readfirstlane(own[i]);
    dead[i] = __builtin_amdgcn_readfirstlane(dead[i]);
    merged |= own[i] >= 0;
  }
  const uint32_t nv = ((uint32_t)player << 9) | (uint32_t)p;
  int total_capture = 0, capture_p = -1;
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    uint32_t v = lab[k];
    int r = (int)(v & 511);
    uint32_t col = v >> 9;
    bool mine = col == (uint32_t)player && (r == own[0] || r == own[1] || r == own[2] || r == own[3]);
    bool cap = col == opp && (r == dead[0] || r == dead[1] || r == dead[2] || r == dead[3]);
    if (mine) v = nv;
    if (cap) v = 0;
    if (g.p[k] == p) { v = nv; lastp[k] = (uint32_t)s.ply & 0xffffu; }  // board.cc:532,1157
    unsigned long long m = __ballot(cap);
    if (m) { total_capture += popc64(m); capture_p = 64 * k + (int)__builtin_ctzll(m); }
    if (v != lab[k]) { L.lab[g.c[k]] = (uint16_t)v; lab[k] = v; }
  }
  if (player == S_BLACK) s.bcap += total_capture; else s.wcap += total_capture;  // board.cc:1132-1133
  WSYNC();
  rebuild_lib(L, g, lab);
  // simple ko, board.cc:1163-1171
  if (!merged && total_capture == 1 && L.lib[p] == 1) {
    s.ko = capture_p; s.kocolor = (int)opp; s.koage = 0;
  } else {
    s.koage = (s.koage + 1) & 0xffff;
  }
  advance(s, p, player);
}

// counter-based policy RNG, identical in oracle/go_oracle.c (goor_policy_rand)
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27; z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return z;
}
__device__ __forceinline__ uint64_t policy_rand(uint64_t seed, uint64_t board_id, uint64_t game, uint64_t ply) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (board_id + 1);
  z = mix64(z ^ (game * 0xD1B54A32D192ED03ull));
  return mix64(z + ply * 0x9E3779B97F4A7C15ull);
}

// cfg-2 policy: uniform over (legal AND NOT own true eye), pass if none.  Wave-uniform result.
__device__ __forceinline__ int pick_random_move(const WaveLds& L, const Geo& g, int lane,
                                                const uint32_t lab[KCH], const Scalars& s,
                                                uint64_t seed, uint64_t board_id) {
  bool legal[KCH], eye[KCH];
  move_masks<true>(L, g, lab, s, s.next, legal, eye);
  unsigned long long w[KCH];
  int n = 0;
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) { w[k] = __ballot(legal[k] && !eye[k]); n += popc64(w[k]); }
  if (n == 0) return -1;
  uint64_t u = policy_rand(seed, board_id, (uint64_t)s.game, (uint64_t)s.ply);
  int idx = (int)(((u >> 32) * (uint64_t)n) >> 32);
  int chosen = -1;
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    int cnt = popc64(w[k]);
    if (chosen < 0) {
      if (idx < cnt) {
        int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(w[k] >> 32),
                                                  __builtin_amdgcn_mbcnt_lo((uint32_t)w[k], 0u));
        unsigned long long hit = __ballot(((w[k] >> lane) & 1ull) && rank == idx);
        chosen = 64 * k + (int)__builtin_ctzll(hit);
      } else {
        idx -= cnt;
      }
    }
  }
  return chosen;
}

// GetTrompTaylorScore(board, NULL, NULL), go/board.cc:1584-1688: stones plus empty regions
// that touch exactly one colour; 0 on an empty board.  Monotone OR-propagation of
// "touches black / touches white" bits over the empties, in place, until a fixed point.
__device__ __forceinline__ int tromp_taylor(WaveLds& L, const Geo& g, int lane, const uint32_t lab[KCH]) {
  uint16_t* flag = reinterpret_cast<uint16_t*>(L.desc);  // padded image, LAB_N u16 <= TAB_N u32
  for (int i = lane; i < LAB_N; i += 64) flag[i] = 0;
  WSYNC();
  uint32_t f[KCH];
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    f[k] = 0;
    if (g.p[k] < NPTS && lab[k] == 0) {
      uint32_t n[4];
      nbr4(L, g.c[k], n);
      GO_UNROLL
      for (int i = 0; i < 4; ++i) {
        uint32_t col = n[i] >> 9;
        if (col == 1 || col == 2) f[k] |= col;
      }
      flag[g.c[k]] = (uint16_t)f[k];
    }
  }
  WSYNC();
  for (int iter = 0; iter < 2 * NPTS; ++iter) {  // bounded: a fixed point is reached long before
    bool changed = false;
    GO_UNROLL
    for (int k = 0; k < KCH; ++k) {
      if (g.p[k] < NPTS && lab[k] == 0) {
        int c = g.c[k];
        uint32_t nf = f[k] | flag[c - PW] | flag[c - 1] | flag[c + PW] | flag[c + 1];
        if (nf != f[k]) { f[k] = nf; flag[c] = (uint16_t)nf; changed = true; }
      }
      WSYNC();
    }
    if (!__any(changed)) break;
  }
  int black = 0, white = 0;
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    uint32_t col = lab[k] >> 9;
    bool e = g.p[k] < NPTS && lab[k] == 0;
    black += popc64(__ballot(col == 1 || (e && f[k] == 1)));
    white += popc64(__ballot(col == 2 || (e && f[k] == 2)));
  }
  WSYNC();
  return black - white;
}

// GetFastScore(board, RULE_CHINESE), go/board.cc:1560-1582: stones + true eyes per colour.
__device__ __forceinline__ int fast_score_cn(const WaveLds& L, const Geo& g, const uint32_t lab[KCH],
                                             const Scalars& s) {
  bool lg[KCH], eb[KCH], ew[KCH];
  move_masks<true>(L, g, lab, s, S_BLACK, lg, eb);
  move_masks<true>(L, g, lab, s, S_WHITE, lg, ew);
  int sc = 0;
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    uint32_t col = lab[k] >> 9;
    sc += popc64(__ballot(col == 1 || (eb[k] && !ew[k])));
    sc -= popc64(__ballot(col == 2 || ew[k]));
  }
  return sc;
}

// Exact L1 distance to the nearest black / white stone (GetDistanceMap + DistanceTransform,
// go/board_feature.cc:17-38,148-160).  Pass 1 (along y) is a bit trick on the 19-bit column
// occupancy, pass 2 (along x) the reference's forward/backward min-plus sweep, one lane per
// (colour, y) line, fully unrolled in registers.  Returns has_black | has_white<<1.
__device__ __forceinline__ int distance_maps(WaveLds& L, int lane, const uint32_t lab[KCH]) {
  unsigned long long wb[KCH + 1], ww[KCH + 1];
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    wb[k] = __ballot((lab[k] >> 9) == 1);
    ww[k] = __ballot((lab[k] >> 9) == 2);
  }
  wb[KCH] = 0; ww[KCH] = 0;
  int has = 0;
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) { if (wb[k]) has |= 1; if (ww[k]) has |= 2; }
  const bool white_lane = lane >= 32;
  const int y = lane & 31;
  int a[BS];
  GO_UNROLL
  for (int x = 0; x < BS; ++x) {
    const int bit = x * BS, w = bit >> 6, off = bit & 63;
    unsigned long long vb = wb[w] >> off, vw = ww[w] >> off;
    if (off > 64 - BS) { vb |= wb[w + 1] << (64 - off); vw |= ww[w + 1] << (64 - off); }
    uint32_t m = (uint32_t)(white_lane ? vw : vb) & 0x7ffffu;
    uint32_t lo = m & ((2u << y) - 1u);
    uint32_t hi = m >> y;
    int dl = lo ? y - (31 - __clz((int)lo)) : DIST_INF;
    int dh = hi ? (int)__builtin_ctz(hi) : DIST_INF;
    a[x] = min(dl, dh);
  }
  GO_UNROLL
  for (int x = 1; x < BS; ++x) a[x] = min(a[x], a[x - 1] + 1);
  GO_UNROLL
  for (int x = BS - 2; x >= 0; --x) a[x] = min(a[x], a[x + 1] + 1);
  if (y < BS) {
    uint8_t* d = L.dist[white_lane ? 1 : 0];
    GO_UNROLL
    for (int x = 0; x < BS; ++x) d[x * BS + y] = (uint8_t)a[x];
  }
  WSYNC();
  return has;
}

// BoardFeature::Transform inverse (go/board_feature.h:50-60): source point of output index q.
__device__ __forceinline__ int d4_source(int code, int q) {
  int a = q / BS, b = q - a * BS;
  if ((code >> 2) == 1) { int t = a; a = b; b = t; }
  int rot = code & 3, x, y;
  if (rot == 1) { x = BS - 1 - b; y = a; }
  else if (rot == 2) { x = BS - 1 - a; y = BS - 1 - b; }
  else if (rot == 3) { x = b; y = BS - 1 - a; }
  else { x = a; y = b; }
  return x * BS + y;
}

// BoardFeature::Extract (go/board_feature.cc:175-195): 25 planes x 361 floats, written
// lane-contiguously.  `lut` = host-built (float)exp(-d/10.0) table (board_feature.cc:141).
__device__ __forceinline__ void write_features(WaveLds& L, const Geo& g, int lane, const uint32_t lab[KCH],
                                               const uint32_t lastp[KCH], const Scalars& s, int aug,
                                               const float* __restrict__ lut, float* __restrict__ out) {
  const int has = distance_maps(L, lane, lab);
  const int ko_p = (s.koage == 0 && s.ko >= 0) ? s.ko : -1;  // GetSimpleKoLocation, board.cc:337-344
  // descriptor: colour(2) | libclass(2) | ko(1) | d_black(7) | d_white(7) | hist(11)
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    if (g.p[k] < NPTS) {
      uint32_t v = lab[k], col = v >> 9, d = col;
      if (col) {
        uint32_t lib = L.lib[v & 511];
        d |= (lib == 1 ? 0u : (lib == 2 ? 1u : 2u)) << 2;  // GetLibertyMap3binary :83-99
        int age = s.ply - (int)lastp[k];
        age = age < 0 ? 0 : (age > HIST_LUT_N - 1 ? HIST_LUT_N - 1 : age);
        d |= (uint32_t)age << 19;
      }
      if (g.p[k] == ko_p) d |= 16u;
      d |= (uint32_t)L.dist[0][g.p[k]] << 5;
      d |= (uint32_t)L.dist[1][g.p[k]] << 12;
      L.desc[g.p[k]] = d;
    }
  }
  WSYNC();
  const uint32_t our = (uint32_t)s.next, opp = 3u - our;
  const bool has_our = (has >> (our - 1)) & 1, has_opp = (has >> (opp - 1)) & 1;
  const int sh_our = our == 1 ? 5 : 12, sh_opp = our == 1 ? 12 : 5;
  uint32_t d[KCH];
  float h[KCH];
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    int q = g.p[k];
    d[k] = 0; h[k] = 0.f;
    if (q < NPTS) {
      d[k] = L.desc[aug ? d4_source(aug, q) : q];
      if (d[k] & 3u) h[k] = lut[d[k] >> 19];
    }
  }
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    const int q = g.p[k];
    if (q < NPTS) {
      const uint32_t col = d[k] & 3u, cls = (d[k] >> 2) & 3u;
      const bool mine = col == our, theirs = col == opp;
      float* o = out + q;
      o[0 * NPTS] = (mine && cls == 0) ? 1.f : 0.f;
      o[1 * NPTS] = (mine && cls == 1) ? 1.f : 0.f;
      o[2 * NPTS] = (mine && cls == 2) ? 1.f : 0.f;
      o[3 * NPTS] = (theirs && cls == 0) ? 1.f : 0.f;
      o[4 * NPTS] = (theirs && cls == 1) ? 1.f : 0.f;
      o[5 * NPTS] = (theirs && cls == 2) ? 1.f : 0.f;
      o[6 * NPTS] = (d[k] & 16u) ? 1.f : 0.f;
      o[7 * NPTS] = mine ? 1.f : 0.f;
      o[8 * NPTS] = theirs ? 1.f : 0.f;
      o[9 * NPTS] = col == 0 ? 1.f : 0.f;
      o[10 * NPTS] = mine ? h[k] : 0.f;
      o[11 * NPTS] = theirs ? h[k] : 0.f;
      o[12 * NPTS] = 0.f;
      o[13 * NPTS] = 0.f;
      o[14 * NPTS] = has_our ? (float)((d[k] >> sh_our) & 127u) : 10000.f;
      o[15 * NPTS] = has_opp ? (float)((d[k] >> sh_opp) & 127u) : 10000.f;
      GO_UNROLL
      for (int pl = 16; pl < 25; ++pl) o[pl * NPTS] = 0.f;
    }
  }
}

__device__ __forceinline__ void write_legal(const WaveLds& L, const Geo& g, const uint32_t lab[KCH],
                                            const Scalars& s, uint8_t* __restrict__ out) {
  bool legal[KCH], eye[KCH];
  move_masks<false>(L, g, lab, s, s.next, legal, eye);
  GO_UNROLL
  for (int k = 0; k < KCH; ++k)
    if (g.p[k] < NPTS) out[g.p[k]] = legal[k] ? 1 : 0;
}

}  // namespace goenv
