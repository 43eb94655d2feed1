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
// 13-bit label  root | colour<<9 | cls<<11  where root is the point index of the
// most recently added stone of its group and cls = min(liberties of the group, 3).
// * Merging groups and removing captured groups are single lane-parallel relabel
//   passes (replacing the linked-list walks of MergeGroups/EmptyGroup,
//   go/board.cc:559-593,412-434).
// * The liberty table lib[root] is rebuilt from scratch each ply: every empty
//   point adds 1 to each distinct neighbouring root with LDS atomics (replacing
//   the incremental bookkeeping of go/board.cc:393-409,529-555,595-620).  The
//   reference's stored liberties always equal this from-scratch count
//   (VerifyBoard, go/board.cc:1455-1462), so the two models are observably
//   identical.
// * Because every neighbour read returns colour AND liberty class, legality,
//   eye and feature tests are branch-free table lookups on one LDS read per
//   neighbour (no dependent group-table gather as in StoneLibertyAnalysis,
//   go/board.cc:95-129).
// * The legal mask and the playout-candidate mask of the side to move are
//   computed once per ply and kept in the record as 2 x 361-bit ballots.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace goenv {

constexpr int BS = 19;
constexpr int NPTS = 361;
constexpr int PW = 21;       // padded width
constexpr int KCH = 6;       // 64-lane chunks covering 361 points
constexpr int LAB_N = 448;   // padded label image entries (441 used)
constexpr int TAB_N = 368;   // per-point tables, 361 rounded up
constexpr int LIB_N = TAB_N + 64;  // + one atomic sink slot per lane
constexpr int HIST_LUT_N = 2048;

constexpr int S_EMPTY = 0, S_BLACK = 1, S_WHITE = 2;
constexpr uint32_t BORDER = 3u << 9;
constexpr uint32_t ROOT_MASK = 511u;
constexpr uint32_t LABEL_MASK = 0x7ffu;  // root | colour
constexpr int DIST_INF = 64;

// scalar slots inside the record (same order as the canonical view, include/goenv.h)
enum { SC_PLY = 0, SC_NEXT, SC_BCAP, SC_WCAP, SC_L1, SC_L2, SC_L3, SC_L4, SC_KOAGE, SC_KO,
       SC_KOCOLOR, SC_DONE, SC_GAME, SC_COUNTED, SC_SEQ, SC_SEQTOT };

// Packed per-board record in HBM: 1632 B, 16-B aligned, one contiguous read per wave.
struct __attribute__((aligned(16))) BoardRec {
  uint16_t lab[TAB_N];    // root | colour<<9 | cls<<11, 0 = empty       (export order)
  uint16_t lastp[TAB_N];  // ply of placement (go/board.h:46); stale under empties
  uint64_t legal[KCH];    // FindAllValidMoves of the side to move, bit p%64 of word p/64
  uint64_t cand[KCH];     // legal AND NOT own true eye (cfg-2 playout policy)
  int32_t sc[16];
};
static_assert(sizeof(BoardRec) == 1632, "record layout");

// Per-wave LDS working set (~4.9 KB -> 32 waves/CU fit in 160 KB).
struct WaveLds {
  uint32_t lib[LIB_N];      // liberties by root (+ per-lane sink slots for masked atomics)
  uint32_t desc[TAB_N];     // per-point feature descriptor; also flood-fill scratch
  uint16_t lab[LAB_N];      // padded label image
  uint8_t dist[2][TAB_N];   // L1 distance to nearest black / white stone
};

struct Scalars {
  int ply, next, bcap, wcap, l1, l2, l3, l4, koage, ko, kocolor, done, game, counted, seq, seqtot;
};

struct Geo {
  int p[KCH];  // point index (may be >= 361 in the last chunk)
  int c[KCH];  // padded index; an interior cell (never written) for invalid lanes
};

struct Masks {
  unsigned long long legal[KCH];
  unsigned long long cand[KCH];
};

#define GO_UNROLL _Pragma("unroll")
#define WSYNC() __syncthreads()  // block == one wave: lowers to a wave-level fence

__device__ __forceinline__ int popc64(unsigned long long v) { return __popcll(v); }
__device__ __forceinline__ uint32_t col_of(uint32_t v) { return (v >> 9) & 3u; }
__device__ __forceinline__ uint32_t cls_of(uint32_t v) { return (v >> 11) & 3u; }
__device__ __forceinline__ uint32_t typ_of(uint32_t v) { return (v >> 9) & 15u; }  // colour | cls<<2

__device__ __forceinline__ void geo_init(Geo& g, int lane) {
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    int p = lane + 64 * k;
    int x = p / BS, y = p - x * BS;
    g.p[k] = p;
    g.c[k] = (p < NPTS) ? (x + 1) * PW + (y + 1) : PW + 1;
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
  s.done = sc[SC_DONE]; s.game = sc[SC_GAME]; s.counted = sc[SC_COUNTED]; s.seq = sc[SC_SEQ];
  s.seqtot = sc[SC_SEQTOT];
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
    case SC_SEQ: v = s.seq; break;       case SC_SEQTOT: v = s.seqtot; break;
    default: v = 0;
  }
  if (lane < 16) rec->sc[lane] = v;
}

// ClearBoard, go/board.cc:30-46 (scalars only; the caller zeroes the tables)
__device__ __forceinline__ void clear_scalars(Scalars& s) {
  s.ply = 1; s.next = S_BLACK; s.bcap = 0; s.wcap = 0;
  s.l1 = s.l2 = s.l3 = s.l4 = -1;
  s.koage = 0; s.ko = -1; s.kocolor = 0; s.done = 0; s.counted = 0; s.seq = 0;
}

__device__ __forceinline__ void fill_rim(WaveLds& L, int lane) {
  for (int i = lane; i < LAB_N; i += 64) L.lab[i] = (uint16_t)BORDER;
}

// HBM record -> LDS image + registers.
__device__ __forceinline__ void load_board(const BoardRec* rec, WaveLds& L, const Geo& g, int lane,
                                           uint32_t lab[KCH], uint32_t lastp[KCH]) {
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    bool v = g.p[k] < NPTS;
    lab[k] = v ? rec->lab[g.p[k]] : 0u;
    lastp[k] = v ? rec->lastp[g.p[k]] : 0u;
  }
  fill_rim(L, lane);
  WSYNC();
  GO_UNROLL
  for (int k = 0; k < KCH; ++k)
    if (g.p[k] < NPTS) L.lab[g.c[k]] = (uint16_t)lab[k];
  WSYNC();
}

__device__ __forceinline__ void load_masks(const BoardRec* rec, Masks& m) {
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) { m.legal[k] = rec->legal[k]; m.cand[k] = rec->cand[k]; }
}

__device__ __forceinline__ void clear_board_lds(WaveLds& L, const Geo& g, int lane, uint32_t lab[KCH],
                                                uint32_t lastp[KCH]) {
  fill_rim(L, lane);
  WSYNC();
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    lab[k] = 0; lastp[k] = 0;
    if (g.p[k] < NPTS) L.lab[g.c[k]] = 0;
  }
  WSYNC();
}

// masks of an empty board with BLACK to move: every point legal and a candidate
__device__ __forceinline__ void full_masks(Masks& m) {
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    unsigned long long w = (k < KCH - 1) ? ~0ull : ((1ull << (NPTS - 64 * (KCH - 1))) - 1ull);
    m.legal[k] = w; m.cand[k] = w;
  }
}

__device__ __forceinline__ void store_board(BoardRec* rec, const Geo& g, int lane, const uint32_t lab[KCH],
                                            const uint32_t lastp[KCH], const Masks& m, const Scalars& s) {
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    if (g.p[k] < NPTS) {
      rec->lab[g.p[k]] = (uint16_t)lab[k];
      rec->lastp[g.p[k]] = (uint16_t)lastp[k];
    }
  }
  // 12 mask words, one lane each
  unsigned long long w = 0;
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    if (lane == k) w = m.legal[k];
    if (lane == KCH + k) w = m.cand[k];
  }
  if (lane < 2 * KCH) rec->legal[lane] = w;  // legal[6] and cand[6] are contiguous
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

__device__ __forceinline__ bool is_stone(uint32_t v) { uint32_t c = col_of(v); return c == 1 || c == 2; }

// Liberty table from scratch: every empty point adds 1 to each DISTINCT neighbouring root
// (== the distinct-liberty count RecomputeGroupLiberties produces, go/board.cc:595-620), then
// every stone's cls bits are refreshed to min(liberties, 3).  Branch-free: masked-off adds go to
// a per-lane sink slot.
__device__ __forceinline__ void rebuild_lib(WaveLds& L, const Geo& g, int lane, uint32_t lab[KCH]) {
  GO_UNROLL
  for (int k = 0; k < KCH; ++k)
    if (g.p[k] < NPTS) L.lib[g.p[k]] = 0;
  WSYNC();
  const int sink = TAB_N + lane;
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    if (g.p[k] < NPTS && lab[k] == 0) {
      uint32_t n[4];
      nbr4(L, g.c[k], n);
      int r0 = is_stone(n[0]) ? (int)(n[0] & ROOT_MASK) : sink;
      int r1 = is_stone(n[1]) ? (int)(n[1] & ROOT_MASK) : sink;
      int r2 = is_stone(n[2]) ? (int)(n[2] & ROOT_MASK) : sink;
      int r3 = is_stone(n[3]) ? (int)(n[3] & ROOT_MASK) : sink;
      r1 = (r1 == r0) ? sink : r1;
      r2 = (r2 == r0 || r2 == r1) ? sink : r2;
      r3 = (r3 == r0 || r3 == r1 || r3 == r2) ? sink : r3;
      atomicAdd(&L.lib[r0], 1u);
      atomicAdd(&L.lib[r1], 1u);
      atomicAdd(&L.lib[r2], 1u);
      atomicAdd(&L.lib[r3], 1u);
    }
  }
  WSYNC();
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    if (g.p[k] < NPTS && lab[k] != 0) {
      uint32_t lib = L.lib[lab[k] & ROOT_MASK];
      uint32_t v = (lab[k] & LABEL_MASK) | ((lib > 3u ? 3u : lib) << 11);
      lab[k] = v;
      L.lab[g.c[k]] = (uint16_t)v;
    }
  }
  WSYNC();
}

// 16-entry truth tables over typ = colour | cls<<2 of a neighbour.
//   OK   : neighbour makes the move non-suicidal (IsSuicideMove go/board.cc:131-154):
//          empty, own group with > 1 liberty, or enemy group with exactly 1 liberty
//   OWNR : neighbour is own stone or rim (IsEye go/board.cc:1501-1508)
__device__ __forceinline__ uint32_t tab_ok(int player) {
  // typ bits: colour in 0..1, cls in 2..3.  empty: typ 0.
  uint32_t t = 1u << 0;
  const uint32_t own = (uint32_t)player, opp = 3u - own;
  t |= (1u << (own | (2u << 2))) | (1u << (own | (3u << 2)));
  t |= 1u << (opp | (1u << 2));
  return t;
}
__device__ __forceinline__ uint32_t tab_col(uint32_t colour) {  // any cls
  return (1u << colour) | (1u << (colour | 4u)) | (1u << (colour | 8u)) | (1u << (colour | 12u));
}

// Legal mask (FindAllValidMoves, go/board.cc:763-779: empty, not a simple-ko violation :156-161,
// not suicide :131-154) and own-true-eye mask (IsTrueEye :1501-1552) for `player`, as ballots.
__device__ __forceinline__ void move_masks(const WaveLds& L, const Geo& g, const uint32_t lab[KCH],
                                           const Scalars& s, int player, Masks& m,
                                           unsigned long long eye_w[KCH]) {
  const uint32_t OK = tab_ok(player);
  const uint32_t OWNR = tab_col((uint32_t)player) | tab_col(3u);
  const uint32_t OPP = tab_col(3u - (uint32_t)player);
  const uint32_t RIM = tab_col(3u);
  const int ko_p = (s.koage == 0 && s.kocolor == player) ? s.ko : -1;
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    const int c = g.c[k];
    uint32_t n[4];
    nbr4(L, c, n);
    const uint32_t d0 = typ_of(L.lab[c - PW - 1]), d1 = typ_of(L.lab[c - PW + 1]);
    const uint32_t d2 = typ_of(L.lab[c + PW - 1]), d3 = typ_of(L.lab[c + PW + 1]);
    const uint32_t t0 = typ_of(n[0]), t1 = typ_of(n[1]), t2 = typ_of(n[2]), t3 = typ_of(n[3]);
    const bool empty = g.p[k] < NPTS && lab[k] == 0;
    const uint32_t ok = ((OK >> t0) | (OK >> t1) | (OK >> t2) | (OK >> t3)) & 1u;
    const uint32_t all_own = (OWNR >> t0) & (OWNR >> t1) & (OWNR >> t2) & (OWNR >> t3) & 1u;
    const int n_opp = (int)(((OPP >> d0) & 1u) + ((OPP >> d1) & 1u) + ((OPP >> d2) & 1u) + ((OPP >> d3) & 1u));
    const int n_rim = (int)(((RIM >> d0) & 1u) + ((RIM >> d1) & 1u) + ((RIM >> d2) & 1u) + ((RIM >> d3) & 1u));
    const bool fake = n_rim > 0 ? n_opp >= 1 : n_opp >= 2;  // IsFakeEye go/board.cc:1530-1544
    const bool legal = empty && ok && g.p[k] != ko_p;
    const bool eye = empty && all_own && !fake;
    m.legal[k] = __ballot(legal);
    eye_w[k] = __ballot(eye);
    m.cand[k] = m.legal[k] & ~eye_w[k];
  }
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

// Legality of one board move for `player` from the image (TryPlay, go/board.cc:626-662).
// Wave-uniform.  Used where the stored masks do not apply (handicap stones are BLACK's
// whoever is to move).
__device__ __forceinline__ bool move_is_legal(const WaveLds& L, const Scalars& s, int p, int player) {
  const int c = pad_of(p);
  if (L.lab[c] != 0) return false;
  if (s.ko == p && s.koage == 0 && s.kocolor == player) return false;
  uint32_t n[4];
  nbr4(L, c, n);
  const uint32_t OK = tab_ok(player);
  return (((OK >> typ_of(n[0])) | (OK >> typ_of(n[1])) | (OK >> typ_of(n[2])) | (OK >> typ_of(n[3]))) & 1u) != 0;
}

// Play (go/board.cc:1083-1179) of a LEGAL board move p for `player`; leaves the liberty table and
// cls bits rebuilt.
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
    const uint32_t col = col_of(n[i]);
    const int r = (int)(n[i] & ROOT_MASK);
    own[i] = __builtin_amdgcn_readfirstlane((col == (uint32_t)player) ? r : -1);
    dead[i] = __builtin_amdgcn_readfirstlane((col == opp && cls_of(n[i]) == 1) ? r : -1);
    merged |= own[i] >= 0;
  }
  const uint32_t nv = ((uint32_t)player << 9) | (uint32_t)p;
  int total_capture = 0, capture_p = -1;
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    uint32_t v = lab[k];
    const int r = (int)(v & ROOT_MASK);
    const uint32_t col = col_of(v);
    const bool mine = col == (uint32_t)player && (r == own[0] || r == own[1] || r == own[2] || r == own[3]);
    const bool cap = col == opp && (r == dead[0] || r == dead[1] || r == dead[2] || r == dead[3]);
    if (mine) v = nv;
    if (cap) v = 0;
    if (g.p[k] == p) { v = nv; lastp[k] = (uint32_t)s.ply & 0xffffu; }  // board.cc:532,1157
    const unsigned long long m = __ballot(cap);
    if (m) { total_capture += popc64(m); capture_p = 64 * k + (int)__builtin_ctzll(m); }
    if (v != lab[k]) { L.lab[g.c[k]] = (uint16_t)v; lab[k] = v; }
  }
  if (player == S_BLACK) s.bcap += total_capture; else s.wcap += total_capture;  // board.cc:1132-1133
  WSYNC();
  rebuild_lib(L, g, lane, lab);
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

// cfg-2 policy: uniform over the candidate ballots (legal AND NOT own true eye), pass if none.
// Wave-uniform result.
__device__ __forceinline__ int pick_random_move(const Masks& m, int lane, const Scalars& s, uint64_t seed,
                                                uint64_t board_id) {
  int n = 0;
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) n += popc64(m.cand[k]);
  if (n == 0) return -1;
  const uint64_t u = policy_rand(seed, board_id, (uint64_t)s.game, (uint64_t)s.ply);
  int idx = (int)(((u >> 32) * (uint64_t)n) >> 32);
  int chosen = -1;
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    const unsigned long long w = m.cand[k];
    const int cnt = popc64(w);
    if (chosen < 0) {
      if (idx < cnt) {
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(w >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)w, 0u));
        const unsigned long long hit = __ballot(((w >> lane) & 1ull) && rank == idx);
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
        const uint32_t col = col_of(n[i]);
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
        const int c = g.c[k];
        const uint32_t nf = f[k] | flag[c - PW] | flag[c - 1] | flag[c + PW] | flag[c + 1];
        if (nf != f[k]) { f[k] = nf; flag[c] = (uint16_t)nf; changed = true; }
      }
      WSYNC();
    }
    if (!__any(changed)) break;
  }
  int black = 0, white = 0;
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    const uint32_t col = col_of(lab[k]);
    const bool e = g.p[k] < NPTS && lab[k] == 0;
    black += popc64(__ballot(col == 1 || (e && f[k] == 1)));
    white += popc64(__ballot(col == 2 || (e && f[k] == 2)));
  }
  WSYNC();
  return black - white;
}

// GetFastScore(board, RULE_CHINESE), go/board.cc:1560-1582: stones + true eyes per colour
// (GetEyeColor tests white first, :1554-1558).
__device__ __forceinline__ int fast_score_cn(const WaveLds& L, const Geo& g, const uint32_t lab[KCH],
                                             const Scalars& s) {
  Masks m;
  unsigned long long eb[KCH], ew[KCH];
  move_masks(L, g, lab, s, S_BLACK, m, eb);
  move_masks(L, g, lab, s, S_WHITE, m, ew);
  int sc = 0;
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    const uint32_t col = col_of(lab[k]);
    sc += popc64(__ballot(col == 1)) + popc64(eb[k] & ~ew[k]);
    sc -= popc64(__ballot(col == 2)) + popc64(ew[k]);
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
    wb[k] = __ballot(col_of(lab[k]) == 1);
    ww[k] = __ballot(col_of(lab[k]) == 2);
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
    const uint32_t m = (uint32_t)(white_lane ? vw : vb) & 0x7ffffu;
    const uint32_t lo = m & ((2u << y) - 1u);
    const uint32_t hi = m >> y;
    const int dl = lo ? y - (31 - __clz((int)lo)) : DIST_INF;
    const int dh = hi ? (int)__builtin_ctz(hi) : DIST_INF;
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

// BoardFeature::Transform (go/board_feature.h:38-48): output index of point p under a D4 code.
__device__ __forceinline__ int d4_dest(int code, int p) {
  const int x = p / BS, y = p - x * BS;
  const int rot = code & 3;
  int ox, oy;
  if (rot == 1) { ox = y; oy = BS - 1 - x; }
  else if (rot == 2) { ox = BS - 1 - x; oy = BS - 1 - y; }
  else if (rot == 3) { ox = BS - 1 - y; oy = x; }
  else { ox = x; oy = y; }
  if ((code >> 2) == 1) { const int t = ox; ox = oy; oy = t; }
  return ox * BS + oy;
}

// 4 floats with only 4-byte alignment: planes start every 1444 B and boards every 36100 B, so
// 16-B alignment is not available; gfx950 takes dwordx4 global stores at dword alignment at full
// rate (tools/store_bench.hip).
struct __attribute__((packed, aligned(4))) float4u { float x, y, z, w; };

__device__ __forceinline__ void zero_range(float* __restrict__ o, int count, int lane) {
  const float4u z = {0.f, 0.f, 0.f, 0.f};
  const int body = count & ~3;
  for (int i = 4 * lane; i < body; i += 256) *reinterpret_cast<float4u*>(o + i) = z;
  if (lane < count - body) o[body + lane] = 0.f;
}

// BoardFeature::Extract (go/board_feature.cc:175-195): 25 planes x 361 floats.
// Store shape: plane-major, each plane as two dwordx4 wave-stores (lane l: q = 4l..4l+3, then
// q = 256+4l..; lane 26 covers 357..360, re-writing 3 floats with the same values), so a board's
// 36,100 B leave the wave as ascending contiguous runs -- 45 store instructions instead of 150
// dword stores, and 1.6x the HBM rate of point-major dword stores at 65,536 boards.
// `lut` = host-built (float)exp(-d/10.0) table (board_feature.cc:141).
__device__ __forceinline__ void write_features(WaveLds& L, const Geo& g, int lane, const uint32_t lab[KCH],
                                               const uint32_t lastp[KCH], const Scalars& s, int aug,
                                               const float* __restrict__ lut, float* __restrict__ out) {
  // history values first: their global-load latency hides under the distance transform
  float hv[KCH];
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    hv[k] = 0.f;
    if (g.p[k] < NPTS && lab[k]) {
      int age = s.ply - (int)lastp[k];
      age = age < 0 ? 0 : (age > HIST_LUT_N - 1 ? HIST_LUT_N - 1 : age);
      hv[k] = lut[age];
    }
  }
  const int has = distance_maps(L, lane, lab);
  const int ko_p = (s.koage == 0 && s.ko >= 0) ? s.ko : -1;  // GetSimpleKoLocation, board.cc:337-344
  // per-point descriptor, scattered to its OUTPUT index (D4 applied here):
  //   colour(2) | cls(2) | ko(1) | d_black(7) | d_white(7);  history value in a parallel table
  float* hist = reinterpret_cast<float*>(L.lib);  // the liberty table is dead from here on
  GO_UNROLL
  for (int k = 0; k < KCH; ++k) {
    if (g.p[k] < NPTS) {
      uint32_t d = typ_of(lab[k]);  // cls 1,2,3 <-> GetLibertyMap3binary :83-99
      if (g.p[k] == ko_p) d |= 16u;
      d |= (uint32_t)L.dist[0][g.p[k]] << 5;
      d |= (uint32_t)L.dist[1][g.p[k]] << 12;
      const int q = aug ? d4_dest(aug, g.p[k]) : g.p[k];
      L.desc[q] = d;
      hist[q] = hv[k];
    }
  }
  WSYNC();
  const uint32_t our = (uint32_t)s.next, opp = 3u - our;
  const bool has_our = (has >> (our - 1)) & 1, has_opp = (has >> (opp - 1)) & 1;
  const int sh_our = our == 1 ? 5 : 12, sh_opp = our == 1 ? 12 : 5;
  const int qa = 4 * lane;
  const int qb = lane < 26 ? 256 + 4 * lane : NPTS - 4;
  uint32_t d[8];
  float h[8];
  GO_UNROLL
  for (int i = 0; i < 4; ++i) {
    d[i] = L.desc[qa + i];
    h[i] = hist[qa + i];
    d[4 + i] = L.desc[qb + i];
    h[4 + i] = hist[qb + i];
  }
  const bool act_b = lane <= 26;

#define GO_PLANE(pl, EXPR)                                                          \
  do {                                                                              \
    float v_[8];                                                                    \
    GO_UNROLL                                                                       \
    for (int i = 0; i < 8; ++i) {                                                   \
      const uint32_t dd = d[i];                                                     \
      const uint32_t col = dd & 3u, cls = (dd >> 2) & 3u;                           \
      const bool mine = col == our, theirs = col == opp;                            \
      const float hh = h[i];                                                        \
      (void)cls; (void)mine; (void)theirs; (void)hh;                                \
      v_[i] = (EXPR);                                                               \
    }                                                                               \
    const float4u a_ = {v_[0], v_[1], v_[2], v_[3]};                                \
    const float4u b_ = {v_[4], v_[5], v_[6], v_[7]};                                \
    *reinterpret_cast<float4u*>(out + (pl) * NPTS + qa) = a_;                       \
    if (act_b) *reinterpret_cast<float4u*>(out + (pl) * NPTS + qb) = b_;            \
  } while (0)

  GO_PLANE(0, (mine && cls == 1) ? 1.f : 0.f);
  GO_PLANE(1, (mine && cls == 2) ? 1.f : 0.f);
  GO_PLANE(2, (mine && cls != 1 && cls != 2) ? 1.f : 0.f);  // literally "else", board_feature.cc:93
  GO_PLANE(3, (theirs && cls == 1) ? 1.f : 0.f);
  GO_PLANE(4, (theirs && cls == 2) ? 1.f : 0.f);
  GO_PLANE(5, (theirs && cls != 1 && cls != 2) ? 1.f : 0.f);
  GO_PLANE(6, (dd & 16u) ? 1.f : 0.f);
  GO_PLANE(7, mine ? 1.f : 0.f);
  GO_PLANE(8, theirs ? 1.f : 0.f);
  GO_PLANE(9, col == 0 ? 1.f : 0.f);
  GO_PLANE(10, mine ? hh : 0.f);
  GO_PLANE(11, theirs ? hh : 0.f);
  zero_range(out + 12 * NPTS, 2 * NPTS, lane);               // BORDER, POSITION_MARK: never filled
  GO_PLANE(14, has_our ? (float)((dd >> sh_our) & 127u) : 10000.f);
  GO_PLANE(15, has_opp ? (float)((dd >> sh_opp) & 127u) : 10000.f);
  zero_range(out + 16 * NPTS, 9 * NPTS, lane);               // planes 16..24: never filled
#undef GO_PLANE
}

__device__ __forceinline__ void write_legal(const Geo& g, int lane, const Masks& m, uint8_t* __restrict__ out) {
  GO_UNROLL
  for (int k = 0; k < KCH; ++k)
    if (g.p[k] < NPTS) out[g.p[k]] = (uint8_t)((m.legal[k] >> lane) & 1ull);
}

}  // namespace goenv
