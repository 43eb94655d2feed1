// goenv_tactics.h -- the heuristic move filters of go/board.cc for the side to move, one
// wavefront per board:
//
//   IsSelfAtari            go/board.cc:168-200     self_atari[p] = stones of the group in atari, 0 = no
//   IsMoveGivingSimpleKo   go/board.cc:320-335     gives_ko[p]
//   FindAllCandidateMoves  go/board.cc:685-715     cand[p] (legal, no own true eye, no self-atari of
//                                                  >= thres stones)
//   find_only_liberty / find_two_liberties  :1042-1081   libs[p][2] of the group of the stone at p
//   CheckLadder / CheckLadderUseSearch      :203-317,348-391   ladder[p] = depth, 0 = none
//
// The reference answers the first and the last by copying the board and playing the move.  Here:
//
// * Self-atari is decided for all 361 points at once from three per-group tables built with LDS
//   atomics (stone count, smallest and largest liberty point -- for a group of <= 2 liberties
//   these ARE its liberties): the liberties of the group a move would form are the move's empty
//   neighbours, the other liberty of each own neighbour group of two, and the stones it
//   captures next to it.  Only one shape needs a look beyond the neighbourhood (a capture of a
//   multi-stone group by a move with no other liberty, whose stones may touch the mover's other
//   groups); those points are settled one at a time by a wave-wide scan.
// * A ladder is a forced line, so it IS played out -- on the LDS image, with the same apply_move
//   as the step kernel -- and the rare two-way branch keeps its alternative on a small LDS stack
//   instead of a recursive board copy.
#pragma once

#include "goenv_device.h"

namespace goenv {

constexpr int LADDER_FRAMES = 16;      // pending alternatives (measured maximum on game records: 4)
constexpr int MAX_LADDER_SEARCH = 1024;  // go/board.cc:202

struct LadderFrame {
  uint16_t lab[TAB_N];
  int32_t sc[12];  // next, l1, l2, l3, l4, ko, koage, kocolor, ply, move, depth, -
};

struct TacticsLds {
  uint32_t lmin[TAB_N + 64];
  uint32_t lmax[TAB_N + 64];
  uint32_t stones[TAB_N + 64];
  LadderFrame stack[LADDER_FRAMES];
};

struct TacticsOut {
  int16_t* self_atari;  // [n][361]
  uint8_t* gives_ko;    // [n][361]
  uint8_t* cand;        // [n][361]
  int16_t* libs;        // [n][361][2]
  int16_t* ladder;      // [n][361]
};

// point index of neighbour i (L,T,R,B) of point p; only meaningful when that neighbour is on the board
__device__ __forceinline__ int nbr_point(int p, int i) {
  return i == 0 ? p - BS : (i == 1 ? p - 1 : (i == 2 ? p + BS : p + 1));
}

// uniform: first / all empty neighbours of point q, in L,T,R,B order
__device__ __forceinline__ int empty_nbrs(const BoardLds<1>& L, int q, int out[4]) {
  const int c = pad_of(q);
  uint32_t n[4];
  nbr4(L, c, n);
  int k = 0;
  GO_UNROLL
  for (int i = 0; i < 4; ++i) {
    const bool e = __builtin_amdgcn_readfirstlane(n[i]) == 0u;
    if (e) out[k++] = nbr_point(q, i);
  }
  return k;
}

__device__ __forceinline__ void restore_image(BoardLds<1>& L, const Geo<1>& g, const uint32_t lab[]) {
  GO_UNROLL
  for (int k = 0; k < Geo<1>::K; ++k)
    if (g.p[k] < NPTS) L.lab[g.c[k]] = (uint16_t)lab[k];
  wsync<1>();
}

// CheckLadderUseSearch (go/board.cc:203-317) as a loop: `lab`/L hold the position AFTER the
// victim's first move.  Returns the depth (> 0), 0 for "no ladder", -1 when more than
// LADDER_FRAMES alternatives were pending (the search is then abandoned).
__device__ __forceinline__ int ladder_search(BoardLds<1>& L, TacticsLds& X, const Geo<1>& g, uint32_t lab[],
                                             Scalars& s, int victim) {
  constexpr int K = Geo<1>::K;
  uint32_t dummy[K];
  int num_call = 0, depth = 1, sp = 0;
  for (;;) {  // every iteration plays one stone or ends the search
    // hard bound every wave reaches: captures can re-open points, so the board alone does not bound
    // the line; past 4 x 361 stones the search is abandoned with the same code as a full stack
    if (++num_call > 4 * NPTS) return -1;
    const int c = s.l1, c2 = s.l2;
    const uint32_t lib = cls_of(__builtin_amdgcn_readfirstlane((uint32_t)L.lab[pad_of(c)]));
    int mv = -1;        // the stone to play on this path, -1 = this path fails
    bool victim_moves = false;
    if (victim == 3 - s.next) {  // capturer to play
      if (lib == 1) return depth;
      if (lib == 2) {
        int esc[4];
        const int n_esc = empty_nbrs(L, c, esc);
        if (n_esc > 1) {
          int must = -1, tmp[4];
          if (empty_nbrs(L, esc[0], tmp) == 3) must = esc[0];
          else if (empty_nbrs(L, esc[1], tmp) == 3) must = esc[1];
          if (must < 0 && num_call >= MAX_LADDER_SEARCH) must = esc[0];
          if (must >= 0) {
            mv = must;
          } else {
            if (sp == LADDER_FRAMES) return -1;
            LadderFrame& f = X.stack[sp++];
            GO_UNROLL
            for (int k = 0; k < K; ++k)
              if (g.p[k] < NPTS) f.lab[g.p[k]] = (uint16_t)lab[k];
            if (g.lane == 0) {
              f.sc[0] = s.next; f.sc[1] = s.l1; f.sc[2] = s.l2; f.sc[3] = s.l3; f.sc[4] = s.l4;
              f.sc[5] = s.ko; f.sc[6] = s.koage; f.sc[7] = s.kocolor; f.sc[8] = s.ply;
              f.sc[9] = esc[1]; f.sc[10] = depth;
            }
            wsync<1>();
            mv = esc[0];
          }
        }
      }
    } else {  // victim to play
      if (lib != 1) {
        int esc[4];
        if (c2 >= 0 && empty_nbrs(L, c2, esc) > 0) { mv = esc[0]; victim_moves = true; }
      }
    }
    for (;;) {  // play mv; on failure fall back to the most recent pending alternative
      bool ok = mv >= 0 && move_is_legal(L, s, mv, s.next);
      if (ok) {
        GO_UNROLL
        for (int k = 0; k < K; ++k) dummy[k] = 0;
        apply_move<1>(L, g, lab, dummy, s, mv, s.next);
        if (victim_moves) {  // go/board.cc:299-311
          const int cf = pad_of(mv);
          const uint32_t l = cls_of(__builtin_amdgcn_readfirstlane((uint32_t)L.lab[cf]));
          if (l >= 3) ok = false;
          if (l == 2) {
            uint32_t n[4];
            nbr4(L, cf, n);
            GO_UNROLL
            for (int i = 0; i < 4; ++i) {
              const uint32_t v = __builtin_amdgcn_readfirstlane(n[i]);
              if (col_of(v) == (uint32_t)(3 - victim) && cls_of(v) == 1) ok = false;
            }
          }
        }
      }
      if (ok) { ++depth; break; }
      if (sp == 0) return 0;
      const LadderFrame& f = X.stack[--sp];
      GO_UNROLL
      for (int k = 0; k < K; ++k) lab[k] = g.p[k] < NPTS ? (uint32_t)f.lab[g.p[k]] : 0u;
      s.next = __builtin_amdgcn_readfirstlane(f.sc[0]); s.l1 = __builtin_amdgcn_readfirstlane(f.sc[1]);
      s.l2 = __builtin_amdgcn_readfirstlane(f.sc[2]); s.l3 = __builtin_amdgcn_readfirstlane(f.sc[3]);
      s.l4 = __builtin_amdgcn_readfirstlane(f.sc[4]); s.ko = __builtin_amdgcn_readfirstlane(f.sc[5]);
      s.koage = __builtin_amdgcn_readfirstlane(f.sc[6]); s.kocolor = __builtin_amdgcn_readfirstlane(f.sc[7]);
      s.ply = __builtin_amdgcn_readfirstlane(f.sc[8]);
      mv = __builtin_amdgcn_readfirstlane(f.sc[9]);
      depth = __builtin_amdgcn_readfirstlane(f.sc[10]);
      victim_moves = false;
      restore_image(L, g, lab);
    }
  }
}

// All five filters of one board.  `lab`/L = the loaded position (cls bits valid), `o` = this
// position's legal / eye ballots for the side to move.  `row` = output row.
__device__ __forceinline__ void tactics(BoardLds<1>& L, TacticsLds& X, const Geo<1>& g, uint32_t lab[], Scalars& s,
                                        const Own<1>& o, int thres, const TacticsOut& out, size_t row) {
  constexpr int K = Geo<1>::K;
  const int lane = g.lane;
  const uint32_t P = (uint32_t)s.next, O = 3u - P;
  // --- group tables
  for (int i = g.tid; i < TAB_N + 64; i += 64) { X.lmin[i] = 0xffffu; X.lmax[i] = 0u; X.stones[i] = 0u; }
  wsync<1>();
  const int sink = TAB_N + g.tid;
  GO_UNROLL
  for (int k = 0; k < K; ++k) {
    if (g.p[k] >= NPTS) continue;
    if (lab[k] == 0) {
      uint32_t n[4];
      nbr4(L, g.c[k], n);
      int r0 = is_stone(n[0]) ? (int)(n[0] & ROOT_MASK) : sink;
      int r1 = is_stone(n[1]) ? (int)(n[1] & ROOT_MASK) : sink;
      int r2 = is_stone(n[2]) ? (int)(n[2] & ROOT_MASK) : sink;
      int r3 = is_stone(n[3]) ? (int)(n[3] & ROOT_MASK) : sink;
      const uint32_t p = (uint32_t)g.p[k];
      atomicMin(&X.lmin[r0], p); atomicMax(&X.lmax[r0], p);
      atomicMin(&X.lmin[r1], p); atomicMax(&X.lmax[r1], p);
      atomicMin(&X.lmin[r2], p); atomicMax(&X.lmax[r2], p);
      atomicMin(&X.lmin[r3], p); atomicMax(&X.lmax[r3], p);
    } else {
      atomicAdd(&X.stones[lab[k] & ROOT_MASK], 1u);
    }
  }
  wsync<1>();
  // --- per point
  int sa[K];
  bool slow[K], lad[K];
  GO_UNROLL
  for (int k = 0; k < K; ++k) {
    sa[k] = 0; slow[k] = false; lad[k] = false;
    const int p = g.p[k];
    const bool valid = p < NPTS;
    const bool legal = (o.legal[k] >> lane) & 1ull;
    uint32_t n[4];
    nbr4(L, g.c[k], n);
    int e_cnt = 0, e_pt = -1, n_own = 0, n_opp = 0, dn = 0, d_root = -1;
    int own_r[4], other[4];
    bool big = false, own_atari = false, opp_three = false;
    int k_own = 0, sum_stones = 1, ko_hits = 0;
    GO_UNROLL
    for (int i = 0; i < 4; ++i) {
      own_r[i] = -1; other[i] = -1;
      const uint32_t v = n[i], col = col_of(v), cls = cls_of(v);
      const int r = (int)(v & ROOT_MASK);
      if (v == 0) { ++e_cnt; e_pt = nbr_point(p, i); }
      if (col == P) {
        ++n_own;
        own_atari = cls == 1;
        bool dup = false;
        GO_UNROLL
        for (int j = 0; j < i; ++j) dup |= own_r[j] == r;
        if (!dup) {
          own_r[i] = r;
          ++k_own;
          sum_stones += (int)X.stones[r];
          if (cls == 3) big = true;
          if (cls == 2) { const int a = (int)X.lmin[r], b = (int)X.lmax[r]; other[i] = a == p ? b : a; }
        }
      }
      if (col == O) {
        ++n_opp;
        opp_three = cls == 3;
        if (cls == 1) { ++dn; d_root = r; if (X.stones[r] == 1u) ++ko_hits; }
      }
    }
    // distinct liberties the new group has without captures: the empty neighbour (if one) and
    // the other liberty of every own neighbour group of two
    int u_cnt = e_cnt == 1 ? 1 : 0;
    GO_UNROLL
    for (int i = 0; i < 4; ++i) {
      if (other[i] < 0) continue;
      bool dup = e_cnt == 1 && other[i] == e_pt;
      GO_UNROLL
      for (int j = 0; j < i; ++j) dup |= other[j] == other[i];
      if (!dup) ++u_cnt;
    }
    if (valid && legal && e_cnt < 2 && !big) {
      const int total = u_cnt + dn;
      if (total == 1) {
        sa[k] = sum_stones;
        slow[k] = dn == 1 && k_own > 0 && X.stones[d_root < 0 ? 0 : d_root] > 1u;
      }
    }
    if (out.gives_ko && valid) out.gives_ko[row * NPTS + p] = (uint8_t)(legal && e_cnt == 0 && n_own == 0 && ko_hits == 1);
    lad[k] = valid && legal && e_cnt == 2 && n_own == 1 && own_atari && n_opp == 1 && opp_three;
    if (out.libs && valid) {
      int a = -1, b = -1;
      if (lab[k] != 0) {
        const uint32_t cls = cls_of(lab[k]);
        const int r = (int)(lab[k] & ROOT_MASK);
        if (cls == 1) a = (int)X.lmin[r];
        if (cls == 2) { a = (int)X.lmin[r]; b = (int)X.lmax[r]; }
      }
      out.libs[(row * NPTS + p) * 2] = (int16_t)a;
      out.libs[(row * NPTS + p) * 2 + 1] = (int16_t)b;
    }
  }
  // --- the one non-local shape: the move captures a multi-stone group D through a single
  // neighbour stone and has no other liberty; stones of D next to the mover's neighbour groups
  // become liberties as well.  Settled point by point (rare).
  GO_UNROLL
  for (int k = 0; k < K; ++k) {
    unsigned long long todo = __ballot(slow[k]);
    while (todo) {  // one bit cleared per iteration
      const int bit = (int)__builtin_ctzll(todo);
      todo &= todo - 1;
      const int p0 = 64 * k + bit, c0 = pad_of(p0);
      uint32_t n[4];
      nbr4(L, c0, n);
      int G[4], rd = -1, d_pt = -1;
      GO_UNROLL
      for (int i = 0; i < 4; ++i) {
        const uint32_t v = __builtin_amdgcn_readfirstlane(n[i]);
        G[i] = col_of(v) == P ? (int)(v & ROOT_MASK) : -1;
        if (col_of(v) == O && cls_of(v) == 1) { rd = (int)(v & ROOT_MASK); d_pt = nbr_point(p0, i); }
      }
      bool hit = false;
      GO_UNROLL
      for (int kk = 0; kk < K; ++kk) {
        const uint32_t v = lab[kk];
        if (g.p[kk] < NPTS && col_of(v) == O && (int)(v & ROOT_MASK) == rd && g.p[kk] != d_pt) {
          uint32_t m[4];
          nbr4(L, g.c[kk], m);
          GO_UNROLL
          for (int i = 0; i < 4; ++i) {
            const int r = (int)(m[i] & ROOT_MASK);
            if (col_of(m[i]) == P && (r == G[0] || r == G[1] || r == G[2] || r == G[3])) hit = true;
          }
        }
      }
      if (__any(hit) && lane == bit) sa[k] = 0;
    }
  }
  GO_UNROLL
  for (int k = 0; k < K; ++k) {
    const int p = g.p[k];
    if (p >= NPTS) continue;
    if (out.self_atari) out.self_atari[row * NPTS + p] = (int16_t)sa[k];
    if (out.cand) {
      const bool legal = (o.legal[k] >> lane) & 1ull, eye = (o.eye[k] >> lane) & 1ull;
      out.cand[row * NPTS + p] = (uint8_t)(legal && !eye && !(sa[k] > 0 && sa[k] >= thres));
    }
  }
  // --- ladders: CheckLadder's precondition (go/board.cc:350-377) is local; the search is played out
  if (!out.ladder) return;
  uint32_t lab0[K];
  GO_UNROLL
  for (int k = 0; k < K; ++k) lab0[k] = lab[k];
  const Scalars s0 = s;
  int depth_out[K];
  GO_UNROLL
  for (int k = 0; k < K; ++k) {
    depth_out[k] = 0;
    unsigned long long todo = __ballot(lad[k]);
    while (todo) {
      const int bit = (int)__builtin_ctzll(todo);
      todo &= todo - 1;
      const int p0 = 64 * k + bit;
      uint32_t dummy[K];
      GO_UNROLL
      for (int kk = 0; kk < K; ++kk) dummy[kk] = 0;
      apply_move<1>(L, g, lab, dummy, s, p0, (int)P);
      const int d = ladder_search(L, X, g, lab, s, (int)P);
      if (lane == bit) depth_out[k] = d;
      GO_UNROLL
      for (int kk = 0; kk < K; ++kk) lab[kk] = lab0[kk];
      s = s0;
      restore_image(L, g, lab);
    }
  }
  GO_UNROLL
  for (int k = 0; k < K; ++k)
    if (g.p[k] < NPTS) out.ladder[row * NPTS + g.p[k]] = (int16_t)depth_out[k];
}

}  // namespace goenv
