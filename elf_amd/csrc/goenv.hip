// goenv.hip -- kernels and C-ABI of libgoenv.so (see include/goenv.h).
//
// Launch shape: one workgroup of NW wavefronts per board (NW in {1,2,3,6}, goenv_device.h), so a
// batch of B boards is B workgroups.  NW is chosen per launch from the batch size (small batches
// are latency-bound and take more waves per board; large ones are HBM-bound and take one) or
// pinned with goenv_set_waves_per_board.  All kernels are stream-ordered on the caller's stream.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/goenv.h"
#include "goenv_device.h"
#include "gen_tune.h"
#include "goenv_tactics.h"

using namespace goenv;

// ----------------------------------------------------------------------------------------------
// kernels
// ----------------------------------------------------------------------------------------------

struct PlyParams {
  BoardRec* state;
  int n;            // rows of this launch
  int n_total;      // boards in the context
  const int32_t* idx;  // board of row b = idx ? idx[b] : first + b
  int first;
  goenv_sample_out meta;
  uint32_t flags;
  const int64_t* actions;
  const float* pi;         // MODE 2: policy weights [n][361], row b in the frame of pi_aug[b]
  const int32_t* pi_aug;
  const int32_t* aug;
  float* feat;
  uint8_t* legal;
  int32_t* actions_out;
  int32_t* status;
  unsigned long long* stats;
  const float* lut;
  unsigned long long* stamps;  // diagnostic build only (GOENV_STAMPS): s_memtime per phase
  uint64_t seed;
  int64_t id_base;
  float komi;
  int max_ply;
  float lut1;       // lut[1]
  int zero_first;   // constant planes are stored at kernel start (small batches)
  int xcd_map;      // rows dealt to workgroups so that each XCD owns a contiguous run (xcd_row)
};

#ifdef GOENV_STAMPS
#define STAMP(i)                                                                   \
  do {                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                             \
    unsigned long long t_ = __builtin_amdgcn_s_memtime();                          \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                            \
    __builtin_amdgcn_sched_barrier(0);                                             \
    if (P.stamps && (threadIdx.x & 63) == 0 && threadIdx.x < 128)                  \
      P.stamps[(size_t)blockIdx.x * 32 + (threadIdx.x >> 6) * 16 + (i)] = t_;      \
  } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// End-of-game accounting into the stats vector; thread 0 issues the atomics.
__device__ __forceinline__ void account_game(unsigned long long* stats, int tid, const Scalars& s,
                                             int score, float komi) {
  if (tid == 0) {
    const unsigned long long len = (unsigned long long)(s.ply - 1);
    atomicAdd(&stats[GOENV_ST_GAMES], 1ull);
    const float margin = (float)score - komi;
    if (margin > 0.f) atomicAdd(&stats[GOENV_ST_BLACK_WINS], 1ull);
    else if (margin < 0.f) atomicAdd(&stats[GOENV_ST_WHITE_WINS], 1ull);
    else atomicAdd(&stats[GOENV_ST_DRAWS], 1ull);
    atomicAdd(&stats[GOENV_ST_PLIES], len);
    atomicAdd(&stats[GOENV_ST_BCAP], (unsigned long long)s.bcap);
    atomicAdd(&stats[GOENV_ST_WCAP], (unsigned long long)s.wcap);
    atomicAdd(&stats[GOENV_ST_SUM_LEN2], len * len);
    if (s.done == 2) atomicAdd(&stats[GOENV_ST_PLY_CAP_HITS], 1ull);
    atomicMax(&stats[GOENV_ST_MAX_LEN], len);
    atomicAdd(&stats[GOENV_ST_SUM_SCORE], (unsigned long long)(long long)score);
  }
}

// Row of a launch that workgroup w takes.  Workgroups are dealt to the 8 XCDs round-robin (w % 8), and a
// kernel's output rows are neighbours in memory (planes, mask: 36,461 B per row).  With row = w every XCD
// writes every eighth row; with this map XCD x owns one CONTIGUOUS run of rows, [x n/8, (x+1) n/8) when 8
// divides n.  Same bytes, same stores per workgroup -- but the chip-wide write stream then consists of 8
// dense regions instead of 8 interleaved combs, which the memory side takes 7 % faster (pure-store probe,
// tools/store_bench.hip run_map: 5.10 -> 5.49 TB/s at 1,024 boards, 4.87 -> 5.22 TB/s at 8,192).  A
// bijection on [0, n) for every n.
__device__ __forceinline__ int xcd_row(int w, int n, int on) {
  if (!on) return w;
  const int x = w & 7, k = w >> 3, q = n >> 3, r = n & 7;
  return x * q + (x < r ? x : r) + k;
}

// Board of launch row b (selection by index list or by contiguous range); -1 if out of range.
__device__ __forceinline__ int board_of(const int32_t* idx, int first, int n_total, int b) {
  const int board = idx ? idx[b] : first + b;
  return (board >= 0 && board < n_total) ? board : -1;
}

// A game restarts in place: empty board, next game of this slot, seq 0 (AICommT::Restart ->
// SeqInfo::NewEpisode, elf/ai_comm.h:91-95, elf/common.h:18).
template <int NW>
__device__ __forceinline__ void restart_in_place(BoardLds<NW>& L, const Geo<NW>& g, uint32_t lab[], uint32_t lastp[],
                                                 Scalars& s, Masks& m, Own<NW>& o) {
  clear_board_lds(L, g, lab, lastp);
  const int game = s.game + 1, seqtot = s.seqtot;
  clear_scalars(s);
  s.game = game;
  s.seqtot = seqtot;
  full_masks(g, m, o);
}

// One sample's sequencing record (GameState::Prepare, go/go_game_specific.h:91-98 +
// SeqInfo, elf/common.h:11-19): written for row b, then the board's seq advances.
__device__ __forceinline__ void emit_sample_meta(const goenv_sample_out& o, int b, int board, int tid,
                                                 Scalars& s, int aug) {
  if (tid == 0) {
    const bool run_on = o.mode == GOENV_SEQ_RUN_ON;
    if (o.id) o.id[b] = board;
    if (o.seq) o.seq[b] = run_on ? s.seqtot : s.seq;
    if (o.game_counter) o.game_counter[b] = run_on ? 0 : s.game;
    if (o.last_terminal) o.last_terminal[b] = (int8_t)((!run_on && s.seq == 0 && s.game > 0) ? 1 : 0);
    if (o.move_idx) o.move_idx[b] = s.ply;  // gs.move_idx = state.GetPly(), go/go_ai.h:82
    if (o.winner) o.winner[b] = 0;
    if (o.aug_code) o.aug_code[b] = aug;
    if (o.game_record_idx) o.game_record_idx[b] = -1;
  }
  s.seq += 1;
  s.seqtot += 1;
}

// Inclusive prefix sum over the 64 lanes of a wave with DPP adds (full-rate VALU; __shfl_up would be
// six LDS-crossbar permutes): row_shr 1/2/4/8 inside each row of 16 lanes, then row_bcast:15 into
// rows 1 and 3 and row_bcast:31 into rows 2 and 3.  Lanes without a source add 0.
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
  return v;
}

// Move selection from a policy head: sample p proportional to pi[p] over the LEGAL points, in exact
// integer arithmetic (weights floor(pi * 2^24), r = floor(u * total / 2^64) with the counter RNG's u,
// the point whose running sum first exceeds r); all-zero weights fall back to the cfg-2 random
// policy.  ONE wave does it for its board (lane l weighs points l + 64 j): with several waves per
// board the others wait at the caller's barrier -- sharing the six words between the waves was
// measured slower (every wave then pays the uniform part, and the masks live in scalar registers
// that the kernel is short of: 83 us against 55 us per ply at 8,192 boards).
// pv[j] = pi of point lane + 64 j in the frame the caller indexed it in (policy_preload).
__device__ __forceinline__ void policy_preload(int lane, const float* __restrict__ row, int aug, float pv[NWORDS]) {
  GO_UNROLL
  for (int j = 0; j < NWORDS; ++j) {
    const int p = lane + 64 * j;
    pv[j] = p < NPTS ? row[aug ? d4_dest(aug, p) : p] : 0.f;
  }
}

__device__ __forceinline__ int pick_policy_move(const Masks& m, int lane, const Scalars& s, const float pv[NWORDS],
                                                uint64_t seed, uint64_t board_id) {
  uint32_t pre[NWORDS], sum[NWORDS];
  unsigned long long total = 0;
  GO_UNROLL
  for (int j = 0; j < NWORDS; ++j) {
    uint32_t w = 0;
    if ((m.legal[j] >> lane) & 1ull) {   // bits beyond point 360 are never set
      const float f = pv[j];
      if (f > 0.f) {
        const float sc = f * 16777216.0f;
        w = sc >= 16777216.0f ? 16777216u : (uint32_t)sc;
      }
    }
    pre[j] = wave_scan_add(w);
    sum[j] = (uint32_t)__builtin_amdgcn_readlane((int)pre[j], 63);
    total += sum[j];
  }
  if (total == 0) return pick_random_move(m, lane, s, seed, board_id);
  const uint64_t u = policy_rand(seed ^ 0xA5A5A5A55A5A5A5Aull, board_id, (uint64_t)s.game, (uint64_t)s.ply);
  unsigned long long r = __umul64hi(u, total);
  int a = -1;
  GO_UNROLL
  for (int j = 0; j < NWORDS; ++j) {
    if (a < 0) {
      if (r < sum[j]) {
        const unsigned long long hit = __ballot((unsigned long long)pre[j] > r);
        a = 64 * j + (int)__builtin_ctzll(hit);
      } else {
        r -= sum[j];
      }
    }
  }
  return a;
}

// The sampler as a FUNCTION CALL from the ply kernel, with the masks and the two scalars it needs read
// again from the record (scalar loads that hit the cache: load_record has just fetched them).  Inlined,
// it shares the kernel's scalar registers, which the kernel has none to spare of: the allocator then
// spills them through VGPR lanes all over the plane stores (971 v_readlane against 226 in the ISA, 3,200
// against 2,530 vector instructions per board-ply, 82 us against 55 us per ply at 8,192 boards).
__device__ __attribute__((noinline)) int sample_move_of_record(const BoardRec* rec, float p0, float p1, float p2,
                                                               float p3, float p4, float p5, uint64_t seed,
                                                               uint64_t board_id) {
  Masks m;
  load_masks(rec, m);
  Scalars s;
  s.game = rec->sc[SC_GAME];
  s.ply = rec->sc[SC_PLY];
  const float pv[NWORDS] = {p0, p1, p2, p3, p4, p5};
  return pick_policy_move(m, (int)(threadIdx.x & 63), s, pv, seed, board_id);
}

// One ply for one board held in LDS/registers.  `m` = masks of the position (all words), `o`
// receives this wave's words of the NEW position's masks.  Returns the action applied (-1 pass)
// or -4 if the external action was illegal (state and `o` untouched).
template <int NW, bool RANDOM, bool MASKS = true>
__device__ __forceinline__ int ply_body(BoardLds<NW>& L, const Geo<NW>& g, uint32_t lab[], uint32_t lastp[],
                                        Scalars& s, const Masks& m, Own<NW>& o, const PlyParams& P, int board,
                                        long long av) {
  int a;
  if (RANDOM) {
    a = pick_random_move(m, g.lane, s, P.seed, (uint64_t)(P.id_base + board));
    a = __builtin_amdgcn_readfirstlane(a);
    STAMP(2);
  } else {
    a = (av < -2 || av >= NPTS) ? -4 : (int)av;
    a = __builtin_amdgcn_readfirstlane(a);
    if (a >= 0) {  // TryPlay: the stored legal ballot of the side to move
      unsigned long long w = 0;
      GO_UNROLL
      for (int j = 0; j < NWORDS; ++j)
        if ((a >> 6) == j) w = m.legal[j];
      if (!((w >> (a & 63)) & 1ull)) a = -4;
    }
    if (a == -4) return a;
  }
#ifdef GOENV_ABLATE_RULES
  advance(s, a, s.next);
#else
  if (a < 0) {
    advance(s, a, s.next);  // pass / resign: Play's early return, go/board.cc:1092-1095
  } else {
    apply_move(L, g, lab, lastp, s, a, s.next);
  }
#endif
  STAMP(3);
  if (MASKS) move_masks(L, g, lab, s, s.next, o);
  s.done = game_end(s) ? 1 : ((P.max_ply > 0 && s.ply > P.max_ply) ? 2 : 0);
  s.counted = 0;
  return a;
}

// MODE 0: the move is actions[b]; 1: the on-device cfg-2 policy; 2: sampled from pi[b] over the legal
// points (pick_policy_move), after any restart of a finished board, so one launch is a whole
// selfplay step: reply -> move -> rules -> next sample's mask, planes and sequencing record.
template <int NW, int MODE>
__global__ __launch_bounds__(64 * NW) void k_ply(PlyParams P) {
  constexpr bool RANDOM = MODE == 1;
  __shared__ BoardLds<NW> L;
  if ((int)blockIdx.x >= P.n) return;
  const int b = xcd_row(blockIdx.x, P.n, P.xcd_map);
  Geo<NW> g;
  geo_init(g);
  constexpr int K = Geo<NW>::K;
  const int board = board_of(P.idx, P.first, P.n_total, b);
  if (board < 0) {
    if (g.tid == 0 && P.status) P.status[b] = -2;
    return;
  }
  STAMP(0);
#ifdef GOENV_ABLATE_ALL
  return;
#endif
  BoardRec* rec = P.state + board;
  // the row's inputs are requested together with the record: one trip to memory, not one per use
  const int aug = P.aug ? (P.aug[b] & 7) : 0;
  long long av = 0;
  if (MODE == 0) av = (long long)P.actions[b];
  float pv[NWORDS];
  if (MODE == 2 && g.wave == 0) policy_preload(g.lane, P.pi + (size_t)b * NPTS, P.pi_aug ? (P.pi_aug[b] & 7) : 0, pv);
  Scalars s;
  uint32_t lab[K], lastp[K];
  Masks m;
  Own<NW> o;
  load_record(rec, L, g, lab, lastp, s, m);
  own_from_masks(g, m, o);
  float hv_pre[K];
  const int pre_ply = s.ply + 1;  // every accepted move, pass included, advances the ply by one
  if (P.feat) {
    history_preload(g, lastp, pre_ply, P.lut, hv_pre);
    if ((P.zero_first & 5) == 1) {
      if (P.zero_first & 2) zero_planes<NW, true>(g, P.feat + (size_t)b * GOENV_FEAT_FLOATS);
      else zero_planes<NW, false>(g, P.feat + (size_t)b * GOENV_FEAT_FLOATS);
    }
  }
  STAMP(1);

  bool skip = false, restarted = false;
  if (s.done) {
    if (P.flags & GOENV_F_AUTO_RESTART) {
      if (!s.counted) account_game(P.stats, g.tid, s, tromp_taylor(L, g, lab), P.komi);
      restart_in_place(L, g, lab, lastp, s, m, o);
      restarted = true;  // the record must be stored even if the move is then rejected
    } else if (P.flags & GOENV_F_SKIP_DONE) {
      skip = true;
    }
  }
  int a = -3, status = -1;
  if (!skip) {
    // the masks of the new position are computed AFTER the planes have been issued: the plane
    // stores then drain to HBM while the workgroup finishes its mask / record tail
    if (MODE == 2) {  // wave 0 samples; the other waves of the board take the move from LDS
      int pa = -1;
      if (g.wave == 0)
        pa = sample_move_of_record(rec, pv[0], pv[1], pv[2], pv[3], pv[4], pv[5], P.seed, (uint64_t)(P.id_base + board));
      if (NW > 1) {
        if (g.tid == 0) L.xflag[3] = (uint32_t)pa;
        WSYNC();
        pa = (int)L.xflag[3];
      }
      av = __builtin_amdgcn_readfirstlane(pa);
      STAMP(2);
    }
    a = ply_body<NW, RANDOM, false>(L, g, lab, lastp, s, m, o, P, board, av);
    status = (a == -4) ? 0 : 1;
    bool fresh = false;  // restarted: `o` already holds the masks of the empty board
    if (a == -4 && (P.flags & GOENV_F_RESET_ILLEGAL)) {
      // "No valid move ..., restarting the game", go/game.cc:81-90
      restart_in_place(L, g, lab, lastp, s, m, o);
      fresh = true;
    } else if (a != -4 && s.done && (P.flags & GOENV_F_RESTART_AFTER_END)) {
      account_game(P.stats, g.tid, s, tromp_taylor(L, g, lab), P.komi);
      restart_in_place(L, g, lab, lastp, s, m, o);
      status = 2;
      fresh = true;
    }
    STAMP(4);
    const int ply_of_sample = s.ply;
    if (P.flags & GOENV_F_COLLECT) emit_sample_meta(P.meta, b, board, g.tid, s, aug);
    (void)ply_of_sample;
#ifdef GOENV_ABLATE_FEAT
    if (false) {
#else
    if (P.feat) {
#endif
      float* out = P.feat + (size_t)b * GOENV_FEAT_FLOATS;
      switch (P.zero_first & 3) {
        case 0: write_features<NW, true, false>(L, g, lab, lastp, s, aug, P.lut, out, hv_pre, pre_ply, P.lut1, P.stamps); break;
        case 1: write_features<NW, false, false>(L, g, lab, lastp, s, aug, P.lut, out, hv_pre, pre_ply, P.lut1, P.stamps); break;
        case 2: write_features<NW, true, true>(L, g, lab, lastp, s, aug, P.lut, out, hv_pre, pre_ply, P.lut1, P.stamps); break;
        default: write_features<NW, false, true>(L, g, lab, lastp, s, aug, P.lut, out, hv_pre, pre_ply, P.lut1, P.stamps); break;
      }
    }
    STAMP(5);
#ifndef GOENV_ABLATE_MASKS
    if (a != -4 && !fresh) move_masks(L, g, lab, s, s.next, o);
#endif
    STAMP(6);
#ifndef GOENV_ABLATE_RECSTORE
    if (a != -4 || restarted || (P.flags & (GOENV_F_RESET_ILLEGAL | GOENV_F_COLLECT)))
      store_board(rec, g, lab, lastp, o, s);
#endif
  }
  if (g.tid == 0) {
    if (P.actions_out) P.actions_out[b] = a;
    if (P.status) P.status[b] = status;
  }
  if (skip) return;
#ifndef GOENV_ABLATE_LEGAL
  if (P.legal) write_legal(g, o, P.legal + (size_t)b * NPTS);
#endif
  STAMP(7);
}

// Planes (+ mask, + sequencing record) of the CURRENT positions of the selected boards.
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_features(BoardRec* state, int n_total, const int32_t* idx, int first,
                                                      int n, const int32_t* aug_codes, const float* lut, float* feat,
                                                      uint8_t* legal, goenv_sample_out meta, int collect,
                                                      float lut1, int zero_first, int xcd_map) {
  __shared__ BoardLds<NW> L;
  if ((int)blockIdx.x >= n) return;
  const int b = xcd_row(blockIdx.x, n, xcd_map);
  const int board = board_of(idx, first, n_total, b);
  if (board < 0) return;
  Geo<NW> g;
  geo_init(g);
  constexpr int K = Geo<NW>::K;
  BoardRec* rec = state + board;
  Scalars s;
  uint32_t lab[K], lastp[K];
  Masks m;
  load_record(rec, L, g, lab, lastp, s, m);
  float hv_pre[K];
  if (feat) {
    history_preload(g, lastp, s.ply, lut, hv_pre);
    if ((zero_first & 5) == 1) {
      if (zero_first & 2) zero_planes<NW, true>(g, feat + (size_t)b * GOENV_FEAT_FLOATS);
      else zero_planes<NW, false>(g, feat + (size_t)b * GOENV_FEAT_FLOATS);
    }
  }
  const int aug = aug_codes ? (aug_codes[b] & 7) : 0;
  if (collect) {
    emit_sample_meta(meta, b, board, g.tid, s, aug);
    if (g.tid == 0) { rec->sc[SC_SEQ] = s.seq; rec->sc[SC_SEQTOT] = s.seqtot; }
  }
  if (legal) {
    Own<NW> o;
    own_from_masks(g, m, o);
    write_legal(g, o, legal + (size_t)b * NPTS);
  }
  if (feat) {
    float* out = feat + (size_t)b * GOENV_FEAT_FLOATS;
    switch (zero_first & 3) {
      case 0: write_features<NW, true, false>(L, g, lab, lastp, s, aug, lut, out, hv_pre, s.ply, lut1); break;
      case 1: write_features<NW, false, false>(L, g, lab, lastp, s, aug, lut, out, hv_pre, s.ply, lut1); break;
      case 2: write_features<NW, true, true>(L, g, lab, lastp, s, aug, lut, out, hv_pre, s.ply, lut1); break;
      default: write_features<NW, false, true>(L, g, lab, lastp, s, aug, lut, out, hv_pre, s.ply, lut1); break;
    }
  }
}

// Move selection from a policy head: sample p proportional to pi[p] over the LEGAL points, in exact
// integer arithmetic (weights floor(pi * 2^24)), keyed by the counter RNG; all-zero weights fall
// back to the cfg-2 random policy.  pi is indexed in the augmented frame of aug_codes[b].
// One wavefront per board (reads only the record's masks; no board image needed).
constexpr int SAMPLER_WAVES = 4;   // boards per workgroup: the waves share nothing, fewer workgroups to dispatch
__global__ __launch_bounds__(64 * SAMPLER_WAVES) void k_sample_policy(const BoardRec* state, int n_total, const int32_t* idx,
                                                      int first, int n, const float* pi, const int32_t* aug_codes,
                                                      int64_t* actions_out, uint64_t seed, int64_t id_base) {
  const int b = blockIdx.x * SAMPLER_WAVES + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= n) return;
  const int board = board_of(idx, first, n_total, b);
  if (board < 0) {
    if (lane == 0) actions_out[b] = -1;
    return;
  }
  const BoardRec* rec = state + board;
  Scalars s;
  load_scalars(rec, s);
  Masks m;
  load_masks(rec, m);
  const int aug = aug_codes ? (aug_codes[b] & 7) : 0;
  float pv[NWORDS];
  policy_preload(lane, pi + (size_t)b * NPTS, aug, pv);
  const int a = pick_policy_move(m, lane, s, pv, seed, (uint64_t)(id_base + board));
  if (lane == 0) actions_out[b] = a;
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void k_score(const BoardRec* state, int n, float* tt, float* fast, int rule) {
  __shared__ BoardLds<NW> L;
  const int b = blockIdx.x;
  if (b >= n) return;
  Geo<NW> g;
  geo_init(g);
  constexpr int K = Geo<NW>::K;
  const BoardRec* rec = state + b;
  Scalars s;
  load_scalars(rec, s);
  uint32_t lab[K], lastp[K];
  load_board(rec, L, g, lab, lastp);
  if (fast) {
    int f = fast_score(L, g, lab, s, rule);
    if (g.tid == 0) fast[b] = (float)f;
  }
  if (tt) {
    int sc = tromp_taylor(L, g, lab);
    if (g.tid == 0) tt[b] = (float)sc;
  }
}

// Handicap stones, go/go_state.cc:32-70: effective point lists per stone count ("*n" tokens
// copy the n-stone list; the token itself decodes off-board and TryPlay rejects it).
__constant__ int8_t c_handi_xy[9][2] = {{3, 3}, {15, 15}, {3, 15}, {15, 3}, {3, 9}, {15, 9}, {9, 15}, {9, 3}, {9, 9}};
__constant__ int8_t c_handi_sets[10][10] = {
    {-1}, {-1}, {0, 1, -1}, {0, 1, 3, -1}, {0, 1, 2, 3, -1}, {0, 1, 2, 3, 8, -1},
    {0, 1, 2, 3, 4, 5, -1}, {0, 1, 2, 3, 4, 5, 8, -1}, {0, 1, 2, 3, 4, 5, 6, 7, -1},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, -1}};

template <int NW>
__global__ __launch_bounds__(64 * NW) void k_reset(BoardRec* state, int n_boards, const int32_t* idx, int n,
                                                   const int32_t* handicap) {
  __shared__ BoardLds<NW> L;
  const int i = blockIdx.x;
  if (i >= n) return;
  const int b = idx ? idx[i] : i;
  if (b < 0 || b >= n_boards) return;
  Geo<NW> g;
  geo_init(g);
  constexpr int K = Geo<NW>::K;
  BoardRec* rec = state + b;
  Scalars s;
  load_scalars(rec, s);
  const int game = s.game, seqtot = s.seqtot;
  uint32_t lab[K], lastp[K];
  clear_board_lds(L, g, lab, lastp);
  clear_scalars(s);
  s.game = game;
  s.seqtot = seqtot;
  int h = handicap ? handicap[i] : 0;
  h = __builtin_amdgcn_readfirstlane(h);
  if (h >= 2 && h <= 9) {
    for (int j = 0; j < 10; ++j) {
      const int e = c_handi_sets[h][j];
      if (e < 0) break;
      const int p = c_handi_xy[e][0] * BS + c_handi_xy[e][1];
      // PlaceHandicap, go/board.cc:48-64: TryPlay/Play as BLACK, then ply and last moves reset
      bool ok = s.ply <= 1 && move_is_legal(L, s, p, S_BLACK);
      ok = __builtin_amdgcn_readfirstlane((int)ok) != 0;
      if (ok) {
        apply_move(L, g, lab, lastp, s, p, S_BLACK);
        s.ply = 1;
        s.l1 = s.l2 = s.l3 = s.l4 = -1;
      }
    }
  }
  Own<NW> o;
  move_masks(L, g, lab, s, s.next, o);
  store_board(rec, g, lab, lastp, o, s);
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void k_get_state(const BoardRec* state, int n_total, const int32_t* idx,
                                                       int first, int n, uint8_t* color, int16_t* libs,
                                                       int16_t* stones, uint16_t* last_placed, int32_t* scalars) {
  __shared__ BoardLds<NW> L;
  const int b = blockIdx.x;
  if (b >= n) return;
  const int board = board_of(idx, first, n_total, b);
  if (board < 0) return;
  Geo<NW> g;
  geo_init(g);
  constexpr int K = Geo<NW>::K;
  const BoardRec* rec = state + board;
  uint32_t lab[K], lastp[K];
  load_board(rec, L, g, lab, lastp);
  rebuild_lib(L, g, lab);  // exact liberty counts (the record keeps only min(lib,3))
  // stones per root
  GO_UNROLL
  for (int k = 0; k < K; ++k)
    if (g.p[k] < NPTS) L.desc[g.p[k]] = 0;
  WSYNC();
  GO_UNROLL
  for (int k = 0; k < K; ++k)
    if (g.p[k] < NPTS && lab[k]) atomicAdd(&L.desc[lab[k] & ROOT_MASK], 1u);
  WSYNC();
  const size_t o = (size_t)b * NPTS;
  GO_UNROLL
  for (int k = 0; k < K; ++k) {
    const int p = g.p[k];
    if (p < NPTS) {
      const uint32_t v = lab[k];
      if (color) color[o + p] = (uint8_t)col_of(v);
      if (libs) libs[o + p] = v ? (int16_t)L.lib[v & ROOT_MASK] : 0;
      if (stones) stones[o + p] = v ? (int16_t)L.desc[v & ROOT_MASK] : 0;
      if (last_placed) last_placed[o + p] = v ? (uint16_t)lastp[k] : 0;
    }
  }
  if (scalars && g.tid < 16) {
    int v = rec->sc[g.tid];
    if (g.tid == SC_COUNTED) v = 0;
    scalars[(size_t)b * 16 + g.tid] = v;
  }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void k_set_state(BoardRec* state, int n, const uint8_t* color,
                                                       const uint16_t* last_placed, const int32_t* scalars) {
  __shared__ BoardLds<NW> L;
  const int b = blockIdx.x;
  if (b >= n) return;
  Geo<NW> g;
  geo_init(g);
  constexpr int K = Geo<NW>::K;
  BoardRec* rec = state + b;
  const size_t o = (size_t)b * NPTS;
  uint32_t lab[K], lastp[K];
  fill_rim(L, g.tid);
  board_any_init(L, g);
  WSYNC();
  GO_UNROLL
  for (int k = 0; k < K; ++k) {
    lab[k] = 0; lastp[k] = 0;
    if (g.p[k] < NPTS) {
      uint32_t col = color[o + g.p[k]];
      col = (col == 1 || col == 2) ? col : 0;
      lab[k] = col ? ((col << 9) | (uint32_t)g.p[k]) : 0u;
      lastp[k] = last_placed ? last_placed[o + g.p[k]] : 0u;
      L.lab[g.c[k]] = (uint16_t)lab[k];
    }
  }
  WSYNC();
  // connected components: min-label propagation to a fixed point (root = smallest member)
  for (int iter = 0; iter < 2 * NPTS; ++iter) {
    bool changed = false;
    GO_UNROLL
    for (int k = 0; k < K; ++k) {
      if (g.p[k] < NPTS && lab[k]) {
        uint32_t n4[4];
        nbr4(L, g.c[k], n4);
        uint32_t m = lab[k];
        GO_UNROLL
        for (int i = 0; i < 4; ++i)
          if (col_of(n4[i]) == col_of(lab[k]) && n4[i] < m) m = n4[i];
        if (m != lab[k]) { lab[k] = m; L.lab[g.c[k]] = (uint16_t)m; changed = true; }
      }
    }
    if (!board_any(L, g, changed, iter)) break;
  }
  WSYNC();
  rebuild_lib(L, g, lab);
  Scalars s;
  const int32_t* sc = scalars + (size_t)b * 16;
  s.ply = sc[0]; s.next = sc[1]; s.bcap = sc[2]; s.wcap = sc[3];
  s.l1 = sc[4]; s.l2 = sc[5]; s.l3 = sc[6]; s.l4 = sc[7];
  s.koage = sc[8] & 0xffff; s.ko = sc[9]; s.kocolor = sc[10];
  s.done = sc[11]; s.game = sc[12]; s.counted = 0; s.seq = sc[14]; s.seqtot = sc[15];
  Own<NW> mk;
  move_masks(L, g, lab, s, s.next, mk);
  store_board(rec, g, lab, lastp, mk, s);
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void k_finish(BoardRec* state, int n, unsigned long long* stats, float komi) {
  __shared__ BoardLds<NW> L;
  const int b = blockIdx.x;
  if (b >= n) return;
  BoardRec* rec = state + b;
  Scalars s;
  load_scalars(rec, s);
  if (!s.done || s.counted) return;  // uniform over the workgroup: no barrier is skipped by a part of it
  Geo<NW> g;
  geo_init(g);
  constexpr int K = Geo<NW>::K;
  uint32_t lab[K], lastp[K];
  load_board(rec, L, g, lab, lastp);
  account_game(stats, g.tid, s, tromp_taylor(L, g, lab), komi);
  if (g.tid == 0) rec->sc[SC_COUNTED] = 1;
}

// Persistent random playout: the board stays in LDS/registers for the whole game.
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_playout(PlyParams P, int16_t* moves, int max_record, int32_t* nmoves) {
  __shared__ BoardLds<NW> L;
  const int b = blockIdx.x;
  if (b >= P.n) return;
  Geo<NW> g;
  geo_init(g);
  constexpr int K = Geo<NW>::K;
  BoardRec* rec = P.state + b;
  Scalars s;
  uint32_t lab[K], lastp[K];
  Masks m;
  Own<NW> o;
  load_record(rec, L, g, lab, lastp, s, m);
  own_from_masks(g, m, o);
  int cnt = 0;
  // exit condition every wave reaches: done flag (uniform), or the hard bound on iterations
  for (int it = 0; it < 4096 && !s.done; ++it) {
    int a = ply_body<NW, true>(L, g, lab, lastp, s, m, o, P, b, 0ll);
    gather_masks(L, g, o, m);
    if (moves && cnt < max_record && g.tid == 0) moves[(size_t)b * max_record + cnt] = (int16_t)a;
    ++cnt;
  }
  store_board(rec, g, lab, lastp, o, s);
  if (nmoves && g.tid == 0) nmoves[b] = cnt;
}

// Persistent generator: `n_plies` plies of the on-device policy per board inside one launch, ply
// t's mask / planes / move written to slot t of the caller's ring buffers.  The board stays in
// LDS/registers; boards drift apart in time, so the rules phase of one overlaps the plane stores
// of another (what a launch per ply cannot do at small batch sizes).
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_generate(PlyParams P, int n_plies, int32_t* actions_ring) {
  __shared__ BoardLds<NW> L;
  if ((int)blockIdx.x >= P.n) return;
  const int b = xcd_row(blockIdx.x, P.n, P.xcd_map);
  Geo<NW> g;
  geo_init(g);
  constexpr int K = Geo<NW>::K;
  BoardRec* rec = P.state + b;
  Scalars s;
  uint32_t lab[K], lastp[K];
  Masks m;
  Own<NW> o;
  load_record(rec, L, g, lab, lastp, s, m);
  own_from_masks(g, m, o);
  float hv_none[K];
  GO_UNROLL
  for (int k = 0; k < K; ++k) hv_none[k] = 0.f;
  for (int t = 0; t < n_plies; ++t) {  // host-supplied bound, the same for every wave
    if (NW > 1) WSYNC();  // the previous ply's plane stores have taken their LDS operands
    STAMP(8);
    if (s.done) {
      if (!s.counted) account_game(P.stats, g.tid, s, tromp_taylor(L, g, lab), P.komi);
      restart_in_place(L, g, lab, lastp, s, m, o);
    }
    const int a = ply_body<NW, true>(L, g, lab, lastp, s, m, o, P, b, 0ll);
    STAMP(9);
    gather_masks(L, g, o, m);
    STAMP(10);
    const size_t row = (size_t)t * P.n + b;
    if (actions_ring && g.tid == 0) actions_ring[row] = a;
    if (P.legal) write_legal(g, o, P.legal + row * NPTS);
    STAMP(11);
    if (P.feat) {
      const int aug = P.aug ? (P.aug[b] & 7) : 0;
      if (P.zero_first & 4)   // constant planes preset by the caller (goenv_set_constant_planes_preset)
        write_features<NW, false, false>(L, g, lab, lastp, s, aug, P.lut, P.feat + row * GOENV_FEAT_FLOATS, hv_none, -1,
                                         P.lut1, P.stamps);
      else
        write_features<NW, true, false>(L, g, lab, lastp, s, aug, P.lut, P.feat + row * GOENV_FEAT_FLOATS, hv_none, -1,
                                        P.lut1, P.stamps);
    }
    STAMP(12);
  }
  store_board(rec, g, lab, lastp, o, s);
}

#ifndef GOENV_PIPE_ABLATE
#define GOENV_PIPE_ABLATE 0
#endif
#ifndef GOENV_PACE_RESET
// periods a feature wave may fall behind its schedule before the schedule starts over.  It catches up by leaving waits out, so that
// all boards end a launch together (2 periods: 6.57 us per ply at a pinned 6.4 us; 16: 6.48; profiles/r03_pipe_generator.md section 14)
#define GOENV_PACE_RESET 16
#endif
// The same generator as a two-stage pipeline INSIDE the board's workgroup (goenv_device.h, PipeSnap): wave 0 plays ply
// t + 1 while wave 1 writes the sample of ply t.  Same ring contents, bit for bit.
__global__ __launch_bounds__(128) void k_generate_pipe(PlyParams P, int n_plies, int32_t* actions_ring, int pace_ticks) {
  __shared__ BoardLds<1> L;
  __shared__ PipeLds X;
  if ((int)blockIdx.x >= P.n) return;
  const int b = xcd_row(blockIdx.x, P.n, P.xcd_map);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int aug = P.aug ? (P.aug[b] & 7) : 0;
  if (threadIdx.x == 0) { X.produced = 0; X.consumed = 0; }
  for (int i = threadIdx.x; i < PIPE_LUT_N; i += 128) X.lut[i] = P.lut[i];
  __syncthreads();   // the only barrier of the kernel
#ifdef GOENV_STAMPS
  unsigned long long acc_[6] = {0, 0, 0, 0, 0, 0}, tk_ = wall_clock64();
#define PTICK(i) do { const unsigned long long n_ = wall_clock64(); acc_[i] += n_ - tk_; tk_ = n_; } while (0)
#else
#define PTICK(i) do { } while (0)
#endif
  if (wave == 0) {
    Geo<1> g;
    geo_init(g);
    constexpr int K = Geo<1>::K;
    BoardRec* rec = P.state + b;
    Scalars s;
    uint32_t lab[K], lastp[K];
    Masks m;
    Own<1> o;
    load_record(rec, L, g, lab, lastp, s, m);
    own_from_masks(g, m, o);
    PTICK(0);
    // the rules wave goes before the feature waves that share its SIMD: its chain is the dependent one, theirs waits for the
    // store queue anyway (512 boards: 5.9 -> 5.3 us per ply, 1,024: -2 %, 4,096: -4 %; profiles/r03_pipe_generator.md)
    // (up to five workgroups per CU; with six the feature waves need their turns: 1,536 boards 11.9 -> 12.5 us per ply)
#ifndef GOENV_NO_RULES_PRIO
    if (gridDim.x <= 1280) __builtin_amdgcn_s_setprio(3);
#endif
    for (int t = 0; t < n_plies; ++t) {  // host-supplied bound, the same for both waves
#if GOENV_PIPE_ABLATE == 2   // probe: no rules (the feature wave and the store stream alone)
      const int a = -1;
#else
      if (s.done) {
        if (!s.counted) account_game(P.stats, g.tid, s, tromp_taylor(L, g, lab), P.komi);
        restart_in_place(L, g, lab, lastp, s, m, o);
      }
      const int a = ply_body<1, true>(L, g, lab, lastp, s, m, o, P, b, 0ll);
      gather_masks(L, g, o, m);
#endif
      PTICK(1);
      if (t >= PIPE_DEPTH) pipe_wait(&X.consumed, t - PIPE_DEPTH + 1, P.stats + GOENV_NUM_STATS + 1);   // slot t % PIPE_DEPTH has been taken
      PTICK(2);
      snapshot_write(X.snap[t % PIPE_DEPTH], g, lab, lastp, s, o, a);
      pipe_post(&X.produced, t + 1, g.lane);
      PTICK(3);
    }
    store_board(rec, g, lab, lastp, o, s);
  } else {
    const int lane = (int)(threadIdx.x & 63);
    int src[8];
    for (int i = 0; i < 8; ++i) {
      const int q = i < 4 ? 4 * lane + i : (lane < 26 ? 256 + 4 * lane : NPTS - 4) + (i - 4);
      src[i] = aug ? d4_src(aug, q) : q;
    }
    // the schedule starts when the wave does, not with its first sample (which comes a ply later): the first two samples
    // go out back to back, into a memory system that the launch has not loaded yet
    unsigned long long next = wall_clock64();
    for (int t = 0; t < n_plies; ++t) {
      PTICK(0);
      pipe_wait(&X.produced, t + 1, P.stats + GOENV_NUM_STATS + 1);
      PTICK(1);
      SnapRegs r;
      snapshot_load(X.snap[t % PIPE_DEPTH], lane, src, r);
      pipe_post(&X.consumed, t + 1, lane);
      PTICK(2);
      if (pace_ticks) {
        // rate limiter on the 100-MHz wall clock: a board starts a sample at most once per `pace_ticks`.  The memory side
        // takes this store pattern FASTER when it is offered just below what it can sustain than when every wave
        // pushes as hard as it can (tools/store_bench.hip mode 8: 5.5 TB/s free-running, 6.0 TB/s paced).
        unsigned long long now = wall_clock64();
        for (int spin = 0; spin < 4096 && now < next; ++spin) {   // bounded: a wave always gets out
          __builtin_amdgcn_s_sleep(1);
          now = wall_clock64();
        }
        // a fixed schedule, so that a sample that started late does not delay those after it; dropped after a long gap
        next = (now > next + (unsigned long long)GOENV_PACE_RESET * (unsigned long long)pace_ticks) ? now + (unsigned long long)pace_ticks
                                                                     : next + (unsigned long long)pace_ticks;
      }
      PTICK(3);
      const size_t row = (size_t)t * P.n_total + P.first + b;   // (P.first: a launch over a part of the boards, as in the _multi kernel)
      if (actions_ring && lane == 0) actions_ring[row] = r.action;
      uint8_t* lg = P.legal ? P.legal + row * NPTS : nullptr;
#if GOENV_PIPE_ABLATE == 1   // probe: no samples written (the rules wave alone)
      if (false) {
#else
      if (P.feat) {
#endif
        if (P.zero_first & 4) snapshot_expand<false>(r, X, lane, aug, src, P.feat + row * GOENV_FEAT_FLOATS, lg);
        else snapshot_expand<true>(r, X, lane, aug, src, P.feat + row * GOENV_FEAT_FLOATS, lg);
      } else if (lg) {
        for (int k = 0; k < NWORDS; ++k) {
          const int p = lane + 64 * k;
          if (p < NPTS) lg[p] = (uint8_t)((r.lg[k] >> lane) & 1ull);
        }
      }
    }
  }
#ifdef GOENV_STAMPS
  PTICK(4);   // the feature wave: expansion of the last sample; the rules wave: the record store
  if (P.stamps && (threadIdx.x & 63) == 0)
    for (int i = 0; i < 6; ++i) P.stamps[(size_t)blockIdx.x * 32 + wave * 16 + i] = acc_[i];
#endif
#undef PTICK
}

// The pipelined generator for MORE boards than the device holds workgroups: a grid of resident workgroups, each with the
// boards w, w + G, w + 2G, ... waiting in LDS in the record's own layout (dynamic LDS: 1,632 B per board).  The rules wave
// takes them in turn, one ply each; the feature wave writes their samples at an even pace.  Same rings as k_generate.
__global__ __launch_bounds__(128) void k_generate_pipe_multi(PlyParams P, int n_plies, int32_t* actions_ring,
                                                             int ply_period_ticks) {
  __shared__ BoardLds<1> L;
  __shared__ PipeLds X;
  extern __shared__ __align__(16) unsigned char saved_raw[];
  SavedBoard* saved = reinterpret_cast<SavedBoard*>(saved_raw);
  const int G = (int)gridDim.x, w = (int)blockIdx.x, n = P.n;
  const int mine = w < n ? (n - 1 - w) / G + 1 : 0;          // boards of this workgroup
  // its i-th board: turn i of the grid covers the rows [i G, (i + 1) G) of a slot -- one window of the output at a time, each XCD
  // a contiguous part of it (GOENV_MULTI_ROWS=1, probe: the rows w + i G dealt per XCD over the whole slot instead)
  auto board_at = [&](int i) {
#ifdef GOENV_MULTI_ROWS
    return xcd_row(w + i * G, n, P.xcd_map);
#else
    const int left = n - i * G;
    return i * G + xcd_row(w, left < G ? left : G, P.xcd_map);
#endif
  };
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = (int)(threadIdx.x & 63);
  if (threadIdx.x == 0) { X.produced = 0; X.consumed = 0; }
  for (int i = threadIdx.x; i < PIPE_LUT_N; i += 128) X.lut[i] = P.lut[i];
  for (int i = 0; i < mine; ++i) saved_from_record(saved[i], P.state + board_at(i), (int)threadIdx.x, 128);
  __syncthreads();   // the only barrier of the kernel
#ifdef GOENV_STAMPS
  unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tk_ = wall_clock64();
#define PTICK(i) do { const unsigned long long n_ = wall_clock64(); acc_[i] += n_ - tk_; tk_ = n_; } while (0)
#else
#define PTICK(i) do { } while (0)
#endif
  const int items = mine * n_plies;
  if (wave == 0) {
    Geo<1> g;
    geo_init(g);
    fill_rim(L, g.tid);
    wsync<1>();
    int j = 0;
    SavedRegs fetched;
    if (mine > 0) lds_fetch_board(saved[0], g, fetched);
#ifndef GOENV_NO_RULES_PRIO
    __builtin_amdgcn_s_setprio(3);   // as in k_generate_pipe
#endif
    for (int t = 0; t < n_plies; ++t) {      // host-supplied bounds, the same for both waves
      for (int i = 0; i < mine; ++i, ++j) {
        const int b = board_at(i);
        Scalars s;
        uint32_t lab[NWORDS], lastp[NWORDS];
        Masks m;
        Own<1> o;
        PTICK(0);
        lds_load_board(fetched, L, g, lab, lastp, s, m);
        own_from_masks(g, m, o);
        PTICK(1);
        if (s.done) {
          if (!s.counted) account_game(P.stats, g.tid, s, tromp_taylor(L, g, lab), P.komi);
          restart_in_place(L, g, lab, lastp, s, m, o);
        }
        const int a = ply_body<1, true>(L, g, lab, lastp, s, m, o, P, b, 0ll);
        gather_masks(L, g, o, m);
        PTICK(2);
        lds_store_board(saved[i], g, lab, lastp, s, m);
        lds_fetch_board(saved[i + 1 < mine ? i + 1 : 0], g, fetched);   // the next turn's board, under the hand-over below
        PTICK(3);
        if (j >= PIPE_DEPTH) pipe_wait(&X.consumed, j - PIPE_DEPTH + 1, P.stats + GOENV_NUM_STATS + 1);
        PTICK(4);
        PipeSnap& S = X.snap[j % PIPE_DEPTH];
        snapshot_write(S, g, lab, lastp, s, o, a);
        if (g.lane == 0) { S.aug = P.aug ? (P.aug[b] & 7) : 0; S.row = b; }
        pipe_post(&X.produced, j + 1, g.lane);
        PTICK(5);
      }
    }
    wsync<1>();
    for (int i = 0; i < mine; ++i) record_from_saved(P.state + board_at(i), saved[i], lane, 64);
  } else {
    const int period = mine > 0 ? ply_period_ticks / mine : 0;
    int src[8], cur_aug = -1;
    unsigned long long next = wall_clock64();
    int t = 0, i = 0;
    for (int j = 0; j < items; ++j) {
      PTICK(0);
      pipe_wait(&X.produced, j + 1, P.stats + GOENV_NUM_STATS + 1);
      PTICK(1);
      const PipeSnap& S = X.snap[j % PIPE_DEPTH];
      const int aug = __builtin_amdgcn_readfirstlane(S.aug), b = __builtin_amdgcn_readfirstlane(S.row);
      if (aug != cur_aug) {
        for (int q8 = 0; q8 < 8; ++q8) {
          const int q = q8 < 4 ? 4 * lane + q8 : (lane < 26 ? 256 + 4 * lane : NPTS - 4) + (q8 - 4);
          src[q8] = aug ? d4_src(aug, q) : q;
        }
        cur_aug = aug;
      }
      SnapRegs r;
      snapshot_load(S, lane, src, r);
      pipe_post(&X.consumed, j + 1, lane);
      PTICK(2);
      if (period) {   // the rate limiter of k_generate_pipe, one sample per period
        unsigned long long now = wall_clock64();
        for (int spin = 0; spin < 4096 && now < next; ++spin) {   // bounded: a wave always gets out
          __builtin_amdgcn_s_sleep(1);
          now = wall_clock64();
        }
        next = (now > next + (unsigned long long)GOENV_PACE_RESET * (unsigned long long)period) ? now + (unsigned long long)period
                                                                 : next + (unsigned long long)period;
      }
      PTICK(3);
      // (a launch over a part of the context's boards -- goenv_generate in chunks -- has P.state, P.aug and P.id_base moved to
      // its first board; a slot of the rings holds all P.n_total boards)
      const size_t row = (size_t)t * P.n_total + P.first + b;
      if (actions_ring && lane == 0) actions_ring[row] = r.action;
      uint8_t* lg = P.legal ? P.legal + row * NPTS : nullptr;
      if (P.feat) {
        if (P.zero_first & 4) snapshot_expand<false>(r, X, lane, aug, src, P.feat + row * GOENV_FEAT_FLOATS, lg);
        else snapshot_expand<true>(r, X, lane, aug, src, P.feat + row * GOENV_FEAT_FLOATS, lg);
      } else if (lg) {
        for (int k = 0; k < NWORDS; ++k) {
          const int p = lane + 64 * k;
          if (p < NPTS) lg[p] = (uint8_t)((r.lg[k] >> lane) & 1ull);
        }
      }
      if (++i == mine) { i = 0; ++t; }
    }
  }
#ifdef GOENV_STAMPS
  PTICK(6);
  if (P.stamps && (threadIdx.x & 63) == 0)
    for (int i = 0; i < 8; ++i) P.stamps[(size_t)blockIdx.x * 32 + wave * 16 + i] = acc_[i];
#endif
#undef PTICK
}

// Replay of given move lists (SGF records packed on the host): row b applies
// moves[b*stride + 0 .. nmoves[b]) to its board with GoState::forward semantics and stops at the
// first illegal move (go/offpolicy_loader.cc:120-126); napplied[b] = moves applied.  The board
// stays in LDS/registers for the whole list.
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_replay(PlyParams P, const int16_t* moves, int stride,
                                                    const int32_t* nmoves, int32_t* napplied) {
  __shared__ BoardLds<NW> L;
  const int b = blockIdx.x;
  if (b >= P.n) return;
  Geo<NW> g;
  geo_init(g);
  constexpr int K = Geo<NW>::K;
  const int board = board_of(P.idx, P.first, P.n_total, b);
  if (board < 0) {
    if (g.tid == 0 && napplied) napplied[b] = -1;
    return;
  }
  BoardRec* rec = P.state + board;
  Scalars s;
  uint32_t lab[K], lastp[K];
  Masks m;
  Own<NW> o;
  load_record(rec, L, g, lab, lastp, s, m);
  own_from_masks(g, m, o);
  int n = nmoves[b];
  n = __builtin_amdgcn_readfirstlane(n < 0 ? 0 : (n > stride ? stride : n));
  int cnt = 0;
  for (; cnt < n; ++cnt) {  // bounded by the host-supplied list length
    const int a = ply_body<NW, false>(L, g, lab, lastp, s, m, o, P, board, (long long)moves[(size_t)b * stride + cnt]);
    if (a == -4) break;
    gather_masks(L, g, o, m);
  }
  if (cnt > 0) store_board(rec, g, lab, lastp, o, s);
  if (napplied && g.tid == 0) napplied[b] = cnt;
}

// One offline (SGF replay) batch in ONE launch: the device half of OfflineLoader's reload / next / extract
// for every row (go/offpolicy_loader.cc:70-108,120-145).  Row b is ROW_W(k) int32 words of a staging block
// that the host packed (device memory or pinned host memory, read in place):
//   [0] board  [1] op = aug | reset << 3 | handicap << 4 | nmoves << 8  [2] offset of the row's moves (int16
//   units from `moves`)  [3] reserved  [4..12) the eight labels in goenv_sample_out order
//   [12..12 + 2k) offline_a, k int64
// The board is reset (ClearBoard + HandicapTable::Apply, as k_reset) when the op says so, the row's moves
// are applied with GoState::forward semantics (the host knows them to be legal: every record is replayed
// once when it is first loaded; a move the device rejects all the same stops the row and is counted in
// *bad_moves), then the planes of the position under the row's D4 code, its legal mask and the labels go
// straight into the batch tensors.
constexpr int OFFLINE_ROW_HEAD = 12;
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_offline(PlyParams P, const int32_t* __restrict__ rows, int row_w, int k_future,
                                                     const int16_t* __restrict__ moves, int64_t* offline_a_out,
                                                     int32_t* bad_moves) {
  __shared__ BoardLds<NW> L;
  if ((int)blockIdx.x >= P.n) return;
  const int b = xcd_row(blockIdx.x, P.n, P.xcd_map);
  Geo<NW> g;
  geo_init(g);
  constexpr int K = Geo<NW>::K;
  const int32_t* row = rows + (size_t)b * row_w;
  const int board = __builtin_amdgcn_readfirstlane(row[0]);
  const uint32_t op = (uint32_t)__builtin_amdgcn_readfirstlane(row[1]);
  const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane(row[2]);
  // labels: one lane per word, requested before anything else so that the trip to the staging block
  // (PCIe when it is host memory) overlaps the record load
  int32_t label = 0;
  if (g.tid < 8 + 2 * k_future) label = row[4 + g.tid];
  if (board < 0 || board >= P.n_total) return;
  const int aug = (int)(op & 7u);
  const bool reset = (op >> 3) & 1u;
  const int handi = (int)((op >> 4) & 15u);
  const int n = (int)(op >> 8);
  BoardRec* rec = P.state + board;
  Scalars s;
  uint32_t lab[K], lastp[K];
  Masks m;
  Own<NW> o;
  float hv_pre[K];
  int pre_ply = -1;
  if (reset) {
    load_scalars(rec, s);
    const int game = s.game, seqtot = s.seqtot;
    clear_board_lds(L, g, lab, lastp);
    clear_scalars(s);
    s.game = game;
    s.seqtot = seqtot;
    if (handi >= 2 && handi <= 9) {
      for (int j = 0; j < 10; ++j) {
        const int e = c_handi_sets[handi][j];
        if (e < 0) break;
        const int p = c_handi_xy[e][0] * BS + c_handi_xy[e][1];
        bool ok = s.ply <= 1 && move_is_legal(L, s, p, S_BLACK);   // PlaceHandicap, go/board.cc:48-64
        ok = __builtin_amdgcn_readfirstlane((int)ok) != 0;
        if (ok) {
          apply_move(L, g, lab, lastp, s, p, S_BLACK);
          s.ply = 1;
          s.l1 = s.l2 = s.l3 = s.l4 = -1;
        }
      }
    }
    move_masks(L, g, lab, s, s.next, o);
    gather_masks(L, g, o, m);
    GO_UNROLL
    for (int k = 0; k < K; ++k) hv_pre[k] = 0.f;
  } else {
    load_record(rec, L, g, lab, lastp, s, m);
    own_from_masks(g, m, o);
    if (P.feat && n == 1) {   // the common row: one move, history values requested with the record
      pre_ply = s.ply + 1;
      history_preload(g, lastp, pre_ply, P.lut, hv_pre);
    } else {
      GO_UNROLL
      for (int k = 0; k < K; ++k) hv_pre[k] = 0.f;
    }
  }
  int cnt = 0;
  bool dirty = reset;
  for (; cnt < n; ++cnt) {  // bounded by the host-supplied count
    const int a = ply_body<NW, false, false>(L, g, lab, lastp, s, m, o, P, board, (long long)moves[(size_t)off + cnt]);
    if (a == -4) break;
    dirty = true;
    if (cnt + 1 < n) {      // the next move's TryPlay needs this position's masks; the last one's come after the planes
      move_masks(L, g, lab, s, s.next, o);
      gather_masks(L, g, o, m);
    }
  }
  if (cnt < n) {
    if (g.tid == 0 && bad_moves) atomicAdd(bad_moves, 1);
    pre_ply = -1;
  }
  // labels (GameState fields, go/go_game_specific.h:59-114), written as the host computed them
  {
    const goenv_sample_out& o8 = P.meta;
    int32_t* dst32[8] = {o8.id, o8.seq, o8.game_counter, nullptr, o8.move_idx, o8.winner, o8.aug_code, o8.game_record_idx};
    GO_UNROLL
    for (int i = 0; i < 8; ++i)
      if (g.tid == i && dst32[i]) dst32[i][b] = label;
    if (g.tid == 3 && o8.last_terminal) o8.last_terminal[b] = (int8_t)label;
    if (offline_a_out && g.tid >= 8 && g.tid < 8 + 2 * k_future)
      reinterpret_cast<int32_t*>(offline_a_out)[(size_t)b * 2 * k_future + (g.tid - 8)] = label;
  }
  if (P.feat) {
    float* out = P.feat + (size_t)b * GOENV_FEAT_FLOATS;
    switch (P.zero_first & 6) {
      case 0: write_features<NW, true, false>(L, g, lab, lastp, s, aug, P.lut, out, hv_pre, pre_ply, P.lut1); break;
      case 2: write_features<NW, true, true>(L, g, lab, lastp, s, aug, P.lut, out, hv_pre, pre_ply, P.lut1); break;
      case 4: write_features<NW, false, false>(L, g, lab, lastp, s, aug, P.lut, out, hv_pre, pre_ply, P.lut1); break;
      default: write_features<NW, false, true>(L, g, lab, lastp, s, aug, P.lut, out, hv_pre, pre_ply, P.lut1); break;
    }
  }
  if (cnt > 0 && cnt == n) move_masks(L, g, lab, s, s.next, o);   // (a rejected move left `o` at the position's masks)
  if (dirty) store_board(rec, g, lab, lastp, o, s);
  if (P.legal) write_legal(g, o, P.legal + (size_t)b * NPTS);
}

// Record copies between slots of one context: what a host-side tree search needs to branch
// (child slot = copy of the parent slot, then one goenv_ply_sel on the children).
__global__ __launch_bounds__(128) void k_copy_boards(BoardRec* state, int n_total, const int32_t* src,
                                                     const int32_t* dst, int n) {
  const int b = blockIdx.x;
  if (b >= n) return;
  const int from = src[b], to = dst[b];
  if (from < 0 || from >= n_total || to < 0 || to >= n_total || from == to) return;
  const uint4* in = reinterpret_cast<const uint4*>(state + from);
  uint4* out = reinterpret_cast<uint4*>(state + to);
  for (int i = threadIdx.x; i < (int)(sizeof(BoardRec) / 16); i += blockDim.x) out[i] = in[i];
}

// Row gather / scatter between two row-major tables: row dst_rows[i] of dst <- row src_rows[i] of src
// (NULL list: row i).  The device form of the collector's per-key memcpy (elf/copier.hh:97-101) between
// a game's history and a batch tensor.  Dword copies when both rows are 4-byte aligned, bytes otherwise.
__global__ __launch_bounds__(256) void k_copy_rows(const unsigned char* src, const int32_t* src_rows, unsigned char* dst,
                                                   const int32_t* dst_rows, int n, size_t row_bytes) {
  const int i = blockIdx.x;
  if (i >= n) return;
  const long long sr = src_rows ? src_rows[i] : i, dr = dst_rows ? dst_rows[i] : i;
  if (sr < 0 || dr < 0) return;
  const unsigned char* a = src + (size_t)sr * row_bytes;
  unsigned char* b = dst + (size_t)dr * row_bytes;
  if ((((uintptr_t)a | (uintptr_t)b | row_bytes) & 3) == 0) {
    const uint32_t* a4 = reinterpret_cast<const uint32_t*>(a);
    uint32_t* b4 = reinterpret_cast<uint32_t*>(b);
    for (size_t k = threadIdx.x; k < row_bytes / 4; k += blockDim.x) b4[k] = a4[k];
  } else {
    for (size_t k = threadIdx.x; k < row_bytes; k += blockDim.x) b[k] = a[k];
  }
}

// The k legal points of largest policy weight, in descending order: what DirectPredictAI::get_last_pi
// (go/go_ai.h:24-68) hands to the tree search (it sorts all 361 by pi and keeps the first five that
// pass CheckMove).  One wavefront per board; k rounds of a wave-wide arg-max over the legal
// ballots (ties: lowest point index; the reference's std::sort leaves them unspecified).
__global__ __launch_bounds__(64) void k_topk_policy(const BoardRec* state, int n_total, const int32_t* idx, int first,
                                                    int n, const float* pi, const int32_t* aug_codes, int k,
                                                    int32_t* actions_out, float* probs_out) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= n) return;
  const int board = board_of(idx, first, n_total, b);
  Masks m;
  if (board >= 0) load_masks(state + board, m);
  const int aug = aug_codes ? (aug_codes[b] & 7) : 0;
  const float* row = pi + (size_t)b * NPTS;
  float v[NWORDS];
  bool live[NWORDS];
  GO_UNROLL
  for (int j = 0; j < NWORDS; ++j) {
    const int p = lane + 64 * j;
    live[j] = board >= 0 && p < NPTS && ((m.legal[j] >> lane) & 1ull);
    v[j] = live[j] ? row[aug ? d4_dest(aug, p) : p] : 0.f;
    if (v[j] != v[j]) live[j] = false;  // NaN weights never compare: leave them out
  }
  for (int r = 0; r < k; ++r) {  // host-checked bound
    // lane-local best, then a wave reduction on (value, lowest point)
    float best = 0.f;
    int bp = -1;
    GO_UNROLL
    for (int j = 0; j < NWORDS; ++j)
      if (live[j] && (bp < 0 || v[j] > best)) { best = v[j]; bp = lane + 64 * j; }
    GO_UNROLL
    for (int d = 32; d >= 1; d >>= 1) {
      const float ob = __shfl_xor(best, d);
      const int op = __shfl_xor(bp, d);
      if (op >= 0 && (bp < 0 || ob > best || (ob == best && op < bp))) { best = ob; bp = op; }
    }
    if (lane == 0) {
      actions_out[(size_t)b * k + r] = bp;
      if (probs_out) probs_out[(size_t)b * k + r] = bp >= 0 ? best : 0.f;
    }
    GO_UNROLL
    for (int j = 0; j < NWORDS; ++j)
      if (bp == lane + 64 * j) live[j] = false;
  }
}

// Heuristic move filters (goenv_tactics.h): always one wavefront per board -- the work is either
// per point (one pass) or a played-out line (serial), neither gains from more waves.
__global__ __launch_bounds__(64) void k_tactics(const BoardRec* state, int n_total, const int32_t* idx, int first,
                                                int count, int thres, TacticsOut out) {
  __shared__ BoardLds<1> L;
  __shared__ TacticsLds X;
  const int b = blockIdx.x;
  if (b >= count) return;
  const int board = board_of(idx, first, n_total, b);
  if (board < 0) return;
  Geo<1> g;
  geo_init(g);
  constexpr int K = Geo<1>::K;
  Scalars s;
  uint32_t lab[K], lastp[K];
  Masks m;
  Own<1> o;
  load_record(state + board, L, g, lab, lastp, s, m);
  own_from_masks(g, m, o);
  tactics(L, X, g, lab, s, o, thres, out, (size_t)b);
}

// Launch of a kernel template on the waves-per-board count chosen for this call.
// probe (tuning only): GOENV_LDS_PAD=<bytes> of unused dynamic LDS per workgroup, to cap the workgroups a CU holds
static size_t g_lds_pad = getenv("GOENV_LDS_PAD") ? (size_t)atoi(getenv("GOENV_LDS_PAD")) : 0;
#define GO_LAUNCH_LDS(NWSEL, KERNEL, GRID, LDS, STREAM, ...)                                           \
  do {                                                                                               \
    switch (NWSEL) {                                                                                 \
      case 2: hipLaunchKernelGGL((KERNEL<2>), dim3(GRID), dim3(128), LDS, STREAM, __VA_ARGS__); break; \
      case 3: hipLaunchKernelGGL((KERNEL<3>), dim3(GRID), dim3(192), LDS, STREAM, __VA_ARGS__); break; \
      case 6: hipLaunchKernelGGL((KERNEL<6>), dim3(GRID), dim3(384), LDS, STREAM, __VA_ARGS__); break; \
      default: hipLaunchKernelGGL((KERNEL<1>), dim3(GRID), dim3(64), LDS, STREAM, __VA_ARGS__); break; \
    }                                                                                                \
  } while (0)
#define GO_LAUNCH(NWSEL, KERNEL, GRID, STREAM, ...) GO_LAUNCH_LDS(NWSEL, KERNEL, GRID, g_lds_pad, STREAM, __VA_ARGS__)

#define GO_LAUNCH_PLY_MODE(NWSEL, MODE, GRID, STREAM, P)                                              \
  do {                                                                                                \
    switch (NWSEL) {                                                                                  \
      case 2: hipLaunchKernelGGL((k_ply<2, MODE>), dim3(GRID), dim3(128), g_lds_pad, STREAM, P); break;       \
      case 3: hipLaunchKernelGGL((k_ply<3, MODE>), dim3(GRID), dim3(192), g_lds_pad, STREAM, P); break;       \
      case 6: hipLaunchKernelGGL((k_ply<6, MODE>), dim3(GRID), dim3(384), g_lds_pad, STREAM, P); break;       \
      default: hipLaunchKernelGGL((k_ply<1, MODE>), dim3(GRID), dim3(64), g_lds_pad, STREAM, P); break;       \
    }                                                                                                 \
  } while (0)

// MODESEL: 0 external actions, 1 on-device random policy, 2 sampled from pi
#define GO_LAUNCH_PLY(NWSEL, MODESEL, GRID, STREAM, P)                  \
  do {                                                                  \
    if ((MODESEL) == 1) GO_LAUNCH_PLY_MODE(NWSEL, 1, GRID, STREAM, P);  \
    else if ((MODESEL) == 2) GO_LAUNCH_PLY_MODE(NWSEL, 2, GRID, STREAM, P); \
    else GO_LAUNCH_PLY_MODE(NWSEL, 0, GRID, STREAM, P);                 \
  } while (0)

// ----------------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------------

constexpr int STAGE_GROUP = 4;   // staging ring: blocks per event (stage_commit)
static_assert(GOENV_STAGE_SLOTS % STAGE_GROUP == 0 && GOENV_STAGE_SLOTS >= 2 * STAGE_GROUP, "staging ring: whole groups, at least two");
// plane outputs of the last launches (planes_in_flight below)
constexpr int OUT_SEEN = 16;
constexpr size_t WRITE_THROUGH_MAX_BYTES = 160u << 20;   // 4,096 boards' planes are 141 MiB
struct OutSeen { const void* p; size_t bytes; uint64_t when; };

static_assert(GenTuneCore::NC == GOENV_GEN_PACE_CANDIDATES, "include/goenv.h: size of goenv_generate_pace's est_out");
// the tuner's logic (gen_tune.h) + the events of launches whose time has not been read back yet
struct GenTune : GenTuneCore {
  static constexpr int NPEND = 8;
  struct Pend { hipEvent_t e0, e1; int cand, plies, busy; } pend[NPEND];
  int events_ready;
};

struct goenv_ctx {
  int device;
  int n;
  uint64_t seed;
  int64_t id_base;
  float komi;
  int max_ply;
  BoardRec* state;
  unsigned long long* stats;
  float* lut;
  hipEvent_t ev0, ev1;
  int nw;  // waves per board: 0 = by batch size, else 1 / 2 / 3 / 6
  int zero_first;  // -1 = by batch size, 0 / 1 pinned (GOENV_ZERO_FIRST, tuning only)
  int xcd_map;     // 1 = xcd_row dealing of output rows (default); GOENV_XCD_MAP=0 turns it off (tuning only)
  float lut1;
  int fuse_max;    // goenv_ply_policy_sel: largest selection whose sampler runs inside the ply kernel
  int64_t* sampled;      // [n] moves of the two-launch form of goenv_ply_policy_sel (allocated by goenv_create)
  hipStream_t part_stream[GOENV_MAX_PARTS];  // goenv_ply_repeat_parts; created on first use
  hipEvent_t part_event[GOENV_MAX_PARTS + 1];
  int parts_ready;
  // goenv_stage_acquire: ring of pinned host blocks that kernels read in place
  void* stage_host[GOENV_STAGE_SLOTS];
  size_t stage_cap[GOENV_STAGE_SLOTS];
  hipEvent_t stage_event[GOENV_STAGE_SLOTS];
  int stage_busy[GOENV_STAGE_SLOTS];   // handed to a kernel since it was acquired last
  int stage_recorded[GOENV_STAGE_SLOTS];   // ... and stage_event[i] has been recorded behind that kernel
  void* stage_stream[GOENV_STAGE_SLOTS];   // the stream of that kernel
  int stage_next;
  int32_t* bad_moves;                  // device counter of goenv_offline_batch rows that hit a rejected move
  int planes_preset;                   // goenv_set_constant_planes_preset
  int gen_pipe;                        // goenv_generate through k_generate_pipe: -1 = where all boards are resident
                                       // (default), 0 / 1 pinned (GOENV_GEN_PIPE)
  int gen_pace_ns;                     // its rate limiter, ns between two samples of a board: -1 = tuned (default),
                                       // else pinned, 0 = none (GOENV_GEN_PACE_NS)
  int gen_resident;                    // boards of k_generate_pipe that the device holds at once (0: not asked yet)
  int wall_khz;                        // rate of the wall clock that the rate limiter reads (s_memrealtime: 100 MHz on gfx950)
  int gen_last_pipe;                   // the last goenv_generate went through k_generate_pipe (1) / _multi (2)
  int gen_multi_m, gen_multi_grid;     // k_generate_pipe_multi: boards per workgroup (0: not asked yet, -1: none fits), grid
  int gen_chunk_n;                     // more boards than k_generate_pipe_multi holds: launches over this many boards each, one after
  int gen_last_m, gen_last_grid;       // the other (0: the whole context in one launch); the shape of the last, shorter one
  int gen_chunk_min_plies;             // ... for launches of at least this many plies (8; GOENV_GEN_CHUNK_MIN_PLIES), else k_generate
  GenTune tune;
  OutSeen out_seen[OUT_SEEN];          // plane outputs of the last launches (planes_in_flight)
  uint64_t out_clock;
};

// The standalone feature kernel has no rules phase that staggers its waves' stores: mid-size
// launches do best with one wave per board (8,192 boards: 60 us vs 70 us with three).
static int pick_nw_features(const goenv_ctx* c, int count) {
  if (c->nw) return c->nw;
  if (count <= 512) return 6;
  if (count <= 4096) return 3;
  if (count <= 8192) return 1;   // with the per-XCD row dealing three waves win again from 16,384 boards on
  return 3;
}

// Bytes of planes that the last OUT_SEEN launches wrote to DISTINCT buffers, this launch included: what has to stay in the
// 256-MB Infinity Cache for write-through plane stores to pay.  A caller that writes every step's planes to the same batch
// tensor (the collectors of a GameContext) keeps one entry; one that rotates through a ring of outputs, or steps the batch
// in parts, adds them up (tools/time_parts_ring.py: 4,096 boards into a ring of eight buffers 61 us per ply with
// write-through stores, 34 us with plain ones; 8,192 boards as two parts 89 -> 56 us).
static size_t planes_in_flight(goenv_ctx* c, const float* feat, int count) {
  const uint64_t now = ++c->out_clock;
  int slot = -1, oldest = 0;
  for (int i = 0; i < OUT_SEEN; ++i) {
    if (c->out_seen[i].p == feat) slot = i;
    if (c->out_seen[i].when < c->out_seen[oldest].when) oldest = i;
  }
  if (slot < 0) slot = oldest;
  c->out_seen[slot].p = feat;
  c->out_seen[slot].bytes = (size_t)count * GOENV_FEAT_FLOATS * sizeof(float);
  c->out_seen[slot].when = now;
  size_t sum = 0;
  for (int i = 0; i < OUT_SEEN; ++i)
    if (c->out_seen[i].p && now - c->out_seen[i].when < (uint64_t)OUT_SEEN) sum += c->out_seen[i].bytes;
  return sum;
}

// How the planes are stored: bit 0 = the 11 constant planes first (tuning knob, no gain
// measured), bit 1 = write-through (`sc1`) plane stores.  Write-through pays for launches whose
// whole output fits the caches (no end-of-kernel write-back of tens of MB: 13.2 -> 12.1 us at
// 1,024 boards) and costs 1.5x on an HBM-sized stream, so it is used for launches of up to 4,096 boards.
static int pick_zero_first(goenv_ctx* c, int count, const float* feat_out) {
  // tools/tune_waves.py under GOENV_ZERO_FIRST=0 / 2: write-through wins up to 6,144 boards (16.5 -> 14.6 us at
  // 1,536, 18.8 -> 17.0 at 2,048, 31.5 -> 30.5 at 4,096, 42.9 -> 42.4 at 6,144) and loses badly once a ply's planes
  // no longer fit the 256-MB Infinity Cache (8,192 boards = 296 MB: 57 -> 89 us) -- or the planes of the last few
  // launches together, when they go to different buffers (planes_in_flight)
  bool through = count <= 4096;
  if (feat_out && planes_in_flight(c, feat_out, count) > WRITE_THROUGH_MAX_BYTES) through = false;
  // bit 2 (+ bit 0: write_features then skips them too): the caller keeps the 11 constant planes zero itself
  if (c->planes_preset) return ((c->zero_first >= 0 ? c->zero_first : (through ? 2 : 0)) & 2) | 5;
  if (c->zero_first >= 0) return c->zero_first;
  return through ? 2 : 0;
}

// Waves per board of a launch over `count` boards.  Small launches are bound by the dependent
// chain of one board, which more waves per board shorten (1,024 boards: 17.4 us with one wave,
// 13.2 us with three); large ones are HBM-bound and still gain from three waves sharing the plane
// stores of a board (65,536 boards: 585 us with two waves, 492 us with three).
static int pick_nw(const goenv_ctx* c, int count) {
  if (c->nw) return c->nw;   // measured: tools/tune_waves.py, profiles/r01_tune_waves.txt
  // six waves while every board has a CU to itself (256 boards: 8.0 us against 8.9 with three); from 384 boards on three
  // (384: 9.7 against 10.0 with six, 512: 9.95 against 10.1; whole playouts 1.37 against 1.54 ms)
  if (count <= 256) return 6;
  return 3;
}

// Launches that write no planes (goenv_step, goenv_playout, goenv_replay, the moves of tree-search leaves, an offline batch
// without `s`) have no store stream for several waves to share; once the device is full they are bound by instruction
// issue, and one wave per board issues least (no barriers, the scalar mask arithmetic once instead of once per wave).
// Measured (tools/playout_prof.py, tools/time_rules_waves.py; 1 / 2 / 3 waves): whole playouts 65,536 boards 763 / 607 /
// 482 M moves/s, 8,192 boards 587 / 515 / 438, 4,096 boards 495 / 459 / 384, 3,072 boards 444 / 443 / 365, 2,048 boards
// 343 / 381 / 347, 1,536 boards 261 / 292 / 305, 1,024 boards 210 / 264 / 266; one rules-only ply per launch 65,536 boards
// 185 / 191 / 234 us, 8,192 boards 28.6 / 29.6 / 34.3, 4,096 boards 16.4 / 17.4 / 19.6, 2,048 boards 12.0 / 10.9 / 11.4,
// 1,024 boards 10.5 / 8.4 / 8.2; goenv_score 65,536 boards 108 / 117 / 142 us, 16,384 boards 34 / 34 / 39 (tools/time_score_waves.py).
static int pick_nw_rules(const goenv_ctx* c, int count) {
  if (c->nw) return c->nw;
  if (count <= 256) return 6;
  if (count <= 1536) return 3;
  if (count < 3072) return 2;
  return 1;
}

// the stats vector is followed by one scratch word (the MAX slot travels through it in
// goenv_stats_allreduce)
constexpr int STATS_ALLOC = GOENV_NUM_STATS + 2;

static thread_local std::string g_err;

static int fail(const char* what, hipError_t e = hipSuccess) {
  g_err = what;
  if (e != hipSuccess) {
    g_err += ": ";
    g_err += hipGetErrorString(e);
  }
  return -1;
}

#define HIPCHK(x)                                  \
  do {                                             \
    hipError_t e_ = (x);                           \
    if (e_ != hipSuccess) return fail(#x, e_);     \
  } while (0)

static inline hipStream_t S(void* s) { return (hipStream_t)s; }

// Entry points that launch or copy run with the context's device current and put the caller's
// device back on return (one host thread may drive several GPUs; a kernel launched on the null
// stream of another device would be handed pointers that device cannot reach).
struct DeviceScope {
  int prev = -1;
  bool switched = false;
  hipError_t err;
  explicit DeviceScope(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) {
      err = hipSetDevice(dev);
      switched = err == hipSuccess;
    }
  }
  ~DeviceScope() {
    if (switched) (void)hipSetDevice(prev);
  }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
};
#define ON_DEVICE_OF(c)                       \
  DeviceScope dev_scope_((c)->device);        \
  if (dev_scope_.err != hipSuccess) return fail("selecting the context's device", dev_scope_.err)

extern "C" {

int goenv_version(void) { return 4; }

const char* goenv_last_error(void) { return g_err.c_str(); }

int goenv_create(int device, int n_boards, uint64_t seed, int64_t board_id_base, float komi, int max_ply,
                 goenv_ctx** out) {
  if (!out) return fail("goenv_create: out is NULL");
  *out = nullptr;
  if (n_boards <= 0 || n_boards > (1 << 24)) return fail("goenv_create: n_boards out of range");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail("goenv_create: no such HIP device");
  DeviceScope scope(device);
  if (scope.err != hipSuccess) return fail("goenv_create: hipSetDevice", scope.err);
  goenv_ctx* c = new goenv_ctx();  // value-initialised: every pointer and handle starts as zero
  c->device = device; c->n = n_boards; c->seed = seed; c->id_base = board_id_base;
  c->komi = komi; c->max_ply = max_ply; c->nw = 0; c->zero_first = -1;
  if (const char* e = getenv("GOENV_ZERO_FIRST")) c->zero_first = atoi(e) & 3;
  c->xcd_map = 1;
  if (const char* e = getenv("GOENV_XCD_MAP")) c->xcd_map = atoi(e) ? 1 : 0;
  // measured (tools/time_api_sweep.py, profiles/r02_api_sweep.txt): up to 3,072 boards one launch wins
  // (13.1 against 15.3 us at 1,024, 18.1 against 20.3 at 2,048, 28.4 against 28.3 at 3,072); beyond, the policy
  // row is one more read in front of every workgroup's chain while the chip is saturated with plane stores, and
  // the sampler as a small launch of its own is faster (4,096: 34.2 against 35.0 us; 8,192: 59.7 against 78.2)
  c->fuse_max = 3072;
  if (const char* e = getenv("GOENV_FUSE_MAX")) c->fuse_max = atoi(e);
  c->gen_pipe = -1;
  c->gen_pace_ns = -1;
  if (const char* e = getenv("GOENV_GEN_PIPE")) c->gen_pipe = atoi(e) ? 1 : 0;
  if (const char* e = getenv("GOENV_GEN_PACE_NS")) c->gen_pace_ns = atoi(e) < 0 ? -1 : atoi(e);
  if (const char* e = getenv("GOENV_WAVES_PER_BOARD")) {
    const int v = atoi(e);
    if (v == 1 || v == 2 || v == 3 || v == 6) c->nw = v;
  }
  // GetHistoryExp, go/board_feature.cc:141: exp((last_placed - ply) / 10.0) in double, narrowed to
  // float -- evaluated on the host with the same libm call so no device exp rounding enters.
  std::vector<float> lut(HIST_LUT_N);
  for (int d = 0; d < HIST_LUT_N; ++d) lut[d] = (float)exp((double)(-d) / 10.0);
  lut[HIST_LUT_N - 1] = 0.f;  // ages >= 2047 plies: exp(-204.7) underflows float to 0
  for (int d = PIPE_LUT_N - 1; d < HIST_LUT_N; ++d)
    if (lut[d] != 0.f) { delete c; return fail("goenv_create: history table is not zero beyond PIPE_LUT_N"); }
  c->lut1 = lut[1];
  hipError_t e;
  const char* what = "goenv_create: hipMalloc";
  if ((e = hipMalloc((void**)&c->state, (size_t)n_boards * sizeof(BoardRec))) == hipSuccess &&
      (e = hipMalloc((void**)&c->stats, STATS_ALLOC * sizeof(unsigned long long))) == hipSuccess &&
      (e = hipMalloc((void**)&c->lut, HIST_LUT_N * sizeof(float))) == hipSuccess &&
      // scratch of the step path, held from the start so that no step ever allocates (a step must be
      // capturable into a HIP graph): the sampled moves of goenv_ply_policy_sel's two-launch form and the
      // rejected-move counter of goenv_offline_batch
      (e = hipMalloc((void**)&c->sampled, (size_t)n_boards * sizeof(int64_t))) == hipSuccess &&
      (e = hipMalloc((void**)&c->bad_moves, sizeof(int32_t))) == hipSuccess) {
    what = "goenv_create: initialising the boards";
    if ((e = hipMemcpy(c->lut, lut.data(), HIST_LUT_N * sizeof(float), hipMemcpyHostToDevice)) == hipSuccess &&
        (e = hipMemset(c->stats, 0, STATS_ALLOC * sizeof(unsigned long long))) == hipSuccess &&
        (e = hipMemset(c->bad_moves, 0, sizeof(int32_t))) == hipSuccess &&
        (e = hipMemset(c->state, 0, (size_t)n_boards * sizeof(BoardRec))) == hipSuccess &&
        (e = hipEventCreate(&c->ev0)) == hipSuccess && (e = hipEventCreate(&c->ev1)) == hipSuccess) {
      GO_LAUNCH(pick_nw(c, n_boards), k_reset, n_boards, 0, c->state, n_boards, nullptr, n_boards, nullptr);
      if ((e = hipGetLastError()) == hipSuccess) e = hipDeviceSynchronize();
    }
  }
  if (e != hipSuccess) {  // nothing of a half-built context is left behind
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->state) (void)hipFree(c->state);
    if (c->stats) (void)hipFree(c->stats);
    if (c->lut) (void)hipFree(c->lut);
    if (c->sampled) (void)hipFree(c->sampled);
    if (c->bad_moves) (void)hipFree(c->bad_moves);
    delete c;
    return fail(what, e);
  }
  *out = c;
  return 0;
}

int goenv_destroy(goenv_ctx* c) {
  if (!c) return 0;
  DeviceScope scope(c->device);
  (void)hipDeviceSynchronize();
  (void)hipFree(c->state);
  (void)hipFree(c->stats);
  (void)hipFree(c->lut);
  if (c->sampled) (void)hipFree(c->sampled);
  (void)hipEventDestroy(c->ev0);
  (void)hipEventDestroy(c->ev1);
  for (int i = 0; i < GOENV_STAGE_SLOTS; ++i) {
    if (c->stage_host[i]) (void)hipHostFree(c->stage_host[i]);
    if (c->stage_event[i]) (void)hipEventDestroy(c->stage_event[i]);
  }
  if (c->bad_moves) (void)hipFree(c->bad_moves);
  if (c->tune.events_ready)
    for (int i = 0; i < GenTune::NPEND; ++i) { (void)hipEventDestroy(c->tune.pend[i].e0); (void)hipEventDestroy(c->tune.pend[i].e1); }
  if (c->parts_ready) {
    for (int i = 0; i < GOENV_MAX_PARTS; ++i) (void)hipStreamDestroy(c->part_stream[i]);
    for (int i = 0; i <= GOENV_MAX_PARTS; ++i) (void)hipEventDestroy(c->part_event[i]);
  }
  delete c;
  return 0;
}

int goenv_num_boards(const goenv_ctx* c) { return c ? c->n : 0; }

int goenv_set_waves_per_board(goenv_ctx* c, int nw) {
  if (!c) return fail("goenv_set_waves_per_board: ctx is NULL");
  if (!(nw == 0 || nw == 1 || nw == 2 || nw == 3 || nw == 6))
    return fail("goenv_set_waves_per_board: waves per board must be 0 (automatic), 1, 2, 3 or 6");
  c->nw = nw;
  return 0;
}
void* goenv_state_ptr(goenv_ctx* c) { return c ? (void*)c->state : nullptr; }
int goenv_get_fuse_max(const goenv_ctx* c) { return c ? c->fuse_max : 0; }

int goenv_set_constant_planes_preset(goenv_ctx* c, int on) {
  if (!c) return fail("goenv_set_constant_planes_preset: ctx is NULL");
  c->planes_preset = on ? 1 : 0;
  return 0;
}

int goenv_reset(goenv_ctx* c, const int32_t* idx, int n, const int32_t* handicap, void* stream) {
  if (!c) return fail("goenv_reset: ctx is NULL");
  ON_DEVICE_OF(c);
  const int cnt = idx ? n : c->n;
  if (cnt <= 0) return idx ? 0 : fail("goenv_reset: empty context");
  if (idx && n > c->n) return fail("goenv_reset: n exceeds the number of boards");
  GO_LAUNCH(pick_nw(c, cnt), k_reset, cnt, S(stream), c->state, c->n, idx, cnt, handicap);
  HIPCHK(hipGetLastError());
  return 0;
}

int goenv_get_state(goenv_ctx* c, uint8_t* color, int16_t* liberties, int16_t* stones, uint16_t* last_placed,
                    int32_t* scalars, void* stream) {
  if (!c) return fail("goenv_get_state: ctx is NULL");
  ON_DEVICE_OF(c);
  GO_LAUNCH(pick_nw(c, c->n), k_get_state, c->n, S(stream), c->state, c->n, nullptr, 0, c->n, color, liberties, stones,
            last_placed, scalars);
  HIPCHK(hipGetLastError());
  return 0;
}

int goenv_set_state(goenv_ctx* c, const uint8_t* color, const uint16_t* last_placed, const int32_t* scalars,
                    void* stream) {
  if (!c) return fail("goenv_set_state: ctx is NULL");
  ON_DEVICE_OF(c);
  if (!color || !scalars) return fail("goenv_set_state: color and scalars are required");
  GO_LAUNCH(pick_nw(c, c->n), k_set_state, c->n, S(stream), c->state, c->n, color, last_placed, scalars);
  HIPCHK(hipGetLastError());
  return 0;
}

static PlyParams make_params(goenv_ctx* c, uint32_t flags, const int64_t* actions, const int32_t* aug, float* feat,
                             uint8_t* legal, int32_t* actions_out, int32_t* status) {
  PlyParams P;
  memset(&P, 0, sizeof(P));
  P.state = c->state; P.n = c->n; P.n_total = c->n; P.idx = nullptr; P.first = 0;
  P.flags = flags; P.actions = actions; P.aug = aug; P.feat = feat;
  P.legal = legal; P.actions_out = actions_out; P.status = status; P.stats = c->stats; P.lut = c->lut;
  P.seed = c->seed; P.id_base = c->id_base; P.komi = c->komi; P.max_ply = c->max_ply;
  P.stamps = nullptr;
  P.lut1 = c->lut1;
  P.zero_first = pick_zero_first(c, c->n, nullptr);
  P.xcd_map = c->xcd_map;
  return P;
}

static int check_sel(goenv_ctx* c, const int32_t* idx, int first, int count, const char* who) {
  if (!c) { fail(who); g_err += ": ctx is NULL"; return -1; }
  if (count < 0 || count > c->n) { fail(who); g_err += ": count out of range"; return -1; }
  if (!idx && (first < 0 || first + count > c->n)) { fail(who); g_err += ": board range out of bounds"; return -1; }
  return 0;
}

static int ply_sel_impl(goenv_ctx* c, const char* who, const int32_t* idx, int first, int count, uint32_t flags,
                        const int64_t* actions, const float* pi, const int32_t* pi_aug, const int32_t* aug_codes,
                        float* feat_out, uint8_t* legal_out, const goenv_sample_out* meta, int32_t* actions_out,
                        int32_t* status_out, void* stream) {
  if (check_sel(c, idx, first, count, who)) return -1;
  ON_DEVICE_OF(c);
  int mode = pi ? 2 : ((flags & GOENV_F_RANDOM_POLICY) ? 1 : 0);
  if (mode == 0 && !actions) return fail("goenv_ply: actions required unless GOENV_F_RANDOM_POLICY is set");
  if ((flags & GOENV_F_COLLECT) && !meta) return fail("goenv_ply: GOENV_F_COLLECT needs a goenv_sample_out");
  if (count == 0) return 0;
  if (mode == 2 && count > c->fuse_max) {
    // large selections: the sampler as a launch of its own (one wave per board, all resident at once),
    // then the ply kernel on the sampled moves -- same moves, same outputs as the fused kernel
    hipLaunchKernelGGL(k_sample_policy, dim3((count + SAMPLER_WAVES - 1) / SAMPLER_WAVES), dim3(64 * SAMPLER_WAVES), 0, S(stream), c->state, c->n, idx, first, count, pi,
                       pi_aug, c->sampled, c->seed, c->id_base);
    HIPCHK(hipGetLastError());
    actions = c->sampled; pi = nullptr; pi_aug = nullptr; mode = 0;
  }
  PlyParams P = make_params(c, flags, actions, aug_codes, feat_out, legal_out, actions_out, status_out);
  P.n = count; P.idx = idx; P.first = first;
  P.pi = pi; P.pi_aug = pi_aug;
  P.zero_first = pick_zero_first(c, count, feat_out);
  if (meta) P.meta = *meta;
  GO_LAUNCH_PLY(feat_out ? pick_nw(c, count) : pick_nw_rules(c, count), mode, count, S(stream), P);
  HIPCHK(hipGetLastError());
  return 0;
}

int goenv_ply_sel(goenv_ctx* c, const int32_t* idx, int first, int count, uint32_t flags, const int64_t* actions,
                  const int32_t* aug_codes, float* feat_out, uint8_t* legal_out, const goenv_sample_out* meta,
                  int32_t* actions_out, int32_t* status_out, void* stream) {
  return ply_sel_impl(c, "goenv_ply_sel", idx, first, count, flags, actions, nullptr, nullptr, aug_codes, feat_out,
                      legal_out, meta, actions_out, status_out, stream);
}

int goenv_ply_policy_sel(goenv_ctx* c, const int32_t* idx, int first, int count, uint32_t flags, const float* pi,
                         const int32_t* pi_aug_codes, const int32_t* aug_codes, float* feat_out, uint8_t* legal_out,
                         const goenv_sample_out* meta, int32_t* actions_out, int32_t* status_out, void* stream) {
  if (!pi) return fail("goenv_ply_policy_sel: pi is required");
  if (flags & (GOENV_F_RANDOM_POLICY | GOENV_F_AUTO_RESTART))
    return fail("goenv_ply_policy_sel: GOENV_F_RANDOM_POLICY and GOENV_F_AUTO_RESTART do not apply (the move is "
                "sampled for the stored position; restart finished games with GOENV_F_RESTART_AFTER_END)");
  return ply_sel_impl(c, "goenv_ply_policy_sel", idx, first, count, flags, nullptr, pi, pi_aug_codes, aug_codes,
                      feat_out, legal_out, meta, actions_out, status_out, stream);
}

int goenv_ply(goenv_ctx* c, uint32_t flags, const int64_t* actions, const int32_t* aug_codes, float* feat_out,
              uint8_t* legal_out, int32_t* actions_out, int32_t* status_out, void* stream) {
  if (!c) return fail("goenv_ply: ctx is NULL");
  return goenv_ply_sel(c, nullptr, 0, c->n, flags & ~GOENV_F_COLLECT, actions, aug_codes, feat_out, legal_out,
                       nullptr, actions_out, status_out, stream);
}

int goenv_ply_repeat(goenv_ctx* c, int n_steps, uint32_t flags, const int64_t* actions, const int32_t* aug_codes,
                     float* feat_out, uint8_t* legal_out, int32_t* actions_out, int32_t* status_out, void* stream) {
  if (n_steps < 0) return fail("goenv_ply_repeat: n_steps is negative");
  for (int i = 0; i < n_steps; ++i) {
    const int rc = goenv_ply(c, flags, actions, aug_codes, feat_out, legal_out, actions_out, status_out, stream);
    if (rc) return rc;
  }
  return 0;
}

// n_steps plies of every board with the batch cut into n_parts contiguous parts, each part a
// launch sequence of its own on a stream of its own: the parts drift out of phase, so the
// latency-bound front of one part's launch (launch, record round trip, rules) runs under the
// store drain of another's.  Fork / join around the whole sequence on the caller's stream.
int goenv_ply_repeat_parts(goenv_ctx* c, int n_steps, int n_parts, uint32_t flags, const int64_t* actions,
                           const int32_t* aug_codes, float* feat_out, uint8_t* legal_out, int32_t* actions_out,
                           int32_t* status_out, void* stream) {
  if (!c) return fail("goenv_ply_repeat_parts: ctx is NULL");
  if (n_steps < 0) return fail("goenv_ply_repeat_parts: n_steps is negative");
  if (n_parts < 1 || n_parts > GOENV_MAX_PARTS || n_parts > c->n)
    return fail("goenv_ply_repeat_parts: n_parts must be in 1..min(GOENV_MAX_PARTS, boards)");
  if (n_parts == 1) return goenv_ply_repeat(c, n_steps, flags, actions, aug_codes, feat_out, legal_out, actions_out,
                                            status_out, stream);
  ON_DEVICE_OF(c);
  if (!c->parts_ready) {
    for (int i = 0; i < GOENV_MAX_PARTS; ++i) HIPCHK(hipStreamCreateWithFlags(&c->part_stream[i], hipStreamNonBlocking));
    for (int i = 0; i <= GOENV_MAX_PARTS; ++i) HIPCHK(hipEventCreateWithFlags(&c->part_event[i], hipEventDisableTiming));
    c->parts_ready = 1;
  }
  HIPCHK(hipEventRecord(c->part_event[GOENV_MAX_PARTS], S(stream)));
  for (int p = 0; p < n_parts; ++p) HIPCHK(hipStreamWaitEvent(c->part_stream[p], c->part_event[GOENV_MAX_PARTS], 0));
  for (int i = 0; i < n_steps; ++i) {
    for (int p = 0; p < n_parts; ++p) {
      const int first = (int)((long long)c->n * p / n_parts), last = (int)((long long)c->n * (p + 1) / n_parts);
      const int rc = goenv_ply_sel(c, nullptr, first, last - first, flags & ~GOENV_F_COLLECT,
                                   actions ? actions + first : nullptr, aug_codes ? aug_codes + first : nullptr,
                                   feat_out ? feat_out + (size_t)first * GOENV_FEAT_FLOATS : nullptr,
                                   legal_out ? legal_out + (size_t)first * GOENV_NUM_POINTS : nullptr, nullptr,
                                   actions_out ? actions_out + first : nullptr, status_out ? status_out + first : nullptr,
                                   c->part_stream[p]);
      if (rc) return rc;
    }
  }
  for (int p = 0; p < n_parts; ++p) {
    HIPCHK(hipEventRecord(c->part_event[p], c->part_stream[p]));
    HIPCHK(hipStreamWaitEvent(S(stream), c->part_event[p], 0));
  }
  return 0;
}

#ifdef GOENV_STAMPS
// Diagnostic build only: stamps of one fused selfplay step (goenv_ply_policy_sel over all boards).
extern "C" int goenv_diag_ply_policy(goenv_ctx* c, uint32_t flags, const float* pi, float* feat_out, uint8_t* legal_out,
                                     unsigned long long* stamps, void* stream) {
  PlyParams P = make_params(c, flags, nullptr, nullptr, feat_out, legal_out, nullptr, nullptr);
  P.pi = pi;
  P.stamps = stamps;
  GO_LAUNCH_PLY(pick_nw(c, c->n), 2, c->n, S(stream), P);
  HIPCHK(hipGetLastError());
  return 0;
}
// Diagnostic build only: stamps [n][2 waves][16] of the LAST ply of a generator launch (slots 8..12: loop
// top, rules + masks done, masks gathered, legal written, planes issued; 2, 3 inside the rules).
extern "C" int goenv_diag_generate(goenv_ctx* c, int n_plies, float* feat_ring, uint8_t* legal_ring,
                                   unsigned long long* stamps, void* stream) {
  PlyParams P = make_params(c, GOENV_F_RANDOM_POLICY | GOENV_F_AUTO_RESTART, nullptr, nullptr, feat_ring, legal_ring,
                            nullptr, nullptr);
  P.stamps = stamps;
  const int nw = c->nw ? c->nw : (c->n <= 4096 ? 2 : 3);
  if (const char* e = getenv("GOENV_GEN_MULTI_M")) {   // stamps of k_generate_pipe_multi with M boards per workgroup
    const int m = atoi(e), grid = (((c->n + m - 1) / m + 7) / 8) * 8;
    hipLaunchKernelGGL(k_generate_pipe_multi, dim3(grid), dim3(128), (size_t)m * sizeof(SavedBoard), S(stream), P, n_plies,
                       (int32_t*)nullptr, (c->gen_pace_ns > 0 ? c->gen_pace_ns : 0) / 10);
  } else if (c->gen_pipe > 0) hipLaunchKernelGGL(k_generate_pipe, dim3(c->n), dim3(128), 0, S(stream), P, n_plies, (int32_t*)nullptr, (c->gen_pace_ns > 0 ? c->gen_pace_ns : 0) / 10);
  else GO_LAUNCH(nw, k_generate, c->n, S(stream), P, n_plies, (int32_t*)nullptr);
  HIPCHK(hipGetLastError());
  return 0;
}
// Diagnostic build only (libgoenv_diag.so, tools/stamps.py): per-board s_memtime stamps
// [n][2 waves][16] of one fused random ply.  Never part of libgoenv.so.
extern "C" int goenv_diag_ply(goenv_ctx* c, uint32_t flags, float* feat_out, uint8_t* legal_out,
                              unsigned long long* stamps, void* stream) {
  PlyParams P = make_params(c, flags | GOENV_F_RANDOM_POLICY, nullptr, nullptr, feat_out, legal_out, nullptr, nullptr);
  P.stamps = stamps;
  GO_LAUNCH_PLY(pick_nw(c, c->n), 1, c->n, S(stream), P);
  HIPCHK(hipGetLastError());
  return 0;
}
#endif

int goenv_step(goenv_ctx* c, const int64_t* actions, uint8_t* legal_out, int32_t* status_out, void* stream) {
  return goenv_ply(c, 0u, actions, nullptr, nullptr, legal_out, nullptr, status_out, stream);
}

int goenv_collect_sel(goenv_ctx* c, const int32_t* idx, int first, int count, const int32_t* aug_codes,
                      float* feat_out, uint8_t* legal_out, const goenv_sample_out* meta, void* stream) {
  if (check_sel(c, idx, first, count, "goenv_collect_sel")) return -1;
  ON_DEVICE_OF(c);
  if (count == 0) return 0;
  goenv_sample_out m;
  memset(&m, 0, sizeof(m));
  if (meta) m = *meta;
  // From 16,384 rows on a launch of free-running workgroups offers the memory side more stores than it takes well: 16,000 B
  // of unused LDS per workgroup hold a CU at 7 boards instead of 10 (65,536 boards: 460 -> 443 us, 32,768: 231 -> 225,
  // 16,384: 118 -> 116; profiles/r03_occupancy_cap_probe.txt).  The ply kernel loses from the same cap and keeps none.
  const int nw = pick_nw_features(c, count);
  const size_t lds_pad = g_lds_pad ? g_lds_pad : ((count >= 16384 && nw == 3 && feat_out) ? 16000 : 0);
  GO_LAUNCH_LDS(nw, k_features, count, lds_pad, S(stream), c->state, c->n, idx, first, count, aug_codes, c->lut,
                feat_out, legal_out, m, meta ? 1 : 0, c->lut1, pick_zero_first(c, count, feat_out), c->xcd_map);
  HIPCHK(hipGetLastError());
  return 0;
}

int goenv_features(goenv_ctx* c, const int32_t* aug_codes, float* feat_out, uint8_t* legal_out, void* stream) {
  if (!c) return fail("goenv_features: ctx is NULL");
  if (!feat_out && !legal_out) return fail("goenv_features: no output buffer");
  return goenv_collect_sel(c, nullptr, 0, c->n, aug_codes, feat_out, legal_out, nullptr, stream);
}

// ---- goenv_features_static: the one entry point that allocates nothing ------------------------
// Positions in, planes + legal mask out, with no context: the packed records that k_set_state
// builds and k_features reads, and the history table, live in `scratch`, a file-scope device array
// (one copy per GPU, held until the process ends).  Host code never uses the array's NAME as an
// address: g_static[dev].base is what hipGetSymbolAddress returned on that device.
constexpr size_t STATIC_LUT_BYTES = HIST_LUT_N * sizeof(float);  // 8 KB, keeps the records 256-B aligned
static __device__ __align__(256) unsigned char scratch[STATIC_LUT_BYTES +
                                                       (size_t)GOENV_STATIC_MAX_BOARDS * sizeof(BoardRec)];

struct StaticDev {
  unsigned char* base = nullptr;  // device address of `scratch` on this GPU
  float lut1 = 0.f;
  bool ready = false;
  // in-use guard: the stream of the last call and an event recorded behind its kernels.  A call from
  // ANOTHER stream while that event has not passed would race on the array and is refused.
  hipStream_t last_stream = nullptr;
  hipEvent_t done = nullptr;
  bool armed = false;
};
static StaticDev g_static[64];
static std::mutex g_static_mu;

// Set-up, once per device, with that device current: the array's address there, then the history
// table (the same host-evaluated exp values goenv_create uploads) copied to its first 8 KB.
static int static_setup(int device, hipStream_t stream, StaticDev** out) {
  std::lock_guard<std::mutex> lock(g_static_mu);
  StaticDev& d = g_static[device];
  if (!d.ready) {
    if (stream) {   // the set-up copies synchronously: it cannot be part of a capture
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
        return fail("goenv_features_static: the first call on a device sets the library's array up with a synchronous "
                    "copy; make one call before starting a stream capture");
    }
    void* p = nullptr;
    HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return fail("goenv_features_static: hipGetSymbolAddress returned NULL");
    std::vector<float> lut(HIST_LUT_N);
    for (int i = 0; i < HIST_LUT_N; ++i) lut[i] = (float)exp((double)(-i) / 10.0);
    lut[HIST_LUT_N - 1] = 0.f;
    HIPCHK(hipMemcpy(p, lut.data(), STATIC_LUT_BYTES, hipMemcpyHostToDevice));
    HIPCHK(hipEventCreateWithFlags(&d.done, hipEventDisableTiming));
    d.base = (unsigned char*)p;
    d.lut1 = lut[1];
    d.ready = true;
  }
  *out = &d;
  return 0;
}

int goenv_features_static(int device, int n, const uint8_t* color, const uint16_t* last_placed,
                          const int32_t* scalars, const int32_t* aug_codes, float* feat_out, uint8_t* legal_out,
                          void* stream) {
  if (n < 0 || n > GOENV_STATIC_MAX_BOARDS)
    return fail("goenv_features_static: n exceeds GOENV_STATIC_MAX_BOARDS (4096 positions per call)");
  if (device < 0 || device >= 64) return fail("goenv_features_static: device out of range");
  if (!color || !scalars) return fail("goenv_features_static: color and scalars are required");
  if (!feat_out && !legal_out) return fail("goenv_features_static: no output buffer");
  DeviceScope scope(device);
  if (scope.err != hipSuccess) return fail("goenv_features_static: selecting the device", scope.err);
  StaticDev* d = nullptr;
  if (static_setup(device, S(stream), &d)) return -1;
  if (n == 0) return 0;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (stream) (void)hipStreamIsCapturing(S(stream), &cap);
  const bool capturing = cap != hipStreamCaptureStatusNone;
  if (!capturing) {   // (no event query inside a capture: a captured call is the graph owner's to order)
    // calls on ONE stream are ordered by the stream; a call from another stream (or thread) must find the
    // previous call's kernels finished, or it would overwrite records they are still reading
    std::lock_guard<std::mutex> lock(g_static_mu);
    if (d->armed && d->last_stream != S(stream)) {
      const hipError_t q = hipEventQuery(d->done);
      if (q == hipErrorNotReady)
        return fail("goenv_features_static: the previous call on this GPU (another stream) is still in flight and "
                    "all calls share one array; order the calls (one stream, or wait for the first)");
      if (q != hipSuccess) return fail("goenv_features_static: hipEventQuery", q);
    }
    d->last_stream = S(stream);
    d->armed = false;   // set again below, once this call's kernels are enqueued
  }
  BoardRec* recs = reinterpret_cast<BoardRec*>(d->base + STATIC_LUT_BYTES);
  const float* lut = reinterpret_cast<const float*>(d->base);
  const int nw = n <= 512 ? 6 : 3;
  // every record row [0, n) is rewritten in full by k_set_state before k_features reads it (same
  // stream): nothing of an earlier call is read, so nothing needs clearing
  GO_LAUNCH(nw, k_set_state, n, S(stream), recs, n, color, last_placed, scalars);
  HIPCHK(hipGetLastError());
  goenv_sample_out none;
  memset(&none, 0, sizeof(none));
  GO_LAUNCH(nw, k_features, n, S(stream), recs, n, (const int32_t*)nullptr, 0, n, aug_codes, lut, feat_out, legal_out,
            none, 0, d->lut1, 2, 1);   // write-through plane stores: n <= 4,096 fits the Infinity Cache
  HIPCHK(hipGetLastError());
  if (!capturing) {   // (a captured call is replayed by the graph's owner, who orders its launches)
    std::lock_guard<std::mutex> lock(g_static_mu);
    HIPCHK(hipEventRecord(d->done, S(stream)));
    d->armed = true;
  }
  return 0;
}

int goenv_sample_policy(goenv_ctx* c, const int32_t* idx, int first, int count, const float* pi,
                        const int32_t* aug_codes, int64_t* actions_out, void* stream) {
  if (check_sel(c, idx, first, count, "goenv_sample_policy")) return -1;
  ON_DEVICE_OF(c);
  if (!pi || !actions_out) return fail("goenv_sample_policy: pi and actions_out are required");
  if (count == 0) return 0;
  hipLaunchKernelGGL(k_sample_policy, dim3((count + SAMPLER_WAVES - 1) / SAMPLER_WAVES), dim3(64 * SAMPLER_WAVES), 0, S(stream), c->state, c->n, idx, first, count, pi,
                     aug_codes, actions_out, c->seed, c->id_base);
  HIPCHK(hipGetLastError());
  return 0;
}

int goenv_topk_policy(goenv_ctx* c, const int32_t* idx, int first, int count, const float* pi,
                      const int32_t* aug_codes, int k, int32_t* actions_out, float* probs_out, void* stream) {
  if (check_sel(c, idx, first, count, "goenv_topk_policy")) return -1;
  ON_DEVICE_OF(c);
  if (!pi || !actions_out) return fail("goenv_topk_policy: pi and actions_out are required");
  if (k < 1 || k > GOENV_NUM_POINTS) return fail("goenv_topk_policy: k must be in 1..361");
  if (count == 0) return 0;
  hipLaunchKernelGGL(k_topk_policy, dim3(count), dim3(64), 0, S(stream), c->state, c->n, idx, first, count, pi,
                     aug_codes, k, actions_out, probs_out);
  HIPCHK(hipGetLastError());
  return 0;
}

int goenv_pointer_is_device(const void* p) {
  if (!p) return 0;
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();  // plain malloc memory: not registered with HIP
    return 0;
  }
  return attr.type == hipMemoryTypeDevice ? 1 : 0;
}

// Both run with the context's device current: `stream` (NULL included: every device has a null stream
// of its own) must be the stream the context's kernels were launched on, or nothing orders the two.
int goenv_copy(goenv_ctx* c, void* dst, const void* src, size_t bytes, void* stream) {
  if (!c) return fail("goenv_copy: ctx is NULL");
  ON_DEVICE_OF(c);
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, S(stream)));
  return 0;
}

int goenv_memset(goenv_ctx* c, void* dst, int value, size_t bytes, void* stream) {
  if (!c) return fail("goenv_memset: ctx is NULL");
  ON_DEVICE_OF(c);
  if (bytes) HIPCHK(hipMemsetAsync(dst, value, bytes, S(stream)));
  return 0;
}

int goenv_sync(goenv_ctx* c, void* stream) {
  if (!c) return fail("goenv_sync: ctx is NULL");
  ON_DEVICE_OF(c);
  HIPCHK(hipStreamSynchronize(S(stream)));
  return 0;
}

int goenv_device_alloc(goenv_ctx* c, size_t bytes, void** out) {
  if (!c || !out) return fail("goenv_device_alloc: NULL argument");
  ON_DEVICE_OF(c);
  HIPCHK(hipMalloc(out, bytes ? bytes : 1));
  return 0;
}

int goenv_device_free(goenv_ctx* c, void* p) {
  if (!c) return fail("goenv_device_free: ctx is NULL");
  ON_DEVICE_OF(c);
  if (p) HIPCHK(hipFree(p));
  return 0;
}

int goenv_get_state_sel(goenv_ctx* c, const int32_t* idx, int first, int count, uint8_t* color, int16_t* liberties,
                        int16_t* stones, uint16_t* last_placed, int32_t* scalars, void* stream) {
  if (check_sel(c, idx, first, count, "goenv_get_state_sel")) return -1;
  ON_DEVICE_OF(c);
  if (count == 0) return 0;
  GO_LAUNCH(pick_nw(c, count), k_get_state, count, S(stream), c->state, c->n, idx, first, count, color, liberties,
            stones, last_placed, scalars);
  HIPCHK(hipGetLastError());
  return 0;
}

int goenv_replay(goenv_ctx* c, const int32_t* idx, int first, int count, const int16_t* moves, int stride,
                 const int32_t* nmoves, int32_t* napplied_out, void* stream) {
  if (check_sel(c, idx, first, count, "goenv_replay")) return -1;
  ON_DEVICE_OF(c);
  if (!moves || !nmoves || stride <= 0) return fail("goenv_replay: moves, nmoves and a positive stride are required");
  if (count == 0) return 0;
  PlyParams P = make_params(c, 0u, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  P.n = count; P.idx = idx; P.first = first;
  P.max_ply = 0;  // a replayed record is not subject to the playout ply cap
  GO_LAUNCH(pick_nw_rules(c, count), k_replay, count, S(stream), P, moves, stride, nmoves, napplied_out);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- offline (SGF replay) batches -----------------------------------------------------------------
// Staging ring: GOENV_STAGE_SLOTS pinned host blocks, handed out in turn.  A block is written by the host,
// read IN PLACE by the one kernel launched with it (no copy engine, no extra stream operation in front
// of the kernel: the rows of a batch are a few KB) and handed out again only after that kernel is done.
int goenv_stage_acquire(goenv_ctx* c, size_t bytes, void** host_block) {
  if (!c || !host_block) return fail("goenv_stage_acquire: NULL argument");
  ON_DEVICE_OF(c);
  const int i = c->stage_next;
  if (c->stage_busy[i]) {   // the kernel that read this block last (GOENV_STAGE_SLOTS launches ago)
    // the event behind it: its own, or the one recorded behind a later block of its group on the same stream (stage_commit)
    int j = i;
    const int end = i | (STAGE_GROUP - 1);
    while (j <= end && !(c->stage_recorded[j] && c->stage_stream[j] == c->stage_stream[i])) ++j;
    if (j <= end) HIPCHK(hipEventSynchronize(c->stage_event[j]));
    else HIPCHK(hipDeviceSynchronize());   // (a group cut short: not reached while blocks are committed in ring order)
    c->stage_busy[i] = 0;
  }
  if (c->stage_cap[i] < bytes) {
    if (c->stage_host[i]) HIPCHK(hipHostFree(c->stage_host[i]));
    c->stage_host[i] = nullptr; c->stage_cap[i] = 0;
    const size_t cap = bytes < 65536 ? 65536 : bytes + bytes / 2;
    HIPCHK(hipHostMalloc(&c->stage_host[i], cap, hipHostMallocDefault));
    c->stage_cap[i] = cap;
  }
  if (!c->stage_event[i]) HIPCHK(hipEventCreateWithFlags(&c->stage_event[i], hipEventDisableTiming));
  *host_block = c->stage_host[i];
  return 0;
}

// Marks the block acquired last as read by everything enqueued on `stream` so far.  An event behind a kernel costs the stream
// a release of its own (the generator's tuner found 25 us behind a long kernel), so only every STAGE_GROUP-th block gets one:
// streams run in order, and the event behind block i also says that the readers of the blocks before it on that stream are done.
// A block whose successor goes to another stream gets its event then.
static int stage_commit(goenv_ctx* c, const void* block, void* stream) {
  const int i = c->stage_next;
  if (block != c->stage_host[i]) return 0;   // not a ring block (device memory, or the caller's own)
  c->stage_stream[i] = stream; c->stage_busy[i] = 1; c->stage_recorded[i] = 0;
  if (i % STAGE_GROUP != 0 && c->stage_busy[i - 1] && !c->stage_recorded[i - 1] && c->stage_stream[i - 1] != stream) {
    HIPCHK(hipEventRecord(c->stage_event[i - 1], S(c->stage_stream[i - 1])));
    c->stage_recorded[i - 1] = 1;
  }
  if (i % STAGE_GROUP == STAGE_GROUP - 1) {
    HIPCHK(hipEventRecord(c->stage_event[i], S(stream)));
    c->stage_recorded[i] = 1;
  }
  c->stage_next = (i + 1) % GOENV_STAGE_SLOTS;
  return 0;
}

int goenv_offline_batch(goenv_ctx* c, const void* block, int count, int k_future, size_t moves_offset,
                        float* feat_out, uint8_t* legal_out, const goenv_sample_out* labels, int64_t* offline_a_out,
                        void* stream) {
  if (!c || !block) return fail("goenv_offline_batch: NULL argument");
  if (count < 0 || count > c->n) return fail("goenv_offline_batch: count out of range");
  if (k_future < 0 || k_future > GOENV_MAX_FUTURE) return fail("goenv_offline_batch: k_future out of range");
  if (moves_offset & 1) return fail("goenv_offline_batch: moves_offset must be even");
  ON_DEVICE_OF(c);
  if (count == 0) return stage_commit(c, block, stream);
  const int row_w = OFFLINE_ROW_HEAD + 2 * k_future;
  if (block == c->stage_host[c->stage_next]) {
    // a block of the staging ring is host memory of known size: check what the kernel will index with before
    // it does (an out-of-range row would be a device fault, not an error code)
    const size_t cap = c->stage_cap[c->stage_next];
    if (moves_offset < (size_t)count * row_w * 4 || moves_offset > cap)
      return fail("goenv_offline_batch: moves_offset does not lie behind the rows, inside the block");
    const size_t n_moves_max = (cap - moves_offset) / 2;
    const int32_t* rows = static_cast<const int32_t*>(block);
    for (int r = 0; r < count; ++r) {
      const int32_t* row = rows + (size_t)r * row_w;
      const uint32_t n = (uint32_t)row[1] >> 8, off = (uint32_t)row[2];
      if (row[0] < 0 || row[0] >= c->n) return fail("goenv_offline_batch: a row names a board outside the context");
      if ((size_t)off + n > n_moves_max) return fail("goenv_offline_batch: a row's moves run past the end of the block");
    }
  }
  PlyParams P = make_params(c, 0u, nullptr, nullptr, feat_out, legal_out, nullptr, nullptr);
  P.n = count;
  P.max_ply = 0;  // a replayed record is not subject to the playout ply cap
  P.zero_first = pick_zero_first(c, count, feat_out);
  if (labels) P.meta = *labels;
  const int nw = feat_out ? pick_nw_features(c, count) : pick_nw_rules(c, count);
  GO_LAUNCH(nw, k_offline, count, S(stream), P, (const int32_t*)block, row_w, k_future,
            (const int16_t*)((const char*)block + moves_offset), offline_a_out, c->bad_moves);
  HIPCHK(hipGetLastError());
  return stage_commit(c, block, stream);
}

int goenv_offline_bad_moves(goenv_ctx* c, int32_t* out_host, void* stream) {
  if (!c || !out_host) return fail("goenv_offline_bad_moves: NULL argument");
  ON_DEVICE_OF(c);
  *out_host = 0;
  HIPCHK(hipMemcpyAsync(out_host, c->bad_moves, sizeof(int32_t), hipMemcpyDeviceToHost, S(stream)));
  HIPCHK(hipStreamSynchronize(S(stream)));
  return 0;
}

int goenv_tactics(goenv_ctx* c, const int32_t* idx, int first, int count, int self_atari_thres,
                  const goenv_tactics_out* out, void* stream) {
  if (check_sel(c, idx, first, count, "goenv_tactics")) return -1;
  ON_DEVICE_OF(c);
  if (!out) return fail("goenv_tactics: out is NULL");
  if (self_atari_thres < 0) return fail("goenv_tactics: self_atari_thres must be >= 0");
  if (count == 0) return 0;
  TacticsOut t;
  t.self_atari = out->self_atari; t.gives_ko = out->gives_ko; t.cand = out->cand; t.libs = out->libs;
  t.ladder = out->ladder;
  hipLaunchKernelGGL(k_tactics, dim3(count), dim3(64), 0, S(stream), c->state, c->n, idx, first, count,
                     self_atari_thres, t);
  HIPCHK(hipGetLastError());
  return 0;
}

int goenv_copy_boards(goenv_ctx* c, const int32_t* src_idx, const int32_t* dst_idx, int n, void* stream) {
  if (!c) return fail("goenv_copy_boards: ctx is NULL");
  if (n < 0 || n > c->n) return fail("goenv_copy_boards: n out of range");
  ON_DEVICE_OF(c);
  if (n == 0) return 0;
  if (!src_idx || !dst_idx) return fail("goenv_copy_boards: src_idx and dst_idx are required");
  hipLaunchKernelGGL(k_copy_boards, dim3(n), dim3(128), 0, S(stream), c->state, c->n, src_idx, dst_idx, n);
  HIPCHK(hipGetLastError());
  return 0;
}

int goenv_copy_rows(goenv_ctx* c, const void* src, const int32_t* src_rows, void* dst, const int32_t* dst_rows, int n,
                    size_t row_bytes, void* stream) {
  if (!c) return fail("goenv_copy_rows: ctx is NULL");
  if (n < 0 || row_bytes == 0) return fail("goenv_copy_rows: n must be >= 0 and row_bytes positive");
  if (n == 0) return 0;
  if (!src || !dst) return fail("goenv_copy_rows: src and dst are required");
  ON_DEVICE_OF(c);
  hipLaunchKernelGGL(k_copy_rows, dim3(n), dim3(256), 0, S(stream), (const unsigned char*)src, src_rows,
                     (unsigned char*)dst, dst_rows, n, row_bytes);
  HIPCHK(hipGetLastError());
  return 0;
}

int goenv_host_alloc(size_t bytes, void** out) {
  if (!out) return fail("goenv_host_alloc: out is NULL");
  HIPCHK(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
  return 0;
}

int goenv_host_free(void* p) {
  if (p) HIPCHK(hipHostFree(p));
  return 0;
}

int goenv_playout(goenv_ctx* c, int16_t* moves_out, int max_record, int32_t* nmoves_out, void* stream) {
  if (!c) return fail("goenv_playout: ctx is NULL");
  ON_DEVICE_OF(c);
  if (c->max_ply <= 0) return fail("goenv_playout: the context needs a ply cap (max_ply > 0)");
  if (moves_out && max_record <= 0) return fail("goenv_playout: max_record must be positive");
  PlyParams P = make_params(c, GOENV_F_RANDOM_POLICY, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  GO_LAUNCH(pick_nw_rules(c, c->n), k_playout, c->n, S(stream), P, moves_out, max_record, nmoves_out);
  HIPCHK(hipGetLastError());
  return 0;
}

// The cheapest launch shape of the pipelined generator over n boards by the cost model described in goenv_generate:
// m = 1 is k_generate_pipe, 2 .. 16 boards per workgroup k_generate_pipe_multi, -1 = none fits; cost in ns per ply.
struct GenShape { int m, grid; double cost; };
static int gen_shape(goenv_ctx* c, int n, int cus, GenShape* out) {
  out->m = -1; out->grid = 0; out->cost = 0;
  int forced = 0;
  if (const char* e = getenv("GOENV_GEN_MULTI_M")) forced = atoi(e);   // tuning only
  for (int m = 1; m <= 16; ++m) {
    if (forced && m != forced) continue;
    const int grid = m == 1 ? n : (((n + m - 1) / m + 7) / 8) * 8;
    int per_cu = 0;
    if (m == 1) per_cu = c->gen_resident > 0 ? c->gen_resident / (cus > 0 ? cus : 1) : 0;
    else HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_generate_pipe_multi, 128, (size_t)m * sizeof(SavedBoard)));
    if (grid > per_cu * cus || (m > 1 && !forced && grid > 4 * cus)) continue;
    const int k = (grid + cus - 1) / cus;                               // workgroups on the fullest CU
    const double imbalance = (double)k * cus / grid;
    double cost = n * (k <= 4 ? 6.5 : 6.9) * imbalance;                 // ns per ply, store side
    if (m > 1 && cost < m * 6900.0) cost = m * 6900.0;                  // rules side
    if (out->m < 0 || cost < out->cost) { out->m = m; out->grid = grid; out->cost = cost; }
  }
  return 0;
}

// results of timed launches that have finished (never waits)
static void gen_tune_collect(GenTune& t) {
  bool not_ready = false;
  for (int i = 0; i < GenTune::NPEND; ++i) {
    GenTune::Pend& p = t.pend[i];
    if (!p.busy) continue;
    const hipError_t q = hipEventQuery(p.e1);
    if (q == hipErrorNotReady) { not_ready = true; continue; }
    if (q != hipSuccess) continue;                 // a real error: left for the caller's next checked call to report
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess && ms > 0.f) t.result(p.cand, ms * 1e3f / (float)p.plies);
    p.busy = 0;
  }
  if (not_ready) (void)hipGetLastError();   // hipEventQuery leaves "not ready" behind as the last error: that one is ours
}

int goenv_generate(goenv_ctx* c, int n_plies, const int32_t* aug_codes, float* feat_ring, uint8_t* legal_ring,
                   int32_t* actions_ring, void* stream) {
  if (!c) return fail("goenv_generate: ctx is NULL");
  ON_DEVICE_OF(c);
  if (n_plies < 0 || n_plies > (1 << 20)) return fail("goenv_generate: n_plies out of range");
  if (n_plies == 0) return 0;
  PlyParams P = make_params(c, GOENV_F_RANDOM_POLICY | GOENV_F_AUTO_RESTART, nullptr, aug_codes, feat_ring, legal_ring,
                            nullptr, nullptr);
  // A rules wave and a feature wave per workgroup, pipelined through LDS, the feature waves rate-limited: one board per
  // workgroup while that keeps the device's storing waves few (k_generate_pipe; 1,024 boards: 7.7 -> 6.9 us per ply, 256 ..
  // 1,536 boards: -18 %), several boards per workgroup beyond (k_generate_pipe_multi, up to about 14,000 boards).  Where
  // neither fits, workgroups would run in rounds and a clock-driven schedule has no meaning: two or three waves per board
  // that share rules and stores (k_generate; tools/time_generate.py).
  if (!c->gen_resident) {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) != hipSuccess || khz <= 0) khz = 100000;
    c->wall_khz = khz;
    int per_cu = 0, cus = 0;
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_generate_pipe, 128, 0));
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
    c->gen_resident = per_cu * cus > 0 ? per_cu * cus : -1;
  }
  // How many boards a workgroup holds (gen_multi_m: 1 = k_generate_pipe; 2 .. 16 = k_generate_pipe_multi, the others waiting
  // in LDS; -1 = neither fits).  Measured (profiles/r03_pipe_generator.md, section 11): the memory side takes the plane stream
  // best from about 1,024 storing waves -- four workgroups on every CU, one wave pair per SIMD -- at 6.5 ns per board-ply;
  // from five to eight workgroups per CU it takes 6.9; an uneven number per CU costs its imbalance; and a rules wave needs
  // 6.9 us for a board's turn.  The cheapest of the shapes that fit (gen_shape).  More boards than any shape holds go through
  // the same kernel in CHUNKS, one launch after the other on the stream, each over a contiguous part of the boards and its
  // rows of every ring slot (boards are independent: the same rings): 16,384 boards as 2 x 8,192, 65,536 as 8 x 8,192.
  if (!c->gen_multi_m) {
    int cus = 0;
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
    GenShape whole;
    if (gen_shape(c, c->n, cus, &whole)) return -1;
    c->gen_multi_m = whole.m; c->gen_multi_grid = whole.grid; c->gen_chunk_n = 0;
    // (a launch boundary costs a chunk what it costs any launch of the kernel: 16,384 boards x 8 plies per launch 124 -> 113 us
    // per ply, 32,768 boards 244 -> 223, but 65,536 boards x 4 plies 478 -> 490: shorter launches stay with k_generate)
    c->gen_chunk_min_plies = 8;
    if (const char* e = getenv("GOENV_GEN_CHUNK_MIN_PLIES")) c->gen_chunk_min_plies = atoi(e);
    if (whole.m < 0 && !getenv("GOENV_GEN_NO_CHUNKS")) {
      double best = 0;
      for (int k = 2; k <= 256 && c->n / k >= 1024; ++k) {
        const int chunk = (((c->n + k - 1) / k + 7) / 8) * 8, last = c->n - (k - 1) * chunk;
        if (last <= 0) continue;
        GenShape a, b;
        if (gen_shape(c, chunk, cus, &a) || gen_shape(c, last, cus, &b)) return -1;
        if (a.m < 2 || b.m < 1) continue;
        const double cost = (k - 1) * a.cost + b.cost;
        if (!c->gen_chunk_n || cost < best) {
          best = cost;
          c->gen_chunk_n = chunk; c->gen_multi_m = a.m; c->gen_multi_grid = a.grid; c->gen_last_m = b.m; c->gen_last_grid = b.grid;
        }
      }
    }
  }
  const bool resident = c->gen_multi_m == 1;
  const bool multi = c->gen_multi_m > 1 && !c->nw && c->gen_pipe != 0 && !(c->gen_chunk_n && n_plies < c->gen_chunk_min_plies);
  // (GOENV_GEN_PIPE=1 with no shape that fits, chunks turned off: k_generate_pipe over all boards, in rounds, unpaced -- a probe)
  const bool pipe = !c->nw && c->gen_pipe != 0 && (resident || multi || (c->gen_pipe > 0 && c->gen_multi_m < 0));
  c->gen_last_pipe = pipe ? (multi ? 2 : 1) : 0;
  if (!pipe) {
    const int nw = c->nw ? c->nw : (c->n <= 4096 ? 2 : 3);
    GO_LAUNCH(nw, k_generate, c->n, S(stream), P, n_plies, actions_ring);
    HIPCHK(hipGetLastError());
    return 0;
  }
  int pace_ticks = 0;   // ticks of the wall clock (10 ns)
  float pace_per_board = 0.f;   // tuned: ns per board-ply
  GenTune::Pend* timed = nullptr;
  if (c->gen_pace_ns >= 0) {
    pace_ticks = (resident || multi) ? (int)((double)c->gen_pace_ns * c->wall_khz * 1e-6) : 0;
  } else if ((resident || multi) && P.feat) {
    GenTune& t = c->tune;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stream) (void)hipStreamIsCapturing(S(stream), &cap);
    int cand = GenTune::SAFE;
    if (cap == hipStreamCaptureStatusNone && n_plies >= 8) {   // (no event inside a capture; short launches say little)
      if (!t.events_ready) {
        for (int i = 0; i < GenTune::NPEND; ++i) {
          HIPCHK(hipEventCreate(&t.pend[i].e0));
          HIPCHK(hipEventCreate(&t.pend[i].e1));
        }
        t.events_ready = 1;
      }
      gen_tune_collect(t);
      GenTune::Pend* slot = nullptr;
      for (int i = 0; i < GenTune::NPEND && !slot; ++i)
        if (!t.pend[i].busy) slot = &t.pend[i];
      bool want = false;
      cand = t.next(slot != nullptr, want);
      if (want) { timed = slot; timed->cand = cand; timed->plies = n_plies; }
      t.enqueued(cand, timed != nullptr);
    } else if (t.best() >= 0) {
      cand = t.best();
    }
    pace_per_board = GenTune::cand[cand];
  }
  if (timed) HIPCHK(hipEventRecord(timed->e0, S(stream)));
  const int chunks = c->gen_chunk_n ? (c->n + c->gen_chunk_n - 1) / c->gen_chunk_n : 1;
  for (int k = 0; k < chunks; ++k) {
    const int first = k * c->gen_chunk_n, count = chunks == 1 ? c->n : (c->n - first < c->gen_chunk_n ? c->n - first : c->gen_chunk_n);
    const bool tail = chunks > 1 && k == chunks - 1;
    const int m = tail ? c->gen_last_m : c->gen_multi_m, grid = tail ? c->gen_last_grid : c->gen_multi_grid;
    PlyParams Q = P;   // this launch's boards: rows first .. first + count of every ring slot
    Q.n = count; Q.first = first; Q.state = P.state + first; Q.id_base = P.id_base + first;
    if (Q.aug) Q.aug += first;
    // the tuned pace is per board-ply: the period of one ply of the boards of THIS launch
    const int ticks = pace_per_board > 0.f ? (int)((double)pace_per_board * count * c->wall_khz * 1e-6) : pace_ticks;
    if (m > 1)   // ticks: the time of one ply of all its boards; a workgroup divides it by the boards it holds
      hipLaunchKernelGGL(k_generate_pipe_multi, dim3(grid), dim3(128), (size_t)m * sizeof(SavedBoard), S(stream), Q, n_plies,
                         actions_ring, ticks);
    else
      hipLaunchKernelGGL(k_generate_pipe, dim3(count), dim3(128), 0, S(stream), Q, n_plies, actions_ring, ticks);
  }
  HIPCHK(hipGetLastError());
  if (timed) {
    HIPCHK(hipEventRecord(timed->e1, S(stream)));
    timed->busy = 1;
  }
  return 0;
}

// What the generator's self-tuning rate limiter has settled on: ns per board-ply offered (0: not in use or no result
// yet), its estimates per candidate in est_out[GOENV_GEN_PACE_CANDIDATES] (us per ply, 0 = none), and whether the last goenv_generate went through
// the pipelined kernel.  Diagnostic.
int goenv_generate_pace(goenv_ctx* c, float* pace_ns_per_board_out, float* est_out, int* pipelined_out) {
  if (!c) return fail("goenv_generate_pace: ctx is NULL");
  if (pipelined_out) *pipelined_out = c->gen_last_pipe;
  GenTune& t = c->tune;
  if (t.events_ready) gen_tune_collect(t);
  const int best = t.best();
  for (int i = 0; i < GenTune::NC; ++i)
    if (est_out) est_out[i] = t.samples[i] ? t.est[i] : 0.f;
  if (pace_ns_per_board_out) *pace_ns_per_board_out = best < 0 ? 0.f : GenTune::cand[best];
  return 0;
}

int goenv_score(goenv_ctx* c, float* score_out, float* fast_out, void* stream) {
  if (!c) return fail("goenv_score: ctx is NULL");
  ON_DEVICE_OF(c);
  GO_LAUNCH(pick_nw_rules(c, c->n), k_score, c->n, S(stream), c->state, c->n, score_out, fast_out, 0);
  HIPCHK(hipGetLastError());
  return 0;
}

int goenv_fast_score(goenv_ctx* c, int rule, float* out, void* stream) {
  if (!c || !out) return fail("goenv_fast_score: NULL argument");
  ON_DEVICE_OF(c);
  if (rule != 0 && rule != 1) return fail("goenv_fast_score: rule must be 0 (Chinese) or 1 (Japanese)");
  GO_LAUNCH(pick_nw_rules(c, c->n), k_score, c->n, S(stream), c->state, c->n, (float*)nullptr, out, rule);
  HIPCHK(hipGetLastError());
  return 0;
}

int goenv_finish(goenv_ctx* c, void* stream) {
  if (!c) return fail("goenv_finish: ctx is NULL");
  ON_DEVICE_OF(c);
  GO_LAUNCH(pick_nw_rules(c, c->n), k_finish, c->n, S(stream), c->state, c->n, c->stats, c->komi);
  HIPCHK(hipGetLastError());
  return 0;
}

int64_t* goenv_stats_ptr(goenv_ctx* c) { return c ? (int64_t*)c->stats : nullptr; }

int goenv_stats_host(goenv_ctx* c, int64_t* out16_host, void* stream) {
  if (!c || !out16_host) return fail("goenv_stats_host: NULL argument");
  ON_DEVICE_OF(c);
  int64_t all[STATS_ALLOC];
  HIPCHK(hipMemcpyAsync(all, c->stats, sizeof(all), hipMemcpyDeviceToHost, S(stream)));
  HIPCHK(hipStreamSynchronize(S(stream)));
  memcpy(out16_host, all, GOENV_NUM_STATS * sizeof(int64_t));
  // the word behind the scratch slot counts waits of the pipelined generator that ran into their bound (pipe_wait)
  if (all[GOENV_NUM_STATS + 1] != 0)
    return fail("goenv_stats_host: a wave of the pipelined generator gave up waiting for its partner (the rings of that launch "
                "are not to be trusted); this does not happen on a healthy device");
  return 0;
}

// RCCL, opened on first use: the library stays loadable (and every other entry point usable) on a
// host without RCCL, and a single-GPU process never maps it.  The build does not need RCCL's headers
// either: the handful of ABI names used are declared here with the values of rccl.h (NCCL's public,
// stable ABI: ncclDataType_t ncclInt64 = 4, ncclRedOp_t ncclSum = 0 / ncclMax = 2, ncclSuccess = 0).
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclDataType_t ncclInt64 = 4;
constexpr ncclRedOp_t ncclSum = 0, ncclMax = 2;
struct RcclApi {
  ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*error_string)(ncclResult_t) = nullptr;
  bool tried = false;
};
static RcclApi g_rccl;
static std::mutex g_rccl_mu;

static int rccl_load() {
  std::lock_guard<std::mutex> lock(g_rccl_mu);
  if (!g_rccl.tried) {
    g_rccl.tried = true;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (h) {
      g_rccl.all_reduce = reinterpret_cast<decltype(g_rccl.all_reduce)>(dlsym(h, "ncclAllReduce"));
      g_rccl.error_string = reinterpret_cast<decltype(g_rccl.error_string)>(dlsym(h, "ncclGetErrorString"));
    }
  }
  if (!g_rccl.all_reduce) return fail("goenv_stats_allreduce: librccl.so (ncclAllReduce) cannot be loaded");
  return 0;
}

// End-of-episode collective: the stats vector summed over the ranks of `rccl_comm`, slot
// GOENV_ST_MAX_LEN reduced with MAX, in place, ordered on `stream`.  Two tiny all-reduces: latency only.
int goenv_stats_allreduce(goenv_ctx* c, void* rccl_comm, void* stream) {
  if (!c) return fail("goenv_stats_allreduce: ctx is NULL");
  if (!rccl_comm) return fail("goenv_stats_allreduce: communicator is NULL");
  if (rccl_load()) return -1;
  ON_DEVICE_OF(c);
  unsigned long long* st = c->stats;
  unsigned long long* mx = c->stats + GOENV_NUM_STATS;
  HIPCHK(hipMemcpyAsync(mx, st + GOENV_ST_MAX_LEN, sizeof(*mx), hipMemcpyDeviceToDevice, S(stream)));
  ncclResult_t r = g_rccl.all_reduce(st, st, GOENV_NUM_STATS, ncclInt64, ncclSum, (ncclComm_t)rccl_comm, S(stream));
  if (r == ncclSuccess) r = g_rccl.all_reduce(mx, mx, 1, ncclInt64, ncclMax, (ncclComm_t)rccl_comm, S(stream));
  if (r != ncclSuccess) {
    fail("goenv_stats_allreduce: ncclAllReduce");
    g_err += ": ";
    g_err += g_rccl.error_string ? g_rccl.error_string(r) : "RCCL error";
    return -1;
  }
  HIPCHK(hipMemcpyAsync(st + GOENV_ST_MAX_LEN, mx, sizeof(*mx), hipMemcpyDeviceToDevice, S(stream)));
  return 0;
}

int goenv_stats_clear(goenv_ctx* c, void* stream) {
  if (!c) return fail("goenv_stats_clear: ctx is NULL");
  ON_DEVICE_OF(c);
  HIPCHK(hipMemsetAsync(c->stats, 0, GOENV_NUM_STATS * sizeof(int64_t), S(stream)));
  return 0;
}

int goenv_timer_begin(goenv_ctx* c, void* stream) {
  if (!c) return fail("goenv_timer_begin: ctx is NULL");
  ON_DEVICE_OF(c);
  HIPCHK(hipEventRecord(c->ev0, S(stream)));
  return 0;
}
int goenv_timer_end(goenv_ctx* c, void* stream) {
  if (!c) return fail("goenv_timer_end: ctx is NULL");
  ON_DEVICE_OF(c);
  HIPCHK(hipEventRecord(c->ev1, S(stream)));
  return 0;
}
// Busy-polls the end event instead of sleeping on it: a blocking synchronise wakes the host tens of
// microseconds after the GPU is done, which a caller timing a sub-millisecond region would count.
int goenv_timer_spin(goenv_ctx* c) {
  if (!c) return fail("goenv_timer_spin: ctx is NULL");
  ON_DEVICE_OF(c);
  for (;;) {
    const hipError_t e = hipEventQuery(c->ev1);
    if (e == hipSuccess) return 0;
    if (e != hipErrorNotReady) return fail("goenv_timer_spin: hipEventQuery", e);
  }
}
float goenv_timer_ms(goenv_ctx* c) {
  if (!c) return -1.f;
  DeviceScope scope(c->device);
  float ms = -1.f;
  if (hipEventSynchronize(c->ev1) != hipSuccess) return -1.f;
  if (hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) return -1.f;
  return ms;
}

static int d4_fwd(int code, int x, int y) {
  const int rot = code % 4;
  const bool flip = (code >> 2) == 1;
  int ox, oy;
  if (rot == 1) { ox = y; oy = BS - x - 1; }
  else if (rot == 2) { ox = BS - x - 1; oy = BS - y - 1; }
  else if (rot == 3) { ox = BS - y - 1; oy = x; }
  else { ox = x; oy = y; }
  if (flip) { int t = ox; ox = oy; oy = t; }
  return ox * BS + oy;
}

int goenv_aug_action(int aug_code, int action) {
  if (action < 0 || action >= NPTS) return action;
  return d4_fwd(aug_code, action / BS, action % BS);
}

int goenv_aug_action_inv(int aug_code, int action) {
  if (action < 0 || action >= NPTS) return action;
  int a = action / BS, b = action % BS;
  if ((aug_code >> 2) == 1) { int t = a; a = b; b = t; }
  const int rot = aug_code % 4;
  int x, y;
  if (rot == 1) { x = BS - 1 - b; y = a; }
  else if (rot == 2) { x = BS - 1 - a; y = BS - 1 - b; }
  else if (rot == 3) { x = b; y = BS - 1 - a; }
  else { x = a; y = b; }
  return x * BS + y;
}

}  // extern "C"
