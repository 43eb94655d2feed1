// goenv_device.h -- device code of the gfx950 Go environment: one workgroup of NW wavefronts
// (NW in {1,2,3,6}) owns one 19x19 board.
//
// Thread t of the workgroup handles the points p = t + 64*NW*k, k = 0..K-1 with K = 6/NW
// (p = x*19 + y, the reference's export order, go/board.h:155), so every batch-buffer store is
// lane-contiguous and the ballot of chunk k taken by wave w is word NW*k + w of the board's
// 361-bit mask (bit p%64 of word p/64).  NW = 1 is "one wavefront per board": everything is
// wave-local.  NW > 1 splits the per-point work of one board over several waves -- at small
// batches (1,024 boards on 256 CUs) the chip is latency-bound with one wave per SIMD, and several
// waves per board cut the dependent LDS/VALU chain per wave by that factor; the few values that
// are properties of the whole board (capture count, colour masks) cross waves through LDS.
//
// Neighbour lookups go through a 21x21 padded label image in LDS (index c = (x+1)*21 + (y+1))
// whose rim is a sentinel colour -- the same trick as the reference's 21x21 `_infos`
// (go/board.h:80) -- so no edge tests are needed.
//
// State model (differs from the reference on purpose): every stone carries the 13-bit label
// root | colour<<9 | cls<<11  where root is the point index of the most recently added stone of
// its group and cls = min(liberties of the group, 3).
// * Merging groups and removing captured groups are single thread-parallel relabel passes
//   (replacing the linked-list walks of MergeGroups/EmptyGroup, go/board.cc:559-593,412-434).
// * The liberty table lib[root] is rebuilt from scratch each ply: every empty point adds 1 to
//   each distinct neighbouring root with LDS atomics (replacing the incremental bookkeeping of
//   go/board.cc:393-409,529-555,595-620).  The reference's stored liberties always equal this
//   from-scratch count (VerifyBoard, go/board.cc:1455-1462), so the two models are observably
//   identical.
// * Because every neighbour read returns colour AND liberty class, legality, eye and feature
//   tests are branch-free table lookups on one LDS read per neighbour (no dependent group-table
//   gather as in StoneLibertyAnalysis, go/board.cc:95-129).
// * The legal mask and the playout-candidate mask of the side to move are computed once per ply
//   and kept in the record as 2 x 361-bit ballots.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace goenv {

constexpr int BS = 19;
constexpr int NPTS = 361;
constexpr int PW = 21;       // padded width
constexpr int NWORDS = 6;    // 64-bit words covering 361 points
constexpr int LAB_N = 448;   // padded label image entries (441 used)
constexpr int TAB_N = 368;   // per-point tables, 361 rounded up
constexpr int HIST_LUT_N = 2048;

constexpr int S_EMPTY = 0, S_BLACK = 1, S_WHITE = 2;
constexpr uint32_t BORDER = 3u << 9;
constexpr uint32_t ROOT_MASK = 511u;
constexpr uint32_t LABEL_MASK = 0x7ffu;  // root | colour
constexpr int DIST_INF = 64;

// scalar slots inside the record (same order as the canonical view, include/goenv.h)
enum { SC_PLY = 0, SC_NEXT, SC_BCAP, SC_WCAP, SC_L1, SC_L2, SC_L3, SC_L4, SC_KOAGE, SC_KO,
       SC_KOCOLOR, SC_DONE, SC_GAME, SC_COUNTED, SC_SEQ, SC_SEQTOT };

// Packed per-board record in HBM: 1632 B, 16-B aligned, one contiguous read per workgroup.
struct __attribute__((aligned(16))) BoardRec {
  uint16_t lab[TAB_N];    // root | colour<<9 | cls<<11, 0 = empty       (export order)
  uint16_t lastp[TAB_N];  // ply of placement (go/board.h:46); stale under empties
  uint64_t legal[NWORDS]; // FindAllValidMoves of the side to move, bit p%64 of word p/64
  uint64_t cand[NWORDS];  // legal AND NOT own true eye (cfg-2 playout policy)
  int32_t sc[16];
};
static_assert(sizeof(BoardRec) == 1632, "record layout");

// Per-board LDS working set (~5 KB -> 32 single-wave boards per CU fit in 160 KB).
template <int NW>
struct __attribute__((aligned(16))) BoardLds {
  // liberties by root (+ one sink slot per thread for masked atomics); while no liberty count is
  // live it doubles as the staging image of the record's per-point tables (load_record) and as
  // the history-value table of write_features
  uint32_t lib[TAB_N + 64 * NW];
  uint32_t desc[TAB_N];           // per-point feature descriptor; also flood-fill scratch
  uint16_t lab[LAB_N];            // padded label image
  uint8_t dist[2][TAB_N];         // L1 distance to nearest black / white stone
  // cross-wave exchange (unused when NW == 1)
  unsigned long long xcol[2 * NWORDS];   // colour ballots
  unsigned long long xmask[2 * NWORDS];  // legal / candidate ballots
  uint32_t xcap[8];                      // per-wave capture count | (point+1) << 16
  int32_t xsum[2][8];                    // per-wave partial sums
  uint32_t xflag[4];                     // "something changed" flags of fixed-point loops
};

struct Scalars {
  int ply, next, bcap, wcap, l1, l2, l3, l4, koage, ko, kocolor, done, game, counted, seq, seqtot;
};

struct Masks {  // all six words, uniform
  unsigned long long legal[NWORDS];
  unsigned long long cand[NWORDS];
};

template <int NW>
struct Own {  // the K words of each mask that this wave's ballots produce (word NW*k + wave)
  unsigned long long legal[NWORDS / NW];
  unsigned long long cand[NWORDS / NW];
  unsigned long long eye[NWORDS / NW];
};

template <int NW>
struct Geo {
  static constexpr int K = NWORDS / NW;
  static constexpr int T = 64 * NW;
  int tid, lane, wave;
  int p[K];  // point index (may be >= 361 in the last chunk)
  int c[K];  // padded index; an interior cell (never written) for invalid threads
};

#define GO_UNROLL _Pragma("unroll")
// Board-wide synchronisation point.  NW == 1: the board is one wave, whose LDS operations are executed in order;
// only the compiler and the wave's own outstanding LDS operations must be fenced -- no `s_barrier`, so that a
// one-wave board can live inside a LARGER workgroup (k_generate_pipe: a rules wave beside feature waves).
template <int NW>
__device__ __forceinline__ void wsync() {
  if (NW == 1) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  } else {
    __syncthreads();
  }
}
#define WSYNC() wsync<NW>()

__device__ __forceinline__ int popc64(unsigned long long v) { return __popcll(v); }
__device__ __forceinline__ uint32_t col_of(uint32_t v) { return (v >> 9) & 3u; }
__device__ __forceinline__ uint32_t cls_of(uint32_t v) { return (v >> 11) & 3u; }
__device__ __forceinline__ uint32_t typ_of(uint32_t v) { return (v >> 9) & 15u; }  // colour | cls<<2

__device__ __forceinline__ unsigned long long uni64(unsigned long long v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

template <int NW>
__device__ __forceinline__ void geo_init(Geo<NW>& g) {
  g.tid = threadIdx.x;
  g.lane = g.tid & 63;
  g.wave = __builtin_amdgcn_readfirstlane(g.tid >> 6);
  GO_UNROLL
  for (int k = 0; k < Geo<NW>::K; ++k) {
    const int p = g.tid + Geo<NW>::T * k;
    const int x = p / BS, y = p - x * BS;
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

__device__ __forceinline__ void store_scalars(BoardRec* rec, const Scalars& s, int tid) {
  // one 64-B line, written by the first 16 threads
  int v = 0;
  switch (tid) {
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
  if (tid < 16) rec->sc[tid] = v;
}

// ClearBoard, go/board.cc:30-46 (scalars only; the caller zeroes the tables)
__device__ __forceinline__ void clear_scalars(Scalars& s) {
  s.ply = 1; s.next = S_BLACK; s.bcap = 0; s.wcap = 0;
  s.l1 = s.l2 = s.l3 = s.l4 = -1;
  s.koage = 0; s.ko = -1; s.kocolor = 0; s.done = 0; s.counted = 0; s.seq = 0;
}

template <int NW>
__device__ __forceinline__ void fill_rim(BoardLds<NW>& L, int tid) {
  for (int i = tid; i < LAB_N; i += 64 * NW) L.lab[i] = (uint16_t)BORDER;
}

// HBM record -> LDS image + registers.
template <int NW>
__device__ __forceinline__ void load_board(const BoardRec* rec, BoardLds<NW>& L, const Geo<NW>& g,
                                           uint32_t lab[], uint32_t lastp[]) {
  GO_UNROLL
  for (int k = 0; k < Geo<NW>::K; ++k) {
    bool v = g.p[k] < NPTS;
    lab[k] = v ? rec->lab[g.p[k]] : 0u;
    lastp[k] = v ? rec->lastp[g.p[k]] : 0u;
  }
  fill_rim(L, g.tid);
  WSYNC();
  GO_UNROLL
  for (int k = 0; k < Geo<NW>::K; ++k)
    if (g.p[k] < NPTS) L.lab[g.c[k]] = (uint16_t)lab[k];
  WSYNC();
}

__device__ __forceinline__ void load_masks(const BoardRec* rec, Masks& m) {
  GO_UNROLL
  for (int j = 0; j < NWORDS; ++j) { m.legal[j] = rec->legal[j]; m.cand[j] = rec->cand[j]; }
}

// HBM record -> LDS -> registers: the two per-point tables with ONE 16-B load per thread (92 of
// them) instead of 2-byte loads per point: the vector-memory pipe takes as long for a
// 128-B wave-instruction as for a 1-KB one, and a record arrives in a single round trip.
constexpr int LABS_VEC = (2 * TAB_N * 2) / 16;         // 92: the two per-point tables
template <int NW>
__device__ __forceinline__ void load_record(const BoardRec* rec, BoardLds<NW>& L, const Geo<NW>& g,
                                            uint32_t lab[], uint32_t lastp[], Scalars& s, Masks& m) {
  load_scalars(rec, s);   // scalar loads: back early, the move choice starts from them
  load_masks(rec, m);
  uint4* stage = reinterpret_cast<uint4*>(L.lib);
  const uint4* src = reinterpret_cast<const uint4*>(rec);
#ifdef GOENV_REC_LD_SC1
  // probe (tools/ablate.py): the record read past the XCD's L2 (agent-scope loads, `sc1`)
  for (int i = g.tid; i < LABS_VEC; i += 64 * NW) {
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(src + i);
    const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    stage[i] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
  }
#else
  for (int i = g.tid; i < LABS_VEC; i += 64 * NW) stage[i] = src[i];
#endif
  fill_rim(L, g.tid);
  WSYNC();
  const uint16_t* raw = reinterpret_cast<const uint16_t*>(L.lib);
  GO_UNROLL
  for (int k = 0; k < Geo<NW>::K; ++k) {
    const bool v = g.p[k] < NPTS;
    lab[k] = v ? raw[g.p[k]] : 0u;
    lastp[k] = v ? raw[TAB_N + g.p[k]] : 0u;
    if (v) L.lab[g.c[k]] = (uint16_t)lab[k];
  }
  WSYNC();  // image complete; the staging area (the liberty table) may be rewritten from here on
}

// this wave's words of the uniform six-word masks
template <int NW>
__device__ __forceinline__ void own_from_masks(const Geo<NW>& g, const Masks& m, Own<NW>& o) {
  GO_UNROLL
  for (int k = 0; k < Geo<NW>::K; ++k) {
    unsigned long long wl = 0, wc = 0;
    GO_UNROLL
    for (int j = 0; j < NWORDS; ++j)
      if (j == NW * k + g.wave) { wl = m.legal[j]; wc = m.cand[j]; }
    o.legal[k] = wl; o.cand[k] = wc; o.eye[k] = wl & ~wc;
  }
}

template <int NW>
__device__ __forceinline__ void clear_board_lds(BoardLds<NW>& L, const Geo<NW>& g, uint32_t lab[], uint32_t lastp[]) {
  fill_rim(L, g.tid);
  WSYNC();
  GO_UNROLL
  for (int k = 0; k < Geo<NW>::K; ++k) {
    lab[k] = 0; lastp[k] = 0;
    if (g.p[k] < NPTS) L.lab[g.c[k]] = 0;
  }
  WSYNC();
}

__device__ __forceinline__ unsigned long long full_word(int j) {
  return (j < NWORDS - 1) ? ~0ull : ((1ull << (NPTS - 64 * (NWORDS - 1))) - 1ull);
}

// masks of an empty board with BLACK to move: every point legal and a candidate
template <int NW>
__device__ __forceinline__ void full_masks(const Geo<NW>& g, Masks& m, Own<NW>& o) {
  GO_UNROLL
  for (int j = 0; j < NWORDS; ++j) { m.legal[j] = full_word(j); m.cand[j] = full_word(j); }
  own_from_masks(g, m, o);
}

// Each wave's own words -> all six words in every wave (through LDS when NW > 1).
template <int NW>
__device__ __forceinline__ void gather_masks(BoardLds<NW>& L, const Geo<NW>& g, const Own<NW>& o, Masks& m) {
  if (NW == 1) {
    GO_UNROLL
    for (int j = 0; j < NWORDS / NW; ++j) { m.legal[j] = o.legal[j]; m.cand[j] = o.cand[j]; }
    return;
  }
  if (g.lane == 0) {
    GO_UNROLL
    for (int k = 0; k < Geo<NW>::K; ++k) {
      L.xmask[NW * k + g.wave] = o.legal[k];
      L.xmask[NWORDS + NW * k + g.wave] = o.cand[k];
    }
  }
  WSYNC();
  GO_UNROLL
  for (int j = 0; j < NWORDS; ++j) { m.legal[j] = uni64(L.xmask[j]); m.cand[j] = uni64(L.xmask[NWORDS + j]); }
  WSYNC();  // the array may be rewritten right after
}

// Sum of a wave-uniform value over the waves of the board.
template <int NW>
__device__ __forceinline__ int board_sum(BoardLds<NW>& L, const Geo<NW>& g, int v, int slot) {
  if (NW == 1) return v;
  if (g.lane == 0) L.xsum[slot][g.wave] = v;
  WSYNC();
  int s = 0;
  GO_UNROLL
  for (int w = 0; w < NW; ++w) s += L.xsum[slot][w];
  s = __builtin_amdgcn_readfirstlane(s);
  WSYNC();
  return s;
}

// "did any thread of the board change something" inside a fixed-point loop; three rotating flag
// words so that clearing never races with a late reader.  Contains the loop's barrier.
template <int NW>
__device__ __forceinline__ bool board_any(BoardLds<NW>& L, const Geo<NW>& g, bool changed, int iter) {
  if (NW == 1) {
    WSYNC();
    return __any(changed);
  }
  const int cur = iter % 3, nxt = (iter + 1) % 3;
  if (g.tid == 0) L.xflag[nxt] = 0;
  if (__any(changed) && g.lane == 0) L.xflag[cur] = 1;
  WSYNC();
  return __builtin_amdgcn_readfirstlane(L.xflag[cur]) != 0;
}

template <int NW>
__device__ __forceinline__ void board_any_init(BoardLds<NW>& L, const Geo<NW>& g) {
  if (NW > 1) {
    if (g.tid < 3) L.xflag[g.tid] = 0;
    WSYNC();
  }
}

// registers -> HBM record.  (Staging the record through LDS for 16-B stores, as load_record does for
// the load, was measured slower in the fused ply: its three extra barriers cost more than the
// narrow stores, which are fire-and-forget.)
template <int NW>
__device__ __forceinline__ void store_board(BoardRec* rec, const Geo<NW>& g, const uint32_t lab[],
                                            const uint32_t lastp[], const Own<NW>& o, const Scalars& s) {
  constexpr int K = Geo<NW>::K;
  GO_UNROLL
  for (int k = 0; k < K; ++k) {
    if (g.p[k] < NPTS) {
#ifdef GOENV_REC_ST_SC1
      // probe (tools/ablate.py): the record written through the XCD's L2 (agent-scope stores, `sc1`)
      __hip_atomic_store(&rec->lab[g.p[k]], (uint16_t)lab[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&rec->lastp[g.p[k]], (uint16_t)lastp[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
      rec->lab[g.p[k]] = (uint16_t)lab[k];
      rec->lastp[g.p[k]] = (uint16_t)lastp[k];
#endif
    }
  }
  // this wave's mask words: lane k writes the legal word, lane K + k the candidate word
  unsigned long long w = 0;
  int j = 0;
  GO_UNROLL
  for (int k = 0; k < K; ++k) {
    if (g.lane == k) { w = o.legal[k]; j = NW * k + g.wave; }
    if (g.lane == K + k) { w = o.cand[k]; j = NWORDS + NW * k + g.wave; }
  }
  uint64_t* words = rec->legal;  // legal[6] and cand[6] are contiguous
  if (g.lane < 2 * K) words[j] = w;
  store_scalars(rec, s, g.tid);
}

// 4 neighbours of padded index c in the reference's L,T,R,B order (go/board.h:183):
// x-1, y-1, x+1, y+1.
template <int NW>
__device__ __forceinline__ void nbr4(const BoardLds<NW>& L, int c, uint32_t n[4]) {
  n[0] = L.lab[c - PW];
  n[1] = L.lab[c - 1];
  n[2] = L.lab[c + PW];
  n[3] = L.lab[c + 1];
}

__device__ __forceinline__ bool is_stone(uint32_t v) { uint32_t c = col_of(v); return c == 1 || c == 2; }

// Liberty table from scratch: every empty point adds 1 to each DISTINCT neighbouring root
// (== the distinct-liberty count RecomputeGroupLiberties produces, go/board.cc:595-620), then
// every stone's cls bits are refreshed to min(liberties, 3).  Branch-free: masked-off adds go to
// a per-thread sink slot.
// PREZEROED: the caller cleared lib[0..360] before its own last barrier.  Returns the liberty
// count of root `probe` (-1: none), read before the closing barrier so that the table is dead
// once this function returns (write_features reuses it without another barrier).
template <int NW, bool PREZEROED = false>
__device__ __forceinline__ int rebuild_lib(BoardLds<NW>& L, const Geo<NW>& g, uint32_t lab[], int probe = -1) {
  if (!PREZEROED) {
    GO_UNROLL
    for (int k = 0; k < Geo<NW>::K; ++k)
      if (g.p[k] < NPTS) L.lib[g.p[k]] = 0;
    WSYNC();
  }
  const int sink = TAB_N + g.tid;
  GO_UNROLL
  for (int k = 0; k < Geo<NW>::K; ++k) {
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
  for (int k = 0; k < Geo<NW>::K; ++k) {
    if (g.p[k] < NPTS && lab[k] != 0) {
      uint32_t lib = L.lib[lab[k] & ROOT_MASK];
      uint32_t v = (lab[k] & LABEL_MASK) | ((lib > 3u ? 3u : lib) << 11);
      lab[k] = v;
      L.lab[g.c[k]] = (uint16_t)v;
    }
  }
  const int probed = probe >= 0 ? (int)L.lib[probe] : 0;
  WSYNC();
  return __builtin_amdgcn_readfirstlane(probed);
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
// not suicide :131-154) and own-true-eye mask (IsTrueEye :1501-1552) for `player`, as this wave's
// ballot words.
template <int NW>
__device__ __forceinline__ void move_masks(const BoardLds<NW>& L, const Geo<NW>& g, const uint32_t lab[],
                                           const Scalars& s, int player, Own<NW>& o) {
  const uint32_t OK = tab_ok(player);
  const uint32_t OWNR = tab_col((uint32_t)player) | tab_col(3u);
  const uint32_t OPP = tab_col(3u - (uint32_t)player);
  const uint32_t RIM = tab_col(3u);
  const int ko_p = (s.koage == 0 && s.kocolor == player) ? s.ko : -1;
  GO_UNROLL
  for (int k = 0; k < Geo<NW>::K; ++k) {
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
    o.legal[k] = __ballot(legal);
    o.eye[k] = __ballot(eye);
    o.cand[k] = o.legal[k] & ~o.eye[k];
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
// Uniform.  Used where the stored masks do not apply (handicap stones are BLACK's whoever is
// to move).
template <int NW>
__device__ __forceinline__ bool move_is_legal(const BoardLds<NW>& L, const Scalars& s, int p, int player) {
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
template <int NW>
__device__ __forceinline__ void apply_move(BoardLds<NW>& L, const Geo<NW>& g, uint32_t lab[], uint32_t lastp[],
                                           Scalars& s, int p, int player) {
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
  if (NW > 1) WSYNC();  // every wave has read the neighbourhood before any wave rewrites it
  const uint32_t nv = ((uint32_t)player << 9) | (uint32_t)p;
  int total_capture = 0, capture_p = -1;
  GO_UNROLL
  for (int k = 0; k < Geo<NW>::K; ++k) {
    uint32_t v = lab[k];
    const int r = (int)(v & ROOT_MASK);
    const uint32_t col = col_of(v);
    const bool mine = col == (uint32_t)player && (r == own[0] || r == own[1] || r == own[2] || r == own[3]);
    const bool cap = col == opp && (r == dead[0] || r == dead[1] || r == dead[2] || r == dead[3]);
    if (mine) v = nv;
    if (cap) v = 0;
    if (g.p[k] == p) { v = nv; lastp[k] = (uint32_t)s.ply & 0xffffu; }  // board.cc:532,1157
    const unsigned long long m = __ballot(cap);
    if (m) { total_capture += popc64(m); capture_p = 64 * (NW * k + g.wave) + (int)__builtin_ctzll(m); }
    if (v != lab[k]) { L.lab[g.c[k]] = (uint16_t)v; lab[k] = v; }
    if (g.p[k] < NPTS) L.lib[g.p[k]] = 0;  // liberty table cleared under the same barrier
  }
  if (NW > 1 && g.lane == 0) L.xcap[g.wave] = (uint32_t)total_capture | ((uint32_t)(capture_p + 1) << 16);
  WSYNC();
  if (NW > 1) {
    total_capture = 0; capture_p = -1;
    GO_UNROLL
    for (int w = 0; w < NW; ++w) {
      const uint32_t v = __builtin_amdgcn_readfirstlane(L.xcap[w]);
      total_capture += (int)(v & 0xffffu);
      if (v & 0xffffu) capture_p = (int)(v >> 16) - 1;
    }
  }
  if (player == S_BLACK) s.bcap += total_capture; else s.wcap += total_capture;  // board.cc:1132-1133
  const int new_libs = rebuild_lib<NW, true>(L, g, lab, p);
  // simple ko, board.cc:1163-1171
  if (!merged && total_capture == 1 && new_libs == 1) {
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
// Works on the uniform six-word mask, so every wave computes the same result by itself.
__device__ __forceinline__ int pick_random_move(const Masks& m, int lane, const Scalars& s, uint64_t seed,
                                                uint64_t board_id) {
  int n = 0;
  GO_UNROLL
  for (int j = 0; j < NWORDS; ++j) n += popc64(m.cand[j]);
  if (n == 0) return -1;
  const uint64_t u = policy_rand(seed, board_id, (uint64_t)s.game, (uint64_t)s.ply);
  int idx = (int)(((u >> 32) * (uint64_t)n) >> 32);
  int chosen = -1;
  GO_UNROLL
  for (int j = 0; j < NWORDS; ++j) {
    const unsigned long long w = m.cand[j];
    const int cnt = popc64(w);
    if (chosen < 0) {
      if (idx < cnt) {
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(w >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)w, 0u));
        const unsigned long long hit = __ballot(((w >> lane) & 1ull) && rank == idx);
        chosen = 64 * j + (int)__builtin_ctzll(hit);
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
template <int NW>
__device__ __forceinline__ int tromp_taylor(BoardLds<NW>& L, const Geo<NW>& g, const uint32_t lab[]) {
  constexpr int K = Geo<NW>::K;
  uint16_t* flag = reinterpret_cast<uint16_t*>(L.desc);  // padded image, LAB_N u16 <= TAB_N u32
  for (int i = g.tid; i < LAB_N; i += 64 * NW) flag[i] = 0;
  board_any_init(L, g);
  WSYNC();
  uint32_t f[K];
  GO_UNROLL
  for (int k = 0; k < K; ++k) {
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
    for (int k = 0; k < K; ++k) {
      if (g.p[k] < NPTS && lab[k] == 0) {
        const int c = g.c[k];
        const uint32_t nf = f[k] | flag[c - PW] | flag[c - 1] | flag[c + PW] | flag[c + 1];
        if (nf != f[k]) { f[k] = nf; flag[c] = (uint16_t)nf; changed = true; }
      }
    }
    if (!board_any(L, g, changed, iter)) break;
  }
  int diff = 0;
  GO_UNROLL
  for (int k = 0; k < K; ++k) {
    const uint32_t col = col_of(lab[k]);
    const bool e = g.p[k] < NPTS && lab[k] == 0;
    diff += popc64(__ballot(col == 1 || (e && f[k] == 1)));
    diff -= popc64(__ballot(col == 2 || (e && f[k] == 2)));
  }
  WSYNC();
  return board_sum(L, g, diff, 0);
}

// GetFastScore, go/board.cc:1560-1582 (GetEyeColor tests white first, :1554-1558).
// rule 0 = RULE_CHINESE: stones + true eyes per colour; rule 1 = RULE_JAPANESE: true eyes +
// captures (`_rollout_passes` is 0 on every path that reaches this build: only the reference's
// own CPU rollouts change it).  Both are narrowed to `short` like the reference.
template <int NW>
__device__ __forceinline__ int fast_score(BoardLds<NW>& L, const Geo<NW>& g, const uint32_t lab[],
                                          const Scalars& s, int rule) {
  Own<NW> ob, ow;
  move_masks(L, g, lab, s, S_BLACK, ob);
  move_masks(L, g, lab, s, S_WHITE, ow);
  int sc = 0;
  GO_UNROLL
  for (int k = 0; k < Geo<NW>::K; ++k) {
    const uint32_t col = col_of(lab[k]);
    sc += popc64(ob.eye[k] & ~ow.eye[k]) - popc64(ow.eye[k]);
    if (rule == 0) sc += popc64(__ballot(col == 1)) - popc64(__ballot(col == 2));
  }
  sc = board_sum(L, g, sc, 1);
  if (rule != 0) sc += s.bcap - s.wcap;
  return (int)(short)sc;
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

// Exact L1 distance to the nearest black / white stone (GetDistanceMap + DistanceTransform,
// go/board_feature.cc:17-38,148-160) into L.dist, by ONE wave (lanes = (colour, y) lines).
// Pass 1 (along y) is a bit trick on the 19-bit column occupancy, pass 2 (along x) the
// reference's forward/backward min-plus sweep, fully unrolled in registers.
// Results are stored at the OUTPUT index of each point (D4 code `aug` applied here).
template <class LDS>
__device__ __forceinline__ void distance_lines(LDS& L, int lane, const unsigned long long wb[NWORDS + 1],
                                               const unsigned long long ww[NWORDS + 1], int aug) {
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
    if (aug == 0) {
      GO_UNROLL
      for (int x = 0; x < BS; ++x) d[x * BS + y] = (uint8_t)a[x];
    } else {
      GO_UNROLL
      for (int x = 0; x < BS; ++x) d[d4_dest(aug, x * BS + y)] = (uint8_t)a[x];
    }
  }
}

// 4 floats with only 4-byte alignment: planes start every 1444 B and boards every 36100 B, so
// 16-B alignment is not available; gfx950 takes dwordx4 global stores at dword alignment at full
// rate (tools/store_bench.hip).
struct __attribute__((packed, aligned(4))) float4u { float x, y, z, w; };
typedef float f32x4 __attribute__((ext_vector_type(4)));

// One dwordx4 plane store.  WT = write-through (`sc1`): the bytes leave the XCD's L2 at once
// instead of staying dirty there until the end-of-kernel write-back, so a launch that writes
// tens of MB does not pay that write-back after its last wave has finished (the "publish-large"
// row of MI355X_MICROARCH.md), and stores issued early really overlap the rules phase.
template <bool WT>
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
#ifdef GOENV_ABLATE_STORES
  if (a == 123.456f) *p = b + c + d;  // keeps the operands live, never stores
  return;
#endif
  if (WT) {
    // The two wait states after the store are part of the instruction as far as the compiler is
    // concerned: a VMEM store of more than 64 bits must not be followed at once by a VALU write to
    // its data registers (the store would pick up the new value), and the hazard recogniser does
    // not look inside inline asm.  Without them about one board-ply in a thousand had one wrong float.
    const f32x4 v = {a, b, c, d};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v));
  } else {
    const float4u v = {a, b, c, d};
    *reinterpret_cast<float4u*>(p) = v;
  }
}

template <bool WT>
__device__ __forceinline__ void zero_range(float* __restrict__ o, int count, int lane) {
  const int body = count & ~3;
  for (int i = 4 * lane; i < body; i += 256) store4<WT>(o + i, 0.f, 0.f, 0.f, 0.f);
  if (lane < count - body) o[body + lane] = 0.f;
}

// The 11 planes that Extract never fills (12, 13: BORDER / POSITION_MARK; 16..24,
// go/board_feature.h:13-28, board_feature.cc:178-179) are constant zeros.  They are 44 % of a
// sample's bytes and depend on nothing, so a kernel issues them FIRST: the stores drain to HBM
// while the workgroup is still in its latency-bound rules phase.
template <int NW, bool WT>
__device__ __forceinline__ void zero_planes(const Geo<NW>& g, float* __restrict__ out) {
  constexpr int BIG = 9 * NPTS;                       // planes 16..24
  constexpr int PART = ((BIG / NW) + 3) & ~3;         // per-wave slice, multiple of 4 floats
  const int lo = g.wave * PART;
  const int hi = (g.wave == NW - 1) ? BIG : lo + PART;
  zero_range<WT>(out + 16 * NPTS + lo, hi - lo, g.lane);
  if (g.wave == 0) zero_range<WT>(out + 12 * NPTS, 2 * NPTS, g.lane);
}

// BoardFeature::Extract (go/board_feature.cc:175-195): the 14 planes that carry data and, when
// ZEROS, the 11 constant ones too (otherwise the caller has issued zero_planes for them).
// Store shape: plane-major, each plane as two dwordx4 wave-stores (lane l: q = 4l..4l+3, then
// q = 256+4l..; lane 26 covers 357..360, re-writing 3 floats with the same values), so a board's
// 36,100 B leave as ascending contiguous runs -- 45 store instructions instead of 150 dword
// stores, and 1.6x the HBM rate of point-major dword stores at 65,536 boards.
// Work split when NW > 1: wave 0 runs the distance transform while the other waves already
// stream out the 12 planes that do not depend on it; planes 14/15 follow.
// `lut` = host-built (float)exp(-d/10.0) table (board_feature.cc:141).
// History values are table loads; they are issued right after the record load (history_preload),
// for the ply the position will have when the planes are written, so that they neither wait
// behind the kernel's own stores nor sit on the critical path.  `pre_ply` = that assumed ply;
// if the position ended up elsewhere (restart, rejected move) the values are loaded again.
// A stone of age 1 is the one just placed (its preload used a stale last_placed): lut1 = lut[1].
template <int NW>
__device__ __forceinline__ void history_preload(const Geo<NW>& g, const uint32_t lastp[], int pre_ply,
                                                const float* __restrict__ lut, float hv_pre[]) {
  GO_UNROLL
  for (int k = 0; k < Geo<NW>::K; ++k) {
    int age = pre_ply - (int)lastp[k];
    age = age < 0 ? 0 : (age > HIST_LUT_N - 1 ? HIST_LUT_N - 1 : age);
    hv_pre[k] = (g.p[k] < NPTS) ? lut[age] : 0.f;
  }
}

template <int NW, bool ZEROS, bool WT>
__device__ __forceinline__ void write_features(BoardLds<NW>& L, const Geo<NW>& g, const uint32_t lab[],
                                               const uint32_t lastp[], const Scalars& s, int aug,
                                               const float* __restrict__ lut, float* __restrict__ out,
                                               const float hv_pre[], int pre_ply, float lut1,
                                               unsigned long long* dbg_stamps = nullptr) {
#ifdef GOENV_STAMPS
#define FSTAMP(i)                                                                                     \
  do {                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    unsigned long long t_ = __builtin_amdgcn_s_memtime();                                             \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    if (dbg_stamps && (threadIdx.x & 63) == 0 && threadIdx.x < 128)                                   \
      dbg_stamps[(size_t)blockIdx.x * 32 + (threadIdx.x >> 6) * 16 + (i)] = t_;                       \
  } while (0)
#else
#define FSTAMP(i) do { (void)dbg_stamps; } while (0)
#endif
  constexpr int K = Geo<NW>::K;
  const int lane = g.lane;
  float hv[K];
  GO_UNROLL
  for (int k = 0; k < K; ++k) {
    hv[k] = 0.f;
    if (g.p[k] < NPTS && lab[k]) {
      int age = s.ply - (int)lastp[k];
      age = age < 0 ? 0 : (age > HIST_LUT_N - 1 ? HIST_LUT_N - 1 : age);
      if (s.ply == pre_ply) hv[k] = age == 1 ? lut1 : hv_pre[k];
      else hv[k] = lut[age];
    }
  }
  // colour ballots (published for the other waves) and the per-point descriptor, scattered to its
  // OUTPUT index (D4 applied here): colour(2) | cls(2) | ko(1); history value in a parallel table
  // (the liberty table is dead).  One barrier covers both.
  unsigned long long wb[NWORDS + 1], ww[NWORDS + 1];
  GO_UNROLL
  for (int k = 0; k < K; ++k) {
    const unsigned long long bb = __ballot(col_of(lab[k]) == 1), bw = __ballot(col_of(lab[k]) == 2);
    if (NW == 1) { wb[k] = bb; ww[k] = bw; }
    else if (g.lane == 0) { L.xcol[NW * k + g.wave] = bb; L.xcol[NWORDS + NW * k + g.wave] = bw; }
  }
  const int ko_p = (s.koage == 0 && s.ko >= 0) ? s.ko : -1;  // GetSimpleKoLocation, board.cc:337-344
  float* hist = reinterpret_cast<float*>(L.lib);
  GO_UNROLL
  for (int k = 0; k < K; ++k) {
    if (g.p[k] < NPTS) {
      uint32_t d = typ_of(lab[k]);  // cls 1,2,3 <-> GetLibertyMap3binary :83-99
      if (g.p[k] == ko_p) d |= 16u;
      const int q = aug ? d4_dest(aug, g.p[k]) : g.p[k];
      L.desc[q] = d;
      hist[q] = hv[k];
    }
  }
  FSTAMP(13);
  WSYNC();
  FSTAMP(14);
  if (NW > 1) {
    GO_UNROLL
    for (int j = 0; j < NWORDS; ++j) { wb[j] = uni64(L.xcol[j]); ww[j] = uni64(L.xcol[NWORDS + j]); }
  }
  wb[NWORDS] = 0; ww[NWORDS] = 0;
  int has = 0;
  GO_UNROLL
  for (int j = 0; j < NWORDS; ++j) { if (wb[j]) has |= 1; if (ww[j]) has |= 2; }
  const uint32_t our = (uint32_t)s.next, opp = 3u - our;
  const bool has_our = (has >> (our - 1)) & 1, has_opp = (has >> (opp - 1)) & 1;
  const int qa = 4 * lane;
  const int qb = lane < 26 ? 256 + 4 * lane : NPTS - 4;
  const bool act_b = lane <= 26;
  uint32_t d[8];
  float h[8];

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
    store4<WT>(out + (pl) * NPTS + qa, v_[0], v_[1], v_[2], v_[3]);                 \
    if (act_b) store4<WT>(out + (pl) * NPTS + qb, v_[4], v_[5], v_[6], v_[7]);      \
  } while (0)

  // the 12 planes that need no distance, as 4 store groups dealt round-robin to the storing
  // waves (all waves but wave 0 when NW > 1)
  constexpr int NS = NW > 1 ? NW - 1 : 1;                 // storing waves in this phase
  // two waves per board: the one storing wave has 12 planes to compute and store while wave 0 runs the
  // distance transform, which takes half as long -- wave 0 takes the 11 constant planes after it
  constexpr bool ZEROS_WAVE0 = NW == 2;
#ifdef GOENV_SPLIT_ASC
  // probe: with two waves per board each wave's stores form ONE ascending contiguous run -- wave 1 planes 0..13,
  // wave 0 (after the distance transform) planes 14..24 -- the "contiguous halves" pattern of tools/store_bench.hip
  constexpr bool ASC = NW == 2;
#else
  constexpr bool ASC = false;
#endif
  const int sw = NW > 1 ? g.wave - 1 : 0;                 // this wave's index among them (-1: wave 0)
  if (NW == 1 || g.wave > 0) {
    GO_UNROLL
    for (int i = 0; i < 4; ++i) {
      d[i] = L.desc[qa + i];
      h[i] = hist[qa + i];
      d[4 + i] = L.desc[qb + i];
      h[4 + i] = hist[qb + i];
    }
    if (0 % NS == sw) {
      GO_PLANE(0, (mine && cls == 1) ? 1.f : 0.f);
      GO_PLANE(1, (mine && cls == 2) ? 1.f : 0.f);
      GO_PLANE(2, (mine && cls != 1 && cls != 2) ? 1.f : 0.f);  // literally "else", board_feature.cc:93
    }
    if (1 % NS == sw) {
      GO_PLANE(3, (theirs && cls == 1) ? 1.f : 0.f);
      GO_PLANE(4, (theirs && cls == 2) ? 1.f : 0.f);
      GO_PLANE(5, (theirs && cls != 1 && cls != 2) ? 1.f : 0.f);
    }
    if (2 % NS == sw) {
      GO_PLANE(6, (dd & 16u) ? 1.f : 0.f);
      GO_PLANE(7, mine ? 1.f : 0.f);
      GO_PLANE(8, theirs ? 1.f : 0.f);
    }
    if (3 % NS == sw) {
      GO_PLANE(9, col == 0 ? 1.f : 0.f);
      GO_PLANE(10, mine ? hh : 0.f);
      GO_PLANE(11, theirs ? hh : 0.f);
      if (ZEROS && (ASC || !ZEROS_WAVE0)) zero_range<WT>(out + 12 * NPTS, 2 * NPTS, lane);   // BORDER, POSITION_MARK: never filled
    }
    if (ZEROS && !ZEROS_WAVE0) {                                // planes 16..24: never filled
      if (4 % NS == sw) zero_range<WT>(out + 16 * NPTS, 4 * NPTS, lane);
      if (5 % NS == sw) zero_range<WT>(out + 20 * NPTS, 5 * NPTS, lane);
    }
  }
#ifndef GOENV_ABLATE_DIST
  if (g.wave == 0) distance_lines(L, lane, wb, ww, aug);
#endif
  if (ZEROS && ZEROS_WAVE0 && !ASC && g.wave == 0) {
    zero_range<WT>(out + 12 * NPTS, 2 * NPTS, lane);
    zero_range<WT>(out + 16 * NPTS, 9 * NPTS, lane);
  }
  FSTAMP(15);
  WSYNC();
  FSTAMP(11);
  // planes 14 / 15 straight from the distance tables (already in output order)
  if (g.wave == 0 || (NW > 1 && !ASC && g.wave == 1)) {
    const bool second = NW > 1 && !ASC && g.wave == 1;               // wave 1 takes plane 15
    for (int pass = 0; pass < ((NW == 1 || ASC) ? 2 : 1); ++pass) {
      const bool opp_plane = second || pass == 1;
      const uint8_t* dt = L.dist[(opp_plane ? opp : our) - 1];
      const bool present = opp_plane ? has_opp : has_our;
      GO_UNROLL
      for (int i = 0; i < 4; ++i) { d[i] = dt[qa + i]; d[4 + i] = dt[qb + i]; h[i] = 0.f; h[4 + i] = 0.f; }
      if (opp_plane) GO_PLANE(15, present ? (float)dd : 10000.f);
      else GO_PLANE(14, present ? (float)dd : 10000.f);
    }
    if (ASC && ZEROS) zero_range<WT>(out + 16 * NPTS, 9 * NPTS, lane);
  }
#undef GO_PLANE
#undef FSTAMP
}

// ---------------------------------------------------------------------------------------------------------
// Pipelined generator (k_generate_pipe): a board's workgroup is ONE rules wave (the NW == 1 code above, no
// `s_barrier` inside) and ONE feature wave.  The rules wave leaves the position it has just reached as a
// snapshot in LDS (its label and age tables, raw) and goes on with the next ply; the feature wave expands the snapshot into
// the sample's planes, mask and move.  The two waves meet at no barrier: a ring of PIPE_DEPTH snapshots and two LDS
// counters (produced / consumed) let the rules wave run ahead through the plies that are cheap for it.  The feature wave
// issues no load from memory at all (`vmcnt` counts loads and stores in one queue on this part: a wave that
// waits for a load waits for every plane store in front of it), and it writes a sample as ONE ascending run.
struct __attribute__((aligned(16))) PipeSnap {
  uint32_t raw[TAB_N];                  // label (colour, liberty class) | last_placed << 16 of every point, board order
  unsigned long long legal[NWORDS];     // legal mask of the side to move
  int32_t ply, next, ko, action;        // ko = GetSimpleKoLocation or -1
  int32_t aug, row, pad[2];             // k_generate_pipe_multi: the sample's D4 code and its row inside a ring slot
};

constexpr int PIPE_DEPTH = 4;           // snapshots in flight between the two waves
// (float)exp(-d / 10.0) is 0 from d = 1040 on: the part of the history table that is not, kept in LDS so that the rules
// wave's loop holds no load from memory (its loads would wait behind its own plane stores: one `vmcnt` queue)
constexpr int PIPE_LUT_N = 1056;
struct __attribute__((aligned(16))) PipeLds {
  float lut[PIPE_LUT_N];
  PipeSnap snap[PIPE_DEPTH];
  uint8_t dist[2][TAB_N];               // the feature wave's distance tables
  int32_t produced, consumed;           // snapshots written by the rules wave / taken by the feature wave so far
};

// Wait until the LDS counter has reached `want`.  Bounded, so that a wave always gets out: the two waves of a workgroup are
// resident together and neither waits for anything else, so the bound (seconds) is never met on a healthy device; if it is,
// the wave counts it into `stalls` (the word behind the context's statistics, which goenv_stats_host / goenv_finish turn
// into an error) and goes on, so the grid drains.
__device__ __forceinline__ void pipe_wait(const int32_t* counter, int want, unsigned long long* stalls) {
  int v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
  int spin = 0;
  for (; v < want && spin < (1 << 24); ++spin) {
    __builtin_amdgcn_s_sleep(2);
    v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
  }
  if (v < want && stalls && (threadIdx.x & 63) == 0) atomicAdd(stalls, 1ull);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void pipe_post(int32_t* counter, int value, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // a wave's LDS operations are executed in order
  if (lane == 0) __hip_atomic_store(counter, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Inverse of d4_dest: the board point that lands on output index q under a D4 code.
__device__ __forceinline__ int d4_src(int code, int q) {
  int rx = q / BS, ry = q - rx * BS;
  if ((code >> 2) == 1) { const int t = rx; rx = ry; ry = t; }
  const int rot = code & 3;
  int x, y;
  if (rot == 1) { x = BS - 1 - ry; y = rx; }
  else if (rot == 2) { x = BS - 1 - rx; y = BS - 1 - ry; }
  else if (rot == 3) { x = ry; y = BS - 1 - rx; }
  else { x = rx; y = ry; }
  return x * BS + y;
}

// What the feature wave keeps of a snapshot (the slot is handed back as soon as this is loaded).
struct SnapRegs {
  unsigned long long wb[NWORDS + 1], ww[NWORDS + 1], lg[NWORDS];
  uint32_t our;
  int action, ply, ko;
  uint32_t raw[8];   // the lane's eight OUTPUT points (4 lane .. 4 lane + 3, then 256 + 4 lane ..)
};
// src[i] = board point of the lane's i-th output point (constant over the launch: d4_src of the board's code)
__device__ __forceinline__ void snapshot_load(const PipeSnap& S, int lane, const int src[8], SnapRegs& r) {
  GO_UNROLL
  for (int j = 0; j < NWORDS; ++j) r.lg[j] = uni64(S.legal[j]);
  r.ply = __builtin_amdgcn_readfirstlane(S.ply);
  r.our = (uint32_t)__builtin_amdgcn_readfirstlane(S.next);
  r.ko = __builtin_amdgcn_readfirstlane(S.ko);
  r.action = __builtin_amdgcn_readfirstlane(S.action);
  GO_UNROLL
  for (int i = 0; i < 8; ++i) r.raw[i] = S.raw[src[i]];
  // colour ballots in board order for the distance transform
  GO_UNROLL
  for (int k = 0; k < NWORDS; ++k) {
    const int p = lane + 64 * k;
    const uint32_t col = p < NPTS ? col_of(S.raw[p]) : 0u;
    r.wb[k] = __ballot(col == 1);
    r.ww[k] = __ballot(col == 2);
  }
  r.wb[NWORDS] = 0; r.ww[NWORDS] = 0;
}

// A board that waits for its next turn in LDS (k_generate_pipe_multi: several boards per workgroup): the record with its two
// per-point tables packed into one dword per point (label | last_placed << 16), so that a turn starts with six LDS reads per
// lane, and the 40 uniform dwords behind them (scalars, legal and candidate words) read ONCE, by lane, and handed out with
// v_readlane.  Same size as the record: 1,632 B.
struct __attribute__((aligned(16))) SavedBoard {
  uint32_t raw[TAB_N];
  uint32_t uni[40];       // legal[6] as 12 dwords | cand[6] as 12 dwords | sc[16]: where the record has them
};
static_assert(sizeof(SavedBoard) == sizeof(BoardRec) && offsetof(BoardRec, legal) == offsetof(SavedBoard, uni) &&
                  offsetof(BoardRec, sc) == offsetof(SavedBoard, uni) + 24 * 4,
              "a saved board is a record with its two tables packed");

// One wave; the label image's rim is the caller's.  In two steps, so that the reads of the NEXT board can be in flight while
// the wave still finishes the current one: lds_fetch_board issues them, lds_load_board takes what they brought.
struct SavedRegs { uint32_t raw[NWORDS]; int uni; };
__device__ __forceinline__ void lds_fetch_board(const SavedBoard& R, const Geo<1>& g, SavedRegs& f) {
  GO_UNROLL
  for (int k = 0; k < NWORDS; ++k) f.raw[k] = g.p[k] < NPTS ? R.raw[g.p[k]] : 0u;
  f.uni = (int)R.uni[g.lane < 40 ? g.lane : 0];
}
__device__ __forceinline__ void lds_load_board(const SavedRegs& f, BoardLds<1>& L, const Geo<1>& g, uint32_t lab[],
                                               uint32_t lastp[], Scalars& s, Masks& m) {
  GO_UNROLL
  for (int k = 0; k < NWORDS; ++k) {
    lab[k] = f.raw[k] & 0xffffu;
    lastp[k] = f.raw[k] >> 16;
    if (g.p[k] < NPTS) L.lab[g.c[k]] = (uint16_t)lab[k];
  }
  const int mine = f.uni;
  int u[40];
  GO_UNROLL
  for (int i = 0; i < 40; ++i) u[i] = __builtin_amdgcn_readlane(mine, i);
  const int* sc = u + 24;
  s.ply = sc[SC_PLY]; s.next = sc[SC_NEXT]; s.bcap = sc[SC_BCAP]; s.wcap = sc[SC_WCAP];
  s.l1 = sc[SC_L1]; s.l2 = sc[SC_L2]; s.l3 = sc[SC_L3]; s.l4 = sc[SC_L4];
  s.koage = sc[SC_KOAGE]; s.ko = sc[SC_KO]; s.kocolor = sc[SC_KOCOLOR];
  s.done = sc[SC_DONE]; s.game = sc[SC_GAME]; s.counted = sc[SC_COUNTED]; s.seq = sc[SC_SEQ]; s.seqtot = sc[SC_SEQTOT];
  GO_UNROLL
  for (int j = 0; j < NWORDS; ++j) {
    m.legal[j] = ((unsigned long long)(uint32_t)u[1 + 2 * j] << 32) | (uint32_t)u[2 * j];
    m.cand[j] = ((unsigned long long)(uint32_t)u[13 + 2 * j] << 32) | (uint32_t)u[12 + 2 * j];
  }
  wsync<1>();
}
__device__ __forceinline__ void lds_store_board(SavedBoard& R, const Geo<1>& g, const uint32_t lab[], const uint32_t lastp[],
                                                const Scalars& s, const Masks& m) {
  GO_UNROLL
  for (int k = 0; k < NWORDS; ++k)
    if (g.p[k] < NPTS) R.raw[g.p[k]] = (lab[k] & 0xffffu) | (lastp[k] << 16);
  if (g.lane == 0) {
    unsigned long long* w = reinterpret_cast<unsigned long long*>(R.uni);
    GO_UNROLL
    for (int j = 0; j < NWORDS; ++j) { w[j] = m.legal[j]; w[NWORDS + j] = m.cand[j]; }
  }
  store_scalars(reinterpret_cast<BoardRec*>(&R), s, g.tid);   // sc[] sits where the record has it
}
// record <-> saved board, by `nthreads` threads of which this is thread `tid`
__device__ __forceinline__ void saved_from_record(SavedBoard& S, const BoardRec* rec, int tid, int nthreads) {
  for (int p = tid; p < TAB_N; p += nthreads) S.raw[p] = (uint32_t)rec->lab[p] | ((uint32_t)rec->lastp[p] << 16);
  const uint32_t* src = reinterpret_cast<const uint32_t*>(rec->legal);   // legal[6], cand[6], sc[16] are contiguous in the record
  for (int i = tid; i < 40; i += nthreads) S.uni[i] = src[i];
}
__device__ __forceinline__ void record_from_saved(BoardRec* rec, const SavedBoard& S, int tid, int nthreads) {
  for (int p = tid; p < TAB_N; p += nthreads) {
    const uint32_t r = S.raw[p];
    rec->lab[p] = (uint16_t)(r & 0xffffu);
    rec->lastp[p] = (uint16_t)(r >> 16);
  }
  uint32_t* dst = reinterpret_cast<uint32_t*>(rec->legal);
  for (int i = tid; i < 40; i += nthreads) dst[i] = S.uni[i];
}

// Rules wave: the snapshot of the position (lab, lastp, s) whose masks are `o`, reached by `action`.
__device__ __forceinline__ void snapshot_write(PipeSnap& S, const Geo<1>& g, const uint32_t lab[], const uint32_t lastp[],
                                               const Scalars& s, const Own<1>& o, int action) {
  GO_UNROLL
  for (int k = 0; k < NWORDS; ++k)
    if (g.p[k] < NPTS) S.raw[g.p[k]] = (lab[k] & 0xffffu) | (lastp[k] << 16);
  if (g.lane == 0) {
    GO_UNROLL
    for (int k = 0; k < NWORDS; ++k) S.legal[k] = o.legal[k];
    S.ply = s.ply;
    S.next = s.next;
    S.ko = (s.koage == 0 && s.ko >= 0) ? s.ko : -1;  // GetSimpleKoLocation, board.cc:337-344
    S.action = action;
  }
}

// Feature wave: BoardFeature::Extract (go/board_feature.cc:175-195) of a snapshot -- all 25 planes (ZEROS) or the 14
// that carry data, in ascending order -- then the legal mask as bytes.
template <bool ZEROS>
__device__ __forceinline__ void snapshot_expand(const SnapRegs& r, PipeLds& X, int lane, int aug, const int src[8],
                                                float* __restrict__ out, uint8_t* __restrict__ legal_out) {
  const uint32_t our = r.our, opp = 3u - our;
  int has = 0;
  GO_UNROLL
  for (int j = 0; j < NWORDS; ++j) { if (r.wb[j]) has |= 1; if (r.ww[j]) has |= 2; }
  const bool has_our = (has >> (our - 1)) & 1, has_opp = (has >> (opp - 1)) & 1;
  const int qa = 4 * lane;
  const int qb = lane < 26 ? 256 + 4 * lane : NPTS - 4;
  const bool act_b = lane <= 26;
  uint32_t d[8];
  float h[8];
  GO_UNROLL
  for (int i = 0; i < 8; ++i) {
    const uint32_t lab = r.raw[i] & 0xffffu;
    d[i] = typ_of(lab);   // cls 1,2,3 <-> GetLibertyMap3binary :83-99
    if (src[i] == r.ko) d[i] |= 16u;
    int age = r.ply - (int)(r.raw[i] >> 16);
    age = age < 0 ? 0 : (age > PIPE_LUT_N - 1 ? PIPE_LUT_N - 1 : age);   // lut[PIPE_LUT_N - 1 ..] == 0 (checked on the host)
    const float hv = X.lut[age];
    h[i] = lab ? hv : 0.f;
  }
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
    store4<false>(out + (pl) * NPTS + qa, v_[0], v_[1], v_[2], v_[3]);              \
    if (act_b) store4<false>(out + (pl) * NPTS + qb, v_[4], v_[5], v_[6], v_[7]);   \
  } while (0)
  GO_PLANE(0, (mine && cls == 1) ? 1.f : 0.f);
  GO_PLANE(1, (mine && cls == 2) ? 1.f : 0.f);
  GO_PLANE(2, (mine && cls != 1 && cls != 2) ? 1.f : 0.f);
  GO_PLANE(3, (theirs && cls == 1) ? 1.f : 0.f);
  GO_PLANE(4, (theirs && cls == 2) ? 1.f : 0.f);
  GO_PLANE(5, (theirs && cls != 1 && cls != 2) ? 1.f : 0.f);
  GO_PLANE(6, (dd & 16u) ? 1.f : 0.f);
  GO_PLANE(7, mine ? 1.f : 0.f);
  GO_PLANE(8, theirs ? 1.f : 0.f);
  GO_PLANE(9, col == 0 ? 1.f : 0.f);
  GO_PLANE(10, mine ? hh : 0.f);
  GO_PLANE(11, theirs ? hh : 0.f);
  if (ZEROS) zero_range<false>(out + 12 * NPTS, 2 * NPTS, lane);
#if !defined(GOENV_PIPE_ABLATE) || GOENV_PIPE_ABLATE != 4   // probe 4: no distance transform
  distance_lines(X, lane, r.wb, r.ww, aug);
#endif
  wsync<1>();
  for (int pass = 0; pass < 2; ++pass) {
    const bool opp_plane = pass == 1;
    const uint8_t* dt = X.dist[(opp_plane ? opp : our) - 1];
    const bool present = opp_plane ? has_opp : has_our;
    GO_UNROLL
    for (int i = 0; i < 4; ++i) { d[i] = dt[qa + i]; d[4 + i] = dt[qb + i]; h[i] = 0.f; h[4 + i] = 0.f; }
    if (opp_plane) GO_PLANE(15, present ? (float)dd : 10000.f);
    else GO_PLANE(14, present ? (float)dd : 10000.f);
  }
#undef GO_PLANE
  if (ZEROS) zero_range<false>(out + 16 * NPTS, 9 * NPTS, lane);
  if (legal_out) {
    GO_UNROLL
    for (int k = 0; k < NWORDS; ++k) {
      const int p = lane + 64 * k;
      if (p < NPTS) legal_out[p] = (uint8_t)((r.lg[k] >> lane) & 1ull);
    }
  }
  wsync<1>();   // the distance tables have been read before the next sample's transform rewrites them
}

template <int NW>
__device__ __forceinline__ void write_legal(const Geo<NW>& g, const Own<NW>& o, uint8_t* __restrict__ out) {
  GO_UNROLL
  for (int k = 0; k < Geo<NW>::K; ++k)
    if (g.p[k] < NPTS) out[g.p[k]] = (uint8_t)((o.legal[k] >> g.lane) & 1ull);
}

}  // namespace goenv
