/*
 * goenv.h -- C-ABI of libgoenv.so, the MI355X (gfx950) batched 19x19 Go
 * environment that replaces the per-game CPU hot path of ELF's DarkForest Go:
 *
 *   rules update   go/board.cc   TryPlay (:626-662) + Play (:1083-1179)
 *   legal moves    go/board.cc   FindAllValidMoves (:763-779)
 *   feature planes go/board_feature.cc  BoardFeature::Extract (:175-195)
 *   scoring        go/board.cc   GetTrompTaylorScore (:1584-1688)
 *
 * Every board lives in HBM as one packed record (GOENV_REC_BYTES); one gfx950
 * wavefront owns one board per launch.  All array arguments are DEVICE
 * pointers (hipMalloc / torch-ROCm data_ptr()) unless a comment says "host".
 * All launches are ordered on the caller's stream (`stream` = hipStream_t cast
 * to void*, NULL = default stream).  No function synchronises the device
 * except goenv_stats_host / goenv_last_error.
 *
 * Conventions shared with the reference:
 *   action      = x*19 + y  (go/board.h:155 EXPORT_OFFSET_XY), -1 = pass
 *                 (go/board_feature.h:67-71 Action2Coord(-1)=M_PASS),
 *                 -2 = resign (go/common.h:41 M_RESIGN; never produced by
 *                 Action2Coord, kept for completeness).
 *   colour      = 0 empty, 1 black, 2 white   (go/common.h:34-36)
 *   aug code    = rot (code%4: 0,CCW90,CCW180,CCW270) then flip (code>>2==1)
 *                 (go/go_state.h:81-86, go/board_feature.h:38-48)
 *
 * Return value of every int function: 0 = ok, <0 = error (goenv_last_error()
 * gives the text).  The library never aborts the process; the reference
 * reports the same conditions by bool return or by the deliberate segfault in
 * go/common.cc:36-48.
 */
#ifndef GOENV_H_
#define GOENV_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GOENV_BOARD_SIZE 19
#define GOENV_NUM_POINTS 361
#define GOENV_NUM_PLANES 25                       /* go/board_feature.h:13 */
#define GOENV_FEAT_FLOATS (25 * 361)              /* 9025 floats / board   */
#define GOENV_NUM_SCALARS 16
#define GOENV_NUM_STATS 16
#define GOENV_REC_BYTES 1632                      /* packed board record   */

#define GOENV_PASS (-1)
#define GOENV_RESIGN (-2)

/* indices into the int32 scalars[16] of the canonical view (goenv_get_state /
 * goenv_set_state).  They mirror the Board fields of go/board.h:92-122. */
enum {
  GOENV_S_PLY = 0,       /* _ply, starts at 1                       */
  GOENV_S_NEXT = 1,      /* _next_player                            */
  GOENV_S_BCAP = 2,      /* _b_cap: stones captured BY black        */
  GOENV_S_WCAP = 3,      /* _w_cap                                  */
  GOENV_S_LAST1 = 4,     /* _last_move  as action (-1 pass)         */
  GOENV_S_LAST2 = 5,
  GOENV_S_LAST3 = 6,
  GOENV_S_LAST4 = 7,
  GOENV_S_KO_AGE = 8,    /* _ko_age (u16 arithmetic)                */
  GOENV_S_KO = 9,        /* _simple_ko as action, -1 if never set   */
  GOENV_S_KO_COLOR = 10, /* _simple_ko_color                        */
  GOENV_S_DONE = 11,     /* ext: 1 = IsGameEnd (board.cc:1690), 2 = ply cap */
  GOENV_S_GAME = 12,     /* ext: games finished on this slot (game_counter) */
  GOENV_S_RSVD13 = 13,
  GOENV_S_SEQ = 14,      /* ext: samples emitted for the current game (SeqInfo.seq) */
  GOENV_S_SEQ_TOTAL = 15 /* ext: samples emitted on this slot since creation        */
};

/* indices into the int64 stats[16] vector (end-of-episode statistics that the
 * multi-GPU job all-reduces).  Slot 9 reduces with MAX, the rest
 * with SUM. */
enum {
  GOENV_ST_GAMES = 0,
  GOENV_ST_BLACK_WINS = 1,
  GOENV_ST_WHITE_WINS = 2,
  GOENV_ST_DRAWS = 3,
  GOENV_ST_PLIES = 4,
  GOENV_ST_BCAP = 5,
  GOENV_ST_WCAP = 6,
  GOENV_ST_SUM_LEN2 = 7,
  GOENV_ST_PLY_CAP_HITS = 8,
  GOENV_ST_MAX_LEN = 9,
  GOENV_ST_SUM_SCORE = 10 /* slots 11..15 are reserved (zero) */
};

/* flags for goenv_ply */
#define GOENV_F_RANDOM_POLICY 1u /* pick the move on device (cfg-2 policy)   */
#define GOENV_F_AUTO_RESTART 2u  /* finished boards are scored, counted in the
                                    stats vector and restarted in place      */
#define GOENV_F_SKIP_DONE 4u     /* finished boards are left untouched       */
#define GOENV_F_RESET_ILLEGAL 8u /* an illegal external action restarts the
                                    game in place ("No valid move ..., restarting
                                    the game", go/game.cc:81-90); status 0    */
#define GOENV_F_RESTART_AFTER_END 16u /* a move that ends the game (IsGameEnd /
                                    ply cap) is followed at once by scoring,
                                    stats and a restart in place; status 2    */
#define GOENV_F_COLLECT 32u      /* also emit the sample's sequencing record
                                    (goenv_sample_out) and advance seq         */

/* Per-sample sequencing / label record that the reference copies field by field
 * out of GameState (go/go_game_specific.h:59-114, keys of EntryFunc
 * go/game_context.h:67-83).  Each pointer is a device array [count] or NULL.
 * seq / game_counter / last_terminal follow SeqInfo (elf/common.h:11-19):
 * seq counts samples of the current game, game_counter games finished on the
 * slot, last_terminal = 1 on the first sample after a restart. */
typedef struct goenv_sample_out {
  int32_t* id;              /* board index inside this context              */
  int32_t* seq;
  int32_t* game_counter;
  int8_t* last_terminal;    /* type "char" in the reference                  */
  int32_t* move_idx;        /* _ply of the position (go/go_ai.h:82)          */
  int32_t* winner;          /* 0 for live play (go/go_ai.h:83)               */
  int32_t* aug_code;
  int32_t* game_record_idx; /* -1 for live play                              */
  uint32_t mode;            /* GOENV_SEQ_* below                             */
} goenv_sample_out;

/* goenv_sample_out.mode.  EPISODES (0): a restart in place starts a new episode -- seq back to
 * 0, game_counter + 1, last_terminal = 1 on its first sample (what AICommT::Restart does,
 * elf/ai_comm.h:91-95).  RUN_ON: the reference's Go selfplay loop restarts a game WITHOUT
 * calling Restart (go/game.cc:81-90), so there seq runs on, game_counter stays 0 and
 * last_terminal stays 0; this mode reproduces that. */
#define GOENV_SEQ_EPISODES 0u
#define GOENV_SEQ_RUN_ON 1u

typedef struct goenv_ctx goenv_ctx;

/* Library / lifetime ------------------------------------------------------ */

/* Version number of this ABI (GameContext::Version analogue,
 * elf/pybind_interface.h:86). */
int goenv_version(void);

/* Text of the last error raised on the calling thread ("" if none).  host. */
const char* goenv_last_error(void);

/* Create n_boards empty boards on HIP device `device` (replaces the
 * per-game `GoState` objects owned by GoGame, go/game.h:23-33 and the
 * ClearBoard call in GoState::Reset, go/go_state.cc:89-91).
 * board_id_base is the global id of board 0 (rank * n_boards in a sharded
 * job): the on-device random policy is keyed by the GLOBAL id so results do
 * not depend on the partition.  komi is used only for win statistics
 * (default of the reference: 7.5, go/sgf.h:92).  max_ply = ply cap of the
 * random playouts (0 = none). */
int goenv_create(int device, int n_boards, uint64_t seed, int64_t board_id_base,
                 float komi, int max_ply, goenv_ctx** out);
int goenv_destroy(goenv_ctx* ctx);
int goenv_num_boards(const goenv_ctx* ctx);
/* Launch shape: wavefronts that share one board (1, 2, 3 or 6; 0 = chosen per launch from the
 * batch size and from whether the launch writes planes, the default).  Results do not depend
 * on it.  No counterpart in the reference. */
int goenv_set_waves_per_board(goenv_ctx* ctx, int nw);
/* goenv_get_fuse_max: the largest selection that goenv_ply_policy_sel steps in ONE launch (above
 * it the sampler is a launch of its own; 3,072 unless GOENV_FUSE_MAX is set).  No counterpart in the
 * reference. */
int goenv_get_fuse_max(const goenv_ctx* ctx);
/* goenv_set_constant_planes_preset(ctx, 1): from now on no entry point of this context writes the 11
 * planes that BoardFeature::Extract never fills (12, 13, 16..24: go/board_feature.h:13-28,
 * go/board_feature.cc:178-179 -- constant zeros, 44 % of a sample's bytes).  For callers whose batch
 * tensors are long-lived and were zero-filled once (ELF allocates its batch tensors once, at
 * start-up, elf/utils_elf.py:52-99): the planes stay bit-identical to Extract's output while a ply
 * writes 20,216 B of planes per board instead of 36,100.  The caller owns that invariant: a tensor
 * that held anything else in those planes keeps it.  Off by default; no counterpart in the
 * reference (its Extract clears the whole vector on every call). */
int goenv_set_constant_planes_preset(goenv_ctx* ctx, int on);
/* device pointer to the packed records [n_boards][GOENV_REC_BYTES] */
void* goenv_state_ptr(goenv_ctx* ctx);

/* Reset / handicap ---------------------------------------------------------
 * ClearBoard (go/board.cc:30-46) followed by HandicapTable::Apply
 * (go/go_state.cc:61-70, PlaceHandicap go/board.cc:48-64) for the boards
 * idx[0..n) (idx == NULL: all boards, n ignored).  handicap[i] in 0..9 (NULL:
 * no handicap).  idx / handicap: device int32. */
int goenv_reset(goenv_ctx* ctx, const int32_t* idx, int n, const int32_t* handicap,
                void* stream);

/* Canonical view -----------------------------------------------------------
 * The observable part of `Board`: colours, per-stone group
 * liberties and stones, last_placed of stones, and the scalars above.  Group
 * ids, linked-list order and stale bytes are NOT part of it.  Arrays are in
 * export order x*19+y, [n_boards][361]; scalars [n_boards][16].  Any output
 * pointer may be NULL.  set_state rebuilds groups and liberties from colours
 * (connected components) -- used to stage arbitrary positions. */
int goenv_get_state(goenv_ctx* ctx, uint8_t* color, int16_t* liberties, int16_t* stones,
                    uint16_t* last_placed, int32_t* scalars, void* stream);
int goenv_set_state(goenv_ctx* ctx, const uint8_t* color, const uint16_t* last_placed,
                    const int32_t* scalars, void* stream);

/* Hot path -----------------------------------------------------------------
 * goenv_step: GoState::forward (go/go_state.cc:73-78) = TryPlay2 + Play for
 * every board: actions[b] is applied for the side to move; an illegal move
 * leaves the board untouched and sets status[b] = 0 (legal: 1; the caller
 * decides what an illegal move means, e.g. restart, go/game.cc:81-90).
 * legal_out (u8 [n][361], may be NULL) receives FindAllValidMoves
 * (go/board.cc:763-779) of the NEW position for the NEW side to move. */
int goenv_step(goenv_ctx* ctx, const int64_t* actions, uint8_t* legal_out, int32_t* status_out,
               void* stream);

/* goenv_features: BoardFeature::Extract (go/board_feature.cc:175-195) of the
 * current positions into float32 [n][25][19][19] (NCHW, inner index x'*19+y'
 * after the D4 code, go/board_feature.h:80-91).  aug_codes: int32 [n] or NULL
 * (code 0).  legal_out: optional legal mask of the current position (not
 * augmented, export order). */
int goenv_features(goenv_ctx* ctx, const int32_t* aug_codes, float* feat_out,
                   uint8_t* legal_out, void* stream);

/* goenv_features_static: goenv_set_state + goenv_features without a context and WITHOUT ANY
 * ALLOCATION.  It has no counterpart in the reference; it is for callers that cannot allocate
 * device memory (code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc) and so cannot call goenv_create.  Positions in (canonical view, as goenv_set_state:
 * color u8 [n][361], last_placed u16 [n][361] or NULL, scalars int32 [n][16]), out
 * BoardFeature::Extract (go/board_feature.cc:175-195) into feat_out float32 [n][25][19][19] and
 * FindAllValidMoves (go/board.cc:763-779) into legal_out u8 [n][361]; aug_codes int32 [n] or NULL;
 * either output may be NULL, not both.  The packed records in between and the history table live
 * in a static device array of the library (8 KB + 4,096 x 1,632 B = 6.7 MB per GPU, held until
 * the process ends): n > GOENV_STATIC_MAX_BOARDS is an error.  The first call on a device looks up
 * that array's address and uploads the 8-KB history table (a synchronous copy; make one call
 * before starting a graph capture); after that the call only enqueues two kernels on `stream`.
 * All calls on one GPU share the array, so calls must not overlap on one GPU: calls on one stream
 * are ordered by it; a call from ANOTHER stream while the previous call's kernels are still in
 * flight is refused with an error (it would overwrite records they read).  A first call made
 * inside a stream capture is refused too (the set-up copy is synchronous).  Replays of a captured
 * call are the graph owner's to order. */
#define GOENV_STATIC_MAX_BOARDS 4096
int goenv_features_static(int device, int n, const uint8_t* color, const uint16_t* last_placed,
                          const int32_t* scalars, const int32_t* aug_codes, float* feat_out,
                          uint8_t* legal_out, void* stream);

/* goenv_ply: the fused per-ply hot path (one launch): choose (flag RANDOM:
 * uniform over legal moves that do not fill an own true eye,
 * go/board.cc:1501-1552, pass if none) or take actions[b]; apply it; then
 * write the legal mask and the 25 planes of the new position straight into
 * the batch buffers.  actions_out (int32 [n], may be NULL) receives the move
 * applied (-1 pass, -3 = board was finished and skipped).  Any of legal_out /
 * feat_out / status_out / aug_codes may be NULL. */
int goenv_ply(goenv_ctx* ctx, uint32_t flags, const int64_t* actions, const int32_t* aug_codes,
              float* feat_out, uint8_t* legal_out, int32_t* actions_out, int32_t* status_out,
              void* stream);

/* goenv_ply_repeat: n_steps successive goenv_ply launches enqueued from native code (the same
 * arguments every time: meant for the on-device policy, where each ply reads the state the
 * previous one wrote).  Saves the per-call cost of a scripting-language binding when a caller
 * steps many plies without looking at the outputs in between. */
int goenv_ply_repeat(goenv_ctx* ctx, int n_steps, uint32_t flags, const int64_t* actions,
                     const int32_t* aug_codes, float* feat_out, uint8_t* legal_out, int32_t* actions_out,
                     int32_t* status_out, void* stream);

/* goenv_ply_repeat_parts: as goenv_ply_repeat, with the batch cut into n_parts (1..GOENV_MAX_PARTS)
 * contiguous parts that are stepped on streams of their own (the library's), forked from and
 * joined to `stream` around the whole sequence.  Every board gets exactly n_steps plies and the
 * results equal goenv_ply_repeat's; the parts are not in step with each other in between, so
 * the front of one part's launch (launch latency, record round trip, rules) runs under the
 * store drain of another's.  The counterpart in the reference is its set of collectors, each
 * batching whichever games are ready (elf/comm_template.h:170-181, num_collectors). */
#define GOENV_MAX_PARTS 8
int goenv_ply_repeat_parts(goenv_ctx* ctx, int n_steps, int n_parts, uint32_t flags, const int64_t* actions,
                           const int32_t* aug_codes, float* feat_out, uint8_t* legal_out, int32_t* actions_out,
                           int32_t* status_out, void* stream);

/* Selections.  The three calls below act on `count` boards: board of row b is
 * idx[b] (device int32) when idx != NULL, else first + b.  Every batch array is
 * indexed by the ROW b, so a collector's batch tensors can be filled from any
 * subset of the resident boards (this is what replaces CollectorGroupT's
 * per-key memcpy, elf/state_collector.h:251-301, elf/hist.h:71-121).  An index outside
 * [0, n_boards) leaves its row untouched (status -2 where a status is written); the indices of
 * a call that MODIFIES boards (goenv_ply_sel, goenv_replay) must be distinct -- two rows on one
 * board would race.
 *
 * goenv_ply_sel     = goenv_ply on the selection (+ optional sample record).
 * goenv_collect_sel = AIWithCommT::Act's Prepare + extract (elf/ai.h:69-80,
 *                     go/go_ai.h:79-87): planes, mask and, when meta != NULL, the
 *                     sequencing record of the CURRENT positions; seq advances.
 * goenv_sample_policy: the move selection the reference leaves to MCTS /
 *                     Python (SURVEY 3.1 note ii): sample a legal point with
 *                     probability proportional to floor(pi*2^24) (exact integer
 *                     arithmetic, counter RNG); uniform cfg-2 policy when every
 *                     weight is 0.  pi float32 [count][361] in the frame of
 *                     aug_codes; actions_out int64 [count] in board frame. */
int goenv_ply_sel(goenv_ctx* ctx, const int32_t* idx, int first, int count, uint32_t flags,
                  const int64_t* actions, const int32_t* aug_codes, float* feat_out,
                  uint8_t* legal_out, const goenv_sample_out* meta, int32_t* actions_out,
                  int32_t* status_out, void* stream);
/* goenv_ply_policy_sel: a whole selfplay step of the selected boards in ONE launch -- what
 * AIWithCommT::Act does around one reply (elf/ai.h:69-80) and GoGame::Act after it
 * (go/game.cc:80-93): the move is sampled from the policy head `pi` over the legal points
 * (as goenv_sample_policy; pi float32 [count][361], row b in the frame of pi_aug_codes[b],
 * NULL = code 0), applied (flags as goenv_ply_sel: RESET_ILLEGAL, RESTART_AFTER_END, COLLECT ...;
 * not RANDOM_POLICY or AUTO_RESTART: the move is sampled for the position as stored),
 * and the NEXT sample's planes (frame aug_codes), legal mask and sequencing record are written
 * straight into the batch arrays.  actions_out int32 [count] receives the sampled move.  One
 * launch up to 3,072 boards; above, the sampler runs as a small launch of its own in front of
 * the ply kernel (measured faster there; same moves and outputs).  Neither form allocates: the
 * call can be captured into a HIP graph. */
int goenv_ply_policy_sel(goenv_ctx* ctx, const int32_t* idx, int first, int count, uint32_t flags,
                         const float* pi, const int32_t* pi_aug_codes, const int32_t* aug_codes,
                         float* feat_out, uint8_t* legal_out, const goenv_sample_out* meta,
                         int32_t* actions_out, int32_t* status_out, void* stream);
int goenv_collect_sel(goenv_ctx* ctx, const int32_t* idx, int first, int count,
                      const int32_t* aug_codes, float* feat_out, uint8_t* legal_out,
                      const goenv_sample_out* meta, void* stream);
int goenv_sample_policy(goenv_ctx* ctx, const int32_t* idx, int first, int count, const float* pi,
                        const int32_t* aug_codes, int64_t* actions_out, void* stream);

/* goenv_topk_policy: the k legal points of largest policy weight of every selected board, in
 * descending order -- what DirectPredictAI::get_last_pi (go/go_ai.h:24-68) gives the tree search
 * (k = 5 there: all 361 sorted by pi, the first five that pass CheckMove kept).  pi float32
 * [count][361], indexed in the frame of aug_codes (int32 [count] or NULL = 0); actions_out int32
 * [count][k] board points p = x*19+y (-1 once the legal points run out); probs_out float32
 * [count][k] their weights (may be NULL).  Ties go to the lower point (the reference's std::sort
 * leaves them unspecified); NaN weights are skipped. */
int goenv_topk_policy(goenv_ctx* ctx, const int32_t* idx, int first, int count, const float* pi,
                      const int32_t* aug_codes, int k, int32_t* actions_out, float* probs_out, void* stream);

/* goenv_get_state_sel: goenv_get_state for a selection (rows = selection rows). */
int goenv_get_state_sel(goenv_ctx* ctx, const int32_t* idx, int first, int count, uint8_t* color,
                        int16_t* liberties, int16_t* stones, uint16_t* last_placed, int32_t* scalars,
                        void* stream);

/* goenv_replay: replay of packed move lists, the device half of OfflineLoader::next
 * (go/offpolicy_loader.cc:120-126: s().forward(curr().GetCoord()), stop on an illegal
 * move) and of the pre-move fast-forward in after_reload (:94-101).  Row b applies
 * moves[b*stride + 0 .. nmoves[b]) (actions, -1 pass) to its board in ONE launch and
 * stops at the first illegal move; napplied_out[b] = moves applied.  moves: device
 * int16 [count][stride]; nmoves / napplied_out: device int32 [count]. */
int goenv_replay(goenv_ctx* ctx, const int32_t* idx, int first, int count, const int16_t* moves,
                 int stride, const int32_t* nmoves, int32_t* napplied_out, void* stream);

/* Offline (SGF replay) batches -------------------------------------------------
 * goenv_offline_batch: the device half of ONE batch of the reference's OfflineLoader
 * (go/offpolicy_loader.cc) in one launch.  For every row: Reload's Reset + ApplyHandicap
 * (elf/replay_loader.h:28-35, go/go_state.cc:85-91) when the row says so, then next()'s
 * s().forward(move) for each of the row's moves (:120-126, and the pre-move loop of after_reload
 * :94-101), then extract (:128-145): planes under the row's D4 code into feat_out[row], the legal
 * mask into legal_out[row], and the labels the host computed -- id / seq / game_counter /
 * last_terminal / move_idx / winner / aug_code / game_record_idx and offline_a[k_future] -- into
 * `labels` / offline_a_out (device arrays [count] / [count][k_future]; any may be NULL).
 * `block` is what the host packed: count rows of (12 + 2 k_future) int32 words
 *   [0] board  [1] aug | reset << 3 | handicap << 4 | nmoves << 8  [2] first move of the row,
 *   in int16 units from block + moves_offset  [3] 0  [4..12) the eight labels in goenv_sample_out
 *   order  [12..) offline_a as k_future int64
 * followed at moves_offset (even) by the rows' moves as int16 actions.  It may be device memory
 * or a block of the staging ring below, which the kernel reads in place.  Boards of one call must be
 * distinct.  The caller is expected to pass moves it knows to be legal (a record's legality is a
 * property of the record: replay it once with goenv_replay when it is loaded); a move the device
 * rejects all the same stops its row and is counted (goenv_offline_bad_moves, which synchronises).
 * goenv_stage_acquire: the next block of a ring of GOENV_STAGE_SLOTS pinned host blocks of at least
 * `bytes` bytes (host pointer; waits until the launch that read this block last has finished).  The
 * block is taken by the next goenv_offline_batch call that is given it. */
#define GOENV_STAGE_SLOTS 16
#define GOENV_MAX_FUTURE 16
int goenv_stage_acquire(goenv_ctx* ctx, size_t bytes, void** host_block);
int goenv_offline_batch(goenv_ctx* ctx, const void* block, int count, int k_future, size_t moves_offset,
                        float* feat_out, uint8_t* legal_out, const goenv_sample_out* labels,
                        int64_t* offline_a_out, void* stream);
int goenv_offline_bad_moves(goenv_ctx* ctx, int32_t* out_host, void* stream);

/* goenv_copy_boards: board dst_idx[i] becomes a copy of board src_idx[i] (device int32 [n];
 * the destinations must be distinct and must not appear among the sources of the same call).
 * This is CopyBoard (go/board.h, used by GoState's copy constructor go/go_state.h:47-49) for
 * callers that branch positions, e.g. a host-side tree search evaluating its leaves in device
 * batches. */
int goenv_copy_boards(goenv_ctx* ctx, const int32_t* src_idx, const int32_t* dst_idx, int n,
                      void* stream);

/* goenv_copy_rows: row dst_rows[i] of `dst` <- row src_rows[i] of `src`, i in [0, n), rows of row_bytes
 * bytes in row-major tables on the context's device (a NULL list means row i; a negative row is
 * skipped; destination rows must be distinct).  The per-key copy of the reference's collectors between
 * a game's history and the batch memory (CopyToMem / CopyFromMem, elf/hist.h:71-121,
 * elf/copier.hh:97-101), for histories that live on the device. */
int goenv_copy_rows(goenv_ctx* ctx, const void* src, const int32_t* src_rows, void* dst,
                    const int32_t* dst_rows, int n, size_t row_bytes, void* stream);

/* Memory helpers for hosts that do not link the HIP runtime themselves (the
 * GameContext layer): pointer classification (EntryInfo.p may be host or
 * device, SURVEY 8b "Ownership"), stream-ordered copy in either direction,
 * stream sync, staging allocation on the context's device. */
int goenv_pointer_is_device(const void* p);
/* copy and sync run with the context's device current, on `stream` of THAT device (NULL = its null
 * stream): pass the stream the context's launches use, so copies and kernels are ordered */
int goenv_copy(goenv_ctx* ctx, void* dst, const void* src, size_t bytes, void* stream);
int goenv_sync(goenv_ctx* ctx, void* stream);
/* stream-ordered byte fill of device memory on the context's device */
int goenv_memset(goenv_ctx* ctx, void* dst, int value, size_t bytes, void* stream);
int goenv_device_alloc(goenv_ctx* ctx, size_t bytes, void** out);
int goenv_device_free(goenv_ctx* ctx, void* p);
/* pinned host staging memory (for small index / action / status arrays) */
int goenv_host_alloc(size_t bytes, void** out);
int goenv_host_free(void* p);

/* goenv_playout: random playouts to the end inside ONE persistent launch
 * (state stays in LDS between plies); stops per board at game end or at the
 * context's ply cap.  moves_out (int16 [n][max_record], may be NULL) records
 * each board's move sequence for replay through the reference; nmoves_out
 * int32 [n]. */
int goenv_playout(goenv_ctx* ctx, int16_t* moves_out, int max_record, int32_t* nmoves_out,
                  void* stream);

/* goenv_generate: n_plies plies of the on-device cfg-2 policy on EVERY board inside one
 * persistent launch (finished games are scored, counted and restarted in place, as
 * GOENV_F_AUTO_RESTART does), for callers without a model in the loop (random-playout data
 * generation).  Ply t's outputs go to slot t of the caller's ring buffers:
 * feat_ring float32 [n_plies][n][25][19][19], legal_ring u8 [n_plies][n][361], actions_ring
 * int32 [n_plies][n]; any may be NULL.  Results equal n_plies calls of
 * goenv_ply(RANDOM_POLICY | AUTO_RESTART).  aug_codes: int32 [n] or NULL.  No counterpart in
 * the reference (its playouts run one thread per game on the CPU). */
int goenv_generate(goenv_ctx* ctx, int n_plies, const int32_t* aug_codes, float* feat_ring,
                   uint8_t* legal_ring, int32_t* actions_ring, void* stream);

/* goenv_generate_pace (diagnostic): with all boards resident on the device, goenv_generate
 * rate-limits its plane stores and tunes that rate by timing its own launches (events, never
 * a wait).  pace_ns_per_board_out: the pace it has settled on, in ns per board-ply (0: none
 * yet / not in use); est_out: float[GOENV_GEN_PACE_CANDIDATES] or NULL, its time estimates per candidate pace in us
 * per ply (0 = not measured); pipelined_out: 1 if the last goenv_generate of this context
 * went through the pipelined kernel (k_generate_pipe), 0 if through k_generate.  Any may be
 * NULL.  The pace changes when samples are written, never their bytes.
 * GOENV_GEN_PACE_NS=<ns between two samples of a board> pins it (0 = no limiter),
 * GOENV_GEN_PIPE=0 selects the non-pipelined generator kernel.  No counterpart in the
 * reference. */
#define GOENV_GEN_PACE_CANDIDATES 9
int goenv_generate_pace(goenv_ctx* ctx, float* pace_ns_per_board_out, float* est_out,
                        int* pipelined_out);

/* goenv_score: GetTrompTaylorScore(board, NULL, NULL) (go/board.cc:1584-1688)
 * into score_out float32 [n] (black minus white, no komi; 0 on an empty
 * board); fast_out (may be NULL) = GetFastScore(board, RULE_CHINESE)
 * (go/board.cc:1560-1582). */
int goenv_score(goenv_ctx* ctx, float* score_out, float* fast_out, void* stream);
/* goenv_fast_score: GetFastScore(board, rule) (go/board.cc:1560-1582) into out float32 [n];
 * rule 0 = RULE_CHINESE, 1 = RULE_JAPANESE. */
int goenv_fast_score(goenv_ctx* ctx, int rule, float* out, void* stream);

/* goenv_tactics: the heuristic move filters of go/board.cc for the side to move of the selected
 * boards (idx / first / count as in goenv_collect_sel), one value per point in export order
 * p = x*19+y; row r of every output belongs to the r-th selected board; any pointer may be NULL.
 *   self_atari int16 [count][361]    IsSelfAtari(board, NULL, c, next, &n) (go/board.cc:168-200):
 *                                    n = stones of the group left with one liberty, 0 = not a
 *                                    self-atari (or not a legal move)
 *   gives_ko   u8    [count][361]    IsMoveGivingSimpleKo (go/board.cc:320-335) of a legal move
 *   cand       u8    [count][361]    FindAllCandidateMoves(board, next, self_atari_thres)
 *                                    (go/board.cc:685-715): legal, not an own true eye, not a
 *                                    self-atari of >= self_atari_thres stones
 *   libs       int16 [count][361][2] the liberties of the group of the stone at p when it has one
 *                                    (find_only_liberty, go/board.cc:1042-1058: {m, -1}) or two
 *                                    (find_two_liberties, :1060-1081: ascending; the reference's
 *                                    order follows its stone list), else {-1, -1}
 *   ladder     int16 [count][361]    CheckLadder (go/board.cc:348-391, search :203-317) of a legal
 *                                    move: depth of the ladder it runs into, 0 = none; -1 = the
 *                                    search had more than 16 two-way branches pending at once, or
 *                                    ran past 4 x 361 stones, and was abandoned (the reference recurses without that bound; on
 *                                    game records the maximum seen is 4)
 * Boards are not modified. */
typedef struct goenv_tactics_out {
  int16_t* self_atari;
  uint8_t* gives_ko;
  uint8_t* cand;
  int16_t* libs;
  int16_t* ladder;
} goenv_tactics_out;
int goenv_tactics(goenv_ctx* ctx, const int32_t* idx, int first, int count, int self_atari_thres,
                  const goenv_tactics_out* out, void* stream);

/* Statistics ---------------------------------------------------------------
 * goenv_finish: score every finished board that has not been counted yet and
 * add it to the device stats vector (boards keep their state).
 * goenv_stats_ptr: device int64[GOENV_NUM_STATS] -- this is the buffer a
 * multi-GPU job hands to RCCL all-reduce (torch.distributed).  goenv_stats_host
 * copies it to a host int64[16] (synchronises the stream); it fails, after
 * filling out16_host, if a wait inside the pipelined generator ever ran into
 * its bound (a device-side counter behind the vector: never on a healthy
 * device; the rings of such a launch are not to be trusted). */
int goenv_finish(goenv_ctx* ctx, void* stream);
int64_t* goenv_stats_ptr(goenv_ctx* ctx);
int goenv_stats_host(goenv_ctx* ctx, int64_t* out16_host, void* stream);
int goenv_stats_clear(goenv_ctx* ctx, void* stream);
/* goenv_stats_allreduce: the one collective of a sharded job for hosts that hold an RCCL
 * communicator themselves (a C++ host keeping ELF's own GameContext; Python callers use
 * torch.distributed on goenv_stats_ptr, elf_amd/sharding.py).  rccl_comm = ncclComm_t of a
 * communicator with one rank per GPU; the context's stats vector becomes the SUM over the ranks
 * (slot GOENV_ST_MAX_LEN: the MAX), in place, ordered on `stream`.  librccl.so is opened by the first
 * call.  No counterpart in the reference (it is single-process). */
int goenv_stats_allreduce(goenv_ctx* ctx, void* rccl_comm, void* stream);

/* Timing helper for bench.py: HIP events on the stream the kernels run on
 * (torch.cuda.Event only sees torch's current stream).  goenv_timer_begin /
 * _end bracket a region; goenv_timer_ms synchronises and returns its length. */
int goenv_timer_begin(goenv_ctx* ctx, void* stream);
int goenv_timer_end(goenv_ctx* ctx, void* stream);
/* busy-polls until the end event has passed (a blocking wait wakes up tens of microseconds late) */
int goenv_timer_spin(goenv_ctx* ctx);
float goenv_timer_ms(goenv_ctx* ctx);

/* D4 / action helpers (host, pure): go/board_feature.h:38-71.  Actions are
 * export indices; goenv_aug_action maps a board action to the action index in
 * the augmented frame (Coord2Action), goenv_aug_action_inv back
 * (Action2Coord).  -1 stays -1. */
int goenv_aug_action(int aug_code, int action);
int goenv_aug_action_inv(int aug_code, int action);

#ifdef __cplusplus
}
#endif
#endif /* GOENV_H_ */
