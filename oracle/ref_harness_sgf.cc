// TEST INFRASTRUCTURE -- not part of the product path.
//
// Second half of the reference harness (see ref_harness.cc): the handicap table
// (go/go_state.cc:32-70) and the SGF parser (go/sgf.cc) of the unmodified
// reference, exposed so tests can pin our own handicap table and SGF reader
// against them.  Compiled with go/go_state.cc, go/sgf.cc, elf/tar_loader.cc and
// vendor/microtar.c where they lie under /root/reference.

#include <stdint.h>

#include "go_state.h"  // -I/root/reference/go -I/root/reference
#include "sgf.h"

static inline int coord_to_action2(Coord c) {
  if (c == M_PASS) return -1;
  if (c == M_RESIGN) return -2;
  return EXPORT_OFFSET(c);
}

extern "C" {

// HandicapTable::Apply (go/go_state.cc:61-70) on a raw Board.
void goref_apply_handicap(void* b, int handi) { GoState::handi_table().Apply(handi, (Board*)b); }

// Sgf::Load + main-variation walk (go/sgf.cc:27-57, go/sgf.h:130-177).
// Returns the number of nodes on the main variation (including passes), or -1.
// moves/players receive up to max_out entries; *num_moves = Sgf::NumMoves() (stops at
// the first pass); *winner = +1 black / -1 white / 0 unknown as OfflineLoader::extract
// maps it (go/offpolicy_loader.cc:132-133).
int goref_sgf_load(const char* path, int16_t* moves, uint8_t* players, int max_out, int* num_moves,
                   int* winner, int* handi, int* board_size) {
  Sgf sgf;
  if (!sgf.Load(path)) return -1;
  int n = 0;
  for (auto it = sgf.begin(); !it.done(); ++it) {
    SgfMove m = it.GetCurrMove();
    if (n < max_out) {
      if (moves) moves[n] = (int16_t)coord_to_action2(m.move);
      if (players) players[n] = m.player;
    }
    ++n;
  }
  if (num_moves) *num_moves = sgf.NumMoves();
  if (winner) *winner = sgf.GetWinner() == S_BLACK ? 1 : (sgf.GetWinner() == S_WHITE ? -1 : 0);
  if (handi) *handi = sgf.GetHandicapStones();
  if (board_size) *board_size = sgf.GetBoardSize();
  return n;
}

}  // extern "C"
