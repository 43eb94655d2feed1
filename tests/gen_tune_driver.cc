// CPU driver for elf_amd/csrc/gen_tune.h (tests/test_gen_tune.py): the tuner against a made-up cost curve with a cliff.
// cost(pace) = a cliff below `cliff` (offered faster than the memory side sustains: +10 %), else the pace itself + overhead; results
// come back `lag` launches late, as the events of goenv_generate do.
// Usage: gen_tune_driver <cliff_ns> <launches> <noise_permille> <lag> [<shift_at> <new_cliff_ns> [<slots>]]
// slots: event pairs available (goenv_generate has 8); with lag >= slots a burst of launches outruns them
// Prints: launches spent per candidate in the last quarter, the final best candidate, the mean cost of the last quarter.
#include <cstdio>
#include <cstdlib>
#include <deque>

#include "../elf_amd/csrc/gen_tune.h"

static unsigned long long rng = 88172645463325252ull;
static double noise(int permille) {
  rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
  return 1.0 + ((double)(rng % 2001) / 1000.0 - 1.0) * permille / 1000.0;
}

int main(int argc, char** argv) {
  double cliff = argc > 1 ? atof(argv[1]) : 6.3;
  const int launches = argc > 2 ? atoi(argv[2]) : 400, permille = argc > 3 ? atoi(argv[3]) : 5, lag = argc > 4 ? atoi(argv[4]) : 3;
  const int shift_at = argc > 6 ? atoi(argv[5]) : -1;
  const double new_cliff = argc > 6 ? atof(argv[6]) : cliff;
  const int slots = argc > 7 ? atoi(argv[7]) : 1 << 20;
  GenTuneCore t = GenTuneCore();
  struct Flight { int cand, at; double cost; };
  std::deque<Flight> in_flight;
  int used[GenTuneCore::NC] = {0};
  double tail_cost = 0;
  int tail_n = 0;
  for (int i = 0; i < launches; ++i) {
    if (i == shift_at) cliff = new_cliff;
    while (!in_flight.empty() && i - in_flight.front().at > lag) { t.result(in_flight.front().cand, (float)in_flight.front().cost); in_flight.pop_front(); }
    bool timed = false;
    const int c = t.next((int)in_flight.size() < slots, timed);
    const double pace = GenTuneCore::cand[c];
    const double cost = (pace < cliff ? cliff * 1.10 + (cliff - pace) : pace) * 1.024 * noise(permille) + 0.3;   // us per ply at 1,024 boards
    t.enqueued(c, timed);
    if (timed) in_flight.push_back({c, i, cost});
    if (i >= launches - launches / 4) { used[c] += 1; tail_cost += cost; tail_n += 1; }
  }
  for (int c = 0; c < GenTuneCore::NC; ++c) printf("%d ", used[c]);
  printf("| best %d | mean %.4f\n", t.best(), tail_cost / tail_n);
  return 0;
}
