// gen_tune.h -- the self-tuning pace of the pipelined generator's rate limiter (host logic only: no HIP in here, so that
// tests/test_gen_tune.py can drive it on the CPU against a made-up cost curve).
//
// The offered store rate that the memory side takes best differs from card to card by a few per cent (cliff measured at
// 6.0 .. 6.5 ns per board-ply), and offering more than that costs 10 % (profiles/r03_pipe_generator.md).  goenv_generate
// times its own launches (events recorded around the kernel, read back later without waiting) and moves between a few
// candidate paces: every candidate twice in turn, then the best one by a running estimate -- re-timed every sixteenth launch,
// a neighbour tried every sixteenth because a card's state drifts, the other launches untimed.  The pace changes WHEN samples are written, never what is written.
#pragma once

struct GenTuneCore {
  static constexpr int NC = 9;
  // ns per board-ply offered, finer where the cliffs have been seen (6.05 .. 6.5): 36,465 B / 6.80 ns = 5.36 TB/s is the safe
  // side of all of them
  static constexpr float cand[NC] = {6.00f, 6.10f, 6.18f, 6.25f, 6.33f, 6.40f, 6.55f, 6.80f, 7.20f};
  static constexpr int SAFE = 7;
  float est[NC];          // us per ply, running estimate (valid where samples > 0)
  int samples[NC], launched[NC];
  int launches;

  // best candidate by the estimates so far; -1: nothing has come back yet
  int best() const {
    int b = -1;
    for (int i = 0; i < NC; ++i)
      if (samples[i] && (b < 0 || est[i] < est[b])) b = i;
    return b;
  }
  // the candidate of the next timed launch
  int pick() const {
    for (int round = 1; round <= 2; ++round)          // every candidate twice, in turn
      for (int i = 0; i < NC; ++i)
        if (launched[i] < round) return i;
    const int b = best();
    if (b < 0) return SAFE;
    if (launches % 16 == 15) {                        // now and then a neighbour
      const int nb = b + (((launches >> 4) & 1) ? 1 : -1);
      if (nb >= 0 && nb < NC) return nb;
    }
    return b;
  }
  // Whether the next launch is to be timed.  An event after a kernel costs the stream a system-scope release (about 25 us per
  // launch at 1,024 boards: 0.2 us per ply of a 128-ply launch), so once every candidate has been tried only one launch in
  // eight is timed: alternately the best candidate and, every sixteenth launch, the neighbour that pick() probes.
  bool want_timing() const {
    for (int i = 0; i < NC; ++i)
      if (launched[i] < 2) return true;
    return launches % 8 == 7;
  }
  // The candidate of the next launch and whether it is to be timed.  slot_free: an event pair is available.  A burst of
  // launches can outrun the events (their results come back only when the first launches have finished): a launch that
  // should be timed and cannot be teaches nothing, so it runs at the best pace known so far instead of an untried one.
  int next(bool slot_free, bool& timed) const {
    timed = want_timing() && slot_free;
    if (want_timing() && !slot_free) {
      const int b = best();
      return b < 0 ? SAFE : b;
    }
    return pick();
  }
  // a launch with candidate c has been enqueued (timed: its duration will be reported by result())
  void enqueued(int c, bool timed) {
    if (timed) launched[c] += 1;
    launches += 1;
  }
  // a timed launch with candidate c took us_per_ply
  void result(int c, float us_per_ply) {
    est[c] = samples[c] ? 0.5f * (est[c] + us_per_ply) : us_per_ply;
    samples[c] += 1;
  }
};
constexpr float GenTuneCore::cand[GenTuneCore::NC];
