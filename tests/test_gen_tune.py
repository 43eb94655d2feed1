"""Host logic of the generator's self-tuning rate limiter (elf_amd/csrc/gen_tune.h), on the CPU: a small C++ driver plays the tuner
against a made-up cost curve -- a 10 % cliff below the pace that the memory side sustains, the pace itself above it, noisy times that
come back a few launches late -- and the tuner must settle on the first candidate above the cliff, spend nearly all launches there, and
follow the cliff when it moves."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAND = [6.00, 6.10, 6.18, 6.25, 6.33, 6.40, 6.55, 6.80, 7.20]


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("gen_tune") / "gen_tune_driver")
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "gen_tune_driver.cc")], check=True)
    return exe


def run(driver, *args):
    out = subprocess.run([driver] + [str(a) for a in args], capture_output=True, text=True, check=True).stdout
    used, best, mean = out.split("|")
    return [int(x) for x in used.split()], int(best.split()[1]), float(mean.split()[1])


@pytest.mark.parametrize("cliff", [6.0, 6.2, 6.3, 6.5, 6.7, 7.0])
def test_settles_just_above_the_cliff(driver, cliff):
    want = next(i for i, c in enumerate(CAND) if c >= cliff)
    used, best, mean = run(driver, cliff, 400, 5, 3)
    assert best == want, (cliff, used, best)
    assert used[want] >= 0.9 * sum(used), (cliff, used)             # the rest: a neighbour every sixteenth launch
    assert mean <= (CAND[want] * 1.024 + 0.3) * 1.01, (cliff, mean)


def first_above(cliff):
    return next(i for i, c in enumerate(CAND) if c >= cliff)


def test_noise_and_late_results(driver):
    used, best, mean = run(driver, 6.3, 800, 15, 7)                  # +-1.5 % noise, results seven launches late
    want = first_above(6.3)
    assert want <= best <= want + 2, (used, best)                    # neighbours are 1 % apart, inside the noise
    assert sum(used[:want]) <= 0.06 * sum(used), used                # hardly ever over the cliff


def test_follows_a_cliff_that_moves(driver):
    used, best, mean = run(driver, 6.2, 1600, 5, 3, 400, 6.6)        # the card gets slower after 400 launches
    want = first_above(6.6)
    assert best == want, (used, best)
    assert used[want] >= 0.85 * sum(used), used
    used, best, mean = run(driver, 6.6, 2400, 5, 3, 400, 6.2)        # ... or faster: a step down per probe of the lower neighbour
    want = first_above(6.2)
    assert best == want, (used, best)
    assert used[want] >= 0.85 * sum(used), used


def test_a_burst_that_outruns_the_events(driver):
    """Results 20 launches late, 8 event pairs: launches that cannot be timed run at the best pace known so far (the safe one
    before any result), never at an untried candidate; the walk finishes all the same."""
    used, best, mean = run(driver, 6.3, 1200, 5, 20, -1, 6.3, 8)
    want = first_above(6.3)
    assert best == want, (used, best)
    assert used[want] >= 0.9 * sum(used), used
