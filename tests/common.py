"""Helpers shared by the CPU and GPU parity tests."""
import hashlib
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

CANON_KEYS = ("color", "liberties", "stones", "last_placed")
N_CANON_SCALARS = 11  # ply .. ko_color: the Board fields; slots 11+ are extensions


def position_digest(canon, legal, planes):
    """16-hex-digit digest of one position's observable state: canonical view
     + legal mask + the 25 planes' float bit patterns."""
    h = hashlib.blake2b(digest_size=8)
    for k in CANON_KEYS:
        h.update(np.ascontiguousarray(canon[k]).tobytes())
    h.update(np.ascontiguousarray(canon["scalars"][:N_CANON_SCALARS], np.int32).tobytes())
    h.update(np.ascontiguousarray(legal, np.uint8).tobytes())
    h.update(np.ascontiguousarray(planes, np.float32).tobytes())
    return h.hexdigest()


def board_digest(board, aug=0):
    return position_digest(board.canonical(), board.legal(), board.extract(aug))


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def assert_canon_equal(a, b, where=""):
    for k in CANON_KEYS:
        if not np.array_equal(np.asarray(a[k]), np.asarray(b[k])):
            bad = np.flatnonzero(np.asarray(a[k]) != np.asarray(b[k]))
            raise AssertionError("%s: canonical field %s differs at points %s: %s vs %s" % (
                where, k, bad[:8], np.asarray(a[k])[bad[:8]], np.asarray(b[k])[bad[:8]]))
    sa = np.asarray(a["scalars"])[:N_CANON_SCALARS]
    sb = np.asarray(b["scalars"])[:N_CANON_SCALARS]
    if not np.array_equal(sa, sb):
        raise AssertionError("%s: scalars differ: %s vs %s" % (where, sa, sb))


def assert_planes_equal(a, b, where=""):
    ua = np.ascontiguousarray(a, np.float32).reshape(-1).view(np.uint32)
    ub = np.ascontiguousarray(b, np.float32).reshape(-1).view(np.uint32)
    if not np.array_equal(ua, ub):
        bad = np.flatnonzero(ua != ub)
        raise AssertionError("%s: planes differ at %d floats, first (plane,idx)=%s: %r vs %r" % (
            where, bad.size, [(int(i) // 361, int(i) % 361) for i in bad[:6]],
            np.asarray(a).reshape(-1)[bad[:6]], np.asarray(b).reshape(-1)[bad[:6]]))


# hand-built edge cases: (name, list of (x, y) moves alternating from black unless 'pass'),
# each followed by probes.  Coordinates are (x, y) with action = x*19+y.
def A(x, y):
    return x * 19 + y


_KO_SETUP = [A(3, 2), A(3, 3), A(2, 3), A(2, 4), A(4, 3), A(4, 4), A(10, 10), A(3, 5), A(3, 4)]  # B takes: ko at (3,3)

EDGE_CASES = {
    # black captures a corner stone
    "corner_capture": [A(0, 1), A(0, 0), A(1, 0), A(0, 0)],
    # ko: W may not retake at once (illegal probe), retakes after a tenuki pair, then B is barred
    "simple_ko": _KO_SETUP + [A(3, 3), A(15, 15), A(15, 16), A(3, 3), A(3, 4), -1, A(16, 16), A(3, 4)],
    # passes do not age a ko (Play's early return, board.cc:1092-1095): after W pass + B pass the
    # ko point is STILL barred for W
    "ko_then_pass": _KO_SETUP + [-1, -1, A(3, 3), A(16, 16), A(16, 15), A(3, 3)],
    # suicide attempts (single stone and multi-stone) are rejected, state untouched
    "suicide": [A(0, 1), A(10, 10), A(1, 0), A(0, 0), A(11, 11), A(12, 12), A(0, 0),
                A(2, 0), A(13, 13), A(1, 1), A(0, 0), A(14, 14), A(0, 2), A(0, 0)],
    # one move captures two separate groups
    "multi_capture": [A(0, 1), A(0, 0), A(10, 10), A(0, 2), A(1, 2), A(11, 11), A(0, 3), A(12, 12), A(1, 0),
                      A(13, 13), A(5, 5), A(1, 1), A(2, 1), A(14, 14), A(6, 6)],
    # snapback in the corner
    "snapback": [A(1, 0), A(0, 0), A(1, 1), A(0, 1), A(1, 2), A(10, 10), A(0, 3), A(11, 11), A(0, 2),
                 A(12, 12), A(2, 2)],
    # two passes end the game; a single opening pass also "ends" it (IsGameEnd quirk, board.cc:1690)
    "opening_pass": [-1, A(3, 3), -1, -1, A(4, 4)],
    "resign": [A(3, 3), -2, A(4, 4)],
    # occupied point and out-of-range actions
    "occupied": [A(3, 3), A(3, 3), A(3, 4), 361, -5, A(3, 5)],
}


def dump_json(path, obj):
    with open(path, "w") as f:
        json.dump(obj, f, separators=(",", ":"))
        f.write("\n")


def mcts_policy(s):
    """Deterministic stand-in for the policy/value net of the tree-search tests: a function of the
    25 planes alone.  pi is a permutation of 1..361 (all weights distinct, so the reference's
    unstable sort has no ties), normalised in float32; V in [0, 1)."""
    import zlib
    rs = np.random.RandomState(zlib.crc32(np.ascontiguousarray(s, np.float32).tobytes()) & 0x7FFFFFFF)
    pi = (rs.permutation(361) + 1).astype(np.float32)
    pi /= np.float32(pi.sum())
    return pi, np.float32(rs.rand())
