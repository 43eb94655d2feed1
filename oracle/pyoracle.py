"""TEST INFRASTRUCTURE -- ctypes front-end to the two CPU checkers.

* ``OracleBoard``  -> oracle/libgooracle.so (our plain-C restatement, go_oracle.c)
* ``RefBoard``     -> oracle/_ref/libgoref.so (the unmodified reference compiled by
  oracle/Makefile from /root/reference; present only where it was built)

Both expose the same methods so tests can run one body against either.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libgooracle.so")
REF_SO = os.path.join(HERE, "_ref", "libgoref.so")

NP_ = 361
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C")
_i16p = np.ctypeslib.ndpointer(np.int16, flags="C")
_u16p = np.ctypeslib.ndpointer(np.uint16, flags="C")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C")


def _load(path, prefix):
    lib = C.CDLL(path)

    def f(name, res, *args):
        fn = getattr(lib, prefix + name)
        fn.restype = res
        fn.argtypes = list(args)
        return fn

    api = {}
    api["sizeof_board"] = f("sizeof_board", C.c_int)
    api["clear"] = f("clear", None, C.c_void_p)
    api["forward"] = f("forward", C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int))
    api["check_move"] = f("check_move", C.c_int, C.c_void_p, C.c_int)
    api["place_handicap"] = f("place_handicap", C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int)
    api["apply_handicap"] = f("apply_handicap", None, C.c_void_p, C.c_int)
    api["is_game_end"] = f("is_game_end", C.c_int, C.c_void_p)
    api["canonical"] = f("canonical", None, C.c_void_p, _u8p, _i16p, _i16p, _u16p, _i32p)
    api["legal"] = f("legal", C.c_int, C.c_void_p, _u8p)
    api["candidates"] = f("candidates", C.c_int, C.c_void_p, _u8p)
    api["extract"] = f("extract", None, C.c_void_p, C.c_int, _f32p)
    api["tt_score"] = f("tt_score", C.c_float, C.c_void_p)
    api["fast_score"] = f("fast_score", C.c_float, C.c_void_p, C.c_int)
    api["tactics"] = f("tactics", None, C.c_void_p, C.c_int, _i16p, _u8p, _u8p, _i16p, _i16p)
    api["policy_move"] = f("policy_move", C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64)
    api["bench_playouts"] = f("bench_playouts", C.c_int64, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int,
                              C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double))
    return lib, api


class _Board:
    _lib = None
    _api = None
    _path = None
    _prefix = None

    @classmethod
    def available(cls):
        return os.path.exists(cls._path)

    @classmethod
    def api(cls):
        if cls._api is None:
            if not cls.available():
                raise RuntimeError("%s not built (run `make -C oracle`)" % cls._path)
            cls._lib, cls._api = _load(cls._path, cls._prefix)
            cls._post_load()
        return cls._api

    @classmethod
    def _post_load(cls):
        pass

    def __init__(self):
        a = self.api()
        self._buf = C.create_string_buffer(a["sizeof_board"]() + 64)
        self.ptr = C.cast(self._buf, C.c_void_p)
        a["clear"](self.ptr)

    def clone(self):
        other = type(self).__new__(type(self))
        other._buf = C.create_string_buffer(self._buf.raw, len(self._buf))
        other.ptr = C.cast(other._buf, C.c_void_p)
        return other

    def clear(self):
        self.api()["clear"](self.ptr)

    def forward(self, action):
        """GoState::forward: returns (legal, game_end)."""
        end = C.c_int(0)
        ok = self.api()["forward"](self.ptr, int(action), C.byref(end))
        return bool(ok), bool(end.value)

    def check_move(self, action):
        return bool(self.api()["check_move"](self.ptr, int(action)))

    def place_handicap(self, x, y, player=1):
        return bool(self.api()["place_handicap"](self.ptr, x, y, player))

    def apply_handicap(self, n):
        self.api()["apply_handicap"](self.ptr, int(n))

    def is_game_end(self):
        return bool(self.api()["is_game_end"](self.ptr))

    def canonical(self):
        color = np.zeros(NP_, np.uint8)
        libs = np.zeros(NP_, np.int16)
        stones = np.zeros(NP_, np.int16)
        lastp = np.zeros(NP_, np.uint16)
        sc = np.zeros(16, np.int32)
        self.api()["canonical"](self.ptr, color, libs, stones, lastp, sc)
        return dict(color=color, liberties=libs, stones=stones, last_placed=lastp, scalars=sc)

    def legal(self):
        m = np.zeros(NP_, np.uint8)
        self.api()["legal"](self.ptr, m)
        return m

    def candidates(self):
        m = np.zeros(NP_, np.uint8)
        self.api()["candidates"](self.ptr, m)
        return m

    def extract(self, aug=0):
        out = np.zeros(25 * NP_, np.float32)
        self.api()["extract"](self.ptr, int(aug), out)
        return out

    def tt_score(self):
        return float(self.api()["tt_score"](self.ptr))

    def fast_score(self, rule=0):
        return float(self.api()["fast_score"](self.ptr, rule))

    def tactics(self, thres=1):
        """Heuristic move filters for the side to move: dict of per-point arrays."""
        sa = np.zeros(NP_, np.int16)
        ko = np.zeros(NP_, np.uint8)
        cand = np.zeros(NP_, np.uint8)
        libs = np.zeros((NP_, 2), np.int16)
        ladder = np.zeros(NP_, np.int16)
        self.api()["tactics"](self.ptr, int(thres), sa, ko, cand, libs.reshape(-1), ladder)
        return dict(self_atari=sa, gives_ko=ko, cand=cand, libs=libs, ladder=ladder)

    def policy_move(self, seed, board_id, game=0):
        return int(self.api()["policy_move"](self.ptr, seed, board_id, game))

    @classmethod
    def bench_playouts(cls, n_boards, threads, seed, max_ply, with_features, games_per_board=1, budget_sec=0.0):
        sec = C.c_double(0)
        cs = C.c_double(0)
        plies = cls.api()["bench_playouts"](n_boards, threads, seed, max_ply, int(with_features),
                                            games_per_board, float(budget_sec), C.byref(sec), C.byref(cs))
        return int(plies), float(sec.value)


class OracleBoard(_Board):
    _path = ORACLE_SO
    _prefix = "goor_"

    @classmethod
    def _post_load(cls):
        lib = cls._lib
        lib.goor_set_state.restype = None
        lib.goor_set_state.argtypes = [C.c_void_p, _u8p, _u16p, _i32p]
        lib.goor_aug_action.restype = C.c_int
        lib.goor_aug_action.argtypes = [C.c_int, C.c_int]
        lib.goor_aug_action_inv.restype = C.c_int
        lib.goor_aug_action_inv.argtypes = [C.c_int, C.c_int]
        lib.goor_sample_policy.restype = C.c_int
        lib.goor_sample_policy.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, _f32p, C.c_int]
        lib.goor_playout.restype = C.c_int
        lib.goor_playout.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, _i16p, C.c_int]

    def set_state(self, color, last_placed, scalars):
        self.api()
        self._lib.goor_set_state(self.ptr, np.ascontiguousarray(color, np.uint8),
                                 np.ascontiguousarray(last_placed, np.uint16),
                                 np.ascontiguousarray(scalars, np.int32))

    @classmethod
    def aug_action(cls, code, action):
        cls.api()
        return cls._lib.goor_aug_action(code, action)

    @classmethod
    def aug_action_inv(cls, code, action):
        cls.api()
        return cls._lib.goor_aug_action_inv(code, action)

    def sample_policy(self, seed, board_id, game, pi, aug=0):
        return self._lib.goor_sample_policy(self.ptr, seed, board_id, game, np.ascontiguousarray(pi, np.float32), aug)

    def playout(self, seed, board_id, game=0, max_ply=722):
        moves = np.zeros(2048, np.int16)
        n = self._lib.goor_playout(self.ptr, seed, board_id, game, max_ply, moves, 2048)
        return moves[:n].copy()


class RefBoard(_Board):
    _path = REF_SO
    _prefix = "goref_"

    @classmethod
    def _post_load(cls):
        lib = cls._lib
        lib.goref_coord2action.restype = C.c_int
        lib.goref_coord2action.argtypes = [C.c_int, C.c_int]
        lib.goref_action2coord.restype = C.c_int
        lib.goref_action2coord.argtypes = [C.c_int, C.c_int]
        lib.goref_num_groups.restype = C.c_int
        lib.goref_num_groups.argtypes = [C.c_void_p]
        lib.goref_sgf_load.restype = C.c_int
        lib.goref_sgf_load.argtypes = [C.c_char_p, _i16p, _u8p, C.c_int, C.POINTER(C.c_int),
                                       C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]

    @classmethod
    def aug_action(cls, code, action):
        cls.api()
        return cls._lib.goref_coord2action(code, action)

    @classmethod
    def aug_action_inv(cls, code, action):
        cls.api()
        return cls._lib.goref_action2coord(code, action)

    def num_groups(self):
        return self._lib.goref_num_groups(self.ptr)

    @classmethod
    def sgf_load(cls, path):
        cls.api()
        moves = np.zeros(4096, np.int16)
        players = np.zeros(4096, np.uint8)
        nm, w, h, bs = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        n = cls._lib.goref_sgf_load(path.encode(), moves, players, 4096, C.byref(nm), C.byref(w),
                                    C.byref(h), C.byref(bs))
        if n < 0:
            raise RuntimeError("reference Sgf::Load failed for " + path)
        return dict(moves=moves[:n].copy(), players=players[:n].copy(), num_moves=nm.value,
                    winner=w.value, handi=h.value, board_size=bs.value)


class RefBatch:
    """n reference boards (the unmodified board.cc `Board`) in one array, stepped and compared natively
    (goref_batch_check, oracle/ref_harness.cc) so that a GPU test can replay EVERY board of a launch
    through the reference on every ply.  `board(b)` is a RefBoard view of one of them."""
    WHAT = ("", "legality", "colours", "liberties", "group sizes", "last_placed", "scalars", "legal mask", "planes")

    def __init__(self, n, threads=None):
        api = RefBoard.api()
        lib = RefBoard._lib
        self.lib = lib
        self.n = n
        self.threads = threads or min(16, os.cpu_count() or 1)
        self.stride = (api["sizeof_board"]() + 63) // 64 * 64
        self._buf = C.create_string_buffer(self.stride * n + 64)
        self.base = (C.addressof(self._buf) + 63) // 64 * 64
        for b in range(n):
            api["clear"](C.c_void_p(self.base + b * self.stride))
        P = C.c_void_p
        lib.goref_batch_check.restype = C.c_int
        lib.goref_batch_check.argtypes = [P, C.c_int, C.c_int] + [P] * 10 + [C.c_int, P]
        lib.goref_batch_policy.restype = None
        lib.goref_batch_policy.argtypes = [P, C.c_int, C.c_int, C.c_uint64, C.c_uint64, P, P, C.c_int]
        lib.goref_batch_ended.restype = C.c_int
        lib.goref_batch_ended.argtypes = [P, C.c_int, C.c_int, C.c_int, P]

    def board(self, b):
        o = RefBoard.__new__(RefBoard)
        o._buf = self._buf                     # keeps the array alive
        o.ptr = C.c_void_p(self.base + b * self.stride)
        return o

    @staticmethod
    def _p(a, dtype):
        if a is None:
            return None, None
        a = np.ascontiguousarray(a, dtype)
        return a, a.ctypes.data_as(C.c_void_p)

    def check(self, actions=None, want_legal=None, state=None, legal=None, planes=None, aug=None, where=""):
        """Play actions (int32 [n], -3 = none) on the reference boards, then compare whatever is given:
        state = dict of the canonical arrays as goenv.get_state() returns them (numpy), legal u8 [n, 361],
        planes f32 [n, 9025] under aug int32 [n].  Raises AssertionError naming the first difference."""
        keep = []

        def p(a, dtype, rows=None):
            a, ptr = self._p(a, dtype)
            if a is not None:
                assert a.shape[0] == self.n, "batch array of %d rows for %d boards" % (a.shape[0], self.n)
                keep.append(a)
            return ptr
        st = state or {}
        sc = st.get("scalars")
        if sc is not None:
            assert np.asarray(sc).shape == (self.n, 16)
        bad3 = np.zeros(3, np.int32)
        nbad = self.lib.goref_batch_check(
            C.c_void_p(self.base), self.stride, self.n, p(actions, np.int32), p(want_legal, np.int32),
            p(st.get("color"), np.uint8), p(st.get("liberties"), np.int16), p(st.get("stones"), np.int16),
            p(st.get("last_placed"), np.uint16), p(sc, np.int32), p(legal, np.uint8),
            p(None if planes is None else np.asarray(planes).reshape(self.n, -1), np.float32), p(aug, np.int32),
            self.threads, bad3.ctypes.data_as(C.c_void_p))
        if nbad:
            raise AssertionError("%s: %d of %d boards differ from the reference; first: board %d, %s, index %d" % (
                where, nbad, self.n, bad3[0], self.WHAT[bad3[1]], bad3[2]))

    def policy(self, seed, games=None, id_base=0):
        out = np.zeros(self.n, np.int32)
        g, gp = self._p(games, np.int32)
        self.lib.goref_batch_policy(C.c_void_p(self.base), self.stride, self.n, seed, id_base, gp,
                                    out.ctypes.data_as(C.c_void_p), self.threads)
        return out

    def ended(self, max_ply):
        out = np.zeros(self.n, np.uint8)
        self.lib.goref_batch_ended(C.c_void_p(self.base), self.stride, self.n, max_ply, out.ctypes.data_as(C.c_void_p))
        return out


DIFF_SO = os.path.join(HERE, "_ref", "libgodiff.so")
DIFF_COUNTS = ("plies", "games", "applied", "illegal", "passes", "resigns", "ko_set", "captures", "extracts",
               "scores", "tactics", "handicap_games", "dropped_maxgroup", "game_ends", "checkmove_sweeps",
               "max_groups")


def diff_fuzz(seed, target_plies, threads):
    """Native reference-vs-restatement loop (oracle/diff_fuzz.cc).  Returns (error or None, counts)."""
    lib = C.CDLL(DIFF_SO)
    n = lib.godiff_num_counts()
    assert n == len(DIFF_COUNTS)
    lib.godiff_run.restype = C.c_int
    lib.godiff_run.argtypes = [C.c_uint64, C.c_int64, C.c_int, C.POINTER(C.c_int64), C.c_char_p, C.c_int]
    cnt = (C.c_int64 * n)()
    err = C.create_string_buffer(512)
    rc = lib.godiff_run(seed, target_plies, threads, cnt, err, 512)
    return (err.value.decode() if rc else None), dict(zip(DIFF_COUNTS, list(cnt)))


def fnv_planes(h, planes):
    """64-bit FNV-1a-style hash over the float bit patterns:
    h = (h ^ u32) * 0x100000001b3, seed 0xcbf29ce484222325."""
    u = np.ascontiguousarray(planes, np.float32).view(np.uint32)
    prime = 0x100000001B3
    mask = (1 << 64) - 1
    for v in u.tolist():
        h = ((h ^ v) * prime) & mask
    return h


FNV_SEED = 0xCBF29CE484222325
