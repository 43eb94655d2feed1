"""ctypes binding of libgoenv.so (include/goenv.h) for Python callers.

PyTorch is used only as the owner of device memory and streams: every array
handed to the library is a torch-ROCm tensor's ``data_ptr()``.  There is no CPU
fallback: if the HIP library is missing the import of :class:`GoEnv` raises.
"""
import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libgoenv.so")

NUM_POINTS = 361
NUM_PLANES = 25
FEAT_FLOATS = NUM_PLANES * NUM_POINTS
GEN_PACE_CANDIDATES = 9       # GOENV_GEN_PACE_CANDIDATES, include/goenv.h
NUM_STATS = 16
REC_BYTES = 1632
PASS = -1
RESIGN = -2

F_RANDOM_POLICY = 1
F_AUTO_RESTART = 2
F_SKIP_DONE = 4
F_RESET_ILLEGAL = 8
F_RESTART_AFTER_END = 16
F_COLLECT = 32

# scalar slots of the canonical view
S_PLY, S_NEXT, S_BCAP, S_WCAP, S_LAST1, S_LAST2, S_LAST3, S_LAST4, S_KO_AGE, S_KO, S_KO_COLOR, S_DONE, S_GAME = range(13)
S_SEQ = 14
# stats slots
(ST_GAMES, ST_BLACK_WINS, ST_WHITE_WINS, ST_DRAWS, ST_PLIES, ST_BCAP, ST_WCAP, ST_SUM_LEN2, ST_PLY_CAP_HITS,
 ST_MAX_LEN, ST_SUM_SCORE) = range(11)

_lib = None

_SIGS = {
    "goenv_version": (C.c_int, []),
    "goenv_last_error": (C.c_char_p, []),
    "goenv_create": (C.c_int, [C.c_int, C.c_int, C.c_uint64, C.c_int64, C.c_float, C.c_int, C.POINTER(C.c_void_p)]),
    "goenv_destroy": (C.c_int, [C.c_void_p]),
    "goenv_num_boards": (C.c_int, [C.c_void_p]),
    "goenv_set_waves_per_board": (C.c_int, [C.c_void_p, C.c_int]),
    "goenv_get_fuse_max": (C.c_int, [C.c_void_p]),
    "goenv_set_constant_planes_preset": (C.c_int, [C.c_void_p, C.c_int]),
    "goenv_state_ptr": (C.c_void_p, [C.c_void_p]),
    "goenv_reset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "goenv_get_state": (C.c_int, [C.c_void_p] + [C.c_void_p] * 5 + [C.c_void_p]),
    "goenv_set_state": (C.c_int, [C.c_void_p] + [C.c_void_p] * 3 + [C.c_void_p]),
    "goenv_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "goenv_features": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "goenv_features_static": (C.c_int, [C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_void_p]),
    "goenv_ply": (C.c_int, [C.c_void_p, C.c_uint32] + [C.c_void_p] * 6 + [C.c_void_p]),
    "goenv_ply_repeat": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32] + [C.c_void_p] * 6 + [C.c_void_p]),
    "goenv_ply_repeat_parts": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_uint32] + [C.c_void_p] * 6 + [C.c_void_p]),
    "goenv_ply_sel": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint32] + [C.c_void_p] * 8),
    "goenv_ply_policy_sel": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint32] + [C.c_void_p] * 9),
    "goenv_collect_sel": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5),
    "goenv_sample_policy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 4),
    "goenv_get_state_sel": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_void_p]),
    "goenv_replay": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                               C.c_void_p]),
    "goenv_stage_acquire": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "goenv_offline_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t] + [C.c_void_p] * 5),
    "goenv_offline_bad_moves": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "goenv_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "goenv_host_free": (C.c_int, [C.c_void_p]),
    "goenv_copy_boards": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "goenv_copy_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t,
                                  C.c_void_p]),
    "goenv_pointer_is_device": (C.c_int, [C.c_void_p]),
    "goenv_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "goenv_memset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]),
    "goenv_sync": (C.c_int, [C.c_void_p, C.c_void_p]),
    "goenv_device_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "goenv_device_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "goenv_playout": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "goenv_generate": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "goenv_generate_pace": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "goenv_score": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "goenv_fast_score": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "goenv_topk_policy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "goenv_tactics": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "goenv_finish": (C.c_int, [C.c_void_p, C.c_void_p]),
    "goenv_stats_ptr": (C.c_void_p, [C.c_void_p]),
    "goenv_stats_host": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]),
    "goenv_stats_clear": (C.c_int, [C.c_void_p, C.c_void_p]),
    "goenv_stats_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "goenv_timer_begin": (C.c_int, [C.c_void_p, C.c_void_p]),
    "goenv_timer_end": (C.c_int, [C.c_void_p, C.c_void_p]),
    "goenv_timer_spin": (C.c_int, [C.c_void_p]),
    "goenv_timer_ms": (C.c_float, [C.c_void_p]),
    "goenv_aug_action": (C.c_int, [C.c_int, C.c_int]),
    "goenv_aug_action_inv": (C.c_int, [C.c_int, C.c_int]),
}


def exported_symbols():
    """Names include/goenv.h declares; tests check the .so exports each of them."""
    return sorted(_SIGS)


def load_library(path=LIB_PATH):
    """Load libgoenv.so and attach signatures.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise RuntimeError(
            "libgoenv.so is missing (%s): build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
            "there is no CPU fallback for the Go environment" % path)
    lib = C.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class GoEnvError(RuntimeError):
    pass


class SampleOut(C.Structure):
    """goenv_sample_out: device arrays [count] (or NULL) for the per-sample sequencing record."""
    _fields_ = [(k, C.c_void_p) for k in ("id", "seq", "game_counter", "last_terminal", "move_idx", "winner",
                                          "aug_code", "game_record_idx")] + [("mode", C.c_uint32)]
    DTYPES = dict(id=torch.int32, seq=torch.int32, game_counter=torch.int32, last_terminal=torch.int8,
                  move_idx=torch.int32, winner=torch.int32, aug_code=torch.int32, game_record_idx=torch.int32)

    @classmethod
    def from_tensors(cls, tensors, mode=0, count=None, device=None):
        o = cls()
        o.mode = mode
        for k, t in tensors.items():
            if t.dtype != cls.DTYPES[k] or not t.is_contiguous():
                raise GoEnvError("sample field %s must be a contiguous %s tensor" % (k, cls.DTYPES[k]))
            if count is not None and t.numel() < count:
                raise GoEnvError("sample field %s holds %d rows, the call writes %d" % (k, t.numel(), count))
            if device is not None and t.device != device:
                raise GoEnvError("sample field %s is on %s, the boards are on %s" % (k, t.device, device))
            setattr(o, k, t.data_ptr())
        return o


class TacticsOut(C.Structure):
    """goenv_tactics_out: device arrays (or NULL) of the heuristic move filters."""
    _fields_ = [(k, C.c_void_p) for k in ("self_atari", "gives_ko", "cand", "libs", "ladder")]
    DTYPES = dict(self_atari=torch.int16, gives_ko=torch.uint8, cand=torch.uint8, libs=torch.int16, ladder=torch.int16)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class GoEnv:
    """n_boards concurrent 19x19 games resident in HBM on one GPU.

    Mirrors what a `GoState` (go/go_state.h:31-110) offers per game -- forward,
    CheckMove via the legal mask, Reset, ApplyHandicap, extractor(code).Extract --
    for the whole batch at once.
    """

    def __init__(self, n_boards, device=0, seed=1, board_id_base=0, komi=7.5, max_ply=722):
        self.lib = load_library()
        self.n = int(n_boards)
        self.device = torch.device("cuda", device)
        self._h = C.c_void_p()
        rc = self.lib.goenv_create(device, self.n, seed, board_id_base, komi, max_ply, C.byref(self._h))
        self._chk(rc)

    # -- plumbing ---------------------------------------------------------
    def _chk(self, rc):
        if rc != 0:
            raise GoEnvError(self.lib.goenv_last_error().decode())

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check(self, t, dtype, shape, name):
        if t is None:
            return
        if t.device != self.device or t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_contiguous():
            raise GoEnvError("%s must be a contiguous %s tensor of shape %s on %s (got %s %s on %s)" % (
                name, dtype, tuple(shape), self.device, t.dtype, tuple(t.shape), t.device))

    def _check_rows(self, count, aug_codes=None, feat_out=None, legal_out=None, actions_out=None, status_out=None,
                    actions=None):
        """Batch arrays of a selection call: one row per selected board."""
        self._check(actions, torch.int64, (count,), "actions")
        self._check(aug_codes, torch.int32, (count,), "aug_codes")
        self._check(feat_out, torch.float32, (count, NUM_PLANES, 19, 19), "feat_out")
        self._check(legal_out, torch.uint8, (count, NUM_POINTS), "legal_out")
        self._check(actions_out, torch.int32, (count,), "actions_out")
        self._check(status_out, torch.int32, (count,), "status_out")

    def close(self):
        if self._h:
            self.lib.goenv_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def fuse_max(self):
        return int(self.lib.goenv_get_fuse_max(self._h))

    def set_constant_planes_preset(self, on=True):
        """The caller keeps planes 12, 13, 16..24 of its (zero-filled) tensors: no kernel writes them."""
        self._chk(self.lib.goenv_set_constant_planes_preset(self._h, 1 if on else 0))

    def set_waves_per_board(self, nw):
        self._chk(self.lib.goenv_set_waves_per_board(self._h, int(nw)))

    def new(self, dtype, *shape):
        return torch.zeros((self.n,) + tuple(shape), dtype=dtype, device=self.device)

    # -- API --------------------------------------------------------------
    def reset(self, idx=None, handicap=None):
        n = self.n
        if idx is not None:
            self._check(idx, torch.int32, (idx.numel(),), "idx")
            n = idx.numel()
        if handicap is not None:
            self._check(handicap, torch.int32, (n,), "handicap")
        self._chk(self.lib.goenv_reset(self._h, _ptr(idx), n, _ptr(handicap), self._stream()))

    def get_state(self):
        """Canonical view as a dict of device tensors."""
        out = dict(color=self.new(torch.uint8, NUM_POINTS), liberties=self.new(torch.int16, NUM_POINTS),
                   stones=self.new(torch.int16, NUM_POINTS), last_placed=self.new(torch.uint16, NUM_POINTS),
                   scalars=self.new(torch.int32, 16))
        self._chk(self.lib.goenv_get_state(self._h, _ptr(out["color"]), _ptr(out["liberties"]), _ptr(out["stones"]),
                                           _ptr(out["last_placed"]), _ptr(out["scalars"]), self._stream()))
        return out

    def set_state(self, color, last_placed, scalars):
        self._check(color, torch.uint8, (self.n, NUM_POINTS), "color")
        self._check(last_placed, torch.uint16, (self.n, NUM_POINTS), "last_placed")
        self._check(scalars, torch.int32, (self.n, 16), "scalars")
        self._chk(self.lib.goenv_set_state(self._h, _ptr(color), _ptr(last_placed), _ptr(scalars), self._stream()))

    def step(self, actions, legal_out=None, status_out=None):
        self._check(actions, torch.int64, (self.n,), "actions")
        self._check(legal_out, torch.uint8, (self.n, NUM_POINTS), "legal_out")
        self._check(status_out, torch.int32, (self.n,), "status_out")
        self._chk(self.lib.goenv_step(self._h, _ptr(actions), _ptr(legal_out), _ptr(status_out), self._stream()))

    def features(self, feat_out=None, aug_codes=None, legal_out=None):
        self._check(feat_out, torch.float32, (self.n, NUM_PLANES, 19, 19), "feat_out")
        self._check(aug_codes, torch.int32, (self.n,), "aug_codes")
        self._check(legal_out, torch.uint8, (self.n, NUM_POINTS), "legal_out")
        self._chk(self.lib.goenv_features(self._h, _ptr(aug_codes), _ptr(feat_out), _ptr(legal_out), self._stream()))

    def ply(self, flags=0, actions=None, aug_codes=None, feat_out=None, legal_out=None, actions_out=None,
            status_out=None):
        self._check(actions, torch.int64, (self.n,), "actions")
        self._check(aug_codes, torch.int32, (self.n,), "aug_codes")
        self._check(feat_out, torch.float32, (self.n, NUM_PLANES, 19, 19), "feat_out")
        self._check(legal_out, torch.uint8, (self.n, NUM_POINTS), "legal_out")
        self._check(actions_out, torch.int32, (self.n,), "actions_out")
        self._check(status_out, torch.int32, (self.n,), "status_out")
        self._chk(self.lib.goenv_ply(self._h, flags, _ptr(actions), _ptr(aug_codes), _ptr(feat_out), _ptr(legal_out),
                                     _ptr(actions_out), _ptr(status_out), self._stream()))

    def ply_repeat(self, n_steps, flags=0, actions=None, aug_codes=None, feat_out=None, legal_out=None,
                   actions_out=None, status_out=None):
        self._check(actions, torch.int64, (self.n,), "actions")
        self._check(aug_codes, torch.int32, (self.n,), "aug_codes")
        self._check(feat_out, torch.float32, (self.n, NUM_PLANES, 19, 19), "feat_out")
        self._check(legal_out, torch.uint8, (self.n, NUM_POINTS), "legal_out")
        self._check(actions_out, torch.int32, (self.n,), "actions_out")
        self._check(status_out, torch.int32, (self.n,), "status_out")
        self._chk(self.lib.goenv_ply_repeat(self._h, int(n_steps), flags, _ptr(actions), _ptr(aug_codes),
                                            _ptr(feat_out), _ptr(legal_out), _ptr(actions_out), _ptr(status_out),
                                            self._stream()))

    def ply_repeat_parts(self, n_steps, n_parts, flags=0, actions=None, aug_codes=None, feat_out=None, legal_out=None,
                         actions_out=None, status_out=None):
        self._check(actions, torch.int64, (self.n,), "actions")
        self._check(aug_codes, torch.int32, (self.n,), "aug_codes")
        self._check(feat_out, torch.float32, (self.n, NUM_PLANES, 19, 19), "feat_out")
        self._check(legal_out, torch.uint8, (self.n, NUM_POINTS), "legal_out")
        self._check(actions_out, torch.int32, (self.n,), "actions_out")
        self._check(status_out, torch.int32, (self.n,), "status_out")
        self._chk(self.lib.goenv_ply_repeat_parts(self._h, int(n_steps), int(n_parts), flags, _ptr(actions),
                                                  _ptr(aug_codes), _ptr(feat_out), _ptr(legal_out), _ptr(actions_out),
                                                  _ptr(status_out), self._stream()))

    def _sel(self, idx, first, count):
        if idx is not None:
            self._check(idx, torch.int32, (idx.numel(),), "idx")
            return _ptr(idx), 0, idx.numel()
        return None, int(first), int(self.n - first if count is None else count)

    def get_state_sel(self, idx=None, first=0, count=None):
        ip, first, count = self._sel(idx, first, count)
        dev = self.device
        out = dict(color=torch.zeros((count, NUM_POINTS), dtype=torch.uint8, device=dev),
                   liberties=torch.zeros((count, NUM_POINTS), dtype=torch.int16, device=dev),
                   stones=torch.zeros((count, NUM_POINTS), dtype=torch.int16, device=dev),
                   last_placed=torch.zeros((count, NUM_POINTS), dtype=torch.uint16, device=dev),
                   scalars=torch.zeros((count, 16), dtype=torch.int32, device=dev))
        self._chk(self.lib.goenv_get_state_sel(self._h, ip, first, count, _ptr(out["color"]), _ptr(out["liberties"]),
                                               _ptr(out["stones"]), _ptr(out["last_placed"]), _ptr(out["scalars"]),
                                               self._stream()))
        return out

    def ply_sel(self, idx=None, first=0, count=None, flags=0, actions=None, aug_codes=None, feat_out=None,
                legal_out=None, meta=None, actions_out=None, status_out=None):
        """goenv_ply_sel: batch arrays are indexed by the selection row."""
        ip, first, count = self._sel(idx, first, count)
        self._check_rows(count, aug_codes, feat_out, legal_out, actions_out, status_out, actions)
        m = SampleOut.from_tensors(meta, count=count, device=self.device) if meta is not None else None
        self._chk(self.lib.goenv_ply_sel(self._h, ip, first, count, flags, _ptr(actions), _ptr(aug_codes),
                                         _ptr(feat_out), _ptr(legal_out), C.byref(m) if m is not None else None,
                                         _ptr(actions_out), _ptr(status_out), self._stream()))

    def ply_policy_sel(self, pi, idx=None, first=0, count=None, flags=0, pi_aug_codes=None, aug_codes=None,
                       feat_out=None, legal_out=None, meta=None, actions_out=None, status_out=None):
        """goenv_ply_policy_sel: sample from pi, apply, write the next sample -- one launch."""
        ip, first, count = self._sel(idx, first, count)
        self._check(pi, torch.float32, (count, NUM_POINTS), "pi")
        self._check(pi_aug_codes, torch.int32, (count,), "pi_aug_codes")
        self._check_rows(count, aug_codes, feat_out, legal_out, actions_out, status_out)
        m = SampleOut.from_tensors(meta, count=count, device=self.device) if meta is not None else None
        self._chk(self.lib.goenv_ply_policy_sel(self._h, ip, first, count, flags, _ptr(pi), _ptr(pi_aug_codes),
                                                _ptr(aug_codes), _ptr(feat_out), _ptr(legal_out),
                                                C.byref(m) if m is not None else None, _ptr(actions_out),
                                                _ptr(status_out), self._stream()))

    def collect_sel(self, idx=None, first=0, count=None, aug_codes=None, feat_out=None, legal_out=None, meta=None):
        ip, first, count = self._sel(idx, first, count)
        self._check_rows(count, aug_codes, feat_out, legal_out)
        m = SampleOut.from_tensors(meta, count=count, device=self.device) if meta is not None else None
        self._chk(self.lib.goenv_collect_sel(self._h, ip, first, count, _ptr(aug_codes), _ptr(feat_out),
                                             _ptr(legal_out), C.byref(m) if m is not None else None, self._stream()))

    def sample_policy(self, pi, actions_out, idx=None, first=0, count=None, aug_codes=None):
        ip, first, count = self._sel(idx, first, count)
        self._check(pi, torch.float32, (count, NUM_POINTS), "pi")
        self._check(actions_out, torch.int64, (count,), "actions_out")
        self._check(aug_codes, torch.int32, (count,), "aug_codes")
        self._chk(self.lib.goenv_sample_policy(self._h, ip, first, count, _ptr(pi), _ptr(aug_codes),
                                               _ptr(actions_out), self._stream()))

    def replay(self, moves, nmoves, napplied_out=None, idx=None, first=0, count=None):
        """goenv_replay: moves int16 [count][stride], nmoves int32 [count]."""
        ip, first, count = self._sel(idx, first, count)
        self._check(moves, torch.int16, (count, moves.shape[1]), "moves")
        self._check(nmoves, torch.int32, (count,), "nmoves")
        self._check(napplied_out, torch.int32, (count,), "napplied_out")
        self._chk(self.lib.goenv_replay(self._h, ip, first, count, _ptr(moves), moves.shape[1], _ptr(nmoves),
                                        _ptr(napplied_out), self._stream()))

    def copy_boards(self, src_idx, dst_idx):
        n = src_idx.numel()
        self._check(src_idx, torch.int32, (n,), "src_idx")
        self._check(dst_idx, torch.int32, (n,), "dst_idx")
        self._chk(self.lib.goenv_copy_boards(self._h, _ptr(src_idx), _ptr(dst_idx), n, self._stream()))

    def copy_rows(self, src, dst, src_rows=None, dst_rows=None, n=None):
        """goenv_copy_rows between two contiguous tensors whose rows ([rows, ...]) have the same byte size."""
        rb = src[0].numel() * src.element_size()
        if not (src.is_contiguous() and dst.is_contiguous()) or dst[0].numel() * dst.element_size() != rb:
            raise GoEnvError("copy_rows: contiguous tensors with rows of equal byte size are required")
        for t, rows in ((src_rows, src.shape[0]), (dst_rows, dst.shape[0])):
            if t is not None:
                self._check(t, torch.int32, (t.numel(),), "row list")
                if t.numel() and (int(t.max()) >= rows):
                    raise GoEnvError("copy_rows: a row index exceeds the table")
        if n is None:
            n = src_rows.numel() if src_rows is not None else (dst_rows.numel() if dst_rows is not None else
                                                               min(src.shape[0], dst.shape[0]))
        self._chk(self.lib.goenv_copy_rows(self._h, _ptr(src), _ptr(src_rows), _ptr(dst), _ptr(dst_rows), int(n), rb,
                                           self._stream()))

    def stage_acquire(self, nbytes):
        """goenv_stage_acquire: the next pinned host block of the staging ring, as a writable numpy uint8 view."""
        import numpy as np
        p = C.c_void_p()
        self._chk(self.lib.goenv_stage_acquire(self._h, int(nbytes), C.byref(p)))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(int(nbytes),)), p.value

    def offline_batch(self, block_ptr, count, k_future, moves_offset, feat_out=None, legal_out=None, labels=None,
                      offline_a_out=None):
        """goenv_offline_batch: block_ptr = address of the packed rows (a stage_acquire block, or device memory)."""
        self._check_rows(count, feat_out=feat_out, legal_out=legal_out)
        self._check(offline_a_out, torch.int64, (count, k_future), "offline_a_out")
        m = SampleOut.from_tensors(labels, count=count, device=self.device) if labels is not None else None
        self._chk(self.lib.goenv_offline_batch(self._h, C.c_void_p(block_ptr), int(count), int(k_future), int(moves_offset),
                                               _ptr(feat_out), _ptr(legal_out), C.byref(m) if m is not None else None,
                                               _ptr(offline_a_out), self._stream()))

    def offline_bad_moves(self):
        n = C.c_int32(0)
        self._chk(self.lib.goenv_offline_bad_moves(self._h, C.byref(n), self._stream()))
        return n.value

    def playout(self, moves_out=None, nmoves_out=None):
        max_record = 0
        if moves_out is not None:
            max_record = moves_out.shape[1]
            self._check(moves_out, torch.int16, (self.n, max_record), "moves_out")
        self._check(nmoves_out, torch.int32, (self.n,), "nmoves_out")
        self._chk(self.lib.goenv_playout(self._h, _ptr(moves_out), max_record, _ptr(nmoves_out), self._stream()))

    def generate(self, n_plies, feat_ring=None, legal_ring=None, actions_ring=None, aug_codes=None):
        """goenv_generate: n_plies plies in one persistent launch into [n_plies][n]... ring buffers."""
        self._check(feat_ring, torch.float32, (n_plies, self.n, NUM_PLANES, 19, 19), "feat_ring")
        self._check(legal_ring, torch.uint8, (n_plies, self.n, NUM_POINTS), "legal_ring")
        self._check(actions_ring, torch.int32, (n_plies, self.n), "actions_ring")
        self._check(aug_codes, torch.int32, (self.n,), "aug_codes")
        self._chk(self.lib.goenv_generate(self._h, int(n_plies), _ptr(aug_codes), _ptr(feat_ring), _ptr(legal_ring),
                                          _ptr(actions_ring), self._stream()))

    def generate_pace(self):
        """goenv_generate_pace: (ns per board-ply that the generator's rate limiter has settled on, [us per ply per
        candidate pace])."""
        pace, est = C.c_float(0), (C.c_float * GEN_PACE_CANDIDATES)()
        self._chk(self.lib.goenv_generate_pace(self._h, C.byref(pace), est, None))
        return pace.value, list(est)

    def generate_kernel(self):
        """Name of the kernel that the last generate() of this context launched."""
        pipe = C.c_int(0)
        self._chk(self.lib.goenv_generate_pace(self._h, None, None, C.byref(pipe)))
        return ("k_generate", "k_generate_pipe", "k_generate_pipe_multi")[pipe.value]

    def score(self, score_out=None, fast_out=None):
        self._check(score_out, torch.float32, (self.n,), "score_out")
        self._check(fast_out, torch.float32, (self.n,), "fast_out")
        self._chk(self.lib.goenv_score(self._h, _ptr(score_out), _ptr(fast_out), self._stream()))

    def fast_score(self, rule, out):
        self._check(out, torch.float32, (self.n,), "out")
        self._chk(self.lib.goenv_fast_score(self._h, int(rule), _ptr(out), self._stream()))

    def topk_policy(self, pi, k, actions_out, probs_out=None, aug_codes=None, idx=None, first=0, count=None):
        ip, first, count = self._sel(idx, first, count)
        self._check(pi, torch.float32, (count, NUM_POINTS), "pi")
        self._check(actions_out, torch.int32, (count, k), "actions_out")
        self._check(probs_out, torch.float32, (count, k), "probs_out")
        self._check(aug_codes, torch.int32, (count,), "aug_codes")
        self._chk(self.lib.goenv_topk_policy(self._h, ip, first, count, _ptr(pi), _ptr(aug_codes), int(k),
                                             _ptr(actions_out), _ptr(probs_out), self._stream()))

    def tactics(self, self_atari_thres=1, idx=None, first=0, count=None, **outs):
        """goenv_tactics: outs = self_atari / gives_ko / cand / libs / ladder tensors ([count, 361],
        libs [count, 361, 2]); returns the dict passed in."""
        if idx is not None:
            count = idx.numel()
            self._check(idx, torch.int32, (count,), "idx")
        elif count is None:
            count = self.n - first
        o = TacticsOut()
        for k, t in outs.items():
            shape = (count, NUM_POINTS, 2) if k == "libs" else (count, NUM_POINTS)
            self._check(t, TacticsOut.DTYPES[k], shape, k)
            setattr(o, k, t.data_ptr())
        self._chk(self.lib.goenv_tactics(self._h, _ptr(idx), first, count, int(self_atari_thres), C.byref(o),
                                         self._stream()))
        return outs

    def finish(self):
        self._chk(self.lib.goenv_finish(self._h, self._stream()))

    def stats_tensor(self):
        """The device int64[16] stats vector as a torch view (what the multi-GPU job all-reduces)."""
        ptr = self.lib.goenv_stats_ptr(self._h)
        # wrap foreign memory without copying: go through __cuda_array_interface__
        holder = _CudaArray(ptr, (NUM_STATS,), "<i8", self)
        return torch.as_tensor(holder, device=self.device)

    def stats(self):
        buf = (C.c_int64 * NUM_STATS)()
        self._chk(self.lib.goenv_stats_host(self._h, buf, self._stream()))
        return list(buf)

    def stats_allreduce(self, rccl_comm):
        """goenv_stats_allreduce with a raw ncclComm_t (an int / c_void_p)."""
        self._chk(self.lib.goenv_stats_allreduce(self._h, C.c_void_p(rccl_comm), self._stream()))

    def stats_clear(self):
        self._chk(self.lib.goenv_stats_clear(self._h, self._stream()))

    def timer_begin(self):
        self._chk(self.lib.goenv_timer_begin(self._h, self._stream()))

    def timer_end(self):
        self._chk(self.lib.goenv_timer_end(self._h, self._stream()))

    def timer_spin(self):
        self._chk(self.lib.goenv_timer_spin(self._h))

    def timer_ms(self):
        return float(self.lib.goenv_timer_ms(self._h))


class _CudaArray:
    def __init__(self, ptr, shape, typestr, owner):
        self.owner = owner
        self.__cuda_array_interface__ = dict(shape=shape, typestr=typestr, data=(int(ptr), False), version=2)


STATIC_MAX_BOARDS = 4096


def features_static(color, last_placed, scalars, feat_out=None, legal_out=None, aug_codes=None):
    """goenv_features_static: positions -> planes + legal mask with no context and no allocation
    (the caller owns every tensor; n <= STATIC_MAX_BOARDS)."""
    lib = load_library()
    dev = color.device
    n = color.shape[0]

    def check(t, dtype, shape, name):
        if t is None:
            return
        if t.device != dev or t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_contiguous():
            raise GoEnvError("%s must be a contiguous %s tensor of shape %s on %s" % (name, dtype, tuple(shape), dev))
    if dev.type != "cuda":
        raise GoEnvError("features_static needs device tensors")
    check(color, torch.uint8, (n, NUM_POINTS), "color")
    check(last_placed, torch.uint16, (n, NUM_POINTS), "last_placed")
    check(scalars, torch.int32, (n, 16), "scalars")
    check(aug_codes, torch.int32, (n,), "aug_codes")
    check(feat_out, torch.float32, (n, NUM_PLANES, 19, 19), "feat_out")
    check(legal_out, torch.uint8, (n, NUM_POINTS), "legal_out")
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    rc = lib.goenv_features_static(index, n, _ptr(color), _ptr(last_placed), _ptr(scalars), _ptr(aug_codes),
                                   _ptr(feat_out), _ptr(legal_out),
                                   C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise GoEnvError(lib.goenv_last_error().decode())


def aug_action(code, action):
    return load_library().goenv_aug_action(code, action)


def aug_action_inv(code, action):
    return load_library().goenv_aug_action_inv(code, action)
