"""Batch-level golden vectors (cfg 3: offline SGF replay -> feature planes + labels).

Two steps, both run HERE (the container that mounts /root/reference):

1. corpus: a small SGF corpus of OUR OWN games (oracle playouts written as SGF text, plus a few
   deliberately awkward files: handicap, comments with escaped brackets, a variation, an illegal
   move, a 9x9 game, a too-short game) next to the reference's only Go fixture go/sample1.sgf.
   Written to tests/golden/corpus/ with the list file train.lst.
2. streams: the UNMODIFIED reference `go_game` pybind module (oracle/_ref/go_game*.so, built by
   `make -C oracle refmod` from the sources under /root/reference) is driven in offline mode over
   that corpus with numpy buffers; for every game id the stream of samples it emits is recorded:
   (seq, game_counter, last_terminal, move_idx, winner, aug_code, game_record_idx, offline_a[k],
   FNV of the 9,025 plane floats).  Batch composition in the reference depends on thread timing,
   per-game streams do not (each game owns its RNGs), so streams are what is pinned.

Output: tests/golden/batch_streams.json (data only).
"""
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from common import GOLDEN, dump_json  # noqa: E402
from oracle.pyoracle import FNV_SEED, OracleBoard, fnv_planes  # noqa: E402

CORPUS = os.path.join(GOLDEN, "corpus")


def sgf_coord(a):
    return "" if a < 0 else chr(ord("a") + a // 19) + chr(ord("a") + a % 19)


def write_sgf(path, moves, handi=0, result="B+3.5", size=19, first="B", extra_head="", decorate=None):
    col = first
    body = []
    for i, a in enumerate(moves):
        node = ";%s[%s]" % (col, sgf_coord(a))
        if decorate and i in decorate:
            node += decorate[i]
        body.append(node)
        col = "W" if col == "B" else "B"
    head = "(;GM[1]FF[4]SZ[%d]KM[7.5]%sRE[%s]PB[oracle black]PW[oracle white]%s\n" % (
        size, ("HA[%d]" % handi) if handi else "", result, extra_head)
    lines = ["".join(body[i:i + 10]) for i in range(0, len(body), 10)]
    with open(path, "w") as f:
        f.write(head + "\n".join(lines) + ")\n")


def make_corpus():
    os.makedirs(CORPUS, exist_ok=True)
    names = ["sample1.sgf"]
    with open(os.path.join(GOLDEN, "sample1.sgf")) as f, open(os.path.join(CORPUS, "sample1.sgf"), "w") as g:
        g.write(f.read())

    def playout(seed, bid, handi=0, cap=400):
        b = OracleBoard()
        if handi:
            b.apply_handicap(handi)
        return [int(m) for m in b.playout(seed, bid, 0, cap)]

    # plain games; the playout ends in two passes so NumMoves stops at the first one (sgf.cc:45-51)
    for i in range(4):
        mv = playout(4242, i, 0, 300 + 40 * i)
        write_sgf(os.path.join(CORPUS, "plain%d.sgf" % i), mv, result=["B+3.5", "W+R", "b+0.5", "W+12"][i])
        names.append("plain%d.sgf" % i)
    # handicap games: white moves first (PlaceHandicap leaves WHITE to move, board.cc:48-64)
    for h in (2, 5, 9):
        mv = playout(777, h, h, 260)
        write_sgf(os.path.join(CORPUS, "handi%d.sgf" % h), mv, handi=h, result="W+R", first="W")
        names.append("handi%d.sgf" % h)
    # comments with escaped brackets, a side property and a variation (flattened by the parser)
    mv = playout(99, 3, 0, 200)
    deco = {5: "C[a comment with an escaped \\] bracket]", 9: "TR[aa][bb]", 17: "C[semi;colon (paren) inside]",
            30: "(;B[aa];W[bb])"}
    write_sgf(os.path.join(CORPUS, "decorated.sgf"), mv, result="B+R", decorate=deco,
              extra_head="C[header comment \\] with escape]")
    names.append("decorated.sgf")
    # an illegal move (an occupied point) in the middle: next() reloads (offpolicy_loader.cc:123-125)
    mv = playout(31337, 9, 0, 220)
    mv = mv[:60] + [mv[10]] + mv[60:]
    write_sgf(os.path.join(CORPUS, "illegal.sgf"), mv, result="W+2.5")
    names.append("illegal.sgf")
    # rejected by after_reload: wrong board size, too short (offpolicy_loader.cc:78)
    write_sgf(os.path.join(CORPUS, "small9.sgf"), [a for a in playout(5, 5, 0, 60) if a // 19 < 9 and a % 19 < 9],
              size=9)
    names.append("small9.sgf")
    write_sgf(os.path.join(CORPUS, "short.sgf"), playout(6, 6, 0, 300)[:7])
    names.append("short.sgf")
    # a pass in the middle truncates the record there
    mv = playout(8, 8, 0, 200)
    write_sgf(os.path.join(CORPUS, "midpass.sgf"), mv[:80] + [-1] + mv[80:])
    names.append("midpass.sgf")
    with open(os.path.join(CORPUS, "train.lst"), "w") as f:
        f.write("\n".join(names) + "\n")
    with open(os.path.join(CORPUS, "one.lst"), "w") as f:
        f.write("sample1.sgf\n")
    # the same corpus as a tar archive (OfflineLoader::InitSharedBuffer takes either,
    # go/offpolicy_loader.cc:27-30); plain ustar headers, members in list order, fixed metadata
    import tarfile
    tar_path = os.path.join(CORPUS, "corpus.tar")
    with tarfile.open(tar_path, "w", format=tarfile.USTAR_FORMAT) as tf:
        for nm in names:
            ti = tf.gettarinfo(os.path.join(CORPUS, nm), arcname=nm)
            ti.mtime, ti.uid, ti.gid, ti.uname, ti.gname, ti.mode = 0, 0, 0, "", "", 0o644
            with open(os.path.join(CORPUS, nm), "rb") as f:
                tf.addfile(ti, f)
    return names


RUNS = [
    dict(name="aug_random", seed=7, num_games=3, games_per_thread=3, data_aug=-1, start_ratio=0.5, ratio=0.0,
         cutoff=-1, k=3, samples=600, lst="train.lst"),
    dict(name="aug0_cutoff", seed=12345, num_games=1, games_per_thread=1, data_aug=0, start_ratio=0.9, ratio=0.3,
         cutoff=70, k=3, samples=400, lst="train.lst"),
    dict(name="k5_aug6", seed=99, num_games=2, games_per_thread=2, data_aug=6, start_ratio=0.2, ratio=0.2,
         cutoff=-1, k=5, samples=300, lst="train.lst"),
    dict(name="tar_corpus", seed=31, num_games=2, games_per_thread=2, data_aug=-1, start_ratio=0.5, ratio=0.2,
         cutoff=-1, k=3, samples=300, lst="corpus.tar"),
    dict(name="multi_game_one_file", seed=2024, num_games=3, games_per_thread=3, data_aug=-1, start_ratio=0.5,
         ratio=0.1, cutoff=-1, k=3, samples=200, lst="one.lst"),
]

# BASELINE.json configs[2] at its stated size: 256 games in ONE batch of 256 (one collector,
# num_games == batchsize), go/game.py's default of 5 records per game thread, random D4.  Kept in a
# file of its own (batch256_streams.json): `make_golden_batch.py --batch256` writes only that.
RUNS_256 = [
    dict(name="cfg3_batch256", seed=7, num_games=256, games_per_thread=5, data_aug=-1, start_ratio=0.5, ratio=0.0,
         cutoff=-1, k=3, samples=12, lst="train.lst"),
]

# History length T > 1 (elf/hist.h:19-66, gating elf/comm_template.h:68-91): every delivered row is
# a window of the game thread's last T samples, oldest first, consecutive windows overlap by one.
# `make_golden_batch.py --hist` writes batch_hist_streams.json (a stream entry = one window = T samples).
RUNS_HIST = [
    dict(name="hist3", T=3, seed=7, num_games=3, games_per_thread=2, data_aug=-1, start_ratio=0.5, ratio=0.0,
         cutoff=-1, k=3, samples=60, lst="train.lst"),
    dict(name="hist4_aug0", T=4, seed=11, num_games=2, games_per_thread=1, data_aug=0, start_ratio=0.9, ratio=0.3,
         cutoff=40, k=3, samples=50, lst="train.lst"),
]

QUIRK_TEXTS = [
    "(;GM[1]SZ[19]RE[b+R]HA[4];B[pd];W[];B[tt];W[dd])", "(;GM[1];B[aa]\nW[bb];W[cc])",
    "(;GM[1];B[aa]C[x \\] ;B[bb] y];W[cc])", "(;GM[1];B[aa](;W[bb])(;W[cc]))", "(;GM[1];B[aa];(;W[bb]);W[cc])",
    "(;GM[1]\n SZ[19]  RE[ W+3.5];B[aa] ;W[ bb];B[cc\n])", "(;GM[1]C[a;b)c];B[aa];W[bb])",
    "(;GM[1];B[aa]AB[cc][dd];W[bb])", "(;GM[1]SZ[13]KM[0.5]RE[W+R]\n;W[aa];B[sa];W[as];B[ss];W[ta])",
    "(;GM[1])", "(;FF[4]RE[];B[jj];W[kk];B[];W[ll])",
]

NP_TYPES = {"int32_t": "i4", "int64_t": "i8", "float": "f4", "unsigned char": "u1", "char": "i1"}
KEYS = ["s", "offline_a", "id", "seq", "game_counter", "last_terminal", "move_idx", "winner", "aug_code",
        "game_record_idx"]


_KEEP_ALIVE = []


def run_reference(go, cfg):
    co = go.ContextOptions()
    co.num_games = cfg["num_games"]
    T = cfg.get("T", 1)
    co.T = T
    co.num_collectors = 1
    opt = go.GameOptions()
    opt.seed = cfg["seed"]
    opt.mode = "offline"
    opt.list_filename = os.path.join(CORPUS, cfg["lst"])
    opt.data_aug = cfg["data_aug"]
    opt.start_ratio_pre_moves = cfg["start_ratio"]
    opt.ratio_pre_moves = cfg["ratio"]
    opt.move_cutoff = cfg["cutoff"]
    opt.num_games_per_thread = cfg["games_per_thread"]
    opt.num_future_actions = cfg["k"]
    opt.verbose = False
    GC = go.GameContext(co, opt)
    gstat = GC.CreateGroupStat()
    gstat.hist_len = T
    B = cfg["num_games"]
    gid = GC.AddCollectors(B, 0, 0, gstat)
    bufs = {}
    for key in KEYS:
        info = GC.GetTensorSpec(gid, key, T)
        assert info.key == key, key
        v = np.zeros(info.sz, dtype=NP_TYPES[info.type])
        info.p = v.ctypes.data
        info.byte_size = v.size * v.dtype.itemsize
        GC.AddTensor(gid, "input", info)
        bufs[key] = v
    GC.Start()
    streams = {g: [] for g in range(B)}
    while min(len(v) for v in streams.values()) < cfg["samples"]:
        infos = GC.Wait(0)
        n = len(infos.s)
        for r in range(n):
            g = int(bufs["id"][T - 1, r])

            def entry(t):
                return ([int(bufs[k][t, r]) for k in KEYS[2:]] + [[int(a) for a in bufs["offline_a"][t, r]]] +
                        ["%016x" % fnv_planes(FNV_SEED, bufs["s"][t, r].reshape(-1))])
            streams[g].append(entry(0) if T == 1 else [entry(t) for t in range(T)])
        GC.Steps(infos)
    # no GC.Stop(): the reference's teardown blocks on its worker threads.  Its threads keep writing
    # into the buffers, so context and buffers stay alive until the caller (one process per
    # configuration) leaves with os._exit
    _KEEP_ALIVE.append((GC, bufs))
    return {str(g): v[:cfg["samples"]] for g, v in streams.items()}


def main():
    so = glob.glob(os.path.join(ROOT, "oracle", "_ref", "go_game*.so"))
    if not so:
        raise SystemExit("oracle/_ref/go_game*.so missing: run `make -C oracle refmod` where /root/reference is mounted")
    if len(sys.argv) > 2 and sys.argv[1] == "--run":     # child: one configuration, then leave hard
        sys.path.insert(0, os.path.dirname(so[0]))
        import go_game as go
        st = run_reference(go, (RUNS + RUNS_256 + RUNS_HIST)[int(sys.argv[2])])
        dump_json(sys.argv[3], st)
        sys.stdout.flush()
        os._exit(0)
    import subprocess
    import tempfile
    if "--batch256" in sys.argv or "--hist" in sys.argv:  # the corpus stays as committed
        hist = "--hist" in sys.argv
        runs, first = (RUNS_HIST, len(RUNS) + len(RUNS_256)) if hist else (RUNS_256, len(RUNS))
        fname = "batch_hist_streams.json" if hist else "batch256_streams.json"
        with open(os.path.join(CORPUS, "train.lst")) as f:
            names = f.read().split()
        out = dict(corpus=names, fields=KEYS[2:] + ["offline_a", "s_fnv"], runs=[])
        for i, cfg in enumerate(runs):
            with tempfile.TemporaryDirectory() as td:
                tmp = os.path.join(td, "run.json")
                for attempt in range(4):
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--run", str(first + i), tmp],
                                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
                    if r.returncode == 0 and os.path.exists(tmp):
                        break
                else:
                    raise SystemExit("reference run %s failed" % cfg["name"])
                with open(tmp) as f:
                    st = json.load(f)
            out["runs"].append(dict(cfg=cfg, streams=st))
            print(cfg["name"], len(st), "games x", cfg["samples"], "samples", flush=True)
        dump_json(os.path.join(GOLDEN, fname), out)
        print("written", os.path.join(GOLDEN, fname))
        return
    names = make_corpus()
    # what the reference's Sgf::Load makes of every corpus file (oracle/_ref/libgoref.so)
    from oracle.pyoracle import RefBoard
    parsed = {}
    for nm in names:
        sg = RefBoard.sgf_load(os.path.join(CORPUS, nm))
        parsed[nm] = dict(moves=[int(m) for m in sg["moves"]], num_moves=sg["num_moves"], winner=sg["winner"],
                          handi=sg["handi"], board_size=sg["board_size"])
    dump_json(os.path.join(GOLDEN, "corpus_parse.json"), parsed)
    # small texts that exercise the parser's quirks, with the reference's reading of each
    quirks = []
    for text in QUIRK_TEXTS:
        tmp = os.path.join(CORPUS, "_quirk_tmp.sgf")
        with open(tmp, "w") as f:
            f.write(text)
        try:
            sg = RefBoard.sgf_load(tmp)
            res = dict(moves=[int(m) for m in sg["moves"]], num_moves=sg["num_moves"], winner=sg["winner"],
                       handi=sg["handi"], board_size=sg["board_size"])
        except RuntimeError:
            res = None
        os.unlink(tmp)
        quirks.append(dict(text=text, ref=res))
    dump_json(os.path.join(GOLDEN, "sgf_quirks.json"), quirks)
    out = dict(corpus=names, fields=KEYS[2:] + ["offline_a", "s_fnv"], runs=[])
    for i, cfg in enumerate(RUNS):
        with tempfile.TemporaryDirectory() as td:
            tmp = os.path.join(td, "run.json")
            for attempt in range(4):   # the reference's thread teardown occasionally crashes the child
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--run", str(i), tmp],
                                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
                if r.returncode == 0 and os.path.exists(tmp):
                    break
            else:
                raise SystemExit("reference run %d failed" % i)
            with open(tmp) as f:
                st = json.load(f)
        out["runs"].append(dict(cfg=cfg, streams=st))
        print(cfg["name"], {g: len(v) for g, v in st.items()}, flush=True)
    dump_json(os.path.join(GOLDEN, "batch_streams.json"), out)
    print("written", os.path.join(GOLDEN, "batch_streams.json"))


if __name__ == "__main__":
    main()
