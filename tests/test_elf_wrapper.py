"""CPU tests of the Python caller's glue (elf_amd/elf_wrapper.py) against a stand-in context:
the semantics callers of elf/utils_elf.py rely on -- first_k, hist, last_ fallback, copy_from
checks, collector counts, callback dispatch."""
import types

import pytest
import torch

from elf_amd.elf_wrapper import Batch, GCWrapper


class FakeGC:
    SPECS = {"s": ("float", [3]), "a": ("int64_t", []), "last_r": ("float", []), "pi": ("float", [5])}

    def __init__(self):
        self.collectors, self.tensors, self.stepped, self.batches = [], [], [], []

    def CreateGroupStat(self):
        return types.SimpleNamespace(hist_len=1, name="")

    def AddCollectors(self, batchsize, exclusive_id, timeout_usec, gstat):
        self.collectors.append((batchsize, exclusive_id, timeout_usec, gstat.hist_len, gstat.name))
        return len(self.collectors) - 1

    def GetTensorSpec(self, gid, key, T):
        e = types.SimpleNamespace(key="", type="", sz=[T, self.collectors[gid][0]], p=0, byte_size=0)
        if key in self.SPECS:
            e.key, e.type = key, self.SPECS[key][0]
            e.sz = e.sz + self.SPECS[key][1]
        return e

    def AddTensor(self, gid, which, info):
        assert info.p != 0 and info.byte_size > 0
        self.tensors.append((gid, which, info.key))

    def Start(self):
        self.started = True

    def Wait(self, t):
        return self.batches.pop(0) if self.batches else types.SimpleNamespace(gid=-1, s=[])

    def Steps(self, infos):
        self.stepped.append(infos.gid)

    def Stop(self):
        self.stopped = True

    def PrintSummary(self):
        pass


DESC = {"actor": dict(batchsize=4, input=dict(T=1, keys=["s", "r"]), reply=dict(T=1, keys=["a", "pi"]), name="actor",
                      timeout_usec=10),
        "train": dict(batchsize=2, input=dict(T=2, keys=["s"]), reply=None)}


def test_collectors_batches_and_last_fallback():
    gc = FakeGC()
    co = types.SimpleNamespace(num_games=12, num_collectors=0)
    w = GCWrapper(gc, co, DESC)
    # 12 games // (4 + 2) = 2 collectors per descriptor, one exclusive id per descriptor
    assert gc.collectors == [(4, 0, 10, 1, "actor")] * 2 + [(2, 1, 0, 2, "")] * 2
    assert w.name2idx == {"actor": [0, 1], "train": [2, 3]} and w.gpu2gid == [[0, 1], [2, 3]]
    assert ("r" not in w.inputs[0].batch) and ("last_r" in w.inputs[0].batch) and "r" in w.inputs[0]
    assert tuple(w.inputs[0]["s"].shape) == (1, 4, 3) and tuple(w.inputs[2]["s"].shape) == (2, 2, 3)
    assert w.replies[2] is None and float(w.replies[0]["pi"].abs().sum()) == 0.0       # replies start zeroed
    assert float(w.inputs[0]["s"].min()) == 1.0                                        # inputs filled with ones
    assert w.inputs[0].first_k(3)["s"].shape[1] == 3
    assert tuple(w.inputs[2].hist(1)["s"].shape) == (2, 3) and tuple(w.inputs[2].hist(0, "s").shape) == (2, 3)
    with pytest.raises(KeyError):
        w.inputs[0]["nope"]
    co.num_collectors = 3
    assert len(GCWrapper(FakeGC(), co, DESC).inputs) == 6


def test_unknown_key_raises():
    gc = FakeGC()
    co = types.SimpleNamespace(num_games=4, num_collectors=1)
    with pytest.raises(ValueError, match="not specified"):
        GCWrapper(gc, co, {"x": dict(batchsize=2, input=dict(T=1, keys=["zzz"]), reply=None)})


def test_run_dispatches_and_writes_reply():
    gc = FakeGC()
    co = types.SimpleNamespace(num_games=4, num_collectors=1)
    w = GCWrapper(gc, co, DESC)
    seen = []

    def actor(batch):
        seen.append((batch.batchsize, batch.max_batchsize, batch.infos.gid, tuple(batch["s"].shape)))
        return dict(a=[7, 8, 9], pi=0.25)

    assert w.reg_callback_if_exists("nope", actor) is False
    with pytest.raises(ValueError):
        w.reg_callback("nope", actor)
    w.reg_callback("actor", actor)
    with pytest.raises(ValueError, match="No callback"):
        w.Start()
    w.reg_callback("train", None)
    w.Start()
    gc.batches = [types.SimpleNamespace(gid=0, s=[5, 6, 7]), types.SimpleNamespace(gid=1, s=[1, 2])]
    w.Run()
    assert seen == [(3, 4, 0, (1, 3, 3))] and gc.stepped == [0]
    assert w.replies[0]["a"][0].tolist() == [7, 8, 9, 0] and w.replies[0]["pi"][0, :3].eq(0.25).all()
    assert w.Run() is None and gc.stepped == [0, 1]          # callback None: batch released untouched
    assert w.Run() is None and gc.stepped == [0, 1]          # nothing ready: gid -1, nothing stepped
    with pytest.raises(ValueError, match="missing in batch specification"):
        w.replies[0].copy_from(dict(a=1, pi=1, V=0))
    with pytest.raises(ValueError, match="not assigned"):
        w.replies[0].copy_from(dict(a=1))
    w.replies[0].copy_from(dict(a=None, pi=torch.full((1, 4, 5), 0.5)))
    assert w.replies[0]["pi"].eq(0.5).all() and w.replies[0]["a"][0].tolist() == [7, 8, 9, 0]
