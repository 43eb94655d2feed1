"""Python caller's side of the GameContext boundary: ``Batch`` and ``GCWrapper``.

Public surface of the reference's ``elf/utils_elf.py`` (which no longer parses on Python 3: it
uses ``async`` as an identifier, elf/utils_elf.py:191) -- descriptor dicts in; ``reg_callback`` /
``reg_callback_if_exists`` / ``Start`` / ``Run`` / ``Stop`` / ``PrintSummary`` out; batches that
answer ``batch["key"]`` (with the ``last_`` fallback), ``hist(t)``, ``first_k(n)``,
``copy_from(reply)``; ``batchsize`` / ``max_batchsize`` / ``infos`` on what a callback receives.

What differs is where tensors live.  With ``gpu`` set, every batch tensor is allocated ON that
device and registered with the context by ``data_ptr()``: the environment's kernels write planes,
masks and labels into them in place, so the pinned host copy and the cpu->gpu transfer of the
reference (elf/utils_elf.py:191-195,229,340-343) do not exist.  With ``gpu=None`` tensors are host
memory, as in the reference, and the context stages them through the device.
"""
import numpy as np
import torch

# type strings of EntryInfo.type (elf/copier.hh:44-52) -> array dtypes
_TORCH = {"float": torch.float32, "int64_t": torch.int64, "int32_t": torch.int32, "char": torch.int8,
          "unsigned char": torch.uint8}
_NUMPY = {"float": "f4", "int64_t": "i8", "int32_t": "i4", "char": "i1", "unsigned char": "u1"}


def _allocate(info, device, use_numpy, zero=False):
    """Storage for one EntryInfo; fills in its pointer and byte size.  Filled with ones like the
    reference does (elf/utils_elf.py:58,63) so that untouched entries are visible in debugging;
    `zero`: filled with zeros instead (the planes tensor of a context that never writes the constant
    planes, GameOptions.preset_zero_planes)."""
    shape = tuple(info.sz)
    if use_numpy:
        arr = (np.zeros if zero else np.ones)(shape, dtype=_NUMPY[info.type])
        info.p, info.byte_size = arr.ctypes.data, arr.nbytes
        return arr
    t = (torch.zeros if zero else torch.ones)(shape, dtype=_TORCH[info.type], device=device or "cpu")
    if device is None and torch.cuda.is_available():
        t = t.pin_memory()
    info.p, info.byte_size = t.data_ptr(), t.numel() * t.element_size()
    return t


def _spec(GC, gid, key, T):
    """Tensor spec of `key`, or of `last_<key>` when the context does not know `key`."""
    for name in (key, "last_" + key):
        info = GC.GetTensorSpec(gid, name, T)
        if info.key:
            return info
    raise ValueError("key[%s] or last_key[last_%s] is not specified!" % (key, key))


class Batch:
    """A dict of [T, B, ...] tensors with the reference's access helpers (elf/utils_elf.py:13-211)."""

    def __init__(self, _batchsize=None, **tensors):
        if isinstance(_batchsize, int):
            tensors = {k: v[:, :_batchsize] for k, v in tensors.items()}
        self.batch = tensors

    @classmethod
    def load(cls, GC, input_reply, desc, group_id, device=None, use_numpy=False, zero_planes=False):
        """Allocate every key of a descriptor half ({"T":..., "keys":...}) and register it."""
        self = cls()
        self.GC, self.dims = GC, {}
        for key in desc["keys"]:
            info = _spec(GC, group_id, key, desc["T"])
            self.batch[info.key] = _allocate(info, device, use_numpy, zero=zero_planes and info.key == "s")
            self.dims[info.key] = info
            GC.AddTensor(group_id, input_reply, info)
        return self

    # -- views ---------------------------------------------------------------------------------
    def first_k(self, batchsize):
        return Batch(_batchsize=batchsize, **self.batch)

    def hist(self, s, key=None):
        """Time slice `s` of every key (a Batch), or of one key (a tensor)."""
        if key is not None:
            return self[key][s]
        view = Batch(**{k: v[s] for k, v in self.batch.items()})
        if hasattr(self, "GC"):
            view.GC = self.GC
        return view

    def __contains__(self, key):
        return key in self.batch or ("last_" + key) in self.batch

    def __getitem__(self, key):
        try:
            return self.batch[key]
        except KeyError:
            pass
        shifted = self.batch.get("last_" + key)
        if shifted is None:
            raise KeyError("Batch(): specified key: %s or last_%s not found!" % (key, key))
        return shifted[1:]

    # -- updates -------------------------------------------------------------------------------
    def add(self, key, value):
        self.batch[key] = value
        return self

    def setzero(self):
        for v in self.batch.values():
            v[:] = 0

    def copy_from(self, src, batch_key=""):
        """Write a callback's reply into the registered reply tensors.  Every registered key must
        be present (None = leave as is); unknown keys are an error (elf/utils_elf.py:132-163)."""
        reply = src.batch if isinstance(src, Batch) else src
        unknown = [k for k in reply if k not in self.batch]
        if unknown:
            raise ValueError("Batch[%s]: \"%s\" in reply is missing in batch specification" % (batch_key, unknown[0]))
        missing = [k for k in self.batch if k not in reply]
        if missing:
            raise ValueError("Batch[%s].copy_from: Reply[%s] is not assigned" % (batch_key, missing[0]))
        for k, v in reply.items():
            dst = self.batch[k]
            if v is None:
                continue
            if isinstance(v, (int, float)):
                dst.fill_(v)
            elif isinstance(v, list) and dst.numel() == len(v):
                dst.view(-1).copy_(torch.as_tensor(v, dtype=dst.dtype))
            else:
                dst[:] = v

    def to_numpy(self):
        return {k: (v if isinstance(v, np.ndarray) else v.cpu().numpy()) for k, v in self.batch.items()}


class GCWrapper:
    """Collectors, batches and the Wait -> callback -> Steps loop (elf/utils_elf.py:214-402)."""

    def __init__(self, GC, co, descriptions, use_numpy=False, gpu=None, params=None, zero_planes=False):
        self.GC = GC
        self.gpu = gpu
        self.device = None if gpu is None else torch.device("cuda", gpu)
        self.params = {} if params is None else params
        self._cb = {}
        self.inputs, self.replies = [], []
        self.idx2name, self.name2idx = {}, {}
        self.gid2gpu, self.gpu2gid = {}, []
        # collectors per descriptor: co.num_collectors, or as many batches as the games fill
        per_desc = co.num_collectors
        if per_desc <= 0:
            per_desc = co.num_games // max(1, sum(d["batchsize"] for d in descriptions.values()))
        per_desc = max(1, per_desc)
        for exclusive_id, (name, d) in enumerate(descriptions.items()):
            reply = d["reply"]
            gstat = GC.CreateGroupStat()
            gstat.hist_len = max(d["input"]["T"], reply["T"] if reply is not None else 0)
            gstat.name = d.get("name", "")
            self.gpu2gid.append([])
            self.name2idx[name] = []
            for _ in range(per_desc):
                gid = GC.AddCollectors(d["batchsize"], exclusive_id, d.get("timeout_usec", 0), gstat)
                inp = Batch.load(GC, "input", d["input"], gid, device=self.device, use_numpy=use_numpy,
                                 zero_planes=zero_planes)
                inp.batchsize = d["batchsize"]
                self.inputs.append(inp)
                rep = None
                if reply is not None:
                    rep = Batch.load(GC, "reply", reply, gid, device=self.device, use_numpy=use_numpy)
                    rep.batchsize = d["batchsize"]
                    rep.setzero()
                self.replies.append(rep)
                self.idx2name[gid] = name
                self.name2idx[name].append(gid)
                self.gpu2gid[-1].append(gid)
                self.gid2gpu[gid] = exclusive_id

    # -- callbacks -----------------------------------------------------------------------------
    def reg_has_callback(self, key):
        return key in self.name2idx

    def reg_callback(self, key, cb):
        if key not in self.name2idx:
            raise ValueError("Callback[%s] is not in the specification" % key)
        for gid in self.name2idx[key]:
            self._cb[gid] = cb
        return True

    def reg_callback_if_exists(self, key, cb):
        return self.reg_has_callback(key) and self.reg_callback(key, cb)

    def _call(self, infos):
        gid = infos.gid
        if gid not in self._cb:
            raise ValueError("info.gid[%d] is not in callback functions" % gid)
        cb = self._cb[gid]
        if cb is None:
            return None
        # Infos.batchsize() (elf/state_collector.h:30-40) instead of len(infos.s): the latter converts the
        # whole id list to Python objects on every batch (0.1 ms at 8,192 games)
        n = infos.batchsize() if hasattr(infos, "batchsize") else len(infos.s)
        picked = self.inputs[gid].first_k(n)      # lives on the device already when gpu is set
        picked.GC, picked.infos = self.GC, infos
        picked.batchsize, picked.max_batchsize = n, self.inputs[gid].batchsize
        out = cb(picked)
        if isinstance(out, dict) and self.replies[gid] is not None:
            self.replies[gid].first_k(n).copy_from(out, batch_key="%s-%d" % (self.idx2name[gid], gid))
        return out

    # -- life cycle ------------------------------------------------------------------------------
    def Start(self):
        for name, gids in self.name2idx.items():
            for gid in gids:
                if gid not in self._cb:
                    raise ValueError("GCWrapper.Start(): No callback function for key = %s and gid = %d" % (name, gid))
        self.GC.Start()

    def Run(self):
        """One batch: wait for it, hand it to its callback, write the reply back, release it."""
        self.infos = self.GC.Wait(0)
        if self.infos.gid < 0:
            return None
        res = self._call(self.infos)
        self.GC.Steps(self.infos)
        return res

    def Stop(self):
        self.GC.Stop()

    def PrintSummary(self):
        self.GC.PrintSummary()
