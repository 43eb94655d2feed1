"""Python caller's side of the GameContext boundary: ``Batch`` and ``GCWrapper``.

Same surface as the reference's ``elf/utils_elf.py`` (which no longer parses on Python 3: it uses
``async`` as an identifier, elf/utils_elf.py:191): descriptor dicts in, ``reg_callback`` /
``Start`` / ``Run`` / ``Stop`` out, ``batch["key"]`` with the ``last_`` fallback, ``hist(t)``,
``first_k``.  The difference is where tensors live: with ``gpu`` set, batch tensors are allocated
ON the device and registered with the context by ``data_ptr()`` -- the environment writes planes,
masks and labels into them directly, so there is no pinned host copy and no cpu->gpu transfer
(elf/utils_elf.py:191-195,340-343 become no-ops).  With ``gpu=None`` tensors are host memory, as
in the reference, and the context stages through the device.
"""
import math
from collections import defaultdict

import numpy as np
import torch


class Batch:
    """elf/utils_elf.py:13-211."""
    torch_types = {"int32_t": torch.int32, "int64_t": torch.int64, "float": torch.float32,
                   "unsigned char": torch.uint8, "char": torch.int8}
    numpy_types = {"int32_t": "i4", "int64_t": "i8", "float": "f4", "unsigned char": "u1", "char": "i1"}

    def __init__(self, _batchsize=None, **kwargs):
        if isinstance(_batchsize, int):
            self.batch = {k: v[:, :_batchsize] for k, v in kwargs.items()}
        else:
            self.batch = kwargs

    def first_k(self, batchsize):
        return Batch(_batchsize=batchsize, **self.batch)

    @staticmethod
    def _request(GC, group_id, key, T):
        info = GC.GetTensorSpec(group_id, key, T)
        if info.key == "":
            last_key = "last_" + key
            info = GC.GetTensorSpec(group_id, last_key, T)
            if info.key == "":
                raise ValueError("key[%s] or last_key[%s] is not specified!" % (key, last_key))
        return info

    @staticmethod
    def _alloc(info, device=None, use_numpy=False):
        if use_numpy:
            v = np.zeros(info.sz, dtype=Batch.numpy_types[info.type])
            v[:] = 1
            info.p = v.ctypes.data
            info.byte_size = v.size * v.dtype.itemsize
        else:
            v = torch.ones(*info.sz, dtype=Batch.torch_types[info.type], device=device if device is not None else "cpu")
            if device is None and torch.cuda.is_available():
                v = v.pin_memory()
            info.p = v.data_ptr()
            info.byte_size = v.numel() * v.element_size()
        return v, info

    @staticmethod
    def load(GC, input_reply, desc, group_id, device=None, use_numpy=False):
        batch = Batch()
        batch.dims = {}
        batch.GC = GC
        for key in desc["keys"]:
            info = Batch._request(GC, group_id, key, desc["T"])
            v, info = Batch._alloc(info, device=device, use_numpy=use_numpy)
            batch.batch[info.key] = v
            batch.dims[info.key] = info
            GC.AddTensor(group_id, input_reply, info)
        return batch

    def __getitem__(self, key):
        if key in self.batch:
            return self.batch[key]
        key_with_last = "last_" + key
        if key_with_last in self.batch:
            return self.batch[key_with_last][1:]
        raise KeyError("Batch(): specified key: %s or %s not found!" % (key, key_with_last))

    def add(self, key, value):
        self.batch[key] = value
        return self

    def __contains__(self, key):
        return key in self.batch or "last_" + key in self.batch

    def setzero(self):
        for v in self.batch.values():
            v[:] = 0

    def copy_from(self, src, batch_key=""):
        this_src = src if isinstance(src, dict) else src.batch
        key_assigned = {k: False for k in self.batch}
        for k, v in this_src.items():
            if k not in self.batch:
                raise ValueError("Batch[%s]: \"%s\" in reply is missing in batch specification" % (batch_key, k))
            bk = self.batch[k]
            key_assigned[k] = True
            if v is None:
                continue
            if isinstance(v, list) and bk.numel() == len(v):
                bk.view(-1).copy_(torch.as_tensor(v, dtype=bk.dtype))
            elif isinstance(v, (int, float)):
                bk.fill_(v)
            else:
                bk[:] = v
        for k, assigned in key_assigned.items():
            if not assigned:
                raise ValueError("Batch[%s].copy_from: Reply[%s] is not assigned" % (batch_key, k))

    def hist(self, s, key=None):
        if key is None:
            new_batch = Batch(**{k: v[s] for k, v in self.batch.items()})
            if hasattr(self, "GC"):
                new_batch.GC = self.GC
            return new_batch
        return self[key][s]

    def to_numpy(self):
        return {k: (v.cpu().numpy() if not isinstance(v, np.ndarray) else v) for k, v in self.batch.items()}


class GCWrapper:
    """elf/utils_elf.py:214-402."""

    def __init__(self, GC, co, descriptions, use_numpy=False, gpu=None, params=None):
        self.gpu = gpu
        self.device = torch.device("cuda", gpu) if gpu is not None else None
        self._init_collectors(GC, co, descriptions, use_numpy=use_numpy)
        self.params = params if params is not None else {}
        self._cb = {}

    def _init_collectors(self, GC, co, descriptions, use_numpy=False):
        total_batchsize = sum(v["batchsize"] for v in descriptions.values())
        if co.num_collectors > 0:
            num_recv_thread = co.num_collectors
        else:
            num_recv_thread = math.floor(co.num_games / total_batchsize)
        num_recv_thread = max(num_recv_thread, 1)
        inputs, replies = [], []
        idx2name, name2idx = {}, defaultdict(list)
        gid2gpu, gpu2gid = {}, []
        for key, v in descriptions.items():
            inp, reply, batchsize = v["input"], v["reply"], v["batchsize"]
            T = inp["T"]
            if reply is not None and reply["T"] > T:
                T = reply["T"]
            gstat = GC.CreateGroupStat()
            gstat.hist_len = T
            gstat.name = v.get("name", "")
            timeout_usec = v.get("timeout_usec", 0)
            gpu2gid.append([])
            for _ in range(num_recv_thread):
                group_id = GC.AddCollectors(batchsize, len(gpu2gid) - 1, timeout_usec, gstat)
                input_batch = Batch.load(GC, "input", inp, group_id, device=self.device, use_numpy=use_numpy)
                input_batch.batchsize = batchsize
                inputs.append(input_batch)
                if reply is not None:
                    reply_batch = Batch.load(GC, "reply", reply, group_id, device=self.device, use_numpy=use_numpy)
                    reply_batch.batchsize = batchsize
                    replies.append(reply_batch)
                else:
                    replies.append(None)
                idx2name[group_id] = key
                name2idx[key].append(group_id)
                gpu2gid[-1].append(group_id)
                gid2gpu[group_id] = len(gpu2gid) - 1
        for reply in replies:
            if reply is not None:
                reply.setzero()
        self.GC, self.inputs, self.replies = GC, inputs, replies
        self.idx2name, self.name2idx, self.gid2gpu, self.gpu2gid = idx2name, name2idx, gid2gpu, gpu2gid

    def reg_has_callback(self, key):
        return key in self.name2idx

    def reg_callback_if_exists(self, key, cb):
        if self.reg_has_callback(key):
            self.reg_callback(key, cb)
            return True
        return False

    def reg_callback(self, key, cb):
        if key not in self.name2idx:
            raise ValueError("Callback[%s] is not in the specification" % key)
        for gid in self.name2idx[key]:
            self._cb[gid] = cb
        return True

    def _call(self, infos):
        if infos.gid not in self._cb:
            raise ValueError("info.gid[%d] is not in callback functions" % infos.gid)
        if self._cb[infos.gid] is None:
            return None
        batchsize = len(infos.s)
        picked = self.inputs[infos.gid].first_k(batchsize)   # already on the device when gpu is set
        picked.infos = infos
        picked.batchsize = batchsize
        picked.max_batchsize = self.inputs[infos.gid].batchsize
        picked.GC = self.GC
        sel_reply = None
        if len(self.replies) > infos.gid and self.replies[infos.gid] is not None:
            sel_reply = self.replies[infos.gid].first_k(batchsize)
        reply = self._cb[infos.gid](picked)
        if isinstance(reply, dict) and sel_reply is not None:
            sel_reply.copy_from(reply, batch_key="%s-%d" % (self.idx2name[infos.gid], infos.gid))
        return reply

    def _check_callbacks(self):
        for key, gids in self.name2idx.items():
            for gid in gids:
                if gid not in self._cb:
                    raise ValueError("GCWrapper.Start(): No callback function for key = %s and gid = %d" % (key, gid))

    def Run(self):
        self.infos = self.GC.Wait(0)
        if self.infos.gid < 0:
            return None
        res = self._call(self.infos)
        self.GC.Steps(self.infos)
        return res

    def Start(self):
        self._check_callbacks()
        self.GC.Start()

    def Stop(self):
        self.GC.Stop()

    def PrintSummary(self):
        self.GC.PrintSummary()
