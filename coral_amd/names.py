"""Read names as ONE byte blob + offsets.

The reference keys its dicts and sets by ``query_name`` strings (/root/reference/src/infer_breakpoint_graph.py:141-151,
:379-384); the graph build itself needs only a name's IDENTITY (name id = order of first appearance in the file), the
``hash()`` of the names of chimeric reads (the set-order replay) and, for whatever is finally printed or iterated, the ``str``.
A 2 M-read sample has 2 M names; a Python ``str`` per name costs a second to create and is touched by every garbage
collection.  ``NameTable`` therefore keeps the bytes the decoder delivered and makes ``str`` objects only for the ids somebody
asks for (``coral_amd._pyobjects``: C loops, no per-item interpreter work).  It behaves as a read-only sequence of ``str``.
"""
from __future__ import annotations

from collections.abc import Sequence
from typing import Iterable, List, Optional

import numpy as np

from . import _pyobjects


class NameTable(Sequence):
    __slots__ = ("blob", "off", "_list", "_index")

    def __init__(self, blob: np.ndarray, off: np.ndarray):
        self.blob = np.ascontiguousarray(blob, dtype=np.uint8)
        self.off = np.ascontiguousarray(off, dtype=np.int64)
        assert self.off.ndim == 1 and len(self.off) >= 1 and int(self.off[-1]) <= len(self.blob)
        self._list: Optional[List[str]] = None
        self._index = None

    # -- construction ------------------------------------------------------------------------------------
    @classmethod
    def from_list(cls, names: Iterable[str]) -> "NameTable":
        enc = [s.encode() for s in names]
        off = np.zeros(len(enc) + 1, dtype=np.int64)
        if enc:
            np.cumsum(np.fromiter(map(len, enc), dtype=np.int64, count=len(enc)), out=off[1:])
        return cls(np.frombuffer(b"".join(enc), dtype=np.uint8), off)

    @classmethod
    def coerce(cls, names) -> "NameTable":
        return names if isinstance(names, NameTable) else cls.from_list(names)

    @classmethod
    def from_decimal(cls, prefix: str, values: np.ndarray, width: int) -> Optional["NameTable"]:
        """``[prefix + "%0{width}d" % v for v in values]`` without a Python loop; None if a value needs more digits."""
        v = np.ascontiguousarray(values, dtype=np.int64)
        if len(v) and (int(v.min()) < 0 or int(v.max()) >= 10 ** width):
            return None
        p = np.frombuffer(prefix.encode(), dtype=np.uint8)
        rows = np.empty((len(v), len(p) + width), dtype=np.uint8)
        rows[:, :len(p)] = p
        for d in range(width):
            rows[:, len(p) + width - 1 - d] = (v // 10 ** d) % 10 + 48
        return cls(rows.reshape(-1), np.arange(len(v) + 1, dtype=np.int64) * (len(p) + width))

    # -- sequence of str ---------------------------------------------------------------------------------
    def __len__(self):
        return len(self.off) - 1

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self.take(np.arange(len(self))[i])
        if self._list is not None:
            return self._list[i]
        i = int(i)
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError("name id out of range")
        return bytes(self.blob[self.off[i]:self.off[i + 1]]).decode()

    def __iter__(self):
        return iter(self.tolist())

    def __eq__(self, other):
        if isinstance(other, NameTable):
            return np.array_equal(self.off - self.off[0], other.off - other.off[0]) and \
                np.array_equal(self.blob[self.off[0]:self.off[-1]], other.blob[other.off[0]:other.off[-1]])
        if isinstance(other, (list, tuple)):
            return len(other) == len(self) and self.tolist() == list(other)
        return NotImplemented

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    __hash__ = None

    def __repr__(self):
        return "NameTable(%d names, %d bytes)" % (len(self), len(self.blob))

    def tolist(self) -> List[str]:
        """Every name as a ``str`` (made once, kept: later single lookups return these objects)."""
        if self._list is None:
            self._list = _pyobjects.blob_names(self.blob, self.off, None)
        return self._list

    # -- what the build asks for -------------------------------------------------------------------------
    def take(self, ids) -> List[str]:
        """``[names[k] for k in ids]``"""
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        if self._list is not None:
            return _pyobjects.names_of(self._list, ids)
        return _pyobjects.blob_names(self.blob, self.off, ids)

    def tuples(self, ids, i, j) -> list:
        """``[(names[ids[k]], int(i[k]), int(j[k])) ...]`` — the ``(name, i, j)`` read tuples of bu:81 / ibg:772."""
        ids, i, j = (np.ascontiguousarray(a, dtype=np.int64) for a in (ids, i, j))
        if self._list is not None:
            return _pyobjects.read_tuples(self._list, ids, i, j)
        return _pyobjects.blob_tuples(self.blob, self.off, ids, i, j)

    def hashes(self, ids) -> np.ndarray:
        """``hash(names[k])`` of this interpreter for every k in ids (int64), without creating the strings."""
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        out = np.empty(len(ids), dtype=np.int64)
        _pyobjects.blob_hashes(self.blob, self.off, ids, out)
        return out

    def index(self, name, *a):
        return self.index_map()[name] if not a else self.tolist().index(name, *a)

    def index_map(self) -> dict:
        """name -> id (built once; only the steps after the graph build look names up)."""
        if self._index is None:
            self._index = {nm: k for k, nm in enumerate(self.tolist())}
        return self._index
