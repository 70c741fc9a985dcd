"""Set-valued fields of the object surface, kept as index arrays until somebody looks inside.

The reference stores, per breakpoint, the ``set`` of supporting ``(read name, i, j)`` tuples (``new_bp_list[k][-1]`` =
``discordant_edges[k][10]``, /root/reference/src/infer_breakpoint_graph.py:330-335, :1000) and, per concordant edge, the
``set`` of read names covering it (``concordant_edges[k][9]``, ibg:1054).  The graph build itself only ever needs their
SIZE (``lr_count = len(reads)``, ibg:1000), unions (``|=``, ibg:330) and membership of read names (ibg:1047-1055); the
elements are read by the step AFTER the build (``compute_path_constraints``, ibg:1071, :1298; the cycle step itself never
touches them — cycle_decomposition.py has no reference to field 9 / 10).  At 2 M reads that is ≈164 000 tuples and as many
name strings per build whose construction alone cost more than every GPU kernel together.

``ReadSupportSet`` / ``ReadNameSet`` therefore hold integer arrays (name id, i, j) and build the real Python ``set`` — by
exactly the reference's sequence of ``set(list)`` and ``|= set(other)`` operations, so that even the ITERATION ORDER of the
materialised set is the reference's — the first time an element is asked for.  They implement the full
``collections.abc.MutableSet`` protocol (comparison with real sets included); they are deliberately NOT subclasses of
``set``: CPython's C-level fast paths (``set(x)``, ``s.update(x)``, ``s |= x``) read a subclass's hash table directly and
would silently see an empty one, whereas a non-``set`` iterable always goes through ``__iter__``.  ``isinstance(x, set)`` is
therefore False (``isinstance(x, collections.abc.Set)`` is True); ``x.as_set()`` returns the plain ``set``.
"""
from __future__ import annotations

from collections.abc import MutableSet, Set

import numpy as np

from . import _pyobjects


class _LazySet(MutableSet):
    __slots__ = ("_names", "_set", "_size")

    def _build(self) -> set:
        raise NotImplementedError

    def _count(self) -> int:
        raise NotImplementedError

    # -- materialisation -----------------------------------------------------------------------------
    def as_set(self) -> set:
        """The real ``set`` (built on first use; the same object afterwards, so in-place updates are kept)."""
        if self._set is None:
            self._set = self._build()
            self._drop_arrays()
        return self._set

    @property
    def materialised(self) -> bool:
        return self._set is not None

    def _drop_arrays(self):
        pass

    # -- Set protocol ----------------------------------------------------------------------------------
    def __len__(self):
        if self._set is not None:
            return len(self._set)
        if self._size is None:
            self._size = self._count()
        return self._size

    def __iter__(self):
        return iter(self.as_set())

    def __contains__(self, x):
        return x in self.as_set()

    def add(self, x):
        self.as_set().add(x)

    def discard(self, x):
        self.as_set().discard(x)

    def __eq__(self, other):
        if isinstance(other, _LazySet):
            other = other.as_set()
        if not isinstance(other, (Set, set, frozenset)):
            return NotImplemented
        return self.as_set() == other

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    __hash__ = None

    @classmethod
    def _from_iterable(cls, it):                 # results of | & - ^ are plain sets, as with the reference's objects
        return set(it)

    def __ior__(self, other):
        self.as_set().update(other)
        return self

    def __ror__(self, other):                    # real_set | lazy  (and real_set |= lazy): a copy of the left operand, updated
        if isinstance(other, (set, frozenset)):
            r = set(other)
            r.update(self.as_set())
            return r
        return NotImplemented

    def update(self, *others):
        for o in others:
            self.__ior__(o)

    def copy(self):
        return set(self.as_set())

    def union(self, *others):
        return self.as_set().union(*others)

    def intersection(self, *others):
        return self.as_set().intersection(*others)

    def difference(self, *others):
        return self.as_set().difference(*others)

    def issubset(self, other):
        return self.as_set().issubset(other)

    def issuperset(self, other):
        return self.as_set().issuperset(other)

    def __repr__(self):
        return repr(self.as_set())

    def __reduce__(self):
        return (set, (list(self.as_set()),))


class ReadSupportSet(_LazySet):
    """``set`` of ``(read name, i, j)`` tuples given as chunks of (name id, i, j) arrays.

    Chunk 0 is what the reference passes to ``addbp`` as ``set(bpr)`` and stores; every later chunk arrived through
    ``bp[-1] |= set(bpr_)`` (ibg:330).  Materialisation repeats exactly that: ``s = set(tuples_0); s |= set(set(tuples_k))``.
    """
    __slots__ = ("_chunks",)

    def __init__(self, names, name_ids, i, j):
        self._names = names
        self._set = None
        self._size = None
        self._chunks = [(np.ascontiguousarray(name_ids, dtype=np.int64), np.ascontiguousarray(i, dtype=np.int64),
                         np.ascontiguousarray(j, dtype=np.int64))]

    def _tuples(self, chunk):
        return self._names.tuples(*chunk)

    def _build(self):
        s = set(self._tuples(self._chunks[0]))
        for c in self._chunks[1:]:
            s |= set(set(self._tuples(c)))
        return s

    def _drop_arrays(self):
        self._chunks = None

    def _count(self):
        if len(self._chunks) == 1:
            return _pyobjects.count_distinct3(*self._chunks[0])
        return _pyobjects.count_distinct3(*(np.concatenate([c[k] for c in self._chunks]) for k in range(3)))

    def __ior__(self, other):
        if self._set is None and isinstance(other, ReadSupportSet) and other._set is None and other._names is self._names:
            self._chunks += other._chunks
            self._size = None
            return self
        return super().__ior__(other)

    def name_ids(self):
        """Name ids of the supporting reads (with repeats), or None once the set has been materialised and possibly edited."""
        if self._set is not None:
            return None
        return np.concatenate([c[0] for c in self._chunks])


class ReadNameSet(_LazySet):
    """``set`` of read names given as the record ordinals of the two point fetches whose union it is (``rls | rrs``, ibg:1054)
    and the records' name ids: materialised as ``set(names of the first) | set(names of the second)``, each in fetch order."""
    __slots__ = ("_rec_name", "_left", "_right", "_keep")

    def __init__(self, names, rec_name, left_recs, right_recs, keep=None):
        self._names = names
        self._set = None
        self._size = None
        self._rec_name = rec_name
        self._left = left_recs
        self._right = right_recs
        self._keep = keep                 # owner of the memory the two record lists are views of

    def _ids(self, recs):
        return np.ascontiguousarray(self._rec_name[recs], dtype=np.int64)

    def _build(self):
        return set(self._names.take(self._ids(self._left))) | set(self._names.take(self._ids(self._right)))

    def _drop_arrays(self):
        self._left = self._right = self._rec_name = self._keep = None

    def _count(self):
        return len(self.name_ids())

    def name_ids(self):
        if self._set is not None:
            return None
        return np.union1d(self._ids(self._left), self._ids(self._right))
