"""Canonical JSON-able form shared by golden files and the objects under test."""
import hashlib
import json
from collections.abc import Set

import numpy as np


def canon(o):
    """sets -> tagged sorted lists, dicts -> tagged ordered pair lists, tuples -> lists."""
    if isinstance(o, (set, frozenset, Set)):           # incl. the lazily materialised sets of coral_amd.lazysets
        return {"__set__": sorted((canon(x) for x in o), key=lambda v: json.dumps(v))}
    if isinstance(o, dict):
        return {"__dict__": [[canon(k), canon(v)] for k, v in o.items()]}
    if isinstance(o, (list, tuple)):
        return [canon(x) for x in o]
    if isinstance(o, np.bool_):
        return bool(o)
    if isinstance(o, np.integer):
        return int(o)
    if isinstance(o, np.floating):
        return float(o)
    return o


def uncanon_unit(o):
    """Inverse used for unit_vectors.json (its tuples are tagged so they can be rebuilt)."""
    if isinstance(o, dict):
        if "__tuple__" in o:
            return tuple(uncanon_unit(x) for x in o["__tuple__"])
        if "__set__" in o:
            return set(uncanon_unit(x) for x in o["__set__"])
        if "__dict__" in o:
            return {uncanon_unit(k): uncanon_unit(v) for k, v in o["__dict__"]}
        return {k: uncanon_unit(v) for k, v in o.items()}
    if isinstance(o, list):
        return [uncanon_unit(x) for x in o]
    return o


def graph_snapshot(g):
    return dict(sequence_edges=canon(g.sequence_edges), concordant_edges=canon(g.concordant_edges),
                discordant_edges=canon(g.discordant_edges), source_edges=canon(g.source_edges),
                nodes=[[list(k), canon(v)] for k, v in g.nodes.items()],
                endnodes=[[list(k), canon(v)] for k, v in g.endnodes.items()],
                amplicon_intervals=canon(g.amplicon_intervals), max_cn=g.max_cn)


def records_digest(rec) -> str:
    h = hashlib.sha256()
    for k in ("tid", "pos", "end", "flag", "mapq", "qlen", "has_seq", "nm", "name_id", "n_cigar", "cigar_off",
              "cigar", "sa_off", "sa", "sa_nm", "nonacgt_rec", "nonacgt_pos", "name_gid"):
        h.update(getattr(rec, k).cpu().numpy().tobytes())
    return h.hexdigest()


def strip_cn(snapshot):
    """Graph snapshot with every float CN field removed (compared separately with a tolerance)."""
    s = json.loads(json.dumps(snapshot))
    cns = []
    for key in ("sequence_edges", "concordant_edges", "discordant_edges", "source_edges"):
        for e in s[key]:
            cns.append(e[-1])
            e[-1] = None
    cns.append(s["max_cn"])
    s["max_cn"] = None
    return s, cns
