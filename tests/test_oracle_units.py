"""Oracle unit functions vs known-answer vectors obtained from the reference's own cp/bu modules."""
import json
import os

import pytest

from oracle import coral_oracle as O
from tests.canon import canon, uncanon_unit


@pytest.fixture(scope="module")
def vec(golden_dir):
    with open(os.path.join(golden_dir, "unit_vectors.json")) as fp:
        return json.load(fp)


def test_survey_known_answers():
    # SURVEY.md §8(c) vectors obtained from the imported reference
    assert O.cigar2pos("2000S4990M30D3000S", "-", 10000) == (3000, 7999, 5020)
    out = O.alignment_from_satags(["chr8,127000001,+,5000S10000M,60,12", "chr8,128000001,-,10000S5000M,60,7"], 15000)
    assert out[0] == [[0, 4999], [5000, 14999]]
    assert out[1] == [["chr8", 128004999, 128000000, "-"], ["chr8", 127000000, 127009999, "+"]]
    assert O.interval2bp(["chr8", 127000000, 127009999, "+"], ["chr8", 128004999, 128000000, "-"], ("r1", 0, 1), 0) == \
        ["chr8", 128004999, "+", "chr8", 127009999, "+", ("r1", 1, 0), 0, 1]


def test_cigar2pos(vec):
    for v in vec["cigar2pos"]:
        assert list(O.cigar2pos(v["cigar"], v["strand"], v["read_length"])) == v["out"], v


def test_cigar2pos_unknown_shape_raises():
    with pytest.raises(KeyError):
        O.cigar2pos("10S5M3S2M", "+", 100)


def test_alignment_from_satags(vec):
    n_fail = 0
    for v in vec["alignment_from_satags"]:
        got = O.alignment_from_satags(list(v["sa_list"]), v["read_length"])
        assert got == uncanon_unit(v["out"]), v
        n_fail += len(got) == 3
    assert n_fail > 0          # the 3-tuple failure shape is exercised


def test_interval_predicates(vec):
    for v in vec["interval_predicates"]:
        assert bool(O.interval_overlap(v["a"], v["b"])) == v["overlap"]
        assert bool(O.interval_include(v["a"], v["b"])) == v["include"]
        assert bool(O.interval_adjacent(v["a"], v["b"])) == v["adjacent"]
    for v in vec["interval_exclusive"]:
        ov, rem = O.interval_exclusive(v["a"], v["L"])
        assert sorted(ov) == v["overlap_ints"] and rem == v["remaining"]


def test_interval2bp(vec):
    for v in vec["interval2bp"]:
        got = O.interval2bp(v["R1"], v["R2"], tuple(v["r"]), v["rgap"])
        assert got == uncanon_unit(v["out"])


def test_alignment2bp(vec):
    n = 0
    for k, v in enumerate(vec["alignment2bp"]):
        ca = uncanon_unit(v["ca"])
        got = O.alignment2bp("rd%d" % k, ca, 100, 20, v["i1"][:3], v["i2"])
        assert got == uncanon_unit(v["out"])
        n += len(got)
    assert n > 10


def test_alignment2bp_l(vec):
    n = 0
    for k, v in enumerate(vec["alignment2bp_l"]):
        ca = uncanon_unit(v["ca"])
        got = O.alignment2bp_l("rd%d" % k, ca, 100, 20, 100, v["intervals"])
        assert got == uncanon_unit(v["out"])
        n += len(got)
    assert n > 10


def test_cluster_and_bpc2bp(vec):
    for v in vec["cluster_bp_list"]:
        got = O.cluster_bp_list(uncanon_unit(v["bp_list"]), v["min_cluster_size"], v["cutoff"])
        assert got == uncanon_unit(v["out"])
    for v in vec["bpc2bp"]:
        bp, bpr, st, rest = O.bpc2bp(uncanon_unit(v["cluster"]), v["cutoff"])
        assert bp == uncanon_unit(v["bp"])
        assert bpr == uncanon_unit(v["bpr"])
        assert st == v["stats"]
        assert rest == uncanon_unit(v["rest"])


def test_bp_match(vec):
    seen = set()
    for v in vec["bp_match"]:
        got = bool(O.bp_match(v["bp1"], v["bp2"], v["rgap"], v["cutoff"]))
        assert got == v["out"]
        seen.add(got)
    assert seen == {True, False}
