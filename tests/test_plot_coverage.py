"""Coverage track of the plot mode against goldens from the REAL reference
(plot_amplicons.graph_vis.plot_graph run by oracle/refharness/run_reference_plotcov.py)."""
import json
import os

import pytest

from coral_amd import synth
from oracle import plotcov_oracle
from oracle.hostrecords import HostRecords
from oracle.refharness.run_reference import records_digest
from tests.product_check import install_cpu_kernel_fakes

CASES = ["tiny", "tiny_region", "tiny_edge_region", "ultra"]


def _load(golden_dir, case):
    with open(os.path.join(golden_dir, "plotcov_%s.json" % case)) as fp:
        gold = json.load(fp)
    cfg, rec = synth.dataset(gold["config"], "cpu")
    assert records_digest(rec) == gold["records_sha256"]
    with open(os.path.join(golden_dir, "e2e_%s.json" % gold["config"])) as fp:
        text = json.load(fp)["files"][gold["graph_file"]]
    bounds = None
    if gold["region"]:
        c, r = gold["region"].split(":")
        bounds = (c, int(r.split("-")[0]), int(r.split("-")[1]))
    want = [(c, a + k * w, a + k * w + w, tot) for c, a, w, totals in gold["tracks"] for k, tot in enumerate(totals)]
    assert len(want) == gold["n_windows"]
    return gold, rec, text, bounds, want


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_reference(case, golden_dir):
    gold, rec, text, bounds, want = _load(golden_dir, case)
    assert plotcov_oracle.intervals_from_graph_text(text) == gold["intervals_from_graph"]
    assert plotcov_oracle.coverage_track(HostRecords(rec), text, bounds) == want


def _product(rec, text, bounds, device, tmp_path):
    from coral_amd import plot_coverage
    from coral_amd.records import DeviceRecords
    fn = str(tmp_path / "g_graph.txt")
    with open(fn, "w") as fp:
        fp.write(text)
    iv = plot_coverage.parse_graph_intervals(fn)
    return iv, plot_coverage.coverage_track(DeviceRecords(rec, device), iv, bounds)


@pytest.mark.parametrize("case", ["tiny", "tiny_region"])       # the stand-in kernel is slow per window; all cases run with -m gpu
def test_product_host_logic_matches_reference(case, golden_dir, tmp_path, monkeypatch):
    install_cpu_kernel_fakes(monkeypatch)
    gold, rec, text, bounds, want = _load(golden_dir, case)
    iv, got = _product(rec, text, bounds, "cpu", tmp_path)
    assert iv == gold["intervals_from_graph"] and got == want


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_product_on_gpu_matches_reference(case, golden_dir, tmp_path):
    """All windows of a plot in one coral_segment_coverage launch, equal to the reference's per-window pysam calls."""
    gold, rec, text, bounds, want = _load(golden_dir, case)
    iv, got = _product(rec, text, bounds, "cuda:0", tmp_path)
    assert iv == gold["intervals_from_graph"] and got == want
