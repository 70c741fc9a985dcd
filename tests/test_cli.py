"""CLI surface of `reconstruct` (same flags as the reference's parser) + the full BAM -> graph path on the GPU."""
import json
import os
import subprocess
import sys

import pytest

from coral_amd import CoRAL, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parser_accepts_reference_flags():
    a = CoRAL.build_parser().parse_args(["reconstruct", "--lr_bam", "x.bam", "--cnv_seed", "s.bed", "--cn_seg", "c.bed",
                                         "--output_prefix", "o", "--skip_cycle_decomp", "--min_bp_support", "2.5",
                                         "--cycle_decomp_alpha", "0.05", "--cycle_decomp_time_limit", "10",
                                         "--cycle_decomp_threads", "4", "--postprocess_greedy_sol", "--log_fn", "l.log",
                                         "--output_all_path_constraints"])
    assert a.mode == "reconstruct" and a.min_bp_support == 2.5 and a.skip_cycle_decomp and not a.output_bp
    assert a.log_fn == "l.log" and a.cycle_decomp_threads == 4
    with pytest.raises(SystemExit):
        CoRAL.build_parser().parse_args(["reconstruct", "--lr_bam", "x.bam"])      # required flags, as the reference


@pytest.mark.gpu
def test_cli_reconstruct_from_bam_matches_golden(golden_dir, tmp_path):
    """BAM file on disk -> native decode -> HIP kernels -> *_graph.txt, compared with the reference's golden text."""
    from coral_amd import bam
    from tests.product_check import compare_graph_text
    cfg = synth.named_config("tiny")
    rec = synth.generate(cfg, "cpu")
    bam_path = str(tmp_path / "tiny.bam")
    bam.write_bam(rec, bam_path, seed=cfg.seed)
    cn, seeds = str(tmp_path / "cn.bed"), str(tmp_path / "seeds.bed")
    synth.write_cn_bed(cfg, cn)
    synth.write_seed_bed(cfg, seeds)
    prefix = str(tmp_path / "out")
    env = dict(os.environ, PYTHONHASHSEED="0")
    r = subprocess.run([sys.executable, "-m", "coral_amd.CoRAL", "reconstruct", "--lr_bam", bam_path, "--cnv_seed", seeds,
                        "--cn_seg", cn, "--output_prefix", prefix, "--skip_cycle_decomp", "--log_fn", str(tmp_path / "run.log")],
                       cwd=ROOT, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "Completed reconstruction." in r.stdout
    with open(os.path.join(golden_dir, "e2e_tiny.json")) as fp:
        gold = json.load(fp)
    for name, text in gold["files"].items():
        compare_graph_text(open(prefix + name[3:]).read(), text)
    log = open(tmp_path / "run.log").read()
    assert "LR normal cov" in log and "Wrote breakpoint graph" in log


def test_parser_accepts_reference_hsr_flags():
    a = CoRAL.build_parser().parse_args(["hsr", "--lr_bam", "x.bam", "--cycles", "c.bed", "--cn_seg", "cn.bed",
                                         "--output_prefix", "o", "--normal_cov", "31.5"])
    assert a.mode == "hsr" and a.normal_cov == "31.5" and a.bp_match_cutoff == 100 and a.bp_match_cutoff_clustering == 2000


@pytest.mark.gpu
def test_cli_hsr_from_bam_matches_golden(golden_dir, tmp_path):
    """BAM file on disk -> native decode -> K3 -> junction candidates -> stdout of the reference's hsr mode."""
    from coral_amd import bam
    from oracle.refharness.run_reference_hsr import hsr_inputs
    cfg, rec, ecdna = hsr_inputs("hsr_edge")
    bam_path = str(tmp_path / "x.bam")
    bam.write_bam(rec, bam_path, seed=cfg.seed)
    cn, cyc = str(tmp_path / "cn.bed"), str(tmp_path / "ecdna.bed")
    synth.write_cn_bed(cfg, cn)
    with open(cyc, "w") as fp:
        for c, s, e in ecdna:
            fp.write("%s\t%d\t%d\t+\t1\tTrue\t1.000000\n" % (c, s, e))
    r = subprocess.run([sys.executable, "-m", "coral_amd.CoRAL", "hsr", "--lr_bam", bam_path, "--cycles", cyc, "--cn_seg", cn,
                        "--output_prefix", "golden", "--normal_cov", "4"], cwd=str(tmp_path),
                       env=dict(os.environ, PYTHONHASHSEED="0", PYTHONPATH=ROOT), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    with open(os.path.join(golden_dir, "hsr_hsr_edge_4.json")) as fp:
        gold = json.load(fp)
    assert r.stdout.endswith(gold["stdout"])
    assert os.path.exists(tmp_path / "integration_sites_golden.png")
