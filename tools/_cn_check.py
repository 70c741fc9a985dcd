import os, sys, time, tempfile
sys.path.insert(0, ".")
import numpy as np, torch
from coral_amd import synth, sharding
from coral_amd.breakpoint_graph import cn_problem, solve_cn_lr
from oracle import coral_oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
cfg = synth.scaled_config("cfg3", n)
work = tempfile.mkdtemp()
cn, seeds = os.path.join(work, "cn.bed"), os.path.join(work, "seeds.bed")
synth.write_cn_bed(cfg, cn); synth.write_seed_bed(cfg, seeds)
rec = synth.generate(cfg, "cuda:0", chunk_pieces=200000)
dr = sharding.shard_records(rec, 0, 1, "cuda:0")
b = sharding.build_graph_sharded(dr, seeds, cn, os.path.join(work, "gpu"))
np.set_printoptions(precision=10, linewidth=200)
for g in b.lr_graph:
    w_inv, w_lin, w_log, A = cn_problem(g, b.normal_cov)
    print("normal_cov", b.normal_cov, "vars", len(w_lin), "rows", A.shape[0], "rank", np.linalg.matrix_rank(A))
    def kkt(x):
        # multiplier by least squares; report stationarity + feasibility
        gvec = w_lin - w_log / x - w_inv / (x * x)
        nu = np.linalg.lstsq(A.T, -gvec, rcond=None)[0]
        return np.linalg.norm(gvec + A.T @ nu) / np.linalg.norm(gvec), np.linalg.norm(A @ x)
    def obj(x):
        return float(np.sum(w_inv / x + w_lin * x - w_log * np.log(x)))
    for it in (200, 2000):
        xp = solve_cn_lr(w_inv, w_lin, w_log, A, max_iter=it)
        xo = O.solve_cn(w_inv, w_lin, w_log, A.astype(float), max_iter=it)
        print("max_iter", it, "| product: rel stationarity %.3e feas %.3e obj %.12e | oracle: %.3e %.3e %.12e | max rel diff %.3e" % (
            *kkt(xp), obj(xp), *kkt(xo), obj(xo), float(np.max(np.abs(xp - xo) / xo))))
    print("x product[:8]", xp[:8] * 2)
    print("x oracle [:8]", xo[:8] * 2)
