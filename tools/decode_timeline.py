"""GPU timeline of a BAM decode from a rocprofv3 --kernel-trace CSV: where the wall time of the pipeline goes.

    python tools/decode_timeline.py <dir with *_kernel_trace.csv> [out.md]

Groups the kernels into decode runs (a run = a maximal sequence of k_bgzf_inflate launches less than 50 ms apart), and for every
run reports: wall span, the union of all kernel intervals (GPU busy), the union of the inflate kernels alone, time with NO kernel
running (bubbles), per-kernel totals, and the largest gaps with the kernels on either side of them."""
import csv
import glob
import sys

d = sys.argv[1]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
rows = []
for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]))
rows.sort()
infl = [r for r in rows if "k_bgzf_inflate" in r[2]]
runs, cur = [], []
for r in infl:
    if cur and r[0] - cur[-1][1] > 5e7:
        runs.append(cur)
        cur = []
    cur.append(r)
if cur:
    runs.append(cur)


def union(iv):
    tot, end = 0, None
    for s, e in sorted(iv):
        if end is None or s > end:
            tot += e - s
            end = e
        elif e > end:
            tot += e - end
            end = e
    return tot


P = lambda *a: print(*a, file=out)
P("# GPU timeline of the BAM decode (rocprofv3 --kernel-trace)\n")
for k, run in enumerate(runs):
    t0, t1 = run[0][0], max(r[1] for r in run)
    inside = [r for r in rows if r[0] >= t0 - 5e6 and r[0] <= t1 + 2e8 and any(t in r[2] for t in ("k_bgzf", "k_bam", "DeviceScan", "Cat", "cat", "copy"))]
    t1 = max(r[1] for r in inside)
    span = (t1 - t0) / 1e6
    busy = union([(r[0], r[1]) for r in inside]) / 1e6
    ib = union([(r[0], r[1]) for r in run]) / 1e6
    P("## decode run %d: %d batches, first inflate to last kernel %.1f ms; some kernel running %.1f ms (%.0f %%), inflate running %.1f ms, NO kernel "
      "running %.1f ms\n" % (k, len(run), span, busy, 100 * busy / span, ib, span - busy))
    tot = {}
    for r in inside:
        tot.setdefault(r[2], [0, 0.0])
        tot[r[2]][0] += 1
        tot[r[2]][1] += (r[1] - r[0]) / 1e6
    P("| kernel | calls | total ms | avg ms |\n|---|---|---|---|")
    for name, (n, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:14]:
        P("| `%s` | %d | %.1f | %.3f |" % (name[:70], n, ms, ms / n))
    # gaps between consecutive kernels (no kernel running), largest first
    ev = sorted((r[0], r[1], r[2]) for r in inside)
    gaps, end, last = [], ev[0][1], ev[0][2]
    for s, e, nm in ev[1:]:
        if s > end:
            gaps.append(((s - end) / 1e6, last, nm))
        if e > end:
            end, last = e, nm
    gaps.sort(reverse=True)
    P("\nidle gaps: %d, total %.1f ms; by the kernel that FOLLOWS the gap:" % (len(gaps), sum(g[0] for g in gaps)))
    by = {}
    for g, a, b in gaps:
        by.setdefault(b[:40], [0, 0.0])
        by[b[:40]][0] += 1
        by[b[:40]][1] += g
    for nm, (n, ms) in sorted(by.items(), key=lambda kv: -kv[1][1])[:8]:
        P("  before `%s`: %d gaps, %.1f ms" % (nm, n, ms))
    P("largest: " + "; ".join("%.2f ms (%s -> %s)" % (g, a[:24], b[:24]) for g, a, b in gaps[:6]))
    d_inf = [(r[1] - r[0]) / 1e6 for r in run]
    P("\ninflate launches: min %.2f / median %.2f / max %.2f ms; the first five %s, the last five %s\n" % (
        min(d_inf), sorted(d_inf)[len(d_inf) // 2], max(d_inf), ["%.1f" % x for x in d_inf[:5]], ["%.1f" % x for x in d_inf[-5:]]))
    # the parse kernels one batch at a time: (start relative to the run's first inflate, duration), and which inflate launch was
    # running when they started — a kernel that shares the chip with an inflate launch is slower than it is alone
    for want in ("k_bam_find", "k_bam_verify", "k_bgzf_crc"):
        ks = sorted(r for r in inside if r[2].startswith(want))
        if not ks:
            continue
        P("`%s` per batch (start ms, duration ms, [overlapping inflate: its start, duration]); first 8 and last 3:" % want)
        for r in ks[:8] + ks[-3:]:
            ov = [i for i in run if i[0] < r[1] and i[1] > r[0]]
            P("  %8.1f  %6.2f   %s" % ((r[0] - t0) / 1e6, (r[1] - r[0]) / 1e6, "; ".join("inflate @%.1f %.1f ms" % ((i[0] - t0) / 1e6, (i[1] - i[0]) / 1e6) for i in ov) or "alone"))
