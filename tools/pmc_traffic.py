"""Summarise the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh into profiles-ready JSON + markdown.

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact."""
import collections, csv, glob, json, sys

fetch_dir, write_dir, n_reads, records, alg_bytes, out_json, out_md = sys.argv[1:8]
workload = sys.argv[8] if len(sys.argv) > 8 else "cfg3"
library = sys.argv[9] if len(sys.argv) > 9 else None          # coral_version(): carries the hash of coral_kernels.hip


def collect(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        with open(f) as fp:
            for r in csv.DictReader(fp):
                k = r["Kernel_Name"]
                if r["Counter_Name"] == counter and ("k_cigar_scan" in k or "k_seg_" in k or "k_point_cover" in k):
                    acc[k.split("(")[0].replace("void ", "").strip()].append(float(r["Counter_Value"]))
    return acc


def durations(d):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
        with open(f) as fp:
            for r in csv.DictReader(fp):
                k = r["Kernel_Name"]
                if "k_cigar_scan" in k or "k_seg_" in k or "k_point_cover" in k:
                    acc[k.split("(")[0].replace("void ", "").strip()].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return acc


F, W, D = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE"), durations(fetch_dir)
kernels = {}
for k in sorted(F):
    f = sum(F[k]) / len(F[k])
    w = sum(W[k]) / len(W[k]) if k in W else 0.0
    kernels[k] = dict(launches=len(F[k]), fetch_kib=f, write_kib=w, hbm_bytes=2 * f * 1024 + w * 1024,
                      avg_ns_under_pmc=sum(D[k]) / len(D[k]) if k in D else None)
scan = [k for k in kernels if k.startswith("k_cigar_scan")]
doc = dict(workload=workload, library=library, n_reads=int(n_reads), records=int(records), algorithmic_bytes_per_scan_launch=int(alg_bytes),
           scan_kernel=scan[0] if scan else None, kernels=kernels,
           method="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (with --kernel-trace only); "
                  "hbm_bytes = 2 * FETCH_SIZE KiB * 1024 + WRITE_SIZE KiB * 1024 (gfx950 FETCH_SIZE halves wide streaming reads)")
json.dump(doc, open(out_json, "w"), indent=1)
with open(out_md, "w") as fp:
    fp.write("# PMC traffic — rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes with --kernel-trace only\n")
    fp.write("Target: `python tools/pmc_target.py %s %s` (%s reads, %s records, algorithmic bytes per scan launch %s); library `%s`.\n" % (n_reads, workload, n_reads, records, alg_bytes, library))
    fp.write("FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950 counts 128-B streaming requests as 64 B), WRITE_SIZE is exact.\n\n")
    fp.write("| kernel | launches | FETCH_SIZE KiB (avg) | WRITE_SIZE KiB (avg) | HBM bytes / launch | avg ns (under PMC) |\n|---|---|---|---|---|---|\n")
    for k, v in kernels.items():
        fp.write("| `%s` | %d | %.1f | %.1f | %.4g | %s |\n" % (k, v["launches"], v["fetch_kib"], v["write_kib"], v["hbm_bytes"],
                                                              "%.0f" % v["avg_ns_under_pmc"] if v["avg_ns_under_pmc"] else "-"))
    if scan:
        hb = kernels[scan[0]]["hbm_bytes"]
        fp.write("\n`%s`: HBM traffic %.3f GB per launch vs %.3f GB algorithmic => traffic / algorithmic = %.3f\n"
                 % (scan[0], hb / 1e9, int(alg_bytes) / 1e9, hb / int(alg_bytes)))
print(open(out_md).read())
