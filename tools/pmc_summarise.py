"""Average the rocprofv3 --pmc rows per kernel (scan variants and probes only) -> small text table on stdout."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    files = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)
    if not files:
        print(d, "no counter_collection.csv")
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(files[0]) as fp:
        for r in csv.DictReader(fp):
            k = r["Kernel_Name"]
            if "scan" in k or "probe" in k or "seg_" in k or "point_cover" in k or "inflate" in k or "k_bam_" in k:
                agg[k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for tf in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
        with open(tf) as fp:
            for r in csv.DictReader(fp):
                k = r["Kernel_Name"]
                if "scan" in k or "probe" in k or "seg_" in k or "point_cover" in k or "inflate" in k or "k_bam_" in k:
                    agg[k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")]["~duration_us"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    names = sorted({c for k in agg for c in agg[k]})
    print("#", d)
    print("%-44s" % "kernel", " ".join("%26s" % c for c in names))
    for k in sorted(agg):
        print("%-44s" % k[:44], " ".join("%26.5g" % (sum(agg[k][c]) / len(agg[k][c])) for c in names))
