"""Constants shared by the graph-build path (same role as /root/reference/src/global_names.py)."""
TSTART = 0

neg_plus_minus = {"+": "-", "-": "+"}

_CHROMS = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY", "chrM"]
chr_idx = {c: i for i, c in enumerate(_CHROMS)}
