"""Synthetic long-read amplicon data (SURVEY.md §8(d) input model).

Everything here is *input construction* for tests, golden generation and
``bench.py`` — it is not on the graded path.  The generator is a pure function of
``(SynthConfig, seed)``: all randomness comes from a counter-based 32-bit
integer hash evaluated with int64 tensor arithmetic that never overflows, so the
CPU (golden fixtures) and the GPU (bench-size data) produce bit-identical
records.

Model (one "read" = one query name):
  * amplicons are circles of reference segments ``(tid, start, end, strand)``;
    a read walks ``W`` circle bases from a random offset and is cut into one
    alignment record per segment it touches (chimeric reads, SA tags restricted
    to the nine S/M/I/D shapes of the reference's SA parser,
    /root/reference/src/cigar_parsing.py:219-229);
  * background reads fall uniformly in per-chromosome windows;
  * each record's CIGAR is ``M`` runs broken by short I/D events on a jittered
    20-bp grid (≈0.1 ops/base); planted and random large deletions become one
    long ``D`` op inside a record (exercises
    /root/reference/src/infer_breakpoint_graph.py:750-762).

The hard input constraints of SURVEY.md §8(d)(i)–(viii) are honoured by
construction (enough CN≈2 tiles, NM varies, contigs are chr1..22,X,Y,M, ...).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

CHROMS = [f"chr{i}" for i in range(1, 23)] + ["chrX", "chrY", "chrM"]
CHR_SIZES = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
             138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
             83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415, 16569]

# BAM CIGAR op codes
OP_M, OP_I, OP_D, OP_N, OP_S, OP_H, OP_P, OP_EQ, OP_X = range(9)
OP_PAD = 15          # layout padding op used by the SoA store (consumes nothing)

# hash streams
S_KIND, S_LEN, S_LENJ, S_DIR, S_CIRCLE, S_START, S_CARRIER, S_NOISE, S_NOISE_LEN, S_NOISE_POS, \
    S_EVJ, S_EVT, S_EVL, S_MAPQ, S_MAPQV, S_NM, S_NBASE, S_JIT, S_SEC, S_CN, S_SEQ = range(21)

_EV_LENS = (1, 1, 1, 1, 2, 2, 3, 5)
EV_SPACING = 20


# --------------------------------------------------------------------------------------
# counter-based hash (murmur3 fmix32 evaluated in int64 without overflow)
# --------------------------------------------------------------------------------------
def _mul32(x: torch.Tensor, c: int) -> torch.Tensor:
    """(x * c) mod 2^32 for 0 <= x < 2^32, 0 <= c < 2^32 using only < 2^63 intermediates."""
    lo = (x & 0xFFFF) * c
    hi = (((x >> 16) * c) & 0xFFFF) << 16
    return (lo + hi) & 0xFFFFFFFF


def _fmix32(x: torch.Tensor) -> torch.Tensor:
    x = x ^ (x >> 16)
    x = _mul32(x, 0x85EBCA6B)
    x = x ^ (x >> 13)
    x = _mul32(x, 0xC2B2AE35)
    x = x ^ (x >> 16)
    return x


def hash_u32(seed: int, stream: int, idx: torch.Tensor) -> torch.Tensor:
    """Uniform 32-bit value (as int64 in [0, 2^32)) for every int64 key in ``idx`` (idx >= 0)."""
    h = (seed * 0x9E3779B1 + stream * 0x85EBCA77 + 0x165667B1) & 0xFFFFFFFF
    lo = idx & 0xFFFFFFFF
    hi = (idx >> 32) & 0xFFFFFFFF
    x = _fmix32(lo ^ h)
    x = x ^ _mul32((hi + 0x27D4EB2F) & 0xFFFFFFFF, 0x9E3779B1)
    return _fmix32(x)


def _bounded(u: torch.Tensor, n) -> torch.Tensor:
    """Map 32-bit uniform ``u`` to [0, n) (n < 2^31, tensor or int)."""
    return (u * n) >> 32


def _thresh(frac: float) -> int:
    return int(max(0.0, min(1.0, frac)) * 4294967296.0)


# --------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------
@dataclass
class Segment:
    tid: int
    start: int   # 0-based inclusive
    end: int     # exclusive
    strand: int  # 0 '+', 1 '-'

    def __len__(self):
        return self.end - self.start


@dataclass
class SynthConfig:
    name: str
    n_reads: int
    mean_len: int
    seed: int
    windows: List[Tuple[int, int, int]]              # (tid, start, end) background windows
    circles: List[List[Segment]]
    circle_weight: List[float]
    seeds: List[Tuple[int, int, int]]                # (tid, start, end) inclusive ends, as the seed bed
    planted: List[Tuple[int, int, int, float]] = field(default_factory=list)  # (tid, start, length, carrier frac)
    sigma_log: float = 0.5
    min_len: int = 1000
    max_len: int = 400000
    amp_frac: float = 0.85
    cn_tile: int = 250000
    min_piece: int = 300
    noise_del_rate: float = 0.002
    lowmapq_frac: float = 0.03
    midmapq_frac: float = 0.02
    bg_lowmapq_frac: float = 0.02
    nbase_frac: float = 0.003
    secondary_frac: float = 0.004

    def total_circle_len(self) -> int:
        return sum(sum(len(s) for s in c) for c in self.circles)

    def total_window_len(self) -> int:
        return sum(e - s for _, s, e in self.windows)


def _pyhash(seed: int, stream: int, i: int) -> int:
    return int(hash_u32(seed, stream, torch.tensor([i], dtype=torch.int64))[0])


def build_config(name: str, n_reads: int, mean_len: int, seed: int, chrom_tids: Sequence[int],
                 n_circles: int, segs_per_circle: int, seg_len: Tuple[int, int], n_seeds: int,
                 window_len: int = 12_000_000, n_planted: int = 3, inverted_frac: float = 0.3,
                 planted_frac: float = 0.25, **kw) -> SynthConfig:
    """Lay out windows, amplicon circles, seeds and planted deletions deterministically."""
    windows = []
    for k, tid in enumerate(chrom_tids):
        centre = (CHR_SIZES[tid] // 2 // 1_000_000 + 7 * k + 3) * 1_000_000
        ws = max(1_000_000, centre - window_len // 2)
        ws -= ws % 250_000
        windows.append((tid, ws, ws + window_len))
    # candidate disjoint slots inside the central part of every window
    n_segs = n_circles * segs_per_circle
    per_win = -(-n_segs // len(windows))
    slots: List[Segment] = []
    ctr = 0
    for (tid, ws, we) in windows:
        lo = ws + window_len // 4
        hi = we - window_len // 4
        pitch = (hi - lo) // per_win
        assert pitch > seg_len[1] + 20_000, "segments do not fit the window; widen window_len"
        for j in range(per_win):
            ln = seg_len[0] + _pyhash(seed, 101, ctr) % (seg_len[1] - seg_len[0] + 1)
            off = 5_000 + _pyhash(seed, 102, ctr) % (pitch - ln - 10_000)
            st = lo + j * pitch + off
            strand = 1 if (_pyhash(seed, 103, ctr) % 1000) < inverted_frac * 1000 else 0
            slots.append(Segment(tid, st, st + ln, strand))
            ctr += 1
    # deterministic shuffle, deal to circles
    order = sorted(range(len(slots)), key=lambda i: _pyhash(seed, 104, i))
    slots = [slots[i] for i in order][:n_segs]
    circles = [slots[c * segs_per_circle:(c + 1) * segs_per_circle] for c in range(n_circles)]
    weights = [float(sum(len(s) for s in c)) for c in circles]
    # seeds: pick segments round-robin over circles, trimmed 10 % inward
    seeds = []
    k = 0
    while len(seeds) < n_seeds:
        c = circles[k % n_circles]
        s = c[(k // n_circles) % len(c)]
        trim = len(s) // 10
        cand = (s.tid, s.start + trim, s.end - trim - 1)
        if cand not in seeds:
            seeds.append(cand)
        k += 1
        if k > 10 * n_segs:
            break
    seeds.sort()
    # planted deletions: inside the longest segments, 1-5 kb, away from the ends
    planted = []
    by_len = sorted(slots, key=lambda s: -len(s))
    for j in range(min(n_planted, len(by_len))):
        s = by_len[j]
        dl = 1000 + _pyhash(seed, 105, j) % 4000
        st = s.start + len(s) // 3 + _pyhash(seed, 106, j) % max(1, len(s) // 3 - dl)
        planted.append((s.tid, st, dl, planted_frac + 0.1 * (j % 3)))
    return SynthConfig(name=name, n_reads=n_reads, mean_len=mean_len, seed=seed, windows=windows,
                       circles=circles, circle_weight=weights, seeds=seeds, planted=planted, **kw)


def named_config(name: str) -> SynthConfig:
    """Configurations used by tests, goldens and bench (BASELINE.json `configs`)."""
    if name == "tiny":       # golden fixture: one circle, chr8 only
        return build_config("tiny", 4000, 4000, 11, [7], 1, 4, (150_000, 260_000), 1,
                            window_len=20_000_000, n_planted=2, amp_frac=0.62, min_len=600, planted_frac=0.5,
                            inverted_frac=0.5)
    if name == "small":      # golden fixture: two circles over chr7/chr8/chr12, inter-chromosomal junctions
        return build_config("small", 9000, 5000, 12, [6, 7, 11], 2, 5, (120_000, 300_000), 3,
                            window_len=12_000_000, n_planted=3, amp_frac=0.6, min_len=600, planted_frac=0.45)
    if name == "ultra":      # golden fixture with long reads / many pieces per read
        return build_config("ultra", 1500, 40000, 15, [7, 11], 1, 8, (60_000, 110_000), 2,
                            window_len=14_000_000, n_planted=2, amp_frac=0.5, min_len=3000, planted_frac=0.5)
    if name == "cfg1":       # 50k reads @15 kb, 1 seed (chr8 MYC-like, 2 Mb) — CPU-runnable config
        return build_config("cfg1", 50_000, 15_000, 1001, [7], 1, 5, (300_000, 480_000), 1,
                            window_len=24_000_000, n_planted=3)
    if name == "cfg2":       # 500k reads, 3 seeds on chr8
        return build_config("cfg2", 500_000, 20_000, 1002, [7], 1, 12, (250_000, 400_000), 3,
                            window_len=28_000_000, n_planted=4)
    if name == "cfg3":       # headline: 2M reads x 20 kb, 10 seeds over chr7/chr8/chr12
        return build_config("cfg3", 2_000_000, 20_000, 1003, [6, 7, 11], 3, 8, (160_000, 260_000), 10,
                            window_len=12_000_000, n_planted=5)
    if name == "cfg5":       # ultra-long: 200k reads x 100 kb, 30 % chimeric
        return build_config("cfg5", 200_000, 100_000, 1005, [7, 11], 1, 14, (250_000, 400_000), 3,
                            window_len=16_000_000, n_planted=4, min_len=5000)
    raise KeyError(name)


def scaled_config(name: str, n_reads: int) -> SynthConfig:
    """Same layout as ``name`` with a different read count (bench subsamples / smoke)."""
    cfg = named_config(name)
    cfg.n_reads = n_reads
    return cfg


# --------------------------------------------------------------------------------------
# CN segment file / seed file
# --------------------------------------------------------------------------------------
def cn_segments(cfg: SynthConfig) -> List[Tuple[str, int, int, float]]:
    """250-kb tiles over every window: background CN ≈ 2 (with ties), amplified tiles carry their CN.

    Rows are (chrom, start, end_exclusive, cn) in the `.bed` flavour read at
    /root/reference/src/infer_breakpoint_graph.py:96-98.
    """
    amp_depth = cfg.amp_frac * cfg.n_reads * cfg.mean_len / max(1, cfg.total_circle_len())
    bg_depth = (1 - cfg.amp_frac) * cfg.n_reads * cfg.mean_len / max(1, cfg.total_window_len())
    amp_cn = 2.0 * amp_depth / bg_depth
    rows = []
    k = 0
    for (tid, ws, we) in sorted(cfg.windows):
        for ts in range(ws, we, cfg.cn_tile):
            te = min(we, ts + cfg.cn_tile)
            ov = 0
            for c in cfg.circles:
                for s in c:
                    if s.tid == tid:
                        ov += max(0, min(te, s.end) - max(ts, s.start))
            cn = 2.0 + (_pyhash(cfg.seed, S_CN, k) % 21 - 10) * 0.01
            if ov > 0:
                cn = round(2.0 + amp_cn * ov / (te - ts), 3)
            rows.append((CHROMS[tid], ts, te, cn))
            k += 1
    return rows


def write_cn_bed(cfg: SynthConfig, path: str) -> None:
    with open(path, "w") as fp:
        for c, s, e, cn in cn_segments(cfg):
            fp.write(f"{c}\t{s}\t{e}\t{cn}\n")


def write_cn_cns(cfg: SynthConfig, path: str) -> None:
    """The same segments in cnvkit's ``.cns`` flavour (ibg:94-95): header line, log2 ratio in column 5, CN = 2 * 2**log2."""
    with open(path, "w") as fp:
        fp.write("chromosome\tstart\tend\tgene\tlog2\tdepth\tprobes\tweight\n")
        for c, s, e, cn in cn_segments(cfg):
            fp.write(f"{c}\t{s}\t{e}\t-\t{np.log2(cn / 2.0):.6f}\t{cn * 10:.3f}\t{(e - s) // 1000}\t1.0\n")


def write_seed_bed(cfg: SynthConfig, path: str) -> None:
    with open(path, "w") as fp:
        for tid, s, e in cfg.seeds:
            fp.write(f"{CHROMS[tid]}\t{s}\t{e}\n")


# --------------------------------------------------------------------------------------
# the record store produced by the generator (and by the BAM decoder)
# --------------------------------------------------------------------------------------
@dataclass
class Records:
    """Structure-of-arrays alignment records in BAM (tid, pos) order.

    All tensors live on one device.  ``cigar`` holds BAM-packed ops (len << 4 | op); every record's
    ops start at a multiple of 4 (16-byte aligned) and are padded with ``OP_PAD``.
    SA rows (one per SA entry of every record that has an SA tag) are the numeric tokenisation of
    ``rname,pos,strand,CIGAR,mapQ,NM`` where the CIGAR is ``[c5 S] m M [x I|D] [c3 S]``.
    """
    n: int
    tid: torch.Tensor        # i32
    pos: torch.Tensor        # i32 0-based
    end: torch.Tensor        # i32 htslib bam_endpos
    flag: torch.Tensor       # i32
    mapq: torch.Tensor       # i32
    qlen: torch.Tensor       # i32 pysam query_length (l_seq, or CIGAR-implied when SEQ is '*')
    has_seq: torch.Tensor    # i32 0/1
    nm: torch.Tensor         # i32 NM tag (0 when absent)
    name_id: torch.Tensor    # i32 index into names (first-appearance order)
    n_cigar: torch.Tensor    # i32 real op count
    cigar_off: torch.Tensor  # i64 [n+1] offsets in ops (multiples of 4)
    cigar: torch.Tensor      # i32 packed ops (bit pattern of BAM u32)
    sa_off: torch.Tensor     # i64 [n+1] offsets into SA rows
    sa: torch.Tensor         # i32 [n_sa, 8]: tid, pos1 (1-based), strand(0/1), c5, m, x(+I / -D), c3, mapq ; nm kept apart
    sa_nm: torch.Tensor      # i32 [n_sa]
    nonacgt_rec: torch.Tensor  # i64 record index of every aligned non-ACGT base
    nonacgt_pos: torch.Tensor  # i32 reference position of that base
    n_names: int
    name_gid: Optional[torch.Tensor] = None   # i64 [n_names] synthetic read id behind every name id
    names: Optional[List[str]] = None         # materialised lazily for synthetic data
    header_chroms: List[str] = field(default_factory=lambda: list(CHROMS))
    header_lens: List[int] = field(default_factory=lambda: list(CHR_SIZES))

    def name_of(self, nid: int) -> str:
        if self.names is not None:
            return self.names[nid]
        return "read%08d" % int(self.name_gid[nid])

    def materialise_names(self) -> List[str]:
        if self.names is None:
            g = self.name_gid.cpu().numpy()
            self.names = ["read%08d" % int(x) for x in g]
        return self.names

    def name_table(self):
        """The names as ``coral_amd.names.NameTable`` (one byte blob + offsets): what a decoded BAM brings, and what synthetic
        records build without a Python loop over the reads."""
        from .names import NameTable
        if isinstance(self.names, NameTable):
            return self.names
        if self.names is not None:
            return NameTable.from_list(self.names)
        t = NameTable.from_decimal("read", self.name_gid.cpu().numpy(), 8)
        return t if t is not None else NameTable.from_list(self.materialise_names())

    def to(self, device) -> "Records":
        kw = {}
        for k, v in self.__dict__.items():
            kw[k] = v.to(device) if isinstance(v, torch.Tensor) else v
        return Records(**kw)

    def algorithmic_bytes(self) -> int:
        """Σ_rec (32 + 4·n_cigar) + 32·N_SA."""
        return int(32 * self.n + 4 * int(self.n_cigar.sum()) + 32 * self.sa.shape[0])


# --------------------------------------------------------------------------------------
# generator
# --------------------------------------------------------------------------------------
def _length_table(cfg: SynthConfig) -> np.ndarray:
    """1024-entry quantile table of a log-normal with the configured mean (host float64, then ints)."""
    mu = math.log(cfg.mean_len) - 0.5 * cfg.sigma_log ** 2
    q = (np.arange(1024, dtype=np.float64) + 0.5) / 1024.0
    # Acklam's rational approximation of the normal quantile (basic IEEE ops only)
    a = [-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02, 1.383577518672690e+02,
         -3.066479806614716e+01, 2.506628277459239e+00]
    b = [-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02, 6.680131188771972e+01,
         -1.328068155288572e+01]
    c = [-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00, -2.549732539343734e+00,
         4.374664141464968e+00, 2.938163982698783e+00]
    d = [7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00]
    z = np.empty_like(q)
    lo = q < 0.02425
    hi = q > 1 - 0.02425
    mid = ~(lo | hi)
    ql = np.sqrt(-2 * np.log(q[lo]))
    z[lo] = (((((c[0] * ql + c[1]) * ql + c[2]) * ql + c[3]) * ql + c[4]) * ql + c[5]) / \
            ((((d[0] * ql + d[1]) * ql + d[2]) * ql + d[3]) * ql + 1)
    qh = np.sqrt(-2 * np.log(1 - q[hi]))
    z[hi] = -(((((c[0] * qh + c[1]) * qh + c[2]) * qh + c[3]) * qh + c[4]) * qh + c[5]) / \
            ((((d[0] * qh + d[1]) * qh + d[2]) * qh + d[3]) * qh + 1)
    qm = q[mid] - 0.5
    r = qm * qm
    z[mid] = (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * qm / \
             (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1)
    ln = np.exp(mu + cfg.sigma_log * z)
    # round to multiples of 16 so that 1-ulp libm differences cannot change the table
    t = (np.floor(ln / 16.0 + 0.5) * 16).astype(np.int64)
    return np.clip(t, cfg.min_len, cfg.max_len)


def _ragged_arange(counts: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """For counts [c0, c1, ...] return (owner index, position within owner) of every element."""
    n = counts.numel()
    dev = counts.device
    owner = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), counts)
    starts = torch.cumsum(counts, 0) - counts
    within = torch.arange(owner.numel(), device=dev, dtype=torch.int64) - starts[owner]
    return owner, within


class _Layout:
    """Device tensors describing circles (unrolled) and windows on one global axis."""

    def __init__(self, cfg: SynthConfig, device):
        unroll = 2 + cfg.max_len // min(sum(len(s) for s in c) for c in cfg.circles)
        bounds, seg_tid, seg_s, seg_e, seg_strand = [], [], [], [], []
        self.circle_base, self.circle_len = [], []
        g = 0
        for c in cfg.circles:
            L = sum(len(s) for s in c)
            self.circle_base.append(g)
            self.circle_len.append(L)
            for _ in range(unroll):
                for s in c:
                    bounds.append(g)
                    seg_tid.append(s.tid); seg_s.append(s.start); seg_e.append(s.end); seg_strand.append(s.strand)
                    g += len(s)
            g += 1 << 22       # gap between circles on the global axis
        bounds.append(g)       # sentinel (start of nothing)
        seg_tid.append(0); seg_s.append(0); seg_e.append(1); seg_strand.append(0)
        t = lambda x: torch.tensor(x, dtype=torch.int64, device=device)
        self.bounds, self.seg_tid, self.seg_s, self.seg_e, self.seg_strand = map(t, (bounds, seg_tid, seg_s, seg_e, seg_strand))
        self.circle_base_t, self.circle_len_t = t(self.circle_base), t(self.circle_len)
        w = np.array(cfg.circle_weight, dtype=np.float64)
        cw = np.cumsum(w / w.sum())
        self.circle_cum = t([int(x * 4294967296.0) for x in cw[:-1]] + [1 << 32])
        wl = np.array([e - s for _, s, e in cfg.windows], dtype=np.float64)
        ww = np.cumsum(wl / wl.sum())
        self.win_cum = t([int(x * 4294967296.0) for x in ww[:-1]] + [1 << 32])
        self.win_tid = t([w_[0] for w_ in cfg.windows])
        self.win_s = t([w_[1] for w_ in cfg.windows])
        self.win_e = t([w_[2] for w_ in cfg.windows])


def _pieces_for_reads(cfg: SynthConfig, lay: _Layout, gid: torch.Tensor, len_table: torch.Tensor):
    """Read-level draw + cut into pieces.  Returns a dict of piece-level int64 tensors (walk order)."""
    seed = cfg.seed
    dev = gid.device
    is_amp = hash_u32(seed, S_KIND, gid) < _thresh(cfg.amp_frac)
    W = len_table[hash_u32(seed, S_LEN, gid) & 1023] + (hash_u32(seed, S_LENJ, gid) & 15)
    flip = hash_u32(seed, S_DIR, gid) & 1
    # ---- amplicon reads
    ci = torch.searchsorted(lay.circle_cum, hash_u32(seed, S_CIRCLE, gid), right=True)
    ci = torch.clamp(ci, max=lay.circle_len_t.numel() - 1)
    L = lay.circle_len_t[ci]
    a = lay.circle_base_t[ci] + _bounded(hash_u32(seed, S_START, gid), torch.clamp(L, max=(1 << 31) - 1))
    jf = torch.searchsorted(lay.bounds, a, right=True) - 1
    jl = torch.searchsorted(lay.bounds, a + W, right=False) - 1
    multi = jl > jf
    first_len = lay.bounds[jf + 1] - a
    fix = multi & (first_len < cfg.min_piece)
    a = torch.where(fix, lay.bounds[jf + 1], a)
    W = torch.where(fix, W - first_len, W)
    jf = torch.where(fix, jf + 1, jf)
    multi = jl > jf
    last_len = a + W - lay.bounds[jl]
    fix = multi & (last_len < cfg.min_piece)
    W = torch.where(fix, lay.bounds[jl] - a, W)
    jl = torch.where(fix, jl - 1, jl)
    npieces = torch.where(is_amp, jl - jf + 1, torch.ones_like(jl))
    # ---- background reads
    wi = torch.searchsorted(lay.win_cum, hash_u32(seed, S_CIRCLE, gid), right=True)
    wi = torch.clamp(wi, max=lay.win_tid.numel() - 1)
    span = torch.clamp(lay.win_e[wi] - lay.win_s[wi] - W - 2, min=1)
    bstart = lay.win_s[wi] + 1 + _bounded(hash_u32(seed, S_START, gid), torch.clamp(span, max=(1 << 31) - 1))
    # ---- ragged piece table
    owner, i = _ragged_arange(npieces)
    amp = is_amp[owner]
    b = jf[owner] + i
    u = torch.maximum(a[owner], lay.bounds[b])
    v = torch.minimum(a[owner] + W[owner], lay.bounds[b + 1])
    ou, ov = u - lay.bounds[b], v - lay.bounds[b]
    sst = lay.seg_strand[b]
    rs = torch.where(sst == 0, lay.seg_s[b] + ou, lay.seg_e[b] - ov)
    re = torch.where(sst == 0, lay.seg_s[b] + ov, lay.seg_e[b] - ou)
    tid = lay.seg_tid[b]
    rs = torch.where(amp, rs, bstart[owner])
    re = torch.where(amp, re, bstart[owner] + W[owner])
    tid = torch.where(amp, tid, lay.win_tid[wi][owner])
    strand = torch.where(amp, sst ^ flip[owner], flip[owner])
    order = torch.where(flip[owner] == 1, npieces[owner] - 1 - i, i)   # position along the read
    return dict(owner=owner, gid=gid[owner], i=i, order=order, npieces=npieces[owner], tid=tid, rs=rs, re=re,
                strand=strand, amp=amp.to(torch.int64)), npieces


def _split_for_deletions(cfg: SynthConfig, P: Dict[str, torch.Tensor]):
    """Decide per piece whether it carries one large deletion; returns (del_start, del_len) (len 0 = none)."""
    seed = cfg.seed
    pk = P["gid"] * 16 + P["i"]
    R = P["re"] - P["rs"]
    dlen = torch.zeros_like(R)
    dstart = torch.zeros_like(R)
    for k, (tid, ds, dl, frac) in enumerate(cfg.planted):
        inside = (P["tid"] == tid) & (P["rs"] + 300 <= ds) & (ds + dl + 300 <= P["re"]) & (dlen == 0)
        carrier = hash_u32(seed, S_CARRIER, P["gid"] * 64 + k) < _thresh(frac)
        j = hash_u32(seed, S_JIT, pk * 64 + k)
        jit = torch.where((j & 7) < 2, ((j >> 3) % 7) - 3, torch.zeros_like(j))   # 25 % of carriers jitter ±3 bp
        m = inside & carrier
        dlen = torch.where(m, torch.full_like(R, dl), dlen)
        dstart = torch.where(m, ds + jit, dstart)
    noise = (hash_u32(seed, S_NOISE, pk) < _thresh(cfg.noise_del_rate)) & (R > 4000) & (dlen == 0)
    nlen = 700 + hash_u32(seed, S_NOISE_LEN, pk) % 2300
    room = torch.clamp(R - 600 - nlen, min=1)
    npos = P["rs"] + 300 + _bounded(hash_u32(seed, S_NOISE_POS, pk), torch.clamp(room, max=(1 << 31) - 1))
    dlen = torch.where(noise, nlen, dlen)
    dstart = torch.where(noise, npos, dstart)
    return dstart, dlen


def _subpiece_events(seed: int, spk: torch.Tensor, R: torch.Tensor):
    """Small-indel events of every sub-piece (key ``spk``, reference length ``R``).

    Returns (owner, k, c, is_del, ev_len, n_ev) where ``c`` is the reference offset of the event inside
    the sub-piece.
    """
    n_ev = torch.clamp(R // EV_SPACING - 1, min=0)
    owner, k = _ragged_arange(n_ev)
    key = spk[owner] * 32768 + k
    Ro, no = R[owner], n_ev[owner]
    c = ((k + 1) * Ro) // (no + 1) + (hash_u32(seed, S_EVJ, key) % 13) - 6
    is_del = (hash_u32(seed, S_EVT, key) % 5) < 3
    lens = torch.tensor(_EV_LENS, dtype=torch.int64, device=R.device)
    ev_len = lens[hash_u32(seed, S_EVL, key) & 7]
    return owner, k, c, is_del, ev_len, n_ev


def _segment_sum(vals: torch.Tensor, owner: torch.Tensor, n: int) -> torch.Tensor:
    out = torch.zeros(n, dtype=torch.int64, device=vals.device)
    out.index_add_(0, owner, vals)
    return out


def generate(cfg: SynthConfig, device="cpu", gid_range: Optional[Tuple[int, int]] = None,
             chunk_pieces: int = 40000) -> Records:
    """Generate the alignment records of reads ``gid_range`` (default: all ``cfg.n_reads``)."""
    dev = torch.device(device)
    seed = cfg.seed
    lay = _Layout(cfg, dev)
    len_table = torch.tensor(_length_table(cfg), dtype=torch.int64, device=dev)
    g0, g1 = gid_range if gid_range is not None else (0, cfg.n_reads)
    gid = torch.arange(g0, g1, dtype=torch.int64, device=dev)

    P, npieces = _pieces_for_reads(cfg, lay, gid, len_table)
    nP = P["gid"].numel()
    pk = P["gid"] * 16 + P["i"]
    dstart, dlen = _split_for_deletions(cfg, P)
    has_del = dlen > 0
    # sub-pieces: sp0 = [rs, dstart) (or whole piece), sp1 = [dstart+dlen, re) when a deletion is carried
    R0 = torch.where(has_del, dstart - P["rs"], P["re"] - P["rs"])
    R1 = torch.where(has_del, P["re"] - dstart - dlen, torch.zeros_like(R0))

    # ---- pass 1: per sub-piece totals (ΣI, ΣD, n_ev) without materialising ops for the whole data set
    def totals(spk, R):
        sI = torch.zeros_like(R); sD = torch.zeros_like(R); nev = torch.zeros_like(R)
        for s in range(0, R.numel(), chunk_pieces):
            e = min(R.numel(), s + chunk_pieces)
            owner, _, _, is_del, ev_len, n_ev = _subpiece_events(seed, spk[s:e], R[s:e])
            sD[s:e] = _segment_sum(torch.where(is_del, ev_len, torch.zeros_like(ev_len)), owner, e - s)
            sI[s:e] = _segment_sum(torch.where(is_del, torch.zeros_like(ev_len), ev_len), owner, e - s)
            nev[s:e] = n_ev
        return sI, sD, nev
    sI0, sD0, nev0 = totals(pk * 2, R0)
    sI1, sD1, nev1 = totals(pk * 2 + 1, R1)
    sI1 = torch.where(has_del, sI1, torch.zeros_like(sI1)); sD1 = torch.where(has_del, sD1, torch.zeros_like(sD1))
    nev1 = torch.where(has_del, nev1, torch.zeros_like(nev1))
    q_piece = (R0 - sD0 + sI0) + torch.where(has_del, R1 - sD1 + sI1, torch.zeros_like(R1))

    # ---- read-level query geometry
    nreads = gid.numel()
    rl = _segment_sum(q_piece, P["owner"], nreads)
    # exclusive cumsum of q_piece in read order: sort key (owner, order)
    perm = torch.argsort(P["owner"] * 64 + P["order"], stable=True)
    qs_sorted = torch.cumsum(q_piece[perm], 0) - q_piece[perm]
    first_of_read = torch.cumsum(npieces, 0) - npieces
    qs_sorted = qs_sorted - qs_sorted[first_of_read][P["owner"][perm]]
    qs = torch.empty_like(qs_sorted); qs[perm] = qs_sorted
    qe = qs + q_piece - 1
    rlp = rl[P["owner"]]
    lead = torch.where(P["strand"] == 0, qs, rlp - 1 - qe)
    trail = rlp - q_piece - lead
    # primary = first piece (walk order) with the maximal query length
    score = q_piece * 64 + (63 - P["i"])
    best = torch.zeros(nreads, dtype=torch.int64, device=dev)
    best.scatter_reduce_(0, P["owner"], score, reduce="amax", include_self=True)
    primary = score == best[P["owner"]]
    chim = P["npieces"] > 1
    # mapq
    hm = hash_u32(seed, S_MAPQ, pk)
    hv = hash_u32(seed, S_MAPQV, pk)
    mapq = torch.full_like(pk, 60)
    low = chim & (hm < _thresh(cfg.lowmapq_frac))
    mid = chim & ~low & (hm < _thresh(cfg.lowmapq_frac + cfg.midmapq_frac))
    mapq = torch.where(low, hv % 10, mapq)
    mapq = torch.where(mid, 10 + hv % 10, mapq)
    bglow = ~chim & (hm < _thresh(cfg.bg_lowmapq_frac))
    mapq = torch.where(bglow, hv % 20, mapq)
    # secondary, SEQ-less copies: a few single-piece background reads are re-labelled as a secondary
    # alignment of the previous read name (exercises flag >= 256 / no-SEQ handling)
    sec = (~chim) & (P["amp"] == 0) & (P["gid"] > 0) & (hash_u32(seed, S_SEC, P["gid"]) < _thresh(cfg.secondary_frac))
    flag = (P["strand"] * 16) | torch.where(primary, torch.zeros_like(pk), torch.full_like(pk, 2048))
    flag = torch.where(sec, (P["strand"] * 16) | 256, flag)
    mapq = torch.where(sec, torch.zeros_like(mapq), mapq)
    has_seq = torch.where(sec, torch.zeros_like(pk), torch.ones_like(pk))
    name_gid_piece = torch.where(sec, P["gid"] - 1, P["gid"])
    nm = sI0 + sD0 + sI1 + sD1 + dlen + hash_u32(seed, S_NM, pk) % (q_piece // 50 + 1)
    qlen_field = torch.where(primary | sec, rlp, q_piece)   # SEQ-less secondaries still carry the CIGAR-implied length here
    clip_op = torch.where(primary | sec, torch.full_like(pk, OP_S), torch.full_like(pk, OP_H))

    # ---- record (== piece) order: stable sort by (tid, pos)
    rperm = torch.argsort(P["tid"] * (1 << 32) + P["rs"], stable=True)
    inv = torch.empty_like(rperm); inv[rperm] = torch.arange(nP, device=dev)
    n_ops_real = (lead > 0).to(torch.int64) + (trail > 0).to(torch.int64) + 2 * nev0 + 1 + \
        torch.where(has_del, 2 * nev1 + 2, torch.zeros_like(nev1))
    n_ops_pad = (n_ops_real + 3) // 4 * 4
    cig_off = torch.zeros(nP + 1, dtype=torch.int64, device=dev)
    cig_off[1:] = torch.cumsum(n_ops_pad[rperm], 0)
    total_ops = int(cig_off[-1])
    cigar = torch.full((total_ops,), OP_PAD, dtype=torch.int32, device=dev)
    base = torch.empty(nP, dtype=torch.int64, device=dev)   # first op slot of every piece (generation order)
    base[rperm] = cig_off[:-1]

    def pack(length, op):
        v = (length << 4) | op
        return torch.where(v >= (1 << 31), v - (1 << 32), v).to(torch.int32)

    # clips and the big deletion op
    m = lead > 0
    cigar[base[m]] = pack(lead[m], clip_op[m])
    sp0_base = base + m.to(torch.int64)
    del_slot = sp0_base + 2 * nev0 + 1
    cigar[del_slot[has_del]] = pack(dlen[has_del], torch.full_like(dlen[has_del], OP_D))
    sp1_base = del_slot + 1
    tr_slot = base + n_ops_real - 1
    m2 = trail > 0
    cigar[tr_slot[m2]] = pack(trail[m2], clip_op[m2])

    # ---- pass 2: materialise the M / I / D ops of every sub-piece
    def emit(spk, R, sp_base, active):
        idx = torch.nonzero(active).squeeze(1)
        for s in range(0, idx.numel(), chunk_pieces):
            sel = idx[s:s + chunk_pieces]
            Rs, bs = R[sel], sp_base[sel]
            owner, k, c, is_del, ev_len, n_ev = _subpiece_events(seed, spk[sel], Rs)
            # event ops
            cigar[bs[owner] + 2 * k + 1] = pack(ev_len, torch.where(is_del, torch.full_like(k, OP_D), torch.full_like(k, OP_I)))
            # reference offset where the M run after event k starts
            after = c + torch.where(is_del, ev_len, torch.zeros_like(ev_len))
            # M run k (k>=1) = c_k - after_{k-1}; M_0 = c_0
            prev_after = torch.zeros_like(after)
            if after.numel() > 1:
                prev_after[1:] = after[:-1]
            prev_after = torch.where(k == 0, torch.zeros_like(after), prev_after)
            cigar[bs[owner] + 2 * k] = pack(c - prev_after, torch.full_like(k, OP_M))
            # last M run of every sub-piece: R - after_{n-1} (or R when there is no event)
            last = torch.zeros(sel.numel(), dtype=torch.int64, device=dev)
            is_last = (k == n_ev[owner] - 1)
            last.index_add_(0, owner[is_last], after[is_last])
            cigar[bs + 2 * n_ev] = pack(Rs - last, torch.full_like(Rs, OP_M))
    emit(pk * 2, R0, sp0_base, torch.ones_like(has_del))
    emit(pk * 2 + 1, R1, sp1_base, has_del)

    # ---- SA rows: for each chimeric record, the other pieces of the read (primary first, then walk order)
    # compressed SA cigar of every piece (minimap2 rule)
    rlen_piece = P["re"] - P["rs"]
    sa_m = torch.minimum(q_piece, rlen_piece)
    sa_x = q_piece - rlen_piece          # >0: I, <0: D
    n_sa_rec = torch.where(chim, P["npieces"] - 1, torch.zeros_like(pk))
    rec_first = first_of_read[P["owner"]]          # piece index (generation order) of the read's first piece
    # index of the primary piece inside its read
    prim_i = torch.zeros(nreads, dtype=torch.int64, device=dev)
    prim_i.index_add_(0, P["owner"][primary], P["i"][primary])
    sa_owner_s, j = _ragged_arange(n_sa_rec[rperm])            # rows in sorted-record order
    src = rperm[sa_owner_s]                                    # generation index of the record carrying the tag
    pi_, ii_ = prim_i[P["owner"][src]], P["i"][src]
    # candidate list: [primary] + [0..n-1 without primary]; drop self
    self_is_prim = ii_ == pi_
    # entries for a non-primary record: j=0 -> primary; j>=1 -> (j-1)-th of the others excluding self and primary
    # entries for the primary record: j -> j-th of the others
    kk = torch.where(self_is_prim, j, j - 1)                   # rank among non-primary pieces (excluding self if needed)
    # map rank among "non-primary, non-self" pieces to a walk index
    lo_, hi_ = torch.minimum(pi_, ii_), torch.maximum(pi_, ii_)
    t = kk
    t = torch.where(t >= lo_, t + 1, t)
    t = torch.where((~self_is_prim) & (t >= hi_), t + 1, t)
    tgt_i = torch.where((~self_is_prim) & (j == 0), pi_, t)
    tgt = rec_first[src] + tgt_i                               # generation index of the described piece
    sa = torch.stack([P["tid"][tgt], P["rs"][tgt] + 1, P["strand"][tgt], lead[tgt], sa_m[tgt], sa_x[tgt],
                      trail[tgt], mapq[tgt]], dim=1).to(torch.int32)
    sa_nm = nm[tgt].to(torch.int32)
    sa_off = torch.zeros(nP + 1, dtype=torch.int64, device=dev)
    sa_off[1:] = torch.cumsum(n_sa_rec[rperm], 0)

    # ---- non-ACGT bases: a few records carry N at the start of their first M run
    nb = (hash_u32(seed, S_NBASE, pk) < _thresh(cfg.nbase_frac)) & (R0 >= 40) & (has_seq == 1)
    nb_cnt = torch.where(nb, 1 + hash_u32(seed, S_NBASE, pk + 7) % 3, torch.zeros_like(pk))
    nb_owner_s, nb_j = _ragged_arange(nb_cnt[rperm])
    nb_src = rperm[nb_owner_s]
    nonacgt_pos = (P["rs"][nb_src] + 1 + 3 * nb_j).to(torch.int32)

    # ---- names: id in first-appearance order over sorted records
    ng_sorted = name_gid_piece[rperm]
    uniq, inverse = torch.unique(ng_sorted, return_inverse=True)
    first_pos = torch.full((uniq.numel(),), nP, dtype=torch.int64, device=dev)
    first_pos.scatter_reduce_(0, inverse, torch.arange(nP, device=dev), reduce="amin", include_self=True)
    order_names = torch.argsort(first_pos, stable=True)
    rank = torch.empty_like(order_names); rank[order_names] = torch.arange(uniq.numel(), device=dev)
    name_id = rank[inverse]
    name_gid = uniq[order_names]

    i32 = lambda x: x[rperm].to(torch.int32)
    return Records(
        n=nP, tid=i32(P["tid"]), pos=i32(P["rs"]), end=i32(P["re"]), flag=i32(flag), mapq=i32(mapq),
        qlen=i32(qlen_field), has_seq=i32(has_seq), nm=i32(nm), name_id=name_id.to(torch.int32),
        n_cigar=i32(n_ops_real), cigar_off=cig_off, cigar=cigar, sa_off=sa_off, sa=sa, sa_nm=sa_nm,
        nonacgt_rec=nb_owner_s, nonacgt_pos=nonacgt_pos, n_names=int(uniq.numel()), name_gid=name_gid)


# --------------------------------------------------------------------------------------
# helpers for host-side consumers (fake pysam, BAM writer, oracle)
# --------------------------------------------------------------------------------------
def sa_cigar_string(c5: int, m: int, x: int, c3: int) -> str:
    s = ""
    if c5 > 0:
        s += f"{c5}S"
    s += f"{m}M"
    if x > 0:
        s += f"{x}I"
    elif x < 0:
        s += f"{-x}D"
    if c3 > 0:
        s += f"{c3}S"
    return s


def sa_entry_string(row: Sequence[int], nm: int, chroms: Sequence[str] = CHROMS) -> str:
    tid, pos1, strand, c5, m, x, c3, mapq = [int(v) for v in row]
    return f"{chroms[tid]},{pos1},{'+-'[strand]},{sa_cigar_string(c5, m, x, c3)},{mapq},{nm}"


def seq_of_record(cfg_seed: int, rec_index: int, length: int) -> np.ndarray:
    """Deterministic ACGT sequence (uint8 ASCII) for a record; only used when a BAM file is written."""
    k = torch.arange(length, dtype=torch.int64) + rec_index * (1 << 22)
    h = hash_u32(cfg_seed, S_SEQ, k) & 3
    return np.frombuffer(b"ACGT", dtype=np.uint8)[h.numpy()]


def records_from_alignments(alns: Sequence[dict], device="cpu") -> Records:
    """Build a ``Records`` store from hand-written alignments (tests of ragged / odd CIGARs).

    Every alignment is a dict with keys tid, pos, cigar=[(op, len), ...] and optional flag, mapq, name, has_seq,
    nm, qlen, sa=[(tid, pos1, strand, c5, m, x, c3, mapq, nm), ...], nonacgt=[refpos, ...].  Input must already be
    in (tid, pos) order.
    """
    n = len(alns)
    ref_ops = {OP_M, OP_D, OP_N, OP_EQ, OP_X}
    qry_ops = {OP_M, OP_I, OP_S, OP_EQ, OP_X}
    tid, pos, end, flag, mapq, qlen, has_seq, nm, name_id, n_cig = ([] for _ in range(10))
    cig, cig_off, sa_rows, sa_nm, sa_off, na_rec, na_pos = [], [0], [], [], [0], [], []
    names: Dict[str, int] = {}
    for i, a in enumerate(alns):
        ops = a.get("cigar", [])
        unmapped = bool(a.get("flag", 0) & 4)
        rlen = 0 if unmapped else sum(l for o, l in ops if o in ref_ops)
        tid.append(a["tid"]); pos.append(a["pos"]); end.append(a["pos"] + max(1, rlen))
        flag.append(a.get("flag", 0)); mapq.append(a.get("mapq", 60)); has_seq.append(a.get("has_seq", 1))
        qlen.append(a.get("qlen", sum(l for o, l in ops if o in qry_ops))); nm.append(a.get("nm", 0))
        nm_ = a.get("name", "r%d" % i)
        name_id.append(names.setdefault(nm_, len(names)))
        n_cig.append(len(ops))
        for o, l in ops:
            cig.append((l << 4) | o)
        while len(cig) % 4:
            cig.append(OP_PAD)
        cig_off.append(len(cig))
        for row in a.get("sa", []):
            sa_rows.append(list(row[:8])); sa_nm.append(row[8])
        sa_off.append(len(sa_rows))
        for p in a.get("nonacgt", []):
            na_rec.append(i); na_pos.append(p)
    dev = torch.device(device)
    t32 = lambda x: torch.tensor(x, dtype=torch.int32, device=dev)
    t64 = lambda x: torch.tensor(x, dtype=torch.int64, device=dev)
    cg = np.array(cig, dtype=np.uint32).view(np.int32) if cig else np.zeros(0, dtype=np.int32)
    return Records(n=n, tid=t32(tid), pos=t32(pos), end=t32(end), flag=t32(flag), mapq=t32(mapq), qlen=t32(qlen),
                   has_seq=t32(has_seq), nm=t32(nm), name_id=t32(name_id), n_cigar=t32(n_cig), cigar_off=t64(cig_off),
                   cigar=torch.tensor(cg, dtype=torch.int32, device=dev), sa_off=t64(sa_off),
                   sa=torch.tensor(sa_rows, dtype=torch.int32, device=dev).reshape(-1, 8), sa_nm=t32(sa_nm),
                   nonacgt_rec=t64(na_rec), nonacgt_pos=t32(na_pos), n_names=len(names),
                   name_gid=torch.arange(len(names), dtype=torch.int64, device=dev),
                   names=[k for k, _ in sorted(names.items(), key=lambda kv: kv[1])])


def merge_sorted(a: Records, b: Records) -> Records:
    """Concatenate two record stores (same header) and restore (tid, pos) order (stable: ``a`` before ``b``)."""
    assert a.header_chroms == b.header_chroms
    na = a.materialise_names()
    nb = b.materialise_names()
    n = a.n + b.n
    cat = lambda k: torch.cat([getattr(a, k).cpu(), getattr(b, k).cpu()])
    tid, pos = cat("tid").to(torch.int64), cat("pos").to(torch.int64)
    perm = torch.argsort(tid * (1 << 32) + pos, stable=True)
    # names: first-appearance order over the merged, sorted records
    all_names = [na[i] for i in a.name_id.tolist()] + [nb[i] for i in b.name_id.tolist()]
    ids: Dict[str, int] = {}
    name_id = []
    for i in perm.tolist():
        name_id.append(ids.setdefault(all_names[i], len(ids)))
    # ragged pieces
    def ragged(off_a, off_b, data_a, data_b):
        cnt = torch.cat([off_a[1:] - off_a[:-1], off_b[1:] - off_b[:-1]]).cpu()
        start = torch.cat([off_a[:-1].cpu(), off_b[:-1].cpu() + data_a.shape[0]])
        data = torch.cat([data_a.cpu(), data_b.cpu()])
        cp, sp = cnt[perm], start[perm]
        owner, within = _ragged_arange(cp)
        new_off = torch.zeros(n + 1, dtype=torch.int64)
        new_off[1:] = torch.cumsum(cp, 0)
        return new_off, data[sp[owner] + within]
    cig_off, cig = ragged(a.cigar_off, b.cigar_off, a.cigar, b.cigar)
    sa_off, sa = ragged(a.sa_off, b.sa_off, a.sa, b.sa)
    _, sa_nm = ragged(a.sa_off, b.sa_off, a.sa_nm, b.sa_nm)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(n)
    na_rec = inv[torch.cat([a.nonacgt_rec.cpu(), b.nonacgt_rec.cpu() + a.n])]
    na_pos = cat("nonacgt_pos")
    o = torch.argsort(na_rec, stable=True)
    g = lambda k: cat(k)[perm]
    return Records(n=n, tid=g("tid"), pos=g("pos"), end=g("end"), flag=g("flag"), mapq=g("mapq"), qlen=g("qlen"),
                   has_seq=g("has_seq"), nm=g("nm"), name_id=torch.tensor(name_id, dtype=torch.int32), n_cigar=g("n_cigar"),
                   cigar_off=cig_off, cigar=cig, sa_off=sa_off, sa=sa, sa_nm=sa_nm, nonacgt_rec=na_rec[o], nonacgt_pos=na_pos[o],
                   n_names=len(ids), name_gid=torch.arange(len(ids), dtype=torch.int64),
                   names=[k for k, _ in sorted(ids.items(), key=lambda kv: kv[1])],
                   header_chroms=list(a.header_chroms), header_lens=list(a.header_lens))


def _edge_alignments(cfg: SynthConfig) -> List[dict]:
    """Hand-written records injected into the 'tiny' data set: the corner cases of the reference's loops."""
    M, I, D, N, S, H, P, EQ, X = range(9)
    seg = cfg.circles[0]
    a, b = seg[0], seg[1]                     # two amplicon segments on chr8
    t = a.tid
    base = a.start + 60_000
    out = []
    # (1) chimeric read WITHOUT a primary record: two supplementary records pointing at each other -> dropped (ibg:163-173)
    out += [dict(tid=t, pos=base, flag=2048, name="edge_noprimary", cigar=[(M, 3000), (H, 2000)],
                 sa=[(t, b.start + 501, 0, 3000, 2000, 0, 0, 60, 3)]),
            dict(tid=t, pos=b.start + 500, flag=2048, name="edge_noprimary", cigar=[(H, 3000), (M, 2000)],
                 sa=[(t, base + 1, 0, 0, 3000, 0, 2000, 60, 2)])]
    # (2) SA entry without soft clip ("2000M"): the read's value becomes ([], [], []) (cp:248-253)
    out += [dict(tid=t, pos=base + 100, flag=0, name="edge_noS", cigar=[(M, 3000), (S, 2000)], nm=5,
                 sa=[(t, b.start + 801, 0, 0, 0, 0, 0, 60, 1)])]
    # (3) placed unmapped read without CIGAR inside the amplicon, and an unplaced one (tid -1 is not emitted: whole-file
    #     fetch skips it) -> name sets of point fetches include the placed one, read counts do not
    out += [dict(tid=t, pos=base + 200, flag=4, name="edge_unmapped", cigar=[], qlen=500, mapq=0)]
    # (4) N (reference skip) and =/X ops; a 700-bp N gap counts as a large gap (get_blocks advances over N)
    out += [dict(tid=t, pos=base + 300, name="edge_N_%d" % k, nm=9 + k,
                 cigar=[(S, 10), (EQ, 800), (X, 3), (EQ, 400), (N, 700 + k), (M, 900), (I, 2), (M, 600)]) for k in range(4)]
    # (5) two large deletions in ONE record (indices 0 and 1 of its list), shared by four reads
    out += [dict(tid=t, pos=base + 5000 + k, name="edge_2del_%d" % k, nm=20,
                 cigar=[(M, 1500 - k), (D, 900), (M, 2000), (D, 1200), (M, 1500)]) for k in range(4)]
    # (6) the same big deletion in a MAPQ-19 record: ignored (ibg:754)
    out += [dict(tid=t, pos=base + 5000, name="edge_lowmapq", mapq=19, cigar=[(M, 1500), (D, 900), (M, 2000)])]
    # (7) chimeric read with one piece on a chromosome that has no CN segments (chr1): cniset {-1} (ibg:209-210)
    out += [dict(tid=t, pos=base + 9000, flag=0, name="edge_chr1", cigar=[(M, 4000), (S, 3000)], nm=4,
                 sa=[(0, 1_000_001, 1, 0, 3000, 0, 4000, 60, 2)]),
            dict(tid=0, pos=1_000_000, flag=2064, name="edge_chr1", cigar=[(M, 3000), (H, 4000)], nm=2,
                 sa=[(t, base + 9001, 0, 0, 4000, 0, 3000, 60, 4)])]
    # (8) adjacent D ops that only TOGETHER exceed 600 bp, and a 600-bp gap that must not count
    out += [dict(tid=t, pos=base + 12000, name="edge_adjD", cigar=[(M, 700), (D, 300), (D, 301), (M, 700)]),
            dict(tid=t, pos=base + 12010, name="edge_600", cigar=[(M, 700), (D, 600), (M, 700)])]
    return sorted(out, key=lambda r: (r["tid"], r["pos"]))


def _hsr_edge_alignments(cfg: SynthConfig) -> List[dict]:
    """Hand-written three-piece reads for the ``hsr`` mode (/root/reference/src/hsr.py:117-147): integration junctions
    between a locus outside the ecDNA and ecDNA segment ``a``, including the "skip a low-MAPQ middle piece" rule."""
    M, I, D, N, S, H, P, EQ, X = range(9)
    a = cfg.circles[0][0]                     # ecDNA segment (the hsr goldens use the first half of circle 0 as ecDNA)
    t = a.tid
    out_pos = cfg.windows[0][1] + 1_000_000   # inside the window (CN segments exist), far from every circle segment
    mid_pos = cfg.windows[0][1] + 3_000_000
    in_pos = a.start + 20_000
    out = []

    def three_piece(name, k, mq_mid, mq_out=60, mq_in=60, shift=0):
        # query layout: [0, 3000) outside locus, [3000, 4000) middle piece, [4000, 7000) inside the ecDNA
        A = (t, out_pos + shift + k + 1, 0, 0, 3000, 0, 4000, mq_out, 3)
        B = (t, mid_pos + 7 * k + 1, 0, 3000, 1000, 0, 3000, mq_mid, 1)
        Cc = (t, in_pos + shift + k + 1, 0, 4000, 3000, 0, 0, mq_in, 2)
        return [dict(tid=t, pos=out_pos + shift + k, flag=0, mapq=mq_out, name=name, nm=3, cigar=[(M, 3000), (S, 4000)], sa=[B, Cc]),
                dict(tid=t, pos=mid_pos + 7 * k, flag=2048, mapq=mq_mid, name=name, nm=1, cigar=[(H, 3000), (M, 1000), (H, 3000)], sa=[A, Cc]),
                dict(tid=t, pos=in_pos + shift + k, flag=2048, mapq=mq_in, name=name, nm=2, cigar=[(H, 4000), (M, 3000)], sa=[A, B])]

    # (1) low-MAPQ middle piece: the junction is called between pieces 0 and 2 (hsr.py:135-147)
    for k in range(5):
        out += three_piece("hsr_skip_%d" % k, k, mq_mid=5)
    # (2) the same junction 60 bp away (same cluster, merged into the first refined breakpoint by the <= / < rule, hsr.py:160-163)
    for k in range(3):
        out += three_piece("hsr_near_%d" % k, k, mq_mid=3, shift=60)
    # (3) middle piece with MAPQ 15: neither rule applies (>= 10 blocks the skip, < 20 blocks the adjacent pairs)
    for k in range(3):
        out += three_piece("hsr_mid15_%d" % k, k, mq_mid=15, shift=5000)
    # (4) middle piece with MAPQ 60 outside the ecDNA: only the adjacent pair (1, 2) is a junction (hsr.py:122-131)
    for k in range(4):
        out += three_piece("hsr_adj_%d" % k, k, mq_mid=60, shift=9000)
    # (5) outside piece with MAPQ 19: nothing is called
    for k in range(2):
        out += three_piece("hsr_lowout_%d" % k, k, mq_mid=5, mq_out=19, shift=13000)
    return sorted(out, key=lambda r: (r["tid"], r["pos"]))


def with_decoy_contigs(rec: Records, n_decoys: int = 100) -> Records:
    """The same alignments under a header that lists `n_decoys` short contigs BEFORE chr1 (as references with decoys / alt
    contigs sorted first do): every target id — records and SA rows — moves up by n_decoys, names and coordinates stay."""
    kw = dict(rec.__dict__)
    kw["tid"] = torch.where(rec.tid >= 0, rec.tid + n_decoys, rec.tid)
    sa = rec.sa.clone()
    if sa.numel():
        sa[:, 0] += n_decoys
    kw["sa"] = sa
    kw["header_chroms"] = ["decoy%03d" % k for k in range(n_decoys)] + list(rec.header_chroms)
    kw["header_lens"] = [5000 + k for k in range(n_decoys)] + list(rec.header_lens)
    return Records(**kw)


def dataset(name: str, device="cpu") -> Tuple[SynthConfig, Records]:
    """Named data set = configuration + records ('tiny_edge' = 'tiny' plus hand-written corner-case records)."""
    if name == "tiny_edge":
        cfg = named_config("tiny")
        cfg.name = "tiny_edge"
        rec = merge_sorted(generate(cfg, "cpu"), records_from_alignments(_edge_alignments(cfg)))
        return cfg, (rec if str(device) == "cpu" else rec.to(device))
    if name == "hsr_edge":
        cfg = named_config("tiny")
        cfg.name = "hsr_edge"
        rec = merge_sorted(generate(cfg, "cpu"), records_from_alignments(_hsr_edge_alignments(cfg)))
        return cfg, (rec if str(device) == "cpu" else rec.to(device))
    if name in ("cfg3_12k", "cfg3_2amp"):
        # the headline configuration's layout (3 chromosomes, 3 circles x 8 segments, 10 seeds) at a read count the unmodified
        # reference finishes in seconds: 12 000 reads x 20 kb reproduce config 3's graph shape (1 amplicon, 28 discordant
        # edges); 6 000 reads x 8 kb leave the amplicon in two connected components (two ccids -> two graph files)
        cfg = scaled_config("cfg3", 12000 if name == "cfg3_12k" else 6000)
        if name == "cfg3_2amp":
            cfg.mean_len = 8000
        cfg.name = name
        return cfg, generate(cfg, device)
    cfg = named_config(name)
    return cfg, generate(cfg, device)
