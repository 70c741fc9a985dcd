/*
 * coral_hip.h — C ABI of libcoral_hip.so: the MI355X (gfx950) data-parallel hot path of CoRAL's
 * breakpoint-graph construction.
 *
 * The reference (suhas-r/CoRAL @ 2024-10-24) is pure Python and has no FFI of its own; the calls below
 * replace, one for one, the per-record loops the reference runs through pysam.
 * Each entry point cites the reference lines it stands in for.  INTEGRATION.md shows the ctypes
 * binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every pointer is a plain device (HBM) or host pointer as stated; no framework types;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); launches are asynchronous
 *     unless a function says it returns a value read back from the device;
 *   - return value 0 = success, negative = error (see coral_last_error());
 *   - the library allocates nothing on the device: the caller owns all buffers and workspaces;
 *   - results are integers and order-free (atomics only ever add integers), so they are bit-exact and
 *     independent of wave scheduling; compacted lists are returned unordered together with a sort key
 *     (record ordinal, within-record index) that defines the reference's iteration order.
 *
 * Record layout in HBM (structure of arrays, BAM file order = (tid, pos) order):
 *   tid, pos, end      int32   reference id, 0-based start, htslib bam_endpos
 *   flagmq             int32   flag | mapq << 16 | has_seq << 24
 *   n_cigar            int32   number of real CIGAR ops
 *   cigar_off          int64   offset (in ops) of the record's first op; ALWAYS a multiple of 4
 *   cigar              uint32  BAM-packed ops (len << 4 | op); every record is padded with op 15 to a
 *                              multiple of 4 ops so a wave reads whole 16-byte quads
 */
#ifndef CORAL_HIP_H
#define CORAL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CORAL_OK 0
#define CORAL_ERR_ARG (-1)        /* bad argument (null pointer, unsorted / overlapping segments, ...) */
#define CORAL_ERR_HIP (-2)        /* a HIP runtime call failed */
#define CORAL_ERR_CAPACITY (-3)   /* an output list overflowed; *count holds the needed size */
#define CORAL_ERR_FORMAT (-4)     /* malformed BAM / BGZF input */
#define CORAL_ERR_ZERODIV (-5)    /* a division the reference performs has a zero divisor (ZeroDivisionError there) */

typedef struct coral_records {
    int64_t n_rec;
    const int32_t *tid;
    const int32_t *pos;
    const int32_t *end;
    const int32_t *flagmq;
    const int32_t *n_cigar;
    const int64_t *cigar_off;
    const uint32_t *cigar;
} coral_records_t;

const char *coral_version(void);
const char *coral_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * coral_cigar_scan — ONE fused pass over every CIGAR op in HBM.
 *
 * Replaces the per-record `get_blocks()` walk and block loop of find_smalldel_breakpoints
 * (/root/reference/src/infer_breakpoint_graph.py:750-762), and pre-computes what the later
 * coverage calls need per record so the CIGAR bytes are streamed once.  summary[i] is one 16-byte row:
 *   [0] mbases     Σ length of M/=/X ops        (what count_coverage adds for a fully covered record,
 *                                                 infer_breakpoint_graph.py:131, :1033)
 *   [1] qinfer     Σ length of M/I/S/H/=/X ops  (pysam infer_read_length(), :1031)
 *   [2] blk_first  start of the first aligned block, [3] blk_last end of the last one (-1: none)
 *                                                (blocks[0][0], blocks[-1][1] at :760)
 * and one row (record, op index of the next block, prev block end, next block start) is appended to
 * `gaps` for every pair of consecutive blocks further apart than `min_gap` in a record with
 * mapq >= min_mapq (:754, :757-758).  `counters` are TWO device words the caller zeroes beforehand: [0] counts the
 * gap rows (rows beyond gap_cap are dropped, the counter still counts them), [1] is the kernel's work cursor.
 * summary and gaps must be 16-byte aligned.
 * ------------------------------------------------------------------------------------------------ */
int coral_cigar_scan(const coral_records_t *rec, int32_t min_gap, int32_t min_mapq, int32_t *summary /* [n_rec][4] */,
                     int32_t *gaps /* [gap_cap][4] */, uint32_t *counters /* [2] */, uint32_t gap_cap, void *stream);
/* Name of the kernel behind coral_cigar_scan as rocprofv3 prints it (reported next to the roofline figures). */
const char *coral_scan_kernel_name(void);

/* ------------------------------------------------------------------------------------------------
 * coral_segment_coverage — per-segment record count and aligned-base count.
 *
 * Replaces, for S sorted, pairwise disjoint half-open segments (seg_tid, seg_start, seg_end):
 *   n_reads[j] = #records overlapping the segment with infer_read_length() > 0
 *                (/root/reference/src/infer_breakpoint_graph.py:1031-1032)
 *   n_bases[j] = Σ of the four pysam count_coverage arrays with quality_threshold=0,
 *                read_callback='nofilter' BEFORE the non-ACGT correction (:130-132, :1033-1034):
 *                aligned (M/=/X) bases of records with SEQ that fall inside the segment.
 * Records fully inside one segment use the `summary` rows of coral_cigar_scan; only records straddling a
 * segment boundary have their CIGAR walked again.  n_reads / n_bases are ADDED to (caller zeroes).
 * `strad` is a workspace of n_rec uint32 and `strad_count` a zeroed device counter.
 * ------------------------------------------------------------------------------------------------ */
int coral_segment_coverage(const coral_records_t *rec, const int32_t *summary /* [n_rec][4] */,
                           int32_t n_seg, const int32_t *seg_tid, const int32_t *seg_start,
                           const int32_t *seg_end, unsigned long long *n_reads,
                           unsigned long long *n_bases, uint32_t *strad, uint32_t *strad_count,
                           void *stream);

/* ------------------------------------------------------------------------------------------------
 * coral_point_cover — which records cover which query points.
 *
 * Replaces the four single-position region fetches per concordant edge
 * (/root/reference/src/infer_breakpoint_graph.py:1043-1046): for P points sorted by (tid, pos) appends
 * the packed pair (point index << 32 | record ordinal) for every record with pos <= p < end.
 * *pair_count is a zeroed device counter; pairs beyond pair_cap are dropped (still counted).
 * max_span: an upper bound of end - pos over the records (the longest reference span of an alignment), which confines the
 * search to the records starting within max_span in front of a point; <= 0 = unknown (the window starts at the contig's
 * first record).  Relies on the (tid, pos) order of the record layout.
 * ------------------------------------------------------------------------------------------------ */
int coral_point_cover(const coral_records_t *rec, int32_t n_pts, const int32_t *pt_tid,
                      const int32_t *pt_pos, int32_t max_span, unsigned long long *pairs, uint32_t *pair_count,
                      uint32_t pair_cap, void *stream);

/* ------------------------------------------------------------------------------------------------
 * coral_sa_table — the chimeric-alignment table of ALL reads from the tokenised SA rows (device in, device out).
 *
 * Replaces the SA-tag half of fetch() (/root/reference/src/infer_breakpoint_graph.py:139-174: first-seen
 * de-duplication of SA entries per read name, read_length = query_length of the first record with flag < 256,
 * reads without a primary dropped) and alignment_from_satags with the nine cigar2pos* shapes
 * (/root/reference/src/cigar_parsing.py:17-269), including the stable (qs, qe) sort.
 *   inputs  rec_tid / rec_flagmq / rec_qlen (pysam query_length) / rec_name : int32[n_rec];
 *           sa int32[n_sa][8] (layout of coral_bam_decode_*), sa_nm int32[n_sa], sa_rec int32[n_sa] owner record
 *   outputs out_rows int32[<= n_sa][8] = qs, qe, tid, rint[1], rint[2], strand, mapq, NM (rows of read i are
 *           out_off[i]..out_off[i+1], sorted by (qs, qe)); out_name / out_failed int32[<= n_sa] per read (failed = the
 *           read's value is the reference's ([], [], [])); reads are in the insertion order of the reference's dict;
 *           out_read_length int32[n_names] (-1 = no primary seen);  counts (HOST) = {n_reads, n_rows}
 * `workspace` is device scratch; when it is too small the call returns CORAL_ERR_CAPACITY with counts[0] = MiB needed.
 * Errors mirror the reference: CORAL_ERR_FORMAT = SA CIGAR outside the nine shapes (KeyError, cp:255),
 * CORAL_ERR_ZERODIV = zero-length query interval (ZeroDivisionError, cp:268).  Synchronises `stream`.
 * ------------------------------------------------------------------------------------------------ */
int coral_sa_table(int32_t n_rec, const int32_t *rec_tid, const int32_t *rec_flagmq, const int32_t *rec_qlen,
                   const int32_t *rec_name, int32_t n_names, int32_t n_sa, const int32_t *sa, const int32_t *sa_nm,
                   const int32_t *sa_rec, void *workspace, int64_t workspace_bytes, int32_t *out_rows, int32_t *out_off,
                   int32_t *out_name, int32_t *out_failed, int32_t *out_read_length, int32_t *counts, void *stream);
const char *coral_sa_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * coral_hash_rows — hash_alignment_to_seg (/root/reference/src/infer_breakpoint_graph.py:181-210) on coral_sa_table's device
 * rows: for both ends of every local alignment the index of the CN segment containing it (the reference's IntervalTree
 * point queries), and the inverted index (contig, segment) -> alignments in the reference's append order.
 *   inputs  rows int32[n_rows][8] (coral_sa_table layout); the CN segments as a device table sorted by (contig id, start),
 *           pairwise disjoint: seg_tid / seg_start / seg_end (exclusive) / seg_idx (the segment's index within its contig,
 *           file order) int32[n_seg]; tid_has_segs int32[n_tid] (0: contig absent from the CN file)
 *   outputs cni0 / cni1 int32[n_rows] (-1 no segment, -3 contig absent); e_key int64[2 * n_rows] = contig << 32 | segment
 *           ascending and e_row int32[2 * n_rows] = table row — the first *n_ent (HOST) entries are valid
 * `workspace`: device scratch; too small -> CORAL_ERR_CAPACITY with *n_ent = MiB needed.  Synchronises `stream`.
 * ------------------------------------------------------------------------------------------------ */
int coral_hash_rows(int32_t n_rows, const int32_t *rows, int32_t n_seg, const int32_t *seg_tid, const int32_t *seg_start,
                    const int32_t *seg_end, const int32_t *seg_idx, const int32_t *tid_has_segs, int32_t n_tid,
                    void *workspace, int64_t workspace_bytes, int32_t *cni0, int32_t *cni1, int64_t *e_key, int32_t *e_row,
                    int32_t *n_ent, void *stream);

/* ------------------------------------------------------------------------------------------------
 * coral_bp_pair_table — the breakpoint candidate of EVERY pair of local alignments of every chimeric read (K4).
 *
 * Replaces the arithmetic of alignment2bp (/root/reference/src/breakpoint_utilities.py:70-96, called per read inside the
 * interval search, infer_breakpoint_graph.py:432-434), of alignment2bp_l (bu:129-186, infer_breakpoint_graph.py:687-688)
 * and of interval2bp (bu:289-295): which pairs yield a candidate depends on the amplicon intervals of the moment, but the
 * candidate itself and every interval-independent test are pure functions of two table rows, so they are computed once.
 *   inputs  off int32[n_reads + 1], rows int32[n_rows][8] — exactly coral_sa_table's out_off / out_rows (device);
 *           chr_rank int32[n_tid]: rank of every BAM contig in chr1..22,X,Y,M (global_names.py:13-18), -1 = other contig
 *   output  pairs int32[2 * n_rows][8] (device, 16-byte aligned).  Slot 2 * g + 0 = pair (g, g + 1), slot 2 * g + 1 = pair
 *           (g - 1, g + 1) around row g ("skip one low-MAPQ alignment"); a slot whose pair leaves the read has bits = 0.
 *           Fields: c1, p1, c2, p2, query gap, bits, row a, row b with
 *           bits: 1 valid | 2 passes the MAPQ / query-gap tests | 4 o1 is '-' | 8 o2 is '-' | 16 ends swapped by interval2bp
 *                 | 32 strands differ | 64 |gr - grr| > max(gap_, |0.2 gr|) | 128 a contig outside chr1..22,X,Y,M (KeyError
 *                 in the reference) | mapq(a) << 8 | mapq(b) << 16
 * Asynchronous on `stream`.  No limit on the number of alignments per read.
 * ------------------------------------------------------------------------------------------------ */
int coral_bp_pair_table(int32_t n_reads, int32_t n_rows, const int32_t *off, const int32_t *rows, const int32_t *chr_rank,
                        int32_t n_tid, int32_t min_bp_match_cutoff, int32_t min_mapq, int32_t gap_, int32_t gap_mapq,
                        int32_t *pairs, void *stream);

/* ------------------------------------------------------------------------------------------------
 * coral_search_* — host side of the amplicon-interval search (no device work): one step of find_interval_i
 * (infer_breakpoint_graph.py:362-434) in one call, alignment2bp_l over all reads (ibg:676-690) in another.
 *
 * coral_search_create borrows HOST arrays (they must outlive the handle): the chimeric table (off int64[n_reads + 1];
 * per row: owning read, contig id, rint[1], rint[2], CN-segment index of both ends (-1 none, -3 contig without CN
 * segments)), per read hash(read name) and name id, the inverted index of hash_alignment_to_seg (e_key = contig << 32 |
 * segment, ascending; e_row = table row), the pair table (host copy of coral_bp_pair_table's output) and the CN segments
 * of every contig in file order (seg_off int64[n_tid + 1]; start, inclusive end).
 * coral_search_params (once, before the first step): min_cluster_cutoff / max_seq_len (ibg:385-419), the arguments of
 * coral_call_breakpoints (cluster distance, match cutoff, acceptance floor) and the number of look-ahead threads.
 * coral_search_step(tid, s, e, si, ei): reach sets (ibg:369-384, replayed as CPython sets: size and iteration order),
 * segments with fewer reads than the cutoff dropped, runs of neighbouring segments (ibg:392-419), for every run alignment2bp
 * of the united set's reads, in the set's iteration order, between the run and (tid, s, e), and coral_call_breakpoints on
 * every run's candidates (ibg:436-457; sub-cluster counter not advanced, Appendix A Q4).  A step is a pure function of its arguments: coral_search_prefetch asks for it to be computed
 * ahead on a worker thread, coral_search_step then waits for / takes over / computes it; results do not depend on timing.
 * coral_search_within: alignment2bp_l of every read in table order.  coral_search_between: alignment2bp
 * of the listed reads between two intervals.  coral_search_result then gives (valid until the next call on the handle)
 *   cand int64[n_cand][13] = c1, p1, o1, c2, p2, o2, read name id, i, j, query gap, swapped, mapq a, mapq b (the 11 fields
 *        of bu:81 / bu:294-295), the runs' candidates one after the other;
 *   meta int64[n_meta] = n_groups, then per run: contig id, first segment, last segment, n candidates, n clusters, n calls,
 *        the cluster sizes, and per call: head candidate, p1, p2, flags, begin and end of its support in `sup` (head and
 *        support indices are relative to the run's first candidate; flags as coral_call_breakpoints);
 *   stats double[6 per call], sup int64[n_sup];
 *   order_off int64[n_groups + 1] / order int32[] = the reads of every run in iteration order (table indices).
 * CORAL_ERR_FORMAT: a candidate touches a contig outside chr1..22,X,Y,M (KeyError at bu:293 in the reference).
 * ------------------------------------------------------------------------------------------------ */
void *coral_search_create(int64_t n_reads, int64_t n_rows, const int64_t *off, const int64_t *row_read, const int64_t *row_tid,
                          const int64_t *ra, const int64_t *rb, const int64_t *cni0, const int64_t *cni1,
                          const int64_t *read_hash, const int64_t *read_name, int64_t n_ent, const int64_t *e_key,
                          const int64_t *e_row, const int32_t *pairs, int32_t n_tid, const int64_t *seg_off,
                          const int64_t *seg_start, const int64_t *seg_end);
int coral_search_free(void *handle);
const char *coral_search_error(void *handle);
int coral_search_params(void *handle, double min_cluster_cutoff, int64_t max_seq_len, int64_t bp_distance_cutoff,
                        int64_t match_cutoff, double accept_floor, int32_t n_threads);
int coral_search_prefetch(void *handle, int64_t tid, int64_t s, int64_t e, int64_t si, int64_t ei);
int coral_search_step(void *handle, int64_t tid, int64_t s, int64_t e, int64_t si, int64_t ei);
int coral_search_within(void *handle, int32_t n_int, const int64_t *int_tid, const int64_t *int_start, const int64_t *int_end);
int coral_search_between(void *handle, int64_t n_sel, const int32_t *reads, int64_t t1, int64_t s1, int64_t e1, int64_t t2,
                         int64_t s2, int64_t e2);
int coral_search_result(void *handle, int64_t *n_meta, const int64_t **meta, int64_t *n_cand, const int64_t **cand,
                        int64_t *n_sup, const int64_t **sup, const double **stats, const int64_t **order_off,
                        const int32_t **order);
/* coral_search_bfs — the WHOLE interval search of a build in one call: the loop over the seed intervals and the breadth-first
 * search of find_interval_i (/root/reference/src/infer_breakpoint_graph.py:343-673) incl. its order-dependent half: addbp
 * (ibg:326-340), the refinement of the reached segments into new intervals (ibg:459-612), interval_exclusive (bu:54-67) and
 * the connection bookkeeping (ibg:614-673); the pure steps run ahead on the handle's worker threads as with
 * coral_search_prefetch / _step.  iv int64[n_seed][4] = contig id, start, end, ccid (-1) of the seeds after the CN-segment snap;
 * seg_cn / seg_ix = CN value and the reference's per-chromosome index of every CN segment (layout of seg_start / seg_end);
 * chr_rank[contig] = rank in chr1..22,X,Y,M or -1; tid_has_rows[contig] = 1 if some alignment is hashed to the contig's
 * segments; log_level 0 none, 1 warnings, 2 + debug events.  Errors: CORAL_ERR_FORMAT / -10 / -11 = the reference's KeyError
 * cases, -12 = its IndexError (coral_search_error has the text).  coral_search_bfs_get(which) returns the result arrays:
 *   0 intervals int64[n][5] (contig, start, end, ccid, start-is-a-bool — Appendix A Q2)
 *   1 breakpoints int64[n][11] (c1 p1 o1 c2 p2 o2, read name id / i / j of the head candidate, query gap, swapped)
 *   2 int64[n][4] (flags of coral_call_breakpoints, ccid, first and end chunk)   3 statistics double[n][6]
 *   4 chunks int64[n][2] (begin, end into 5 / 6 / 7)   5 / 6 / 7 support triples (read name id, i, j) — chunk 0 of a breakpoint
 *     is the set it was created with, every later chunk arrived through `|=` (ibg:330)
 *   8 connection keys int64[n][2] in insertion order   9 offsets int64[n + 1] into 10   10 breakpoint indices in order of addition
 *   11 events int64[n][6] for the log (type, arguments). */
int coral_search_bfs(void *handle, int32_t n_seed, const int64_t *iv, const double *seg_cn, const int64_t *seg_ix,
                     const int32_t *chr_rank, const uint8_t *tid_has_rows, double cn_gain, int64_t interval_delta,
                     int32_t log_level);
int coral_search_bfs_get(void *handle, int32_t which, const void **ptr, int64_t *n);

/* ------------------------------------------------------------------------------------------------
 * coral_read_counter — copy a device counter to the host (synchronises `stream`).
 * ------------------------------------------------------------------------------------------------ */
int coral_read_counter(const uint32_t *dev_counter, uint32_t *host_value, void *stream);

/* ------------------------------------------------------------------------------------------------
 * coral_cluster_first_fit — HOST function (no device work).
 *
 * Greedy first-fit clustering of one (chr1, chr2, o1, o2) group of breakpoint candidates, exactly as
 * /root/reference/src/breakpoint_utilities.py:268-282: candidate i joins the FIRST existing cluster that
 * has any member with |p1 - p1'| < cutoff and |p2 - p2'| < cutoff, else opens a new cluster.
 * cluster_of[i] receives the cluster ordinal (creation order), *n_clusters the number of clusters.
 * ------------------------------------------------------------------------------------------------ */
int coral_cluster_first_fit(int64_t n, const int64_t *p1, const int64_t *p2, int64_t cutoff,
                            int32_t *cluster_of, int32_t *n_clusters);

/* ------------------------------------------------------------------------------------------------
 * coral_first_seen_rows — HOST function.  is_first[i] = 1 iff row i of the row-major int64 matrix rows[n][ncols]
 * differs from every earlier row.  With the read-name id in column 0 and the tokenised SA entry in the others this
 * is the "if sa not in chimeric_alignments[rn]: append" de-duplication of
 * /root/reference/src/infer_breakpoint_graph.py:146-151, for all reads at once, exact (no hash-only equality).
 * ------------------------------------------------------------------------------------------------ */
int coral_first_seen_rows(int64_t n, int32_t ncols, const int64_t *rows, uint8_t *is_first);

/* ------------------------------------------------------------------------------------------------
 * coral_names_unify — HOST function: the read-name tables of `n_pieces` consecutive byte ranges of ONE BAM file (one per
 * rank, each decoded on its own GPU; piece p has n_names[p] names as blob[p] + off[p][n_names[p] + 1], numbered in order of
 * first appearance WITHIN the piece) -> the numbering a single decode of the whole file gives: name id = order of first
 * appearance over the file = insertion order of the reference's dicts keyed by query_name
 * (/root/reference/src/infer_breakpoint_graph.py:141-151).  lut[p][local id] receives the global id; out_blob (capacity:
 * the pieces' blob bytes together) / out_off (capacity: Σ n_names + 1) the global table; *n_global its size.
 * Exact: names are compared as bytes, a 64-bit hash only routes them; n_threads workers join hash partitions in parallel.
 * ------------------------------------------------------------------------------------------------ */
int coral_names_unify(int32_t n_pieces, const int64_t *n_names, const uint8_t *const *blob, const int64_t *const *off,
                      int32_t *const *lut, uint8_t *out_blob, int64_t *out_off, int64_t *n_global, int32_t n_threads);

/* ------------------------------------------------------------------------------------------------
 * coral_pyset_* — HOST functions: the iteration order of the Python sets of read names the reference builds and iterates
 * in its interval search (/root/reference/src/infer_breakpoint_graph.py:379-384 .add() per reached CN segment,
 * :405-419 `|=` unions, :428/432 iteration), obtained by replaying CPython 3.10's set algorithm on (item id, str hash)
 * pairs instead of creating the sets.  `create` builds one set per key from entries in insertion order and reports the
 * distinct counts; `union_order` returns list(set() | sets[keys[0]] | sets[keys[1]] | ...) as item ids in iteration
 * order (out_items needs room for the sum of the counts).
 * ------------------------------------------------------------------------------------------------ */
void *coral_pyset_batch_create(int64_t n_entries, const int32_t *key_of_entry, const int32_t *item,
                               const int64_t *item_hash, int32_t n_keys, int32_t *out_count);
int coral_pyset_union_order(void *handle, int32_t n_union, const int32_t *keys, int32_t *out_items, int32_t *out_n);
int coral_pyset_batch_free(void *handle);

/* Candidates -> clusters -> exact breakpoints in one call: cluster_bp_list (bu:252-286), then for every cluster the
 * sub-cluster loop of ibg:436-457 / :693-718 / :777-802 around bpc2bp (bu:299-388) and bp_match (bu:391-416).
 * field_ptr[13] / field_stride[13] (stride in elements): the candidate columns c1, p1, o1, c2, p2, o2, read, i, j, gap,
 * swapped, mapq_a, mapq_b as int64 (chromosomes as BAM tids < 64, orientation 0 '+' / 1 '-').
 * min_cluster_cutoff: minimum cluster size and first-sub-cluster support; accept_floor: support needed by later sub-clusters
 * (max(normal_cov * min_bp_cov_factor, 3)); advance_subcluster = 0 reproduces the BFS call site, which never increments its
 * sub-cluster counter (Appendix A Q4).
 * Outputs (caller allocates n entries each, call_sup_off n + 1, call_stats 6 n): cluster_size[*n_clusters] for the log;
 * for each accepted breakpoint k < *n_calls: call_head (candidate whose fields name the breakpoint), call_p1/p2 (positions),
 * call_stats[6k..] (mean p1, mean p2, sd p1, sd p2, mean mapq 1, mean mapq 2), call_flags bit0/bit1 = the sd was the
 * reference's "ValueError -> integer 0" case, and sup_idx[call_sup_off[k] .. call_sup_off[k+1]) = supporting candidates. */
int coral_call_breakpoints(int64_t n, const int64_t *const *field_ptr, const int64_t *field_stride, double min_cluster_cutoff,
                           int64_t bp_distance_cutoff, int64_t match_cutoff, double accept_floor, int32_t advance_subcluster,
                           int32_t *n_clusters, int32_t *cluster_size, int32_t *n_calls, int64_t *call_head, int64_t *call_p1,
                           int64_t *call_p2, double *call_stats, int32_t *call_flags, int64_t *call_sup_off, int64_t *sup_idx);

/* NM statistics of the mapped, non-chimeric (no SA tag) MAPQ-60 records, ibg:153-157: their count and the sums of
 * e = NM / query_length and of e * e, added in record order with one rounding per addition (the reference's sequential
 * `+=`).  Host arrays: tid / mapq / nm / qlen int32[n], sa_off int64[n + 1] (SA rows per record).
 * CORAL_ERR_ZERODIV when a counted record has query_length 0 (no SEQ): the reference raises ZeroDivisionError at ibg:154. */
int coral_nm_stats(int64_t n, const int32_t *tid, const int64_t *sa_off, const int32_t *mapq, const int32_t *nm,
                   const int32_t *qlen, int64_t *count, double *sum_e, double *sum_e2);

/* Long-read support of the concordant edges (ibg:1043-1055): for edge q, pt_rec[pt_begin[4q + d] .. pt_end[4q + d]) are the
 * record ordinals covering its position d (p, p + 1, p - 101, p + 101: coral_point_cover; the four ranges may be any slices of
 * the one int32 array pt_rec[n_pt_rec], equal points share theirs), rec_name int32[n_rec] maps records to read-name ids,
 * sup_name[sup_off[q] .. sup_off[q + 1]) are the name ids supporting a discordant edge at either node of the edge;
 * count[q] = number of distinct names covering all four positions and not among those.  Host arrays. */
int coral_concordant_counts(int32_t n_edges, const int64_t *pt_begin, const int64_t *pt_end, const int32_t *pt_rec, int64_t n_pt_rec,
                            const int32_t *rec_name, int64_t n_rec, int64_t n_names, const int64_t *sup_off, const int64_t *sup_name,
                            int64_t *count);

/* coral_independent_rows — HOST function: keep[i] = 1 for a maximal linearly independent subset of the rows of A (double[m][n],
 * taken in order; a row is kept when it is not in the span of the rows kept before it — Gram-Schmidt, re-orthogonalised once,
 * residual > tol * max(1, |row|)).  The redundant balance rows of the CN program (bg:526-541) are dropped with it before
 * coral_cn_solve.  Returns the number of rows kept, negative = bad arguments. */
int coral_independent_rows(int32_t m, int32_t n, const double *A, uint8_t *keep, double tol);

/* coral_cn_solve — HOST function: the CN assignment of one amplicon graph,  argmin Σ w_inv/x + w_lin·x − w_log·log x  s.t.
 * A x = 0, x > 0,  started at x = 1 — the convex program compute_cn_lr gives to cvxopt.solvers.cp
 * (/root/reference/src/breakpoint_graph.py:495-606; one variable per edge, one balance row per interior node).  A double[p][n]
 * row-major with linearly independent rows.  Infeasible-start Newton with a backtracking search on the KKT residual, until the
 * relative step is below 1e-13.  Returns 0 (x[n] and nu[p] filled, *n_iter = iterations), 1 = the reduced Newton system was singular
 * (the caller takes its general path), negative = bad arguments.  CN parity against cvxopt itself is unpinned. */
int coral_cn_solve(int32_t n, int32_t p, const double *w_inv, const double *w_lin, const double *w_log, const double *A,
                   int32_t max_iter, double *x, double *nu /* [p]: the multipliers of the balance rows */, int32_t *n_iter);

/* Reachable CN segments of one amplicon interval — the traversal of ibg:369-384 with the read-name sets replayed natively.
 * visit_rows[n_visit]: rows of the chimeric table hashed to segments si..ei of chromosome `tid`, in the reference's visiting
 * order (segment ascending, then append order).  row_read/row_tid/cni0/cni1 are per table row, off[n_reads + 1] the row
 * ranges per read, read_hash[n_reads] = hash(read name).  Every read is expanded once (first visit); it is added to the set
 * of each (chromosome, segment) it touches outside (si, ei) on `tid` (boundary segments count as outside, Appendix A Q9).
 * Returns a handle usable with coral_pyset_union_order / coral_pyset_batch_free (NULL on bad arguments); keys are numbered
 * in order of first appearance = insertion order of the reference's nested dicts. */
void *coral_reach_create(int64_t n_visit, const int64_t *visit_rows, const int64_t *row_read, const int64_t *off,
                         const int64_t *row_tid, const int64_t *cni0, const int64_t *cni1, int64_t n_reads, int64_t tid,
                         int64_t si, int64_t ei, const int64_t *read_hash, int32_t *n_keys_out);
/* codes[k] = tid << 32 | segment index, counts[k] = distinct reads of key k. */
int coral_reach_keys(void *handle, int64_t *codes, int32_t *counts);

/* ------------------------------------------------------------------------------------------------
 * coral_bam_decode_* — HOST functions: BAM/BGZF file -> structure-of-arrays records, decoded ONCE.
 *
 * Replaces pysam.AlignmentFile(path, 'rb') + the whole-file fetch() loop
 * (/root/reference/src/infer_breakpoint_graph.py:65, :140-158).  `open` inflates (n_threads zlib workers) and
 * parses the whole file; `sizes` reports {n_rec, n_cigar_words (padded), n_sa_rows, n_nonacgt, n_names,
 * names_bytes, n_ref, ref_names_bytes}; `fill` copies everything into caller-allocated arrays (read names as ONE blob of
 * names_bytes bytes without terminators + name_off[n_names + 1], name id = order of first appearance in the file, i.e. the
 * insertion order of the reference's dicts keyed by query_name, ibg:148-151; reference names as consecutive NUL-terminated
 * strings); `close` frees the handle.
 * SA rows are 8 ints: ref id, 1-based pos, strand (0 '+', 1 '-'), leading S, M, +I/-D, trailing S, mapq
 * (leading S = -2 marks a CIGAR that contains S and M but is not one of the nine shapes of
 * cigar_parsing.py:219-229).  qlen is l_seq, or the CIGAR-implied query length when SEQ is '*'.
 * ------------------------------------------------------------------------------------------------ */
int coral_bam_decode_open(const char *path, int32_t n_threads, void **handle);
int coral_bam_decode_sizes(void *handle, int64_t sizes[8]);
int coral_bam_decode_fill(void *handle, int32_t *tid, int32_t *pos, int32_t *end, int32_t *flag, int32_t *mapq,
                          int32_t *qlen, int32_t *has_seq, int32_t *nm, int32_t *name_id, int32_t *n_cigar,
                          int64_t *cigar_off, uint32_t *cigar, int64_t *sa_off, int32_t *sa, int32_t *sa_nm,
                          int64_t *nonacgt_rec, int32_t *nonacgt_pos, char *names, int64_t *name_off,
                          char *ref_names, int32_t *ref_lens);
int coral_bam_decode_close(void *handle);
/* The same for the rank-th of `world` byte ranges of the file (one process per GPU decodes only its share): the range starts
 * at the first BGZF block at or after its first byte and at the first record that starts in that block's inflated bytes or
 * later, and ends with the record that straddles into the next range; read-name ids are local to the range. */
int coral_bam_decode_range(const char *path, int32_t n_threads, int32_t rank, int32_t world, void **handle);
/* stats = compressed bytes, uncompressed bytes, BGZF blocks of the decoded range; *seconds = wall time of the decode. */
int coral_bam_decode_stats(void *handle, int64_t stats[3], double *seconds);
/* SoA records -> coordinate-sorted BAM (tests and benchmarks only; the product reads BAM).  Arrays as coral_bam_decode_fill
 * delivers them (names / ref_names: one C string per name id / contig); SEQ is hash-made ACGT with N at the listed aligned
 * positions, QUAL absent, tags NM:i, SA:Z, and CG:B,I for CIGARs of more than 65535 operations. */
int coral_bam_write(const char *path, int64_t n_rec, const int32_t *tid, const int32_t *pos, const int32_t *flag,
                    const int32_t *mapq, const int32_t *qlen, const int32_t *has_seq, const int32_t *nm, const int32_t *name_id,
                    const int32_t *n_cigar, const int64_t *cigar_off, const uint32_t *cigar, const int64_t *sa_off,
                    const int32_t *sa, const int32_t *sa_nm, int64_t n_nonacgt, const int64_t *nonacgt_rec,
                    const int32_t *nonacgt_pos, const char *const *names, int32_t n_ref, const char *const *ref_names,
                    const int32_t *ref_lens, uint32_t seed, int32_t level, int32_t n_threads);
const char *coral_bam_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * coral_bamgpu_* — the same decode with the inflate and the record parsing ON THE GPU: the host reads the file and sends
 * COMPRESSED bytes over PCIe; BGZF blocks are inflated one per wave (k_bgzf_inflate), record boundaries found, CIGARs laid
 * out in the padded SoA form of coral_records_t directly in HBM, and only a few hundred bytes per record (fixed fields,
 * read name, SA text) come back for the host-side fields.  Same replacement as coral_bam_decode_*
 * (/root/reference/src/infer_breakpoint_graph.py:65, :140-158), same results (tests/test_bam_gpu.py), same byte-range
 * rule for rank / world.  Every inflated block's CRC-32 is checked against its BGZF trailer (k_bgzf_crc), as htslib does.
 * The library allocates no device memory: the caller provides one workspace.
 *
 *   open   parse the header, size the batches (`batch_bytes` inflated bytes per batch; 0 = default 2.52 GiB, at most 3.5 GiB, never more than
 *          the byte range needs) -> *workspace_bytes the caller must allocate on the current device (256-byte aligned)
 *   start  take the workspace, start reading / uploading / inflating
 *   next   parse the next batch up to its sizes: out[0] records, out[1] padded CIGAR words, out[2] 1 = a batch is pending
 *          (0 = the file is done), on `stream` (synchronises it)
 *   emit   write the pending batch's CIGAR words to `cigar_dst` (device, out[1] words) and, if not NULL, its batch-local
 *          op offsets to `cigar_off_dst` (device, out[0] + 1 int64); takes the batch's host-side fields in
 *   host   the host-side half of the result as a handle for coral_bam_decode_sizes / _fill / _stats (cigar: pass NULL,
 *          n_cigar_words is 0; cigar_off covers the whole decoded range); owned by the decoder, do not close it
 *   stats  stats = batches, segments whose speculative record start was replaced by the exact walk, records fetched
 *          whole for the non-ACGT list, batch capacity; seconds = total wall time, host-side field handling, file reads,
 *          set-up of pinned buffers and streams, time the caller waited for the file feeder, ... for the GPU
 * coral_bgzf_inflate: one inflate launch over caller-provided device buffers — desc = n_blocks x {src_off, src_len,
 * dst_off, isize} uint32 (raw DEFLATE streams in `comp`, which must be readable 4096 bytes beyond the last stream);
 * status[b] = 0 or the decoder's error code.
 * ------------------------------------------------------------------------------------------------ */
int coral_bamgpu_open(const char *path, int32_t n_threads, int32_t rank, int32_t world, int64_t batch_bytes, void **handle,
                      int64_t *workspace_bytes);
int coral_bamgpu_start(void *handle, void *workspace, int64_t workspace_bytes);
int coral_bamgpu_next(void *handle, int64_t out[4], void *stream);
int coral_bamgpu_emit(void *handle, uint32_t *cigar_dst, int64_t *cigar_off_dst, void *stream);
int coral_bamgpu_host(void *handle, void **decoded);
int coral_bamgpu_stats(void *handle, int64_t stats[4], double seconds[6]);
int coral_bamgpu_close(void *handle);
int coral_bgzf_inflate(const uint8_t *comp, const uint32_t *desc, int32_t n_blocks, uint8_t *out, int32_t *status,
                       void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CORAL_HIP_H */
