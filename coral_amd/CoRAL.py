#!/usr/bin/env python3
"""`CoRAL.py reconstruct` with the MI355X graph build (same flags as /root/reference/src/CoRAL.py:80-109).

    python -m coral_amd.CoRAL reconstruct --lr_bam x.bam --cnv_seed seeds.bed --cn_seg cn.bed --output_prefix out \\
        --skip_cycle_decomp

The `reconstruct` mode and the `hsr` mode (§8(f) item 3) run on the MI355X path; the other modes of the
reference (seed, plot, cycle2bed) are untouched and are delegated to the reference's own modules when they are
importable (set CORAL_REFERENCE_SRC to the reference's src/ directory).  The cycle-decomposition step after the graph build is the
reference's (Gurobi); it runs on the object this module returns.
"""
import argparse
import os
import sys

# host side = one thread + a few native workers: keep the per-core OpenMP / BLAS pools from spinning (coral_amd/hostpools.py)
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1" if _v.startswith("OPENBLAS") else "4")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # (ROCm's default of 4 hardware queues makes the decoder's and the build's streams share queues)


def print_args(args_dict):
    for key, value in vars(args_dict).items():
        print(f"{key}: {value}")
    print()


def build_parser():
    parser = argparse.ArgumentParser(description="Long-read amplicon reconstruction pipeline and associated utilities.")
    sub = parser.add_subparsers(dest="mode", help="Select mode.")
    rp = sub.add_parser("reconstruct", help="Reconstruct focal amplifications")
    rp.add_argument("--lr_bam", help="Sorted indexed (long read) bam file.", required=True)
    rp.add_argument("--cnv_seed", help="Bed file of CNV seed intervals.", required=True)
    rp.add_argument("--output_prefix", help="Prefix of output files.", required=True)
    rp.add_argument("--cn_seg", help="Long read segmented whole genome CN calls (.bed or CNVkit .cns file).", required=True)
    rp.add_argument("--output_bp", help="If specified, only output the list of breakpoints.", action='store_true')
    rp.add_argument("--skip_cycle_decomp", help="If specified, only reconstruct and output the breakpoint graph for all amplicons.",
                    action='store_true')
    rp.add_argument("--output_all_path_constraints", help="If specified, output all path constraints in *.cycles file.",
                    action='store_true')
    rp.add_argument("--min_bp_support", help="Ignore breakpoints with less than (min_bp_support * normal coverage) long read support.",
                    type=float, default=1.0)
    rp.add_argument("--cycle_decomp_alpha", help="Parameter used to balance CN weight and path constraints in greedy cycle extraction.",
                    type=float, default=0.01)
    rp.add_argument("--cycle_decomp_time_limit", help="Maximum running time (in seconds) reserved for integer program solvers.",
                    type=int, default=7200)
    rp.add_argument("--cycle_decomp_threads", help="Number of threads reserved for integer program solvers.", type=int)
    rp.add_argument("--postprocess_greedy_sol", help="Postprocess the cycles/paths returned in greedy cycle extraction.",
                    action='store_true')
    rp.add_argument("--log_fn", help="Name of log file.")
    rp.add_argument("--device", help="GPU to use (MI355X build only option).", default="cuda:0")
    hp = sub.add_parser("hsr", help="Detect possible integration points of ecDNA HSR amplifications.")     # CoRAL.py:112-120
    hp.add_argument("--lr_bam", help="Sorted indexed long read bam file.", required=True)
    hp.add_argument("--cycles", help="AmpliconSuite-formatted cycles file", required=True)
    hp.add_argument("--cn_seg", help="Long read segmented whole genome CN calls (.bed or CNVkit .cns file).", required=True)
    hp.add_argument("--output_prefix", help="Prefix of output file name.", required=True)
    hp.add_argument("--normal_cov", help="Estimated diploid coverage.", required=True)
    hp.add_argument("--bp_match_cutoff", help="Breakpoint matching cutoff.", type=int, default=100)
    hp.add_argument("--bp_match_cutoff_clustering", help="Crude breakpoint matching cutoff for clustering.", type=int, default=2000)
    hp.add_argument("--device", help="GPU to use (MI355X build only option).", default="cuda:0")
    for mode in ("seed", "plot", "cycle2bed"):
        sub.add_parser(mode, help="(reference implementation; not part of the MI355X path)", add_help=False)
    return parser


def reconstruct_mode(args):
    print("Performing reconstruction with options:")
    print_args(args)
    from coral_amd import infer_breakpoint_graph
    b2bn = infer_breakpoint_graph.reconstruct_graph(args)
    if not (args.output_bp or args.skip_cycle_decomp):
        ref = os.environ.get("CORAL_REFERENCE_SRC")
        if ref and ref not in sys.path:
            sys.path.insert(0, ref)
        try:
            import cycle_decomposition            # the reference's module (needs gurobipy)
        except ImportError as e:
            raise SystemExit("cycle decomposition is the reference's Gurobi step and is not available here (%s); "
                             "re-run with --skip_cycle_decomp or --output_bp" % e)
        cycle_decomposition.reconstruct_cycles(args, b2bn)
    b2bn.closebam()
    infer_breakpoint_graph.print_complete_message()
    print("\nCompleted reconstruction.")
    return b2bn


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if argv and argv[0] in ("seed", "plot", "cycle2bed"):
        ref = os.environ.get("CORAL_REFERENCE_SRC")
        if not ref:
            raise SystemExit("mode '%s' is the reference's own code; set CORAL_REFERENCE_SRC to its src/ directory" % argv[0])
        import runpy
        sys.argv = [os.path.join(ref, "CoRAL.py")] + list(argv)
        sys.path.insert(0, ref)
        runpy.run_path(os.path.join(ref, "CoRAL.py"), run_name="__main__")
        return None
    parser = build_parser()
    args = parser.parse_args(argv)
    if args.mode == "reconstruct":
        return reconstruct_mode(args)
    if args.mode == "hsr":
        print("Performing HSR mode with options:")              # CoRAL.py:34-38
        print_args(args)
        from coral_amd import hsr
        return hsr.locate_hsrs(args)
    parser.print_help()
    return None


if __name__ == '__main__':
    main()
