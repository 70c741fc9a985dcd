#!/usr/bin/env python3
"""ThreadSanitizer run of the library's HOST code (CPU only; the pool refuses GPU sanitizer runs).

Builds coral_host / coral_search / coral_bpcall / coral_bam with `g++ -fsanitize=thread` into a scratch library — the device
entry points of include/coral_hip.h are stubbed (they are never called by the `-m "not gpu"` tests) — and runs the tests of
the threaded host paths against it: the interval search's look-ahead threads + helper pool + native BFS (tests/test_search*.py),
the name join of the multi-GPU merge (tests/test_names.py), the host BAM decoder's inflate threads and the CN solver.  Reports that lie wholly inside libtorch /
libgomp (un-instrumented OpenMP) are listed apart: they are not this library's.

    python tools/tsan_host.py [--out profiles/r03_tsan_host.txt] [tests ...]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_SOURCES = ("coral_host.cpp", "coral_search.cpp", "coral_bpcall.cpp", "coral_bam.cpp")
DEFAULT_TESTS = ("tests/test_search_bfs.py", "tests/test_search.py", "tests/test_names.py", "tests/test_cn_solver.py",
                 "tests/test_bam_spec_shapes.py", "tests/test_bam_io.py", "tests/test_bpcall.py")


def write_stubs(path):
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "coral_hip.h")).read(), flags=re.S)
    protos = re.findall(r"\n((?:int|void \*|const char \*)\s*coral_[a-z0-9_]+\s*\([^;]*\);)", hdr)
    src = "".join(open(os.path.join(ROOT, "coral_amd", "csrc", f)).read() for f in HOST_SOURCES)
    defined = set(re.findall(r'extern "C"[^\n]*?\b(coral_[a-z0-9_]+)\s*\(', src))
    out = ['#include "%s"' % os.path.join(ROOT, "include", "coral_hip.h"),
           "// device entry points: absent from the sanitizer build, never called by the CPU tests"]
    n = 0
    for p in protos:
        name = re.search(r"(coral_[a-z0-9_]+)\s*\(", p).group(1)
        if name in defined:
            continue
        body = 'return "tsan host build";' if p.startswith("const char") else "return 0;"
        out.append('extern "C" %s { %s }' % (p.rstrip(";"), body))
        n += 1
    open(path, "w").write("\n".join(out) + "\n")
    return len(protos), n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("tests", nargs="*", default=list(DEFAULT_TESTS))
    a = ap.parse_args()
    work = tempfile.mkdtemp(prefix="coral_tsan_")
    stubs = os.path.join(work, "stubs.cpp")
    n_protos, n_stubs = write_stubs(stubs)
    lib = os.path.join(work, "libcoral_tsan.so")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-fPIC", "-shared", "-ffp-contract=off", "-o", lib]
    cmd += [os.path.join(ROOT, "coral_amd", "csrc", f) for f in HOST_SOURCES] + [stubs, "-lz", "-lpthread"]
    subprocess.check_call(cmd)
    runner = os.path.join(work, "run.py")
    open(runner, "w").write(
        "import sys\nsys.path.insert(0, %r)\nimport coral_amd._lib as L\nL.LIB_PATH = %r\nimport pytest\n"
        "sys.exit(pytest.main(['-x', '-q', '-m', 'not gpu', '-p', 'no:cacheprovider'] + sys.argv[1:]))\n" % (ROOT, lib))
    tsan = subprocess.check_output(["gcc", "-print-file-name=libtsan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=tsan,
               TSAN_OPTIONS="halt_on_error=0 exitcode=0 report_signal_unsafe=0 history_size=4 log_path=%s" % os.path.join(work, "report"))
    r = subprocess.run([sys.executable, runner] + a.tests, cwd=ROOT, env=env, text=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    ours, theirs = [], []
    for f in sorted(os.listdir(work)):
        if not f.startswith("report."):
            continue
        for rep in open(os.path.join(work, f)).read().split("==================\n"):
            if "WARNING: ThreadSanitizer" not in rep:
                continue
            summary = next((ln for ln in rep.splitlines() if ln.startswith("SUMMARY:")), "SUMMARY: ?")
            (ours if "libcoral_tsan" in rep else theirs).append(summary)
    lines = ["ThreadSanitizer, host code of the library (%s; %d of %d entry points stubbed as device-only)" % (", ".join(HOST_SOURCES), n_stubs, n_protos),
             "tests: %s" % " ".join(a.tests), "pytest: %s (exit %d)" % (tail, r.returncode),
             "reports with a frame in the library: %d" % len(ours)] + ["  " + s for s in ours]
    lines += ["reports wholly inside other libraries (libtorch_cpu / libgomp, not instrumented): %d" % len(theirs)] + ["  " + s for s in sorted(set(theirs))]
    text = "\n".join(lines) + "\n"
    sys.stdout.write(text)
    if a.out:
        open(os.path.join(ROOT, a.out), "w").write(text)
    return 1 if (ours or r.returncode) else 0


if __name__ == "__main__":
    sys.exit(main())
