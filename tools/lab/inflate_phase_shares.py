"""Where a wave of k_bgzf_inflate spends a block's time: dynamic header / table build / symbol loop.

Lab tool, two steps (the TIMING build is not the product library: its kernel writes clock sums where the status goes):

    python tools/lab/inflate_phase_shares.py --build     # here: patched copies of the two sources -> variants/libcoral_timing.so
    CORAL_LIB=variants/libcoral_timing.so python tools/lab/inflate_phase_shares.py      # on the GPU box

The patch puts `w.mark(k)` calls (s_memrealtime, 100 MHz) around read_dynamic_header / build / codes in Inflater::run and makes the
kernel store, for block b, the sum number b % 3 instead of the status.  Round 3: header 5.4 %, table build 2.3 %, symbol loop 92.2 % of a
block's time (profiles/r03_pmc_inflate.md).
"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build():
    csrc = os.path.join(ROOT, "coral_amd", "csrc")
    core = open(os.path.join(csrc, "coral_inflate_core.h")).read()
    old = """                rc = type == 1 ? fixed_tables() : read_dynamic_header(&n_ll, &n_dist);
                if (rc == OK) rc = build(T->lens, n_ll, T->ll_count, T->ll_sym, T->ll, nullptr, LL_BITS);
                if (rc == OK) rc = build(T->lens + n_ll, n_dist, T->d_count, T->ll_sym + 288, nullptr, T->dt, D_BITS);
                if (rc == OK) rc = codes();"""
    assert old in core, "Inflater::run changed: adapt the patch"
    core = core.replace(old, """                w.mark(0);
                rc = type == 1 ? fixed_tables() : read_dynamic_header(&n_ll, &n_dist);
                w.mark(1);
                if (rc == OK) rc = build(T->lens, n_ll, T->ll_count, T->ll_sym, T->ll, nullptr, LL_BITS);
                if (rc == OK) rc = build(T->lens + n_ll, n_dist, T->d_count, T->ll_sym + 288, nullptr, T->dt, D_BITS);
                w.mark(2);
                if (rc == OK) rc = codes();
                w.mark(3);""")
    work = os.path.join(ROOT, "build", "variants")
    os.makedirs(work, exist_ok=True)
    os.makedirs(os.path.join(ROOT, "variants"), exist_ok=True)
    core_path = os.path.join(work, "coral_inflate_core_timing.h")
    open(core_path, "w").write(core)
    hip = open(os.path.join(csrc, "coral_bamgpu.hip")).read()
    assert '#include "coral_inflate_core.h"' in hip
    hip = hip.replace('#include "coral_inflate_core.h"', '#include "%s"' % core_path)
    old = "    static constexpr bool vector_loop = true;\n    int lane;"
    assert old in hip
    hip = hip.replace(old, """    static constexpr bool vector_loop = true;
    long long t_last = 0, t_acc[3] = {0, 0, 0};
    __device__ __forceinline__ void mark(int k) {
        const long long now = (long long)wall_clock64();
        if (k > 0) t_acc[k - 1] += now - t_last;
        t_last = now;
    }
    int lane;""")
    old = "    if (lane == 0) status[b] = rc;\n}"
    assert hip.count(old) == 1
    hip = hip.replace(old, "    if (lane == 0) status[b] = (int)w.t_acc[b % 3];\n    (void)rc;\n}")
    hip_path = os.path.join(csrc, "coral_bamgpu_timing_tmp.hip")          # (next to its includes)
    open(hip_path, "w").write(hip)
    try:
        obj = os.path.join(work, "bamgpu_timing.o")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-ffp-contract=off",
                               "-c", hip_path, "-o", obj], cwd=ROOT)
    finally:
        os.remove(hip_path)
    others = [o for o in sorted(glob.glob(os.path.join(ROOT, "build", "obj", "*.o"))) if "bamgpu" not in o]
    assert others, "build the product first (__graft_entry__.build())"
    out = os.path.join(ROOT, "variants", "libcoral_timing.so")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj] + others + ["-lz", "-lpthread"], cwd=ROOT)
    print("built", out)


def measure():
    import runpy
    os.environ["CORAL_INFLATE_ABLATE"] = "1"            # (bench_inflate.py then does not read the status words as statuses)
    assert os.environ.get("CORAL_LIB"), "CORAL_LIB=variants/libcoral_timing.so"
    sys.argv = ["bench_inflate.py", "30000", "1", "2"]
    g = runpy.run_path(os.path.join(ROOT, "tools", "bench_inflate.py"))
    st = g["status"].cpu().numpy().astype("float64")
    h, b, c = st[0::3].mean(), st[1::3].mean(), st[2::3].mean()
    print("ticks of 10 ns per block (means): header %.0f  table build %.0f  symbol loop %.0f  -> shares %.3f %.3f %.3f"
          % (h, b, c, h / (h + b + c), b / (h + b + c), c / (h + b + c)))


if __name__ == "__main__":
    build() if "--build" in sys.argv else measure()
