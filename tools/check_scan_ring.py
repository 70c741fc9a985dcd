"""Build-time proof obligations of the scan kernel's hand-counted load ring.

`k_cigar_scan_v4<RING>` (coral_amd/csrc/coral_kernels.hip) issues its 1 KiB wave loads through `asm volatile` statements the
compiler does not see as loads, keeps RING - 1 of them in flight and retires them with its own `s_waitcnt vmcnt(RING - 1)`.
That is correct only if, in the code the compiler actually emitted,

  (1) no instruction reads or writes a register of a ring quad between the quad's load and the point where that load is
      guaranteed to have completed (a copy, a spill, a re-used temporary or an early use would all be silent corruption);
  (2) the kernel has no scratch (a spilled ring register is a read of an in-flight register);
  (3) the ring is what the source says: RING - 1 loads in the prologue, RING loads + RING counted waits per unrolled round.

This script checks exactly that on the device code of the built library: it disassembles the gfx950 code object of
coral_kernels.hip (llvm-objdump), rebuilds the control-flow graph of the kernel from the branch targets and runs a forward
data-flow analysis over ALL paths.  The abstract state is the ordered list of ring quads whose loads may still be in flight.
Vector-memory loads return in order, so after `s_waitcnt vmcnt(N)` every load with at least N younger loads has completed —
stores and atomics are ignored (they return out of order and can only make a wait more conservative), other loads count as
younger loads.  Any mention of a VGPR of an in-flight quad is an error.  Run by __graft_entry__.build(); a failure fails the
build here, in the GPU-less container, instead of on the GPU box.

    python tools/check_scan_ring.py <object or shared library> [RING ...]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def device_code_objects(path):
    """The gfx950 code objects embedded in a host object / shared library (one per .hip translation unit)."""
    out = []
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, path], check=True)
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        for k, a in enumerate(starts):
            piece = os.path.join(d, "bundle%d" % k)
            with open(piece, "wb") as fp:
                fp.write(blob[a:starts[k + 1] if k + 1 < len(starts) else len(blob)])
            co = os.path.join(d, "co%d" % k)
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + piece,
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True)
            if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co):
                dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
                notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
                out.append((dis, notes))
    return out


_INS = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def parse_kernel(dis, symbol_part):
    """[(addr, mnemonic, operand text)] of the first function whose mangled name contains ``symbol_part``."""
    ins, on = [], False
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            if on:
                break
            on = symbol_part in m.group(1)
            continue
        if on:
            m = _INS.match(line)
            if m:
                ops = re.sub(r"<[^>]*>", "", m.group(2))
                ins.append((int(m.group(3), 16), m.group(1), ops))
    return ins


def vregs(text):
    s = set()
    for m in _VREG.finditer(text):
        if m.group(1) is not None:
            s.add(int(m.group(1)))
        else:
            s.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return s


def check(ins, ring):
    addr_index = {a: k for k, (a, _, _) in enumerate(ins)}
    # the ring's loads are the kernel's only non-temporal loads (the exact path's compiler-tracked load is a plain one)
    is_ring_load = lambda k: ins[k][1] == "global_load_dwordx4" and ins[k][2].rstrip().endswith(" nt")
    errors = []
    n = len(ins)

    def succ(k):
        a, mn, ops = ins[k]
        if mn == "s_endpgm":
            return []
        if mn.startswith("s_cbranch") or mn == "s_branch":
            off = int(ops.split()[0])
            if off >= 0x8000:
                off -= 0x10000
            t = a + 4 + 4 * off
            if t not in addr_index:
                errors.append("branch at %#x to %#x leaves the kernel" % (a, t))
                return [k + 1] if mn != "s_branch" and k + 1 < n else []
            return [addr_index[t]] + ([k + 1] if mn != "s_branch" and k + 1 < n else [])
        if mn in ("s_setpc_b64", "s_swappc_b64"):
            errors.append("indirect jump / call at %#x: the analysis needs a closed control-flow graph" % a)
            return []
        return [k + 1] if k + 1 < n else []

    # The AMDGPU structurizer routes control flow through boolean SGPR pairs: `s_mov_b64 s[a:b], 0 | -1` on the way in,
    # `s_and_b64 vcc, exec, s[a:b]` + `s_cbranch_vccz / vccnz` at the join.  Half of those branches are infeasible on any given
    # path, so the analysis propagates exactly these constants (and nothing else) and follows a vcc branch one way when vcc is
    # known: flag pairs = the SGPR pairs that feed such an s_and_b64 somewhere in the kernel.
    flag_pairs = set()
    for _, mn, ops in ins:
        m = re.match(r"vcc, exec, s\[(\d+):(\d+)\]$", ops) if mn == "s_and_b64" else None
        if m:
            flag_pairs.add((int(m.group(1)), int(m.group(2))))
    flag_regs = {r: p for p in flag_pairs for r in range(p[0], p[1] + 1)}
    _sreg = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")

    def sregs_of(text):
        out = set()
        for m in _sreg.finditer(text):
            if m.group(1) is not None:
                out.add(int(m.group(1)))
            else:
                out.update(range(int(m.group(2)), int(m.group(3)) + 1))
        return out

    # forward data flow: state = (in-flight loads oldest first, known flag constants, what is known of vcc); a load entry is a
    # frozenset of VGPRs (ring quad) or None (a compiler-tracked load: only counts as a younger load)
    seen = [set() for _ in range(n)]
    work = [(0, ((), frozenset(), None))]
    parent = {}                                  # (instruction, state) at a block entry -> the (instruction, state) it was reached from
    trace = os.environ.get("CORAL_RING_TRACE") == "1"

    def path_to(k, full, entry):
        out, cur = ["%#x" % ins[k][0]], entry
        while cur in parent and len(out) < 60:
            out.append("%#x" % ins[cur[0]][0])
            cur = parent[cur]
        return " <- ".join(out)

    n_states = 0
    while work:
        k, full = work.pop()
        entry = (k, full)
        while True:
            if full in seen[k]:
                break
            seen[k].add(full)
            n_states += 1
            if n_states > 4000000:
                errors.append("state space of the analysis exploded")
                return errors
            st, consts, vcc = full
            a, mn, ops = ins[k]
            used = vregs(ops)
            if is_ring_load(k):
                dest = frozenset(vregs(ops.split(",")[0]))
                for q in st:
                    if q is not None and q & (used | dest):
                        errors.append("%#x: ring load touches v%s while its previous load may be in flight" % (a, sorted(q & (used | dest))))
                st = st + (dest,)
            else:
                for q in st:
                    if q is not None and q & used:
                        errors.append("%#x: `%s %s` touches v%s of a ring quad whose load may still be in flight" % (a, mn, ops, sorted(q & used))
                                      + ("\n      block entries on the way (newest first): " + path_to(k, full, entry) +
                                         "\n      in flight: %s" % [sorted(x) if x else None for x in st] if trace else ""))
                if mn == "s_waitcnt":
                    m = re.search(r"vmcnt\((\d+)\)", ops)
                    if m:
                        keep = int(m.group(1))
                        st = st[len(st) - keep:] if keep else ()
                elif re.match(r"(global|flat|buffer|scratch)_load", mn):
                    st = st + (None,)
            while st and st[0] is None:          # a tracked load with nothing of ours in front of it is nobody's business
                st = st[1:]
            # ---- flag constants and vcc
            is_branch = mn.startswith("s_cbranch") or mn == "s_branch"
            if not is_branch and mn not in ("s_waitcnt", "s_nop"):
                first = ops.split(",")[0] if ops else ""
                m = re.match(r"s\[(\d+):(\d+)\], (0|-1)$", ops) if mn == "s_mov_b64" else None
                if m and (int(m.group(1)), int(m.group(2))) in flag_pairs:
                    pr = (int(m.group(1)), int(m.group(2)))
                    consts = frozenset({c for c in consts if c[0] != pr} | {(pr, int(m.group(3)))})
                else:
                    hit = {flag_regs[r] for r in sregs_of(first) if r in flag_regs}
                    if "saveexec" in mn or mn.startswith("s_swap"):
                        hit |= {flag_regs[r] for r in sregs_of(ops) if r in flag_regs}
                    if hit:
                        consts = frozenset(c for c in consts if c[0] not in hit)
                m = re.match(r"vcc, exec, s\[(\d+):(\d+)\]$", ops) if mn == "s_and_b64" else None
                if m:
                    known = dict(consts).get((int(m.group(1)), int(m.group(2))))
                    vcc = None if known is None else ("zero" if known == 0 else "nonzero")
                elif "vcc" in ops:
                    vcc = None
            if len(st) > 4 * ring + 8:
                errors.append("%#x: more than %d loads in flight without a wait" % (a, len(st)))
                break
            if len(errors) > 20:
                return errors
            nx = succ(k)
            if len(nx) == 2 and vcc is not None and mn in ("s_cbranch_vccz", "s_cbranch_vccnz"):
                taken = (vcc == "zero") == (mn == "s_cbranch_vccz")
                nx = [nx[0]] if taken else [nx[1]]
            full = (st, consts, vcc)
            if not nx:
                if mn == "s_endpgm" and any(q is not None for q in st):
                    errors.append("%#x: kernel ends with ring loads in flight" % a + ("\n      block entries on the way (newest first): " +
                                  path_to(k, full, entry) + "\n      in flight: %s" % [sorted(x) if x else None for x in st] if trace else ""))
                break
            for j in nx[1:]:
                work.append((j, full))
                parent.setdefault((j, full), entry)
            if len(nx) > 1 or is_branch:
                parent.setdefault((nx[0], full), entry)
                entry = (nx[0], full)
            k = nx[0]
    return errors


def verify(dis, notes, ring, verbose=True):
    sym = "k_cigar_scan_v4ILi%dEE" % ring
    ins = parse_kernel(dis, sym)
    if not ins:
        return ["kernel %s not found in the code object" % sym]
    errs = []
    ring_loads = [k for k, (a, mn, ops) in enumerate(ins) if mn == "global_load_dwordx4" and ops.rstrip().endswith(" nt")]
    waits = [k for k, (a, mn, ops) in enumerate(ins) if mn == "s_waitcnt" and re.search(r"vmcnt\(%d\)" % (ring - 1), ops)]
    if len(ring_loads) != 2 * ring - 1:
        errs.append("expected %d ring loads (%d prologue + %d per unrolled round), found %d" % (2 * ring - 1, ring - 1, ring, len(ring_loads)))
    if len(waits) != ring:
        errs.append("expected %d counted waits s_waitcnt vmcnt(%d), found %d" % (ring, ring - 1, len(waits)))
    # in the unrolled round every counted wait directly follows its ring load (issue, then wait for the oldest)
    for w in waits:
        if w - 1 not in ring_loads:
            errs.append("%#x: counted wait is not directly behind a ring load" % ins[w][0])
    quads = {frozenset(vregs(ins[k][2].split(",")[0])) for k in ring_loads}
    if len(quads) != ring or any(len(q) != 4 for q in quads):
        errs.append("expected %d distinct ring register quads, found %s" % (ring, sorted(sorted(q) for q in quads)))
    if any(mn.startswith("scratch_") or "buffer_" in mn and "offen" in ops and "s[0:3]" in ops for _, mn, ops in ins):
        errs.append("the kernel has scratch traffic (a spilled ring register is a read of an in-flight register)")
    m = re.search(r"\.name:\s+_Z15%s.*?\n(?:.*\n)*?.*?\.private_segment_fixed_size:\s+(\d+)" % sym, notes)
    m2 = None
    for blk in notes.split("- .agpr_count")[1:] if "- .agpr_count" in notes else notes.split("  - .args")[1:]:
        if sym in blk:
            m2 = re.search(r"\.private_segment_fixed_size:\s*(\d+)", blk)
    priv = int(m2.group(1)) if m2 else (int(m.group(1)) if m else None)
    if priv is None:
        errs.append("private segment size of the kernel not found in the code object's metadata")
    elif priv != 0:
        errs.append("private segment (scratch) size is %d bytes, expected 0" % priv)
    errs += check(ins, ring)
    if verbose and not errs:
        print("check_scan_ring: k_cigar_scan_v4<%d>: %d instructions, %d ring loads into %d quads, %d counted waits, no scratch, "
              "no access to an in-flight ring register on any path" % (ring, len(ins), len(ring_loads), len(quads), len(waits)))
    return errs


def main(path, rings=(6,)):
    objs = device_code_objects(path)
    cand = [(d, nt) for d, nt in objs if "k_cigar_scan_v4" in d]
    if not cand:
        raise SystemExit("check_scan_ring: no code object with k_cigar_scan_v4 in %s" % path)
    bad = []
    for r in rings:
        for e in verify(cand[0][0], cand[0][1], r):
            bad.append("k_cigar_scan_v4<%d>: %s" % (r, e))
    if bad:
        raise SystemExit("check_scan_ring FAILED (the compiled scan kernel does not keep its load ring intact):\n  " + "\n  ".join(bad))


if __name__ == "__main__":
    main(sys.argv[1], tuple(int(x) for x in sys.argv[2:]) or (6,))
