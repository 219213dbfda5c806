"""Instruction mix of a kernel's persistent loop, from the compiler's own assembly (no GPU needed).

usage: python tools/isa_phase_counts.py vae_segmentation_amd/csrc/igemm_k3_bf16.hip 'k3t_kernel<0, false, 8, true, unsigned short, false>' [-DVS_DET_BUILD=1 ...]

Device-only compile to assembly (hipcc -S), the named kernel's body cut out, its LAST loop (the backward branch that spans the most s_barrier
instructions: the tile loop of the persistent kernels) split at its barriers, instructions counted by class.  What it is for: telling "the loop is bound by
instruction issue" from "the loop waits" before building a deeper pipeline — issue cycles per tile and wave = 4 x (VALU + packed VALU + MFMA issue slots) on a
16-lane SIMD, matrix-pipe cycles = 16 per v_mfma_f32_16x16x32 (8 passes would be 32 for the 32x32x16 shape); compare with the measured time per tile.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_pk_"):
        return "valu_packed"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_barrier"):
        return "s_barrier"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    src, want = sys.argv[1], sys.argv[2]
    extra = [a for a in sys.argv[3:] if a.startswith("-")]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "dev.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-variable", "--cuda-device-only", "-S",
                            "-I" + os.path.join(root, "include"), src, "-o", out] + extra, capture_output=True, text=True)
        if r.returncode:
            print(r.stderr[-3000:])
            sys.exit(r.returncode)
        text = open(out).read().split("\n")
    # kernel bodies: from "<mangled>:" to the matching .size directive
    starts = [(i, ln.split(":")[0]) for i, ln in enumerate(text) if re.match(r"^_Z\w+:", ln)]
    names = subprocess.run(["c++filt"] + [m for _, m in starts], capture_output=True, text=True).stdout.split("\n")
    norm = lambda s: re.sub(r"\s+", "", s)
    hit = [(i, m, n) for (i, m), n in zip(starts, names) if norm(want) in norm(n)]
    if len(hit) != 1:
        print("%d kernels match %r:" % (len(hit), want))
        for _, _, n in (hit or [(0, 0, n) for n in names if "kernel" in n][:80]):
            print("  ", n)
        sys.exit(1)
    i0, mangled, name = hit[0]
    i1 = next(i for i in range(i0, len(text)) if text[i].startswith("\t.size\t" + mangled))
    body = text[i0:i1]
    # instruction stream with label positions
    labels, ins = {}, []
    for ln in body:
        s = ln.strip()
        if not s or s.startswith((";", ".")) and not re.match(r"^\.LBB\d+_\d+:", s):
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if re.match(r"^[a-z]", s):
            ins.append(s)
    # backward branches = loops; pick the one containing the most barriers (ties: the longest)
    best = None
    for k, s in enumerate(ins):
        m = re.match(r"^s_c?branch\w*\s+(\.LBB\d+_\d+)", s)
        if m and m.group(1) in labels and labels[m.group(1)] <= k:
            a = labels[m.group(1)]
            nb = sum(1 for t in ins[a:k + 1] if t.startswith("s_barrier"))
            key = (nb, k - a)
            if best is None or key > best[0]:
                best = (key, a, k)
    if best is None:
        print("no loop found")
        sys.exit(1)
    _, a, b = best
    loop = ins[a:b + 1]
    print("%s\nloop: %d instructions, %d barriers (kernel body: %d instructions)" % (name.strip(), len(loop), best[0][0], len(ins)))
    phases, cur = [], collections.Counter()
    for s in loop:
        c = classify(s.split()[0])
        if c == "s_barrier":
            phases.append(cur)
            cur = collections.Counter()
        else:
            cur[c] += 1
    phases.append(cur)
    keys = ["valu", "valu_packed", "mfma", "lds", "vmem", "salu", "s_waitcnt", "other"]
    print("%-28s" % "phase (between barriers)" + "".join("%12s" % k for k in keys))
    tot = collections.Counter()
    for n, p in enumerate(phases):
        print("%-28s" % ("%d" % n) + "".join("%12d" % p[k] for k in keys))
        tot.update(p)
    print("%-28s" % "loop total" + "".join("%12d" % tot[k] for k in keys))
    issue = 4 * (tot["valu"] + tot["valu_packed"] + tot["mfma"])
    print("per iteration and wave: %d vector/matrix issue cycles (4 per wave64 instruction), %d matrix-pipe cycles (16 per 16x16x32 MFMA); at 2.4 GHz: %.2f / %.2f us"
          % (issue, 16 * tot["mfma"], issue / 2400.0, 16 * tot["mfma"] / 2400.0))


if __name__ == "__main__":
    main()
