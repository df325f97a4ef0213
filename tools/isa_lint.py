#!/usr/bin/env python3
"""ISA lint for one code-generation defect of the gfx950 compiler of this image (ROCm 7.2 clang) that this kernel family has
hit twice in rounds 2-3 on hardware only and twice more in round 4 at build time (docs/EXPERIMENTS.md, "The hardware-only failures"):

    s_cbranch_execz .LBB9_156          ; no lane takes the `if`: jump to the join block with EXEC = 0
    ...
  .LBB9_156:                           ; join block of the `if`
    v_accvgpr_write_b32 a86, v152      ; <-- live-range split copy (VGPR parked in an AGPR across a call) ...
    s_or_b64 exec, exec, s[2:3]        ; <-- ... placed BEFORE the EXEC mask of the join is restored

The copy executes for the lanes that took the `if` only (none, typically: a rarely violated limit), the reload after
the call executes for all of them, and those lanes read whatever the AGPR held before -- here the lane id itself came
back as 0 for every active lane, so fifty-two lanes passed `lane == 0`.  Register pressure decides which value is parked
where, which is why the symptom moved with every unrelated source change (and vanished under printf).

The lint reads device assembly (hipcc -S --cuda-device-only) and reports every control-flow join block -- a label some
`s_cbranch_execz` jumps to (the skip edge of an `if`: such a block has two predecessors, so the compiler can never have
merged the body of the `if` into it) -- that writes vector registers or memory lane-wise before its first
`s_or_b64 exec, exec, ...`.  `--build` compiles topay_amd/csrc/topay_hip.hip with the product flags first.  Exit status
1 when anything is found (the build refuses such a library: __graft_entry__.build()).
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LANEWISE = re.compile(r"^(v_(?!readlane|readfirstlane|writelane)|ds_|scratch_|global_|flat_|buffer_)")
EXEC_WRITE = re.compile(r"^s_\w+\s+(exec\b|s\[\d+:\d+\],\s*(exec|-1|s\[\d+:\d+\])\s*$)|^s_(and|or|xor|andn2|orn2|nand|nor|xnor)_saveexec")


def lint(path):
    lines = open(path).read().split("\n")
    targets = set()
    for ln in lines:
        m = re.match(r"\s*s_cbranch_exec()z\s+(\.L\w+)", ln)   # (execnz: loop back edges, whose targets legitimately run lane-wise code first)
        if m:
            targets.add(m.group(2))
    hits = []
    func = "?"
    i = 0
    while i < len(lines):
        ln = lines[i]
        mf = re.match(r"^(_Z\w+|k_\w+):", ln)
        if mf:
            func = mf.group(1)
        ml = re.match(r"^(\.L\w+):", ln)
        if ml and ml.group(1) in targets:
            pending = []
            j = i + 1
            while j < len(lines):
                t = lines[j].strip()
                j += 1
                if not t or t.startswith(";") or t.startswith("."):
                    if re.match(r"^\.L\w+:", t):
                        break
                    continue
                ins = t.split(";")[0].strip()
                if re.match(r"^s_or_b64\s+exec,\s*exec,", ins):
                    for p, q in pending:
                        hits.append((func, ml.group(1), p, q))
                    break
                if re.match(r"^s_(cbranch|branch|setpc|swappc|endpgm)", ins) or "exec" in ins.split()[1:2] or EXEC_WRITE.match(ins):
                    break
                if LANEWISE.match(ins):
                    pending.append((j, ins))
        i += 1
    return hits


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--build" in sys.argv:
        out = "/tmp/topay_isa_lint.s"
        src = os.path.join(ROOT, "topay_amd", "csrc", "topay_hip.hip")
        extra = [a[len("--flag="):] for a in sys.argv if a.startswith("--flag=")]
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                               "--cuda-device-only", "-o", out, src] + extra, stderr=subprocess.DEVNULL)
        args.append(out)
    total = 0
    for p in args:
        hits = lint(p)
        total += len(hits)
        byblock = {}
        for f, lab, line, ins in hits:
            byblock.setdefault((f, lab), []).append((line, ins))
        for (f, lab), v in byblock.items():
            print(f"{p}: {f}: join block {lab}: {len(v)} lane-wise instruction(s) ahead of the EXEC restore, first: line {v[0][0]}: {v[0][1]}")
        print(f"{p}: {len(byblock)} join block(s) with lane-wise code ahead of the EXEC restore")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
