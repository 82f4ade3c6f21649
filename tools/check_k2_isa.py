#!/usr/bin/env python3
"""Checks the generated code of K2's gather pipeline (frave_amd/csrc/k2_predict.hip, p3_issue / p3_issue_wait / p3_wait).

The gathers are inline-asm ds_read_u16_d16_hi whose completion the compiler does not track: a block issues the NEXT node's six
gathers and waits (lgkmcnt(6)) for the current node's. Between the block that issues a register's load and the block that waits for
it, no instruction may read or write that register - a register copy there would carry stale data on. This script compiles the
kernel file to assembly and scans the product kernel linearly: registers in flight, any mention of them outside an asm block is an
error (a compiler-inserted `s_waitcnt lgkmcnt(0)` lands everything and is fine).

    python tools/check_k2_isa.py          # exit code 0 = clean
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "frave_amd", "csrc", "k2_predict.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"),
         "-I" + os.path.join(ROOT, "frave_amd", "csrc"), "-S", "--cuda-device-only", "-x", "hip"]


def regs_of(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def main():
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k2.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [SRC, "-o", asm], check=True, stderr=subprocess.DEVNULL)
        lines = open(asm).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN.*predict_histogram_kernel3", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    in_flight, errors, blocks, i = set(), [], 0, start
    while i < end:
        line = lines[i].split(";")[0].strip()
        if lines[i].strip().startswith(";;#ASMSTART"):
            j = i + 1
            body = []
            while not lines[j].strip().startswith(";;#ASMEND"):
                body.append(lines[j].strip())
                j += 1
            loads = [b for b in body if b.startswith("ds_read_u16_d16_hi")]
            if loads or any("lgkmcnt" in b for b in body):
                blocks += 1
                issued = {int(re.match(r"ds_read_u16_d16_hi v(\d+)", b).group(1)) for b in loads}
                waits = [b for b in body if b.startswith("s_waitcnt")]
                if waits and "lgkmcnt(0)" in waits[-1]:
                    in_flight = set()
                elif waits and "lgkmcnt(6)" in waits[-1]:
                    if len(loads) != 6:
                        errors.append(f"line {j}: lgkmcnt(6) behind {len(loads)} loads")
                    in_flight = set(issued)
                else:
                    in_flight |= issued
            i = j + 1
            continue
        if line and not line.startswith(".") and not line.endswith(":"):
            if line.startswith("s_waitcnt") and "lgkmcnt(0)" in line:
                in_flight = set()
            elif in_flight:
                hit = regs_of(line) & in_flight
                if hit:
                    errors.append(f"line {i + 1}: `{line}` touches v{sorted(hit)} while its gather is in flight")
        i += 1
    print(f"{blocks} gather blocks checked, {len(errors)} problem(s)")
    for e in errors[:20]:
        print("  " + e)
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main())
