#!/usr/bin/env python3
"""Checks the generated code of K2's gather pipeline (frave_amd/csrc/k2_predict.hip, p3_issue / p3_issue_wait / p3_wait).

The gathers are inline-asm ds_read_u16_d16_hi whose completion the compiler does not track: a block issues the NEXT node's six
gathers and waits (lgkmcnt(6)) for the current node's. Between the block that issues a register's load and the block that waits for
it, no instruction may read or write that register - a register copy there would carry stale data on. The bucket-table read of the previous
node (ds_read_u16) rides at the FRONT of a block, where lgkmcnt(6) covers it. This script scans the
product kernel's assembly linearly: registers in flight, any mention of them outside an asm block is an error (a compiler-inserted
`s_waitcnt lgkmcnt(0)` lands everything and is fine).

The assembly is produced by frave_amd/csrc/Makefile with the very flags the object file is built with (FLAGS + EXTRA_k2_predict);
the Makefile runs this scan before it links libfri_hip.so, and tests/test_k2_isa.py runs it in the CPU suite.

The scan is linear, so it must not meet control flow while a gather is in flight: a label (a merge point: the state of the other predecessor is unknown) or a
branch / end of program with a non-empty in-flight set is reported as an error rather than followed - the kernel keeps every gather block, the
node arithmetic between two blocks included, in straight-line code.

    python tools/check_k2_isa.py                 # make the assembly (csrc/build/k2_predict.s), then scan it; exit code 0 = clean
    python tools/check_k2_isa.py <file.s>        # scan an existing assembly file
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "frave_amd", "csrc")
KERNEL = "(predict_histogram_kernel3)"  # every kernel of the file with hand-pipelined gather blocks
MIN_BLOCKS = 32  # per instantiation: two roles x two cells x (1 + 4 + 1 + ...) blocks; fewer means the scan no longer matches the kernel


def make_assembly():
    """csrc/build/k2_predict.s through the Makefile (same compiler, same flags as the object that is linked)."""
    asm = os.path.join(CSRC, "build", "k2_predict.s")
    subprocess.run(["make", "-s", "-C", CSRC, asm], check=True)
    return asm


def regs_of(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def scan(asm):
    """(number of gather blocks seen, list of problems) for the product kernel in assembly file `asm`."""
    lines = open(asm).read().splitlines()
    blocks, errors = 0, []
    for start in [i for i, l in enumerate(lines) if re.match(r"^_ZN.*" + KERNEL + r".*:", l)]:  # every instantiation of the kernel templates
        end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
        b, e = scan_function(lines, start, end)
        if b < MIN_BLOCKS:
            e.append(f"{lines[start].split(':')[0][:90]}: only {b} gather blocks (expected >= {MIN_BLOCKS}): the scan no longer matches this instantiation")
        blocks += b
        errors += e
    return blocks, errors


def scan_function(lines, start, end):
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
            table = [k for k, b in enumerate(body) if re.match(r"ds_read_u16 v\d+", b)]  # the previous node's bucket-table read rides in front of the gathers
            if table and loads and table[-1] > body.index(loads[0]):
                errors.append(f"line {j}: a table read behind the block's gathers is not covered by lgkmcnt(6)")
            if loads or table or any("lgkmcnt" in b for b in body):
                blocks += 1
                issued = {int(re.match(r"ds_read_u16_d16_hi v(\d+)", b).group(1)) for b in loads}
                if table and not any(b.startswith("s_waitcnt") for b in body):  # issued alone (the last node's): in flight until the next lgkmcnt(0)
                    issued |= {int(re.match(r"ds_read_u16 v(\d+)", body[k]).group(1)) for k in table}
                waits = [b for b in body if b.startswith("s_waitcnt")]
                if waits and "lgkmcnt(0)" in waits[-1]:
                    in_flight = set()
                elif waits and re.search(r"lgkmcnt\((\d+)\)", waits[-1]):
                    # lgkmcnt(N) behind N gathers: exactly the block's own loads stay in flight (K2: 6 per node, the fit's value pass: 7)
                    n = int(re.search(r"lgkmcnt\((\d+)\)", waits[-1]).group(1))
                    if len(loads) != n:
                        errors.append(f"line {j}: lgkmcnt({n}) behind {len(loads)} loads")
                    in_flight = set(issued)
                else:
                    in_flight |= issued
            i = j + 1
            continue
        if in_flight and (re.match(r"^\.LBB\S*:", lines[i].strip()) or re.match(r"^(s_cbranch|s_branch|s_setpc|s_endpgm|s_call|s_swappc)", line)):
            errors.append(f"line {i + 1}: `{lines[i].strip()[:60]}` - control flow while v{sorted(in_flight)} are in flight (the linear scan cannot follow it)")
            in_flight = set()  # (reported once)
        if line and not line.startswith(".") and not line.endswith(":"):
            if line.startswith("s_waitcnt") and "lgkmcnt(0)" in line:
                in_flight = set()
            elif in_flight:
                hit = regs_of(line) & in_flight
                if hit:
                    errors.append(f"line {i + 1}: `{line}` touches v{sorted(hit)} while its gather is in flight")
        i += 1
    return blocks, errors


def main():
    asm = sys.argv[1] if len(sys.argv) > 1 else make_assembly()
    blocks, errors = scan(asm)
    print(f"{blocks} gather blocks checked, {len(errors)} problem(s)")
    for e in errors[:20]:
        print("  " + e)
    if blocks == 0:
        print("  no gather block found: the scan no longer matches the kernel")
        return 1
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main())
