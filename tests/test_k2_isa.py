"""K2's hand-counted gather pipeline (k2_predict.hip: p3_issue / p3_issue_wait, `s_waitcnt lgkmcnt(6)`) is only correct if the compiler
never touches a register whose inline-asm LDS load is still in flight. tools/check_k2_isa.py scans the generated code for exactly that;
the library's Makefile refuses to link without a clean scan, and this test runs the same scan in the CPU suite (hipcc cross-compiles
gfx950 here), with the Makefile's own flags."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("check_k2_isa", os.path.join(ROOT, "tools", "check_k2_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_no_instruction_touches_a_gather_register_in_flight():
    tool = _tool()
    blocks, errors = tool.scan(tool.make_assembly())
    assert blocks >= 32, f"only {blocks} gather blocks found: the scan no longer matches the kernel"
    assert errors == [], "\n".join(errors[:10])


def test_the_scan_finds_a_planted_hazard(tmp_path):
    """The scan is worth something only if it fires: a copy of a register in flight, planted behind the first pipelined block."""
    tool = _tool()
    lines = open(tool.make_assembly()).read().splitlines()
    import re

    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and re.search(tool.KERNEL, l))
    for i in range(start, len(lines)):
        if "lgkmcnt(6)" in lines[i]:
            reg = next(l.split()[1].rstrip(",") for l in lines[i - 6:i] if l.strip().startswith("ds_read_u16_d16_hi"))
            end = next(j for j in range(i, len(lines)) if lines[j].strip().startswith(";;#ASMEND"))
            lines.insert(end + 1, f"\tv_mov_b32_e32 v0, {reg}")
            break
    bad = tmp_path / "planted.s"
    bad.write_text("\n".join(lines))
    _, errors = tool.scan(str(bad))
    assert errors and "in flight" in errors[0]


def test_the_scan_refuses_control_flow_under_a_gather_in_flight(tmp_path):
    """The scan is linear: a branch or a label between the block that issues a gather and the block that waits for it must be reported, not followed."""
    tool = _tool()
    lines = open(tool.make_assembly()).read().splitlines()
    import re

    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and re.search(tool.KERNEL, l))
    for i in range(start, len(lines)):
        if "lgkmcnt(6)" in lines[i]:
            end = next(j for j in range(i, len(lines)) if lines[j].strip().startswith(";;#ASMEND"))
            lines.insert(end + 1, "\ts_cbranch_scc0 .LBB0_planted")
            break
    bad = tmp_path / "planted_branch.s"
    bad.write_text("\n".join(lines))
    _, errors = tool.scan(str(bad))
    assert errors and "control flow" in errors[0]

