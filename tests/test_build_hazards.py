"""The build's hazard pass (eeyore_amd/csrc/mfma_load_hazard.py): the gfx950 compiler lets an LDS load
into the SrcC registers of a running f64 MFMA follow it directly, which corrupts the product (measured:
tools/mfma_war_probe.hip; it was the cause of the wrong -O1 builds of ey_fused16.hip).  The pass pads such loads in the
device assembly of every translation unit.  CPU tests: the pass on the failing sequence and on control flow; the
assembly the shipped library was built from has no such load left; the -O1 reproducer still shows the pattern before the
pass and none after it."""
import importlib.util
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "eeyore_amd", "csrc")
spec = importlib.util.spec_from_file_location("mfma_load_hazard", os.path.join(CSRC, "mfma_load_hazard.py"))
hz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(hz)

FAILING = """k:
\tv_accvgpr_mov_b32 a24, a16
\ts_waitcnt lgkmcnt(0)
\ts_nop 0
\tv_mfma_f64_16x16x4_f64 a[16:23], v[26:27], v[34:35], a[24:31]
.LBB2_146:
\tds_read_b64 v[16:17], v160 offset:20480
\tds_read_b128 a[24:27], v183 offset:128
\tds_read_b128 a[28:31], v183 offset:144
\ts_and_b64 vcc, exec, s[38:39]
\ts_waitcnt lgkmcnt(0)
\tv_mfma_f64_16x16x4_f64 a[24:31], v[16:17], v[24:25], a[24:31]
\ts_endpgm
"""


def _ws_before(lines, needle):
    """wait states of s_nop directly in front of the first line containing `needle`"""
    i = next(k for k, ln in enumerate(lines) if needle in ln)
    ws = 0
    while i > 0 and lines[i - 1].strip().startswith("s_nop"):
        ws += int(lines[i - 1].split()[1]) + 1
        i -= 1
    return ws


def test_the_failing_sequence_is_padded_for_the_run_time_of_the_mfma():
    lines = FAILING.split("\n")
    _, need, found = hz.find(lines)
    assert len(found) == 1 and found[0][4] == 0 and found[0][5] == 18   # distance 0: the LDS read between does not count
    out, _ = hz.fix(lines)
    assert _ws_before(out, "ds_read_b128 a[24:27]") == 18
    assert _ws_before(out, "ds_read_b128 a[28:31]") == 0                # behind the first pad it is outside the window
    assert hz.find(out)[2] == []                                         # idempotent: nothing left to find
    assert [ln for ln in out if "mfma_load_hazard.py" not in ln] == lines  # nothing else moved


def test_distance_counts_valu_and_nops_but_not_memory_instructions_and_follows_branches():
    src = """k:
\tv_mfma_f64_16x16x4_f64 a[0:7], v[0:1], v[2:3], a[8:15]
\tv_mov_b32 v9, v8
\ts_nop 3
\tglobal_load_dwordx2 v[20:21], v[30:31], off
\ts_cbranch_vccnz .LBB0_2
\tv_add_f64 v[4:5], v[4:5], v[6:7]
\ts_branch .LBB0_3
.LBB0_2:
\tds_read_b64 a[10:11], v40
.LBB0_3:
\tscratch_load_dwordx2 a[0:1], off, off offset:16
\ts_endpgm
"""
    lines = src.split("\n")
    _, need, found = hz.find(lines)
    got = {f[3].split()[0]: (f[4], f[5]) for f in found}
    # taken branch: v_mov 1 + s_nop 4 + cbranch 1 = 6 wait states to the LDS read of SrcC
    assert got["ds_read_b64"] == (6, 18)
    # the load into vDst: over the taken branch it lies behind the LDS read's pad (outside the window by then), over the
    # fall-through path v_mov 1 + s_nop 4 + cbranch 1 + v_add 1 + s_branch 1 = 8
    assert got["scratch_load_dwordx2"] == (8, 18)
    out, _ = hz.fix(lines)
    assert hz.find(out)[2] == []


def test_f32_and_bf16_mfmas_are_left_alone_and_f64_4x4x4_gets_its_own_window():
    src = """k:
\tv_mfma_f32_32x32x2_f32 a[0:15], v0, v1, a[16:31]
\tds_read_b128 a[16:19], v9
\tv_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[16:31]
\tds_read_b128 a[16:19], v9
\tv_mfma_f64_4x4x4_4b_f64 v[106:107], v[24:25], v[118:119], v[18:19]
\tds_read_b128 v[18:21], v213 offset:2064
\ts_endpgm
"""
    _, need, found = hz.find(src.split("\n"))
    assert [(f[1].split()[0], f[4], f[5]) for f in found] == [("v_mfma_f64_4x4x4_4b_f64", 0, 6)]


def test_the_assembly_the_library_was_built_from_has_no_such_load_left():
    obj = os.path.join(ROOT, "eeyore_amd", "lib", "obj")
    files = sorted(f for f in (os.listdir(obj) if os.path.isdir(obj) else []) if f.endswith(".fixed.s"))
    if not files:
        pytest.skip("library not built in this tree yet (python -c 'import __graft_entry__ as g; g.build()')")
    assert {"ey_fused16.fixed.s", "ey_fused16_d32.fixed.s", "ey_large.fixed.s", "ey_mfma32.fixed.s"} <= set(files)
    for f in files:
        _, _, found = hz.find(open(os.path.join(obj, f)).read().split("\n"))
        assert found == [], (f, found[:3])
    # the pass had work to do on the f64 kernels of the fused16 family (if a later compiler stops producing the pattern
    # this number goes to zero and the pass is idle, which is fine -- reported, not asserted)
    for unit in ("ey_fused16", "ey_fused16_d32"):
        _, _, before = hz.find(open(os.path.join(obj, unit + ".dev.s")).read().split("\n"))
        print(f"{unit}: {len(before)} load(s) padded by the build")


def test_the_o1_reproducer_shows_the_pattern_before_the_pass_and_none_after(tmp_path):
    """k_fused16<double, 32, 4, one hidden layer | padded> at -O1: the build whose MLP(13-29-4) BCE results were wrong
    (tools/f16_asm_bisect.py found this very load).  Seconds to compile: one instantiation."""
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    dev = tmp_path / "dev.s"
    cmd = ["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-fast-math", "-w",
           "-DF16_ONLY_SIZE=8", "-DF16_ONLY_H=32", "-DF16_ONLY_V=3", "-DF16_ONLY_MODE=1", "-DEY_F16_PART=1", "--offload-device-only", "-S",
           os.path.join(CSRC, "ey_fused16.hip"), "-o", str(dev)]
    subprocess.check_call(cmd)
    lines = dev.read_text().split("\n")
    _, _, found = hz.find(lines)
    adjacent = [f for f in found if f[1].startswith("v_mfma_f64_16x16x4") and f[4] == 0]
    if not adjacent:
        pytest.skip("this compiler no longer schedules the load directly behind the MFMA")
    out, _ = hz.fix(lines)
    assert hz.find(out)[2] == []
