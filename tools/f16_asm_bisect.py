"""Where does a wrong-result build of one fused16 instantiation need more time?  Runs ON a GPU box.

The -O1 build of k_fused16<double, 32, 4, 3> computes a wrong log-likelihood for MLP(13-29-4) BCE, and passes when the
compiler pads every instruction with s_nop (-amdgpu-snop-padding) or with s_waitcnt 0 (-amdgpu-waitcnt-forcezero): some
instruction pair is closer than the hardware needs.  This tool finds the pair: it takes the kernel's assembly, pads chosen
instructions with `s_nop 7` (before and after), rebuilds the library from the edited assembly and runs tools/f16_check.py:
first by instruction class, then by bisection over the occurrences of the first class whose padding repairs the result.

  python tools/f16_asm_bisect.py [-O1] [extra compiler flags ...]       -> gpurun_out/f16_asm_bisect.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LL = "/opt/rocm/lib/llvm/bin"
SRC = os.path.join(ROOT, "eeyore_amd/csrc/ey_fused16.hip")
OBJ = os.path.join(ROOT, "eeyore_amd/lib/obj")
WORK = os.environ.get("WORK", "/tmp/f16_asm")
CASE = os.environ.get("F16_CASE", "13,29,4 1,1 0 f64 7").split()
INST = os.environ.get("F16_INST", "8 32 3").split()   # element size, H, V
KERNEL = os.environ.get("F16_KERNEL", "_Z9k_fused16IdLi32ELi4ELi3EEv7F16ArgsIT_E")
OUT = os.path.join(ROOT, "gpurun_out", os.environ.get("OUTNAME", "f16_asm_bisect.txt"))
PAD = os.environ.get("F16_PAD", "s_nop 7")

CLASSES = [
    ("mfma", r"v_mfma"),
    ("accvgpr", r"v_accvgpr"),
    ("lane", r"v_readlane|v_writelane|v_readfirstlane"),
    ("dpp", r"_dpp\b|row_|quad_perm"),
    ("ds", r"ds_"),
    ("vmem", r"global_|scratch_|buffer_|flat_"),
    ("smem", r"s_load|s_buffer_load"),
    ("trans", r"v_rcp|v_rsq|v_sqrt|v_exp|v_log|v_sin|v_cos"),
    ("cmp", r"v_cmp|v_cndmask|v_div_fmas|v_div_scale"),
    ("salu", r"s_(?!nop|waitcnt|cbranch|branch|endpgm|barrier)"),
    ("branch", r"s_cbranch|s_branch"),
    ("valu64", r"v_fma_f64|v_mul_f64|v_add_f64|v_ldexp_f64|v_max_f64|v_min_f64|v_rndne_f64|v_cvt"),
    ("valu_other", r"v_(?!mfma|accvgpr|readlane|writelane|readfirstlane|rcp|rsq|sqrt|exp|log|cmp|cndmask|div_|fma_f64|mul_f64|add_f64|ldexp_f64|max_f64|min_f64|rndne_f64|cvt)"),
]


def log(msg):
    print(msg, flush=True)
    with open(OUT, "a") as f:
        f.write(msg + "\n")


def sh(cmd, **kw):
    return subprocess.run(cmd, shell=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, **kw)


def base_flags(extra):
    return (f"-std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -w -DF16_ONLY_SIZE={INST[0]} -DF16_ONLY_H={INST[1]} "
            f"-DF16_ONLY_V={INST[2]} " + " ".join(extra))


def build_from_asm(asm_path, tag, flags):
    d = os.path.join(WORK, tag)
    os.makedirs(d, exist_ok=True)
    steps = [
        f"{LL}/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c {asm_path} -o {d}/dev.o",
        f"{LL}/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o {d}/dev.out {d}/dev.o",
        f"{LL}/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,"
        f"hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input={d}/dev.out -output={d}/dev.hipfb",
        f"/opt/rocm/bin/hipcc {flags} --offload-host-only -Xclang -fcuda-include-gpubinary -Xclang {d}/dev.hipfb -c {SRC} -o {d}/f16.o",
        f"/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o {d}/lib.so {d}/f16.o {OBJ}/ey_api.o {OBJ}/ey_generic.o "
        f"{OBJ}/ey_mfma32.o {OBJ}/ey_large.o {OBJ}/ey_stats.o",
    ]
    for c in steps:
        r = sh(c)
        if r.returncode:
            return None, r.stdout[-400:]
    return f"{d}/lib.so", ""


def check(lib):
    if os.environ.get("F16_DRY"):
        return False, "dry run"
    env = dict(os.environ, EEYORE_AMD_LIB=lib)
    if not os.environ.get("F16_FULL"):  # F16_FULL=1: every mode (a wrong MALA log-rate needs the draws), else value + gradient
        env["F16_CHECK_QUICK"] = "1"
    r = sh(f"timeout -k 10 120 python {ROOT}/tools/f16_check.py " + " ".join(CASE), env=env)
    last = r.stdout.strip().split("\n")[-1] if r.stdout.strip() else ""
    if r.returncode >= 124:
        log("timeout in the check: stopping")
        sys.exit(1)
    return last.startswith("PASS"), last


def main():
    extra = sys.argv[1:] or ["-O1"]
    os.makedirs(WORK, exist_ok=True)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    flags = base_flags(extra)
    log(f"# flags: {flags}\n# case: {' '.join(CASE)}  pad: {PAD}")
    r = sh(f"/opt/rocm/bin/hipcc {flags} --offload-device-only -S {SRC} -o {WORK}/dev.s")
    if r.returncode:
        log("device compile failed: " + r.stdout[-400:])
        return 1
    lines = open(f"{WORK}/dev.s").read().split("\n")
    # instruction lines of the kernel
    start = next(i for i, ln in enumerate(lines) if ln.startswith(KERNEL + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = [i for i in range(start + 1, end) if re.match(r"^\t[a-z]", lines[i])]
    log(f"kernel lines {start + 1}..{end + 1}, {len(body)} instructions")

    def run(pad_idx, tag):
        pad = set(pad_idx)
        out = []
        for i, ln in enumerate(lines):
            if i in pad:
                out.append("\t" + PAD)
            out.append(ln)
            if i in pad and not re.match(r"^\ts_(cbranch|branch|setpc|endpgm)", ln):
                out.append("\t" + PAD)
        p = f"{WORK}/{tag}.s"
        open(p, "w").write("\n".join(out))
        lib, err = build_from_asm(p, tag, flags)
        if lib is None:
            return None, "build failed: " + err.replace("\n", " ")[-300:]
        return check(lib)

    ok, msg = run([], "plain")
    log(f"unpadded: {'PASS' if ok else 'FAIL'}  {msg[:120]}")
    if ok:
        log("the unpadded build passes: nothing to find")
        return 0
    ok, msg = run(body, "all")
    log(f"all padded: {ok}  {msg[:200]}")
    found = None
    for name, rx in CLASSES:
        idx = [i for i in body if re.match(r"^\t(" + rx + ")", lines[i]) or (name == "dpp" and re.search(rx, lines[i]))]
        if not idx:
            continue
        ok, msg = run(idx, "cls_" + name)
        log(f"class {name:11s} ({len(idx):5d} instructions): {'PASS' if ok else ('FAIL' if ok is False else 'n/a')}  {msg[:100] if not ok else ''}")
        if ok and found is None:
            found = (name, idx)
    if found is None:
        log("no single class repairs it")
        return 0
    name, idx = found
    log(f"bisecting class {name}")
    lo, hi = 0, len(idx)   # invariant: padding idx[lo:hi] repairs
    step = 0
    while hi - lo > 1:
        mid = (lo + hi) // 2
        step += 1
        ok1, _ = run(idx[lo:mid], f"b{step}a")
        if ok1:
            hi = mid
            log(f"  step {step}: [{lo}, {mid}) repairs")
            continue
        ok2, _ = run(idx[mid:hi], f"b{step}b")
        if ok2:
            lo = mid
            log(f"  step {step}: [{mid}, {hi}) repairs")
            continue
        log(f"  step {step}: neither half of [{lo}, {hi}) repairs alone (both needed): stopping with {hi - lo} instructions")
        break
    log(f"smallest repairing set: occurrences [{lo}, {hi}) of class {name}")
    for k in range(lo, min(hi, lo + 6)):
        i = idx[k]
        log(f"--- dev.s line {i + 1}:")
        for j in range(max(start, i - 14), min(end, i + 15)):
            log(f"{'>>' if j == i else '  '} {j + 1}: {lines[j]}")
    sh(f"cp {WORK}/dev.s {ROOT}/gpurun_out/f16_asm_bisect_dev.s")
    return 0


if __name__ == "__main__":
    sys.exit(main())
