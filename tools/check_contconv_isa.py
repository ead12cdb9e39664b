"""Checks, in the gfx950 disassembly of csrc/contconv_fused.hip, the two properties of the fused ContinuousConv kernel
that hipcc cannot see because the instructions involved are issued from inline asm (ADVICE round 2, and the comments at
`load_b` / `CC_FRAGMENT_LANDED` in the source):

  1. consumer waves -- the filter fragment of the NEXT cell is requested a whole cell ahead by eight
     `global_load_dwordx4` and declared landed by a hand-written `s_waitcnt vmcnt(8)`: between a fragment load and the
     wait that covers it NO instruction may read or write the load's destination registers (no copy, no spill, no
     early use);
  2. the kernel spills at most the two registers the build is known to spill (a spill slot's `s_waitcnt vmcnt(0)` would
     drain the prefetch), and allocates <= 128 VGPRs (four waves per SIMD: one 16-wave workgroup per CU).

The scan is linear over the consumer's region of the function (first fragment load .. last MFMA): loads retire in
order, `s_waitcnt vmcnt(N)` retires all but the youngest N. Run by tests/test_cabi.py (no GPU needed: hipcc
cross-compiles).   python tools/check_contconv_isa.py [path/to/contconv_fused.s]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "nbody-deep-sim_amd", "csrc", "contconv_fused.hip")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-S"]


def disassemble(out_path):
    subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-o", out_path, SRC], check=True, capture_output=True)


def _refs(text):
    text = text.split(";")[0]
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def check_kernel(asm, mangled):
    i = asm.index(mangled + ":")
    body = asm[i:asm.index(".Lfunc_end", i)].split("\n")
    insts = [ln.strip() for ln in body if ln.strip() and not ln.strip().startswith((".", ";", "//")) and not ln.strip().endswith(":")]
    first = next(k for k, t in enumerate(insts) if t.startswith("global_load_dwordx4"))
    last = max(k for k, t in enumerate(insts) if t.startswith("v_mfma"))
    pending = []            # destination register sets of the fragment loads in flight, oldest first
    problems, waits = [], 0
    for t in insts[first:last + 1]:
        op = t.split()[0]
        if op == "global_load_dwordx4":
            m = re.match(r"global_load_dwordx4 v\[(\d+):(\d+)\]", t)
            pending.append(set(range(int(m.group(1)), int(m.group(2)) + 1)))
            continue
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m:
                keep = int(m.group(1))
                pending = pending[len(pending) - keep:] if keep else []
                waits += keep == 8
            continue
        if op.startswith(("global_load", "buffer_load", "scratch_load")):
            problems.append(f"another vector-memory load inside the consumer region breaks the hand count: {t}")
        hit = _refs(t) & set().union(*pending) if pending else set()
        if hit:
            problems.append(f"touches fragment registers {sorted(hit)} before their wait: {t}")
    return problems, waits, len(insts)


def metadata(asm, mangled):
    i = re.search(rf"\.name:\s+{re.escape(mangled)}\b", asm).start()
    # the fields of ONE kernel's metadata entry sit between two '- .agpr_count' markers
    a = asm.rfind("- .agpr_count", 0, i)
    b = asm.find("- .agpr_count", i)
    ent = asm[a:b if b > 0 else len(asm)]
    return {k: int(re.search(rf"\.{k}:\s+(\d+)", ent).group(1)) for k in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count")}


def main(path=None):
    tmp = None
    if path is None:
        tmp = tempfile.mkdtemp(prefix="nbd_isa_")
        path = os.path.join(tmp, "contconv_fused.s")
        disassemble(path)
    asm = open(path).read()
    report = {}
    ok = True
    for kg, mangled in ((8, "_ZN12_GLOBAL__N_122contconv_stream_kernelILi8EEEvNS_6CCArgsE"),
                        (2, "_ZN12_GLOBAL__N_122contconv_stream_kernelILi2EEEvNS_6CCArgsE")):
        problems, waits, n = check_kernel(asm, mangled)
        md = metadata(asm, mangled)
        if kg == 8 and waits < 2:
            problems.append(f"expected the two hand-written s_waitcnt vmcnt(8) (one per fragment set), found {waits}")
        if md["vgpr_count"] > 128:
            problems.append(f"{md['vgpr_count']} VGPRs: a 16-wave workgroup no longer fits a CU")
        if md["vgpr_spill_count"] > 2:
            problems.append(f"{md['vgpr_spill_count']} spilled VGPRs (2 known)")
        report[f"contconv_stream_kernel<{kg}>"] = {"instructions": n, "vmcnt8_waits": waits, **md, "problems": problems}
        ok = ok and not problems
    return ok, report


if __name__ == "__main__":
    import json
    good, rep = main(sys.argv[1] if len(sys.argv) > 1 else None)
    print(json.dumps(rep, indent=1))
    sys.exit(0 if good else 1)
