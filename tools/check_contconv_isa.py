"""Checks, in the gfx950 disassembly of csrc/contconv_fused.hip, the properties of the fused ContinuousConv kernel that
hipcc cannot see because the instructions involved are issued from inline asm (see `CC_LOAD_QUAD` / `CC_QUAD_LANDED` in the
source):

  1. consumer waves -- a filter-fragment quad is re-requested IN PLACE (`global_load_dwordx4` from inline asm into the
     registers the MFMAs just read) and declared landed by a hand-counted `s_waitcnt vmcnt(N)`: between a fragment load and
     the wait that covers it NO instruction may read or write the load's destination registers (no copy, no spill, no early
     use) on ANY path through the kernel's control-flow graph;
  2. every hand-written wait is reached with the number of outstanding fragment loads its count assumes: a step that
     reloads sees 3 NS of them before every quad's first use, a step that does not sees 3 NS - j or none;
  3. the kernel does not spill and allocates <= 128 VGPRs (four waves per SIMD: one 16-wave workgroup per CU).

The walk is over basic blocks with the list of fragment loads in flight (destination register sets, oldest first) as the
state: loads retire in order, `s_waitcnt vmcnt(N)` retires all but the youngest N (other vector-memory operations can only
make it retire more: the list over-approximates what is pending). Inline-asm instructions are recognised by the
`;;#ASMSTART` / `;;#ASMEND` brackets hipcc writes around them. Run by tests/test_cabi.py (no GPU needed: hipcc
cross-compiles).   python tools/check_contconv_isa.py [path/to/contconv_fused.s]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "nbody-deep-sim_amd", "csrc", "contconv_fused.hip")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-S"]
KERNEL = "_ZN12_GLOBAL__N_122contconv_stream_kernelILi{ns}EEEvNS_6CCArgsE"


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


def blocks_of(asm, mangled):
    """[(label, [(text, in_asm)], successors)] of one function, in layout order."""
    i = asm.index(mangled + ":")
    body = asm[i:asm.index(".Lfunc_end", i)].split("\n")[1:]
    blocks, cur, label, in_asm, anon = [], [], "entry", False, 0
    for ln in body:
        t = ln.strip()
        if not t:
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if t.startswith((";", "//", ".")) and not re.match(r"^\.LBB\d+_\d+:", t):
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            blocks.append([label, cur])
            label, cur = m.group(1), []
            continue
        cur.append((t, in_asm))
        if t.split()[0].startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):      # a branch ends its basic block
            blocks.append([label, cur])
            anon += 1
            label, cur = f"{label}+{anon}", []
    blocks.append([label, cur])
    index = {b[0]: k for k, b in enumerate(blocks)}
    out = []
    for k, (lab, insts) in enumerate(blocks):
        succ, fall = [], True
        if insts:
            t = insts[-1][0]
            op = t.split()[0]
            if op.startswith("s_cbranch"):
                succ.append(index[t.split()[1]])
            elif op == "s_branch":
                succ.append(index[t.split()[1]])
                fall = False
            elif op in ("s_endpgm", "s_setpc_b64"):
                fall = False
        if fall and k + 1 < len(blocks):
            succ.append(k + 1)
        out.append((lab, insts, succ))
    return out


def check_kernel(asm, mangled, nq):
    blocks = blocks_of(asm, mangled)
    problems, seen_waits = [], {}
    n_loads = sum(1 for _, insts, _ in blocks for t, a in insts if a and t.startswith("global_load_dwordx4"))
    # states per block: tuple of frozensets (pending fragment loads, oldest first)
    work = [(0, ())]
    visited = set()
    steps = 0
    while work:
        k, state = work.pop()
        if (k, state) in visited:
            continue
        visited.add((k, state))
        steps += 1
        if len(visited) > 200000:
            problems.append("state explosion in the control-flow walk")
            break
        pending = list(state)
        lab, insts, succ = blocks[k]
        for t, in_asm in insts:
            op = t.split()[0]
            if in_asm and op == "global_load_dwordx4":
                m = re.match(r"global_load_dwordx4 v\[(\d+):(\d+)\]", t)
                pending.append(frozenset(range(int(m.group(1)), int(m.group(2)) + 1)))
                if len(pending) > 2 * nq:
                    problems.append(f"{lab}: more than {2 * nq} fragment loads in flight")
                    pending = pending[-2 * nq:]
                continue
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", t)
                if m:
                    keep = int(m.group(1))
                    if in_asm:
                        seen_waits.setdefault(keep, set()).add(len(pending))
                    pending = pending[len(pending) - keep:] if keep < len(pending) else pending
                    if keep == 0:
                        pending = []
                continue
            if pending and op.startswith("scratch_") and any(x.startswith("v_mfma") for x, _ in insts):
                problems.append(f"{lab}: spill traffic inside a step (its vmcnt(0) drains the fragment loads): {t}")
            if pending:
                hit = _refs(t) & frozenset().union(*pending)
                if hit:
                    problems.append(f"{lab}: touches fragment registers {sorted(hit)} before their wait: {t}")
        for s in succ:
            work.append((s, tuple(pending)))
    # every hand-written wait vmcnt(N): reached with nothing in flight, with what a non-reloading step leaves (N + 1),
    # or in a reloading step (3 NS in flight, N = 3 NS - 1); a pass's first step may also see the previous pass's last
    # request still in flight on top (up to 2 x 3 NS): more in flight only makes the wait retire more
    for keep, counts in sorted(seen_waits.items()):
        for c in sorted(counts):
            if c > keep + 1 and keep not in (0, nq - 1):        # vmcnt(0): the drains (pass start / end, flush_acc)
                problems.append(f"s_waitcnt vmcnt({keep}) from inline asm reached with {c} fragment loads in flight (expects <= {keep + 1})")
    dedup = []
    for p in problems:
        if p not in dedup:
            dedup.append(p)
    return dedup, {k: sorted(v) for k, v in sorted(seen_waits.items())}, n_loads, len(visited)


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
    for ns in (4,):                 # the instantiated K depths (cc_slabs() in the source)
        mangled = KERNEL.format(ns=ns)
        nq = 3 * ns
        problems, waits, n_loads, states = check_kernel(asm, mangled, nq)
        md = metadata(asm, mangled)
        if n_loads < 2 * nq:
            problems.append(f"expected at least {2 * nq} inline-asm fragment loads (a pass's first request + one reloading step), found {n_loads}")
        if nq - 1 not in waits:
            problems.append(f"no hand-written s_waitcnt vmcnt({nq - 1}) found")
        if md["vgpr_count"] > 128:
            problems.append(f"{md['vgpr_count']} VGPRs: a 16-wave workgroup no longer fits a CU")
        if md["vgpr_spill_count"] > 8:          # a handful outside the step loop (setup, between passes) is tolerated; inside it a
            problems.append(f"{md['vgpr_spill_count']} spilled VGPRs")      # spill would show up as a scratch_* touching the walk
        report[f"contconv_stream_kernel<{ns}>"] = {"fragment_loads": n_loads, "asm_waits(vmcnt: in flight)": waits, "cfg_states": states,
                                                  **md, "problems": problems[:20]}
        ok = ok and not problems
    return ok, report


if __name__ == "__main__":
    import json
    good, rep = main(sys.argv[1] if len(sys.argv) > 1 else None)
    print(json.dumps(rep, indent=1))
    sys.exit(0 if good else 1)
