#!/usr/bin/env python3
"""Static check of the inline-assembly loads of the streaming MLP kernels in the compiled ISA (runs on the CPU;
tests/test_cabi_and_host.py runs it on the PRODUCT build's source with the product flags).

hipcc does not model an `asm volatile("global_load ...")`: it emits no wait for it and treats the destination as
written at the end of the statement (cdna_hip_programming.md §5.7), so the kernel places its own
`s_waitcnt vmcnt(N)`.  What can go wrong after a compiler change: an instruction between the load and the wait
that covers it READS the destination (data that has not landed: a copy, a spill, a use hoisted above the wait) or
WRITES it (the landing load then overwrites a live value).  This script finds, for every asm load inside a kernel's
main loop, the first `s_waitcnt vmcnt(M)` behind it that guarantees its completion — vmcnt counts loads, stores and
atomics in issue order, so that is the first wait with M <= (vector-memory instructions issued after the load) — and
reports every instruction in between (in layout order, wrapping once over the loop's back-edge: the prefetch of tile
t is consumed in tile t+1) that touches a destination register.

usage: check_asm_loads.py file.s [kernel-substring]     exit status 1 on findings"""
import re
import sys

VMEM = re.compile(r"^(global_|buffer_|flat_|scratch_)(load|store|atomic)")


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def parse(line):
    """-> (opcode, [operand register sets]) of an instruction line, or None"""
    t = line.split(";")[0].strip()
    if not t or t.startswith(".") or t.endswith(":"):
        return None
    parts = t.split(None, 1)
    ops = [o for o in (parts[1].split(",") if len(parts) > 1 else [])]
    return parts[0], [regs(o.split()[0]) if o.split() else set() for o in ops]


def check_kernel(name, body):
    marks = [i for i, l in enumerate(body) if "Loop Header" in l or "Loop: Header" in l or "This Inner Loop" in l]
    if not marks:
        return None
    start = marks[0]
    end = len(body)
    for i in range(len(body) - 1, start, -1):      # the loop ends at its last backward branch
        if re.search(r"s_cbranch_\w+\s+\.LBB", body[i]) or re.search(r"s_branch\s+\.LBB", body[i]):
            end = i + 1
            break
    loop = body[start:end]
    # instruction stream with asm-region flags
    stream = []
    in_asm = False
    for l in loop:
        if "#ASMSTART" in l:
            in_asm = True
            continue
        if "#ASMEND" in l:
            in_asm = False
            continue
        p = parse(l)
        if p:
            stream.append((p[0], p[1], in_asm, l.strip()))
    n = len(stream)
    findings = []
    loads = [i for i, (op, _, a, _) in enumerate(stream) if a and op.startswith("global_load")]
    for i in loads:
        dest = stream[i][1][0]
        issued = 0
        covered = False
        for step in range(1, 2 * n):
            op, ops, a, text = stream[(i + step) % n]
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", text)
                if m and int(m.group(1)) <= issued:
                    covered = True
                    break
                continue
            if VMEM.match(op):
                if a and op.startswith("global_load") and (ops[0] & dest) and (i + step) % n != i:
                    findings.append(f"load into {text.split()[1]} re-issued before the wait that covers the previous one")
                    break
                issued += 1
            if a:
                continue
            touched = set().union(*ops) if ops else set()
            if touched & dest:
                findings.append(f"between an asm load of v{min(dest)}.. and its wait: {text}")
        if not covered and not findings:
            findings.append(f"no covering s_waitcnt found for the asm load into v{min(dest)}..")
    return len(loop), len(loads), findings


def main(path, want):
    src = open(path).read().split("\n")
    bad = 0
    seen = 0
    kern, body = None, []
    for line in src:
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kern, body = m.group(1), []
        if kern:
            body.append(line)
        if kern and "s_endpgm" in line:
            if want in kern:
                r = check_kernel(kern, body)
                if r is not None:
                    seen += 1
                    nl, nloads, findings = r
                    for f in findings[:12]:
                        print(f"{kern[:70]}: {f}")
                    print(f"{kern[:90]}: loop of {nl} lines, {nloads} asm loads, {len(findings)} findings")
                    bad += len(findings)
            kern = None
    if not seen:
        print(f"no kernel matching {want!r} with a loop found")
        return 2
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "mlp_stream_"))
