#!/usr/bin/env python3
"""Static check of the hand-scheduled loads of mlp_stream_wgrad_kernel in the compiled ISA (run on the CPU):
inside the kernel's main loop no instruction other than the inline-assembly loads themselves may WRITE a
register that one of those loads targets, and none may read one outside the hand-placed waits' shadow as a
plain copy (v_mov / v_accvgpr of a load destination = a copy of data that may not have landed).
usage: check_asm_loads.py file.s [kernel-substring]"""
import re
import sys

src = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else "mlp_stream_wgrad_kernel"


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


bad = 0
kern = None
body = []
for line in src:
    m = re.match(r"^(_Z\w+):", line)
    if m:
        kern, body = m.group(1), []
    if kern:
        body.append(line)
    if kern and "s_endpgm" in line:
        if want in kern:
            # loop = from the first '=>This Inner Loop' label to the last branch back to it
            marks = [i for i, l in enumerate(body) if "Loop: Header" in l or "This Inner Loop" in l]
            start = marks[0] if marks else None
            if start is not None:
                # the loop's blocks carry a "Loop" annotation; it ends at the next label without one
                end = next((i for i in range(marks[-1] + 1, len(body)) if re.match(r"^\.LBB", body[i])), len(body))
                loop = body[start:end]
                dests = set()
                in_asm = False
                for l in loop:
                    if "#ASMSTART" in l:
                        in_asm = True
                    elif "#ASMEND" in l:
                        in_asm = False
                    elif in_asm and "global_load" in l:
                        dests |= regs(l.split()[1].rstrip(","))
                n_bad = 0
                in_asm = False
                for l in loop:
                    if "#ASMSTART" in l:
                        in_asm = True
                        continue
                    if "#ASMEND" in l:
                        in_asm = False
                        continue
                    t = l.strip()
                    if in_asm or not t or t.startswith(";") or t.startswith("."):
                        continue
                    ops = [o.strip().rstrip(",") for o in t.split(None, 1)[1].split(",")] if " " in t else []
                    if not ops:
                        continue
                    op = t.split()[0]
                    wr = regs(ops[0]) if not op.startswith(("ds_write", "global_store", "s_", "v_cmp", "global_atomic")) else set()
                    if wr & dests:
                        print(f"{kern[:70]}: writes a load destination: {t}")
                        n_bad += 1
                    if op.startswith(("v_mov", "v_accvgpr")) and any(regs(o) & dests for o in ops[1:]):
                        print(f"{kern[:70]}: copies a load destination: {t}")
                        n_bad += 1
                print(f"{kern[:90]}: loop of {len(loop)} lines, {len(dests)} load-destination registers, {n_bad} findings")
                bad += n_bad
        kern = None
sys.exit(1 if bad else 0)
