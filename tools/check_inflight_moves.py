#!/usr/bin/env python3
"""Static check of the generated gfx950 assembly: a register that an inline-asm load has in flight must not be read by a
compiler-generated move before the next wait of its counter.

Inline-asm loads (global_load_* / ds_read_* written in asm volatile statements) are invisible to the compiler's wait-count
pass: it treats their results as available at once and may copy them (live-range splits, phi copies at loop back-edges).  A
copy issued before the hand-written s_waitcnt reads stale data whenever the latency is not covered by chance - the failure
mode of kmeans_screen_kernel in round 3.  The scan is linear in text order (it does not follow branches) and therefore a
heuristic: it reports v_mov / v_accvgpr moves whose source overlaps a register loaded inside an ASMSTART/ASMEND block, with no
s_waitcnt of the matching counter (vmcnt for global / buffer loads, lgkmcnt for ds reads) in between.

usage: tools/check_inflight_moves.py file.s [...]   (hipcc -S --cuda-device-only output); exit status 1 if anything is reported
"""
import re, sys

REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def scan(path):
    bad = 0
    func = "?"
    in_asm = False
    pending = {"vm": {}, "lgkm": {}}  # register -> line number of the load
    for ln, line in enumerate(open(path), 1):
        t = line.strip()
        if t.endswith(":") and not t.startswith(".") and not t.startswith(";"):
            func = t[:-1]
            pending = {"vm": {}, "lgkm": {}}
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        op = t.split()[0]
        if op == "s_waitcnt":
            if "vmcnt" in t:
                pending["vm"].clear()
            if "lgkmcnt" in t:
                pending["lgkm"].clear()
            continue
        if op in ("s_endpgm",):
            pending = {"vm": {}, "lgkm": {}}
            continue
        if in_asm:
            if (op.startswith("global_load") or op.startswith("buffer_load")) and "lds" not in op:
                for r in regs(t.split(",")[0]):
                    pending["vm"][r] = ln
            elif op.startswith("ds_read"):
                for r in regs(t.split(",")[0]):
                    pending["lgkm"][r] = ln
            continue
        if op.startswith("v_mov") or op.startswith("v_accvgpr_write"):
            parts = t.split(None, 1)[1].split(",")
            src = set()
            for p_ in parts[1:]:
                src |= regs(p_)
            for kind in ("vm", "lgkm"):
                hit = src & set(pending[kind])
                if hit:
                    bad += 1
                    print("%s:%d: %s: `%s` reads v%s loaded by asm at line %d with no %s wait in between"
                          % (path, ln, func[:60], t, sorted(hit)[0], pending[kind][sorted(hit)[0]], "vmcnt" if kind == "vm" else "lgkmcnt"))
        else:
            # any other instruction that overwrites a pending register ends its in-flight window (the register was reused)
            pass
    return bad


if __name__ == "__main__":
    total = sum(scan(p) for p in sys.argv[1:])
    print("%d suspicious move(s)" % total)
    sys.exit(1 if total else 0)
