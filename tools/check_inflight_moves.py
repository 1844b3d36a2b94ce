#!/usr/bin/env python3
"""Static check of the generated gfx950 assembly: a vector register that an inline-asm load has in flight must not be touched
by a compiler-generated instruction before the next wait of its counter.

Inline-asm loads (global_load_* / buffer_load_* / ds_read_* written in asm volatile statements) are invisible to the compiler's
wait-count pass: it treats their results as available at once and may copy them (live-range splits, phi copies at loop
back-edges) or schedule their consumers early.  An instruction issued before the hand-written s_waitcnt reads stale data
whenever the latency is not covered by chance - the failure mode of kmeans_screen_kernel in round 3 (v_mov at the back-edge in
front of the wait at the top of the next iteration: wrong labels in 1 run of the GPU suite in 6).

The check is a forward data-flow over the control-flow graph of every function: the state is the set of registers with an asm
load in flight (per counter: vmcnt for global / buffer loads, lgkmcnt for ds reads); an asm load adds its destination, any
s_waitcnt of the counter clears the set (counted waits are taken as complete waits: the check can miss, it does not invent),
block entries take the union of their predecessors.  Every instruction outside ASMSTART/ASMEND that names a register of the set
(as a source or as the destination it would overwrite) is reported.

usage: tools/check_inflight_moves.py file.s [...]   (hipcc -S --cuda-device-only output); exit status 1 if anything is reported
"""
import re
import sys

REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")
FUNC = re.compile(r"^([A-Za-z_][\w$]*):")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def functions(path):
    """[(name, [(line number, text, in_asm)])] for every function body of the file."""
    out, cur, name, in_asm = [], None, None, False
    for ln, line in enumerate(open(path), 1):
        t = line.strip()
        if not t or t.startswith(";") and not t.startswith(";;#ASM"):
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = FUNC.match(t)
        if m:
            name, cur = m.group(1), []
            out.append((name, cur))
            continue
        if cur is None or (t.startswith(".") and not LABEL.match(t)):
            continue
        cur.append((ln, t.split(";")[0].strip() if not LABEL.match(t) else t, in_asm))
        if t.startswith("s_endpgm"):
            cur = None
    return out


def check_function(path, name, ins):
    # basic blocks
    blocks, labels = [[]], {}
    for item in ins:
        m = LABEL.match(item[1])
        if m:
            if blocks[-1]:
                blocks.append([])
            labels[m.group(1)] = len(blocks) - 1
            continue
        blocks[-1].append(item)
        op = item[1].split()[0]
        if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            blocks.append([])
    succ = []
    for i, b in enumerate(blocks):
        s = []
        last = b[-1][1] if b else ""
        op = last.split()[0] if last else ""
        if op == "s_branch":
            tgt = last.split()[1]
            if tgt in labels:
                s.append(labels[tgt])
        elif op.startswith("s_cbranch"):
            tgt = last.split()[1]
            if tgt in labels:
                s.append(labels[tgt])
            if i + 1 < len(blocks):
                s.append(i + 1)
        elif op in ("s_endpgm", "s_setpc_b64"):
            pass
        elif i + 1 < len(blocks):
            s.append(i + 1)
        succ.append(s)

    def transfer(state, block, report):
        vm, lg = dict(state[0]), dict(state[1])
        bad = 0
        for ln, t, in_asm in block:
            parts = t.split(None, 1)
            op, body = parts[0], (parts[1] if len(parts) > 1 else "")
            if op == "s_waitcnt":
                if "vmcnt" in t:
                    vm.clear()
                if "lgkmcnt" in t:
                    lg.clear()
                continue
            if in_asm:
                if (op.startswith("global_load") or op.startswith("buffer_load")) and "lds" not in op:
                    for r in regs(body.split(",")[0]):
                        vm[r] = ln
                elif op.startswith("ds_read"):
                    for r in regs(body.split(",")[0]):
                        lg[r] = ln
                continue
            named = regs(body)
            for kind, pend in (("vmcnt", vm), ("lgkmcnt", lg)):
                hit = named & set(pend)
                if hit:
                    if report:
                        r0 = sorted(hit)[0]
                        print("%s:%d: %s: `%s` touches v%d, loaded by asm at line %d, with no %s wait in between"
                              % (path, ln, name[:60], t, r0, pend[r0], kind))
                        bad += 1
                    for r in hit:
                        del pend[r]
        return (vm, lg), bad

    state_in = [({}, {}) for _ in blocks]
    work = list(range(len(blocks)))
    rounds = 0
    while work and rounds < 100000:
        rounds += 1
        i = work.pop(0)
        out, _ = transfer(state_in[i], blocks[i], False)
        for j in succ[i]:
            changed = False
            for k in (0, 1):
                for r, ln in out[k].items():
                    if r not in state_in[j][k]:
                        state_in[j][k][r] = ln
                        changed = True
            if changed and j not in work:
                work.append(j)
    return sum(transfer(state_in[i], blocks[i], True)[1] for i in range(len(blocks)))


def scan(path):
    return sum(check_function(path, name, ins) for name, ins in functions(path))


if __name__ == "__main__":
    total = sum(scan(p) for p in sys.argv[1:])
    print("%d suspicious instruction(s)" % total)
    sys.exit(1 if total else 0)
