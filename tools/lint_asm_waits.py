#!/usr/bin/env python3
"""Static check of the "inline-asm LDS read + counted s_waitcnt" idiom in the compiled kernels.

The attention kernels issue ds_read_b128 / ds_read_b64_tr_b16 through inline asm and wait for them with counted
`s_waitcnt lgkmcnt(N)` asm statements whose "+v" operands tie the fragments to the wait.  The compiler does not know that the
asm outputs are not there yet: when the register allocator splits such a live range it places the copy BEFORE the wait and the
copy reads a register whose LDS data has not arrived (seen in attn_bwd3_dq_kernel: a v_mov_b64 of half a V fragment ahead of the
wait in the ragged-tile path -> rare huge / NaN dQ rows).  Nothing at the source level rules that out, so the build checks the
ISA instead.

For every inline-asm LDS read this walks all paths of the function's control-flow graph from the read until a
`s_waitcnt lgkmcnt(N)` retires it (LDS operations complete in order: the read is complete once N <= number of LGKM operations
issued after it) and reports any instruction on the way that mentions one of its destination registers.

The same walk covers the one asynchronous VMEM result the GEMM kernels keep in a register across their main loop: the tile-claim
`global_atomic_add ... sc0` issued through inline asm in front of the loop and consumed behind it (gemm.hip, claim_issue /
claim_resolve).  It is retired by an `s_waitcnt vmcnt(0)`; a counted vmcnt(N > 0) on the way is ignored (stricter than needed).

usage: lint_asm_waits.py file.s [--kernel substr]      (file.s from `hipcc -S --cuda-device-only`)
exit status 1 when a violation is found.
"""
import re
import sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
LGKM_OP = re.compile(r"^\s*(ds_|s_load_|s_buffer_load_|s_sendmsg|s_memtime|s_memrealtime)")
WAIT = re.compile(r"s_waitcnt\b(.*)")
LGKM_N = re.compile(r"lgkmcnt\((\d+)\)")
VM_N = re.compile(r"vmcnt\((\d+)\)")
NEAR = 48
BRANCH = re.compile(r"^\s*(s_branch|s_cbranch_\w+)\s+(\S+)")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")


ALL_SOURCES = re.compile(r"^(ds_write|ds_store|global_store|buffer_store|flat_store|scratch_store|global_atomic|buffer_atomic|ds_add|ds_max|ds_min|"
                         r"v_cmp|v_cmpx|v_readfirstlane|v_readlane|v_permlane|s_|v_swap|v_nop|exp)")


def split_operands(text):
    """(destination registers, source registers) of one instruction: the first operand is the destination unless the mnemonic only reads."""
    parts = text.split(None, 1)
    if len(parts) < 2:
        return set(), set()
    ops = parts[1].split(",")
    if parts[0].startswith(("global_atomic", "buffer_atomic")) and "sc0" in text.split():
        return regs_of(ops[0]), regs_of(",".join(ops[1:]))        # returning form: the first operand receives the old value
    if ALL_SOURCES.match(parts[0]):
        return set(), regs_of(parts[1])
    return regs_of(ops[0]), regs_of(",".join(ops[1:]))


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def parse_functions(path):
    funcs, cur, name, in_asm = {}, None, None, False
    for raw in open(path):
        line = raw.rstrip("\n")
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, cur = m.group(1), []
            funcs[name] = cur
            continue
        if cur is None:
            continue
        if re.match(r"^\.Lfunc_end", line):
            cur, name = None, None
            continue
        s = line.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        lm = LABEL.match(line)
        if lm:
            cur.append(("label", lm.group(1), False))
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        code = s.split(";")[0].strip()
        if code:
            cur.append(("ins", code, in_asm))
    return funcs


def lint_function(name, items):
    labels = {it[1]: i for i, it in enumerate(items) if it[0] == "label"}
    bad = []

    def successors(i):
        kind, text, _ = items[i]
        if kind == "ins":
            if text.startswith("s_endpgm"):
                return []
            b = BRANCH.match(text)
            if b:
                tgt = labels.get(b.group(2))
                nxt = [tgt] if tgt is not None else []
                if b.group(1) != "s_branch" and i + 1 < len(items):
                    nxt.append(i + 1)
                return nxt
        return [i + 1] if i + 1 < len(items) else []

    for i, (kind, text, in_asm) in enumerate(items):
        if kind != "ins" or not in_asm or not (text.startswith("ds_read") or text.startswith("global_atomic")):
            continue
        vmem = text.startswith("global_atomic")
        if vmem and "sc0" not in text.split():
            continue                                  # no returned value, nothing to wait for
        dest = regs_of(text.split(",")[0])
        seen, stack = set(), [(j, 0, 1) for j in successors(i)]
        while stack:
            j, k, steps = stack.pop()
            if vmem:
                k = 0                                 # retired by vmcnt(0) only: the path state is the position alone
            if (j, k) in seen or k > 40 or steps > (200000 if vmem else 2000):
                continue
            seen.add((j, k))
            kind2, t2, asm2 = items[j]
            if kind2 == "ins":
                w = WAIT.match(t2)
                if w:
                    n = (VM_N if vmem else LGKM_N).search(t2)
                    # a wait without the counter's field leaves it alone
                    if n is not None and int(n.group(1)) <= k:
                        continue                      # retired on this path
                else:
                    d2, s2 = split_operands(t2)
                    if s2 & dest:
                        bad.append((i, j, text, t2, "reads"))
                        continue
                    if d2 & dest:
                        # the register is reused while the read may still be in flight: only a problem if the data lands afterwards;
                        # reported when it happens within NEAR instructions of the read (an abandoned prefetch that is overwritten
                        # hundreds of instructions later has long landed)
                        if steps <= NEAR:
                            bad.append((i, j, text, t2, "overwrites"))
                        continue
                if LGKM_OP.match(t2):
                    k += 1
            for nx in successors(j):
                stack.append((nx, k, steps + (1 if kind2 == "ins" else 0)))
    return bad


def main():
    args = sys.argv[1:]
    if not args:
        print(__doc__)
        return 2
    flt = None
    if "--kernel" in args:
        flt = args[args.index("--kernel") + 1]
        args = [a for a in args if a not in ("--kernel", flt)]
    total = 0
    for path in args:
        for name, items in parse_functions(path).items():
            if flt and flt not in name:
                continue
            n_reads = sum(1 for k, t, a in items if k == "ins" and a and (t.startswith("ds_read") or t.startswith("global_atomic")))
            if not n_reads:
                continue
            bad = lint_function(name, items)
            uniq = sorted({(b[2], b[3], b[4]) for b in bad})
            print(f"{name}: {n_reads} asm LDS reads / returning atomics, {len(uniq)} premature uses")
            for rd, use, how in uniq[:20]:
                print(f"    {rd}\n        before its wait, {how} it: {use}")
            total += len(uniq)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
