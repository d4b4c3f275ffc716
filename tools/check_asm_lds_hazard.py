"""Static check of the kernels that issue LDS transpose reads from inline asm (binhip_wgrad.hip: tr_issue* / the manual
s_waitcnt lgkmcnt(0)): on every control-flow path between a `ds_read_b64_tr_b16 vX, ...` and the next
`s_waitcnt ... lgkmcnt(0)` no instruction may touch the destination registers, and no s_barrier may be crossed — the compiler
does not know the read is asynchronous, so a copy or spill scheduled in that window would move stale data.
Forward dataflow over the basic blocks of each kernel (in-flight set = union over predecessors).
usage: check_asm_lds_hazard.py file.s [kernel-name-substring]; exit code 1 on a hazard."""
import re, sys


def regs_of(tok):
    """v12 -> {12}; v[4:7] -> {4,5,6,7}; a-registers live in a separate number space"""
    out = set()
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b", tok):
        if m.group(1):
            base = 0 if m.group(1) == "v" else 1000
            out |= {base + i for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        else:
            base = 0 if m.group(4) == "v" else 1000
            out.add(base + int(m.group(5)))
    return out


def kernels_of(path, want):
    name, ins = None, []
    for ln, line in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, ins = m.group(1), []
            continue
        if name is None:
            continue
        if line.startswith(".Lfunc_end"):          # (a kernel may contain several s_endpgm)
            if want in name:
                yield name, ins
            name = None
            continue
        lab = re.match(r"^(\.LBB\w+):", line)
        if lab:
            ins.append((ln, "label", lab.group(1)))
            continue
        code = line.split(";")[0].strip()
        if not code or code.startswith(".") or code.endswith(":"):
            continue
        op, _, rest = code.partition(" ")
        ins.append((ln, op, rest.strip()))


def check_kernel(name, ins):
    # basic blocks
    starts = {0}
    for i, (ln, op, rest) in enumerate(ins):
        if op == "label":
            starts.add(i)
        if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm"):
            starts.add(i + 1)
    starts = sorted(x for x in starts if x < len(ins))
    label_at = {rest: i for i, (ln, op, rest) in enumerate(ins) if op == "label"}
    block_of = {}
    for bi, st in enumerate(starts):
        en = starts[bi + 1] if bi + 1 < len(starts) else len(ins)
        for i in range(st, en):
            block_of[i] = bi
    succ = []
    for bi, st in enumerate(starts):
        en = starts[bi + 1] if bi + 1 < len(starts) else len(ins)
        ln, op, rest = ins[en - 1]
        s = []
        if op == "s_branch":
            s = [block_of[label_at[rest]]]
        elif op.startswith("s_cbranch"):
            s = [block_of[label_at[rest]]] + ([bi + 1] if bi + 1 < len(starts) else [])
        elif op != "s_endpgm" and bi + 1 < len(starts):
            s = [bi + 1]
        succ.append(s)
    # in-flight state: list of (line of the read instruction, destination registers) in ISSUE order along the path; LDS reads
    # retire in issue order, so `s_waitcnt lgkmcnt(N)` keeps the last N entries
    IN = [None for _ in starts]
    IN[0] = []
    bad, seen_bad = [], set()
    work = [0]

    def in_flight(pending):
        return {r: l for l, regs in pending for r in regs}

    while work:
        bi = work.pop()
        pending = list(IN[bi])
        st = starts[bi]
        en = starts[bi + 1] if bi + 1 < len(starts) else len(ins)
        for i in range(st, en):
            ln, op, rest = ins[i]
            if op == "label":
                continue
            if op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", rest)
                if m:
                    keep = int(m.group(1))
                    pending = pending[len(pending) - keep:] if keep else []
                continue
            fl = in_flight(pending)
            if op == "s_barrier" and fl and (ln, "b") not in seen_bad:
                seen_bad.add((ln, "b"))
                bad.append((name, ln, "barrier with reads in flight", op))
            if op.startswith("s_cbranch") or op == "s_branch":
                continue
            touched = regs_of(rest)
            if op == "ds_read_b64_tr_b16":
                dst = regs_of(rest.split(",")[0])
                hit = (regs_of(",".join(rest.split(",")[1:])) | dst) & set(fl)
                if hit and (ln, "a") not in seen_bad:
                    seen_bad.add((ln, "a"))
                    bad.append((name, ln, f"register {sorted(hit)} is an in-flight destination", rest))
                pending.append((ln, dst))
                continue
            hit = touched & set(fl)
            if hit and (ln, "t") not in seen_bad:
                seen_bad.add((ln, "t"))
                bad.append((name, ln, f"touches in-flight destination {sorted(hit)} (read at line {min(fl[r] for r in hit)})",
                            op + " " + rest))
        for sb in succ[bi]:
            if IN[sb] is None:
                IN[sb] = list(pending)
                work.append(sb)
            else:
                have = {l for l, _ in IN[sb]}
                extra = [e for e in pending if e[0] not in have]
                if extra:                       # paths disagree: conservative union, the newcomers count as most recent
                    IN[sb] = IN[sb] + extra
                    work.append(sb)
    n_reads = sum(1 for (_, op, _) in ins if op == "ds_read_b64_tr_b16")
    return bad, n_reads


def check(path, want=""):
    bad, kernels, n_reads = [], 0, 0
    for name, ins in kernels_of(path, want):
        b, n = check_kernel(name, ins)
        bad += b
        n_reads += n
        kernels += 1
    return bad, kernels, n_reads


if __name__ == "__main__":
    bad, kernels, n_reads = check(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
    for b in bad[:20]:
        print("HAZARD", *b)
    print(f"{kernels} kernels, {n_reads} transpose reads, {len(bad)} hazards")
    sys.exit(1 if bad else 0)
