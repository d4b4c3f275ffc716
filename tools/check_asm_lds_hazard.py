"""Static check of the kernels that issue LDS transpose reads from inline asm (binhip_wgrad.hip: tr_issue / the manual
s_waitcnt lgkmcnt(0)): between a `ds_read_b64_tr_b16 vX, ...` and the next `s_waitcnt lgkmcnt(0)` no instruction may touch
the destination registers — the compiler does not know the read is asynchronous, so a copy or spill scheduled in that window
would move stale data.  usage: check_asm_lds_hazard.py file.s [kernel-name-substring]; exit code 1 on a hazard."""
import re, sys


def regs_of(tok):
    """v12 -> {12}; v[4:7] -> {4,5,6,7}; a-registers are tracked in a separate number space"""
    out = set()
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b", tok):
        if m.group(1):
            base = 0 if m.group(1) == "v" else 1000
            out |= {base + i for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        else:
            base = 0 if m.group(4) == "v" else 1000
            out.add(base + int(m.group(5)))
    return out


def check(path, want=""):
    bad, kernels, name, pending, n_reads = [], 0, None, {}, 0
    for ln, line in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, pending = m.group(1), {}
            continue
        if name is None or want not in name:
            continue
        code = line.split(";")[0].strip()
        if not code or code.startswith(".") or code.endswith(":"):
            continue
        op, _, rest = code.partition(" ")
        if op == "s_endpgm":
            kernels += 1
        if op == "s_waitcnt" and "lgkmcnt(0)" in rest:
            pending = {}
            continue
        if op.startswith("s_cbranch") or op == "s_branch" or op == "s_barrier":
            if pending and op == "s_barrier":
                bad.append((name, ln, "barrier with reads in flight", code))
            continue
        touched = regs_of(rest)
        if op == "ds_read_b64_tr_b16":
            dst = regs_of(rest.split(",")[0])
            srcs = regs_of(",".join(rest.split(",")[1:]))
            hit = srcs & set(pending)
            if hit:
                bad.append((name, ln, f"address register {sorted(hit)} is an in-flight destination", code))
            for r in dst:
                pending[r] = ln
            n_reads += 1
            continue
        hit = touched & set(pending)
        if hit:
            bad.append((name, ln, f"touches in-flight destination {sorted(hit)} (read issued at line {min(pending[r] for r in hit)})", code))
    return bad, kernels, n_reads


if __name__ == "__main__":
    bad, kernels, n_reads = check(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
    for b in bad[:20]:
        print("HAZARD", *b)
    print(f"{kernels} kernels, {n_reads} transpose reads, {len(bad)} hazards")
    sys.exit(1 if bad else 0)
