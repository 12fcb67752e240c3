"""Audit of hand-issued loads in compiler output (test infrastructure).

ranked_stream.hip issues its range-table gathers from inline asm and waits for them by hand (`rs_gissue` / `rs_gwait`
markers), because hipcc would otherwise drain vmcnt at every control-flow join. For the compiler the destination VGPR
of such a load is "written" at the asm statement, so nothing stops it from copying, spilling or reusing that register
before the data has landed. This walks the control-flow graph of every kernel in a `hipcc -S` listing from each issue
point and reports any instruction that names the destination register before a wait that names it (or a full
`s_waitcnt vmcnt(0)` inside an asm statement) is reached.
"""
import re


def kernels(asm_text):
    out, cur, name = {}, None, None
    for line in asm_text.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m and any(k in m.group(1) for k in ("k_ranked_stream", "k_freq_stream", "k_and_stream", "k_union_stream")):
            name, cur = m.group(1), []
            out[name] = cur
            continue
        if cur is not None:
            cur.append(line)
            if line.strip().startswith("s_endpgm"):
                cur = None
    return out


def audit(lines):
    """returns a list of violations (strings)"""
    label_at = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            label_at[m.group(1)] = i
    in_asm = [False] * len(lines)
    flag = False
    for i, l in enumerate(lines):
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            flag = True
        in_asm[i] = flag
        if t.startswith(";;#ASMEND"):
            flag = False

    def succ(i):
        t = lines[i].strip().split(";")[0].strip()
        if t.startswith("s_endpgm"):
            return []
        m = re.match(r"s_branch\s+(\.LBB\w+)", t)
        if m:
            return [label_at[m.group(1)]]
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\w+)", t)
        if m:
            return [label_at[m.group(1)], i + 1]
        return [i + 1] if i + 1 < len(lines) else []

    bad = []
    issues = [i for i, l in enumerate(lines) if "rs_gissue" in l]
    for i0 in issues:
        m = re.search(r"global_load_ubyte\s+(v\d+),", lines[i0])
        reg = m.group(1)
        pat = re.compile(r"\b%s\b" % reg)
        seen, work = set(), list(succ(i0))
        while work:
            i = work.pop()
            if i in seen or i >= len(lines):
                continue
            seen.add(i)
            raw = lines[i]
            t = raw.strip()
            code = t.split(";")[0]
            if "rs_gwait" in raw and pat.search(raw):
                continue  # waited for: the register is the compiler's again
            if in_asm[i] and re.search(r"s_waitcnt vmcnt\(0\)", code):
                continue  # everything has landed
            if "rs_gissue" in raw:
                if pat.search(code.split(",")[0]):
                    continue  # re-issued into the same register
            if pat.search(code) and not t.startswith((";", ".")):
                bad.append("line %d (issue at %d, %s): %s" % (i, i0, reg, t))
                continue
            work.extend(succ(i))
    return bad
