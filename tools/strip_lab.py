"""One-way: derive the SHIPPED source of a kernel file from its lab copy (smalltts_amd/csrc/exp/*_lab.hip) by resolving the experiment
switches as "not defined" and dropping the timeline stamps.  The lab copy keeps every branch (tools build it: make LAB=1).
    python tools/strip_lab.py smalltts_amd/csrc/exp/codec_ffn_stream_lab.hip smalltts_amd/csrc/codec_ffn_stream.hip SYM [SYM ...]
"""
import re
import sys


def strip(lines, undef, drop_calls):
    out, stack = [], []   # stack entries: [kind, keep_this_branch, parent_keep]  kind: "x" = one of ours, "o" = other (passed through)
    keep = True
    for ln in lines:
        s = ln.strip()
        m = re.match(r"#\s*(ifdef|ifndef|if|else|elif|endif)\b\s*(\w+)?", s)
        if m:
            d, sym = m.group(1), m.group(2)
            if d in ("ifdef", "ifndef") and sym in undef:
                stack.append(["x", d == "ifndef", keep])
                keep = keep and (d == "ifndef")
                continue
            if d in ("ifdef", "ifndef", "if"):
                stack.append(["o", True, keep])
                if keep:
                    out.append(ln)
                continue
            if d in ("else", "elif"):
                top = stack[-1]
                if top[0] == "x":
                    assert d == "else"
                    top[1] = not top[1]
                    keep = top[2] and top[1]
                elif keep:
                    out.append(ln)
                continue
            if d == "endif":
                top = stack.pop()
                if top[0] == "x":
                    keep = top[2]
                elif keep:
                    out.append(ln)
                continue
        if not keep:
            continue
        if any(re.match(r"\s*%s\(.*\);\s*(//.*)?$" % c, ln) for c in drop_calls):
            continue
        out.append(ln)
    assert not stack
    return out


if __name__ == "__main__":
    src, dst, syms = sys.argv[1], sys.argv[2], set(sys.argv[3:])
    calls = [s[5:] for s in syms if s.startswith("CALL:")]
    syms = {s for s in syms if not s.startswith("CALL:")}
    with open(src) as f:
        lines = f.readlines()
    with open(dst, "w") as f:
        f.writelines(strip(lines, syms, calls))
