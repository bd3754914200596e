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


def _main():
    src, dst, syms = sys.argv[1], sys.argv[2], (sys.argv[3:] or STREAM_FFN)
    with open(src) as f:
        lines = f.readlines()
    with open(dst, "w") as f:
        f.writelines(shipped(lines, syms))


def shipped(lines, syms):
    """syms: switch names, "CALL:<macro>" (statement lines calling it are dropped), "DEF:<macro>" (its #define lines are dropped)."""
    syms = set(syms)
    calls = [s[5:] for s in syms if s.startswith("CALL:")]
    defs = [s[4:] for s in syms if s.startswith("DEF:")]
    plain = {s for s in syms if ":" not in s}
    while lines and lines[0].startswith("// LAB:"):   # the lab copy's own header
        lines = lines[1:]
    out = strip(lines, plain, calls)
    return [ln for ln in out if not any(re.match(r"\s*#\s*define\s+%s\b" % d, ln) for d in defs)]


STREAM_FFN = ["FS_TIMELINE", "FS_PHASE_TICKS", "FS_ELIM_XIN", "FS_NOBARRIER", "FS_ELIM_FRAG", "FS_ELIM_MFMA", "FS_ELIM_GELU", "FS_EPI_NOFENCE",
              "FS_ELIM_XOUT", "FS_LIN_STORE", "FS_XNEXT", "FS_XO_EARLY", "CALL:FS_STAMP", "CALL:FS_STAMPR", "DEF:FS_STAMP", "DEF:FS_STAMPR"]


if __name__ == "__main__":
    _main()
