#!/usr/bin/env python3
"""Resolve developer-only preprocessor switches of a source file to fixed values and delete their branches (round 4
clean-up: ablation / stamp / probe builds leave the product's translation units).

    python tools/strip_dev_macros.py file.hip MACRO=VALUE [MACRO=VALUE ...]

For every listed macro: its `#ifndef M / #define M v / #endif` default block goes, and every `#if <expr>` whose
expression names only listed macros is evaluated and replaced by the branch taken (`#else` honoured; nested
conditionals on other macros are kept verbatim).  Other uses of the macro name in code are replaced by the value."""
import re
import sys


def main():
    path, defs = sys.argv[1], dict(a.split("=", 1) for a in sys.argv[2:])
    lines = open(path).read().split("\n")
    out, i = [], 0
    names = sorted(defs, key=len, reverse=True)

    def known(expr):
        ids = set(re.findall(r"[A-Za-z_]\w*", expr)) - {"defined"}
        return ids and ids <= set(defs)

    def ev(expr):
        e = re.sub(r"//.*", "", expr)
        for n in names:
            e = re.sub(r"\b%s\b" % n, defs[n], e)
        e = e.replace("&&", " and ").replace("||", " or ")
        e = re.sub(r"!(?!=)", " not ", e)
        return bool(eval(e))

    def skip_block(j):
        """index after the #endif matching the conditional that starts at j; also the top-level #else position"""
        depth, k, els = 0, j, None
        while True:
            s = lines[k].strip()
            if re.match(r"#\s*if", s):
                depth += 1
            elif re.match(r"#\s*else", s) and depth == 1:
                els = k
            elif re.match(r"#\s*endif", s):
                depth -= 1
                if depth == 0:
                    return k, els
            k += 1

    while i < len(lines):
        s = lines[i].strip()
        m = re.match(r"#\s*ifndef\s+(\w+)", s)
        if m and m.group(1) in defs:
            end, _ = skip_block(i)
            i = end + 1
            continue
        m = re.match(r"#\s*if\s+(.*)", s)
        if m and known(re.sub(r"//.*", "", m.group(1))):
            end, els = skip_block(i)
            take = ev(m.group(1))
            if take:
                body = lines[i + 1: els if els is not None else end]
            else:
                body = lines[els + 1: end] if els is not None else []
            lines[i: end + 1] = body
            continue  # re-scan the kept body (it may hold further conditionals)
        out.append(lines[i])
        i += 1
    text = "\n".join(out)
    for n in names:
        text = re.sub(r"\b%s\b" % n, defs[n], text)
    open(path, "w").write(text)


if __name__ == "__main__":
    main()
