#!/usr/bin/env python3
"""Turn every markdown table that has a row longer than WIDTH into a bullet list (first cell = the item, the other
cells sub-bullets labelled with their column header), so that the file can be wrapped.  usage: FILE [WIDTH=120]"""
import sys


def cells(line):
    parts, cur, i = [], "", 0
    s = line.strip()
    if s.startswith("|"):
        s = s[1:]
    if s.endswith("|"):
        s = s[:-1]
    while i < len(s):
        if s[i] == "\\" and i + 1 < len(s) and s[i + 1] == "|":
            cur += "|"
            i += 2
            continue
        if s[i] == "|":
            parts.append(cur.strip())
            cur = ""
        else:
            cur += s[i]
        i += 1
    parts.append(cur.strip())
    return parts


def convert(path, width=120):
    lines = open(path).read().split("\n")
    out, i = [], 0
    while i < len(lines):
        if lines[i].startswith("|") and i + 1 < len(lines) and set(lines[i + 1].replace("|", "").strip()) <= set("-: "):
            j = i
            while j < len(lines) and lines[j].startswith("|"):
                j += 1
            block = lines[i:j]
            if max(len(b) for b in block) > width:
                head = cells(block[0])
                for row in block[2:]:
                    c = cells(row)
                    out.append("* **%s**" % c[0].strip("*").strip() if not c[0].startswith("`") else "* %s" % c[0])
                    for h, v in zip(head[1:], c[1:]):
                        if v and v != "—":
                            out.append("  - *%s:* %s" % (h, v))
                out.append("")
            else:
                out.extend(block)
            i = j
            continue
        out.append(lines[i])
        i += 1
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    convert(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 120)
