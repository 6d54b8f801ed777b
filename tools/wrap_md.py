#!/usr/bin/env python3
"""Re-wrap a markdown file to a column limit without touching code blocks, tables, headings or indented code.
Bullets and numbered items keep their hanging indent.  usage: tools/wrap_md.py FILE [WIDTH=118]"""
import re
import sys
import textwrap


def wrap_file(path, width=118):
    out, para, indent_first, indent_rest = [], [], "", ""
    in_code = False

    def flush():
        nonlocal para
        if para:
            text = " ".join(s.strip() for s in para)
            out.extend(textwrap.wrap(text, width=width, initial_indent=indent_first, subsequent_indent=indent_rest,
                                     break_long_words=False, break_on_hyphens=False))
            para = []

    for line in open(path).read().split("\n"):
        if line.strip().startswith("```"):
            flush()
            in_code = not in_code
            out.append(line)
            continue
        if in_code or line.startswith("|") or line.startswith("#") or line.startswith("    ") and not para:
            flush()
            out.append(line)
            continue
        if not line.strip():
            flush()
            out.append("")
            continue
        m = re.match(r"^(\s*)([*+-]|\d+\.)\s+", line)
        if m:
            flush()
            indent_first = m.group(0)
            indent_rest = " " * len(m.group(0))
            para = [line[len(m.group(0)):]]
            indent_first = m.group(0)
            continue
        if not para:
            lead = re.match(r"^\s*", line).group(0)
            indent_first = indent_rest = lead
        para.append(line)
    flush()
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    wrap_file(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 118)
