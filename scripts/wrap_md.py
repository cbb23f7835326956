#!/usr/bin/env python3
"""Re-wrap the prose of a Markdown file to a maximum line width (default 120): paragraphs and list items (hanging indent kept),
nothing inside ``` fences, no table rows, no headings.  usage: wrap_md.py FILE [width]   (in place)"""
import re
import sys
import textwrap


def wrap(text, width):
    out, para, fence = [], [], False
    first_indent = rest_indent = ""

    def flush():
        nonlocal para
        if para:
            body = " ".join(s.strip() for s in para)
            out.extend(textwrap.wrap(body, width=width, initial_indent=first_indent, subsequent_indent=rest_indent, break_long_words=False, break_on_hyphens=False))
            para = []

    for line in text.split("\n"):
        if line.lstrip().startswith("```"):
            flush()
            fence = not fence
            out.append(line)
            continue
        if fence or line.startswith("|") or line.startswith("#") or not line.strip():
            flush()
            out.append(line)
            continue
        m = re.match(r"^(\s*)((?:[*\-+]|\d+\.)\s+)", line)
        if m:  # a new list item
            flush()
            first_indent, rest_indent = m.group(1), " " * len(m.group(1) + m.group(2))
            para = [m.group(2) + line[m.end():]]
            continue
        indent = re.match(r"^\s*", line).group(0)
        if not para:
            first_indent = rest_indent = indent
        elif len(indent) < len(rest_indent):  # a paragraph that leaves the list
            flush()
            first_indent = rest_indent = indent
        para.append(line)
    flush()
    return "\n".join(out)


if __name__ == "__main__":
    path, width = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 120
    with open(path) as fh:
        src = fh.read()
    with open(path, "w") as fh:
        fh.write(wrap(src, width))
