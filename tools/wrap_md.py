#!/usr/bin/env python
"""Re-wrap the prose of a markdown file to a column limit (default 120) so that it can be diffed: paragraphs and list items
are wrapped with their hanging indent, table rows / headings / code fences are left alone, and a table whose cells hold
paragraphs (`--tables N`: tables with at least N columns of which the first names the item) is turned into one `####` section
per row with one wrapped bullet per remaining column.
    python tools/wrap_md.py DESIGN.md [--width 120] [--tables 5]"""
import re
import sys
import textwrap


def wrap_line(line, width):
    if len(line) <= width:
        return [line]
    m = re.match(r"^(\s*)((?:[*+-]|\d+\.)\s+)?", line)
    lead, bullet = m.group(1), m.group(2) or ""
    body = line[len(lead) + len(bullet):]
    return textwrap.wrap(body, width=width, initial_indent=lead + bullet, subsequent_indent=lead + " " * len(bullet),
                         break_long_words=False, break_on_hyphens=False) or [line]


def split_row(row):
    cells, cur, depth = [], "", 0
    body = row.strip()
    body = body[1:] if body.startswith("|") else body
    body = body[:-1] if body.endswith("|") else body
    i = 0
    while i < len(body):
        c = body[i]
        if c == "`":
            depth ^= 1
        if c == "|" and not depth and (i == 0 or body[i - 1] != "\\"):
            cells.append(cur.strip())
            cur = ""
        else:
            cur += c
        i += 1
    cells.append(cur.strip())
    return cells


def main():
    path = sys.argv[1]
    width = int(sys.argv[sys.argv.index("--width") + 1]) if "--width" in sys.argv else 120
    tcols = int(sys.argv[sys.argv.index("--tables") + 1]) if "--tables" in sys.argv else 0
    lines = open(path).read().split("\n")
    out, i, fence = [], 0, False
    while i < len(lines):
        ln = lines[i]
        if ln.strip().startswith("```"):
            fence = not fence
        if fence or ln.startswith("#"):
            out.append(ln)
            i += 1
            continue
        if ln.lstrip().startswith("|"):
            j = i
            while j < len(lines) and lines[j].lstrip().startswith("|"):
                j += 1
            rows = lines[i:j]
            head = split_row(rows[0])
            long_cells = any(len(r) > 400 for r in rows)
            if tcols and len(head) >= tcols and long_cells and len(rows) > 2 and set(rows[1].replace("|", "").strip()) <= set("-: "):
                for r in rows[2:]:
                    cells = split_row(r)
                    out.append("#### " + cells[0])
                    out.append("")
                    for h, c in zip(head[1:], cells[1:]):
                        if c:
                            out.extend(wrap_line("* **%s**: %s" % (h, c), width))
                    out.append("")
            else:
                out.extend(rows)
            i = j
            continue
        if not ln.strip():
            out.append(ln)
            i += 1
            continue
        # a prose block: up to the next blank line / table / heading / fence; items start at a bullet, other lines continue the
        # item before them (re-flowed: a source that was hard-wrapped at another width does not end up with stub lines)
        items = []
        while i < len(lines) and lines[i].strip() and not lines[i].lstrip().startswith("|") and not lines[i].startswith("#") \
                and not lines[i].strip().startswith("```"):
            cur = lines[i]
            if re.match(r"^\s*(?:[*+-]|\d+\.)\s+", cur) or not items:
                items.append(cur.rstrip())
            else:
                items[-1] += " " + cur.strip()
            i += 1
        for it in items:
            out.extend(wrap_line(it, width))
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
