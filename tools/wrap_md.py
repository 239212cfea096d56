"""Wrap the prose of a markdown file to a column limit (tables, code fences and headings are left alone; list items keep their hanging indent).
    python tools/wrap_md.py FILE [width=150]"""
import re
import sys
import textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 150
out, fence = [], False
for line in open(path).read().split("\n"):
    if line.lstrip().startswith("```"):
        fence = not fence
        out.append(line); continue
    if fence or len(line) <= width or line.lstrip().startswith(("|", "#")):
        out.append(line); continue
    m = re.match(r"^(\s*(?:[-*+]|\d+\.)\s+|\s*)", line)
    lead = m.group(1)
    hang = " " * len(lead)
    out.extend(textwrap.wrap(line[len(lead):], width=width, initial_indent=lead, subsequent_indent=hang, break_long_words=False, break_on_hyphens=False))
open(path, "w").write("\n".join(out))
