"""Turn the wide tables of a markdown file (any table with a row longer than LIMIT characters) into definition lists: one item per row,
'**first cell** -- ' followed by the other cells, each introduced by its column header.   python tools/detable_md.py FILE [LIMIT=400]"""
import sys

path = sys.argv[1]
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 400
lines = open(path).read().split("\n")
out, i = [], 0


def cells(row):
    # split on unescaped pipes outside `code`
    parts, cur, code = [], "", False
    for ch in row.strip():
        if ch == "`":
            code = not code
        if ch == "|" and not code and not cur.endswith("\\"):
            parts.append(cur.strip()); cur = ""
        else:
            cur += ch
    parts.append(cur.strip())
    return parts[1:-1] if len(parts) >= 2 else parts


while i < len(lines):
    if lines[i].startswith("|") and i + 1 < len(lines) and set(lines[i + 1].replace("|", "").strip()) <= set("-: "):
        j = i
        while j < len(lines) and lines[j].startswith("|"):
            j += 1
        block = lines[i:j]
        if max(len(l) for l in block) > limit:
            hdr = cells(block[0])
            for row in block[2:]:
                c = cells(row)
                item = "- **%s**" % c[0].strip("* ")
                rest = []
                for h, v in zip(hdr[1:], c[1:]):
                    v = v.strip()
                    if v and v not in ("—", "-", "--"):
                        rest.append(("*%s:* " % h if h else "") + v)
                out.append(item + (" -- " + "  ".join(rest) if rest else ""))
            out.append("")
        else:
            out.extend(block)
        i = j
        continue
    out.append(lines[i]); i += 1
open(path, "w").write("\n".join(out))
