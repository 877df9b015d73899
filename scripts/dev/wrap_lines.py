#!/usr/bin/env python3
"""Development: wrap the long lines of a Python source at a column limit WITHOUT changing its meaning -- breaks only after commas (or before a string
literal that follows another one) inside brackets, and splits an over-long plain / f-string literal into adjacent literals at a space outside {...};
the result must have the same AST as the input (checked).   usage: wrap_lines.py FILE [LIMIT=140]"""
import ast
import io
import re
import sys
import tokenize

path, limit = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 140
src = open(path).read()


def split_string_token(tok_text, room):
    """'...long...' -> ['...', '...'] as adjacent literals of the same prefix / quote, each at most `room` wide; None when it cannot be done safely."""
    m = re.match(r"^([rbfRBF]*)(\"\"\"|'''|\"|')", tok_text)
    if not m or len(m.group(2)) == 3 or "\\\n" in tok_text:
        return None
    prefix, q = m.group(1), m.group(2)
    body = tok_text[len(prefix) + 1:-1]
    is_f = "f" in prefix.lower()
    parts, cur, depth, last_space, i = [], "", 0, -1, 0
    while i < len(body):
        ch = body[i]
        if ch == "\\":                                   # keep an escape together
            cur += body[i:i + 2]; i += 2
            continue
        if is_f and ch == "{":
            depth += 0 if body[i:i + 2] == "{{" else 1
            if body[i:i + 2] == "{{":
                cur += "{{"; i += 2
                continue
        elif is_f and ch == "}":
            if body[i:i + 2] == "}}" and depth == 0:
                cur += "}}"; i += 2
                continue
            depth -= 1
        cur += ch
        if ch == " " and depth == 0:
            last_space = len(cur)
        if len(cur) + len(prefix) + 2 >= room and last_space > 0 and depth == 0:
            parts.append(cur[:last_space]); cur = cur[last_space:]; last_space = -1
        i += 1
    parts.append(cur)
    return [f"{prefix}{q}{p}{q}" for p in parts if p] if len(parts) > 1 else None


out_lines = src.split("\n")
for _ in range(12):                                       # a few passes: a wrapped line's tail may still be long
    changed = False
    toks = list(tokenize.generate_tokens(io.StringIO("\n".join(out_lines)).readline))
    by_row = {}
    depth = 0
    for t in toks:
        if t.type == tokenize.OP and t.string in "([{":
            depth += 1
        by_row.setdefault(t.start[0], []).append((t, depth))
        if t.type == tokenize.OP and t.string in ")]}":
            depth -= 1
    edits = {}
    for row, items in by_row.items():
        line = out_lines[row - 1]
        if len(line) <= limit or row in edits:
            continue
        indent = len(line) - len(line.lstrip())
        cont = " " * (indent + 8)
        best = None
        for k, (t, d) in enumerate(items):                # the last comma inside brackets that ends before the limit
            if t.start[0] != t.end[0]:
                continue
            if t.type == tokenize.OP and t.string == "," and d >= 1 and t.end[1] <= limit - 1 and t.end[1] > indent + 20:
                best = t.end[1]
            if t.type == tokenize.STRING and k > 0 and items[k - 1][0].type == tokenize.STRING and d >= 1 and indent + 20 < t.start[1] <= limit - 1:
                best = t.start[1]
        if best is not None and line[best:].strip():
            edits[row] = [line[:best].rstrip(), cont + line[best:].lstrip()]
            continue
        for t, d in items:                                # no such place: an over-long string literal inside brackets
            if t.type == tokenize.STRING and d >= 1 and t.start[0] == t.end[0] == row and t.end[1] > limit and len(t.string) > 40:
                room = max(40, limit - max(t.start[1], indent + 8) - 2)
                pieces = split_string_token(t.string, room)
                if pieces:
                    head, tail = line[:t.start[1]], line[t.end[1]:]
                    new = [head + pieces[0]] if head.strip() else [head + pieces[0]]
                    for p_ in pieces[1:]:
                        new.append(" " * max(t.start[1], indent + 8) + p_)
                    new[-1] += tail
                    edits[row] = new
                    break
    if not edits:
        break
    new_lines = []
    for i, ln in enumerate(out_lines, 1):
        new_lines.extend(edits.get(i, [ln]))
    out_lines = new_lines
    changed = True
result = "\n".join(out_lines)
assert ast.dump(ast.parse(src)) == ast.dump(ast.parse(result)), "the AST changed: nothing written"
open(path, "w").write(result)
print(path, "lines over the limit:", sum(len(x) > limit for x in src.split("\n")), "->", sum(len(x) > limit for x in out_lines))
