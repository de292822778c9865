#!/usr/bin/env python3
"""Re-wraps the paragraphs of DESIGN.md that hold a line over 120 columns (tables, headings and code blocks are left alone)."""
import re, sys, textwrap
p = sys.argv[1] if len(sys.argv) > 1 else "DESIGN.md"
lines = open(p, encoding="utf-8").read().split("\n")
out, i, incode = [], 0, False
par = lambda l: l.strip() != "" and not l.startswith("|") and not l.startswith("#") and not l.startswith("```")
while i < len(lines):
    l = lines[i]
    if l.startswith("```"):
        incode = not incode
    if incode or l.startswith("```") or not par(l):
        out.append(l); i += 1; continue
    blk = [l]; i += 1
    while i < len(lines) and par(lines[i]) and not re.match(r"^(\s*)([*-]|\d+\.)\s", lines[i]):
        blk.append(lines[i]); i += 1
    if max(len(x) for x in blk) <= 120:
        out.extend(blk); continue
    lead = re.match(r"^(\s*(?:[*-]|\d+\.)\s+|\s*)", blk[0]).group(1)
    hang = re.match(r"^(\s*)", blk[1]).group(1) if len(blk) > 1 else " " * len(lead)
    text = " ".join(x.strip() for x in blk)
    if lead.strip() and text.startswith(lead.strip()):
        text = text[len(lead.strip()):].strip()
    out.extend(textwrap.wrap(text, width=118, initial_indent=lead, subsequent_indent=hang, break_long_words=False, break_on_hyphens=False))
open(p, "w", encoding="utf-8").write("\n".join(out))
