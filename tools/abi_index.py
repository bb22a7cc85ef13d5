#!/usr/bin/env python3
"""Regenerates the entry-point index at the end of INTEGRATION.md from the doc comments of include/gradslam_hip.h
(one row per declaration group: symbols, and the first sentences of the comment that cites what they replace)."""
import os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MARK = "## 7. Entry-point index"
h = open(os.path.join(ROOT, "include", "gradslam_hip.h")).read()
pat = re.compile(r"/\*(.*?)\*/\s*((?:(?:int|size_t|void|const char \*)\s*\*?gs_[a-z0-9_]+\([^;]*\);\s*)+)", re.S)
rows = []
for m in pat.finditer(h):
    text = " ".join(x.strip(" *") for x in m.group(1).strip().splitlines())
    text = re.sub(r"-{8,}[^A-Za-z]*", "", text).strip()
    if text.startswith("gradslam_hip.h --"):  # the file header precedes the first declaration
        text = "vertex / normal maps of a frame batch, local and global (structures/rgbdimages.py:643-762), one launch"
    names = re.findall(r"(gs_[a-z0-9_]+)\(", m.group(2))
    sent = re.split(r"(?<=[.)])\s+(?=[A-Z*])", text)
    short = sent[0] if len(sent[0]) > 60 or len(sent) == 1 else sent[0] + " " + sent[1]
    rows.append((names, short[:330].replace("|", "/")))
declared = set(re.findall(r"\b(gs_[a-z0-9_]+)\(", h))
listed = {n for names, _ in rows for n in names}
rest = sorted(declared - listed)
out = [MARK, "", "Generated from the doc comments of `include/gradslam_hip.h` (`python tools/abi_index.py`); file:line are the",
       "reference's.  `*_ws_bytes` / `*_tape_bytes` size the caller-provided workspace of the call they are named after.", "",
       "| entry points | what they compute / replace |", "|---|---|"]
for names, short in rows:
    out.append("| " + ", ".join("`%s`" % n for n in names) + " | " + short + " |")
if rest:
    out.append("| " + ", ".join("`%s`" % n for n in rest) + " | library housekeeping: ABI version, text of the last error of the calling thread |")
p = os.path.join(ROOT, "INTEGRATION.md")
s = open(p).read()
if MARK in s:
    s = s[: s.index(MARK)].rstrip("\n") + "\n\n"
else:
    s = s.rstrip("\n") + "\n\n"
open(p, "w").write(s + "\n".join(out) + "\n")
print("rows", len(rows), "symbols", len(listed | set(rest)), "of", len(declared))
