#!/usr/bin/env python3
"""Transcribe the reference's known-answer vectors into JSON fixtures.

Reads the `-- ==` / `-- entry:` / `-- input {..}` / `-- output {..}` doc-comment
blocks of a Futhark test file (the `futhark test` data format: `2i64`,
`[0i64,1i64]`, un-suffixed ints are i32, `empty([0][1]i32)`) and writes the
inputs and expected outputs as plain data.  Only literals are extracted; no
reference source text is stored.

Run in the build container (the reference is not present on the GPU box):
    python tests/golden/transcribe_kat.py \
        /root/reference/lib/github.com/diku-dk/sparse/compressed_test.fut \
        tests/golden/compressed_kat.json
    python tests/golden/transcribe_kat.py \
        /root/reference/lib/github.com/diku-dk/sparse/mono_test.fut \
        tests/golden/mono_kat.json
"""
import json
import re
import sys


class P:
    def __init__(self, s):
        self.s = s
        self.i = 0

    def ws(self):
        while self.i < len(self.s) and self.s[self.i].isspace():
            self.i += 1

    def value(self):
        self.ws()
        s = self.s
        if s.startswith("empty(", self.i):
            j = s.index(")", self.i)
            body = s[self.i + 6:j]
            dims = [int(d) for d in re.findall(r"\[(\d+)\]", body)]
            ty = re.sub(r"\[\d+\]", "", body).strip()
            self.i = j + 1
            return {"empty": dims, "type": ty}
        if s[self.i] == "[":
            self.i += 1
            out = []
            self.ws()
            while s[self.i] != "]":
                out.append(self.value())
                self.ws()
                if s[self.i] == ",":
                    self.i += 1
                self.ws()
            self.i += 1
            return out
        m = re.match(r"-?\d+(i64|i32)?", s[self.i:])
        assert m, "cannot parse at: " + s[self.i:self.i + 30]
        self.i += m.end()
        tok = m.group(0)
        ty = "i64" if tok.endswith("i64") else "i32"
        return {"v": int(re.sub(r"i(64|32)$", "", tok)), "type": ty}

    def values(self):
        out = []
        self.ws()
        while self.i < len(self.s):
            out.append(self.value())
            self.ws()
        return out


def simplify(v):
    """{'v':..,'type':..} leaves -> (python value, leaf type)."""
    if isinstance(v, dict) and "v" in v:
        return v["v"], v["type"]
    if isinstance(v, dict) and "empty" in v:
        return {"empty": v["empty"]}, v["type"]
    vals, ty = [], None
    for e in v:
        x, t = simplify(e)
        vals.append(x)
        ty = ty or t
    return vals, ty


def main(src, dst):
    lines = open(src).read().split("\n")
    cases = []
    entry = None
    i = 0
    while i < len(lines):
        ln = lines[i]
        m = re.match(r"--\s*entry:\s*(\w+)", ln)
        if m:
            entry = m.group(1)
        m = re.match(r"--\s*(input|output)\s*\{(.*)", ln)
        if m and entry:
            kind, rest = m.group(1), m.group(2)
            first = i + 1
            body = rest
            while "}" not in body:
                i += 1
                body += " " + re.sub(r"^--", "", lines[i])
            body = body[:body.index("}")]
            vals = [dict(zip(("value", "type"), simplify(v))) for v in P(body).values()]
            if kind == "input":
                cases.append({"entry": entry, "ref": "%s:%d" % (src.split("/")[-1], first),
                              "input": vals})
            else:
                cases[-1]["output"] = vals
                cases[-1]["ref"] += "-%d" % (i + 1)
        i += 1
    doc = {"source": "diku-dk/sparse lib/github.com/diku-dk/sparse/" + src.split("/")[-1],
           "note": "literals of the reference's futhark-test blocks; element type of the module under "
                   "test is i32 (line 5 of the test file)",
           "cases": cases}
    json.dump(doc, open(dst, "w"), indent=1)
    print("wrote %d cases to %s" % (len(cases), dst))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
