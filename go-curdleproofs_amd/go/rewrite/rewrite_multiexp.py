#!/usr/bin/env python3
"""Routes every MSM of jsign/go-curdleproofs through ONE function: rewrites, across a checkout of the module,

    X.MultiExp(points, scalars, common.MultiExpConf)   ->   common.MultiExp(&X, points, scalars)
    X.MultiExp(points, scalars, MultiExpConf)           ->   MultiExp(&X, points, scalars)          (inside package common)

which is step 3 of INTEGRATION.md section 2 (the funnel `common.MultiExp` itself is step 2: add it AFTER running
this, or it is rewritten into a call of itself).  Expected on the reference as surveyed:
39 non-test call sites -- 37 outside package common, 2 inside (common/util.go) -- and none left afterwards.

    python3 rewrite_multiexp.py --check /path/to/go-curdleproofs      # dry run: per-file counts, exit 1 unless 39
    python3 rewrite_multiexp.py --write /path/to/go-curdleproofs      # in place (then: gofmt -l . ; go build ./... ; go vet ./...)

The same change as two gofmt rules, for a box with Go (single lower-case letters are gofmt's wildcards):

    gofmt -w -r 'a.MultiExp(b, c, common.MultiExpConf) -> common.MultiExp(&a, b, c)' .
    gofmt -w -r 'a.MultiExp(b, c, MultiExpConf) -> MultiExp(&a, b, c)' common/

This script holds no text of the reference: it is a pattern and a replacement.  Every receiver at the 39 sites is an
addressable bls12381.G1Jac variable (the method has a pointer receiver, so Go takes its address implicitly today).
Test files are left alone (they do not call MultiExp)."""
import argparse
import os
import re
import sys

# an argument: anything without commas at nesting depth 0, with one level of parentheses allowed
# (bls12381.BatchJacobianToAffineG1(proof.L_Cs) is the deepest the module has)
ARG = r"(?:[^(),]|\([^()]*\))+"
CALL = re.compile(r"\b([A-Za-z_][A-Za-z0-9_]*)\.MultiExp\(\s*(" + ARG + r")\s*,\s*(" + ARG + r")\s*,\s*(common\.)?MultiExpConf\s*\)")


def rewrite(text):
    """-> (new text, sites rewritten)."""
    return CALL.subn(lambda m: f"{m.group(4) or ''}MultiExp(&{m.group(1)}, {m.group(2).strip()}, {m.group(3).strip()})", text)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    g = ap.add_mutually_exclusive_group(required=True)
    g.add_argument("--check", metavar="DIR")
    g.add_argument("--write", metavar="DIR")
    ap.add_argument("--expect", type=int, default=39, help="call sites the module is expected to have (default 39)")
    a = ap.parse_args()
    root = a.check or a.write
    total, left = 0, 0
    for d, _, files in os.walk(root):
        if os.sep + "." in d:
            continue
        for f in sorted(files):
            if not f.endswith(".go") or f.endswith("_test.go"):
                continue
            path = os.path.join(d, f)
            text = open(path, encoding="utf-8").read()
            new, n = rewrite(text)
            left += len([m for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\.MultiExp\(", new) if m.group(1) != "common"])   # another shape of the call
            if n:
                print(f"{n:3d}  {os.path.relpath(path, root)}")
                total += n
                if a.write:
                    open(path, "w", encoding="utf-8").write(new)
    print(f"{total} call sites {'rewritten' if a.write else 'found'}; {left} X.MultiExp( calls of another shape left")
    return 0 if total == a.expect and left == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
