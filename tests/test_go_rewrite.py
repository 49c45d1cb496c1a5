"""The call-site rewrite a maintainer runs on the Go module (go-curdleproofs_amd/go/rewrite/rewrite_multiexp.py;
INTEGRATION.md section 2 step 3): the pattern on call shapes written here, and -- where the reference checkout is
present (this container; never the GPU box) -- the count SURVEY.md section 8a gives, 39 non-test sites, taken in
memory: nothing is written, no reference text is kept."""
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("rewrite_multiexp", os.path.join(ROOT, "go-curdleproofs_amd", "go", "rewrite", "rewrite_multiexp.py"))
rw = importlib.util.module_from_spec(spec)
spec.loader.exec_module(rw)


def test_the_rule_on_the_shapes_the_module_uses():
    src = """
	if _, err := acc.MultiExp(bases, scalars, common.MultiExpConf); err != nil {
	if _, err := l.MultiExp(pkg.ToAffine(proof.Points), gammaInv, common.MultiExpConf); err != nil {
	if _, err := M.MultiExp(gs, frs, MultiExpConf); err != nil {
	res, err := other.MultiExp(a, b, ecc.MultiExpConfig{})   // another configuration: not this module's funnel
	x.MultiExpBatch(a, b, common.MultiExpConf)
"""
    out, n = rw.rewrite(src)
    assert n == 3
    assert "common.MultiExp(&acc, bases, scalars)" in out
    assert "common.MultiExp(&l, pkg.ToAffine(proof.Points), gammaInv)" in out
    assert "\tif _, err := MultiExp(&M, gs, frs); err != nil {" in out          # inside package common: unqualified
    assert "other.MultiExp(a, b, ecc.MultiExpConfig{})" in out                   # untouched
    assert "x.MultiExpBatch(a, b, common.MultiExpConf)" in out                  # a different method
    again, m = rw.rewrite(out)
    assert m == 0 and again == out                                               # idempotent


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference checkout is not on this machine")
def test_the_reference_has_the_39_sites_of_the_survey():
    per_dir, total, left = {}, 0, 0
    for d, _, files in os.walk("/root/reference"):
        if os.sep + "." in d:
            continue
        for f in files:
            if not f.endswith(".go") or f.endswith("_test.go"):
                continue
            text = open(os.path.join(d, f), encoding="utf-8").read()
            new, n = rw.rewrite(text)
            total += n
            left += len([m for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\.MultiExp\(", new) if m.group(1) != "common"])
            if n:
                key = os.path.relpath(d, "/root/reference")
                per_dir[key] = per_dir.get(key, 0) + n
    # msmaccumulator 1, IPA 10, GPA 6, SameMSM 15, SamePerm 1, root 4, common 2
    assert per_dir == {"msmaccumulator": 1, "innerproductargument": 10, "grandproductargument": 6, "samemultiscalarargument": 15,
                       "samepermutationargument": 1, ".": 4, "common": 2}
    assert total == 39 and left == 0
