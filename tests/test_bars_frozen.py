"""The parity bars are frozen (round 6): tests/bars.py is the one table, tests/golden/bars_frozen.json its snapshot at the freeze.
A bar may only get LOOSER with a `BARS_CHANGELOG: <name> <old> -> <new> <why>` line in DESIGN.md section 3 -- the review of round 5
counted five widenings in one round, each argued, all one-way."""
import io
import json
import os
import re
import tokenize

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _changelog():
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    return {m.group(1): (float(m.group(2)), float(m.group(3))) for m in re.finditer(r"BARS_CHANGELOG:\s*`?(\w+)`?\s+([0-9.eE+-]+)\s*->\s*([0-9.eE+-]+)", text)}


def test_no_bar_is_looser_than_the_frozen_table_without_a_changelog_entry():
    from tests.bars import BARS
    frozen = json.load(open(os.path.join(HERE, "golden", "bars_frozen.json")))
    log = _changelog()
    problems = []
    for name, old in frozen.items():
        if name not in BARS:
            problems.append(f"{name}: frozen bar no longer in tests/bars.py")
            continue
        new = BARS[name][0]
        if new > old and not (name in log and log[name] == (old, new)):
            problems.append(f"{name}: {old} -> {new} is looser and DESIGN.md has no `BARS_CHANGELOG: {name} {old} -> {new} ...` line")
    for name, (value, commit, what) in BARS.items():
        assert isinstance(value, float) and commit and what, name
        if name not in frozen and name not in log:
            problems.append(f"{name}: a bar that is neither in the frozen snapshot nor in the changelog (new bars need an entry: `BARS_CHANGELOG: {name} 0 -> {value} ...`)")
    assert not problems, "\n".join(problems)


def _float_literals(path):
    """Float literals in CODE (comments and strings do not count) -> [(line, text)]."""
    out = []
    for tok in tokenize.generate_tokens(io.StringIO(open(path).read()).readline):
        if tok.type == tokenize.NUMBER and re.search(r"[.eE]", tok.string) and not tok.string.lower().startswith("0x"):
            out.append((tok.start[0], tok.string, tok.line))
    return out


def test_no_tolerance_literal_outside_the_table():
    """gpu_util.py and test_gpu_fullsize.py take every tolerance from tests/bars.py.  What may stay a literal: division guards (1e-20, 1e-30),
    the constants of formulas (0.5, 1.0, 2.0 ...: no exponent), and scene parameters (lines that build a scene)."""
    allowed_values = {"1e-20", "1e-30"}
    scene_words = ("scale_lo", "scale_hi", "synthetic_gaussians", "posed_scene", "clustered_gaussians", "spread", "seed")
    bad = []
    for f in ("gpu_util.py", "test_gpu_fullsize.py"):
        for line, text, src in _float_literals(os.path.join(HERE, f)):
            if "e" not in text.lower() or text in allowed_values:
                continue
            if any(w in src for w in scene_words):
                continue
            bad.append(f"{f}:{line}: {text}   {src.strip()[:120]}")
    assert not bad, "tolerance literals outside tests/bars.py:\n" + "\n".join(bad)
