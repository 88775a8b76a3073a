"""Documentation rot guard (CPU): every `profiles/...` artefact, `scripts/...` file and `tests/...` file that DESIGN.md,
README.md, INTEGRATION.md or profiles/README.md cite must exist in the tree (brace lists `r4_gemm_{a3,wide}_probe.txt` and
`*` globs are expanded).  References to files that live only in the history must say so by not using a path."""
from __future__ import annotations

import itertools
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
DOCS = ["DESIGN.md", "README.md", "INTEGRATION.md", "profiles/README.md"]
PATH_RE = re.compile(r"(?<![\w/.-])((?:profiles|scripts|tests|oracle|include|diarizen_amd|testkit)/[\w./{},*-]+)")


def expand(token: str):
    m = re.search(r"\{([^{}]*)\}", token)
    if not m:
        return [token]
    return list(itertools.chain.from_iterable(expand(token[:m.start()] + alt + token[m.end():]) for alt in m.group(1).split(",")))


def cited_paths(doc: str):
    text = (ROOT / doc).read_text()
    out = set()
    for tok in PATH_RE.findall(text):
        tok = tok.rstrip(".,:;)")
        if tok.endswith("/") or "." not in tok.rsplit("/", 1)[-1].strip("*"):
            continue                       # directories, bare names without an extension
        out.update(expand(tok))
    return sorted(out)


@pytest.mark.parametrize("doc", DOCS)
def test_cited_files_exist(doc):
    missing = []
    for p in cited_paths(doc):
        if "*" in p:
            if not list(ROOT.glob(p)):
                missing.append(p)
        elif not (ROOT / p).exists():
            # kernels' sources are cited as csrc/<file>; build products (lib/*.so) are not in the tree on purpose
            if p.endswith(".so") or "/build/" in p or "/lib/" in p:
                continue
            missing.append(p)
    assert not missing, f"{doc} cites files that are not in the tree: {missing}"


@pytest.mark.parametrize("doc", DOCS)
def test_cited_test_names_are_defined(doc):
    """`test_...` names quoted in the documents are functions in tests/ (a trailing `*` or `_` quotes a family by prefix);
    file names (`test_x.py`, `test_x.c`) and the reference's own tests (`pyannote-audio/tests/...`) are not names of ours."""
    defs, files = set(), set()
    for f in (ROOT / "tests").glob("*.py"):
        files.add(f.stem)
        defs.update(re.findall(r"^def (test_\w+)", f.read_text(), re.M))
    text = (ROOT / doc).read_text()
    unknown = set()
    for m in re.finditer(r"(?<![\w/])(test_[a-z0-9_]+)(\*?)", text):
        name, star = m.group(1), m.group(2)
        if text[m.end():m.end() + 3] in (".py", ".c ", ".c)", ".c`") or text[m.end():m.end() + 2] == ".c" or name in files:
            continue
        if "pyannote-audio/tests" in text[max(0, m.start() - 48):m.start()]:
            continue
        family = bool(star) or name.endswith("_")
        if name in defs or (family and any(d.startswith(name) for d in defs)):
            continue
        unknown.add(name)
    assert not unknown, f"{doc} quotes tests that do not exist: {sorted(unknown)}"
