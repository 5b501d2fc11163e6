"""The Julia files under julia/ cannot be executed here (no Julia in the image).  This is a crude structural guard, not a parser: every
block opener at statement level has its `end`, brackets balance, and each `ccall` names a symbol that include/cmblens.h declares with
the same number of arguments -- the classes of slip an unexecuted file accumulates silently."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = [os.path.join(ROOT, "julia", f) for f in ("CMBLensingHIPExt.jl", "make_reference_fixtures.jl", "test_hipext.jl")]
OPENERS = {"function", "struct", "if", "for", "while", "let", "do", "begin", "module", "try", "quote", "macro"}
IDENT = re.compile("[A-Za-z_-￿][\\w-￿!]*|[()\\[\\]{}]")


def strip(src):
    """remove comments, string and char literals (keeping line structure)"""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if c == "#":
            while i < n and src[i] != "\n":
                i += 1
        elif c == '"':
            if src.startswith('"""', i):
                j = src.index('"""', i + 3) + 3
            else:
                j = i + 1
                while src[j] != '"':
                    j += 2 if src[j] == "\\" else 1
                j += 1
            out.append('""' + "\n" * src[i:j].count("\n"))
            i = j
        else:
            out.append(c)
            i += 1
    return "".join(out)


@pytest.mark.parametrize("path", FILES)
def test_blocks_and_brackets_balance(path):
    src = strip(open(path, encoding="utf-8").read())
    depth, stack = 0, []
    pairs = {")": "(", "]": "[", "}": "{"}
    name = os.path.basename(path)
    for tok in IDENT.finditer(src):
        t = tok.group()
        if t in "([{":
            stack.append(t)
        elif t in ")]}":
            assert stack and stack[-1] == pairs[t], f"{name}: unbalanced {t!r} near offset {tok.start()}"
            stack.pop()
        elif not stack:                                    # statement level only: `a[end]`, generators and comprehensions are inside brackets
            prev = src[max(0, tok.start() - 1):tok.start()]
            if prev in (":", "."):                         # a Symbol literal such as :if, or a field access
                continue
            if t in OPENERS:
                depth += 1
            elif t == "end":
                depth -= 1
                assert depth >= 0, f"{name}: `end` without an opener near offset {tok.start()}"
    assert not stack, f"{name}: unclosed {stack[-1]!r}"
    assert depth == 0, f"{name}: {depth} block(s) left open"


def test_ccalls_match_the_header():
    hdr = open(os.path.join(ROOT, "include", "cmblens.h")).read()
    decl = {}
    for m in re.finditer(r"\b(?:int|const char\*)\s+(cmbl_\w+)\s*\(([^;]*?)\)\s*;", hdr, flags=re.S):
        args = m.group(2).strip()
        decl[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    src = strip(open(FILES[0], encoding="utf-8").read())
    seen = set()
    for m in re.finditer(r"ccall\(\(:(cmbl_\w+),\s*lib\),\s*\w+,\s*\(", src):
        name, i, d = m.group(1), m.end(), 1
        j = i
        while d:                                           # the tuple of argument types
            d += {"(": 1, ")": -1}.get(src[j], 0)
            j += 1
        types = src[i:j - 1].strip().rstrip(",")
        depth, nargs = 0, (1 if types else 0)
        for ch in types:
            depth += {"{": 1, "(": 1, "}": -1, ")": -1}.get(ch, 0)
            nargs += ch == "," and depth == 0
        assert name in decl, f"ccall of {name}, which include/cmblens.h does not declare"
        assert nargs == decl[name], f"{name}: ccall passes {nargs} argument types, the header declares {decl[name]}"
        seen.add(name)
    must = {"cmbl_ctx_create", "cmbl_lenseflow_apply", "cmbl_lenseflow_grad", "cmbl_wiener_cg", "cmbl_gradientf_logpdf", "cmbl_logpdf_mixed",
            "cmbl_grad_logpdf_mixed", "cmbl_dataset_set_op", "cmbl_dot", "cmbl_randn"}
    assert must <= seen, must - seen


def test_the_glue_defaults_to_the_reference_arithmetic():
    """A Julia user who loads the glue unmodified must get the reference's results: `HIPLenseFlow` defaults to the aliased
    δϕ velocity of src/lenseflow.jl:198-200, a context starts with working-precision sums (src/util.jl:288-316), the consistent form is a
    keyword, and julia/test_hipext.jl compares DEFAULT-constructed operators with the reference and asserts the default."""
    ext = strip(open(FILES[0], encoding="utf-8").read())
    assert re.search(r"reference_exact\(\)\s*=\s*get\(ENV,\s*\"\",\s*\"\"\)\s*in\s*\(\"\",\s*\"\"\)", ext)             # default true; CMBL_CONSISTENT=1 flips it
    assert len(re.findall(r"HIPLenseFlow\([^)]*;\s*alias_quirk=reference_exact\(\)\)", ext)) == 2                  # both constructors
    assert re.search(r"reference_exact\(\)\s*&&\s*chk\(ccall\(\(:cmbl_set_sum_accuracy_mode,\s*lib\)[^\n]*h\[\],\s*0\)\)", ext)
    tst = strip(open(FILES[2], encoding="utf-8").read())
    assert "L = HIPLenseFlow(ϕg, 7)" in tst and "L.alias_quirk === true" in tst
    assert "alias_quirk=true" not in tst                                                    # no test forces the mode: the default is what is compared
