"""The test suite audits itself.  Round 4 spliced 255 lines of tests/test_fullsize_gpu.py in twice; Python keeps the LAST definition of a name, so the lenient round-3 body
of one test shadowed its strict successor for two rounds while the file still collected and passed.  A repeated top-level name in a test module -- or a
repeated method name in a test class -- is an error here, on the CPU, in every round's first check."""
import ast
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _python_files(top):
    for d, _dirs, files in os.walk(top):
        if "__pycache__" in d:
            continue
        for f in sorted(files):
            if f.endswith(".py"):
                yield os.path.join(d, f)


def _repeated_names(body):
    """names bound more than once by def / class statements directly in this body (an `if` / `try` at module level is walked too: a conditional redefinition shadows just the same)"""
    seen, dup = {}, []
    def walk(stmts):
        for n in stmts:
            if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
                if n.name in seen:
                    dup.append((n.name, seen[n.name], n.lineno))
                seen[n.name] = n.lineno
            elif isinstance(n, (ast.If, ast.Try, ast.With)):
                for fld in ("body", "orelse", "finalbody"):
                    walk(getattr(n, fld, []) or [])
                for h in getattr(n, "handlers", []) or []:
                    walk(h.body)
    walk(body)
    return dup


def _audit(path):
    tree = ast.parse(open(path).read(), filename=path)
    bad = [("%s:%d" % (os.path.relpath(path, ROOT), l2), "`%s` already defined at line %d" % (name, l1)) for name, l1, l2 in _repeated_names(tree.body)]
    for n in ast.walk(tree):
        if isinstance(n, ast.ClassDef):
            bad += [("%s:%d" % (os.path.relpath(path, ROOT), l2), "`%s.%s` already defined at line %d" % (n.name, name, l1)) for name, l1, l2 in _repeated_names(n.body)]
    return bad


def test_no_test_module_defines_a_name_twice():
    bad = []
    for path in _python_files(HERE):
        bad += _audit(path)
    assert not bad, "shadowed definitions (the later one is the only one pytest runs):\n" + "\n".join("  %s  %s" % b for b in bad)


def test_the_audit_sees_a_shadowed_test(tmp_path):
    # the round-4 accident in miniature: a strict test followed by its lenient twin
    p = tmp_path / "test_x.py"
    p.write_text("def test_a():\n    assert 1 == 2\n\n\ndef helper():\n    pass\n\n\ndef test_a():\n    pass\n\n\nclass TestK:\n    def test_m(self):\n        pass\n    def test_m(self):\n        pass\n")
    found = _audit(str(p))
    assert len(found) == 2 and "`test_a` already defined at line 1" in found[0][1] and "`TestK.test_m`" in found[1][1]
    q = tmp_path / "test_y.py"
    q.write_text("def test_a():\n    pass\n\n\ndef test_b():\n    def inner():\n        pass\n    def inner2():\n        pass\n")
    assert _audit(str(q)) == []


def test_product_and_tool_modules_define_no_name_twice():
    # the same accident in the product package, the oracle, bench.py or a probe script would be as silent
    bad = []
    for top in ("unitex_amd", "oracle", "tools"):
        for path in _python_files(os.path.join(ROOT, top)):
            bad += _audit(path)
    for f in ("bench.py", "__graft_entry__.py"):
        bad += _audit(os.path.join(ROOT, f))
    assert not bad, "\n".join("  %s  %s" % b for b in bad)
