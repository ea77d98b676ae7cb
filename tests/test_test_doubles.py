"""Static check (CPU, no GPU): every function a test installs over a PRODUCT function with
`monkeypatch.setattr(Target, 'name', replacement)` accepts the calls the product makes of the real one.

Round 3 was handed in red because `PassportLayerBase._layer` had grown a sixth positional argument and a GPU-only test
still installed a five-argument replacement; nothing that runs in the CPU container noticed.  This test parses every
file under tests/, resolves `Target` through the file's own imports and compares the two signatures, so a signature
change in the product fails HERE, before any GPU session.  (tests/conftest.py applies the same rule at patch time.)"""
import ast
import importlib
import inspect
import os

import pytest

from tests.conftest import _accepts_call_of

HERE = os.path.dirname(os.path.abspath(__file__))


def _patch_sites():
    sites = []
    for fn in sorted(os.listdir(HERE)):
        if not fn.endswith('.py') or fn == os.path.basename(__file__):
            continue
        tree = ast.parse(open(os.path.join(HERE, fn)).read(), fn)
        imports = {}
        for node in ast.walk(tree):
            if isinstance(node, ast.ImportFrom) and node.module and node.level == 0:
                for a in node.names:
                    imports[a.asname or a.name] = (node.module, a.name)
            elif isinstance(node, ast.Import):
                for a in node.names:
                    imports[a.asname or a.name.split('.')[0]] = (a.name if a.asname else a.name.split('.')[0], None)
        defs = {}
        for node in ast.walk(tree):
            if isinstance(node, (ast.FunctionDef, ast.Lambda)):
                defs.setdefault(getattr(node, 'name', None), []).append(node)
        for node in ast.walk(tree):
            if not (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == 'setattr'
                    and isinstance(node.func.value, ast.Name) and node.func.value.id == 'monkeypatch'
                    and len(node.args) == 3):
                continue
            target, name, value = node.args
            if not (isinstance(target, ast.Name) and isinstance(name, ast.Constant) and isinstance(name.value, str)):
                continue
            if not (isinstance(value, ast.Name) and value.id in defs):
                continue                                  # an object / a call result: judged at patch time (conftest)
            sites.append((fn, node.lineno, imports.get(target.id), name.value, defs[value.id][-1]))
    return sites


def _positional(fdef):
    a = fdef.args
    n = len(a.posonlyargs) + len(a.args)
    return n, n - len(a.defaults), a.vararg is not None


def test_the_scan_sees_the_known_patch_site():
    sites = [(f, attr) for f, _l, _imp, attr, _d in _patch_sites()]
    assert any(attr == '_layer' for _f, attr in sites), sites      # the whole-net backward test's gate


@pytest.mark.parametrize('site', _patch_sites(), ids=lambda s: '%s:%d:%s' % (s[0], s[1], s[3]))
def test_replacement_accepts_the_products_calls(site):
    fn, line, imp, attr, fdef = site
    assert imp is not None, '%s:%d: cannot resolve the patched object' % (fn, line)
    module, obj = imp
    target = importlib.import_module(module)
    if obj is not None:
        target = getattr(target, obj)
    real = getattr(target, attr)
    if not inspect.isfunction(real):
        pytest.skip('not a python function')
    sig = inspect.signature(real)
    P = inspect.Parameter
    pos_real = [p for p in sig.parameters.values() if p.kind in (P.POSITIONAL_ONLY, P.POSITIONAL_OR_KEYWORD)]
    n_new, required_new, var = _positional(fdef)
    assert var or n_new >= len(pos_real), (
        '%s:%d: %s takes %d positional arguments, the product calls %s%s' % (fn, line, fdef.name, n_new,
                                                                             real.__qualname__, sig))
    assert required_new <= len(pos_real)


def test_the_rule_itself():
    def real(self, a, b, c=None): pass
    def ok(self, a, b, c=None): pass
    def ok2(self, *args): pass
    def short(self, a, b): pass
    assert _accepts_call_of(ok, real) and _accepts_call_of(ok2, real) and not _accepts_call_of(short, real)


def test_patch_time_guard_refuses_a_short_replacement(monkeypatch):
    class Product:
        def method(self, a, b, c=None):
            return a, b, c

    def short(self, a, b):
        return None
    with pytest.raises(TypeError, match='cannot take the calls'):
        monkeypatch.setattr(Product, 'method', short)
    monkeypatch.setattr(Product, 'method', lambda self, a, b, c=None: 'ok')
    assert Product().method(1, 2, 3) == 'ok'
    monkeypatch.setattr(Product, 'flag', None, raising=False)      # non-callables pass through untouched
