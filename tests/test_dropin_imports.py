"""CPU: the documented drop-in recipe resolves every `lvmhelpers.*` and
`entmax` name the reference's three train.py scripts import -- the hot-path modules from this
repository, the sampling baselines (not rebuilt) from the reference checkout that follows on the
path.  Skipped where /root/reference is absent (the GPU box)."""
import ast
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "sparse-marginalization-lvm_amd")
REF = "/root/reference"
SCRIPTS = ["experiments/signal-game/train.py", "experiments/bit_vector-vae/train.py",
           "experiments/semi_supervised-vae/train.py"]
OURS = {"lvmhelpers.marg", "lvmhelpers.structmarg", "lvmhelpers.sparsemap", "lvmhelpers.utils",
        "lvmhelpers.pbinary_topk", "lvmhelpers.pbernoulli", "lvmhelpers.pbudget", "entmax"}
THEIRS = {"lvmhelpers.sfe", "lvmhelpers.nvil", "lvmhelpers.gumbel", "lvmhelpers.sum_and_sample"}


def _wanted():
    """{module: set(names)} for every `from lvmhelpers.x import a, b` / `from entmax import ...`."""
    want = {}
    for rel in SCRIPTS:
        tree = ast.parse(open(os.path.join(REF, rel)).read())
        for node in ast.walk(tree):
            if isinstance(node, ast.ImportFrom) and node.module and (
                    node.module.split(".")[0] in ("lvmhelpers", "entmax")):
                want.setdefault(node.module, set()).update(a.name for a in node.names)
    return want


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout")
def test_train_scripts_imports_resolve():
    want = _wanted()
    assert OURS & set(want) and THEIRS <= set(want)
    prog = ["import importlib, json, sys", "out = {}"]
    prog.append(f"want = {dict((k, sorted(v)) for k, v in want.items())!r}")
    prog.append("for mod, names in want.items():\n"
                "    m = importlib.import_module(mod)\n"
                "    missing = [n for n in names if not hasattr(m, n)]\n"
                "    out[mod] = [m.__file__, missing]")
    prog.append("print(json.dumps(out))")
    env = dict(os.environ, PYTHONPATH=f"{PKG}:{REF}")
    res = subprocess.run([sys.executable, "-c", "\n".join(prog)], env=env, capture_output=True,
                         text=True, cwd="/tmp")
    assert res.returncode == 0, res.stderr[-2000:]
    import json
    got = json.loads(res.stdout.strip().splitlines()[-1])
    for mod, (path, missing) in got.items():
        assert not missing, (mod, missing)
        if mod in THEIRS:
            assert path.startswith(REF), (mod, path)
        else:
            assert path.startswith(PKG), (mod, path)
    # and the modules only we ship still win over same-named reference files
    for mod in ("lvmhelpers.marg", "lvmhelpers.structmarg", "lvmhelpers.utils"):
        assert got[mod][0].startswith(PKG)


def test_package_is_a_fall_through_package():
    import lvmhelpers
    assert any(p.startswith(PKG) for p in lvmhelpers.__path__)
    assert list(lvmhelpers.__path__)[0].startswith(PKG)
