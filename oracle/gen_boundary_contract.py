"""Extract the BOUNDARY CONTRACT of the hot path from the reference's own callers -- TEST INFRASTRUCTURE ONLY.

Parses (ast, nothing is imported or executed) the reference scripts that drive the path
    pretrain_gpt2.py, generate_samples.py, generation/sampling.py, utils.py, fp16/fp16.py, fp16/fp16util.py,
    model/gpt2_modeling.py, model/distributed.py, preprocess/preprocess_text_image_data.py, preprocess/utils.py
and records every name they take from the packages this repo mirrors (mpu, model, fp16, vqvae): `from pkg import name`,
`pkg.name` attribute reads, and for every CALL of such a name the number of positional arguments and the keyword names
used -- i.e. what a drop-in replacement must resolve and accept.  Writes
tests/golden/boundary_contract.json; tests/test_boundary_contract.py checks cogview_amd against it (and, where
/root/reference is present, that the committed file is current).

    python oracle/gen_boundary_contract.py [/root/reference] [out.json]
"""
import ast
import json
import os
import sys

PACKAGES = ("mpu", "model", "fp16", "vqvae")
FILES = ("pretrain_gpt2.py", "generate_samples.py", "generation/sampling.py", "utils.py", "fp16/fp16.py",
         "fp16/fp16util.py", "model/gpt2_modeling.py", "model/distributed.py",
         "preprocess/preprocess_text_image_data.py", "preprocess/utils.py")


def extract(ref_root):
    uses = {}          # (pkg, name) -> {"files": {file: [lines]}, "calls": [{"file", "line", "nargs", "kwargs"}]}

    def rec(pkg, name, fname, line):
        e = uses.setdefault((pkg, name), {"files": {}, "calls": []})
        e["files"].setdefault(fname, [])
        if line not in e["files"][fname]:
            e["files"][fname].append(line)
        return e

    for fname in FILES:
        path = os.path.join(ref_root, fname)
        if not os.path.exists(path):
            continue
        tree = ast.parse(open(path).read(), filename=path)
        own_pkg = fname.split("/")[0] if "/" in fname else None
        alias = {}         # local name -> (pkg, name) for `from pkg import name [as alias]`
        mods = {}          # local name -> pkg for `import pkg [as alias]`
        for node in ast.walk(tree):
            if isinstance(node, ast.Import):
                for a in node.names:
                    if a.name in PACKAGES:
                        mods[a.asname or a.name] = a.name
            elif isinstance(node, ast.ImportFrom) and node.module:
                top = node.module.split(".")[0]
                if node.level == 0 and top in PACKAGES and top != own_pkg:
                    for a in node.names:
                        if a.name == "*":
                            continue
                        if node.module == top:
                            alias[a.asname or a.name] = (top, a.name)
                            rec(top, a.name, fname, node.lineno)
        # attribute reads / calls, scope-aware: a function parameter or local variable that shadows an imported package
        # name (utils.py: `def save_checkpoint(iteration, model, ...)` under `import model`) is not the package
        def local_names(fn):
            names = {a.arg for a in fn.args.args + fn.args.kwonlyargs + fn.args.posonlyargs}
            if fn.args.vararg:
                names.add(fn.args.vararg.arg)
            if fn.args.kwarg:
                names.add(fn.args.kwarg.arg)
            for n in ast.walk(fn):
                if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store):
                    names.add(n.id)
            return names

        def visit(node, shadow):
            if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
                shadow = shadow | (local_names(node) if not isinstance(node, ast.Lambda) else {a.arg for a in node.args.args})
            if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id in mods \
                    and node.value.id not in shadow:
                rec(mods[node.value.id], node.attr, fname, node.lineno)
            if isinstance(node, ast.Call):
                tgt = None
                f = node.func
                if isinstance(f, ast.Attribute) and isinstance(f.value, ast.Name) and f.value.id in mods \
                        and f.value.id not in shadow:
                    tgt = (mods[f.value.id], f.attr)
                elif isinstance(f, ast.Name) and f.id in alias and f.id not in shadow:
                    tgt = alias[f.id]
                if tgt is not None and not any(isinstance(a, ast.Starred) for a in node.args):
                    rec(tgt[0], tgt[1], fname, node.lineno)["calls"].append(
                        {"file": fname, "line": node.lineno, "nargs": len(node.args),
                         "kwargs": sorted(k.arg for k in node.keywords if k.arg is not None)})
            for child in ast.iter_child_nodes(node):
                visit(child, shadow)

        visit(tree, frozenset())
    out = []
    for (pkg, name), e in sorted(uses.items()):
        out.append({"package": pkg, "name": name,
                    "where": [f"{f}:{','.join(str(l) for l in sorted(ls))}" for f, ls in sorted(e["files"].items())],
                    "calls": e["calls"]})
    return out


INSTANCE_FILES = ("pretrain_gpt2.py", "utils.py", "generate_samples.py", "generation/sampling.py")
INSTANCE_VARS = ("optimizer", "model", "lr_scheduler")


def extract_instance_uses(ref_root):
    """Attributes the callers read on the objects they get back from the mirrored constructors -- the variables they
    name `optimizer` (FP16_Optimizer or the bare optimizer), `model` (DDP(FP16_Module(GPT2Model)) or a DeepSpeed engine)
    and `lr_scheduler` -- with the call shapes.  Which class must carry which attribute is decided by the test (the
    DeepSpeed-engine-only ones are listed there)."""
    out = {v: {} for v in INSTANCE_VARS}
    for fname in INSTANCE_FILES:
        path = os.path.join(ref_root, fname)
        if not os.path.exists(path):
            continue
        tree = ast.parse(open(path).read(), filename=path)
        parents = {}
        for node in ast.walk(tree):
            for ch in ast.iter_child_nodes(node):
                parents[ch] = node
        for node in ast.walk(tree):
            if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id in INSTANCE_VARS:
                e = out[node.value.id].setdefault(node.attr, {"where": [], "calls": []})
                loc = f"{fname}:{node.lineno}"
                if loc not in e["where"]:
                    e["where"].append(loc)
                par = parents.get(node)
                if isinstance(par, ast.Call) and par.func is node and not any(isinstance(a, ast.Starred) for a in par.args):
                    e["calls"].append({"where": loc, "nargs": len(par.args),
                                       "kwargs": sorted(k.arg for k in par.keywords if k.arg is not None)})
    return out


if __name__ == "__main__":
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                            "tests", "golden", "boundary_contract.json")
    data = {"source": "THUDM/CogView reference scripts, parsed with ast by oracle/gen_boundary_contract.py",
            "files": list(FILES), "uses": extract(ref), "instance_uses": extract_instance_uses(ref)}
    with open(dst, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
        f.write("\n")
    print(f"{len(data['uses'])} names -> {dst}")
