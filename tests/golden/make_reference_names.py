#!/usr/bin/env python
"""Writes tests/golden/reference_names.json: the public top-level names of the reference modules
the shim stands in for, read from their SOURCE under /root/reference (ast for .py, a regex for the
two Cython modules) -- names only, no code.  `hot_path` marks the ones SURVEY.md section 8 puts on
the path (rows a1-a17, f2, f4); tests/test_host_logic.py checks that every one of them resolves
after `taiyaki_amd.shim.install()`, and that `taiyaki.ctc` is complete.

    python tests/golden/make_reference_names.py        (in the build container only)"""
import ast
import json
import os
import re

REF = "/root/reference/taiyaki"
HERE = os.path.dirname(os.path.abspath(__file__))

MODULES = {
    "ctc": "ctc/ctc.pyx",
    "decodeutil": "decodeutil/decodeutil.pyx",
    "layers": "layers.py",
    "decode": "decode.py",
    "flipflopfings": "flipflopfings.py",
    "flipflop_remap": "flipflop_remap.py",
    "qscores": "qscores.py",
    "basecall_helpers": "basecall_helpers.py",
}
# what of each module is on the path (None = every public name)
HOT = {
    "ctc": None,
    "decodeutil": None,
    "layers": ["flipflop_logpartition", "log_partition_flipflop", "global_norm_flipflop", "GlobalNormFlipFlop",
               "GlobalNormFlipFlopCatMod", "Convolution", "Lstm", "GruMod", "Reverse", "Serial"],
    "decode": None,
    "flipflopfings": ["move_indices", "stay_indices", "flopmask", "flipflop_code", "path_to_str", "nstate_flipflop",
                      "nbase_flipflop"],
    "flipflop_remap": None,
    "qscores": None,
    "basecall_helpers": ["stitch_chunks"],
}


def names_py(path):
    tree = ast.parse(open(path).read())
    out = []
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and not node.name.startswith("_"):
            out.append(node.name)
        elif isinstance(node, ast.Assign):
            for t in node.targets:
                # aliases of operators (`crf_flipflop_loss = FlipFlopCRF.apply`)
                if isinstance(t, ast.Name) and not t.id.startswith("_") and t.id.islower() and \
                        isinstance(node.value, ast.Attribute) and node.value.attr == "apply":
                    out.append(t.id)
    return out


def names_pyx(path):
    out = []
    for line in open(path):
        m = re.match(r"^(?:def|cpdef|class)\s+([A-Za-z]\w*)", line) or re.match(r"^([a-z]\w*)\s*=\s*\w+\.apply\s*$", line)
        if m and not m.group(1).startswith("_"):
            out.append(m.group(1))
    return out


def main():
    doc = {}
    for mod, rel in MODULES.items():
        path = os.path.join(REF, rel)
        names = names_pyx(path) if rel.endswith(".pyx") else names_py(path)
        hot = names if HOT[mod] is None else [n for n in names if n in HOT[mod]]
        missing = [n for n in (HOT[mod] or []) if n not in names]
        assert not missing, (mod, missing)
        doc[mod] = dict(source="taiyaki/" + rel, public=sorted(names), hot_path=sorted(hot))
    with open(os.path.join(HERE, "reference_names.json"), "w") as fh:
        json.dump(doc, fh, indent=1, sort_keys=True)
    for mod, d in doc.items():
        print(mod, len(d["public"]), "public,", len(d["hot_path"]), "on the path")


if __name__ == "__main__":
    main()
