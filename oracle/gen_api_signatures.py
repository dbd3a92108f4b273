"""TEST INFRASTRUCTURE (container only: reads /root/reference).  Writes tests/golden/api_signatures.json:
for every public function / method of this package that mirrors one of the reference (the table `PUBLIC`), the
reference's parameter names in order, which of them are keyword-only, and their defaults
where the default is a literal.  Data about an interface, no source text.  tests/test_api_signatures.py holds the
package to it: a call written for kikuchipy - positional or by keyword - must bind the same way here.

    python oracle/gen_api_signatures.py
"""
import ast
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/kikuchipy"


def collect(root):
    out = {}
    for p in sorted(glob.glob(root + "/**/*.py", recursive=True)):
        if "/tests/" in p:
            continue
        t = ast.parse(open(p).read())
        for n in t.body:
            if isinstance(n, ast.FunctionDef) and not n.name.startswith("_"):
                out.setdefault(n.name, (os.path.relpath(p, root), n))
            if isinstance(n, ast.ClassDef):
                for m in n.body:
                    if isinstance(m, ast.FunctionDef) and (not m.name.startswith("_") or m.name == "__init__"):
                        out.setdefault(n.name + "." + m.name, (os.path.relpath(p, root), m))
    return out


def literal(node):
    try:
        return {"value": ast.literal_eval(node)}
    except Exception:
        return None  # (an expression: not compared)


def describe(fn):
    a = fn.args
    pos = [x.arg for x in a.posonlyargs + a.args]
    defaults = {}
    for name, d in zip(pos[len(pos) - len(a.defaults):], a.defaults):
        lit = literal(d)
        if lit is not None:
            defaults[name] = lit["value"]
    for x, d in zip(a.kwonlyargs, a.kw_defaults):
        if d is not None:
            lit = literal(d)
            if lit is not None:
                defaults[x.arg] = lit["value"]
    return {"positional": pos, "keyword_only": [x.arg for x in a.kwonlyargs], "defaults": defaults,
            "var_positional": a.vararg is not None, "var_keyword": a.kwarg is not None}


# reference name -> where this package exposes its counterpart (dotted path under `kikuchipy_amd`)
PUBLIC = {
    "EBSD.dictionary_indexing": "EBSD.dictionary_indexing",
    "EBSD.remove_static_background": "EBSD.remove_static_background",
    "EBSD.remove_dynamic_background": "EBSD.remove_dynamic_background",
    "EBSD.refine_orientation": "EBSD.refine_orientation",
    "EBSD.refine_projection_center": "EBSD.refine_projection_center",
    "EBSD.refine_orientation_projection_center": "EBSD.refine_orientation_projection_center",
    "EBSDMasterPattern.get_patterns": "EBSDMasterPattern.get_patterns",
    "EBSDDetector.__init__": "EBSDDetector.__init__",
    "SimilarityMetric.__init__": "indexing.similarity_metrics.SimilarityMetric.__init__",
    "load": "load",
    "orientation_similarity_map": "indexing.orientation_similarity_map",
    "merge_crystal_maps": "indexing.merge_crystal_maps",
    "compute_refine_orientation_results": "indexing.compute_refine_orientation_results",
    "compute_refine_projection_center_results": "indexing.compute_refine_projection_center_results",
    "compute_refine_orientation_projection_center_results": "indexing.compute_refine_orientation_projection_center_results",
    "distance_to_origin": "filters.distance_to_origin",
}
# Not in the table on purpose: `kikuchipy_amd.remove_*_background(patterns, ...)` are this package's functions over a
# STACK of patterns (the reference's `kikuchipy.pattern.remove_dynamic_background` takes one pattern and a `dtype_out`;
# SURVEY.md 8 a-pre2 is the signal method); `_refinement.rotation_from_euler` is a private helper that shares a name.


def main():
    ref = collect(REF)
    table = {}
    for name, ours in PUBLIC.items():
        path, fn = ref[name]
        table[name] = dict(describe(fn), reference_file=path, line=fn.lineno, ours=ours)
    with open(os.path.join(ROOT, "tests", "golden", "api_signatures.json"), "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
    print(len(table), "signatures:", ", ".join(table))


if __name__ == "__main__":
    main()
