#!/usr/bin/env python3
"""The cheapest compile check available without a Go toolchain.

  python tests/tools/go_symbols.py            regenerates tests/golden/go_reference_symbols.json from /root/reference

Two halves, both used by tests/test_go_shim_symbols.py:

* `reference_symbols(import_path)`: the EXPORTED package-level names of a Go package of the reference tree (func / type / var /
  const; for funcs the parameter count and "variadic"), read with a line-level scanner from the package's non-test sources.
  The fixture holds them for every reference package integration/go/gpubinpacking imports: /root/reference does not travel, the
  fixture does; where the reference is present the test re-derives the fixture and compares.
* `scan_go_package(dir)`: what the shim's own files use — `alias.Name` selectors per imported package (with call arity),
  `C.name` references, package-level definitions and bare calls.

Not a compiler: no types, no method sets.  It catches what broke the shim before: a call into an unexported function of another
package (`observeBinpackingHeterogeneity`), a name that does not exist, a wrong argument count, a `C.` symbol the header lacks."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/cluster-autoscaler"
SHIM = os.path.join(ROOT, "integration", "go", "gpubinpacking")
FIXTURE = os.path.join(ROOT, "tests", "golden", "go_reference_symbols.json")


def package_dir(import_path):
    """import path -> directory under the reference tree (module k8s.io/autoscaler/cluster-autoscaler, everything else vendored)."""
    mod = "k8s.io/autoscaler/cluster-autoscaler"
    if import_path == mod or import_path.startswith(mod + "/"):
        return os.path.join(REF, import_path[len(mod):].lstrip("/"))
    return os.path.join(REF, "vendor", import_path)


def strip_go(src):
    """Comments and string / rune literals blanked out (lengths kept, newlines kept)."""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith("//", i):
            j = src.find("\n", i)
            j = n if j < 0 else j
            out.append(" " * (j - i)); i = j
        elif src.startswith("/*", i):
            j = src.find("*/", i + 2)
            j = n if j < 0 else j + 2
            out.append("".join(ch if ch == "\n" else " " for ch in src[i:j])); i = j
        elif c == '"':
            j = i + 1
            while j < n and src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            out.append('"' + " " * (j - i - 1) + '"'); i = j + 1
        elif c == "`":
            j = src.find("`", i + 1)
            j = n if j < 0 else j
            out.append("`" + "".join(ch if ch == "\n" else " " for ch in src[i + 1:j]) + "`"); i = j + 1
        elif c == "'":
            j = i + 1
            while j < n and src[j] != "'":
                j += 2 if src[j] == "\\" else 1
            out.append("'" + " " * (j - i - 1) + "'"); i = j + 1
        else:
            out.append(c); i += 1
    return "".join(out)


def _matching(src, i, open_ch="(", close_ch=")"):
    depth = 0
    for j in range(i, len(src)):
        if src[j] == open_ch:
            depth += 1
        elif src[j] == close_ch:
            depth -= 1
            if depth == 0:
                return j
    return -1


def split_top(s):
    """comma-separated items at nesting depth 0 of ( [ {"""
    items, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            items.append("".join(cur)); cur = []
        else:
            cur.append(ch)
    tail = "".join(cur)
    if tail.strip() or items:
        items.append(tail)
    return [x for x in (y.strip() for y in items)]


def reference_symbols(import_path):
    d = package_dir(import_path)
    syms = {}
    if not os.path.isdir(d):
        return None
    for fn in sorted(os.listdir(d)):
        if not fn.endswith(".go") or fn.endswith("_test.go"):
            continue
        src = strip_go(open(os.path.join(d, fn), encoding="utf-8", errors="replace").read())
        for m in re.finditer(r"^func ([A-Za-z_]\w*)\s*(\[[^\]]*\])?\(", src, re.M):      # package-level functions (not methods)
            name = m.group(1)
            lp = src.index("(", m.end() - 1)
            rp = _matching(src, lp)
            params = [p for p in split_top(src[lp + 1:rp]) if p]
            syms[name] = {"kind": "func", "params": len(params), "variadic": bool(params) and "..." in params[-1]}
        for m in re.finditer(r"^type ([A-Za-z_]\w*)\b", src, re.M):
            syms.setdefault(m.group(1), {"kind": "type"})
        for kw in ("var", "const"):
            for m in re.finditer(r"^%s ([A-Za-z_]\w*)\b" % kw, src, re.M):
                syms.setdefault(m.group(1), {"kind": kw})
        for m in re.finditer(r"^(var|const|type) \(", src, re.M):                         # grouped declarations
            rp = _matching(src, m.end() - 1)
            for line in src[m.end():rp].split("\n"):
                mm = re.match(r"^\t([A-Za-z_]\w*(?:\s*,\s*[A-Za-z_]\w*)*)\b", line)
                if mm:
                    for nm in re.split(r"\s*,\s*", mm.group(1)):
                        syms.setdefault(nm, {"kind": m.group(1)})
    return {k: v for k, v in sorted(syms.items()) if k[0].isupper()}


GO_BUILTINS = {"append", "cap", "clear", "close", "complex", "copy", "delete", "imag", "len", "make", "max", "min", "new", "panic", "print",
               "println", "real", "recover", "string", "int", "int8", "int16", "int32", "int64", "uint", "uint8", "uint16", "uint32", "uint64",
               "uintptr", "float32", "float64", "bool", "byte", "rune", "error", "any"}
GO_KEYWORDS = {"func", "if", "for", "switch", "return", "go", "defer", "select", "range", "case", "else", "map", "chan", "struct", "interface", "type", "var", "const"}


def scan_go_package(d=SHIM):
    """-> dict(imports {file: {alias: path}}, selectors [(file, line, alias, name, n_args or None)], c_refs [(file, line, name, n_args or None)],
    defs set, bare_calls [(file, line, name)])"""
    out = {"imports": {}, "selectors": [], "c_refs": [], "defs": set(), "bare_calls": [], "locals": {}}
    for fn in sorted(os.listdir(d)):
        if not fn.endswith(".go"):
            continue
        raw = open(os.path.join(d, fn), encoding="utf-8").read()
        # (the cgo preamble is a comment directly above `import "C"`: blanked with the other comments)
        src = strip_go(raw)
        imports = {}
        for m in re.finditer(r'^import \(\n(.*?)^\)', raw, re.M | re.S):
            for line in m.group(1).split("\n"):
                mm = re.match(r'^\s*(?:([A-Za-z_]\w*)\s+)?"([^"]+)"', line)
                if mm:
                    imports[mm.group(1) or mm.group(2).rsplit("/", 1)[-1]] = mm.group(2)
        for m in re.finditer(r'^import (?:([A-Za-z_]\w*)\s+)?"([^"]+)"', raw, re.M):
            imports[m.group(1) or m.group(2).rsplit("/", 1)[-1]] = m.group(2)
        out["imports"][fn] = imports
        line_of = lambda pos: src.count("\n", 0, pos) + 1
        # package-level definitions
        for m in re.finditer(r"^func (?:\([^)]*\)\s*)?([A-Za-z_]\w*)\s*\(", src, re.M):
            out["defs"].add(m.group(1))
        for m in re.finditer(r"^(?:type|var|const) ([A-Za-z_]\w*)\b", src, re.M):
            out["defs"].add(m.group(1))
        # methods declared by an interface type of the package (`Name(args) result` lines inside `type X interface { ... }`)
        for m in re.finditer(r"^type [A-Za-z_]\w* interface \{\n(.*?)^\}", src, re.M | re.S):
            for mm in re.finditer(r"^\s+([A-Za-z_]\w*)\(", m.group(1), re.M):
                out["defs"].add(mm.group(1))
        # names that hold function values inside function bodies: `name := func`, `name = func`, parameters `name func(`
        loc = set(re.findall(r"\b([A-Za-z_]\w*)\s*:?=\s*func\b", src)) | set(re.findall(r"\b([A-Za-z_]\w*)\s+func\(", src))
        out["locals"][fn] = loc
        # selectors alias.Name (alias not preceded by '.' or an identifier character)
        for m in re.finditer(r"(?<![\w.])([A-Za-z_]\w*)\.([A-Za-z_]\w*)", src):
            alias, name = m.group(1), m.group(2)
            n_args = None
            k = m.end()
            while k < len(src) and src[k] in " \t":
                k += 1
            if k < len(src) and src[k] == "(":
                rp = _matching(src, k)
                n_args = len([a for a in split_top(src[k + 1:rp]) if a]) if rp > 0 else None
            if alias == "C":
                out["c_refs"].append((fn, line_of(m.start()), name, n_args))
            elif alias in imports:
                out["selectors"].append((fn, line_of(m.start()), alias, name, n_args))
        # bare calls name( — not a selector, not a definition, not a keyword
        for m in re.finditer(r"(?<![\w.])([A-Za-z_]\w*)\s*\(", src):
            name = m.group(1)
            if name in GO_KEYWORDS:
                continue
            before = src[max(0, m.start() - 6):m.start()]
            if re.search(r"func\s+$", before) or re.search(r"\)\s*$", src[max(0, m.start() - 3):m.start()]) and src.rfind("func (", 0, m.start()) > src.rfind("\n", 0, m.start()):
                continue   # the name of a function / method being declared
            out["bare_calls"].append((fn, line_of(m.start()), name))
    return out


def shim_reference_packages():
    pk = set()
    for imports in scan_go_package()["imports"].values():
        for path in imports.values():
            if path != "C" and os.path.isdir(package_dir(path)):
                pk.add(path)
    return sorted(pk)


def main():
    if not os.path.isdir(REF):
        raise SystemExit("the reference tree is not here: the committed fixture stays as it is")
    data = {"_generated_by": "tests/tools/go_symbols.py from /root/reference/cluster-autoscaler (exported package-level names only)"}
    for path in shim_reference_packages():
        data[path] = reference_symbols(path)
    with open(FIXTURE, "w") as f:
        json.dump(data, f, indent=0, sort_keys=True)
        f.write("\n")
    print(f"{FIXTURE}: {len(data) - 1} packages, {sum(len(v) for k, v in data.items() if not k.startswith('_'))} exported names")


if __name__ == "__main__":
    sys.exit(main())
