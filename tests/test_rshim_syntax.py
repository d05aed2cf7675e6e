"""The R .Call shim (rshim/bwgr_shim.c, rshim/bwgr_hip.R) is this repository's own code but cannot be built here (no R).  What CAN be checked:

* it is syntactically valid C against declarations-only stubs of the R API names it uses (tests/r_api_stub/: test infrastructure, pins nothing --
  the real compile happens where R's headers exist);
* every entry of its registration table names a defined function with that many SEXP parameters (what R_registerRoutines enforces at load time:
  the reference registers the same way, src/RcppExports.cpp:1152-1233), and every .Call in bwgr_hip.R passes that many arguments;
* it calls only C-ABI entry points that include/bwgr.h declares."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "rshim", "bwgr_shim.c")
RFILE = os.path.join(ROOT, "rshim", "bwgr_hip.R")


def test_shim_is_valid_c_against_the_r_api_declarations():
    r = subprocess.run(["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Wextra", "-Wno-cast-function-type", "-Werror=implicit-function-declaration",
                        "-I" + os.path.join(ROOT, "tests", "r_api_stub"), "-I" + os.path.join(ROOT, "include"), SHIM], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "warning" not in r.stderr, r.stderr


def _split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def test_registration_table_matches_the_definitions_and_the_r_calls():
    src = open(SHIM).read()
    defs = {m.group(1): len(_split_args(m.group(2))) for m in re.finditer(r"^SEXP\s+(bwgrhip_\w+)\s*\(([^)]*)\)\s*\{", src, re.M)}
    table = {m.group(1): int(m.group(3)) for m in re.finditer(r'\{"(bwgrhip_\w+)",\s*\(DL_FUNC\)\s*&(\w+),\s*(\d+)\}', src)}
    assert table and set(table) <= set(defs), (sorted(table), sorted(defs))
    for name, nargs in table.items():
        assert defs[name] == nargs, (name, defs[name], nargs)
    rsrc = open(RFILE).read()
    calls = []
    for m in re.finditer(r'\.Call\(\s*"(bwgrhip_\w+)"', rsrc):
        i, depth = rsrc.index("(", m.start()), 0
        for k in range(i, len(rsrc)):
            depth += rsrc[k] == "("; depth -= rsrc[k] == ")"
            if depth == 0:
                break
        args = [a for a in _split_args(rsrc[i + 1:k]) if not re.match(r"\s*PACKAGE\s*=", a)]
        calls.append((m.group(1), len(args) - 1))
    assert calls
    for name, nargs in calls:
        assert name in table and table[name] == nargs, (name, nargs, table.get(name))


def test_shim_calls_only_declared_entry_points():
    hdr = open(os.path.join(ROOT, "include", "bwgr.h")).read()
    declared = set(re.findall(r"\b(bwgr_\w+)\s*\(", hdr))
    used = set(re.findall(r"\b(bwgr_[a-z0-9_]+)\s*\(", open(SHIM).read()))
    assert used and used <= declared, sorted(used - declared)
