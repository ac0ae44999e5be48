#!/usr/bin/env python
"""Generate the ctypes binding of include/smrt_dort.h -- struct smrt_batch and every entry point -- from the header
itself.  The block between the `<!-- stub:begin -->` / `<!-- stub:end -->` markers of INTEGRATION.md is this output
(`python tools/gen_ctypes_stub.py --update` rewrites it; tests/test_host_logic.py checks that it is current and that
the declaration matches the compiled library through smrt_dort_abi).

    python tools/gen_ctypes_stub.py            # print the stub
    python tools/gen_ctypes_stub.py --update   # refresh INTEGRATION.md
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "smrt_dort.h")
DOC = os.path.join(ROOT, "INTEGRATION.md")
BEGIN, END = "<!-- stub:begin -->", "<!-- stub:end -->"

SCALARS = {"int32_t": "C.c_int32", "int64_t": "C.c_int64", "double": "C.c_double", "char": "C.c_char"}


def ctype_of(decl):
    """ctypes spelling of a C parameter / field type such as `const double*`, `smrt_dort_ctx**`, `int32_t`."""
    t = decl.replace("const", " ").strip()
    stars = t.count("*")
    base = t.replace("*", " ").split()[0]
    if base in ("smrt_dort_ctx", "void"):
        return "C.c_void_p" if stars <= 1 else "C.POINTER(C.c_void_p)"
    if base == "char" and stars == 1:
        return "C.c_char_p"
    if base == "smrt_batch":
        return "C.POINTER(SmrtBatch)"
    if base == "smrt_gather_op":   # {int32 peer, int32 reserved, int64 offset_rows, int64 rows}: passed as an opaque array
        return "C.c_void_p"
    c = SCALARS[base]
    for _ in range(stars):
        c = "C.POINTER(%s)" % c
    return c


def parse(header_text):
    text = re.sub(r"/\*.*?\*/", "", header_text, flags=re.S)
    body = text[text.index("typedef struct smrt_batch {") + len("typedef struct smrt_batch {"):text.index("} smrt_batch;")]
    fields = []
    for stmt in body.split(";"):
        stmt = " ".join(stmt.split())
        if stmt:
            m = re.match(r"(.+?)\s*(\w+)$", stmt)
            fields.append((m.group(2), ctype_of(m.group(1))))
    functions = []
    for m in re.finditer(r"^([\w\s\*]+?)\b(smrt_\w+)\s*\(([^)]*)\)\s*;", text[text.index("} smrt_batch;"):], flags=re.M):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.+?)\s*(\w+)(\[\w*\])?$", a)
                argtypes.append(ctype_of(mm.group(1) + ("*" if mm.group(3) else "")))
        functions.append((name, "None" if ret == "void" else ctype_of(ret), argtypes))
    return fields, functions


def stub():
    fields, functions = parse(open(HEADER).read())
    out = ["import ctypes as C", "", "", "class SmrtBatch(C.Structure):   # struct smrt_batch, field for field",
           "    _fields_ = ["]
    out += ['        ("%s", %s),' % f for f in fields]
    out += ["    ]", "", "", 'lib = C.CDLL("libsmrt_dort.so")']
    for name, ret, argtypes in functions:
        out.append("lib.%s.argtypes = [%s]" % (name, ", ".join(argtypes)))
        out.append("lib.%s.restype = %s" % (name, ret))
    out += ["",
            "# the declaration above against the library as compiled: sizeof(smrt_batch), then every field offset",
            "n = lib.smrt_dort_abi(None, 0)",
            "layout = (C.c_int32 * n)()",
            "lib.smrt_dort_abi(layout, n)",
            "assert list(layout) == [C.sizeof(SmrtBatch)] + [getattr(SmrtBatch, f).offset for f, _ in SmrtBatch._fields_]"]
    return "\n".join(out)


def main():
    text = stub()
    if "--update" not in sys.argv:
        print(text)
        return
    doc = open(DOC).read()
    a, b = doc.index(BEGIN) + len(BEGIN), doc.index(END)
    open(DOC, "w").write(doc[:a] + "\n```python\n" + text + "\n```\n" + doc[b:])
    print("updated", os.path.relpath(DOC))


if __name__ == "__main__":
    main()
