"""The drop-in boundary, checked against the reference's own header TEXT (no reference code runs).

1. Every class / struct / enum / member that simian-spacemonkey_amd/host/gluvv_compat.h declares is
   looked up in MetaVolume.h, TLUT.h, gluvv.h and gluvvPrimitive.h of the reference and must carry the
   same type / return type / parameter types / constness there (compat may declare a subset, never
   something the reference lacks); enumerators must match in order.
2. The adapter (HipVolumeRenderer.cpp) is compiled with -DSMK_USE_REFERENCE_HEADERS against the
   reference's real MetaVolume.h / TLUT.h / gluvvPrimitive.h / gluvv.h, the build INTEGRATION.md
   prescribes.  gluvv.h includes TFWindow.h and LTWidgetRen.h (GLUT/GLUI, not installable here): the
   test puts two one-line forward declarations of those classes in front of them, in a temporary
   directory; nothing else is substituted.

Both need /root/reference and are skipped where it does not exist (the GPU box).
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
HOST = os.path.join(ROOT, "simian-spacemonkey_amd", "host")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")

KEYWORDS = {"unsigned", "signed", "int", "char", "float", "double", "long", "short", "void", "bool", "const"}


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"^\s*#[^\n]*", " ", text, flags=re.M)
    return text


def _match_brace(text, i):
    depth = 0
    for j in range(i, len(text)):
        if text[j] == "{":
            depth += 1
        elif text[j] == "}":
            depth -= 1
            if depth == 0:
                return j
    raise ValueError("unbalanced braces")


def _flatten_body(body):
    """replace every nested {...} (inline function bodies) by ';'; nested enums are returned apart"""
    out, enums, i = [], {}, 0
    while i < len(body):
        if body[i] == "{":
            j = _match_brace(body, i)
            head = "".join(out)
            m = re.search(r"typedef\s+enum\s*$", head)
            if m:
                tail = re.match(r"\s*(\w+)\s*;", body[j + 1:])
                enums[tail.group(1)] = _enumerators(body[i + 1:j])
                out = [head[:m.start()]]
                i = j + 1 + tail.end()
                continue
            out.append(";")
            i = j + 1
        else:
            out.append(body[i])
            i += 1
    return "".join(out), enums


def _enumerators(body):
    return [re.sub(r"\s+", "", e) for e in body.split(",") if e.strip()]


def _norm_type(t):
    t = re.sub(r"\b(inline|static|explicit)\b", " ", t)
    t = re.sub(r"\s*([\*&])\s*", r"\1", t)
    return " ".join(t.split())


def _param_type(p):
    p = p.split("=")[0].strip()
    if not p:
        return ""
    arr = "".join(re.findall(r"\[[^\]]*\]", p))
    p = re.sub(r"\[[^\]]*\]", "", p)
    toks = re.findall(r"[\w:]+|[\*&]", p)
    words = [t for t in toks if t not in "*&"]
    if len(words) >= 2 and toks[-1] not in "*&" and toks[-1] not in KEYWORDS:
        toks = toks[:-1]  # the parameter's name
    return _norm_type(" ".join(toks)) + arr


def _members(body):
    """{name or signature: normalised declaration} of one class/struct body"""
    flat, enums = _flatten_body(body)
    flat = re.sub(r"\b(public|protected|private)\s*:", " ", flat)
    out = {}
    for st in flat.split(";"):
        st = " ".join(st.split())
        if not st:
            continue
        if "(" in st:
            m = re.match(r"^(?P<ret>.*?)(?P<name>~?\w+)\s*\((?P<params>.*)\)\s*(?P<const>const)?\s*(:.*)?$", st)
            assert m, st
            params = [p for p in (_param_type(p) for p in m.group("params").split(",")) if p and p != "void"]
            key = "%s(%s)%s" % (m.group("name"), ",".join(params), " const" if m.group("const") else "")
            out[key] = _norm_type(m.group("ret"))
        else:
            parts = [p.strip() for p in st.split(",")]
            m = re.match(r"^(?P<type>.*?)(?P<ptr>[\*&\s]*)(?P<name>\w+)(?P<arr>(\[\w*\])*)$", parts[0])
            assert m, st
            base = _norm_type(m.group("type"))
            for k, part in enumerate(parts):
                if k:
                    m = re.match(r"^(?P<ptr>[\*&\s]*)(?P<name>\w+)(?P<arr>(\[\w*\])*)$", part)
                    assert m, st
                out[m.group("name")] = base + m.group("ptr").replace(" ", "") + m.group("arr")
    return out, enums


def parse_header(path):
    text = _strip_comments(open(path, errors="replace").read())
    classes, enums = {}, {}
    for m in re.finditer(r"\b(class|struct)\s+(\w+)\s*(?::[^{;]*)?\{", text):
        end = _match_brace(text, m.end() - 1)
        members, nested = _members(text[m.end():end])
        classes[m.group(2)] = members
        for k, v in nested.items():
            enums["%s::%s" % (m.group(2), k)] = v
    for m in re.finditer(r"\btypedef\s+enum\s*\{", text):
        end = _match_brace(text, m.end() - 1)
        tail = re.match(r"\s*(\w+)\s*;", text[end + 1:])
        name = tail.group(1)
        if not any(name == k.split("::")[-1] for k in enums):
            enums[name] = _enumerators(text[m.end():end])
    return classes, enums


def _reference():
    classes, enums = {}, {}
    for h in ("MetaVolume.h", "TLUT.h", "gluvv.h", "gluvvPrimitive.h"):
        c, e = parse_header(os.path.join(REF, h))
        classes.update(c)
        enums.update(e)
    return classes, enums


def test_every_compat_declaration_exists_in_the_reference_with_the_same_signature():
    cc, ce = parse_header(os.path.join(HOST, "gluvv_compat.h"))
    rc, re_ = _reference()
    assert {"Volume", "MetaVolume", "TLUT", "gluvvPrimitive", "gluvvGlobal", "gluvvVolRen", "gluvvPert"} <= set(cc)
    problems = []
    for cls, members in cc.items():
        if cls not in rc:
            problems.append("class %s does not exist in the reference" % cls)
            continue
        for key, decl in members.items():
            if key not in rc[cls]:
                problems.append("%s::%s is not a member of the reference's %s" % (cls, key, cls))
            elif rc[cls][key] != decl:
                problems.append("%s::%s is '%s' here, '%s' in the reference" % (cls, key, decl, rc[cls][key]))
    for name, vals in ce.items():
        if name not in re_:
            problems.append("enum %s does not exist in the reference" % name)
        elif re_[name] != vals:
            problems.append("enum %s: %s here, %s in the reference" % (name, vals, re_[name]))
    assert not problems, "\n".join(problems)
    # the two signatures the round-1 adapter got wrong stay pinned
    assert rc["TLUT"]["scaleAlpha(float)"] == "void" and cc["TLUT"]["scaleAlpha(float)"] == "void"
    assert "~gluvvPrimitive()" in rc["gluvvPrimitive"] and rc["gluvvPrimitive"]["~gluvvPrimitive()"] == ""  # not virtual


def test_adapter_compiles_against_the_reference_headers(tmp_path):
    inc = tmp_path / "inc"
    inc.mkdir()
    # gluvv.h's quoted includes are looked up beside it first: give it a directory where the two
    # GLUT/GLUI-dependent widget headers are forward declarations and nothing else exists
    shutil.copy(os.path.join(REF, "gluvv.h"), inc / "gluvv.h")
    (inc / "TFWindow.h").write_text("class TFWindow;\n")
    (inc / "LTWidgetRen.h").write_text("class LTWidgetRen;\n")
    obj = tmp_path / "HipVolumeRenderer.o"
    cmd = ["g++", "-std=c++17", "-Wall", "-Wno-comment", "-DSMK_USE_REFERENCE_HEADERS", "-I" + str(inc), "-I" + REF,
           "-I" + os.path.join(ROOT, "include"), "-I" + HOST, "-c", os.path.join(HOST, "HipVolumeRenderer.cpp"), "-o", str(obj)]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    # what it leaves for the reference's own objects to provide: gluvvPrimitive's defaults and TLUT's ctor/dtor
    syms = subprocess.run(["nm", "-C", "--undefined-only", str(obj)], capture_output=True, text=True).stdout
    assert "TLUT::TLUT(int)" in syms and "gluvvPrimitive::" in syms
    assert "TLUT::scaleAlpha" not in syms and "loadTransferTableRGBA" not in syms.replace("HipVolumeRenderer::loadTransferTableRGBA", "")


def test_hip_volume_renderer_has_the_inner_interface_of_volume_renderer():
    """what VolumeRenderable needs from VolumeRenderer (VolumeRenderer.h:86-123).  Every public method of the
    reference class that is not GL/NRRD plumbing must exist in HipVolumeRenderer with the same return type and parameter
    types (GLdouble is double, GL/gl.h)."""
    rc, _ = parse_header(os.path.join(REF, "VolumeRenderer.h"))
    hc, _ = parse_header(os.path.join(HOST, "HipVolumeRenderer.h"))
    ref, mine = rc["VolumeRenderer"], hc["HipVolumeRenderer"]
    gl = lambda k: k.replace("GLdouble", "double")   # noqa: E731
    ref = {gl(k): v for k, v in ref.items()}
    wanted = ["createVolume(int,Volume*)", "createVolume(int,Volume*,int)", "createTLUT()", "getColorMap()",
              "renderVolume(float,double[16])", "renderVolume(float,double[16],float[2],float[2],float[2])",
              "renderSlice(float[4][3],float)", "useBBox(int)", "useBBoxBrackets(int)"]
    problems = []
    for key in wanted:
        if key not in ref:
            problems.append("the reference's VolumeRenderer has no %s (the test's list is stale)" % key)
        elif key not in mine:
            problems.append("HipVolumeRenderer lacks %s" % key)
        elif mine[key] != ref[key]:
            problems.append("%s returns '%s' here, '%s' in the reference" % (key, mine[key], ref[key]))
    assert not problems, "\n".join(problems)
