"""CPU restatement of the reference's on-disk volume formats -- TEST INFRASTRUCTURE ONLY (the
product is simian-spacemonkey_amd/host/VolumeFiles.cpp; nothing there imports this).

    parse_trex   MetaVolume::parse           MetaVolume.cpp:233-627
    brick_file   readVol's file name         MetaVolume.cpp:756-760
    quantize     quantize<T> + affine        VectorMath.h:70-74, 1441-1552
    read_nrrd    readNrrd / parseNrrd        MetaVolume.cpp:1006-1105, 1518-1566

Pinned by the reference's own sample description TT.trex (copied as data to tests/golden/).
"""
import numpy as np

_TYPES = {"float": "f4", "double": "f8", "int": "i4", "uint": "u4", "short": "i2", "ushort": "u2", "uchar": "u1"}


def _trim(s):
    return s.rstrip(" \t\n\r").lstrip(" \t")


def _three(value, conv):
    parts = [_trim(p) for p in value.split(",", 2)]
    return [conv(p) for p in parts if p != ""]


def _atoi(s):                       # C atoi: leading integer, 0 when there is none
    import re
    m = re.match(r"\s*[-+]?\d+", s)
    return int(m.group(0)) if m else 0


def _atof(s):
    import re
    m = re.match(r"\s*[-+]?(\d+\.?\d*([eE][-+]?\d+)?|\.\d+([eE][-+]?\d+)?)", s)
    return float(np.float32(float(m.group(0)))) if m else 0.0


def parse_trex(text):
    h = dict(name="", native_name="", files="", tlut="", bane="", nrrd="", tsteps=[0, 0, 0], cache=0,
             isize=[0, 0, 0], fsize=[0.0, 0.0, 0.0], type="uchar", big_endian=False, append=True,
             bricks=[], displays=[], declared=0)
    lines = text.split("\n")
    i = 0
    subv = 0
    while i < len(lines):
        line = lines[i]
        i += 1
        cut = min([c for c in (line.find(":"), line.find("{")) if c >= 0], default=-1)
        key = _trim(line if cut < 0 else line[:cut])
        value = "" if cut < 0 else _trim(line[cut + 1:])
        if key == "":
            continue
        if key == "Data Type":
            for t in _TYPES:
                if value in (t, t.upper(), t.capitalize(), {"uint": "UInt", "ushort": "UShort"}.get(t, "")):
                    h["type"] = t
        elif key in ("Time Step Cache", "Time step cache", "time step cache"):
            h["cache"] = _atoi(value)
        elif key in ("ENDIAN", "Endian", "endian"):
            if value in ("BIG", "big", "Big"):
                h["big_endian"] = True
            if value in ("LITTLE", "little", "Little"):
                h["big_endian"] = False
        elif key == "Displays":
            h["displays"] = [_trim(p) for p in value.split(",") if _trim(p)]
        elif key == "Don't append numbers":
            h["append"] = False
        elif key == "Data Set Name":
            h["name"] = value
        elif key == "Native Data Set Name":
            h["native_name"] = value
        elif key == "Data Set Files":
            h["files"] = value
        elif key == "Number of Time Steps":
            v = _three(value, _atoi)
            h["tsteps"][:len(v)] = v
        elif key == "TLUT File":
            h["tlut"] = value
        elif key == "Bane File":
            h["bane"] = value
        elif key == "Nrrd File":
            h["nrrd"] = value
        elif key == "Volume Size int":
            v = _three(value, _atoi)
            h["isize"][:len(v)] = v
        elif key == "Volume Size float":
            v = _three(value, _atof)
            h["fsize"][:len(v)] = v
        elif key == "Number of Sub Volumes":
            h["declared"] = _atoi(value)
            h["bricks"] = [dict(isize=[0, 0, 0], fsize=[0.0] * 3, ipos=[0, 0, 0], fpos=[0.0] * 3)
                           for _ in range(h["declared"])]
        elif key == "SubVolume":
            if not h["bricks"]:
                raise ValueError("Number of subvolumes not known")
            b = h["bricks"][subv]
            while True:
                if i >= len(lines):
                    raise ValueError("SubVolume{ not closed")
                line = lines[i]
                i += 1
                if _trim(line) == "}":
                    break
                c = line.find(":")
                k2 = _trim(line if c < 0 else line[:c])
                v2 = "" if c < 0 else line[c + 1:]
                if k2 == "Size int":
                    b["isize"] = _three(v2, _atoi)
                elif k2 == "Size float":
                    b["fsize"] = _three(v2, _atof)
                elif k2 == "Pos int":
                    b["ipos"] = _three(v2, _atoi)
                elif k2 == "Pos float":
                    b["fpos"] = _three(v2, _atof)
            subv += 1
    return h


def brick_file(h, timestep, brick):
    return h["files"] + (".%04d.%02d" % (timestep, brick) if h["append"] else "")


def quantize(native):
    """(uchar) affine(min, x, max, 0, 255) in double, truncated; a constant volume -> 0"""
    a = np.asarray(native)
    if a.dtype == np.uint8:
        return a.copy()
    if a.dtype == np.float32:       # the reference starts max at -1e10 and min at +1e10
        lo = np.float32(min(np.float32(1e10), a.min()))
        hi = np.float32(max(np.float32(-1e10), a.max()))
    else:
        lo, hi = a.min(), a.max()
    lo, hi = float(lo), float(hi)
    if hi == lo:
        return np.zeros(a.shape, np.uint8)
    q = 255.0 * (a.astype(np.float64) - lo) / (hi - lo)
    return np.trunc(q).astype(np.uint8)


def read_raw_brick(h, path, brick):
    b = h["bricks"][brick]
    n = b["isize"][0] * b["isize"][1] * b["isize"][2]
    dt = np.dtype(_TYPES[h["type"]])
    if h["big_endian"]:
        dt = dt.newbyteorder(">")
    raw = np.fromfile(path, dtype=dt, count=n)
    assert raw.size == n
    return quantize(raw.astype(dt.newbyteorder("=")))


def read_nrrd(path):
    with open(path, "rb") as f:
        blob = f.read()
    head, _, body = blob.partition(b"\n\n")
    info = dict(dims=3, nelts=1, isize=[0, 0, 0], spacing=[1.0, 1.0, 1.0], type="uchar")
    for line in head.decode("latin-1").split("\n"):
        key, _, value = line.partition(":")
        if key == "dimension":
            info["dims"] = int(value)
        elif key == "sizes":
            v = [int(t) for t in value.split()]
            if info["dims"] == 4:
                info["nelts"], info["isize"] = v[0], v[1:4]
            else:
                info["isize"] = v[:3]
        elif key == "spacings":
            t = value.split()
            info["spacing"] = [float(np.float32(float(x))) for x in (t[1:4] if info["dims"] == 4 else t[:3])]
        elif key == "type":
            info["type"] = {"unsigned char": "uchar", "unsigned short": "ushort"}[value.strip()]
    n = info["nelts"] * info["isize"][0] * info["isize"][1] * info["isize"][2]
    if info["type"] == "ushort":
        data = quantize(np.frombuffer(body, np.uint16, n))
    else:
        data = np.frombuffer(body, np.uint8, n).copy()
    fs = np.array([np.float32(info["spacing"][a]) * np.float32(info["isize"][a]) for a in range(3)], np.float32)
    info["fsize"] = (fs / fs.max()).tolist()
    info["data"] = data
    return info
