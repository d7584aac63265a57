"""Reader / writer for the reference's on-disk depth format: an OpenCV FileStorage XML holding
cv::Mat_<float> matrices named "averaged_depth" and "depth" (main.cpp:112-114 writes it in capture mode,
main.cpp:146-149 reads it).  Only what that file needs: <name type_id="opencv-matrix"> with <rows>, <cols>,
<dt>f</dt> (also d, i, u, w, s, c) and whitespace-separated <data>.  No OpenCV involved."""
from __future__ import annotations

import re
from typing import Dict

import numpy as np

_DT = {"f": np.float32, "d": np.float64, "i": np.int32, "u": np.uint8, "c": np.int8, "w": np.uint16, "s": np.int16}
_SPECIAL = {".Inf": "inf", "+.Inf": "inf", "-.Inf": "-inf", ".Nan": "nan", "-.Nan": "nan", ".NaN": "nan"}
_MAT = re.compile(r"<(?P<name>[A-Za-z_][\w\-]*)\s+type_id=\"opencv-matrix\"\s*>(?P<body>.*?)</(?P=name)\s*>", re.S)


def _field(body: str, tag: str) -> str:
    m = re.search(rf"<{tag}>\s*(.*?)\s*</{tag}>", body, re.S)
    if not m:
        raise ValueError(f"opencv-matrix without <{tag}>")
    return m.group(1)


def read_matrices(path: str) -> Dict[str, np.ndarray]:
    """all opencv-matrix nodes of the file as {name: array [rows, cols(, channels)]}"""
    text = open(path, "r", encoding="utf-8", errors="replace").read()
    if "<opencv_storage>" not in text:
        raise ValueError(f"{path}: not an OpenCV FileStorage XML")
    out = {}
    for m in _MAT.finditer(text):
        body = m.group("body")
        rows, cols = int(_field(body, "rows")), int(_field(body, "cols"))
        dt = _field(body, "dt").strip().strip('"')
        ch = 1
        mm = re.fullmatch(r"(\d*)([a-z])", dt)
        if not mm or mm.group(2) not in _DT:
            raise ValueError(f"{path}: unsupported <dt>{dt}</dt>")
        if mm.group(1):
            ch = int(mm.group(1))
        toks = _field(body, "data").split()
        if mm.group(2) in "fd":
            vals = np.array([float(_SPECIAL.get(t, t)) for t in toks], dtype=np.float64)
        else:
            vals = np.array([int(t) for t in toks], dtype=np.int64)
        if vals.size != rows * cols * ch:
            raise ValueError(f"{path}: <{m.group('name')}> has {vals.size} values, expected {rows * cols * ch}")
        arr = vals.astype(_DT[mm.group(2)])
        out[m.group("name")] = arr.reshape((rows, cols) if ch == 1 else (rows, cols, ch))
    return out


def read_depth_xml(path: str):
    """(depth, averaged_depth) float32 [H,W] in millimetres, as main.cpp:146-149 reads them"""
    mats = read_matrices(path)
    for need in ("depth", "averaged_depth"):
        if need not in mats:
            raise ValueError(f"{path}: no <{need}> matrix")
    return np.ascontiguousarray(mats["depth"], np.float32), np.ascontiguousarray(mats["averaged_depth"], np.float32)


def _fmt(v: float) -> str:
    if v != v:
        return ".Nan"
    if v in (float("inf"), float("-inf")):
        return ".Inf" if v > 0 else "-.Inf"
    if float(v).is_integer() and abs(v) < 1e9:
        return f"{int(v)}."                      # OpenCV writes integral floats as "1234."
    return f"{v:.8e}"                            # 9 significant digits round-trip binary32


def write_matrices(path: str, mats: Dict[str, np.ndarray]) -> None:
    with open(path, "w", encoding="utf-8") as f:
        f.write('<?xml version="1.0"?>\n<opencv_storage>\n')
        for name, a in mats.items():
            a = np.asarray(a)
            if a.ndim != 2:
                raise ValueError("only single-channel matrices are written")
            code = {np.dtype(np.float32): "f", np.dtype(np.float64): "d", np.dtype(np.int32): "i", np.dtype(np.uint8): "u"}[a.dtype]
            f.write(f'<{name} type_id="opencv-matrix">\n  <rows>{a.shape[0]}</rows>\n  <cols>{a.shape[1]}</cols>\n  <dt>{code}</dt>\n  <data>\n')
            flat = a.reshape(-1)
            toks = [_fmt(float(v)) for v in flat] if code in "fd" else [str(int(v)) for v in flat]
            for i in range(0, len(toks), 8):
                f.write("    " + " ".join(toks[i:i + 8]) + "\n")
            f.write(f"  </data></{name}>\n")
        f.write("</opencv_storage>\n")


def write_depth_xml(path: str, depth: np.ndarray, averaged_depth: np.ndarray) -> None:
    """the file main.cpp's capture branch writes (main.cpp:112-114): averaged_depth first, then depth"""
    write_matrices(path, {"averaged_depth": np.asarray(averaged_depth, np.float32), "depth": np.asarray(depth, np.float32)})
