"""Minimal stand-in for the `plyfile` package, covering exactly what the reference's Gaussian model and dataset
readers use ("import enablers"):

    PlyElement.describe(structured_array, "vertex")                 scene/gaussian_model.py:444, dataset_readers.py:176
    PlyData([el]).write(path)                                       scene/gaussian_model.py:445, dataset_readers.py:177-178
    PlyData.read(path).elements[0]["x"], .elements[0].properties    scene/gaussian_model.py:456-505
    PlyData.read(path)["vertex"]["red"]                             scene/dataset_readers.py:156-160

`plyfile` is a third-party dependency of the reference (requirements.txt, unpinned) that is not in this image.  The
file format is the public PLY 1.0 format: a text header (`ply`, `format`, `element <name> <count>`,
`property <type> <name>` ...) followed by the element tables.  Files written here are `binary_little_endian 1.0`, the
same bytes `plyfile` writes for a single scalar-property element, so checkpoints interchange with the reference.
Only scalar properties are supported (3DGS point clouds have no list properties); a list property raises.
`gaussianeditor_amd.install()` registers this module as `plyfile` only when the real package is not importable.
"""
from __future__ import annotations

import io
from typing import Dict, Iterable, List, Union

import numpy as np

__all__ = ["PlyData", "PlyElement", "PlyProperty", "PlyParseError"]

# PLY type name -> numpy type code (both the classic and the sized spellings are legal in headers)
_PLY2NP = {
    "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1",
    "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
    "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
    "float": "f4", "float32": "f4", "double": "f8", "float64": "f8",
}
_NP2PLY = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float",
           "f8": "double"}
_BYTE_ORDER = {"binary_little_endian": "<", "binary_big_endian": ">"}


class PlyParseError(Exception):
    pass


class PlyProperty:
    """One scalar column of an element: `.name`, `.val_dtype` (numpy type code without byte order)."""

    def __init__(self, name: str, val_dtype: str):
        self.name = str(name)
        self.val_dtype = np.dtype(val_dtype).str[1:]

    def __repr__(self):
        return f"PlyProperty({self.name!r}, {self.val_dtype!r})"

    def __str__(self):
        return f"property {_NP2PLY[self.val_dtype]} {self.name}"


class PlyElement:
    """A named table: `.name`, `.data` (1-D structured array), `.properties`, `el[column]`."""

    def __init__(self, name: str, properties: List[PlyProperty], data: np.ndarray):
        self.name = str(name)
        self.properties = tuple(properties)
        self.data = data

    @staticmethod
    def describe(data: np.ndarray, name: str) -> "PlyElement":
        if not isinstance(data, np.ndarray) or data.ndim != 1 or data.dtype.names is None:
            raise TypeError("PlyElement.describe needs a one-dimensional structured numpy array")
        props = []
        for col in data.dtype.names:
            dt = data.dtype.fields[col][0]
            if dt.shape != () or dt.str[1:] not in _NP2PLY:
                raise ValueError(f"column {col!r}: only scalar int/uint/float properties are supported, got {dt}")
            props.append(PlyProperty(col, dt.str[1:]))
        return PlyElement(name, props, data)

    @property
    def count(self) -> int:
        return int(self.data.shape[0])

    def __len__(self):
        return self.count

    def __getitem__(self, key):
        return self.data[key]

    def __contains__(self, key):
        return key in (self.data.dtype.names or ())

    def dtype(self, byte_order: str = "<") -> np.dtype:
        return np.dtype([(p.name, byte_order + p.val_dtype) for p in self.properties])

    def header(self) -> str:
        return "\n".join([f"element {self.name} {self.count}"] + [str(p) for p in self.properties])


class PlyData:
    """A PLY file: `.elements`, `ply[name]`, `PlyData.read(path_or_stream)`, `.write(path_or_stream)`."""

    def __init__(self, elements: Iterable[PlyElement] = (), text: bool = False, byte_order: str = "<",
                 comments: Iterable[str] = ()):
        self.elements = list(elements)
        self.text = bool(text)
        self.byte_order = byte_order
        self.comments = list(comments)

    def __getitem__(self, name: str) -> PlyElement:
        for el in self.elements:
            if el.name == name:
                return el
        raise KeyError(name)

    def __contains__(self, name: str) -> bool:
        return any(el.name == name for el in self.elements)

    def __len__(self):
        return len(self.elements)

    def __iter__(self):
        return iter(self.elements)

    # -- reading --------------------------------------------------------------------------------------------
    @staticmethod
    def _parse_header(stream) -> tuple:
        line = stream.readline()
        if line.strip() != b"ply":
            raise PlyParseError("not a PLY file (missing 'ply' magic)")
        fmt = None
        comments: List[str] = []
        decl: List[Dict] = []
        while True:
            raw = stream.readline()
            if not raw:
                raise PlyParseError("unexpected end of file in PLY header")
            try:
                tok = raw.decode("ascii").strip().split()
            except UnicodeDecodeError as e:
                raise PlyParseError("non-ASCII bytes in PLY header") from e
            if not tok:
                continue
            if tok[0] == "end_header":
                break
            if tok[0] == "format":
                if len(tok) != 3 or tok[2] != "1.0" or tok[1] not in ("ascii", *_BYTE_ORDER):
                    raise PlyParseError(f"unsupported format line: {' '.join(tok)}")
                fmt = tok[1]
            elif tok[0] in ("comment", "obj_info"):
                comments.append(raw.decode("ascii").strip()[len(tok[0]) + 1:])
            elif tok[0] == "element":
                if len(tok) != 3:
                    raise PlyParseError(f"malformed element line: {' '.join(tok)}")
                decl.append({"name": tok[1], "count": int(tok[2]), "props": []})
            elif tok[0] == "property":
                if not decl:
                    raise PlyParseError("property before any element")
                if tok[1] == "list":
                    raise PlyParseError(f"list property {tok[-1]!r} is not supported by this PLY reader")
                if len(tok) != 3 or tok[1] not in _PLY2NP:
                    raise PlyParseError(f"malformed property line: {' '.join(tok)}")
                decl[-1]["props"].append(PlyProperty(tok[2], _PLY2NP[tok[1]]))
            else:
                raise PlyParseError(f"unknown header keyword {tok[0]!r}")
        if fmt is None:
            raise PlyParseError("PLY header has no format line")
        return fmt, comments, decl

    @classmethod
    def read(cls, stream: Union[str, "io.IOBase"]) -> "PlyData":
        own = isinstance(stream, (str, bytes)) or hasattr(stream, "__fspath__")
        f = open(stream, "rb") if own else stream
        try:
            fmt, comments, decl = cls._parse_header(f)
            elements = []
            for d in decl:
                el = PlyElement(d["name"], d["props"], None)
                if fmt == "ascii":
                    dt = el.dtype("=")
                    data = np.empty(d["count"], dtype=dt)
                    for r in range(d["count"]):
                        tok = f.readline().split()
                        if len(tok) != len(d["props"]):
                            raise PlyParseError(f"element {d['name']} row {r}: expected {len(d['props'])} values")
                        data[r] = tuple(np.dtype(p.val_dtype).type(float(t) if p.val_dtype[0] == "f" else int(t))
                                        for p, t in zip(d["props"], tok))
                else:
                    dt = el.dtype(_BYTE_ORDER[fmt])
                    nbytes = dt.itemsize * d["count"]
                    buf = f.read(nbytes)
                    if len(buf) != nbytes:
                        raise PlyParseError(f"element {d['name']}: file is truncated ({len(buf)} of {nbytes} bytes)")
                    data = np.frombuffer(buf, dtype=dt, count=d["count"]).copy()
                el.data = data
                elements.append(el)
            return cls(elements, text=(fmt == "ascii"), byte_order=_BYTE_ORDER.get(fmt, "="), comments=comments)
        finally:
            if own:
                f.close()

    # -- writing --------------------------------------------------------------------------------------------
    def header(self) -> str:
        fmt = "ascii" if self.text else ("binary_little_endian" if self.byte_order in ("<", "=") else "binary_big_endian")
        lines = ["ply", f"format {fmt} 1.0"] + [f"comment {c}" for c in self.comments]
        lines += [el.header() for el in self.elements] + ["end_header"]
        return "\n".join(lines)

    def write(self, stream: Union[str, "io.IOBase"]) -> None:
        own = isinstance(stream, (str, bytes)) or hasattr(stream, "__fspath__")
        f = open(stream, "wb") if own else stream
        try:
            f.write((self.header() + "\n").encode("ascii"))
            order = "<" if self.byte_order in ("<", "=") else ">"
            for el in self.elements:
                if self.text:
                    for row in el.data:
                        f.write((" ".join(repr(v.item()) for v in row) + "\n").encode("ascii"))
                else:
                    f.write(np.ascontiguousarray(el.data.astype(el.dtype(order), copy=False)).tobytes())
        finally:
            if own:
                f.close()
