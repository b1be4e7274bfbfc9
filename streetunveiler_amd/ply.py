"""Gaussian checkpoint PLY in the reference's layout, numpy only.

Layout [REF /root/reference/scene/gaussian_model.py:226-259 save_ply, :338-382 load_ply]: one `vertex` element,
binary little-endian, float32 properties x y z nx ny nz f_dc_0..2 f_rest_0..44 opacity scale_0..1 rot_0..3 and an
int32 `semantics`.  f_dc / f_rest are stored CHANNEL-major (the [P,K,3] feature tensors transposed to [P,3,K] and
flattened), all values are the raw (pre-activation) parameters.  The reader takes the properties by name, so files
with a different property order or extra properties load too; ASCII PLY is accepted as well.
"""
from __future__ import annotations

import numpy as np

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
              "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
              "double": "f8", "float64": "f8"}


def attribute_names(n_dc: int = 3, n_rest: int = 45, n_scale: int = 2, n_rot: int = 4):
    """Float properties in file order [REF scene/gaussian_model.py:226-239]."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)] + [f"f_rest_{i}" for i in range(n_rest)] + ["opacity"]
    names += [f"scale_{i}" for i in range(n_scale)] + [f"rot_{i}" for i in range(n_rot)]
    return names


def save_ply(path, xyz, features_dc, features_rest, opacity, scaling, rotation, semantics) -> None:
    """xyz [P,3], features_dc [P,1,3], features_rest [P,K-1,3], opacity [P,1], scaling [P,2], rotation [P,4],
    semantics [P] or [P,1] (integer class ids)."""
    xyz = np.asarray(xyz, np.float32)
    P = xyz.shape[0]
    f_dc = np.asarray(features_dc, np.float32).reshape(P, -1, 3).transpose(0, 2, 1).reshape(P, -1)
    f_rest = np.asarray(features_rest, np.float32).reshape(P, -1, 3).transpose(0, 2, 1).reshape(P, -1)
    scaling = np.asarray(scaling, np.float32).reshape(P, -1)
    rotation = np.asarray(rotation, np.float32).reshape(P, -1)
    names = attribute_names(f_dc.shape[1], f_rest.shape[1], scaling.shape[1], rotation.shape[1])
    dtype = np.dtype([(n, "<f4") for n in names] + [("semantics", "<i4")])
    rows = np.empty(P, dtype=dtype)
    cols = np.concatenate([xyz, np.zeros_like(xyz), f_dc, f_rest, np.asarray(opacity, np.float32).reshape(P, 1), scaling, rotation], axis=1)
    assert cols.shape[1] == len(names)
    for k, n in enumerate(names):
        rows[n] = cols[:, k]
    rows["semantics"] = np.asarray(semantics).reshape(P).astype(np.int32)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {P}"]
    header += [f"property float {n}" for n in names] + ["property int semantics", "end_header"]
    with open(path, "wb") as fh:
        fh.write(("\n".join(header) + "\n").encode("ascii"))
        fh.write(rows.tobytes())


def _read_vertex_table(path):
    with open(path, "rb") as fh:
        if fh.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, props, count, in_vertex, elements_before = None, [], None, False, 0
        while True:
            line = fh.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    count = int(tok[2])
                elif count is None:
                    elements_before += 1
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list property in the vertex element")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if count is None or fmt is None:
            raise ValueError(f"{path}: no vertex element")
        if elements_before:
            raise ValueError(f"{path}: elements before `vertex` are not supported")
        if fmt == "ascii":
            flat = np.loadtxt(fh, dtype=np.float64, max_rows=count, ndmin=2)
            return {n: flat[:, k] for k, (n, _) in enumerate(props)}, count
        end = "<" if fmt == "binary_little_endian" else ">"
        dtype = np.dtype([(n, end + t) for n, t in props])
        table = np.frombuffer(fh.read(dtype.itemsize * count), dtype=dtype, count=count)
        return {n: table[n] for n, _ in props}, count


def load_ply(path, max_sh_degree: int = 3):
    """-> dict of float32 arrays in the model's layout: xyz [P,3], features_dc [P,1,3], features_rest [P,K-1,3],
    opacity [P,1], scaling [P,S], rotation [P,4], semantics [P] int32 [REF scene/gaussian_model.py:338-382]."""
    col, P = _read_vertex_table(path)
    f32 = lambda names: np.stack([np.asarray(col[n], np.float32) for n in names], axis=1) if names else np.zeros((P, 0), np.float32)
    numbered = lambda prefix: sorted((n for n in col if n.startswith(prefix)), key=lambda n: int(n.split("_")[-1]))
    rest_names = numbered("f_rest_")
    K = (max_sh_degree + 1) ** 2
    if len(rest_names) != 3 * K - 3:
        raise ValueError(f"{path}: {len(rest_names)} f_rest properties, SH degree {max_sh_degree} needs {3 * K - 3}")
    out = {
        "xyz": f32(["x", "y", "z"]),
        "features_dc": f32(["f_dc_0", "f_dc_1", "f_dc_2"]).reshape(P, 3, 1).transpose(0, 2, 1).copy(),
        "features_rest": f32(rest_names).reshape(P, 3, K - 1).transpose(0, 2, 1).copy(),
        "opacity": f32(["opacity"]),
        "scaling": f32(numbered("scale_")),
        "rotation": f32(numbered("rot")),
        "semantics": np.asarray(col["semantics"], np.int32) if "semantics" in col else np.zeros(P, np.int32),
    }
    return out
