"""PLY ingest / output for the clouds either side of the ICP path (utilities/ply_io.hpp:43-143,
utilities/point_cloud.hpp:501-541): vertex x y z [nx ny nz] [red green blue], ascii or binary_little_endian.

    cloud = read_ply(path)          # dict(points (N,3) f32, normals (N,3) f32 | None, colors (N,3) f32 in [0,1] | None)
    write_ply(path, points, normals=None, colors=None, binary=True)
"""
import numpy as np

_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1", "char": "i1",
          "int8": "i1", "ushort": "<u2", "uint16": "<u2", "short": "<i2", "int16": "<i2", "uint": "<u4", "uint32": "<u4",
          "int": "<i4", "int32": "<i4"}


def read_ply(path):
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elems = None, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "end_header":
                break
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elems.append((tok[1], int(tok[2]), []))
            elif tok[0] == "property":
                if tok[1] == "list":
                    elems[-1][2].append((tok[4], None))          # list properties are not readable as a table
                else:
                    elems[-1][2].append((tok[2], tok[1]))
        if fmt not in ("ascii", "binary_little_endian"):
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
        out = {"points": np.zeros((0, 3), np.float32), "normals": None, "colors": None}
        import os

        data_bytes = os.fstat(f.fileno()).st_size - f.tell()
        for name, count, props in elems:
            # the counts are untrusted: a row takes at least one byte per property
            if count < 0 or count * max(1, len(props)) > data_bytes:
                raise ValueError(f"{path}: element count exceeds the file size")
            if any(t is None for _, t in props):
                if name == "vertex":
                    raise ValueError(f"{path}: list property in the vertex element")
                break                                              # faces etc. after the vertices: not needed
            dt = np.dtype([(n, _TYPES[t]) for n, t in props])
            if fmt == "ascii":
                rows = np.loadtxt(f, max_rows=count, ndmin=2) if count else np.zeros((0, len(props)))
                v = {n: rows[:, i] for i, (n, _) in enumerate(props)}
            else:
                v = np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)
            if name != "vertex":
                continue
            names = [n for n, _ in props]

            def cols(keys):
                return np.stack([np.asarray(v[k], np.float32) for k in keys], 1) if all(k in names for k in keys) else None
            out["points"] = cols(("x", "y", "z"))
            out["normals"] = cols(("nx", "ny", "nz"))
            c = cols(("red", "green", "blue"))
            out["colors"] = None if c is None else c * np.float32(1.0 / 255.0)
            break
        if out["points"] is None:
            raise ValueError(f"{path}: no vertex x/y/z properties")
        return out


def write_ply(path, points, normals=None, colors=None, binary=True):
    p = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
    n = len(p)
    fields, cols = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")], [p[:, 0], p[:, 1], p[:, 2]]
    if normals is not None:
        q = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
        fields += [("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4")]; cols += [q[:, 0], q[:, 1], q[:, 2]]
    if colors is not None:
        c = np.clip(np.asarray(colors, np.float32).reshape(-1, 3) * np.float32(255.0), 0, 255).astype(np.uint8)   # cast, as point_cloud.hpp:535
        fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]; cols += [c[:, 0], c[:, 1], c[:, 2]]
    names = {"<f4": "float", "u1": "uchar"}
    header = "ply\nformat %s 1.0\nelement vertex %d\n" % ("binary_little_endian" if binary else "ascii", n)
    header += "".join(f"property {names[t]} {k}\n" for k, t in fields) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        if binary:
            rec = np.zeros(n, dtype=np.dtype(fields))
            for (k, _), col in zip(fields, cols):
                rec[k] = col
            f.write(rec.tobytes())
        else:
            for i in range(n):
                f.write((" ".join(("%d" % col[i]) if t == "u1" else ("%.9g" % col[i]) for (_, t), col in zip(fields, cols)) + "\n").encode("ascii"))
