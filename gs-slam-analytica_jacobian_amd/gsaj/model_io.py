"""Parameter I/O of the Gaussian map in the reference's on-disk formats, without third-party readers and without
executing anything a file contains:

  * PLY  -- the layout GaussianModel.save_ply writes (reference gaussian_splatting/scene/gaussian_model.py:383-436: vertex
            element, float32 properties x y z nx ny nz f_dc_* f_rest_* opacity scale_* rot_*; plyfile's default encoding is
            binary little-endian) and load_ply reads (:453-542).  plyfile is not installed here: the 30-line header is parsed
            by hand, the body with numpy.frombuffer; ascii bodies are read too.
  * .pt  -- GaussianModel.load_tensors (:70-138) calls torch.jit.load and takes the module's parameters in registration order
            (xyz, f_dc, f_rest, opacity, scaling, rotation).  torch.jit.load deserialises and RUNS TorchScript code; here the
            archive is opened as the zip it is and its data.pkl is decoded by a restricted unpickler that can only build
            tensors from the archive's raw storages and plain containers -- module classes become inert attribute bags.
            Plain torch.save files (list / tuple / dict of tensors) go through torch.load(weights_only=True).
"""
import io
import pickle
import zipfile

import numpy as np
import torch

PLY_FIXED = ["x", "y", "z", "nx", "ny", "nz"]


def ply_attributes(n_dc, n_rest, n_scale, n_rot):
    """construct_list_of_attributes (gaussian_model.py:383-394)."""
    return (PLY_FIXED + ["f_dc_%d" % i for i in range(n_dc)] + ["f_rest_%d" % i for i in range(n_rest)] + ["opacity"]
            + ["scale_%d" % i for i in range(n_scale)] + ["rot_%d" % i for i in range(n_rot)])


def write_ply(path, xyz, f_dc, f_rest, opacity, scaling, rotation):
    """f_dc [P,1,3], f_rest [P,M-1,3] (the model's layout): stored channel-major like save_ply (transpose(1,2).flatten)."""
    xyz = np.asarray(xyz, np.float32)
    P = xyz.shape[0]
    dc = np.asarray(f_dc, np.float32).transpose(0, 2, 1).reshape(P, -1)
    rest = np.asarray(f_rest, np.float32).transpose(0, 2, 1).reshape(P, -1)
    cols = [xyz, np.zeros_like(xyz), dc, rest, np.asarray(opacity, np.float32).reshape(P, -1), np.asarray(scaling, np.float32).reshape(P, -1),
            np.asarray(rotation, np.float32).reshape(P, -1)]
    body = np.ascontiguousarray(np.concatenate(cols, axis=1).astype("<f4"))
    names = ply_attributes(dc.shape[1], rest.shape[1], cols[5].shape[1], cols[6].shape[1])
    assert body.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % P + "".join("property float %s\n" % n for n in names) + "end_header\n"
    with open(path, "wb") as fh:
        fh.write(header.encode("ascii"))
        fh.write(body.tobytes())


_PLY_TYPES = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "uchar": "u1", "uint8": "u1", "char": "i1", "int8": "i1",
              "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4"}


def read_ply_vertices(path):
    """-> dict name -> float64 column of the first (vertex) element.  Formats: binary_little_endian, binary_big_endian, ascii."""
    with open(path, "rb") as fh:
        raw = fh.read()
    end = raw.index(b"end_header")
    end = raw.index(b"\n", end) + 1
    lines = raw[:end].decode("ascii", "replace").split("\n")
    if lines[0].strip() != "ply":
        raise ValueError("%s is not a PLY file" % path)
    fmt, count, props, in_first = None, None, [], False
    for ln in lines[1:]:
        tok = ln.split()
        if not tok:
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            if count is None:
                count, in_first = int(tok[2]), True
            else:
                in_first = False
        elif tok[0] == "property" and in_first:
            if tok[1] == "list":
                raise ValueError("list properties are not part of the Gaussian-map layout")
            props.append((tok[2], _PLY_TYPES[tok[1]]))
    if fmt == "ascii":
        vals = np.loadtxt(io.BytesIO(raw[end:]), max_rows=count, ndmin=2)
        return {n: vals[:, i].astype(np.float64) for i, (n, _) in enumerate(props)}
    order = "<" if fmt == "binary_little_endian" else ">"
    dt = np.dtype([(n, order + t) for n, t in props])
    rec = np.frombuffer(raw, dtype=dt, count=count, offset=end)
    return {n: rec[n].astype(np.float64) for n, _ in props}


def read_gaussian_ply(path, max_sh_degree):
    """load_ply (gaussian_model.py:453-542) -> dict of float32 arrays in the model's layout: xyz [P,3], f_dc [P,1,3],
    f_rest [P,M-1,3], opacity [P,1], scaling [P,S], rotation [P,4], normals [P,3]."""
    v = read_ply_vertices(path)
    P = v["x"].shape[0]
    by_index = lambda prefix: sorted((k for k in v if k.startswith(prefix)), key=lambda s: int(s.split("_")[-1]))  # noqa: E731
    xyz = np.stack([v["x"], v["y"], v["z"]], axis=1)
    dc = np.stack([v["f_dc_0"], v["f_dc_1"], v["f_dc_2"]], axis=1).reshape(P, 3, 1)
    rest_names = by_index("f_rest_")
    if len(rest_names) != 3 * (max_sh_degree + 1) ** 2 - 3:
        raise ValueError("PLY holds %d f_rest_* properties, SH degree %d needs %d" % (len(rest_names), max_sh_degree, 3 * (max_sh_degree + 1) ** 2 - 3))
    rest = np.stack([v[n] for n in rest_names], axis=1).reshape(P, 3, (max_sh_degree + 1) ** 2 - 1) if rest_names else np.zeros((P, 3, 0))
    scales = np.stack([v[n] for n in by_index("scale_")], axis=1)
    rots = np.stack([v[n] for n in by_index("rot")], axis=1)
    normals = np.stack([v.get("nx", np.zeros(P)), v.get("ny", np.zeros(P)), v.get("nz", np.zeros(P))], axis=1)
    f = np.float32
    return dict(xyz=xyz.astype(f), f_dc=dc.transpose(0, 2, 1).astype(f).copy(), f_rest=rest.transpose(0, 2, 1).astype(f).copy(),
                opacity=v["opacity"][:, None].astype(f), scaling=scales.astype(f), rotation=rots.astype(f), normals=normals.astype(f))


# ---- .pt -----------------------------------------------------------------------------------------------------------------
_STORAGE_DTYPES = {"FloatStorage": torch.float32, "DoubleStorage": torch.float64, "HalfStorage": torch.float16, "BFloat16Storage": torch.bfloat16,
                   "LongStorage": torch.int64, "IntStorage": torch.int32, "ShortStorage": torch.int16, "CharStorage": torch.int8,
                   "ByteStorage": torch.uint8, "BoolStorage": torch.bool}


class _Bag:
    """Inert stand-in for a TorchScript module class (`__torch__....`): keeps the attributes, runs nothing."""

    def __init__(self, *a, **k):
        self.state = {}

    def __setstate__(self, state):
        self.state = state


class _StorageTag:
    def __init__(self, dtype):
        self.dtype = dtype


def _rebuild_tensor(storage, offset, size, stride, requires_grad=False, hooks=None, metadata=None):
    t = torch.empty(0, dtype=storage.dtype)
    t.set_(storage.untyped_storage() if hasattr(storage, "untyped_storage") else storage, offset, tuple(size), tuple(stride))
    return t


def _rebuild_parameter(data, requires_grad=False, hooks=None):
    return data


class _TensorOnlyUnpickler(pickle.Unpickler):
    """data.pkl of a torch zip archive with NOTHING callable from the file: tensors (from the archive's raw records), plain
    containers, and attribute bags for module classes."""

    def __init__(self, fh, zf, prefix):
        super().__init__(fh)
        self.zf, self.prefix = zf, prefix

    def find_class(self, module, name):
        if module == "torch._utils" and name in ("_rebuild_tensor_v2", "_rebuild_tensor"):
            return _rebuild_tensor
        if module == "torch._utils" and name == "_rebuild_parameter":
            return _rebuild_parameter
        if module == "torch" and name in _STORAGE_DTYPES:
            return _StorageTag(_STORAGE_DTYPES[name])
        if module == "collections" and name == "OrderedDict":
            import collections
            return collections.OrderedDict
        if module.startswith("__torch__") or module.startswith("torch.nn.modules") or module.startswith("torch.jit"):
            return _Bag
        raise pickle.UnpicklingError("refusing to load %s.%s from a parameter file" % (module, name))

    def persistent_load(self, pid):
        kind, tag, key, _location, numel = pid[0], pid[1], pid[2], pid[3], pid[4]
        if kind != "storage":
            raise pickle.UnpicklingError("unknown persistent id %r" % (kind,))
        dtype = tag.dtype if isinstance(tag, _StorageTag) else torch.float32
        raw = self.zf.read("%s/data/%s" % (self.prefix, key))
        return torch.frombuffer(bytearray(raw), dtype=dtype, count=int(numel)).clone() if numel else torch.empty(0, dtype=dtype)


def _tensors_in_order(obj, out):
    if torch.is_tensor(obj):
        out.append(obj)
    elif isinstance(obj, _Bag):
        _tensors_in_order(obj.state, out)
    elif isinstance(obj, dict):
        for k, v in obj.items():
            if isinstance(k, str) and k in ("training", "_is_full_backward_hook"):
                continue
            _tensors_in_order(v, out)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _tensors_in_order(v, out)
    return out


def read_parameter_tensors(path):
    """The tensors of a saved parameter file, in stored order (load_tensors takes them as xyz, f_dc, f_rest, opacity, scaling,
    rotation).  TorchScript archives (what the reference's optimized_params*.pt are) and plain torch.save files."""
    if zipfile.is_zipfile(path):
        with zipfile.ZipFile(path) as zf:
            names = zf.namelist()
            pkl = [n for n in names if n.endswith("/data.pkl") and n.count("/") == 1]
            if pkl and any(n.endswith("/constants.pkl") or "/code/" in n for n in names):  # TorchScript archive
                prefix = pkl[0].split("/")[0]
                obj = _TensorOnlyUnpickler(io.BytesIO(zf.read(pkl[0])), zf, prefix).load()
                return _tensors_in_order(obj, [])
    obj = torch.load(path, map_location="cpu", weights_only=True)
    return _tensors_in_order(obj, [])
