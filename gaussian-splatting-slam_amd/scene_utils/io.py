"""On-disk formats of the reference's Gaussian model, written against the format, not against `plyfile`:

* PLY (reference scene/gaussian_model.py:193-224 save_ply, :231-272 load_ply): one `vertex` element, binary little-endian,
  all properties `float`, in the order  x y z nx ny nz  f_dc_0..2  f_rest_0..(3(M-1)-1)  opacity  scale_0..2  rot_0..3 ;
  f_dc / f_rest are stored CHANNEL-major (`transpose(1, 2).flatten`), normals are zeros, values are the RAW (pre-activation)
  parameters.  Files written here load in the upstream viewers / `GaussianModel.load_ply`, and vice versa.
* checkpoint tuple (reference :67-99 capture / restore): (active_sh_degree, xyz, f_dc, f_rest, scaling, rotation, opacity,
  max_radii2D, xyz_gradient_accum, denom, optimizer.state_dict(), spatial_lr_scale).
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn


def ply_attribute_names(n_dc: int, n_rest: int):
    """reference construct_list_of_attributes (:101-114)"""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)]
    names += [f"f_rest_{i}" for i in range(n_rest)]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(3)]
    names += [f"rot_{i}" for i in range(4)]
    return names


def save_ply(model, path: str):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    xyz = model._xyz.detach().cpu().numpy()
    f_dc = model._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
    f_rest = model._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
    attrs = np.concatenate((xyz, np.zeros_like(xyz), f_dc, f_rest, model._opacity.detach().cpu().numpy(),
                            model._scaling.detach().cpu().numpy(), model._rotation.detach().cpu().numpy()), axis=1)
    names = ply_attribute_names(f_dc.shape[1], f_rest.shape[1])
    assert attrs.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % xyz.shape[0]
    header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(attrs, dtype="<f4").tobytes())


_PLY_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1",
              "char": "i1", "int8": "i1", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2", "int": "<i4",
              "int32": "<i4", "uint": "<u4", "uint32": "<u4"}


def read_ply_vertices(path: str):
    """-> dict name -> float64 array [P].  Binary little-endian and ascii PLY with scalar vertex properties.  Elements
    declared BEFORE `vertex` (their data precedes the vertices in the body) are skipped when all their properties are scalars;
    a list property there makes the offset of the vertex data unknowable without parsing it, which is refused."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elements = None, []          # elements: [name, count, [(prop, dtype)], has_list]
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated header")
            tok = line.decode("ascii").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append([tok[1], int(tok[2]), [], False])
            elif tok[0] == "property" and elements:
                if tok[1] == "list":
                    elements[-1][3] = True
                else:
                    if tok[1] not in _PLY_TYPES:
                        raise ValueError(f"{path}: unknown property type {tok[1]}")
                    elements[-1][2].append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        names = [e[0] for e in elements]
        if "vertex" not in names:
            raise ValueError(f"{path}: no vertex element")
        vi = names.index("vertex")
        _, count, props, has_list = elements[vi]
        if has_list:
            raise ValueError("list properties on vertices are not supported")
        for name, n, pr, lst in elements[:vi]:
            if lst:
                raise ValueError(f"{path}: element '{name}' with a list property precedes the vertices")
        if fmt == "binary_little_endian":
            for name, n, pr, lst in elements[:vi]:
                f.seek(n * np.dtype(pr).itemsize if pr else 0, 1)
            data = np.frombuffer(f.read(count * np.dtype(props).itemsize), dtype=np.dtype(props), count=count)
            return {n: data[n].astype(np.float64) for n, _ in props}
        if fmt == "ascii":
            for name, n, pr, lst in elements[:vi]:
                for _ in range(n):
                    f.readline()
            arr = np.loadtxt(f, max_rows=count, ndmin=2)
            return {n: arr[:, i].astype(np.float64) for i, (n, _) in enumerate(props)}
        raise ValueError(f"{path}: unsupported PLY format {fmt}")


def load_ply(model, path: str, device="cuda"):
    """Fills `model` (a scene_utils.GaussianModel) exactly as reference load_ply (:231-272) does."""
    v = read_ply_vertices(path)
    P = v["x"].shape[0]
    xyz = np.stack((v["x"], v["y"], v["z"]), axis=1)
    f_dc = np.stack((v["f_dc_0"], v["f_dc_1"], v["f_dc_2"]), axis=1)[:, :, None]                  # [P,3,1]
    rest_names = sorted((n for n in v if n.startswith("f_rest_")), key=lambda s: int(s.split("_")[-1]))
    assert len(rest_names) == 3 * (model.max_sh_degree + 1) ** 2 - 3, "PLY SH degree does not match the model"
    f_rest = np.stack([v[n] for n in rest_names], axis=1).reshape(P, 3, (model.max_sh_degree + 1) ** 2 - 1)
    scales = np.stack([v[n] for n in sorted((n for n in v if n.startswith("scale_")), key=lambda s: int(s.split("_")[-1]))], 1)
    rots = np.stack([v[n] for n in sorted((n for n in v if n.startswith("rot")), key=lambda s: int(s.split("_")[-1]))], 1)

    def par(a):
        return nn.Parameter(torch.tensor(a, dtype=torch.float, device=device).contiguous().requires_grad_(True))
    model._xyz = par(xyz)
    model._features_dc = nn.Parameter(torch.tensor(f_dc, dtype=torch.float, device=device).transpose(1, 2).contiguous()
                                      .requires_grad_(True))
    model._features_rest = nn.Parameter(torch.tensor(f_rest, dtype=torch.float, device=device).transpose(1, 2).contiguous()
                                        .requires_grad_(True))
    model._opacity = par(v["opacity"][:, None])
    model._scaling = par(scales)
    model._rotation = par(rots)
    model.active_sh_degree = model.max_sh_degree
    return model


def capture(model):
    """reference :67-82"""
    return (model.active_sh_degree, model._xyz, model._features_dc, model._features_rest, model._scaling, model._rotation,
            model._opacity, model.max_radii2D, model.xyz_gradient_accum, model.denom, model.optimizer.state_dict(),
            getattr(model, "spatial_lr_scale", 1.0))


def restore(model, model_args, optimizer="hip"):
    """reference :83-99"""
    (model.active_sh_degree, model._xyz, model._features_dc, model._features_rest, model._scaling, model._rotation,
     model._opacity, max_radii2D, xyz_gradient_accum, denom, opt_dict, model.spatial_lr_scale) = model_args
    model.training_setup(optimizer=optimizer)
    model.max_radii2D, model.xyz_gradient_accum, model.denom = max_radii2D, xyz_gradient_accum, denom
    model.optimizer.load_state_dict(opt_dict)
    return model
