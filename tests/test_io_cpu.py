"""CPU tests of the on-disk formats: PLY layout of reference scene/gaussian_model.py:193-272 and the checkpoint
tuple of :67-99."""
import os

import numpy as np
import torch

from scene_utils import GaussianModel, make_gaussians, save_ply, load_ply, read_ply_vertices, capture, restore
from scene_utils.io import ply_attribute_names


def test_ply_layout_matches_reference_format(tmp_path):
    raw = make_gaussians(37, 3, seed=1)
    m = GaussianModel.from_raw(raw)
    path = os.path.join(tmp_path, "point_cloud", "iteration_7", "point_cloud.ply")     # reference scene/__init__.py:104-106
    save_ply(m, path)
    blob = open(path, "rb").read()
    head, body = blob.split(b"end_header\n", 1)
    lines = head.decode().splitlines()
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 37"]
    names = [l.split()[2] for l in lines[3:]]
    assert all(l.startswith("property float ") for l in lines[3:])
    assert names == ply_attribute_names(3, 45)
    assert names[:6] == ["x", "y", "z", "nx", "ny", "nz"] and names[6:9] == ["f_dc_0", "f_dc_1", "f_dc_2"]
    assert names[9] == "f_rest_0" and names[53] == "f_rest_44" and names[54:] == ["opacity", "scale_0", "scale_1", "scale_2",
                                                                                "rot_0", "rot_1", "rot_2", "rot_3"]
    data = np.frombuffer(body, dtype="<f4").reshape(37, 62)
    assert np.array_equal(data[:, 0:3], raw.xyz.numpy()) and not data[:, 3:6].any()
    # f_rest is channel-major: column 9 + c*15 + k  <-  features_rest[:, k, c]
    assert np.array_equal(data[:, 9 + 1 * 15 + 4], raw.features_rest[:, 4, 1].numpy())
    assert np.array_equal(data[:, 54], raw.opacity[:, 0].numpy())


def test_ply_round_trip(tmp_path):
    raw = make_gaussians(101, 3, seed=2)
    m = GaussianModel.from_raw(raw)
    path = os.path.join(tmp_path, "pc.ply")
    save_ply(m, path)
    m2 = load_ply(GaussianModel(3), path, device="cpu")
    for a in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        assert torch.equal(getattr(m, a).detach(), getattr(m2, a).detach()), a
        assert getattr(m2, a).requires_grad and getattr(m2, a).is_contiguous()
    assert m2.active_sh_degree == 3
    v = read_ply_vertices(path)
    assert set(v) == set(ply_attribute_names(3, 45))


def test_ascii_ply_is_readable(tmp_path):
    path = os.path.join(tmp_path, "a.ply")
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 2\nproperty float x\nproperty float y\nproperty float z\nend_header\n")
        f.write("1 2 3\n4 5 6\n")
    v = read_ply_vertices(path)
    assert np.array_equal(v["y"], [2.0, 5.0])


def test_checkpoint_capture_restore():
    raw = make_gaussians(50, 1, seed=3)
    m = GaussianModel.from_raw(raw)
    m.training_setup(optimizer="torch")
    for p in m.parameters():
        p.grad = torch.ones_like(p) * 0.01
    m.optimizer.step()
    m.xyz_gradient_accum += 1.5
    ck = capture(m)
    assert len(ck) == 12 and ck[0] == m.active_sh_degree
    m2 = restore(GaussianModel(1), ck, optimizer="torch")
    assert torch.equal(m2._xyz, m._xyz) and torch.equal(m2.xyz_gradient_accum, m.xyz_gradient_accum)
    s1 = m.optimizer.state[m._xyz]["exp_avg"]
    s2 = m2.optimizer.state[m2._xyz]["exp_avg"]
    assert torch.equal(s1, s2)


def test_ply_reader_skips_elements_declared_before_vertex(tmp_path):
    """A PLY whose header declares another element BEFORE `vertex` keeps that element's data in front of the vertices
    (the reader used to take the bytes right after the header for vertices).  Hand-written files, both encodings;
    a list property in front of the vertices is refused rather than mis-parsed."""
    import numpy as np
    from scene_utils import read_ply_vertices
    verts = np.arange(12, dtype="<f4").reshape(4, 3) + 0.5
    hdr = ("ply\nformat binary_little_endian 1.0\nelement camera 2\nproperty float fx\nproperty uchar id\n"
           "element vertex 4\nproperty float x\nproperty float y\nproperty float z\nend_header\n")
    p = tmp_path / "pre.ply"
    cam = np.zeros(2, dtype=[("fx", "<f4"), ("id", "u1")])
    cam["fx"] = (7.0, 9.0)
    with open(p, "wb") as f:
        f.write(hdr.encode("ascii")); f.write(cam.tobytes()); f.write(verts.tobytes())
    v = read_ply_vertices(str(p))
    assert np.array_equal(np.stack((v["x"], v["y"], v["z"]), 1), verts.astype(np.float64))
    pa = tmp_path / "pre_ascii.ply"
    with open(pa, "w") as f:
        f.write(hdr.replace("binary_little_endian", "ascii"))
        f.write("7.0 1\n9.0 2\n")
        for r in verts:
            f.write(" ".join(str(float(x)) for x in r) + "\n")
    v = read_ply_vertices(str(pa))
    assert np.array_equal(np.stack((v["x"], v["y"], v["z"]), 1), verts.astype(np.float64))
    pl = tmp_path / "list.ply"
    with open(pl, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement face 1\nproperty list uchar int vertex_indices\n"
                 "element vertex 1\nproperty float x\nend_header\n").encode("ascii"))
        f.write(b"\x03" + np.zeros(3, "<i4").tobytes() + np.zeros(1, "<f4").tobytes())
    import pytest
    with pytest.raises(ValueError, match="precedes"):
        read_ply_vertices(str(pl))
