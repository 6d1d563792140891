"""The C++ host surface (raytracer_amd/host/raytracer.hpp through the rt_host_* C entry points) against the
reference's own results (tests/golden/host.npz): camera construction / rotation / move, plane normalisation,
rotation matrices, the PLY loader with its transform chain, the bounding sphere, and the known answers of the
reference's check_op_overloading(). Host-only: no GPU call is made."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import gold, same_bits
from raytracer_amd import cabi
from raytracer_amd.cabi import fptr, rt_plane
from raytracer_amd.scene import Mesh, camera_to_array, host_lib, make_camera


def test_library_exports_every_declared_symbol():
    lib = cabi.load()
    for name in cabi.EXPORTS + cabi.HOST_EXPORTS:
        assert hasattr(lib, name), name
    # and the headers declare nothing else
    inc = os.path.join(os.path.dirname(cabi.HERE), "include")
    import re
    declared = set()
    for h in ("raytracer_amd.h", "raytracer_amd_host.h"):
        text = open(os.path.join(inc, h)).read()
        declared |= set(re.findall(r"\b(rt_(?:host_)?[a-z_0-9]+)\s*\(", text))
    assert declared == set(cabi.EXPORTS + cabi.HOST_EXPORTS), declared ^ set(cabi.EXPORTS + cabi.HOST_EXPORTS)
    assert b"gfx950" in lib.rt_version()


def test_known_answers():
    out = np.zeros(16, np.float32)
    lib = host_lib()
    assert lib.rt_host_known_answers(fptr(out)) == 0
    # the values printed as expectations in the reference: 1 2 3 | 2 4 6 | 1 4 9 | 2 4 6 | 0 0 1 | 1
    assert out.tolist() == [1, 2, 3, 2, 4, 6, 1, 4, 9, 2, 4, 6, 0, 0, 1, 1]
    assert same_bits(out, gold("host.npz")["known_answers"])


def test_camera_make():
    g = gold("host.npz")
    for cin, cout in zip(g["cam_in"], g["cam_out"]):
        rot = None if np.isnan(cin[12]) else cin[12:15]
        move = None if np.isnan(cin[15]) else cin[15:18]
        cam = make_camera(int(cin[10]), int(cin[11]), fov=float(cin[9]), eye=cin[0:3], up=cin[3:6], at=cin[6:9],
                          rotate=rot, move=move)
        assert same_bits(camera_to_array(cam)[:14], cout[:14])
        assert (cam.imWidth, cam.imHeight) == (int(cout[14]), int(cout[15]))


def test_plane_and_rotation():
    g = gold("host.npz")
    lib = host_lib()
    for nrm, d, want in zip(g["plane_normals"], g["plane_dists"], g["planes"]):
        p = rt_plane()
        assert lib.rt_host_plane_make(fptr(np.ascontiguousarray(nrm)), float(d), C.byref(p)) == 0
        assert same_bits(np.array([p.a, p.b, p.c, p.d], np.float32), want)
    for ang, want in zip(g["rot_angles"], g["rot_mats"]):
        m = np.zeros(9, np.float32)
        assert lib.rt_host_mat3_rotation(float(ang[0]), float(ang[1]), float(ang[2]), fptr(m)) == 0
        assert same_bits(m, want)


def test_ply_loader_and_transform(tmp_path):
    g = gold("host.npz")
    path = tmp_path / "blob.ply"
    path.write_bytes(g["ply_text"].tobytes())
    raw = Mesh.load_ply(path, transform=False)
    assert same_bits(np.stack(raw.v), g["ply_raw"]) and same_bits(raw.bb, g["ply_raw_bb"])
    xf = Mesh.load_ply(path, rotation=g["ply_rot"], scale=float(g["ply_scale"]), translation=g["ply_trans"])
    assert same_bits(np.stack(xf.v), g["ply_xf"]) and same_bits(xf.bb, g["ply_xf_bb"])
    # n0 is the (rotated) flat face normal, n1/n2 the smooth vertex normals: the reference's constructor quirk
    assert not same_bits(xf.v[3], xf.v[4])
    bb = np.zeros(4, np.float32)
    lib = host_lib()
    assert lib.rt_host_bbvol(fptr(xf.v[0]), fptr(xf.v[1]), fptr(xf.v[2]), xf.num_faces, fptr(bb)) == 0
    assert same_bits(bb, g["bbvol_of_xf"])


def test_ply_loader_errors(tmp_path):
    lib = host_lib()
    h = C.c_void_p()
    assert lib.rt_host_model_load_ply(str(tmp_path / "missing.ply").encode(), C.byref(h)) != 0
    assert b"could not be loaded" in lib.rt_last_error()
    quad = tmp_path / "quad.ply"
    quad.write_text("ply\nformat ascii 1.0\nelement vertex 4\nelement face 1\nend_header\n"
                    "0 0 0\n1 0 0\n1 1 0\n0 1 0\n4 0 1 2 3\n")
    assert lib.rt_host_model_load_ply(str(quad).encode(), C.byref(h)) != 0
    assert b"Only triangles are supported" in lib.rt_last_error()
    nohdr = tmp_path / "nohdr.ply"
    nohdr.write_text("ply\nelement vertex 1\n0 0 0\n")
    assert lib.rt_host_model_load_ply(str(nohdr).encode(), C.byref(h)) != 0


def test_scene_create_needs_a_gpu_or_fails_loudly():
    """No CPU fallback: without a device rt_scene_create reports an error instead of rendering elsewhere."""
    import torch
    if torch.cuda.is_available():
        return
    from raytracer_amd import workloads as wl
    from raytracer_amd.scene import DeviceScene
    try:
        DeviceScene(wl.scene_analytic())
    except cabi.RtError as e:
        assert "rt error" in str(e)
    else:
        raise AssertionError("rt_scene_create succeeded without a GPU")


def test_write_image_png_and_ppm(tmp_path):
    """write_image (host surface, replaces cv::imwrite of main()): the PNG decodes (zlib + CRCs) to the frame's RGB
    bytes, the PPM holds them verbatim."""
    import struct
    import zlib
    from raytracer_amd.scene import write_image
    rng = np.random.default_rng(5)
    frame = rng.integers(0, 256, (37, 301, 3), dtype=np.uint8)        # BGR, more than one 64-KiB deflate block
    rgb = frame[:, :, ::-1]
    png, ppm = tmp_path / "f.png", tmp_path / "f.ppm"
    write_image(png, frame)
    write_image(ppm, frame)
    b = png.read_bytes()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, seen = 8, b"", []
    while pos < len(b):
        n, typ = struct.unpack(">I4s", b[pos:pos + 8])
        data = b[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(typ + data)
        seen.append(typ)
        if typ == b"IHDR":
            assert struct.unpack(">IIBBBBB", data) == (301, 37, 8, 2, 0, 0, 0)
        if typ == b"IDAT":
            idat += data
        pos += 12 + n
    assert seen == [b"IHDR", b"IDAT", b"IEND"]
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(37, 1 + 301 * 3)
    assert not raw[:, 0].any() and np.array_equal(raw[:, 1:].reshape(37, 301, 3), rgb)
    p = ppm.read_bytes()
    head = b"P6\n301 37\n255\n"
    assert p.startswith(head) and np.array_equal(np.frombuffer(p[len(head):], np.uint8).reshape(37, 301, 3), rgb)
    with pytest.raises(Exception):
        write_image(tmp_path / "f.jpg", frame)
