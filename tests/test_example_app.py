"""examples/raytracer_main: the reference's main() (scene file -> frame -> image file) on the C++ host surface.
The binary is built by __graft_entry__.build() (examples/Makefile)."""
import os
import subprocess

import numpy as np
import pytest

from raytracer_amd.scene import DeviceScene, FlatScene, Params, reference_camera
from test_scene_yaml import write_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
APP = os.path.join(ROOT, "examples", "raytracer_main")


def test_example_builds_and_reports_usage():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    r = subprocess.run([APP], capture_output=True, text=True)
    assert r.returncode != 0 and "You have to provide a scene file." in r.stderr     # ref src/raytracer.cpp:632-635


@pytest.mark.gpu
def test_example_renders_scene_file(tmp_path):
    assert os.path.exists(APP), "examples/raytracer_main missing: run __graft_entry__.build()"
    path, _ = write_scene(tmp_path)
    out = tmp_path / "frame.ppm"
    r = subprocess.run([APP, str(path), "-o", str(out), "-w", "320", "-h", "240"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "... start ray tracer" in r.stdout and "pix/s" in r.stdout
    b = out.read_bytes()
    head = b"P6\n320 240\n255\n"
    assert b.startswith(head)
    rgb = np.frombuffer(b[len(head):], np.uint8).reshape(240, 320, 3)
    want, _ = DeviceScene(FlatScene.load_yaml(path)).render(reference_camera(320, 240), Params())
    assert np.array_equal(rgb[:, :, ::-1], want)
    # the same frame spread over two replicas (Scene::setDevices -> rt_multi_render; the box has one GPU, named twice)
    out2 = tmp_path / "frame2.ppm"
    r = subprocess.run([APP, str(path), "-o", str(out2), "-w", "320", "-h", "240", "-devices", "0,0"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert out2.read_bytes() == b
    bad = subprocess.run([APP, str(tmp_path / "missing.yaml")], capture_output=True, text=True)
    assert bad.returncode == 1 and "could not be opened" in bad.stderr
