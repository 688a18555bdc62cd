"""Helpers shared by the GPU parity tests, smoke() and bench.py: build a python_raytracer_amd.Camera from the
dense fixture scenes and from oracle-style settings dicts."""
from python_raytracer_amd import Camera, PackedScene
from python_raytracer_amd.data import finalize_settings
from python_raytracer_amd.lib import store, vec3, quaternion


def settings_store(d):
    """oracle_lib.make_settings() dict -> python_raytracer_amd settings store (with the pixel partition)."""
    s = store(**{k: v for k, v in d.items() if k not in ("proportions", "chunk_radius")})
    if not hasattr(s, "culling"):
        s.culling = False
    return finalize_settings(s)


def camera_for(scene, settings, pos, rot, lens, grid=None, device=None):
    """Camera over a dense oracle_lib.Scene.  grid defaults to the scene's camera grid."""
    cam = Camera(settings=settings, device=device)
    cam.pos = vec3(*[float(v) for v in pos])
    cam.rot = quaternion(*[float(v) for v in rot])
    cam.lens = float(lens)
    g = scene.grid if grid is None else grid
    cam.set_packed_scene(PackedScene.from_dense(scene.origin, scene.dims, scene.chunk_size, scene.present, scene.res,
                                                g, scene.materials))
    return cam
