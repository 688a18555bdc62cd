"""MI355X-native voxel ray-march renderer: the Camera.tile -> Camera.trace hot path of
MirceaKitsune/python_raytracer behind the reference's Camera / Material API.

    from python_raytracer_amd import Camera, Material, Frame, data
    from python_raytracer_amd.lib import vec3, quaternion, rgb, material, material_background

The compute path is the HIP library python_raytracer_amd/_vrt.so (include/vrt.h); there is no CPU fallback.
"""
from . import data, lib
from .data import Material, Frame, make_settings, load_settings, pixel_partition
from .lib import vec3, quaternion, rgb, store, material, material_background
from .scene import PackedScene
from .camera import Camera, RenderResult, release_caches
from .canvas import Canvas
from . import world

__all__ = ["Camera", "RenderResult", "release_caches", "Canvas", "world", "Material", "Frame", "PackedScene", "data", "lib", "vec3", "quaternion", "rgb",
           "store", "material", "material_background", "make_settings", "load_settings", "pixel_partition"]
