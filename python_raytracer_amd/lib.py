"""Host-side value types of the drop-in boundary.

The reference passes `vec3`, `quaternion` and `rgb` objects across its Camera / Material API
(reference lib.py:166-395) and a `store` attribute bag for settings and rays (lib.py:7-10).  This module
provides the same names with the members the boundary needs, so code written against the reference's
`Camera.pos`, `Camera.rot`, `Material(albedo=rgb(...))` keeps working.  Any object exposing the same
attributes (`.x/.y/.z`, `.x/.y/.z/.w`, `.r/.g/.b`) is accepted as well, including the reference's own.

The per-ray arithmetic itself (vector ops, shaders) runs on the GPU; `material` and
`material_background` below are the identities a Material / background selects the on-device default
shaders with (reference lib.py:448-476) -- they are not executed on the host.
"""
import math


class store:
    """Attribute bag (reference lib.py:7-10)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __repr__(self):
        return "store(%s)" % ", ".join("%s=%r" % kv for kv in self.__dict__.items())


class vec3:
    __slots__ = ("x", "y", "z")

    def __init__(self, x, y, z):
        self.x, self.y, self.z = x, y, z

    def _map(self, other, op):
        if isinstance(other, vec3):
            return vec3(op(self.x, other.x), op(self.y, other.y), op(self.z, other.z))
        return vec3(op(self.x, other), op(self.y, other), op(self.z, other))

    def __add__(self, o):
        return self._map(o, lambda a, b: a + b)

    def __sub__(self, o):
        return self._map(o, lambda a, b: a - b)

    def __mul__(self, o):
        return self._map(o, lambda a, b: a * b)

    def __truediv__(self, o):
        return self._map(o, lambda a, b: a / b)

    def __floordiv__(self, o):
        return self._map(o, lambda a, b: a // b)

    def __neg__(self):
        return vec3(-self.x, -self.y, -self.z)

    def __eq__(self, o):
        if isinstance(o, vec3):
            return (self.x, self.y, self.z) == (o.x, o.y, o.z)
        return self.x == o and self.y == o and self.z == o

    def __hash__(self):
        return hash((self.x, self.y, self.z))

    def __iter__(self):
        return iter((self.x, self.y, self.z))

    def __repr__(self):
        return "vec3(%r, %r, %r)" % (self.x, self.y, self.z)

    def __str__(self):
        return "%s,%s,%s" % (self.x, self.y, self.z)

    def array(self):
        return [self.x, self.y, self.z]

    def tuple(self):
        return (self.x, self.y, self.z)

    def snapped(self, unit):
        """Floor each component to a multiple of `unit` (reference lib.py:316-320)."""
        return self._map(unit, lambda a, u: (a // u) * u)

    def distance(self, other):
        return math.dist(self.array(), other.array())

    def quaternion(self):
        """Euler degrees -> quaternion with the reference's axis convention (lib.py:322-338)."""
        hx, hy, hz = (math.radians(a) / 2 for a in (self.x, self.y, self.z))
        sx, cx, sy, cy, sz, cz = math.sin(hx), math.cos(hx), math.sin(hy), math.cos(hy), math.sin(hz), math.cos(hz)
        return quaternion(sx * cy * cz - cx * sy * sz, cx * sy * cz - sx * cy * sz,
                          cx * cy * sz + sx * sy * cz, cx * cy * cz + sx * sy * sz)


class quaternion:
    __slots__ = ("x", "y", "z", "w")

    def __init__(self, x, y, z, w):
        self.x, self.y, self.z, self.w = x, y, z, w

    def __repr__(self):
        return "quaternion(%r, %r, %r, %r)" % (self.x, self.y, self.z, self.w)

    def array(self):
        return [self.x, self.y, self.z, self.w]

    def multiply(self, o):
        """The reference's quaternion product (lib.py:353-358), term for term.  It is NOT the Hamilton product (two signs
        differ) and does not preserve the norm: a rotated camera's primary velocity is not unit (Camera._velocity_bound)."""
        a, b = self, o
        return quaternion(a.w * b.x + a.z * b.y - a.y * b.z + a.x * b.w,
                          a.z * b.x + a.w * b.y + a.x * b.z + a.y * b.w,
                          a.y * b.x - a.x * b.y + a.w * b.z + a.z * b.w,
                          a.x * b.x - a.y * b.y - a.z * b.z + a.w * b.w)

    def vec_forward(self):
        """(lib.py:372-376)"""
        return vec3(2 * (self.z * self.x + self.w * self.y), 2 * (self.y * self.x - self.w * self.z),
                    1 - 2 * (self.z ** 2 + self.y ** 2))


class rgb:
    __slots__ = ("r", "g", "b")

    def __init__(self, r, g, b):
        self.r, self.g, self.b = r, g, b

    def __repr__(self):
        return "rgb(%r, %r, %r)" % (self.r, self.g, self.b)

    def array(self):
        return [self.r, self.g, self.b]

    def tuple(self):
        return (self.r, self.g, self.b)


def material(ray, mat, settings):
    """Identity of the default PBR shader (reference lib.py:448-460).

    Assign it as `Material(function=material, ...)`.  The shader body runs inside the HIP march kernel
    (python_raytracer_amd/csrc/vrt_kernels.hip, hit_body); it cannot be called on the host."""
    raise RuntimeError("lib.material is evaluated on the GPU by Camera.tile/trace; it is not callable on the host")


def material_background(ray, settings):
    """Identity of the default sky shader (reference lib.py:463-476); runs on the GPU, see `material`."""
    raise RuntimeError("lib.material_background is evaluated on the GPU by Camera.tile/trace")


def is_default_material_function(f):
    """True for this module's `material` or the reference's own lib.material."""
    return f is material or (callable(f) and getattr(f, "__name__", "") == "material"
                             and getattr(f, "__module__", "") in ("lib", __name__))


def is_default_background(f):
    return f is material_background or (callable(f) and getattr(f, "__name__", "") == "material_background"
                                        and getattr(f, "__module__", "") in ("lib", __name__))
