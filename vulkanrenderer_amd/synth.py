"""Seeded synthetic inputs for the shading path: a G-buffer, lights, shadow maps and IBL images
of the shapes the reference's geometry / shadow passes would hand to the composition pass
(gpass.frag:16-24 for the G-buffer channels, point_shadow.frag:12-14 for the cube depth encoding
length/25, Program.cs:84,125-132 for the default camera and directional light).

Everything is generated with numpy from a seed, and every stored value is already rounded to the
storage format (binary16 for RGBA16F images), so a CPU checker and the GPU see identical bits.
"""
from __future__ import annotations

import dataclasses
import numpy as np

FAR_PLANE = 25.0


def _f16(a: np.ndarray) -> np.ndarray:
    """float array -> binary16 bit patterns (round to nearest even)."""
    return np.ascontiguousarray(a.astype(np.float16)).view(np.uint16)


def _smooth_field(rng: np.random.Generator, n: int, cells: int = 16, count: int = 1) -> np.ndarray:
    """`count` smooth random n x n fields in [0,1]: a (cells+1)^2 random grid, bilinearly upsampled."""
    g = rng.random((count, cells + 1, cells + 1), dtype=np.float32)
    t = (np.arange(n, dtype=np.float32) + 0.5) / n * cells
    i0 = np.minimum(t.astype(np.int64), cells - 1)
    fr = t - i0
    a = np.zeros((n, cells + 1), dtype=np.float32)
    a[np.arange(n), i0] = 1.0 - fr
    a[np.arange(n), i0 + 1] = fr
    return np.einsum("ik,ckl,jl->cij", a, g, a, optimize=True).astype(np.float32)


def look_at(eye, target, up) -> np.ndarray:
    """Right-handed look-at matrix in the row-vector convention of Silk.NET's Matrix4X4.CreateLookAt."""
    eye, target, up = (np.asarray(v, dtype=np.float64) for v in (eye, target, up))
    z = eye - target
    z /= np.linalg.norm(z)
    x = np.cross(up, z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2] = x, y, z
    m[3, :3] = [-x @ eye, -y @ eye, -z @ eye]
    return m


def ortho_off_center(left, right, bottom, top, near, far) -> np.ndarray:
    """Matrix4X4.CreateOrthographicOffCenter (row-vector convention, depth 0..1)."""
    m = np.eye(4)
    m[0, 0] = 2.0 / (right - left)
    m[1, 1] = 2.0 / (top - bottom)
    m[2, 2] = 1.0 / (near - far)
    m[3, 0] = (left + right) / (left - right)
    m[3, 1] = (top + bottom) / (bottom - top)
    m[3, 2] = near / (near - far)
    return m


def light_volume_radius(color) -> float:
    """Light.cs:20-26."""
    return 16.0 * float(np.sqrt(np.float32(max(color)) / np.float32(5.0)))


SCENE_DTYPE = np.dtype([
    ("camera_view", np.float32, 16), ("camera_projection", np.float32, 16),
    ("camera_position", np.float32, 3), ("_pad0", np.float32),
    ("dir_light_direction", np.float32, 3), ("_pad1", np.float32),
    ("dir_light_color", np.float32, 3), ("_pad2", np.float32),
    ("light_space_matrix", np.float32, 16)])
LIGHT_DTYPE = np.dtype([
    ("model", np.float32, 16), ("position", np.float32, 3), ("_pad0", np.float32), ("color", np.float32, 3)])
assert SCENE_DTYPE.itemsize == 240 and LIGHT_DTYPE.itemsize == 92  # VulkanRenderer.cs:11-28


def perspective(fov_degrees: float, aspect: float, near: float, far: float) -> np.ndarray:
    """Camera.GetProjectionMatrix (Camera.cs:30-35): right-handed CreatePerspectiveFieldOfView, depth 0..1,
    row-vector convention, with M22 negated because Vulkan's y points down."""
    f = 1.0 / np.tan(np.radians(fov_degrees) * 0.5)
    m = np.zeros((4, 4), dtype=np.float64)
    m[0, 0] = f / aspect
    m[1, 1] = -f
    m[2, 2] = far / (near - far)
    m[2, 3] = -1.0
    m[3, 2] = near * far / (near - far)
    return m


def reference_light_volume():
    """The reference's own light-volume mesh -- positions and triangles of assets/models/sphere/sphere.glb
    (VulkanRenderer.cs:78,269), extracted by tools/make_golden.py --only-light-volume into tests/golden/light_volume.npz:
    559 vertices (482 positions, repeated along the texture seam), 960 triangles, unit radius, counter-clockwise seen
    from outside.  Returns (verts [559,3] f32, indices [960,3] u32)."""
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "light_volume.npz")
    with np.load(path) as d:
        return np.ascontiguousarray(d["positions"], dtype=np.float32), np.ascontiguousarray(d["triangles"], dtype=np.uint32)


def make_sphere_mesh(segments: int = 32, rings: int = 16):
    """A unit UV sphere with the reference light volume's topology (assets/models/sphere: 32 segments,
    16 rings, 960 triangles), counter-clockwise seen from outside.  Returns (verts [n,3] f32, indices [960,3] u32)."""
    verts = [(0.0, 1.0, 0.0)]
    for r in range(1, rings):
        th = np.pi * r / rings
        for s_ in range(segments):
            ph = 2.0 * np.pi * s_ / segments
            verts.append((np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)))
    verts.append((0.0, -1.0, 0.0))
    ring = lambda r, s_: 1 + (r - 1) * segments + s_ % segments
    south = len(verts) - 1
    tris = []
    for s_ in range(segments):
        tris.append((0, ring(1, s_ + 1), ring(1, s_)))
        tris.append((south, ring(rings - 1, s_), ring(rings - 1, s_ + 1)))
    for r in range(1, rings - 1):
        for s_ in range(segments):
            a, b, c, d = ring(r, s_), ring(r, s_ + 1), ring(r + 1, s_), ring(r + 1, s_ + 1)
            tris.append((a, b, d))
            tris.append((a, d, c))
    v = np.asarray(verts, dtype=np.float32)
    t = np.asarray(tris, dtype=np.uint32)
    # orientation check: normals of counter-clockwise triangles point away from the centre
    n = np.cross(v[t[:, 1]] - v[t[:, 0]], v[t[:, 2]] - v[t[:, 0]])
    flip = (n * (v[t[:, 0]] + v[t[:, 1]] + v[t[:, 2]])).sum(-1) < 0
    t[flip] = t[flip][:, [0, 2, 1]]
    return v, t


def height_field(x, z):
    """The synthetic scene's surface: y = 0.3 sin(1.7 x) cos(1.3 z)."""
    return 0.3 * np.sin(1.7 * x) * np.cos(1.3 * z)


def make_scene_mesh(n: int = 64, thickness: float = 0.3, extent: float = 3.0):
    """The height field over [-extent, extent]^2 as a closed slab (top surface, a bottom `thickness` below,
    side walls), counter-clockwise seen from outside, so that a shadow pass that culls front faces keeps its
    far side.  Returns (verts [2(n+1)^2, 3] f32, triangles [m, 3] u32)."""
    g = np.linspace(-extent, extent, n + 1)
    xx, zz = np.meshgrid(g, g, indexing="xy")          # [row z, col x]
    top = np.stack([xx, height_field(xx, zz), zz], -1).reshape(-1, 3)
    bottom = top - np.array([0.0, thickness, 0.0])
    verts = np.concatenate([top, bottom]).astype(np.float32)
    nb = (n + 1) * (n + 1)
    idx = lambda r, c: r * (n + 1) + c
    tris = []
    for r in range(n):
        for c in range(n):
            a, b, cc, d = idx(r, c), idx(r, c + 1), idx(r + 1, c), idx(r + 1, c + 1)
            tris += [(a, cc, b), (b, cc, d)]                       # top: normal +y
            tris += [(nb + a, nb + b, nb + cc), (nb + b, nb + d, nb + cc)]  # bottom: normal -y
    for k in range(n):
        for (p, q) in ((idx(0, k), idx(0, k + 1)), (idx(n, k + 1), idx(n, k)), (idx(k + 1, 0), idx(k, 0)), (idx(k, n), idx(k + 1, n))):
            tris += [(p, q, nb + q), (p, nb + q, nb + p)]          # walls (orientation fixed below)
    t = np.asarray(tris, dtype=np.uint32)
    v = verts.astype(np.float64)
    nrm = np.cross(v[t[:, 1]] - v[t[:, 0]], v[t[:, 2]] - v[t[:, 0]])
    centre = np.array([0.0, -0.5 * thickness, 0.0])
    mid = (v[t[:, 0]] + v[t[:, 1]] + v[t[:, 2]]) / 3.0
    walls = np.arange(len(t)) >= 4 * n * n
    flip = walls & ((nrm * (mid - centre)).sum(-1) < 0)
    t[flip] = t[flip][:, [0, 2, 1]]
    return verts, t


def mesh_attributes(verts: np.ndarray, tris: np.ndarray, uv_scale: float = 1.0):
    """Per-vertex normals (area-weighted), planar texture coordinates and tangents for a mesh: what gpass.vert
    reads beside the position (Mesh.cs vertex layout: position, texcoord, normal, tangent)."""
    v = verts.astype(np.float64)
    fn = np.cross(v[tris[:, 1]] - v[tris[:, 0]], v[tris[:, 2]] - v[tris[:, 0]])
    n = np.zeros_like(v)
    for k in range(3):
        np.add.at(n, tris[:, k], fn)
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-20)
    uv = np.stack([v[:, 0], v[:, 2]], -1) * uv_scale + 0.5 * v[:, 1:2]
    t = np.cross(np.array([0.0, 0.0, 1.0]), n)
    small = np.linalg.norm(t, axis=1) < 1e-3
    t[small] = np.cross(np.array([1.0, 0.0, 0.0]), n[small])
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    return uv.astype(np.float32), n.astype(np.float32), t.astype(np.float32)


def make_material(rng: np.random.Generator, size: int = 64):
    """Three RGBA8 images (albedo, normal map, ao/roughness/metalness) with smooth content."""
    f = _smooth_field(rng, size, 6, 9)
    albedo = np.concatenate([0.1 + 0.8 * f[0:3], np.ones((1, size, size))]).transpose(1, 2, 0)
    tilt = 0.35 * (f[3:5] - 0.5)
    nz = np.sqrt(np.maximum(1.0 - (tilt ** 2).sum(0), 0.0))
    normal = np.concatenate([0.5 + 0.5 * tilt, (0.5 + 0.5 * nz)[None], np.ones((1, size, size))]).transpose(1, 2, 0)
    arm = np.concatenate([0.6 + 0.4 * f[5:6], 0.2 + 0.7 * f[6:7], f[7:8], np.ones((1, size, size))]).transpose(1, 2, 0)
    q = lambda a: np.ascontiguousarray(np.clip(np.rint(a * 255.0), 0, 255).astype(np.uint8))
    return q(albedo), q(normal), q(arm)


def default_scene_info() -> np.ndarray:
    """SceneInfo of the reference's demo frame (Program.cs:84,120-132)."""
    s = np.zeros((), dtype=SCENE_DTYPE)
    cam = (2.0, 1.0, -4.0)
    s["camera_view"] = look_at(cam, (0, 0, 0), (0, 1, 0)).astype(np.float32).ravel()
    s["camera_projection"] = perspective(60.0, 16.0 / 9.0, 0.1, 100.0).astype(np.float32).ravel()  # Camera.cs:12,30-35
    s["camera_position"] = cam
    s["dir_light_direction"] = (2.0, -4.0, 1.0)
    s["dir_light_color"] = (0.35, 0.25, 0.2)
    view = look_at((-2.0, 4.0, -1.0), (0, 0, 0), (0, 1, 0))
    proj = ortho_off_center(-10.0, 10.0, 10.0, -10.0, 1.0, 7.5)
    s["light_space_matrix"] = (view @ proj).astype(np.float32).ravel()
    return s


def make_lights(rng: np.random.Generator, n: int) -> np.ndarray:
    """n LightInfo records: positions over the scene, colours hue * U[1,8] (so every light volume,
    radius 16*sqrt(max/5) >= 7.15, contains the whole scene and the camera)."""
    lights = np.zeros(n, dtype=LIGHT_DTYPE)
    pos = np.stack([rng.uniform(-3, 3, n), rng.uniform(0.5, 3, n), rng.uniform(-3, 3, n)], axis=1)
    hue = rng.uniform(0.2, 1.0, (n, 3))
    hue /= hue.max(axis=1, keepdims=True)
    col = hue * rng.uniform(1.0, 8.0, (n, 1))
    for i in range(n):
        r = light_volume_radius(col[i])
        m = np.diag([r, r, r, 1.0])
        m[3, :3] = pos[i]
        lights[i]["model"] = m.astype(np.float32).ravel()
    lights["position"] = pos.astype(np.float32)
    lights["color"] = col.astype(np.float32)
    return lights


@dataclasses.dataclass
class FrameInputs:
    width: int
    height: int
    albedo: np.ndarray      # uint16 [H, W, 4] binary16 bits
    normal: np.ndarray
    arm: np.ndarray
    position: np.ndarray
    scene: np.ndarray       # SCENE_DTYPE scalar
    lights: np.ndarray      # LIGHT_DTYPE [L]
    dir_shadow: np.ndarray  # float32 [R, R]
    cube_shadow: np.ndarray  # float32 [L, 6, S, S]
    irradiance: np.ndarray  # uint16 [6, n, n, 4]
    prefilter: np.ndarray   # uint16 [6, n, n, 4]
    brdf_lut: np.ndarray    # uint16 [n, n, 2]

    def rows(self, y0: int, y1: int) -> "FrameInputs":
        """The same frame restricted to rows [y0, y1) (strips share everything but the G-buffer)."""
        return dataclasses.replace(
            self, height=y1 - y0,
            albedo=np.ascontiguousarray(self.albedo[y0:y1]), normal=np.ascontiguousarray(self.normal[y0:y1]),
            arm=np.ascontiguousarray(self.arm[y0:y1]), position=np.ascontiguousarray(self.position[y0:y1]))


def make_gbuffer(rng: np.random.Generator, width: int, height: int, background_fraction: float = 0.1):
    """Height field y = 0.3 sin(1.7x) cos(1.3z) over [-3,3]^2, one texel per pixel; 8x8-pixel blocks of
    constant material with smooth variation on top; ~10% of the blocks are background (cleared texels,
    alpha 0: VulkanRenderer.cs:1155-1161)."""
    u = (np.arange(width, dtype=np.float32) + 0.5) / width
    v = (np.arange(height, dtype=np.float32) + 0.5) / height
    px = (-3.0 + 6.0 * u)[None, :].repeat(height, 0)
    pz = (-3.0 + 6.0 * v)[:, None].repeat(width, 1)
    py = 0.3 * np.sin(1.7 * px) * np.cos(1.3 * pz)
    dydx = 0.3 * 1.7 * np.cos(1.7 * px) * np.cos(1.3 * pz)
    dydz = -0.3 * 1.3 * np.sin(1.7 * px) * np.sin(1.3 * pz)
    n = np.stack([-dydx, np.ones_like(dydx), -dydz], axis=-1)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    n *= (0.9 + 0.2 * u)[None, :, None]  # deliberately not unit length: the shader renormalises

    bw, bh = (width + 7) // 8, (height + 7) // 8

    def blocks(lo, hi, ch):
        b = rng.uniform(lo, hi, (bh, bw, ch)).astype(np.float32)
        return b.repeat(8, 0).repeat(8, 1)[:height, :width]

    smooth = _smooth_field(rng, max(width, height), 8, 1)[0][:height, :width]
    albedo = np.clip(blocks(0.05, 0.95, 3) * (0.85 + 0.3 * smooth[..., None]), 0.0, 1.0)
    ao = blocks(0.5, 1.0, 1)[..., 0]
    rough = np.clip(blocks(0.05, 1.0, 1)[..., 0] * (0.9 + 0.2 * smooth), 0.03, 1.0)
    metal = np.where(blocks(0.0, 1.0, 1)[..., 0] > 0.7, 1.0, 0.0) * (0.8 + 0.2 * smooth)
    bg = (rng.random((bh, bw)) < background_fraction).repeat(8, 0).repeat(8, 1)[:height, :width]

    def rgba(rgb, a):
        out = np.concatenate([rgb, np.broadcast_to(np.asarray(a, dtype=np.float32), rgb.shape[:2])[..., None]], axis=-1)
        out = np.where(bg[..., None], 0.0, out).astype(np.float32)
        return _f16(out)

    return (rgba(albedo, 1.0), rgba(n, 1.0), rgba(np.stack([ao, rough, metal], -1), 1.0),
            rgba(np.stack([px, py, pz], -1), 1.0))


def make_cube_shadows(rng: np.random.Generator, n_lights: int, res: int) -> np.ndarray:
    """Per light a depth cube holding distance/25 with distance = 3 + 2*smooth noise."""
    out = np.empty((n_lights, 6, res, res), dtype=np.float32)
    for i in range(n_lights):
        out[i] = (3.0 + 2.0 * _smooth_field(rng, res, 12, 6)) / FAR_PLANE
    return out


def make_dir_shadow(rng: np.random.Generator, res: int) -> np.ndarray:
    return (0.2 + 0.7 * _smooth_field(rng, res, 24, 1)[0]).astype(np.float32)


def make_env_cube(rng: np.random.Generator, n: int, sun_peak: float = 50.0) -> np.ndarray:
    """RGBA16F environment cube [6, n, n, 4]: sky gradient + three Gaussian suns + smooth noise."""
    s = (2.0 * (np.arange(n, dtype=np.float32) + 0.5) / n - 1.0)
    ss, tt = np.meshgrid(s, s)  # ss varies along i (x), tt along j (y)
    one = np.ones_like(ss)
    dirs = [(one, -tt, -ss), (-one, -tt, ss), (ss, one, tt), (ss, -one, -tt), (ss, -tt, one), (-ss, -tt, -one)]
    suns = rng.normal(size=(3, 3))
    suns /= np.linalg.norm(suns, axis=1, keepdims=True)
    sun_col = rng.uniform(0.5, 1.0, (3, 3))
    noise = _smooth_field(rng, n, 8, 6)
    out = np.empty((6, n, n, 4), dtype=np.float32)
    for f, (x, y, z) in enumerate(dirs):
        inv = 1.0 / np.sqrt(x * x + y * y + z * z)
        x, y, z = x * inv, y * inv, z * inv
        sky = 0.15 + 0.6 * np.clip(0.5 + 0.5 * y, 0, 1)
        rgb = np.stack([sky * 0.7, sky * 0.85, sky * 1.1], -1) * (0.9 + 0.2 * noise[f][..., None])
        for k in range(3):
            c = x * suns[k, 0] + y * suns[k, 1] + z * suns[k, 2]
            rgb += sun_peak * np.exp((c - 1.0) * 400.0)[..., None] * sun_col[k]
        out[f, ..., :3] = rgb
        out[f, ..., 3] = 1.0
    return _f16(out)


def make_smooth_cube(rng: np.random.Generator, n: int, lo: float, hi: float) -> np.ndarray:
    """A smooth RGBA16F cube with values in [lo, hi] (stand-in for an irradiance / prefiltered cube)."""
    out = np.ones((6, n, n, 4), dtype=np.float32)
    for c in range(3):
        out[..., c] = lo + (hi - lo) * _smooth_field(rng, n, 4, 6)
    return _f16(out)


def make_smooth_lut(rng: np.random.Generator, n: int) -> np.ndarray:
    """A smooth RG16F image with the rough shape of the split-sum LUT (scale falls, bias small)."""
    a = 0.3 + 0.7 * _smooth_field(rng, n, 4, 1)[0]
    b = 0.2 * _smooth_field(rng, n, 4, 1)[0]
    return _f16(np.stack([a, b], -1))


def make_equirect(rng: np.random.Generator, w: int, h: int, hdr: bool) -> np.ndarray:
    """Equirectangular source: RGBA16F bits [h, w, 4] (hdr) or RGBA8 [h, w, 4] (the reference's 8-bit skybox)."""
    f = _smooth_field(rng, max(w, h), 16, 3)[:, :h, :w]
    rgb = np.stack([f[0], f[1], f[2]], -1)
    if hdr:
        rgb = rgb * 4.0
        return _f16(np.concatenate([rgb, np.ones((h, w, 1), np.float32)], -1))
    return np.ascontiguousarray(np.concatenate([np.round(rgb * 255.0), np.full((h, w, 1), 255.0)], -1).astype(np.uint8))


def make_frame(seed: int, width: int, height: int, n_lights: int, cube_res: int, dir_res: int,
               irr_res: int = 32, pref_res: int = 128, lut_res: int = 512,
               ibl: tuple | None = None) -> FrameInputs:
    """One frame's worth of inputs.  `ibl` = (irradiance, prefilter, brdf_lut) to use precomputed IBL
    images instead of smooth stand-ins."""
    rng = np.random.default_rng(seed)
    albedo, normal, arm, position = make_gbuffer(rng, width, height)
    lights = make_lights(rng, n_lights)
    dir_shadow = make_dir_shadow(rng, dir_res)
    cube_shadow = make_cube_shadows(rng, n_lights, cube_res)
    if ibl is None:
        ibl = (make_smooth_cube(rng, irr_res, 0.2, 1.5), make_smooth_cube(rng, pref_res, 0.1, 3.0), make_smooth_lut(rng, lut_res))
    return FrameInputs(width, height, albedo, normal, arm, position, default_scene_info(), lights,
                       dir_shadow, cube_shadow, ibl[0], ibl[1], ibl[2])
