"""ctypes wrapper of the CPU oracle (oracle/libvkr_oracle.so).  Test infrastructure: imported only by
tests, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libvkr_oracle.so")

WRAP_REPEAT, WRAP_CLAMP = 0, 1
SRC_RGBA8_SRGB, SRC_RGBA16F = 0, 1

_lib = None


class ShadeArgs(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32),
                ("albedo", C.c_void_p), ("normal", C.c_void_p), ("arm", C.c_void_p), ("position", C.c_void_p),
                ("background", C.c_void_p), ("scene", C.c_void_p),
                ("lights", C.c_void_p), ("n_lights", C.c_int32),
                ("dir_shadow", C.c_void_p), ("dir_res", C.c_int32),
                ("cube_shadow", C.c_void_p), ("cube_res", C.c_int32),
                ("irradiance", C.c_void_p), ("irr_res", C.c_int32),
                ("prefilter", C.c_void_p), ("pref_res", C.c_int32),
                ("brdf_lut", C.c_void_p), ("lut_res", C.c_int32),
                ("out_color", C.c_void_p), ("out_threshold", C.c_void_p), ("coverage", C.c_void_p)]


class CoverageArgs(C.Structure):
    _fields_ = [("width", C.c_int32), ("rows", C.c_int32), ("frame_rows", C.c_int32), ("row0", C.c_int32),
                ("scene", C.c_void_p), ("lights", C.c_void_p), ("n_lights", C.c_int32),
                ("verts", C.c_void_p), ("n_verts", C.c_int32),
                ("indices", C.c_void_p), ("n_tris", C.c_int32),
                ("masks", C.c_void_p)]


class MeshDraw(C.Structure):
    _fields_ = [("positions", C.c_void_p), ("n_verts", C.c_int32), ("indices", C.c_void_p), ("n_tris", C.c_int32),
                ("model", C.c_float * 16)]


class PointShadowArgs(C.Structure):
    _fields_ = [("draws", C.c_void_p), ("n_draws", C.c_int32), ("lights", C.c_void_p), ("n_lights", C.c_int32),
                ("res", C.c_int32), ("far_plane", C.c_float), ("out", C.c_void_p)]


class DirShadowArgs(C.Structure):
    _fields_ = [("draws", C.c_void_p), ("n_draws", C.c_int32), ("light_space", C.c_void_p), ("res", C.c_int32), ("out", C.c_void_p)]


class Texture(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32)]


class GpassDraw(C.Structure):
    _fields_ = [("positions", C.c_void_p), ("texcoords", C.c_void_p), ("normals", C.c_void_p), ("tangents", C.c_void_p),
                ("n_verts", C.c_int32), ("indices", C.c_void_p), ("n_tris", C.c_int32), ("model", C.c_float * 16),
                ("albedo", Texture), ("normal", Texture), ("arm", Texture)]


class GpassArgs(C.Structure):
    _fields_ = [("draws", C.c_void_p), ("n_draws", C.c_int32), ("scene", C.c_void_p),
                ("width", C.c_int32), ("rows", C.c_int32), ("frame_rows", C.c_int32), ("row0", C.c_int32),
                ("albedo", C.c_void_p), ("normal", C.c_void_p), ("arm", C.c_void_p), ("position", C.c_void_p)]


class SkyboxArgs(C.Structure):
    _fields_ = [("env", C.c_void_p), ("env_n", C.c_int32), ("scene", C.c_void_p),
                ("width", C.c_int32), ("rows", C.c_int32), ("frame_rows", C.c_int32), ("row0", C.c_int32), ("out", C.c_void_p)]


def build() -> None:
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_aces.restype = C.c_float
        _lib.orc_aces.argtypes = [C.c_float]
        _lib.orc_radical_inverse.restype = C.c_float
        _lib.orc_radical_inverse.argtypes = [C.c_uint32]
        _lib.orc_normal_distribution.restype = C.c_float
        _lib.orc_normal_distribution.argtypes = [C.c_float, C.c_float]
        _lib.orc_geometry_schlick.restype = C.c_float
        _lib.orc_geometry_schlick.argtypes = [C.c_float, C.c_float]
        _lib.orc_texcube_depth_nearest.restype = C.c_float
        _lib.orc_prefilter.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float]
    return _lib


def _p(a: np.ndarray | None) -> C.c_void_p:
    if a is None:
        return C.c_void_p(0)
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data)


def omp_threads() -> int:
    return load().orc_omp_threads()


def f32_to_f16(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    out = np.empty(a.shape, dtype=np.uint16)
    load().orc_f32_to_f16(_p(a), _p(out), C.c_int64(a.size))
    return out


def f16_to_f32(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint16)
    out = np.empty(a.shape, dtype=np.float32)
    load().orc_f16_to_f32(_p(a), _p(out), C.c_int64(a.size))
    return out


def light_coverage(scene: np.ndarray, lights: np.ndarray, verts: np.ndarray, indices: np.ndarray,
                   width: int, rows: int, frame_rows: int | None = None, row0: int = 0) -> np.ndarray:
    """Light-volume coverage masks uint64 [n_lights, ceil(rows/8), ceil(width/8)]: bit (y&7)*8 + (x&7)."""
    scene = np.ascontiguousarray(scene).reshape(1)
    lights = np.ascontiguousarray(lights)
    verts = np.ascontiguousarray(verts, dtype=np.float32)
    indices = np.ascontiguousarray(indices, dtype=np.uint32)
    masks = np.zeros((len(lights), (rows + 7) // 8, (width + 7) // 8), dtype=np.uint64)
    a = CoverageArgs()
    a.width, a.rows, a.frame_rows, a.row0 = width, rows, rows if frame_rows is None else frame_rows, row0
    a.scene, a.lights, a.n_lights = _p(scene), (_p(lights) if len(lights) else C.c_void_p(0)), len(lights)
    a.verts, a.n_verts = _p(verts), len(verts)
    a.indices, a.n_tris = _p(indices), len(indices) // 3 if indices.ndim == 1 else len(indices)
    a.masks = _p(masks)
    load().orc_light_coverage(C.byref(a))
    return masks


def coverage_to_image(masks: np.ndarray, width: int, rows: int) -> np.ndarray:
    """bool [n_lights, rows, width] from the tile masks."""
    n, ty, tx = masks.shape
    bits = (masks[..., None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)
    img = bits.reshape(n, ty, tx, 8, 8).transpose(0, 1, 3, 2, 4).reshape(n, ty * 8, tx * 8)
    return img[:, :rows, :width].astype(bool)


def light_clip_vertex(scene: np.ndarray, light: np.ndarray, v) -> np.ndarray:
    scene = np.ascontiguousarray(scene).reshape(1)
    light = np.ascontiguousarray(light).reshape(1)
    v = np.ascontiguousarray(v, dtype=np.float32)
    out = np.empty(4, dtype=np.float32)
    load().orc_light_clip_vertex(_p(scene), _p(light), _p(v), _p(out))
    return out


def _mesh_draws(draws):
    """draws: list of (positions [n,3] f32, triangles [m,3] u32, model [4,4] or 16 floats); returns (array, keepalive)."""
    arr = (MeshDraw * len(draws))()
    keep = []
    for k, (pos, tri, model) in enumerate(draws):
        pos = np.ascontiguousarray(pos, dtype=np.float32)
        tri = np.ascontiguousarray(tri, dtype=np.uint32)
        keep += [pos, tri]
        arr[k].positions, arr[k].n_verts = pos.ctypes.data, len(pos)
        arr[k].indices, arr[k].n_tris = tri.ctypes.data, len(tri)
        arr[k].model[:] = [float(v) for v in np.asarray(model, dtype=np.float32).ravel()]
    return arr, keep


def point_shadow_maps(draws, lights: np.ndarray, res: int, far_plane: float = 25.0) -> np.ndarray:
    """Depth cubes float32 [n_lights, 6, res, res] of the point-shadow pass (VulkanRenderer.cs:325-376)."""
    arr, keep = _mesh_draws(draws)
    lights = np.ascontiguousarray(lights)
    out = np.empty((len(lights), 6, res, res), dtype=np.float32)
    a = PointShadowArgs()
    a.draws, a.n_draws = C.addressof(arr), len(draws)
    a.lights, a.n_lights = (_p(lights) if len(lights) else C.c_void_p(0)), len(lights)
    a.res, a.far_plane, a.out = res, far_plane, _p(out)
    load().orc_point_shadow_maps(C.byref(a))
    return out


def dir_shadow_map(draws, light_space: np.ndarray, res: int) -> np.ndarray:
    """Depth map float32 [res, res] of the directional shadow pass (VulkanRenderer.cs:406-428)."""
    arr, keep = _mesh_draws(draws)
    ls = np.ascontiguousarray(light_space, dtype=np.float32).ravel()
    out = np.empty((res, res), dtype=np.float32)
    a = DirShadowArgs()
    a.draws, a.n_draws, a.light_space, a.res, a.out = C.addressof(arr), len(draws), _p(ls), res, _p(out)
    load().orc_dir_shadow_map(C.byref(a))
    return out


def shadow_face_matrix(eye, face: int, far_plane: float = 25.0) -> np.ndarray:
    eye = np.ascontiguousarray(eye, dtype=np.float32)
    out = np.empty(16, dtype=np.float32)
    load().orc_shadow_face_matrix(_p(eye), face, C.c_float(far_plane), _p(out))
    return out.reshape(4, 4)


def gbuffer(draws, scene: np.ndarray, width: int, rows: int, frame_rows: int | None = None, row0: int = 0):
    """Geometry pass (gpass.vert/.frag).  draws: list of dicts with positions, texcoords, normals, tangents,
    triangles, model and the RGBA8 images albedo, normal, arm [h, w, 4].  Returns (albedo, normal, arm,
    position) uint16 [rows, width, 4]."""
    arr = (GpassDraw * len(draws))()
    keep = []
    for k, d in enumerate(draws):
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        pos, uv, nrm, tan = f32(d["positions"]), f32(d["texcoords"]), f32(d["normals"]), f32(d["tangents"])
        tri = np.ascontiguousarray(d["triangles"], dtype=np.uint32)
        tex = [np.ascontiguousarray(d[n], dtype=np.uint8) for n in ("albedo", "normal", "arm")]
        keep += [pos, uv, nrm, tan, tri] + tex
        arr[k].positions, arr[k].texcoords, arr[k].normals, arr[k].tangents = (x.ctypes.data for x in (pos, uv, nrm, tan))
        arr[k].n_verts, arr[k].indices, arr[k].n_tris = len(pos), tri.ctypes.data, len(tri)
        arr[k].model[:] = [float(v) for v in np.asarray(d["model"], dtype=np.float32).ravel()]
        for field, t in zip(("albedo", "normal", "arm"), tex):
            getattr(arr[k], field).rgba8 = t.ctypes.data
            getattr(arr[k], field).width, getattr(arr[k], field).height = t.shape[1], t.shape[0]
    scene = np.ascontiguousarray(scene).reshape(1)
    out = [np.zeros((rows, width, 4), dtype=np.uint16) for _ in range(4)]
    a = GpassArgs()
    a.draws, a.n_draws, a.scene = C.addressof(arr), len(draws), _p(scene)
    a.width, a.rows, a.frame_rows, a.row0 = width, rows, rows if frame_rows is None else frame_rows, row0
    a.albedo, a.normal, a.arm, a.position = (_p(o) for o in out)
    load().orc_gbuffer(C.byref(a))
    return tuple(out)


def _one_gpass_draw(d):
    arr = (GpassDraw * 1)()
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    keep = [f32(d["positions"]), f32(d["texcoords"]), f32(d["normals"]), f32(d["tangents"])]
    arr[0].positions, arr[0].texcoords, arr[0].normals, arr[0].tangents = (x.ctypes.data for x in keep)
    arr[0].n_verts = len(keep[0])
    arr[0].model[:] = [float(v) for v in np.asarray(d["model"], dtype=np.float32).ravel()]
    for field in ("albedo", "normal", "arm"):
        if field in d:
            t = np.ascontiguousarray(d[field], dtype=np.uint8)
            keep.append(t)
            getattr(arr[0], field).rgba8 = t.ctypes.data
            getattr(arr[0], field).width, getattr(arr[0], field).height = t.shape[1], t.shape[0]
    return arr, keep


def gpass_vertex(scene: np.ndarray, d: dict, i: int):
    """gpass.vert for vertex i of a draw: (outPosition [4], gl_Position [4], TBN columns T,B,N [9])."""
    arr, keep = _one_gpass_draw(d)
    scene = np.ascontiguousarray(scene).reshape(1)
    pos, clip, tbn = np.empty(4, np.float32), np.empty(4, np.float32), np.empty(9, np.float32)
    load().orc_gpass_vertex(_p(scene), C.byref(arr[0]), C.c_uint32(i), _p(pos), _p(clip), _p(tbn))
    return pos, clip, tbn


def gpass_fragment(d: dict, uv, tbn):
    """gpass.frag for one fragment: (albedo [4], world normal [3], arm [4]) before the binary16 store."""
    arr, keep = _one_gpass_draw(d)
    tbn = np.ascontiguousarray(tbn, dtype=np.float32)
    alb, wn, arm = np.empty(4, np.float32), np.empty(3, np.float32), np.empty(4, np.float32)
    load().orc_gpass_fragment(C.byref(arr[0]), C.c_float(uv[0]), C.c_float(uv[1]), _p(tbn), _p(alb), _p(wn), _p(arm))
    return alb, wn, arm


def skybox(env: np.ndarray, scene: np.ndarray, width: int, rows: int, frame_rows: int | None = None, row0: int = 0) -> np.ndarray:
    """The skybox draw (cubemap.vert / skybox.frag): RGBA16F bits [rows, width, 4]."""
    env = np.ascontiguousarray(env, dtype=np.uint16)
    scene = np.ascontiguousarray(scene).reshape(1)
    out = np.empty((rows, width, 4), dtype=np.uint16)
    a = SkyboxArgs()
    a.env, a.env_n, a.scene = _p(env), env.shape[1], _p(scene)
    a.width, a.rows, a.frame_rows, a.row0, a.out = width, rows, rows if frame_rows is None else frame_rows, row0, _p(out)
    load().orc_skybox(C.byref(a))
    return out


def shade(f, background: np.ndarray | None = None, rows: tuple[int, int] | None = None, coverage: np.ndarray | None = None):
    """Composition pass on a synth.FrameInputs; returns (color, threshold) uint16 [H, W, 4].
    `coverage`: masks from light_coverage (None: every light covers every pixel)."""
    lib = load()
    color = np.zeros((f.height, f.width, 4), dtype=np.uint16)
    thr = np.zeros((f.height, f.width, 4), dtype=np.uint16)
    scene = np.ascontiguousarray(f.scene).reshape(1)
    lights = np.ascontiguousarray(f.lights)
    a = ShadeArgs()
    a.width, a.height = f.width, f.height
    a.albedo, a.normal, a.arm, a.position = _p(f.albedo), _p(f.normal), _p(f.arm), _p(f.position)
    a.background = _p(background)
    a.scene = _p(scene)
    a.lights, a.n_lights = (_p(lights) if len(lights) else C.c_void_p(0)), len(lights)
    a.dir_shadow, a.dir_res = _p(f.dir_shadow), f.dir_shadow.shape[-1]
    a.cube_shadow, a.cube_res = _p(f.cube_shadow), f.cube_shadow.shape[-1]
    a.irradiance, a.irr_res = _p(f.irradiance), f.irradiance.shape[1]
    a.prefilter, a.pref_res = _p(f.prefilter), f.prefilter.shape[1]
    a.brdf_lut, a.lut_res = _p(f.brdf_lut), f.brdf_lut.shape[0]
    a.out_color, a.out_threshold = _p(color), _p(thr)
    a.coverage = _p(None if coverage is None else np.ascontiguousarray(coverage, dtype=np.uint64))
    r0, r1 = (0, f.height) if rows is None else rows
    lib.orc_shade(C.byref(a), r0, r1)
    return color, thr


def bloom_blur(src: np.ndarray, horizontal: bool, wrap: int, true_bilinear: bool = False) -> np.ndarray:
    src = np.ascontiguousarray(src, dtype=np.uint16)
    h, w = src.shape[:2]
    dst = np.empty_like(src)
    load().orc_bloom_blur(_p(src), _p(dst), w, h, int(horizontal), wrap, int(true_bilinear))
    return dst


def postprocess(hdr: np.ndarray, bloom: np.ndarray, srgb_target: bool = False) -> np.ndarray:
    hdr = np.ascontiguousarray(hdr, dtype=np.uint16)
    bloom = np.ascontiguousarray(bloom, dtype=np.uint16)
    h, w = hdr.shape[:2]
    out = np.empty((h, w, 4), dtype=np.uint8)
    load().orc_postprocess(_p(hdr), _p(bloom), _p(out), w, h, int(srgb_target))
    return out


def frame_chain(f, background=None):
    """The whole per-frame chain (VulkanRenderer.cs:430-557): returns dict of every target."""
    color, thr = shade(f, background)
    b1 = bloom_blur(thr, True, WRAP_REPEAT)
    b2 = bloom_blur(b1, False, WRAP_CLAMP)
    ldr = postprocess(color, b2)
    return {"color": color, "threshold": thr, "bloom1": b1, "bloom2": b2, "ldr": ldr}


def equirect_to_cube(src: np.ndarray, n: int) -> np.ndarray:
    src = np.ascontiguousarray(src)
    fmt = SRC_RGBA8_SRGB if src.dtype == np.uint8 else SRC_RGBA16F
    cube = np.empty((6, n, n, 4), dtype=np.uint16)
    load().orc_equirect_to_cube(_p(src), src.shape[1], src.shape[0], fmt, _p(cube), n)
    return cube


def irradiance(env: np.ndarray, n: int, order: int = 0) -> np.ndarray:
    env = np.ascontiguousarray(env, dtype=np.uint16)
    out = np.empty((6, n, n, 4), dtype=np.uint16)
    load().orc_irradiance(_p(env), env.shape[1], _p(out), n, order)
    return out


def prefilter(env: np.ndarray, n: int, roughness: float) -> np.ndarray:
    env = np.ascontiguousarray(env, dtype=np.uint16)
    out = np.empty((6, n, n, 4), dtype=np.uint16)
    load().orc_prefilter(_p(env), env.shape[1], _p(out), n, C.c_float(roughness))
    return out


def brdf_lut(n: int) -> np.ndarray:
    out = np.empty((n, n, 2), dtype=np.uint16)
    load().orc_brdf_lut(_p(out), n)
    return out


def face_dir(f: int, i: int, j: int, n: int) -> np.ndarray:
    out = np.empty(3, dtype=np.float32)
    load().orc_face_dir(f, i, j, n, _p(out))
    return out


def cube_fold(n: int, f: int, i: int, j: int):
    out = np.empty(3, dtype=np.int32)
    ok = load().orc_cube_fold(n, f, i, j, _p(out))
    return (int(ok), tuple(int(v) for v in out))


def texcube_rgb_linear(cube: np.ndarray, d) -> np.ndarray:
    cube = np.ascontiguousarray(cube, dtype=np.uint16)
    d = np.ascontiguousarray(d, dtype=np.float32)
    out = np.empty(3, dtype=np.float32)
    load().orc_texcube_rgb_linear(_p(cube), cube.shape[1], _p(d), _p(out))
    return out


def texcube_depth_nearest(cube: np.ndarray, d) -> float:
    cube = np.ascontiguousarray(cube, dtype=np.float32)
    d = np.ascontiguousarray(d, dtype=np.float32)
    return float(load().orc_texcube_depth_nearest(_p(cube), cube.shape[1], _p(d)))


def tex2d_rg_linear_repeat(img: np.ndarray, s: float, t: float) -> np.ndarray:
    img = np.ascontiguousarray(img, dtype=np.uint16)
    out = np.empty(2, dtype=np.float32)
    load().orc_tex2d_rg_linear_repeat(_p(img), img.shape[1], img.shape[0], C.c_float(s), C.c_float(t), _p(out))
    return out


def equirect_uv(d) -> np.ndarray:
    d = np.ascontiguousarray(d, dtype=np.float32)
    out = np.empty(2, dtype=np.float32)
    load().orc_equirect_uv(_p(d), _p(out))
    return out


def point_shadow_vertex(model: np.ndarray, face_matrix: np.ndarray, pos: np.ndarray):
    """point_shadow.vert + point_shadow.geom for one vertex and one face: (world [4], clip [4])."""
    m = np.ascontiguousarray(model, dtype=np.float32).ravel()
    fm = np.ascontiguousarray(face_matrix, dtype=np.float32).ravel()
    p = np.ascontiguousarray(pos, dtype=np.float32)
    world, clip = np.zeros(4, np.float32), np.zeros(4, np.float32)
    load().orc_point_shadow_vertex(_p(m), _p(fm), _p(p), _p(world), _p(clip))
    return world, clip


def point_shadow_depth(world: np.ndarray, light_pos: np.ndarray, far_plane: float) -> float:
    """point_shadow.frag for one fragment."""
    lib = load()
    lib.orc_point_shadow_depth.restype = C.c_float
    w, l = np.ascontiguousarray(world, dtype=np.float32), np.ascontiguousarray(light_pos, dtype=np.float32)
    return float(lib.orc_point_shadow_depth(_p(w), _p(l), C.c_float(far_plane)))


def depth_vertex(model: np.ndarray, light_space: np.ndarray, pos: np.ndarray) -> np.ndarray:
    """depth.vert for one vertex: clip position [4]."""
    m = np.ascontiguousarray(model, dtype=np.float32).ravel()
    ls = np.ascontiguousarray(light_space, dtype=np.float32).ravel()
    p = np.ascontiguousarray(pos, dtype=np.float32)
    clip = np.zeros(4, np.float32)
    load().orc_depth_vertex(_p(m), _p(ls), _p(p), _p(clip))
    return clip
