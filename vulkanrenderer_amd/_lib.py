"""ctypes binding of libvkr_hip.so (include/vkr_hip.h).

The library is the product: there is no CPU fallback.  Loading fails loudly when the shared object
is missing (run `python -c "import __graft_entry__ as g; g.build()"` or `make -C vulkanrenderer_amd/csrc`).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VKR_HIP_LIB selects another build of the same library (kernel experiments); default: the in-tree one
LIB_PATH = os.environ.get("VKR_HIP_LIB") or os.path.join(_HERE, "libvkr_hip.so")

VKR_OK = 0
ABI_VERSION = 4  # VKR_ABI_VERSION of include/vkr_hip.h as this binding was written against
VKR_ERR_INVALID, VKR_ERR_HIP, VKR_ERR_UNSUPPORTED, VKR_ERR_ORDER, VKR_ERR_NO_DEVICE = -1, -2, -3, -4, -5
COVERAGE_FULLSCREEN, COVERAGE_MESH = 0, 1
CUBE_LAYOUT_LINEAR, CUBE_LAYOUT_TILED_8X4, CUBE_LAYOUT_TILED_8X4_X25 = 0, 1, 2
ENV_LAYOUT_LINEAR, ENV_LAYOUT_TILED_4X4, ENV_LAYOUT_FOOTPRINT_TILES = 0, 1, 2
WRAP_REPEAT, WRAP_CLAMP = 0, 1
SRC_RGBA8_SRGB, SRC_RGBA16F = 0, 1
POST_SRGB_TARGET = 1
SHADOW_RANGES_ALWAYS, SHADOW_RANGES_NEVER, SHADOW_RANGES_MIN_LIGHTS = 0x100, 0x200, 16
GBUFFER_STATIC_MAX_PIXELS, GBUFFER_STATIC_MAX_DRAWS = 2097152, 64
IRRADIANCE_SEQUENTIAL, IRRADIANCE_PHI_SLICES, IRRADIANCE_BY_REGION, IRRADIANCE_BY_REGION_ROWS = 0, 1, 2, 3
IRRADIANCE_FASTEST = -1
(TARGET_COLOR, TARGET_THRESHOLD, TARGET_BLOOM1, TARGET_BLOOM2, TARGET_LDR,
 TARGET_ENV, TARGET_IRRADIANCE, TARGET_PREFILTER, TARGET_BRDF_LUT) = range(9)
(STAGE_POINT_SHADOW, STAGE_GEOMETRY, STAGE_DIR_SHADOW, STAGE_COMPOSITION, STAGE_BLOOM1, STAGE_BLOOM2, STAGE_POSTPROCESS) = range(7)
STAGE_COUNT = 7
STRIP_HALO_ROWS, STRIP_ABOVE, STRIP_BELOW = 4, 0, 1


class VkrError(RuntimeError):
    """A libvkr_hip call returned a negative status (the reference throws `Exception` on any non-Success VkResult)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libvkr_hip error {code}: {message}")
        self.code = code


class SceneInfo(C.Structure):
    """VulkanRenderer.cs:11-20 (240 bytes)."""
    _fields_ = [("camera_view", C.c_float * 16), ("camera_projection", C.c_float * 16),
                ("camera_position", C.c_float * 3), ("_pad0", C.c_float),
                ("dir_light_direction", C.c_float * 3), ("_pad1", C.c_float),
                ("dir_light_color", C.c_float * 3), ("_pad2", C.c_float),
                ("light_space_matrix", C.c_float * 16)]


class LightInfo(C.Structure):
    """VulkanRenderer.cs:22-28 (92 bytes)."""
    _fields_ = [("model", C.c_float * 16), ("position", C.c_float * 3), ("_pad0", C.c_float), ("color", C.c_float * 3)]


class ShadeArgs(C.Structure):
    _fields_ = [("width", C.c_int32), ("rows", C.c_int32),
                ("albedo", C.c_void_p), ("normal", C.c_void_p), ("arm", C.c_void_p), ("position", C.c_void_p),
                ("background", C.c_void_p),
                ("scene", SceneInfo),
                ("lights", C.c_void_p), ("n_lights", C.c_int32), ("coverage_mode", C.c_int32),
                ("coverage_masks", C.c_void_p),
                ("dir_shadow", C.c_void_p), ("dir_res", C.c_int32),
                ("cube_shadow", C.c_void_p), ("cube_res", C.c_int32), ("cube_layout", C.c_int32),
                ("irradiance", C.c_void_p), ("irr_res", C.c_int32),
                ("prefilter", C.c_void_p), ("pref_res", C.c_int32),
                ("brdf_lut", C.c_void_p), ("lut_res", C.c_int32),
                ("out_color", C.c_void_p), ("out_threshold", C.c_void_p), ("coverage_flags", C.c_void_p),
                ("cube_ranges", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64)]


class CoverageArgs(C.Structure):
    _fields_ = [("width", C.c_int32), ("rows", C.c_int32), ("frame_rows", C.c_int32), ("row0", C.c_int32),
                ("scene", SceneInfo),
                ("lights", C.c_void_p), ("n_lights", C.c_int32),
                ("verts", C.c_void_p), ("n_verts", C.c_int32),
                ("indices", C.c_void_p), ("n_tris", C.c_int32),
                ("masks", C.c_void_p), ("light_flags", C.c_void_p)]


class MeshDraw(C.Structure):
    _fields_ = [("positions", C.c_void_p), ("n_verts", C.c_int32), ("indices", C.c_void_p), ("n_tris", C.c_int32),
                ("model", C.c_float * 16)]


class PointShadowArgs(C.Structure):
    _fields_ = [("draws", C.POINTER(MeshDraw)), ("n_draws", C.c_int32), ("lights", C.c_void_p), ("n_lights", C.c_int32),
                ("res", C.c_int32), ("far_plane", C.c_float), ("layout", C.c_int32), ("out", C.c_void_p)]


class DirShadowArgs(C.Structure):
    _fields_ = [("draws", C.POINTER(MeshDraw)), ("n_draws", C.c_int32), ("light_space", C.c_float * 16),
                ("res", C.c_int32), ("out", C.c_void_p)]


class Texture(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32)]


class GpassDraw(C.Structure):
    _fields_ = [("positions", C.c_void_p), ("texcoords", C.c_void_p), ("normals", C.c_void_p), ("tangents", C.c_void_p),
                ("n_verts", C.c_int32), ("indices", C.c_void_p), ("n_tris", C.c_int32), ("model", C.c_float * 16),
                ("albedo", Texture), ("normal", Texture), ("arm", Texture)]


class GpassArgs(C.Structure):
    _fields_ = [("draws", C.POINTER(GpassDraw)), ("n_draws", C.c_int32), ("scene", SceneInfo),
                ("width", C.c_int32), ("rows", C.c_int32), ("frame_rows", C.c_int32), ("row0", C.c_int32),
                ("albedo", C.c_void_p), ("normal", C.c_void_p), ("arm", C.c_void_p), ("position", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64)]


class BloomArgs(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("width", C.c_int32), ("rows", C.c_int32),
                ("src_rows", C.c_int32), ("src_row_offset", C.c_int32), ("horizontal", C.c_int32),
                ("wrap", C.c_int32), ("clamp_lo", C.c_int32), ("clamp_hi", C.c_int32)]


class BloomPostArgs(C.Structure):
    _fields_ = [("bloom1", C.c_void_p), ("hdr", C.c_void_p), ("out_rgba8", C.c_void_p),
                ("width", C.c_int32), ("rows", C.c_int32), ("src_rows", C.c_int32), ("src_row_offset", C.c_int32),
                ("wrap", C.c_int32), ("clamp_lo", C.c_int32), ("clamp_hi", C.c_int32), ("flags", C.c_int32)]


class BloomChainArgs(C.Structure):
    _fields_ = [("threshold", C.c_void_p), ("hdr", C.c_void_p), ("out_rgba8", C.c_void_p),
                ("width", C.c_int32), ("rows", C.c_int32), ("src_rows", C.c_int32), ("src_row_offset", C.c_int32),
                ("wrap_h", C.c_int32), ("clamp_lo_h", C.c_int32), ("clamp_hi_h", C.c_int32),
                ("wrap_v", C.c_int32), ("clamp_lo_v", C.c_int32), ("clamp_hi_v", C.c_int32), ("flags", C.c_int32)]


PREFILTER_MAX_MIPS = 8


class PrefilterMipsArgs(C.Structure):
    _fields_ = [("env", C.c_void_p), ("env_res", C.c_int32), ("env_layout", C.c_int32), ("base_res", C.c_int32), ("mips", C.c_int32),
                ("out", C.c_void_p * PREFILTER_MAX_MIPS), ("row_begin", C.c_int32 * PREFILTER_MAX_MIPS), ("row_end", C.c_int32 * PREFILTER_MAX_MIPS)]


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "device", "width", "height", "max_lights", "dir_shadow_res", "point_shadow_res", "env_res",
        "irradiance_res", "prefilter_res", "prefilter_mips", "brdf_lut_res", "frames_in_flight", "flags",
        "frame_rows", "row0")]


class Timing(C.Structure):
    _fields_ = [("shade_ms", C.c_float), ("bloom_h_ms", C.c_float), ("bloom_v_post_ms", C.c_float), ("frame_ms", C.c_float)]


assert C.sizeof(SceneInfo) == 240 and C.sizeof(LightInfo) == 92

_i32, _vp, _f = C.c_int32, C.c_void_p, C.c_float
# name -> argument types; every function returns int32 except vkr_last_error
PROTOTYPES = {
    "vkr_abi_version": [],
    "vkr_device_count": [],
    "vkr_shade": [C.POINTER(ShadeArgs), _vp],
    "vkr_tile_depth_cubes": [_vp, _vp, C.c_int32, C.c_int32, _vp],
    "vkr_relayout_depth_cubes": [_vp, _vp, _i32, _i32, _i32, _vp],
    "vkr_cube_shadow_ranges_bytes": [_i32, _i32],
    "vkr_shade_workspace_bytes": [_i32, _i32, _i32],
    "vkr_cube_shadow_ranges": [_vp, _i32, _i32, _i32, _vp, _vp],
    "vkr_light_coverage": [C.POINTER(CoverageArgs), _vp],
    "vkr_point_shadow_maps": [C.POINTER(PointShadowArgs), _vp],
    "vkr_gbuffer": [C.POINTER(GpassArgs), _vp],
    "vkr_gbuffer_static": [C.POINTER(GpassArgs), _vp],
    "vkr_skybox": [_vp, _i32, _i32, C.POINTER(SceneInfo), _i32, _i32, _i32, _i32, _vp, _vp],
    "vkr_gbuffer_workspace_bytes": [_i32, _i32, _i32],
    "vkr_gbuffer_workspace_bytes_with_triangles": [_i32, _i32, _i32, C.c_int64],
    "vkr_dir_shadow_map": [C.POINTER(DirShadowArgs), _vp],
    "vkr_bloom_blur": [C.POINTER(BloomArgs), _vp],
    "vkr_postprocess": [_vp, _vp, _vp, _i32, _i32, _i32, _vp],
    "vkr_bloomv_postprocess": [C.POINTER(BloomPostArgs), _vp],
    "vkr_bloom_postprocess": [C.POINTER(BloomChainArgs), _vp],
    "vkr_ibl_equirect_to_cube": [_vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _vp],
    "vkr_ibl_irradiance": [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp],
    "vkr_ibl_irradiance_by_region": [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp, C.c_int64, _vp],
    "vkr_ibl_irradiance_by_region_workspace_bytes": [_i32, _i32, _i32, _i32],
    "vkr_env_footprint_tiles_bytes": [_i32],
    "vkr_tile_rgba16f_footprints": [_vp, _vp, _i32, _vp],
    "vkr_ibl_irradiance_by_region_rows": [_vp, _i32, _vp, _i32, _i32, _i32, _vp, C.c_int64, _vp],
    "vkr_ibl_irradiance_by_region_rows_workspace_bytes": [_i32, _i32, _i32, _i32],
    "vkr_ibl_prefilter": [_vp, _i32, _i32, _vp, _i32, _f, _i32, _i32, _vp],
    "vkr_ibl_prefilter_mips": [_vp, _vp],
    "vkr_tile_rgba16f_faces": [_vp, _vp, _i32, _i32, _vp],
    "vkr_measure_peaks": [C.POINTER(C.c_float), C.POINTER(C.c_float)],
    "vkr_ibl_brdf_lut": [_vp, _i32, _i32, _i32, _vp],
    "vkr_create": [C.POINTER(Config), C.POINTER(_vp)],
    "vkr_destroy": [_vp],
    "vkr_resize": [_vp, _i32, _i32],
    "vkr_update_scene_info": [_vp, C.POINTER(SceneInfo)],
    "vkr_set_lights": [_vp, _vp, _i32, _i32],
    "vkr_set_light_volume_mesh": [_vp, _vp, _i32, _vp, _i32],
    "vkr_create_mesh": [_vp, _vp, _i32, _vp, _i32, C.POINTER(C.c_int32)],
    "vkr_set_mesh_attributes": [_vp, _i32, _vp, _vp, _vp],
    "vkr_create_material": [_vp, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _i32, C.POINTER(C.c_int32)],
    "vkr_draw_model": [_vp, _i32, _i32, _vp],
    "vkr_begin_frame": [_vp],
    "vkr_end_frame": [_vp],
    "vkr_current_frame": [_vp],
    "vkr_upload_gbuffer": [_vp, _vp, _vp, _vp, _vp],
    "vkr_upload_background": [_vp, _vp],
    "vkr_upload_dir_shadow": [_vp, _vp],
    "vkr_upload_point_shadow": [_vp, _i32, _vp],
    "vkr_create_skybox_cubemap": [_vp, _vp, _i32, _i32, _i32],
    "vkr_upload_env_cube": [_vp, _vp],
    "vkr_create_irradiance_cubemap": [_vp, _i32],
    "vkr_create_prefiltered_cubemap": [_vp],
    "vkr_create_brdf_lut": [_vp],
    "vkr_pass_composition": [_vp],
    "vkr_pass_bloom": [_vp, _i32],
    "vkr_stage_reset_commands": [_vp, _i32, _i32],
    "vkr_stage_begin_commands": [_vp, _i32, _i32],
    "vkr_stage_begin_render_pass": [_vp, _i32, _i32],
    "vkr_stage_draw": [_vp, _i32, _i32],
    "vkr_stage_end_render_pass": [_vp, _i32, _i32],
    "vkr_stage_end_commands": [_vp, _i32, _i32],
    "vkr_submit_frame": [_vp],
    "vkr_strip_plan": [_i32, _i32, _i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
    "vkr_strip_shade": [_vp],
    "vkr_strip_halo": [_vp, _i32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)],
    "vkr_strip_exchange": [_vp, _vp],
    "vkr_strip_finish": [_vp],
    "vkr_resize_strip": [_vp, _i32, _i32, _i32, _i32],
    "vkr_pass_postprocess": [_vp],
    "vkr_read": [_vp, _i32, _i32, _i32, _vp, C.c_int64],
    "vkr_sync": [_vp],
    "vkr_get_timing": [_vp, C.POINTER(Timing)],
    "vkr_selftest_math": [C.POINTER(C.c_uint64)],
}

_lib = None


RETURNS_INT64 = {"vkr_gbuffer_workspace_bytes", "vkr_gbuffer_workspace_bytes_with_triangles", "vkr_ibl_irradiance_by_region_workspace_bytes",
                 "vkr_ibl_irradiance_by_region_rows_workspace_bytes", "vkr_env_footprint_tiles_bytes",
                 "vkr_cube_shadow_ranges_bytes", "vkr_shade_workspace_bytes"}


def load() -> C.CDLL:
    """Loads libvkr_hip.so once and sets the prototypes of every entry point in include/vkr_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C vulkanrenderer_amd/csrc` "
                          "(there is no CPU fallback for the shading path)")
    lib = C.CDLL(LIB_PATH)
    lib.vkr_last_error.restype = C.c_char_p
    lib.vkr_last_error.argtypes = []
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = C.c_int64 if name in RETURNS_INT64 else C.c_int32
        fn.argtypes = argtypes
    if lib.vkr_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has interface version {lib.vkr_abi_version()}, this binding needs {ABI_VERSION}: rebuild the library")
    _lib = lib
    return lib


def check(code: int) -> int:
    if code < 0:
        raise VkrError(code, load().vkr_last_error().decode("utf-8", "replace"))
    return code
