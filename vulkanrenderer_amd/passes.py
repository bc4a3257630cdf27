"""Layer-1 passes of libvkr_hip.so on torch-owned device memory.

torch is used for what it is good at here -- allocating HBM, streams, and `torch.distributed` --
while every pass is one C-ABI call into the hand-written kernels, launched on torch's current
stream.  RGBA16F images are float16 tensors [rows, width, 4]; depth maps float32; the final image
uint8 [rows, width, 4]; LightInfo arrays uint8 [n * 92].
"""
from __future__ import annotations

import ctypes as C
import dataclasses

import numpy as np
import torch

from . import _lib
from .synth import FrameInputs, SCENE_DTYPE


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: torch.Tensor | None) -> C.c_void_p:
    if t is None:
        return C.c_void_p(0)
    assert t.is_cuda and t.is_contiguous(), "passes take contiguous device tensors"
    return C.c_void_p(t.data_ptr())


def to_device_f16(bits: np.ndarray, device) -> torch.Tensor:
    """binary16 bit patterns (uint16) -> float16 device tensor of the same shape."""
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.float16)).to(device)


def f16_bits(t: torch.Tensor) -> np.ndarray:
    """float16 device tensor -> uint16 bit patterns on the host."""
    return t.detach().cpu().numpy().view(np.uint16)


def _cube_layout_for(tiled: bool | None, res: int) -> int:
    if tiled is None:
        return _lib.CUBE_LAYOUT_TILED_8X4_X25 if res % 8 == 0 else _lib.CUBE_LAYOUT_LINEAR
    return _lib.CUBE_LAYOUT_TILED_8X4 if tiled else _lib.CUBE_LAYOUT_LINEAR


@dataclasses.dataclass
class DeviceFrame:
    """A frame's inputs resident in HBM (what the geometry / shadow passes would have left there)."""
    width: int
    rows: int
    albedo: torch.Tensor
    normal: torch.Tensor
    arm: torch.Tensor
    position: torch.Tensor
    scene: np.ndarray
    lights: torch.Tensor
    n_lights: int
    dir_shadow: torch.Tensor
    cube_shadow: torch.Tensor
    irradiance: torch.Tensor
    prefilter: torch.Tensor
    brdf_lut: torch.Tensor
    background: torch.Tensor | None = None
    cube_layout: int = _lib.CUBE_LAYOUT_LINEAR
    coverage: torch.Tensor | None = None  # masks from light_coverage(); None: every light covers every pixel
    coverage_flags: torch.Tensor | None = None  # per-light "covers every pixel" words of the same light_coverage() call
    cube_ranges: torch.Tensor | None = None  # cube_shadow_ranges(cube_shadow, cube_layout), or None: every tap is read
    # the per-tile light classes, written by every shade() of a frame with cube_ranges: made on first use; two shade()
    # calls that may overlap on the device (frames in flight on different streams) need a workspace each
    shade_workspace: torch.Tensor | None = None
    # whether shade() sorts the lights per tile by the ranges first: None = when there are at least
    # _lib.SHADOW_RANGES_MIN_LIGHTS lights (with few, the sorting pass costs more than the taps it saves)
    tile_classes: bool | None = None

    @staticmethod
    def upload(f: FrameInputs, device, tiled: bool | None = None, ranges: bool = True) -> "DeviceFrame":
        """`tiled`: how the point-shadow cubes are stored -- None: the layout the shading kernel is fastest on
        (8x4-texel tiles of 25 * depth) whenever the face size allows; True: 8x4-texel tiles; False: rows.
        `ranges`: also build the depth ranges of the cubes, with which the shading kernel settles most lights of most
        pixels without reading their 20 taps (same image either way)."""
        d = DeviceFrame._upload_linear(f, device)
        layout = _cube_layout_for(tiled, d.cube_shadow.shape[-1])
        if layout != _lib.CUBE_LAYOUT_LINEAR and d.n_lights > 0:
            d.cube_shadow = tile_depth_cubes(d.cube_shadow, layout)
            d.cube_layout = layout
        if ranges and d.n_lights > 0:
            d.cube_ranges = cube_shadow_ranges(d.cube_shadow, d.cube_layout)
        return d

    @staticmethod
    def _upload_linear(f: FrameInputs, device) -> "DeviceFrame":
        lights = torch.from_numpy(np.frombuffer(f.lights.tobytes(), dtype=np.uint8).copy()).to(device)
        return DeviceFrame(
            f.width, f.height,
            to_device_f16(f.albedo, device), to_device_f16(f.normal, device),
            to_device_f16(f.arm, device), to_device_f16(f.position, device),
            f.scene, lights, len(f.lights),
            torch.from_numpy(f.dir_shadow).to(device), torch.from_numpy(f.cube_shadow).to(device),
            to_device_f16(f.irradiance, device), to_device_f16(f.prefilter, device), to_device_f16(f.brdf_lut, device))


def light_coverage(f: DeviceFrame, verts: torch.Tensor, indices: torch.Tensor,
                   frame_rows: int | None = None, row0: int = 0, flags: torch.Tensor | None = None) -> torch.Tensor:
    """Light-volume coverage of the frame's lights: int64 tensor [n_lights, ceil(rows/8), ceil(width/8)] of
    64-bit tile masks (bit (y&7)*8 + (x&7)).  `verts` [n,3] float32, `indices` [m,3] int32 on the device;
    `frame_rows` / `row0` place a strip inside the whole frame.  `flags`: int32 tensor of n_lights + 1 words that
    receives the per-light "covers every pixel" decisions (see light_coverage_with_flags)."""
    lib = _lib.load()
    masks = torch.empty((f.n_lights, (f.rows + 7) // 8, (f.width + 7) // 8), dtype=torch.int64, device=f.albedo.device)
    a = _lib.CoverageArgs()
    a.width, a.rows, a.frame_rows, a.row0 = f.width, f.rows, f.rows if frame_rows is None else frame_rows, row0
    a.scene = _scene_struct(f.scene)
    a.lights, a.n_lights = _ptr(f.lights), f.n_lights
    a.verts, a.n_verts = _ptr(verts), verts.shape[0]
    a.indices, a.n_tris = _ptr(indices), indices.shape[0]
    a.masks = _ptr(masks)
    a.light_flags = _ptr(flags)
    _lib.check(lib.vkr_light_coverage(C.byref(a), _stream()))
    return masks


def light_coverage_with_flags(f: DeviceFrame, verts: torch.Tensor, indices: torch.Tensor,
                              frame_rows: int | None = None, row0: int = 0) -> tuple[torch.Tensor, torch.Tensor]:
    """light_coverage() plus the per-light words: flags[i] = 1 when light i provably covers every pixel (its masks
    are then all ones, written without rasterising), flags[n_lights] = the number of lights with a 0.  A frame that
    carries both (DeviceFrame.coverage, .coverage_flags) is shaded without reading the masks when that number is 0."""
    flags = torch.empty((f.n_lights + 1,), dtype=torch.int32, device=f.albedo.device)
    return light_coverage(f, verts, indices, frame_rows, row0, flags), flags


def _mesh_draws(draws):
    """draws: list of (positions [n,3] float32, triangles [m,3] int32, model 16 floats), tensors on the device."""
    arr = (_lib.MeshDraw * len(draws))()
    for k, (pos, tri, model) in enumerate(draws):
        assert pos.dtype == torch.float32 and tri.dtype == torch.int32 and pos.is_contiguous() and tri.is_contiguous()
        arr[k].positions, arr[k].n_verts = pos.data_ptr(), pos.shape[0]
        arr[k].indices, arr[k].n_tris = tri.data_ptr(), tri.shape[0]
        arr[k].model[:] = [float(v) for v in np.asarray(model, dtype=np.float32).ravel()]
    return arr


def point_shadow_maps(draws, lights: torch.Tensor, n_lights: int, res: int, far_plane: float = 25.0,
                      tiled: bool | None = None) -> tuple[torch.Tensor, int]:
    """The point-shadow passes: (depth cubes float32 [n_lights, 6, res, res], layout).  `lights` is the
    device LightInfo array of a DeviceFrame; `tiled` as in DeviceFrame.upload."""
    lib = _lib.load()
    out = torch.empty((n_lights, 6, res, res), dtype=torch.float32, device=lights.device)
    arr = _mesh_draws(draws)
    a = _lib.PointShadowArgs()
    a.draws, a.n_draws = arr, len(draws)
    a.lights, a.n_lights = _ptr(lights), n_lights
    a.res, a.far_plane = res, far_plane
    a.layout = _cube_layout_for(tiled, res)
    a.out = _ptr(out)
    _lib.check(lib.vkr_point_shadow_maps(C.byref(a), _stream()))
    return out, a.layout


def dir_shadow_map(draws, light_space, res: int, device) -> torch.Tensor:
    """The directional depth pass: float32 [res, res]."""
    lib = _lib.load()
    out = torch.empty((res, res), dtype=torch.float32, device=device)
    arr = _mesh_draws(draws)
    a = _lib.DirShadowArgs()
    a.draws, a.n_draws = arr, len(draws)
    a.light_space[:] = [float(v) for v in np.asarray(light_space, dtype=np.float32).ravel()]
    a.res, a.out = res, _ptr(out)
    _lib.check(lib.vkr_dir_shadow_map(C.byref(a), _stream()))
    return out


def gbuffer(draws, scene: np.ndarray, width: int, rows: int, device, frame_rows: int | None = None, row0: int = 0,
            triangle_records: bool = True):
    """The geometry pass: (albedo, normal, arm, position) float16 [rows, width, 4].  draws: list of dicts of
    device tensors -- positions [n,3], texcoords [n,2], normals [n,3], tangents [n,3] float32, triangles [m,3]
    int32, the RGBA8 images albedo / normal / arm [h,w,4] uint8 -- and `model` (16 floats).
    `triangle_records`: give the call the larger workspace with which it shades every triangle's corners once
    (False: the smallest workspace; the resolve step then shades the corners per pixel -- same image)."""
    lib = _lib.load()
    arr = (_lib.GpassDraw * len(draws))()
    for k, d in enumerate(draws):
        for name in ("positions", "texcoords", "normals", "tangents"):
            t = d[name]
            assert t.dtype == torch.float32 and t.is_contiguous()
            setattr(arr[k], name, t.data_ptr())
        tri = d["triangles"]
        assert tri.dtype == torch.int32 and tri.is_contiguous()
        arr[k].n_verts, arr[k].indices, arr[k].n_tris = d["positions"].shape[0], tri.data_ptr(), tri.shape[0]
        arr[k].model[:] = [float(v) for v in np.asarray(d["model"], dtype=np.float32).ravel()]
        for name in ("albedo", "normal", "arm"):
            t = d[name]
            assert t.dtype == torch.uint8 and t.is_contiguous() and t.shape[-1] == 4
            tex = getattr(arr[k], name)
            tex.rgba8, tex.width, tex.height = t.data_ptr(), t.shape[1], t.shape[0]
    out = [torch.empty((rows, width, 4), dtype=torch.float16, device=device) for _ in range(4)]
    # with room for a record per triangle: the corners are shaded once per triangle instead of three times per pixel
    if triangle_records:
        need = lib.vkr_gbuffer_workspace_bytes_with_triangles(width, rows, len(draws), sum(int(d["triangles"].shape[0]) for d in draws))
    else:
        need = lib.vkr_gbuffer_workspace_bytes(width, rows, len(draws))
    workspace = torch.empty(need, dtype=torch.uint8, device=device)
    a = _lib.GpassArgs()
    a.draws, a.n_draws = arr, len(draws)
    a.scene = _scene_struct(scene)
    a.width, a.rows, a.frame_rows, a.row0 = width, rows, rows if frame_rows is None else frame_rows, row0
    a.albedo, a.normal, a.arm, a.position = (_ptr(o) for o in out)
    a.workspace, a.workspace_bytes = _ptr(workspace), need
    _lib.check(lib.vkr_gbuffer(C.byref(a), _stream()))
    return tuple(out)


def skybox(env: torch.Tensor, scene: np.ndarray, width: int, rows: int, frame_rows: int | None = None, row0: int = 0,
           env_layout: int = _lib.ENV_LAYOUT_LINEAR) -> torch.Tensor:
    """The skybox draw of the composition pass: float16 [rows, width, 4], the `background` of a DeviceFrame."""
    lib = _lib.load()
    out = torch.empty((rows, width, 4), dtype=torch.float16, device=env.device)
    sc = _scene_struct(scene)
    _lib.check(lib.vkr_skybox(_ptr(env), env.shape[1], env_layout, C.byref(sc), width, rows,
                              rows if frame_rows is None else frame_rows, row0, _ptr(out), _stream()))
    return out


def tile_depth_cubes(linear: torch.Tensor, layout: int = _lib.CUBE_LAYOUT_TILED_8X4) -> torch.Tensor:
    """[..., res, res] float32 depth faces, row-major -> the same faces as 8x4-texel tiles (new tensor), holding
    the depth (CUBE_LAYOUT_TILED_8X4) or 25 * depth (CUBE_LAYOUT_TILED_8X4_X25)."""
    lib = _lib.load()
    res = linear.shape[-1]
    out = torch.empty_like(linear)
    _lib.check(lib.vkr_relayout_depth_cubes(_ptr(linear), _ptr(out), linear.numel() // (res * res), res, layout, _stream()))
    return out


def cube_shadow_ranges(cubes: torch.Tensor, layout: int, out: torch.Tensor | None = None) -> torch.Tensor:
    """Depth ranges of the point-shadow cubes [n_lights, 6, res, res] stored in `layout` (DeviceFrame.cube_ranges):
    a uint8 tensor [n_lights, bytes per light] of vkr_cube_shadow_ranges_bytes() bytes.  Rebuild whenever the cubes change."""
    lib = _lib.load()
    n_lights, res = cubes.shape[0], cubes.shape[-1]
    need = lib.vkr_cube_shadow_ranges_bytes(n_lights, res)
    if need < 0:
        raise ValueError(f"cube_shadow_ranges: bad size ({n_lights} lights, faces of {res})")
    if out is None:
        out = torch.empty((n_lights, need // max(n_lights, 1)), dtype=torch.uint8, device=cubes.device)  # one row per light
    assert out.numel() * out.element_size() >= need
    _lib.check(lib.vkr_cube_shadow_ranges(_ptr(cubes), n_lights, res, layout, _ptr(out), _stream()))
    return out


def _scene_struct(scene: np.ndarray) -> _lib.SceneInfo:
    assert scene.dtype == SCENE_DTYPE
    return _lib.SceneInfo.from_buffer_copy(scene.tobytes())


def shade_workspace_bytes(width: int, rows: int, n_lights: int) -> int:
    """Size of DeviceFrame.shade_workspace for a strip of this shape."""
    return int(_lib.load().vkr_shade_workspace_bytes(width, rows, n_lights))


def shade(f: DeviceFrame, out_color: torch.Tensor, out_threshold: torch.Tensor) -> None:
    """Composition pass (composition.frag + one light.frag draw per light) over the frame's rows."""
    lib = _lib.load()
    a = _lib.ShadeArgs()
    a.width, a.rows = f.width, f.rows
    a.albedo, a.normal, a.arm, a.position = _ptr(f.albedo), _ptr(f.normal), _ptr(f.arm), _ptr(f.position)
    a.background = _ptr(f.background)
    a.scene = _scene_struct(f.scene)
    a.lights, a.n_lights = _ptr(f.lights), f.n_lights
    a.coverage_mode = _lib.COVERAGE_FULLSCREEN if f.coverage is None else _lib.COVERAGE_MESH
    a.coverage_masks = _ptr(f.coverage)
    a.coverage_flags = _ptr(f.coverage_flags) if f.coverage is not None else None
    a.dir_shadow, a.dir_res = _ptr(f.dir_shadow), f.dir_shadow.shape[-1]
    a.cube_shadow, a.cube_res, a.cube_layout = _ptr(f.cube_shadow), f.cube_shadow.shape[-1], f.cube_layout
    a.cube_ranges = _ptr(f.cube_ranges)
    classes = f.n_lights >= _lib.SHADOW_RANGES_MIN_LIGHTS if f.tile_classes is None else f.tile_classes
    if f.cube_ranges is not None and f.n_lights > 0 and classes:
        need = lib.vkr_shade_workspace_bytes(f.width, f.rows, f.n_lights)
        if f.shade_workspace is None or f.shade_workspace.numel() < need:
            f.shade_workspace = torch.empty(need, dtype=torch.uint8, device=f.albedo.device)
        a.workspace, a.workspace_bytes = _ptr(f.shade_workspace), f.shade_workspace.numel()
    a.irradiance, a.irr_res = _ptr(f.irradiance), f.irradiance.shape[1]
    a.prefilter, a.pref_res = _ptr(f.prefilter), f.prefilter.shape[1]
    a.brdf_lut, a.lut_res = _ptr(f.brdf_lut), f.brdf_lut.shape[0]
    a.out_color, a.out_threshold = _ptr(out_color), _ptr(out_threshold)
    _lib.check(lib.vkr_shade(C.byref(a), _stream()))


def bloom_blur(src: torch.Tensor, dst: torch.Tensor, horizontal: bool, wrap: int, *,
               rows: int | None = None, src_row_offset: int = 0,
               clamp_lo: int | None = None, clamp_hi: int | None = None) -> None:
    """One bloom_blur.frag draw.  Defaults blur a whole image onto an image of the same size."""
    lib = _lib.load()
    a = _lib.BloomArgs()
    a.src, a.dst = _ptr(src), _ptr(dst)
    a.width = src.shape[1]
    a.rows = dst.shape[0] if rows is None else rows
    a.src_rows, a.src_row_offset = src.shape[0], src_row_offset
    a.horizontal, a.wrap = int(bool(horizontal)), wrap
    n = a.width if horizontal else a.src_rows
    a.clamp_lo = 0 if clamp_lo is None else clamp_lo
    a.clamp_hi = n - 1 if clamp_hi is None else clamp_hi
    _lib.check(lib.vkr_bloom_blur(C.byref(a), _stream()))


def postprocess(hdr: torch.Tensor, bloom: torch.Tensor, out: torch.Tensor, flags: int = 0) -> None:
    lib = _lib.load()
    _lib.check(lib.vkr_postprocess(_ptr(hdr), _ptr(bloom), _ptr(out), hdr.shape[1], hdr.shape[0], flags, _stream()))


def bloomv_postprocess(bloom1: torch.Tensor, hdr: torch.Tensor, out: torch.Tensor, flags: int = 0, *,
                       src_row_offset: int = 0, clamp_lo: int | None = None, clamp_hi: int | None = None,
                       wrap: int = _lib.WRAP_CLAMP) -> None:
    """Bloom pass 2 fused with the post-process draw."""
    lib = _lib.load()
    a = _lib.BloomPostArgs()
    a.bloom1, a.hdr, a.out_rgba8 = _ptr(bloom1), _ptr(hdr), _ptr(out)
    a.width, a.rows = hdr.shape[1], hdr.shape[0]
    a.src_rows, a.src_row_offset = bloom1.shape[0], src_row_offset
    a.wrap = wrap
    a.clamp_lo = 0 if clamp_lo is None else clamp_lo
    a.clamp_hi = bloom1.shape[0] - 1 if clamp_hi is None else clamp_hi
    a.flags = flags
    _lib.check(lib.vkr_bloomv_postprocess(C.byref(a), _stream()))


def bloom_postprocess(threshold: torch.Tensor, hdr: torch.Tensor, out: torch.Tensor, flags: int = 0, *,
                      src_row_offset: int = 0, wrap_h: int = _lib.WRAP_REPEAT, wrap_v: int = _lib.WRAP_CLAMP,
                      clamp_h: tuple[int, int] | None = None, clamp_v: tuple[int, int] | None = None) -> None:
    """Both bloom passes and the post-process draw in one kernel.  `threshold` may carry halo rows: output row r
    is centred on its row r + src_row_offset; `clamp_v` is the valid range of its rows (default: all)."""
    lib = _lib.load()
    a = _lib.BloomChainArgs()
    a.threshold, a.hdr, a.out_rgba8 = _ptr(threshold), _ptr(hdr), _ptr(out)
    a.width, a.rows = hdr.shape[1], hdr.shape[0]
    a.src_rows, a.src_row_offset = threshold.shape[0], src_row_offset
    a.wrap_h, a.wrap_v = wrap_h, wrap_v
    a.clamp_lo_h, a.clamp_hi_h = (0, a.width - 1) if clamp_h is None else clamp_h
    a.clamp_lo_v, a.clamp_hi_v = (0, a.src_rows - 1) if clamp_v is None else clamp_v
    a.flags = flags
    _lib.check(lib.vkr_bloom_postprocess(C.byref(a), _stream()))


def _rows(n_rows_total: int, rows: tuple[int, int] | None) -> tuple[int, int]:
    return (0, n_rows_total) if rows is None else rows


def equirect_to_cube(src: torch.Tensor, src_format: int, cube: torch.Tensor, rows=None) -> None:
    lib = _lib.load()
    n = cube.shape[1]
    r0, r1 = _rows(6 * n, rows)
    _lib.check(lib.vkr_ibl_equirect_to_cube(_ptr(src), src.shape[1], src.shape[0], src_format, _ptr(cube), n, r0, r1, _stream()))


def tile_env_cube(linear: torch.Tensor) -> torch.Tensor:
    """[6, n, n, 4] float16 cube, row-major -> the same cube as 4x4-texel tiles (new tensor, same shape)."""
    lib = _lib.load()
    out = torch.empty_like(linear)
    _lib.check(lib.vkr_tile_rgba16f_faces(_ptr(linear), _ptr(out), linear.shape[0], linear.shape[1], _stream()))
    return out


def tile_env_footprints(linear: torch.Tensor) -> torch.Tensor:
    """[6, n, n, 4] float16 cube, row-major -> the cube in footprint tiles (VKR_ENV_LAYOUT_FOOTPRINT_TILES): a flat
    uint8 tensor of vkr_env_footprint_tiles_bytes(n) bytes."""
    lib = _lib.load()
    n = linear.shape[1]
    need = lib.vkr_env_footprint_tiles_bytes(n)
    if need < 0:
        _lib.check(_lib.VKR_ERR_INVALID)
    out = torch.empty(need, dtype=torch.uint8, device=linear.device)
    _lib.check(lib.vkr_tile_rgba16f_footprints(_ptr(linear), _ptr(out), n, _stream()))
    return out


def irradiance(env: torch.Tensor, out: torch.Tensor, order: int = _lib.IRRADIANCE_SEQUENTIAL, rows=None,
               env_layout: int = _lib.ENV_LAYOUT_LINEAR) -> None:
    """One of the summation orders of the irradiance pass.  The two by-region orders allocate their workspace here
    (the rows order also its footprint tiles, from a row-major `env`) and check that every sample was accounted for,
    which synchronises; a caller that wants neither uses irradiance_by_region() / irradiance_by_region_rows()."""
    if order == _lib.IRRADIANCE_FASTEST:
        order = _lib.IRRADIANCE_BY_REGION_ROWS
    if order == _lib.IRRADIANCE_BY_REGION:
        ws = irradiance_by_region(env, out, rows, env_layout)
        _require_all_samples(ws, "VKR_IRRADIANCE_BY_REGION")
        return
    if order == _lib.IRRADIANCE_BY_REGION_ROWS:
        if env_layout == _lib.ENV_LAYOUT_FOOTPRINT_TILES:
            raise ValueError("pass the row-major cube here, or call irradiance_by_region_rows() with the tiles and the face size")
        if env_layout != _lib.ENV_LAYOUT_LINEAR:
            raise ValueError("the rows order reads footprint tiles, which are made from the row-major cube")
        ws = irradiance_by_region_rows(tile_env_footprints(env), env.shape[1], out, rows)
        _require_all_samples(ws, "VKR_IRRADIANCE_BY_REGION_ROWS")
        return
    lib = _lib.load()
    n = out.shape[1]
    r0, r1 = _rows(6 * n, rows)
    _lib.check(lib.vkr_ibl_irradiance(_ptr(env), env.shape[1], env_layout, _ptr(out), n, order, r0, r1, _stream()))


def irradiance_by_region(env: torch.Tensor, out: torch.Tensor, rows=None, env_layout: int = _lib.ENV_LAYOUT_LINEAR,
                         workspace: torch.Tensor | None = None) -> torch.Tensor:
    """The irradiance cube summed region by region of the environment (VKR_IRRADIANCE_BY_REGION).  Returns the
    workspace (a uint8 tensor, allocated here unless given); irradiance_uncovered() reads its check word."""
    lib = _lib.load()
    n = out.shape[1]
    r0, r1 = _rows(6 * n, rows)
    need = lib.vkr_ibl_irradiance_by_region_workspace_bytes(env.shape[1], n, r0, r1)
    if need < 0:
        _lib.check(_lib.VKR_ERR_INVALID)
    if workspace is None:
        workspace = torch.empty(need, dtype=torch.uint8, device=out.device)
    _lib.check(lib.vkr_ibl_irradiance_by_region(_ptr(env), env.shape[1], env_layout, _ptr(out), n, r0, r1,
                                                _ptr(workspace), workspace.numel(), _stream()))
    return workspace


def irradiance_by_region_rows(env_tiles: torch.Tensor, env_n: int, out: torch.Tensor, rows=None,
                              workspace: torch.Tensor | None = None) -> torch.Tensor:
    """The irradiance cube in the VKR_IRRADIANCE_BY_REGION_ROWS order from the environment in footprint tiles
    (tile_env_footprints).  Returns the workspace; irradiance_uncovered() reads its check word."""
    lib = _lib.load()
    n = out.shape[1]
    r0, r1 = _rows(6 * n, rows)
    need = lib.vkr_ibl_irradiance_by_region_rows_workspace_bytes(env_n, n, r0, r1)
    if need < 0:
        _lib.check(_lib.VKR_ERR_INVALID)
    if workspace is None:
        workspace = torch.empty(need, dtype=torch.uint8, device=out.device)
    _lib.check(lib.vkr_ibl_irradiance_by_region_rows(_ptr(env_tiles), env_n, _ptr(out), n, r0, r1,
                                                     _ptr(workspace), workspace.numel(), _stream()))
    return workspace


def _require_all_samples(workspace: torch.Tensor, what: str) -> None:
    missing = irradiance_uncovered(workspace)
    if missing:
        raise RuntimeError(f"{what}: {missing} texels did not find all of their samples (written as NaN); "
                           "the kernel's culling bounds have a flaw at these sizes -- use VKR_IRRADIANCE_SEQUENTIAL and report it")


def irradiance_uncovered(workspace: torch.Tensor, need: int | None = None) -> int:
    """Texels of the last irradiance_by_region() call on this workspace whose samples were not all found (0 when
    the kernel is sound).  `need` = the size the call asked for, when the workspace is larger.  Synchronises."""
    end = workspace.numel() if need is None else need
    return int(workspace[end - 256:end - 252].cpu().view(torch.int32).item())


def prefilter(env: torch.Tensor, out: torch.Tensor, roughness: float, rows=None,
              env_layout: int = _lib.ENV_LAYOUT_LINEAR, env_n: int | None = None) -> None:
    """`env`: the cube [6, n, n, 4] row-major or in 4x4 tiles, or (ENV_LAYOUT_FOOTPRINT_TILES) the flat tensor
    tile_env_footprints() returns, with the face size in `env_n`."""
    lib = _lib.load()
    n = out.shape[1]
    r0, r1 = _rows(6 * n, rows)
    if env_layout == _lib.ENV_LAYOUT_FOOTPRINT_TILES:
        if env_n is None or env.numel() != lib.vkr_env_footprint_tiles_bytes(env_n):
            raise ValueError("footprint tiles: pass the face size of the cube they were made from in env_n")
    else:
        env_n = env.shape[1]
    _lib.check(lib.vkr_ibl_prefilter(_ptr(env), env_n, env_layout, _ptr(out), n, C.c_float(roughness), r0, r1, _stream()))


def prefilter_mips(env: torch.Tensor, outs: list, rows: list | None = None, env_layout: int = _lib.ENV_LAYOUT_LINEAR,
                   env_n: int | None = None) -> None:
    """Every roughness mip of the prefiltered cube (outs[m]: [6, n >> m, n >> m, 4] float16, roughness m / (mips - 1)) in as
    few launches as serve them best: the images of one prefilter() call per mip.  rows[m] = (begin, end) of mip m's
    6 (n >> m) rows, default all."""
    lib = _lib.load()
    if env_layout == _lib.ENV_LAYOUT_FOOTPRINT_TILES:
        if env_n is None or env.numel() != lib.vkr_env_footprint_tiles_bytes(env_n):
            raise ValueError("footprint tiles: pass the face size of the cube they were made from in env_n")
    else:
        env_n = env.shape[1]
    a = _lib.PrefilterMipsArgs()
    a.env, a.env_res, a.env_layout = _ptr(env), env_n, env_layout
    a.base_res, a.mips = outs[0].shape[1], len(outs)
    for m, o in enumerate(outs):
        if o.shape[1] != a.base_res >> m:
            raise ValueError(f"mip {m} has face size {o.shape[1]}, expected {a.base_res >> m}")
        a.out[m] = _ptr(o).value
        a.row_begin[m], a.row_end[m] = _rows(6 * o.shape[1], None if rows is None else rows[m])
    _lib.check(lib.vkr_ibl_prefilter_mips(C.byref(a), _stream()))


def brdf_lut(out: torch.Tensor, rows=None) -> None:
    lib = _lib.load()
    n = out.shape[0]
    r0, r1 = _rows(n, rows)
    _lib.check(lib.vkr_ibl_brdf_lut(_ptr(out), n, r0, r1, _stream()))
