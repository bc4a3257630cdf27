"""Host-side mirror of the reference's renderer interface for the shading path.

`HipRenderer` keeps the method names, call order and error behaviour of `Renderer.VulkanRenderer`
(src/Renderer/VulkanRenderer.cs) for the part of a frame this package replaces, and forwards every
call to the renderer object of libvkr_hip.so (layer 2 of include/vkr_hip.h):

    UpdateSceneInfo(info)          VulkanRenderer.cs:635   per-frame 240-byte SceneInfo
    Lights                         VulkanRenderer.cs:65    list of Light; index = shadow cube index
    BeginFrame() / EndFrame()      VulkanRenderer.cs:286 / :320   raise if called out of order
    CreateSkyboxCubemap(...) etc.  VulkanRenderer.cs:653,734,815,966   synchronous IBL precompute
    WaitIdle()                     blocks until the device is idle

    CreateMesh / CreateMaterial / DrawModel(...)   Mesh.cs, Material.cs, VulkanRenderer.cs:630

EndFrame renders the shadow maps and the G-buffer from the frame's draw calls and the skybox from the
environment cube; a caller that renders those itself hands the images over with UploadGBuffer /
UploadDirectionalShadowMap / UploadPointShadowMap / UploadBackground.
"""
from __future__ import annotations

import ctypes as C
import dataclasses

import numpy as np

from . import _lib
from .synth import LIGHT_DTYPE, SCENE_DTYPE, light_volume_radius


@dataclasses.dataclass
class Light:
    """Light.cs:5-27."""
    Position: tuple
    Color: tuple

    def ToInfo(self) -> np.ndarray:
        """Light.cs:10-18: Model = Scale(volume radius) * Translate(position)."""
        info = np.zeros((), dtype=LIGHT_DTYPE)
        r = light_volume_radius(self.Color)
        m = np.diag([r, r, r, 1.0])
        m[3, :3] = self.Position
        info["model"] = m.astype(np.float32).ravel()
        info["position"] = self.Position
        info["color"] = self.Color
        return info


def _host_ptr(a: np.ndarray) -> C.c_void_p:
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data)


class HipRenderer:
    """Drop-in for the composition / bloom / post-process part of VulkanRenderer.EndFrame and for the
    IBL precompute of its constructor, running on one MI355X."""

    MaxFramesInFlight = 2  # VulkanRenderer.cs:60
    MaxLights = 32         # VulkanRenderer.cs:63

    def __init__(self, width: int = 1920, height: int = 1080, *, device: int = 0, max_lights: int = 0,
                 dir_shadow_res: int = 0, point_shadow_res: int = 0, env_res: int = 0, irradiance_res: int = 0,
                 prefilter_res: int = 0, prefilter_mips: int = 0, brdf_lut_res: int = 0, srgb_target: bool = False,
                 frame_rows: int = 0, row0: int = 0, shadow_ranges: bool | None = None):
        """`frame_rows` / `row0`: the renderer owns rows [row0, row0 + height) of a frame of frame_rows rows (one
        strip of a frame rendered on several GPUs: StripShade / strip_exchange / StripFinish end its frames).
        `shadow_ranges`: the depth ranges of the point-shadow cubes in the composition pass -- None: used by frames with
        at least 16 point lights, True / False: always / never (the image is the same)."""
        self._lib = _lib.load()
        cfg = _lib.Config(device=device, width=width, height=height, max_lights=max_lights,
                          dir_shadow_res=dir_shadow_res, point_shadow_res=point_shadow_res, env_res=env_res,
                          irradiance_res=irradiance_res, prefilter_res=prefilter_res, prefilter_mips=prefilter_mips,
                          brdf_lut_res=brdf_lut_res, frames_in_flight=self.MaxFramesInFlight,
                          flags=(_lib.POST_SRGB_TARGET if srgb_target else 0) |
                                {None: 0, True: _lib.SHADOW_RANGES_ALWAYS, False: _lib.SHADOW_RANGES_NEVER}[shadow_ranges],
                          frame_rows=frame_rows, row0=row0)
        self._ctx = C.c_void_p()
        _lib.check(self._lib.vkr_create(C.byref(cfg), C.byref(self._ctx)))
        self.width, self.height = width, height
        self.max_lights = max_lights or self.MaxLights
        self.dir_shadow_res = dir_shadow_res or 1024
        self.point_shadow_res = point_shadow_res or 256
        self.env_res = env_res or 512
        self.irradiance_res = irradiance_res or 32
        self.prefilter_res = prefilter_res or 128
        self.prefilter_mips = prefilter_mips or 5
        self.brdf_lut_res = brdf_lut_res or 512
        self.Lights: list[Light] = []
        self._explicit_lights = False   # SetLightInfos was called for the frame being built: EndFrame keeps those records
        self._has_volume_mesh = False

    # ---- lifetime (IDisposable in the reference) ----------------------------------------------
    def Dispose(self) -> None:
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._lib.vkr_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        self.Dispose()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.Dispose()

    def _call(self, fn, *args):
        if not self._ctx.value:
            raise _lib.VkrError(_lib.VKR_ERR_INVALID, "renderer is disposed")
        return _lib.check(fn(self._ctx, *args))

    # ---- per-frame interface --------------------------------------------------------------------
    @property
    def CurrentFrame(self) -> int:
        return self._call(self._lib.vkr_current_frame)

    def UpdateSceneInfo(self, scene_info: np.ndarray) -> None:
        assert scene_info.dtype == SCENE_DTYPE
        s = _lib.SceneInfo.from_buffer_copy(scene_info.tobytes())
        self._call(self._lib.vkr_update_scene_info, C.byref(s))

    def BeginFrame(self) -> None:
        self._call(self._lib.vkr_begin_frame)

    def EndFrame(self) -> None:
        """Uploads the light list (draw order = list order), then runs composition, bloom 1, bloom 2 and
        post-process for the current frame and advances the frame index.  Light records given with
        SetLightInfos since the last EndFrame take the place of the `Lights` list for this frame."""
        self._set_lights()
        self._call(self._lib.vkr_end_frame)
        self._explicit_lights = False

    def _set_lights(self) -> None:
        """`Lights` -> LightInfo records; once a light-volume mesh is set, each light's draw covers the pixels
        of its volume (what rasterising sphereMesh per light does in the reference), before that every pixel."""
        if self._explicit_lights:
            return
        infos = np.zeros(len(self.Lights), dtype=LIGHT_DTYPE)
        for i, light in enumerate(self.Lights):
            infos[i] = light.ToInfo()
        self._push_lights(infos, _lib.COVERAGE_MESH if self._has_volume_mesh else _lib.COVERAGE_FULLSCREEN)

    def _push_lights(self, infos: np.ndarray, coverage: int) -> None:
        infos = np.ascontiguousarray(infos)
        self._call(self._lib.vkr_set_lights, _host_ptr(infos) if len(infos) else C.c_void_p(0), len(infos), coverage)

    def SetLightInfos(self, infos: np.ndarray, coverage: int = _lib.COVERAGE_FULLSCREEN) -> None:
        """Raw LightInfo records and a coverage mode for the frame being built, instead of the `Lights` list."""
        assert infos.dtype == LIGHT_DTYPE
        self._push_lights(infos, coverage)
        self._explicit_lights = True

    def CreateMesh(self, verts: np.ndarray, triangles: np.ndarray, texcoords=None, normals=None, tangents=None) -> int:
        """Keeps a mesh on the device (Mesh.cs); returns its handle for DrawModel.  With texture coordinates,
        normals and tangents the mesh can take part in the geometry pass."""
        verts = np.ascontiguousarray(verts, dtype=np.float32)
        triangles = np.ascontiguousarray(triangles, dtype=np.uint32)
        handle = C.c_int32(-1)
        self._call(self._lib.vkr_create_mesh, _host_ptr(verts), len(verts), _host_ptr(triangles), len(triangles), C.byref(handle))
        if texcoords is not None:
            uv, nrm, tan = (np.ascontiguousarray(a, dtype=np.float32) for a in (texcoords, normals, tangents))
            assert len(uv) == len(nrm) == len(tan) == len(verts)
            self._call(self._lib.vkr_set_mesh_attributes, handle.value, _host_ptr(uv), _host_ptr(nrm), _host_ptr(tan))
        return handle.value

    def CreateMaterial(self, albedo: np.ndarray, normal: np.ndarray, arm: np.ndarray) -> int:
        """Material.cs: three RGBA8 images [h, w, 4] (albedo sRGB, normal map, ao/roughness/metalness sRGB)."""
        imgs = [np.ascontiguousarray(a, dtype=np.uint8) for a in (albedo, normal, arm)]
        handle = C.c_int32(-1)
        args = []
        for im in imgs:
            assert im.ndim == 3 and im.shape[2] == 4
            args += [_host_ptr(im), im.shape[1], im.shape[0]]
        self._call(self._lib.vkr_create_material, *args, C.byref(handle))
        return handle.value

    def DrawModel(self, mesh: int, model_matrix, material: int = -1) -> None:
        """VulkanRenderer.DrawModel (VulkanRenderer.cs:630-633): queues a draw call for this frame.  EndFrame
        renders the shadow maps from the frame's draw calls, and the G-buffer too when every draw has a material."""
        m = np.ascontiguousarray(model_matrix, dtype=np.float32).ravel()
        assert m.size == 16
        self._call(self._lib.vkr_draw_model, mesh, material, _host_ptr(m))

    def SetLightVolumeMesh(self, verts: np.ndarray, triangles: np.ndarray) -> None:
        """The mesh each light volume is an instance of (the reference's sphereMesh, VulkanRenderer.cs:78,269);
        with it, SetLightInfos(..., COVERAGE_MESH) restricts each light's draw to the pixels its volume covers."""
        verts = np.ascontiguousarray(verts, dtype=np.float32)
        triangles = np.ascontiguousarray(triangles, dtype=np.uint32)
        self._call(self._lib.vkr_set_light_volume_mesh, _host_ptr(verts), len(verts), _host_ptr(triangles), len(triangles))
        self._has_volume_mesh = True

    def WaitIdle(self) -> None:
        self._call(self._lib.vkr_sync)

    def Resize(self, width: int, height: int) -> None:
        """RecreateSwapchain (VulkanRenderer.cs:1087-1127): every frame-sized image is recreated."""
        self._call(self._lib.vkr_resize, width, height)
        self.width, self.height = width, height

    # ---- inputs produced by passes outside this path -------------------------------------------
    def UploadGBuffer(self, albedo, normal, arm, position) -> None:
        imgs = [np.ascontiguousarray(a) for a in (albedo, normal, arm, position)]
        for a in imgs:
            assert a.dtype == np.uint16 and a.shape == (self.height, self.width, 4)
        self._call(self._lib.vkr_upload_gbuffer, *[_host_ptr(a) for a in imgs])

    def UploadBackground(self, rgba16f: np.ndarray | None) -> None:
        if rgba16f is None:
            self._call(self._lib.vkr_upload_background, C.c_void_p(0))
        else:
            a = np.ascontiguousarray(rgba16f)
            assert a.dtype == np.uint16 and a.shape == (self.height, self.width, 4)
            self._call(self._lib.vkr_upload_background, _host_ptr(a))

    def UploadDirectionalShadowMap(self, depth: np.ndarray) -> None:
        a = np.ascontiguousarray(depth, dtype=np.float32)
        assert a.shape == (self.dir_shadow_res, self.dir_shadow_res)
        self._call(self._lib.vkr_upload_dir_shadow, _host_ptr(a))

    def UploadPointShadowMap(self, light: int, depth_cube: np.ndarray) -> None:
        a = np.ascontiguousarray(depth_cube, dtype=np.float32)
        assert a.shape == (6, self.point_shadow_res, self.point_shadow_res)
        self._call(self._lib.vkr_upload_point_shadow, light, _host_ptr(a))

    # ---- IBL precompute (constructor work in the reference, VulkanRenderer.cs:271-275) ----------
    def CreateSkyboxCubemap(self, equirect: np.ndarray) -> None:
        a = np.ascontiguousarray(equirect)
        fmt = _lib.SRC_RGBA8_SRGB if a.dtype == np.uint8 else _lib.SRC_RGBA16F
        assert a.ndim == 3 and a.shape[2] == 4 and a.dtype in (np.uint8, np.uint16)
        self._call(self._lib.vkr_create_skybox_cubemap, _host_ptr(a), a.shape[1], a.shape[0], fmt)

    def UploadEnvironmentCubemap(self, cube: np.ndarray) -> None:
        a = np.ascontiguousarray(cube)
        assert a.dtype == np.uint16 and a.shape == (6, self.env_res, self.env_res, 4)
        self._call(self._lib.vkr_upload_env_cube, _host_ptr(a))

    def CreateIrradianceCubemap(self, order: int = _lib.IRRADIANCE_FASTEST) -> None:
        """VulkanRenderer.cs:734.  The default order is the library's fastest (by region of the environment with row sums),
        which differs from the shader's own running sum (IRRADIANCE_SEQUENTIAL) in rounding only."""
        self._call(self._lib.vkr_create_irradiance_cubemap, order)

    def CreatePrefilteredCubemap(self) -> None:
        self._call(self._lib.vkr_create_prefiltered_cubemap)

    def CreateBRDFLUTTexture(self) -> None:
        self._call(self._lib.vkr_create_brdf_lut)

    # ---- strips: one frame on several GPUs ----------------------------------------------------------
    def StripShade(self) -> None:
        """Every pass of the frame up to the composition pass, for this renderer's rows."""
        self._set_lights()
        self._call(self._lib.vkr_strip_shade)

    def StripFinish(self) -> None:
        """Bloom 1, bloom 2 and post-process for this renderer's rows (after the halo rows of the neighbouring
        strips have arrived: strip_exchange); advances the frame index."""
        self._call(self._lib.vkr_strip_finish)
        self._explicit_lights = False

    def StripHalo(self, side: int) -> tuple[int, int, int]:
        """Device addresses (send from, receive into) and size in bytes of the rows this strip swaps with its
        neighbour on `side` (_lib.STRIP_ABOVE / STRIP_BELOW), for hosts that move them with a transport of their own."""
        send, recv, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        self._call(self._lib.vkr_strip_halo, side, C.byref(send), C.byref(recv), C.byref(n))
        return send.value, recv.value, n.value

    # ---- RenderStage mirror (RenderStage.cs:99-190) -----------------------------------------------
    def Stage(self, stage: int) -> "HipRenderStage":
        """The reference's stage object for _lib.STAGE_*: record with its five calls, run with SubmitFrame()."""
        return HipRenderStage(self, stage)

    def PushLights(self) -> None:
        """What EndFrame does before its stages: the light records of the frame go to the device."""
        self._set_lights()

    def SubmitFrame(self) -> None:
        """Runs the recorded stages in the reference's submission order (VulkanRenderer.cs:559-595) and advances
        the frame index (:622)."""
        self._call(self._lib.vkr_submit_frame)
        self._explicit_lights = False

    # ---- single passes, run at once ----------------------------------------------------------------
    def CompositionPass(self) -> None:
        self._set_lights()
        self._call(self._lib.vkr_pass_composition)

    def BloomPass(self, horizontal: bool) -> None:
        self._call(self._lib.vkr_pass_bloom, int(bool(horizontal)))

    def PostProcessPass(self) -> None:
        self._call(self._lib.vkr_pass_postprocess)

    # ---- read-back -------------------------------------------------------------------------------
    def Read(self, target: int, frame: int = 0, mip: int = 0) -> np.ndarray:
        if target <= _lib.TARGET_BLOOM2:
            out = np.empty((self.height, self.width, 4), dtype=np.uint16)
        elif target == _lib.TARGET_LDR:
            out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        elif target == _lib.TARGET_ENV:
            out = np.empty((6, self.env_res, self.env_res, 4), dtype=np.uint16)
        elif target == _lib.TARGET_IRRADIANCE:
            out = np.empty((6, self.irradiance_res, self.irradiance_res, 4), dtype=np.uint16)
        elif target == _lib.TARGET_PREFILTER:
            n = self.prefilter_res >> mip
            out = np.empty((6, n, n, 4), dtype=np.uint16)
        elif target == _lib.TARGET_BRDF_LUT:
            out = np.empty((self.brdf_lut_res, self.brdf_lut_res, 2), dtype=np.uint16)
        else:
            raise _lib.VkrError(_lib.VKR_ERR_INVALID, f"unknown target {target}")
        self._call(self._lib.vkr_read, frame, target, mip, _host_ptr(out), out.nbytes)
        return out

    def Timing(self) -> dict:
        t = _lib.Timing()
        self._call(self._lib.vkr_get_timing, C.byref(t))
        return {"shade_ms": t.shade_ms, "bloom_h_ms": t.bloom_h_ms, "bloom_v_post_ms": t.bloom_v_post_ms, "frame_ms": t.frame_ms}


def strip_plan(frame_rows: int, n_strips: int, k: int) -> tuple[int, int]:
    """(row0, rows) of strip k when a frame of frame_rows rows is cut into n_strips near-equal strips."""
    row0, rows = C.c_int32(), C.c_int32()
    _lib.check(_lib.load().vkr_strip_plan(frame_rows, n_strips, k, C.byref(row0), C.byref(rows)))
    return row0.value, rows.value


def strip_exchange(upper: HipRenderer, lower: HipRenderer) -> None:
    """Swaps the boundary rows of two neighbouring strips of one process (device-to-device, asynchronous)."""
    _lib.check(upper._lib.vkr_strip_exchange(upper._ctx, lower._ctx))


class HipRenderStage:
    """RenderStage.cs:99-190 over the C ABI: the reference's five calls with their arguments (commandBufferIndex is
    the current frame), plus Draw() for the draws VulkanRenderer.EndFrame records into the stage."""

    def __init__(self, renderer: HipRenderer, stage: int):
        self._r, self._stage = renderer, stage

    def _do(self, fn, command_buffer_index: int) -> None:
        self._r._call(fn, self._stage, command_buffer_index)

    def ResetCommandBuffer(self, commandBufferIndex: int) -> None:
        self._do(self._r._lib.vkr_stage_reset_commands, commandBufferIndex)

    def BeginCommands(self, commandBufferIndex: int) -> None:
        self._do(self._r._lib.vkr_stage_begin_commands, commandBufferIndex)

    def BeginRenderPass(self, commandBufferIndex: int, framebufferIndex: int) -> None:
        self._do(self._r._lib.vkr_stage_begin_render_pass, commandBufferIndex)

    def Draw(self, commandBufferIndex: int) -> None:
        self._do(self._r._lib.vkr_stage_draw, commandBufferIndex)

    def EndRenderPass(self, commandBufferIndex: int) -> None:
        self._do(self._r._lib.vkr_stage_end_render_pass, commandBufferIndex)

    def EndCommands(self, commandBufferIndex: int) -> None:
        self._do(self._r._lib.vkr_stage_end_commands, commandBufferIndex)
