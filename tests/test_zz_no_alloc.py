"""vkr_gbuffer_static: the geometry pass for callers that cannot allocate device memory.  Its scratch memory
(visibility keys, draw records) is a fixed array inside the library instead of a workspace; this file checks,
through the C ABI, that its four images equal vkr_gbuffer's bit for bit on two calls in a row -- the second
on a smaller image, so whatever the first call left in the shared array must not leak into it -- and that
sizes beyond the fixed array are refused."""
import ctypes as C

import numpy as np
import pytest

from vulkanrenderer_amd import _lib

from tests.test_gpu_parity import _camera_scene, _gpass_device_draws, _gpass_scene, assert_bits_equal

pytestmark = pytest.mark.gpu


def _gpass_args(P, torch, draws, scene, w, h, frame_rows, row0, device, workspace=None):
    """GpassArgs over fresh output images; the ctypes draw array is returned too so that it outlives the call."""
    arr = (_lib.GpassDraw * len(draws))()
    for k, d in enumerate(draws):
        for name in ("positions", "texcoords", "normals", "tangents"):
            setattr(arr[k], name, d[name].data_ptr())
        arr[k].n_verts, arr[k].indices, arr[k].n_tris = d["positions"].shape[0], d["triangles"].data_ptr(), d["triangles"].shape[0]
        arr[k].model[:] = [float(v) for v in np.asarray(d["model"], dtype=np.float32).ravel()]
        for name in ("albedo", "normal", "arm"):
            tex = getattr(arr[k], name)
            tex.rgba8, tex.width, tex.height = d[name].data_ptr(), d[name].shape[1], d[name].shape[0]
    out = [torch.full((h, w, 4), 7.0, dtype=torch.float16, device=device) for _ in range(4)]
    a = _lib.GpassArgs()
    a.draws, a.n_draws = arr, len(draws)
    a.scene = P._scene_struct(scene)
    a.width, a.rows, a.frame_rows, a.row0 = w, h, frame_rows, row0
    a.albedo, a.normal, a.arm, a.position = (o.data_ptr() for o in out)
    if workspace is not None:
        a.workspace, a.workspace_bytes = workspace.data_ptr(), workspace.numel()
    return a, out, arr


def test_gbuffer_static_equals_gbuffer_on_two_calls(gpu):
    import torch
    from vulkanrenderer_amd import passes as P
    lib = _lib.load()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    # first a whole frame, then a smaller strip of another frame: the second call reuses the same scratch array
    for w, h, frame_rows, row0 in ((192, 108, 108, 0), (157, 41, 108, 30)):
        draws = _gpass_device_draws(_gpass_scene(w, h), gpu)
        scene = _camera_scene((2.0, 1.6, -4.0), aspect=w / frame_rows)
        need = lib.vkr_gbuffer_workspace_bytes(w, h, len(draws))
        ws = torch.empty(need, dtype=torch.uint8, device=gpu)
        a_ref, want, keep_ref = _gpass_args(P, torch, draws, scene, w, h, frame_rows, row0, gpu, ws)
        _lib.check(lib.vkr_gbuffer(C.byref(a_ref), stream))
        a, got, keep = _gpass_args(P, torch, draws, scene, w, h, frame_rows, row0, gpu)
        _lib.check(lib.vkr_gbuffer_static(C.byref(a), stream))
        torch.cuda.synchronize()
        covered = P.f16_bits(want[0])[..., 3] != 0
        assert 0.2 < covered.mean() < 0.95, "the scene should cover part of the image"
        for name, g, wv in zip(("albedo", "normal", "arm", "position"), got, want):
            assert_bits_equal(P.f16_bits(g), P.f16_bits(wv), f"{w}x{h} G-buffer {name} (static scratch vs workspace)")
        del keep_ref, keep


def test_gbuffer_static_refuses_sizes_beyond_its_scratch(gpu):
    import torch
    from vulkanrenderer_amd import passes as P
    lib = _lib.load()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    draws = _gpass_device_draws(_gpass_scene(64, 64), gpu)
    scene = _camera_scene((2.0, 1.6, -4.0), aspect=1.0)
    # one row more than the fixed array holds (the output images are never touched: the call must fail first)
    w = 2048
    h = _lib.GBUFFER_STATIC_MAX_PIXELS // w + 1
    a, _, keep = _gpass_args(P, torch, draws, scene, 8, 8, 8, 0, gpu)
    a.width, a.rows, a.frame_rows = w, h, h
    assert lib.vkr_gbuffer_static(C.byref(a), stream) == _lib.VKR_ERR_INVALID
    assert b"exceed the fixed scratch" in lib.vkr_last_error()
    many = draws * (_lib.GBUFFER_STATIC_MAX_DRAWS // len(draws) + 1)
    a, _, keep2 = _gpass_args(P, torch, many, scene, 8, 8, 8, 0, gpu)
    assert lib.vkr_gbuffer_static(C.byref(a), stream) == _lib.VKR_ERR_INVALID
    assert b"draws exceed the fixed scratch" in lib.vkr_last_error()
    del keep, keep2
