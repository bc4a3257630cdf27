"""GPU parity: every pass of libvkr_hip.so, called through the C ABI, against the CPU oracle on the
same seeded inputs.

Bars: the composition pass, both bloom passes, the prefilter, the BRDF LUT and the irradiance
convolution are bit-exact (both sides follow one arithmetic profile built from correctly rounded
IEEE operations).  The 8-bit image may differ by 1 LSB (pow(x, 1/2.2) comes from the hardware
log2/exp2 units), and the equirect -> cube resampling by max(1e-3, one binary16 ulp) (atan/asin
come from different math libraries).
"""
import numpy as np
import pytest

from vulkanrenderer_amd import _lib, synth

pytestmark = pytest.mark.gpu


def _passes():
    from vulkanrenderer_amd import passes
    return passes


def _torch():
    import torch
    return torch


def assert_bits_equal(got: np.ndarray, want: np.ndarray, what: str):
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        first = tuple(bad[0])
        raise AssertionError(f"{what}: {len(bad)} of {got.size} values differ; first at {first}: got 0x{int(got[first]):04x} want 0x{int(want[first]):04x}")


def f16_ulp(x: np.ndarray) -> np.ndarray:
    """Size of one binary16 ulp at |x| (normal range)."""
    e = np.floor(np.log2(np.maximum(np.abs(x), 2.0 ** -14)))
    return 2.0 ** (e - 10)


FRAMES = [
    # seed, width, height, lights, cube_res, dir_res
    (11, 160, 96, 5, 64, 256),     # multiple of the 16x16 tile
    (12, 201, 37, 3, 32, 128),     # ragged: odd width, height not a multiple of anything
    (13, 64, 64, 0, 16, 64),       # no point lights at all
    (14, 96, 80, 33, 16, 64),      # more lights than the reference's MaxLights
    (16, 88, 48, 2, 24, 64),       # cube faces whose size is not a power of two
    (17, 72, 40, 100, 8, 64),      # enough lights for two launches (64 + 36): the second continues from the stored targets
    (18, 56, 24, 161, 8, 32),      # three launches (64 + 97 would exceed a launch and a half: 64 + 64 + 33 lights)
]


@pytest.mark.parametrize("seed,w,h,nl,cube,dres", FRAMES)
def test_shade_bit_exact(gpu, oracle, seed, w, h, nl, cube, dres):
    torch, P = _torch(), _passes()
    f = synth.make_frame(seed, w, h, nl, cube, dres, irr_res=8, pref_res=16, lut_res=32)
    want_c, want_t = oracle.shade(f)
    d = P.DeviceFrame.upload(f, gpu)
    color = torch.empty((h, w, 4), dtype=torch.float16, device=gpu)
    thr = torch.empty_like(color)
    P.shade(d, color, thr)
    torch.cuda.synchronize()
    assert_bits_equal(P.f16_bits(color), want_c, "composition colour")
    assert_bits_equal(P.f16_bits(thr), want_t, "thresholded colour")


@pytest.mark.parametrize("fixture", ["frame_composition_lights.npz", "frame_composition_lights_48x32.npz"])
def test_shade_against_reference_shader_vectors(gpu, fixture):
    """The shading kernel against the outputs recorded from the reference's own composition.frag.spv / light.frag.spv
    (tests/golden, written by tools/make_golden.py), without the oracle in between.  The recorded values come from an
    interpreter that evaluates GLSL built-ins in numpy binary32, so agreement is to rounding: max(1e-3, 2 binary16
    ulps), thresholded pixels except those whose luminance is within 2e-3 of the threshold."""
    import os
    torch, P = _torch(), _passes()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", fixture))
    h, w = g["albedo"].shape[:2]
    f = synth.FrameInputs(w, h, g["albedo"], g["normal"], g["arm"], g["position"], g["scene"][()], g["lights"],
                          g["dir_shadow"], g["cube_shadow"], g["irradiance"], g["prefilter"], g["brdf_lut"])
    d = P.DeviceFrame.upload(f, gpu)
    color = torch.empty((h, w, 4), dtype=torch.float16, device=gpu)
    thr = torch.empty_like(color)
    P.shade(d, color, thr)
    torch.cuda.synchronize()
    got_c = color.float().cpu().numpy()[..., :3]
    got_t = thr.float().cpu().numpy()[..., :3]
    close = lambda a, b: np.abs(a - b) <= np.maximum(1e-3, 2.0 * f16_ulp(b))
    assert close(got_c, g["color"]).all(), f"colour: max abs diff {np.abs(got_c - g['color']).max()}"
    sure = ~g["near_threshold"]
    assert sure.mean() > 0.99   # the exclusion is a handful of pixels (0 of 192 and 1 of 1536 in the two fixtures), not a loophole
    assert close(got_t, g["threshold"])[sure].all()
    # cleared texels: what light.frag.spv yields where max() of a NaN returns the other operand (exactly 0 per draw)
    assert (g["cleared_light_out_max_is_maxnum"] == 0.0).all()
    for _, y, x in g["cleared_px"]:
        assert (got_c[y, x] == 0.0).all() and (got_t[y, x] == 0.0).all()


def test_post_chain_and_ibl_against_reference_shader_vectors(gpu):
    """The one-kernel post chain against bloom_blur.frag.spv (x2) + postprocess.frag.spv, and the four IBL kernels
    against equirectangular_to_cubemap / irradiance / prefilter / brdf .frag.spv (tests/golden), without the oracle."""
    import os
    torch, P = _torch(), _passes()
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    close = lambda a, b, ulps=2.0: np.abs(a - b) <= np.maximum(1e-3, ulps * f16_ulp(b))
    g = np.load(os.path.join(gold, "post_chain.npz"))
    src, hdr = P.to_device_f16(g["src"], gpu), P.to_device_f16(g["hdr"], gpu)
    b1 = torch.empty_like(src)
    P.bloom_blur(src, b1, True, _lib.WRAP_REPEAT)
    assert close(b1.float().cpu().numpy(), g["bloom1"], 1.0).all()
    b1_ref = P.to_device_f16(np.ascontiguousarray(g["bloom1"].astype(np.float16)).view(np.uint16), gpu)
    b2 = torch.empty_like(src)
    P.bloom_blur(b1_ref, b2, False, _lib.WRAP_CLAMP)
    assert close(b2.float().cpu().numpy(), g["bloom2"], 1.0).all()
    # the recorded 8-bit image was made from the recorded bloom-2 image: feed the separate post-process draw with it,
    # and the one-kernel chain with the source (its own bloom images agree with the recorded ones to 1 ulp: 1 LSB more)
    b2_ref = P.to_device_f16(np.ascontiguousarray(g["bloom2"].astype(np.float16)).view(np.uint16), gpu)
    want = np.floor(np.clip(g["ldr_float"], 0, 1) * 255.0 + 0.5).astype(np.int32)
    ldr = torch.empty(src.shape, dtype=torch.uint8, device=gpu)
    P.postprocess(hdr, b2_ref, ldr)
    assert np.abs(ldr.cpu().numpy().astype(np.int32) - want).max() <= 1
    P.bloom_postprocess(src, hdr, ldr)
    assert np.abs(ldr.cpu().numpy().astype(np.int32) - want).max() <= 2

    g = np.load(os.path.join(gold, "ibl.npz"))
    n = g["equirect_cube"].shape[1]
    cube = torch.empty((6, n, n, 4), dtype=torch.float16, device=gpu)
    esrc = g["equirect_src"]
    fmt = _lib.SRC_RGBA8_SRGB if esrc.dtype == np.uint8 else _lib.SRC_RGBA16F
    P.equirect_to_cube(torch.from_numpy(esrc).to(gpu) if esrc.dtype == np.uint8 else P.to_device_f16(esrc, gpu), fmt, cube)
    assert close(cube.float().cpu().numpy(), g["equirect_cube"]).all()
    env = P.to_device_f16(g["env"], gpu)
    irr = torch.empty((6, 2, 2, 4), dtype=torch.float16, device=gpu)
    for order in (_lib.IRRADIANCE_SEQUENTIAL, _lib.IRRADIANCE_PHI_SLICES, _lib.IRRADIANCE_BY_REGION, _lib.IRRADIANCE_BY_REGION_ROWS):
        P.irradiance(env, irr, order)
        got = irr.float().cpu().numpy()
        for (f, i, j), want in zip(g["irradiance_texels"], g["irradiance"]):
            assert close(got[f, j, i], want).all(), (order, f, i, j)
    pref = {}
    for (r, f, i, j), want in zip(g["prefilter_cases"], g["prefilter"]):
        r = float(r)
        if r not in pref:
            out = torch.empty((6, 2, 2, 4), dtype=torch.float16, device=gpu)
            P.prefilter(env, out, r)
            pref[r] = out.float().cpu().numpy()
        assert close(pref[r][int(f), int(j), int(i)], want).all(), (r, f, i, j)
    ln = g["brdf_lut"].shape[0]
    lut = torch.empty((ln, ln, 2), dtype=torch.float16, device=gpu)
    P.brdf_lut(lut)
    assert close(lut.float().cpu().numpy(), g["brdf_lut"]).all()


def test_ibl_kernels_against_reference_shader_vectors_on_1024_faces(gpu):
    """The four IBL kernels against the reference's .frag.spv on an environment with an interior (tests/golden/ibl_1024.npz):
    all four irradiance summation orders, the row-major and both tiled environment layouts, the prefilter at its five
    roughnesses, a 32^2 LUT, the equirect pass -- the recorded shader outputs directly, without the oracle in between."""
    import os
    from tests.test_golden_reference_spirv import _env_1024
    torch, P = _torch(), _passes()
    close = lambda a, b, ulps=2.0: np.abs(a - b) <= np.maximum(1e-3, ulps * f16_ulp(b))
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ibl_1024.npz"))
    env = P.to_device_f16(_env_1024(g), gpu)
    tiled = P.tile_env_cube(env)
    n = int(g["irradiance_n"])
    irr = torch.empty((6, n, n, 4), dtype=torch.float16, device=gpu)
    cases = [(order, env, _lib.ENV_LAYOUT_LINEAR) for order in (_lib.IRRADIANCE_SEQUENTIAL, _lib.IRRADIANCE_PHI_SLICES, _lib.IRRADIANCE_BY_REGION,
                                                               _lib.IRRADIANCE_BY_REGION_ROWS, _lib.IRRADIANCE_FASTEST)]
    cases += [(order, tiled, _lib.ENV_LAYOUT_TILED_4X4) for order in (_lib.IRRADIANCE_SEQUENTIAL, _lib.IRRADIANCE_BY_REGION)]
    for order, cube, layout in cases:
        irr.zero_()
        P.irradiance(cube, irr, order, env_layout=layout)   # the rows order makes its footprint tiles from the row-major cube
        got = irr.float().cpu().numpy()
        for (f, i, j), want in zip(g["irradiance_texels"], g["irradiance"]):
            assert close(got[f, j, i], want).all(), (order, layout, f, i, j, got[f, j, i], want)
    pn, pref = int(g["prefilter_n"]), {}
    for (r, f, i, j), want in zip(g["prefilter_cases"], g["prefilter"]):
        r = float(r)
        for layout, cube in ((_lib.ENV_LAYOUT_LINEAR, env), (_lib.ENV_LAYOUT_TILED_4X4, tiled)):
            if (r, layout) not in pref:
                out = torch.empty((6, pn, pn, 4), dtype=torch.float16, device=gpu)
                P.prefilter(cube, out, r, env_layout=layout)
                pref[(r, layout)] = out.float().cpu().numpy()
            assert close(pref[(r, layout)][int(f), int(j), int(i)], want).all(), (r, layout, f, i, j)
    ln = g["brdf_lut"].shape[0]
    lut = torch.empty((ln, ln, 2), dtype=torch.float16, device=gpu)
    P.brdf_lut(lut)
    assert close(lut.float().cpu().numpy(), g["brdf_lut"]).all()
    en = g["equirect_cube"].shape[1]
    cube = torch.empty((6, en, en, 4), dtype=torch.float16, device=gpu)
    P.equirect_to_cube(P.to_device_f16(g["equirect_src"], gpu), _lib.SRC_RGBA16F, cube)
    assert close(cube.float().cpu().numpy(), g["equirect_cube"]).all()


@pytest.mark.parametrize("env_n,base,mips", [(64, 16, 5), (36, 12, 3), (128, 128, 2)])
def test_prefilter_mips_in_one_call_equal_one_launch_per_mip(gpu, oracle, env_n, base, mips):
    """vkr_ibl_prefilter_mips (the small mips share a launch, a large one takes sixteen texels per workgroup) against one
    vkr_ibl_prefilter call per mip and against the oracle, from all three environment layouts, whole images and row
    ranges as a rank of a sharded precompute passes them."""
    torch, P = _torch(), _passes()
    env = synth.make_env_cube(np.random.default_rng(77), env_n, sun_peak=8.0)
    dev = P.to_device_f16(env, gpu)
    layouts = [(dev, _lib.ENV_LAYOUT_LINEAR, None), (P.tile_env_footprints(dev), _lib.ENV_LAYOUT_FOOTPRINT_TILES, env_n)]
    if env_n % 4 == 0:
        layouts.append((P.tile_env_cube(dev), _lib.ENV_LAYOUT_TILED_4X4, None))
    want = [oracle.prefilter(env, base >> m, m / max(mips - 1, 1)) for m in range(mips)]
    for cube, layout, n_arg in layouts:
        outs = [torch.zeros((6, base >> m, base >> m, 4), dtype=torch.float16, device=gpu) for m in range(mips)]
        P.prefilter_mips(cube, outs, env_layout=layout, env_n=n_arg)
        for m in range(mips):
            assert_bits_equal(P.f16_bits(outs[m]), want[m], f"mip {m} of {mips}, layout {layout}")
            one = torch.zeros_like(outs[m])
            P.prefilter(cube, one, m / max(mips - 1, 1), env_layout=layout, env_n=n_arg)
            assert torch.equal(one, outs[m])
        # rows [1/3, 2/3) of every mip, the rest untouched
        part = [torch.full((6, base >> m, base >> m, 4), 7.0, dtype=torch.float16, device=gpu) for m in range(mips)]
        rows = [(2 * (base >> m), 4 * (base >> m)) for m in range(mips)]
        P.prefilter_mips(cube, part, rows, env_layout=layout, env_n=n_arg)
        for m in range(mips):
            n = base >> m
            got = P.f16_bits(part[m]).reshape(6 * n, n, 4)
            assert np.array_equal(got[rows[m][0]:rows[m][1]], want[m].reshape(6 * n, n, 4)[rows[m][0]:rows[m][1]])
            assert (part[m].reshape(6 * n, n, 4)[:rows[m][0]] == 7.0).all() and (part[m].reshape(6 * n, n, 4)[rows[m][1]:] == 7.0).all()


def test_shade_background_image(gpu, oracle):
    torch, P = _torch(), _passes()
    f = synth.make_frame(21, 80, 48, 2, 16, 64, irr_res=8, pref_res=16, lut_res=32)
    bg = synth.make_smooth_cube(np.random.default_rng(5), 80, 0.0, 2.0)[0][:48]  # any RGBA16F image
    bg = np.ascontiguousarray(bg)
    want_c, want_t = oracle.shade(f, background=bg)
    d = P.DeviceFrame.upload(f, gpu)
    d.background = P.to_device_f16(bg, gpu)
    color = torch.empty((48, 80, 4), dtype=torch.float16, device=gpu)
    thr = torch.empty_like(color)
    P.shade(d, color, thr)
    assert_bits_equal(P.f16_bits(color), want_c, "colour over background")
    assert_bits_equal(P.f16_bits(thr), want_t, "threshold over background")


def _random_hdr(seed, h, w, scale=4.0):
    rng = np.random.default_rng(seed)
    img = rng.random((h, w, 4), dtype=np.float32) * scale
    img[rng.random((h, w)) < 0.5] = 0.0   # the thresholded target is mostly zero
    img[..., 3] = 1.0
    return np.ascontiguousarray(img.astype(np.float16)).view(np.uint16)


@pytest.mark.parametrize("w,h", [(512, 40), (520, 33), (37, 19), (1030, 5), (3, 3)])
def test_bloom_passes_bit_exact(gpu, oracle, w, h):
    torch, P = _torch(), _passes()
    src = _random_hdr(w * 1000 + h, h, w)
    want1 = oracle.bloom_blur(src, True, oracle.WRAP_REPEAT)
    want2 = oracle.bloom_blur(want1, False, oracle.WRAP_CLAMP)
    s = P.to_device_f16(src, gpu)
    b1 = torch.empty_like(s)
    b2 = torch.empty_like(s)
    P.bloom_blur(s, b1, True, _lib.WRAP_REPEAT)
    P.bloom_blur(b1, b2, False, _lib.WRAP_CLAMP)
    assert_bits_equal(P.f16_bits(b1), want1, "bloom pass 1 (horizontal, REPEAT)")
    assert_bits_equal(P.f16_bits(b2), want2, "bloom pass 2 (vertical, CLAMP)")
    # the other wrap mode on each axis
    want1c = oracle.bloom_blur(src, True, oracle.WRAP_CLAMP)
    want2r = oracle.bloom_blur(src, False, oracle.WRAP_REPEAT)
    P.bloom_blur(s, b1, True, _lib.WRAP_CLAMP)
    P.bloom_blur(s, b2, False, _lib.WRAP_REPEAT)
    assert_bits_equal(P.f16_bits(b1), want1c, "horizontal, CLAMP")
    assert_bits_equal(P.f16_bits(b2), want2r, "vertical, REPEAT")


@pytest.mark.parametrize("w,h,srgb", [(256, 32, False), (101, 7, False), (64, 16, True)])
def test_postprocess_within_one_lsb(gpu, oracle, w, h, srgb):
    torch, P = _torch(), _passes()
    hdr = _random_hdr(7, h, w, scale=12.0)
    bloom = _random_hdr(8, h, w, scale=1.0)
    want = oracle.postprocess(hdr, bloom, srgb)
    out = torch.empty((h, w, 4), dtype=torch.uint8, device=gpu)
    P.postprocess(P.to_device_f16(hdr, gpu), P.to_device_f16(bloom, gpu), out, _lib.POST_SRGB_TARGET if srgb else 0)
    got = out.cpu().numpy()
    diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert diff.max() <= 1, f"8-bit image differs by {diff.max()} LSB"
    assert (diff != 0).mean() < 0.01, f"{(diff != 0).mean():.4f} of the bytes differ by 1 LSB"
    assert (got[..., 3] == 255).all()


@pytest.mark.parametrize("w,h", [(256, 50), (77, 31)])
def test_fused_bloomv_postprocess_equals_unfused(gpu, w, h):
    torch, P = _torch(), _passes()
    b1 = P.to_device_f16(_random_hdr(3, h, w), gpu)
    hdr = P.to_device_f16(_random_hdr(4, h, w, 8.0), gpu)
    b2 = torch.empty_like(b1)
    out_a = torch.empty((h, w, 4), dtype=torch.uint8, device=gpu)
    out_b = torch.empty_like(out_a)
    P.bloom_blur(b1, b2, False, _lib.WRAP_CLAMP)
    P.postprocess(hdr, b2, out_a)
    P.bloomv_postprocess(b1, hdr, out_b)
    assert torch.equal(out_a, out_b)


@pytest.mark.parametrize("w,h", [(256, 50), (77, 31), (3, 3), (130, 9), (1, 1), (384, 100), (258, 20)])
@pytest.mark.parametrize("wrap_h,wrap_v", [(0, 1), (1, 0)])
def test_one_kernel_bloom_chain_equals_three_passes(gpu, w, h, wrap_h, wrap_v):
    """vkr_bloom_postprocess against vkr_bloom_blur x2 + vkr_postprocess (themselves held to the oracle above):
    ragged widths, images smaller than the blur radius, tiles that cross the right edge, both wraps on both axes."""
    torch, P = _torch(), _passes()
    thr = P.to_device_f16(_random_hdr(5 + w, h, w), gpu)
    hdr = P.to_device_f16(_random_hdr(6 + h, h, w, 8.0), gpu)
    b1, b2 = torch.empty_like(thr), torch.empty_like(thr)
    out_a = torch.empty((h, w, 4), dtype=torch.uint8, device=gpu)
    out_b = torch.full_like(out_a, 7)
    P.bloom_blur(thr, b1, True, wrap_h)
    P.bloom_blur(b1, b2, False, wrap_v)
    P.postprocess(hdr, b2, out_a)
    P.bloom_postprocess(thr, hdr, out_b, wrap_h=wrap_h, wrap_v=wrap_v)
    assert torch.equal(out_a, out_b)


def test_one_kernel_bloom_chain_on_a_strip_with_halo_rows(gpu):
    """A strip hands the kernel its threshold rows plus four halo rows on each side; rows beyond the frame clamp."""
    torch, P = _torch(), _passes()
    w, h = 200, 96
    thr = P.to_device_f16(_random_hdr(11, h, w), gpu)
    hdr = P.to_device_f16(_random_hdr(12, h, w, 8.0), gpu)
    whole = torch.empty((h, w, 4), dtype=torch.uint8, device=gpu)
    P.bloom_postprocess(thr, hdr, whole)
    for (a, b) in ((0, 40), (40, 77), (77, 96)):
        lo, hi = max(a - 4, 0), min(b + 4, h)
        src = thr[lo:hi].contiguous()
        out = torch.empty((b - a, w, 4), dtype=torch.uint8, device=gpu)
        P.bloom_postprocess(src, hdr[a:b].contiguous(), out, src_row_offset=a - lo)
        assert torch.equal(out, whole[a:b]), f"rows {a}..{b}"
    # srgb target flag
    out_a, out_b = torch.empty_like(whole), torch.empty_like(whole)
    b1, b2 = torch.empty_like(thr), torch.empty_like(thr)
    P.bloom_blur(thr, b1, True, _lib.WRAP_REPEAT)
    P.bloom_blur(b1, b2, False, _lib.WRAP_CLAMP)
    P.postprocess(hdr, b2, out_a, _lib.POST_SRGB_TARGET)
    P.bloom_postprocess(thr, hdr, out_b, _lib.POST_SRGB_TARGET)
    assert torch.equal(out_a, out_b)


def test_brdf_lut_bit_exact(gpu, oracle):
    torch, P = _torch(), _passes()
    n = 48
    want = oracle.brdf_lut(n)
    out = torch.empty((n, n, 2), dtype=torch.float16, device=gpu)
    P.brdf_lut(out)
    assert_bits_equal(P.f16_bits(out), want, "BRDF LUT")


@pytest.mark.parametrize("rough", [0.0, 0.25, 0.5, 1.0])
def test_prefilter_bit_exact(gpu, oracle, rough):
    torch, P = _torch(), _passes()
    env = synth.make_env_cube(np.random.default_rng(3), 16)
    want = oracle.prefilter(env, 8, rough)
    out = torch.empty((6, 8, 8, 4), dtype=torch.float16, device=gpu)
    P.prefilter(P.to_device_f16(env, gpu), out, rough)
    assert_bits_equal(P.f16_bits(out), want, f"prefilter roughness {rough}")


@pytest.mark.parametrize("n,footprints", [(7, False), (12, True), (110, True)])
def test_prefilter_at_roughness_zero(gpu, oracle, n, footprints):
    """Roughness 0: the kernel evaluates the samples whose half vector is exactly N once per texel and only the other 137 one by
    one, then adds all 1024 terms in sample order.  Odd face sizes have texels with a zero component of N, which take the general
    path, as do launches of two texels per workgroup (the 7^2 and 12^2 faces here); 110^2 faces are a launch of sixteen texels per workgroup, the form that shares the terms."""
    torch, P = _torch(), _passes()
    env = synth.make_env_cube(np.random.default_rng(21), 24)
    want = oracle.prefilter(env, n, 0.0)
    out = torch.empty((6, n, n, 4), dtype=torch.float16, device=gpu)
    dev_env = P.to_device_f16(env, gpu)
    if footprints:
        P.prefilter(P.tile_env_footprints(dev_env), out, 0.0, env_layout=_lib.ENV_LAYOUT_FOOTPRINT_TILES, env_n=24)
    else:
        P.prefilter(dev_env, out, 0.0)
    assert_bits_equal(P.f16_bits(out), want, f"prefilter {n}^2 at roughness 0")


def test_prefilter_large_launch_equals_small_launches(gpu, oracle):
    """A prefilter launch of 65536 texels or more takes sixteen texels per workgroup and the samples in eight passes, a smaller
    one two texels and all samples at once: the whole cube in one launch against the same cube row range by row range,
    every texel, and sampled texels against the oracle (odd face size: the last group is cut short)."""
    torch, P = _torch(), _passes()
    n = 107
    env = synth.make_env_cube(np.random.default_rng(12), 32)
    dev_env = P.to_device_f16(env, gpu)
    for rough in (0.0, 0.6):
        whole = torch.zeros((6, n, n, 4), dtype=torch.float16, device=gpu)
        parts = torch.zeros_like(whole)
        P.prefilter(dev_env, whole, rough)
        for r0 in range(0, 6 * n, 150):
            P.prefilter(dev_env, parts, rough, rows=(r0, min(r0 + 150, 6 * n)))
        assert np.array_equal(P.f16_bits(whole), P.f16_bits(parts)), rough
    want = oracle.prefilter(env, n, 0.6).reshape(6 * n * n, 4)
    assert_bits_equal(P.f16_bits(whole).reshape(6 * n * n, 4), want, "prefilter 107^2, roughness 0.6")


@pytest.mark.parametrize("order", [0, 1, 2, 3])
def test_irradiance_bit_exact(gpu, oracle, order):
    torch, P = _torch(), _passes()
    env = synth.make_env_cube(np.random.default_rng(4), 16)
    want = oracle.irradiance(env, 4, order)
    out = torch.empty((6, 4, 4, 4), dtype=torch.float16, device=gpu)
    P.irradiance(P.to_device_f16(env, gpu), out, order)
    assert_bits_equal(P.f16_bits(out), want, f"irradiance order {order}")


def test_irradiance_by_region_of_a_large_environment(gpu, oracle):
    """The by-region order where a face has several regions (1024^2 faces: 2 x 2 regions of 512^2 on each), a last
    region that is cut short (640^2: regions of 512 and 128 texels), rows of one rank, a workspace that is reused,
    too small or missing, and the word that counts texels with samples unaccounted for."""
    torch, P = _torch(), _passes()
    lib = _lib.load()
    for env_n, n, tiled in ((1024, 4, True), (640, 3, False), (1024, 3, False)):
        env = synth.make_env_cube(np.random.default_rng(env_n + n), env_n)
        dev_env = P.to_device_f16(env, gpu)
        layout = _lib.ENV_LAYOUT_LINEAR
        if tiled:
            dev_env, layout = P.tile_env_cube(dev_env), _lib.ENV_LAYOUT_TILED_4X4
        want = oracle.irradiance(env, n, 2)
        out = torch.zeros((6, n, n, 4), dtype=torch.float16, device=gpu)
        ws = P.irradiance_by_region(dev_env, out, env_layout=layout)
        assert P.irradiance_uncovered(ws) == 0
        assert_bits_equal(P.f16_bits(out), want, f"irradiance by region, environment {env_n}, cube {n}")
        # it is the shader's sum but for rounding: half a binary16 ulp of rounding error in the sum, then the store
        seq = oracle.f16_to_f32(oracle.irradiance(env, n, 0))
        got = out.float().cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(seq))   # odd n: the centre texels of the +-Y faces have no frame
        assert (np.isnan(seq) | (np.abs(got - seq) <= 1.0 * f16_ulp(seq))).all()
        # rows [r0, r1) only, into an image that holds something else, reusing the (larger) workspace
        r0, r1 = n + 1, 4 * n - 1
        part = torch.full((6, n, n, 4), 7.0, dtype=torch.float16, device=gpu)
        P.irradiance_by_region(dev_env, part, (r0, r1), layout, workspace=ws)
        need = lib.vkr_ibl_irradiance_by_region_workspace_bytes(env_n, n, r0, r1)
        assert 0 < need <= ws.numel() and P.irradiance_uncovered(ws, need) == 0
        got = P.f16_bits(part).reshape(6 * n, n, 4)
        assert np.array_equal(got[r0:r1], want.reshape(6 * n, n, 4)[r0:r1])
        assert (part.view(6 * n, n, 4)[:r0] == 7.0).all() and (part.view(6 * n, n, 4)[r1:] == 7.0).all()
        with pytest.raises(_lib.VkrError):
            P.irradiance_by_region(dev_env, out, env_layout=layout, workspace=ws[:need - 1])
    # more shapes of regions and texel grids, against the kernel in the shader's order: every sample accounted for, and
    # the two sums within rounding of each other (three regions per side, eight per side, one; odd and even cubes)
    for env_n, n in ((1536, 8), (4096, 6), (512, 16), (2048, 9), (72, 5)):
        gen = torch.Generator(device=gpu).manual_seed(env_n)
        dev_env = (torch.rand((6, env_n, env_n, 4), generator=gen, device=gpu) * 4.0).to(torch.float16)  # made on the device: 4096^2 faces are 0.8 GB
        layout = _lib.ENV_LAYOUT_LINEAR
        if env_n % 8 == 0:
            dev_env, layout = P.tile_env_cube(dev_env), _lib.ENV_LAYOUT_TILED_4X4
        out = torch.zeros((6, n, n, 4), dtype=torch.float16, device=gpu)
        seq = torch.zeros_like(out)
        ws = P.irradiance_by_region(dev_env, out, env_layout=layout)
        P.irradiance(dev_env, seq, 0, env_layout=layout)
        assert P.irradiance_uncovered(ws) == 0, (env_n, n)
        a, b = seq.float().cpu().numpy(), out.float().cpu().numpy()
        assert np.array_equal(np.isnan(a), np.isnan(b))
        assert (np.isnan(a) | (np.abs(a - b) <= 1.0 * f16_ulp(a))).all(), (env_n, n)
        del dev_env, ws
    env_n, n = 1024, 3
    env = synth.make_env_cube(np.random.default_rng(env_n + n), env_n)
    dev_env, layout = P.to_device_f16(env, gpu), _lib.ENV_LAYOUT_LINEAR
    out = torch.zeros((6, n, n, 4), dtype=torch.float16, device=gpu)
    with pytest.raises(_lib.VkrError):
        _lib.check(lib.vkr_ibl_irradiance(P._ptr(dev_env), env_n, layout, P._ptr(out), n, 2, 0, 6 * n, None))
    assert lib.vkr_ibl_irradiance_by_region_workspace_bytes(0, 4, 0, 24) == -1
    assert lib.vkr_ibl_irradiance_by_region_workspace_bytes(64, 4, 0, 25) == -1
    empty = torch.zeros((6, n, n, 4), dtype=torch.float16, device=gpu)
    P.irradiance_by_region(dev_env, empty, (5, 5), layout)  # no rows: nothing is written
    assert (empty == 0).all()


def test_footprint_tiles_hold_every_footprint_whole(gpu, oracle):
    """vkr_tile_rgba16f_footprints against the layout's definition: tile (k, m) of a face holds the texels
    (3k-1 .. 3k+2) x (3m-1 .. 3m+2), those beyond an edge folded from the neighbouring face as the oracle's cube_fold
    folds them, those beyond a corner of the cube (or beyond the apron) zero."""
    torch, P = _torch(), _passes()
    lib = _lib.load()
    for n in (8, 9, 10, 37):
        env = synth.make_env_cube(np.random.default_rng(n), n)
        tiles = P.tile_env_footprints(P.to_device_f16(env, gpu))
        k = n // 3 + 1
        assert tiles.numel() == lib.vkr_env_footprint_tiles_bytes(n) == 6 * k * k * 128
        got = tiles.cpu().numpy().view(np.uint16).reshape(6, k, k, 4, 4, 4)   # face, m, k, row, column, channel
        for f in range(6):
            for m in range(k):
                for kk in range(k):
                    for r in range(4):
                        for c in range(4):
                            i, j = 3 * kk + c - 1, 3 * m + r - 1
                            want = np.zeros(4, np.uint16)
                            if i <= n and j <= n:
                                ok, (of, oi, oj) = oracle.cube_fold(n, f, i, j)
                                if ok:
                                    want = env[of, oj, oi]
                            assert np.array_equal(got[f, m, kk, r, c], want), (n, f, m, kk, r, c)
    assert lib.vkr_env_footprint_tiles_bytes(0) == -1 and lib.vkr_env_footprint_tiles_bytes(8193) == -1


def test_irradiance_by_region_rows_of_a_large_environment(gpu, oracle):
    """The by-region order with row sums where a face has several regions (520^2 faces: 2 x 2 regions of 87 cells;
    1024^2: 3 x 3 of 114), rows of one rank, a workspace that is reused, too small or missing, the word that counts
    texels with samples unaccounted for, cube corners and face edges (small cubes look across them all the time)."""
    torch, P = _torch(), _passes()
    lib = _lib.load()
    for env_n, n in ((1024, 4), (520, 3), (640, 5), (16, 6)):
        env = synth.make_env_cube(np.random.default_rng(env_n + n), env_n)
        dev_env = P.to_device_f16(env, gpu)
        tiles = P.tile_env_footprints(dev_env)
        want = oracle.irradiance(env, n, 3)
        out = torch.zeros((6, n, n, 4), dtype=torch.float16, device=gpu)
        ws = P.irradiance_by_region_rows(tiles, env_n, out)
        assert P.irradiance_uncovered(ws) == 0
        assert_bits_equal(P.f16_bits(out), want, f"irradiance by region with row sums, environment {env_n}, cube {n}")
        seq = oracle.f16_to_f32(oracle.irradiance(env, n, 0))
        got = out.float().cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(seq))   # odd n: the centre texels of the +-Y faces have no frame
        assert (np.isnan(seq) | (np.abs(got - seq) <= 1.0 * f16_ulp(seq))).all()
        r0, r1 = n + 1, 4 * n - 1
        part = torch.full((6, n, n, 4), 7.0, dtype=torch.float16, device=gpu)
        P.irradiance_by_region_rows(tiles, env_n, part, (r0, r1), workspace=ws)
        need = lib.vkr_ibl_irradiance_by_region_rows_workspace_bytes(env_n, n, r0, r1)
        assert 0 < need <= ws.numel() and P.irradiance_uncovered(ws, need) == 0
        got = P.f16_bits(part).reshape(6 * n, n, 4)
        assert np.array_equal(got[r0:r1], want.reshape(6 * n, n, 4)[r0:r1])
        assert (part.view(6 * n, n, 4)[:r0] == 7.0).all() and (part.view(6 * n, n, 4)[r1:] == 7.0).all()
        with pytest.raises(_lib.VkrError):
            P.irradiance_by_region_rows(tiles, env_n, out, workspace=ws[:need - 1])
        # the convenience form: tiles and workspace made inside, the sample count checked
        again = torch.zeros_like(out)
        P.irradiance(dev_env, again, _lib.IRRADIANCE_FASTEST)
        assert torch.equal(again, out)
    # more shapes of regions and texel grids, against the kernel in the shader's order: every sample accounted for, and
    # the two sums within rounding of each other
    for env_n, n in ((1536, 8), (4096, 6), (512, 16), (2048, 9), (72, 5), (383, 7), (385, 4), (1, 2), (5, 1)):
        gen = torch.Generator(device=gpu).manual_seed(env_n)
        dev_env = (torch.rand((6, env_n, env_n, 4), generator=gen, device=gpu) * 4.0).to(torch.float16)  # made on the device: 4096^2 faces are 0.8 GB
        out = torch.zeros((6, n, n, 4), dtype=torch.float16, device=gpu)
        seq = torch.zeros_like(out)
        ws = P.irradiance_by_region_rows(P.tile_env_footprints(dev_env), env_n, out)
        P.irradiance(dev_env, seq, 0)
        assert P.irradiance_uncovered(ws) == 0, (env_n, n)
        a, b = seq.float().cpu().numpy(), out.float().cpu().numpy()
        assert np.array_equal(np.isnan(a), np.isnan(b))
        assert (np.isnan(a) | (np.abs(a - b) <= 1.0 * f16_ulp(a))).all(), (env_n, n)
        del dev_env, ws
    assert lib.vkr_ibl_irradiance_by_region_rows_workspace_bytes(0, 4, 0, 24) == -1
    assert lib.vkr_ibl_irradiance_by_region_rows_workspace_bytes(64, 4, 0, 25) == -1
    env = P.to_device_f16(synth.make_env_cube(np.random.default_rng(1), 16), gpu)
    empty = torch.zeros((6, 3, 3, 4), dtype=torch.float16, device=gpu)
    P.irradiance_by_region_rows(P.tile_env_footprints(env), 16, empty, (5, 5))  # no rows: nothing is written
    assert (empty == 0).all()


@pytest.mark.parametrize("hdr", [True, False])
def test_equirect_to_cube_within_tolerance(gpu, oracle, hdr):
    torch, P = _torch(), _passes()
    src = synth.make_equirect(np.random.default_rng(9), 256, 128, hdr)
    want = oracle.f16_to_f32(oracle.equirect_to_cube(src, 32))
    out = torch.empty((6, 32, 32, 4), dtype=torch.float16, device=gpu)
    dev_src = P.to_device_f16(src, gpu) if hdr else torch.from_numpy(src).to(gpu)
    P.equirect_to_cube(dev_src, _lib.SRC_RGBA16F if hdr else _lib.SRC_RGBA8_SRGB, out)
    got = out.float().cpu().numpy()
    tol = np.maximum(1e-3, f16_ulp(want))
    assert (np.abs(got - want) <= tol).all(), f"max abs error {np.abs(got - want).max()}"


def test_ibl_row_ranges_compose(gpu):
    """Computing a cube in two row ranges gives the same image as one call (the multi-GPU split)."""
    torch, P = _torch(), _passes()
    env = P.to_device_f16(synth.make_env_cube(np.random.default_rng(6), 16), gpu)
    whole = torch.empty((6, 8, 8, 4), dtype=torch.float16, device=gpu)
    parts = torch.zeros_like(whole)
    P.prefilter(env, whole, 0.5)
    P.prefilter(env, parts, 0.5, rows=(0, 19))
    P.prefilter(env, parts, 0.5, rows=(19, 48))
    assert torch.equal(whole, parts)
    lut_a = torch.empty((32, 32, 2), dtype=torch.float16, device=gpu)
    lut_b = torch.zeros_like(lut_a)
    P.brdf_lut(lut_a)
    P.brdf_lut(lut_b, rows=(0, 5))
    P.brdf_lut(lut_b, rows=(5, 32))
    assert torch.equal(lut_a, lut_b)


def test_renderer_end_frame_matches_oracle_chain(gpu, oracle):
    """Layer 2: HipRenderer driven the way Program.OnRender drives VulkanRenderer."""
    from vulkanrenderer_amd.renderer import HipRenderer
    w, h, nl = 120, 72, 4
    f = synth.make_frame(31, w, h, nl, 32, 128, irr_res=8, pref_res=16, lut_res=32)
    want = oracle.frame_chain(f)
    with HipRenderer(w, h, max_lights=8, dir_shadow_res=128, point_shadow_res=32, env_res=16,
                     irradiance_res=8, prefilter_res=16, prefilter_mips=2, brdf_lut_res=32) as r:
        # IBL images computed by the library from an environment cube, then the oracle shades with the same images
        env = synth.make_env_cube(np.random.default_rng(2), 16)
        r.UploadEnvironmentCubemap(env)
        for order in (_lib.IRRADIANCE_PHI_SLICES, _lib.IRRADIANCE_BY_REGION, _lib.IRRADIANCE_BY_REGION_ROWS, _lib.IRRADIANCE_SEQUENTIAL):
            r.CreateIrradianceCubemap(order)
            assert_bits_equal(r.Read(_lib.TARGET_IRRADIANCE), oracle.irradiance(env, 8, order), f"irradiance cube, order {order}")
        r.CreateIrradianceCubemap()  # the default: the library's fastest order
        assert_bits_equal(r.Read(_lib.TARGET_IRRADIANCE), oracle.irradiance(env, 8, _lib.IRRADIANCE_BY_REGION_ROWS), "irradiance cube, default order")
        r.CreatePrefilteredCubemap()
        r.CreateBRDFLUTTexture()
        f.irradiance = r.Read(_lib.TARGET_IRRADIANCE)
        f.prefilter = r.Read(_lib.TARGET_PREFILTER, mip=0)
        f.brdf_lut = r.Read(_lib.TARGET_BRDF_LUT)
        want = oracle.frame_chain(f, oracle.skybox(env, f.scene, w, h))  # EndFrame opens with the skybox draw
        for _ in range(3):  # cycles through both frames in flight
            frame = r.CurrentFrame
            r.UpdateSceneInfo(f.scene)
            r.UploadGBuffer(f.albedo, f.normal, f.arm, f.position)
            r.UploadDirectionalShadowMap(f.dir_shadow)
            for i in range(nl):
                r.UploadPointShadowMap(i, f.cube_shadow[i])
            r.SetLightInfos(f.lights)
            r.BeginFrame()
            r._call(r._lib.vkr_end_frame)
            assert_bits_equal(r.Read(_lib.TARGET_COLOR, frame), want["color"], "colour")
            assert_bits_equal(r.Read(_lib.TARGET_THRESHOLD, frame), want["threshold"], "threshold")
            with pytest.raises(_lib.VkrError):  # EndFrame stores neither bloom image
                r.Read(_lib.TARGET_BLOOM1, frame)
            ldr = r.Read(_lib.TARGET_LDR, frame)
            assert np.abs(ldr.astype(np.int32) - want["ldr"].astype(np.int32)).max() <= 1
        # order errors, as the reference throws them (VulkanRenderer.cs:288,322)
        with pytest.raises(_lib.VkrError):
            r._call(r._lib.vkr_end_frame)
        r.BeginFrame()
        with pytest.raises(_lib.VkrError):
            r.BeginFrame()


def test_render_stages_record_and_submit_like_the_reference(gpu, oracle):
    """Layer 2 through the RenderStage mirror: the five calls per stage, one submission; every attachment of the
    frame (both bloom images included) equals the oracle's chain; order errors; a render pass without draws leaves
    its attachments at the clear values; a stage that records nothing leaves its images alone."""
    from vulkanrenderer_amd.renderer import HipRenderer
    w, h, nl = 120, 72, 4
    f = synth.make_frame(31, w, h, nl, 32, 128, irr_res=8, pref_res=16, lut_res=32)
    with HipRenderer(w, h, max_lights=8, dir_shadow_res=128, point_shadow_res=32, env_res=16,
                     irradiance_res=8, prefilter_res=16, prefilter_mips=2, brdf_lut_res=32) as r:
        env = synth.make_env_cube(np.random.default_rng(2), 16)
        r.UploadEnvironmentCubemap(env)
        r.CreateIrradianceCubemap(); r.CreatePrefilteredCubemap(); r.CreateBRDFLUTTexture()
        f.irradiance = r.Read(_lib.TARGET_IRRADIANCE)
        f.prefilter = r.Read(_lib.TARGET_PREFILTER, mip=0)
        f.brdf_lut = r.Read(_lib.TARGET_BRDF_LUT)
        want = oracle.frame_chain(f, oracle.skybox(env, f.scene, w, h))
        screen_stages = (_lib.STAGE_COMPOSITION, _lib.STAGE_BLOOM1, _lib.STAGE_BLOOM2, _lib.STAGE_POSTPROCESS)

        def upload():
            r.UpdateSceneInfo(f.scene)
            r.UploadGBuffer(f.albedo, f.normal, f.arm, f.position)
            r.UploadDirectionalShadowMap(f.dir_shadow)
            for i in range(nl):
                r.UploadPointShadowMap(i, f.cube_shadow[i])
            r.SetLightInfos(f.lights)

        def record(stage, cf, draw=True):
            st = r.Stage(stage)
            st.BeginCommands(cf)
            st.BeginRenderPass(cf, cf)
            if draw:
                st.Draw(cf)
            st.EndRenderPass(cf)
            st.EndCommands(cf)

        for _ in range(3):  # both frames in flight
            cf = r.CurrentFrame
            r.BeginFrame()
            upload()
            for stage in screen_stages:  # the geometry and shadow stages record nothing: their images are the uploads
                record(stage, cf)
            r.SubmitFrame()
            assert r.CurrentFrame == (cf + 1) % 2
            for target, name in ((_lib.TARGET_COLOR, "color"), (_lib.TARGET_THRESHOLD, "threshold"), (_lib.TARGET_BLOOM1, "bloom1"), (_lib.TARGET_BLOOM2, "bloom2")):
                assert_bits_equal(r.Read(target, cf), want[name], name)
            assert np.abs(r.Read(_lib.TARGET_LDR, cf).astype(np.int32) - want["ldr"].astype(np.int32)).max() <= 1

        cf = r.CurrentFrame
        with pytest.raises(_lib.VkrError):      # no frame begun
            r.SubmitFrame()
        r.BeginFrame()
        upload()
        st = r.Stage(_lib.STAGE_COMPOSITION)
        with pytest.raises(_lib.VkrError):      # render pass before BeginCommands
            st.BeginRenderPass(cf, cf)
        with pytest.raises(_lib.VkrError):      # another frame's command buffer
            st.BeginCommands(cf ^ 1)
        st.BeginCommands(cf)
        with pytest.raises(_lib.VkrError):      # already recording
            st.BeginCommands(cf)
        with pytest.raises(_lib.VkrError):      # draw outside a render pass
            st.Draw(cf)
        st.BeginRenderPass(cf, cf)
        with pytest.raises(_lib.VkrError):      # EndCommands inside a render pass
            st.EndCommands(cf)
        with pytest.raises(_lib.VkrError):      # a stage still recording cannot be submitted
            r.SubmitFrame()
        st.Draw(cf)
        st.EndRenderPass(cf)
        st.EndCommands(cf)
        record(_lib.STAGE_GEOMETRY, cf, draw=False)   # cleared G-buffer: the composition draw discards every pixel
        record(_lib.STAGE_BLOOM1, cf, draw=False)     # bloom 1 at its clear value (0, 0, 0, 1)
        record(_lib.STAGE_BLOOM2, cf)
        record(_lib.STAGE_POSTPROCESS, cf)
        r.SubmitFrame()
        sky = oracle.skybox(env, f.scene, w, h)
        assert_bits_equal(r.Read(_lib.TARGET_COLOR, cf)[..., :3], sky[..., :3], "skybox behind a cleared G-buffer")
        b1 = r.Read(_lib.TARGET_BLOOM1, cf)
        assert (b1[..., :3] == 0).all() and (b1[..., 3] == 0x3C00).all()
        assert (r.Read(_lib.TARGET_BLOOM2, cf)[..., :3] == 0).all()
        want_ldr = oracle.postprocess(np.ascontiguousarray(r.Read(_lib.TARGET_COLOR, cf)), np.ascontiguousarray(r.Read(_lib.TARGET_BLOOM2, cf)))
        assert np.abs(r.Read(_lib.TARGET_LDR, cf).astype(np.int32) - want_ldr.astype(np.int32)).max() <= 1

        # a host that mirrors EndFrame literally calls Draw() in every stage: with no DrawModel calls in the frame the
        # geometry and shadow passes hold zero draws and clear (VulkanRenderer.cs:357-366, 391-400, 417-424), whatever
        # was uploaded -- the frame is the skybox alone
        cf = r.CurrentFrame
        r.BeginFrame()
        upload()
        for stage in range(_lib.STAGE_COUNT):
            record(stage, cf)
        r.SubmitFrame()
        assert_bits_equal(r.Read(_lib.TARGET_COLOR, cf)[..., :3], sky[..., :3], "Draw() without draw calls clears the G-buffer")
        assert (r.Read(_lib.TARGET_THRESHOLD, cf)[..., :3] == 0).all()


def test_strip_pair_with_local_halo_copy_matches_whole_frame(gpu, oracle):
    """Two strips of one frame rendered by two StripRenderers on one GPU, the halo rows copied
    between them where the ranks of a real run would send them: exercises the split launches
    (interior rows first, rows next to a neighbour afterwards) of the multi-GPU path."""
    torch = _torch()
    from vulkanrenderer_amd.strips import HALO, StripPlan, StripRenderer
    P = _passes()
    w, h, world = 200, 45, 2
    f = synth.make_frame(41, w, h, 3, 32, 128, irr_res=8, pref_res=16, lut_res=32)
    want = oracle.frame_chain(f)

    class Paired(StripRenderer):
        peers = None

        def start_halo_exchange(self):
            p, r = self.plan, self.plan.rows
            if not p.is_top:
                up = self.peers[p.rank - 1]
                self.threshold_padded[0:HALO] = up.threshold_padded[up.plan.rows:up.plan.rows + HALO]
            if not p.is_bottom:
                self.threshold_padded[r + HALO:r + 2 * HALO] = self.peers[p.rank + 1].threshold_padded[HALO:2 * HALO]
            return []

    plans = [StripPlan.make(k, world, h) for k in range(world)]
    frames = [P.DeviceFrame.upload(f.rows(p.row0, p.row0 + p.rows), gpu) for p in plans]
    rs = [Paired(p, w, device=gpu) for p in plans]
    for r in rs:
        r.peers = rs
    for r, d in zip(rs, frames):  # thresholded images first: a strip reads its neighbour's boundary rows
        r.backend.shade(d, r.color, r.threshold)
    ldr = torch.cat([r.render(d) for r, d in zip(rs, frames)], 0).cpu().numpy()
    color = np.concatenate([P.f16_bits(r.color) for r in rs], 0)
    assert_bits_equal(color, want["color"], "colour of the two strips")
    assert np.abs(ldr.astype(np.int32) - want["ldr"].astype(np.int32)).max() <= 1
    # and the split launches give the very bytes of a single launch over the whole frame
    whole = StripRenderer(StripPlan.make(0, 1, h), w, device=gpu)
    assert np.array_equal(whole.render(P.DeviceFrame.upload(f, gpu)).cpu().numpy(), ldr)


def test_ibl_precompute_matches_oracle_at_cube_edges_and_corners(gpu, oracle):
    """The whole precompute through vulkanrenderer_amd.ibl on sizes where most irradiance and
    prefilter samples cross face edges and some hit cube corners (64^2 environment)."""
    torch, P = _torch(), _passes()
    import dataclasses
    from vulkanrenderer_amd.ibl import IblConfig, IblPrecompute
    cfg = IblConfig(env_res=64, irradiance_res=8, prefilter_res=16, prefilter_mips=5, lut_res=48)
    src = synth.make_equirect(np.random.default_rng(21), 256, 128, hdr=True)
    res = IblPrecompute(cfg, device=gpu).run(P.to_device_f16(src, gpu), _lib.SRC_RGBA16F)
    env = P.f16_bits(res["env"])  # the GPU's environment cube feeds both sides from here on
    # IblConfig's default order is the library's fastest: by region of the environment with row sums
    assert_bits_equal(P.f16_bits(res["irradiance"]), oracle.irradiance(env, 8, _lib.IRRADIANCE_BY_REGION_ROWS), "irradiance")
    seq = IblPrecompute(dataclasses.replace(cfg, irradiance_order=_lib.IRRADIANCE_SEQUENTIAL), device=gpu).run(P.to_device_f16(src, gpu), _lib.SRC_RGBA16F)
    assert_bits_equal(P.f16_bits(seq["irradiance"]), oracle.irradiance(env, 8, 0), "irradiance in the shader's order")
    for m, cube in enumerate(res["prefilter"]):
        assert_bits_equal(P.f16_bits(cube), oracle.prefilter(env, 16 >> m, m / 4.0), f"prefilter mip {m}")
    assert_bits_equal(P.f16_bits(res["brdf_lut"]), oracle.brdf_lut(48), "BRDF LUT")


@pytest.mark.parametrize("mode", ["end_frame", "stages"])
def test_cpp_host_program_matches_oracle(gpu, oracle, tmp_path, mode):
    """tests/cpp/renderer_roundtrip (built by __graft_entry__.build()) drives vkr::HipRenderer from
    C++ the way the reference's render loop drives VulkanRenderer; its frame equals the oracle's.
    "stages": every pass recorded through vkr::HipRenderStage's five calls (RenderStage.cs:99-190), then one
    submission, instead of EndFrame."""
    import os
    import subprocess
    exe = _cpp_host()
    w, h, nl, dres, cres, eres, ires, pres, lres = 104, 56, 3, 64, 16, 16, 4, 8, 16
    f = synth.make_frame(52, w, h, nl, cres, dres, irr_res=ires, pref_res=pres, lut_res=lres)
    env = synth.make_env_cube(np.random.default_rng(8), eres)
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as fh:
        fh.write(np.array([w, h, nl, dres, cres, eres, ires, pres, lres], dtype=np.int32).tobytes())
        fh.write(f.scene.tobytes())
        for l in f.lights:
            fh.write(np.asarray(l["position"][:3], dtype=np.float32).tobytes())
            fh.write(np.asarray(l["color"][:3], dtype=np.float32).tobytes())
        for a in (f.albedo, f.normal, f.arm, f.position, f.dir_shadow, f.cube_shadow, env):
            fh.write(np.ascontiguousarray(a).tobytes())
    proc = subprocess.run([exe, str(inp), str(outp)] + (["stages"] if mode == "stages" else []), capture_output=True, text=True, timeout=100)
    assert proc.returncode == 0, proc.stderr
    raw = np.fromfile(outp, dtype=np.uint8)
    sizes = [w * h * 8, w * h * 8, w * h * 4, 6 * ires * ires * 8, 6 * pres * pres * 8, lres * lres * 4]
    assert raw.size == sum(sizes)
    parts = np.split(raw, np.cumsum(sizes)[:-1])
    color, thr = (p.view(np.uint16).reshape(h, w, 4) for p in parts[:2])
    ldr = parts[2].reshape(h, w, 4)
    f.irradiance = parts[3].view(np.uint16).reshape(6, ires, ires, 4)
    f.prefilter = parts[4].view(np.uint16).reshape(6, pres, pres, 4)
    f.brdf_lut = parts[5].view(np.uint16).reshape(lres, lres, 2)
    assert_bits_equal(f.irradiance, oracle.irradiance(env, ires, _lib.IRRADIANCE_BY_REGION_ROWS), "irradiance")  # the wrapper's default order
    assert_bits_equal(f.prefilter, oracle.prefilter(env, pres, 0.0), "prefilter")
    assert_bits_equal(f.brdf_lut, oracle.brdf_lut(lres), "BRDF LUT")
    want = oracle.frame_chain(f, oracle.skybox(env, f.scene, w, h))
    assert_bits_equal(color, want["color"], "colour")
    assert_bits_equal(thr, want["threshold"], "threshold")
    assert np.abs(ldr.astype(np.int32) - want["ldr"].astype(np.int32)).max() <= 1


@pytest.mark.parametrize("seed,w,h,nl,cube,dres", FRAMES)
def test_shade_bit_exact_with_row_major_cubes(gpu, oracle, seed, w, h, nl, cube, dres):
    """The same frames with the point-shadow cubes left row-major, and as 8x4 tiles of the plain depth
    (test_shade_bit_exact uses the default: tiles of 25 * depth where the face size allows)."""
    torch, P = _torch(), _passes()
    f = synth.make_frame(seed, w, h, nl, cube, dres, irr_res=8, pref_res=16, lut_res=32)
    want_c, want_t = oracle.shade(f)
    for tiled in (False, True):
        if tiled and (cube % 8 != 0 or nl == 0):
            continue
        d = P.DeviceFrame.upload(f, gpu, tiled=tiled)
        assert d.cube_layout == (_lib.CUBE_LAYOUT_TILED_8X4 if tiled else _lib.CUBE_LAYOUT_LINEAR)
        color = torch.empty((h, w, 4), dtype=torch.float16, device=gpu)
        thr = torch.empty_like(color)
        P.shade(d, color, thr)
        assert_bits_equal(P.f16_bits(color), want_c, f"composition colour (tiled: {tiled})")
        assert_bits_equal(P.f16_bits(thr), want_t, f"thresholded colour (tiled: {tiled})")


def test_tiled_cube_layout(gpu, oracle):
    """vkr_tile_depth_cubes against the layout's definition; a face size that is not a multiple of 8
    stays row-major and is refused for the tiled layout."""
    torch, P = _torch(), _passes()
    res = 24
    lin = torch.arange(2 * 6 * res * res, dtype=torch.float32, device=gpu).reshape(2, 6, res, res)
    got = P.tile_depth_cubes(lin).cpu().numpy().reshape(12, -1)
    src = lin.cpu().numpy().reshape(12, res, res)
    j, i = np.meshgrid(np.arange(res), np.arange(res), indexing="ij")
    index = (i >> 3) * (res * 8) + j * 8 + (i & 7)  # 8x4-texel tiles, stored column by column
    want = np.empty_like(got)
    want[:, index.ravel()] = src.reshape(12, -1)
    assert np.array_equal(got, want)
    got25 = P.tile_depth_cubes(lin, _lib.CUBE_LAYOUT_TILED_8X4_X25).cpu().numpy().reshape(12, -1)
    assert np.array_equal(got25, want * np.float32(25.0))          # the product light.frag forms per tap, rounded once
    with pytest.raises(_lib.VkrError):
        P.tile_depth_cubes(lin, _lib.CUBE_LAYOUT_LINEAR)

    f = synth.make_frame(15, 72, 40, 2, 12, 64, irr_res=8, pref_res=16, lut_res=32)  # 12^2 faces
    d = P.DeviceFrame.upload(f, gpu)
    assert d.cube_layout == _lib.CUBE_LAYOUT_LINEAR
    color = torch.empty((40, 72, 4), dtype=torch.float16, device=gpu)
    thr = torch.empty_like(color)
    P.shade(d, color, thr)
    want_c, want_t = oracle.shade(f)
    assert_bits_equal(P.f16_bits(color), want_c, "composition colour, 12^2 cube faces")
    d.cube_layout = _lib.CUBE_LAYOUT_TILED_8X4
    with pytest.raises(_lib.VkrError):
        P.shade(d, color, thr)
    with pytest.raises(_lib.VkrError):
        P.tile_depth_cubes(d.cube_shadow)


def test_tiled_environment_cube(gpu, oracle):
    """vkr_tile_rgba16f_faces against the layout's definition, and the irradiance / prefilter passes
    reading the tiled cube give the bits they give on the row-major one (= the oracle's)."""
    torch, P = _torch(), _passes()
    n = 64
    env = synth.make_env_cube(np.random.default_rng(33), n)
    dev_env = P.to_device_f16(env, gpu)
    tiled = P.tile_env_cube(dev_env)
    got = P.f16_bits(tiled).reshape(6, n * n, 4)
    j, i = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    index = ((j >> 2) * (n >> 2) + (i >> 2)) * 16 + (j & 3) * 4 + (i & 3)
    want = np.empty_like(got)
    want[:, index.ravel()] = env.reshape(6, n * n, 4)
    assert np.array_equal(got, want)

    T = _lib.ENV_LAYOUT_TILED_4X4
    irr = torch.empty((6, 8, 8, 4), dtype=torch.float16, device=gpu)
    for order in (0, 1, 2):
        P.irradiance(tiled, irr, order, env_layout=T)
        assert_bits_equal(P.f16_bits(irr), oracle.irradiance(env, 8, order), f"irradiance from the tiled cube, order {order}")
    pre = torch.empty((6, 16, 16, 4), dtype=torch.float16, device=gpu)
    for rough in (0.0, 0.5, 1.0):
        P.prefilter(tiled, pre, rough, env_layout=T)
        assert_bits_equal(P.f16_bits(pre), oracle.prefilter(env, 16, rough), f"prefilter from the tiled cube, roughness {rough}")
    odd = P.to_device_f16(synth.make_env_cube(np.random.default_rng(34), 6), gpu)  # 6 is not a multiple of 4
    with pytest.raises(_lib.VkrError):
        P.tile_env_cube(odd)
    with pytest.raises(_lib.VkrError):
        P.irradiance(odd, irr, 0, env_layout=T)


def _cpp_host() -> str:
    """The C++ host program: tests/cpp/renderer_roundtrip, or the build VKR_TEST_CPP_HOST names (scratch/asan/build.sh makes
    one whose host code, the library's included, runs under the address and undefined-behaviour sanitizers)."""
    import os
    exe = os.environ.get("VKR_TEST_CPP_HOST") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "renderer_roundtrip")
    assert os.path.exists(exe), f"{exe} is not built: run __graft_entry__.build()"
    return exe


def _camera_scene(cam, target=(0.0, 0.0, 0.0), aspect=16.0 / 9.0):
    s = synth.default_scene_info().copy()
    s["camera_view"] = synth.look_at(cam, target, (0, 1, 0)).astype(np.float32).ravel()
    s["camera_projection"] = synth.perspective(60.0, aspect, 0.1, 100.0).astype(np.float32).ravel()
    s["camera_position"] = cam
    return s


def _toward_origin(cam):
    d = -np.array(cam, dtype=np.float64)
    return d / np.linalg.norm(d)


def _small_lights(rng, n, scale):
    """Lights whose volumes are much smaller than the reference's (colour maximum `scale` * U[0.2,1])."""
    lights = synth.make_lights(rng, n)
    for i in range(n):
        lights["color"][i] *= scale
        r = synth.light_volume_radius(lights["color"][i])
        m = np.diag([r, r, r, 1.0])
        m[3, :3] = lights["position"][i]
        lights["model"][i] = m.astype(np.float32).ravel()
    return lights


@pytest.mark.parametrize("w,h,row0,frame_rows", [(200, 120, 0, 120), (131, 77, 0, 77), (160, 45, 37, 120)])
def test_light_coverage_masks_bit_exact(gpu, oracle, w, h, row0, frame_rows):
    """vkr_light_coverage against the oracle's rasterisation: volumes in front of, around and behind
    the camera, partly off screen, one beyond the far plane; whole frames and a strip of a frame."""
    torch, P = _torch(), _passes()
    verts, tris = synth.reference_light_volume()
    f = synth.make_frame(61, w, h, 9, 16, 64, irr_res=8, pref_res=16, lut_res=32)
    f.scene = _camera_scene((2.0, 1.0, -4.0), aspect=w / frame_rows)
    f.lights = _small_lights(np.random.default_rng(5), 9, 0.01)
    # hand-placed cases: around the camera, behind it, far away, beyond the far plane
    for i, (pos, r) in enumerate([((2.0, 1.0, -4.0), 1.0), ((4.0, 2.0, -8.0), 1.5), ((-30.0, -10.0, 60.0), 6.0), ((-60.0, -25.0, 118.0), 5.0)]):
        m = np.diag([r, r, r, 1.0]); m[3, :3] = pos
        f.lights["model"][i] = m.astype(np.float32).ravel()
        f.lights["position"][i] = pos
    want = oracle.light_coverage(f.scene, f.lights, verts, tris, w, h, frame_rows, row0)
    img = oracle.coverage_to_image(want, w, h)
    assert img[0].all() and not img[1].any() and img[2].any() and not img[3].any()
    assert 0 < img[4:].sum() < img[4:].size
    d = P.DeviceFrame.upload(f, gpu)
    got = P.light_coverage(d, torch.from_numpy(verts).to(gpu), torch.from_numpy(tris.astype(np.int32)).to(gpu), frame_rows, row0)
    got = got.cpu().numpy().view(np.uint64)
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"{(got != want).sum()} of {got.size} tile masks differ"


@pytest.mark.parametrize("w,h,row0,frame_rows", [(200, 120, 0, 120), (203, 61, 40, 120)])
def test_light_coverage_flags_name_only_lights_that_cover_every_pixel(gpu, oracle, w, h, row0, frame_rows):
    """The per-light words of vkr_light_coverage: set for volumes that hold the camera well inside, clear for a
    camera outside, near the surface, for a volume cut by the far plane and for one the near plane cuts inside the
    view; wherever a word is set the oracle's rasterisation has every pixel covered, and the masks written with the
    words equal the oracle's bit for bit (ragged widths and a strip included: bits beyond the image stay clear)."""
    torch, P = _torch(), _passes()
    verts, tris = synth.reference_light_volume()
    cam = (2.0, 1.0, -4.0)
    cases = [(cam, 8.0),                      # camera at the centre
             ((3.0, 2.0, -2.0), 12.0),        # camera well inside
             (tuple(np.array(cam) - 7.95 * _toward_origin(cam)), 8.0),   # camera inside, the surface 0.05 ahead of it: the near plane (0.1) cuts the volume in view
             ((2.0, 1.0, 8.0), 3.0),          # camera outside
             ((2.0, 1.0, 40.0), 70.0),        # reaches beyond the far plane (100)
             ((2.5, 1.0, -4.0), 0.45),        # camera outside a small volume beside it
             ((2.0, 1.0, -4.02), 0.05)]       # camera inside a volume smaller than the near plane distance (0.1)
    nl = len(cases)
    f = synth.make_frame(63, w, h, nl, 16, 64, irr_res=8, pref_res=16, lut_res=32)
    f.scene = _camera_scene(cam, aspect=w / frame_rows)
    for i, (pos, r) in enumerate(cases):
        m = np.diag([r, r, r, 1.0]); m[3, :3] = pos
        f.lights["model"][i] = m.astype(np.float32).ravel()
        f.lights["position"][i] = pos
    want = oracle.light_coverage(f.scene, f.lights, verts, tris, w, h, frame_rows, row0)
    full = oracle.coverage_to_image(want, w, h).reshape(nl, -1).all(axis=1)
    d = P.DeviceFrame.upload(f, gpu)
    got, flags = P.light_coverage_with_flags(d, torch.from_numpy(verts).to(gpu), torch.from_numpy(tris.astype(np.int32)).to(gpu), frame_rows, row0)
    flags = flags.cpu().numpy()
    assert np.array_equal(got.cpu().numpy().view(np.uint64), want), "masks written with the flags differ from the oracle's"
    assert flags[0] == 1 and flags[1] == 1, flags
    assert not flags[2:nl].any(), flags
    assert flags[nl] == nl - int(flags[:nl].sum())
    assert all(full[i] for i in range(nl) if flags[i]), "a flagged light leaves a pixel uncovered"
    assert not full[2] and not full[6]  # the near plane clips what the camera sees of these two


def test_shade_skips_the_masks_when_every_light_covers_every_pixel(gpu, oracle):
    """With the reference's light radii and the camera inside every volume the words say so, the masks are all
    ones, and the shaded images equal the ones shaded without masks, whether or not the words are passed on."""
    torch, P = _torch(), _passes()
    w, h, nl = 176, 104, 7
    verts, tris = synth.reference_light_volume()
    dverts, dtris = torch.from_numpy(verts).to(gpu), torch.from_numpy(tris.astype(np.int32)).to(gpu)
    f = synth.make_frame(62, w, h, nl, 32, 128, irr_res=8, pref_res=16, lut_res=32)
    f.scene = _camera_scene((2.0, 1.0, -4.0), aspect=w / h)
    color, thr = torch.empty((h, w, 4), dtype=torch.float16, device=gpu), torch.empty((h, w, 4), dtype=torch.float16, device=gpu)
    d = P.DeviceFrame.upload(f, gpu)
    P.shade(d, color, thr)
    full_c, full_t = P.f16_bits(color).copy(), P.f16_bits(thr).copy()
    d.coverage, d.coverage_flags = P.light_coverage_with_flags(d, dverts, dtris)
    assert d.coverage_flags.cpu().tolist() == [1] * nl + [0]
    assert bool((d.coverage == -1).all())
    for with_flags in (True, False):
        if not with_flags:
            d.coverage_flags = None
        color.zero_(); thr.zero_()
        P.shade(d, color, thr)
        assert np.array_equal(P.f16_bits(color), full_c) and np.array_equal(P.f16_bits(thr), full_t)
    # one local light among them: the count is 1 and the masked loop runs (bit-exact against the oracle with the same masks)
    f.lights["color"][3] *= 0.004
    r = synth.light_volume_radius(f.lights["color"][3])
    m = np.diag([r, r, r, 1.0]); m[3, :3] = f.lights["position"][3]
    f.lights["model"][3] = m.astype(np.float32).ravel()
    d = P.DeviceFrame.upload(f, gpu)
    d.coverage, d.coverage_flags = P.light_coverage_with_flags(d, dverts, dtris)
    assert int(d.coverage_flags[nl]) == 1 and int(d.coverage_flags[3]) == 0
    masks = oracle.light_coverage(f.scene, f.lights, verts, tris, w, h)
    assert np.array_equal(d.coverage.cpu().numpy().view(np.uint64), masks)
    P.shade(d, color, thr)
    want_c, want_t = oracle.shade(f, coverage=masks)
    assert np.array_equal(P.f16_bits(color), want_c) and np.array_equal(P.f16_bits(thr), want_t)


def test_shade_with_light_volume_coverage(gpu, oracle):
    """Composition with VKR_COVERAGE_MESH: small light volumes, so most draws touch part of the frame
    only; bit-exact against the oracle using the same masks, and equal to full coverage when every
    volume contains the camera."""
    torch, P = _torch(), _passes()
    w, h, nl = 176, 104, 7
    verts, tris = synth.reference_light_volume()
    dverts, dtris = torch.from_numpy(verts).to(gpu), torch.from_numpy(tris.astype(np.int32)).to(gpu)
    f = synth.make_frame(62, w, h, nl, 32, 128, irr_res=8, pref_res=16, lut_res=32)
    f.scene = _camera_scene((2.0, 1.0, -4.0), aspect=w / h)
    color = torch.empty((h, w, 4), dtype=torch.float16, device=gpu)
    thr = torch.empty_like(color)

    d = P.DeviceFrame.upload(f, gpu)          # the reference's radii: every volume contains the camera
    P.shade(d, color, thr)
    full_c, full_t = P.f16_bits(color).copy(), P.f16_bits(thr).copy()
    d.coverage = P.light_coverage(d, dverts, dtris)
    assert bool((d.coverage == -1).all())     # every bit set
    P.shade(d, color, thr)
    assert np.array_equal(P.f16_bits(color), full_c) and np.array_equal(P.f16_bits(thr), full_t)

    f.lights = _small_lights(np.random.default_rng(8), nl, 0.02)
    d = P.DeviceFrame.upload(f, gpu)
    d.coverage = P.light_coverage(d, dverts, dtris)
    masks = oracle.light_coverage(f.scene, f.lights, verts, tris, w, h)
    assert np.array_equal(d.coverage.cpu().numpy().view(np.uint64), masks)
    frac = oracle.coverage_to_image(masks, w, h).mean()
    assert 0.02 < frac < 0.9, frac
    P.shade(d, color, thr)
    want_c, want_t = oracle.shade(f, coverage=masks)
    assert_bits_equal(P.f16_bits(color), want_c, "composition colour with light-volume coverage")
    assert_bits_equal(P.f16_bits(thr), want_t, "thresholded colour with light-volume coverage")
    assert not np.array_equal(want_c, oracle.shade(f)[0])  # the masks do change the image


def test_renderer_with_light_volume_mesh(gpu, oracle):
    """Layer 2 with VKR_COVERAGE_MESH: SetLightVolumeMesh + small volumes, EndFrame equals the oracle
    chain shaded under the oracle's masks; order errors without a mesh."""
    from vulkanrenderer_amd.renderer import HipRenderer
    w, h, nl = 136, 88, 5
    verts, tris = synth.reference_light_volume()
    f = synth.make_frame(71, w, h, nl, 16, 64, irr_res=8, pref_res=16, lut_res=32)
    f.scene = _camera_scene((2.0, 1.0, -4.0), aspect=w / h)
    f.lights = _small_lights(np.random.default_rng(3), nl, 0.03)
    masks = oracle.light_coverage(f.scene, f.lights, verts, tris, w, h)
    with HipRenderer(w, h, max_lights=8, dir_shadow_res=64, point_shadow_res=16, env_res=16,
                     irradiance_res=8, prefilter_res=16, prefilter_mips=1, brdf_lut_res=32) as r:
        env = synth.make_env_cube(np.random.default_rng(2), 16)
        r.UploadEnvironmentCubemap(env)
        r.CreateIrradianceCubemap(); r.CreatePrefilteredCubemap(); r.CreateBRDFLUTTexture()
        f.irradiance = r.Read(_lib.TARGET_IRRADIANCE)
        f.prefilter = r.Read(_lib.TARGET_PREFILTER, mip=0)
        f.brdf_lut = r.Read(_lib.TARGET_BRDF_LUT)
        with pytest.raises(_lib.VkrError):
            r.SetLightInfos(f.lights, _lib.COVERAGE_MESH)   # no mesh yet
        bad = tris.copy(); bad[5, 1] = len(verts)
        with pytest.raises(_lib.VkrError):
            r.SetLightVolumeMesh(verts, bad)               # index past the vertex array
        r.SetLightVolumeMesh(verts, tris)
        r.UpdateSceneInfo(f.scene)
        r.UploadGBuffer(f.albedo, f.normal, f.arm, f.position)
        r.UploadDirectionalShadowMap(f.dir_shadow)
        for i in range(nl):
            r.UploadPointShadowMap(i, f.cube_shadow[i])
        r.SetLightInfos(f.lights, _lib.COVERAGE_MESH)
        r.BeginFrame()
        r._call(r._lib.vkr_end_frame)
        want_c, want_t = oracle.shade(f, oracle.skybox(env, f.scene, w, h), coverage=masks)
        assert_bits_equal(r.Read(_lib.TARGET_COLOR, 0), want_c, "colour")
        assert_bits_equal(r.Read(_lib.TARGET_THRESHOLD, 0), want_t, "threshold")


def _device_draws(draws, gpu):
    torch = _torch()
    return [(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(gpu),
             torch.from_numpy(np.ascontiguousarray(t).astype(np.int32)).to(gpu), m) for v, t, m in draws]


@pytest.mark.parametrize("res,tiled", [(64, False), (64, True), (64, None), (90, False)])
def test_point_shadow_maps_bit_exact(gpu, oracle, res, tiled):
    """vkr_point_shadow_maps against the oracle: the closed height-field slab plus a moved, scaled sphere,
    lights above, beside and inside the geometry; row-major output, tiles, and tiles of 25 * depth (None)."""
    torch, P = _torch(), _passes()
    slab = synth.make_scene_mesh(24)
    ball = synth.make_sphere_mesh(16, 8)
    model = np.diag([0.6, 0.6, 0.6, 1.0]); model[3, :3] = (1.0, 0.9, -0.5)
    draws = [(slab[0], slab[1], np.eye(4)), (ball[0], ball[1], model)]
    lights = synth.make_lights(np.random.default_rng(17), 5)
    lights["position"][0] = (0.2, 2.5, 0.1)
    lights["position"][1] = (1.0, 0.9, -0.5)      # inside the sphere
    lights["position"][2] = (-2.9, 0.15, 2.0)     # just above the slab near its rim
    want = oracle.point_shadow_maps(draws, lights, res)
    assert 0.05 < (want < 1.0).mean() < 0.95
    dev_lights = torch.from_numpy(np.frombuffer(lights.tobytes(), dtype=np.uint8).copy()).to(gpu)
    got, layout = P.point_shadow_maps(_device_draws(draws, gpu), dev_lights, len(lights), res, tiled=tiled)
    got = got.cpu().numpy()
    if tiled is not False:
        j, i = np.meshgrid(np.arange(res), np.arange(res), indexing="ij")
        index = (i >> 3) * (res * 8) + j * 8 + (i & 7)  # 8x4-texel tiles, stored column by column
        got = got.reshape(len(lights), 6, -1)[:, :, index.ravel()].reshape(want.shape)
    if tiled is None:
        assert layout == _lib.CUBE_LAYOUT_TILED_8X4_X25
        want = want * np.float32(25.0)
    bad = got.view(np.uint32) != want.view(np.uint32)
    assert not bad.any(), f"{bad.sum()} of {bad.size} texels differ, first at {np.argwhere(bad)[0]}"


def test_dir_shadow_map_bit_exact(gpu, oracle):
    torch, P = _torch(), _passes()
    slab = synth.make_scene_mesh(24)
    ball = synth.make_sphere_mesh(16, 8)
    model = np.diag([0.6, 0.6, 0.6, 1.0]); model[3, :3] = (1.0, 0.9, -0.5)
    draws = [(slab[0], slab[1], np.eye(4)), (ball[0], ball[1], model)]
    ls = synth.default_scene_info()["light_space_matrix"]
    want = oracle.dir_shadow_map(draws, ls, 192)
    assert 0.02 < (want < 1.0).mean() < 0.9
    got = P.dir_shadow_map(_device_draws(draws, gpu), ls, 192, gpu).cpu().numpy()
    bad = got.view(np.uint32) != want.view(np.uint32)
    assert not bad.any(), f"{bad.sum()} of {bad.size} texels differ"


def _sliver_soup(seed, n, centre_radius=(0.8, 3.0), spread=3.0):
    """n triangles of every size from a thousandth of a sample to tens of samples, many of them far thinner than a
    sample: the ones whose box of sample centres is empty or a single row or column."""
    rng = np.random.default_rng(seed)
    direction = rng.normal(size=(n, 3)); direction /= np.linalg.norm(direction, axis=1, keepdims=True)
    centre = direction * rng.uniform(*centre_radius, size=(n, 1)) * np.array([spread / 3.0, 1.0, spread / 3.0])
    size = 10.0 ** rng.uniform(-3.5, -0.3, size=(n, 1))
    thin = 10.0 ** rng.uniform(-3.0, 0.0, size=(n, 1))
    e1 = rng.normal(size=(n, 3)) * size
    e2 = rng.normal(size=(n, 3)) * size * thin
    verts = np.stack([centre, centre + e1, centre + e1 * rng.uniform(0, 1, size=(n, 1)) + e2], axis=1).reshape(-1, 3).astype(np.float32)
    tris = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    return verts, tris


@pytest.mark.parametrize("res,tiled", [(64, False), (136, None)])
def test_shadow_maps_of_slivers(gpu, oracle, res, tiled):
    """Triangles thinner and smaller than a sample, around two lights and under the directional one.  The kernels walk
    the box of sample centres a triangle's extent can hold (empty for many of these); the oracle walks that box with a
    margin of two samples on every side, so equal bits also say that the tight boxes lose no sample."""
    torch, P = _torch(), _passes()
    verts, tris = _sliver_soup(5, 6000)
    draws = [(verts, tris, np.eye(4))]
    lights = synth.make_lights(np.random.default_rng(3), 2)
    lights["position"][0] = (0.0, 0.0, 0.0)
    lights["position"][1] = (0.3, -0.2, 0.25)
    want = oracle.point_shadow_maps(draws, lights, res)
    assert 0.01 < (want < 1.0).mean() < 0.9
    dev_lights = torch.from_numpy(np.frombuffer(lights.tobytes(), dtype=np.uint8).copy()).to(gpu)
    got, layout = P.point_shadow_maps(_device_draws(draws, gpu), dev_lights, len(lights), res, tiled=tiled)
    got = got.cpu().numpy()
    if tiled is not False:
        j, i = np.meshgrid(np.arange(res), np.arange(res), indexing="ij")
        index = (i >> 3) * (res * 8) + j * 8 + (i & 7)
        got = got.reshape(len(lights), 6, -1)[:, :, index.ravel()].reshape(want.shape)
        want = want * np.float32(25.0)
    bad = got.view(np.uint32) != want.view(np.uint32)
    assert not bad.any(), f"{bad.sum()} of {bad.size} texels differ, first at {np.argwhere(bad)[0]}"
    ls = synth.default_scene_info()["light_space_matrix"]
    want = oracle.dir_shadow_map(draws, ls, 200)
    assert (want < 1.0).any()
    got = P.dir_shadow_map(_device_draws(draws, gpu), ls, 200, gpu).cpu().numpy()
    bad = got.view(np.uint32) != want.view(np.uint32)
    assert not bad.any(), f"directional map: {bad.sum()} of {bad.size} texels differ"


def test_dir_shadow_map_with_corners_on_sample_centres(gpu, oracle):
    """Triangles whose corners project onto sample centres of the directional map (as closely as binary32 positions
    allow): samples on the rim of a box of sample centres, from inside and from outside."""
    torch, P = _torch(), _passes()
    res = 96
    ls = np.asarray(synth.default_scene_info()["light_space_matrix"], dtype=np.float64).reshape(4, 4)
    inv = np.linalg.inv(ls)
    rng = np.random.default_rng(8)
    n = 1500
    ij = rng.integers(4, res - 4, size=(n, 3, 2)) + rng.integers(-3, 4, size=(n, 1, 2))
    ndc = (ij + 0.5) * (2.0 / res) - 1.0
    z = rng.uniform(0.2, 0.8, size=(n, 3, 1))
    clip = np.concatenate([ndc, z, np.ones((n, 3, 1))], axis=2)  # orthographic: w = 1
    world = clip @ inv                                            # row-vector convention
    verts = (world[..., :3] / world[..., 3:]).reshape(-1, 3).astype(np.float32)
    tris = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    draws = [(verts, tris, np.eye(4))]
    want = oracle.dir_shadow_map(draws, ls.astype(np.float32).ravel(), res)
    assert 0.05 < (want < 1.0).mean()
    got = P.dir_shadow_map(_device_draws(draws, gpu), ls.astype(np.float32).ravel(), res, gpu).cpu().numpy()
    bad = got.view(np.uint32) != want.view(np.uint32)
    assert not bad.any(), f"{bad.sum()} of {bad.size} texels differ, first at {np.argwhere(bad)[0]}"


def test_generated_shadow_maps_feed_the_shading_pass(gpu, oracle):
    """End to end: shadow maps made by the library from the scene mesh (tiled cubes straight from the pass)
    shade to the bits the oracle gets from its own maps."""
    torch, P = _torch(), _passes()
    w, h, nl = 128, 72, 3
    slab = synth.make_scene_mesh(24)
    draws = [(slab[0], slab[1], np.eye(4))]
    f = synth.make_frame(81, w, h, nl, 64, 128, irr_res=8, pref_res=16, lut_res=32)
    f.cube_shadow = oracle.point_shadow_maps(draws, f.lights, 64)
    f.dir_shadow = oracle.dir_shadow_map(draws, f.scene["light_space_matrix"], 128)
    want_c, want_t = oracle.shade(f)
    d = P.DeviceFrame.upload(f, gpu)
    ddraws = _device_draws(draws, gpu)
    d.cube_shadow, d.cube_layout = P.point_shadow_maps(ddraws, d.lights, nl, 64)
    assert d.cube_layout == _lib.CUBE_LAYOUT_TILED_8X4_X25
    d.dir_shadow = P.dir_shadow_map(ddraws, f.scene["light_space_matrix"], 128, gpu)
    color = torch.empty((h, w, 4), dtype=torch.float16, device=gpu)
    thr = torch.empty_like(color)
    P.shade(d, color, thr)
    assert_bits_equal(P.f16_bits(color), want_c, "colour under generated shadow maps")
    assert_bits_equal(P.f16_bits(thr), want_t, "threshold under generated shadow maps")
    lit = oracle.shade(dataclasses_replace(f))[0]
    assert (f.cube_shadow < 1.0).any() and lit.shape == want_c.shape


def dataclasses_replace(f):
    import dataclasses
    return dataclasses.replace(f)


def test_renderer_draw_calls_make_the_shadow_maps(gpu, oracle):
    """Layer 2: CreateMesh + DrawModel between BeginFrame and EndFrame; EndFrame renders the depth cubes and
    the directional map from the draw calls and shades with them (VulkanRenderer.EndFrame's order)."""
    from vulkanrenderer_amd.renderer import HipRenderer
    w, h, nl = 112, 64, 3
    slab = synth.make_scene_mesh(20)
    ball = synth.make_sphere_mesh(12, 6)
    model = np.diag([0.5, 0.5, 0.5, 1.0]); model[3, :3] = (-0.8, 0.8, 0.6)
    f = synth.make_frame(95, w, h, nl, 32, 64, irr_res=8, pref_res=16, lut_res=32)
    draws = [(slab[0], slab[1], np.eye(4)), (ball[0], ball[1], model)]
    f.cube_shadow = oracle.point_shadow_maps(draws, f.lights, 32)
    f.dir_shadow = oracle.dir_shadow_map(draws, f.scene["light_space_matrix"], 64)
    with HipRenderer(w, h, max_lights=4, dir_shadow_res=64, point_shadow_res=32, env_res=16,
                     irradiance_res=8, prefilter_res=16, prefilter_mips=1, brdf_lut_res=32) as r:
        env = synth.make_env_cube(np.random.default_rng(2), 16)
        r.UploadEnvironmentCubemap(env)
        r.CreateIrradianceCubemap(); r.CreatePrefilteredCubemap(); r.CreateBRDFLUTTexture()
        f.irradiance = r.Read(_lib.TARGET_IRRADIANCE)
        f.prefilter = r.Read(_lib.TARGET_PREFILTER, mip=0)
        f.brdf_lut = r.Read(_lib.TARGET_BRDF_LUT)
        m_slab, m_ball = r.CreateMesh(*slab), r.CreateMesh(*ball)
        with pytest.raises(_lib.VkrError):
            r.DrawModel(m_slab, np.eye(4))          # outside a frame
        want_c, want_t = oracle.shade(f, oracle.skybox(env, f.scene, w, h))
        for _ in range(2):                           # both frames in flight
            frame = r.CurrentFrame
            r.UpdateSceneInfo(f.scene)
            r.UploadGBuffer(f.albedo, f.normal, f.arm, f.position)
            r.SetLightInfos(f.lights)
            r.BeginFrame()
            r.DrawModel(m_slab, np.eye(4))
            r.DrawModel(m_ball, model)
            r._call(r._lib.vkr_end_frame)
            assert_bits_equal(r.Read(_lib.TARGET_COLOR, frame), want_c, "colour")
            assert_bits_equal(r.Read(_lib.TARGET_THRESHOLD, frame), want_t, "threshold")
        with pytest.raises(_lib.VkrError):
            r.DrawModel(99, np.eye(4))


def _gpass_scene(w, h):
    slab = synth.make_scene_mesh(20)
    ball = synth.make_sphere_mesh(16, 8)
    model = np.diag([0.7, 0.7, 0.7, 1.0]); model[3, :3] = (0.6, 0.8, -1.2)
    draws = []
    for k, ((verts, tris), m, scale) in enumerate(((slab, np.eye(4), 0.25), (ball, model, 0.5))):
        uv, nrm, tan = synth.mesh_attributes(verts, tris, uv_scale=scale)
        alb, nmap, arm = synth.make_material(np.random.default_rng(40 + k), 32 + 16 * k)
        draws.append(dict(positions=verts, texcoords=uv, normals=nrm, tangents=tan, triangles=tris, model=m,
                          albedo=alb, normal=nmap, arm=arm))
    return draws


def _gpass_device_draws(draws, gpu):
    torch = _torch()
    out = []
    for d in draws:
        dd = {k: torch.from_numpy(np.ascontiguousarray(d[k], dtype=np.float32)).to(gpu) for k in ("positions", "texcoords", "normals", "tangents")}
        dd["triangles"] = torch.from_numpy(np.ascontiguousarray(d["triangles"]).astype(np.int32)).to(gpu)
        for k in ("albedo", "normal", "arm"):
            dd[k] = torch.from_numpy(np.ascontiguousarray(d[k])).to(gpu)
        dd["model"] = d["model"]
        out.append(dd)
    return out


@pytest.mark.parametrize("w,h,row0,frame_rows", [(192, 108, 0, 108), (157, 41, 30, 108)])
def test_gbuffer_bit_exact(gpu, oracle, w, h, row0, frame_rows):
    """vkr_gbuffer against the oracle's geometry pass: two draws with different materials, one occluding the
    other; a whole frame and a strip of it."""
    torch, P = _torch(), _passes()
    draws = _gpass_scene(w, h)
    scene = _camera_scene((2.0, 1.6, -4.0), aspect=w / frame_rows)
    want = oracle.gbuffer(draws, scene, w, h, frame_rows, row0)
    covered = want[0][..., 3] != 0
    assert 0.2 < covered.mean() < 0.95
    got = P.gbuffer(_gpass_device_draws(draws, gpu), scene, w, h, gpu, frame_rows, row0)
    for name, g, wv in zip(("albedo", "normal", "arm", "position"), got, want):
        assert_bits_equal(P.f16_bits(g), wv, f"G-buffer {name}")


@pytest.mark.parametrize("w,h,row0,frame_rows", [(157, 41, 30, 108), (64, 200, 0, 200)])
def test_gbuffer_of_slivers(gpu, oracle, w, h, row0, frame_rows):
    """The geometry pass over triangles thinner and smaller than a pixel, on frames far from square (the box of
    sample centres is taken on the frame's own width x rows grid) and on a strip of one."""
    torch, P = _torch(), _passes()
    verts, tris = _sliver_soup(12, 5000, centre_radius=(0.0, 1.5))
    uv, nrm, tan = synth.mesh_attributes(verts, tris, uv_scale=0.5)
    alb, nmap, arm = synth.make_material(np.random.default_rng(77), 32)
    draws = [dict(positions=verts, texcoords=uv, normals=nrm, tangents=tan, triangles=tris, model=np.eye(4), albedo=alb, normal=nmap, arm=arm)]
    scene = _camera_scene((2.0, 1.6, -4.0), aspect=w / frame_rows)
    want = oracle.gbuffer(draws, scene, w, h, frame_rows, row0)
    assert 0.02 < (want[0][..., 3] != 0).mean() < 0.95
    got = P.gbuffer(_gpass_device_draws(draws, gpu), scene, w, h, gpu, frame_rows, row0)
    for name, g, wv in zip(("albedo", "normal", "arm", "position"), got, want):
        assert_bits_equal(P.f16_bits(g), wv, f"G-buffer {name}")


def test_gbuffer_with_and_without_triangle_records(gpu):
    """vkr_gbuffer with the larger workspace (corners shaded once per triangle, records read by the resolve step) and
    with the smallest one (corners shaded per pixel): the same four images, bit for bit."""
    torch, P = _torch(), _passes()
    w, h = 200, 120
    draws = _gpass_device_draws(_gpass_scene(w, h), gpu)
    scene = _camera_scene((2.0, 1.6, -4.0), aspect=w / h)
    with_records = P.gbuffer(draws, scene, w, h, gpu, triangle_records=True)
    without = P.gbuffer(draws, scene, w, h, gpu, triangle_records=False)
    assert (with_records[0][..., 3] != 0).float().mean() > 0.2
    for name, a, b in zip(("albedo", "normal", "arm", "position"), with_records, without):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), f"G-buffer {name} differs between the two forms"


def test_gbuffer_draws_of_few_triangles(gpu, oracle):
    """Draws of 1, 2, 3, 5 and 12 triangles cut from the slab (the visibility kernel hands lane s the triangle
    s * stride mod n: the stride is 1 below four triangles and coprime to n above), overlapping each other, so that
    equal depths are settled by submission order."""
    torch, P = _torch(), _passes()
    w, h = 96, 54
    verts, tris = synth.make_scene_mesh(6)
    uv, nrm, tan = synth.mesh_attributes(verts, tris, uv_scale=0.5)
    draws, start = [], 0
    for k, n in enumerate((1, 2, 3, 5, 12, 5)):
        sel = tris[start:start + n] if k < 5 else tris[8:13]   # the last draw repeats triangles of an earlier one
        start += n if k < 5 else 0
        alb, nmap, arm = synth.make_material(np.random.default_rng(60 + k), 16)
        draws.append(dict(positions=verts, texcoords=uv, normals=nrm, tangents=tan, triangles=np.ascontiguousarray(sel), model=np.eye(4),
                          albedo=alb, normal=nmap, arm=arm))
    scene = _camera_scene((1.0, 3.0, -3.5), aspect=w / h)
    want = oracle.gbuffer(draws, scene, w, h, h, 0)
    assert (want[0][..., 3] != 0).any()
    got = P.gbuffer(_gpass_device_draws(draws, gpu), scene, w, h, gpu, h, 0)
    for name, g, wv in zip(("albedo", "normal", "arm", "position"), got, want):
        assert_bits_equal(P.f16_bits(g), wv, f"G-buffer {name}")


def test_whole_frame_from_meshes(gpu, oracle):
    """Meshes in, image out: geometry pass, shadow passes, composition, bloom and tonemap all by the library,
    against the same chain in the oracle."""
    torch, P = _torch(), _passes()
    w, h, nl = 160, 90, 3
    draws = _gpass_scene(w, h)
    f = synth.make_frame(77, w, h, nl, 32, 64, irr_res=8, pref_res=16, lut_res=32)
    f.scene = _camera_scene((2.0, 1.6, -4.0), aspect=w / h)
    f.albedo, f.normal, f.arm, f.position = oracle.gbuffer(draws, f.scene, w, h)
    shadow_draws = [(d["positions"], d["triangles"], d["model"]) for d in draws]
    f.cube_shadow = oracle.point_shadow_maps(shadow_draws, f.lights, 32)
    f.dir_shadow = oracle.dir_shadow_map(shadow_draws, f.scene["light_space_matrix"], 64)
    want = oracle.frame_chain(f)

    dev = P.DeviceFrame.upload(f, gpu)
    ddraws = _gpass_device_draws(draws, gpu)
    ga, gn, gm, gp = P.gbuffer(ddraws, f.scene, w, h, gpu)
    dev.albedo, dev.normal, dev.arm, dev.position = ga, gn, gm, gp
    sdraws = [(d["positions"], d["triangles"], d["model"]) for d in ddraws]
    dev.cube_shadow, dev.cube_layout = P.point_shadow_maps(sdraws, dev.lights, nl, 32)
    dev.dir_shadow = P.dir_shadow_map(sdraws, f.scene["light_space_matrix"], 64, gpu)
    color = torch.empty((h, w, 4), dtype=torch.float16, device=gpu)
    thr, b1 = torch.empty_like(color), torch.empty_like(color)
    ldr = torch.empty((h, w, 4), dtype=torch.uint8, device=gpu)
    P.shade(dev, color, thr)
    P.bloom_blur(thr, b1, True, _lib.WRAP_REPEAT)
    P.bloomv_postprocess(b1, color, ldr)
    assert_bits_equal(P.f16_bits(color), want["color"], "colour")
    assert np.abs(ldr.cpu().numpy().astype(np.int32) - want["ldr"].astype(np.int32)).max() <= 1
    assert (want["color"][..., :3] != 0).any()


def test_renderer_renders_a_frame_from_draw_calls_only(gpu, oracle):
    """Layer 2 with nothing uploaded but scene constants, lights, meshes and materials: EndFrame runs the
    shadow passes, the geometry pass, the light-volume coverage, composition, bloom and tonemap."""
    from vulkanrenderer_amd.renderer import HipRenderer
    w, h, nl = 144, 80, 3
    draws = _gpass_scene(w, h)
    f = synth.make_frame(78, w, h, nl, 32, 64, irr_res=8, pref_res=16, lut_res=32)
    f.scene = _camera_scene((2.0, 1.6, -4.0), aspect=w / h)
    f.lights = _small_lights(np.random.default_rng(6), nl, 0.05)
    sphere = synth.reference_light_volume()
    with HipRenderer(w, h, max_lights=4, dir_shadow_res=64, point_shadow_res=32, env_res=16,
                     irradiance_res=8, prefilter_res=16, prefilter_mips=1, brdf_lut_res=32) as r:
        env = synth.make_env_cube(np.random.default_rng(2), 16)
        r.UploadEnvironmentCubemap(env)
        r.CreateIrradianceCubemap(); r.CreatePrefilteredCubemap(); r.CreateBRDFLUTTexture()
        f.irradiance = r.Read(_lib.TARGET_IRRADIANCE)
        f.prefilter = r.Read(_lib.TARGET_PREFILTER, mip=0)
        f.brdf_lut = r.Read(_lib.TARGET_BRDF_LUT)
        r.SetLightVolumeMesh(*sphere)
        handles = []
        for d in draws:
            mesh = r.CreateMesh(d["positions"], d["triangles"], d["texcoords"], d["normals"], d["tangents"])
            handles.append((mesh, r.CreateMaterial(d["albedo"], d["normal"], d["arm"]), d["model"]))
        # the oracle's frame
        f.albedo, f.normal, f.arm, f.position = oracle.gbuffer(draws, f.scene, w, h)
        sd = [(d["positions"], d["triangles"], d["model"]) for d in draws]
        f.cube_shadow = oracle.point_shadow_maps(sd, f.lights, 32)
        f.dir_shadow = oracle.dir_shadow_map(sd, f.scene["light_space_matrix"], 64)
        masks = oracle.light_coverage(f.scene, f.lights, sphere[0], sphere[1], w, h)
        sky = oracle.skybox(env, f.scene, w, h)
        color, thr = oracle.shade(f, sky, coverage=masks)
        b2 = oracle.bloom_blur(oracle.bloom_blur(thr, True, oracle.WRAP_REPEAT), False, oracle.WRAP_CLAMP)
        want_ldr = oracle.postprocess(color, b2)
        for _ in range(2):
            frame = r.CurrentFrame
            r.UpdateSceneInfo(f.scene)
            r.SetLightInfos(f.lights, _lib.COVERAGE_MESH)
            r.BeginFrame()
            for mesh, material, model in handles:
                r.DrawModel(mesh, model, material)
            r._call(r._lib.vkr_end_frame)
            assert_bits_equal(r.Read(_lib.TARGET_COLOR, frame), color, "colour")
            assert_bits_equal(r.Read(_lib.TARGET_THRESHOLD, frame), thr, "threshold")
            ldr = r.Read(_lib.TARGET_LDR, frame)
            assert np.abs(ldr.astype(np.int32) - want_ldr.astype(np.int32)).max() <= 1
        assert (color[..., :3] != 0).any()
        bare = f.albedo[..., 3] == 0  # texels no triangle covers show the sky
        assert bare.any() and not bare.all() and np.array_equal(color[bare], sky[bare])


@pytest.mark.parametrize("n_strips", [2, 3])
def test_strip_contexts_of_one_process_equal_one_context(gpu, oracle, n_strips):
    """Layer 2 on several GPUs from one process, rehearsed with every context on this GPU: a frame from draw calls
    (shadow passes, geometry pass, skybox, light volumes) rendered as strips by `n_strips` renderer objects whose
    boundary rows move with vkr_strip_exchange, against the same frame rendered by one object; two frames per
    slot so that a strip's lent rows are overwritten while frames are in flight.  Also the ragged plan, the
    order errors and a strip whose halo rows travel through pointers from vkr_strip_halo."""
    from vulkanrenderer_amd.renderer import HipRenderer, strip_exchange, strip_plan
    torch = _torch()
    w, h, nl = 144, 83, 3
    draws = _gpass_scene(w, h)
    f = synth.make_frame(78, w, h, nl, 32, 64, irr_res=8, pref_res=16, lut_res=32)
    # one light set per frame: bright ones (bloom to spread across the cuts) and small volumes (masks) in turn
    lights = [synth.make_lights(np.random.default_rng(6 + k), nl) if k % 2 == 0 else _small_lights(np.random.default_rng(6 + k), nl, 0.3) for k in range(4)]
    bloom_seen = 0
    sphere = synth.reference_light_volume()
    env = synth.make_env_cube(np.random.default_rng(2), 16)
    kw = dict(max_lights=4, dir_shadow_res=64, point_shadow_res=32, env_res=16, irradiance_res=8, prefilter_res=16, prefilter_mips=1, brdf_lut_res=32)
    plans = [strip_plan(h, n_strips, k) for k in range(n_strips)]
    assert plans[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(plans, plans[1:])) and plans[-1][0] + plans[-1][1] == h
    with pytest.raises(_lib.VkrError):
        strip_plan(10, 3, 0)  # strips of fewer than four rows
    whole = HipRenderer(w, h, **kw)
    strips = [HipRenderer(w, rows, frame_rows=h, row0=row0, **kw) for row0, rows in plans]
    try:
        for r in [whole] + strips:
            r.UploadEnvironmentCubemap(env)
            r.CreateIrradianceCubemap(); r.CreatePrefilteredCubemap(); r.CreateBRDFLUTTexture()
            r.SetLightVolumeMesh(*sphere)
            r.handles = []
            for d in draws:
                mesh = r.CreateMesh(d["positions"], d["triangles"], d["texcoords"], d["normals"], d["tangents"])
                r.handles.append((mesh, r.CreateMaterial(d["albedo"], d["normal"], d["arm"]), d["model"]))

        def begin(r, k):
            scene = _camera_scene((2.0 + 0.1 * k, 1.6, -4.0), aspect=w / h)
            if k % 2 == 0:
                scene["dir_light_color"] = (40.0, 36.0, 30.0)  # a sun bright enough to pass the bloom threshold
            r.UpdateSceneInfo(scene)
            r.SetLightInfos(lights[k], _lib.COVERAGE_MESH)
            r.BeginFrame()
            for mesh, material, model in r.handles:
                r.DrawModel(mesh, model, material)

        for k in range(4):
            slot = whole.CurrentFrame
            begin(whole, k)
            whole._call(whole._lib.vkr_end_frame)
            for r in strips:
                begin(r, k)
            with pytest.raises(_lib.VkrError):      # a strip cannot end its frame alone
                strips[0]._call(strips[0]._lib.vkr_end_frame)
            with pytest.raises(_lib.VkrError):      # nothing shaded yet
                strip_exchange(strips[0], strips[1])
            for r in strips:
                r._call(r._lib.vkr_strip_shade)
            with pytest.raises(_lib.VkrError):      # not neighbours in this order
                strip_exchange(strips[1], strips[0])
            if k % 2 == 0:
                for up, lo in zip(strips, strips[1:]):
                    strip_exchange(up, lo)
            else:  # the same rows through the pointers a host with its own transport would use
                for up, lo in zip(strips, strips[1:]):
                    up.WaitIdle(); lo.WaitIdle()
                    us, ur, n = up.StripHalo(_lib.STRIP_BELOW)
                    ls, lr, n2 = lo.StripHalo(_lib.STRIP_ABOVE)
                    assert n == n2 == w * 4 * 8
                    _lib.check(_hip_memcpy_d2d(lr, us, n)); _lib.check(_hip_memcpy_d2d(ur, ls, n))
            with pytest.raises(_lib.VkrError):
                strips[0].StripHalo(_lib.STRIP_ABOVE)   # the top strip has no neighbour above
            for r in strips:
                r.StripFinish()
            for target in (_lib.TARGET_COLOR, _lib.TARGET_THRESHOLD, _lib.TARGET_LDR):
                got = np.concatenate([r.Read(target, slot)[:rows] for r, (_, rows) in zip(strips, plans)], 0)
                assert np.array_equal(got, whole.Read(target, slot)), f"frame {k}, target {target}: the stacked strips differ from the whole frame"
            thr = whole.Read(_lib.TARGET_THRESHOLD, slot)[..., :3]
            cuts = [row0 for row0, _ in plans[1:]]
            bloom_seen += int(any((thr[c - 4:c + 4] != 0).any() for c in cuts))  # bright pixels within reach of a cut
        assert bloom_seen >= 2
    finally:
        for r in [whole] + strips:
            r.Dispose()


def _hip_memcpy_d2d(dst, src, n):
    """hipMemcpy device to device through the HIP runtime the library links (test plumbing for vkr_strip_halo)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    return -abs(hip.hipMemcpy(dst, src, n, 3))  # 3 = hipMemcpyDeviceToDevice; 0 on success


def test_large_triangles_are_shared_by_the_wave(gpu, oracle):
    """Triangles whose bounding boxes hold far more samples than one lane should walk (a floor of two
    triangles under the scene, seen by the camera and by the lights) give the oracle's bits too."""
    torch, P = _torch(), _passes()
    w, h = 256, 144
    floor = (np.array([[-6, -0.6, -6], [6, -0.6, -6], [6, -0.6, 6], [-6, -0.6, 6], [-6, -0.9, -6], [6, -0.9, -6], [6, -0.9, 6], [-6, -0.9, 6]], dtype=np.float32),
             np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [1, 2, 6], [1, 6, 5], [2, 3, 7], [2, 7, 6], [3, 0, 4], [3, 4, 7]], dtype=np.uint32))
    draws = _gpass_scene(w, h)
    uv, nrm, tan = synth.mesh_attributes(*floor, uv_scale=0.1)
    alb, nmap, arm = synth.make_material(np.random.default_rng(50), 32)
    draws.append(dict(positions=floor[0], texcoords=uv, normals=nrm, tangents=tan, triangles=floor[1], model=np.eye(4),
                      albedo=alb, normal=nmap, arm=arm))
    scene = _camera_scene((2.0, 1.6, -4.0), aspect=w / h)
    want = oracle.gbuffer(draws, scene, w, h)
    assert (want[0][..., 3] != 0).mean() > 0.6
    got = P.gbuffer(_gpass_device_draws(draws, gpu), scene, w, h, gpu)
    for name, g, wv in zip(("albedo", "normal", "arm", "position"), got, want):
        assert_bits_equal(P.f16_bits(g), wv, f"G-buffer {name}")
    lights = synth.make_lights(np.random.default_rng(18), 2)
    sd = [(d["positions"], d["triangles"], d["model"]) for d in draws]
    want_cube = oracle.point_shadow_maps(sd, lights, 128)
    dev_lights = torch.from_numpy(np.frombuffer(lights.tobytes(), dtype=np.uint8).copy()).to(gpu)
    got_cube, _ = P.point_shadow_maps(_device_draws(sd, gpu), dev_lights, 2, 128, tiled=False)
    assert np.array_equal(got_cube.cpu().numpy().view(np.uint32), want_cube.view(np.uint32))
    ls = synth.default_scene_info()["light_space_matrix"]
    want_dir = oracle.dir_shadow_map(sd, ls, 256)
    got_dir = P.dir_shadow_map(_device_draws(sd, gpu), ls, 256, gpu)
    assert np.array_equal(got_dir.cpu().numpy().view(np.uint32), want_dir.view(np.uint32))


@pytest.mark.parametrize("mode", ["end_frame", "stages"])
def test_cpp_host_program_renders_from_draw_calls(gpu, oracle, tmp_path, mode):
    """The C++ mirror driven like the reference's render loop with models only: CreateMesh, CreateMaterial,
    DrawModel between BeginFrame and EndFrame; its frame equals the oracle's geometry + shadow + shading chain.
    "stages": the seven passes recorded stage by stage (point shadows, geometry, directional shadow, composition,
    bloom 1, bloom 2, post-process) and submitted once, as VulkanRenderer.EndFrame is written."""
    import os
    import subprocess
    exe = _cpp_host()
    w, h, nl, dres, cres, eres, ires, pres, lres = 120, 64, 2, 64, 16, 16, 4, 8, 16
    f = synth.make_frame(53, w, h, nl, cres, dres, irr_res=ires, pref_res=pres, lut_res=lres)
    f.scene = _camera_scene((2.0, 1.6, -4.0), aspect=w / h)
    env = synth.make_env_cube(np.random.default_rng(8), eres)
    draws = _gpass_scene(w, h)
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as fh:
        fh.write(np.array([w, h, nl, dres, cres, eres, ires, pres, lres], dtype=np.int32).tobytes())
        fh.write(f.scene.tobytes())
        for l in f.lights:
            fh.write(np.asarray(l["position"][:3], dtype=np.float32).tobytes())
            fh.write(np.asarray(l["color"][:3], dtype=np.float32).tobytes())
        for a in (f.albedo, f.normal, f.arm, f.position, f.dir_shadow, f.cube_shadow, env):
            fh.write(np.ascontiguousarray(a).tobytes())
        fh.write(np.int32(len(draws)).tobytes())
        for d in draws:
            fh.write(np.array([len(d["positions"]), len(d["triangles"])], dtype=np.int32).tobytes())
            for k in ("positions", "texcoords", "normals", "tangents"):
                fh.write(np.ascontiguousarray(d[k], dtype=np.float32).tobytes())
            fh.write(np.ascontiguousarray(d["triangles"], dtype=np.uint32).tobytes())
            fh.write(np.ascontiguousarray(d["model"], dtype=np.float32).tobytes())
            for k in ("albedo", "normal", "arm"):
                fh.write(np.array([d[k].shape[1], d[k].shape[0]], dtype=np.int32).tobytes())
                fh.write(np.ascontiguousarray(d[k], dtype=np.uint8).tobytes())
    proc = subprocess.run([exe, str(inp), str(outp)] + (["stages"] if mode == "stages" else []), capture_output=True, text=True, timeout=100)
    assert proc.returncode == 0, proc.stderr
    raw = np.fromfile(outp, dtype=np.uint8)
    sizes = [w * h * 8, w * h * 8, w * h * 4, 6 * ires * ires * 8, 6 * pres * pres * 8, lres * lres * 4]
    parts = np.split(raw, np.cumsum(sizes)[:-1])
    color, thr = (p.view(np.uint16).reshape(h, w, 4) for p in parts[:2])
    ldr = parts[2].reshape(h, w, 4)
    f.irradiance = parts[3].view(np.uint16).reshape(6, ires, ires, 4)
    f.prefilter = parts[4].view(np.uint16).reshape(6, pres, pres, 4)
    f.brdf_lut = parts[5].view(np.uint16).reshape(lres, lres, 2)
    # vkr::Light::ToInfo gives the lights the reference's volume radius; the shading only reads position and colour
    f.albedo, f.normal, f.arm, f.position = oracle.gbuffer(draws, f.scene, w, h)
    sd = [(d["positions"], d["triangles"], d["model"]) for d in draws]
    f.cube_shadow = oracle.point_shadow_maps(sd, f.lights, cres)
    f.dir_shadow = oracle.dir_shadow_map(sd, f.scene["light_space_matrix"], dres)
    want = oracle.frame_chain(f, oracle.skybox(env, f.scene, w, h))
    assert_bits_equal(color, want["color"], "colour")
    assert_bits_equal(thr, want["threshold"], "threshold")
    assert np.abs(ldr.astype(np.int32) - want["ldr"].astype(np.int32)).max() <= 1


@pytest.mark.parametrize("w,h,row0,frame_rows", [(160, 90, 0, 90), (131, 37, 19, 77), (64, 1, 63, 64)])
def test_skybox_bit_exact(gpu, oracle, w, h, row0, frame_rows):
    """The skybox draw from a row-major and from a 4x4-tiled environment cube, whole frames and strips, for
    cameras that see face edges and corners of the cube; and as the background of the composition pass."""
    torch, P = _torch(), _passes()
    n = 32
    env = synth.make_env_cube(np.random.default_rng(91), n)
    dev_env = P.to_device_f16(env, gpu)
    tiled = P.tile_env_cube(dev_env)
    for cam, target in (((2.0, 1.6, -4.0), (0.0, 0.0, 0.0)), ((0.0, 0.0, 0.0), (1.0, 1.0, 1.0)), ((1.0, 5.0, 1.0), (1.0, 0.0, 1.001))):
        scene = _camera_scene(cam, target, aspect=w / frame_rows)
        want = oracle.skybox(env, scene, w, h, frame_rows, row0)
        got = P.skybox(dev_env, scene, w, h, frame_rows, row0)
        assert_bits_equal(P.f16_bits(got), want, f"skybox, camera {cam}")
        got_t = P.skybox(tiled, scene, w, h, frame_rows, row0, env_layout=_lib.ENV_LAYOUT_TILED_4X4)
        assert_bits_equal(P.f16_bits(got_t), want, f"skybox from the tiled cube, camera {cam}")
        assert (want[..., 3] == 0x3C00).all() and len(np.unique(want[..., 0])) > 1
    if row0 == 0 and h == frame_rows:
        f = synth.make_frame(92, w, h, 2, 16, 64, irr_res=8, pref_res=16, lut_res=32)
        f.scene = _camera_scene((2.0, 1.6, -4.0), aspect=w / h)
        sky = oracle.skybox(env, f.scene, w, h)
        want_c, want_t = oracle.shade(f, background=sky)
        d = P.DeviceFrame.upload(f, gpu)
        d.background = P.skybox(tiled, f.scene, w, h, env_layout=_lib.ENV_LAYOUT_TILED_4X4)
        color = torch.empty((h, w, 4), dtype=torch.float16, device=gpu)
        thr = torch.empty_like(color)
        P.shade(d, color, thr)
        assert_bits_equal(P.f16_bits(color), want_c, "colour over the sky")
        assert_bits_equal(P.f16_bits(thr), want_t, "threshold over the sky")
    bad = _camera_scene((0.0, 0.0, 1.0))
    bad["camera_projection"] = np.zeros(16, dtype=np.float32)
    with pytest.raises(_lib.VkrError):
        P.skybox(dev_env, bad, w, h, frame_rows, row0)


def test_fast_reciprocal_and_square_root_are_correctly_rounded(gpu):
    """vkr_selftest_math sweeps all 2^32 binary32 patterns on the device: the fast reciprocal (one Newton
    step), the fast square root and the pair the shading kernel uses return IEEE's bits on their domains."""
    import ctypes
    counts = (ctypes.c_uint64 * 7)()
    _lib.check(_lib.load().vkr_selftest_math(counts))
    rcp1, _, sqrt_, rcp_of_sqrt, pair, n_rcp, n_sqrt = list(counts)
    assert n_rcp > 4.2e9 and n_sqrt > 2.1e9
    assert (rcp1, sqrt_, rcp_of_sqrt) == (0, 0, 0)
    assert pair == 0


def test_shade_lights_on_cube_edges_corners_and_surfaces(gpu, oracle):
    """Lights placed so that many pixels see them along cube edges and corners (|x| = |y| = |z| directions), a
    hand's breadth above the surface (every tap on another face), and exactly on a pixel's position
    (distance zero: the slow branches of the reciprocal and the root)."""
    torch, P = _torch(), _passes()
    w, h, nl, cube = 128, 72, 8, 32
    f = synth.make_frame(97, w, h, nl, cube, 64, irr_res=8, pref_res=16, lut_res=32)
    pos = oracle.f16_to_f32(f.position)[..., :3]
    lights = f.lights.copy()
    anchors = [(10, 20), (40, 64), (60, 100), (30, 30), (50, 90), (20, 110), (65, 5), (5, 70)]
    offsets = [(0.7, 0.7, 0.7), (-0.5, 0.5, 0.5), (0.6, 0.6, 0.0), (0.0, -0.4, 0.4), (0.02, 0.03, 0.01), (0.0, 0.0, 0.0),
               (1.5, 1.5, -1.5), (0.3, 0.0, 0.3)]
    for i, ((y, x), o) in enumerate(zip(anchors, offsets)):
        lights["position"][i] = pos[y, x] + np.asarray(o, dtype=np.float32)
    f.lights = lights
    want_c, want_t = oracle.shade(f)
    for tiled in (True, False):
        d = P.DeviceFrame.upload(f, gpu, tiled=tiled)
        color = torch.empty((h, w, 4), dtype=torch.float16, device=gpu)
        thr = torch.empty_like(color)
        P.shade(d, color, thr)
        assert_bits_equal(P.f16_bits(color), want_c, f"colour (tiled cubes: {tiled})")
        assert_bits_equal(P.f16_bits(thr), want_t, f"threshold (tiled cubes: {tiled})")


def test_graft_entry_smoke(gpu):
    """The round-end smoke check itself (one frame through layer 2 + a small IBL precompute against the oracle)."""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    importlib.import_module("__graft_entry__").smoke()


def test_shade_with_non_finite_and_huge_inputs(gpu, oracle):
    """Infinite and NaN positions in the G-buffer and lights 1e30 .. 3e38 away: the kernel stays inside its
    images (the unclamped one-face path is only taken for tame coordinates) and every pixel whose own inputs
    are finite still equals the oracle bit for bit."""
    torch, P = _torch(), _passes()
    w, h, nl, cube = 96, 64, 4, 32
    f = synth.make_frame(98, w, h, nl, cube, 64, irr_res=8, pref_res=16, lut_res=32)
    pos = f.position.copy()
    pos[10:14, 20:30, 0] = 0x7C00          # +inf
    pos[30:33, 40:48, 1] = 0xFC00          # -inf
    pos[50:52, 60:70, 2] = 0x7E00          # NaN
    pos[5, 5, :3] = 0x7BFF                 # largest finite binary16
    f.position = pos
    lights = f.lights.copy()
    lights["position"][0] = (1e30, -2e30, 3e30)
    lights["position"][1] = (3e38, 3e38, -3e38)
    f.lights = lights
    want_c, want_t = oracle.shade(f)
    finite = np.isfinite(oracle.f16_to_f32(pos)[..., :3]).all(-1)
    for tiled in (True, False):
        d = P.DeviceFrame.upload(f, gpu, tiled=tiled)
        color = torch.empty((h, w, 4), dtype=torch.float16, device=gpu)
        thr = torch.empty_like(color)
        P.shade(d, color, thr)
        torch.cuda.synchronize()
        got_c, got_t = P.f16_bits(color), P.f16_bits(thr)
        assert_bits_equal(got_c[finite], want_c[finite], f"colour of the finite pixels (tiled cubes: {tiled})")
        assert_bits_equal(got_t[finite], want_t[finite], f"threshold of the finite pixels (tiled cubes: {tiled})")


def test_renderer_without_a_projection_draws_no_skybox(gpu, oracle):
    """A SceneInfo whose projection matrix is all zero (never set): EndFrame leaves the discarded texels at the
    clear colour instead of failing on the skybox draw."""
    from vulkanrenderer_amd.renderer import HipRenderer
    w, h, nl = 64, 40, 1
    f = synth.make_frame(99, w, h, nl, 16, 64, irr_res=8, pref_res=16, lut_res=32)
    f.scene = f.scene.copy()
    f.scene["camera_projection"] = np.zeros(16, dtype=np.float32)
    with HipRenderer(w, h, max_lights=2, dir_shadow_res=64, point_shadow_res=16, env_res=16,
                     irradiance_res=8, prefilter_res=16, prefilter_mips=1, brdf_lut_res=32) as r:
        r.UploadEnvironmentCubemap(synth.make_env_cube(np.random.default_rng(2), 16))
        r.CreateIrradianceCubemap(); r.CreatePrefilteredCubemap(); r.CreateBRDFLUTTexture()
        f.irradiance = r.Read(_lib.TARGET_IRRADIANCE)
        f.prefilter = r.Read(_lib.TARGET_PREFILTER, mip=0)
        f.brdf_lut = r.Read(_lib.TARGET_BRDF_LUT)
        r.UpdateSceneInfo(f.scene)
        r.UploadGBuffer(f.albedo, f.normal, f.arm, f.position)
        r.UploadDirectionalShadowMap(f.dir_shadow)
        r.UploadPointShadowMap(0, f.cube_shadow[0])
        r.SetLightInfos(f.lights)
        r.BeginFrame()
        r._call(r._lib.vkr_end_frame)
        want_c, _ = oracle.shade(f)
        assert_bits_equal(r.Read(_lib.TARGET_COLOR, 0), want_c, "colour")


def test_gbuffer_rejects_fragments_on_the_far_plane(gpu, oracle):
    """CompareOp.Less against the depth buffer's clear value: a triangle at depth exactly 1.0 leaves the clear values,
    one a binary32 step nearer is drawn -- kernel and oracle alike."""
    from tests.test_oracle_kat import _flat_triangle_draw, _identity_scene
    P = _passes()
    scene = _identity_scene()
    flip = bool((oracle.gbuffer([_flat_triangle_draw(0.5, True)], scene, 32, 32)[0][..., 3] != 0).any())
    for z in (1.0, np.float32(1.0) - np.float32(2.0 ** -24), 0.25):
        draws = [_flat_triangle_draw(z, flip)]
        want = oracle.gbuffer(draws, scene, 32, 32)
        got = P.gbuffer(_gpass_device_draws(draws, gpu), scene, 32, 32, gpu)
        for name, g, wv in zip(("albedo", "normal", "arm", "position"), got, want):
            assert_bits_equal(P.f16_bits(g), wv, f"z = {z}: G-buffer {name}")
        assert bool((want[0][..., 3] != 0).any()) == (z < 1.0)


def _small_renderer(w, h, **kw):
    from vulkanrenderer_amd.renderer import HipRenderer
    args = dict(max_lights=8, dir_shadow_res=64, point_shadow_res=16, env_res=16, irradiance_res=8, prefilter_res=16, prefilter_mips=1, brdf_lut_res=32)
    args.update(kw)
    return HipRenderer(w, h, **args)


def test_renderer_end_frame_takes_the_lights_list_and_the_volume_mesh(gpu, oracle):
    """The public EndFrame(): the `Lights` list is what is drawn, with per-light volume coverage once a volume mesh is
    set; light records given with SetLightInfos for a frame are not overwritten by the list."""
    import dataclasses
    from vulkanrenderer_amd.renderer import Light
    w, h, nl = 136, 88, 5
    verts, tris = synth.reference_light_volume()
    f = synth.make_frame(72, w, h, nl, 16, 64, irr_res=8, pref_res=16, lut_res=32)
    f.scene = _camera_scene((2.0, 1.0, -4.0), aspect=w / h)
    f.lights = _small_lights(np.random.default_rng(3), nl, 0.03)
    masks = oracle.light_coverage(f.scene, f.lights, verts, tris, w, h)
    with _small_renderer(w, h) as r:
        env = synth.make_env_cube(np.random.default_rng(2), 16)
        r.UploadEnvironmentCubemap(env)
        r.CreateIrradianceCubemap(); r.CreatePrefilteredCubemap(); r.CreateBRDFLUTTexture()
        f.irradiance, f.prefilter, f.brdf_lut = r.Read(_lib.TARGET_IRRADIANCE), r.Read(_lib.TARGET_PREFILTER, mip=0), r.Read(_lib.TARGET_BRDF_LUT)
        sky = oracle.skybox(env, f.scene, w, h)
        r.Lights = [Light(tuple(float(v) for v in l["position"]), tuple(float(v) for v in l["color"])) for l in f.lights]
        for i, l in enumerate(r.Lights):
            assert l.ToInfo().tobytes() == f.lights[i].tobytes()

        def frame_inputs():
            r.UpdateSceneInfo(f.scene)
            r.UploadGBuffer(f.albedo, f.normal, f.arm, f.position)
            r.UploadDirectionalShadowMap(f.dir_shadow)
            for i in range(nl):
                r.UploadPointShadowMap(i, f.cube_shadow[i])

        # slot 0, no volume mesh: every light covers every pixel
        frame_inputs(); r.BeginFrame(); r.EndFrame()
        want_c, want_t = oracle.shade(f, sky)
        assert_bits_equal(r.Read(_lib.TARGET_COLOR, 0), want_c, "Lights list, full coverage: colour")
        assert_bits_equal(r.Read(_lib.TARGET_THRESHOLD, 0), want_t, "Lights list, full coverage: threshold")
        # slot 1, volume mesh set: each light's draw covers its volume
        r.SetLightVolumeMesh(verts, tris)
        frame_inputs(); r.BeginFrame(); r.EndFrame()
        mask_c, mask_t = oracle.shade(f, sky, coverage=masks)
        assert not np.array_equal(mask_c, want_c)
        assert_bits_equal(r.Read(_lib.TARGET_COLOR, 1), mask_c, "Lights list + volume mesh: colour")
        assert_bits_equal(r.Read(_lib.TARGET_THRESHOLD, 1), mask_t, "Lights list + volume mesh: threshold")
        # slot 0 again (its uploads are still there): two explicit records with full coverage survive EndFrame
        r.UpdateSceneInfo(f.scene)
        r.SetLightInfos(f.lights[:2], _lib.COVERAGE_FULLSCREEN)
        r.BeginFrame(); r.EndFrame()
        two = dataclasses.replace(f, lights=f.lights[:2], cube_shadow=np.ascontiguousarray(f.cube_shadow[:2]))
        two_c, _ = oracle.shade(two, sky)
        assert_bits_equal(r.Read(_lib.TARGET_COLOR, 0), two_c, "explicit light records: colour")
        # and the next frame falls back to the list
        frame_inputs(); r.BeginFrame(); r.CompositionPass()
        r.WaitIdle()
        assert_bits_equal(r.Read(_lib.TARGET_COLOR, 1), mask_c, "CompositionPass with the Lights list: colour")


def test_renderer_without_uploads_shades_an_empty_scene(gpu, oracle):
    """Nothing uploaded, nothing drawn: every frame slot holds the reference's clear values (G-buffer alpha 0, depth
    1.0), so both slots give the cleared composition image, not whatever the allocations held."""
    from vulkanrenderer_amd.renderer import Light
    w, h = 72, 40
    with _small_renderer(w, h) as r:
        r.UploadEnvironmentCubemap(synth.make_env_cube(np.random.default_rng(2), 16))
        r.CreateIrradianceCubemap(); r.CreatePrefilteredCubemap(); r.CreateBRDFLUTTexture()
        r.Lights = [Light((0.0, 1.0, 0.0), (5.0, 5.0, 5.0))]
        for slot in (0, 1, 0):
            assert r.CurrentFrame == slot
            r.BeginFrame(); r.EndFrame()   # the zeroed SceneInfo has no projection: no skybox either
            color = r.Read(_lib.TARGET_COLOR, slot)
            assert (color[..., :3] == 0).all() and (color[..., 3] == 0x3C00).all()
            assert (r.Read(_lib.TARGET_THRESHOLD, slot)[..., :3] == 0).all()
            ldr = r.Read(_lib.TARGET_LDR, slot)
            assert (ldr[..., :3] == 0).all() and (ldr[..., 3] == 255).all()
        r.Resize(40, 24)
        r.BeginFrame(); r.EndFrame()
        color = r.Read(_lib.TARGET_COLOR, 1)
        assert color.shape == (24, 40, 4) and (color[..., :3] == 0).all() and (color[..., 3] == 0x3C00).all()


def test_renderer_refuses_a_frame_whose_draws_are_only_partly_shadeable(gpu):
    w, h = 64, 40
    slab = synth.make_scene_mesh(8)
    uv, nrm, tan = synth.mesh_attributes(*slab, uv_scale=0.25)
    alb, nmap, arm = synth.make_material(np.random.default_rng(4), 16)
    with _small_renderer(w, h) as r:
        r.UploadEnvironmentCubemap(synth.make_env_cube(np.random.default_rng(2), 16))
        r.CreateIrradianceCubemap(); r.CreatePrefilteredCubemap(); r.CreateBRDFLUTTexture()
        full = r.CreateMesh(slab[0], slab[1], uv, nrm, tan)
        bare = r.CreateMesh(slab[0], slab[1])
        mat = r.CreateMaterial(alb, nmap, arm)
        r.UpdateSceneInfo(_camera_scene((2.0, 1.6, -4.0), aspect=w / h))
        r.BeginFrame()
        r.DrawModel(full, np.eye(4), mat)
        r.DrawModel(bare, np.eye(4))
        with pytest.raises(_lib.VkrError) as e:
            r.EndFrame()
        assert e.value.code == _lib.VKR_ERR_INVALID and "draw calls have a material" in str(e.value)
