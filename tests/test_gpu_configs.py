"""GPU parity at the real shapes of the benchmark configurations (not scaled-down stand-ins):

  C1  1280x720, 1 directional + 4 point lights, 256^2 shadow cubes, 1024^2 directional map
  C2  1920x1080 full frame chain (the reference's window size and default maps, 2 point lights)
  C3  3840x2160, 64 point lights with 1024^2 cubes + 4096^2 directional map
  C4  IBL precompute: 4096x2048 HDR equirect -> 2048^2 cube -> irradiance 64^2, prefilter 128^2 x 5 mips, LUT 512^2
  C5  16384-pixel-wide rows with 256 shadowed lights (a 64-row band of the 16384^2 frame on one GPU), and a
      16384-wide frame cut over three ranks with halo exchange

Everything goes through the C ABI (vulkanrenderer_amd.passes -> libvkr_hip.so) and is compared with the CPU
oracle on the same seeded inputs: bit for bit for the binary16 images, within 1 LSB for the 8-bit image,
within max(1e-3, 1 binary16 ulp) for the equirect resampling (atan / asin come from different libraries).
The oracle takes seconds at these sizes on the GPU box's 16 host cores (the 4K frame about 12 s)."""
import os
import socket

import numpy as np
import pytest

from vulkanrenderer_amd import _lib, synth

from tests.test_gpu_parity import assert_bits_equal, f16_ulp

pytestmark = pytest.mark.gpu


def _chain_on_gpu(f, gpu):
    """Whole frame chain through layer 1; returns dict of host arrays like oracle.frame_chain."""
    import torch
    from vulkanrenderer_amd import passes as P
    d = P.DeviceFrame.upload(f, gpu)
    color, thr, b1, b2 = (torch.empty((f.height, f.width, 4), dtype=torch.float16, device=gpu) for _ in range(4))
    ldr, ldr_fused = (torch.empty((f.height, f.width, 4), dtype=torch.uint8, device=gpu) for _ in range(2))
    P.shade(d, color, thr)
    P.bloom_blur(thr, b1, True, _lib.WRAP_REPEAT)
    P.bloom_blur(b1, b2, False, _lib.WRAP_CLAMP)
    P.postprocess(color, b2, ldr)
    P.bloomv_postprocess(b1, color, ldr_fused)
    torch.cuda.synchronize()
    return {"color": P.f16_bits(color), "threshold": P.f16_bits(thr), "bloom1": P.f16_bits(b1), "bloom2": P.f16_bits(b2),
            "ldr": ldr.cpu().numpy(), "ldr_fused": ldr_fused.cpu().numpy()}


def _check_chain(got, want, what):
    for name in ("color", "threshold", "bloom1", "bloom2"):
        assert_bits_equal(got[name], want[name], f"{what}: {name}")
    assert np.array_equal(got["ldr"], got["ldr_fused"]), f"{what}: fused and unfused post-process differ"
    diff = np.abs(got["ldr"].astype(np.int32) - want["ldr"].astype(np.int32))
    assert diff.max() <= 1, f"{what}: 8-bit image differs by {diff.max()} LSB"
    lit = want["threshold"].view(np.float16)[..., :3].astype(np.float32).sum(-1) > 0
    return float(lit.mean())


def test_c1_1280x720_one_directional_four_point_lights(gpu, oracle):
    f = synth.make_frame(0x5EED0001, 1280, 720, 4, 256, 1024)
    got = _chain_on_gpu(f, gpu)
    bloom_fraction = _check_chain(got, oracle.frame_chain(f), "C1")
    assert 0.005 < bloom_fraction < 0.9, "the bloom path should be exercised"


def test_c2_1920x1080_full_frame_chain(gpu, oracle):
    f = synth.make_frame(0x5EED0002, 1920, 1080, 2, 256, 1024)
    got = _chain_on_gpu(f, gpu)
    _check_chain(got, oracle.frame_chain(f), "C2")


def test_c3_3840x2160_64_lights_whole_frame(gpu, oracle):
    """The headline configuration, whole frame against the oracle (531 M pixel-light pairs on the CPU)."""
    import torch
    from vulkanrenderer_amd import passes as P
    f = synth.make_frame(0x5EED0003, 3840, 2160, 64, 1024, 4096)
    got = _chain_on_gpu(f, gpu)
    want = oracle.frame_chain(f)
    _check_chain(got, want, "C3")
    # the same frame shaded as three unequal strips (independent launches) gives the same bits
    d = P.DeviceFrame.upload(f, gpu)
    import dataclasses
    for y0, y1 in ((0, 8), (8, 1336), (1336, 2160)):
        part = dataclasses.replace(d, rows=y1 - y0, albedo=d.albedo[y0:y1], normal=d.normal[y0:y1], arm=d.arm[y0:y1], position=d.position[y0:y1])
        c, t = (torch.empty((y1 - y0, 3840, 4), dtype=torch.float16, device=gpu) for _ in range(2))
        P.shade(part, c, t)
        assert_bits_equal(P.f16_bits(c), want["color"][y0:y1], f"C3 strip rows {y0}..{y1}: colour")
        assert_bits_equal(P.f16_bits(t), want["threshold"][y0:y1], f"C3 strip rows {y0}..{y1}: threshold")


def test_c4_ibl_precompute_at_2048_64_128_512(gpu, oracle):
    import torch
    from vulkanrenderer_amd import passes as P
    src = synth.make_equirect(np.random.default_rng(0x5EED0004), 4096, 2048, hdr=True)
    dsrc = P.to_device_f16(src, gpu)
    env = torch.empty((6, 2048, 2048, 4), dtype=torch.float16, device=gpu)
    P.equirect_to_cube(dsrc, _lib.SRC_RGBA16F, env)
    got_env = P.f16_bits(env)
    want_env = oracle.equirect_to_cube(src, 2048)
    a, b = got_env.view(np.float16).astype(np.float32), want_env.view(np.float16).astype(np.float32)
    tol = np.maximum(1e-3, f16_ulp(b))
    assert (np.abs(a - b) <= tol).all(), f"C4 equirect: max error {np.abs(a - b).max()}"
    # The convolutions read one and the same cube on both sides (the GPU's), so that they can be held to bit equality.
    # An HDR environment with suns, not the smooth equirect image, so that the sums see a large dynamic range.
    henv = synth.make_env_cube(np.random.default_rng(0x5EED0004), 2048)
    env = P.to_device_f16(henv, gpu)
    tiled = P.tile_env_cube(env)
    irr = torch.empty((6, 64, 64, 4), dtype=torch.float16, device=gpu)
    P.irradiance(tiled, irr, _lib.IRRADIANCE_SEQUENTIAL, env_layout=_lib.ENV_LAYOUT_TILED_4X4)
    assert_bits_equal(P.f16_bits(irr), oracle.irradiance(henv, 64), "C4 irradiance 64^2 (all 24 576 texels, 15 876 samples each)")
    for m in range(5):
        n = 128 >> m
        pre = torch.empty((6, n, n, 4), dtype=torch.float16, device=gpu)
        P.prefilter(tiled, pre, m / 4.0, env_layout=_lib.ENV_LAYOUT_TILED_4X4)
        assert_bits_equal(P.f16_bits(pre), oracle.prefilter(henv, n, m / 4.0), f"C4 prefilter mip {m} ({n}^2, roughness {m / 4})")
    lut = torch.empty((512, 512, 2), dtype=torch.float16, device=gpu)
    P.brdf_lut(lut)
    assert_bits_equal(P.f16_bits(lut), oracle.brdf_lut(512), "C4 BRDF LUT 512^2")


def _c5_band(rows, n_lights, cube_res, reps):
    """A band of `rows` 16384-pixel rows with n_lights * reps lights; the depth cubes of the first n_lights
    lights are repeated for the others (positions and colours are all different)."""
    rng = np.random.default_rng(0x5EED0005)
    g = synth.make_gbuffer(rng, 16384, rows)
    lights = synth.make_lights(rng, n_lights * reps)
    cubes = synth.make_cube_shadows(rng, n_lights, cube_res)
    cubes = np.ascontiguousarray(np.tile(cubes, (reps, 1, 1, 1)))
    return synth.FrameInputs(16384, rows, *g, synth.default_scene_info(), lights, synth.make_dir_shadow(rng, 4096), cubes,
                             synth.make_smooth_cube(rng, 32, 0.2, 1.5), synth.make_smooth_cube(rng, 128, 0.1, 3.0), synth.make_smooth_lut(rng, 512))


def test_c5_16384_wide_band_with_256_lights(gpu, oracle):
    f = _c5_band(64, 64, 1024, 4)
    assert len(f.lights) == 256 and f.cube_shadow.shape == (256, 6, 1024, 1024)
    got = _chain_on_gpu(f, gpu)
    _check_chain(got, oracle.frame_chain(f), "C5 band")


def test_c5_full_width_256_rows_of_the_per_rank_shape(gpu, oracle):
    """The shape one rank of eight shades in the 16384^2 configuration, at its full width and with all 256 lights and
    1024^2 cubes (6 GiB, the 64 synthetic cubes repeated), 256 of its 2048 rows -- more than the oracle can check in a
    test's time, so the whole image is held to properties that do not depend on the size: shaded as one strip and as two
    strips with halo rows it is the same image byte for byte; the one-kernel post chain equals the three passes; the
    lights are drawn in four launches of 64 (the kernel's own cut); and a band of 16 rows across the strip boundary equals the oracle's bit for bit."""
    import dataclasses
    import torch
    from vulkanrenderer_amd import passes as P
    from vulkanrenderer_amd.strips import StripPlan, StripRenderer
    rows = 256
    f = _c5_band(rows, 64, 1024, 4)
    d = P.DeviceFrame.upload(f, gpu)
    whole = StripRenderer(StripPlan.make(0, 1, rows), 16384, device=gpu)
    ldr = whole.render(d).clone()
    color, thr = P.f16_bits(whole.color), P.f16_bits(whole.threshold)
    # (a) two strips of 128 rows, the halo rows handed over by hand (what the exchange moves), against the one strip
    parts = []
    strips = [StripRenderer(StripPlan.make(k, 2, rows), 16384, device=gpu) for k in range(2)]
    frames = [dataclasses.replace(d, rows=128, albedo=d.albedo[k * 128:(k + 1) * 128], normal=d.normal[k * 128:(k + 1) * 128],
                                  arm=d.arm[k * 128:(k + 1) * 128], position=d.position[k * 128:(k + 1) * 128]) for k in range(2)]
    for r, fr in zip(strips, frames):
        r.backend.shade(fr, r.color, r.threshold)
    from vulkanrenderer_amd.strips import HALO
    strips[0].threshold_padded[128 + HALO:] = strips[1].threshold[:HALO]
    strips[1].threshold_padded[:HALO] = strips[0].threshold[128 - HALO:]
    for k, r in enumerate(strips):
        lo, hi = (HALO, 128 + 2 * HALO - 1) if k == 0 else (0, HALO + 128 - 1)
        r.backend.bloom_post(r.threshold_padded, r.color, r.ldr, lo, hi)
        parts.append(r.ldr)
    assert torch.equal(torch.cat(parts, 0), ldr), "two strips with halo rows differ from one strip"
    # (b) the one-kernel post chain against bloom pass 1, then pass 2 fused with the post-process draw
    b1 = torch.empty_like(whole.threshold)
    ldr3 = torch.empty_like(ldr)
    P.bloom_blur(whole.threshold, b1, True, _lib.WRAP_REPEAT)
    P.bloomv_postprocess(b1, whole.color, ldr3)
    assert torch.equal(ldr3, ldr), "the one-kernel post chain differs from its three passes"
    # (c) rows 120..136 against the oracle (the composition pass is per pixel: a band of the frame is a frame)
    band = f.rows(120, 136)
    want_c, want_t = oracle.shade(band)
    assert_bits_equal(color[120:136], want_c, "C5 rows 120..136: colour")
    assert_bits_equal(thr[120:136], want_t, "C5 rows 120..136: threshold")


W5, H5, L5 = 16384, 72, 6


def _c5_strip_frame():
    return synth.make_frame(0x5EED0055, W5, H5, L5, 64, 256, irr_res=8, pref_res=16, lut_res=32)


def _c5_worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    from vulkanrenderer_amd import passes as P
    from vulkanrenderer_amd.strips import StripPlan, StripRenderer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        f = _c5_strip_frame()
        plan = StripPlan.make(rank, world, H5)
        r = StripRenderer(plan, W5, device=dev, frames_in_flight=2)
        for _ in range(2):  # two frames through the ring of output strips
            r.render(P.DeviceFrame.upload(f.rows(plan.row0, plan.row0 + plan.rows), dev))
            parts = r.gather(0)
        torch.cuda.synchronize()
        if rank == 0:
            np.save(out_path, torch.cat([p.cpu() for p in parts], 0).numpy())
    finally:
        dist.destroy_process_group()


def test_c5_16384_wide_frame_over_three_ranks(gpu, oracle, tmp_path):
    """Three ranks (sharing the one GPU of the test box, talking over gloo) each own 24 rows of a 16384-wide frame:
    512 KiB halo messages per direction, as in the 16384^2 configuration."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out_path = str(tmp_path / "ldr.npy")
    mp.spawn(_c5_worker, args=(3, port, out_path), nprocs=3, join=True)
    got = np.load(out_path)
    f = _c5_strip_frame()
    one = _chain_on_gpu(f, gpu)
    assert np.array_equal(got, one["ldr_fused"]), "three strips differ from the single-GPU image"
    want = oracle.frame_chain(f)["ldr"]
    assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1
