"""bench.py started from a bare shell with --gpus N > 1 starts its own ranks.  On a machine without a GPU
the transport rehearsal (--transport-only: gloo, CPU tensors, no kernels) drives the launcher, the halo
exchange in both directions and the gather through the same StripRenderer the GPU run uses; with a GPU
(marked gpu) the real kernels run on ranks that share the device over gloo (VKR_DIST_BACKEND=gloo)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, env=None, timeout=240):
    clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    clean.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=clean, capture_output=True, text=True, timeout=timeout)


def _json_line(stdout: str) -> dict:
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line expected, got {len(lines)}:\n{stdout}"
    return json.loads(lines[0])


@pytest.mark.parametrize("world,height", [(2, 97), (3, 41), (8, 97)])
def test_self_launched_ranks_exchange_halos_and_gather(world, height):
    p = _run("--gpus", str(world), "--transport-only", "--steps", "2", "--warmup", "1", "--height", str(height))
    assert p.returncode == 0, p.stderr[-2000:]
    line = _json_line(p.stdout)
    assert line["n_gpus"] == world and line["transport_only"] is True
    assert line["distributed"] == {"backend": "gloo", "world": world}
    assert line["gathered_equals_single_rank"] is True
    assert line["value"] is None  # nothing was rendered: no throughput claim


def test_a_failing_rank_fails_the_launch():
    # strips of 1 row are refused by StripPlan (a strip holds at least the 4 halo rows): every rank exits non-zero
    p = _run("--gpus", "2", "--transport-only", "--height", "3", "--launch-timeout", "60")
    assert p.returncode != 0
    assert "every strip needs at least" in p.stderr


def test_rank_environment_must_match_gpus():
    p = _run("--gpus", "2", "--transport-only", env={"WORLD_SIZE": "3", "RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=3" in p.stderr


@pytest.mark.gpu
def test_self_launched_ranks_render_one_frame_in_strips(gpu):
    """Two ranks sharing cuda:0 over gloo, HIP kernels, the strong-scaling split of one frame: rank 0's line says
    the gathered image equals the image it renders alone."""
    p = _run("--gpus", "2", "--steps", "2", "--warmup", "1", "--width", "256", "--height", "96", "--lights", "3", "--cube-res", "64",
             "--dir-res", "128", "--no-extra", env={"VKR_DIST_BACKEND": "gloo"})
    assert p.returncode == 0, p.stderr[-3000:]
    line = _json_line(p.stdout)
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["distributed"]["backend"] == "gloo" and line["distributed"]["world"] == 2
    assert line["gathered_equals_single_gpu"] is True
    assert line["value"] > 0 and set(line["phases_ms"]) >= {"halo_wait", "gather", "shade", "interior_bloom_post"}
    assert "48 rows" in line["config"]["workload"]
    # the IBL precompute with its rows split over the two ranks (2048^2 configuration), and rank 0's own timing of it
    assert line["ibl_precompute_sharded"]["world"] == 2 and line["ibl_precompute_sharded"]["ms"] > 0
    assert 0 < line["ibl_precompute_sharded"]["ms"] < line["ibl_precompute_sharded"]["ms_with_the_shader_summation_order"]
    orders = line["ibl_precompute"]["irradiance_orders"]
    assert orders["by_region_rows (default, timed in total_ms)"]["texels_with_samples_unaccounted_for"] == 0
    assert orders["by_region"]["texels_with_samples_unaccounted_for"] == 0
    assert orders["by_region_rows (default, timed in total_ms)"]["against_the_shader_order"]["max_difference_binary16_ulps"] <= 1.0
    assert line["ibl_precompute"]["total_ms"] > 0


@pytest.mark.gpu
@pytest.mark.timeout(420)
def test_eight_ranks_render_the_headline_frame_in_strips(gpu):
    """The headline frame itself -- 3840 x 2160, 64 lights with 1024^2 cubes, a 4096^2 directional map -- cut into the
    eight 270-row strips of an 8-GPU run, the eight ranks sharing cuda:0 over gloo: the launcher, the halo exchange
    between seven neighbour pairs and the gather of eight strips at full size, and the gathered image equals the image
    rank 0 renders alone, byte for byte.  (RCCL itself needs eight GPUs; the transport is the only thing that differs.)"""
    p = _run("--gpus", "8", "--steps", "2", "--warmup", "1", "--no-extra", "--no-ibl", "--no-predict", "--launch-timeout", "360",
             env={"VKR_DIST_BACKEND": "gloo"}, timeout=400)
    assert p.returncode == 0, p.stderr[-3000:]
    line = _json_line(p.stdout)
    assert line["n_gpus"] == 8 and line["scaling"] == "strong"
    assert line["distributed"]["backend"] == "gloo" and line["distributed"]["world"] == 8
    assert "8 strips of 270 rows" in line["config"]["workload"] and "3840x2160" in line["config"]["workload"]
    assert line["gathered_equals_single_gpu"] is True
    assert line["value"] > 0
