"""Bounding-box statistics of the point-shadow pass on the benchmark scene: how many (light, face, triangle)
boxes there are, how many exceed the wave-cooperative threshold, and how many samples each class holds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from tests import oracle
from vulkanrenderer_amd import synth

res = 1024
verts, tris = synth.make_scene_mesh(128)
lights = synth.make_lights(np.random.default_rng(bench.SEED), 64)
v4 = np.concatenate([verts, np.ones((len(verts), 1), np.float32)], 1).astype(np.float64)
small_n = big_n = 0
small_s = big_s = 0
big_hist = []
wave_max = []
for li in range(0, 64, 8):
    for face in range(6):
        m = oracle.shadow_face_matrix(lights["position"][li], face, 25.0).astype(np.float64)
        clip = v4 @ m
        c = clip[tris]                                   # [T, 3, 4]
        inside = (c[..., 3] > 1e-6).all(1) & (c[..., 2] >= 0).all(1)
        ndc = c[..., :2] / np.maximum(c[..., 3:4], 1e-6)
        lo = np.floor((ndc.min(1) + 1) * res / 2 - 0.5) - 1
        hi = np.ceil((ndc.max(1) + 1) * res / 2 - 0.5) + 1
        lo = np.clip(lo, 0, res - 1); hi = np.clip(hi, 0, res - 1)
        vis = inside & (hi >= lo).all(1) & (ndc.max(1) > -1).all(1) & (ndc.min(1) < 1).all(1)
        d = (c[:, 1, 0] * c[:, 2, 1] - c[:, 1, 1] * c[:, 2, 0])  # not the exact facing rule: statistics only
        area = np.where(vis, (hi[:, 0] - lo[:, 0] + 1) * (hi[:, 1] - lo[:, 1] + 1), 0)
        big = area > 128
        small_n += int((vis & ~big).sum()); big_n += int(big.sum())
        small_s += int(area[vis & ~big].sum()); big_s += int(area[big].sum())
        big_hist.append(area[big])
        a = np.where(big, 0, area)
        pad = (-len(a)) % 64
        wave_max.append(np.pad(a, (0, pad)).reshape(-1, 64).max(1))
big_hist = np.concatenate(big_hist); wave_max = np.concatenate(wave_max)
print("sampled 8 lights x 6 faces;", len(tris), "triangles")
print("small boxes", small_n, "samples", small_s, "mean", small_s / max(small_n, 1))
print("big boxes  ", big_n, "samples", big_s, "mean", big_s / max(big_n, 1), "median", np.median(big_hist), "p90", np.percentile(big_hist, 90))
print("small path: sum over waves of the largest box", int(wave_max.sum()), "x64 =", int(wave_max.sum()) * 64, "lane-samples for", small_s, "useful")
print("big path: iterations", int(np.ceil(big_hist / 64).sum()), "for", big_n, "triangles")
