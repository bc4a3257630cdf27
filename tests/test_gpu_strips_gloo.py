"""Two ranks, the HIP kernels and a real halo exchange: both ranks share cuda:0 and talk over gloo
(RCCL refuses two ranks on one device, and the test box has one GPU).  The assembled image must be
the image of a single-rank run, byte for byte."""
import os
import socket

import numpy as np
import pytest

from vulkanrenderer_amd import synth

pytestmark = pytest.mark.gpu

W, H, LIGHTS = 136, 75, 3


def _frame():
    return synth.make_frame(91, W, H, LIGHTS, 32, 128, irr_res=8, pref_res=16, lut_res=32)


def _worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    from vulkanrenderer_amd import passes as P
    from vulkanrenderer_amd.strips import StripPlan, StripRenderer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        f = _frame()
        plan = StripPlan.make(rank, world, H)
        r = StripRenderer(plan, W, device=dev)
        r.render(P.DeviceFrame.upload(f.rows(plan.row0, plan.row0 + plan.rows), dev))
        parts = r.gather(0)
        if rank == 0:
            np.save(out_path, torch.cat([p.cpu() for p in parts], 0).numpy())
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_hip_strips_over_gloo_equal_single_rank(gpu, oracle, world, tmp_path):
    import torch
    import torch.multiprocessing as mp
    from vulkanrenderer_amd import passes as P
    from vulkanrenderer_amd.strips import StripPlan, StripRenderer
    f = _frame()
    whole = StripRenderer(StripPlan.make(0, 1, H), W, device=gpu)
    want = whole.render(P.DeviceFrame.upload(f, gpu)).cpu().numpy()
    out_path = str(tmp_path / "ldr.npy")
    mp.spawn(_worker, args=(world, _free_port(), out_path), nprocs=world, join=True)
    got = np.load(out_path)
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    ref = oracle.frame_chain(f)["ldr"]
    assert np.abs(got.astype(np.int32) - ref.astype(np.int32)).max() <= 1
