"""The RCCL path on the GPU that is there: torch.distributed backend "nccl" (= RCCL) at world size 1 drives the REAL
PCDSensorUpdaterHip through rmcl_amd.distributed.ShardedSensorUpdate / ShardedResample and must reproduce the
unsharded update bit for bit (the sharded classes had never executed on a GPU).
World sizes 2 and 3 (ragged) run on CPU with gloo in tests/test_distributed_cpu.py; 8 GPUs are the driver's to launch."""
import math
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rccl_world1_sharded_update_equals_unsharded():
    """runs in a fresh process: torch (which bundles its own HIP runtime) has to initialise the device BEFORE
    librmclhip.so does -- the order bench.py uses -- and the pytest process has already created HIP contexts."""
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests")]))
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def _main():
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    torch.cuda.init()
    import rmcl_amd as ra
    from rmcl_amd import distributed as D, synthetic as syn, types as T
    ctx = ra.Context(0)
    meshes = {"room30k": syn.noisy_room(30000)}.__getitem__
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        v, f = meshes("room30k")
        hm = ra.import_hip_map(ctx, v, f)
        n = 3001
        poses, attrs = syn.uniform_particles(n, seed=21, bb_min=(-8, -8, 0.2, 0, 0, -math.pi), bb_max=(8, 8, 3, 0, 0, math.pi))
        beams = ra.beams_from_points(syn.model_directions(syn.model_pf16())[::4] * np.float32(5.0))
        upd = ra.PCDSensorUpdaterHip(hm)
        upd.init()
        upd.setInput(beams, syn.tsb_offset())
        # unsharded
        d_p, d_a = ra.DeviceArray.from_host(ctx, poses), ra.DeviceArray.from_host(ctx, attrs)
        upd.update(d_p, d_a)
        ref = d_a.download()
        # sharded (one shard = everything) + the RCCL all-gather of the weights
        d_a2 = ra.DeviceArray.from_host(ctx, attrs)
        sh = D.ShardedSensorUpdate(upd, n, 0, 1)
        w_local = torch.empty(n, dtype=torch.float32, device="cuda")
        for _ in range(2):
            d_a2.upload(attrs)
            gathered = sh.update(d_p, d_a2, w_local)
        torch.cuda.synchronize()
        assert gathered.is_cuda and gathered.numel() == n
        assert np.array_equal(gathered.cpu().numpy(), ref["likelihood"]["mean"])
        assert d_a2.download().tobytes() == ref.tobytes()
        ssum, smax = D.allreduce_sum_max(w_local)
        assert abs(ssum - float(ref["likelihood"]["mean"].astype(np.float64).sum())) < 1e-6 * max(1.0, abs(ssum))
        assert smax == float(ref["likelihood"]["mean"].max())
        # record all-gather (the distributed tournament's exchange) through RCCL
        rec = torch.from_numpy(poses.view(np.uint8).reshape(n, 32).copy()).cuda()
        allp = D.allgather_records(rec, n)
        assert allp.shape == (n, 32) and allp.cpu().numpy().tobytes() == poses.tobytes()
        upd.close()
        print("RCCL_WORLD1_OK", flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    _main()
