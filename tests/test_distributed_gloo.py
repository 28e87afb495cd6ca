"""N>1 path on CPU: world_size-2 gloo run of the env-sharding + obs all-gather logic (raisimlib_amd/dist.py).

Each rank steps ITS shard with the oracle standing in for the device world (this is a test; the product path has
no CPU fallback), builds the obs block and all-gathers it; the result must equal the single-process run over all
envs — i.e. sharding by global env index changes nothing, bit for bit (every workload quantity is a function of the
global env index).  Both issue policies of raisimlib_amd.dist.ObsGatherer (in line / double-buffered overlap) are run."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from common import ROOT


def obs_block(model, q, u, contacts, counts, feet, dt):
    n = q.shape[0]
    f = np.zeros((n, 3 * len(feet)))
    for e in range(n):
        for c in contacts[e][:counts[e]]:
            if c["collision"] in feet:
                k = feet.index(c["collision"])
                f[e, 3 * k:3 * k + 3] = c["impulse"] / dt
    return np.concatenate([q, u, f], axis=1).astype(np.float32)


def run_shard(lo, hi):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle.pyoracle import Oracle
    from raisimlib_amd import Model, rsc_path, workload
    model = Model(urdf_path=rsc_path("anymal_c_like.urdf"))
    feet = model.collision_indices("_foot")
    orc = Oracle(model.blob)
    n = hi - lo
    gc, gv = workload.anymal_initial_state(n, env_offset=lo, height=0.56)
    kp, kd = workload.anymal_gains()
    q, u = gc, gv
    for cs in range(3):
        pt = workload.anymal_targets(n, cs, env_offset=lo)
        r = orc.step_batch(q, u, 4, kp.astype(np.float64), kd.astype(np.float64), pt, np.zeros((n, 18)), want_contacts=True)
        q, u = r["q"], r["u"]
    return obs_block(model, q, u, r["contacts"], r["n_contacts"], feet, 0.0025)


def worker(rank, world, n, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from raisimlib_amd.dist import ObsGatherer, env_range, gather_obs
    lo, hi = env_range(rank, n)
    local = torch.from_numpy(run_shard(lo, hi))
    full = gather_obs(local)
    assert full.shape == (world * n, local.shape[1])
    # the bench's issue policies: three "control steps" whose obs blocks differ by a known offset
    for overlap in (False, True):
        g = ObsGatherer(n, local.shape[1], torch.device("cpu"), overlap=overlap)
        assert g.active and g.nbuf == (2 if overlap else 1)
        seen = []
        for k in range(3):
            g.acquire(k)
            g.local(k).copy_(local + float(k))          # stands for the step kernel writing this rank's block
            g.gather(k)
            if not overlap:
                seen.append(g.gathered(k).clone())
            elif k >= 1:                                # the previous step's gather is complete once its slot is re-acquired
                g.acquire(k + 1)
                seen.append(g.gathered(k - 1).clone())
        g.drain()
        if overlap:
            seen.append(g.gathered(2).clone())
        for k, blk in enumerate(seen[:3]):
            assert torch.equal(blk, full + float(k)), (overlap, k)
    if rank == 0:
        np.save(out, full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_obs_gather_equals_single_process(tmp_path, built_lib):
    n, world = 6, 2
    out = str(tmp_path / "gathered.npy")
    mp.spawn(worker, args=(world, n, 29517, out), nprocs=world, join=True)
    gathered = np.load(out)
    ref = run_shard(0, 2 * n)                           # ONE process stepping all 12 envs: sharded == unsharded, bit for bit
    assert gathered.shape == (12, 19 + 18 + 12)
    assert np.array_equal(gathered, ref)
    assert np.array_equal(ref, np.concatenate([run_shard(0, n), run_shard(n, 2 * n)], axis=0))
    assert np.abs(gathered[:, 37:]).max() > 0          # feet were in contact: the force slots are populated
