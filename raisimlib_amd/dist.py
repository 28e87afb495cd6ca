"""Env sharding across ranks + the one collective of the path (obs all-gather), on torch.distributed.

One process per GPU; backend "nccl" (= RCCL on ROCm, over xGMI) on GPUs, "gloo" in the CPU tests.  Envs are
independent, so rank r owns the contiguous global env range [r*n, (r+1)*n) and nothing inside integrate()
communicates; once per control step every rank contributes its [n, obs_dim] observation block
(q 19, u 18, foot force 12 = 196 B per ANYmal env: 0.8 MB per rank at n = 4096).

Every workload quantity is a function of the GLOBAL env index (raisimlib_amd/workload.py), so a sharded run produces
exactly the rows of the unsharded one (tests/test_distributed_gloo.py asserts it bit for bit).

`ObsGatherer` owns the per-rank and gathered buffers and the two ways of issuing the collective:
  in line      the all-gather of control step k is enqueued behind the step's kernel on the same stream (default);
  overlapped   double-buffered: the all-gather of step k runs on the backend's own stream while the kernel of step k+1
               writes the other buffer; a buffer is only rewritten after the gather that reads it has finished.
  pipelined    (pipeline_world = the BatchedWorld whose control steps are pipelined, rsb_set_step_pipelining) double-buffered on a stream of
               its own: the gather of step k is ordered behind step k ALONE (rsb_step_pipeline_publish), step k + 2 behind the gather that
               still reads its buffer (rsb_step_pipeline_wait_event); the pipeline of control steps is never joined.
The C++ host side has the same collective without Python: rsb_comm_* / rsb_allgather_obs in include/rsb.h.
"""
import torch
import torch.distributed as dist


def env_range(rank, envs_per_rank):
    """Global env indices owned by `rank` (seeds are derived from the GLOBAL index, so results are shard-invariant)."""
    return rank * envs_per_rank, (rank + 1) * envs_per_rank


def gather_obs(local_obs: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """All-gather of the per-rank observation block along the env axis: [n, d] -> [world*n, d] (rank-major)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local_obs
    world = dist.get_world_size()
    if out is None:
        out = torch.empty((world * local_obs.shape[0], local_obs.shape[1]), dtype=local_obs.dtype, device=local_obs.device)
    dist.all_gather_into_tensor(out, local_obs.contiguous())
    return out


class ObsGatherer:
    """Buffers + issue policy of the per-control-step obs all-gather (see the module docstring)."""

    def __init__(self, n_local, obs_dim, device, overlap=False, force=False, dtype=torch.float32, pipeline_world=None):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.pipe = pipeline_world if self.active else None
        if self.pipe is not None:
            self.side = torch.cuda.Stream(device=device)
            self.done = [torch.cuda.Event(), torch.cuda.Event()]     # recorded behind the gather that reads buffer b
            self.used = [False, False]
        self.nbuf = 2 if (self.active and (overlap or self.pipe is not None)) else 1
        self.local_bufs = [torch.empty((n_local, obs_dim), dtype=dtype, device=device) for _ in range(self.nbuf)]
        self.all_bufs = ([torch.empty((self.world * n_local, obs_dim), dtype=dtype, device=device) for _ in range(self.nbuf)]
                         if self.active else self.local_bufs)
        self.pending = [None] * self.nbuf

    def slot(self, k):
        return k % self.nbuf

    def local(self, k):
        """The buffer control step k writes its obs block into."""
        return self.local_bufs[self.slot(k)]

    def gathered(self, k):
        """[world*n, d] block of control step k (valid after drain(), or after the next acquire() of the same slot)."""
        return self.all_bufs[self.slot(k)]

    def acquire(self, k):
        """Before step k's kernel is enqueued: wait (stream-side) for the gather that still reads this slot."""
        b = self.slot(k)
        if self.pipe is not None:
            if self.used[b]:
                self.pipe.step_pipeline_wait_event(self.done[b].cuda_event)
            return
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None

    def gather(self, k):
        """After step k's kernel is enqueued: issue its all-gather."""
        if not self.active:
            return
        b = self.slot(k)
        if self.pipe is not None:
            self.pipe.step_pipeline_publish(self.side.cuda_stream)
            with torch.cuda.stream(self.side):
                dist.all_gather_into_tensor(self.all_bufs[b], self.local_bufs[b])
                self.done[b].record(self.side)
            self.used[b] = True
            return
        if self.nbuf == 2:
            self.pending[b] = dist.all_gather_into_tensor(self.all_bufs[b], self.local_bufs[b], async_op=True)
        else:
            dist.all_gather_into_tensor(self.all_bufs[b], self.local_bufs[b])

    def drain(self):
        if self.pipe is not None:
            torch.cuda.current_stream().wait_stream(self.side)
            return
        for b in range(self.nbuf):
            if self.pending[b] is not None:
                self.pending[b].wait()
                self.pending[b] = None

    def describe(self):
        if not self.active:
            return "none (1 rank)"
        if self.pipe is not None:
            return "on a stream of its own behind each pipelined control step (double-buffered; publish / wait-event, the pipeline is never joined)"
        return "overlapped with the next control step (double-buffered)" if self.nbuf == 2 else "in line"


class PeerObsGatherer:
    """The same job without a collective: every rank's gathered block is mapped by the other ranks (hipIpc handles exchanged once
    through torch.distributed), the step kernel's epilogue stores each env's obs row into all of them (write-through stores) and
    the last wave of the launch writes the step number into every rank's flag word; `gather` only enqueues the stream-side wait (rsb_obs_peer_wait).  Interface of ObsGatherer, so bench.py can
    switch with a flag; `local_bufs` is [None]: the control step needs no obs block of its own."""

    def __init__(self, world, force_collisions, force=False, no_wait=False):
        self.world = world
        self.no_wait = no_wait          # diagnostic: the producer side only (rows + flags), nobody waits for them
        self.ranks = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.active = self.ranks > 1 or force
        self.nbuf = 1
        self.local_bufs = [None]
        self._last = None
        if not self.active:
            return
        handle = world.obs_peer_create(self.ranks, self.rank, force_collisions)
        if self.ranks > 1:
            every = [None] * self.ranks
            dist.all_gather_object(every, handle)
            world.obs_peer_connect(b"".join(every))
        else:
            world.obs_peer_connect(handle)          # one rank: the own entry is the only one (and is ignored)

    def slot(self, k):
        return 0

    def acquire(self, k):
        pass                                        # the library double-buffers by control-step parity

    def gather(self, k):
        if self.active and not self.no_wait:
            self._last = self.world.obs_peer_wait()

    def gathered_ptr(self):
        """device pointer of the last complete [ranks * n, obs_dim] block"""
        return self._last

    def drain(self):
        pass

    def describe(self):
        if not self.active:
            return "none (1 rank)"
        return "peer-mapped buffers: rows stored by the step kernel's epilogue into every rank's block, stream-side flag wait (no collective, no copy kernel)"
