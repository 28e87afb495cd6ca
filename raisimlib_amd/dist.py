"""Env sharding across ranks + the one collective of the path (obs all-gather), on torch.distributed.

One process per GPU; backend "nccl" (= RCCL on ROCm, over xGMI) on GPUs, "gloo" in the CPU tests.  Envs are
independent, so rank r owns the contiguous global env range [r*n, (r+1)*n) and nothing inside integrate()
communicates; once per control step every rank contributes its [n, obs_dim] observation block.
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
