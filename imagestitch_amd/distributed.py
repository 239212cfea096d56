"""torch.distributed plumbing for the pair-sharded path: one process per GPU, one collective per dataset.

backend "nccl" IS RCCL on ROCm (xGMI between the 8 GPUs of a node); "gloo" is used by the CPU tests.
The payload is the int32 offset table of grid.GridRegistrar.shard_payload -- at most a few tens of KB, so
the collective is latency-bound and is issued exactly once per registered path.
"""
import numpy as np


def make_all_gather(device):
    """-> all_gather(int32 ndarray [C]) -> int32 ndarray [world, C] over the default process group."""
    import torch
    import torch.distributed as dist

    def all_gather(payload):
        world = dist.get_world_size()
        t = torch.from_numpy(np.ascontiguousarray(payload, np.int32)).to(device)
        out = torch.empty(world * t.numel(), dtype=torch.int32, device=device)
        dist.all_gather_into_tensor(out, t)
        return out.view(world, t.numel()).cpu().numpy()
    return all_gather


def single_process_all_gather(payload):
    return np.asarray(payload, np.int32)[None, :]
