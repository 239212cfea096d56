"""Worker of test_gpu_parity.test_rccl_all_gather_at_world_size_1 (GPU box): the pair-sharded registration path with the REAL collective --
torch.distributed backend "nccl" (= RCCL on ROCm) on cuda:LOCAL_RANK -- against the single-process registration of the same tiles.
Runs with any world size the box has GPUs for (the driver's 8-GPU node: torchrun --nproc-per-node N tests/rccl_worker.py out.json);
the GPU test starts it at world size 1.  Writes {"rows", "rows_single", "backend", "world", "device", "gathered_shape"} on rank 0."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import imagestitch_amd as isa  # noqa: E402
from imagestitch_amd.distributed import make_all_gather, single_process_all_gather  # noqa: E402
from imagestitch_amd.grid import GridRegistrar  # noqa: E402
from imagestitch_amd.synthetic import SyntheticGrid  # noqa: E402

rank = int(os.environ.get("RANK", "0"))
world = int(os.environ.get("WORLD_SIZE", "1"))
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
torch.cuda.set_device(local_rank)
device = torch.device("cuda", local_rank)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
assert dist.get_backend() == "nccl"

eng = isa.Engine(local_rank)
g = SyntheticGrid(2, 3, 768, overlap=0.15)                 # 5 pairs on a serpentine with two turns
tiles = g.tiles(threads=1)
shapes = [t.shape for t in tiles]
hs = [eng.tile_upload(t) for t in tiles]


def registrar():
    return GridRegistrar(eng, method="surf", roiRatio=0.2, searchRatio=0.75, offsetEvaluate=3, directIncre=1, surfParams=eng.surf_params(), window=8)


gather = make_all_gather(device)
probe = gather(np.arange(7, dtype=np.int32) + 100 * rank)   # the collective itself: device tensor in, [world, C] host table out
assert probe.shape == (world, 7) and all((probe[r] == np.arange(7) + 100 * r).all() for r in range(world)), probe
full, d = registrar().register_sharded(hs, shapes, 1, rank, world, gather)
# second path on the same registrar form: path memory primed (one hinted chain per rank, repair round if the hint were wrong)
reg2 = registrar()
reg2.register_sharded(hs, shapes, 1, rank, world, gather)
full2, d2 = reg2.register_sharded(hs, shapes, 1, rank, world, gather)
single, ds = registrar().register_sharded(hs, shapes, 1, 0, 1, single_process_all_gather)
plain, dp = registrar().register(hs, shapes, 1)
if rank == 0:
    json.dump(dict(rows=np.asarray(full).tolist(), rows_primed=np.asarray(full2).tolist(), rows_single=np.asarray(single).tolist(),
                   rows_register=np.asarray(plain).tolist(), direction=[int(d), int(d2), int(ds), int(dp)],
                   truth=[list(map(int, o)) for o in g.true_offsets()], backend=dist.get_backend(), world=world, device=str(device),
                   gathered_shape=list(probe.shape), repairs=int(getattr(reg2, "hint_repairs", 0))), open(sys.argv[1], "w"))
dist.barrier()
dist.destroy_process_group()
eng.close()
