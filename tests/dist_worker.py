"""torchrun worker for test_grid_registrar.test_two_process_gloo_all_gather (CPU, gloo)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from imagestitch_amd.grid import GridRegistrar  # noqa: E402
from imagestitch_amd.distributed import make_all_gather  # noqa: E402
from scripted import ScriptedAttemptEngine, random_truth  # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(77)
accept = random_truth(rng, 23, 0.2)
SHAPE = (1000, 1400)
reg = GridRegistrar(ScriptedAttemptEngine(SHAPE, 0.2, accept), roiRatio=0.2, directIncre=1, window=4)
full, d = reg.register_sharded(list(range(24)), [SHAPE] * 24, 1, rank, world, make_all_gather(torch.device("cpu")))
if rank == 0:
    json.dump(dict(rows=full.tolist(), direction=int(d)), open(sys.argv[1], "w"))
dist.barrier()
dist.destroy_process_group()
