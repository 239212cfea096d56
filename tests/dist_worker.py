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
from scripted import ScriptedAttemptEngine, random_truth, serpentine_truth  # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 23
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 77
window = int(sys.argv[4]) if len(sys.argv) > 4 else 4
rng = np.random.default_rng(seed)
accept = random_truth(rng, n_pairs, 0.2) if n_pairs < 100 else serpentine_truth(32, 32, 0.2)
SHAPE = (1000, 1400)
eng = ScriptedAttemptEngine(SHAPE, 0.2, accept)
reg = GridRegistrar(eng, roiRatio=0.2, directIncre=1, window=window)
# argv[5]: "hint" = the serpentine's own directions as the scan-pattern hint, "badhint" = the same with wrong entries (forces the repair round)
hint = None
mode = sys.argv[5] if len(sys.argv) > 5 else ""
if mode in ("hint", "badhint"):
    hint = [min(a_, key=lambda c: (c[1], c[0]))[0] if a_ else 1 for a_ in accept]
    if mode == "badhint":
        for k in range(len(hint) // 2 - 120, len(hint) // 2 + 120):   # wrong wherever a 2-rank split can put its cut
            hint[k] = hint[k] % 4 + 1
full, d = reg.register_sharded(list(range(len(accept) + 1)), [SHAPE] * (len(accept) + 1), 1, rank, world, make_all_gather(torch.device("cpu")), hint=hint)
stats = torch.tensor([reg.stats["attempts"], reg.stats["batches"]], dtype=torch.int64)
allstats = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
dist.all_gather(allstats, stats)
if rank == 0:
    json.dump(dict(rows=full.tolist(), direction=int(d), attempts=[int(s[0]) for s in allstats], batches=[int(s[1]) for s in allstats],
                   repairs=int(getattr(reg, "hint_repairs", 0))),
              open(sys.argv[1], "w"))
dist.barrier()
dist.destroy_process_group()
