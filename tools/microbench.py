"""Stage-level micro benchmark: a fixed batch of N in-column pairs, K repetitions, per-stage HIP-event times.
(A fixed batch keeps per-stage timings of kernel variants comparable.)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("VFSMS_LIB"):
    from imagestitch_amd import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ["VFSMS_LIB"])
import imagestitch_amd as isa
from imagestitch_amd.synthetic import SyntheticGrid

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
TILE = int(sys.argv[3]) if len(sys.argv) > 3 else 2048           # 4096: configs[4]'s strips (37 k keypoints each)
eng = isa.Engine(0)
g = SyntheticGrid(10, 9, TILE)
tiles = g.tiles(range(N + 1))
hs = [eng.tile_upload(t) for t in tiles]
ra = isa.roi_rect(tiles[0].shape, 1, "first", 0.2); rb = isa.roi_rect(tiles[0].shape, 1, "second", 0.2)
jobs = [(hs[k], hs[k + 1], ra[0], ra[1], rb[0], rb[1], ra[2], ra[3]) for k in range(N)]
rows = eng.attempt_surf_batch(jobs)
eng.set_keypoint_capacity(int(rows[:, 4:6].max() * 1.5) + 1024)
rows = eng.attempt_surf_batch(jobs)
eng.profile_enable(True); eng.profile_read()
t0 = time.perf_counter()
for _ in range(K):
    rows = eng.attempt_surf_batch(jobs)
dt = time.perf_counter() - t0
prof = eng.profile_read()
print("N=%d pairs/batch  %.2f ms/batch  (%.3f ms/pair)  ok=%d nA=%d" % (N, dt / K * 1e3, dt / K / N * 1e3, rows[:, 0].sum(), rows[0, 4]))
print("  " + "  ".join("%s=%.3f" % (k, v[0] / K) for k, v in prof.items()))
