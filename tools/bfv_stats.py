"""Debug: candidate-filter statistics of k_bf_verify_d64 (library built with -DVFSMS_DESC_TIMING -> tools/libvfsms_timing.so)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagestitch_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvfsms_timing.so")
import imagestitch_amd as isa
from imagestitch_amd.synthetic import SyntheticGrid
eng = isa.Engine(0)
g = SyntheticGrid(10, 9, 2048)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
tiles = g.tiles(range(N + 1)); hs = [eng.tile_upload(t) for t in tiles]
ra = isa.roi_rect(tiles[0].shape, 1, "first", 0.2); rb = isa.roi_rect(tiles[0].shape, 1, "second", 0.2)
jobs = [(hs[k], hs[k + 1], ra[0], ra[1], rb[0], rb[1], ra[2], ra[3]) for k in range(N)]
rows = eng.attempt_surf_batch(jobs); eng.set_keypoint_capacity(int(rows[:, 4:6].max() * 1.5) + 1024)
a = np.zeros(4, np.uint32); eng.lib.vfsms_debug_bfv_stats(a.ctypes.data_as(ctypes.c_void_p))
rows = eng.attempt_surf_batch(jobs)
b = np.zeros(4, np.uint32); eng.lib.vfsms_debug_bfv_stats(b.ctypes.data_as(ctypes.c_void_p))
d = (b - a).astype(np.int64)
print("queries %d  overflowed %d  exact evaluations per query %.2f" % (d[0], d[1], d[3] / max(d[0], 1)))
