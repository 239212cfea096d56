"""CPU tests of the C-ABI boundary: the shared library loads without a GPU and exports exactly what
include/vfsms.h declares; the Python binding covers every declared entry point; no compute is called."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

import imagestitch_amd as isa
from imagestitch_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "vfsms.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vfsms_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(isa.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    lib = ctypes.CDLL(isa.LIB_PATH)
    names = _declared()
    assert len(names) >= 28
    for n in names:
        assert hasattr(lib, n), "libvfsms.so lacks %s declared in include/vfsms.h" % n
    out = subprocess.check_output(["nm", "-D", "--defined-only", isa.LIB_PATH]).decode()
    exported = sorted(set(re.findall(r" T (vfsms_[a-z0-9_]+)", out)))
    assert exported == names, "exported symbols and header declarations differ"


def test_python_binding_covers_the_header():
    assert sorted(_lib.exported_names()) == _declared()
    lib = isa.load_library()
    assert lib.vfsms_version() >= 100
    buf = ctypes.create_string_buffer(64)
    assert lib.vfsms_last_error(buf, 64) == 0


def test_no_cpu_fallback_without_a_device():
    """Without a visible GPU the product refuses to run: there is no silent CPU path."""
    lib = isa.load_library()
    if lib.vfsms_device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(isa.VfsmsError):
        isa.Engine(0)
    m = isa.Method()
    import numpy as np
    with pytest.raises(isa.VfsmsError):
        m.detectAndDescribe(np.zeros((64, 64), np.uint8), "surf")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "imagestitch_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in src and "from oracle" not in src and "vfsms_oracle" not in src, f


def test_reference_module_names_resolve_to_the_engine():
    """with imagestitch_amd/compat on sys.path the reference's imports work unchanged (Main.py:1 `from Stitcher import Stitcher`,
    Stitcher.py:10-11 `import ImageUtility as Utility`, `import ImageFusion`)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("from Stitcher import Stitcher; import ImageUtility as Utility; import ImageFusion; import imagestitch_amd as isa; "
            "assert Stitcher is isa.Stitcher and Utility.Method is isa.Method and ImageFusion.ImageFusion is isa.ImageFusion; "
            "assert issubclass(Stitcher, Utility.Method); print('ok')")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([root, os.path.join(root, "imagestitch_amd", "compat")]))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd="/tmp")
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr


def test_phase_plan_picks_the_lds_transforms_for_the_strips_of_the_path():
    """vfsms_phase_plan does no device work: which strips the hand-written LDS transforms take (csrc/phase_kernels.hip), in which orientation, and
    which stay with rocFFT.  The column axis is the SHORTER padded one (a tall strip is correlated as its transpose), the row length must be even,
    the padded sizes are cv2.getOptimalDFTSize's (409 -> 432, 819 -> 864, 387 -> 400, 2584 -> 2592)."""
    import numpy as np
    lib = isa.load_library()

    def plan(h, w):
        info = np.zeros(8, np.int32)
        assert lib.vfsms_phase_plan(h, w, info.ctypes.data_as(ctypes.c_void_p)) == 0
        return [int(v) for v in info]
    assert plan(409, 2048)[:4] == [1, 0, 432, 2048]               # strip above / below at 2048^2 tiles, roiRatio 0.2
    assert plan(2048, 409)[:4] == [1, 1, 432, 2048]               # strip left / right: transposed first
    assert plan(819, 4096)[:4] == [1, 0, 864, 4096]               # configs[4]
    assert plan(387, 2584)[:4] == [1, 0, 400, 2592]               # the dendriticCrystal tiles (1936 x 2584)
    assert plan(97, 131)[:4] == [1, 1, 135, 100]                  # 131 pads to the odd 135: only the other axis can be the packed real one
    assert plan(625, 625)[0] == 0 and plan(625, 625)[2:4] == [625, 625]      # odd both ways: rocFFT
    assert plan(3000, 3000)[0] == 0                               # a 3000-point column does not fit LDS twice over: rocFFT
    p = plan(409, 2048)
    assert p[4] in (2, 4, 8) and p[5] >= 1 and p[6] % 64 == 0 and p[7] % 64 == 0 and 2 * p[4] * p[2] <= 8 * p[7] and p[5] * p[3] // 2 <= 8 * p[6]
    assert lib.vfsms_phase_plan(0, 5, np.zeros(8, np.int32).ctypes.data_as(ctypes.c_void_p)) != 0
