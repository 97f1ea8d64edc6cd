"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol include/orbx.h declares, and
fails loudly (no CPU fallback) when no HIP device exists."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "orbx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(orbx_[a-z0-9_]+)\s*\(", text))
    names.discard("orbx_hamming256")  # static inline in the header
    return names


def test_header_symbols_exported(pkg):
    lib = pkg.load_library()
    declared = _declared_symbols()
    assert declared == set(pkg.ABI_SYMBOLS), declared ^ set(pkg.ABI_SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s


def test_header_cites_reference_interfaces():
    text = open(os.path.join(ROOT, "include", "orbx.h")).read()
    for cite in ["include/ORBextractor.h:58-59", "include/ORBextractor.h:66-68", "src/ORBextractor.cc:1227-1307",
                 "src/ORBmatcher3.cc:637-653", "src/Frame.cc:45,1174", "src/Frame.cc:841-925", "include/ORBextractor.h:92"]:
        assert cite in text, cite


def test_keypoint_layout(pkg):
    assert pkg.KEYPOINT_DTYPE.itemsize == 28
    assert [pkg.KEYPOINT_DTYPE.fields[n][1] for n in pkg.KEYPOINT_DTYPE.names] == [0, 4, 8, 12, 16, 20, 24]


def test_invalid_arguments_rejected(pkg):
    lib = pkg.load_library()
    h = ctypes.c_void_p()
    p = pkg._Params(0, 1.2, 8, 20, 7, 0, 640, 480, 1, 0, 0)
    assert lib.orbx_create(ctypes.byref(p), ctypes.byref(h)) == -1 and not h.value
    assert b"nfeatures" in lib.orbx_last_error()
    p = pkg._Params(1000, 1.0, 8, 20, 7, 0, 640, 480, 1, 0, 0)
    assert lib.orbx_create(ctypes.byref(p), ctypes.byref(h)) == -1
    assert lib.orbx_create(None, ctypes.byref(h)) == -1
    assert lib.orbx_sync(None) == -1
    lib.orbx_destroy(None)  # no-op


def test_no_cpu_fallback(pkg, gpu_available):
    if gpu_available:
        pytest.skip("GPU present")
    with pytest.raises(pkg.OrbxError, match="no HIP device"):
        pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    with pytest.raises(pkg.OrbxError, match="no HIP device"):
        pkg.ORBmatcher(16, 16)


def test_host_hamming_leaf_matches_oracle(pkg, orc):
    rng = np.random.default_rng(3)
    for _ in range(50):
        a = rng.integers(0, 256, 32, dtype=np.uint8)
        b = rng.integers(0, 256, 32, dtype=np.uint8)
        assert pkg.ORBmatcher.DescriptorDistance(a, b) == orc.hamming256(a, b)


def test_synth_generator_identical_in_both_libraries(pkg, orc):
    for idx, (w, h) in [(0, (640, 480)), (5, (752, 480)), (11, (512, 512))]:
        assert np.array_equal(pkg.synth_frame(idx, w, h), orc.synth_frame(idx, w, h))


def test_inline_hamming_in_header_compiles_and_matches(tmp_path, orc):
    src = tmp_path / "t.c"
    src.write_text('#include "orbx.h"\n#include <stdio.h>\nint main(void){unsigned char a[32],b[32];for(int i=0;i<32;i++){a[i]=(unsigned char)(i*7+1);b[i]=(unsigned char)(255-i*3);}printf("%d\\n",orbx_hamming256(a,b));return 0;}\n')
    exe = tmp_path / "t"
    import subprocess
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = int(subprocess.check_output([str(exe)]).decode())
    a = np.array([(i * 7 + 1) & 255 for i in range(32)], np.uint8)
    b = np.array([(255 - i * 3) & 255 for i in range(32)], np.uint8)
    assert got == orc.hamming256(a, b)


def test_known_switch_list_matches_the_sources(pkg):
    """Every getenv("ORBX_..") of the native sources is in the library's list of known switches (orbx_known_switches) and nothing else is -
    apart from the three names of the Python side / tooling; under ORBX_STRICT_ENV=1 an unknown ORBX_* variable makes orbx_create fail, so
    the list must not rot."""
    import ctypes, glob, os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    used = set()
    for f in glob.glob(os.path.join(root, "wut-cuda-orb-slam3_amd", "csrc", "*.*")):
        if f.endswith((".cpp", ".hip", ".h")):
            used |= set(re.findall(r'getenv\("(ORBX_[A-Z0-9_]+)"\)', open(f).read()))
    lib = pkg.load_library()
    arr = (ctypes.c_char_p * 256)()
    lib.orbx_known_switches.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.c_int]
    n = lib.orbx_known_switches(arr, 256)
    known = {arr[i].decode() for i in range(n)}
    assert known - {"ORBX_LIB", "ORBX_BENCH_BACKEND", "ORBX_COMMIT"} == used, (sorted(known - used), sorted(used - known))
    design = open(os.path.join(root, "DESIGN.md")).read()
    missing = [k for k in sorted(used) if k not in design]
    assert not missing, f"switches not described in DESIGN.md's appendix: {missing}"


def test_unknown_switch_is_reported():
    """ORBX_PF_KERNAL=2 (a typo) must not silently run the default: with ORBX_STRICT_ENV=1 (tests, A / B scripts) orbx_create fails with the
    switch's name - before it looks for a device, so this runs on the CPU box too (where, without the typo, the same call fails with 'no HIP
    device'); without it (production: some other tool's ORBX_HOME must not take the path down) the library warns once on stderr and goes on."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import __graft_entry__ as g\npkg = g.load_package()\n"
            "try:\n    pkg.ORBextractor(1000, 1.2, 8, 20, 7)\n    print('CREATED')\nexcept pkg.OrbxError as e:\n    print('ERR', e)\n")
    for env_extra, expect in (({"ORBX_PF_KERNAL": "2", "ORBX_STRICT_ENV": "1"}, "orbx_create: unknown environment switch ORBX_PF_KERNAL"),
                              ({"ORBX_PF_KERNAL": "2", "ORBX_STRICT_ENV": "0"}, "warning: unknown environment switch ORBX_PF_KERNAL is ignored"),
                              ({"ORBX_PF_KERNEL": "1", "ORBX_STRICT_ENV": "1"}, None)):
        env = dict(os.environ, **env_extra)
        out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120).stdout
        if expect:
            assert expect in out, out[-500:]
            if env_extra["ORBX_STRICT_ENV"] == "0":
                assert "orbx_create: unknown" not in out and out.count("warning: unknown environment switch") == 1, out[-500:]
        else:
            assert "unknown environment switch" not in out, out[-500:]
