"""The shading device functions (kernels/shade.hip.h, dmath.hip.h, qmc.hip.h) compiled as host C++ behind a stand-in for
the HIP names they use (profiles/host_shade/) and run on the CPU under MemorySanitizer and ASan + UBSan: no use of an
uninitialised value, no undefined behaviour on 200 000 random vertices covering every OpenPBR lobe, emission, all four
light kinds (lights at infinity included), escaped-ray lookup, camera, filter and the interior-medium functions.
CPU-only; GPU sanitizers are not available on this pool."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm clang++ for -fsanitize=memory")
def test_shading_functions_are_clean_under_msan_and_ubsan(tmp_path):
    env = dict(os.environ, TMPDIR=str(tmp_path))
    res = subprocess.run(["bash", os.path.join(ROOT, "profiles", "host_shade", "run.sh")], capture_output=True, text=True,
                         timeout=900, env=env)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    lines = [l for l in res.stdout.splitlines() if l.startswith("scatter ")]
    assert len(lines) == 2 and lines[0] == lines[1]          # the MSan and the ASan + UBSan build compute the same thing
    counts = dict(zip(("scatter", "eval", "light", "escaped"), (int(lines[0].split()[k]) for k in (1, 3, 6, 8))))
    assert counts["scatter"] > 100000 and counts["eval"] > 100000 and counts["light"] > 100000 and counts["escaped"] > 10000
    assert "MemorySanitizer" not in res.stderr and "runtime error" not in res.stderr and "AddressSanitizer" not in res.stderr
