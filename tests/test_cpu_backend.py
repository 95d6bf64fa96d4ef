"""The CPU build of the C ABI (libsd_cpu.so: BASELINE.json configs[0], "plumbing, no GPU").
The backend is chosen per process (SD_AMD_BACKEND), so the cases run in a child process; this
process keeps the GPU library, which is what every other test loads."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_library_exports_the_whole_abi():
    import ctypes
    from vllm_omni_draft_amd import _lib
    path = os.path.join(ROOT, "vllm_omni_draft_amd", "libsd_cpu.so")
    if not os.path.exists(path):
        subprocess.run(["make", "-C", ROOT, "vllm_omni_draft_amd/libsd_cpu.so"], check=True)
    lib = ctypes.CDLL(path)
    for name in _lib.SIGNATURES:
        assert hasattr(lib, name), name


def test_cpu_backend_cases_in_their_own_process():
    env = dict(os.environ, SD_AMD_BACKEND="cpu", OMP_NUM_THREADS=str(min(8, os.cpu_count() or 1)))
    env.pop("SD_AMD_LIB", None)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-o", "python_files=case_*.py",
                        "-p", "no:cacheprovider", os.path.join("tests", "cpu_backend")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
