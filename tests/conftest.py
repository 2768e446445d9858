import os, subprocess, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle_bin():
    """The plain-C oracle (test infrastructure).  Built on demand with gcc."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True)
    return os.path.join(ROOT, "oracle", "build", "dwgsim_oracle")


@pytest.fixture(scope="session")
def oracle_lib(oracle_bin):
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "build", "liboracle.so"))
    lib.oracle_det_log.restype = ctypes.c_double
    lib.oracle_det_log.argtypes = [ctypes.c_double]
    lib.oracle_drand48_next.restype = ctypes.c_double
    lib.oracle_drand48_next.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    lib.oracle_philox_uniform.restype = ctypes.c_double
    lib.oracle_philox_uniform.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64,
                                          ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    return lib


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
