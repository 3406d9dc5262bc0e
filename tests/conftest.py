import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def mvs():
    """The product package (directory name has a hyphen)."""
    return importlib.import_module("multiview-simulation_amd")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("multiview-simulation_amd.synthetic")


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def ctx(mvs):
    """A GPU context; GPU tests fail (not skip) when the HIP library or device is missing."""
    c = mvs.Context(0)
    # the small test volumes go through the fused rotate+attenuate kernel of the 512^3 workload as well (the default,
    # "auto", hands volumes below 2 waves per SIMD to the two separate kernels)
    c.set_option("fused_rotate", 1)
    yield c
    c.close()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def psf51_tif(tmp_path_factory, mvs, synth):
    """An ImageJ float32 stack like the reference's `Angle0.tif` (51 planes of 51 x 51, big-endian), synthesised and
    written with Tools.save: the reference's PSF files are GPL data and are not copied into this repository."""
    path = str(tmp_path_factory.mktemp("psf") / "Angle0.tif")
    mvs.Tools.save(synth.measured_like_psf(51), path)
    return path


def rel_to_max(a, b):
    """max|a-b| / max|b| -- the range-normalised error of SURVEY.md H2."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
