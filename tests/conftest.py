"""Shared fixtures.  `-m "not gpu"` runs anywhere; `-m gpu` needs an MI355X and the in-tree libraries."""
import importlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


def _built():
    need = [os.path.join(ROOT, "ray-tracer_amd", "lib", n) for n in ("librtamd.so", "librtamd_host.so")]
    need.append(os.path.join(ROOT, "oracle", "build", "liboracle.so"))
    need.append(os.path.join(ROOT, "oracle", "build", "kat_tests"))
    return all(os.path.exists(p) for p in need)


@pytest.fixture(scope="session", autouse=True)
def _libraries():
    """The libraries are built in-tree beforehand (`python -c 'import __graft_entry__ as g; g.build()'`);
    a fresh checkout without them gets one incremental build here."""
    if not _built():
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def rt(_libraries):
    return importlib.import_module("ray-tracer_amd")


@pytest.fixture(scope="session")
def oracle(_libraries):
    import oracle_lib
    return oracle_lib


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)["scenes"]


def sg_path(name):
    return os.path.join(GOLDEN, "scenes", name + ".sg")


def golden_frame(name, size):
    return np.load(os.path.join(GOLDEN, "oracle", f"{name}_{size[0]}x{size[1]}.npy"))


@pytest.fixture(scope="session")
def gpu_ctx(rt):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: there is no CPU fallback for the render path")
    ctx = rt.Context(0)
    yield ctx
    ctx.close()


ALL_SCENES = ["default_world", "reflect_refract", "rr_code", "cover", "cylinders", "tables", "glass_spheres", "groups",
              "puppets", "csg", "teapot_lowres_room", "teapot_room"]
