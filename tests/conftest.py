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


# The command-line renderer is a program of its own, and a process that holds the GPU may not start another one: it is run
# here, at the start of a session that asked for GPU tests and before anything in this process has touched the GPU; the
# test that checks its output (tests/test_gpu_cli.py) finds the record below.
CLI_RUNS = {}


def pytest_sessionstart(session):
    import subprocess
    import tempfile
    expr = session.config.getoption("-m") or ""
    cli = os.path.join(ROOT, "ray-tracer_amd", "bin", "yaml_renderer")
    if "gpu" not in expr or expr.strip().startswith("not") or not os.path.exists("/dev/kfd") or not os.path.exists(cli):
        return
    from test_host import YAML  # a scene of this repository's own (no file of the reference travels)
    d = tempfile.mkdtemp(prefix="rtamd_cli_")
    path = os.path.join(d, "own_scene.yaml")
    with open(path, "w") as f:
        f.write(YAML.replace("width: 20", "width: 200").replace("height: 10", "height: 100"))
    CLI_RUNS["dir"], CLI_RUNS["yaml"] = d, path
    for key, args in (("default", [path]), ("flags", [path, "--width", "96", "--height", "64", "--strict", "--ppm", "-o", os.path.join(d, "small.ppm")])):
        try:
            CLI_RUNS[key] = subprocess.run([cli] + args, cwd=d, capture_output=True, text=True, timeout=120)
        except Exception as e:  # the test reports it
            CLI_RUNS[key] = e
    # a 3840x2160 frame: the 8-bit channels alone (what the PNG holds), then Canvas.pixels as well; five more frames each
    for key, extra in (("4k_u8", []), ("4k_f64", ["--f64"])):
        try:
            CLI_RUNS[key] = subprocess.run([cli, path, "--width", "3840", "--height", "2160", "--repeat", "5", "-o", os.path.join(d, key + ".png")] + extra,
                                           cwd=d, capture_output=True, text=True, timeout=120)
        except Exception as e:
            CLI_RUNS[key] = e
    # a host in plain C that hands its object tree over call by call (examples/build_world_c.c)
    try:
        CLI_RUNS["build_world_c"] = subprocess.run([os.path.join(ROOT, "ray-tracer_amd", "bin", "build_world_c"), "100", "50", os.path.join(d, "csg.f64")],
                                                   cwd=d, capture_output=True, text=True, timeout=120)
    except Exception as e:
        CLI_RUNS["build_world_c"] = e
    # the host in plain C (examples/render_c.c): one GPU, then the same GPU three times as the ranks of rtamd_multi_*
    try:
        CLI_RUNS["render_c"] = subprocess.run([os.path.join(ROOT, "ray-tracer_amd", "bin", "render_c"), "400", "225", "0", "0", "0"], cwd=d,
                                              capture_output=True, text=True, timeout=120)
    except Exception as e:
        CLI_RUNS["render_c"] = e


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
