"""The CPU sanitizer leg (host ASan/UBSan for the CPU twin; GPU sanitizers do not exist on this pool).

* oracle/liboracle_san.so: oracle_f64.c + oracle_threads.c built with -fsanitize=address,undefined -fno-sanitize-recover;
  the golden-render, analytic, Philox, PNG-pin and multi-thread tests run against it in a child interpreter started with
  libasan preloaded.  A heap/stack/global overflow, a use after free, signed overflow, a misaligned or null access in
  the oracle aborts that run.
* host/rtiow_render_san: the C++ host mirror (host/rtiow_host.hpp: Vec3, Camera, random_scene, the flat-scene writer, and
  host/rtiow_multi.hpp's reassembly plan) under the same sanitizers; its scene bytes must still equal the Python mirror's.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import rtiow_amd as rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN_ENV = {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1:halt_on_error=1", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"}


def _gcc_file(name):
    return subprocess.run(["gcc", "-print-file-name=" + name], check=True, capture_output=True, text=True).stdout.strip()


def test_oracle_under_asan_ubsan_passes_its_pins():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle_san.so"], check=True)
    san = os.path.join(ROOT, "oracle", "liboracle_san.so")
    needed = subprocess.run(["readelf", "-d", san], capture_output=True, text=True).stdout
    assert "libasan" in needed and "libubsan" in needed                 # it IS the instrumented build
    env = dict(os.environ, ORACLE_LIB=san, LD_PRELOAD=_gcc_file("libasan.so"), **SAN_ENV)
    tests = ["tests/test_oracle_golden.py", "tests/test_oracle_analytic.py", "tests/test_oracle_philox.py",
             "tests/test_oracle_sky_png.py", "tests/test_oracle_png_spheres.py", "tests/test_uniform53.py", "tests/test_distributed_gloo.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider", *tests],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail


def test_cpp_host_under_asan_ubsan_builds_the_same_scene(tmp_path):
    exe = os.path.join(ROOT, "host", "rtiow_render_san")
    import __graft_entry__ as g
    g.build_host_cli(sanitize=True)
    env = dict(os.environ, **SAN_ENV)
    for args, seed, grid in ([], 1, (-11, 11)), (["--grid", "-50", "49", "--scene-seed", "3"], 3, (-50, 49)):
        path = str(tmp_path / "scene.bin")
        r = subprocess.run([exe, "--dump-scene", path, *args], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-2000:]
        got = np.fromfile(path, dtype=rt.SPHERE_DTYPE)
        assert got.tobytes() == rt.random_scene(seed, grid=grid).flatten().tobytes()
    r = subprocess.run([exe, "--reassembly-plan", "4320", "1", "8"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and len(r.stdout.splitlines()) == 8 and "runtime error" not in r.stderr, r.stderr[-2000:]
    # the PNG writer (own CRC-32 / Adler-32 / stored-block framing) under the sanitizers: several blocks, and the smallest image
    for w, h in ((301, 207), (1, 1)):
        out = str(tmp_path / "p.png")
        r = subprocess.run([exe, "--test-png", str(w), str(h), out], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-2000:]
        assert rt.read_png(out).shape == (h, w, 4)
    # a truncated scene file is refused, not read past its end
    bad = str(tmp_path / "bad.bin")
    open(bad, "wb").write(b"\0" * 100)
    r = subprocess.run([exe, "--scene", bad, "--out", str(tmp_path / "x.ppm")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "72-byte records" in r.stderr and "AddressSanitizer" not in r.stderr
