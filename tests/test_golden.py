"""Committed fixture (tests/golden/cornell_40x24.npz, written by tests/golden/make_golden.py from the
CPU oracle): the oracle must keep reproducing it (CPU test) and the HIP path must match it (GPU test)."""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _golden():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.golden_inputs(), np.load(os.path.join(HERE, "golden", "cornell_40x24.npz"))


def test_oracle_reproduces_fixture(pt, po):
    (scene, sky, st, w, h, times), ref = _golden()
    osc = po.from_scene(scene, w, h, st, sky)
    acc = np.zeros((h, w, 4), np.float32)
    rays = 0
    for t in times:
        acc, prim, tt, stats = osc.render(t, acc, want_aov=True, threads=3)
        rays += stats["rays"]
    assert np.array_equal(prim, ref["prim"]) and rays == int(ref["rays"])
    assert np.array_equal(tt.view(np.uint32), ref["t"].view(np.uint32))
    assert np.array_equal(acc.view(np.uint32), ref["accum"].view(np.uint32))


@pytest.mark.gpu
def test_hip_matches_fixture(pt):
    (scene, sky, st, w, h, times), ref = _golden()
    r = pt.Renderer(scene, w, h, settings=st, flags=pt.PT_FLAG_AOV)
    r.set_sky_cube(sky)
    for t in times:
        r.render(t)
    prim, tt = r.read_aov()
    assert np.array_equal(prim, ref["prim"]) and r.stats()["rays"] == int(ref["rays"])
    np.testing.assert_allclose(tt, ref["t"], rtol=1e-4)
    np.testing.assert_allclose(r.read_sum(), ref["accum"], rtol=1e-4, atol=1e-6)
    assert np.array_equal(r.read_sum().view(np.uint32), ref["accum"].view(np.uint32)), "within 1e-4 but not bit-identical"
    r.close()
