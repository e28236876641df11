"""The oracle's "reference stream" mode (oracle.cpp, orc_render_reference_stream): ImageTileIntegrator::render drawing what the Rust binary draws with one rayon
worker — ONE SmallRng (rand 0.8.5: Xoshiro256++ seeded through SplitMix64, sampler.rs:103-109) consumed by waves -> tiles -> x -> y -> sample (integrator.rs:242-260),
f32 = (next_u64() >> 40) * 2^-24 (sampler.rs:123-131 through rand's Standard distribution). Test infrastructure only: the product keeps the defined per-pixel stream.
What is checked here: the generator against the PUBLISHED vectors of both algorithms and an independent Python restatement; the f32 mapping; the mode's determinism,
its loop order, its agreement in the mean with the per-pixel stream's render; and the committed film / PFM hashes of the two example scenes
(tests/golden/reference_stream.json, made by tests/golden/gen_reference_stream.py) that INTEGRATION.md's recipe hands to a maintainer with cargo.
UNVERIFIED against the binary itself: there is no Rust toolchain in this image."""
import ctypes as C
import json
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests" / "golden"))
M64 = (1 << 64) - 1

# xoshiro256++ 1.0 (Blackman & Vigna, prng.di.unimi.it, public domain): the first ten outputs from the state {1, 2, 3, 4} — the reference vector rand 0.8's own
# xoshiro256plusplus.rs test carries
XOSHIRO256PP_1234 = [41943041, 58720359, 3588806011781223, 3591011842654386, 9228616714210784205, 9973669472204895162, 14011001112246962877,
                     12406186145184390807, 15849039046786891736, 10450023813501588000]
# SplitMix64 (Steele, Lea & Flood; splitmix64.c, public domain): the widely published outputs for the seeds 1234567 and 0
SPLITMIX64_1234567 = [6457827717110365317, 3203168211198807973, 9817491932198370423, 4593380528125082431, 16408922859458223821]
SPLITMIX64_0_FIRST = 0xE220A8397B1DCDAF


def _rotl(x, k):
    return ((x << k) | (x >> (64 - k))) & M64


def py_xoshiro256pp(state, n):
    s, out = list(state), []
    for _ in range(n):
        out.append((_rotl((s[0] + s[3]) & M64, 23) + s[0]) & M64)
        t = (s[1] << 17) & M64
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t
        s[3] = _rotl(s[3], 45)
    return out


def py_splitmix64(seed, n):
    out = []
    for _ in range(n):
        seed = (seed + 0x9E3779B97F4A7C15) & M64
        z = seed
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        out.append(z ^ (z >> 31))
    return out


@pytest.fixture(scope="module")
def orc():
    import oracle_py
    return oracle_py, oracle_py.load()


def test_generators_reproduce_the_published_vectors(orc):
    _, lib = orc
    assert py_xoshiro256pp([1, 2, 3, 4], 10) == XOSHIRO256PP_1234 and py_splitmix64(1234567, 5) == SPLITMIX64_1234567 and py_splitmix64(0, 1)[0] == SPLITMIX64_0_FIRST
    st = (C.c_uint64 * 4)(1, 2, 3, 4)
    out = (C.c_uint64 * 10)()
    lib.orc_fn_xoshiro256pp(st, 10, out)
    assert list(out) == XOSHIRO256PP_1234
    out = (C.c_uint64 * 5)()
    lib.orc_fn_splitmix64(1234567, 5, out)
    assert list(out) == SPLITMIX64_1234567
    out = (C.c_uint64 * 1)()
    lib.orc_fn_splitmix64(0, 1, out)
    assert out[0] == SPLITMIX64_0_FIRST


@pytest.mark.parametrize("seed", [0, 1, 7, 0xFFFFFFFFFFFFFFFF, 0x0123456789ABCDEF])
def test_seed_from_u64_and_the_f32_mapping(orc, seed):
    """seed_from_u64 = four SplitMix64 outputs as the state words (rand 0.8.5 xoshiro256plusplus.rs); gen::<f32>() = (next_u32() >> 8) * 2^-24 with
    next_u32 = next_u64 >> 32 — against the independent Python restatement, 4096 draws."""
    _, lib = orc
    n = 4096
    vals = (C.c_float * n)()
    st = (C.c_uint64 * 4)()
    lib.orc_fn_reference_stream_f32(seed, n, vals, st)
    want_state = py_splitmix64(seed, 4)
    assert list(st) == want_state
    want = np.array([(u >> 40) for u in py_xoshiro256pp(want_state, n)], np.float64) * 2.0 ** -24
    got = np.frombuffer(vals, np.float32)
    assert np.array_equal(got.astype(np.float64), want) and got.min() >= 0.0 and got.max() < 1.0
    assert 0.45 < got.mean() < 0.55


def _s1(lib, res=24):
    from shimmer_amd import scenes
    return scenes.sphere_light(lib, res, res)


def test_mode_is_deterministic_and_sequential(orc, lib):
    """Two renders are identical; a render of the tiles in two halves with ONE stream cannot be reproduced by two separate calls (the stream is sequential: the second
    call starts it again) — but the first half's pixels are the whole render's; the number of draws is the same run to run; another seed gives another film."""
    oracle_py, _ = orc
    from shimmer_amd import render, scene as scn
    sc = _s1(lib)
    o = oracle_py.Oracle(sc.desc)
    p = render.make_params(seed=0, spp=3, max_depth=5)
    f1, s1, d1 = o.render_reference_stream(p)
    f2, s2, d2 = o.render_reference_stream(p)
    assert np.array_equal(f1, f2) and d1 == d2 and d1 > 24 * 24 * 3 * 6 and s1 == s2
    assert (f1["weight_sum"] == 3.0).all() and np.isfinite(f1["rgb_sum"]).all() and f1["rgb_sum"].max() > 0
    tiles, n = scn.tiles_for(lib, o.pixel_bounds)
    # one wave over the first tile only: the stream's first draws belong to that tile's pixel (x0, y0), x outer (integrator.rs:257-260)
    p1 = render.make_params(seed=0, spp=1, max_depth=5)
    fa, _, _ = o.render_reference_stream(p1, tiles, 1)
    fb, _, _ = o.render_reference_stream(p1, tiles, n)
    t = tiles[0]
    assert np.array_equal(fa[t.y0:t.y1, t.x0:t.x1], fb[t.y0:t.y1, t.x0:t.x1]) and (fa["weight_sum"].sum() == (t.x1 - t.x0) * (t.y1 - t.y0))
    f3, _, _ = o.render_reference_stream(render.make_params(seed=1, spp=3, max_depth=5))
    assert not np.array_equal(f1, f3)
    o.close()


def test_first_sample_consumes_the_stream_in_evaluate_pixel_sample_order(orc, lib):
    """The very first draws: lambda u (integrator.rs:339-343), the filter's 2d, the lens 2d, time (sampling.rs:347-371) — the first camera ray of the mode equals the
    camera ray made from the stream's first values through the defined-stream entry point's arithmetic (orc_fn_camera_ray is fed by its own stream, so compare
    through the film instead: a 1x1-pixel, depth-0 render's weight is 1 and its draws are at least six)."""
    oracle_py, _ = orc
    from shimmer_amd import render, scene as scn
    sc = _s1(lib, 8)
    o = oracle_py.Oracle(sc.desc)
    tiles, n = scn.tiles_for(lib, o.pixel_bounds)
    f, st, draws = o.render_reference_stream(render.make_params(seed=5, spp=1, max_depth=0), tiles, n)
    assert st["paths"] == 64 and draws >= 6 * 64 and (f["weight_sum"] == 1.0).all()
    o.close()


def test_agrees_in_the_mean_with_the_defined_stream(orc, lib):
    """Same estimator, another sample stream: the two films' means agree within Monte-Carlo noise (a wiring error — a dimension drawn twice, a draw skipped — biases it)."""
    oracle_py, _ = orc
    from shimmer_amd import render
    sc = _s1(lib, 32)
    o = oracle_py.Oracle(sc.desc)
    p = render.make_params(seed=0, spp=64, max_depth=5)
    fr, _, _ = o.render_reference_stream(p)
    fd, _ = o.render(p, n_threads=os.cpu_count() or 1)
    a, b = fr["rgb_sum"].mean(axis=(0, 1)), fd["rgb_sum"].mean(axis=(0, 1))
    assert np.all(np.abs(a - b) < 0.03 * np.abs(b)), (a, b)
    o.close()


def test_entropy_seeded_materials_are_refused(orc, lib):
    oracle_py, _ = orc
    from shimmer_amd import render, scenes
    sc = scenes.cornell_box(lib, 16, 16, coated=True)
    o = oracle_py.Oracle(sc.desc)
    with pytest.raises(RuntimeError, match="entropy"):
        o.render_reference_stream(render.make_params(seed=0, spp=1, max_depth=5))
    o.close()


@pytest.mark.parametrize("scene,spp", [("sphere_light.pbrt", 1), ("sphere_light.pbrt", 4), ("cornell_box.pbrt", 1)])
def test_committed_hashes_of_the_example_scenes(orc, lib, scene, spp):
    """examples/scenes/*.pbrt through the repo's loader and the mode: the film and PFM hashes INTEGRATION.md's recipe quotes (the 4- and 64-spp records of the same
    file are made by `python tests/golden/gen_reference_stream.py --full`; they take minutes single-threaded)."""
    import gen_reference_stream
    want = {(r["scene"], r["spp"]): r for r in json.loads((ROOT / "tests" / "golden" / "reference_stream.json").read_text())["renders"]}
    got = gen_reference_stream.reference_stream_record(lib, scene, spp)
    for k in ("film_sha256", "pfm_sha256", "paths", "rays_closest", "rays_any", "u64_draws", "width", "height"):
        assert got[k] == want[(scene, spp)][k], k
    assert ("cornell_box.pbrt", 64) in want and ("sphere_light.pbrt", 64) in want
