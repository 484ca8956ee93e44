"""CPU tests: the oracle's COUNTER-mode renderer (the random-number SPEC shared with the HIP kernels) against per-pixel
statistics of the REAL reference renderer stored in tests/golden/ref_images.npz (16 batches x 64 spp each, reference's own
mt19937 stream).  Same-seed comparison with the reference is impossible (one global sequential generator), so
this layer is statistical -- tolerances from SURVEY §8(d): image mean within 1 % and <= 0.3 % of pixels beyond 4 sigma
(a little slack is left for the finite number of batches behind the variance estimate)."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def images():
    with np.load(os.path.join(G, "ref_images.npz")) as z:
        return {k: z[k] for k in z.files}


def _batched(o, spp_per_batch, batches, seed):
    means = []
    for b in range(batches):
        acc, cnt, _ = o.render(spp_per_batch, seed=seed, first_sample=b * spp_per_batch)
        means.append(acc[..., :3] / acc[..., 3:])
    m = np.stack(means)
    return m.mean(0), m.var(0, ddof=1) / batches, cnt


def _check(mean_a, var_a, mean_b, var_b, mean_tol=0.01, frac_tol=0.006):
    ga, gb = mean_a.mean((0, 1)), mean_b.mean((0, 1))
    assert np.allclose(ga, gb, rtol=mean_tol), (ga, gb)
    z = np.abs(mean_a - mean_b) / np.sqrt(var_a + var_b + 1e-12)
    frac = float((z > 4).mean())
    assert frac <= frac_tol, frac
    return frac


@pytest.mark.parametrize("name,max_depth,scene_fn,res", [("cs_unbounded", 0, "cornell_box_small", 64), ("cs_depth4", 4, "cornell_box_small", 64),
                                                        ("ob_unbounded", 0, "open_box", 48)])
def test_oracle_counter_mode_matches_reference_statistics(pkg, orc, images, name, max_depth, scene_fn, res):
    scene = getattr(pkg.scenes, scene_fn)(res, res)
    o = orc.Oracle(scene, max_depth=max_depth)
    mean, var, cnt = _batched(o, 64, 8, seed=99)
    frac = _check(mean, var, images[name + "_mean"], images[name + "_var"])
    print(name, "image mean oracle", mean.mean((0, 1)), "reference", images[name + "_mean"].mean((0, 1)), "pixels > 4 sigma: %.3f%%" % (100 * frac))


def test_reference_is_darker_than_the_corrected_estimator(pkg, orc, images):
    """SURVEY A-9 made visible: reproducing the reference's shadow-ray self-occlusion darkens the image substantially; the
    corrected estimator (MCPT_FLAG_CORRECT_SHADOW_T2) does NOT match the reference."""
    scene = pkg.scenes.cornell_box_small(64, 64)
    faithful, _, _ = orc.Oracle(scene).render(128, seed=5)
    corrected, _, c2 = orc.Oracle(scene, flags=pkg.FLAG_CORRECT_SHADOW_T2).render(128, seed=5)
    mf = (faithful[..., :3] / faithful[..., 3:]).mean(); mc = (corrected[..., :3] / corrected[..., 3:]).mean()
    ref = images["cs_unbounded_mean"].mean()
    assert abs(mf - ref) / ref < 0.02
    assert mc > 1.2 * ref


def test_sample_range_split_is_exact(pkg, orc):
    """A sample's random numbers depend only on (seed, pixel, sample index): rendering samples [0,8) in one call or as
    [0,3) + [3,8) gives the same film (identical paths, identical per-pixel summation order) -- the property multi-GPU
    sample sharding relies on."""
    scene = pkg.scenes.open_box(24, 24)
    o = orc.Oracle(scene, max_depth=4)
    a, _, _ = o.render(8, seed=3)
    b, _, _ = o.render(3, seed=3, first_sample=0)
    b, _, _ = o.render(5, seed=3, first_sample=3, accum=b)
    assert np.array_equal(a, b)
    c, _, _ = o.render(8, seed=4)
    assert not np.array_equal(a, c)


def test_counter_rng_is_uniform_and_keyed(orc):
    blocks = np.array([orc.Oracle.rng_block(p, s, b, 7) for p in range(64) for s in range(8) for b in range(4)])
    assert blocks.min() >= 0.0 and blocks.max() < 1.0
    assert abs(blocks.mean() - 0.5) < 0.01 and abs(blocks.var() - 1 / 12) < 0.005
    assert not np.array_equal(orc.Oracle.rng_block(1, 2, 3, 7), orc.Oracle.rng_block(1, 2, 3, 8))
    assert not np.array_equal(orc.Oracle.rng_block(1, 2, 3, 7), orc.Oracle.rng_block(1, 2, 3, 7 + (1 << 32)))
    assert np.array_equal(orc.Oracle.rng_block(1, 2, 3, 7), orc.Oracle.rng_block(1, 2, 3, 7))
