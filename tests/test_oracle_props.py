"""Size-independent properties of the oracle itself (CPU): tiling / thread invariance under the
RNG contract, seed sensitivity, sample-count bookkeeping."""
import numpy as np

from conftest import open_scene


def test_tiles_and_threads_do_not_change_the_image(fray, abi, oracle):
    s = open_scene(fray, "cornell_box.fray", 100, 75, numPaths=3)      # ragged: not a multiple of 48
    full, st = oracle.render(s.desc, abi.MODE_RENDER, threads=1)
    again, _ = oracle.render(s.desc, abi.MODE_RENDER, threads=8)
    assert np.array_equal(full, again)
    acc = np.zeros_like(full)
    rays = 0
    for r in range(3):
        part, pst = oracle.render(s.desc, abi.MODE_RENDER, bucket_first=r, bucket_stride=3, threads=4)
        acc += part
        rays += pst["closest_rays"]
    assert np.array_equal(acc, full) and rays == st["closest_rays"]
    assert st["samples"] == 100 * 75 * 3
    other, _ = oracle.render(s.desc, abi.MODE_RENDER, seed=43)
    assert not np.array_equal(other, full)
    s.close()


def test_whitted_sample_counts(fray, abi, oracle):
    s = open_scene(fray, "boxed.fray", 64, 48, wantAA=1)
    _, st = oracle.render(s.desc, abi.MODE_RENDER)
    assert st["samples"] == 64 * 48 * 5 and st["closest_rays"] == st["samples"]
    # every shaded hit fires 2 lights x 16 samples of shadow rays; camera rays that land on a
    # RectLight return its colour without shading (main.cpp:273-275)
    assert st["shadow_rays"] % 32 == 0 and 0.9 * 32 * st["closest_rays"] < st["shadow_rays"] <= 32 * st["closest_rays"]
    s.close()


def test_seed_function_is_the_documented_one(oracle):
    def fmix(h):
        h ^= h >> 16; h = (h * 0x85ebca6b) & 0xffffffff; h ^= h >> 13; h = (h * 0xc2b2ae35) & 0xffffffff; h ^= h >> 16
        return h
    for seed, p, i in [(42, 0, 0), (42, 12345, 7), (1, 2073599, 63)]:
        h = fmix(seed ^ ((p * 0x9e3779b1) & 0xffffffff))
        assert oracle.lib.fray_oracle_sample_seed(seed, p, i) == fmix(h ^ ((i * 0x85ebca77) & 0xffffffff) ^ 0x27d4eb2f)


def test_mt19937_known_answer(oracle):
    # the C++ standard fixes the 10000th output of a default-seeded (5489) mt19937 to 4123659995
    out = np.zeros(10000, np.uint32)
    oracle.lib.fray_oracle_rng_words(5489, 10000, out.ctypes.data)
    assert int(out[-1]) == 4123659995
