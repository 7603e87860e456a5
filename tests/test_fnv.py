"""pyhash.fnv1_32 replacement (host helper): restatement vs the reference's own C core
(oracle/_ref, built from pyhash-0.9.3/src/fnv/hash_32.c), vs golden vectors of that build, and the product
(mdt_fnv1_32 in libmdt_hip.so through mdt_policy_amd.utils.pyhash_compat) vs both.  Bit exact."""
import json
import os

import numpy as np
import pytest

from oracle import fnv_oracle as FO
from tests.helpers import GOLDEN

VEC = json.load(open(os.path.join(GOLDEN, "g10_fnv.json")))["vectors"]


def test_oracle_matches_golden_vectors_of_the_reference_build():
    for v in VEC:
        assert FO.pyhash_fnv1_32(v["s"], seed=v["seed"]) == v["h"], v


@pytest.mark.skipif(not FO.reference_available(), reason="oracle/_ref not built (needs /root/reference: make -C oracle)")
def test_oracle_matches_the_reference_core_on_random_buffers():
    rng = np.random.default_rng(0)
    for n in [0, 1, 2, 3, 7, 64, 1000, 65537]:
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        for seed in (0, 1, 0x811C9DC5, 0xFFFFFFFF):
            assert FO.fnv1_32_bytes(data, seed) == FO.reference_fnv_32_buf(data, seed)


def test_product_matches_golden_and_oracle():
    from mdt_policy_amd.utils.pyhash_compat import fnv1_32, get_validation_window_size, hasher
    for v in VEC:
        assert fnv1_32(seed=v["seed"])(v["s"]) == v["h"], v
        assert hasher(v["s"], seed=v["seed"]) == v["h"]
    rng = np.random.default_rng(1)
    for n in [0, 1, 5, 4096]:
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert hasher(data) == FO.fnv1_32_bytes(data, 0)
    # several arguments thread the running value (Hash.h:169-173); seed attribute like pyhash's def_readwrite
    assert hasher("12", "34") == FO.pyhash_fnv1_32("12", "34") == FO.pyhash_fnv1_32("34", seed=FO.pyhash_fnv1_32("12"))
    h = fnv1_32(7)
    assert h.seed == 7 and h("x") == FO.pyhash_fnv1_32("x", seed=7)
    with pytest.raises(TypeError):
        hasher(12)
    # the call site: window sizes stay inside [min, max] and are a pure function of idx
    sizes = [get_validation_window_size(i, 20, 32) for i in range(200)]
    assert min(sizes) >= 20 and max(sizes) <= 32 and len(set(sizes)) > 5
    assert sizes == [20 + FO.pyhash_fnv1_32(str(i)) % 13 for i in range(200)]
