"""-m gpu: a few seeds of the differential fuzz cases (tests/fuzz_cases.py) on the device; tools/emu/fuzz.py runs the same cases by the
thousand on the emulated one."""
import pytest

import fuzz_cases

pytestmark = pytest.mark.gpu


# (1100002: a chain that alternates forward-delete and plain states byte after byte: two words per byte in the scoring walk)
@pytest.mark.parametrize("seed", [3, 1007, 1019, 1100, 1254, 1900, 1100002])
def test_tokenize_count_serialized_score_decode_against_the_oracle(seed):
    fuzz_cases.one(seed)


@pytest.mark.parametrize("first", [5000, 9000, 20000])
def test_device_normalizer_against_the_host_normalizer(first):
    for seed in range(first, first + 12):
        fuzz_cases.one_norm(seed)


@pytest.mark.parametrize("first", [1, 700, 4000])
def test_device_capcode_decode_against_the_host_decoder(first):
    for seed in range(first, first + 15):
        fuzz_cases.one_decode(seed)


@pytest.mark.parametrize("first", [1, 501])
def test_raw_text_to_ids_in_one_device_pass(first):
    for seed in range(first, first + 6):
        fuzz_cases.one_raw(seed)
