"""Philox4x32-10 known-answer tests (Random123 kat_vectors)."""
import pytest

KATS = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


@pytest.mark.parametrize("ctr,key,want", KATS)
def test_oracle_philox_kat(oracle_mod, ctr, key, want):
    assert oracle_mod.philox(ctr, key) == want


@pytest.mark.parametrize("ctr,key,want", KATS)
def test_host_scene_philox_kat(ctr, key, want):
    from rtiow_amd.philox import philox4x32_10
    assert philox4x32_10(ctr, key) == want


def test_uniform_stream_is_24_bit_grid():
    from rtiow_amd.philox import UniformStream
    s = UniformStream(1)
    for _ in range(64):
        u = s.next()
        assert 0.0 <= u < 1.0 and (u * 16777216.0) == int(u * 16777216.0)


@pytest.mark.gpu
@pytest.mark.parametrize("ctr,key,want", KATS)
def test_device_philox_kat(renderer, ctr, key, want):
    assert renderer.philox(ctr, key) == want
