"""CPU: the schedule oracle (oracle/c/eso_sched.c: AES-128 PN rows, HMAC-SHA256 band hop) against FIPS 197 C.1,
RFC 4231, the reference's known answers (captured from the reference) and the host code that
produces the broadcast schedule."""
import hashlib
import numpy as np
import pytest

from echoseal_amd.crypto import SecureChannel
from echoseal_amd.utils import band_index

KEY = b"\xAA" * 32


@pytest.fixture(scope="module")
def oracle():
    import oracle.oracle as o
    o.build()
    return o


def test_aes_and_hmac_published_vectors(oracle):
    assert oracle.aes128_encrypt(bytes(range(16)), bytes.fromhex("00112233445566778899aabbccddeeff")).hex() == \
        "69c4e0d86a7b0430d8cdb78070b4c55a"                                            # FIPS 197 C.1
    assert oracle.hmac_sha256_short(b"Jefe", b"what do ya want for nothing?").hex() == \
        "5bdcc146bf60754e6a042426089575c75a003f089d2739839dec58b964ec3843"            # RFC 4231 test case 2


def test_schedule_known_answers_of_the_reference(oracle):
    sec = SecureChannel(KEY)
    pn, band = oracle.schedule_rows(sec._prng.sub_key, KEY, list(range(16)))
    assert band.tolist() == [1, 3, 0, 2, 2, 1, 0, 3, 0, 2, 0, 1, 3, 0, 1, 3]
    assert pn[0, :16].tobytes().hex() == "413e2a551d4759db038e35ff269471a9"           # pn_bits(0, 128) packed
    assert pn[5, :4].tobytes().hex() == "da07eb92"                                     # pn_bits(5, 32) packed
    bits = np.unpackbits(pn[5])[:1215]
    assert hashlib.sha256(bits.astype(np.uint8).tobytes()).hexdigest()[:32] == "d5c2ebf78b9cdd5e426495a606f765e4"
    pn0, band0 = oracle.schedule_rows(SecureChannel(b"\x00" * 32)._prng.sub_key, b"\x00" * 32, list(range(16)))
    assert band0.tolist() == [2, 3, 1, 1, 3, 2, 1, 2, 1, 3, 1, 2, 1, 0, 2, 2]


def test_schedule_equals_the_broadcast_schedule(oracle):
    sec = SecureChannel(KEY)
    ctrs = [0, 1, 2, 255, 1024, 65535, 2 ** 31 + 5, 2 ** 32 - 1] + list(np.random.default_rng(3).integers(0, 2 ** 32, 40))
    pn, band = oracle.schedule_rows(sec._prng.sub_key, KEY, ctrs)
    assert np.array_equal(pn, sec.pn_bytes_batch(ctrs, 152))
    assert band.tolist() == [band_index(KEY, int(c)) for c in ctrs]
