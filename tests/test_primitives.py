"""Known-answer tests for the self-contained keyed primitives and the key/PN schedule."""
import hashlib

import numpy as np
import pytest

from echoseal_amd import primitives as P
from echoseal_amd.crypto import SecureChannel
from echoseal_amd.utils import band_index, mseq_63


def test_aes128_fips197_c1():
    ct = P.aes128_encrypt_blocks(bytes(range(16)), np.frombuffer(bytes.fromhex("00112233445566778899aabbccddeeff"), np.uint8))
    assert ct.tobytes().hex() == "69c4e0d86a7b0430d8cdb78070b4c55a"


def test_chacha20poly1305_rfc8439_2_8_2():
    key = bytes(range(0x80, 0xA0)); nonce = bytes.fromhex("070000004041424344454647"); aad = bytes.fromhex("50515253c0c1c2c3c4c5c6c7")
    pt = b"Ladies and Gentlemen of the class of '99: If I could offer you only one tip for the future, sunscreen would be it."
    out = P.chacha20poly1305_encrypt(key, nonce, pt, aad)
    assert out[:16].hex() == "d31a8d34648e60db7b86afbc53ef7ec2" and out[-16:].hex() == "1ae10b594f09e26a7e902ecbd0600691"
    assert P.chacha20poly1305_decrypt(key, nonce, out, aad) == pt
    bad = bytearray(out); bad[3] ^= 1
    with pytest.raises(P.InvalidTag):
        P.chacha20poly1305_decrypt(key, nonce, bytes(bad), aad)


def test_hkdf_rfc5869():
    assert P.hkdf_sha256(bytes.fromhex("0b" * 22), 42).hex() == \
        "8da4e775a563c18f715f802a063c5a31b8a11f5c5ee1879ec3454e5f3c738d2d9d201395faa4b61a96c8"
    assert P.hkdf_sha256(bytes.fromhex("0b" * 22), 42, salt=bytes(range(13)), info=bytes(range(0xF0, 0xFA))).hex() == \
        "3cb25f25faacd57a90434f64d0362f2a2d2d0a90cf1a5a4c5db02d56ecc4c5bf34007208d5b887185865"


def test_schedule_known_answers(golden_detector):
    """Values captured from the reference (SURVEY.md Appendix A + tests/golden/detector.npz)."""
    k = b"\xAA" * 32
    sc = SecureChannel(k)
    assert np.packbits(sc.pn_bits(0, 128)).tobytes().hex() == "413e2a551d4759db038e35ff269471a9"
    assert np.packbits(sc.pn_bits(5, 32)).tobytes().hex() == "da07eb92"
    assert hashlib.sha256(sc.pn_bits(5, 1215).tobytes()).hexdigest()[:32] == "d5c2ebf78b9cdd5e426495a606f765e4"
    assert [band_index(k, c) for c in range(16)] == [1, 3, 0, 2, 2, 1, 0, 3, 0, 2, 0, 1, 3, 0, 1, 3]
    assert [band_index(b"\0" * 32, c) for c in range(16)] == [2, 3, 1, 1, 3, 2, 1, 2, 1, 3, 1, 2, 1, 0, 2, 2]
    assert "".join(map(str, mseq_63())) == "100000100001100010100111101000111001001011011101100110101011111"
    g = golden_detector
    for i in range(int(g["det/count"])):
        key = g[f"det/{i:02d}/key"].tobytes(); ctr = int(g[f"det/{i:02d}/ctr"])
        ch = SecureChannel(key)
        assert np.array_equal(np.packbits(ch.pn_bits(ctr, 1215)), g[f"det/{i:02d}/pn"])
        assert np.array_equal(np.unpackbits(ch.pn_bytes_batch([ctr], 152)[0])[:1215], ch.pn_bits(ctr, 1215))
        assert band_index(key, ctr) == int(g[f"det/{i:02d}/band"])
    assert np.array_equal(np.packbits(SecureChannel(k).pn_bits(0, 128)), g["static/AA/hdr_pn"])


def test_seal_open_roundtrip_and_errors():
    sc = SecureChannel(b"\x01" * 32)
    blob = sc.seal(b"ESAL" + bytes(23))
    assert len(blob) == 55 and sc.open(blob) == b"ESAL" + bytes(23)
    with pytest.raises(ValueError):
        sc.open(b"short")
    with pytest.raises(ValueError):
        SecureChannel(b"x" * 31)
    tam = bytearray(blob); tam[20] ^= 0xFF
    with pytest.raises(Exception):
        sc.open(bytes(tam))
