"""CPU: the AEAD-validator oracle (oracle/c/eso_aead.c) against RFC 8439's own vectors and
against the host primitives the reference's call sites run on (echoseal_amd.primitives, themselves RFC-pinned in
tests/test_primitives.py)."""
import numpy as np
import pytest

from echoseal_amd.crypto import SecureChannel
from echoseal_amd.primitives import chacha20poly1305_encrypt

KEY = b"\xAA" * 32


@pytest.fixture(scope="module")
def oracle():
    import oracle.oracle as o
    o.build()
    return o


def test_chacha20_block_rfc8439_2_3_2(oracle):
    key = bytes(range(32))
    nonce = bytes.fromhex("000000090000004a00000000")
    ks = oracle.chacha20_block(key, 1, nonce)
    assert ks.hex() == ("10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e"
                        "d2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e")


def test_poly1305_rfc8439_2_5_2(oracle):
    otk = bytes.fromhex("85d6be7857556d337f4452fe42d506a80103808afb0db2fd4abff6af4149f51b")
    msg = b"Cryptographic Forum Research Group"
    assert oracle.poly1305(otk, msg).hex() == "a8061dc1305136c6c22b8baf0c0127a9"


def test_aead_rfc8439_2_8_2(oracle):
    key = bytes(range(0x80, 0xa0))
    nonce = bytes.fromhex("070000004041424344454647")
    aad = bytes.fromhex("50515253c0c1c2c3c4c5c6c7")
    pt = (b"Ladies and Gentlemen of the class of '99: If I could offer you only one tip for the future, "
          b"sunscreen would be it.")
    sealed = chacha20poly1305_encrypt(key, nonce, pt, aad)
    assert sealed[-16:].hex() == "1ae10b594f09e26a7e902ecbd0600691"       # the RFC's tag
    assert oracle.aead_open(key, nonce, aad, sealed) == pt
    bad = bytearray(sealed); bad[5] ^= 1
    assert oracle.aead_open(key, nonce, aad, bytes(bad)) is None
    assert oracle.aead_open(key, nonce, aad + b"x", sealed) is None


def test_validator_matches_the_detector_closure(oracle):
    """Blobs sealed the way the embedder does (SecureChannel.seal, 27-byte ESAL plaintext) and mutations of them:
    oracle verdict == the Python closure's verdict (rtwm/detector.py:168-176 restated in detector._validator)."""
    sec = SecureChannel(KEY)
    rng = np.random.default_rng(5)
    blobs, ctrs, want = [], [], []
    for i in range(200):
        ctr = int(rng.integers(0, 2 ** 32))
        magic = b"ESAL" if i % 7 else b"ESAX"
        pt = magic + ctr.to_bytes(4, "big") + rng.bytes(19)
        blob = bytearray(sec.seal(pt, nonce=rng.bytes(12)))
        expect_ctr = ctr if i % 5 else (ctr + 1) % 2 ** 32
        if i % 3 == 0:
            blob[int(rng.integers(0, 55))] ^= 1 << int(rng.integers(0, 8))
        blobs.append(bytes(blob)); ctrs.append(expect_ctr)
        try:
            p = sec.open(bytes(blob)); w = p.startswith(b"ESAL") and int.from_bytes(p[4:8], "big") == expect_ctr
        except Exception:
            w = False
        want.append(w)
    ok, plain = oracle.validate_blobs(sec._aead._key, np.frombuffer(b"".join(blobs), np.uint8).reshape(-1, 55), ctrs)
    assert ok.astype(bool).tolist() == want and 0 < sum(want) < len(want)
    for b, o, p in zip(blobs, ok, plain):
        try:
            assert bytes(p) == sec.open(b)
        except Exception:
            assert not o and not p.any()
