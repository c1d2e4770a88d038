"""Self-contained keyed primitives for the EchoSeal key / PN schedule.

The reference leans on `cryptography` / PyCryptodome for these (rtwm/crypto.py:6-10,
rtwm/utils.py:72-110); neither is installed on the build or GPU boxes, and the schedule is host-side
input preparation, so the primitives are restated here on top of the
standard library (`hashlib`, `hmac`) and NumPy:

* AES-128 single-block encryption (FIPS-197), vectorised over a batch of blocks,
* ChaCha20 (RFC 8439 section 2.3/2.4), vectorised over a batch of (key, nonce) messages,
* Poly1305 + the ChaCha20-Poly1305 AEAD construction (RFC 8439 section 2.5/2.8),
* HKDF-SHA256 (RFC 5869).

Known-answer tests from the RFCs / FIPS-197 live in tests/test_primitives.py.
"""
from __future__ import annotations

import hashlib
import hmac
import struct

import numpy as np

__all__ = [
    "aes128_encrypt_blocks", "chacha20_xor", "poly1305_tag", "chacha20poly1305_encrypt",
    "chacha20poly1305_decrypt", "hkdf_sha256", "InvalidTag",
]


class InvalidTag(Exception):
    """AEAD authentication failure (stands where cryptography.exceptions.InvalidTag would)."""


# ----------------------------------------------------------------------------- AES-128
def _build_sbox() -> np.ndarray:
    # multiplicative inverse in GF(2^8) followed by the affine map (FIPS-197 section 5.1.1)
    exp = [0] * 510
    log = [0] * 256
    x = 1
    for i in range(255):
        exp[i] = x
        log[x] = i
        x ^= (x << 1) ^ (0x11B if x & 0x80 else 0)   # multiply by generator 3
        x &= 0xFF
    for i in range(255, 510):
        exp[i] = exp[i - 255]
    sbox = np.zeros(256, dtype=np.uint8)
    for v in range(256):
        inv = 0 if v == 0 else exp[255 - log[v]]
        r = inv
        for s in (1, 2, 3, 4):
            r ^= ((inv << s) | (inv >> (8 - s))) & 0xFF
        sbox[v] = r ^ 0x63
    return sbox


_SBOX = _build_sbox()
_XT = np.array([((v << 1) ^ (0x1B if v & 0x80 else 0)) & 0xFF for v in range(256)], dtype=np.uint8)
# ShiftRows on a column-major 16-byte state: new[r + 4c] = old[r + 4((c + r) % 4)]
_SHIFT = np.array([(r + 4 * ((c + r) % 4)) for c in range(4) for r in range(4)], dtype=np.intp)


def _aes128_round_keys(key: bytes) -> np.ndarray:
    if len(key) != 16:
        raise ValueError("AES-128 key must be 16 bytes")
    w = [list(key[4 * i:4 * i + 4]) for i in range(4)]
    rcon = 1
    for i in range(4, 44):
        t = list(w[i - 1])
        if i % 4 == 0:
            t = t[1:] + t[:1]
            t = [int(_SBOX[b]) for b in t]
            t[0] ^= rcon
            rcon = int(_XT[rcon])
        w.append([a ^ b for a, b in zip(w[i - 4], t)])
    return np.array(w, dtype=np.uint8).reshape(11, 16)


def aes128_encrypt_blocks(key: bytes, blocks: np.ndarray) -> np.ndarray:
    """Encrypt `blocks` (uint8 array [..., 16]) under a 16-byte key, ECB, one block per row."""
    rk = _aes128_round_keys(key)
    s = np.array(blocks, dtype=np.uint8, copy=True).reshape(-1, 16)
    s ^= rk[0]
    for rnd in range(1, 11):
        s = _SBOX[s][:, _SHIFT]
        if rnd != 10:
            c = s.reshape(-1, 4, 4)                       # [block, column, row]
            rot1 = np.roll(c, -1, axis=2)
            t = c ^ rot1
            allx = c[:, :, 0:1] ^ c[:, :, 1:2] ^ c[:, :, 2:3] ^ c[:, :, 3:4]
            s = (c ^ allx ^ _XT[t]).reshape(-1, 16)
        s = s ^ rk[rnd]
    return s.reshape(np.shape(blocks))


# ----------------------------------------------------------------------------- ChaCha20
_SIGMA = np.frombuffer(b"expand 32-byte k", dtype="<u4")


def _rotl(x: np.ndarray, n: int) -> np.ndarray:
    return (x << np.uint32(n)) | (x >> np.uint32(32 - n))


def _chacha_blocks(keys: np.ndarray, counters: np.ndarray, nonces: np.ndarray) -> np.ndarray:
    """keys [B,8] u32, counters [B] u32, nonces [B,3] u32 -> keystream [B,64] u8."""
    B = keys.shape[0]
    st = np.empty((B, 16), dtype=np.uint32)
    st[:, 0:4] = _SIGMA
    st[:, 4:12] = keys
    st[:, 12] = counters
    st[:, 13:16] = nonces
    x = st.copy()

    def qr(a, b, c, d):
        x[:, a] += x[:, b]; x[:, d] = _rotl(x[:, d] ^ x[:, a], 16)
        x[:, c] += x[:, d]; x[:, b] = _rotl(x[:, b] ^ x[:, c], 12)
        x[:, a] += x[:, b]; x[:, d] = _rotl(x[:, d] ^ x[:, a], 8)
        x[:, c] += x[:, d]; x[:, b] = _rotl(x[:, b] ^ x[:, c], 7)

    with np.errstate(over="ignore"):
        for _ in range(10):
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
        x += st
    return x.astype("<u4").view(np.uint8).reshape(B, 64)


def chacha20_xor(key: bytes, nonces: np.ndarray, data: np.ndarray, counter0: int = 1) -> np.ndarray:
    """XOR `data` [B, n] with the ChaCha20 keystream of (key, nonces[b]) starting at block counter0."""
    data = np.atleast_2d(np.asarray(data, dtype=np.uint8))
    nonces = np.atleast_2d(np.asarray(nonces, dtype=np.uint8))
    B, n = data.shape
    k = np.broadcast_to(np.frombuffer(key, dtype="<u4"), (B, 8))
    nn = np.ascontiguousarray(nonces).view("<u4").reshape(B, 3)
    out = np.empty_like(data)
    for j in range((n + 63) // 64):
        ks = _chacha_blocks(k, np.full(B, counter0 + j, dtype=np.uint32), nn)
        seg = slice(64 * j, min(n, 64 * (j + 1)))
        out[:, seg] = data[:, seg] ^ ks[:, : seg.stop - seg.start]
    return out


_P1305 = (1 << 130) - 5


def poly1305_tag(otk: bytes, msg: bytes) -> bytes:
    r = int.from_bytes(otk[:16], "little") & 0x0FFFFFFC0FFFFFFC0FFFFFFC0FFFFFFF
    s = int.from_bytes(otk[16:32], "little")
    acc = 0
    for i in range(0, len(msg), 16):
        blk = msg[i:i + 16]
        acc = ((acc + int.from_bytes(blk, "little") + (1 << (8 * len(blk)))) * r) % _P1305
    return ((acc + s) & ((1 << 128) - 1)).to_bytes(16, "little")


def _pad16(b: bytes) -> bytes:
    return b"\x00" * (-len(b) % 16)


def _aead_tag(key: bytes, nonce: bytes, aad: bytes, ct: bytes) -> bytes:
    ks0 = _chacha_blocks(np.frombuffer(key, dtype="<u4").reshape(1, 8), np.zeros(1, np.uint32),
                         np.frombuffer(nonce, dtype="<u4").reshape(1, 3))[0]
    mac_data = aad + _pad16(aad) + ct + _pad16(ct) + struct.pack("<QQ", len(aad), len(ct))
    return poly1305_tag(ks0[:32].tobytes(), mac_data)


def chacha20poly1305_encrypt(key: bytes, nonce: bytes, plaintext: bytes, aad: bytes = b"") -> bytes:
    """RFC 8439 AEAD: returns ciphertext || 16-byte tag."""
    if len(key) != 32 or len(nonce) != 12:
        raise ValueError("ChaCha20-Poly1305 needs a 32-byte key and a 12-byte nonce")
    ct = chacha20_xor(key, np.frombuffer(nonce, np.uint8), np.frombuffer(plaintext, np.uint8))[0].tobytes() \
        if plaintext else b""
    return ct + _aead_tag(key, nonce, aad, ct)


def chacha20poly1305_decrypt(key: bytes, nonce: bytes, data: bytes, aad: bytes = b"") -> bytes:
    if len(key) != 32 or len(nonce) != 12:
        raise ValueError("ChaCha20-Poly1305 needs a 32-byte key and a 12-byte nonce")
    if len(data) < 16:
        raise InvalidTag("ciphertext shorter than the tag")
    ct, tag = data[:-16], data[-16:]
    if not hmac.compare_digest(_aead_tag(key, nonce, aad, ct), tag):
        raise InvalidTag("authentication failed")
    if not ct:
        return b""
    return chacha20_xor(key, np.frombuffer(nonce, np.uint8), np.frombuffer(ct, np.uint8))[0].tobytes()


# ----------------------------------------------------------------------------- HKDF
def hkdf_sha256(ikm: bytes, length: int, *, salt: bytes | None = None, info: bytes = b"") -> bytes:
    prk = hmac.new(salt if salt else b"\x00" * 32, ikm, hashlib.sha256).digest()
    okm, t, i = b"", b"", 1
    while len(okm) < length:
        t = hmac.new(prk, t + info + bytes([i]), hashlib.sha256).digest()
        okm += t
        i += 1
    return okm[:length]
