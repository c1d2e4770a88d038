"""SecureChannel: payload AEAD + per-frame PN bits (mirror of rtwm/crypto.py:12-48).

Key schedule (rtwm/crypto.py:19-30): okm = HKDF-SHA256(master, salt=None, info="EchoSeal:KDF:v1",
64 bytes); okm[:32] keys ChaCha20-Poly1305, okm[32:] seeds the AES PN stream.
"""
from __future__ import annotations

import secrets

import numpy as np

from .primitives import chacha20poly1305_decrypt, chacha20poly1305_encrypt, hkdf_sha256
from .utils import StreamPRNG, pn_bits as _pn_bits


class _Aead:
    """Just enough of cryptography's ChaCha20Poly1305 object for the detector's fallback path."""

    def __init__(self, key: bytes) -> None:
        self._key = key

    def encrypt(self, nonce: bytes, data: bytes, aad: bytes | None) -> bytes:
        return chacha20poly1305_encrypt(self._key, nonce, data, aad or b"")

    def decrypt(self, nonce: bytes, data: bytes, aad: bytes | None) -> bytes:
        return chacha20poly1305_decrypt(self._key, nonce, data, aad or b"")


class SecureChannel:
    def __init__(self, master_key: bytes) -> None:
        if len(master_key) != 32:
            raise ValueError("master_key must be 32 bytes (256 bit)")
        okm = hkdf_sha256(master_key, 64, salt=None, info=b"EchoSeal:KDF:v1")
        self._aead = _Aead(okm[:32])
        self._prng = StreamPRNG(okm[32:])

    def seal(self, plaintext: bytes, *, nonce: bytes | None = None) -> bytes:
        nonce = secrets.token_bytes(12) if nonce is None else nonce
        return nonce + self._aead.encrypt(nonce, plaintext, b"")

    def open(self, blob: bytes) -> bytes:
        if len(blob) < 12 + 16:
            raise ValueError("ciphertext too short")
        return self._aead.decrypt(blob[:12], blob[12:], b"")

    def pn_bits(self, frame_ctr: int, n_bits: int) -> np.ndarray:
        return _pn_bits(self._prng, frame_ctr, n_bits)

    def pn_bytes_batch(self, frame_ctrs, n_bytes: int = 152) -> np.ndarray:
        """Packed PN schedule rows (MSB-first bits) for many counters: uint8 [B, n_bytes]."""
        return self._prng.blocks(frame_ctrs, (n_bytes + 15) // 16)[:, :n_bytes]
