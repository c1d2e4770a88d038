"""Load the read-only reference (/root/reference/rtwm) in THIS process so it can be run as the
oracle-of-oracles and golden vectors can be captured from it.  Build-container only: the
reference never travels to the GPU box, and nothing here is imported by the product.

The reference imports `cryptography` (HKDF, ChaCha20Poly1305, AES-ECB, InvalidTag), which is not
installed.  Those names are bound to adapters over echoseal_amd.primitives (RFC 8439 / RFC 5869 /
FIPS-197 KATs: tests/test_primitives.py), injected into sys.modules before the import.
"""
from __future__ import annotations

import sys
import types

REF_ROOT = "/root/reference"


def _install_crypto_adapters() -> None:
    from echoseal_amd import primitives as P
    import numpy as np

    def mod(name: str) -> types.ModuleType:
        m = types.ModuleType(name)
        m.__path__ = []            # behave like a package
        sys.modules[name] = m
        return m

    root = mod("cryptography")
    exc = mod("cryptography.exceptions"); exc.InvalidTag = P.InvalidTag
    haz = mod("cryptography.hazmat"); prim = mod("cryptography.hazmat.primitives")
    ciphers = mod("cryptography.hazmat.primitives.ciphers")
    aead = mod("cryptography.hazmat.primitives.ciphers.aead")
    kdf = mod("cryptography.hazmat.primitives.kdf"); hk = mod("cryptography.hazmat.primitives.kdf.hkdf")
    hashes = mod("cryptography.hazmat.primitives.hashes")
    backends = mod("cryptography.hazmat.backends")
    root.exceptions, root.hazmat = exc, haz
    haz.primitives, haz.backends = prim, backends
    prim.ciphers, prim.kdf, prim.hashes = ciphers, kdf, hashes
    ciphers.aead, kdf.hkdf = aead, hk

    class ChaCha20Poly1305:
        def __init__(self, key): self._k = bytes(key)
        def encrypt(self, nonce, data, aad): return P.chacha20poly1305_encrypt(self._k, bytes(nonce), bytes(data), aad or b"")
        def decrypt(self, nonce, data, aad): return P.chacha20poly1305_decrypt(self._k, bytes(nonce), bytes(data), aad or b"")
    aead.ChaCha20Poly1305 = ChaCha20Poly1305

    class SHA256: pass
    hashes.SHA256 = SHA256

    class HKDF:
        def __init__(self, algorithm, length, salt, info, backend=None):
            self._len, self._salt, self._info = length, salt, info
        def derive(self, ikm): return P.hkdf_sha256(bytes(ikm), self._len, salt=self._salt, info=self._info or b"")
    hk.HKDF = HKDF

    class _AES:
        def __init__(self, key): self.key = bytes(key)
    class _ECB: pass
    class _Enc:
        def __init__(self, key): self._k = key
        def update(self, data):
            return P.aes128_encrypt_blocks(self._k, np.frombuffer(bytes(data), np.uint8).reshape(-1, 16)).tobytes()
        def finalize(self): return b""
    class Cipher:
        def __init__(self, alg, mode, backend=None): self._k = alg.key
        def encryptor(self): return _Enc(self._k)
    algorithms = types.SimpleNamespace(AES=_AES); modes = types.SimpleNamespace(ECB=_ECB)
    ciphers.Cipher, ciphers.algorithms, ciphers.modes = Cipher, algorithms, modes
    backends.default_backend = lambda: None


def load_reference():
    """Return the reference's `rtwm` package (detector, embedder, polar_fast, ...)."""
    if "rtwm" in sys.modules and not getattr(sys.modules["rtwm"], "__file__", "").startswith(REF_ROOT):
        raise RuntimeError("a different `rtwm` is already imported in this process")
    _install_crypto_adapters()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import rtwm  # noqa: F401  (the reference)
    import rtwm.detector, rtwm.embedder, rtwm.polar_fast, rtwm.fastpolar, rtwm.utils, rtwm.crypto  # noqa
    assert rtwm.__file__.startswith(REF_ROOT)
    return rtwm
