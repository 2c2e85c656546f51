"""ctypes bindings for the native crypto library (csrc/host/crypto.cpp).

Provides the primitives the secure aggregators and the update signer need:
``X25519Key`` (key agreement for pair seeds), ``RSAKey`` (generate / public PEM / OAEP-SHA256 chunk encryption / PSS-SHA256
signatures), ``aesgcm_seal`` / ``aesgcm_open`` (AES-256-GCM, 12-byte nonce,
16-byte tag appended) and ``pbkdf2_sha256``.  The library is built on first
use if the shared object is missing and a compiler is available.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_int, c_long, c_void_p
from functools import lru_cache


class CryptoError(Exception):
    """A primitive failed (bad key, failed authentication, malformed input)."""


@lru_cache(maxsize=1)
def _lib() -> ctypes.CDLL:
    from nanofed_amd import native_build

    path = native_build.HOST_LIB
    if not path.exists():
        native_build.build_host()
    lib = ctypes.CDLL(str(path))
    lib.nf_rsa_generate.restype = c_void_p
    lib.nf_rsa_generate.argtypes = [c_int]
    lib.nf_rsa_free.argtypes = [c_void_p]
    lib.nf_rsa_size_bytes.argtypes = [c_void_p]
    lib.nf_rsa_public_pem.restype = c_long
    lib.nf_rsa_public_pem.argtypes = [c_void_p, c_char_p, c_long]
    lib.nf_rsa_from_public_pem.restype = c_void_p
    lib.nf_rsa_from_public_pem.argtypes = [c_char_p, c_long]
    lib.nf_rsa_oaep_encrypt_chunks.restype = c_long
    lib.nf_rsa_oaep_encrypt_chunks.argtypes = [
        c_void_p, c_char_p, c_long, c_long, c_char_p]
    lib.nf_rsa_oaep_decrypt_chunks.restype = c_long
    lib.nf_rsa_oaep_decrypt_chunks.argtypes = [
        c_void_p, c_char_p, c_long, c_long, c_char_p]
    lib.nf_rsa_pss_sign.restype = c_long
    lib.nf_rsa_pss_sign.argtypes = [c_void_p, c_char_p, c_long, c_char_p, c_long]
    lib.nf_rsa_pss_verify.restype = c_int
    lib.nf_rsa_pss_verify.argtypes = [
        c_void_p, c_char_p, c_long, c_char_p, c_long]
    lib.nf_aes256gcm_seal.restype = c_long
    lib.nf_aes256gcm_seal.argtypes = [
        c_char_p, c_char_p, c_char_p, c_long, c_char_p]
    lib.nf_aes256gcm_open.restype = c_long
    lib.nf_aes256gcm_open.argtypes = [
        c_char_p, c_char_p, c_char_p, c_long, c_char_p]
    lib.nf_pbkdf2_sha256.restype = c_int
    lib.nf_pbkdf2_sha256.argtypes = [
        c_char_p, c_long, c_char_p, c_long, c_int, c_char_p, c_long]
    lib.nf_x25519_generate.argtypes = [c_char_p, c_char_p]
    lib.nf_x25519_public.argtypes = [c_char_p, c_char_p]
    lib.nf_x25519_derive.argtypes = [c_char_p, c_char_p, c_char_p, c_long, c_char_p]
    return lib


class X25519Key:
    """An X25519 key pair (RFC 7748).  ``public`` is what travels; ``derive`` turns a
    peer's public key into 32 secret bytes both sides compute alike:
    SHA-256(shared secret || context)."""

    def __init__(self, private: bytes) -> None:
        if len(private) != 32:
            raise CryptoError("an X25519 private key is 32 bytes")
        self._private = bytes(private)
        pub = ctypes.create_string_buffer(32)
        if _lib().nf_x25519_public(self._private, pub) != 0:
            raise CryptoError("X25519 public key derivation failed")
        self.public = pub.raw

    @classmethod
    def generate(cls) -> "X25519Key":
        priv, pub = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
        if _lib().nf_x25519_generate(priv, pub) != 0:
            raise CryptoError("X25519 key generation failed")
        return cls(priv.raw)

    def derive(self, peer_public: bytes, context: bytes = b"") -> bytes:
        if len(peer_public) != 32:
            raise CryptoError("an X25519 public key is 32 bytes")
        out = ctypes.create_string_buffer(32)
        if _lib().nf_x25519_derive(self._private, bytes(peer_public), context, len(context), out) != 0:
            raise CryptoError("X25519 agreement failed (malformed or low-order peer key)")
        return out.raw


class RSAKey:
    """An RSA key pair (or a public key loaded from PEM)."""

    def __init__(self, handle: int, has_private: bool) -> None:
        if not handle:
            raise CryptoError("RSA key creation failed")
        self._h = c_void_p(handle)
        self._private = has_private
        self.size_bytes = int(_lib().nf_rsa_size_bytes(self._h))

    @classmethod
    def generate(cls, bits: int = 2048) -> "RSAKey":
        return cls(_lib().nf_rsa_generate(int(bits)), True)

    @classmethod
    def from_public_pem(cls, pem: bytes) -> "RSAKey":
        handle = _lib().nf_rsa_from_public_pem(pem, len(pem))
        if not handle:
            raise CryptoError("Not an RSA public key in PEM format")
        return cls(handle, False)

    def __del__(self) -> None:
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib().nf_rsa_free(h)
            except Exception:
                pass
            self._h = None

    @property
    def oaep_max_chunk(self) -> int:
        """Largest plaintext OAEP-SHA256 can carry with this key."""
        return self.size_bytes - 2 * 32 - 2

    def public_pem(self) -> bytes:
        buf = ctypes.create_string_buffer(4096)
        n = _lib().nf_rsa_public_pem(self._h, buf, len(buf))
        if n <= 0:
            raise CryptoError("PEM export failed")
        return buf.raw[:n]

    def oaep_encrypt_chunks(self, data: bytes, chunk_len: int) -> list[bytes]:
        """Encrypt ``data`` (a multiple of ``chunk_len`` bytes) chunk by chunk."""
        if chunk_len <= 0 or chunk_len > self.oaep_max_chunk:
            raise CryptoError("chunk length does not fit OAEP with this key")
        if len(data) % chunk_len:
            raise CryptoError("data is not a whole number of chunks")
        n = len(data) // chunk_len
        if n == 0:
            return []
        out = ctypes.create_string_buffer(n * self.size_bytes)
        rc = _lib().nf_rsa_oaep_encrypt_chunks(self._h, data, chunk_len, n, out)
        if rc != n:
            raise CryptoError(f"OAEP encryption failed (code {rc})")
        k = self.size_bytes
        raw = out.raw
        return [raw[i * k : (i + 1) * k] for i in range(n)]

    def oaep_decrypt_chunks(self, chunks: list[bytes], chunk_len: int) -> bytes:
        if not self._private:
            raise CryptoError("decryption needs the private key")
        n = len(chunks)
        if n == 0:
            return b""
        if any(len(c) != self.size_bytes for c in chunks):
            raise CryptoError("ciphertext chunk has the wrong length")
        out = ctypes.create_string_buffer(n * chunk_len)
        rc = _lib().nf_rsa_oaep_decrypt_chunks(
            self._h, b"".join(chunks), n, chunk_len, out
        )
        if rc != n:
            raise CryptoError(f"OAEP decryption failed at chunk {-rc - 2}")
        return out.raw

    def pss_sign(self, message: bytes) -> bytes:
        if not self._private:
            raise CryptoError("signing needs the private key")
        sig = ctypes.create_string_buffer(self.size_bytes)
        n = _lib().nf_rsa_pss_sign(self._h, message, len(message), sig, len(sig))
        if n <= 0:
            raise CryptoError("PSS signing failed")
        return sig.raw[:n]

    def pss_verify(self, message: bytes, signature: bytes) -> bool:
        rc = _lib().nf_rsa_pss_verify(
            self._h, message, len(message), signature, len(signature)
        )
        if rc < 0:
            raise CryptoError("PSS verification could not run")
        return rc == 1


def aesgcm_seal(key: bytes, nonce: bytes, plaintext: bytes) -> bytes:
    """AES-256-GCM: returns ciphertext || 16-byte tag."""
    if len(key) != 32 or len(nonce) != 12:
        raise CryptoError("AES-256-GCM needs a 32-byte key and a 12-byte nonce")
    out = ctypes.create_string_buffer(len(plaintext) + 16)
    n = _lib().nf_aes256gcm_seal(key, nonce, plaintext, len(plaintext), out)
    if n != len(plaintext) + 16:
        raise CryptoError("AES-GCM encryption failed")
    return out.raw


def aesgcm_open(key: bytes, nonce: bytes, sealed: bytes) -> bytes:
    """Inverse of :func:`aesgcm_seal`; raises on authentication failure."""
    if len(key) != 32 or len(nonce) != 12:
        raise CryptoError("AES-256-GCM needs a 32-byte key and a 12-byte nonce")
    if len(sealed) < 16:
        raise CryptoError("ciphertext shorter than the authentication tag")
    out = ctypes.create_string_buffer(max(len(sealed) - 16, 1))
    n = _lib().nf_aes256gcm_open(key, nonce, sealed, len(sealed), out)
    if n < 0:
        raise CryptoError("AES-GCM authentication failed")
    return out.raw[:n]


def pbkdf2_sha256(
    password: bytes, salt: bytes, iterations: int, length: int = 32
) -> bytes:
    out = ctypes.create_string_buffer(length)
    rc = _lib().nf_pbkdf2_sha256(
        password, len(password), salt, len(salt), int(iterations), out, length
    )
    if rc != 0:
        raise CryptoError("PBKDF2 failed")
    return out.raw


def random_bytes(n: int) -> bytes:
    return os.urandom(n)
