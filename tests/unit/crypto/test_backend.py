"""The native crypto primitives (csrc/host/crypto.cpp over OpenSSL's EVP interface)
against published vectors and against each other."""

import hashlib

import pytest

from nanofed_amd.crypto.backend import (CryptoError, RSAKey, X25519Key, aesgcm_open, aesgcm_seal,
                                        pbkdf2_sha256, random_bytes)


def test_x25519_rfc7748_vector():
    # RFC 7748, section 6.1
    alice = bytes.fromhex("77076d0a7318a57d3c16c17251b26645df4c2f87ebc0992ab177fba51db92c2a")
    bob = bytes.fromhex("5dab087e624a8a4b79e17f8b83800ee66f3bb1292618b6fd1c2f8b27ff88e0eb")
    shared = bytes.fromhex("4a5d9d5ba4ce2de1728e3bf480350f25e07e21c947d19e3376f09b3c1e161742")
    a, b = X25519Key(alice), X25519Key(bob)
    assert a.public.hex() == "8520f0098930a754748b7ddcb43ef75a0dbf3a0d26381af4eba4a98eaa9b4e6a"
    assert b.public.hex() == "de9edb7d7b7dc1b4d35b61c2ece435373f8343c85b78674dadfc7e146f882b4f"
    # derive = SHA-256(shared secret || context), the same on both sides
    assert a.derive(b.public) == b.derive(a.public) == hashlib.sha256(shared).digest()
    assert a.derive(b.public, b"ctx") == hashlib.sha256(shared + b"ctx").digest()


def test_x25519_fresh_keys_agree_and_bad_peers_are_refused():
    a, b, c = X25519Key.generate(), X25519Key.generate(), X25519Key.generate()
    assert len({a.public, b.public, c.public}) == 3
    assert a.derive(b.public, b"x") == b.derive(a.public, b"x")
    assert a.derive(b.public, b"x") != a.derive(c.public, b"x")
    assert a.derive(b.public, b"x") != a.derive(b.public, b"y")
    with pytest.raises(CryptoError):
        a.derive(b"\x00" * 32)  # low-order point: all-zero shared secret
    with pytest.raises(CryptoError):
        a.derive(b"short")
    with pytest.raises(CryptoError):
        X25519Key(b"too short")


def test_rsa_pss_and_oaep_round_trip():
    key = RSAKey.generate(2048)
    pub = RSAKey.from_public_pem(key.public_pem())
    sig = key.pss_sign(b"update 7")
    assert pub.pss_verify(b"update 7", sig) and not pub.pss_verify(b"update 8", sig)
    data = random_bytes(3 * 100)
    assert key.oaep_decrypt_chunks(pub.oaep_encrypt_chunks(data, 100), 100) == data


def test_aes_gcm_authenticates_and_pbkdf2_matches_hashlib():
    key, nonce = random_bytes(32), random_bytes(12)
    sealed = aesgcm_seal(key, nonce, b"tensor bytes")
    assert aesgcm_open(key, nonce, sealed) == b"tensor bytes"
    with pytest.raises(CryptoError):
        aesgcm_open(key, nonce, sealed[:-1] + bytes([sealed[-1] ^ 1]))
    assert pbkdf2_sha256(b"pw", b"salt", 1000, 32) == hashlib.pbkdf2_hmac("sha256", b"pw", b"salt", 1000, 32)
