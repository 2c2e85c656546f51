from .backend import (
    CryptoError,
    RSAKey,
    aesgcm_open,
    aesgcm_seal,
    pbkdf2_sha256,
    random_bytes,
)

__all__ = [
    "CryptoError",
    "RSAKey",
    "aesgcm_seal",
    "aesgcm_open",
    "pbkdf2_sha256",
    "random_bytes",
]
