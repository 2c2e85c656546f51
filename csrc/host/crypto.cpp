// Host-side cryptography for the secure-aggregation and update-signing paths.
//
// A thin C ABI over OpenSSL 3 EVP: RSA key generation, RSA-OAEP(SHA-256,
// MGF1-SHA-256) encryption of fixed-size chunks, RSA-PSS(SHA-256, maximum salt)
// signatures, AES-256-GCM sealing, PBKDF2-HMAC-SHA-256 and X25519 key agreement
// (raw 32-byte keys; the shared secret is hashed with a caller-supplied context
// into pair seeds for the blinding masks of secure aggregation).  Loaded from Python
// with ctypes (nanofed_amd/crypto/backend.py).  Every function returns a
// negative value on failure and never throws; no state is kept besides the
// EVP_PKEY handles the caller owns.
//
// The chunked OAEP entry points process a whole tensor per call, so the
// Python side does not loop over 190-byte chunks.

#include <openssl/bio.h>
#include <openssl/evp.h>
#include <openssl/pem.h>
#include <openssl/rand.h>
#include <openssl/rsa.h>

#include <cstdint>
#include <cstring>

namespace {

struct CtxGuard {
    EVP_PKEY_CTX* ctx;
    explicit CtxGuard(EVP_PKEY_CTX* c) : ctx(c) {}
    ~CtxGuard() { if (ctx) EVP_PKEY_CTX_free(ctx); }
};

bool setup_oaep(EVP_PKEY_CTX* ctx) {
    return EVP_PKEY_CTX_set_rsa_padding(ctx, RSA_PKCS1_OAEP_PADDING) > 0 &&
           EVP_PKEY_CTX_set_rsa_oaep_md(ctx, EVP_sha256()) > 0 &&
           EVP_PKEY_CTX_set_rsa_mgf1_md(ctx, EVP_sha256()) > 0;
}

bool setup_pss(EVP_PKEY_CTX* ctx, EVP_PKEY* key) {
    // maximum salt length: emLen - hLen - 2 with emLen = ceil((bits-1)/8)
    const int bits = EVP_PKEY_get_bits(key);
    const int em_len = (bits - 1 + 7) / 8;
    const int salt = em_len - 32 - 2;
    return EVP_PKEY_CTX_set_rsa_padding(ctx, RSA_PKCS1_PSS_PADDING) > 0 &&
           EVP_PKEY_CTX_set_signature_md(ctx, EVP_sha256()) > 0 &&
           EVP_PKEY_CTX_set_rsa_mgf1_md(ctx, EVP_sha256()) > 0 &&
           EVP_PKEY_CTX_set_rsa_pss_saltlen(ctx, salt) > 0;
}

}  // namespace

extern "C" {

void* nf_rsa_generate(int bits) {
    return EVP_RSA_gen(static_cast<unsigned int>(bits));
}

void nf_rsa_free(void* key) {
    if (key) EVP_PKEY_free(static_cast<EVP_PKEY*>(key));
}

int nf_rsa_size_bytes(void* key) {
    return EVP_PKEY_get_size(static_cast<EVP_PKEY*>(key));
}

// SubjectPublicKeyInfo PEM. Returns the length written, or -needed if cap is
// too small.
long nf_rsa_public_pem(void* key, char* out, long cap) {
    BIO* bio = BIO_new(BIO_s_mem());
    if (!bio) return -1;
    long n = -1;
    if (PEM_write_bio_PUBKEY(bio, static_cast<EVP_PKEY*>(key)) > 0) {
        char* data = nullptr;
        const long len = BIO_get_mem_data(bio, &data);
        if (len <= cap) {
            std::memcpy(out, data, static_cast<size_t>(len));
            n = len;
        } else {
            n = -len;
        }
    }
    BIO_free(bio);
    return n;
}

void* nf_rsa_from_public_pem(const char* pem, long len) {
    BIO* bio = BIO_new_mem_buf(pem, static_cast<int>(len));
    if (!bio) return nullptr;
    EVP_PKEY* key = PEM_read_bio_PUBKEY(bio, nullptr, nullptr, nullptr);
    BIO_free(bio);
    if (key && EVP_PKEY_get_base_id(key) != EVP_PKEY_RSA) {
        EVP_PKEY_free(key);
        return nullptr;
    }
    return key;
}

// Encrypts n_chunks plaintext chunks of chunk_len bytes each (contiguous in
// `in`) into n_chunks ciphertexts of key-size bytes each (contiguous in `out`).
long nf_rsa_oaep_encrypt_chunks(void* key, const uint8_t* in, long chunk_len,
                                long n_chunks, uint8_t* out) {
    EVP_PKEY* pkey = static_cast<EVP_PKEY*>(key);
    CtxGuard g(EVP_PKEY_CTX_new(pkey, nullptr));
    if (!g.ctx || EVP_PKEY_encrypt_init(g.ctx) <= 0 || !setup_oaep(g.ctx))
        return -1;
    const size_t ksz = static_cast<size_t>(EVP_PKEY_get_size(pkey));
    for (long i = 0; i < n_chunks; ++i) {
        size_t outlen = ksz;
        if (EVP_PKEY_encrypt(g.ctx, out + i * ksz, &outlen, in + i * chunk_len,
                             static_cast<size_t>(chunk_len)) <= 0 ||
            outlen != ksz)
            return -2 - i;
    }
    return n_chunks;
}

// Inverse of the above. Every plaintext must come out exactly chunk_len long.
// Returns n_chunks, or -(2+i) when chunk i fails to decrypt.
long nf_rsa_oaep_decrypt_chunks(void* key, const uint8_t* in, long n_chunks,
                                long chunk_len, uint8_t* out) {
    EVP_PKEY* pkey = static_cast<EVP_PKEY*>(key);
    CtxGuard g(EVP_PKEY_CTX_new(pkey, nullptr));
    if (!g.ctx || EVP_PKEY_decrypt_init(g.ctx) <= 0 || !setup_oaep(g.ctx))
        return -1;
    const size_t ksz = static_cast<size_t>(EVP_PKEY_get_size(pkey));
    uint8_t scratch[1024];
    if (ksz > sizeof(scratch)) return -1;
    for (long i = 0; i < n_chunks; ++i) {
        size_t outlen = ksz;
        if (EVP_PKEY_decrypt(g.ctx, scratch, &outlen, in + i * ksz, ksz) <= 0 ||
            outlen != static_cast<size_t>(chunk_len))
            return -2 - i;
        std::memcpy(out + i * chunk_len, scratch, outlen);
    }
    return n_chunks;
}

long nf_rsa_pss_sign(void* key, const uint8_t* msg, long len, uint8_t* sig,
                     long cap) {
    EVP_PKEY* pkey = static_cast<EVP_PKEY*>(key);
    EVP_MD_CTX* md = EVP_MD_CTX_new();
    if (!md) return -1;
    long result = -1;
    EVP_PKEY_CTX* pctx = nullptr;
    size_t siglen = static_cast<size_t>(cap);
    if (EVP_DigestSignInit(md, &pctx, EVP_sha256(), nullptr, pkey) > 0 &&
        setup_pss(pctx, pkey) &&
        EVP_DigestSign(md, sig, &siglen, msg, static_cast<size_t>(len)) > 0)
        result = static_cast<long>(siglen);
    EVP_MD_CTX_free(md);
    return result;
}

// 1 = valid, 0 = invalid signature, <0 = error.
int nf_rsa_pss_verify(void* key, const uint8_t* msg, long len,
                      const uint8_t* sig, long siglen) {
    EVP_PKEY* pkey = static_cast<EVP_PKEY*>(key);
    EVP_MD_CTX* md = EVP_MD_CTX_new();
    if (!md) return -1;
    int result = -1;
    EVP_PKEY_CTX* pctx = nullptr;
    if (EVP_DigestVerifyInit(md, &pctx, EVP_sha256(), nullptr, pkey) > 0 &&
        setup_pss(pctx, pkey)) {
        const int rc = EVP_DigestVerify(md, sig, static_cast<size_t>(siglen), msg,
                                        static_cast<size_t>(len));
        result = rc == 1 ? 1 : 0;
    }
    EVP_MD_CTX_free(md);
    return result;
}

// out = ciphertext (len bytes) || tag (16 bytes)
long nf_aes256gcm_seal(const uint8_t* key, const uint8_t* nonce12,
                       const uint8_t* pt, long len, uint8_t* out) {
    EVP_CIPHER_CTX* c = EVP_CIPHER_CTX_new();
    if (!c) return -1;
    long result = -1;
    int n = 0, tail = 0;
    if (EVP_EncryptInit_ex(c, EVP_aes_256_gcm(), nullptr, nullptr, nullptr) > 0 &&
        EVP_CIPHER_CTX_ctrl(c, EVP_CTRL_GCM_SET_IVLEN, 12, nullptr) > 0 &&
        EVP_EncryptInit_ex(c, nullptr, nullptr, key, nonce12) > 0 &&
        EVP_EncryptUpdate(c, out, &n, pt, static_cast<int>(len)) > 0 &&
        EVP_EncryptFinal_ex(c, out + n, &tail) > 0 &&
        EVP_CIPHER_CTX_ctrl(c, EVP_CTRL_GCM_GET_TAG, 16, out + n + tail) > 0)
        result = n + tail + 16;
    EVP_CIPHER_CTX_free(c);
    return result;
}

// in = ciphertext || tag(16). Returns plaintext length or -1 (authentication
// failure included).
long nf_aes256gcm_open(const uint8_t* key, const uint8_t* nonce12,
                       const uint8_t* in, long len, uint8_t* out) {
    if (len < 16) return -1;
    EVP_CIPHER_CTX* c = EVP_CIPHER_CTX_new();
    if (!c) return -1;
    long result = -1;
    const long ct_len = len - 16;
    int n = 0, tail = 0;
    if (EVP_DecryptInit_ex(c, EVP_aes_256_gcm(), nullptr, nullptr, nullptr) > 0 &&
        EVP_CIPHER_CTX_ctrl(c, EVP_CTRL_GCM_SET_IVLEN, 12, nullptr) > 0 &&
        EVP_DecryptInit_ex(c, nullptr, nullptr, key, nonce12) > 0 &&
        EVP_DecryptUpdate(c, out, &n, in, static_cast<int>(ct_len)) > 0 &&
        EVP_CIPHER_CTX_ctrl(c, EVP_CTRL_GCM_SET_TAG, 16,
                            const_cast<uint8_t*>(in + ct_len)) > 0 &&
        EVP_DecryptFinal_ex(c, out + n, &tail) > 0)
        result = n + tail;
    EVP_CIPHER_CTX_free(c);
    return result;
}

int nf_pbkdf2_sha256(const uint8_t* pass, long passlen, const uint8_t* salt,
                     long saltlen, int iters, uint8_t* out, long outlen) {
    return PKCS5_PBKDF2_HMAC(reinterpret_cast<const char*>(pass),
                             static_cast<int>(passlen), salt,
                             static_cast<int>(saltlen), iters, EVP_sha256(),
                             static_cast<int>(outlen), out) > 0
               ? 0
               : -1;
}

// X25519 key pair: 32 raw private and 32 raw public bytes.
int nf_x25519_generate(uint8_t* priv32, uint8_t* pub32) {
    EVP_PKEY* key = EVP_PKEY_Q_keygen(nullptr, nullptr, "X25519");
    if (!key) return -1;
    size_t n_priv = 32, n_pub = 32;
    const bool ok = EVP_PKEY_get_raw_private_key(key, priv32, &n_priv) > 0 && n_priv == 32 &&
                    EVP_PKEY_get_raw_public_key(key, pub32, &n_pub) > 0 && n_pub == 32;
    EVP_PKEY_free(key);
    return ok ? 0 : -2;
}

// Public half of a raw private key.
int nf_x25519_public(const uint8_t* priv32, uint8_t* pub32) {
    EVP_PKEY* key = EVP_PKEY_new_raw_private_key(EVP_PKEY_X25519, nullptr, priv32, 32);
    if (!key) return -1;
    size_t n_pub = 32;
    const bool ok = EVP_PKEY_get_raw_public_key(key, pub32, &n_pub) > 0 && n_pub == 32;
    EVP_PKEY_free(key);
    return ok ? 0 : -2;
}

// out32 = SHA-256(X25519(priv, peer_pub) || context).  Fails (< 0) on a malformed or
// low-order peer key (OpenSSL rejects an all-zero shared secret).
int nf_x25519_derive(const uint8_t* priv32, const uint8_t* peer_pub32, const uint8_t* context,
                     long context_len, uint8_t* out32) {
    EVP_PKEY* mine = EVP_PKEY_new_raw_private_key(EVP_PKEY_X25519, nullptr, priv32, 32);
    EVP_PKEY* peer = EVP_PKEY_new_raw_public_key(EVP_PKEY_X25519, nullptr, peer_pub32, 32);
    int rc = -1;
    uint8_t secret[32];
    size_t n = sizeof(secret);
    if (mine && peer) {
        CtxGuard g(EVP_PKEY_CTX_new(mine, nullptr));
        if (g.ctx && EVP_PKEY_derive_init(g.ctx) > 0 && EVP_PKEY_derive_set_peer(g.ctx, peer) > 0 &&
            EVP_PKEY_derive(g.ctx, secret, &n) > 0 && n == 32) {
            EVP_MD_CTX* md = EVP_MD_CTX_new();
            unsigned int out_len = 0;
            if (md && EVP_DigestInit_ex(md, EVP_sha256(), nullptr) > 0 &&
                EVP_DigestUpdate(md, secret, 32) > 0 &&
                EVP_DigestUpdate(md, context, static_cast<size_t>(context_len)) > 0 &&
                EVP_DigestFinal_ex(md, out32, &out_len) > 0 && out_len == 32)
                rc = 0;
            if (md) EVP_MD_CTX_free(md);
        }
        OPENSSL_cleanse(secret, sizeof(secret));
    }
    if (mine) EVP_PKEY_free(mine);
    if (peer) EVP_PKEY_free(peer);
    return rc;
}

int nf_random_bytes(uint8_t* out, long len) {
    return RAND_bytes(out, static_cast<int>(len)) > 0 ? 0 : -1;
}

}  // extern "C"
