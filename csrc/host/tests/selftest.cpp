// Self-test of the host library, meant to run under AddressSanitizer and
// UndefinedBehaviorSanitizer (tools/host_sanitize.sh).  It drives the JSON
// tensor codec with well-formed, truncated and ragged inputs, the AES-GCM and
// RSA wrappers with valid and tampered data, and the CPU Philox sampler with
// unaligned ranges - the places where a length mistake would read or write
// out of bounds.

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <limits>
#include <vector>

extern "C" {
long nf_json_encode_bound(long numel, int ndim);
long nf_json_encode_f32(const float* data, const long* shape, int ndim, char* out, long cap);
long nf_json_decode_f32(const char* text, long len, float* out, long cap, long* shape_out,
                        int* ndim_out, long* consumed_out);
long nf_json_skip_value(const char* text, long len, long pos);
long nf_aes256gcm_seal(const uint8_t* key, const uint8_t* nonce12, const uint8_t* pt, long len,
                       uint8_t* out);
long nf_aes256gcm_open(const uint8_t* key, const uint8_t* nonce12, const uint8_t* in, long len,
                       uint8_t* out);
int nf_pbkdf2_sha256(const uint8_t* pass, long passlen, const uint8_t* salt, long saltlen,
                     int iters, uint8_t* out, long outlen);
int nf_random_bytes(uint8_t* out, long len);
void nf_philox_cpu(float* out, long n, unsigned long long seed, unsigned long long offset,
                   long first, int mode, float scale, int accumulate);
}

static int failures = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); \
            ++failures;                                                    \
        }                                                                  \
    } while (0)

static void test_json() {
    const long shape[3] = {3, 4, 5};
    std::vector<float> data(60);
    for (int i = 0; i < 60; ++i) data[i] = std::ldexp(static_cast<float>(i * 37 % 101) - 50.f, i % 40 - 20);
    data[7] = 0.f;
    data[8] = -0.f;
    data[9] = 3.4028235e38f;
    data[10] = 1.4e-45f;
    const long bound = nf_json_encode_bound(60, 3);
    std::vector<char> text(bound);
    const long n = nf_json_encode_f32(data.data(), shape, 3, text.data(), bound);
    CHECK(n > 0 && n <= bound);
    // exact-size and too-small buffers
    std::vector<char> tight(n);
    CHECK(nf_json_encode_f32(data.data(), shape, 3, tight.data(), n) == n);
    for (long cap : {0L, 1L, n / 2, n - 1}) {
        std::vector<char> small(cap > 0 ? cap : 1);
        CHECK(nf_json_encode_f32(data.data(), shape, 3, small.data(), cap) < 0);
    }
    std::vector<float> back(60);
    long shp[8];
    int ndim = 0;
    CHECK(nf_json_decode_f32(text.data(), n, back.data(), 60, shp, &ndim, nullptr) == 60);
    CHECK(ndim == 3 && shp[0] == 3 && shp[1] == 4 && shp[2] == 5);
    CHECK(std::memcmp(back.data(), data.data(), 60 * sizeof(float)) == 0);
    // capacity smaller than the value count: count is still reported, nothing overflows
    std::vector<float> few(7);
    CHECK(nf_json_decode_f32(text.data(), n, few.data(), 7, shp, &ndim, nullptr) == 60);
    // every truncation of the text must be rejected or parsed without overrun
    for (long cut = 0; cut < n; cut += 3) {
        std::vector<char> part(text.begin(), text.begin() + cut);
        (void)nf_json_decode_f32(part.data(), cut, back.data(), 60, shp, &ndim, nullptr);
        (void)nf_json_skip_value(part.data(), cut, 0);
    }
    const char* bad[] = {"[[1,2],[3]]", "[1,[2]]", "[1,,2]", "[1 2]", "]", "[[[[[[[[[[1]]]]]]]]]]",
                         "[1e99999]", "[nan]", "", "   ", "[\"a\"]", "[1,2]x"};
    for (const char* b : bad) {
        const long len = static_cast<long>(std::strlen(b));
        const long r = nf_json_decode_f32(b, len, back.data(), 60, shp, &ndim, nullptr);
        CHECK(r == -1 || r >= 0);  // must not crash; most are -1
    }
    CHECK(nf_json_decode_f32("[[1,2],[3]]", 11, back.data(), 60, shp, &ndim, nullptr) == -1);
    long used = 0;
    const char* doc = " [1.5, -2] , \"x\"";
    CHECK(nf_json_decode_f32(doc, static_cast<long>(std::strlen(doc)), back.data(), 60, shp, &ndim,
                             &used) == 2);
    CHECK(used == 10 && back[0] == 1.5f && back[1] == -2.f);
    const char* obj = "{\"a\": [1, {\"b\": \"]\\\"\"}], \"c\": 2} tail";
    CHECK(nf_json_skip_value(obj, static_cast<long>(std::strlen(obj)), 0) == 32);
    CHECK(nf_json_skip_value("\"open", 5, 0) == -1);
    CHECK(nf_json_skip_value("[1,2", 4, 0) == -1);

    // large tensors take the threaded paths (rows split over threads): same text, same
    // values, trailing text left alone, truncation and ragged rows rejected
    const long R = 96, C = 4096;
    std::vector<float> big(static_cast<size_t>(R * C));
    for (long i = 0; i < R * C; ++i) big[i] = 0.001f * static_cast<float>(i % 9973) - 3.25f;
    big[5] = std::numeric_limits<float>::quiet_NaN();
    big[C + 1] = -std::numeric_limits<float>::infinity();
    const long bshape[2] = {R, C};
    const long bbound = nf_json_encode_bound(R * C, 2);
    std::vector<char> btext(static_cast<size_t>(bbound) + 8);
    const long bn = nf_json_encode_f32(big.data(), bshape, 2, btext.data(), bbound);
    CHECK(bn > (1L << 21));
    std::vector<char> exact(static_cast<size_t>(bn));
    CHECK(nf_json_encode_f32(big.data(), bshape, 2, exact.data(), bn) == bn);  // checked writer
    CHECK(std::memcmp(exact.data(), btext.data(), static_cast<size_t>(bn)) == 0);
    std::vector<float> bback(static_cast<size_t>(R * C));
    CHECK(nf_json_decode_f32(btext.data(), bn, bback.data(), R * C, shp, &ndim, nullptr) == R * C);
    CHECK(ndim == 2 && shp[0] == R && shp[1] == C);
    for (long i = 0; i < R * C; ++i)
        CHECK(std::memcmp(&bback[i], &big[i], 4) == 0 || (big[i] != big[i] && bback[i] != bback[i]));
    std::memcpy(btext.data() + bn, ", 7", 3);
    CHECK(nf_json_decode_f32(btext.data(), bn + 3, bback.data(), R * C, shp, &ndim, &used) == R * C);
    CHECK(used == bn);
    CHECK(nf_json_decode_f32(btext.data(), bn - 1, bback.data(), R * C, shp, &ndim, nullptr) == -1);
    CHECK(nf_json_decode_f32(btext.data(), bn, bback.data(), 10, shp, &ndim, nullptr) == R * C);
    // drop one number from a row in the middle: ragged
    {
        std::vector<char> rag(btext.begin(), btext.begin() + bn);
        long at = bn / 2;
        while (rag[at] != ',') ++at;
        long stop = at + 1;
        while (rag[stop] != ',' && rag[stop] != ']') ++stop;
        for (long i = at; i < stop; ++i) rag[i] = ' ';
        CHECK(nf_json_decode_f32(rag.data(), bn, bback.data(), R * C, shp, &ndim, nullptr) == -1);
    }
}

static void test_aes() {
    uint8_t key[32], nonce[12];
    CHECK(nf_random_bytes(key, 32) == 0 && nf_random_bytes(nonce, 12) == 0);
    for (long len : {0L, 1L, 15L, 16L, 17L, 4096L, 100003L}) {
        std::vector<uint8_t> pt(len ? len : 1), ct(len + 16), out(len ? len : 1);
        for (long i = 0; i < len; ++i) pt[i] = static_cast<uint8_t>(i * 131 + 7);
        CHECK(nf_aes256gcm_seal(key, nonce, pt.data(), len, ct.data()) == len + 16);
        CHECK(nf_aes256gcm_open(key, nonce, ct.data(), len + 16, out.data()) == len);
        CHECK(len == 0 || std::memcmp(pt.data(), out.data(), len) == 0);
        ct[len / 2] ^= 1;  // tamper (ciphertext or tag)
        CHECK(nf_aes256gcm_open(key, nonce, ct.data(), len + 16, out.data()) == -1);
    }
    uint8_t tiny[8] = {0};
    uint8_t sink[8];
    CHECK(nf_aes256gcm_open(key, nonce, tiny, 8, sink) == -1);
    uint8_t derived[32], again[32];
    const uint8_t salt[4] = {1, 2, 3, 4};
    CHECK(nf_pbkdf2_sha256(reinterpret_cast<const uint8_t*>("pw"), 2, salt, 4, 1000, derived, 32) == 0);
    CHECK(nf_pbkdf2_sha256(reinterpret_cast<const uint8_t*>("pw"), 2, salt, 4, 1000, again, 32) == 0);
    CHECK(std::memcmp(derived, again, 32) == 0);
}

static void test_philox() {
    // a range fetched in one call equals the same range fetched in unaligned pieces
    const long n = 1003;
    for (int mode = 0; mode < 3; ++mode) {
        std::vector<float> whole(n), pieces(n);
        nf_philox_cpu(whole.data(), n, 42ull, 7ull, 5, mode, 2.0f, 0);
        long pos = 0;
        const long cuts[] = {1, 2, 3, 5, 64, 129, 799};
        for (long c : cuts) {
            nf_philox_cpu(pieces.data() + pos, c, 42ull, 7ull, 5 + pos, mode, 2.0f, 0);
            pos += c;
        }
        CHECK(pos == n);
        CHECK(std::memcmp(whole.data(), pieces.data(), n * sizeof(float)) == 0);
        std::vector<float> acc(n, 1.0f);
        nf_philox_cpu(acc.data(), n, 42ull, 7ull, 5, mode, 2.0f, 1);
        for (long i = 0; i < n; i += 97) CHECK(acc[i] == 1.0f + whole[i]);
        double mean = 0;
        for (float v : whole) { CHECK(std::isfinite(v)); mean += v; }
        (void)mean;
    }
    nf_philox_cpu(nullptr, 0, 1ull, 1ull, 0, 0, 1.0f, 0);
}

int main() {
    test_json();
    test_aes();
    test_philox();
    if (failures) {
        std::fprintf(stderr, "%d check(s) failed\n", failures);
        return 1;
    }
    std::puts("host selftest ok");
    return 0;
}
