// Tensor <-> JSON nested-list codec for the HTTP wire format.
//
// The wire schema carries every tensor as nested JSON lists of numbers.  Going
// through Python float objects (tensor.tolist() -> json.dumps, json.loads ->
// torch.tensor) costs seconds for a 1.2M-parameter model.  These two functions
// convert directly between a contiguous fp32 buffer and JSON text:
//
//  * encode: shortest decimal that round-trips the fp32 value (std::to_chars),
//    so the text is about half the size of repr(float64(x)) and parses back to
//    the identical fp32;
//  * decode: parses a (possibly ragged-free) nested list of numbers, returning
//    the shape and the values as fp32; "NaN", "Infinity" and "-Infinity" (what
//    Python's json emits) are accepted.
//
// C ABI, loaded with ctypes (nanofed_amd/communication/codec.py).

#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <utility>
#include <vector>

namespace {

struct Writer {
    char* out;
    long cap;
    long pos = 0;
    bool overflow = false;
    void put(char c) {
        if (pos < cap) out[pos] = c; else overflow = true;
        ++pos;
    }
    void put(const char* s) { while (*s) put(*s++); }
    void number(float v) {
        if (std::isnan(v)) { put("NaN"); return; }
        if (std::isinf(v)) { put(v > 0 ? "Infinity" : "-Infinity"); return; }
        char buf[48];
        auto r = std::to_chars(buf, buf + sizeof(buf), v);
        bool is_int = true;
        for (char* p = buf; p != r.ptr; ++p)
            if (*p == '.' || *p == 'e' || *p == 'n' || *p == 'i') is_int = false;
        for (char* p = buf; p != r.ptr; ++p) put(*p);
        if (is_int) put(".0");  // keep it a JSON float, as tolist() would
    }
};

void encode_rec(Writer& w, const float*& data, const long* shape, int ndim) {
    if (ndim == 0) { w.number(*data++); return; }
    w.put('[');
    for (long i = 0; i < shape[0]; ++i) {
        if (i) { w.put(','); w.put(' '); }
        encode_rec(w, data, shape + 1, ndim - 1);
    }
    w.put(']');
}

// The same text written without a bounds check per character: the caller has room for
// nf_json_encode_bound() bytes (20 per number: at most 15 of digits / sign / point /
// exponent, ".0", ", " and slack for brackets).
inline char* number_fast(char* p, float v) {
    if (__builtin_expect(!std::isfinite(v), 0)) {
        const char* s = std::isnan(v) ? "NaN" : (v > 0 ? "Infinity" : "-Infinity");
        while (*s) *p++ = *s++;
        return p;
    }
    char* const first = p;
    p = std::to_chars(p, p + 24, v).ptr;
    bool is_int = true;
    for (const char* q = first; q != p; ++q)
        if (*q == '.' || *q == 'e') { is_int = false; break; }
    if (is_int) { *p++ = '.'; *p++ = '0'; }  // keep it a JSON float, as tolist() would
    return p;
}

char* encode_fast(char* p, const float*& data, const long* shape, int ndim) {
    if (ndim == 0) return number_fast(p, *data++);
    *p++ = '[';
    if (ndim == 1) {
        const long n = shape[0];
        const float* d = data;
        for (long i = 0; i < n; ++i) {
            if (i) { *p++ = ','; *p++ = ' '; }
            p = number_fast(p, d[i]);
        }
        data += n;
    } else {
        for (long i = 0; i < shape[0]; ++i) {
            if (i) { *p++ = ','; *p++ = ' '; }
            p = encode_fast(p, data, shape + 1, ndim - 1);
        }
    }
    *p++ = ']';
    return p;
}

// Worker threads of the parallel paths: min(hardware threads, 8), or NANOFED_JSON_THREADS
// (1 switches the parallel paths off).  Read once.
int codec_threads() {
    static const int cached = [] {
        if (const char* env = std::getenv("NANOFED_JSON_THREADS")) {
            const long v = std::strtol(env, nullptr, 10);
            if (v >= 1) return static_cast<int>(v > 64 ? 64 : v);
        }
        const unsigned hw = std::thread::hardware_concurrency();
        return static_cast<int>(hw == 0 ? 4 : (hw > 8 ? 8 : hw));
    }();
    return cached;
}

// Largest text the parallel decoder takes (it holds about five times the text in
// buffers of its own; the bodies come from the network): NANOFED_JSON_PARALLEL_MAX_MB,
// 256 MB by default.  Longer texts are parsed by the sequential reader, in place.
long parallel_decode_limit() {
    static const long cached = [] {
        long mb = 256;
        if (const char* env = std::getenv("NANOFED_JSON_PARALLEL_MAX_MB")) {
            const long v = std::strtol(env, nullptr, 10);
            if (v >= 0) mb = v;
        }
        return mb << 20;
    }();
    return cached;
}

// Joins whatever was started, also when starting the next thread throws.
struct JoinAll {
    std::vector<std::thread>& pool;
    ~JoinAll() {
        for (auto& th : pool)
            if (th.joinable()) th.join();
    }
};

// Large tensors: the rows of the outermost dimension are cut into contiguous ranges, one
// per thread; every thread writes its rows (", " between them) into a buffer of its own
// and the pieces are joined.  The text is the same as the single-threaded one.
// Returns the bytes written, or -1 when buffers or threads could not be had (the caller
// then encodes sequentially); never throws.
long encode_parallel(char* out, const float* data, const long* shape, int ndim, int threads) try {
    const long rows = shape[0];
    long row_numel = 1, row_lists = 0;
    for (int i = 1; i < ndim; ++i) { row_lists += row_numel; row_numel *= shape[i]; }
    const long row_bound = row_numel * 17 + row_lists * 4 + 8;
    std::vector<std::vector<char>> piece(threads);
    std::vector<long> used(threads, 0);
    // (the buffers before the threads: an allocation that fails is caught here, not in a
    // thread, where it would end the process)
    for (int t = 0; t < threads; ++t) {
        const long r0 = rows * t / threads, r1 = rows * (t + 1) / threads;
        piece[t].resize(static_cast<size_t>((r1 - r0) * row_bound + 8));
    }
    {
        std::vector<std::thread> pool;
        JoinAll guard{pool};
        pool.reserve(static_cast<size_t>(threads));
        for (int t = 0; t < threads; ++t) {
            const long r0 = rows * t / threads, r1 = rows * (t + 1) / threads;
            pool.emplace_back([&, t, r0, r1]() noexcept {
                char* p = piece[t].data();
                const float* cursor = data + r0 * row_numel;
                for (long r = r0; r < r1; ++r) {
                    if (r != r0) { *p++ = ','; *p++ = ' '; }
                    p = encode_fast(p, cursor, shape + 1, ndim - 1);
                }
                used[t] = p - piece[t].data();
            });
        }
    }
    char* p = out;
    *p++ = '[';
    bool first = true;
    for (int t = 0; t < threads; ++t) {
        if (used[t] == 0) continue;
        if (!first) { *p++ = ','; *p++ = ' '; }
        first = false;
        std::memcpy(p, piece[t].data(), static_cast<size_t>(used[t]));
        p += used[t];
    }
    *p++ = ']';
    return p - out;
} catch (...) {
    return -1;
}

// Exact powers of ten in double: 10^0 .. 10^22.
const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                           1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// Decimal text -> double for the common case (at most 15 significant digits in total,
// decimal exponent within +-22): significand and power of ten are both exact doubles, so
// one multiplication or division gives the correctly rounded result - what from_chars
// returns, far faster.  Returns nullptr when the text is not of that form (the caller
// falls back to from_chars).
inline const char* parse_fast(const char* p, const char* end, double* out) {
    const char* q = p;
    bool neg = false;
    if (q < end && *q == '-') { neg = true; ++q; }
    uint64_t mant = 0;
    int digits = 0, frac = 0;
    bool any = false, nonzero = false;
    while (q < end && *q >= '0' && *q <= '9') {
        any = true;
        if (nonzero || *q != '0') { nonzero = true; mant = mant * 10 + static_cast<uint64_t>(*q - '0'); ++digits; }
        ++q;
    }
    if (q < end && *q == '.') {
        ++q;
        while (q < end && *q >= '0' && *q <= '9') {
            any = true;
            if (nonzero || *q != '0') { nonzero = true; mant = mant * 10 + static_cast<uint64_t>(*q - '0'); ++digits; }
            ++frac;
            ++q;
        }
    }
    if (!any || digits > 15) return nullptr;
    int exp10 = 0;
    if (q < end && (*q == 'e' || *q == 'E')) {
        ++q;
        bool eneg = false;
        if (q < end && (*q == '+' || *q == '-')) { eneg = *q == '-'; ++q; }
        if (q >= end || *q < '0' || *q > '9') return nullptr;
        int e = 0;
        while (q < end && *q >= '0' && *q <= '9') {
            if (e < 10000) e = e * 10 + (*q - '0');
            ++q;
        }
        exp10 = eneg ? -e : e;
    }
    exp10 -= frac;
    double d = static_cast<double>(mant);
    if (exp10 < -22 || exp10 > 22) return nullptr;
    d = exp10 < 0 ? d / kPow10[-exp10] : d * kPow10[exp10];
    *out = neg ? -d : d;
    return q;
}

struct Reader {
    const char* p;
    const char* end;
    float* out;
    long cap;
    long count = 0;
    long shape[8];
    bool seen[8];
    int max_depth = 0;
    bool fail = false;

    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }

    bool literal(const char* s, float v) {
        const size_t n = std::strlen(s);
        if (static_cast<size_t>(end - p) >= n && std::memcmp(p, s, n) == 0) {
            p += n;
            emit(v);
            return true;
        }
        return false;
    }
    void emit(float v) {
        if (count < cap) out[count] = v;
        ++count;
    }
    void value(int depth) {
        ws();
        if (p >= end) { fail = true; return; }
        if (*p == '[') {
            if (depth >= 8) { fail = true; return; }
            ++p;
            long n = 0;
            ws();
            if (p < end && *p == ']') { ++p; }
            else {
                for (;;) {
                    value(depth + 1);
                    if (fail) return;
                    ++n;
                    ws();
                    if (p < end && *p == ',') { ++p; continue; }
                    if (p < end && *p == ']') { ++p; break; }
                    fail = true; return;
                }
            }
            if (!seen[depth]) { seen[depth] = true; shape[depth] = n; }
            else if (shape[depth] != n) { fail = true; return; }
            if (depth + 1 > max_depth) max_depth = depth + 1;
            return;
        }
        if (literal("NaN", NAN) || literal("Infinity", INFINITY) ||
            literal("-Infinity", -INFINITY))
            return;
        if (literal("true", 1.0f) || literal("false", 0.0f)) return;
        double d = 0;
        if (const char* q = parse_fast(p, end, &d)) {
            p = q;
        } else {
            auto r = std::from_chars(p, end, d);
            if (r.ec != std::errc()) { fail = true; return; }
            p = r.ptr;
        }
        emit(static_cast<float>(d));
    }
};

// A large two-dimensional list: find the rows (memchr for their closing brackets), let a
// few threads parse contiguous ranges of rows into buffers of their own, join.  Returns
// false when the text is not of that plain form (the caller parses it sequentially, which
// also produces the error for malformed input).
bool decode_rows_parallel(const char* text, const char* end, float* out, long cap, long* count_out,
                          long* rows_out, long* cols_out, const char** after, int threads) try {
    auto skip_ws = [&](const char* p) {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
        return p;
    };
    const char* p = skip_ws(text);
    if (p >= end || *p != '[') return false;
    p = skip_ws(p + 1);
    std::vector<std::pair<const char*, const char*>> rows;
    for (;;) {
        if (p >= end || *p != '[') return false;
        const char* close = static_cast<const char*>(std::memchr(p + 1, ']', static_cast<size_t>(end - p - 1)));
        if (close == nullptr) return false;
        if (std::memchr(p + 1, '[', static_cast<size_t>(close - p - 1)) != nullptr) return false;  // deeper nesting
        if (std::memchr(p + 1, '"', static_cast<size_t>(close - p - 1)) != nullptr) return false;
        rows.emplace_back(p, close + 1);
        p = skip_ws(close + 1);
        if (p < end && *p == ',') { p = skip_ws(p + 1); continue; }
        if (p < end && *p == ']') { ++p; break; }
        return false;
    }
    const long nrows = static_cast<long>(rows.size());
    if (nrows < threads) return false;
    std::vector<std::vector<float>> piece(threads);
    std::vector<long> counts(threads, 0), cols(threads, -1);
    std::vector<char> bad(threads, 0);
    // (the buffers before the threads: a failed allocation is caught here and the caller
    // falls back to the sequential reader)
    for (int t = 0; t < threads; ++t) {
        const long i0 = nrows * t / threads, i1 = nrows * (t + 1) / threads;
        const long bytes = rows[i1 - 1].second - rows[i0].first;
        piece[t].resize(static_cast<size_t>(bytes / 2 + 2));
    }
    {
        std::vector<std::thread> pool;
        JoinAll guard{pool};
        pool.reserve(static_cast<size_t>(threads));
        for (int t = 0; t < threads; ++t) {
            const long i0 = nrows * t / threads, i1 = nrows * (t + 1) / threads;
            pool.emplace_back([&, t, i0, i1]() noexcept {
                Reader r{rows[i0].first, rows[i0].second, piece[t].data(),
                         static_cast<long>(piece[t].size())};
                for (int i = 0; i < 8; ++i) { r.seen[i] = false; r.shape[i] = 0; }
                for (long i = i0; i < i1; ++i) {
                    r.p = rows[i].first;
                    r.end = rows[i].second;
                    r.value(1);
                    if (r.fail || r.p != rows[i].second) { bad[t] = 1; return; }
                }
                counts[t] = r.count;
                cols[t] = r.seen[1] ? r.shape[1] : 0;
                if (r.max_depth > 2) bad[t] = 1;
            });
        }
    }
    long total = 0;
    for (int t = 0; t < threads; ++t) {
        if (bad[t] || cols[t] != cols[0]) return false;
        total += counts[t];
    }
    if (total <= cap) {
        float* dst = out;
        for (int t = 0; t < threads; ++t) {
            std::memcpy(dst, piece[t].data(), static_cast<size_t>(counts[t]) * sizeof(float));
            dst += counts[t];
        }
    }
    *count_out = total;
    *rows_out = nrows;
    *cols_out = cols[0];
    *after = p;
    return true;
} catch (...) {
    return false;  // no memory / no threads: the sequential reader takes over
}

}  // namespace

extern "C" {

// Upper bound on the encoded size: 16 chars per number + separators/brackets.
long nf_json_encode_bound(long numel, int ndim) {
    return numel * 20 + 4 * ndim + 16;
}

// Returns bytes written, or -needed when cap is too small.
long nf_json_encode_f32(const float* data, const long* shape, int ndim,
                        char* out, long cap) {
    const float* cursor = data;
    // unchecked writes when the text certainly fits: at most 17 bytes per number (15 of
    // digits, ", ") and 4 per list ("[", "]", ", ")
    long numel = 1, lists = 0;
    for (int i = 0; i < ndim; ++i) { lists += numel; numel *= shape[i]; }
    if (cap >= numel * 17 + lists * 4 + 8) {
        // worth a few threads from about a quarter of a million numbers on
        if (ndim >= 2 && numel >= (1L << 18) && shape[0] >= 8) {
            const int threads = codec_threads();
            if (threads > 1) {
                const long n = encode_parallel(out, data, shape, ndim, threads);
                if (n >= 0) return n;
            }
        }
        return encode_fast(out, cursor, shape, ndim) - out;
    }
    Writer w{out, cap};
    encode_rec(w, cursor, shape, ndim);
    return w.overflow ? -w.pos : w.pos;
}

// Parses one JSON value (number or nested list). On success returns the number
// of values (written to out up to cap; when the count exceeds cap the caller
// retries with a larger buffer), fills shape[0..ndim). Returns -1 on malformed
// or ragged input.
//
// consumed_out == nullptr: the value must span the whole text (trailing white
// space allowed). Otherwise the number of bytes the value occupied is stored
// there and whatever follows is left to the caller.
long nf_json_decode_f32(const char* text, long len, float* out, long cap,
                        long* shape_out, int* ndim_out, long* consumed_out) {
    // worth a few threads from a couple of megabytes on (up to parallel_decode_limit())
    if (len >= (1L << 21) && len <= parallel_decode_limit()) {
        const int threads = codec_threads();
        long count = 0, rows = 0, cols = 0;
        const char* after = nullptr;
        if (threads > 1 && decode_rows_parallel(text, text + len, out, cap, &count, &rows, &cols,
                                                &after, threads)) {
            if (consumed_out) {
                *consumed_out = after - text;
            } else {
                const char* q = after;
                while (q < text + len && (*q == ' ' || *q == '\n' || *q == '\t' || *q == '\r')) ++q;
                if (q != text + len) return -1;
            }
            *ndim_out = 2;
            shape_out[0] = rows;
            shape_out[1] = cols;
            return count;
        }
    }
    Reader r{text, text + len, out, cap};
    for (int i = 0; i < 8; ++i) { r.seen[i] = false; r.shape[i] = 0; }
    r.value(0);
    if (r.fail) return -1;
    if (consumed_out) {
        *consumed_out = r.p - text;
    } else {
        r.ws();
        if (r.p != r.end) return -1;
    }
    *ndim_out = r.max_depth;
    for (int i = 0; i < r.max_depth; ++i) shape_out[i] = r.shape[i];
    return r.count;
}

// Returns the offset one past the JSON value that starts at `pos` (after
// optional white space), or -1 if the text ends inside it. Strings, escapes and
// nesting are honoured; numbers/literals end at the first of ",]} \t\r\n".
long nf_json_skip_value(const char* text, long len, long pos) {
    const char* p = text + pos;
    const char* end = text + len;
    while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
    if (p >= end) return -1;
    if (*p == '"') {
        for (++p; p < end; ++p) {
            if (*p == '\\') { ++p; continue; }
            if (*p == '"') return (p + 1) - text;
        }
        return -1;
    }
    if (*p == '[' || *p == '{') {
        long depth = 0;
        bool in_str = false;
        for (; p < end; ++p) {
            const char c = *p;
            if (in_str) {
                if (c == '\\') { ++p; continue; }
                if (c == '"') in_str = false;
                continue;
            }
            if (c == '"') in_str = true;
            else if (c == '[' || c == '{') ++depth;
            else if (c == ']' || c == '}') {
                if (--depth == 0) return (p + 1) - text;
            }
        }
        return -1;
    }
    while (p < end && *p != ',' && *p != ']' && *p != '}' && *p != ' ' &&
           *p != '\n' && *p != '\t' && *p != '\r')
        ++p;
    return p - text;
}

}  // extern "C"
