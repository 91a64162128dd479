// Host-side columnar reader for the sample tables (demethify/demethify.py:103-143 reads every
// sample file with pandas.read_csv and column_stacks two of its columns).  This reader memory-maps the file, finds the
// two columns the solver needs by header name and parses them, multi-threaded, STRAIGHT INTO the caller's (N x S)
// row-major matrices (column s, row stride S) -- typically page-locked host memory (dmf_host_alloc), from which the
// upload to HBM runs at PCIe speed without a staging copy.
//
// Bit-identical to the pandas path or not at all: the float conversion repeats the arithmetic of pandas' default C
// parser (pandas/_libs/src/parser/tokenizer.c, `precise_xstrtod`: at most 17 digits accumulated in a double, then ONE
// scaling by a tabulated power of ten -- not the correctly rounded strtod), and anything that parser treats specially (quotes,
// NA spellings, empty fields, non-integer coverage, thousands separators ...) makes the call return
// DMF_ERR_UNSUPPORTED so that the caller falls back to pandas itself (demethify_amd/tables.py).
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <charconv>
#include <cmath>
#include <string>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/demethify_hip.h"

namespace {

struct Mapped {
    const char* p = nullptr;
    size_t n = 0;
    int fd = -1;
    bool open(const char* path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        n = (size_t)st.st_size;
        if (n == 0) return true;
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return false;
        p = (const char*)m;
        madvise((void*)p, n, MADV_SEQUENTIAL);
        return true;
    }
    ~Mapped() {
        if (p) munmap((void*)p, n);
        if (fd >= 0) ::close(fd);
    }
};

inline bool is_digit(char c) { return c >= '0' && c <= '9'; }

// pandas tokenizer.c `precise_xstrtod` (the C parser's default "high" converter since pandas 1.2; decimal '.', sci 'E',
// no thousands separator) on the field [b, e): at most 17 digits are accumulated in a double (leading zeros count),
// further integer digits raise the exponent, further decimals are dropped, and the result is scaled by ONE
// multiplication or division with a tabulated power of ten.  Returns the value and "is a plain integer token";
// false if the field is not a complete plain number.
const double kPow10[] = {
    1e0,   1e1,   1e2,   1e3,   1e4,   1e5,   1e6,   1e7,   1e8,   1e9,   1e10,  1e11,  1e12,  1e13,  1e14,  1e15,  1e16,
    1e17,  1e18,  1e19,  1e20,  1e21,  1e22,  1e23,  1e24,  1e25,  1e26,  1e27,  1e28,  1e29,  1e30,  1e31,  1e32,  1e33,
    1e34,  1e35,  1e36,  1e37,  1e38,  1e39,  1e40};

bool parse_number(const char* b, const char* e, double* out, bool* is_int) {
    constexpr int max_digits = 17;
    const char* p = b;
    while (p < e && (*p == ' ' || *p == '\t')) ++p;
    bool negative = false;
    if (p < e && (*p == '-' || *p == '+')) {
        negative = *p == '-';
        ++p;
    }
    int exponent = 0, num_digits = 0, num_decimals = 0;
    double number = 0.0;
    bool integer = true;
    while (p < e && is_digit(*p)) {
        if (num_digits < max_digits) {
            number = number * 10. + (*p - '0');
            ++num_digits;
        } else {
            ++exponent;
        }
        ++p;
    }
    if (p < e && *p == '.') {
        integer = false;
        ++p;
        while (num_digits < max_digits && p < e && is_digit(*p)) {
            number = number * 10. + (*p - '0');
            ++p;
            ++num_digits;
            ++num_decimals;
        }
        if (num_digits >= max_digits)
            while (p < e && is_digit(*p)) ++p;  // extra decimal digits are consumed and ignored
        exponent -= num_decimals;
    }
    if (num_digits == 0) return false;
    if (negative) number = -number;
    if (p < e && (*p == 'e' || *p == 'E')) {
        integer = false;
        ++p;
        bool eneg = false;
        if (p < e && (*p == '-' || *p == '+')) {
            eneg = *p == '-';
            ++p;
        }
        int n = 0, nd = 0;
        while (p < e && is_digit(*p)) {
            n = n * 10 + (*p - '0');
            if (n > 1000) return false;
            ++p;
            ++nd;
        }
        if (nd == 0) return false;
        exponent += eneg ? -n : n;
    }
    while (p < e && (*p == ' ' || *p == '\t')) ++p;
    if (p != e) return false;
    if (exponent > 40 || exponent < -40) return false;  // beyond any methylation table: pandas' own code path
    if (exponent > 0) number *= kPow10[exponent];
    else if (exponent < 0) number /= kPow10[-exponent];
    *out = number;
    *is_int = integer;
    return true;
}

// [line_begin, line_end) without the trailing '\r'; fields split at `sep`; returns false on a quote character
struct Line {
    const char* b;
    const char* e;
};

inline const char* next_line(const char* p, const char* end, Line* ln) {
    const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
    const char* le = nl ? nl : end;
    ln->b = p;
    ln->e = (le > p && le[-1] == '\r') ? le - 1 : le;
    return nl ? nl + 1 : end;
}

inline bool field(const Line& ln, char sep, int index, const char** fb, const char** fe) {
    const char* p = ln.b;
    for (int i = 0; i < index; ++i) {
        const char* s = (const char*)memchr(p, sep, (size_t)(ln.e - p));
        if (!s) return false;
        p = s + 1;
    }
    const char* s = (const char*)memchr(p, sep, (size_t)(ln.e - p));
    *fb = p;
    *fe = s ? s : ln.e;
    return true;
}

// fields of a line (separators + 1)
inline int count_fields(const Line& ln, char sep) {
    int n = 1;
    for (const char* p = ln.b; (p = (const char*)memchr(p, sep, (size_t)(ln.e - p))) != nullptr; ++p) ++n;
    return n;
}

}  // namespace

extern "C" {

int dmf_table_scan(const char* path, char sep, int64_t* n_rows, int* col_percent_modified, int* col_valid_coverage,
                   int* n_cols) {
    if (!path || !n_rows || !col_percent_modified || !col_valid_coverage || !n_cols) return DMF_ERR_BAD_ARG;
    Mapped m;
    if (!m.open(path)) return DMF_ERR_BAD_ARG;
    if (m.n == 0) return DMF_ERR_UNSUPPORTED;
    if (memchr(m.p, '"', m.n) != nullptr) return DMF_ERR_UNSUPPORTED;  // quoting: pandas' business
    const char* end = m.p + m.n;
    Line hdr;
    const char* p = next_line(m.p, end, &hdr);
    *col_percent_modified = *col_valid_coverage = -1;
    int c = 0;
    for (const char* f = hdr.b;; ++c) {
        const char* s = (const char*)memchr(f, sep, (size_t)(hdr.e - f));
        const char* fe = s ? s : hdr.e;
        const size_t len = (size_t)(fe - f);
        if (len == 16 && memcmp(f, "percent_modified", 16) == 0 && *col_percent_modified < 0) *col_percent_modified = c;
        if (len == 14 && memcmp(f, "valid_coverage", 14) == 0 && *col_valid_coverage < 0) *col_valid_coverage = c;
        if (!s) break;
        f = s + 1;
    }
    *n_cols = c + 1;
    int64_t rows = 0;
    while (p < end) {
        Line ln;
        p = next_line(p, end, &ln);
        if (ln.e == ln.b) return DMF_ERR_UNSUPPORTED;  // blank lines: pandas skips them, keep that logic in one place
        // A data row with another field count than the header: pandas has rules of its own for that (one extra field in
        // every row -- a trailing separator -- makes the first column an implicit index and shifts the names; short rows
        // are filled with NaN).  Checked on the first row here and on every row by dmf_table_read.
        if (rows == 0 && count_fields(ln, sep) != *n_cols) return DMF_ERR_UNSUPPORTED;
        ++rows;
    }
    *n_rows = rows;
    return DMF_OK;
}

int dmf_table_read(const char* path, char sep, int col_percent_modified, int col_valid_coverage, int64_t n_rows,
                   double* out_freq, int64_t stride_freq, double divide_by, int64_t* out_cov, int64_t stride_cov,
                   int n_threads) {
    if (!path || !out_freq || col_percent_modified < 0 || n_rows < 0 || divide_by == 0.0) return DMF_ERR_BAD_ARG;
    if (col_valid_coverage >= 0 && !out_cov) return DMF_ERR_BAD_ARG;
    Mapped m;
    if (!m.open(path)) return DMF_ERR_BAD_ARG;
    if (m.n == 0) return DMF_ERR_UNSUPPORTED;
    const char* end = m.p + m.n;
    Line hdr;
    const char* body = next_line(m.p, end, &hdr);
    const int n_cols = count_fields(hdr, sep);
    if (n_threads < 1) n_threads = 1;
    if ((size_t)(end - body) < (size_t)n_threads * 65536) n_threads = 1;
    // chunk boundaries at line starts
    std::vector<const char*> cut(n_threads + 1);
    cut[0] = body;
    cut[n_threads] = end;
    for (int t = 1; t < n_threads; ++t) {
        const char* guess = body + (size_t)(end - body) * t / n_threads;
        const char* nl = (const char*)memchr(guess, '\n', (size_t)(end - guess));
        cut[t] = nl ? nl + 1 : end;
        if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
    }
    std::vector<int64_t> first(n_threads + 1, 0);
    {
        std::vector<std::thread> th;
        std::vector<int64_t> cnt(n_threads, 0);
        for (int t = 0; t < n_threads; ++t)
            th.emplace_back([&, t] {
                int64_t c = 0;
                for (const char* p = cut[t]; p < cut[t + 1];) {
                    const char* nl = (const char*)memchr(p, '\n', (size_t)(cut[t + 1] - p));
                    ++c;
                    p = nl ? nl + 1 : cut[t + 1];
                }
                cnt[t] = c;
            });
        for (auto& x : th) x.join();
        for (int t = 0; t < n_threads; ++t) first[t + 1] = first[t] + cnt[t];
    }
    if (first[n_threads] != n_rows) return DMF_ERR_BAD_SHAPE;
    std::atomic<int> status{DMF_OK};
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t)
        th.emplace_back([&, t] {
            int64_t row = first[t];
            for (const char* p = cut[t]; p < cut[t + 1] && status.load(std::memory_order_relaxed) == DMF_OK; ++row) {
                Line ln;
                p = next_line(p, cut[t + 1], &ln);
                const char *fb, *fe;
                double x;
                bool is_int;
                if (count_fields(ln, sep) != n_cols ||  // (ragged rows: pandas' rules, see dmf_table_scan)
                    !field(ln, sep, col_percent_modified, &fb, &fe) || !parse_number(fb, fe, &x, &is_int)) {
                    status = DMF_ERR_UNSUPPORTED;
                    return;
                }
                out_freq[row * stride_freq] = divide_by == 1.0 ? x : x / divide_by;
                if (col_valid_coverage >= 0) {
                    if (!field(ln, sep, col_valid_coverage, &fb, &fe) || !parse_number(fb, fe, &x, &is_int) || !is_int ||
                        x > 9e15 || x < -9e15) {
                        status = DMF_ERR_UNSUPPORTED;  // NA or fractional coverage: pandas decides the dtype
                        return;
                    }
                    out_cov[row * stride_cov] = (int64_t)x;
                }
            }
        });
    for (auto& x : th) x.join();
    return status.load();
}

}  // extern "C"

namespace {

// repr(float) of CPython / numpy (shortest digits that round-trip; fixed notation while -4 < decimal exponent <= 16,
// else d.ddde+XX with at least two exponent digits; a trailing ".0" on integers).  Appends to `out`.
void append_py_repr(std::string& out, double x) {
    if (std::isnan(x)) {
        out += "nan";
        return;
    }
    if (std::isinf(x)) {
        out += x < 0 ? "-inf" : "inf";
        return;
    }
    if (std::signbit(x)) {
        out += '-';
        x = -x;
    }
    if (x == 0.0) {
        out += "0.0";
        return;
    }
    char buf[48];
    const auto r = std::to_chars(buf, buf + sizeof(buf), x, std::chars_format::scientific);  // shortest round trip
    // buf = d[.ddd]e[+-]XX
    char digits[24];
    int nd = 0;
    const char* p = buf;
    for (; p < r.ptr && *p != 'e'; ++p)
        if (*p != '.') digits[nd++] = *p;
    int e10 = 0;
    {
        ++p;  // 'e'
        const bool neg = *p == '-';
        ++p;
        for (; p < r.ptr; ++p) e10 = e10 * 10 + (*p - '0');
        if (neg) e10 = -e10;
    }
    const int decpt = e10 + 1;  // value = 0.d1d2... x 10^decpt
    if (decpt > -4 && decpt <= 16) {
        if (decpt <= 0) {
            out += "0.";
            out.append((size_t)(-decpt), '0');
            out.append(digits, (size_t)nd);
        } else if (decpt >= nd) {
            out.append(digits, (size_t)nd);
            out.append((size_t)(decpt - nd), '0');
            out += ".0";
        } else {
            out.append(digits, (size_t)decpt);
            out += '.';
            out.append(digits + decpt, (size_t)(nd - decpt));
        }
    } else {
        out += digits[0];
        if (nd > 1) {
            out += '.';
            out.append(digits + 1, (size_t)(nd - 1));
        }
        out += 'e';
        int e = decpt - 1;
        out += e < 0 ? '-' : '+';
        if (e < 0) e = -e;
        char eb[8];
        int ne = 0;
        do {
            eb[ne++] = (char)('0' + e % 10);
            e /= 10;
        } while (e > 0);
        if (ne < 2) eb[ne++] = '0';
        while (ne > 0) out += eb[--ne];
    }
}

}  // namespace

extern "C" {

/* The confidence-interval table of the profile estimates as the reference writes it (demethify/bootstrap.py:85-91: a
 * DataFrame whose cells are (lower, upper) tuples, DataFrame.to_csv): header line, then per CpG row the n_cols cells
 * "(lo, hi)" -- quoted, because of the comma -- with the floats as repr() prints them.  numpy_scalar_repr != 0 writes
 * the cells as numpy >= 2 prints a tuple of float64 scalars, "(np.float64(lo), np.float64(hi))".  Byte-identical to the
 * pandas output (tests/test_host.py); 1e6 rows take a fraction of a second instead of ~20 s. */
int dmf_write_interval_csv(const char* path, const char* header_line, const double* lower, const double* upper,
                           int64_t n_rows, int n_cols, int numpy_scalar_repr, int n_threads) {
    if (!path || !header_line || !lower || !upper || n_rows < 0 || n_cols < 1) return DMF_ERR_BAD_ARG;
    FILE* f = fopen(path, "wb");
    if (!f) return DMF_ERR_BAD_ARG;
    bool ok = fputs(header_line, f) >= 0 && fputc('\n', f) != EOF;
    if (n_threads < 1) n_threads = 1;
    const int64_t kBatch = 65536;  // rows per thread and round
    const char* open_lo = numpy_scalar_repr ? "\"(np.float64(" : "\"(";
    const char* mid = numpy_scalar_repr ? "), np.float64(" : ", ";
    const char* close_hi = numpy_scalar_repr ? "))\"" : ")\"";
    std::vector<std::string> bufs((size_t)n_threads);
    for (int64_t r0 = 0; r0 < n_rows && ok; r0 += kBatch * n_threads) {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t)
            th.emplace_back([&, t] {
                std::string& out = bufs[(size_t)t];
                out.clear();
                const int64_t a = r0 + t * kBatch, b = a + kBatch < n_rows ? a + kBatch : n_rows;
                for (int64_t i = a; i < b; ++i) {
                    for (int c = 0; c < n_cols; ++c) {
                        if (c) out += ',';
                        out += open_lo;
                        append_py_repr(out, lower[i * n_cols + c]);
                        out += mid;
                        append_py_repr(out, upper[i * n_cols + c]);
                        out += close_hi;
                    }
                    out += '\n';
                }
            });
        for (auto& x : th) x.join();
        for (int t = 0; t < n_threads && ok; ++t)
            if (!bufs[(size_t)t].empty()) ok = fwrite(bufs[(size_t)t].data(), 1, bufs[(size_t)t].size(), f) == bufs[(size_t)t].size();
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? DMF_OK : DMF_ERR_BAD_ARG;
}

}  // extern "C"

extern "C" {
/* Page-locked host memory for the input matrices (falls back to plain memory without a GPU runtime). */
void* dmf_host_alloc(size_t bytes, int* pinned) {
    void* p = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) == hipSuccess && count > 0 && hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess) {
        if (pinned) *pinned = 1;
        return p;
    }
    (void)hipGetLastError();
    if (pinned) *pinned = 0;
    return malloc(bytes);
}

void dmf_host_free(void* p, int pinned) {
    if (!p) return;
    if (pinned) (void)hipHostFree(p);
    else free(p);
}

}  // extern "C"
