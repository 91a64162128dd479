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
                if (!field(ln, sep, col_percent_modified, &fb, &fe) || !parse_number(fb, fe, &x, &is_int)) {
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
