// svx_host.hip -- host-side (CPU) pieces of the C ABI that sit in front of and behind the kernels when a
// process has to feed a GPU at hundreds of document pairs per second: the sampled row indices, the candidate
// index table of a document, and the text of an alignment file.  Plain C++; nothing here touches the device,
// and every function may be called from any thread (ctypes releases the GIL around them).
//
// Reference semantics (paths relative to the reference repository):
//   np.random.choice call order      svecalign/vecalign/dp_utils.py:301-302, 345-348  
//   make_overlap / make_doc_embedding svecalign/utils/embedding_utils.py:106-132, 135-203
//   read_in_embeddings (key -> first row) svecalign/utils/embedding_utils.py:79-103
//   load_ignore_index_file           svecalign/vecalign/vecalign.py:187-195
//   print_alignments                 svecalign/vecalign/vecalign.py:174-184
#include <errno.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include <stdint.h>

#include "../../include/svx.h"  // (no HIP header: this file also builds with a host compiler, tests/test_host_sanitized.py)

namespace {

// ---- MT19937, numpy's legacy generator (numpy/random/src/mt19937/mt19937.c: the reference's np.random.choice
// draws from RandomState, whose stream is frozen by NEP 19) -------------------------------------------------
constexpr int MT_N = 624, MT_M = 397;

struct Mt {
    uint32_t* key;
    int pos;
    inline void regen() {
        int i;
        uint32_t y;
        for (i = 0; i < MT_N - MT_M; i++) {
            y = (key[i] & 0x80000000u) | (key[i + 1] & 0x7fffffffu);
            key[i] = key[i + MT_M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
        }
        for (; i < MT_N - 1; i++) {
            y = (key[i] & 0x80000000u) | (key[i + 1] & 0x7fffffffu);
            key[i] = key[i + (MT_M - MT_N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
        }
        y = (key[MT_N - 1] & 0x80000000u) | (key[0] & 0x7fffffffu);
        key[MT_N - 1] = key[MT_M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
        pos = 0;
    }
    inline uint32_t next() {
        if (pos >= MT_N) regen();
        uint32_t y = key[pos++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
};

// RandomState.choice(n, size=cnt, replace=True) == randint(0, n, cnt): masked rejection on 32-bit outputs
// (numpy/random/src/distributions/distributions.c: random_bounded_uint64_fill, use_masked, rng = n - 1 <= 2^32 - 1;
// rng == 0 consumes nothing).
inline void choice_fill(Mt& mt, int64_t n, int64_t cnt, int32_t* out) {
    const uint32_t rng = (uint32_t)(n - 1);
    if (rng == 0) {
        for (int64_t i = 0; i < cnt; i++) out[i] = 0;
        return;
    }
    if (rng == 0xffffffffu) {
        for (int64_t i = 0; i < cnt; i++) out[i] = (int32_t)mt.next();
        return;
    }
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    for (int64_t i = 0; i < cnt; i++) {
        uint32_t v;
        do { v = mt.next() & mask; } while (v > rng);
        out[i] = (int32_t)v;
    }
}

inline bool is_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f); }

std::string_view strip(std::string_view s) {
    size_t a = 0, b = s.size();
    while (a < b && is_space((unsigned char)s[a])) a++;
    while (b > a && is_space((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}

bool read_file(const char* path, std::string* out) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    std::string buf;
    char tmp[1 << 16];
    size_t n;
    while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.append(tmp, n);
    fclose(f);
    out->swap(buf);
    return true;
}

// lines as Python's text-mode iteration yields them ('\n', '\r\n' and '\r' all end a line)
void split_lines(const std::string& buf, std::vector<std::string_view>* lines) {
    size_t i = 0, n = buf.size();
    while (i < n) {
        size_t j = i;
        while (j < n && buf[j] != '\n' && buf[j] != '\r') j++;
        lines->emplace_back(buf.data() + i, j - i);
        if (j < n && buf[j] == '\r' && j + 1 < n && buf[j + 1] == '\n') j++;
        i = j + 1;
    }
}

// first two whitespace-separated tokens of a line (str.split()[0], [1])
bool two_tokens(std::string_view s, std::string_view* t0, std::string_view* t1) {
    size_t i = 0, n = s.size();
    while (i < n && is_space((unsigned char)s[i])) i++;
    size_t a = i;
    while (i < n && !is_space((unsigned char)s[i])) i++;
    if (i == a) return false;
    *t0 = s.substr(a, i - a);
    while (i < n && is_space((unsigned char)s[i])) i++;
    a = i;
    while (i < n && !is_space((unsigned char)s[i])) i++;
    if (i == a) return false;
    *t1 = s.substr(a, i - a);
    return true;
}

}  // namespace

extern "C" {

int svx_num_levels(int n, int m, int max_size_full_dp) {
    long long s0 = n, s1 = m, lim = (long long)max_size_full_dp * max_size_full_dp;
    int depth = 0;
    while (s0 * s1 > lim) {
        depth++;
        s0 /= 2;
        s1 /= 2;
    }
    return depth;
}

int64_t svx_knob_count(int n_l, int m_l, int costs_sample_size) {
    long long p = (long long)n_l * m_l;
    return p < costs_sample_size ? p : costs_sample_size;
}


int svx_mt19937_choice(uint32_t* key, int32_t* pos, int64_t n, int64_t size, int32_t* out) {
    if (!key || !pos || !out || n < 1 || n > 0xffffffffLL || size < 0 || *pos < 0 || *pos > MT_N) return SVX_ERR_ARG;
    Mt mt{key, *pos};
    choice_fill(mt, n, size, out);
    *pos = mt.pos;
    return SVX_OK;
}

int64_t svx_norm_index_count(int n, int m, int k0, int k1, int max_size_full_dp, int num_samps_for_norm, int have_norms0,
                             int have_norms1) {
    const int L = svx_num_levels(n, m, max_size_full_dp);
    int64_t tot = 0;
    for (int l = 0; l <= L; l++) {
        const int s0 = n >> l, s1 = m >> l;
        const int spo1 = k1 > 0 ? (num_samps_for_norm + k1 - 1) / k1 : 0, spo0 = k0 > 0 ? (num_samps_for_norm + k0 - 1) / k0 : 0;
        if (!(l == 0 && have_norms0) && s1 > 0 && spo1 > 0) tot += (int64_t)k1 * spo1;
        if (!(l == 0 && have_norms1) && s0 > 0 && spo0 > 0) tot += (int64_t)k0 * spo0;
    }
    return tot;
}

int64_t svx_knob_index_count(int n, int m, int max_size_full_dp, int costs_sample_size) {
    const int L = svx_num_levels(n, m, max_size_full_dp);
    int64_t tot = 0;
    for (int l = 0; l <= L; l++) tot += 2 * svx_knob_count(n >> l, m >> l, costs_sample_size);
    return tot;
}

int svx_draw_indices(uint32_t* key, int32_t* pos, int n, int m, int k0, int k1, int max_size_full_dp, int costs_sample_size,
                     int num_samps_for_norm, int have_norms0, int have_norms1, int32_t* norm_idx, int32_t* knob_idx) {
    if (!key || !pos || !knob_idx || n < 1 || m < 1 || k0 < 1 || k1 < 1 || *pos < 0 || *pos > MT_N) return SVX_ERR_ARG;
    if (max_size_full_dp < 1 || costs_sample_size < 1 || num_samps_for_norm < 0) return SVX_ERR_ARG;
    Mt mt{key, *pos};
    const int L = svx_num_levels(n, m, max_size_full_dp);
    const int spo1 = (num_samps_for_norm + k1 - 1) / k1, spo0 = (num_samps_for_norm + k0 - 1) / k0;
    int32_t* o = norm_idx;
    for (int l = 0; l <= L; l++) {  // dp_utils.py:423-444: compute_norms(v0, v1) samples side 1, then compute_norms(v1, v0) side 0
        const int s0 = n >> l, s1 = m >> l;
        if (!(l == 0 && have_norms0) && s1 > 0 && spo1 > 0) {
            if (!o) return SVX_ERR_ARG;
            for (int k = 0; k < k1; k++) { choice_fill(mt, s1, spo1, o); o += spo1; }
        }
        if (!(l == 0 && have_norms1) && s0 > 0 && spo0 > 0) {
            if (!o) return SVX_ERR_ARG;
            for (int k = 0; k < k0; k++) { choice_fill(mt, s0, spo0, o); o += spo0; }
        }
    }
    o = knob_idx;
    for (int l = 0; l <= L; l++) {  // dp_utils.py:450-456 -> make_del_knob :286-302
        const int s0 = n >> l, s1 = m >> l;
        if ((int64_t)s0 * s1 < costs_sample_size) {
            const int64_t c = (int64_t)s0 * s1;
            for (int64_t i = 0; i < c; i++) { o[i] = (int32_t)(i / s1); o[c + i] = (int32_t)(i % s1); }
            o += 2 * c;
        } else {
            choice_fill(mt, s0, costs_sample_size, o);
            choice_fill(mt, s1, costs_sample_size, o + costs_sample_size);
            o += 2 * (int64_t)costs_sample_size;
        }
    }
    *pos = mt.pos;
    return SVX_OK;
}

int svx_candidate_table(const char* seg_path, const char* cat_path, const char* ignore_path, int max_overlaps, int32_t* table,
                        int cap_lines, int32_t* n_lines, int64_t* n_candidates, char* err, int err_cap) {
    auto fail = [&](int code, const char* fmt, const char* a) {
        if (err && err_cap > 0) snprintf(err, (size_t)err_cap, fmt, a);
        return code;
    };
    if (!seg_path || !cat_path || !n_lines || max_overlaps < 1) return fail(SVX_ERR_ARG, "svx_candidate_table: %s", "bad argument");
    std::string segbuf, catbuf, ignbuf;
    if (!read_file(seg_path, &segbuf)) return fail(SVX_ERR_ARG, "cannot read %s", seg_path);
    if (!read_file(cat_path, &catbuf)) return fail(SVX_ERR_ARG, "cannot read %s", cat_path);
    std::vector<std::string_view> seg, cat;
    split_lines(segbuf, &seg);
    split_lines(catbuf, &cat);
    const int n = (int)seg.size();
    *n_lines = n;
    if (n_candidates) *n_candidates = (int64_t)cat.size();
    if (!table) return SVX_OK;  // size query
    if (n > cap_lines) return fail(SVX_ERR_ARG, "%s: more lines than the table holds", seg_path);
    // candidate line -> first row (embedding_utils.py:93-99: duplicates keep the first)
    std::unordered_map<std::string_view, int32_t> first;
    first.reserve(cat.size() * 2);
    for (size_t i = 0; i < cat.size(); i++) first.emplace(strip(cat[i]), (int32_t)i);
    // segment lines: start token, end token (embedding_utils.py:30-35 preprocess_line, :128 split()[0] / split()[1])
    std::vector<std::string_view> st(n), en(n);
    static const char kBlank[] = "[BLANK_LINE]";
    for (int i = 0; i < n; i++) {
        std::string_view l = strip(seg[i]);
        if (l.empty()) l = std::string_view(kBlank);
        if (!two_tokens(l, &st[i], &en[i])) return fail(SVX_ERR_ARG, "%s: a line does not hold a start and an end", seg_path);
    }
    // ignore entries (start_id, j): everything from that overlap on is padded (embedding_utils.py:123-126)
    std::vector<int32_t> stop(n, 0x7fffffff);  // smallest ignored j per start
    if (ignore_path && ignore_path[0]) {
        if (!read_file(ignore_path, &ignbuf)) return fail(SVX_ERR_ARG, "cannot read %s", ignore_path);
        std::vector<std::string_view> ign;
        split_lines(ignbuf, &ign);
        for (auto l : ign) {
            std::string_view a, b;
            if (!two_tokens(strip(l), &a, &b)) return fail(SVX_ERR_ARG, "%s: expected 'i j' lines", ignore_path);
            const long i = strtol(std::string(a).c_str(), nullptr, 10), j = strtol(std::string(b).c_str(), nullptr, 10);
            // only entries the loop can reach matter: start <= j < start + max_overlaps
            if (i >= 0 && i < n && j >= i && j < i + max_overlaps && j < stop[i]) stop[i] = (int32_t)j;
        }
    }
    for (int64_t e = 0; e < (int64_t)max_overlaps * cap_lines; e++) table[e] = -1;
    std::string key;
    for (int i = 0; i < n; i++) {
        for (int o = 0; o < max_overlaps; o++) {
            const int j = i + o;
            if (j >= n || j >= stop[i]) break;
            key.assign(st[i].data(), st[i].size());
            key.push_back(' ');
            key.append(en[j].data(), en[j].size());
            auto it = first.find(std::string_view(key));
            table[(int64_t)o * cap_lines + j] = it == first.end() ? -1 : it->second;
        }
    }
    return SVX_OK;
}

int64_t svx_format_alignments(const int32_t* rows, const double* scores, int64_t n, char* out, int64_t cap) {
    if ((!rows && n > 0) || n < 0 || (!out && cap > 0)) return -1;
    int64_t w = 0;
    auto put = [&](const char* s, int64_t len) {
        if (out && w + len <= cap) memcpy(out + w, s, (size_t)len);
        w += len;
    };
    char num[352];  // "%.6f" of the largest double has 309 digits before the point (found by tests/test_host_sanitized.py: 64 was too few)
    auto list = [&](int start, int len) {
        put("[", 1);
        for (int i = 0; i < len; i++) {
            const int c = snprintf(num, sizeof(num), i ? ", %d" : "%d", start + i);
            put(num, c);
        }
        put("]", 1);
    };
    for (int64_t i = 0; i < n; i++) {
        const int32_t* r = rows + 4 * i;
        list(r[0], r[1]);
        put(":", 1);
        list(r[2], r[3]);
        if (scores) {
            const int c = snprintf(num, sizeof(num), ":%.6f", scores[i]);
            put(num, c);
        }
        put("\n", 1);
    }
    return w;  // bytes needed (written when they fit)
}

}  // extern "C"
