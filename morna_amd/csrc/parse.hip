// parse.hip -- native host-side pre-pass of `morna index` (no GPU work here).
//
// Counterpart of the reference's go_index line loop, count_samples and the host
// half of MornaIndex.add_junction (commanderson/morna morna.py:841-861, 789-822,
// 357-382): reads a (gzipped) intropolis file and produces exactly the CSR arrays
// morna_stage_junctions() takes -- kept lines in file order, first-seen internal
// ids, cumulative junction frequencies, idf = log(sample_count / freq) from libm.
// The Python tokenising loop is the reference's real end-to-end cost (SURVEY.md
// section 8f N1); this does the same work at I/O speed.
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "common.hpp"

struct morna_lines {
    std::vector<uint8_t> key_bytes;
    std::vector<int64_t> key_off{0}, row_ptr{0};
    std::vector<int32_t> item_ids, cov;
    std::vector<double> idf;
    std::vector<int64_t> ext_ids;                       // external sample id of each internal id
    std::unordered_map<std::string, int64_t> freq;      // sample_frequencies (morna.py:365)
    std::vector<std::string> freq_keys;                 // insertion order, for the accessor
    std::vector<int64_t> freq_vals;
    int64_t sample_count = 0, skipped = 0, lines_read = 0;
};

namespace {

using morna::set_error;

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

// Python's line.strip(): [b, e) without leading / trailing whitespace
inline void strip(const std::string &s, size_t &b, size_t &e)
{
    b = 0;
    e = s.size();
    while (b < e && is_space(s[b])) b++;
    while (e > b && is_space(s[e - 1])) e--;
}

// int(token) for the decimal tokens of an intropolis file (optional sign, surrounding blanks)
inline bool parse_int(const char *p, const char *end, int64_t &out)
{
    while (p < end && is_space(*p)) p++;
    while (end > p && is_space(end[-1])) end--;
    if (p == end) return false;
    bool neg = false;
    if (*p == '+' || *p == '-') {
        neg = *p == '-';
        p++;
    }
    if (p == end) return false;
    int64_t v = 0;
    for (; p < end; p++) {
        if (*p < '0' || *p > '9') return false;
        v = v * 10 + (*p - '0');
    }
    out = neg ? -v : v;
    return true;
}

// positions of the tab-separated tokens of s[b, e)
inline void split_tabs(const std::string &s, size_t b, size_t e, std::vector<std::pair<size_t, size_t>> &tok)
{
    tok.clear();
    size_t start = b;
    for (size_t i = b; i <= e; i++)
        if (i == e || s[i] == '\t') {
            tok.emplace_back(start, i);
            start = i + 1;
        }
}

const char LINES_MAGIC[8] = {'M', 'O', 'R', 'N', 'A', 'L', 'N', '1'};   // cache blob, see morna_lines_save

template <typename T>
bool put_vec(FILE *f, const std::vector<T> &v)
{
    return v.empty() || fwrite(v.data(), sizeof(T), v.size(), f) == v.size();
}
template <typename T>
bool get_vec(FILE *f, std::vector<T> &v, int64_t n)
{
    if (n < 0) return false;
    v.resize((size_t)n);
    return n == 0 || fread(v.data(), sizeof(T), (size_t)n, f) == (size_t)n;
}

}  // namespace

extern "C" {

int morna_lines_free(morna_lines *L)
{
    delete L;
    return MORNA_OK;
}

// ---- the pipeline ---------------------------------------------------------------------------------------------
// One thread inflates the file into blocks cut at line ends; worker threads tokenise the blocks (the per-character work:
// tabs, commas, decimal numbers); the calling thread merges the blocks IN FILE ORDER, which is where everything that is
// sequential by definition happens -- the threshold, the cumulative frequency of a key and its idf, first-seen internal
// ids (morna.py:357-382).  gzip inflation cannot be spread over threads, so it is the floor of the wall-clock; the
// single-threaded pass it replaces spent as long again on the tokenising.  MORNA_PARSE_THREADS sets the workers (default:
// the cores available, at most 12; 1: everything on the calling thread, no thread is started).

struct ParsedLine {
    int32_t err;                 // 0 ok, 1 fewer than two columns, 2 invalid literal for int()
    uint32_t key_off, key_len;   // into ParsedBlock::keys
    int64_t first;               // into ParsedBlock::samples / covs
    int32_t n_samples, n_covs;
};
struct ParsedBlock {
    std::string text;            // the block's lines (whole lines only)
    int64_t first_line = 0;      // number of the block's first line, from 1
    std::vector<ParsedLine> lines;
    std::string keys;
    std::vector<int64_t> samples, covs;   // per line: n_samples values in samples[first ..], n_covs in covs[first2 ..]
    std::vector<int64_t> cov_first;       // start of each line's coverages in covs
    std::vector<std::string> ids_seen;    // count_samples mode: the sample-id strings of column -2
    std::atomic<bool> done{false};
};

// tokenise one block (what go_index does to each line before add_junction: morna.py:848-853)
static void parse_block(ParsedBlock &B, bool count_only)
{
    const std::string &s = B.text;
    std::vector<std::pair<size_t, size_t>> tok;
    size_t pos = 0;
    while (pos < s.size()) {
        size_t nl = s.find('\n', pos);
        const size_t end_line = nl == std::string::npos ? s.size() : nl;
        ParsedLine L;
        L.err = 0;
        L.key_off = (uint32_t)B.keys.size();
        L.key_len = 0;
        L.first = (int64_t)B.samples.size();
        L.n_samples = L.n_covs = 0;
        if (count_only) {
            // the reference splits the raw line here (no strip): column -2 can never hold the newline
            split_tabs(s, pos, end_line, tok);
            if (tok.size() < 2) {
                L.err = 1;
            } else {
                const auto &t = tok[tok.size() - 2];
                size_t st = t.first;
                for (size_t i = t.first; i <= t.second; i++)
                    if (i == t.second || s[i] == ',') {
                        B.ids_seen.emplace_back(s.data() + st, i - st);
                        st = i + 1;
                    }
            }
            B.lines.push_back(L);
            B.cov_first.push_back(0);
            pos = end_line + 1;
            continue;
        }
        size_t b = pos, e = end_line;   // tokens = line.strip().split('\t')
        while (b < e && is_space(s[b])) b++;
        while (e > b && is_space(s[e - 1])) e--;
        split_tabs(s, b, e, tok);
        B.cov_first.push_back((int64_t)B.covs.size());
        if (tok.size() < 2) {
            L.err = 1;
        } else {
            for (size_t i = 0; i < tok.size() && i < 3; i++) {   // ' '.join(tokens[:3])
                if (i) B.keys.push_back(' ');
                B.keys.append(s, tok[i].first, tok[i].second - tok[i].first);
            }
            L.key_len = (uint32_t)(B.keys.size() - L.key_off);
            auto parse_list = [&](const std::pair<size_t, size_t> &t, std::vector<int64_t> &dst, int32_t &n) {
                size_t st = t.first;
                for (size_t i = t.first; i <= t.second; i++)
                    if (i == t.second || s[i] == ',') {
                        int64_t v;
                        if (!parse_int(s.data() + st, s.data() + i, v)) return false;
                        dst.push_back(v);
                        n++;
                        st = i + 1;
                    }
                return true;
            };
            if (!parse_list(tok[tok.size() - 2], B.samples, L.n_samples) || !parse_list(tok[tok.size() - 1], B.covs, L.n_covs)) L.err = 2;
        }
        B.lines.push_back(L);
        pos = end_line + 1;
    }
    B.done = true;
}

// the reader + workers; next_block() hands the blocks over in file order
struct BlockPipeline {
    static constexpr size_t BLOCK_BYTES = (size_t)4 << 20;
    gzFile f = nullptr;
    bool count_only = false;
    int n_workers = 1;
    std::mutex mu;
    std::condition_variable cv_work, cv_done, cv_room;
    std::deque<std::unique_ptr<ParsedBlock>> blocks;   // in file order; front = next to merge
    size_t next_to_parse = 0;                          // index into blocks of the first block no worker has taken
    bool eof = false;
    std::vector<std::thread> threads;
    size_t max_in_flight = 32;

    bool read_block(std::string &carry, ParsedBlock &B, int64_t &line_no)
    {
        // whole lines only: what follows the last newline is carried into the next block
        std::string &t = B.text;
        t.swap(carry);
        carry.clear();
        B.first_line = line_no;
        for (;;) {
            const size_t old = t.size();
            t.resize(old + BLOCK_BYTES);
            const int got = gzread(f, &t[old], (unsigned)BLOCK_BYTES);
            t.resize(old + (got > 0 ? (size_t)got : 0));
            if (got <= 0) break;                                    // end of file (or a read error: what was read is used)
            const size_t nl = t.rfind('\n');
            if (nl != std::string::npos) {
                carry.assign(t, nl + 1, std::string::npos);
                t.resize(nl + 1);
                break;
            }
        }
        if (t.empty()) return false;
        for (char ch : t) line_no += ch == '\n';
        if (t.back() != '\n') line_no++;                            // the file's last line without a terminator
        return true;
    }

    void start(const char *path_unused, bool count, int workers)
    {
        (void)path_unused;
        count_only = count;
        n_workers = workers;
        if (n_workers <= 1) return;
        threads.emplace_back([this] {   // the reader: inflation
            std::string carry;
            int64_t line_no = 1;
            for (;;) {
                std::unique_ptr<ParsedBlock> B(new ParsedBlock());
                const bool any = read_block(carry, *B, line_no);
                std::unique_lock<std::mutex> lk(mu);
                if (!any) {
                    eof = true;
                    cv_work.notify_all();
                    cv_done.notify_all();
                    return;
                }
                cv_room.wait(lk, [this] { return blocks.size() < max_in_flight; });
                blocks.push_back(std::move(B));
                cv_work.notify_one();
            }
        });
        for (int w = 0; w < n_workers - 1; w++)
            threads.emplace_back([this] {   // the tokenisers
                for (;;) {
                    ParsedBlock *B = nullptr;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv_work.wait(lk, [this] { return next_to_parse < blocks.size() || eof; });
                        if (next_to_parse >= blocks.size()) return;   // eof and nothing left
                        B = blocks[next_to_parse++].get();
                    }
                    parse_block(*B, count_only);
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        cv_done.notify_all();
                    }
                }
            });
    }

    std::string carry1;
    int64_t line1 = 1;
    // the next block in file order, tokenised; null at the end of the file
    std::unique_ptr<ParsedBlock> next_block()
    {
        if (n_workers <= 1) {
            std::unique_ptr<ParsedBlock> B(new ParsedBlock());
            if (!read_block(carry1, *B, line1)) return nullptr;
            parse_block(*B, count_only);
            return B;
        }
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [this] { return (!blocks.empty() && blocks.front()->done) || (eof && blocks.empty()); });
        if (blocks.empty()) return nullptr;
        std::unique_ptr<ParsedBlock> B = std::move(blocks.front());
        blocks.pop_front();
        next_to_parse--;   // indices shift with the pop; the front block was taken by a worker long ago
        cv_room.notify_one();
        return B;
    }

    void stop()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            max_in_flight = (size_t)-1;   // let the reader run to the end if the caller bailed out early
            cv_room.notify_all();
        }
        for (std::thread &t : threads) t.join();
        threads.clear();
    }
};

static int parse_threads()
{
    if (const char *e = getenv("MORNA_PARSE_THREADS")) return std::max(1, atoi(e));
    const unsigned hc = std::thread::hardware_concurrency();
    return (int)std::min<unsigned>(std::max<unsigned>(hc, 1u), 12u);
}

int morna_parse_intropolis(const char *path, int64_t sample_count, int64_t sample_threshold, morna_lines **out)
{
    if (!path || !out) {
        set_error("parse_intropolis: null argument");
        return MORNA_E_INVALID;
    }
    *out = nullptr;
    const int workers = parse_threads();
    if (sample_count <= 0) {
        // count_samples (morna.py:789-822): distinct sample-id STRINGS of column -2
        BlockPipeline P;
        P.f = gzopen(path, "rb");   // transparently reads plain text too
        if (!P.f) {
            set_error("Unable to open %s", path);
            return MORNA_E_IO;
        }
        gzbuffer(P.f, 1 << 20);
        P.start(path, true, workers);
        std::unordered_set<std::string> seen;
        int rc = MORNA_OK;
        while (std::unique_ptr<ParsedBlock> B = P.next_block()) {
            if (rc != MORNA_OK) continue;   // drain
            for (size_t i = 0; i < B->lines.size(); i++)
                if (B->lines[i].err) {
                    set_error("line %lld of %s has fewer than two tab-separated columns", (long long)(B->first_line + (int64_t)i), path);
                    rc = MORNA_E_INVALID;
                    break;
                }
            if (rc == MORNA_OK)
                for (std::string &id : B->ids_seen) seen.insert(std::move(id));
        }
        P.stop();
        gzclose(P.f);
        if (rc != MORNA_OK) return rc;
        sample_count = (int64_t)seen.size();
    }
    BlockPipeline P;
    P.f = gzopen(path, "rb");
    if (!P.f) {
        set_error("Unable to open %s", path);
        return MORNA_E_IO;
    }
    gzbuffer(P.f, 1 << 20);
    P.start(path, false, workers);
    morna_lines *L = new morna_lines();
    L->sample_count = sample_count;
    // internal_id_map (morna.py:377-382): a table indexed by the sample id while the ids are small non-negative numbers
    // (1e8 look-ups are the merge's cost), a hash map for the others
    std::vector<int32_t> dense;
    std::unordered_map<int64_t, int32_t> sparse;
    constexpr int64_t DENSE_MAX = (int64_t)1 << 26;
    std::string key;
    int rc = MORNA_OK;
    int64_t lineno = 0;
    while (std::unique_ptr<ParsedBlock> B = P.next_block()) {
        if (rc != MORNA_OK) continue;   // an error further up the file: drain the pipeline
        for (size_t li = 0; li < B->lines.size() && rc == MORNA_OK; li++) {
            const ParsedLine &PL = B->lines[li];
            lineno = B->first_line + (int64_t)li;
            if (PL.err == 1) {
                set_error("line %lld of %s has fewer than two tab-separated columns", (long long)lineno, path);
                rc = MORNA_E_INVALID;
                break;
            }
            if (PL.err == 2) {
                set_error("invalid literal for int() on line %lld of %s", (long long)lineno, path);
                rc = MORNA_E_INVALID;
                break;
            }
            if ((int64_t)PL.n_samples < sample_threshold) {    // morna.py:361-363
                L->skipped++;
                continue;
            }
            key.assign(B->keys, PL.key_off, PL.key_len);
            auto it = L->freq.find(key);
            if (it == L->freq.end()) {
                it = L->freq.emplace(key, 0).first;
                L->freq_keys.push_back(key);
            }
            it->second += (int64_t)PL.n_samples;               // morna.py:365
            L->idf.push_back(log((double)sample_count / (double)it->second));   // morna.py:372-374
            L->key_bytes.insert(L->key_bytes.end(), key.begin(), key.end());
            L->key_off.push_back((int64_t)L->key_bytes.size());
            const int32_t n = PL.n_samples < PL.n_covs ? PL.n_samples : PL.n_covs;   // zip() truncates
            const int64_t *sm = B->samples.data() + PL.first, *cv = B->covs.data() + B->cov_first[li];
            const size_t at = L->item_ids.size();
            L->item_ids.resize(at + (size_t)n);
            L->cov.resize(at + (size_t)n);
            for (int32_t i = 0; i < n; i++) {
                const int64_t sid = sm[i];
                int32_t id;
                if (sid >= 0 && sid < DENSE_MAX) {
                    if ((size_t)sid >= dense.size()) dense.resize((size_t)std::max<int64_t>(sid + 1, (int64_t)dense.size() * 2), -1);
                    id = dense[(size_t)sid];
                    if (id < 0) {
                        id = dense[(size_t)sid] = (int32_t)L->ext_ids.size();
                        L->ext_ids.push_back(sid);
                    }
                } else {
                    auto f = sparse.find(sid);
                    if (f == sparse.end()) {
                        id = (int32_t)L->ext_ids.size();
                        sparse.emplace(sid, id);
                        L->ext_ids.push_back(sid);
                    } else {
                        id = f->second;
                    }
                }
                if (cv[i] > INT32_MAX || cv[i] < INT32_MIN) {   // the staged arrays are int32; the reference keeps Python ints
                    set_error("coverage %lld on line %lld of %s does not fit 32 bits", (long long)cv[i], (long long)lineno, path);
                    rc = MORNA_E_INVALID;
                    break;
                }
                L->item_ids[at + (size_t)i] = id;
                L->cov[at + (size_t)i] = (int32_t)cv[i];
            }
            if (rc == MORNA_OK) L->row_ptr.push_back((int64_t)L->item_ids.size());
        }
        if (rc == MORNA_OK) lineno = B->first_line + (int64_t)B->lines.size() - 1;
    }
    P.stop();
    gzclose(P.f);
    if (rc != MORNA_OK) {
        delete L;
        return rc;
    }
    L->lines_read = lineno;
    L->freq_vals.reserve(L->freq_keys.size());
    for (const std::string &k : L->freq_keys) L->freq_vals.push_back(L->freq[k]);
    *out = L;
    return MORNA_OK;
}

// counts[8] = {kept lines, nnz, n_items, skipped, sample_count, key bytes, distinct keys, lines read}
int morna_lines_counts(const morna_lines *L, int64_t *counts)
{
    if (!L || !counts) return MORNA_E_INVALID;
    counts[0] = (int64_t)L->idf.size();
    counts[1] = (int64_t)L->item_ids.size();
    counts[2] = (int64_t)L->ext_ids.size();
    counts[3] = L->skipped;
    counts[4] = L->sample_count;
    counts[5] = (int64_t)L->key_bytes.size();
    counts[6] = (int64_t)L->freq_keys.size();
    counts[7] = L->lines_read;
    return MORNA_OK;
}

// borrowed pointers, valid until morna_lines_free
int morna_lines_arrays(const morna_lines *L, const uint8_t **key_bytes, const int64_t **key_off,
                       const int64_t **row_ptr, const int32_t **item_ids, const int32_t **cov, const double **idf,
                       const int64_t **ext_ids)
{
    if (!L) return MORNA_E_INVALID;
    if (key_bytes) *key_bytes = L->key_bytes.data();
    if (key_off) *key_off = L->key_off.data();
    if (row_ptr) *row_ptr = L->row_ptr.data();
    if (item_ids) *item_ids = L->item_ids.data();
    if (cov) *cov = L->cov.data();
    if (idf) *idf = L->idf.data();
    if (ext_ids) *ext_ids = L->ext_ids.data();
    return MORNA_OK;
}

// i-th distinct junction key (first-seen order) and its final cumulative frequency
int morna_lines_freq_entry(const morna_lines *L, int64_t i, const char **key, int64_t *key_len, int64_t *freq)
{
    if (!L || i < 0 || i >= (int64_t)L->freq_keys.size()) return MORNA_E_RANGE;
    *key = L->freq_keys[(size_t)i].data();
    *key_len = (int64_t)L->freq_keys[(size_t)i].size();
    *freq = L->freq_vals[(size_t)i];
    return MORNA_OK;
}

// ---- binary pre-tokenised cache: the parsed arrays as one little-endian blob, so
// that a second `index` run over the same file (other n_trees / features / seed) skips inflate + tokenising.
//   "MORNALN1", tag[4] (caller's identity of the source: size, mtime, sample_count argument, threshold),
//   counts[8] as morna_lines_counts, then key_bytes, key_off, row_ptr, item_ids, cov, idf, ext_ids,
//   freq_vals, freq key lengths (int64 each), freq key bytes.
int morna_lines_save(const morna_lines *L, const char *path, const int64_t *tag)
{
    if (!L || !path || !tag) {
        set_error("lines_save: null argument");
        return MORNA_E_INVALID;
    }
    FILE *f = fopen(path, "wb");
    if (!f) {
        set_error("Unable to open %s for writing", path);
        return MORNA_E_IO;
    }
    int64_t counts[8];
    morna_lines_counts(L, counts);
    std::vector<int64_t> klen;
    std::vector<uint8_t> kbytes;
    for (const std::string &k : L->freq_keys) {
        klen.push_back((int64_t)k.size());
        kbytes.insert(kbytes.end(), k.begin(), k.end());
    }
    const int64_t kb = (int64_t)kbytes.size();
    bool ok = fwrite(LINES_MAGIC, 1, 8, f) == 8 && fwrite(tag, 8, 4, f) == 4 && fwrite(counts, 8, 8, f) == 8 &&
              fwrite(&kb, 8, 1, f) == 1 && put_vec(f, L->key_bytes) && put_vec(f, L->key_off) &&
              put_vec(f, L->row_ptr) && put_vec(f, L->item_ids) && put_vec(f, L->cov) && put_vec(f, L->idf) &&
              put_vec(f, L->ext_ids) && put_vec(f, L->freq_vals) && put_vec(f, klen) && put_vec(f, kbytes);
    ok = (fclose(f) == 0) && ok;
    if (!ok) {
        set_error("short write to %s", path);
        return MORNA_E_IO;
    }
    return MORNA_OK;
}

int morna_lines_load(const char *path, int64_t *tag_out, morna_lines **out)
{
    if (!path || !tag_out || !out) {
        set_error("lines_load: null argument");
        return MORNA_E_INVALID;
    }
    *out = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) {
        set_error("Unable to open %s", path);
        return MORNA_E_IO;
    }
    char magic[8];
    int64_t counts[8], kb = 0;
    morna_lines *L = new morna_lines();
    std::vector<int64_t> klen;
    std::vector<uint8_t> kbytes;
    bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, LINES_MAGIC, 8) == 0 && fread(tag_out, 8, 4, f) == 4 &&
              fread(counts, 8, 8, f) == 8 && fread(&kb, 8, 1, f) == 1;
    if (ok) {
        // the counts must account for the file's size exactly before anything is allocated from them
        for (int i = 0; i < 8; i++) ok = ok && counts[i] >= 0 && counts[i] < ((int64_t)1 << 56);
        ok = ok && kb >= 0 && kb < ((int64_t)1 << 56);
        const long here = ftell(f);
        ok = ok && fseek(f, 0, SEEK_END) == 0;
        const long size = ftell(f);
        ok = ok && fseek(f, here, SEEK_SET) == 0;
        const double want = (double)here + (double)counts[5] + 16.0 * (double)(counts[0] + 1) + 8.0 * (double)counts[1] +
                            8.0 * (double)counts[0] + 8.0 * (double)counts[2] + 16.0 * (double)counts[6] + (double)kb;
        ok = ok && (double)size == want;
    }
    ok = ok && get_vec(f, L->key_bytes, counts[5]) && get_vec(f, L->key_off, counts[0] + 1) &&
         get_vec(f, L->row_ptr, counts[0] + 1) && get_vec(f, L->item_ids, counts[1]) && get_vec(f, L->cov, counts[1]) &&
         get_vec(f, L->idf, counts[0]) && get_vec(f, L->ext_ids, counts[2]) && get_vec(f, L->freq_vals, counts[6]) &&
         get_vec(f, klen, counts[6]) && get_vec(f, kbytes, kb);
    // nothing may follow, and the offsets must describe the arrays that were read
    ok = ok && fgetc(f) == EOF && L->key_off.front() == 0 && L->key_off.back() == counts[5] &&
         L->row_ptr.front() == 0 && L->row_ptr.back() == counts[1];
    fclose(f);
    if (ok) {
        int64_t pos = 0;
        for (int64_t i = 0; ok && i < counts[6]; i++) {
            ok = klen[(size_t)i] >= 0 && pos + klen[(size_t)i] <= kb;
            if (ok) L->freq_keys.emplace_back((const char *)kbytes.data() + pos, (size_t)klen[(size_t)i]);
            pos += klen[(size_t)i];
        }
        ok = ok && pos == kb;
    }
    for (int64_t i = 0; ok && i < counts[0]; i++)
        ok = L->key_off[(size_t)i] <= L->key_off[(size_t)i + 1] && L->row_ptr[(size_t)i] <= L->row_ptr[(size_t)i + 1];
    for (int64_t i = 0; ok && i < counts[1]; i++) ok = L->item_ids[(size_t)i] >= 0 && L->item_ids[(size_t)i] < counts[2];
    if (!ok) {
        delete L;
        set_error("%s is not a pre-tokenised intropolis cache (or is truncated)", path);
        return MORNA_E_IO;
    }
    L->skipped = counts[3];
    L->sample_count = counts[4];
    L->lines_read = counts[7];
    *out = L;
    return MORNA_OK;
}

// Inverse of the parser, for benchmarks and round-trip tests (no counterpart in the reference): the given lines as an
// intropolis text file -- chrom, start, end (the key's three words), strand, donor, acceptor, sample list, coverage list,
// tab separated (tests/tiny_intropolis.tsv) -- gzipped when the path ends in ".gz" (level 1).
int morna_write_intropolis(const char *path, const uint8_t *key_bytes, const int64_t *key_off, int64_t J, const int64_t *row_ptr,
                           const int64_t *samples, const int32_t *cov)
{
    if (!path || J < 0 || (J > 0 && (!key_bytes || !key_off || !row_ptr || !samples || !cov))) {
        set_error("write_intropolis: null argument");
        return MORNA_E_INVALID;
    }
    const size_t plen = strlen(path);
    const bool gz = plen > 3 && strcmp(path + plen - 3, ".gz") == 0;
    gzFile zf = nullptr;
    FILE *pf = nullptr;
    if (gz) zf = gzopen(path, "wb1");
    else pf = fopen(path, "wb");
    if (!zf && !pf) {
        set_error("Unable to open %s for writing", path);
        return MORNA_E_IO;
    }
    if (zf) gzbuffer(zf, 1 << 20);
    std::string line;
    bool ok = true;
    char num[24];
    auto put_num = [&](long long v) {
        int n = 0;
        bool neg = v < 0;
        unsigned long long u = neg ? 0ull - (unsigned long long)v : (unsigned long long)v;
        do { num[n++] = (char)('0' + u % 10); u /= 10; } while (u);
        if (neg) line.push_back('-');
        while (n) line.push_back(num[--n]);
    };
    for (int64_t j = 0; ok && j < J; j++) {
        line.clear();
        for (int64_t i = key_off[j]; i < key_off[j + 1]; i++) line.push_back(key_bytes[i] == ' ' ? '\t' : (char)key_bytes[i]);
        line.append("\t+\tGT\tAG\t");
        for (int64_t t = row_ptr[j]; t < row_ptr[j + 1]; t++) {
            if (t > row_ptr[j]) line.push_back(',');
            put_num(samples[t]);
        }
        line.push_back('\t');
        for (int64_t t = row_ptr[j]; t < row_ptr[j + 1]; t++) {
            if (t > row_ptr[j]) line.push_back(',');
            put_num(cov[t]);
        }
        line.push_back('\n');
        if (zf) ok = gzwrite(zf, line.data(), (unsigned)line.size()) == (int)line.size();
        else ok = fwrite(line.data(), 1, line.size(), pf) == line.size();
    }
    if (zf) ok = (gzclose(zf) == Z_OK) && ok;
    if (pf) ok = (fclose(pf) == 0) && ok;
    if (!ok) {
        set_error("short write to %s", path);
        return MORNA_E_IO;
    }
    return MORNA_OK;
}

int morna_stage_lines(morna_index *h, const morna_lines *L)
{
    if (!h || !L) {
        set_error("stage_lines: null argument");
        return MORNA_E_INVALID;
    }
    MORNA_TRY(morna_stage_junctions(h, L->key_bytes.data(), L->key_off.data(), (int64_t)L->idf.size(), L->row_ptr.data(),
                                    L->item_ids.data(), L->cov.data(), L->idf.data()));
    // the file's lines list their samples in ascending order of the external id: hand that order over with them
    return morna_stage_item_order(h, L->ext_ids.data(), (int64_t)L->ext_ids.size());
}

}  // extern "C"
