// parse.hip -- native host-side pre-pass of `morna index` (no GPU work here).
//
// Counterpart of the reference's go_index line loop, count_samples and the host
// half of MornaIndex.add_junction (commanderson/morna morna.py:841-861, 789-822,
// 357-382): reads a (gzipped) intropolis file and produces exactly the CSR arrays
// morna_stage_junctions() takes -- kept lines in file order, first-seen internal
// ids, cumulative junction frequencies, idf = log(sample_count / freq) from libm.
// The Python tokenising loop is the reference's real end-to-end cost (SURVEY.md
// section 8f N1); this does the same work at I/O speed.
#include <unistd.h>
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
    // a row shard cut by morna_lines_shard: the ids above are LOCAL (global id - id_offset); of a whole parse: 0 / 1 / 0 / n_items
    int64_t shard_rank = 0, shard_world = 1, id_offset = 0, n_items_global = -1;
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

    // 0 / MORNA_E_IO (a damaged or truncated .gz: the reference's gzip.open raises there) / MORNA_E_INVALID (an exception
    // in a thread, e.g. bad_alloc): set by whichever thread hits it, read by the caller after stop()
    std::atomic<int> fatal{0};
    std::string fatal_msg;          // written once, under mu, before `fatal` is set

    void fail(int code, const std::string &msg)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!fatal.load()) {
            fatal_msg = msg;
            fatal = code;
        }
        eof = true;
        cv_work.notify_all();
        cv_done.notify_all();
        cv_room.notify_all();
    }

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
            if (got < (int)BLOCK_BYTES) {
                // a short read is the end of the file -- or of what can be inflated of it: zlib reports a stream that
                // stops early as Z_BUF_ERROR and damaged data as Z_DATA_ERROR, with the bytes before it delivered
                int errnum = Z_OK;
                const char *why = gzerror(f, &errnum);
                if (got < 0 || (errnum != Z_OK && errnum != Z_STREAM_END)) {
                    read_error = true;
                    read_error_msg = why ? why : "read error";
                    return false;
                }
            }
            if (got <= 0) break;                                    // end of file
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
    bool read_error = false;        // reader's side only (one thread), folded into `fatal` by whoever reads
    std::string read_error_msg;

    void start(const char *path_unused, bool count, int workers)
    {
        (void)path_unused;
        count_only = count;
        n_workers = workers;
        if (n_workers <= 1) return;
        threads.emplace_back([this] {   // the reader: inflation
            try {
                std::string carry;
                int64_t line_no = 1;
                for (;;) {
                    std::unique_ptr<ParsedBlock> B(new ParsedBlock());
                    const bool any = read_block(carry, *B, line_no);
                    if (read_error) {
                        fail(MORNA_E_IO, read_error_msg);
                        return;
                    }
                    std::unique_lock<std::mutex> lk(mu);
                    if (!any) {
                        eof = true;
                        cv_work.notify_all();
                        cv_done.notify_all();
                        return;
                    }
                    cv_room.wait(lk, [this] { return blocks.size() < max_in_flight || fatal.load(); });
                    if (fatal.load()) return;
                    blocks.push_back(std::move(B));
                    cv_work.notify_one();
                }
            } catch (const std::exception &e) {
                fail(MORNA_E_INVALID, e.what());
            }
        });
        for (int w = 0; w < n_workers - 1; w++)
            threads.emplace_back([this] {   // the tokenisers
                try {
                    for (;;) {
                        ParsedBlock *B = nullptr;
                        {
                            std::unique_lock<std::mutex> lk(mu);
                            cv_work.wait(lk, [this] { return next_to_parse < blocks.size() || eof; });
                            if (fatal.load() || next_to_parse >= blocks.size()) return;   // eof and nothing left
                            B = blocks[next_to_parse++].get();
                        }
                        parse_block(*B, count_only);
                        {
                            std::lock_guard<std::mutex> lk(mu);
                            cv_done.notify_all();
                        }
                    }
                } catch (const std::exception &e) {
                    fail(MORNA_E_INVALID, e.what());
                }
            });
    }

    std::string carry1;
    int64_t line1 = 1;
    // the next block in file order, tokenised; null at the end of the file
    std::unique_ptr<ParsedBlock> next_block()
    {
        if (n_workers <= 1) {
            if (fatal.load()) return nullptr;
            std::unique_ptr<ParsedBlock> B(new ParsedBlock());
            const bool any = read_block(carry1, *B, line1);
            if (read_error) {
                fail(MORNA_E_IO, read_error_msg);
                return nullptr;
            }
            if (!any) return nullptr;
            parse_block(*B, count_only);
            return B;
        }
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [this] { return fatal.load() || (!blocks.empty() && blocks.front()->done) || (eof && blocks.empty()); });
        if (fatal.load() || blocks.empty()) return nullptr;
        std::unique_ptr<ParsedBlock> B = std::move(blocks.front());
        blocks.pop_front();
        next_to_parse--;   // indices shift with the pop; the front block was taken by a worker long ago
        cv_room.notify_one();
        return B;
    }

    // the caller gives up (an exception on its side): the threads leave at their next look at the queue
    void abandon()
    {
        if (!threads.empty() && !fatal.load()) fail(MORNA_E_INVALID, "abandoned");
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

// stop the threads, close the file and fold what the pipeline or zlib reported into a return code
static int finish_pipeline(BlockPipeline &P, const char *path, int rc)
{
    P.stop();
    const int zrc = P.f ? gzclose(P.f) : Z_OK;
    P.f = nullptr;
    if (P.fatal.load()) {
        if (P.fatal.load() == MORNA_E_IO) set_error("%s: %s (truncated or damaged gzip stream)", path, P.fatal_msg.c_str());
        else set_error("parse_intropolis: %s", P.fatal_msg.c_str());
        return P.fatal.load();
    }
    if (rc == MORNA_OK && zrc != Z_OK) {
        set_error("%s: gzclose reports error %d (truncated or damaged gzip stream)", path, zrc);
        return MORNA_E_IO;
    }
    return rc;
}

static int parse_intropolis_impl(const char *path, int64_t sample_count, int64_t sample_threshold, BlockPipeline &PC,
                                 BlockPipeline &P, std::unique_ptr<morna_lines> &result)
{
    const int workers = parse_threads();
    if (sample_count <= 0) {
        // count_samples (morna.py:789-822): distinct sample-id STRINGS of column -2
        BlockPipeline &P = PC;
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
        rc = finish_pipeline(P, path, rc);
        if (rc != MORNA_OK) return rc;
        sample_count = (int64_t)seen.size();
    }
    P.f = gzopen(path, "rb");
    if (!P.f) {
        set_error("Unable to open %s", path);
        return MORNA_E_IO;
    }
    gzbuffer(P.f, 1 << 20);
    P.start(path, false, workers);
    result.reset(new morna_lines());
    morna_lines *L = result.get();
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
    rc = finish_pipeline(P, path, rc);
    if (rc != MORNA_OK) return rc;
    L->lines_read = lineno;
    L->freq_vals.reserve(L->freq_keys.size());
    for (const std::string &k : L->freq_keys) L->freq_vals.push_back(L->freq[k]);
    return MORNA_OK;
}

int morna_parse_intropolis(const char *path, int64_t sample_count, int64_t sample_threshold, morna_lines **out)
{
    if (!path || !out) {
        set_error("parse_intropolis: null argument");
        return MORNA_E_INVALID;
    }
    *out = nullptr;
    BlockPipeline count_pass, main_pass;
    std::unique_ptr<morna_lines> L;
    int rc;
    try {
        rc = parse_intropolis_impl(path, sample_count, sample_threshold, count_pass, main_pass, L);
    } catch (const std::exception &e) {   // bad_alloc in the merge: nothing may cross the C ABI, no thread may outlive the call
        set_error("parse_intropolis: %s", e.what());
        rc = MORNA_E_INVALID;
    }
    for (BlockPipeline *P : {&count_pass, &main_pass}) {   // whatever path was taken: threads joined, file closed
        P->abandon();
        P->stop();
        if (P->f) gzclose(P->f);
        P->f = nullptr;
    }
    if (rc != MORNA_OK) return rc;
    *out = L.release();
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
    // written under a name of its own and renamed into place: several ranks of `index --shards --cache` may write the
    // cache of the same file at once, and a reader must never see half of one
    const std::string tmp = std::string(path) + ".tmp." + std::to_string((long long)getpid());
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) {
        set_error("Unable to open %s for writing", tmp.c_str());
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
    ok = ok && rename(tmp.c_str(), path) == 0;
    if (!ok) {
        (void)remove(tmp.c_str());
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

// ---- row shards of ONE parsed data set -----------------------------------------------------------
// The global pass above has done everything that is global by definition: the threshold on a line's whole sample list,
// the cumulative frequency and idf = log(sample_count / freq) with the GLOBAL sample count (morna.py:357-374), first-seen
// internal ids over the whole file (morna.py:377-382).  Shard `rank` of `world` owns the global ids
// [rank * ceil(N / world), (rank + 1) * ceil(N / world)) and receives, in file order, the lines restricted to the
// entries of its items, ids renumbered from 0.  A cell of the matrix is the sum, in file order, of the entries of ITS
// sample in the lines of ITS column (morna.py:376-388): restricting the lines to a row range changes neither the terms
// of a cell nor their order, so the shard matrices stacked by rank ARE the matrix of the whole index, bit for bit.
// Lines that lose every entry are dropped (they touch no cell of the shard).  The frequency table stays global.
int morna_lines_shard(const morna_lines *L, int32_t rank, int32_t world, morna_lines **out)
{
    if (!L || !out || world <= 0 || rank < 0 || rank >= world) {
        set_error("lines_shard: need 0 <= rank < world");
        return MORNA_E_INVALID;
    }
    *out = nullptr;
    if (L->shard_world != 1) {
        set_error("lines_shard: the lines are a shard already (cut the whole parse)");
        return MORNA_E_STATE;
    }
    try {
        const int64_t N = (int64_t)L->ext_ids.size(), per = (N + world - 1) / world;
        const int64_t lo = std::min<int64_t>(N, per * rank), hi = std::min<int64_t>(N, lo + per);
        const int64_t J = (int64_t)L->idf.size();
        std::unique_ptr<morna_lines> S(new morna_lines());
        // sizes first (one pass), then the copy: no reallocation of the big arrays
        int64_t nnz = 0, nl = 0, kb = 0;
        for (int64_t j = 0; j < J; j++) {
            int64_t c = 0;
            for (int64_t t = L->row_ptr[(size_t)j]; t < L->row_ptr[(size_t)j + 1]; t++) {
                const int64_t id = L->item_ids[(size_t)t];
                c += (id >= lo && id < hi) ? 1 : 0;
            }
            if (c) {
                nnz += c;
                nl++;
                kb += L->key_off[(size_t)j + 1] - L->key_off[(size_t)j];
            }
        }
        S->item_ids.resize((size_t)nnz);
        S->cov.resize((size_t)nnz);
        S->idf.reserve((size_t)nl);
        S->key_off.reserve((size_t)nl + 1);
        S->row_ptr.reserve((size_t)nl + 1);
        S->key_bytes.reserve((size_t)kb);
        int64_t at = 0;
        for (int64_t j = 0; j < J; j++) {
            const int64_t at0 = at;
            for (int64_t t = L->row_ptr[(size_t)j]; t < L->row_ptr[(size_t)j + 1]; t++) {
                const int64_t id = L->item_ids[(size_t)t];
                if (id >= lo && id < hi) {
                    S->item_ids[(size_t)at] = (int32_t)(id - lo);
                    S->cov[(size_t)at] = L->cov[(size_t)t];
                    at++;
                }
            }
            if (at == at0) continue;
            S->row_ptr.push_back(at);
            S->idf.push_back(L->idf[(size_t)j]);
            S->key_bytes.insert(S->key_bytes.end(), L->key_bytes.begin() + L->key_off[(size_t)j],
                                L->key_bytes.begin() + L->key_off[(size_t)j + 1]);
            S->key_off.push_back((int64_t)S->key_bytes.size());
        }
        S->ext_ids.assign(L->ext_ids.begin() + lo, L->ext_ids.begin() + hi);
        S->freq_keys = L->freq_keys;
        S->freq_vals = L->freq_vals;
        S->sample_count = L->sample_count;
        S->skipped = L->skipped;
        S->lines_read = L->lines_read;
        S->shard_rank = rank;
        S->shard_world = world;
        S->id_offset = lo;
        S->n_items_global = N;
        *out = S.release();
    } catch (const std::exception &e) {
        set_error("lines_shard: %s", e.what());
        return MORNA_E_INVALID;
    }
    return MORNA_OK;
}

// info[4] = {rank, world, id_offset (global id of local id 0), n_items of the whole data set}
int morna_lines_shard_info(const morna_lines *L, int64_t *info)
{
    if (!L || !info) return MORNA_E_INVALID;
    info[0] = L->shard_rank;
    info[1] = L->shard_world;
    info[2] = L->id_offset;
    info[3] = L->n_items_global >= 0 ? L->n_items_global : (int64_t)L->ext_ids.size();
    return MORNA_OK;
}

// Lines handed over as arrays (a Python-side parse, a synthetic data set): the same object the parser makes, so that the
// shard cut and the staging have ONE implementation.  ext_ids[n_items]: external sample id of every internal id.
int morna_lines_from_arrays(const uint8_t *key_bytes, const int64_t *key_off, int64_t J, const int64_t *row_ptr,
                            const int32_t *item_ids, const int32_t *cov, const double *idf, const int64_t *ext_ids,
                            int64_t n_items, int64_t sample_count, morna_lines **out)
{
    if (!out || J < 0 || n_items < 0 || (J > 0 && (!key_bytes || !key_off || !row_ptr || !item_ids || !cov || !idf)) ||
        (n_items > 0 && !ext_ids)) {
        set_error("lines_from_arrays: null argument");
        return MORNA_E_INVALID;
    }
    *out = nullptr;
    if (J > 0 && (key_off[0] != 0 || row_ptr[0] != 0)) {
        set_error("lines_from_arrays: offsets must start at 0");
        return MORNA_E_INVALID;
    }
    for (int64_t j = 0; j < J; j++)
        if (key_off[j + 1] < key_off[j] || row_ptr[j + 1] < row_ptr[j]) {
            set_error("lines_from_arrays: offsets must be non-decreasing (line %lld)", (long long)j);
            return MORNA_E_INVALID;
        }
    const int64_t nnz = J ? row_ptr[J] : 0;
    for (int64_t t = 0; t < nnz; t++)
        if (item_ids[t] < 0 || item_ids[t] >= n_items) {
            set_error("lines_from_arrays: item id %d out of range [0, %lld)", item_ids[t], (long long)n_items);
            return MORNA_E_RANGE;
        }
    try {
        std::unique_ptr<morna_lines> L(new morna_lines());
        if (J > 0) {
            L->key_bytes.assign(key_bytes, key_bytes + key_off[J]);
            L->key_off.assign(key_off, key_off + J + 1);
            L->row_ptr.assign(row_ptr, row_ptr + J + 1);
            L->item_ids.assign(item_ids, item_ids + nnz);
            L->cov.assign(cov, cov + nnz);
            L->idf.assign(idf, idf + J);
        }
        if (n_items > 0) L->ext_ids.assign(ext_ids, ext_ids + n_items);
        L->sample_count = sample_count;
        *out = L.release();
    } catch (const std::exception &e) {
        set_error("lines_from_arrays: %s", e.what());
        return MORNA_E_INVALID;
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
