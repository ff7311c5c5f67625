// pc_jit.cpp -- run-time specialisation of the score-only scan kernel for one adapter pair.
//
// hiprtc is loaded with dlopen (no link-time dependency); if it is missing, or a compile fails,
// pcj::get() returns null and the caller keeps using the ahead-of-time generic kernels -- still
// the GPU, only 9 instead of 5 (fp16) / 6 (int16) packed ops per cell pair.  PC_DISABLE_JIT=1
// forces that path.
#include "pc_jit.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "pc_bounds.h"
#include "pc_jit_source.h"

namespace pcj {

namespace {

typedef struct _hiprtcProgram *hiprtcProgram;
typedef int hiprtcResult;
struct Rtc {
    void *lib = nullptr;
    hiprtcResult (*CreateProgram)(hiprtcProgram *, const char *, const char *, int, const char *const *, const char *const *) = nullptr;
    hiprtcResult (*CompileProgram)(hiprtcProgram, int, const char *const *) = nullptr;
    hiprtcResult (*GetProgramLogSize)(hiprtcProgram, size_t *) = nullptr;
    hiprtcResult (*GetProgramLog)(hiprtcProgram, char *) = nullptr;
    hiprtcResult (*GetCodeSize)(hiprtcProgram, size_t *) = nullptr;
    hiprtcResult (*GetCode)(hiprtcProgram, char *) = nullptr;
    hiprtcResult (*DestroyProgram)(hiprtcProgram *) = nullptr;
    bool ok = false;
};

Rtc &rtc()
{
    static Rtc r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"libhiprtc.so.7", "libhiprtc.so", "libhiprtc.so.6"};
        for (const char *n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (r.lib) break; }
        if (!r.lib) return;
#define PC_SYM(field, sym) *(void **)(&r.field) = dlsym(r.lib, sym); if (!r.field) return;
        PC_SYM(CreateProgram, "hiprtcCreateProgram")
        PC_SYM(CompileProgram, "hiprtcCompileProgram")
        PC_SYM(GetProgramLogSize, "hiprtcGetProgramLogSize")
        PC_SYM(GetProgramLog, "hiprtcGetProgramLog")
        PC_SYM(GetCodeSize, "hiprtcGetCodeSize")
        PC_SYM(GetCode, "hiprtcGetCode")
        PC_SYM(DestroyProgram, "hiprtcDestroyProgram")
#undef PC_SYM
        r.ok = true;
    });
    return r;
}

int dna5(unsigned char c)
{
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': case 'U': case 'u': return 3;
        default: return 4;
    }
}

// IEEE half bit pattern of a small integer (|v| <= 2048 is exact)
uint32_t half_bits(int v)
{
    if (v == 0) return 0;
    const uint32_t sign = v < 0 ? 0x8000u : 0u;
    uint32_t a = (uint32_t)(v < 0 ? -v : v);
    int e = 0;
    while ((a >> (e + 1)) != 0) ++e;                 // floor(log2(a)), <= 11
    const uint32_t mant = e <= 10 ? ((a << (10 - e)) & 0x3FFu) : ((a >> (e - 10)) & 0x3FFu);
    return sign | ((uint32_t)(e + 15) << 10) | mant;
}

// One adapter pair + scheme: its specialised kernel goes through
//   ABSENT (work seen so far below the threshold) -> COMPILING (hiprtc, possibly on a worker
//   thread) -> READY (module loaded, table uploaded) | FAILED (never retried)
struct Entry {
    enum State { ABSENT, COMPILING, READY, FAILED } state = ABSENT;
    double seen = 0;                 // cells scanned with the generic kernel so far
    bool disk_checked = false;       // the kernel caches on disk were looked at (once per entry)
    bool from_disk = false;
    Spec *spec = nullptr;
    // product of the compile step (no HIP calls in it: it may run on a worker thread)
    std::vector<char> code;
    std::vector<uint32_t> table;
    std::string log;
    bool compiled_ok = false;
    std::atomic<bool> compile_done{false};
    std::thread worker;
    int R = 0, K = 0, m_lo = 0, m_hi = 0, waves = 2;
    bool f16 = false;
    long kren = 0;
};

std::mutex g_mu;                               // the cache
std::mutex g_compile_mu;                       // one hiprtc compile at a time
std::map<std::string, Entry *> g_cache;
std::atomic<int> g_async{0};
std::atomic<bool> g_cancel{false};             // set at shutdown: queued compiles are dropped

struct Recipe {                                // everything the compile step needs, by value
    std::string ad_lo, ad_hi;
    int match, mismatch, gap_open, gap_extend;
    int R, m_lo, m_hi, eps, waves;
    bool f16;
    long kren, cen;
};

// Everything about a kernel that follows from its recipe without compiling anything: the hiprtc options (they
// ARE the kernel's identity: the source is one constant string), the distinct letter pairs, the register plan.
struct Derived {
    std::vector<std::string> opts;
    std::vector<int> combos;
    int K = 0, waves = 2;
};

Derived derive(const Recipe &rc)
{
    Derived d;
    const int R = rc.R;
    // letters per register row of each half (bottom-aligned): 0..4 = Dna5 code, 5 = padding row
    std::vector<int> lo(R, 5), hi(R, 5);
    for (int i = 0; i < rc.m_lo; ++i) lo[R - rc.m_lo + i] = dna5((unsigned char)rc.ad_lo[i]);
    for (int i = 0; i < rc.m_hi; ++i) hi[R - rc.m_hi + i] = dna5((unsigned char)rc.ad_hi[i]);
    std::vector<int> combo_of_row(R);
    for (int row = 0; row < R; ++row) {
        const int c = lo[row] * 6 + hi[row];
        int k = -1;
        for (size_t i = 0; i < d.combos.size(); ++i) if (d.combos[i] == c) k = (int)i;
        if (k < 0) { k = (int)d.combos.size(); d.combos.push_back(c); }
        combo_of_row[row] = k;
    }
    const int K = ((int)d.combos.size() + 3) / 4 * 4;
    // Register budget: T, U and four sets of substitution terms (two columns in flight, two being
    // fetched) = 2R + 4K, plus ~50 of bookkeeping.  Up to 120 the column loop fits 168 VGPRs -- three waves
    // per SIMD, the few values the allocator then spills are touched only outside the loop (measured: -5 %
    // on the 28+22-base pair, -4 % on 33+30; a 39+34-base pair, 142, spills inside the loop and loses 10 %).
    // Up to 140: 256 VGPRs, two waves.  Longer adapters (full barcode sequences, 63-111 bases) get the whole
    // register file of a SIMD -- one wave, the rows beyond 256 VGPRs parked in AGPRs -- which is still
    // several times faster than the generic kernel's column in LDS.
    const int waves = (2 * R + 4 * K <= 120) ? 3 : (2 * R + 4 * K <= 140) ? 2 : 1;
    d.K = K; d.waves = waves;
    std::string init;
    for (int row = 0; row < R; ++row) { init += std::to_string(combo_of_row[row]); if (row + 1 < R) init += ","; }
    const char *chk = getenv("PC_JIT_CHECK_RANGE");
    d.opts = {"--offload-arch=gfx950", "-O3", "-std=c++17",
              "-DPC_R=" + std::to_string(R), "-DPC_K=" + std::to_string(K), "-DPC_COMBO_INIT=" + init,
              std::string("-DPC_F16=") + (rc.f16 ? "1" : "0"), "-DPC_EPS=" + std::to_string(rc.eps),
              "-DPC_OE=(" + std::to_string(rc.gap_open + rc.eps) + ")", "-DPC_CEN=(" + std::to_string(rc.cen) + ")",
              "-DPC_KREN=" + std::to_string(rc.kren),
              std::string("-DPC_WAVES=") + (getenv("PC_JIT_WAVES") ? getenv("PC_JIT_WAVES") : std::to_string(waves)),
              std::string("-DPC_CHECK_RANGE=") + ((chk && *chk && *chk != '0') ? "1" : "0"),
              std::string("-DPC_DUAL=") + (rc.ad_lo != rc.ad_hi ? "1" : "0")};
    return d;
}

// S table: [code of the low stream's base * 5 + code of the high stream's base][K letter pairs] of packed
// (sub_lo - open + eps) | (sub_hi - open + eps) << 16
std::vector<uint32_t> substitution_table(const Recipe &rc, const Derived &d)
{
    std::vector<uint32_t> table((size_t)25 * d.K, 0);
    auto term = [&](int letter, int code) -> int {
        const int sub = letter == 5 ? 0 : (letter == code ? rc.match : rc.mismatch);
        return sub - rc.gap_open + rc.eps;
    };
    for (int cl = 0; cl < 5; ++cl)
        for (int ch = 0; ch < 5; ++ch)
            for (size_t k = 0; k < d.combos.size(); ++k) {
                const int l = term(d.combos[k] / 6, cl), h = term(d.combos[k] % 6, ch);
                // the table terms are values the kernel adds in its lane type: they must be exact there too
                // (spec_plan's gate bounds match, mismatch and open for the fp16 variant; asserted here, where they are formed)
                const int lim = rc.f16 ? pcb::kF16Limit : 32000;
                if (l > lim || l < -lim || h > lim || h < -lim) return std::vector<uint32_t>();
                table[(size_t)(cl * 5 + ch) * d.K + k] = rc.f16 ? (half_bits(l) | (half_bits(h) << 16))
                                                                : (((uint32_t)l & 0xFFFFu) | ((uint32_t)h << 16));
            }
    return table;
}

// ---- kernel cache on disk ------------------------------------------------------------------------------
// A specialised kernel is a pure function of (kSpecSource, hiprtc options).  Its code object is kept in a file
// named by a 128-bit hash of both; the file repeats the options, and a load checks them and the source's own
// hash, so a collision or a stale file can only cause a recompile.  Two directories are consulted:
//   <directory of this library>/kernel_cache   read-only: filled at build time for the static panel
//                                              (pc_jit_precompile; `make kernels`, __graft_entry__.build())
//   PC_JIT_CACHE_DIR, else $XDG_CACHE_HOME/porechop_amd, else $HOME/.cache/porechop_amd
//                                              read-write: what this machine compiled at run time
// PC_JIT_CACHE_DIR=off (or 0) disables the second one.
uint64_t fnv64(const void *p, size_t n, uint64_t h)
{
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 0x100000001b3ull; }
    return h;
}

const char kCacheMagic[] = "PCJK3\n";

// The toolchain a cached code object belongs to: the HIP / clang versions THIS LIBRARY was compiled with (hipcc and
// hiprtc of one ROCm install share their compiler; the library is built in-tree by that install).  Part of the file
// name's hash and of the header a load checks, so code objects in ~/.cache/porechop_amd or in a copied kernel_cache/
// do not survive a toolchain upgrade: they simply stop being found and are recompiled.
#define PC_STR2(x) #x
#define PC_STR(x) PC_STR2(x)
const char kToolchainTag[] = "toolchain=hip-" PC_STR(HIP_VERSION_MAJOR) "." PC_STR(HIP_VERSION_MINOR) "." PC_STR(HIP_VERSION_PATCH)
                             "/clang-" __clang_version__;

std::string joined(const std::vector<std::string> &opts)
{
    std::string s;
    for (const std::string &o : opts) { s += o; s += ' '; }
    s += kToolchainTag;
    return s;
}

uint64_t source_hash()
{
    static const uint64_t h = fnv64(kSpecSource, strlen(kSpecSource), 0xcbf29ce484222325ull);
    return h;
}

std::string cache_file_name(const std::vector<std::string> &opts)
{
    const std::string o = joined(opts);
    const uint64_t h1 = fnv64(o.data(), o.size(), source_hash());
    const uint64_t h2 = fnv64(o.data(), o.size(), source_hash() * 0x9E3779B97F4A7C15ull + 0x632be59bd9b4e019ull);
    char buf[64];
    snprintf(buf, sizeof buf, "%016llx%016llx.pcjk", (unsigned long long)h1, (unsigned long long)h2);
    return buf;
}

std::string library_dir()
{
    Dl_info info;
    if (dladdr((void *)&fnv64, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        const size_t k = p.rfind('/');
        return k == std::string::npos ? std::string(".") : p.substr(0, k);
    }
    return ".";
}

std::string intree_cache_dir() { static const std::string d = library_dir() + "/kernel_cache"; return d; }

std::string user_cache_dir()
{
    const char *e = getenv("PC_JIT_CACHE_DIR");
    if (e) {
        if (!*e || !strcmp(e, "0") || !strcmp(e, "off")) return "";
        return e;
    }
    const char *x = getenv("XDG_CACHE_HOME");
    if (x && *x) return std::string(x) + "/porechop_amd";
    const char *h = getenv("HOME");
    if (h && *h) return std::string(h) + "/.cache/porechop_amd";
    return "";
}

void make_dirs(const std::string &dir)
{
    for (size_t k = 1; k <= dir.size(); ++k)
        if (k == dir.size() || dir[k] == '/') (void)mkdir(dir.substr(0, k).c_str(), 0755);
}

bool cache_read(const std::string &dir, const std::vector<std::string> &opts, std::vector<char> &code)
{
    if (dir.empty()) return false;
    const std::string path = dir + "/" + cache_file_name(opts);
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    bool ok = false;
    const std::string want = joined(opts);
    char magic[sizeof kCacheMagic] = {0};
    uint64_t src = 0, olen = 0, clen = 0;
    std::string got;
    if (fread(magic, 1, sizeof kCacheMagic - 1, f) == sizeof kCacheMagic - 1 && !strcmp(magic, kCacheMagic) &&
        fread(&src, 8, 1, f) == 1 && src == source_hash() && fread(&olen, 8, 1, f) == 1 && olen == want.size()) {
        got.resize(olen);
        if (fread(&got[0], 1, olen, f) == olen && got == want && fread(&clen, 8, 1, f) == 1 && clen > 0 && clen < (1ull << 28)) {
            code.resize(clen);
            uint64_t sum = 0;
            ok = fread(code.data(), 1, clen, f) == clen && fread(&sum, 8, 1, f) == 1 &&
                 sum == fnv64(code.data(), clen, 0xcbf29ce484222325ull);
        }
    }
    fclose(f);
    if (!ok) code.clear();
    return ok;
}

bool cache_write(const std::string &dir, const std::vector<std::string> &opts, const std::vector<char> &code)
{
    if (dir.empty() || code.empty()) return false;
    make_dirs(dir);
    const std::string path = dir + "/" + cache_file_name(opts);
    char tmp[64];
    snprintf(tmp, sizeof tmp, ".tmp%ld_%p", (long)getpid(), (const void *)&code);
    const std::string tpath = path + tmp;
    FILE *f = fopen(tpath.c_str(), "wb");
    if (!f) return false;
    const std::string o = joined(opts);
    const uint64_t src = source_hash(), olen = o.size(), clen = code.size(), sum = fnv64(code.data(), code.size(), 0xcbf29ce484222325ull);
    bool ok = fwrite(kCacheMagic, 1, sizeof kCacheMagic - 1, f) == sizeof kCacheMagic - 1 && fwrite(&src, 8, 1, f) == 1 &&
              fwrite(&olen, 8, 1, f) == 1 && fwrite(o.data(), 1, olen, f) == olen && fwrite(&clen, 8, 1, f) == 1 &&
              fwrite(code.data(), 1, clen, f) == clen && fwrite(&sum, 8, 1, f) == 1;
    ok = (fclose(f) == 0) && ok;
    if (ok) ok = rename(tpath.c_str(), path.c_str()) == 0;      // atomic: readers see a whole file or none
    if (!ok) (void)unlink(tpath.c_str());
    return ok;
}

std::atomic<long> g_n_compiled{0}, g_n_disk{0};

// hiprtc compile of one kernel.  No HIP runtime calls (hiprtc needs no device).
bool hiprtc_compile(const std::vector<std::string> &opts, std::vector<char> &code, std::string &log)
{
    Rtc &r = rtc();                      // first use loads libhiprtc (hundreds of ms)
    if (!r.ok) { log = "hiprtc not available"; return false; }
    std::vector<const char *> o;
    for (const std::string &s : opts) o.push_back(s.c_str());
    hiprtcProgram prog = nullptr;
    bool ok = false;
    if (r.CreateProgram(&prog, kSpecSource, "pc_spec_score.hip", 0, nullptr, nullptr) == 0) {
        const hiprtcResult rcode = r.CompileProgram(prog, (int)o.size(), o.data());
        if (rcode != 0) {
            size_t n = 0;
            r.GetProgramLogSize(prog, &n);
            log.assign(n + 1, '\0');
            if (n) r.GetProgramLog(prog, &log[0]);
        } else {
            size_t csz = 0;
            r.GetCodeSize(prog, &csz);
            code.resize(csz);
            r.GetCode(prog, code.data());
            ok = csz > 0;
        }
        r.DestroyProgram(&prog);
    } else {
        log = "hiprtcCreateProgram failed";
    }
    return ok;
}

void fill_entry(Entry *e, const Recipe &rc, const Derived &d)
{
    e->R = rc.R; e->K = d.K; e->m_lo = rc.m_lo; e->m_hi = rc.m_hi; e->f16 = rc.f16; e->waves = d.waves; e->kren = rc.kren;
    e->table = substitution_table(rc, d);           // empty: a table term outside the lane type's exact range (never expected)
}

// The kernel from one of the caches on disk, if it is there.  No HIP runtime calls.
bool load_entry_from_disk(Entry *e, const Recipe &rc)
{
    const Derived d = derive(rc);
    if (!cache_read(intree_cache_dir(), d.opts, e->code) && !cache_read(user_cache_dir(), d.opts, e->code)) return false;
    fill_entry(e, rc, d);
    if (e->table.empty()) { e->code.clear(); return false; }
    e->compiled_ok = true;
    e->from_disk = true;
    g_n_disk.fetch_add(1);
    return true;
}

// hiprtc compile + substitution table (+ a copy of the code object in the user's cache directory).
// No HIP runtime calls: it may run on a worker thread.
void compile_entry(Entry *e, const Recipe rc)
{
    std::lock_guard<std::mutex> one(g_compile_mu);
    if (g_cancel.load()) { e->log = "cancelled"; e->compile_done.store(true, std::memory_order_release); return; }
    const Derived d = derive(rc);
    fill_entry(e, rc, d);
    e->compiled_ok = !e->table.empty() && hiprtc_compile(d.opts, e->code, e->log);
    if (e->table.empty()) e->log = "a substitution-table term lies outside the lane type's exact range";
    if (e->compiled_ok) {
        g_n_compiled.fetch_add(1);
        (void)cache_write(user_cache_dir(), d.opts, e->code);
    }
    e->compile_done.store(true, std::memory_order_release);
}

// Module load + table upload on the caller's thread (the one that owns the device context).
void finalize_entry(Entry *e, bool verbose)
{
    if (e->worker.joinable()) e->worker.join();
    e->state = Entry::FAILED;
    if (!e->compiled_ok) {
        static bool told = false;
        if (!told || e->log != "hiprtc not available")
            fprintf(stderr, "porechop_amd: no specialised kernel (%s), using the generic scan kernels\n", e->log.c_str());
        told = true;
        return;
    }
    Spec *sp = new Spec();
    sp->R = e->R; sp->K = e->K; sp->m_lo = e->m_lo; sp->m_hi = e->m_hi; sp->f16 = e->f16; sp->waves = e->waves;
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    if (hipModuleLoadData(&mod, e->code.data()) != hipSuccess || hipModuleGetFunction(&fn, mod, "pc_spec_score") != hipSuccess) {
        fprintf(stderr, "porechop_amd: loading the specialised kernel failed, using the generic kernels\n");
        delete sp;
        return;
    }
    sp->module = mod; sp->function = fn;
    // resident workgroups (= waves) per CU as the runtime sees them -- registers AND LDS -- for the launch
    // planner's balanced heads (pc_api.cpp plan_score_launches)
    int nb = 0;
    if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64, 0) == hipSuccess && nb > 0) sp->blocks_per_cu = nb;
    else sp->blocks_per_cu = 4 * e->waves;
    void *d = nullptr;
    if (hipMalloc(&d, e->table.size() * 4) != hipSuccess ||
        hipMemcpy(d, e->table.data(), e->table.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        delete sp;
        return;
    }
    sp->d_table = d;
    if (verbose) fprintf(stderr, "porechop_amd: specialised kernel R=%d K=%d f16=%d kren=%ld waves/CU=%d (%s)\n", e->R, e->K, e->f16 ? 1 : 0, e->kren, sp->blocks_per_cu, e->from_disk ? "from the kernel cache on disk" : "compiled now");
    e->code.clear(); e->code.shrink_to_fit();
    e->spec = sp;
    e->state = Entry::READY;
}

}  // namespace

bool disabled()
{
    const char *e = getenv("PC_DISABLE_JIT");
    return e && *e && *e != '0';
}

void set_async(int on) { g_async.store(on ? 1 : 0); }

void wait_idle(bool cancel_queued)
{
    // a worker must not be inside hiprtc when the process image is torn down: callers with
    // asynchronous specialisation on call this before exiting (the Python binding registers it with
    // atexit).  Compiles that have not started yet are dropped; the one in flight is waited for.
    if (cancel_queued) g_cancel.store(true);
    std::vector<Entry *> es;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto &kv : g_cache) es.push_back(kv.second);
    }
    for (Entry *e : es) if (e->worker.joinable()) e->worker.join();
}

namespace {
// a worker thread must not be inside hiprtc while the process image is torn down, whoever loaded the library
struct JoinAtExit { ~JoinAtExit() { wait_idle(true); } } g_join_at_exit;
}  // namespace

Spec *get(int device, const std::string &ad_lo, const std::string &ad_hi, int match, int mismatch, int gap_open,
          int gap_extend, double cells, bool int16_only)
{
    if (disabled()) return nullptr;
    const int m_lo = (int)ad_lo.size(), m_hi = (int)ad_hi.size();
    const int R = m_lo > m_hi ? m_lo : m_hi;
    static const bool verbose = [] { const char *v = getenv("PC_JIT_VERBOSE"); return v && *v && *v != '0'; }();
    if (R < 2 || R > pcb::MAX_ADAPTER) {
        if (verbose) fprintf(stderr, "porechop_amd: no specialised kernel for %d rows\n", R);
        return nullptr;
    }
    // Drifting coordinates (pc_jit_source.h): a value X of row rho, jj columns after the last
    // renormalisation, is held as X + (rho + jj [+1]) * eps - C.  True values lie in
    // [low, high]; C puts `low` at -lim and the renormalisation period KREN keeps the top below
    // +lim, lim = what the lane type holds exactly (fp16: integers up to 2048).  Packed-fp16
    // variant (5 ops per cell pair, v_pk_maximum3_f16) when that leaves a useful period, otherwise
    // packed-int16 (6 ops).  PC_JIT_INT16=1 forces the latter.
    const int eps = -gap_extend;
    const char *force_int = getenv("PC_JIT_INT16");
    const pcb::SpecPlan plan = pcb::spec_plan(match, mismatch, gap_open, gap_extend, R, int16_only || (force_int && *force_int && *force_int != '0'));
    const bool f16 = plan.f16;
    const long kren = plan.kren;
    if (!plan.ok) {
        if (verbose) fprintf(stderr, "porechop_amd: no specialised kernel: scores too large for drifting coordinates\n");
        return nullptr;
    }
    const long cen = plan.cen;
    char keybuf[64];
    snprintf(keybuf, sizeof keybuf, "|%d|%d,%d,%d,%d|%d", device, match, mismatch, gap_open, gap_extend, f16 ? 1 : 0);
    const std::string key = ad_lo + "|" + ad_hi + keybuf;
    std::lock_guard<std::mutex> lk(g_mu);
    Entry *&slot = g_cache[key];
    if (!slot) slot = new Entry();
    Entry *e = slot;
    if (e->state == Entry::READY) return e->spec;
    if (e->state == Entry::FAILED) return nullptr;
    if (e->state == Entry::COMPILING) {
        // asynchronous mode: the generic kernel keeps running until the worker is done
        if (!e->compile_done.load(std::memory_order_acquire)) return nullptr;
        finalize_entry(e, verbose);
        return e->spec;
    }
    const Recipe rc{ad_lo, ad_hi, match, mismatch, gap_open, gap_extend, R, m_lo, m_hi, eps, 2, f16, kren, cen};
    // a kernel that is already on disk (built with the library for the static panel, or compiled by an earlier
    // process on this machine) costs a file read and a module load: use it from the first launch on
    if (!e->disk_checked) {
        e->disk_checked = true;
        if (load_entry_from_disk(e, rc)) {
            e->compile_done.store(true, std::memory_order_release);
            finalize_entry(e, verbose);
            return e->spec;
        }
    }
    // a hiprtc compile costs 0.3-2 s and halves the scan (x6-10 for long adapters): start it once
    // the work seen for this adapter pair, over all launches so far, would have paid for it
    // (PC_JIT_MIN_CELLS overrides the 1e11-cell default)
    static const double min_cells = [] { const char *v = getenv("PC_JIT_MIN_CELLS"); return v ? atof(v) : 1e11; }();
    e->seen += cells;
    if (e->seen < min_cells) return nullptr;
    e->state = Entry::COMPILING;
    // Compile in place only when THIS launch alone is worth the stall; a small launch that merely tipped
    // the pair's running total over the threshold (the mask-and-realign rounds: a millisecond of work)
    // never waits 0.3 s for a compile -- the worker thread builds the kernel, this launch and the next
    // ones use the generic kernel, a later one picks the result up.
    if (g_async.load() || cells < min_cells) {
        e->worker = std::thread(compile_entry, e, rc);
        return nullptr;
    }
    compile_entry(e, rc);
    finalize_entry(e, verbose);
    return e->spec;
}

int precompile(const std::string &ad_lo, const std::string &ad_hi, int match, int mismatch, int gap_open, int gap_extend,
               const std::string &dir)
{
    const int m_lo = (int)ad_lo.size(), m_hi = (int)ad_hi.size();
    const int R = m_lo > m_hi ? m_lo : m_hi;
    if (R < 2 || R > pcb::MAX_ADAPTER || m_lo < 1 || m_hi < 1) return -1;
    if (pcb::is_linear(gap_open, gap_extend) || !pcb::scores_supported(match, mismatch, gap_open, gap_extend, R)) return -1;
    const char *force_int = getenv("PC_JIT_INT16");
    const pcb::SpecPlan plan = pcb::spec_plan(match, mismatch, gap_open, gap_extend, R, force_int && *force_int && *force_int != '0');
    if (!plan.ok) return -1;
    const Recipe rc{ad_lo, ad_hi, match, mismatch, gap_open, gap_extend, R, m_lo, m_hi, -gap_extend, 2, plan.f16, plan.kren, plan.cen};
    const Derived d = derive(rc);
    const std::string where = dir.empty() ? intree_cache_dir() : dir;
    std::vector<char> code;
    if (cache_read(where, d.opts, code)) return 1;
    std::string log;
    if (!hiprtc_compile(d.opts, code, log)) {
        fprintf(stderr, "porechop_amd: precompile failed: %s\n", log.c_str());
        return -2;
    }
    return cache_write(where, d.opts, code) ? 0 : -3;
}

void stats(long *compiled, long *from_disk)
{
    if (compiled) *compiled = g_n_compiled.load();
    if (from_disk) *from_disk = g_n_disk.load();
}

int launch(const Spec *sp, const SpecArgs &a, int grid, void *stream)
{
    SpecArgs args = a;
    args.s_table = (const uint32_t *)sp->d_table;
    args.m_lo = sp->m_lo; args.m_hi = sp->m_hi;
    args.one2 = 0x00010001u;
    void *params[] = {&args};
    const hipError_t e = hipModuleLaunchKernel((hipFunction_t)sp->function, (unsigned)grid, 1, 1, 64, 1, 1, 0,
                                               (hipStream_t)stream, params, nullptr);
    return e == hipSuccess ? 0 : -2;
}

}  // namespace pcj
