// pc_jit.cpp -- run-time specialisation of the score-only scan kernel for one adapter pair.
//
// hiprtc is loaded with dlopen (no link-time dependency); if it is missing, or a compile fails,
// pcj::get() returns null and the caller keeps using the ahead-of-time generic kernels -- still
// the GPU, only 9 instead of 5 (fp16) / 6 (int16) packed ops per cell pair.  PC_DISABLE_JIT=1
// forces that path.
#include "pc_jit.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

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

// hiprtc compile + substitution table.  No HIP runtime calls.
void compile_entry(Entry *e, const Recipe rc)
{
    std::lock_guard<std::mutex> one(g_compile_mu);
    if (g_cancel.load()) { e->log = "cancelled"; e->compile_done.store(true, std::memory_order_release); return; }
    Rtc &r = rtc();                      // first use loads libhiprtc (hundreds of ms): also off the caller's thread
    if (!r.ok) {
        e->log = "hiprtc not available";
        e->compile_done.store(true, std::memory_order_release);
        return;
    }
    const int R = rc.R;
    // letters per register row of each half (bottom-aligned): 0..4 = Dna5 code, 5 = padding row
    std::vector<int> lo(R, 5), hi(R, 5);
    for (int i = 0; i < rc.m_lo; ++i) lo[R - rc.m_lo + i] = dna5((unsigned char)rc.ad_lo[i]);
    for (int i = 0; i < rc.m_hi; ++i) hi[R - rc.m_hi + i] = dna5((unsigned char)rc.ad_hi[i]);
    std::vector<int> combos;             // distinct lo*6+hi
    std::vector<int> combo_of_row(R);
    for (int row = 0; row < R; ++row) {
        const int c = lo[row] * 6 + hi[row];
        int k = -1;
        for (size_t i = 0; i < combos.size(); ++i) if (combos[i] == c) k = (int)i;
        if (k < 0) { k = (int)combos.size(); combos.push_back(c); }
        combo_of_row[row] = k;
    }
    const int K = ((int)combos.size() + 3) / 4 * 4;
    // Register budget: T, U and four sets of substitution terms (two columns in flight, two being
    // fetched) = 2R + 4K, plus ~50 of bookkeeping.  Up to 120 the column loop fits 168 VGPRs -- three waves
    // per SIMD, the few values the allocator then spills are touched only outside the loop (measured: -5 %
    // on the 28+22-base pair, -4 % on 33+30; a 39+34-base pair, 142, spills inside the loop and loses 10 %).
    // Up to 140: 256 VGPRs, two waves.  Longer adapters (full barcode sequences, 63-111 bases) get the whole
    // register file of a SIMD -- one wave, the rows beyond 256 VGPRs parked in AGPRs -- which is still
    // several times faster than the generic kernel's column in LDS.
    const int waves = (2 * R + 4 * K <= 120) ? 3 : (2 * R + 4 * K <= 140) ? 2 : 1;
    e->R = R; e->K = K; e->m_lo = rc.m_lo; e->m_hi = rc.m_hi; e->f16 = rc.f16; e->waves = waves; e->kren = rc.kren;

    std::string init;
    for (int row = 0; row < R; ++row) { init += std::to_string(combo_of_row[row]); if (row + 1 < R) init += ","; }
    const std::string dR = "-DPC_R=" + std::to_string(R), dK = "-DPC_K=" + std::to_string(K),
                      dC = "-DPC_COMBO_INIT=" + init, dF = std::string("-DPC_F16=") + (rc.f16 ? "1" : "0"),
                      dE = "-DPC_EPS=" + std::to_string(rc.eps), dO = "-DPC_OE=(" + std::to_string(rc.gap_open + rc.eps) + ")",
                      dN = "-DPC_CEN=(" + std::to_string(rc.cen) + ")", dP = "-DPC_KREN=" + std::to_string(rc.kren),
                      dW = std::string("-DPC_WAVES=") + (getenv("PC_JIT_WAVES") ? getenv("PC_JIT_WAVES") : std::to_string(waves));
    const char *chk = getenv("PC_JIT_CHECK_RANGE");
    const std::string dG = std::string("-DPC_CHECK_RANGE=") + ((chk && *chk && *chk != '0') ? "1" : "0");
    const std::string dD = std::string("-DPC_DUAL=") + (rc.ad_lo != rc.ad_hi ? "1" : "0");
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", dR.c_str(), dK.c_str(), dC.c_str(), dF.c_str(),
                          dE.c_str(), dO.c_str(), dN.c_str(), dP.c_str(), dW.c_str(), dG.c_str(), dD.c_str()};
    hiprtcProgram prog = nullptr;
    if (r.CreateProgram(&prog, kSpecSource, "pc_spec_score.hip", 0, nullptr, nullptr) == 0) {
        const hiprtcResult rcode = r.CompileProgram(prog, (int)(sizeof(opts) / sizeof(opts[0])), opts);
        if (rcode != 0) {
            size_t n = 0;
            r.GetProgramLogSize(prog, &n);
            e->log.assign(n + 1, '\0');
            if (n) r.GetProgramLog(prog, &e->log[0]);
        } else {
            size_t csz = 0;
            r.GetCodeSize(prog, &csz);
            e->code.resize(csz);
            r.GetCode(prog, e->code.data());
            e->compiled_ok = csz > 0;
        }
        r.DestroyProgram(&prog);
    }
    // S table: [code of the low stream's base * 5 + code of the high stream's base][K letter pairs] of packed
    // (sub_lo - open + eps) | (sub_hi - open + eps) << 16
    e->table.assign((size_t)25 * K, 0);
    auto term = [&](int letter, int code) -> int {
        const int sub = letter == 5 ? 0 : (letter == code ? rc.match : rc.mismatch);
        return sub - rc.gap_open + rc.eps;
    };
    for (int cl = 0; cl < 5; ++cl)
        for (int ch = 0; ch < 5; ++ch)
            for (size_t k = 0; k < combos.size(); ++k) {
                const int l = term(combos[k] / 6, cl), h = term(combos[k] % 6, ch);
                e->table[(size_t)(cl * 5 + ch) * K + k] = rc.f16 ? (half_bits(l) | (half_bits(h) << 16))
                                                                   : (((uint32_t)l & 0xFFFFu) | ((uint32_t)h << 16));
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
    if (verbose) fprintf(stderr, "porechop_amd: specialised kernel R=%d K=%d f16=%d kren=%ld waves/CU=%d\n", e->R, e->K, e->f16 ? 1 : 0, e->kren, sp->blocks_per_cu);
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
          int gap_extend, double cells)
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
    const pcb::SpecPlan plan = pcb::spec_plan(match, mismatch, gap_open, gap_extend, R, force_int && *force_int && *force_int != '0');
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
    // a hiprtc compile costs 0.3-2 s and halves the scan (x6-10 for long adapters): start it once
    // the work seen for this adapter pair, over all launches so far, would have paid for it
    // (PC_JIT_MIN_CELLS overrides the 1e11-cell default)
    static const double min_cells = [] { const char *v = getenv("PC_JIT_MIN_CELLS"); return v ? atof(v) : 1e11; }();
    e->seen += cells;
    if (e->seen < min_cells) return nullptr;
    const Recipe rc{ad_lo, ad_hi, match, mismatch, gap_open, gap_extend, R, m_lo, m_hi, eps, 2, f16, kren, cen};
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
