// pc_jit.cpp -- run-time specialisation of the score-only scan kernel for one adapter pair.
//
// hiprtc is loaded with dlopen (no link-time dependency); if it is missing, or a compile fails,
// pcj::get() returns null and the caller keeps using the ahead-of-time generic kernels -- still
// the GPU, only 11 instead of 5 (fp16) / 6 (int16) packed ops per cell pair.  PC_DISABLE_JIT=1
// forces that path.
#include "pc_jit.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <mutex>
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

std::mutex g_mu;
std::map<std::string, Spec *> g_cache;   // key -> spec (null = failed, do not retry)
std::map<std::string, double> g_seen;    // key -> cells scanned with the generic kernel so far

}  // namespace

bool disabled()
{
    const char *e = getenv("PC_DISABLE_JIT");
    return e && *e && *e != '0';
}

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
    const long low = std::min<long>({2L * gap_open + (long)(R - 1) * gap_extend,
                                     (long)gap_open + (long)(R - 1) * gap_extend + mismatch, (long)gap_open});
    const long high = (long)match * R;
    auto period = [&](long lim) -> long {
        const long k = (2 * lim - (high - low) - (long)(R + 6) * eps) / eps;
        return k < 0 ? 0 : std::min<long>(k / 4 * 4, 1L << 20);
    };
    const char *force_int = getenv("PC_JIT_INT16");
    const bool f16 = !(force_int && *force_int && *force_int != '0') && period(2040) >= 64 &&
                     high + (long)R * eps <= 2040 && -mismatch <= 1000 && -gap_open <= 1000;
    const long lim = f16 ? 2040 : 32000;
    const long kren = period(lim);
    if (kren < 64) {
        if (verbose) fprintf(stderr, "porechop_amd: no specialised kernel: scores too large for drifting coordinates\n");
        return nullptr;
    }
    const long cen = low + lim;
    char keybuf[64];
    snprintf(keybuf, sizeof keybuf, "|%d|%d,%d,%d,%d|%d", device, match, mismatch, gap_open, gap_extend, f16 ? 1 : 0);
    const std::string key = ad_lo + "|" + ad_hi + keybuf;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_cache.find(key);
    if (it != g_cache.end()) return it->second;
    // a hiprtc compile costs 0.3-1 s and halves the scan: compile once the work seen for this
    // adapter pair, over all launches so far, would have paid for it (PC_JIT_MIN_CELLS overrides
    // the 1e11-cell default)
    static const double min_cells = [] { const char *v = getenv("PC_JIT_MIN_CELLS"); return v ? atof(v) : 1e11; }();
    double &seen = g_seen[key];
    seen += cells;
    if (seen < min_cells) return nullptr;
    g_cache[key] = nullptr;
    Rtc &r = rtc();
    if (!r.ok) {
        static bool told = false;
        if (!told) { fprintf(stderr, "porechop_amd: hiprtc not available, using the generic scan kernels\n"); told = true; }
        return nullptr;
    }
    // letters per register row of each half (bottom-aligned): 0..4 = Dna5 code, 5 = padding row
    std::vector<int> lo(R, 5), hi(R, 5);
    for (int i = 0; i < m_lo; ++i) lo[R - m_lo + i] = dna5((unsigned char)ad_lo[i]);
    for (int i = 0; i < m_hi; ++i) hi[R - m_hi + i] = dna5((unsigned char)ad_hi[i]);
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
    // Register budget: T, U and two sets of substitution terms.  Up to 2R + 2K = 120 the kernel fits
    // 256 VGPRs (two waves per SIMD); longer adapters (full barcode sequences, 63-111 bases) get the
    // whole register file of a SIMD -- one wave, the rows beyond 256 VGPRs parked in AGPRs -- which
    // is still several times faster than the generic kernel's column in LDS.
    const int waves = (2 * R + 2 * K <= 120) ? 2 : 1;

    std::string init;
    for (int row = 0; row < R; ++row) { init += std::to_string(combo_of_row[row]); if (row + 1 < R) init += ","; }
    const std::string dR = "-DPC_R=" + std::to_string(R), dK = "-DPC_K=" + std::to_string(K),
                      dC = "-DPC_COMBO_INIT=" + init, dF = std::string("-DPC_F16=") + (f16 ? "1" : "0"),
                      dE = "-DPC_EPS=" + std::to_string(eps), dO = "-DPC_OE=(" + std::to_string(gap_open + eps) + ")",
                      dN = "-DPC_CEN=(" + std::to_string(cen) + ")", dP = "-DPC_KREN=" + std::to_string(kren),
                      dW = std::string("-DPC_WAVES=") + (getenv("PC_JIT_WAVES") ? getenv("PC_JIT_WAVES") : std::to_string(waves));
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", dR.c_str(), dK.c_str(), dC.c_str(), dF.c_str(),
                          dE.c_str(), dO.c_str(), dN.c_str(), dP.c_str(), dW.c_str()};
    hiprtcProgram prog = nullptr;
    if (r.CreateProgram(&prog, kSpecSource, "pc_spec_score.hip", 0, nullptr, nullptr) != 0) return nullptr;
    const hiprtcResult rc = r.CompileProgram(prog, 12, opts);
    if (rc != 0) {
        size_t n = 0;
        r.GetProgramLogSize(prog, &n);
        std::string log(n + 1, '\0');
        if (n) r.GetProgramLog(prog, &log[0]);
        fprintf(stderr, "porechop_amd: hiprtc compile failed (%d), using the generic kernels\n%s\n", rc, log.c_str());
        r.DestroyProgram(&prog);
        return nullptr;
    }
    size_t csz = 0;
    r.GetCodeSize(prog, &csz);
    std::vector<char> code(csz);
    r.GetCode(prog, code.data());
    r.DestroyProgram(&prog);

    Spec *sp = new Spec();
    sp->R = R; sp->K = K; sp->m_lo = m_lo; sp->m_hi = m_hi; sp->f16 = f16; sp->waves = waves;
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    if (hipModuleLoadData(&mod, code.data()) != hipSuccess || hipModuleGetFunction(&fn, mod, "pc_spec_score") != hipSuccess) {
        fprintf(stderr, "porechop_amd: loading the specialised kernel failed, using the generic kernels\n");
        delete sp;
        return nullptr;
    }
    sp->module = mod; sp->function = fn;
    // S table: [256 read bytes][K letter pairs] of packed (sub_lo - open + eps) | (sub_hi - open + eps) << 16
    std::vector<uint32_t> tab((size_t)256 * K, 0);
    auto term = [&](int letter, int code) -> int {
        const int sub = letter == 5 ? 0 : (letter == code ? match : mismatch);
        return sub - gap_open + eps;
    };
    for (int b = 0; b < 256; ++b) {
        const int code = dna5((unsigned char)b);
        for (size_t k = 0; k < combos.size(); ++k) {
            const int l = term(combos[k] / 6, code), h = term(combos[k] % 6, code);
            tab[(size_t)b * K + k] = f16 ? (half_bits(l) | (half_bits(h) << 16))
                                         : (((uint32_t)l & 0xFFFFu) | ((uint32_t)h << 16));
        }
    }
    void *d = nullptr;
    if (hipMalloc(&d, tab.size() * 4) != hipSuccess ||
        hipMemcpy(d, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        delete sp;
        return nullptr;
    }
    sp->d_table = d;
    if (verbose) fprintf(stderr, "porechop_amd: specialised kernel R=%d K=%d f16=%d kren=%ld\n", R, K, f16 ? 1 : 0, kren);
    g_cache[key] = sp;
    return sp;
}

int launch(const Spec *sp, const SpecArgs &a, int grid, void *stream)
{
    SpecArgs args = a;
    args.s_table = (const uint32_t *)sp->d_table;
    args.m_lo = sp->m_lo; args.m_hi = sp->m_hi;
    void *params[] = {&args};
    const hipError_t e = hipModuleLaunchKernel((hipFunction_t)sp->function, (unsigned)grid, 1, 1, 64, 1, 1, 0,
                                               (hipStream_t)stream, params, nullptr);
    return e == hipSuccess ? 0 : -2;
}

}  // namespace pcj
