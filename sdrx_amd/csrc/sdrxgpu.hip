// sdrxgpu.hip -- host side of libsdrxgpu: the C ABI of include/sdrx_gpu.h over the gfx950
// kernels of kernels.hpp.  No torch, no CPU compute path: every sample of the channelizer
// goes through the HIP kernels or the call fails.
#include "../../include/sdrx_gpu.h"
#include "kernels.hpp"
#include "kernels_mfma.hpp"
#include "kernels_mfma3.hpp"
#include "kernels_mfma6.hpp"
#include "kernels_tailagc.hpp"

#include <algorithm>
#include <complex>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

using namespace sdrxgpu;

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(SDRXGPU_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                  \
    } while (0)

extern "C" const char *sdrxgpu_strerror(int code)
{
    switch (code) {
        case SDRXGPU_OK:                return "ok";
        case SDRXGPU_ERR_INVALID_ARG:   return "invalid argument";
        case SDRXGPU_ERR_NO_DEVICE:     return "no usable gfx950 device";
        case SDRXGPU_ERR_NO_MEM:        return "out of memory";
        case SDRXGPU_ERR_HIP:           return "HIP runtime error";
        case SDRXGPU_ERR_UNSUPPORTED:   return "stage chain not supported by the compiled kernels";
        case SDRXGPU_ERR_OUT_TOO_SMALL: return "out_stride smaller than the outputs of this call";
        default:                        return "unknown error";
    }
}

extern "C" const char *sdrxgpu_last_error(void) { return g_err; }
extern "C" const char *sdrxgpu_version(void) { return "sdrxgpu 0.1 (gfx950)"; }

// ------------------------------------------------------------------------------------------
// host-side set-up helpers (reference src/sdrx.cpp:1682-1710)
// ------------------------------------------------------------------------------------------
extern "C" void sdrxgpu_agc_params_sdrx(sdrxgpu_agc_params *agc, sdrxgpu_agc_params *lf)
{
    // Channel's AGC/LfAGC members are default-constructed (agc.hpp:30, :85) and then adjusted by
    // setters (sdrx.cpp:1702-1710); setReference leaves power_ at its constructor value.
    if (agc) {
        agc->attack = 1.0f;
        agc->decay = 0.01f;
        agc->reference = 1.0f;
        agc->max_gain = 300.0f;
        agc->alpha = 0.01f;
        agc->init_gain = 1.0f;
        agc->init_power = 0.25f;
    }
    if (lf) {
        lf->attack = 1.0f;
        lf->decay = 0.01f;
        lf->reference = 1.0f;
        lf->max_gain = 200.0f;
        lf->alpha = 0.005f;
        lf->init_gain = 1.0f;
        lf->init_power = 0.25f * 0.25f;
    }
}

extern "C" int sdrxgpu_make_translator(int N, int z, int ch_offset, sdrxgpu_iq *out)
{
    if (N <= 0 || !out) return fail(SDRXGPU_ERR_INVALID_ARG, "make_translator: N=%d out=%p", N, (void *)out);
    for (int n = 0; n < N; n++) {
        // same expression shape as sdrx.cpp:1687: double phase, narrowed to float, then cexpf
        const double phase = -2.0f * M_PI * n * ch_offset * (float)z / (float)N;
        const std::complex<float> t = std::exp(std::complex<float>(0.0f, (float)phase));
        out[n].re = t.real();
        out[n].im = t.imag();
    }
    return SDRXGPU_OK;
}

// ------------------------------------------------------------------------------------------
// channel names -> offsets (reference src/sdrx.cpp:957-1021, 1398-1408, 1424-1478)
// ------------------------------------------------------------------------------------------
// the step (0..11) and the frequency inside a 100 kHz band that the last two digits of a channel name stand for
// (sub_ch_map, sdrx.cpp:984-989 and 1461-1466: both tables, one row per key)
static bool sub_channel(const char *two, int *step, unsigned *hz)
{
    static const struct { const char *key; int step; unsigned hz; } map[] = {
        {"00", 0, 0},      {"05", 0, 0},      {"10", 1, 8333},   {"15", 2, 16667},
        {"25", 3, 25000},  {"30", 3, 25000},  {"35", 4, 33333},  {"40", 5, 41667},
        {"50", 6, 50000},  {"55", 6, 50000},  {"60", 7, 58333},  {"65", 8, 66667},
        {"75", 9, 75000},  {"80", 9, 75000},  {"85", 10, 83333}, {"90", 11, 91667}};
    for (const auto &e : map)
        if (e.key[0] == two[0] && e.key[1] == two[1]) {
            if (step) *step = e.step;
            if (hz) *hz = e.hz;
            return true;
        }
    return false;
}

// "NNN.NNN": 2-4 digits, a dot, exactly three digits, a known pair at the end (parse_fq's aeronautical branch)
static bool split_channel(const char *name, long *mhz, int *tenth, const char **two)
{
    if (!name) return false;
    const char *dot = strchr(name, '.');
    if (!dot) return false;
    const size_t nint = (size_t)(dot - name), nfrac = strlen(dot + 1);
    if (nint < 2 || nint > 4 || nfrac != 3) return false;
    for (const char *p = name; *p; p++)
        if (p != dot && (*p < '0' || *p > '9')) return false;
    *mhz = atol(std::string(name, nint).c_str());
    *tenth = dot[1] - '0';
    *two = dot + 2;
    return sub_channel(*two, nullptr, nullptr);
}

extern "C" int sdrxgpu_parse_channel(const char *name, uint32_t *fq_hz)
{
    long mhz;
    int tenth;
    const char *two;
    unsigned hz = 0;
    if (!fq_hz || !split_channel(name, &mhz, &tenth, &two) || mhz >= 4000)
        return fail(SDRXGPU_ERR_INVALID_ARG, "parse_channel: '%s' is not an aeronautical channel name", name ? name : "(null)");
    sub_channel(two, nullptr, &hz);
    *fq_hz = (uint32_t)(mhz * 1000000 + tenth * 100000 + (long)hz);     // sdrx.cpp:990-994, 1014-1016
    return SDRXGPU_OK;
}

static int lowest_highest(const char *const *names, size_t n, uint32_t *lo_fq, uint32_t *hi_fq)
{
    if (!names || n == 0) return fail(SDRXGPU_ERR_INVALID_ARG, "no channels");
    // std::sort over Channel::operator< = std::string order of the names (sdrx.cpp:212, 1398-1401)
    std::string lo = names[0] ? names[0] : "", hi = lo;
    for (size_t i = 0; i < n; i++) {
        if (!names[i]) return fail(SDRXGPU_ERR_INVALID_ARG, "null channel name");
        const std::string s = names[i];
        if (s < lo) lo = s;
        if (hi < s) hi = s;
    }
    int rc;
    if ((rc = sdrxgpu_parse_channel(lo.c_str(), lo_fq)) || (rc = sdrxgpu_parse_channel(hi.c_str(), hi_fq))) return rc;
    return SDRXGPU_OK;
}

extern "C" int sdrxgpu_tuner_fq(const char *const *names, size_t n, uint32_t *tuner_fq)
{
    uint32_t lo, hi;
    if (!tuner_fq) return fail(SDRXGPU_ERR_INVALID_ARG, "tuner_fq: null argument");
    const int rc = lowest_highest(names, n, &lo, &hi);
    if (rc) return rc;
    const double lo_fq = (double)lo, hi_fq = (double)hi;
    const double mid_fq = lo_fq + (hi_fq - lo_fq) / 2;                    // sdrx.cpp:1403-1408
    *tuner_fq = (uint32_t)(std::round(mid_fq / 100000.0) * 100000.0);
    return SDRXGPU_OK;
}

extern "C" int sdrxgpu_channel_to_offset(const char *name, int32_t tuner_fq, int *offset)
{
    long mhz;
    int tenth, step = 0;
    const char *two;
    if (!offset || !split_channel(name, &mhz, &tenth, &two))
        return fail(SDRXGPU_ERR_INVALID_ARG, "channel_to_offset: '%s' is not an aeronautical channel name", name ? name : "(null)");
    sub_channel(two, &step, nullptr);
    int32_t fq_base = (int32_t)(mhz * 1000000);                           // sdrx.cpp:1470-1475, int32 arithmetic as there
    fq_base += tenth * 100000;
    const int32_t fq_diff = fq_base - tuner_fq;
    const int32_t offset_diff = (fq_diff / 100000) * 12;
    *offset = offset_diff + step;
    return SDRXGPU_OK;
}

extern "C" int sdrxgpu_channels_fit_bandwidth(const char *const *names, size_t n, uint32_t sample_rate_hz, int *fits)
{
    uint32_t lo, hi;
    if (!fits) return fail(SDRXGPU_ERR_INVALID_ARG, "channels_fit_bandwidth: null argument");
    const int rc = lowest_highest(names, n, &lo, &hi);
    if (rc) return rc;
    const uint32_t ch_bw = hi - lo;
    const uint32_t iq_bw = sample_rate_hz * 8 / 10;                       // 80 % of the sample rate, sdrx.cpp:1437 (uint32, as there)
    *fits = ch_bw <= iq_bw ? 1 : 0;
    return SDRXGPU_OK;
}

// ------------------------------------------------------------------------------------------
// stage-1/2 kernel registry: one entry per (M1,T1,M2,T2) the library was built for
// ------------------------------------------------------------------------------------------
struct K12Launch {
    const float2 *x, *tables;
    float2 *y2, *y3, *seam;
    Stage12Args args;
    const float *h1, *h2, *h3;
    int W, CPW;
    int grid;
};

typedef hipError_t (*k12_launch_fn)(const K12Launch &, hipStream_t);

// level 0: every (W, CPW) geometry is compiled (the chains of BASELINE.json's configs); level 1: the four that
// matter for large and small banks; level 2 (big first-stage decimation, register-heavy): W = 4 only.
static bool k12_geometry_compiled(int level, int W, int CPW)
{
    if (level == 0) return (W == 16 || W == 8 || W == 4) && (CPW == 4 || CPW == 2 || CPW == 1);
    if (level == 1) return (CPW == 4 && (W == 16 || W == 8 || W == 4)) || (W == 4 && CPW == 1);
    return W == 4 && CPW == 4;      // (4 waves x 4 channels: the only geometry whose two tile buffers fit 160 KB of LDS)
}

// MA > 0: the variant with stage 3 (MA, TA) fused behind stages 1-2 (kernels.hpp, k_stage12)
template <int M1, int T1, int M2, int T2, int Y2B, int LEVEL, int MA = 0, int TA = 1>
static hipError_t launch_k12(const K12Launch &L, hipStream_t s)
{
    Stage12Taps<T1, T2, TA> taps{};
    for (int i = 0; i < (T1 + 1) / 2; i++) taps.h1[i] = L.h1[i];
    for (int i = 0; i < (T2 + 1) / 2; i++) taps.h2[i] = L.h2[i];
    if constexpr (MA > 0)
        for (int i = 0; i < (TA + 1) / 2; i++) taps.h3[i] = L.h3[i];
#define SDRX_K12_CASE(W_, C_)                                                                                  \
    if (L.W == W_ && L.CPW == C_) {                                                                             \
        hipLaunchKernelGGL((k_stage12<M1, T1, M2, T2, Y2B, W_, C_, MA, TA>), dim3(L.grid), dim3(64 * W_), 0, s, \
                           L.x, L.tables, L.y2, L.y3, L.seam, L.args, taps);                                            \
        return hipGetLastError();                                                                               \
    }
    SDRX_K12_CASE(4, 4)
    if constexpr (LEVEL <= 1) {
        SDRX_K12_CASE(16, 4) SDRX_K12_CASE(8, 4) SDRX_K12_CASE(4, 1)
    }
    if constexpr (LEVEL == 0) {
        SDRX_K12_CASE(16, 2) SDRX_K12_CASE(16, 1) SDRX_K12_CASE(8, 2) SDRX_K12_CASE(8, 1) SDRX_K12_CASE(4, 2)
    }
#undef SDRX_K12_CASE
    return hipErrorInvalidValue;
}

// two channels per lane (k_stage12p): compiled for the 10 MS/s chain, W = 4, CPW = 4
struct K12PairLaunch {
    K12Launch L;
    PairArgs pa;
    bool fused;
    size_t lds;
};
constexpr int kPairW = 4, kPairCPW = 4, kPairSlots = kPairW * kPairCPW;

template <int MA, int TA>
static hipError_t launch_k12p_one(const K12PairLaunch &P, hipStream_t s)
{
    Stage12Taps<17, 19, TA> taps{};
    for (int i = 0; i < 9; i++) taps.h1[i] = P.L.h1[i];
    for (int i = 0; i < 10; i++) taps.h2[i] = P.L.h2[i];
    if constexpr (MA > 0)
        for (int i = 0; i < (TA + 1) / 2; i++) taps.h3[i] = P.L.h3[i];
    hipLaunchKernelGGL((k_stage12p<17, 19, 4, kPairW, kPairCPW, MA, TA>), dim3(P.L.grid), dim3(64 * kPairW), P.lds, s,
                       P.L.x, P.L.tables, P.L.y2, P.L.y3, P.L.seam, P.L.args, P.pa, taps);
    return hipGetLastError();
}

static hipError_t launch_k12p(const K12PairLaunch &P, hipStream_t s)
{
    return P.fused ? launch_k12p_one<5, 21>(P, s) : launch_k12p_one<0, 1>(P, s);
}

static size_t k12p_lds_bytes() { return 2 * (size_t)(kWave / kPairCPW + 2 * kPairSlots) * (100 * 8 + 16); }   // = the tiles of general groups

static hipError_t k12p_set_attributes()
{
    const int n = (int)k12p_lds_bytes();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stage12p<17, 19, 4, kPairW, kPairCPW, 5, 21>), hipFuncAttributeMaxDynamicSharedMemorySize, n);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stage12p<17, 19, 4, kPairW, kPairCPW, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, n);
    return e;
}

// mixer + stage 1 on the f32 matrix pipe (k_stage12m, kernels_mfma.hpp): 10 MS/s chain; workgroup = GS x GT tiles
struct K12mLaunch {
    K12Launch L;
    PairArgs pa;
    int GS, GT;
    int kind;            // 0: k_stage12m (10 MS/s chain, (5, 17)(5, 19)); 1: k_stage12m6 (6 MS/s chain, (15, 53)(5, 21))
};

static bool k12m_geometry_compiled(int kind, int GS, int GT)
{
    if (kind == 1) return GS == 2 && GT == 2;
    return (GS == 2 && GT == 2) || (GS == 4 && GT == 2) || (GS == 2 && GT == 4);
}
static size_t k12m_lds_bytes(int kind, int GS, int GT)
{
    if (kind == 1) return 2 * (size_t)TileGeomM6<2, 2>::buf_bytes;
    return 2 * (size_t)(16 * GS + 16 * GT) * 528;
}

template <int GS, int GT>
static hipError_t launch_k12m_one(const K12mLaunch &M, const Stage12mTaps &taps, hipStream_t s)
{
    using TG = TileGeomM<GS, GT>;
    static_assert(2 * TG::buf_bytes <= 160 * 1024, "two tile buffers must fit the LDS");
    hipLaunchKernelGGL((k_stage12m<GS, GT>), dim3(M.L.grid), dim3(64 * GS * GT), 2 * TG::buf_bytes, s,
                       M.L.x, M.L.tables, M.L.y2, M.L.args, M.pa, taps);
    return hipGetLastError();
}

static hipError_t launch_k12m(const K12mLaunch &M, hipStream_t s)
{
    if (M.kind == 1) {
        Stage12m6Taps t6{};
        for (int i = 0; i < 53; i++) t6.h1[i] = M.L.h1[i < 27 ? i : 52 - i];   // (the reference reads h[0 .. half] only, msd.hpp:309-331)
        for (int i = 0; i < 11; i++) t6.h2[i] = M.L.h2[i];
        using TG = TileGeomM6<2, 2>;
        static_assert(2 * TG::buf_bytes <= 160 * 1024, "two tile buffers must fit the LDS");
        hipLaunchKernelGGL((k_stage12m6<2, 2>), dim3(M.L.grid), dim3(64 * TG::waves), 2 * TG::buf_bytes, s, M.L.x, M.L.tables, M.L.y2, M.L.args, M.pa, t6);
        return hipGetLastError();
    }
    Stage12mTaps taps{};
    for (int i = 0; i < 17; i++) taps.h1[i] = M.L.h1[i < 9 ? i : 16 - i];     // (the reference reads h[0 .. half] only, msd.hpp:309-331)
    for (int i = 0; i < 10; i++) taps.h2[i] = M.L.h2[i];
    if (M.GS == 2 && M.GT == 2) return launch_k12m_one<2, 2>(M, taps, s);
    if (M.GS == 4 && M.GT == 2) return launch_k12m_one<4, 2>(M, taps, s);
    if (M.GS == 2 && M.GT == 4) return launch_k12m_one<2, 4>(M, taps, s);
    return hipErrorInvalidValue;
}

static hipError_t k12m_set_attributes(int kind, int GS, int GT)
{
    if (kind == 1)
        return hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stage12m6<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)k12m_lds_bytes(1, 2, 2));
    const int n = (int)k12m_lds_bytes(0, GS, GT);
    const void *f = nullptr;
    if (GS == 2 && GT == 2) f = reinterpret_cast<const void *>(&k_stage12m<2, 2>);
    if (GS == 4 && GT == 2) f = reinterpret_cast<const void *>(&k_stage12m<4, 2>);
    if (GS == 2 && GT == 4) f = reinterpret_cast<const void *>(&k_stage12m<2, 4>);
    if (!f) return hipErrorInvalidValue;
    return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, n);
}

// mixer + stages 1-3 without a stage-2 scratch (k_stage123m, kernels_mfma3.hpp): same workgroup shapes as k_stage12m
struct K123mLaunch {
    const float2 *x, *tables;
    float2 *y3c;
    Stage123Args args;
    PairArgs pa;
    const float *h1;
    int GS, GT, grid;
};

static size_t k123m_lds_bytes(int GS, int GT)
{
    if (GS == 2 && GT == 2) return TileGeomM3<2, 2>::lds_bytes;
    if (GS == 4 && GT == 2) return TileGeomM3<4, 2>::lds_bytes;
    return TileGeomM3<2, 4>::lds_bytes;
}

template <int GS, int GT>
static hipError_t launch_k123m_one(const K123mLaunch &M, const Stage123mTaps &taps, hipStream_t s)
{
    using TG = TileGeomM3<GS, GT>;
    static_assert(TG::lds_bytes <= 160 * 1024, "tile buffers + coefficient table must fit the LDS");
    hipLaunchKernelGGL((k_stage123m<GS, GT>), dim3(M.grid), dim3(64 * GS * GT), TG::lds_bytes, s, M.x, M.tables, M.y3c, M.args, M.pa, taps);
    return hipGetLastError();
}

static hipError_t launch_k123m(const K123mLaunch &M, hipStream_t s)
{
    Stage123mTaps taps{};
    for (int i = 0; i < 17; i++) taps.h1[i] = M.h1[i < 9 ? i : 16 - i];       // (the reference reads h[0 .. half] only, msd.hpp:309-331)
    if (M.GS == 2 && M.GT == 2) return launch_k123m_one<2, 2>(M, taps, s);
    if (M.GS == 4 && M.GT == 2) return launch_k123m_one<4, 2>(M, taps, s);
    if (M.GS == 2 && M.GT == 4) return launch_k123m_one<2, 4>(M, taps, s);
    return hipErrorInvalidValue;
}

static hipError_t k123m_set_attributes(int GS, int GT)
{
    const void *f = nullptr;
    if (GS == 2 && GT == 2) f = reinterpret_cast<const void *>(&k_stage123m<2, 2>);
    if (GS == 4 && GT == 2) f = reinterpret_cast<const void *>(&k_stage123m<4, 2>);
    if (GS == 2 && GT == 4) f = reinterpret_cast<const void *>(&k_stage123m<2, 4>);
    if (!f) return hipErrorInvalidValue;
    return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)k123m_lds_bytes(GS, GT));
}

struct K12Entry {
    int M1, T1, M2, T2, Y2B;
    int level;
    k12_launch_fn fn;
    // the chain's tail stages when stage 3 can run fused (fn_fused != nullptr): (MA, TA) inside k_stage12, (MB, TB) in k_tail4
    int MA, TA, MB, TB;
    k12_launch_fn fn_fused;
};

// one entry per first-two-stages pair of the reference's chains (sdrx.cpp:1588-1668)
static const K12Entry k12_table[] = {
    {5, 17, 5, 19, 4, 0, launch_k12<5, 17, 5, 19, 4, 0>, 5, 21, 5, 67, launch_k12<5, 17, 5, 19, 4, 0, 5, 21>},   // 10 MS/s    (sdrx.cpp:1660-1668)
    {2, 9, 3, 13, 4, 0, launch_k12<2, 9, 3, 13, 4, 0>, 5, 19, 5, 45, launch_k12<2, 9, 3, 13, 4, 0, 5, 19>},       // 2.4 MS/s   (sdrx.cpp:1629-1637)
    {3, 13, 4, 25, 4, 1, launch_k12<3, 13, 4, 25, 4, 1>, 0, 0, 0, 0, nullptr},      // 0.96 MS/s  (sdrx.cpp:1589-1596)
    {3, 13, 5, 19, 4, 1, launch_k12<3, 13, 5, 19, 4, 1>, 0, 0, 0, 0, nullptr},      // 1.2 MS/s   (sdrx.cpp:1597-1604)
    {3, 13, 6, 23, 4, 1, launch_k12<3, 13, 6, 23, 4, 1>, 0, 0, 0, 0, nullptr},      // 1.44 MS/s  (sdrx.cpp:1605-1612)
    {4, 15, 5, 19, 4, 1, launch_k12<4, 15, 5, 19, 4, 1>, 0, 0, 0, 0, nullptr},      // 1.6 MS/s   (sdrx.cpp:1613-1620)
    {4, 15, 6, 23, 4, 1, launch_k12<4, 15, 6, 23, 4, 1>, 0, 0, 0, 0, nullptr},      // 1.92 MS/s  (sdrx.cpp:1621-1628)
    {15, 53, 5, 21, 2, 2, launch_k12<15, 53, 5, 21, 2, 2>, 0, 0, 0, 0, nullptr},    // 6 MS/s     (sdrx.cpp:1651-1658)
    {20, 51, 4, 33, 2, 2, launch_k12<20, 51, 4, 33, 2, 2>, 0, 0, 0, 0, nullptr},    // 2.56 MS/s  (sdrx.cpp:1641-1648; N % m != 0, sdrx.cpp:1642)
};

static const K12Entry *find_k12(int M1, int T1, int M2, int T2)
{
    for (const K12Entry &e : k12_table)
        if (e.M1 == M1 && e.T1 == T1 && e.M2 == M2 && e.T2 == T2) return &e;
    return nullptr;
}

// k_tail_agc (kernels_tailagc.hpp) for the tail of a chain: the kernel and the LDS of its fixed part (the index table comes on top)
typedef void (*TailAgcKernel)(const TailAgcArgs);
template <int NT>
static bool find_tail_agc_nt(int mA, int tA, int mB, int tB, bool sq, bool fm, TailAgcKernel *k, size_t *lds)
{
#define SDRX_TA_PICK(MA_, TA_, MB_, TB_)                                                                                  \
    if (mA == MA_ && tA == TA_ && mB == MB_ && tB == TB_) {                                                               \
        *k = sq ? (fm ? &k_tail_agc<MA_, TA_, MB_, TB_, true, true, NT> : &k_tail_agc<MA_, TA_, MB_, TB_, true, false, NT>) \
                : (fm ? &k_tail_agc<MA_, TA_, MB_, TB_, false, true, NT> : &k_tail_agc<MA_, TA_, MB_, TB_, false, false, NT>); \
        *lds = fm ? tail_agc_lds_bytes<MA_, TA_, MB_, TB_, true, NT>() : tail_agc_lds_bytes<MA_, TA_, MB_, TB_, false, NT>(); \
        return true;                                                                                                      \
    }
    SDRX_TA_PICK(5, 21, 5, 67)     // 10 MS/s
    SDRX_TA_PICK(5, 19, 5, 45)     // 2.4 MS/s
    SDRX_TA_PICK(0, 0, 5, 45)      // 0.96 / 1.2 / 1.44 / 1.6 / 1.92 MS/s
    SDRX_TA_PICK(0, 0, 5, 67)      // 6 MS/s
    SDRX_TA_PICK(0, 0, 2, 21)      // 2.56 MS/s
#undef SDRX_TA_PICK
    return false;
}
// nt: producer waves per workgroup (1, 2 or 4)
static bool find_tail_agc(int nt, int mA, int tA, int mB, int tB, bool sq, bool fm, TailAgcKernel *k, size_t *lds)
{
    if (nt == 1) return find_tail_agc_nt<1>(mA, tA, mB, tB, sq, fm, k, lds);
    if (nt == 2) return find_tail_agc_nt<2>(mA, tA, mB, tB, sq, fm, k, lds);
    if (nt == 4) return find_tail_agc_nt<4>(mA, tA, mB, tB, sq, fm, k, lds);
    return false;
}

static long long gcd_ll(long long a, long long b) { return b ? gcd_ll(b, a % b) : a; }

// Banks alive per device.  The one thing it decides is a choice between two bit-identical kernel sequences (k_tail_agc or
// k_tail_s + k_agc, launch_tail): a bank ALONE on its GPU is 4 % faster per block with the fused kernel, two banks sharing a GPU
// (BASELINE configs[4]) are 12 % slower with it -- its long-lived workgroups get in the way of the other bank's stage-1/2 launch,
// under which the two short kernels hide anyway (0.219 against 0.194 ms per step of both banks).  The choice is made ONCE per bank
// (round 5 looked at the counter at every launch: the path could then change between two calls of one stream, and bank_info said
// nothing of it): provisionally in bank_create, finally when the bank's FIRST block is processed -- an application creates its
// banks, then streams (sdrx.cpp:1680-1714, then data_cb), so that is the moment all of a device's banks exist: BASELINE
// configs[4]'s two banks per GPU both keep the two kernels, whichever was created first.  From then on the bank keeps its
// sequence whatever is created or destroyed beside it, and sdrxgpu_bank_info.tail_agc says which.
#include <atomic>
static std::atomic<int> g_banks_on_device[64];

// k_stage12m's scratch stores: the last SDRX_Y2_KEEP_MB of a launch's stage-2 outputs stay ordinary stores (they are still in the
// 256 MB Infinity Cache when the tail kernel reads them, which it does first), everything before them streams (launch_k12_parts)
#ifndef SDRX_Y2_KEEP_MB
#define SDRX_Y2_KEEP_MB 192
#endif

// k_stage12m6 (the 6 MS/s chain's matrix pipe) is a caller's choice, tuning.matrix_pipe = 2, never automatic: on the stress fixtures
// of tests/golden/pairs6 (a carrier of 0.9 of full scale over a floor of 0.05) the carrier's conjugate partner comes out 1.4e-5 from
// the reference's portable build where the reference's own two builds are 0.42e-5 apart (3.3 x their spread, audio 3.4 x; k_stage12m holds 2.1 x
// on the 10 MS/s chain's fixtures, and 1e-5 wherever the builds agree to 0.5e-5) -- tests/test_gpu_matrix6.py.  Where it pays, ms
// per block, vector / matrix (tools/lab/bench_m6_lists.py): 512 channels 0.091 / 0.123, 768: 0.121 / 0.124, 1 024: 0.130 / 0.106,
// 2 048: 0.219 / 0.158, 4 096: 0.393 / 0.281, 10 000: 0.929 / 0.575; without partners 2 048: 0.217 / 0.201, 10 000: 0.924 / 0.952;
// half of the channels paired, 4 096: 0.393 / 0.364 -- from 1 024 channels if at most 80 % as many slots as channels are needed.

// k_tail_agc is taken automatically by banks of SDRX_TAILAGC_AUTO_MIN_CU .. SDRX_TAILAGC_AUTO_MAX_CU channels per CU of the device
// (256 .. 1 024 channels on an MI355X): measured against k_tail_s + k_agc, ms per block, bench.py's channel lists: 256 channels
// 0.0753 / 0.0764, 384: 0.0840 / 0.0869, 512: 0.1003 / 0.1052, 768: 0.1275 / 0.1307, BASELINE configs[2]'s 1 024: 0.1276 / 0.1329; below
// (64: 0.0560 / 0.0536, 128: 0.0611 / 0.0596, 192: 0.0681 / 0.0669) the two kernels' own chains are shorter than its pipeline's fill,
// above (1 280: 0.184 / 0.165, 2 048: 0.256 / 0.237) its workgroups no longer fit the chip at once (profiles/r05_tail_agc.txt)
#ifndef SDRX_TAILAGC_AUTO_MIN_CU
#define SDRX_TAILAGC_AUTO_MIN_CU 1
#endif
#ifndef SDRX_TAILAGC_AUTO_MAX_CU
#define SDRX_TAILAGC_AUTO_MAX_CU 4
#endif

// ------------------------------------------------------------------------------------------
// the bank
// ------------------------------------------------------------------------------------------
// Every entry that touches a bank makes the bank's device current for its own duration and puts the caller's back on return: a
// thread that drives banks on several devices, or does HIP work of its own, keeps the current device it chose.
struct DeviceGuard {
    int prev = -1, dev;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int d) : dev(d)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) err = hipSetDevice(dev);
    }
    ~DeviceGuard()
    {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

struct ProfSpan {
    hipEvent_t a, b;
    int kernel;
};

struct sdrxgpu_bank {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;

    // configuration
    uint32_t n_ch = 0, n_stages = 0, flags = 0;
    std::vector<uint32_t> m, ntaps;
    std::vector<std::vector<float>> taps;
    uint32_t N = 0;                  // translator length (0: nobody mixes)
    uint32_t m_total = 1;
    sdrxgpu_agc_params agc{}, lf{};

    // stage-1/2 geometry
    const K12Entry *k12 = nullptr;
    int MM = 1, SB = 1, WU = 1, HIST = 0;
    int Neff = 0;                    // table length on the device (N, or SB*12 when nobody mixes)
    int spacing = 0, bps = 0;        // lane spacing in samples, bodies per segment
    int nb = 0, P = 0, W = 16, CPW = 4;
    int tbl_stride = 0;
    int y2_hist = 0, y2_stride = 0;
    uint32_t max_chunk = 0;
    size_t x_capacity = 0;           // samples
    int keep_stride = 0;

    // device memory
    // two input rings [history | block | slack], used alternately: while the kernels of block k read ring k & 1, block
    // k+1 may already be arriving (H2D copy, RCCL broadcast) in the other one; the history of block k is copied across
    float2 *d_xr[2] = {nullptr, nullptr};
    int xr = 0;                               // ring the NEXT chunk is read from
    float2 *d_tables = nullptr, *d_y2 = nullptr, *d_keep = nullptr;
    float2 *d_y2b = nullptr;                  // second stage-2 scratch (pipelined mode only)
    // stage 3 fused behind stages 1-2 (k_stage12<.., MA, TA> + k_tail4): interior stage-3 outputs of the call, and the
    // last H2 stage-2 / H3 stage-3 outputs of the stream instead of d_hist
    bool fused = false;
    float2 *d_y3 = nullptr, *d_hist2 = nullptr, *d_hist3 = nullptr, *d_seam = nullptr;
    int H2 = 0, H3 = 0, y3_stride = 0, seam_stride = 0;
    size_t tail4_lds = 0;
    float2 *d_hist[2] = {nullptr, nullptr};   // stage-2 history, ping-pong between calls
    int hist_cur = 0;
    // tail stages + AGC/demod as ONE kernel (k_tail_agc, kernels_tailagc.hpp): banks on the separate tail whose tail is a latency chain
    bool tail_agc = false;
    bool tail_agc_required = false;           //   tuning.tail_agc = 2: whoever else is on the device
    bool tail_agc_final = false;              //   the first block has been processed: the choice stands
    bool counted = false;                     // this bank is in g_banks_on_device
    int ta_cpg = 4;                           //   channels per workgroup of that kernel
    int ta_nt = 2;                            //   its producer waves per workgroup
    TailAgcKernel ta_kernel = nullptr;
    size_t ta_lds = 0;                        //   LDS of its fixed part
    uint32_t *d_gtab = nullptr;               //   where group g of a call's stage-2 outputs lies in a channel's scratch row
    int gtab_len = 0;
    int tile_b = 64;                          // final outputs per k_tail workgroup (64: 8 workgroups per CU)
    bool tail_generic = false;                // tuning.tail_generic: run-time-tap k_tail even for the compiled-in chains
    size_t tail_lds = 0;
    float *d_audio_keep = nullptr;
    float4 *d_state = nullptr;
    float *d_fir_taps = nullptr, *d_fir_hist = nullptr;   // optional audio FIR
    // optional squelch
    bool squelch = false;
    uchar2 *d_sql = nullptr;
    float *d_snr = nullptr, *d_sql_level = nullptr, *d_sq_tabs = nullptr;   // d_sq_tabs: window | passband | ramp_up | ramp_down
    float2 *d_tw = nullptr;
    int fir_T = 0;
    // optional per-channel FM demodulation
    unsigned char *d_fm = nullptr;
    float2 *d_fm_prev = nullptr;
    // optional stereo mix-down
    bool mix = false;
    float2 *d_pan_w = nullptr;
    float *d_mix_partial = nullptr, *d_mix = nullptr, *d_mix_f32 = nullptr, *d_mix_taps = nullptr, *d_mix_hist = nullptr;
    short *d_mix_s16 = nullptr;
    int mix_T = 0, mix_groups = 0, mix_frames = 0;
    size_t device_bytes = 0;

    // pipelined mode: stage-1/2 of block k runs on `stream2` while the tail stages + AGC of block k-1
    // run on `stream`; the tail of the newest block waits in `pending`
    bool ftfir = false;
    bool pipelined = false;
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_copied = nullptr;          // pipelined mode: the caller's input buffer has been copied into the ingest ring
    struct TailJob {
        bool valid = false;
        TailArgs t;
        int c0 = 0, n_y3 = 0, d0 = 0, n_bnd = 0;   // fused banks (Tail4Args)
        K12Launch L;                                // the stage-1/2 launch of this chunk (launch_k12_range)
        K123mLaunch M3;                             // banks on k_stage123m: its launch, and what k_tail_c needs of the call
        int n_y1 = 0, k25 = 0, r0m6 = 0, n_out3 = 0, n_seg = 0;
        const float2 *roll_from = nullptr;   // input-history copy left for k_agc to do (serial mode)
        float2 *roll_to = nullptr;
        int roll_cnt = 0;
    } pending;
    int y2_cur = 0;

    // host entries: copies on their own stream beside the kernels (sdrxgpu_bank_process: D2H of the first channel half
    // beside the kernels of the second; sdrxgpu_bank_process_pinned: H2D of block k+1 beside the kernels of block k)
    // two channels per lane (k_stage12p): slot lists {channel A, channel B or -1; A = -1: empty}, ordered by host part,
    // every part's share padded to whole workgroups of kPairSlots slots
    // mixer + stage 1 on the f32 matrix pipe (k_stage12m): slot list {channel, its conjugate partner or -1; {-1,-1}: empty},
    // ordered by host part, every part's share padded to whole workgroup sets of 16 * mGS slots
    bool mfma = false;
    int mfma_kind = 0;               // 0: k_stage12m (10 MS/s chain); 1: k_stage12m6 (6 MS/s chain)
    int mGS = 2, mGT = 2;
    // ... and stages 2-3 behind it in the same kernel, no stage-2 scratch (k_stage123m + k_tail_c, kernels_mfma3.hpp): the
    // coefficient table of the combined stage-2/3 filter, the stage-3 scratch [channel][segment group][entry][segment] and, per
    // channel, the last H3c stage-3 outputs of the stream followed by the six partial sums in flight (+ 2 of padding)
    bool mfma3 = false;
    float *d_wtab = nullptr;
    float2 *d_y3c = nullptr, *d_hist3c[2] = {nullptr, nullptr};    // (histories: ping-pong between calls, as d_hist)
    int y3c_stride = 0, H3c = 80;
    int2 *d_slots_m = nullptr, *d_slots_mp = nullptr;  // partners from the whole bank / from inside each host part
    int n_slots_m = 0, n_slots_mp = 0, n_pairs_m = 0;
    bool pair = false, want_pair = false;
    int2 *d_slots = nullptr;
    unsigned char *d_group_conj = nullptr;            // per slot group: conjugate partners (one translator row per slot) or any two channels
    int n_slots = 0, n_conj_pairs = 0;
    // Host parts (sdrxgpu_bank_process): part p = channels [ch[p], ch[p+1]); gm = set boundaries of k_stage12m's part launches in
    // the slot list sorted by first channel (d_slots_mp).  Two cuts of the same bank: [0] for calls that take the decimated IQ
    // (61 MB of D2H per 10 000-channel block: the copies are what takes longest, so they have to start early), [1] for calls that
    // take the audio only (a third of that: the kernels take longest, so as few parts as leave little to copy behind the last).
    struct HostPlan {
        std::vector<int> ch, gm;
    } plan[2];
    int cur_plan = 0;
    std::vector<int> part_g;                          // slot-group boundaries of the parts (two channels per lane: plan[0] only)
    hipStream_t copy_stream = nullptr;
    int host_parts = 0;                               // tuning.host_parts (0 = auto)
    size_t y2_keep = 0;                               // bytes of a k_stage12m launch's scratch (the sets written last) stored without the streaming hint
    hipEvent_t ev_part[8] = {};                       // kernels of a channel part are done
    hipEvent_t ev_iq[8] = {};                         // the tail kernel of a channel part is done (its decimated IQ rows are final)
    hipEvent_t mark_after_tail = nullptr;             // launch_tail records it between the tail kernel and the AGC kernel
    hipEvent_t ev_h2d = nullptr;                      // the block is in the input ring
    hipEvent_t ev_d2h = nullptr;                      // the output rows have left the keep buffers
    hipEvent_t ev_ring_free[2] = {nullptr, nullptr};  // the kernels that read input ring r are done
    bool ring_busy[2] = {false, false};
    bool h2d_pending = false;

    // stream position
    uint64_t n_in = 0, n_out = 0;
    long long x_origin = 0;          // absolute sample index held at d_xr[xr][0] (same parity as the next call's s0)

    // profiling
    int prof = 0;                    // 0 off, 1 every kernel + whole call, 2 the stage-1/2 kernel only
    std::vector<ProfSpan> prof_pending;
    std::vector<hipEvent_t> pool;
    double prof_ms[4] = {0, 0, 0, 0};
    uint64_t prof_n[4] = {0, 0, 0, 0};
};

static int dev_alloc(sdrxgpu_bank *b, void **p, size_t bytes)
{
    hipError_t e = hipMalloc(p, bytes ? bytes : 16);
    if (e != hipSuccess) return fail(SDRXGPU_ERR_NO_MEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    b->device_bytes += bytes;
    return SDRXGPU_OK;
}

// Host -> device upload of set-up data (tables, slot lists, taps).  On the BANK'S OWN stream, and complete on return (the source
// is a temporary): a bank enqueues nothing anywhere else, so waiting for its stream is waiting for all of its work.  Rounds 2-3
// used blocking hipMemcpy here, i.e. the NULL stream, which the bank's non-blocking stream neither orders with nor waits for:
// the runtime keeps the last command of a queue alive -- and with it a reference to the command's destination buffer -- until that
// queue is used or synchronised again, so hipFree of the tables at destroy dropped a reference and freed nothing (DESIGN.md 7;
// tools/leak_probe.py reproduces it with lab builds: -DSDRX_LAB_UPLOAD=1 leaks, with -DSDRX_LAB_CREATE_SYNC=1 or =2 it does not).
#ifndef SDRX_LAB_UPLOAD
#define SDRX_LAB_UPLOAD 0
#endif
#ifndef SDRX_LAB_CREATE_SYNC
#define SDRX_LAB_CREATE_SYNC 0
#endif
static hipError_t upload(sdrxgpu_bank *b, void *dst, const void *src, size_t bytes)
{
    if (!bytes) return hipSuccess;
#if SDRX_LAB_UPLOAD == 1
    (void)b;
    return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
#else
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    return e;
#endif
}

static thread_local sdrxgpu_bank *g_creating = nullptr;   // bank under construction (freed if an exception ends bank_create)

// (returns how many hipFree calls failed: sdrxgpu_bank_destroy reports them instead of ignoring them)
static int free_bank(sdrxgpu_bank *b)
{
    if (!b) return 0;
    int n_bad = 0;
    auto fr = [&](void *p) {
        if (p && hipFree(p) != hipSuccess) n_bad++;
    };
    if (g_creating == b) g_creating = nullptr;
    if (b->counted) g_banks_on_device[b->device].fetch_sub(1);
    // every stream of the bank first (a pipelined bank's second chain is not ordered behind the bank's stream any more)
    if (b->stream2) (void)hipStreamSynchronize(b->stream2);
    if (b->copy_stream) (void)hipStreamSynchronize(b->copy_stream);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    for (ProfSpan &s : b->prof_pending) {
        (void)hipEventDestroy(s.a);
        (void)hipEventDestroy(s.b);
    }
    for (hipEvent_t e : b->pool) (void)hipEventDestroy(e);
    fr(b->d_xr[0]);
    fr(b->d_xr[1]);
    fr(b->d_tables);
    fr(b->d_y2);
    fr(b->d_y2b);
    if (b->stream2) {
        (void)hipStreamSynchronize(b->stream2);
        (void)hipStreamDestroy(b->stream2);
    }
    if (b->copy_stream) {
        (void)hipStreamSynchronize(b->copy_stream);
        (void)hipStreamDestroy(b->copy_stream);
    }
    for (hipEvent_t e : {b->ev_h2d, b->ev_d2h, b->ev_ring_free[0], b->ev_ring_free[1]})
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : b->ev_part)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : b->ev_iq)
        if (e) (void)hipEventDestroy(e);
    if (b->ev_fork) (void)hipEventDestroy(b->ev_fork);
    if (b->ev_join) (void)hipEventDestroy(b->ev_join);
    if (b->ev_copied) (void)hipEventDestroy(b->ev_copied);
    fr(b->d_hist[0]);
    fr(b->d_hist[1]);
    fr(b->d_y3);
    fr(b->d_hist2);
    fr(b->d_hist3);
    fr(b->d_seam);
    fr(b->d_slots);
    fr(b->d_slots_m);
    fr(b->d_wtab);
    fr(b->d_y3c);
    fr(b->d_hist3c[0]);
    fr(b->d_hist3c[1]);
    fr(b->d_group_conj);
    fr(b->d_gtab);
    fr(b->d_keep);
    fr(b->d_audio_keep);
    fr(b->d_state);
    fr(b->d_sql);
    fr(b->d_snr);
    fr(b->d_sql_level);
    fr(b->d_sq_tabs);
    fr(b->d_tw);
    fr(b->d_fir_taps);
    fr(b->d_fir_hist);
    fr(b->d_fm);
    fr(b->d_fm_prev);
    fr(b->d_pan_w);
    fr(b->d_mix_partial);
    fr(b->d_mix);
    fr(b->d_mix_f32);
    fr(b->d_mix_s16);
    fr(b->d_mix_taps);
    fr(b->d_mix_hist);
    if (b->owns_stream && b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
    return n_bad;
}

static int init_state(sdrxgpu_bank *b)
{
    if (b->stream2) HIPCHK(hipStreamSynchronize(b->stream2));
    if (b->copy_stream) HIPCHK(hipStreamSynchronize(b->copy_stream));
    b->ring_busy[0] = b->ring_busy[1] = false;
    b->h2d_pending = false;
    b->pending.valid = false;
    b->y2_cur = 0;
    HIPCHK(hipMemsetAsync(b->d_xr[0], 0, b->x_capacity * sizeof(float2), b->stream));
    HIPCHK(hipMemsetAsync(b->d_xr[1], 0, b->x_capacity * sizeof(float2), b->stream));
    b->xr = 0;
    if (b->d_y2) HIPCHK(hipMemsetAsync(b->d_y2, 0, (size_t)b->n_ch * b->y2_stride * sizeof(float2), b->stream));
    if (b->mfma3) {
        HIPCHK(hipMemsetAsync(b->d_hist3c[0], 0, (size_t)b->n_ch * (b->H3c + 8) * sizeof(float2), b->stream));
        HIPCHK(hipMemsetAsync(b->d_hist3c[1], 0, (size_t)b->n_ch * (b->H3c + 8) * sizeof(float2), b->stream));
    } else if (b->fused) {
        HIPCHK(hipMemsetAsync(b->d_y3, 0, (size_t)b->n_ch * b->y3_stride * sizeof(float2), b->stream));
        HIPCHK(hipMemsetAsync(b->d_hist2, 0, (size_t)b->n_ch * b->H2 * sizeof(float2), b->stream));
        HIPCHK(hipMemsetAsync(b->d_hist3, 0, (size_t)b->n_ch * b->H3 * sizeof(float2), b->stream));
        HIPCHK(hipMemsetAsync(b->d_seam, 0, (size_t)b->n_ch * b->seam_stride * sizeof(float2), b->stream));
    } else {
        HIPCHK(hipMemsetAsync(b->d_hist[0], 0, (size_t)b->n_ch * b->y2_hist * sizeof(float2), b->stream));
        HIPCHK(hipMemsetAsync(b->d_hist[1], 0, (size_t)b->n_ch * b->y2_hist * sizeof(float2), b->stream));
    }
    b->hist_cur = 0;
    if (b->squelch) {
        HIPCHK(hipMemsetAsync(b->d_sql, 0, (size_t)b->n_ch * sizeof(uchar2), b->stream));   // SQL_CLOSED, sdrx.cpp:210
        HIPCHK(hipMemsetAsync(b->d_snr, 0, (size_t)b->n_ch * sizeof(float), b->stream));
    }
    if (b->fir_T) HIPCHK(hipMemsetAsync(b->d_fir_hist, 0, (size_t)b->n_ch * (b->fir_T - 1) * sizeof(float), b->stream));
    if (b->d_fm_prev) HIPCHK(hipMemsetAsync(b->d_fm_prev, 0, (size_t)b->n_ch * sizeof(float2), b->stream));   // prev_sample(0, 0), sdrx.cpp:175
    if (b->mix_T) HIPCHK(hipMemsetAsync(b->d_mix_hist, 0, (size_t)2 * (b->mix_T - 1) * sizeof(float), b->stream));
    b->mix_frames = 0;
    std::vector<float4> st(b->n_ch, make_float4(b->agc.init_gain, b->agc.init_power, b->lf.init_gain, b->lf.init_power));
    HIPCHK(hipMemcpyAsync(b->d_state, st.data(), st.size() * sizeof(float4), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    b->n_in = 0;
    b->n_out = 0;
    b->x_origin = -(long long)b->HIST;
    return SDRXGPU_OK;
}

static int bank_create_impl(const sdrxgpu_bank_cfg *cfg, sdrxgpu_bank **out, int *decide_only);

// no C++ exception may cross the C ABI (host-side staging vectors of a very large bank can fail to allocate)
static int bank_create_guarded(const sdrxgpu_bank_cfg *cfg, sdrxgpu_bank **out, int *decide_only)
{
    try {
        const int rc = bank_create_impl(cfg, out, decide_only);
        g_creating = nullptr;
        return rc;
    } catch (const std::bad_alloc &) {
        free_bank(g_creating);
        return fail(SDRXGPU_ERR_NO_MEM, "bank_create: host allocation failed");
    } catch (const std::exception &e) {
        free_bank(g_creating);
        return fail(SDRXGPU_ERR_INVALID_ARG, "bank_create: %s", e.what());
    }
}

extern "C" int sdrxgpu_bank_create(const sdrxgpu_bank_cfg *cfg, sdrxgpu_bank **out)
{
    if (!out) return fail(SDRXGPU_ERR_INVALID_ARG, "bank_create: null argument");
    return bank_create_guarded(cfg, out, nullptr);
}

// decide_only != nullptr: run the validation and the choice of the stage-1/2 kernel family for this channel list (0 vector
// kernels, 1 k_stage12m, 2 k_stage123m -- what bank_info.matrix_pipe would say), allocate nothing on the device, create no bank
// (sdrxgpu_multibank_create: the choice is made ONCE for the whole list and pinned in every shard)
static int bank_create_impl(const sdrxgpu_bank_cfg *cfg, sdrxgpu_bank **out, int *decide_only)
{
    if (!cfg || (!out && !decide_only)) return fail(SDRXGPU_ERR_INVALID_ARG, "bank_create: null argument");
    if (out) *out = nullptr;
    if (cfg->struct_size != sizeof(sdrxgpu_bank_cfg))
        return fail(SDRXGPU_ERR_INVALID_ARG, "bank_create: struct_size %u != %zu", cfg->struct_size, sizeof(sdrxgpu_bank_cfg));
    if (cfg->tuning && cfg->tuning->struct_size != sizeof(sdrxgpu_tuning))
        return fail(SDRXGPU_ERR_INVALID_ARG, "bank_create: tuning.struct_size %u != %zu (built against another version of sdrx_gpu.h?)",
                    cfg->tuning->struct_size, sizeof(sdrxgpu_tuning));
    if (cfg->n_channels == 0 || !cfg->channels) return fail(SDRXGPU_ERR_INVALID_ARG, "bank_create: no channels");
    if (cfg->n_stages < 3 || cfg->n_stages > 4 || !cfg->stages)
        return fail(SDRXGPU_ERR_UNSUPPORTED, "bank_create: %u stages (3 or 4 supported)", cfg->n_stages);
    for (uint32_t s = 0; s < cfg->n_stages; s++) {
        const sdrxgpu_stage &st = cfg->stages[s];
        // MSD::S asserts odd, non-empty taps (msd.hpp:169-170)
        if (st.m == 0 || st.n_taps == 0 || (st.n_taps % 2) == 0 || !st.taps)
            return fail(SDRXGPU_ERR_INVALID_ARG, "bank_create: stage %u needs m>0 and an odd number of taps", s);
        if (s >= 2 && (st.n_taps + 1) / 2 > (uint32_t)kMaxHalfTaps)
            return fail(SDRXGPU_ERR_UNSUPPORTED, "bank_create: stage %u has %u taps (max %d)", s, st.n_taps, 2 * kMaxHalfTaps - 1);
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(SDRXGPU_ERR_NO_DEVICE, "bank_create: no HIP device (this library has no CPU path)");
    int dev = cfg->device;
    if (dev < 0) HIPCHK(hipGetDevice(&dev));
    if (dev >= ndev) return fail(SDRXGPU_ERR_NO_DEVICE, "bank_create: device %d of %d", dev, ndev);
    DeviceGuard on_device(dev);             // (the caller's current device is restored on return)
    HIPCHK(on_device.err);
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SDRXGPU_ERR_NO_DEVICE, "bank_create: device %d is %s, kernels are built for gfx950 only", dev, prop.gcnArchName);

    uint32_t N = 0;
    for (uint32_t c = 0; c < cfg->n_channels; c++) {
        const sdrxgpu_channel &ch = cfg->channels[c];
        if (ch.n_translator == 0) continue;
        if (!ch.translator) return fail(SDRXGPU_ERR_INVALID_ARG, "bank_create: channel %u has a null translator", c);
        if (N == 0) N = ch.n_translator;
        if (ch.n_translator != N)
            return fail(SDRXGPU_ERR_UNSUPPORTED, "bank_create: translator lengths differ (%u vs %u)", ch.n_translator, N);
    }

    const K12Entry *k12 = find_k12(cfg->stages[0].m, cfg->stages[0].n_taps, cfg->stages[1].m, cfg->stages[1].n_taps);
    if (!k12)
        return fail(SDRXGPU_ERR_UNSUPPORTED, "bank_create: no stage-1/2 kernel for (m=%u,taps=%u)(m=%u,taps=%u)",
                    cfg->stages[0].m, cfg->stages[0].n_taps, cfg->stages[1].m, cfg->stages[1].n_taps);

    sdrxgpu_bank *b = new (std::nothrow) sdrxgpu_bank();
    if (!b) return fail(SDRXGPU_ERR_NO_MEM, "bank_create: host allocation failed");
    g_creating = b;
    b->device = dev;
    // the bank's stream first: every upload below goes through it (upload())
    if (cfg->stream) {
        b->stream = (hipStream_t)cfg->stream;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            free_bank(b);
            return fail(SDRXGPU_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
        }
        b->owns_stream = true;
    }
    b->n_ch = cfg->n_channels;
    b->n_stages = cfg->n_stages;
    b->flags = cfg->flags;
    b->agc = cfg->agc;
    b->lf = cfg->lf_agc;
    b->pipelined = (cfg->flags & SDRXGPU_FLAG_PIPELINED) != 0;
    b->ftfir = (cfg->flags & SDRXGPU_FLAG_FTFIR) != 0;
    b->N = N;
    b->k12 = k12;
    for (uint32_t s = 0; s < cfg->n_stages; s++) {
        b->m.push_back(cfg->stages[s].m);
        b->ntaps.push_back(cfg->stages[s].n_taps);
        b->taps.emplace_back(cfg->stages[s].taps, cfg->stages[s].taps + cfg->stages[s].n_taps);
        b->m_total *= cfg->stages[s].m;
    }

    // ---- geometry of the stage-1/2 kernel ----
    b->MM = k12->M1 * k12->M2;
    b->SB = b->MM * k12->Y2B;
    const int halo = k12->M1 * (k12->T2 - k12->M2) + (k12->T1 - k12->M1);
    b->WU = (halo + b->SB - 1) / b->SB;
    b->HIST = ((b->WU * b->SB + b->MM + 7) / 8) * 8;
    // Mixer + stage 1 on the f32 matrix pipe (k_stage12m): the 10 MS/s chain with translator tables.  Automatic for
    // banks of 1 024 channels or more whose launch, counted in rounds of workgroup places x bodies per lane for the bank's own
    // channel list (how many channels found their conjugate partner decides the slot count), comes out under 0.97 of the pair
    // kernel's -- decided further down, once the partners are known; tuning.matrix_pipe overrides.  Its input rows start
    // T1 - M1 samples in front of a body.
    sdrxgpu_tuning tune0 = cfg->tuning ? *cfg->tuning : sdrxgpu_tuning{};
    if (cfg->flags & SDRXGPU_FLAG_REFERENCE_ORDER) {
        if (tune0.matrix_pipe >= 2 || (cfg->flags & SDRXGPU_FLAG_FUSED_STAGE3)) {
            free_bank(b);
            return fail(SDRXGPU_ERR_INVALID_ARG, "bank_create: SDRXGPU_FLAG_REFERENCE_ORDER excludes the matrix-pipe kernels (tuning.matrix_pipe = %u, "
                        "SDRXGPU_FLAG_FUSED_STAGE3 %s)", tune0.matrix_pipe, (cfg->flags & SDRXGPU_FLAG_FUSED_STAGE3) ? "set" : "not set");
        }
        tune0.matrix_pipe = 1;
    }
    {
        const bool can = k12->M1 == 5 && k12->T1 == 17 && k12->M2 == 5 && k12->T2 == 19 && k12->Y2B == 4 && N > 0 && (N % 2) == 0;
        // ... and k_stage123m (stages 2-3 in the same kernel, k_tail_c behind it): the whole 10 MS/s chain with N = 1 200, whole
        // segments per lane, the call cut nowhere else (not pipelined: one scratch)
        const bool can3 = can && N == 1200 && cfg->n_stages == 4 && cfg->stages[2].m == 5 && cfg->stages[2].n_taps == 21 &&
                          cfg->stages[3].m == 5 && cfg->stages[3].n_taps == 67 && !b->pipelined && !tune0.tail_generic &&
                          (cfg->block_hint == 0 || cfg->block_hint == 0xFFFFFFFFu);
        if (tune0.matrix_pipe > 3) {
            free_bank(b);
            return fail(SDRXGPU_ERR_INVALID_ARG, "bank_create: tuning.matrix_pipe %u (0..3)", tune0.matrix_pipe);
        }
        if (tune0.matrix_pipe == 3 && !can3) {
            free_bank(b);
            return fail(SDRXGPU_ERR_UNSUPPORTED, "bank_create: tuning.matrix_pipe = 3, but k_stage123m is built for the whole 10 MS/s chain "
                        "((5,17)(5,19)(5,21)(5,67), N = 1200), not pipelined, whole segments per lane");
        }
        // ... and the 6 MS/s chain's first two stages, (15, 53)(5, 21): k_stage12m6 (kernels_mfma6.hpp)
        const bool can6 = k12->M1 == 15 && k12->T1 == 53 && k12->M2 == 5 && k12->T2 == 21 && k12->Y2B == 2 && N > 0 && (N % 2) == 0 && !b->pipelined;
        if (tune0.matrix_pipe == 2 && !can && !can6) {
            free_bank(b);
            return fail(SDRXGPU_ERR_UNSUPPORTED, "bank_create: tuning.matrix_pipe = 2, but the matrix-pipe kernels are built for the 10 MS/s chain "
                        "(stages (5,17)(5,19): k_stage12m) and the 6 MS/s chain ((15,53)(5,21): k_stage12m6), with translator tables of even length");
        }
        // (automatic only where nothing else about the stage-1/2 kernel is asked for: a required fused tail, required pair lanes
        // or a forced geometry mean the vector kernels)
        const bool others = tune0.fused_tail == 2 || tune0.pair_lanes == 2 || tune0.k12_waves || tune0.k12_cpw;
        if (tune0.matrix_pipe >= 2 && others) {
            free_bank(b);
            return fail(SDRXGPU_ERR_INVALID_ARG, "bank_create: tuning.matrix_pipe = 2 / 3 goes with neither fused_tail = 2, pair_lanes = 2 nor k12_waves / k12_cpw");
        }
        // automatic: from 1 024 channels (where the vector path is k_stage12p), if enough of them have their conjugate partner in
        // the bank (decided below, once the partners are known)
        b->mfma = (can && (tune0.matrix_pipe >= 2 || (tune0.matrix_pipe == 0 && !others && b->n_ch >= 1024))) ||
                  (can6 && !can && tune0.matrix_pipe == 2);
        b->mfma_kind = (b->mfma && can6) ? 1 : 0;
        // (never automatic: 11 % faster per block, but 2 to 3 x the reference's own build-to-build rounding spread on channels next
        // to a strong carrier, where k_stage12m is at 1.1 to 1.4 x: DESIGN.md 4.9 -- a caller's choice, SDRXGPU_FLAG_FUSED_STAGE3)
        b->mfma3 = b->mfma && can3 && (tune0.matrix_pipe == 3 || (tune0.matrix_pipe == 0 && (cfg->flags & SDRXGPU_FLAG_FUSED_STAGE3)));
        if (b->mfma) {
            b->mGS = tune0.m_slot_groups ? (int)tune0.m_slot_groups : 2;
            b->mGT = tune0.m_seg_groups ? (int)tune0.m_seg_groups : 2;
            if (!k12m_geometry_compiled(b->mfma_kind, b->mGS, b->mGT)) {
                free_bank(b);
                return fail(SDRXGPU_ERR_INVALID_ARG, "bank_create: %s is not compiled for %d x %d tiles per workgroup",
                            b->mfma_kind ? "k_stage12m6" : "k_stage12m", b->mGS, b->mGT);
            }
            b->HIST = ((b->WU * b->SB + b->MM + (k12->T1 - k12->M1) + 7) / 8) * 8;
        }
    }
    b->Neff = N ? (int)N : b->SB * 12;
    // lane spacing: smallest multiple of the table length that holds a whole number of bodies
    b->spacing = (int)((long long)b->Neff / gcd_ll(b->Neff, b->SB) * b->SB);
    b->bps = b->spacing / b->SB;
    b->tbl_stride = b->Neff + b->SB;
    // k_stage12m addresses translator rows by 32-bit byte offsets from the table base (LDS-DMA source = scalar base + lane offset)
    if (b->mfma && (uint64_t)b->n_ch * (uint64_t)b->tbl_stride * sizeof(float2) >= (1ull << 32)) {
        if (tune0.matrix_pipe >= 2) {
            free_bank(b);
            return fail(SDRXGPU_ERR_UNSUPPORTED, "bank_create: tuning.matrix_pipe = %u, but %u tables of %d entries exceed the 32-bit row offsets of %s",
                        tune0.matrix_pipe, b->n_ch, b->tbl_stride, tune0.matrix_pipe == 3 ? "k_stage123m" : "k_stage12m");
        }
        b->mfma = b->mfma3 = false;                             // automatic choice: the vector kernels
        b->HIST = ((b->WU * b->SB + b->MM + 7) / 8) * 8;
    }

    // Slots of k_stage12m: a channel and the channel whose table is its exact complex conjugate (+k / -k steps from the
    // tuner); whoever finds no partner gets a slot of its own (the kernel computes the pair all the same and drops the
    // second half: a channel's bits do not depend on who else is in the bank).  The whole-bank list is made here, ahead of the
    // launch geometry: how many workgroups a launch has depends on how many channels found a partner, not on how many there are.
    std::vector<std::string> m_keys, m_fkeys;
    std::vector<int2> m_slots_all;
    int m_np_all = 0;
    const int m_set = 16 * std::max(1, b->mGS);
    auto m_pair_range = [&](int c_lo, int c_hi, std::vector<int2> &slots, int &n_pairs) {
        std::unordered_map<std::string, std::vector<int>> open;
        std::vector<int2> conj;
        std::vector<int> single;
        for (int c = c_lo; c < c_hi; c++) {
            auto it = open.find(m_fkeys[c]);
            if (it != open.end() && !it->second.empty()) {
                conj.push_back(make_int2(it->second.back(), c));
                it->second.pop_back();
            } else {
                open[m_keys[c]].push_back(c);
            }
        }
        for (auto &kv : open)
            for (int c : kv.second) single.push_back(c);
        std::sort(single.begin(), single.end());
        std::sort(conj.begin(), conj.end(), [](const int2 &u, const int2 &v) { return u.x < v.x; });
        n_pairs += (int)conj.size();
        slots.insert(slots.end(), conj.begin(), conj.end());
        for (int c : single) slots.push_back(make_int2(c, -1));
        while (slots.size() % m_set) slots.push_back(make_int2(-1, -1));
    };
    if (b->mfma) {
        auto key_of = [&](uint32_t c, bool flip) {
            const sdrxgpu_channel &ch = cfg->channels[c];
            if (ch.n_translator == 0) return std::string("unit");         // no translator = the unit table, its own conjugate
            std::string k((const char *)ch.translator, (size_t)ch.n_translator * sizeof(sdrxgpu_iq));
            if (flip)
                for (size_t i = 0; i < ch.n_translator; i++) k[i * 8 + 7] ^= (char)0x80;   // sign bit of .im
            return k;
        };
        m_keys.resize(b->n_ch);
        m_fkeys.resize(b->n_ch);
        for (uint32_t c = 0; c < b->n_ch; c++) {
            m_keys[c] = key_of(c, false);
            m_fkeys[c] = key_of(c, true);
        }
        m_pair_range(0, (int)b->n_ch, m_slots_all, m_np_all);
        // Automatic choice: the kernel's work goes with its SLOTS (a channel without partner costs what a pair costs), and what a
        // launch costs is rounds of workgroup places x bodies per lane for either kernel -- the same count the geometry below
        // uses to pick the lane length, in units of one body of a workgroup that shares its CU: 15.0 us for k_stage12m (64 channel
        // places x 32 segments), 5.6 us for k_stage12p (32 channels x 16 segments).  Measured on bench.py's channel list (whole
        // cycles of the 1 200 offsets are pairs, the rest mostly single: 0.50 - 0.67 slots per channel), stage-1/2 kernel ms,
        // matrix / vector, and the ratio the counts predict: 2 048 channels 0.164 / 0.177 (0.92; 0.94), 2 304: 0.164 / 0.185 (0.89;
        // 0.89), 2 560: 0.197 / 0.218 (0.91; 0.89), 2 816: 0.198 / 0.224 (0.89; 0.89), 3 072: 0.253 / 0.257 (0.98; 1.05),
        // 3 328: 0.252 / 0.260 (0.97; 1.03), 3 584: 0.252 / 0.284 (0.89; 0.93), 4 096: 0.293 / 0.324 (0.90; 0.89), 10 000:
        // 0.579 / 0.77 (0.75; 0.76); smaller banks (since k_stage12m deals its units evenly to the XCDs): 1 024: 0.097 / 0.100
        // (0.97; 0.96), 1 152: 0.098 / 0.118 (0.83; 0.94), 1 408: 0.108 / 0.145 (0.75; 0.90), 1 536: 0.148 / 0.139 (1.07; 1.07),
        // 1 792: 0.162 / 0.152 (1.07; 1.16), 896: 0.097 / 0.095 (1.02; 1.33) -- where the prediction is under 0.97 the matrix
        // pipe was ahead at every size measured (profiles/r04_lab_matrix_pipe_choice.txt).  Automatic = predicted ratio under 0.97.
        if (tune0.matrix_pipe == 0 && b->mfma && b->mfma_kind == 0) {
            const long long max_chunk = cfg->max_in_len ? cfg->max_in_len : 512LL * b->m_total;
            const long long U = (max_chunk / b->MM + 1 + k12->Y2B - 1) / k12->Y2B;
            const long long n_cu_ = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            auto launch_cost = [&](long long wgs_whole_segments, long long places) {   // min over lane lengths of rounds x (warm-up + bodies)
                double best = -1.0;
                for (int nb = b->bps; nb >= 1; nb--) {
                    if (b->bps % nb) continue;
                    const long long wgs = wgs_whole_segments * (b->bps / nb);
                    const long long full = wgs / places, rem = wgs % places;
                    const double rounds = (double)full + (rem == 0 ? 0.0 : (2 * rem <= places ? 0.61 : 1.0));
                    const double cost = rounds * (b->WU + nb);
                    if (best < 0 || cost < best) best = cost;
                }
                return best;
            };
            const long long g16 = (U + 16LL * b->bps - 1) / (16LL * b->bps);                        // segment groups of 16
            const long long places_m = n_cu_ * std::max<long long>(1, (long long)(160 * 1024 / k12m_lds_bytes(b->mfma_kind, b->mGS, b->mGT)));
            const double cost_m = launch_cost((long long)(m_slots_all.size() / m_set) * ((g16 + b->mGT - 1) / b->mGT), places_m);
            const double cost_v = launch_cost(((long long)b->n_ch + 2 * kPairSlots - 1) / (2 * kPairSlots) * g16, 2 * n_cu_);
            // (15.0 us: the 2 x 2 workgroup's body; other shapes -- a test knob -- in proportion to their waves)
            if (15.0 * (b->mGS * b->mGT / 4.0) * cost_m >= 0.97 * 5.6 * cost_v) {
                b->mfma = b->mfma3 = false;
                b->HIST = ((b->WU * b->SB + b->MM + 7) / 8) * 8;
                m_slots_all.clear();
            }
        }
    }

    if (decide_only) {
        *decide_only = b->mfma3 ? 2 : b->mfma ? 1 : 0;
        free_bank(b);
        return SDRXGPU_OK;
    }

    // tail-stage history: stage-2 outputs the remaining stages still need from earlier calls
    {
        const int tB = (int)b->ntaps[cfg->n_stages - 1];
        int need = tB - 1;
        if (cfg->n_stages == 4) need = (int)b->m[2] * (tB - 1) + (int)b->ntaps[2] - 1;
        b->y2_hist = ((need + 63) / 64) * 64;
    }

    b->max_chunk = cfg->max_in_len ? cfg->max_in_len : 512u * b->m_total;   // 32 ms block of this chain
    const int max_y2 = (int)(b->max_chunk / b->MM) + 1;
    b->y2_stride = ((max_y2 + 63) / 64) * 64;    // finalised below once the launch geometry is known
    b->keep_stride = (((int)(b->max_chunk / b->m_total) + 1 + 3) / 4) * 4;   // rows stay 16-byte aligned

    // LDS of one k_tail tile: the stage-2 span of tile_b outputs (+ FIR history) and its stage-A outputs
    const sdrxgpu_tuning tune = cfg->tuning ? *cfg->tuning : sdrxgpu_tuning{};
    if (tune.tail_tile) {
        if (tune.tail_tile < 8 || tune.tail_tile > 128) {
            free_bank(b);
            return fail(SDRXGPU_ERR_INVALID_ARG, "bank_create: tuning.tail_tile %u (8..128)", tune.tail_tile);
        }
        b->tile_b = (int)tune.tail_tile;
    }
    b->tail_generic = tune.tail_generic != 0;
    if (tune.fused_tail > 2 || tune.pair_lanes > 2) {
        free_bank(b);
        return fail(SDRXGPU_ERR_INVALID_ARG, "bank_create: tuning.fused_tail %u / tuning.pair_lanes %u (0 = auto, 1 = never, 2 = require)", tune.fused_tail, tune.pair_lanes);
    }
    if (tune.host_parts > 8) {
        free_bank(b);
        return fail(SDRXGPU_ERR_INVALID_ARG, "bank_create: tuning.host_parts %u (0..8)", tune.host_parts);
    }
    b->host_parts = (int)tune.host_parts;
    b->y2_keep = tune.y2_keep_mb == 0 ? (size_t)SDRX_Y2_KEEP_MB << 20 : tune.y2_keep_mb == 0xFFFFFFFFu ? ~(size_t)0 : (size_t)(tune.y2_keep_mb - 1) << 20;
    {
        const int tB = (int)b->ntaps[cfg->n_stages - 1], mB = (int)b->m[cfg->n_stages - 1];
        size_t np, nr = 0;
        if (cfg->n_stages == 4) {
            nr = (size_t)mB * (b->tile_b - 1) + tB;
            np = (size_t)b->m[2] * (nr - 1) + b->ntaps[2];
        } else {
            np = (size_t)mB * (b->tile_b - 1) + tB;
        }
        b->tail_lds = (np + 8 + nr) * sizeof(float2);     // + alignment slack of the staging by groups of Y2B
        if (b->tail_lds > 150 * 1024) {
            free_bank(b);
            return fail(SDRXGPU_ERR_UNSUPPORTED, "bank_create: tail stages need %zu B of LDS per tile (max 153600)", b->tail_lds);
        }
    }

    // Geometry of the stage-1/2 launch: W waves per workgroup, CPW channels per wave (the wave's 64
    // lanes are CPW channels x 64/CPW time segments), nb bodies per lane.  Every choice computes the same
    // outputs bit for bit (tests/test_gpu_parity.py); the choice only decides how much of the machine does
    // useful work.  Picked by a model fitted to a sweep of the 10 MS/s chain over 1 ... 1 023 channels x 9 geometries x
    // 6 lane lengths: workgroups of 4 waves were the fastest or within 2 % of it at every size, so W = 4;
    // the busiest CU gets m = ceil(workgroups / CUs) of them, runs them kmax at a time (LDS, registers), and a body
    // takes body_time(CPW, k) while k workgroups share the CU's SIMDs (fitted, in units of one 4-channel wave alone on
    // its SIMD); cost = (WU + nb) x that sum.  The kernel keeps a channel group's workgroups on one XCD (blockIdx % 8):
    // the last (groups % 8) groups go out by units, so every XCD is reached whatever the group count.
    {
        const long long U = ((long long)max_y2 + k12->Y2B - 1) / k12->Y2B;      // body slots per channel per call
        double best_cost = -1.0;
        const int n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        auto body_time = [](int cpw, int k) {
            if (k <= 0) return 0.0;
            if (cpw == 4) return k == 1 ? 1.0 : k == 2 ? 1.45 : 1.87 + 0.45 * (k - 3);
            if (cpw == 2) return k == 1 ? 0.83 : 1.39 + 0.45 * (k - 2);
            return 0.97 + 0.45 * (k - 1);
        };
        const int W = 4;
        for (int CPW : {4, 2, 1}) {
            if (!k12_geometry_compiled(k12->level, W, CPW)) continue;
            const int SPW = kWave / CPW;
            const size_t lds = 2 * (size_t)(SPW + W * CPW) * (b->SB * 8 + 16);
            if (lds > 160 * 1024) continue;
            const int max_waves_cu = (k12->level == 2) ? 8 : 16;                               // register budget
            const int kmax = (int)std::min<size_t>(160 * 1024 / lds, (size_t)(max_waves_cu / W));
            if (kmax < 1) continue;
            for (int nb = b->bps; nb >= 1; nb--) {
                if (b->bps % nb) continue;
                const int P = b->bps / nb;
                const long long G = (U + (long long)SPW * b->bps - 1) / ((long long)SPW * b->bps);
                const long long cgs = ((long long)b->n_ch + W * CPW - 1) / (W * CPW);
                const long long wgs = cgs * G * P;
                const long long cus = n_cu;      // (every XCD is reached whatever the group count: the last groups % 8 go out by units)
                const long long m = (wgs + cus - 1) / cus;
                const double cost = (double)(b->WU + nb) * ((double)(m / kmax) * body_time(CPW, kmax) + body_time(CPW, (int)(m % kmax)));
                if (best_cost < 0 || cost < best_cost * 0.9999) {
                    best_cost = cost;
                    b->W = W;
                    b->CPW = CPW;
                    b->nb = nb;
                }
            }
        }
        if (best_cost < 0) {
            free_bank(b);
            return fail(SDRXGPU_ERR_UNSUPPORTED, "bank_create: no stage-1/2 geometry fits this chain");
        }
        // Banks that can take two channels per lane (k_stage12p: 10 MS/s chain, compiled for W = 4, CPW = 4) do: measured at
        // 1 024 ... 25 000 channels the pair kernel is as fast as or faster than whatever the model above picks for one
        // channel per lane (1 536: 0.192 against 0.219 ms, 3 000: 0.341 / 0.398, 8 192: 0.811 / 0.942, 25 000: 2.43 / 2.88)
        if (k12->M1 == 5 && k12->T1 == 17 && k12->M2 == 5 && k12->T2 == 19 && k12->Y2B == 4 && !b->pipelined && N > 0 &&
            (tune.pair_lanes == 2 || (tune.pair_lanes == 0 && b->n_ch >= 1024)) && k12_geometry_compiled(k12->level, kPairW, kPairCPW)) {
            b->W = kPairW;
            b->CPW = kPairCPW;
        }
        if (tune.k12_waves) b->W = (int)tune.k12_waves;
        if (tune.k12_cpw) b->CPW = (int)tune.k12_cpw;
        if (b->mfma) {          // segments in groups of 16 (the MFMA's N): the scratch layout of W = 4, CPW = 4
            b->W = 4;
            b->CPW = 4;
        }
        if (!k12_geometry_compiled(k12->level, b->W, b->CPW) ||
            2 * (size_t)(kWave / b->CPW + b->W * b->CPW) * (b->SB * 8 + 16) > 160 * 1024) {
            free_bank(b);
            return fail(SDRXGPU_ERR_INVALID_ARG, "bank_create: stage-1/2 geometry W=%d CPW=%d is not compiled for this chain or does not fit the LDS",
                        b->W, b->CPW);
        }
        // Two channels per lane (k_stage12p) wanted?  Then the bodies per lane are chosen for ITS grid: workgroups of
        // 32 channels, two per CU (same count of rounds x bodies model; measured at 1 024 / 2 048 / 4 096 channels: the
        // choices below are the fastest or within 1 % of it).
        b->want_pair = k12->M1 == 5 && k12->T1 == 17 && k12->M2 == 5 && k12->T2 == 19 && k12->Y2B == 4 && b->W == kPairW &&
                       b->CPW == kPairCPW && !b->pipelined && N > 0 &&
                       (tune.pair_lanes == 2 || (tune.pair_lanes == 0 && b->n_ch >= 1024));
        if (b->want_pair) {
            const int SPW = kWave / b->CPW;
            const long long G = (U + (long long)SPW * b->bps - 1) / ((long long)SPW * b->bps);
            const long long cgs = ((long long)b->n_ch + 2 * kPairSlots - 1) / (2 * kPairSlots);
            // (a last round that leaves every CU with one workgroup at most runs 1.64x faster per workgroup: one wave per
            // SIMD delivers 82 % of what two deliver; ties go to the longer lane: whole segments can run fused)
            const long long slots = 2LL * n_cu;
            double best_cost = -1.0;
            for (int nb = b->bps; nb >= 1; nb--) {
                if (b->bps % nb) continue;
                const long long wgs = cgs * G * (b->bps / nb);
                const long long full = wgs / slots, rem = wgs % slots;
                const double rounds = (double)full + (rem == 0 ? 0.0 : (2 * rem <= slots ? 0.61 : 1.0));
                const double cost = rounds * (b->WU + nb);
                if (best_cost < 0 || cost < best_cost * 0.99) {
                    best_cost = cost;
                    b->nb = nb;
                }
            }
        }
        if (b->mfma) {
            // workgroups of mGS x mGT tiles, two (256 threads) or one (512) per CU; same count of rounds x bodies as above
            const long long Gs = (U + 16LL * b->bps - 1) / (16LL * b->bps);
            const long long sets = (long long)m_slots_all.size() / m_set;      // (exact: partners found, not channels / 2)
            const long long slots = (long long)n_cu * std::max<long long>(1, (long long)(160 * 1024 / k12m_lds_bytes(b->mfma_kind, b->mGS, b->mGT)));
            double best_cost = -1.0;
            for (int nb = b->bps; nb >= 1; nb--) {
                if (b->bps % nb) continue;
                const long long wgs = sets * ((Gs + b->mGT - 1) / b->mGT) * (b->bps / nb);
                const long long full = wgs / slots, rem = wgs % slots;
                const double rounds = (double)full + (rem == 0 ? 0.0 : (2 * rem <= slots ? 0.61 : 1.0));
                const double cost = rounds * (b->WU + nb);
                if (best_cost < 0 || cost < best_cost * 0.99) {
                    best_cost = cost;
                    b->nb = nb;
                }
            }
        }
        if (cfg->block_hint == 0xFFFFFFFFu || b->mfma3) {
            b->nb = b->bps;                   // whole segment per lane (P == 1)
        } else if (cfg->block_hint) {
            if (b->bps % cfg->block_hint) {
                free_bank(b);
                return fail(SDRXGPU_ERR_INVALID_ARG, "bank_create: block_hint %u does not divide %d bodies per segment", cfg->block_hint, b->bps);
            }
            b->nb = (int)cfg->block_hint;
        }
        b->P = b->bps / b->nb;
    }

    // Stage 3 inside the stage-1/2 kernel: possible when the chain's tail is the compiled-in one, every
    // workgroup covers one contiguous range of stage-2 outputs (P == 1), the LDS the tiles leave holds one channel's range at least, and
    // the call's range boundaries fit k_tail4's one pass (4 threads each).  Same outputs either way, bit for bit.
    {
        const int SPW = kWave / b->CPW;
        const long long rlen = (long long)SPW * b->bps * k12->Y2B;
        const size_t lds = 2 * (size_t)(SPW + b->W * b->CPW) * (b->SB * 8 + 16);
        const size_t epilogue_row = ((size_t)rlen / 2 + kWave - 1) / kWave * kWave * 16;     // one channel's range, whole DMA instructions
        const long long n_bnd_max = (max_y2 + rlen - 1) / rlen;
        const int max_y3 = k12->MA ? max_y2 / k12->MA + 2 : 0;
        const size_t t4_lds = ((size_t)((k12->TB - 1 + 15) / 16) * 16 + max_y3 + 2) * sizeof(float2);   // k_tail4: [hist3 | y3] of one channel
        const bool can = k12->fn_fused && cfg->n_stages == 4 && (int)b->m[2] == k12->MA && (int)b->ntaps[2] == k12->TA &&
                         (int)b->m[3] == k12->MB && (int)b->ntaps[3] == k12->TB && b->P == 1 && !b->pipelined &&
                         epilogue_row <= lds && n_bnd_max <= 64 && !b->tail_generic && t4_lds <= 64 * 1024 && !b->mfma;
        if (tune.fused_tail == 2 && !can) {
            free_bank(b);
            return fail(SDRXGPU_ERR_UNSUPPORTED, "bank_create: tuning.fused_tail = 2, but this bank cannot run stage 3 fused "
                        "(chain tail, P = %d, pipelined = %d, range of %lld outputs: %zu B per row, %zu B of LDS)",
                        b->P, (int)b->pipelined, rlen, epilogue_row, lds);
        }
        // automatic: from 6 000 channels (measured with whole segments per lane, fused against separate tail: 3 000 channels
        // 0.345 / 0.340 ms, 5 000: 0.528 / 0.521, 6 144: 0.639 / 0.649, 8 192: 0.825 / 0.853, 10 000: 0.997 / 1.03)
        b->fused = can && (tune.fused_tail == 2 || (tune.fused_tail == 0 && b->n_ch >= 6000));
        if (b->fused) {
            b->H2 = ((k12->TA - 1 + 15) / 16) * 16;
            b->H3 = ((k12->TB - 1 + 15) / 16) * 16;
            b->y3_stride = ((max_y3 + 63) / 64) * 64;
            b->seam_stride = (int)(n_bnd_max + 1) * 2 * (k12->TA - 1);
            b->tail4_lds = t4_lds;
        }
    }

    // Tail stages + AGC/demod as ONE kernel (k_tail_agc, kernels_tailagc.hpp; bit-identical to k_tail_s + k_agc): for banks on the
    // separate tail with one of the compiled-in tail chains.  A workgroup is 1, 2 or 4 channels with two producer waves each and
    // the two AGC waves.  Automatic where it is the faster of the two (mid-size banks: the macro above); tuning.tail_agc overrides.
    {
        const int mA = cfg->n_stages == 4 ? (int)b->m[2] : 0, tA = cfg->n_stages == 4 ? (int)b->ntaps[2] : 0;
        const int mB = (int)b->m[cfg->n_stages - 1], tB = (int)b->ntaps[cfg->n_stages - 1];
        bool any_fm_ = false;                      // (a bank of AM channels only keeps the AM-only kernel, as below for k_agc)
        for (uint32_t c = 0; cfg->modulation && c < b->n_ch; c++) any_fm_ = any_fm_ || cfg->modulation[c] == SDRXGPU_MOD_FM;
        if (tune.tail_agc > 2) {
            free_bank(b);
            return fail(SDRXGPU_ERR_INVALID_ARG, "bank_create: tuning.tail_agc %u (0 = auto, 1 = never, 2 = require) / tail_agc_cpg %u (0 = auto, 1..%d)",
                        tune.tail_agc, tune.tail_agc_cpg, kTaMaxCpg);
        }
        const int n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        // 1, 2 or 4 channels per workgroup (two producer waves each + the two AGC waves): as few as still give every CU at most one
        // workgroup; tuning.tail_agc_cpg sets it instead.  Measured (tools/lab/bench_ta_cpg.sh, the kernel's us with 1 / 2 / 4): 256
        // channels 30.6 / 32.0 / 36.5, 384: 33.9 / 32.0 / 35.6, 512: 34.2 / 32.0 / 35.8, 768: 45.2 / 40.4 / 37.6, 1 024: 67.5 / 41.6 /
        // 37.8.  (A workgroup of 4 channels holds 150 KB of LDS and a CU then has room for nothing else -- which cost two banks on one
        // GPU, BASELINE configs[4], 15 % of their overlap while the kernel was chosen per bank; the automatic choice is for a bank
        // that is alone on its device when it is created, below.)
        b->ta_nt = (int)b->n_ch <= n_cu ? 1 : (int)b->n_ch <= 2 * n_cu ? 2 : 4;
        if (tune.tail_agc_cpg) {
            if (tune.tail_agc_cpg != 1 && tune.tail_agc_cpg != 2 && tune.tail_agc_cpg != 4) {
                free_bank(b);
                return fail(SDRXGPU_ERR_INVALID_ARG, "bank_create: tuning.tail_agc_cpg %u (0 = auto, 1, 2 or 4 channels per workgroup)", tune.tail_agc_cpg);
            }
            b->ta_nt = (int)tune.tail_agc_cpg;
        }
        b->ta_cpg = b->ta_nt;
        const bool compiled = find_tail_agc(b->ta_nt, mA, tA, mB, tB, cfg->sql_level != nullptr, any_fm_, &b->ta_kernel, &b->ta_lds);
        // (the index table: one entry per group of Y2B stage-2 outputs of the longest call, in whole 16-byte pieces)
        b->gtab_len = (((int)max_y2 / k12->Y2B + 2) + 3) / 4 * 4;
        const size_t lds_max = b->ta_lds + (size_t)b->gtab_len * sizeof(uint32_t);
        const bool can = compiled && !b->fused && !b->mfma3 && !b->tail_generic && lds_max <= 160 * 1024;
        if (tune.tail_agc == 2 && !can) {
            free_bank(b);
            return fail(SDRXGPU_ERR_UNSUPPORTED, "bank_create: tuning.tail_agc = 2, but k_tail_agc is built for the separate tail of the reference's "
                        "chains (not with stage 3 fused into the stage-1/2 kernel, not with tail_generic; %zu B of LDS)", lds_max);
        }
        // automatic: mid-size banks that are the only bank on their device -- now, and still when their first block comes (launch_tail)
        b->tail_agc_required = tune.tail_agc == 2;
        const bool alone = !(b->device >= 0 && b->device < 64) || g_banks_on_device[b->device].load() == 0;
        b->tail_agc = can && (tune.tail_agc == 2 || (tune.tail_agc == 0 && alone && (int)b->n_ch >= SDRX_TAILAGC_AUTO_MIN_CU * n_cu &&
                                                     (int)b->n_ch <= SDRX_TAILAGC_AUTO_MAX_CU * n_cu));
        if (b->tail_agc) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(b->ta_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
            if (e != hipSuccess) {
                free_bank(b);
                return fail(SDRXGPU_ERR_HIP, "hipFuncSetAttribute(k_tail_agc, %zu B of LDS) failed: %s", lds_max, hipGetErrorString(e));
            }
        }
    }

    // Host parts (sdrxgpu_bank_process copies the rows of part p out while the kernels of part p+1 run): fixed here
    // because the pairing below keeps partners inside one part.  Vector kernels: about 2 500 channels per part, at most 4 parts
    // (measured at 10 000 channels, page-locked buffers, ms per block: 1 part 2.26, 2: 1.85, 3: 1.73, 4: 1.68, 5: 1.69, 8: 1.79).
    // k_stage12m (round 6, tools/lab/sweep_host_entry.py + the copy traces under profiles/r06_host_entry_*): a part's launch costs
    // whole ROUNDS of the workgroups the chip holds at once (512 = 56 sets of 32 slots at the 10 000-channel bank's lane length,
    // about 0.2 ms each, whatever fraction of the round the part fills) -- four equal parts of a bank of 2.9 rounds paid four
    // rounds, 0.83 ms of stage-1/2 kernels against 0.57 for the whole bank.  So the cuts go by rounds: as many EQUAL parts
    // as the bank has rounds for calls that take the decimated IQ (the 61 MB of copies are what takes longest: they start after the
    // first part's round and then never wait), and two parts -- all whole rounds, then the rest -- for calls that take the audio only
    // (a third of the bytes: the kernels are what takes longest; one cut leaves the last round's rows to copy behind them, more cuts
    // only add launches).  10 000 channels, ms per block: IQ + audio 1.56 (4 equal parts) -> 1.54, audio only 1.31 -> 1.11.
    {
        const int gsz = b->W * b->CPW;
        const int unit = gsz / (int)gcd_ll(gsz, kWave) * kWave;          // whole channel groups and whole k_agc blocks
        const bool may_split = !cfg->sql_level && !cfg->n_audio_fir_taps && !cfg->pan_pos && b->n_ch >= 2048;
        int n_parts = 1;
        if (may_split) n_parts = b->host_parts ? b->host_parts : std::min(4, std::max(1, (int)(b->n_ch / 2500)));
        const int per = (((int)b->n_ch + n_parts - 1) / n_parts + unit - 1) / unit * unit;
        std::vector<int> &pc = b->plan[0].ch;
        pc.clear();
        bool by_rounds = false;
        if (n_parts > 1 && tune.host_part_ch[0]) {
            // explicit boundaries (tuning runs): whole units, strictly rising, inside the bank
            pc.push_back(0);
            for (int i = 0; i < 7 && tune.host_part_ch[i] && (int)pc.size() < n_parts; i++) {
                const int c = (int)std::min<uint32_t>(tune.host_part_ch[i], b->n_ch) / unit * unit;
                if (c <= pc.back() || c >= (int)b->n_ch) {
                    free_bank(b);
                    return fail(SDRXGPU_ERR_INVALID_ARG, "bank_create: tuning.host_part_ch[%d] = %u: boundaries are multiples of %d channels, rising, inside the bank",
                                i, tune.host_part_ch[i], unit);
                }
                pc.push_back(c);
            }
        } else if (may_split && b->mfma && !b->host_parts) {
            by_rounds = true;
        } else {
            for (int c = 0; c < (int)b->n_ch; c += per) pc.push_back(c);
        }
        if (by_rounds) {
            // sets per round for a call of max_chunk samples: the workgroup places of the chip over the time-workgroups of one set
            const long long U = ((long long)max_y2 + k12->Y2B - 1) / k12->Y2B;
            const long long G = (U + (long long)(kWave / b->CPW) * b->bps - 1) / ((long long)(kWave / b->CPW) * b->bps);
            const long long wt = (G + b->mGT - 1) / b->mGT * b->P;
            const long long n_cu_ = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            const long long places = n_cu_ * std::max<long long>(1, (long long)(160 * 1024 / k12m_lds_bytes(b->mfma_kind, b->mGS, b->mGT)));
            const int R = (int)std::max<long long>(1, places / std::max<long long>(1, wt));
            // the first channel of every slot, ascending (a pair belongs to the part of its first channel, further down)
            std::vector<int> first;
            for (const int2 &sl : m_slots_all)
                if (sl.x >= 0) first.push_back(sl.x);
            std::sort(first.begin(), first.end());
            const int n_sets = ((int)first.size() + m_set - 1) / m_set;
            // (whole sets per round: 113 sets of 9 time-workgroups are 1 017 workgroups = two rounds of 512 as ONE launch, but no cut
            // of them into launches of whole sets fits two rounds -- three parts it is: 7 000 channels 1.20 / 0.95 ms per block
            // against 1.30 / 0.98 with two)
            const int n_rounds = (n_sets + R - 1) / R;
            // the channel (whole units) up to which the slots fill `sets` sets
            auto cut_at = [&](int sets) {
                const size_t k = std::min(first.size(), (size_t)sets * (size_t)m_set);
                const int c = k >= first.size() ? (int)b->n_ch : first[k];      // channels below first[k] own at most `sets` sets
                return c / unit * unit;
            };
            auto build = [&](std::vector<int> &out, const std::vector<int> &set_ends) {
                out.assign(1, 0);
                for (int e : set_ends) {
                    const int c = cut_at(e);
                    if (c > out.back() && c < (int)b->n_ch) out.push_back(c);
                }
                out.push_back((int)b->n_ch);
            };
            // [0] IQ + audio: as many parts as the bank has rounds (at most 8: whole rounds per part beyond that), EQUAL in sets -- no
            // part is longer than its rounds, so the launches cost what the whole bank's cost, and the first part, which everything
            // waits for, is no larger than the others (a 5 000-channel bank, 81 sets = 1.45 rounds: [41 | 40] 0.962 ms per block
            // against [56 | 25] 0.985; at 10 000 channels [55 | 54 | 54] is [55 | 56 | 52] to within the noise)
            std::vector<int> ends;
            const int rounds_per_part = (n_rounds + 7) / 8;
            const int n_p = (n_rounds + rounds_per_part - 1) / rounds_per_part;
            for (int i = 1; i < n_p; i++) ends.push_back((int)(((long long)n_sets * i + n_p - 1) / n_p));
            build(pc, ends);
            // [1] audio only: every whole round, then the rest
            ends.clear();
            if (n_rounds >= 2) ends.push_back((n_rounds - 1) * R);
            build(b->plan[1].ch, ends);
        } else {
            pc.push_back((int)b->n_ch);
            b->plan[1].ch = pc;
        }
    }

    // Two channels per lane (k_stage12p): for large banks of the 10 MS/s chain in its large-bank geometry.  Partners are
    // looked for inside each host part: first channels whose tables are exact complex conjugates of each other
    // (+k / -k steps from the tuner: they share the mixer), then whatever is left, two by two.
    {
        // (fused: the waves of a workgroup read their channels' ranges back into LDS rows of their own, one row at least)
        const size_t epilogue_row = ((size_t)(kWave / b->CPW) * b->bps * k12->Y2B / 2 + kWave - 1) / kWave * kWave * 16;
        const bool can = k12->M1 == 5 && k12->T1 == 17 && k12->M2 == 5 && k12->T2 == 19 && k12->Y2B == 4 && b->W == kPairW &&
                         b->CPW == kPairCPW && !b->pipelined && N > 0 && !b->mfma &&
                         (!b->fused || (k12->MA == 5 && k12->TA == 21 && epilogue_row <= k12p_lds_bytes()));
        if (tune.pair_lanes == 2 && !can) {
            free_bank(b);
            return fail(SDRXGPU_ERR_UNSUPPORTED, "bank_create: tuning.pair_lanes = 2, but this bank cannot run two channels per lane "
                        "(10 MS/s chain, W = 4, CPW = 4; W = %d, CPW = %d)", b->W, b->CPW);
        }
        b->pair = can && b->want_pair;
    }
    if (b->pair) {
        std::vector<int2> slots;
        std::vector<unsigned char> group_conj;
        b->part_g.assign(1, 0);
        b->n_conj_pairs = 0;
        auto key_of = [&](uint32_t c, bool flip) {            // the table's bytes, imaginary parts negated when `flip`
            const sdrxgpu_channel &ch = cfg->channels[c];
            std::string k((const char *)ch.translator, (size_t)ch.n_translator * sizeof(sdrxgpu_iq));
            if (flip)
                for (size_t i = 0; i < ch.n_translator; i++) k[i * 8 + 7] ^= (char)0x80;   // sign bit of .im
            return k;
        };
        for (size_t p = 0; p + 1 < b->plan[0].ch.size(); p++) {
            std::unordered_map<std::string, std::vector<int>> open;   // table bytes -> channels of this part still without partner
            std::vector<int2> conj, gen;
            std::vector<int> single;
            for (int c = b->plan[0].ch[p]; c < b->plan[0].ch[p + 1]; c++) {
                if (cfg->channels[c].n_translator == 0) {
                    single.push_back(c);
                    continue;
                }
                auto it = open.find(key_of((uint32_t)c, true));
                if (it != open.end() && !it->second.empty()) {
                    conj.push_back(make_int2(it->second.back(), c));
                    it->second.pop_back();
                } else {
                    open[key_of((uint32_t)c, false)].push_back(c);
                }
            }
            for (auto &kv : open)
                for (int c : kv.second) single.push_back(c);
            std::sort(single.begin(), single.end());
            // conjugate pairs that do not fill a whole group go with the general ones (a conjugate pair is also just a pair)
            while (conj.size() % kPairSlots) {
                gen.push_back(conj.back());
                conj.pop_back();
            }
            for (size_t i = 0; i < single.size(); i += 2) gen.push_back(make_int2(single[i], i + 1 < single.size() ? single[i + 1] : -1));
            while (gen.size() % kPairSlots) gen.push_back(make_int2(-1, -1));
            b->n_conj_pairs += (int)conj.size();
            slots.insert(slots.end(), conj.begin(), conj.end());
            group_conj.insert(group_conj.end(), conj.size() / kPairSlots, 1);
            slots.insert(slots.end(), gen.begin(), gen.end());
            group_conj.insert(group_conj.end(), gen.size() / kPairSlots, 0);
            b->part_g.push_back((int)(slots.size() / kPairSlots));
        }
        b->n_slots = (int)slots.size();
        int rc2;
        if ((rc2 = dev_alloc(b, (void **)&b->d_slots, slots.size() * sizeof(int2))) ||
            (rc2 = dev_alloc(b, (void **)&b->d_group_conj, group_conj.size() + 16))) {
            free_bank(b);
            return rc2;
        }
        hipError_t e = upload(b, b->d_slots, slots.data(), slots.size() * sizeof(int2));
        if (e == hipSuccess) e = upload(b, b->d_group_conj, group_conj.data(), group_conj.size());
        if (e == hipSuccess) e = k12p_set_attributes();
        if (e != hipSuccess) {
            free_bank(b);
            return fail(SDRXGPU_ERR_HIP, "pair-lane set-up failed: %s", hipGetErrorString(e));
        }
    }

    if (b->mfma) {
        // Two slot lists: partners looked for in the whole bank (above; launches that cover every channel: the device and pinned
        // entries), and inside each host part (sdrxgpu_bank_process runs the parts one after the other, every kernel on one
        // part's channels).
        const int set = m_set;
        std::vector<int2> &slots_all = m_slots_all;
        std::vector<int2> slots_parts;
        const int np_all = m_np_all;
        // The parts' launches walk ONE list, the whole bank's slots sorted by their first channel and padded at its end only: part
        // p runs the sets up to the one that holds the last slot whose first channel is below the part's upper bound.  A pair so
        // belongs to the part of its FIRST channel -- its second channel, which may lie in a later part, and the few slots of the
        // next part that share the boundary set are computed ahead of that part's own launch (a part's tail needs its channels'
        // stage-2 rows complete, not computed by its own launch) -- and cutting the bank costs no pair, wherever the cuts fall.
        {
            for (const int2 &sl : slots_all)
                if (sl.x >= 0) slots_parts.push_back(sl);
            std::stable_sort(slots_parts.begin(), slots_parts.end(), [](const int2 &u, const int2 &v) { return u.x < v.x; });
            for (sdrxgpu_bank::HostPlan &hp : b->plan) {
                hp.gm.assign(1, 0);
                size_t i = 0;
                for (size_t p = 0; p + 1 < hp.ch.size(); p++) {
                    while (i < slots_parts.size() && slots_parts[i].x < hp.ch[p + 1]) i++;
                    hp.gm.push_back((int)((i + set - 1) / set));
                }
            }
            while (slots_parts.size() % set) slots_parts.push_back(make_int2(-1, -1));
        }
        b->n_pairs_m = np_all;
        b->n_slots_m = (int)slots_all.size();
        b->n_slots_mp = (int)slots_parts.size();
        int rc2;
        if ((rc2 = dev_alloc(b, (void **)&b->d_slots_m, (slots_all.size() + slots_parts.size()) * sizeof(int2)))) {
            free_bank(b);
            return rc2;
        }
        b->d_slots_mp = b->d_slots_m + slots_all.size();
        hipError_t e = upload(b, b->d_slots_m, slots_all.data(), slots_all.size() * sizeof(int2));
        if (e == hipSuccess) e = upload(b, b->d_slots_mp, slots_parts.data(), slots_parts.size() * sizeof(int2));
        if (e == hipSuccess) e = k12m_set_attributes(b->mfma_kind, b->mGS, b->mGT);
        if (e == hipSuccess && b->mfma3) e = k123m_set_attributes(b->mGS, b->mGT);
        if (e != hipSuccess) {
            free_bank(b);
            return fail(SDRXGPU_ERR_HIP, "k_stage12m set-up failed: %s", hipGetErrorString(e));
        }
        if (b->mfma3) {
            // stages 2 and 3 as one filter on the stage-1 outputs: H[5 e3 + e2] += h3[e3] h2[e2], in double, rounded to fp32 once;
            // table row tau = the six slots' coefficients for a lane in phase tau (fused_index.hpp), rows 150 .. 158 repeat rows 0 .. 8
            double H[fx::NTAP] = {};
            for (size_t e3 = 0; e3 < b->taps[2].size(); e3++)
                for (size_t e2 = 0; e2 < b->taps[1].size(); e2++) H[5 * e3 + e2] += (double)b->taps[2][e3] * (double)b->taps[1][e2];
            std::vector<float> w((size_t)fx::WROWS * 8, 0.0f);
            for (int r = 0; r < fx::WROWS - 1; r++)
                for (int q = 0; q < fx::NSLOT; q++) {
                    const int ee = ((r % fx::PERIOD) - 25 * q + 94 + fx::PERIOD) % fx::PERIOD;
                    w[(size_t)r * 8 + q] = ee < fx::NTAP ? (float)H[ee] : 0.0f;
                }
            if ((rc2 = dev_alloc(b, (void **)&b->d_wtab, w.size() * sizeof(float)))) {
                free_bank(b);
                return rc2;
            }
            e = upload(b, b->d_wtab, w.data(), w.size() * sizeof(float));
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tail_c<5, 67>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            if (e != hipSuccess) {
                free_bank(b);
                return fail(SDRXGPU_ERR_HIP, "k_stage123m set-up failed: %s", hipGetErrorString(e));
            }
        }
    }

    // ---- device memory ----
    // input ring: [HIST history | up to max_chunk new | slack so that the lanes past the end of
    // the data (their results are dropped) still read inside the allocation]
    {
        const int SPW = kWave / b->CPW;
        const long long G = ((long long)max_y2 / k12->Y2B + (long long)SPW * b->bps) / ((long long)SPW * b->bps) + 1;
        // body-major stage-2 rows: whole time-waves of SPW segments x bps bodies x Y2B outputs
        b->y2_stride = (int)((G * SPW * b->bps * k12->Y2B + 63) / 64 * 64);
        if (b->mfma3) {
            // segment groups of 16 segments of 240 stage-1 outputs, rounded up to whole workgroups of mGT groups; 16 entries x 16 segments each
            const long long n_seg = ((long long)b->max_chunk / k12->M1 + 1 + fx::LSEG - 1) / fx::LSEG;
            const long long groups = ((n_seg + 15) / 16 + b->mGT - 1) / b->mGT * b->mGT;
            b->y3c_stride = (int)(groups * fx::NENT * 16);
            if ((uint64_t)(b->n_ch + 1) * (uint64_t)b->y3c_stride * sizeof(float2) >= (1ull << 32)) {
                free_bank(b);
                return fail(SDRXGPU_ERR_UNSUPPORTED, "bank_create: stage-3 scratch of %u x %d entries exceeds k_stage123m's 32-bit byte offsets", b->n_ch, b->y3c_stride);
            }
        }
        b->x_capacity = (size_t)b->HIST + 2 + b->max_chunk + (size_t)G * SPW * b->spacing + (size_t)(b->WU + 2) * b->SB;
    }
    int rc;
    if ((rc = dev_alloc(b, (void **)&b->d_xr[0], b->x_capacity * sizeof(float2))) ||
        (rc = dev_alloc(b, (void **)&b->d_xr[1], b->x_capacity * sizeof(float2))) ||
        (rc = dev_alloc(b, (void **)&b->d_tables, 2 * (size_t)b->n_ch * b->tbl_stride * sizeof(float2))) ||
        (!b->mfma3 && (rc = dev_alloc(b, (void **)&b->d_y2, ((size_t)b->n_ch + (b->mfma ? 1 : 0)) * b->y2_stride * sizeof(float2)))) ||   // (k_stage12m: a spare row for places without channel)
        (b->mfma3 && (rc = dev_alloc(b, (void **)&b->d_y3c, ((size_t)b->n_ch + 1) * b->y3c_stride * sizeof(float2)))) ||
        (b->mfma3 && (rc = dev_alloc(b, (void **)&b->d_hist3c[0], (size_t)b->n_ch * (b->H3c + 8) * sizeof(float2)))) ||
        (b->mfma3 && (rc = dev_alloc(b, (void **)&b->d_hist3c[1], (size_t)b->n_ch * (b->H3c + 8) * sizeof(float2)))) ||
        (b->pipelined && (rc = dev_alloc(b, (void **)&b->d_y2b, ((size_t)b->n_ch + (b->mfma ? 1 : 0)) * b->y2_stride * sizeof(float2)))) ||
        (!b->fused && !b->mfma3 && (rc = dev_alloc(b, (void **)&b->d_hist[0], (size_t)b->n_ch * b->y2_hist * sizeof(float2)))) ||
        (!b->fused && !b->mfma3 && (rc = dev_alloc(b, (void **)&b->d_hist[1], (size_t)b->n_ch * b->y2_hist * sizeof(float2)))) ||
        (b->fused && (rc = dev_alloc(b, (void **)&b->d_y3, (size_t)b->n_ch * b->y3_stride * sizeof(float2)))) ||
        (b->fused && (rc = dev_alloc(b, (void **)&b->d_hist2, (size_t)b->n_ch * b->H2 * sizeof(float2)))) ||
        (b->fused && (rc = dev_alloc(b, (void **)&b->d_hist3, (size_t)b->n_ch * b->H3 * sizeof(float2)))) ||
        (b->fused && (rc = dev_alloc(b, (void **)&b->d_seam, (size_t)b->n_ch * b->seam_stride * sizeof(float2)))) ||
        (rc = dev_alloc(b, (void **)&b->d_keep, (size_t)b->n_ch * b->keep_stride * sizeof(float2))) ||
        (rc = dev_alloc(b, (void **)&b->d_audio_keep, (size_t)b->n_ch * b->keep_stride * sizeof(float))) ||
        (rc = dev_alloc(b, (void **)&b->d_state, (size_t)b->n_ch * sizeof(float4)))) {
        free_bank(b);
        return rc;
    }

    // translator tables: the caller's entries verbatim, wrapped by SB entries so a body never
    // wraps; a channel without translator gets the unit table (x * (1+0j) == x exactly).  Two copies:
    // copy 0 starts at t[0], copy 1 at t[1], so that a body's first entry is always at an even index
    // (16-byte aligned LDS-DMA source) whatever the parity of the stream position.
    {
        std::vector<float2> host(2 * (size_t)b->n_ch * b->tbl_stride);
        for (int copy = 0; copy < 2; copy++) {
            for (uint32_t c = 0; c < b->n_ch; c++) {
                float2 *row = host.data() + ((size_t)copy * b->n_ch + c) * b->tbl_stride;
                const sdrxgpu_channel &ch = cfg->channels[c];
                for (int i = 0; i < b->tbl_stride; i++) {
                    if (ch.n_translator) {
                        const sdrxgpu_iq &t = ch.translator[(i + copy) % b->Neff];
                        // --ftfir: MSD::S builds hk = h * t * 2 (msd.hpp:215); 2*t is exact in fp32, and a channel
                        // without translator takes the plain path un-doubled (msd.hpp:79)
                        const float sc = b->ftfir ? 2.0f : 1.0f;
                        row[i] = make_float2(t.re * sc, t.im * sc);
                    } else {
                        row[i] = make_float2(1.0f, 0.0f);
                    }
                }
            }
        }
        hipError_t e = upload(b, b->d_tables, host.data(), host.size() * sizeof(float2));
        if (e != hipSuccess) {
            free_bank(b);
            return fail(SDRXGPU_ERR_HIP, "table upload failed: %s", hipGetErrorString(e));
        }
    }

    {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_agc<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(AgcLds<false>));
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_agc<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(AgcLds<false>));
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_agc<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(AgcLds<true>));
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_agc<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(AgcLds<true>));
        if (e != hipSuccess) {
            free_bank(b);
            return fail(SDRXGPU_ERR_HIP, "hipFuncSetAttribute(k_agc) failed: %s", hipGetErrorString(e));
        }
    }
    {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tail), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) {
            free_bank(b);
            return fail(SDRXGPU_ERR_HIP, "hipFuncSetAttribute(k_tail) failed: %s", hipGetErrorString(e));
        }
    }
    if (cfg->sql_level) {
        if (!cfg->passband_shape || !cfg->ramp_up || !cfg->ramp_down) {
            free_bank(b);
            return fail(SDRXGPU_ERR_INVALID_ARG, "bank_create: squelch needs passband_shape, ramp_up and ramp_down (512 floats each)");
        }
        b->squelch = true;
        std::vector<float> tabs(4 * 512);
        for (unsigned n = 0; n < 512; n++) {
            tabs[n] = 0.54f - 0.46f * std::cos((2.0f * M_PI * n) / 512);        // Hamming, sdrx.cpp:896-900 verbatim
            tabs[512 + n] = cfg->passband_shape[n];
            tabs[1024 + n] = cfg->ramp_up[n];
            tabs[1536 + n] = cfg->ramp_down[n];
        }
        std::vector<float2> tw(256);
        for (int k = 0; k < 256; k++)
            tw[k] = make_float2((float)std::cos(-2.0 * M_PI * k / 512.0), (float)std::sin(-2.0 * M_PI * k / 512.0));
        if ((rc = dev_alloc(b, (void **)&b->d_sql, (size_t)b->n_ch * sizeof(uchar2))) ||
            (rc = dev_alloc(b, (void **)&b->d_snr, (size_t)b->n_ch * sizeof(float))) ||
            (rc = dev_alloc(b, (void **)&b->d_sql_level, (size_t)b->n_ch * sizeof(float))) ||
            (rc = dev_alloc(b, (void **)&b->d_sq_tabs, tabs.size() * sizeof(float))) ||
            (rc = dev_alloc(b, (void **)&b->d_tw, tw.size() * sizeof(float2)))) {
            free_bank(b);
            return rc;
        }
        hipError_t e = upload(b, b->d_sql_level, cfg->sql_level, (size_t)b->n_ch * sizeof(float));
        if (e == hipSuccess) e = upload(b, b->d_sq_tabs, tabs.data(), tabs.size() * sizeof(float));
        if (e == hipSuccess) e = upload(b, b->d_tw, tw.data(), tw.size() * sizeof(float2));
        if (e != hipSuccess) {
            free_bank(b);
            return fail(SDRXGPU_ERR_HIP, "squelch table upload failed: %s", hipGetErrorString(e));
        }
    }
    if (cfg->n_audio_fir_taps) {
        if (!cfg->audio_fir_taps || cfg->n_audio_fir_taps > 4096) {
            free_bank(b);
            return fail(SDRXGPU_ERR_INVALID_ARG, "bank_create: audio FIR needs 1..4096 taps");
        }
        b->fir_T = (int)cfg->n_audio_fir_taps;
        std::vector<float> c(cfg->audio_fir_taps, cfg->audio_fir_taps + b->fir_T);
        // FIR::setGain, fir.hpp:105-115: c_adj = c * pow(10.0f, gain/20.0f)   (float pow, as the reference)
        for (float &v : c) v = v * std::pow(10.0f, cfg->audio_fir_gain_db / 20.0f);
        if ((rc = dev_alloc(b, (void **)&b->d_fir_taps, c.size() * sizeof(float))) ||
            (rc = dev_alloc(b, (void **)&b->d_fir_hist, (size_t)b->n_ch * (b->fir_T - 1) * sizeof(float) + 16))) {
            free_bank(b);
            return rc;
        }
        hipError_t e = upload(b, b->d_fir_taps, c.data(), c.size() * sizeof(float));
        if (e != hipSuccess) {
            free_bank(b);
            return fail(SDRXGPU_ERR_HIP, "audio FIR tap upload failed: %s", hipGetErrorString(e));
        }
    }
    if (cfg->modulation) {
        bool any_fm = false;
        for (uint32_t c = 0; c < b->n_ch; c++) {
            if (cfg->modulation[c] > SDRXGPU_MOD_FM) {
                free_bank(b);
                return fail(SDRXGPU_ERR_INVALID_ARG, "bank_create: modulation[%u] = %u (0 = AM, 1 = FM)", c, cfg->modulation[c]);
            }
            any_fm = any_fm || cfg->modulation[c] == SDRXGPU_MOD_FM;
        }
        if (any_fm) {       // a bank of AM channels only keeps the AM-only kernel
            if ((rc = dev_alloc(b, (void **)&b->d_fm, b->n_ch)) || (rc = dev_alloc(b, (void **)&b->d_fm_prev, (size_t)b->n_ch * sizeof(float2)))) {
                free_bank(b);
                return rc;
            }
            hipError_t e = upload(b, b->d_fm, cfg->modulation, b->n_ch);
            if (e != hipSuccess) {
                free_bank(b);
                return fail(SDRXGPU_ERR_HIP, "modulation upload failed: %s", hipGetErrorString(e));
            }
        }
    }
    if (cfg->pan_pos) {
        if (cfg->n_mix_fir_taps && (!cfg->mix_fir_taps || cfg->n_mix_fir_taps > 4096)) {
            free_bank(b);
            return fail(SDRXGPU_ERR_INVALID_ARG, "bank_create: mix FIR needs 1..4096 taps");
        }
        b->mix = true;
        b->mix_T = (int)cfg->n_mix_fir_taps;
        b->mix_groups = (int)((b->n_ch + kMixGroup - 1) / kMixGroup);
        std::vector<float2> w(b->n_ch);
        for (uint32_t c = 0; c < b->n_ch; c++) {
            switch (cfg->pan_pos[c]) {                       // sdrx.cpp:471-497
                case -2: w[c] = make_float2(0.8f, 0.2f); break;
                case -1: w[c] = make_float2(0.6f, 0.4f); break;
                case 1:  w[c] = make_float2(0.4f, 0.6f); break;
                case 2:  w[c] = make_float2(0.2f, 0.8f); break;
                default: w[c] = make_float2(0.5f, 0.5f); break;
            }
        }
        std::vector<float> c(cfg->mix_fir_taps, cfg->mix_fir_taps + b->mix_T);
        for (float &v : c) v = v * std::pow(10.0f, cfg->mix_fir_gain_db / 20.0f);   // FIR2::setGain, fir.hpp:160-170
        const size_t ks = (size_t)b->keep_stride;
        if ((rc = dev_alloc(b, (void **)&b->d_pan_w, w.size() * sizeof(float2))) ||
            (rc = dev_alloc(b, (void **)&b->d_mix_partial, (size_t)b->mix_groups * 2 * ks * sizeof(float))) ||
            (rc = dev_alloc(b, (void **)&b->d_mix, 2 * ks * sizeof(float))) ||
            (rc = dev_alloc(b, (void **)&b->d_mix_f32, 2 * ks * sizeof(float))) ||
            (rc = dev_alloc(b, (void **)&b->d_mix_s16, 2 * ks * sizeof(short))) ||
            (rc = dev_alloc(b, (void **)&b->d_mix_taps, c.size() * sizeof(float) + 16)) ||
            (rc = dev_alloc(b, (void **)&b->d_mix_hist, (size_t)2 * (b->mix_T > 0 ? b->mix_T - 1 : 0) * sizeof(float) + 16))) {
            free_bank(b);
            return rc;
        }
        hipError_t e = upload(b, b->d_pan_w, w.data(), w.size() * sizeof(float2));
        if (e == hipSuccess && b->mix_T) e = upload(b, b->d_mix_taps, c.data(), c.size() * sizeof(float));
        if (e != hipSuccess) {
            free_bank(b);
            return fail(SDRXGPU_ERR_HIP, "mix table upload failed: %s", hipGetErrorString(e));
        }
    }
    if (b->pipelined) {
        hipError_t e = hipStreamCreateWithFlags(&b->stream2, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&b->ev_copied, hipEventDisableTiming);
        if (e != hipSuccess) {
            free_bank(b);
            return fail(SDRXGPU_ERR_HIP, "pipelined-mode stream/event creation failed: %s", hipGetErrorString(e));
        }
    }
    if (b->tail_agc) {
        // k_tail_agc's index table: where group gi of a call's stage-2 outputs (time index gi * Y2B) lies in a channel's scratch row --
        // y2_group_index / y2_storage_index of kernels.hpp with exact integer divisions (the bank's geometry, the same for every call)
        std::vector<uint32_t> tab((size_t)b->gtab_len);
        const int y2b = k12->Y2B, seg_len = b->bps * y2b, ph_len = b->nb * y2b, spw = kWave / b->CPW;
        for (int gi = 0; gi < b->gtab_len; gi++) {
            const int j = gi * y2b;
            const int seg = j / seg_len, r = j - seg * seg_len, pp = r / ph_len, r2 = r - pp * ph_len, body = r2 / y2b;
            const int g = seg / spw, sl = seg % spw, wt = g * b->P + pp;
            tab[gi] = (b->mfma && b->mfma_kind == 0) ? (uint32_t)(((wt * 2 * b->nb + 2 * body) * spw + sl) * 2)   // half-body rows (TailArgs.half_split)
                              : (uint32_t)(((wt * b->nb + body) * spw + sl) * y2b);
        }
        if ((rc = dev_alloc(b, (void **)&b->d_gtab, tab.size() * sizeof(uint32_t)))) {
            free_bank(b);
            return rc;
        }
        hipError_t e = upload(b, b->d_gtab, tab.data(), tab.size() * sizeof(uint32_t));
        if (e != hipSuccess) {
            free_bank(b);
            return fail(SDRXGPU_ERR_HIP, "index table upload failed: %s", hipGetErrorString(e));
        }
    }
    rc = init_state(b);
    if (rc) {
        free_bank(b);
        return rc;
    }
    // Everything bank_create has put on the device went through the bank's own stream and is complete (upload(), init_state):
    // no device-wide synchronisation, which would stall every other bank's streams (rounds 2-3 needed one here: see upload()).
#if SDRX_LAB_CREATE_SYNC == 1
    HIPCHK(hipDeviceSynchronize());
#elif SDRX_LAB_CREATE_SYNC == 2
    HIPCHK(hipStreamSynchronize(nullptr));
#endif
    if (b->device >= 0 && b->device < 64) {
        g_banks_on_device[b->device].fetch_add(1);
        b->counted = true;
    }
    *out = b;
    return SDRXGPU_OK;
}

extern "C" int sdrxgpu_bank_destroy(sdrxgpu_bank *b)
{
    if (!b) return fail(SDRXGPU_ERR_INVALID_ARG, "bank_destroy: null bank");
    DeviceGuard on_device(b->device);
    const int n_bad = free_bank(b);
    if (n_bad) return fail(SDRXGPU_ERR_HIP, "bank_destroy: %d hipFree call(s) failed", n_bad);
    return SDRXGPU_OK;
}

extern "C" int sdrxgpu_bank_reset(sdrxgpu_bank *b)
{
    if (!b) return fail(SDRXGPU_ERR_INVALID_ARG, "bank_reset: null bank");
    DeviceGuard on_device(b->device);       // (the caller's current device is the caller's: restored on return)
    HIPCHK(on_device.err);
    return init_state(b);
}

extern "C" int sdrxgpu_bank_synchronize(sdrxgpu_bank *b)
{
    if (!b) return fail(SDRXGPU_ERR_INVALID_ARG, "bank_synchronize: null bank");
    HIPCHK(hipStreamSynchronize(b->stream));
    if (b->stream2) HIPCHK(hipStreamSynchronize(b->stream2));
    if (b->copy_stream) HIPCHK(hipStreamSynchronize(b->copy_stream));
    b->h2d_pending = false;
    return SDRXGPU_OK;
}

extern "C" int sdrxgpu_bank_get_info(sdrxgpu_bank *b, sdrxgpu_bank_info *info)
{
    if (!b || !info) return fail(SDRXGPU_ERR_INVALID_ARG, "bank_get_info: null argument");
    info->n_channels = b->n_ch;
    info->m_total = b->m_total;
    info->n_translator = b->N;
    info->history_len = (uint32_t)b->HIST;
    info->samples_in = b->n_in;
    info->samples_out = b->n_out;
    info->device_bytes = b->device_bytes;
    info->fused_tail = b->fused ? 1u : 0u;
    info->pair_lanes = b->pair ? (uint32_t)b->n_conj_pairs + 1u : 0u;
    info->k12_waves = (uint32_t)b->W;
    info->k12_cpw = (uint32_t)b->CPW;
    info->k12_bodies = (uint32_t)b->nb;
    info->matrix_pipe = b->mfma3 ? 2u : b->mfma ? 1u : 0u;
    info->m_slots = b->mfma ? (uint32_t)b->n_slots_m : 0u;
    info->tail_agc = b->tail_agc ? (uint32_t)b->ta_cpg : 0u;
    return SDRXGPU_OK;
}

extern "C" int sdrxgpu_bank_ingest_buffer(sdrxgpu_bank *b, void **d_ptr, size_t *capacity)
{
    if (!b || !d_ptr) return fail(SDRXGPU_ERR_INVALID_ARG, "bank_ingest_buffer: null argument");
    *d_ptr = b->d_xr[b->xr] + ((long long)b->n_in - b->x_origin);
    if (capacity) *capacity = b->max_chunk;
    return SDRXGPU_OK;
}

// where the history of a chunk ending at absolute sample n1 starts: HIST samples back, one more when the next call's
// first body would otherwise start on an odd index of the ring (16-byte LDS-DMA sources)
static long long origin_after(const sdrxgpu_bank *b, long long n1)
{
    // (k_stage123m's rows start at a stage-1 output's window, any stage-1 output: anchor = M1)
    const long long anchor = b->mfma3 ? b->k12->M1 : b->MM;
    const long long s0_next = (n1 / anchor) * anchor;
    long long origin_next = n1 - b->HIST;
    if ((origin_next - s0_next) & 1) origin_next -= 1;
    return origin_next;
}

extern "C" int sdrxgpu_bank_ingest_buffer_after(sdrxgpu_bank *b, size_t in_len, void **d_ptr)
{
    if (!b || !d_ptr) return fail(SDRXGPU_ERR_INVALID_ARG, "bank_ingest_buffer_after: null argument");
    if (in_len == 0 || in_len > b->max_chunk)
        return fail(SDRXGPU_ERR_INVALID_ARG, "bank_ingest_buffer_after: in_len %zu (1 .. max_in_len = %u)", in_len, b->max_chunk);
    const long long n1 = (long long)b->n_in + (long long)in_len;
    *d_ptr = b->d_xr[b->xr ^ 1] + (n1 - origin_after(b, n1));
    return SDRXGPU_OK;
}

extern "C" int sdrxgpu_bank_get_host_parts(sdrxgpu_bank *b, int audio_only, uint32_t *bounds, size_t cap, size_t *n_parts)
{
    if (!b || !n_parts) return fail(SDRXGPU_ERR_INVALID_ARG, "get_host_parts: null argument");
    const std::vector<int> &pc = b->plan[audio_only ? 1 : 0].ch;
    *n_parts = pc.size() - 1;
    if (bounds) {
        if (cap < pc.size()) return fail(SDRXGPU_ERR_INVALID_ARG, "get_host_parts: %zu entries for %zu bounds", cap, pc.size());
        for (size_t i = 0; i < pc.size(); i++) bounds[i] = (uint32_t)pc[i];
    }
    return SDRXGPU_OK;
}

extern "C" int sdrxgpu_bank_get_gains(sdrxgpu_bank *b, float *agc_gain, float *lf_gain)
{
    if (!b) return fail(SDRXGPU_ERR_INVALID_ARG, "bank_get_gains: null bank");
    DeviceGuard on_device(b->device);       // (the caller's current device is the caller's: restored on return)
    HIPCHK(on_device.err);
    std::vector<float4> st(b->n_ch);
    HIPCHK(hipMemcpyAsync(st.data(), b->d_state, st.size() * sizeof(float4), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    for (uint32_t c = 0; c < b->n_ch; c++) {
        if (agc_gain) agc_gain[c] = st[c].x;
        if (lf_gain) lf_gain[c] = st[c].z;
    }
    return SDRXGPU_OK;
}

extern "C" int sdrxgpu_bank_get_agc_state(sdrxgpu_bank *b, float *agc_gain, float *agc_power, float *lf_gain, float *lf_power)
{
    if (!b) return fail(SDRXGPU_ERR_INVALID_ARG, "bank_get_agc_state: null bank");
    DeviceGuard on_device(b->device);       // (the caller's current device is the caller's: restored on return)
    HIPCHK(on_device.err);
    std::vector<float4> st(b->n_ch);
    HIPCHK(hipMemcpyAsync(st.data(), b->d_state, st.size() * sizeof(float4), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    for (uint32_t c = 0; c < b->n_ch; c++) {
        if (agc_gain) agc_gain[c] = st[c].x;
        if (agc_power) agc_power[c] = st[c].y;
        if (lf_gain) lf_gain[c] = st[c].z;
        if (lf_power) lf_power[c] = st[c].w;
    }
    return SDRXGPU_OK;
}

// ------------------------------------------------------------------------------------------
// profiling (HIP events on the bank's own stream)
// ------------------------------------------------------------------------------------------
static hipEvent_t prof_event(sdrxgpu_bank *b)
{
    if (!b->pool.empty()) {
        hipEvent_t e = b->pool.back();
        b->pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct ProfScope {
    sdrxgpu_bank *b;
    ProfSpan s{};
    bool on;
    hipStream_t st;
    ProfScope(sdrxgpu_bank *bank, int kernel, hipStream_t stream = nullptr)
        : b(bank), on(bank->prof == 1 || (bank->prof == 2 && kernel == SDRXGPU_K_STAGE12)), st(stream ? stream : bank->stream)
    {
        if (!on) return;
        s.kernel = kernel;
        s.a = prof_event(b);
        s.b = prof_event(b);
        (void)hipEventRecord(s.a, st);
    }
    ~ProfScope()
    {
        if (!on) return;
        (void)hipEventRecord(s.b, st);
        b->prof_pending.push_back(s);
    }
};

static void prof_collect(sdrxgpu_bank *b)
{
    if (b->prof_pending.empty()) return;
    (void)hipStreamSynchronize(b->stream);
    if (b->stream2) (void)hipStreamSynchronize(b->stream2);
    for (ProfSpan &s : b->prof_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            b->prof_ms[s.kernel] += ms;
            b->prof_n[s.kernel] += 1;
        }
        b->pool.push_back(s.a);
        b->pool.push_back(s.b);
    }
    b->prof_pending.clear();
}

extern "C" int sdrxgpu_bank_profile_enable(sdrxgpu_bank *b, int enable)
{
    if (!b) return fail(SDRXGPU_ERR_INVALID_ARG, "profile_enable: null bank");
    prof_collect(b);
    b->prof = (enable == 2) ? 2 : (enable != 0);
    return SDRXGPU_OK;
}

extern "C" int sdrxgpu_bank_profile_read(sdrxgpu_bank *b, int kernel, double *total_ms, uint64_t *launches)
{
    if (!b || kernel < 0 || kernel > 3) return fail(SDRXGPU_ERR_INVALID_ARG, "profile_read: bad argument");
    prof_collect(b);
    if (total_ms) *total_ms = b->prof_ms[kernel];
    if (launches) *launches = b->prof_n[kernel];
    return SDRXGPU_OK;
}

extern "C" int sdrxgpu_bank_profile_reset(sdrxgpu_bank *b)
{
    if (!b) return fail(SDRXGPU_ERR_INVALID_ARG, "profile_reset: null bank");
    prof_collect(b);
    for (int i = 0; i < 4; i++) {
        b->prof_ms[i] = 0;
        b->prof_n[i] = 0;
    }
    return SDRXGPU_OK;
}

// ------------------------------------------------------------------------------------------
// one chunk (<= max_chunk samples, already in the current input ring behind its history) through the kernels
// ------------------------------------------------------------------------------------------
// outputs per channel that `len` samples arriving at stream position n0 yield (MSD::decimate's *out_len_ptr, msd.hpp:158)
static size_t outputs_of(const sdrxgpu_bank *b, uint64_t n0, size_t len)
{
    return (size_t)((n0 + len) / b->m_total - n0 / b->m_total);
}

// Stage-1/2 of one chunk (<= max_chunk samples, already in the input ring) on stream `st`; fills `job` with
// everything the tail launch needs except the output pointers and the history ping-pong side.
static int launch_k12_range(sdrxgpu_bank *b, sdrxgpu_bank::TailJob *job, hipStream_t st, int cg_lo, int cg_hi);

// Launch the stage-1/2 kernel of a prepared chunk for the host parts [p_lo, p_hi) (all of them: the whole bank).
static int launch_k123_parts(sdrxgpu_bank *b, sdrxgpu_bank::TailJob *job, hipStream_t st, int p_lo, int p_hi);

static int launch_k12_parts(sdrxgpu_bank *b, sdrxgpu_bank::TailJob *job, hipStream_t st, int p_lo, int p_hi)
{
    if (b->mfma3) return launch_k123_parts(b, job, st, p_lo, p_hi);
    if (b->mfma) {
        ProfScope ps(b, SDRXGPU_K_STAGE12, st);
        const bool whole = p_lo == 0 && p_hi == (int)b->plan[b->cur_plan].ch.size() - 1;
        const int g_lo = whole ? 0 : b->plan[b->cur_plan].gm[p_lo], g_hi = whole ? b->n_slots_m / (16 * b->mGS) : b->plan[b->cur_plan].gm[p_hi];
        if (g_hi <= g_lo) return SDRXGPU_OK;
        K12mLaunch M;
        M.L = job->L;
        M.GS = b->mGS;
        M.GT = b->mGT;
        const int Gs = M.L.args.n_wt / b->P;                            // segment groups of the call
        M.L.args.n_wt = (Gs + b->mGT - 1) / b->mGT * b->P;              // time-waves of mGT segment groups
        M.L.args.cg0 = g_lo;
        M.L.args.n_chgroups = g_hi;
        // (the kernel walks items blockIdx.x, + gridDim.x, ...: a grid capped at the workgroups the chip holds at once makes it
        // persistent -- measured 3 % slower than one workgroup per item, which lets the dispatcher even out the last round)
        // items = 8 x the units of the XCD that has most (kernels_mfma.hpp: whole sets eight at a time, the last (sets % 8) by units)
        M.L.args.n_cgx = ((g_hi - g_lo) / 8) * M.L.args.n_wt + (((g_hi - g_lo) % 8) * M.L.args.n_wt + 7) / 8;   // (k_stage12m reads it as units per XCD)
        M.L.grid = M.L.args.n_cgx * 8;
        M.pa.slots = whole ? b->d_slots_m : b->d_slots_mp;
        M.pa.group_conj = nullptr;
        M.pa.n_slots = whole ? b->n_slots_m : b->n_slots_mp;
        M.kind = b->mfma_kind;
        M.pa.lds_bytes = (int)k12m_lds_bytes(b->mfma_kind, b->mGS, b->mGT);
        if (b->mfma_kind == 0) {
            // The scratch rows of all but the sets written last go out with the streaming hint: what the Infinity Cache cannot hold until
            // the tail kernel comes for it should not push out what it can (the tail starts with the rows written last).
            const size_t keep = b->y2_keep;
            const size_t set_bytes = (size_t)32 * b->mGS * (size_t)M.L.args.n_y2_new * sizeof(float2);
            const long long keep_sets = set_bytes ? (long long)(keep / set_bytes) : 0;
            M.L.args.nt_sets = (int)std::max<long long>(0, (long long)g_hi - keep_sets);
        }
        if ((uint64_t)(b->n_ch + 1) * (uint64_t)b->y2_stride * sizeof(float2) >= (1ull << 32))
            return fail(SDRXGPU_ERR_HIP, "internal: stage-2 scratch of %u x %d entries exceeds k_stage12m's 32-bit byte offsets", b->n_ch, b->y2_stride);
        HIPCHK(launch_k12m(M, st));
        return SDRXGPU_OK;
    }
    if (!b->pair) {
        const int gsz = b->W * b->CPW;
        return launch_k12_range(b, job, st, b->plan[b->cur_plan].ch[p_lo] / gsz, (b->plan[b->cur_plan].ch[p_hi] + gsz - 1) / gsz);
    }
    ProfScope ps(b, SDRXGPU_K_STAGE12, st);
    const int g_lo = b->part_g[p_lo], g_hi = b->part_g[p_hi];
    if (g_hi <= g_lo) return SDRXGPU_OK;
    K12PairLaunch P;
    P.L = job->L;
    P.L.args.cg0 = g_lo;
    P.L.args.n_chgroups = g_hi;
    P.L.args.n_cgx = (g_hi - g_lo + 7) / 8 * 8;
    P.L.grid = 8 * (((g_hi - g_lo) / 8) * P.L.args.n_wt + (((g_hi - g_lo) % 8) * P.L.args.n_wt + 7) / 8);   // (as launch_k12_range)
    P.pa.slots = b->d_slots;
    P.pa.group_conj = b->d_group_conj;
    P.pa.n_slots = b->n_slots;
    P.pa.lds_bytes = (int)k12p_lds_bytes();
    P.lds = k12p_lds_bytes();
    P.fused = b->fused;
    HIPCHK(launch_k12p(P, st));
    return SDRXGPU_OK;
}

// Launch the stage-1/2 kernel of a prepared chunk for the channel groups [cg_lo, cg_hi).
static int launch_k12_range(sdrxgpu_bank *b, sdrxgpu_bank::TailJob *job, hipStream_t st, int cg_lo, int cg_hi)
{
    K12Launch L = job->L;
    L.args.cg0 = cg_lo;
    L.args.n_chgroups = cg_hi;
    L.args.n_cgx = (cg_hi - cg_lo + 7) / 8 * 8;
    // (k_stage12: whole groups eight at a time, the last (groups % 8) by units: 8 x the units of the XCD that has most)
    L.grid = 8 * (((cg_hi - cg_lo) / 8) * L.args.n_wt + (((cg_hi - cg_lo) % 8) * L.args.n_wt + 7) / 8);
    ProfScope ps(b, SDRXGPU_K_STAGE12, st);
    HIPCHK((b->fused ? b->k12->fn_fused : b->k12->fn)(L, st));
    return SDRXGPU_OK;
}

// keep the last HIST (+1 when the next call's first body would otherwise start on an odd index) input samples in front of the
// next block
static int prepare_roll(sdrxgpu_bank *b, sdrxgpu_bank::TailJob *job, long long n1, hipStream_t st)
{
    const long long origin_next = origin_after(b, n1);
    const int src_off = (int)(origin_next - b->x_origin);
    const int cnt = (int)(n1 - origin_next);
    if (src_off < 0) return fail(SDRXGPU_ERR_HIP, "internal: history roll offset %d < 0", src_off);
    if (cnt > 1024) return fail(SDRXGPU_ERR_HIP, "internal: history roll of %d samples", cnt);
    // the history goes to the head of the OTHER ring, which the next chunk is read from (its block may already be
    // arriving there behind the history's place).  Serial mode: the copy rides along with k_agc of the same
    // chunk (launch_tail), when there is one
    job->roll_from = b->d_xr[b->xr] + src_off;
    job->roll_to = b->d_xr[b->xr ^ 1];
    job->roll_cnt = cnt;
    if (b->pipelined || !job->valid || job->t.n_b <= 0) {
        if (job->roll_cnt) {
            hipLaunchKernelGGL(k_roll_history, dim3(1), dim3(1024), 0, st, job->roll_from, job->roll_to, cnt);
            HIPCHK(hipGetLastError());
        }
        job->roll_cnt = 0;
    }
    b->x_origin = origin_next;
    b->xr ^= 1;
    return SDRXGPU_OK;
}

// Launch k_stage123m of a prepared chunk for the host parts [p_lo, p_hi) (all of them: the whole bank).
static int launch_k123_parts(sdrxgpu_bank *b, sdrxgpu_bank::TailJob *job, hipStream_t st, int p_lo, int p_hi)
{
    ProfScope ps(b, SDRXGPU_K_STAGE12, st);
    const bool whole = p_lo == 0 && p_hi == (int)b->plan[b->cur_plan].ch.size() - 1;
    const int g_lo = whole ? 0 : b->plan[b->cur_plan].gm[p_lo], g_hi = whole ? b->n_slots_m / (16 * b->mGS) : b->plan[b->cur_plan].gm[p_hi];
    if (g_hi <= g_lo) return SDRXGPU_OK;
    K123mLaunch M = job->M3;
    M.args.cg0 = g_lo;
    M.args.n_chgroups = g_hi;
    M.args.n_cgx = ((g_hi - g_lo) / 8) * M.args.n_wt + (((g_hi - g_lo) % 8) * M.args.n_wt + 7) / 8;    // (units per XCD, as for k_stage12m)
    M.grid = M.args.n_cgx * 8;
    M.pa.slots = whole ? b->d_slots_m : b->d_slots_mp;
    M.pa.group_conj = nullptr;
    M.pa.n_slots = whole ? b->n_slots_m : b->n_slots_mp;
    M.pa.lds_bytes = (int)k123m_lds_bytes(b->mGS, b->mGT);
    HIPCHK(launch_k123m(M, st));
    return SDRXGPU_OK;
}

// Banks on k_stage123m (kernels_mfma3.hpp): the call's stage-1 outputs are cut into segments of 240 from the call's start; no
// history but the input ring's (the first stage-1 window reaches T1 - M1 samples back).
static int launch_stage123(sdrxgpu_bank *b, size_t in_len, hipStream_t st, sdrxgpu_bank::TailJob *job, bool defer)
{
    const K12Entry *k = b->k12;
    const long long n0 = (long long)b->n_in, n1 = n0 + (long long)in_len;
    const long long k0 = n0 / k->M1, k1 = n1 / k->M1;               // stage-1 outputs before / after this call
    const int n_y1 = (int)(k1 - k0);
    job->valid = false;
    if (n_y1 > 0) {
        K123mLaunch &M = job->M3;
        Stage123Args &a = M.args;
        const long long first = k0 * k->M1 - (k->T1 - k->M1);       // abs index of the first sample of stage-1 output k0's window
        const long long tshift = b->ftfir ? (long long)(k->M1 - 1) : 0;      // (--ftfir: see launch_stage12)
        const long long tph_abs = (((first - tshift) % b->Neff) + b->Neff) % b->Neff;
        const int tcopy = (int)(tph_abs & 1);
        M.x = b->d_xr[b->xr];
        M.tables = b->d_tables + (size_t)tcopy * b->n_ch * b->tbl_stride;
        M.y3c = b->d_y3c;
        M.h1 = b->taps[0].data();
        M.GS = b->mGS;
        M.GT = b->mGT;
        a.n_ch = (int)b->n_ch;
        a.tbl_stride = b->tbl_stride;
        a.N = b->Neff;
        a.n_y1 = n_y1;
        a.x_base = (int)(first - b->x_origin);
        a.tphase0 = (int)(tph_abs - tcopy);
        a.kph = (int)(k0 % fx::PERIOD);
        const int n_seg = (n_y1 + fx::LSEG - 1) / fx::LSEG;
        const int n_grp = (n_seg + 15) / 16;
        a.n_wt = (n_grp + b->mGT - 1) / b->mGT;
        a.y3c_stride = b->y3c_stride;
        a.wtab = b->d_wtab;
        if (first < b->x_origin || (a.x_base & 1) || (a.tphase0 & 1))
            return fail(SDRXGPU_ERR_HIP, "internal: x_base %d / tphase0 %d must be even and >= 0", a.x_base, a.tphase0);
        const size_t x_need = (size_t)a.x_base + (size_t)n_seg * b->Neff + 64;
        if (x_need > b->x_capacity) return fail(SDRXGPU_ERR_HIP, "internal: input ring too small (%zu > %zu)", x_need, b->x_capacity);
        if (a.n_wt * b->mGT * fx::NENT * 16 > b->y3c_stride)
            return fail(SDRXGPU_ERR_HIP, "internal: %d segment groups exceed the stage-3 scratch", a.n_wt * b->mGT);
        job->n_y1 = n_y1;
        job->k25 = (int)(k0 % 25);
        job->r0m6 = (int)((k0 / 25) % fx::NSLOT);
        job->n_seg = n_seg;
        const long long r0 = n0 / (k->M1 * k->M2 * k->MA), r1 = n1 / (k->M1 * k->M2 * k->MA);     // stage-3 outputs before / after
        job->n_out3 = (int)(r1 - r0);
        const long long q0 = r0 / k->MB;
        job->d0 = (int)(k->MB * q0 + k->MB - 1 - r0);
        if (!defer) {
            const int rc = launch_k123_parts(b, job, st, 0, (int)b->plan[b->cur_plan].ch.size() - 1);
            if (rc) return rc;
        }
        TailArgs &t = job->t;
        memset(&t, 0, sizeof(t));
        t.n_ch = (int)b->n_ch;
        t.keep_stride = b->keep_stride;
        t.iq_keep = b->d_keep;
        t.mB = k->MB;
        t.tB = k->TB;
        for (int i = 0; i < (t.tB + 1) / 2; i++) t.hB[i] = b->taps[3][i];
        t.n_b = (int)(n1 / b->m_total - n0 / b->m_total);
        if (t.n_b > b->keep_stride) return fail(SDRXGPU_ERR_HIP, "internal: %d outputs > keep_stride %d", t.n_b, b->keep_stride);
        job->valid = true;
    }
    {
        const int rc = prepare_roll(b, job, n1, st);
        if (rc) return rc;
    }
    b->n_in += in_len;
    return SDRXGPU_OK;
}

// `defer`: prepare only (the caller launches the stage-1/2 kernel itself, by channel ranges: launch_k12_range)
static int launch_stage12(sdrxgpu_bank *b, size_t in_len, hipStream_t st, float2 *y2buf, sdrxgpu_bank::TailJob *job, bool defer = false)
{
    if (b->mfma3) return launch_stage123(b, in_len, st, job, defer);
    const K12Entry *k = b->k12;
    const long long n0 = (long long)b->n_in, n1 = n0 + (long long)in_len;
    const long long j0 = n0 / b->MM, j1 = n1 / b->MM;
    const int n_y2_new = (int)(j1 - j0);
    job->valid = false;

    if (n_y2_new > 0) {
        K12Launch L;
        Stage12Args &a = L.args;
        const long long s0 = j0 * b->MM;                               // abs index of the first input of y2[j0]
        const long long first = s0 - (long long)b->WU * b->SB;         // first warm-up sample of lane 0
        L.x = b->d_xr[b->xr];
        // translator entry met by input sample n: t[n mod N] on the mixer path (trans_pos_, msd.hpp:134-135);
        // with --ftfir stage 0's coefficient sets are laid out so that sample n meets t[(n - (m1-1)) mod N]
        // (msd.hpp:201-203: the set for output 0 starts at index N-(taps-1), its newest tap is t[0] on sample m1-1)
        const long long tshift = b->ftfir ? (long long)(k->M1 - 1) : 0;
        const long long tph_abs = (((first - tshift) % b->Neff) + b->Neff) % b->Neff;
        const int tcopy = (int)(tph_abs & 1);                              // odd phase: table copy shifted by one entry
        L.tables = b->d_tables + (size_t)tcopy * b->n_ch * b->tbl_stride;
        L.y2 = y2buf;
        a.n_ch = (int)b->n_ch;
        a.tbl_stride = b->tbl_stride;
        a.N = b->Neff;
        a.spacing = b->spacing;
        a.y2_stride = b->y2_stride;
        a.y2_hist = 0;                                                 // new outputs start at row index 0
        a.n_y2_new = n_y2_new;
        a.x_base = (int)(first - b->x_origin);                         // the ring's entry 0 holds abs sample x_origin
        a.tphase0 = (int)(tph_abs - tcopy);
        a.nb = b->nb;
        a.P = b->P;
        const long long U = ((long long)n_y2_new + k->Y2B - 1) / k->Y2B;
        const int SPW = kWave / b->CPW;
        const long long G = (U + (long long)SPW * b->bps - 1) / ((long long)SPW * b->bps);
        a.n_wt = (int)(G * b->P);
        a.cg0 = 0;
        a.n_chgroups = (int)((b->n_ch + b->W * b->CPW - 1) / (b->W * b->CPW));
        a.n_cgx = (a.n_chgroups + 7) / 8 * 8;
        if (a.x_base < 0 || (a.x_base & 1) || (a.tphase0 & 1))
            return fail(SDRXGPU_ERR_HIP, "internal: x_base %d / tphase0 %d must be even and >= 0", a.x_base, a.tphase0);
        if (b->mfma && a.x_base < k->T1 - k->M1)
            return fail(SDRXGPU_ERR_HIP, "internal: x_base %d leaves no room for k_stage12m's %d samples of row history", a.x_base, k->T1 - k->M1);
        const size_t x_need = (size_t)a.x_base + (size_t)G * SPW * b->spacing + (size_t)(b->WU + 1) * b->SB;
        if (x_need > b->x_capacity) return fail(SDRXGPU_ERR_HIP, "internal: input ring too small (%zu > %zu)", x_need, b->x_capacity);
        L.h1 = b->taps[0].data();
        L.h2 = b->taps[1].data();
        L.h3 = nullptr;
        L.y3 = nullptr;
        L.seam = nullptr;
        a.seam_stride = 0;
        L.W = b->W;
        L.CPW = b->CPW;
        L.grid = a.n_cgx * a.n_wt;
        a.y3_stride = 0;
        a.c0 = 0;
        a.n_y3 = 0;
        if (b->fused) {
            // stage-3 output R (absolute) has stage-2 outputs MA*R + MA-1 - (TA-1) .. MA*R + MA-1 as its window
            // (S::addSample fires on every MA-th input, msd.hpp:228-247); this call yields R = j0/MA .. j1/MA - 1
            const long long r0 = j0 / k->MA, r1 = j1 / k->MA;
            job->c0 = (int)(k->MA * r0 + k->MA - 1 - j0);
            job->n_y3 = (int)(r1 - r0);
            const long long q0 = r0 / k->MB;
            job->d0 = (int)(k->MB * q0 + k->MB - 1 - r0);
            const long long rlen = (long long)SPW * b->bps * k->Y2B;
            job->n_bnd = (int)((n_y2_new + rlen - 1) / rlen);
            L.h3 = b->taps[2].data();
            L.y3 = b->d_y3;
            L.seam = b->d_seam;
            a.seam_stride = b->seam_stride;
            a.y3_stride = b->y3_stride;
            a.c0 = job->c0;
            a.n_y3 = job->n_y3;
            if (job->n_y3 > b->y3_stride - 2 || job->n_bnd > 64)
                return fail(SDRXGPU_ERR_HIP, "internal: %d stage-3 outputs / %d range boundaries exceed the fused path's buffers", job->n_y3, job->n_bnd);
        }
        job->L = L;
        if (!defer) {
            const int rc = launch_k12_parts(b, job, st, 0, (int)b->plan[b->cur_plan].ch.size() - 1);
            if (rc) return rc;
        }

        // ---- parameters of the remaining stage(s) ----
        TailArgs &t = job->t;
        memset(&t, 0, sizeof(t));
        const int ns = (int)b->n_stages;
        t.y2_new = y2buf;
        t.n_ch = (int)b->n_ch;
        t.y2_stride = b->y2_stride;
        t.hist = b->y2_hist;
        t.n_y2_new = n_y2_new;
        t.keep_stride = b->keep_stride;
        t.iq_keep = b->d_keep;
        t.has_a = (ns == 4);
        t.mB = (int)b->m[ns - 1];
        t.tB = (int)b->ntaps[ns - 1];
        for (int i = 0; i < (t.tB + 1) / 2; i++) t.hB[i] = b->taps[ns - 1][i];
        long long q_prev, q1;
        const long long y2_first = j0 - b->y2_hist;                    // abs stage-2 index of row entry 0
        if (t.has_a) {
            t.mA = (int)b->m[2];
            t.tA = (int)b->ntaps[2];
            for (int i = 0; i < (t.tA + 1) / 2; i++) t.hA[i] = b->taps[2][i];
            q_prev = (j0 / t.mA) / t.mB;
            q1 = (j1 / t.mA) / t.mB;
            const long long i_lo = (long long)t.mB * (q_prev + 1) - t.tB;       // oldest stage-A output needed
            t.pA0 = (int)((long long)t.mA * i_lo + t.mA - t.tA - y2_first);
            t.pB0 = 0;
            if (q1 > q_prev && t.pA0 < 0) return fail(SDRXGPU_ERR_HIP, "internal: pA0 %d < 0", t.pA0);
        } else {
            q_prev = j0 / t.mB;
            q1 = j1 / t.mB;
            t.pB0 = (int)((long long)t.mB * (q_prev + 1) - t.tB - y2_first);
            if (q1 > q_prev && t.pB0 < 0) return fail(SDRXGPU_ERR_HIP, "internal: pB0 %d < 0", t.pB0);
        }
        t.n_b = (int)(q1 - q_prev);
        t.tile_b = b->tile_b;
        t.y2b = k->Y2B;
        t.y2b_shift = (k->Y2B == 4) ? 2 : 1;
        t.seg_len = b->bps * k->Y2B;
        t.ph_len = b->nb * k->Y2B;
        t.spw = kWave / b->CPW;
        t.spw_shift = (t.spw == 64) ? 6 : (t.spw == 32) ? 5 : 4;
        t.nb = b->nb;
        t.P = b->P;
        t.rcp_seg_len = 1.0f / (float)t.seg_len;
        t.rcp_ph_len = 1.0f / (float)t.ph_len;
        t.rcp_y2b = 1.0f / (float)t.y2b;
        t.half_split = (b->mfma && b->mfma_kind == 0) ? 1 : 0;     // k_stage12m stores by half-bodies of two outputs (kernels_mfma.hpp)
        t.n_tiles = std::max(1, (t.n_b + b->tile_b - 1) / b->tile_b);
        if (t.n_b > b->keep_stride) return fail(SDRXGPU_ERR_HIP, "internal: %d outputs > keep_stride %d", t.n_b, b->keep_stride);
        job->valid = true;
    }

    {
        const int rc = prepare_roll(b, job, n1, st);
        if (rc) return rc;
    }
    b->n_in += in_len;
    return SDRXGPU_OK;
}

// k_tail_agc or k_tail_s + k_agc: settled when the bank's first block is processed (the comment at g_banks_on_device)
static bool tail_agc_now(sdrxgpu_bank *b)
{
    if (!b->tail_agc_final) {
        if (b->tail_agc && !b->tail_agc_required && b->device >= 0 && b->device < 64 && g_banks_on_device[b->device].load() > 1) b->tail_agc = false;
        b->tail_agc_final = true;
    }
    return b->tail_agc;
}

// Tail stage(s) + AGC/demod of a prepared job on the bank's stream, outputs to the given device pointers.
// [ch_lo, ch_hi): the channels of this launch (ch_hi < 0: the whole bank); `last`: the last (or only) part of the chunk --
// what concerns the whole bank (history copy, squelch / audio FIR / mix-down, the output count) happens with it.
static int launch_tail(sdrxgpu_bank *b, sdrxgpu_bank::TailJob *job, float2 *d_iq_out, float *d_audio_out, size_t out_stride,
                       size_t out_off, size_t *produced, int ch_lo = 0, int ch_hi = -1, bool last = true)
{
    *produced = 0;
    if (!job->valid) return SDRXGPU_OK;
    if (ch_hi < 0) ch_hi = (int)b->n_ch;
    const unsigned n_part = (unsigned)(ch_hi - ch_lo);
    TailArgs &t = job->t;
    // (process_impl / bank_flush have validated out_stride and the squelch's 512-output rule before anything was
    // launched; these two are internal consistency checks and leave the job untouched)
    if (out_off + (size_t)t.n_b > out_stride && (d_iq_out || d_audio_out))
        return fail(SDRXGPU_ERR_OUT_TOO_SMALL, "internal: %zu outputs per channel but out_stride is %zu", out_off + t.n_b, out_stride);
    if (b->squelch && t.n_b != 512)
        return fail(SDRXGPU_ERR_UNSUPPORTED, "internal: squelch chunk of %d outputs", t.n_b);
    if (last) job->valid = false;
    t.n_ch = ch_hi;
    t.ch_lo = ch_lo;
    t.iq_out = d_iq_out ? d_iq_out + out_off : nullptr;
    t.out_stride = (int)out_stride;
    // k_agc / k_squelch read the decimated IQ rows 16 bytes at a time: from the caller's rows when those are aligned
    // (then no second copy is written), else from the compact keep buffer
    const bool iq_direct = t.iq_out && out_stride % 2 == 0 && ((uintptr_t)t.iq_out & 15) == 0;
    t.iq_keep = iq_direct ? nullptr : b->d_keep;
    const float2 *iq_rows = iq_direct ? t.iq_out : b->d_keep;
    const int iq_rows_stride = iq_direct ? (int)out_stride : b->keep_stride;
    // what the AGC / demodulator needs to know (k_agc, or the consumer waves of k_tail_agc)
    bool agc_done = false;
    auto make_agc_args = [&]() {
        AgcArgs g;
        g.iq = iq_rows;
        g.iq_stride = iq_rows_stride;
        g.audio = d_audio_out ? d_audio_out + out_off : b->d_audio_keep;
        g.out_stride = d_audio_out ? (int)out_stride : b->keep_stride;
        g.state = b->d_state;
        g.n_ch = ch_hi;
        g.ch_lo = ch_lo;
        g.n_out = t.n_b;
        g.lf_active = (b->flags & SDRXGPU_FLAG_LF_AGC_ACTIVE) ? 1 : 0;
        g.audio_vec = (((uintptr_t)g.audio & 15) == 0 && (g.out_stride & 3) == 0) ? 1 : 0;
        auto fill = [](AgcParams &p, const sdrxgpu_agc_params &s) {
            p.attack = s.attack;
            p.decay = s.decay;
            p.reference = s.reference;
            p.max_gain = s.max_gain;
            p.min_gain = (s.max_gain > 0.0f) ? 1.0f / s.max_gain : 1.0f / 200.0f;   // agc.hpp:34, 42
            p.alpha = s.alpha;
            p.sel = (s.decay <= s.attack) ? -INFINITY : INFINITY;
        };
        fill(g.agc, b->agc);
        fill(g.lf, b->lf);
        g.fm = b->d_fm;
        g.fm_prev = b->d_fm_prev;
        g.roll_from = job->roll_from;
        g.roll_to = job->roll_to;
        g.roll_cnt = last ? job->roll_cnt : 0;
        if (last) job->roll_cnt = 0;
        g.sql = b->squelch ? b->d_sql : nullptr;
        g.ramp_up = b->squelch ? b->d_sq_tabs + 1024 : nullptr;      // 16-byte aligned (offsets are multiples of 512 floats)
        g.ramp_down = b->squelch ? b->d_sq_tabs + 1536 : nullptr;
        return g;
    };
    if (b->mfma3) {
        // stage-3 outputs put together from k_stage123m's entries + last stage, one workgroup per channel (k_tail_c)
        TailCArgs f;
        f.t = t;
        f.y3c = b->d_y3c;
        f.y3c_stride = b->y3c_stride;
        f.hist_in = b->d_hist3c[b->hist_cur];
        f.hist_out = b->d_hist3c[b->hist_cur ^ 1];
        f.H3 = b->H3c;
        f.n_y1 = job->n_y1;
        f.k25 = job->k25;
        f.r0m6 = job->r0m6;
        f.n_out3 = job->n_out3;
        f.d0 = job->d0;
        f.n_seg = job->n_seg;
        f.n_tiles = std::max(1, (t.n_b + kTailCTile - 1) / kTailCTile);
        ProfScope ps(b, SDRXGPU_K_TAIL);
        // LDS: the tile's stage-3 outputs (5 x 128 of its own, TB - 1 before them, the next call's history: <= 816) + 16 entries of <= 96 segments
        const size_t lds = (size_t)(816 + kTailCMaxGrp * fx::NENT * 16) * sizeof(float2);
        hipLaunchKernelGGL((k_tail_c<5, 67>), dim3(((n_part + 7) / 8 * 8) * f.n_tiles), dim3(256), lds, b->stream, f);
        HIPCHK(hipGetLastError());
        if (last) b->hist_cur ^= 1;
    } else if (b->fused) {
        // boundary stage-3 outputs + last stage, one workgroup per channel (k_tail4)
        Tail4Args f;
        f.t = t;
        f.y3 = b->d_y3;
        f.seam = b->d_seam;
        f.seam_stride = b->seam_stride;
        f.hist2 = b->d_hist2;
        f.hist3 = b->d_hist3;
        f.y3_stride = b->y3_stride;
        f.n_y3 = job->n_y3;
        f.c0 = job->c0;
        f.d0 = job->d0;
        f.H2 = b->H2;
        f.H3 = b->H3;
        f.rlen = t.spw * t.seg_len;
        f.n_bnd = job->n_bnd;
        ProfScope ps(b, SDRXGPU_K_TAIL);
        const size_t lds = ((size_t)b->H3 + job->n_y3 + 2) * sizeof(float2);
        const K12Entry *k = b->k12;
        if (k->MA == 5 && k->TA == 21 && k->MB == 5 && k->TB == 67)
            hipLaunchKernelGGL((k_tail4<5, 21, 5, 67>), dim3(n_part), dim3(256), lds, b->stream, f);
        else if (k->MA == 5 && k->TA == 19 && k->MB == 5 && k->TB == 45)
            hipLaunchKernelGGL((k_tail4<5, 19, 5, 45>), dim3(n_part), dim3(256), lds, b->stream, f);
        else
            return fail(SDRXGPU_ERR_HIP, "internal: no k_tail4 for this chain");
        HIPCHK(hipGetLastError());
    } else if (tail_agc_now(b) && t.n_b > 0) {
        // tail stages + AGC/demod in one kernel (k_tail_agc): the tile hand-over through LDS replaces the launch boundary
        t.hist_in = b->d_hist[b->hist_cur];
        t.hist_out = b->d_hist[b->hist_cur ^ 1];
        TailAgcArgs f;
        f.t = t;
        f.g = make_agc_args();
        f.cpg = b->ta_cpg;
        f.n_groups = (t.n_y2_new + t.y2b - 1) >> t.y2b_shift;
        const unsigned blocks = (n_part + (unsigned)f.cpg - 1) / (unsigned)f.cpg + (f.g.roll_cnt > 0 ? 1u : 0u);
        ProfScope ps(b, SDRXGPU_K_TAIL);
        f.gtab = b->d_gtab;
        if (f.n_groups > b->gtab_len) return fail(SDRXGPU_ERR_HIP, "internal: %d groups, index table of %d", f.n_groups, b->gtab_len);
        hipLaunchKernelGGL(b->ta_kernel, dim3(blocks), dim3(tail_agc_waves(b->ta_nt) * kWave), b->ta_lds + (size_t)((f.n_groups + 3) / 4 * 4) * sizeof(uint32_t),
                           b->stream, f);
        HIPCHK(hipGetLastError());
        agc_done = true;
        if (last) b->hist_cur ^= 1;
    } else {
        t.hist_in = b->d_hist[b->hist_cur];
        t.hist_out = b->d_hist[b->hist_cur ^ 1];
        ProfScope ps(b, SDRXGPU_K_TAIL);
        const dim3 grid(((n_part + 7) / 8 * 8) * t.n_tiles);
        const int mA = t.has_a ? t.mA : 0, tA = t.has_a ? t.tA : 0;
        // the tail stages of the reference's chains (sdrx.cpp:1588-1668) are compiled in; anything else runs generic
#define SDRX_TAIL_CASE(MA_, TA_, MB_, TB_)                                                                       \
        if (!b->tail_generic && mA == MA_ && tA == TA_ && t.mB == MB_ && t.tB == TB_)                            \
            hipLaunchKernelGGL((k_tail_s<MA_, TA_, MB_, TB_>), grid, dim3(256), b->tail_lds, b->stream, t);      \
        else
        SDRX_TAIL_CASE(5, 21, 5, 67)     // 10 MS/s
        SDRX_TAIL_CASE(5, 19, 5, 45)     // 2.4 MS/s
        SDRX_TAIL_CASE(0, 0, 5, 45)      // 0.96 / 1.2 / 1.44 / 1.6 / 1.92 MS/s
        SDRX_TAIL_CASE(0, 0, 5, 67)      // 6 MS/s
        SDRX_TAIL_CASE(0, 0, 2, 21)      // 2.56 MS/s
            hipLaunchKernelGGL(k_tail, grid, dim3(256), b->tail_lds, b->stream, t);
#undef SDRX_TAIL_CASE
        HIPCHK(hipGetLastError());
        if (last) b->hist_cur ^= 1;
    }

    if (b->mark_after_tail) HIPCHK(hipEventRecord(b->mark_after_tail, b->stream));   // (the decimated IQ rows of [ch_lo, ch_hi) are final)

    // ---- AGC -> |.| -> LF AGC ----
    if (t.n_b > 0) {
        // (with k_tail_agc the recurrence has run already; what follows only needs to know where the audio rows are)
        AgcArgs g;
        if (agc_done) {
            g.audio = d_audio_out ? d_audio_out + out_off : b->d_audio_keep;
            g.out_stride = d_audio_out ? (int)out_stride : b->keep_stride;
        } else {
            g = make_agc_args();
            const unsigned agc_blocks = (n_part + kWave - 1) / kWave + (g.roll_cnt > 0 ? 1u : 0u);
            ProfScope ps(b, SDRXGPU_K_AGC);
            if (b->squelch)
                if (b->d_fm)
                    hipLaunchKernelGGL((k_agc<true, true>), dim3(agc_blocks), dim3(4 * kWave), sizeof(AgcLds<true>), b->stream, g);
                else
                    hipLaunchKernelGGL((k_agc<true, false>), dim3(agc_blocks), dim3(4 * kWave), sizeof(AgcLds<false>), b->stream, g);
            else if (b->d_fm)
                hipLaunchKernelGGL((k_agc<false, true>), dim3(agc_blocks), dim3(4 * kWave), sizeof(AgcLds<true>), b->stream, g);
            else
                hipLaunchKernelGGL((k_agc<false, false>), dim3(agc_blocks), dim3(4 * kWave), sizeof(AgcLds<false>), b->stream, g);
            HIPCHK(hipGetLastError());
        }
        // (banks with squelch, audio FIR or mix-down are never run in parts: these work on the whole bank)
        if (b->squelch) {
            SquelchArgs q;
            q.iq = iq_rows;
            q.iq_stride = iq_rows_stride;
            q.window = b->d_sq_tabs;
            q.pb = b->d_sq_tabs + 512;
            q.tw = b->d_tw;
            q.sql_level = b->d_sql_level;
            q.snr = b->d_snr;
            q.sql = b->d_sql;
            q.n_ch = (int)b->n_ch;
            hipLaunchKernelGGL(k_squelch, dim3((b->n_ch + 3) / 4), dim3(256), 0, b->stream, q);
            HIPCHK(hipGetLastError());
        }
        if (b->fir_T) {
            AudioFirArgs f;
            f.audio = g.audio;
            f.stride = g.out_stride;
            f.hist = b->d_fir_hist;
            f.taps = b->d_fir_taps;
            f.n_ch = (int)b->n_ch;
            f.n_out = t.n_b;
            f.T = b->fir_T;
            const size_t lds = ((size_t)2 * b->fir_T + t.n_b) * sizeof(float);
            hipLaunchKernelGGL(k_audio_fir, dim3(b->n_ch), dim3(256), lds, b->stream, f);
            HIPCHK(hipGetLastError());
        }
        if (b->mix) {
            MixArgs x;
            x.audio = g.audio;
            x.stride = g.out_stride;
            x.w = b->d_pan_w;
            x.partial = b->d_mix_partial;
            x.mix = b->d_mix;
            x.out_f32 = b->d_mix_f32;
            x.out_s16 = b->d_mix_s16;
            x.n_ch = (int)b->n_ch;
            x.n_out = t.n_b;
            x.pstride = b->keep_stride;
            x.n_groups = b->mix_groups;
            hipLaunchKernelGGL(k_mix_partial, dim3((t.n_b + 255) / 256, b->mix_groups), dim3(64), 0, b->stream, x);
            hipLaunchKernelGGL(k_mix_reduce, dim3((t.n_b + 63) / 64, 2), dim3(256), 0, b->stream, x);
            if (b->mix_T) {
                AudioFirArgs f;                         // FIR2 = the same FIR on the two planar rows
                f.audio = b->d_mix;
                f.stride = b->keep_stride;
                f.hist = b->d_mix_hist;
                f.taps = b->d_mix_taps;
                f.n_ch = 2;
                f.n_out = t.n_b;
                f.T = b->mix_T;
                const size_t lds = ((size_t)2 * b->mix_T + t.n_b) * sizeof(float);
                hipLaunchKernelGGL(k_audio_fir, dim3(2), dim3(256), lds, b->stream, f);
            }
            hipLaunchKernelGGL(k_mix_out, dim3((t.n_b + 255) / 256), dim3(256), 0, b->stream, x);
            HIPCHK(hipGetLastError());
            b->mix_frames = t.n_b;
        }
    }
    if (last) {
        *produced = (size_t)t.n_b;
        b->n_out += (uint64_t)t.n_b;
    }
    return SDRXGPU_OK;
}

// one chunk through all kernels, in order, on the bank's stream
static int run_chunk(sdrxgpu_bank *b, size_t in_len, float2 *d_iq_out, float *d_audio_out, size_t out_stride,
                     size_t out_off, size_t *produced)
{
    sdrxgpu_bank::TailJob job;
    const int rc = launch_stage12(b, in_len, b->stream, b->d_y2, &job);
    if (rc) return rc;
    return launch_tail(b, &job, d_iq_out, d_audio_out, out_stride, out_off, produced);
}

// Pipelined mode: two chains that never wait for each other inside a call.  Chain A (stream2): input copy, mixer + stages 1-2 and
// the history roll of THIS block, one block after the other.  Chain B (the bank's stream): tail stages + AGC of the PREVIOUS
// block, behind an event wait for that block's chain-A work.  Outputs (and *produced) are those of the previous block.
//   * stream2 waits (ev_fork) for what the bank's stream holds when the call begins: the caller's producer of the input, and
//     the tail of block k-2 -- the last reader of the stage-2 scratch block k is about to overwrite.  Both are long done when
//     chain A gets there, so the wait costs nothing.
//   * round 4 made the bank's stream wait for chain A again at the END of every call (so that "after the call, in stream
//     order" covered the input): the trace (profiles/r05_trace_pipelined.txt) shows what that cost -- copy -> fork -> stage 1-2
//     -> join -> next copy became ONE chain through two queues, 10 to 25 us per hand-over, and the mode gained nothing.  Now
//     the join is taken at the start of the NEXT call, where the tail needs it anyway.  What the bank's stream does wait for
//     within the call is the COPY of a caller's own input buffer (round 6): the buffer is the caller's again in
//     stream order after the call, as with every other entry.
static int run_chunk_pipelined(sdrxgpu_bank *b, const float2 *src, float2 *d_ingest, size_t in_len, float2 *d_iq_out, float *d_audio_out,
                               size_t out_stride, size_t *produced)
{
    HIPCHK(hipEventRecord(b->ev_fork, b->stream));
    HIPCHK(hipStreamWaitEvent(b->stream2, b->ev_fork, 0));
    // chain B first: the previous block's tail waits for the previous block's chain A (the record of ev_join made a call ago)
    if (b->pending.valid) HIPCHK(hipStreamWaitEvent(b->stream, b->ev_join, 0));
    int rc = launch_tail(b, &b->pending, d_iq_out, d_audio_out, out_stride, 0, produced);
    if (rc) return rc;
    // chain A
    if (src != d_ingest) {
        // a caller's OWN buffer is the caller's again in the bank's stream order once the call has been enqueued (the contract of every
        // other entry, and of this one before round 5): the bank's stream waits for the copy -- the head of chain A, behind nothing
        // but ev_fork -- not for the kernels that follow it.  (A block written straight into the ingest ring is read by those kernels:
        // that ring is free again when the call after next begins, sdrxgpu_bank_ingest_buffer_after.)
        HIPCHK(hipMemcpyAsync(d_ingest, src, in_len * sizeof(float2), hipMemcpyDeviceToDevice, b->stream2));
        HIPCHK(hipEventRecord(b->ev_copied, b->stream2));
        HIPCHK(hipStreamWaitEvent(b->stream, b->ev_copied, 0));
    }
    sdrxgpu_bank::TailJob job;
    float2 *y2buf = b->y2_cur ? b->d_y2b : b->d_y2;
    rc = launch_stage12(b, in_len, b->stream2, y2buf, &job);
    if (rc) return rc;
    HIPCHK(hipEventRecord(b->ev_join, b->stream2));
    if (job.valid) {
        b->pending = job;
        b->y2_cur ^= 1;
    }
    return SDRXGPU_OK;
}

// Everything that can refuse a call is checked HERE, before any launch or state change: a refused call leaves the
// bank exactly as it was (MSD::decimate cannot fail, msd.hpp:75; the next call must still be right).
static int validate_call(const sdrxgpu_bank *b, size_t in_len, bool has_outs, size_t out_stride)
{
    uint64_t pos = b->n_in;
    size_t left = in_len, total_out = 0;
    while (left) {
        const size_t n = std::min(left, (size_t)b->max_chunk);
        const size_t k = outputs_of(b, pos, n);
        if (b->squelch && (pos + n) / b->MM > pos / b->MM && k != 512)
            return fail(SDRXGPU_ERR_UNSUPPORTED, "process: a bank with squelch takes whole 512-output chunks (this call produces %zu)", k);
        total_out += k;
        pos += n;
        left -= n;
    }
    const size_t returned = b->pipelined ? (b->pending.valid ? (size_t)b->pending.t.n_b : 0) : total_out;
    if (has_outs && returned > out_stride)
        return fail(SDRXGPU_ERR_OUT_TOO_SMALL, "process: %zu outputs per channel but out_stride is %zu", returned, out_stride);
    return SDRXGPU_OK;
}

static int ensure_copy_stream(sdrxgpu_bank *b)
{
    if (b->copy_stream) return SDRXGPU_OK;
    HIPCHK(hipStreamCreateWithFlags(&b->copy_stream, hipStreamNonBlocking));
    for (hipEvent_t *e : {&b->ev_h2d, &b->ev_d2h, &b->ev_ring_free[0], &b->ev_ring_free[1]})
        HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    for (hipEvent_t &e : b->ev_part) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (hipEvent_t &e : b->ev_iq) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return SDRXGPU_OK;
}

// One chunk with HOST outputs.  The kernels write the compact keep buffers and the rows are copied out from there.  A
// large plain bank runs as two channel halves: the rows of the first half cross PCIe on the copy stream while the
// kernels of the second half run (61 MB of D2H per 10 000-channel block take as long as its kernels).
static int run_chunk_host(sdrxgpu_bank *b, size_t in_len, void *iq_out, void *audio_out, size_t out_stride, size_t out_off,
                          size_t *produced)
{
    *produced = 0;
    sdrxgpu_bank::TailJob job;
    int rc = launch_stage12(b, in_len, b->stream, b->d_y2, &job, /*defer=*/true);
    if (rc || !job.valid) return rc;
    const int n_b = job.t.n_b;
    const bool copies = n_b > 0 && (iq_out || audio_out);
    struct PlanScope {                                       // which cut of the bank (bank_create's host-parts comment), for this chunk only
        sdrxgpu_bank *b;
        ~PlanScope() { b->cur_plan = 0; b->mark_after_tail = nullptr; }
    } plan_scope{b};
    b->cur_plan = iq_out ? 0 : 1;
    const int n_all = (int)b->plan[b->cur_plan].ch.size() - 1;
    const int n_parts = copies ? n_all : 1;                  // (nothing to copy out: one launch over the whole bank)
    if (n_parts > 1 && (rc = ensure_copy_stream(b))) return rc;
    for (int p = 0; p < n_parts; p++) {
        const int p_lo = copies ? p : 0, p_hi = copies ? p + 1 : n_all;
        const int ch_lo = b->plan[b->cur_plan].ch[p_lo], ch_hi = b->plan[b->cur_plan].ch[p_hi];
        if ((rc = launch_k12_parts(b, &job, b->stream, p_lo, p_hi))) return rc;
        size_t prod = 0;
        // (the decimated IQ rows are final when the tail kernel is done: their copy need not wait for the AGC recurrence behind it)
        b->mark_after_tail = (n_parts > 1 && iq_out) ? b->ev_iq[p] : nullptr;
        rc = launch_tail(b, &job, nullptr, nullptr, b->keep_stride, 0, &prod, ch_lo, ch_hi, p == n_parts - 1);
        b->mark_after_tail = nullptr;
        if (rc) return rc;
        if (!copies) continue;
        hipStream_t cs = b->stream;
        if (n_parts > 1) {
            HIPCHK(hipEventRecord(b->ev_part[p], b->stream));
            cs = b->copy_stream;
        }
        const size_t rows = (size_t)(ch_hi - ch_lo);
        // (rows that touch on both sides -- whole blocks into a [n_channels][512] array -- go as ONE linear copy)
        auto rows_out = [&](void *dst, const void *src, size_t elem) -> hipError_t {
            char *d = (char *)dst + ((size_t)ch_lo * out_stride + out_off) * elem;
            const char *s_ = (const char *)src + (size_t)ch_lo * b->keep_stride * elem;
            if (out_stride == (size_t)n_b && b->keep_stride == n_b)
                return hipMemcpyAsync(d, s_, rows * n_b * elem, hipMemcpyDeviceToHost, cs);
            return hipMemcpy2DAsync(d, out_stride * elem, s_, (size_t)b->keep_stride * elem, (size_t)n_b * elem, rows, hipMemcpyDeviceToHost, cs);
        };
        if (iq_out) {
            if (n_parts > 1) HIPCHK(hipStreamWaitEvent(cs, b->ev_iq[p], 0));
            HIPCHK(rows_out(iq_out, b->d_keep, sizeof(float2)));
        }
        if (audio_out) {
            if (n_parts > 1) HIPCHK(hipStreamWaitEvent(cs, b->ev_part[p], 0));
            HIPCHK(rows_out(audio_out, b->d_audio_keep, sizeof(float)));
        }
    }
    if (n_parts > 1) {      // the keep buffers are free again (next chunk, next call) once the copies have read them
        HIPCHK(hipEventRecord(b->ev_d2h, b->copy_stream));
        HIPCHK(hipStreamWaitEvent(b->stream, b->ev_d2h, 0));
    }
    *produced = (size_t)n_b;
    return SDRXGPU_OK;
}

static int process_impl(sdrxgpu_bank *b, const void *in, bool in_is_host, size_t in_len, void *iq_out, void *audio_out,
                        bool out_is_host, size_t out_stride, size_t *out_len)
{
    if (!b) return fail(SDRXGPU_ERR_INVALID_ARG, "process: null bank");
    if (in_len && !in) return fail(SDRXGPU_ERR_INVALID_ARG, "process: null input");
    if (b->HIST + 1 > 1024) return fail(SDRXGPU_ERR_UNSUPPORTED, "internal: history %d > 1023", b->HIST);
    if (b->pipelined && (in_is_host || out_is_host || in_len > b->max_chunk))
        return fail(SDRXGPU_ERR_UNSUPPORTED, "process: a pipelined bank takes device pointers and at most max_in_len samples per call");
    if (b->mix && in_len > b->max_chunk)
        return fail(SDRXGPU_ERR_UNSUPPORTED, "process: a bank with mix-down takes at most max_in_len = %u samples per call", b->max_chunk);
    {
        const int rc = validate_call(b, in_len, iq_out || audio_out, out_stride);
        if (rc) return rc;
    }
    DeviceGuard on_device(b->device);       // (the caller's current device is the caller's: restored on return)
    HIPCHK(on_device.err);
    if (b->mix && !b->pipelined) b->mix_frames = 0;
    b->ring_busy[0] = b->ring_busy[1] = false;      // (sdrxgpu_bank_process_pinned's bookkeeping does not cover these calls)
    ProfScope total(b, SDRXGPU_K_TOTAL);

    const float2 *src = (const float2 *)in;
    size_t done = 0, produced_total = 0;
    while (done < in_len) {
        float2 *d_ingest = b->d_xr[b->xr] + ((long long)b->n_in - b->x_origin);
        const size_t n = std::min(in_len - done, (size_t)b->max_chunk);
        if (b->pipelined) {
            // (the copy belongs to chain A: run_chunk_pipelined)
        } else if (in_is_host) {
            HIPCHK(hipMemcpyAsync(d_ingest, src + done, n * sizeof(float2), hipMemcpyHostToDevice, b->stream));
        } else if (src + done != d_ingest) {
            HIPCHK(hipMemcpyAsync(d_ingest, src + done, n * sizeof(float2), hipMemcpyDeviceToDevice, b->stream));
        }
        size_t produced = 0;
        int rc;
        if (out_is_host) {
            rc = run_chunk_host(b, n, iq_out, audio_out, out_stride, produced_total, &produced);
            if (rc) return rc;
        } else if (b->pipelined) {
            rc = run_chunk_pipelined(b, src + done, d_ingest, n, (float2 *)iq_out, (float *)audio_out, out_stride, &produced);
            if (rc) return rc;
        } else {
            rc = run_chunk(b, n, (float2 *)iq_out, (float *)audio_out, out_stride, produced_total, &produced);
            if (rc) return rc;
        }
        produced_total += produced;
        done += n;
    }
    if (out_len) *out_len = produced_total;
    if (in_is_host || out_is_host) HIPCHK(hipStreamSynchronize(b->stream));   // borrowed buffers: done before return
    return SDRXGPU_OK;                                                        // (the copy stream's work is ordered before the bank's: ev_d2h)
}

// Throughput entry for a caller that has the block in PAGE-LOCKED host memory and takes the outputs on the device:
// nothing waits.  The H2D copy goes on the bank's copy stream into the input ring the kernels of the previous block
// are NOT reading, so it runs beside them; this block's kernels follow on the bank's stream.
extern "C" int sdrxgpu_bank_process_pinned(sdrxgpu_bank *b, const sdrxgpu_iq *iq_in, size_t in_len, void *d_iq_out,
                                           void *d_audio_out, size_t out_stride, size_t *out_len)
{
    if (!b) return fail(SDRXGPU_ERR_INVALID_ARG, "process_pinned: null bank");
    if (in_len && !iq_in) return fail(SDRXGPU_ERR_INVALID_ARG, "process_pinned: null input");
    if (b->pipelined) return fail(SDRXGPU_ERR_UNSUPPORTED, "process_pinned: not for pipelined banks");
    if (b->mix && in_len > b->max_chunk)
        return fail(SDRXGPU_ERR_UNSUPPORTED, "process_pinned: a bank with mix-down takes at most max_in_len = %u samples per call", b->max_chunk);
    int rc = validate_call(b, in_len, d_iq_out || d_audio_out, out_stride);
    if (rc) return rc;
    DeviceGuard on_device(b->device);       // (the caller's current device is the caller's: restored on return)
    HIPCHK(on_device.err);
    if ((rc = ensure_copy_stream(b))) return rc;
    if (b->h2d_pending) {       // the previous call's host buffer is the caller's again once this call has begun
        HIPCHK(hipEventSynchronize(b->ev_h2d));
        b->h2d_pending = false;
    }
    if (b->mix) b->mix_frames = 0;
    ProfScope total(b, SDRXGPU_K_TOTAL);
    const float2 *src = (const float2 *)iq_in;
    size_t done = 0, produced_total = 0;
    while (done < in_len) {
        const int r = b->xr;
        float2 *d_ingest = b->d_xr[r] + ((long long)b->n_in - b->x_origin);
        const size_t n = std::min(in_len - done, (size_t)b->max_chunk);
        // ring r was last read two chunks ago; if that was not under this entry (no event yet), wait for everything
        // queued on the bank's stream so far
        if (!b->ring_busy[r]) HIPCHK(hipEventRecord(b->ev_ring_free[r], b->stream));
        HIPCHK(hipStreamWaitEvent(b->copy_stream, b->ev_ring_free[r], 0));
        HIPCHK(hipMemcpyAsync(d_ingest, src + done, n * sizeof(float2), hipMemcpyHostToDevice, b->copy_stream));
        HIPCHK(hipEventRecord(b->ev_h2d, b->copy_stream));
        b->h2d_pending = true;
        HIPCHK(hipStreamWaitEvent(b->stream, b->ev_h2d, 0));
        size_t produced = 0;
        rc = run_chunk(b, n, (float2 *)d_iq_out, (float *)d_audio_out, out_stride, produced_total, &produced);
        if (rc) return rc;
        HIPCHK(hipEventRecord(b->ev_ring_free[r], b->stream));
        b->ring_busy[r] = true;
        produced_total += produced;
        done += n;
    }
    if (out_len) *out_len = produced_total;
    return SDRXGPU_OK;
}

extern "C" int sdrxgpu_bank_process(sdrxgpu_bank *b, const sdrxgpu_iq *iq_in, size_t in_len, sdrxgpu_iq *iq_out,
                                    float *audio_out, size_t out_stride, size_t *out_len)
{
    return process_impl(b, iq_in, true, in_len, iq_out, audio_out, true, out_stride, out_len);
}

extern "C" int sdrxgpu_bank_process_device(sdrxgpu_bank *b, const void *d_iq_in, size_t in_len, void *d_iq_out,
                                           void *d_audio_out, size_t out_stride, size_t *out_len)
{
    return process_impl(b, d_iq_in, false, in_len, d_iq_out, d_audio_out, false, out_stride, out_len);
}

extern "C" int sdrxgpu_bank_flush(sdrxgpu_bank *b, void *d_iq_out, void *d_audio_out, size_t out_stride, size_t *out_len)
{
    if (!b) return fail(SDRXGPU_ERR_INVALID_ARG, "flush: null bank");
    if (b->pending.valid && (d_iq_out || d_audio_out) && (size_t)b->pending.t.n_b > out_stride)
        return fail(SDRXGPU_ERR_OUT_TOO_SMALL, "flush: %d outputs per channel but out_stride is %zu", b->pending.t.n_b, out_stride);
    DeviceGuard on_device(b->device);       // (the caller's current device is the caller's: restored on return)
    HIPCHK(on_device.err);
    size_t produced = 0;
    if (b->pipelined && b->pending.valid) HIPCHK(hipStreamWaitEvent(b->stream, b->ev_join, 0));   // the pending block's chain A
    int rc = launch_tail(b, &b->pending, (float2 *)d_iq_out, (float *)d_audio_out, out_stride, 0, &produced);
    if (out_len) *out_len = produced;
    return rc;
}

extern "C" int sdrxgpu_host_register(void *ptr, size_t bytes)
{
    if (!ptr || !bytes) return fail(SDRXGPU_ERR_INVALID_ARG, "host_register: null/empty range");
    HIPCHK(hipHostRegister(ptr, bytes, hipHostRegisterPortable));      // (portable: every device of a multibank copies from / to it)
    return SDRXGPU_OK;
}

extern "C" int sdrxgpu_host_unregister(void *ptr)
{
    if (!ptr) return fail(SDRXGPU_ERR_INVALID_ARG, "host_unregister: null pointer");
    HIPCHK(hipHostUnregister(ptr));
    return SDRXGPU_OK;
}

extern "C" int sdrxgpu_audio_pos(unsigned channel_no, unsigned num_channels)
{
    // sdrx.cpp:1024-1044 with its five positions
    const unsigned half = num_channels / 2;
    if (channel_no >= num_channels) return 0;
    if (channel_no < half) return (int)(std::floor((float)(channel_no * 5u) / (float)num_channels) - 2.0f);
    if (channel_no == half && (num_channels & 1u)) return 0;
    return (int)(2.0f - std::floor((float)((num_channels - 1 - channel_no) * 5u) / (float)num_channels));
}

extern "C" int sdrxgpu_bank_mix_buffers(sdrxgpu_bank *b, void **d_mix_f32, void **d_mix_s16, size_t *n_frames)
{
    if (!b) return fail(SDRXGPU_ERR_INVALID_ARG, "bank_mix_buffers: null bank");
    if (!b->mix) return fail(SDRXGPU_ERR_INVALID_ARG, "bank_mix_buffers: the bank was created without pan_pos");
    if (d_mix_f32) *d_mix_f32 = b->d_mix_f32;
    if (d_mix_s16) *d_mix_s16 = b->d_mix_s16;
    if (n_frames) *n_frames = (size_t)b->mix_frames;
    return SDRXGPU_OK;
}

extern "C" int sdrxgpu_bank_get_mix(sdrxgpu_bank *b, float *mix_f32, int16_t *mix_s16, size_t cap_frames, size_t *n_frames)
{
    if (!b) return fail(SDRXGPU_ERR_INVALID_ARG, "bank_get_mix: null bank");
    if (!b->mix) return fail(SDRXGPU_ERR_INVALID_ARG, "bank_get_mix: the bank was created without pan_pos");
    const size_t n = (size_t)b->mix_frames;
    if (n_frames) *n_frames = n;
    if ((mix_f32 || mix_s16) && cap_frames < n)
        return fail(SDRXGPU_ERR_OUT_TOO_SMALL, "bank_get_mix: %zu frames but room for %zu", n, cap_frames);
    DeviceGuard on_device(b->device);       // (the caller's current device is the caller's: restored on return)
    HIPCHK(on_device.err);
    if (mix_f32 && n) HIPCHK(hipMemcpyAsync(mix_f32, b->d_mix_f32, n * 2 * sizeof(float), hipMemcpyDeviceToHost, b->stream));
    if (mix_s16 && n) HIPCHK(hipMemcpyAsync(mix_s16, b->d_mix_s16, n * 2 * sizeof(short), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return SDRXGPU_OK;
}

extern "C" int sdrxgpu_bank_get_squelch(sdrxgpu_bank *b, float *snr_db, unsigned char *state)
{
    if (!b) return fail(SDRXGPU_ERR_INVALID_ARG, "bank_get_squelch: null bank");
    if (!b->squelch) return fail(SDRXGPU_ERR_INVALID_ARG, "bank_get_squelch: the bank was created without squelch");
    DeviceGuard on_device(b->device);       // (the caller's current device is the caller's: restored on return)
    HIPCHK(on_device.err);
    std::vector<uchar2> st(b->n_ch);
    if (snr_db) HIPCHK(hipMemcpyAsync(snr_db, b->d_snr, (size_t)b->n_ch * sizeof(float), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipMemcpyAsync(st.data(), b->d_sql, st.size() * sizeof(uchar2), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    if (state)
        for (uint32_t c = 0; c < b->n_ch; c++) state[c] = st[c].x;
    return SDRXGPU_OK;
}

#include "multibank.hpp"
