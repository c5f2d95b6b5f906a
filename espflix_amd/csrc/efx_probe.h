// efx_probe.h -- development probes, compiled in only with -DEFX_PROBE (tools/exp/build_variant.sh <tag> "-DEFX_PROBE"); the
// product build sees empty macros and no symbol.  A probed wave claims a record of eight 64-bit words in a ring of its
// translation unit (atomic counter; EFX_PROBE_READER(name) defines the unit's reader efx_probe_read_<name>) and fills it: [0] kernel tag | wave index << 8 | HW_ID << 32, [1..7] whatever the kernel puts there --
// by convention s_memrealtime stamps (100 MHz, one clock for all CUs) and shader-cycle counts.  tools/dbg/probe_waves.py reads the
// ring through efx_probe_read() and turns it into per-CU residency / lifetime tables.
#pragma once
#include <hip/hip_runtime.h>

#ifdef EFX_PROBE
namespace efx {
namespace {
constexpr unsigned kProbeRecords = 1u << 17;
__device__ unsigned long long g_probe[kProbeRecords * 8];
__device__ unsigned int g_probe_next;
__device__ unsigned int g_probe_window[2] = {0u, 0xFFFFFFFFu};  // kernels that pass a `when` record only inside [lo, hi]
}  // namespace
// `seq` < 0: the record comes from the unit's atomic counter; otherwise record (seq mod ring size) -- no atomic: a kernel of
// ten thousand short waves must not queue for one counter (the first k_recon probe stretched its launch 3.5 x)
__device__ inline unsigned long long* probe_claim(unsigned tag, unsigned wave, long long seq = -1, unsigned when = 0)
{
    unsigned slot = 0;
    if (seq >= 0)
        slot = (when >= g_probe_window[0] && when <= g_probe_window[1]) ? (unsigned)((unsigned long long)seq % (kProbeRecords - 1))
                                                                        : kProbeRecords - 1;  // (the last record: a dump)
    else {
        if ((threadIdx.x & 63) == 0)
            slot = atomicAdd(&g_probe_next, 1u) % kProbeRecords;
        slot = (unsigned)__builtin_amdgcn_readfirstlane((int)slot);
    }
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* r = g_probe + (size_t)slot * 8;
    if ((threadIdx.x & 63) == 0) {
        r[0] = tag | ((unsigned long long)wave << 8) | ((unsigned long long)hw << 32);
        r[7] = xcc;
    }
    return r;
}
}  // namespace efx
// a device function that stamps on behalf of the kernel that claimed the record takes it as a trailing parameter
#define EFX_PROBE_PARAM , unsigned long long* const efx_probe_rec
#define EFX_PROBE_PASS , efx_probe_rec
#define EFX_PROBE_PASS_NONE , (efx::g_probe + (size_t)(efx::kProbeRecords - 1) * 8)  // (the dump record: a caller that claimed none)
#define EFX_PROBE_CLAIM(tag, wave) unsigned long long* const efx_probe_rec = efx::probe_claim(tag, wave)
#define EFX_PROBE_CLAIM_AT(tag, wave, seq, when) unsigned long long* const efx_probe_rec = efx::probe_claim(tag, wave, seq, when)
#define EFX_PROBE_STAMP(i)                        \
    do {                                          \
        if ((threadIdx.x & 63) == 0)              \
            efx_probe_rec[i] = wall_clock64();    \
    } while (0)
// a stamp taken only once `v` (a loaded value) is there: the wait is part of what the stamp times
#define EFX_PROBE_STAMP_AFTER(i, v)                                  \
    do {                                                             \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::"v"(v));      \
        if ((threadIdx.x & 63) == 0)                                 \
            efx_probe_rec[i] = wall_clock64();                       \
    } while (0)
// shader cycles between the two (s_memtime): against the 100 MHz stamps this is the clock the CU actually ran at
#define EFX_PROBE_CYCLES_BEGIN() const long long efx_probe_c0 = clock64()
#define EFX_PROBE_CYCLES_END(i)                                      \
    do {                                                             \
        if ((threadIdx.x & 63) == 0)                                 \
            efx_probe_rec[i] = (unsigned long long)(clock64() - efx_probe_c0); \
    } while (0)
#define EFX_PROBE_MAX(i, v) atomicMax(&efx_probe_rec[i], (unsigned long long)(v))
#define EFX_PROBE_SET(i, v)                       \
    do {                                          \
        if ((threadIdx.x & 63) == 0)              \
            efx_probe_rec[i] = (v);               \
    } while (0)
// int efx_probe_read_<name>(dst, max_records, &next): copies the ring (dst == NULL: clears it)
#define EFX_PROBE_READER(name)                                                                                            \
    extern "C" int efx_probe_window_##name(unsigned lo, unsigned hi)                                                      \
    {                                                                                                                     \
        const unsigned w[2] = {lo, hi};                                                                                   \
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(efx::g_probe_window), w, sizeof(w));                                     \
    }                                                                                                                     \
    extern "C" int efx_probe_read_##name(unsigned long long* dst, size_t max_records, unsigned* next)                    \
    {                                                                                                                     \
        if (!dst) {                                                                                                       \
            void *p = nullptr, *q = nullptr;                                                                              \
            if (hipGetSymbolAddress(&p, HIP_SYMBOL(efx::g_probe)) != hipSuccess ||                                        \
                hipGetSymbolAddress(&q, HIP_SYMBOL(efx::g_probe_next)) != hipSuccess)                                     \
                return -1;                                                                                                \
            (void)hipMemset(q, 0, sizeof(unsigned));                                                                      \
            return (int)hipMemset(p, 0, sizeof(unsigned long long) * efx::kProbeRecords * 8);                             \
        }                                                                                                                 \
        if (next && hipMemcpyFromSymbol(next, HIP_SYMBOL(efx::g_probe_next), sizeof(unsigned)) != hipSuccess)             \
            return -1;                                                                                                    \
        const size_t n = max_records < efx::kProbeRecords ? max_records : efx::kProbeRecords;                             \
        return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(efx::g_probe), n * 8 * sizeof(unsigned long long));               \
    }
#else
#define EFX_PROBE_READER(name)
#define EFX_PROBE_PARAM
#define EFX_PROBE_PASS
#define EFX_PROBE_PASS_NONE
#define EFX_PROBE_CLAIM(tag, wave) ((void)0)
#define EFX_PROBE_CLAIM_AT(tag, wave, seq, when) ((void)0)
#define EFX_PROBE_STAMP(i) ((void)0)
#define EFX_PROBE_STAMP_AFTER(i, v) ((void)0)
#define EFX_PROBE_CYCLES_BEGIN() ((void)0)
#define EFX_PROBE_CYCLES_END(i) ((void)0)
#define EFX_PROBE_MAX(i, v) ((void)0)
#define EFX_PROBE_SET(i, v) ((void)0)
#endif
