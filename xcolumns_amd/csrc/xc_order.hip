// xc_order.hip -- the visiting order of a sweep, on the host (no device code in this file).
//
// predict_using_bc_with_0approx draws ONE generator, np.random.default_rng(seed), and shuffles ONE array
// cumulatively, once per sweep (/root/reference/xcolumns/block_coordinate.py:413-419).  numpy's
// Generator.shuffle is a sequential Fisher-Yates walk: for i = n-1 .. 1: j = random_interval(i);
// swap(x[i], x[j]), with random_interval = masked rejection on 32-bit halves of the PCG64 (XSL-RR 128/64)
// output when i < 2^32 (numpy/random/_generator.pyx `_shuffle_raw`, src/distributions/distributions.c
// `random_interval`, src/pcg64/pcg64.h `pcg64_next32`).  At 1 M rows that walk costs 7.6 ms in numpy --
// twelve sweeps' worth of GPU time -- most of it cache misses of the random swap partner in an 8 MB
// int64 array.  This is the same walk on an int32 array (half the footprint) with the draws generated a
// block ahead, so the partner's line is prefetched before the swap needs it: same generator state in,
// same permutation and state out, about 2-3x less time.  The Python side verifies it against numpy itself
// on a small array at first use and falls back to numpy's own shuffle if a single element differs.
#include <stdint.h>

#include "xc_host.h"
#include "xcolumns_amd.h"

namespace {

typedef unsigned __int128 u128;

struct Pcg64 {
    u128 state, inc;
    int has_uint32;
    uint32_t uinteger;
};

// pcg_setseq_128_xsl_rr_64_random_r: step, then output
static inline uint64_t pcg64_next64(Pcg64 &g) {
    const u128 mult = ((u128)2549297995355413924ULL << 64) | (u128)4865540595714422341ULL;
    g.state = g.state * mult + g.inc;
    const uint64_t hi = (uint64_t)(g.state >> 64), lo = (uint64_t)g.state;
    const unsigned rot = (unsigned)(hi >> 58); // state >> 122
    const uint64_t x = hi ^ lo;
    return (x >> rot) | (x << ((64 - rot) & 63));
}

static inline uint64_t pcg64_output(u128 state) { // XSL-RR of a state that has already been stepped
    const uint64_t hi = (uint64_t)(state >> 64), lo = (uint64_t)state;
    const unsigned rot = (unsigned)(hi >> 58);
    const uint64_t x = hi ^ lo;
    return (x >> rot) | (x << ((64 - rot) & 63));
}

static inline uint32_t pcg64_next32(Pcg64 &g) {
    if (g.has_uint32) {
        g.has_uint32 = 0;
        return g.uinteger;
    }
    const uint64_t next = pcg64_next64(g);
    g.has_uint32 = 1;
    g.uinteger = (uint32_t)(next >> 32);
    return (uint32_t)next;
}

static inline uint64_t random_interval(Pcg64 &g, uint64_t max) {
    if (max == 0) return 0;
    uint64_t mask = max, value;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    mask |= mask >> 32;
    if (max <= 0xffffffffULL) {
        while ((value = (pcg64_next32(g) & mask)) > max) {
        }
    } else {
        while ((value = (pcg64_next64(g) & mask)) > max) {
        }
    }
    return value;
}

} // namespace

extern "C" {

// One Generator.shuffle(order) of numpy's PCG64 stream, in place on an int32 array of n entries.
// state_io: {state_hi, state_lo, inc_hi, inc_lo} of rng.bit_generator.state["state"]; has_uint32_io / uinteger_io:
// the buffered 32-bit half.  All updated to the generator's state after the shuffle.
int xc_host_shuffle_pcg64(uint64_t *state_io, int *has_uint32_io, uint32_t *uinteger_io, int64_t n, int32_t *order) {
    if (!state_io || !has_uint32_io || !uinteger_io || n < 0 || (n > 0 && !order))
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_host_shuffle_pcg64: bad argument");
    Pcg64 g;
    g.state = ((u128)state_io[0] << 64) | (u128)state_io[1];
    g.inc = ((u128)state_io[2] << 64) | (u128)state_io[3];
    g.has_uint32 = *has_uint32_io;
    g.uinteger = *uinteger_io;
    enum { BLOCK = 512, AHEAD = 32 };
    uint32_t js[BLOCK + 1];
    int64_t i = n - 1;
    while (i >= 1) {
        // steps i, i-1, ..., i-cnt+1 (all >= 1), cut where the rejection mask changes (at a power of two): inside
        // such a run random_interval's loop is "take the next 32-bit draw & mask, keep it if <= the current bound",
        // which filters a stream of candidates without a data-dependent branch (half of them are rejected just
        // above a power of two: a mispredicted branch per element otherwise)
        int cnt = (int)(i < BLOCK ? i : BLOCK);
        if (i <= 0xffffffffLL) {
            uint64_t mask = (uint64_t)i;
            mask |= mask >> 1;
            mask |= mask >> 2;
            mask |= mask >> 4;
            mask |= mask >> 8;
            mask |= mask >> 16;
            const int64_t low = (int64_t)(mask >> 1) + 1; // smallest bound with this mask
            if (i - cnt + 1 < low) cnt = (int)(i - low + 1);
            const uint32_t m32 = (uint32_t)mask;
            int got = 0;
            while (got < cnt) {
                const uint32_t v = pcg64_next32(g) & m32;
                js[got] = v;
                got += (v <= (uint32_t)(i - got)) ? 1 : 0;
            }
        } else {
            for (int t = 0; t < cnt; ++t) js[t] = (uint32_t)random_interval(g, (uint64_t)(i - t));
        }
        for (int t = 0; t < cnt; ++t) {
            if (t + AHEAD < cnt) __builtin_prefetch(order + js[t + AHEAD], 1, 0);
            const int64_t a = i - t;
            const uint32_t j = js[t];
            const int32_t tmp = order[j];
            order[j] = order[a];
            order[a] = tmp;
        }
        i -= cnt;
    }
    state_io[0] = (uint64_t)(g.state >> 64);
    state_io[1] = (uint64_t)g.state;
    *has_uint32_io = g.has_uint32;
    *uinteger_io = g.uinteger;
    return XC_OK;
}

// The same walk in two halves, so that the draws of sweep j + 1 (sequential PCG64 arithmetic, independent of the
// array) can be generated on one host thread while the swaps of sweep j (memory-bound) are applied on another:
//   xc_host_shuffle_draws  js[t] = the partner of step i = n-1-t, t = 0 .. n-2 (n - 1 entries); advances the generator
//   xc_host_shuffle_apply  the swaps, in place on the int32 array
int xc_host_shuffle_draws(uint64_t *state_io, int *has_uint32_io, uint32_t *uinteger_io, int64_t n, uint32_t *js) {
    if (!state_io || !has_uint32_io || !uinteger_io || n < 0 || (n > 1 && !js))
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_host_shuffle_draws: bad argument");
    if (n > 0xffffffffLL) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_host_shuffle_draws: n must fit 32 bits");
    Pcg64 g;
    g.state = ((u128)state_io[0] << 64) | (u128)state_io[1];
    g.inc = ((u128)state_io[2] << 64) | (u128)state_io[3];
    g.has_uint32 = *has_uint32_io;
    g.uinteger = *uinteger_io;
    int64_t i = n - 1;
    uint32_t *out = js;
    u128 jump_a[4], jump_c[4];
    {
        const u128 mult = ((u128)2549297995355413924ULL << 64) | (u128)4865540595714422341ULL;
        jump_a[0] = mult;
        jump_c[0] = g.inc;
        for (int q = 1; q < 4; ++q) {
            jump_a[q] = jump_a[q - 1] * mult;
            jump_c[q] = jump_c[q - 1] * mult + g.inc;
        }
    }
    while (i >= 1) {
        uint64_t mask = (uint64_t)i;
        mask |= mask >> 1;
        mask |= mask >> 2;
        mask |= mask >> 4;
        mask |= mask >> 8;
        mask |= mask >> 16;
        const int64_t low = (int64_t)(mask >> 1) + 1; // smallest bound with this mask
        const int64_t cnt = i - low + 1;              // steps i .. low share the mask
        const uint32_t m32 = (uint32_t)mask;
        int64_t got = 0;
        // branch-free filter of the candidate stream, as in xc_host_shuffle_pcg64, two candidates per 64-bit
        // output (numpy hands out the low half first and buffers the high half: pcg64_next32)
        if (g.has_uint32 && got < cnt) {
            g.has_uint32 = 0;
            const uint32_t v = g.uinteger & m32;
            out[got] = v;
            got += (v <= (uint32_t)(i - got)) ? 1 : 0;
        }
        // Eight candidates (four 64-bit outputs) at a time.  The four generator steps are computed from ONE state with
        // jump-ahead constants (state_{n+k} = A_k state_n + C_k), so they do not wait for each other; and the eight
        // bounds i - got differ by at most 7, so a candidate <= i - got - 8 is accepted and one > i - got is rejected
        // whatever the others do -- unless a candidate falls into that 8-wide band (rare: the range is ~i wide) the
        // accept flags are independent of `got` and only the one-cycle `got += flag` chain is left.
        while (got + 8 <= cnt) {
            const u128 s0 = g.state;
            const u128 s1 = s0 * jump_a[0] + jump_c[0], s2 = s0 * jump_a[1] + jump_c[1];
            const u128 s3 = s0 * jump_a[2] + jump_c[2], s4 = s0 * jump_a[3] + jump_c[3];
            g.state = s4;
            const uint64_t w0 = pcg64_output(s1), w1 = pcg64_output(s2), w2 = pcg64_output(s3), w3 = pcg64_output(s4);
            const uint32_t c[8] = {(uint32_t)w0 & m32, (uint32_t)(w0 >> 32) & m32, (uint32_t)w1 & m32, (uint32_t)(w1 >> 32) & m32,
                                   (uint32_t)w2 & m32, (uint32_t)(w2 >> 32) & m32, (uint32_t)w3 & m32, (uint32_t)(w3 >> 32) & m32};
            const uint32_t hi_b = (uint32_t)(i - got), lo_b = hi_b - 8u; // i - got >= low + 7 >= 8 here
            unsigned band = 0;
            for (int q = 0; q < 8; ++q) band |= (unsigned)((c[q] > lo_b) & (c[q] <= hi_b));
            if (!band) {
                for (int q = 0; q < 8; ++q) {
                    out[got] = c[q];
                    got += (c[q] <= lo_b) ? 1 : 0;
                }
            } else {
                for (int q = 0; q < 8; ++q) {
                    out[got] = c[q];
                    got += (c[q] <= (uint32_t)(i - got)) ? 1 : 0;
                }
            }
        }
        while (got < cnt) {
            const uint64_t w = pcg64_next64(g);
            const uint32_t v0 = (uint32_t)w & m32;
            out[got] = v0;
            got += (v0 <= (uint32_t)(i - got)) ? 1 : 0;
            if (got < cnt) {
                const uint32_t v1 = (uint32_t)(w >> 32) & m32;
                out[got] = v1;
                got += (v1 <= (uint32_t)(i - got)) ? 1 : 0;
            } else { // the run ended on the low half: the high half stays buffered for the next draw
                g.has_uint32 = 1;
                g.uinteger = (uint32_t)(w >> 32);
            }
        }
        out += cnt;
        i -= cnt;
    }
    state_io[0] = (uint64_t)(g.state >> 64);
    state_io[1] = (uint64_t)g.state;
    *has_uint32_io = g.has_uint32;
    *uinteger_io = g.uinteger;
    return XC_OK;
}

int xc_host_shuffle_apply(int64_t n, const uint32_t *js, int32_t *order) {
    if (n < 0 || (n > 1 && (!js || !order))) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_host_shuffle_apply: bad argument");
    const int64_t steps = n - 1;
    for (int64_t t = 0; t < steps; ++t) {
        if (t + 32 < steps) __builtin_prefetch(order + js[t + 32], 1, 0);
        const int64_t a = n - 1 - t;
        const uint32_t j = js[t];
        const int32_t tmp = order[j];
        order[j] = order[a];
        order[a] = tmp;
    }
    return XC_OK;
}

} // extern "C"
