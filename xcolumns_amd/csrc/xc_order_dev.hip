// xc_order_dev.hip -- the visiting order of a sweep, generated ON the GPU: numpy's own stream.
//
// predict_using_bc_with_0approx draws ONE generator, np.random.default_rng(seed), and shuffles ONE array cumulatively,
// once per sweep (/root/reference/xcolumns/block_coordinate.py:413-419).  Generator.shuffle is a sequential
// Fisher-Yates walk (for i = n-1 .. 1: j = random_interval(i); swap(x[i], x[j])) whose partners come from masked
// rejection on the 32-bit halves of the PCG64 (XSL-RR 128/64) outputs (numpy/random/_generator.pyx `_shuffle_raw`,
// src/distributions/distributions.c `random_interval`, src/pcg64/pcg64.h `pcg64_next32`).  On the host that walk is
// 2-3 ms per million rows (csrc/xc_order.hip) behind a 0.5 ms sweep kernel.  Here the same permutation and the same
// generator position come out of four device steps, no host arithmetic and no host synchronisation per sweep (0.7 ms per
// million rows; 2.2 ms with the rejection walked by one workgroup, the form of this file until the grid-wide walk):
//
//   raw     every thread jumps the 128-bit LCG ahead to its own outputs (state_k = A^k state_0 + C_k) and writes
//           their 32-bit halves: the candidate stream, in order;
//   filter  which candidates does the rejection loop keep?  A candidate is kept iff it is <= the bound i at the time, and i
//           drops by one per kept candidate: a sequential recurrence.  One workgroup per batch of 8192 candidates, all
//           batches of the shuffle in ONE launch, in rounds: a batch is settled exactly from its entering bound (a candidate
//           <= i - 512 is kept and one > i dropped whatever the others of its group do; the few in between one by one),
//           and its entering bound is (n - 1) - what the batches before it kept in the previous round -- first guess: the
//           expectation.  The fixed point is the sequential walk; ~17 rounds at 1 M rows; no grid barrier (a workgroup
//           waits only for lower-numbered ones).  It leaves the partners j_t of the steps t = 0 .. n-2 (i_t = n-1-t) and
//           the number of candidates consumed (the generator's position for the next shuffle);
//   lists   per position q the steps whose partner is q (a linked list, atomicExch on a head);
//   resolve the value at position i_t before step t is the value the latest EARLIER step with partner i_t moved there
//           (or the old array's); these "who wrote it last" links form short chains, followed by pointer jumping; then
//           every position takes its final value directly.
#include "xc_common.h"
#include "xc_host.h"

namespace xc {

typedef unsigned __int128 u128;

// device header (uint64 words) at the start of the workspace
#define XC_OG_STATE_HI 0   /* base state B: output k (k >= 0) = xsl_rr(B stepped k + 1 times) */
#define XC_OG_STATE_LO 1
#define XC_OG_INC_HI 2
#define XC_OG_INC_LO 3
#define XC_OG_CONSUMED 4   /* 32-bit draws consumed so far: draw d is the low (d even) / high (d odd) half of output d / 2 */
#define XC_OG_FLAG 5       /* != 0: a shuffle failed (1: candidate buffer too short, 2: chains too long) */
#define XC_OG_SHUFFLES 6
#define XC_OG_CYCLES 8     /* the last walk's shader cycles and 100 MHz ticks (diagnostics) */
#define XC_OG_TICKS 9
#define XC_OG_BATCHES 10   /* batches (of 64 planes) the last walk left records for */
#define XC_OG_FIRST 11     /* raw index of the last walk's first candidate (0 or 1) */
#define XC_OG_ROUNDS 12    /* rounds of the last walk (diagnostics) */
#define XC_OG_T0 13        /* 100 MHz tick at which the last shuffle began (raw kernel) */
#define XC_OG_FALLBACK 14  /* walks the grid-wide kernel did not settle and the one-wavefront walk redid */
#define XC_OG_EPOCH 15     /* shuffle number the grid-wide walk tags its publications with */
// staged by the grid-wide walk, committed behind it (its late workgroups still read the words above)
#define XC_OG_NEXT_CONSUMED 16
#define XC_OG_NEXT_FLAG 17
#define XC_OG_NEXT_ROUNDS 18
#define XC_OG_UNSETTLED 19 /* != 0: the grid-wide walk gave up (rounds exhausted / a wait timed out): the one-wavefront walk redoes it */
#define XC_OG_ABORT 20
#define XC_OG_WORDS 24

__device__ __forceinline__ u128 og_mult() { return ((u128)2549297995355413924ULL << 64) | (u128)4865540595714422341ULL; }

__device__ __forceinline__ unsigned long long og_output(u128 s) { // XSL-RR of a state that has been stepped
    const unsigned long long hi = (unsigned long long)(s >> 64), lo = (unsigned long long)s;
    const unsigned rot = (unsigned)(hi >> 58);
    const unsigned long long x = hi ^ lo;
    return (x >> rot) | (x << ((64 - rot) & 63));
}

// raw[r] = draw (first + r) of the stream, r < count, where first = consumed & ~1 (the filter skips one if consumed is odd)
__global__ __launch_bounds__(256) void og_raw_kernel(unsigned long long *hdr, unsigned *raw, long long count) {
    constexpr int PER = 8; // outputs per thread
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g == 0) { // the shuffle begins: what the walk behind this kernel counts on
        hdr[XC_OG_T0] = __builtin_amdgcn_s_memrealtime();
        hdr[XC_OG_EPOCH] += 1ull;
        hdr[XC_OG_UNSETTLED] = 0ull;
        hdr[XC_OG_ABORT] = 0ull;
        hdr[XC_OG_NEXT_FLAG] = 0ull;
    }
    const long long r0 = g * PER * 2;
    if (r0 >= count) return;
    const u128 inc = ((u128)hdr[XC_OG_INC_HI] << 64) | hdr[XC_OG_INC_LO];
    u128 s = ((u128)hdr[XC_OG_STATE_HI] << 64) | hdr[XC_OG_STATE_LO];
    // advance by (first output of this thread) steps: acc <- compose
    unsigned long long delta = hdr[XC_OG_CONSUMED] / 2 + (unsigned long long)g * PER;
    u128 cur_mult = og_mult(), cur_plus = inc, acc_mult = 1, acc_plus = 0;
    while (delta) {
        if (delta & 1ull) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta >>= 1;
    }
    s = s * acc_mult + acc_plus;
    const u128 mult = og_mult();
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        s = s * mult + inc;
        const unsigned long long w = og_output(s);
        const long long r = r0 + 2 * q;
        if (r < count) raw[r] = (unsigned)w;
        if (r + 1 < count) raw[r + 1] = (unsigned)(w >> 32);
    }
}

__device__ __forceinline__ unsigned og_mask(unsigned i) {
    unsigned m = i;
    m |= m >> 1;
    m |= m >> 2;
    m |= m >> 4;
    m |= m >> 8;
    m |= m >> 16;
    return m;
}

// The rejection walk: which candidates does numpy's loop keep?  A candidate is kept iff it is <= the bound i at the time,
// and i drops by one per kept candidate -- a sequential recurrence.  It is walked in GROUPS of 512 candidates (8 planes of
// 64: lane l of plane p holds candidate p * 64 + l) by ONE 16-wavefront workgroup, 16 groups at a time:
//   * given the bound when its group begins, a wavefront settles the group exactly (og_group): a candidate <= i - 512 is
//     kept and one > i is dropped whatever the others of the group do, the few in between are settled one by one in
//     order; where the rejection mask changes within the next 512 steps (i crosses a power of two) or the walk ends,
//     the same per plane with a 64-wide margin, and the plane that holds the change itself candidate by candidate;
//   * the 16 entering bounds are guessed (the exact bound of the first group, minus the expected number of kept
//     candidates per group), all 16 groups are settled in parallel, the kept counts give new entering bounds, and the
//     round repeats until no bound moves.  A group's kept count depends on its entering bound only through the few
//     candidates within the error of the guess (512 / 2^20 of a unit per unit at 1 M rows), so two or three rounds
//     settle a batch; by induction over the groups the fixed point is the sequential walk.
// One wavefront alone needs ~2350 cycles per group (2.7 ms per million rows); the 16 of one CU in rounds 2.0 ms (the CU
// settles four groups at a time, one per SIMD: measured, profiles/r03_order_generator.txt).  The walk writes no partners: per plane it leaves a record {bound when the plane
// began, mask of the kept candidates} and og_compact_kernel (all CUs) turns the records into js[t], the partner of step
// t (i_t = n - 1 - t).
#define XC_OG_PLANES 8
#define XC_OG_GROUP (XC_OG_PLANES * 64)
#define XC_OG_WAVES 16 /* groups per batch = wavefronts of the walking workgroup */

#define XC_OG_KIND_PLANES 0u /* the group's 8 plane records say which candidates were kept */
#define XC_OG_KIND_LEAN 1u   /* kept <=> candidate <= bound - 512, or its position is one of the (at most 2) listed extras */
#define XC_OG_EXTRAS 2u

__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) { // DPP row shifts + row broadcasts (xc_common.h's reductions)
    v += dpp_src<0x111, 0xF>(0u, v);
    v += dpp_src<0x112, 0xF>(0u, v);
    v += dpp_src<0x114, 0xF>(0u, v);
    v += dpp_src<0x118, 0xF>(0u, v);
    v += dpp_src<0x142, 0xA>(0u, v);
    v += dpp_src<0x143, 0xC>(0u, v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

struct OgRec { // one plane of 64 candidates
    unsigned bound;     // i when the plane began
    unsigned one_by_one; // 1: the rejection mask changes inside the plane (or the walk ends there)
    unsigned long long kept;
};

__device__ __forceinline__ long long og_uni64(long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}

// one plane with a 64-wide margin (the mask is the same for its 64 candidates): returns the kept mask
__device__ __forceinline__ unsigned long long og_plane64(unsigned x, unsigned i) {
    unsigned long long a = __ballot(x <= i - 64u), b = __ballot(x <= i) & ~a;
    while (b) { // in order of arrival: kept iff x <= i - (candidates kept before it in this plane)
        const int e = __builtin_ctzll(b);
        b &= b - 1;
        const unsigned xe = (unsigned)__builtin_amdgcn_readlane((int)x, e);
        if (xe <= i - (unsigned)__popcll(a & ((1ull << e) - 1ull))) a |= 1ull << e;
    }
    return a;
}

// Settle ONE plane of 64 candidates (u: LDS) from its exact entering bound i, whatever the mask does in it: the kept mask,
// and whether the mask followed the bound through the plane (`follows`: og_compact_kernel's one_by_one).  `used` grows by
// the candidates consumed (fewer than 64 only where the walk ends).
__device__ __forceinline__ unsigned long long og_one_plane(const unsigned *u, int lane, long long steps, long long &got, unsigned &i, int &used,
                                                           unsigned &follows) {
    const unsigned mp = og_mask(i);
    const unsigned lowp = (mp >> 1) + 1u;
    unsigned long long a;
    follows = 0u;
    if (i >= lowp + 64u && steps - got > 64) {
        a = og_plane64(u[lane] & mp, i);
        const unsigned kept = (unsigned)__popcll(a);
        got += kept;
        i -= kept;
        used += 64;
    } else if (lowp >= 128u && i >= 64u && steps - got > 64) {
        // the plane in which the bound leaves its mask (once: the next mask holds for lowp / 2 >= 64 steps): settled as if
        // the mask held, which is right up to the candidate whose keeping takes the bound to lowp - 1; the candidates
        // behind it are settled from there with the next mask
        const unsigned xv = u[lane];
        const unsigned need = i - lowp + 1u; // candidates kept under this mask before the bound leaves it
        a = og_plane64(xv & mp, i);
        if ((unsigned)__popcll(a) >= need) {
            unsigned long long bts = a;
            for (unsigned q = 1u; q < need; ++q) bts &= bts - 1ull;
            const int e_star = __builtin_ctzll(bts); // the candidate that takes the bound to lowp - 1
            a &= (2ull << e_star) - 1ull;
            const unsigned x2 = lane > e_star ? (xv & (mp >> 1)) : 0xFFFFFFFFu;
            a |= og_plane64(x2, lowp - 1u);
        }
        follows = 1u;
        const unsigned kept = (unsigned)__popcll(a);
        got += kept;
        i -= kept;
        used += 64;
    } else { // the last steps of the walk (masks below 128, or its end): one candidate at a time
        a = 0ull;
        int e = 0;
        const unsigned xv = u[lane]; // the plane in a register: a lane read per step, not an LDS round trip
        for (; e < 64 && got < steps; ++e) {
            const unsigned ue = (unsigned)__builtin_amdgcn_readlane((int)xv, e);
            const unsigned xx = ue & og_mask(i);
            if (__builtin_amdgcn_readfirstlane((int)(xx <= i))) {
                a |= 1ull << e;
                ++got;
                --i;
            }
        }
        follows = 1u;
        used += e;
    }
    i = (unsigned)__builtin_amdgcn_readfirstlane((int)i);
    got = og_uni64(got);
    return a;
}

// Settle one group plane by plane: the path of the groups in which the rejection mask changes or the walk ends, and of the
// sequential tail (og_tail_batch).
__device__ __forceinline__ void og_group_planes(const unsigned *u, OgRec *rec, int lane, long long steps, long long &got, unsigned &i,
                                                int &used) {
    used = 0;
#pragma unroll 1
    for (int p = 0; p < XC_OG_PLANES; ++p) {
        if (got < steps) {
            const unsigned i_begin = i;
            unsigned follows = 0u;
            const unsigned long long a = og_one_plane(u + p * 64, lane, steps, got, i, used, follows);
            if (lane == p) {
                rec[p].bound = i_begin;
                rec[p].one_by_one = follows;
                rec[p].kept = a;
            }
        } else if (lane == p) {
            rec[p].bound = 0u;
            rec[p].one_by_one = 0u;
            rec[p].kept = 0ull;
        }
    }
}

// Settle one group of 512 candidates (u: LDS) that begins at bound i with `got` steps done: the plane records (rec: LDS),
// the candidates consumed (`used` < 512 only where the walk ends) and the bound / step count after it.
__device__ __forceinline__ void og_group(const unsigned *u, OgRec *rec, int lane, long long steps, long long &got, unsigned &i,
                                         int &used, unsigned &kind) {
    kind = XC_OG_KIND_PLANES;
    // the walk's counters are the same in all lanes: keep them in scalar registers whatever the compiler's divergence
    // analysis concludes (as vector values behind exec masks the walk took twice as long)
    i = (unsigned)__builtin_amdgcn_readfirstlane((int)i);
    got = og_uni64(got);
    const unsigned m = og_mask(i);
    const unsigned low = (m >> 1) + 1u; // smallest bound with this mask
    if (got < steps && i >= low + XC_OG_GROUP && steps - got > XC_OG_GROUP) {
        // the whole group shares the mask; only candidates in (i - 512, i] depend on the others.  Counted per LANE in
        // vector registers and summed over the wavefront once: the CU has ONE scalar unit for its 16 wavefronts, and a
        // walk that kept its bookkeeping in scalar registers (ballots, population counts: ~650 scalar instructions per
        // group) ran them one after the other
        unsigned x[XC_OG_PLANES];
        const unsigned thr = i - XC_OG_GROUP;
        unsigned cnt = 0u, band_any = 0u;
#pragma unroll
        for (int p = 0; p < XC_OG_PLANES; ++p) {
            x[p] = u[p * 64 + lane] & m;
            cnt += x[p] <= thr ? 1u : 0u;
            band_any |= (x[p] - thr - 1u) < (unsigned)XC_OG_GROUP ? 1u : 0u; // thr < x <= i
        }
        if (__ballot(band_any != 0u) == 0ull) {
            const unsigned kept = wave_sum_u32(cnt);
            kind = XC_OG_KIND_LEAN; // the kept candidates are exactly those <= bound - 512: og_compact_kernel finds them itself
            got += kept;
            i -= kept;
            used = XC_OG_GROUP;
            return;
        }
        // a few candidates lie within 512 of the bound: they are settled in order of arrival -- kept iff x <= i - (candidates
        // kept before it in this group) -- and those kept are listed by position (at most XC_OG_EXTRAS; more: plane records)
        unsigned long long sure[XC_OG_PLANES], band[XC_OG_PLANES];
        unsigned n_band = 0u;
#pragma unroll
        for (int p = 0; p < XC_OG_PLANES; ++p) {
            sure[p] = __ballot(x[p] <= thr);
            band[p] = __ballot((x[p] - thr - 1u) < (unsigned)XC_OG_GROUP);
            n_band += (unsigned)__popcll(band[p]);
        }
        if (n_band <= XC_OG_EXTRAS) {
            unsigned before = 0u, n_extra = 0u, packed = 0u; // packed: 9-bit positions of the kept in-between candidates
#pragma unroll
            for (int p = 0; p < XC_OG_PLANES; ++p) {
                unsigned long long b = band[p];
                unsigned kept_here = 0u;
                while (b) {
                    const int e = __builtin_ctzll(b);
                    b &= b - 1;
                    const unsigned xe = (unsigned)__builtin_amdgcn_readlane((int)x[p], e);
                    // kept before it: the sure ones of the earlier planes and of this plane's earlier lanes, and the extras so far
                    const unsigned bf = before + (unsigned)__popcll(sure[p] & ((1ull << e) - 1ull)) + kept_here;
                    if (xe <= i - bf) {
                        packed |= (unsigned)(p * 64 + e) << (9u * n_extra);
                        ++n_extra;
                        ++kept_here;
                    }
                }
                before += (unsigned)__popcll(sure[p]) + kept_here;
            }
            kind = XC_OG_KIND_LEAN | (n_extra << 4) | (packed << 8); // up to 2 extras ride in the kind word: 4 + 4 + 18 bits
            got += before;
            i -= before;
            used = XC_OG_GROUP;
            return;
        }
        unsigned before = 0u; // candidates of this group kept so far
#pragma unroll
        for (int p = 0; p < XC_OG_PLANES; ++p) {
            unsigned long long a = sure[p];
            unsigned long long b = band[p];
            while (b) {
                const int e = __builtin_ctzll(b);
                b &= b - 1;
                const unsigned xe = (unsigned)__builtin_amdgcn_readlane((int)x[p], e);
                const unsigned bf = before + (unsigned)__popcll(a & ((1ull << e) - 1ull));
                if (xe <= i - bf) a |= 1ull << e;
            }
            if (lane == p) {
                rec[p].bound = i - before;
                rec[p].one_by_one = 0u;
                rec[p].kept = a;
            }
            before += (unsigned)__popcll(a);
        }
        got += before;
        i -= before;
        used = XC_OG_GROUP;
        return;
    }
    // the mask changes within the next 512 steps, or the walk ends (or has ended): plane by plane
    og_group_planes(u, rec, lane, steps, got, i, used);
}

// LDS of a walking workgroup
struct OgShared {
    unsigned raw[XC_OG_WAVES * XC_OG_GROUP]; // the batch: 16 groups
    OgRec rec[XC_OG_WAVES * XC_OG_PLANES];
    unsigned kept[XC_OG_WAVES];              // candidates the group keeps, given its entering bound
    int used[XC_OG_WAVES];
    unsigned enter[XC_OG_WAVES + 1];         // entering bounds of this round (and of the next batch)
    unsigned kind[XC_OG_WAVES];
    int again;
};

// Settle the batch in S.raw from its (exact) entering bound: S.enter[] must hold a guess of the 16 entering bounds with
// S.enter[0] = the batch's; rounds until no bound moves.  Leaves S.rec / S.kind / S.used / S.kept and the exit bound in
// S.enter[16].  All 1024 threads call it; returns the rounds it took.
__device__ __forceinline__ int og_batch(OgShared &S, long long n, long long steps) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned i_batch = S.enter[0];
    int rounds = 0;
    for (;;) {
        unsigned i = S.enter[wv];
        long long got = (long long)(n - 1) - (long long)i;
        int used = 0;
        const unsigned i_in = (unsigned)__builtin_amdgcn_readfirstlane((int)i);
        unsigned kind = 0u;
        og_group(S.raw + wv * XC_OG_GROUP, S.rec + wv * XC_OG_PLANES, lane, steps, got, i, used, kind);
        if (lane == 0) {
            S.kept[wv] = i_in - i;
            S.used[wv] = used;
            S.kind[wv] = kind;
        }
        if (threadIdx.x == 0) S.again = 0;
        __syncthreads();
        if (wv == 0) { // new entering bounds from the kept counts (lane g: group g); has any moved?
            unsigned kq = lane < XC_OG_WAVES ? S.kept[lane] : 0u;
            unsigned incl = kq; // inclusive prefix over the lanes
#pragma unroll
            for (int o = 1; o < XC_OG_WAVES; o <<= 1) {
                const unsigned up = (unsigned)__shfl_up((int)incl, o, XC_WAVE);
                if (lane >= o) incl += up;
            }
            const unsigned enter = i_batch - (incl - kq);
            const bool moved = lane < XC_OG_WAVES && S.enter[lane] != enter;
            if (lane < XC_OG_WAVES) S.enter[lane] = enter;
            if (lane == XC_OG_WAVES - 1) S.enter[XC_OG_WAVES] = enter - kq;
            const unsigned long long mv = __ballot(moved);
            if (lane == 0) S.again = mv != 0ull ? 1 : 0;
        }
        ++rounds;
        __syncthreads();
        if (S.again == 0) break; // every group was settled from its true entering bound
    }
    return rounds;
}

// the settled batch's plane records -> recs[batch]
__device__ __forceinline__ void og_store_records(const OgShared &S, OgRec *recs, long long batch) {
    if (threadIdx.x < XC_OG_WAVES * XC_OG_PLANES) {
        OgRec rc = S.rec[threadIdx.x];
        const int g = threadIdx.x / XC_OG_PLANES;
        if ((S.kind[g] & 15u) == XC_OG_KIND_LEAN) { // one record for the group, in its first plane's slot
            rc.bound = S.enter[g];
            rc.one_by_one = 2u;
            rc.kept = (unsigned long long)(S.kind[g] >> 4); // extras: count (4 bits), then 9-bit positions
        }
        recs[batch * (XC_OG_WAVES * XC_OG_PLANES) + threadIdx.x] = rc;
    }
}

// The expected bound t candidates after bound i0: while the mask holds, (i + 1) decays like exp(-t / (mask + 1)).
__device__ __forceinline__ unsigned og_expected_from(unsigned i0, double t) {
    double i = (double)i0;
    while (t > 0.0 && i >= 1.0) {
        const unsigned m = og_mask((unsigned)i);
        const double range = (double)m + 1.0, low = (double)(m >> 1) + 1.0;
        const double t_plane = range * ::log((i + 1.0) / low); // candidates until the bound leaves this mask
        if (t < t_plane) {
            i = (i + 1.0) * ::exp(-t / range) - 1.0;
            break;
        }
        t -= t_plane;
        i = low - 1.0;
    }
    return i > 0.0 ? (unsigned)i : 0u;
}

// first guess of a batch's 16 entering bounds from its own: the expected bounds (a change of the mask inside the batch included)
__device__ __forceinline__ void og_guess_within(OgShared &S, unsigned i_batch) {
    if (threadIdx.x <= XC_OG_WAVES) S.enter[threadIdx.x] = threadIdx.x == 0 ? i_batch : og_expected_from(i_batch, (double)(XC_OG_GROUP * threadIdx.x));
}

// The batch in S.raw settled by ONE wavefront, plane after plane, from its (exact) entering bound: the tail of the walk
// (bounds below a batch's worth of candidates: the mask changes every few hundred steps and the walk ends -- guesses of the
// groups' entering bounds are worth nothing there, and og_group's own path for such groups is the slow one).  Leaves what
// og_batch leaves.  All threads call it.
__device__ __forceinline__ void og_tail_batch(OgShared &S, long long n, long long steps, unsigned bound) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (wv == 0) {
        unsigned i = (unsigned)__builtin_amdgcn_readfirstlane((int)bound);
        long long got = (long long)(n - 1) - (long long)i;
        for (int g = 0; g < XC_OG_WAVES; ++g) {
            const unsigned i_in = i;
            int used = 0;
            og_group_planes(S.raw + g * XC_OG_GROUP, S.rec + g * XC_OG_PLANES, lane, steps, got, i, used);
            if (lane == 0) {
                S.enter[g] = i_in;
                S.kept[g] = i_in - i;
                S.used[g] = used;
                S.kind[g] = XC_OG_KIND_PLANES;
            }
        }
        if (lane == 0) S.enter[XC_OG_WAVES] = i;
    }
    __syncthreads();
}

// Behind the grid-wide walk (below), ONE wavefront: it commits what that walk staged and returns -- or, if that walk gave up
// (never observed), walks the shuffle itself, plane after plane from the candidates in memory (og_one_plane: exact whatever
// the mask does): ~8 ms per million rows, the safety net that keeps the result exact.  One wavefront without LDS so that
// the launch finds a place at once on a GPU the sweeps keep full (as a 1024-thread workgroup it waited up to 0.65 ms).
__global__ __launch_bounds__(64) void og_walk_kernel(unsigned long long *hdr, const unsigned *raw, long long count, long long n,
                                                     OgRec *recs, int behind_grid, unsigned grid_batches) {
    const int lane = threadIdx.x;
    if (behind_grid && hdr[XC_OG_UNSETTLED] == 0ull) {
        if (lane == 0) { // commit the grid-wide walk's result (its workgroups have all left)
            const unsigned long long first = hdr[XC_OG_CONSUMED] & 1ull;
            if (hdr[XC_OG_NEXT_FLAG] != 0ull) hdr[XC_OG_FLAG] = hdr[XC_OG_NEXT_FLAG];
            hdr[XC_OG_BATCHES] = (unsigned long long)grid_batches;
            hdr[XC_OG_FIRST] = first;
            hdr[XC_OG_CONSUMED] = (hdr[XC_OG_CONSUMED] & ~1ull) + hdr[XC_OG_NEXT_CONSUMED];
            hdr[XC_OG_SHUFFLES] += 1ull;
            hdr[XC_OG_TICKS] = __builtin_amdgcn_s_memrealtime() - hdr[XC_OG_T0];
            hdr[XC_OG_CYCLES] = 24ull * hdr[XC_OG_TICKS]; // nominal (many workgroups)
            hdr[XC_OG_ROUNDS] = hdr[XC_OG_NEXT_ROUNDS];
        }
        return;
    }
    const long long first = (long long)__builtin_amdgcn_readfirstlane((int)(hdr[XC_OG_CONSUMED] & 1ull)); // odd: the low half of that output is spent
    const long long steps = n - 1;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned i = (unsigned)(n - 1);
    long long got = 0, plane = 0, consumed = first;
    bool fail = false;
    while (got < steps) {
        const long long base = first + plane * 64;
        if (base + 64 > count) { // the candidate buffer is used up (sized for the expectation + 8 sigma)
            fail = true;
            break;
        }
        const unsigned i_begin = i;
        int used = 0;
        unsigned follows = 0u;
        const unsigned long long a = og_one_plane(raw + base, lane, steps, got, i, used, follows);
        if (lane == 0) {
            OgRec rc;
            rc.bound = i_begin;
            rc.one_by_one = follows;
            rc.kept = a;
            recs[plane] = rc;
        }
        consumed = base + used;
        ++plane;
    }
    // the compaction works on whole batches of 128 planes: the rest of the last one holds nothing
    const long long batches = (plane + XC_OG_WAVES * XC_OG_PLANES - 1) / (XC_OG_WAVES * XC_OG_PLANES);
    for (long long q = plane + lane; q < batches * (XC_OG_WAVES * XC_OG_PLANES); q += 64) {
        OgRec z;
        z.bound = 0u;
        z.one_by_one = 0u;
        z.kept = 0ull;
        recs[q] = z;
    }
    if (lane == 0) {
        if (fail) hdr[XC_OG_FLAG] = 1ull;
        hdr[XC_OG_BATCHES] = (unsigned long long)batches;
        hdr[XC_OG_FIRST] = (unsigned long long)first;
        hdr[XC_OG_CONSUMED] = (hdr[XC_OG_CONSUMED] & ~1ull) + (unsigned long long)consumed;
        hdr[XC_OG_SHUFFLES] += 1ull;
        hdr[XC_OG_CYCLES] = __builtin_amdgcn_s_memtime() - c0;
        hdr[XC_OG_TICKS] = __builtin_amdgcn_s_memrealtime() - r0;
        hdr[XC_OG_ROUNDS] = (unsigned long long)plane;
        if (behind_grid) hdr[XC_OG_FALLBACK] += 1ull;
    }
}

// ---- the same walk, grid-wide ---------------------------------------------------------------------------------------
// One workgroup per BATCH (16 groups, 8192 candidates), all batches of the shuffle in one launch.  A batch settled from
// its entering bound keeps F_b(bound) candidates (og_batch: exact).  Round r of workgroup b:
//     bound_b(r) = (n - 1) - sum over the batches h < b of what they kept in round r - 1,
// first guess (round 0): the expected bound, in closed form per mask ((i + 1) decays like exp(-t / (mask + 1))).  Batch 0 is
// exact from the start and the error of the guesses dies out like x^k / k! (a batch's kept count depends on its entering
// bound only through the candidates between the guess and the truth: 8192 / 2^20 of a unit per unit at 1 M rows): 18 rounds
// at 1 M rows, 25 at 10 M, ~7 settles per batch (tests/studies/walk_rounds.c) -- against 172 batches one after the other.
//
// No grid barrier: workgroup b reads only what workgroups h < b published, from a history hist[round][workgroup] of
// 64-bit words {shuffle number, kept} written once each -- it never waits for a workgroup that the dispatcher starts after
// it (workgroups start in index order: the assumption of every decoupled look-back scan), so the kernel needs no
// co-residency and cannot deadlock against other kernels.  Workgroup b is FINAL in round r when every h < b published the
// same count in rounds r - 1 and r - 2: then all bounds below b are where they were a round ago, so are their counts, and by
// induction they never move again; it settles from its bound, fills the rest of its history with that count and leaves.
// The workgroup in which the walk ends stages the generator's new position; og_walk_kernel (one wavefront, behind this
// launch) commits it -- or, if this kernel gave up (rounds exhausted, a wait that timed out: never observed), walks the
// shuffle itself: the result is exact whatever happens here.
#define XC_OG_HIST_ROUNDS 96 /* 26 rounds at 1 M rows, 33 at 10 M, 43 at 20 M (tools/order_walk_trace.py) */
#define XC_OG_SPIN_LIMIT (1u << 21) /* polls of ~2 us each */

__device__ __forceinline__ unsigned long long og_ld64(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void og_st64(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(XC_OG_WAVES * 64) void og_walk_grid_kernel(unsigned long long *hdr, const unsigned *raw, long long n, OgRec *recs,
                                                                         unsigned long long *hist, int max_rounds,
                                                                         unsigned long long *dbg) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
    unsigned long long t_wait = 0ull, t_settle = 0ull;
    int settles = 0;
    __shared__ OgShared S;
    __shared__ unsigned s_sum[1], s_same[1], s_bad[1];
    __shared__ unsigned s_prev_kept[XC_OG_WAVES];
    const unsigned b = blockIdx.x, nb = gridDim.x;
    const long long first = (long long)__builtin_amdgcn_readfirstlane((int)(hdr[XC_OG_CONSUMED] & 1ull));
    const unsigned long long epoch = hdr[XC_OG_EPOCH] << 32;
    const long long steps = n - 1;
    {   // this batch's candidates (whole batches only: the host sizes the grid so)
        const long long b0 = first + ((long long)b * XC_OG_WAVES + wv) * XC_OG_GROUP;
#pragma unroll
        for (int p = 0; p < XC_OG_PLANES; ++p) S.raw[wv * XC_OG_GROUP + p * 64 + lane] = raw[b0 + p * 64 + lane];
    }
    unsigned bound_prev = 0xFFFFFFFFu, bound_round = 0u, kept_total = 0u, last_publish = 0x80000000u;
    int inner = 0, r = 0;
    bool gave_up = false, have_prev = false;
    for (;; ++r) {
        // -- this round's entering bound, and whether the batches before are at rest
        unsigned bound;
        bool final_now;
        if (b == 0u) {
            bound = (unsigned)(n - 1);
            final_now = true;
        } else if (r == 0) {
            bound = og_expected_from((unsigned)(n - 1), (double)b * (XC_OG_WAVES * XC_OG_GROUP));
            final_now = false;
        } else {
            // ONE wavefront reads what the batches before kept in round r - 1 and (from round 2 on) in round r - 2, waiting
            // for the words to carry this shuffle's tag; the others wait at the barrier (thousands of polling lanes starve
            // the memory system: 0.5 ms per round with all 1024 threads polling)
            const unsigned long long tw0 = __builtin_amdgcn_s_memrealtime();
            if (wv == 0) {
                unsigned sum = 0u, same = 1u, bad = 0u;
                for (unsigned h0 = 0u; h0 < b && bad == 0u; h0 += 64u) {
                    const unsigned h = h0 + (unsigned)lane;
                    const bool act = h < b;
                    unsigned long long w1 = 0ull, w2 = 0ull;
                    bool ok1 = !act, ok2 = !act || r < 2;
                    unsigned spins = 0u;
                    for (;;) {
                        if (!ok1) {
                            w1 = og_ld64(hist + (size_t)(r - 1) * nb + h);
                            ok1 = (w1 & 0xFFFFFFFF00000000ull) == epoch;
                        }
                        if (!ok2) {
                            w2 = og_ld64(hist + (size_t)(r - 2) * nb + h);
                            ok2 = (w2 & 0xFFFFFFFF00000000ull) == epoch;
                        }
                        if (__ballot(!(ok1 && ok2)) == 0ull) break;
                        __builtin_amdgcn_s_sleep(32);
                        if ((++spins & 255u) == 0u && (spins > XC_OG_SPIN_LIMIT || og_ld64(hdr + XC_OG_ABORT) != 0ull)) {
                            bad = 1u;
                            break;
                        }
                    }
                    if (act && bad == 0u) { // low word: exact flag (bit 31) | candidates kept
                        sum += (unsigned)w1 & 0x7FFFFFFFu;
                        same &= (unsigned)w1 >> 31; // an estimate (a batch of the tail that is not final yet) settles nothing
                        if (r >= 2) same &= (unsigned)w1 == (unsigned)w2 ? 1u : 0u;
                    }
                }
                sum = wave_sum_u32(sum);
                const unsigned long long sm = __ballot(same != 0u);
                if (lane == 0) {
                    s_sum[0] = sum;
                    s_same[0] = sm == ~0ull ? 1u : 0u;
                    s_bad[0] = bad;
                }
            }
            __syncthreads();
            const unsigned long long before = s_sum[0];
            const unsigned all_same = s_same[0], any_bad = s_bad[0];
            __syncthreads();
            t_wait += __builtin_amdgcn_s_memrealtime() - tw0;
            bound = before < (unsigned long long)(n - 1) ? (unsigned)((unsigned long long)(n - 1) - before) : 0u;
            final_now = r >= 2 && all_same != 0u;
            if (any_bad != 0u) gave_up = true;
        }
        if (!gave_up && r >= max_rounds - 1 && !final_now) gave_up = true; // the history is used up
        if (gave_up) break;
        // -- the tail of the walk (bounds within one batch's worth of candidates: the walk ends in this batch or the next;
        //    every batch behind it is tail too): until the batches before are at rest only an ESTIMATE is published -- the
        //    expected count, a function of the bound, not marked exact -- and the one exact settle, plane after plane by one
        //    wavefront, happens when the bound is final.  Batches of the tail become final one round after each other.
        const bool tail = bound <= (unsigned)(XC_OG_WAVES * XC_OG_GROUP);
        // Near the tail (bounds below 2^17: masks so small that every move of the bound changes many candidates' fate, a
        // settle takes ~40 us and the bound keeps moving for 13-16 rounds) a batch whose bound still moved by more than 32
        // since the last round does the same: an estimate now, the exact settle once the batches before have calmed down.
        const unsigned moved_by = bound > bound_round ? bound - bound_round : bound_round - bound;
        const bool restless = r > 0 && bound < (1u << 17) && moved_by > 32u;
        bound_round = bound;
        unsigned publish;
        if ((tail || restless) && !final_now) {
            publish = bound - og_expected_from(bound, (double)(XC_OG_WAVES * XC_OG_GROUP));
        } else {
            // -- settle the batch from that bound (unless it is the bound it was last settled from: same result)
            if (bound != bound_prev) {
                const unsigned long long ts0 = __builtin_amdgcn_s_memrealtime();
                ++settles;
                if (tail) {
                    og_tail_batch(S, n, steps, bound);
                    ++inner;
                } else {
                    if (!have_prev) {
                        og_guess_within(S, bound);
                    } else if (threadIdx.x <= XC_OG_WAVES) { // the groups' bounds move with the batch's: the previous kept counts as the guess
                        unsigned before = 0u;
                        for (unsigned w = 0; w < threadIdx.x; ++w) before += s_prev_kept[w];
                        S.enter[threadIdx.x] = before < bound ? bound - before : 0u;
                    }
                    __syncthreads();
                    inner += og_batch(S, n, steps);
                }
                og_store_records(S, recs, (long long)b);
                if (threadIdx.x < XC_OG_WAVES) s_prev_kept[threadIdx.x] = S.kept[threadIdx.x];
                have_prev = !tail;
                kept_total = bound - S.enter[XC_OG_WAVES];
                bound_prev = bound;
                __syncthreads();
                t_settle += __builtin_amdgcn_s_memrealtime() - ts0;
            }
            publish = kept_total | 0x80000000u;
        }
        last_publish = publish;
        if (final_now) break;
        if (threadIdx.x == 0) og_st64(hist + (size_t)r * nb + b, epoch | publish);
    }
    // -- leaving: the rest of the history (the batches behind read it), and the end of the walk
    if (gave_up && threadIdx.x == 0) {
        og_st64(hdr + XC_OG_UNSETTLED, 1ull);
        og_st64(hdr + XC_OG_ABORT, 1ull);
    }
    for (int q = r + (int)threadIdx.x; q < XC_OG_HIST_ROUNDS; q += blockDim.x) og_st64(hist + (size_t)q * nb + b, epoch | last_publish);
    if (!gave_up && threadIdx.x == 0) {
        const long long got_in = (long long)(n - 1) - (long long)bound_prev, got_out = got_in + kept_total;
        if (got_in < steps && got_out >= steps) { // the walk ends in this batch
            long long use = 0;
            for (int g = 0; g < XC_OG_WAVES; ++g) use += S.used[g];
            hdr[XC_OG_NEXT_CONSUMED] = (unsigned long long)(first + (long long)b * (XC_OG_WAVES * XC_OG_GROUP) + use);
            hdr[XC_OG_NEXT_ROUNDS] = (unsigned long long)(r + 1);
        } else if (b == nb - 1u && got_out < steps) {
            hdr[XC_OG_NEXT_FLAG] = 1ull; // the candidate buffer is used up
            hdr[XC_OG_NEXT_CONSUMED] = 0ull;
        }
    }
    if (dbg != nullptr && threadIdx.x == 0) { // diagnostics (100 MHz ticks since the shuffle began)
        const unsigned long long t0 = hdr[XC_OG_T0];
        dbg[4 * b + 0] = t_begin - t0;
        dbg[4 * b + 1] = __builtin_amdgcn_s_memrealtime() - t0;
        dbg[4 * b + 2] = (t_wait << 32) | (t_settle & 0xFFFFFFFFull);
        dbg[4 * b + 3] = ((unsigned long long)(r + 1) << 40) | ((unsigned long long)settles << 24) | (unsigned long long)inner;
    }
}

// js[t] from the records: a wavefront per plane; candidate e of plane q (raw index first + 64 q + e) that was kept is the
// partner of step t = (n - 1 - bound) + (kept candidates before e).  A LEAN group has one bound for its 8 planes and its
// kept candidates are those <= bound - 512.
__global__ __launch_bounds__(256) void og_compact_kernel(const unsigned long long *hdr, const unsigned *raw, const OgRec *recs,
                                                         long long n, unsigned *js) {
    const long long plane = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (plane >= (long long)hdr[XC_OG_BATCHES] * (XC_OG_WAVES * XC_OG_PLANES)) return;
    const long long group = plane / XC_OG_PLANES;
    const OgRec lead = recs[group * XC_OG_PLANES];
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (lead.one_by_one == 2u) {
        const unsigned m = og_mask(lead.bound), thr = lead.bound - XC_OG_GROUP;
        const long long r0 = (long long)hdr[XC_OG_FIRST] + group * XC_OG_GROUP;
        const int p = (int)(plane - group * XC_OG_PLANES);
        const unsigned n_extra = (unsigned)(lead.kept & 15ull);
        const unsigned ex0 = n_extra > 0 ? (unsigned)((lead.kept >> 4) & 511ull) : 0xFFFFu, ex1 = n_extra > 1 ? (unsigned)((lead.kept >> 13) & 511ull) : 0xFFFFu;
        long long t0 = n - 1 - (long long)lead.bound;
        for (int q = 0; q < p; ++q) { // kept in the planes before
            const unsigned pos = (unsigned)(q * 64 + lane);
            t0 += __popcll(__ballot((raw[r0 + q * 64 + lane] & m) <= thr || pos == ex0 || pos == ex1));
        }
        const unsigned x = raw[r0 + p * 64 + lane] & m;
        const unsigned pos = (unsigned)(p * 64 + lane);
        const unsigned long long kept = __ballot(x <= thr || pos == ex0 || pos == ex1);
        if ((kept >> lane) & 1ull) js[t0 + __popcll(kept & lt)] = x;
        return;
    }
    const OgRec rc = recs[plane];
    if (rc.kept == 0ull) return;
    const unsigned u = raw[(long long)hdr[XC_OG_FIRST] + plane * 64 + lane];
    const long long t0 = n - 1 - (long long)rc.bound;
    if (!rc.one_by_one) {
        if ((rc.kept >> lane) & 1ull) js[t0 + __popcll(rc.kept & lt)] = u & og_mask(rc.bound);
    } else { // the mask follows the bound through the plane
        const unsigned i_e = rc.bound - (unsigned)__popcll(rc.kept & lt); // the bound when candidate `lane` arrived
        if ((rc.kept >> lane) & 1ull) js[t0 + __popcll(rc.kept & lt)] = u & og_mask(i_e);
    }
}

// head[q] = a step with partner q, nxt[t] = another one (or -1): the steps whose partner is q, in any order
__global__ __launch_bounds__(256) void og_lists_kernel(long long steps, const unsigned *js, int *head, int *nxt) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= steps) return;
    nxt[t] = atomicExch(head + js[t], (int)t);
}

__device__ __forceinline__ int og_latest_before(const int *head, const int *nxt, unsigned q, int t) {
    int best = -1;
    for (int s = head[q]; s >= 0; s = nxt[s])
        if (s < t && s > best) best = s;
    return best;
}

// par[t]: the latest earlier step whose partner was position i_t (it moved the value that step t finds there), or t
// itself; src[t]: where position i_t's final value comes from: >= 0 a step s (the value step s found at ITS position
// i_s), -1 - q: the old array's position q
__global__ __launch_bounds__(256) void og_links_kernel(long long n, const unsigned *js, const int *head, const int *nxt, int *root,
                                                       int *src) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n - 1) return;
    const unsigned i = (unsigned)(n - 1 - t), j = js[t];
    const int p = og_latest_before(head, nxt, i, (int)t);
    root[t] = p >= 0 ? p : (int)t;
    if (j == i) {
        src[t] = (int)t; // a self swap: the value the step finds at its own position stays
    } else {
        const int s = og_latest_before(head, nxt, j, (int)t);
        src[t] = s >= 0 ? s : -1 - (int)j;
    }
}

__global__ __launch_bounds__(256) void og_jump_kernel(long long steps, int *root) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= steps) return;
    int r = root[t];
    int rr = root[r];
    if (rr != r) {
        const int r3 = root[rr];
        root[t] = r3; // three hops per pass (in place: every value read is an ancestor, so it stays correct)
    }
}

__global__ __launch_bounds__(256) void og_final_kernel(long long n, const int *root, const int *src, const int *head, const int *nxt,
                                                       const int32_t *a0, int32_t *out, unsigned long long *hdr) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t > n - 1) return;
    if (t == n - 1) { // position 0: the latest step of all with partner 0
        int best = -1;
        for (int s = head[0]; s >= 0; s = nxt[s]) best = s > best ? s : best;
        int32_t v = a0[0];
        if (best >= 0) {
            const int r = root[best];
            if (root[r] != r) hdr[XC_OG_FLAG] = 2ull;
            v = a0[n - 1 - r];
        }
        out[0] = v;
        return;
    }
    const int s = src[t];
    int32_t v;
    if (s >= 0) {
        const int r = root[s];
        if (root[r] != r) hdr[XC_OG_FLAG] = 2ull; // a chain longer than the jumps cover
        v = a0[n - 1 - r];
    } else {
        v = a0[-1 - s];
    }
    out[n - 1 - t] = v;
}

__global__ void og_init_kernel(unsigned long long *hdr, unsigned long long s_hi, unsigned long long s_lo, unsigned long long i_hi,
                               unsigned long long i_lo, unsigned long long consumed) {
    if (threadIdx.x == 0) {
        hdr[XC_OG_STATE_HI] = s_hi;
        hdr[XC_OG_STATE_LO] = s_lo;
        hdr[XC_OG_INC_HI] = i_hi;
        hdr[XC_OG_INC_LO] = i_lo;
        hdr[XC_OG_CONSUMED] = consumed;
        hdr[XC_OG_FLAG] = 0ull;
        hdr[XC_OG_SHUFFLES] = 0ull;
    }
}

__global__ __launch_bounds__(256) void og_arange_kernel(long long n, int32_t *a) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < n) a[t] = (int32_t)t;
}

static int64_t og_align(int64_t b) { return (b + 255) / 256 * 256; }
static int g_og_rounds = 0; // test knob: rounds the grid-wide walk may take (0: default; < 0: the one-wavefront walk only)

} // namespace xc

extern "C" {

// Candidates a shuffle of n entries may draw: the expectation of numpy's masked rejection plus eight standard deviations.
int xc_order_dev_candidates(int64_t n, int64_t *count) {
    if (!count || n < 0 || n > 0x7fffffffLL) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_order_dev_candidates: bad argument");
    double mean = 0.0, var = 0.0;
    int64_t i = n - 1;
    while (i >= 1) { // steps i .. low share a mask
        uint64_t mask = (uint64_t)i;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
        const int64_t low = (int64_t)(mask >> 1) + 1;
        const double range = (double)mask + 1.0;
        // sum over b = low .. i of range / (b + 1) and of (1 - p) / p^2 with p = (b + 1) / range, by the integral (smooth, large)
        if (i - low > 64) {
            const double lo1 = (double)low + 0.5, hi1 = (double)i + 1.5;
            mean += range * (__builtin_log(hi1) - __builtin_log(lo1));
            var += range * range * (1.0 / lo1 - 1.0 / hi1) - range * (__builtin_log(hi1) - __builtin_log(lo1));
        } else {
            for (int64_t b = low; b <= i; ++b) {
                const double p = ((double)b + 1.0) / range;
                mean += 1.0 / p;
                var += (1.0 - p) / (p * p);
            }
        }
        i = low - 1;
    }
    *count = (int64_t)(mean + 8.0 * __builtin_sqrt(var > 0 ? var : 0) + 2.0 * 8192.0); // the walk consumes whole batches of 8192
    return XC_OK;
}

int xc_order_dev_workspace_bytes(int64_t n, int64_t *bytes) {
    int64_t count = 0;
    int rc = xc_order_dev_candidates(n, &count);
    if (rc) return rc;
    if (!bytes) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_order_dev_workspace_bytes: NULL");
    // header | raw[count + 2] | plane records[count / 64 + 256] | js[n] | head[n] | nxt[n] | root[n] | src[n] | history of the
    // grid-wide walk [64 rounds][batches]
    *bytes = 256 + xc::og_align((count + 2) * 4) + xc::og_align((count / 64 + 256) * 16) + 5 * xc::og_align(n * 4 + 4) +
             xc::og_align((int64_t)(XC_OG_HIST_ROUNDS + 4) * (count / (XC_OG_WAVES * XC_OG_GROUP) + 1) * 8) + // + 4 words of diagnostics per batch
             xc::og_align(n * 4 + 4);                                                                          // the second set of partners
    return XC_OK;
}

// Bind the generator: state_inc = {state_hi, state_lo, inc_hi, inc_lo} of rng.bit_generator.state["state"] such that the
// NEXT 64-bit output is xsl_rr(state stepped once); consumed = 0, or 1 when the generator holds a buffered 32-bit half
// (then `state` must be the state BEFORE the step that produced that output).  order (int32[n], device) <- 0 .. n-1.
int xc_order_dev_begin(void *workspace, const uint64_t *state_inc, int consumed, int64_t n, int32_t *order, void *stream) {
    if (!workspace || !state_inc || n < 0 || (n > 0 && !order) || consumed < 0 || consumed > 1)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_order_dev_begin: bad argument");
    hipStream_t st = xc::as_stream(stream);
    int64_t bytes = 0;
    int rc = xc_order_dev_workspace_bytes(n, &bytes);
    if (rc) return rc;
    XC_HIP_TRY(hipMemsetAsync(workspace, 0, (size_t)bytes, st)); // (the grid-wide walk's history must not carry a tag yet)
    hipLaunchKernelGGL(xc::og_init_kernel, dim3(1), dim3(64), 0, st, static_cast<unsigned long long *>(workspace), state_inc[0],
                       state_inc[1], state_inc[2], state_inc[3], (unsigned long long)consumed);
    if (n > 0) hipLaunchKernelGGL(xc::og_arange_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (long long)n, order);
    XC_CHECK_LAUNCH("og_init_kernel");
    return XC_OK;
}

// The two halves of one Generator.shuffle, for two streams (utils.DeviceNumpyOrders): the DRAW -- candidate stream, rejection
// walk, the partners js[slot] of all steps; it advances the generator's position in the workspace, and the draw of the NEXT
// shuffle needs nothing else of this one -- and the APPLY: the Fisher-Yates swaps with those partners, order_out <- the
// shuffle of order_in.  slot 0 / 1: two sets of partners, so that the next draw may run while this apply still reads its set
// (the caller orders draw k + 2 behind apply k).
struct OgLayout {
    unsigned long long *hdr;
    unsigned *raw;
    xc::OgRec *recs;
    unsigned *js[2];
    int *head, *nxt, *root, *src;
    unsigned long long *hist;
    int64_t count;
    unsigned batches;
};

static int og_layout(void *workspace, int64_t n, OgLayout &L) {
    int64_t count = 0;
    int rc = xc_order_dev_candidates(n, &count);
    if (rc) return rc;
    count += 2;
    char *w = static_cast<char *>(workspace);
    L.hdr = reinterpret_cast<unsigned long long *>(w);
    char *q = w + 256;
    L.raw = reinterpret_cast<unsigned *>(q);
    q += xc::og_align(count * 4);
    L.recs = reinterpret_cast<xc::OgRec *>(q);
    q += xc::og_align(((count - 2) / 64 + 256) * 16);
    const int64_t seg = xc::og_align(n * 4 + 4);
    L.js[0] = reinterpret_cast<unsigned *>(q);
    L.head = reinterpret_cast<int *>(q + seg);
    L.nxt = reinterpret_cast<int *>(q + 2 * seg);
    L.root = reinterpret_cast<int *>(q + 3 * seg);
    L.src = reinterpret_cast<int *>(q + 4 * seg);
    L.hist = reinterpret_cast<unsigned long long *>(q + 5 * seg);
    q += 5 * seg + xc::og_align((int64_t)(XC_OG_HIST_ROUNDS + 4) * ((count - 2) / (XC_OG_WAVES * XC_OG_GROUP) + 1) * 8);
    L.js[1] = reinterpret_cast<unsigned *>(q);
    L.count = count;
    L.batches = (unsigned)((count - 2) / (XC_OG_WAVES * XC_OG_GROUP)); // whole batches behind the first candidate (raw index 0 or 1)
    return XC_OK;
}

int xc_order_dev_draw(void *workspace, int64_t n, int slot, void *stream) {
    if (!workspace || n < 2 || slot < 0 || slot > 1) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_order_dev_draw: bad argument");
    hipStream_t st = xc::as_stream(stream);
    OgLayout L;
    int rc = og_layout(workspace, n, L);
    if (rc) return rc;
    int max_rounds = XC_OG_HIST_ROUNDS;
    if (xc::g_og_rounds > 0 && xc::g_og_rounds < XC_OG_HIST_ROUNDS) max_rounds = xc::g_og_rounds;
    hipLaunchKernelGGL(xc::og_raw_kernel, dim3((unsigned)((L.count / 16 + 256) / 256)), dim3(256), 0, st, L.hdr, L.raw, (long long)L.count);
    const bool grid = L.batches >= 2 && xc::g_og_rounds >= 0;
    if (grid)
        hipLaunchKernelGGL(xc::og_walk_grid_kernel, dim3(L.batches), dim3(XC_OG_WAVES * 64), 0, st, L.hdr, L.raw, (long long)n, L.recs, L.hist,
                           max_rounds, L.hist + (size_t)XC_OG_HIST_ROUNDS * L.batches);
    hipLaunchKernelGGL(xc::og_walk_kernel, dim3(1), dim3(64), 0, st, L.hdr, L.raw, (long long)L.count, (long long)n, L.recs, grid ? 1 : 0,
                       L.batches);
    hipLaunchKernelGGL(xc::og_compact_kernel, dim3((unsigned)((L.count / 64 + 4) / 4)), dim3(256), 0, st, L.hdr, L.raw, L.recs, (long long)n,
                       L.js[slot]);
    XC_CHECK_LAUNCH("order generator kernels (draw)");
    return XC_OK;
}

int xc_order_dev_apply(void *workspace, int64_t n, int slot, const int32_t *order_in, int32_t *order_out, void *stream) {
    if (!workspace || n < 2 || slot < 0 || slot > 1 || !order_in || !order_out || order_in == order_out)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_order_dev_apply: bad argument");
    hipStream_t st = xc::as_stream(stream);
    OgLayout L;
    int rc = og_layout(workspace, n, L);
    if (rc) return rc;
    const long long steps = n - 1;
    const unsigned gb = (unsigned)((steps + 255) / 256);
    XC_HIP_TRY(hipMemsetAsync(L.head, 0xff, (size_t)n * 4, st));
    hipLaunchKernelGGL(xc::og_lists_kernel, dim3(gb), dim3(256), 0, st, steps, L.js[slot], L.head, L.nxt);
    hipLaunchKernelGGL(xc::og_links_kernel, dim3(gb), dim3(256), 0, st, (long long)n, L.js[slot], L.head, L.nxt, L.root, L.src);
    for (int pass = 0; pass < 4; ++pass) hipLaunchKernelGGL(xc::og_jump_kernel, dim3(gb), dim3(256), 0, st, steps, L.root);
    hipLaunchKernelGGL(xc::og_final_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (long long)n, L.root, L.src, L.head, L.nxt,
                       order_in, order_out, L.hdr);
    XC_CHECK_LAUNCH("order generator kernels (apply)");
    return XC_OK;
}

// One Generator.shuffle on one stream: order_out <- the shuffle of order_in (both int32[n] on the device, different buffers);
// the generator position in the workspace advances.  Asynchronous on `stream`.
int xc_order_dev_shuffle(void *workspace, int64_t n, const int32_t *order_in, int32_t *order_out, void *stream) {
    if (!workspace || n < 0 || (n > 0 && (!order_in || !order_out)) || order_in == order_out)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_order_dev_shuffle: bad argument");
    if (n == 0) return XC_OK;
    if (n == 1) {
        XC_HIP_TRY(hipMemcpyAsync(order_out, order_in, 4, hipMemcpyDeviceToDevice, xc::as_stream(stream)));
        return XC_OK;
    }
    int rc = xc_order_dev_draw(workspace, n, 0, stream);
    if (rc) return rc;
    return xc_order_dev_apply(workspace, n, 0, order_in, order_out, stream);
}

// Diagnostics of the last grid-wide walk: per batch {start, end (100 MHz ticks since the shuffle began), ticks waiting <<
// 32 | ticks settling, rounds << 40 | settles << 24 | inner rounds}; out must hold 4 * batches words.  Blocks on the stream.
int xc_order_dev_walk_trace(void *workspace, int64_t n, int64_t *out, int64_t *batches_out, void *stream) {
    if (!workspace || !batches_out || n < 2) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_order_dev_walk_trace: bad argument");
    int64_t count = 0;
    int rc = xc_order_dev_candidates(n, &count);
    if (rc) return rc;
    const int64_t batches = count / (XC_OG_WAVES * XC_OG_GROUP);
    *batches_out = batches;
    if (!out) return XC_OK;
    const char *q = static_cast<const char *>(workspace) + 256 + xc::og_align((count + 2) * 4) + xc::og_align((count / 64 + 256) * 16) +
                    5 * xc::og_align(n * 4 + 4);
    const unsigned long long *dbg = reinterpret_cast<const unsigned long long *>(q) + (size_t)XC_OG_HIST_ROUNDS * batches;
    hipStream_t st = xc::as_stream(stream);
    XC_HIP_TRY(hipMemcpyAsync(out, dbg, (size_t)batches * 32, hipMemcpyDeviceToHost, st));
    XC_HIP_TRY(hipStreamSynchronize(st));
    return XC_OK;
}

// Test knob: rounds the grid-wide walk may take (0 = default, 96).  With 1 or 2 it gives up on a large shuffle and the
// one-wavefront walk behind it redoes the shuffle: same result.  Negative: the one-wavefront walk only.
int xc_order_dev_set_rounds(int rounds) {
    xc::g_og_rounds = rounds;
    return XC_OK;
}

// {failure flag (0 = every shuffle so far is numpy's), 32-bit draws consumed, shuffles, shader cycles and 100 MHz ticks of
// the last rejection walk, its rounds and batches}; blocks on the stream.
int xc_order_dev_status(void *workspace, int64_t *out3_host, void *stream) {
    if (!workspace || !out3_host) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_order_dev_status: NULL");
    unsigned long long tmp[XC_OG_WORDS];
    hipStream_t st = xc::as_stream(stream);
    XC_HIP_TRY(hipMemcpyAsync(tmp, workspace, sizeof(tmp), hipMemcpyDeviceToHost, st));
    XC_HIP_TRY(hipStreamSynchronize(st));
    out3_host[0] = (int64_t)tmp[XC_OG_FLAG];
    out3_host[1] = (int64_t)tmp[XC_OG_CONSUMED];
    out3_host[2] = (int64_t)tmp[XC_OG_SHUFFLES];
    out3_host[3] = (int64_t)tmp[XC_OG_CYCLES];
    out3_host[4] = (int64_t)tmp[XC_OG_TICKS];
    out3_host[5] = (int64_t)tmp[XC_OG_ROUNDS];
    out3_host[6] = (int64_t)tmp[XC_OG_BATCHES];
    out3_host[7] = (int64_t)tmp[XC_OG_FALLBACK];
    return XC_OK;
}

} // extern "C"
