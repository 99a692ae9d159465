// Study behind csrc/xc_order_dev.hip's grid-wide rejection walk (numpy's masked rejection, Generator.shuffle): how many
// rounds does "every segment of S candidates settled exactly from its entering bound, entering bounds from the prefix sums
// of the previous round's kept counts (first guess: the expectation)" take to reach the sequential walk's fixed point?
//   gcc -O2 -o /tmp/walk_rounds tests/studies/walk_rounds.c && /tmp/walk_rounds <rows> <segment> <guess 0|1> <seed>
// Measured (segment 512 / 8192 candidates): 50 K rows 14-16 rounds, 150 K 18-22, 1 M 24-27 / 18, 5 M 29-33, 10 M 36 / 25;
// 6.5-10 settles per segment; with no guess (every bound = n - 1) 31 instead of 23 at 1 M.
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
static uint64_t s[2] = {0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull};
static inline uint64_t nxt(void) { uint64_t a = s[0], b = s[1]; s[0] = b; a ^= a << 23; s[1] = a ^ b ^ (a >> 17) ^ (b >> 26); return s[1] + b; }
static inline uint32_t mask_of(uint32_t i) { uint32_t m = i; m |= m >> 1; m |= m >> 2; m |= m >> 4; m |= m >> 8; m |= m >> 16; return m; }
int main(int argc, char **argv) {
    if (argc > 4) { s[0] ^= (uint64_t)atol(argv[4]) * 0x9E3779B97F4A7C15ull; for (int q = 0; q < 8; ++q) nxt(); }
    long n = argc > 1 ? atol(argv[1]) : 1000000; long S = argc > 2 ? atol(argv[2]) : 512; int guess = argc > 3 ? atoi(argv[3]) : 1;
    long T = (long)(n * 1.5) + 4096; T = (T + S - 1) / S * S;
    uint32_t *c = malloc(T * 4); for (long t = 0; t < T; ++t) c[t] = (uint32_t)(nxt() >> 32);
    long NS = T / S; long *enter = malloc((NS + 1) * 8), *kept = malloc(NS * 8), *last = malloc(NS * 8);
    // first guess
    { double i = n - 1; for (long g = 0; g <= NS; ++g) { enter[g] = guess ? (long)i : (g == 0 ? n - 1 : n - 1); double p = (i + 1.0) / ((double)mask_of((uint32_t)(i > 1 ? i : 1)) + 1.0); i -= p * S; if (i < 0) i = 0; } enter[0] = n - 1; }
    for (long g = 0; g < NS; ++g) last[g] = -1;
    int rounds = 0; long work = 0;
    for (;;) {
        ++rounds;
        for (long g = 0; g < NS; ++g) {
            if (enter[g] == last[g]) continue; last[g] = enter[g]; ++work;
            long i = enter[g], k = 0;
            for (long t = g * S; t < (g + 1) * S && i >= 1; ++t) if ((c[t] & mask_of((uint32_t)i)) <= (uint32_t)i) { --i; ++k; }
            kept[g] = k;
        }
        int moved = 0; long i = n - 1;
        for (long g = 0; g < NS; ++g) { if (enter[g] != i) { enter[g] = i; ++moved; } i -= kept[g]; if (i < 0) i = 0; }
        if (!moved) break;
        if (rounds > 100000) break;
    }
    printf("n=%ld segment=%ld segments=%ld guess=%d: rounds %d, segment settles %ld (%.2f per segment)\n", n, S, NS, guess, rounds, work, (double)work / NS);
    return 0;
}
