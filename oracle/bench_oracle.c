/*
 * bench_oracle.c -- times the CPU restatement (toyni_oracle.c) in a plain C loop: bench.py's `cpu_baseline` leg.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY (see toyni_oracle.c).  Built ON THE MACHINE THAT RUNS IT with
 * `gcc -O3 -march=native` (the reference is `cargo build --release` on the host), as its own
 * executable: the portable -O3 libtoyni_oracle.so stays the checker, this binary is only ever timed.
 *
 *   bench_oracle ntt  <log_n> <seconds>   forward + inverse NTT of size 2^log_n (src/ntt.rs:24-66), 1 thread
 *   bench_oracle fold <log_m> <seconds>   fri_fold of a 2^log_m layer (src/math/fri.rs:27-48), 1 thread
 * Prints one line: <kind> <log> <reps> <elapsed seconds> <checksum>.
 */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

int orc_ntt_canonical(uint64_t *values, size_t n);
int orc_intt_canonical(uint64_t *values, size_t n);
int orc_domain_elements(uint64_t *out, size_t n, uint64_t shift);
int orc_fri_fold(uint64_t *out, const uint64_t *evals, size_t len, const uint64_t *xs, uint64_t beta);
void orc_fill_splitmix(uint64_t *out, size_t n, uint64_t seed);

static double now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s ntt|fold <log> <seconds>\n", argv[0]); return 2; }
    const int fold = strcmp(argv[1], "fold") == 0;
    const unsigned lg = (unsigned)atoi(argv[2]);
    const double budget = atof(argv[3]);
    if (lg > 27) return 2;
    const size_t n = (size_t)1 << lg;
    uint64_t *x = malloc(n * sizeof(uint64_t)), *ref = malloc(n * sizeof(uint64_t));
    uint64_t *xs = NULL, *out = NULL;
    if (!x || !ref) return 1;
    orc_fill_splitmix(x, n, 0xB45E);
    memcpy(ref, x, n * sizeof(uint64_t));
    if (fold) {
        xs = malloc(n * sizeof(uint64_t));
        out = malloc(n / 2 * sizeof(uint64_t) + 8);
        if (!xs || !out) return 1;
        orc_domain_elements(xs, n, 7);  /* the prover's coset, COSET_SHIFT = 7 (src/fibonacci.rs:16) */
    }
    long reps = 0;
    uint64_t sum = 0;
    const double t0 = now();
    double el;
    do {
        if (fold) {
            if (orc_fri_fold(out, x, n, xs, 123456789ULL + (uint64_t)reps)) return 1;
            sum += out[(size_t)reps % (n / 2 ? n / 2 : 1)];
        } else {
            if (orc_ntt_canonical(x, n) || orc_intt_canonical(x, n)) return 1;
            sum += x[(size_t)reps % n];
        }
        ++reps;
        el = now() - t0;
    } while (el < budget && reps < 100000);
    if (!fold && memcmp(x, ref, n * sizeof(uint64_t)) != 0) { fprintf(stderr, "round trip changed the data\n"); return 1; }
    printf("%s %u %ld %.6f %llu\n", fold ? "fold" : "ntt", lg, reps, el, (unsigned long long)sum);
    return 0;
}
