/* oracle/exhaust_log.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * The quality normals of mode B (D_QUAL0) are Marsaglia polar tries on two 16-bit operands: v = (h - 32768) * 2^-15.  The
 * reference computes fac = sqrt(-2 log(rsq) / rsq) with glibc's log (dwgsim.c:156-175); the oracle and the kernels use det_log (fdlibm's
 * e_log restated in IEEE operations, oracle_det_log), whose last bit differs from glibc's on a few per cent of the arguments.  The only
 * consumer of such a normal is the truncation (int)(nrm * sigma + 0.5) (dwgsim.c:912).  This program settles, for EVERY one of the 2^32
 * possible tries, whether the two logs can ever give a different quality offset -- for each sigma on the command line.
 *
 * rsq depends on (|s1|, |s2|) only and both variates of a try share fac, so the 2^32 tries fold into the 2^29 pairs 0 <= a <= b <= 32768:
 * where the two logs agree bitwise nothing can differ; where they do not, the four values (+-a, +-b) * 2^-15 * fac * sigma + 0.5 are
 * truncated with both facs and compared.  Output (one line, parsed by tests/test_oracle_units.py):
 *   tries_accepted N  radii R  log_bits_differ D  offsets_differ[sigma] K ...
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

double oracle_det_log(double x);

#define MAXS 16
static int n_sigma;
static double sigma[MAXS];
typedef struct { int t, nt; uint64_t accepted, radii, logdiff, offdiff[MAXS]; } job_t;

static void *work(void *p)
{
    job_t *j = (job_t *)p;
    for (int a = j->t; a <= 32768; a += j->nt) {
        for (int b = a; b <= 32768; ++b) {
            const double v1 = (double)a * 0x1p-15, v2 = (double)b * 0x1p-15;
            const double rsq = v1 * v1 + v2 * v2;                    /* exact: an integer below 2^31 times 2^-30 */
            if (rsq >= 1.0 || rsq == 0.0) continue;
            /* signs: s = -a and s = +a are both operands unless a = 0 (and |s| = 32768 exists only as -32768, which is never accepted) */
            const uint64_t mult = (uint64_t)(a ? 2 : 1) * (uint64_t)(b ? 2 : 1) * (uint64_t)(a != b ? 2 : 1);
            j->accepted += mult; j->radii++;
            const double l1 = log(rsq), l2 = oracle_det_log(rsq);
            if (memcmp(&l1, &l2, 8) == 0) continue;
            j->logdiff++;
            const double f1 = sqrt(-2.0 * l1 / rsq), f2 = sqrt(-2.0 * l2 / rsq);
            if (memcmp(&f1, &f2, 8) == 0) continue;
            const double vs[4] = { v1, -v1, v2, -v2 };
            for (int s = 0; s < n_sigma; ++s)
                for (int k = 0; k < 4; ++k)
                    if ((int)((vs[k] * f1) * sigma[s] + 0.5) != (int)((vs[k] * f2) * sigma[s] + 0.5)) j->offdiff[s]++;
        }
    }
    return 0;
}

int main(int argc, char **argv)
{
    int nt = argc > 1 ? atoi(argv[1]) : 8;
    if (nt < 1) nt = 1;
    if (nt > 256) nt = 256;
    for (int i = 2; i < argc && n_sigma < MAXS; ++i) sigma[n_sigma++] = atof(argv[i]);
    if (!n_sigma) { sigma[0] = 2.0; n_sigma = 1; }
    pthread_t th[256]; static job_t jobs[256];
    for (int t = 0; t < nt; ++t) { jobs[t].t = t; jobs[t].nt = nt; pthread_create(&th[t], 0, work, &jobs[t]); }
    job_t sum; memset(&sum, 0, sizeof sum);
    for (int t = 0; t < nt; ++t) {
        pthread_join(th[t], 0);
        sum.accepted += jobs[t].accepted; sum.radii += jobs[t].radii; sum.logdiff += jobs[t].logdiff;
        for (int s = 0; s < n_sigma; ++s) sum.offdiff[s] += jobs[t].offdiff[s];
    }
    printf("tries_accepted %llu radii %llu log_bits_differ %llu", (unsigned long long)sum.accepted, (unsigned long long)sum.radii, (unsigned long long)sum.logdiff);
    for (int s = 0; s < n_sigma; ++s) printf(" offsets_differ[%g] %llu", sigma[s], (unsigned long long)sum.offdiff[s]);
    printf("\n");
    return 0;
}
