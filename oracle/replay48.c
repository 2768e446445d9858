/* oracle/replay48.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * LD_PRELOAD shim for the UNMODIFIED reference binary (oracle/_ref/dwgsim): its drand48() -- the one source of randomness of src/dwgsim.c,
 * src/mut.c and src/dwgsim_opt.c (ran_normal, dwgsim.c:156-175, draws from it too) -- returns the doubles of the file named by REPLAY48_FILE,
 * in order.  That file is what `dwgsim_oracle --rng philox --dump-draws FILE` wrote: every uniform of the Philox stream ("mode B", the stream the
 * HIP kernels draw from) in the order the oracle consumed it.  If the reference, fed that stream, writes the same five files as the oracle in
 * mode B -- and it must consume exactly the draws that were dumped, no more, no fewer -- then mode B is the reference's own algorithm under
 * another generator, checked directly and not only "by construction" (tests/test_replay_parity.py).  Valid on configurations
 * that draw no normal (-2 0 with -Q 0 or -q): the reference's Box-Muller cache leaks a variate from pair to pair (dwgsim.c:158-159), which a
 * counter-based stream deliberately does not reproduce.
 * At exit the number of draws served / available goes to the file named by REPLAY48_REPORT ("served available\n"). */
#include <stdio.h>
#include <stdlib.h>

static FILE *fp; static unsigned long long served, avail; static int started;

static void report(void)
{
    const char *rp = getenv("REPLAY48_REPORT");
    if (rp) { FILE *f = fopen(rp, "w"); if (f) { fprintf(f, "%llu %llu\n", served, avail); fclose(f); } }
}
static void start(void)
{
    started = 1;
    const char *p = getenv("REPLAY48_FILE");
    if (!p || !(fp = fopen(p, "rb"))) { fprintf(stderr, "replay48: cannot open REPLAY48_FILE\n"); _Exit(97); }
    fseek(fp, 0, SEEK_END); avail = (unsigned long long)ftell(fp) / sizeof(double); fseek(fp, 0, SEEK_SET);
    setvbuf(fp, NULL, _IOFBF, 1 << 20);
    atexit(report);
}
double drand48(void)
{
    double u;
    if (!started) start();
    if (fread(&u, sizeof u, 1, fp) != 1) { fprintf(stderr, "replay48: the reference asked for draw %llu, the stream holds %llu\n", served + 1, avail); report(); _Exit(98); }
    ++served;
    return u;
}
