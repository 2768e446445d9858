/*
 * oracle/nullgz.c -- TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * LD_PRELOAD shim that turns the zlib gz* writer calls of the unmodified reference (oracle/_ref/dwgsim) into no-ops, so that
 * bench.py's cpu_baseline can report what the reference's simulation itself costs without its byte-at-a-time gzip output
 * (SURVEY.md 8(d) "output to a null sink"; timing seams dwgsim.c:919-981, :1150-1158).  Nothing else uses it.
 */
#include <stdarg.h>
#include <stddef.h>

typedef struct gzFile_s *gzFile;
static int dummy;

gzFile gzopen(const char *path, const char *mode) { (void)path; (void)mode; return (gzFile)&dummy; }
gzFile gzopen64(const char *path, const char *mode) { (void)path; (void)mode; return (gzFile)&dummy; }
gzFile gzdopen(int fd, const char *mode) { (void)fd; (void)mode; return (gzFile)&dummy; }
int gzputc(gzFile f, int c) { (void)f; return c; }
int gzputs(gzFile f, const char *s) { (void)f; (void)s; return 1; }
int gzwrite(gzFile f, const void *buf, unsigned len) { (void)f; (void)buf; return (int)len; }
int gzprintf(gzFile f, const char *fmt, ...) { (void)f; (void)fmt; return 1; }
int gzflush(gzFile f, int flush) { (void)f; (void)flush; return 0; }
int gzclose(gzFile f) { (void)f; return 0; }
