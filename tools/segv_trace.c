/* LD_PRELOAD helper for debugging on the GPU box: a backtrace on SIGSEGV (gcc -shared -fPIC -o /tmp/segv.so tools/segv_trace.c) */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
static void h(int s) { void *b[48]; int n = backtrace(b, 48); (void)s; backtrace_symbols_fd(b, n, 2); _exit(139); }
__attribute__((constructor)) static void init(void) { signal(SIGSEGV, h); }
