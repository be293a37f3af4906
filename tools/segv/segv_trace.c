/* Diagnostic aid (not product): prints a native backtrace on SIGSEGV / SIGABRT / SIGBUS and re-raises.
   Loaded into a Python process with ctypes.CDLL before the code under investigation runs.
   gcc -shared -fPIC -O1 -o libsegv_trace.so segv_trace.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void handler(int sig) {
    void* frames[64];
    const char msg[] = "\n=== native backtrace (segv_trace) ===\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    const char end[] = "=== end of native backtrace ===\n";
    (void)!write(2, end, sizeof(end) - 1);
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void) {
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = handler;
    sigemptyset(&sa.sa_mask);
    sa.sa_flags = SA_NODEFER | SA_ONSTACK;
    static char stack[1 << 16];
    stack_t ss = {.ss_sp = stack, .ss_size = sizeof(stack), .ss_flags = 0};
    sigaltstack(&ss, 0);
    sigaction(SIGSEGV, &sa, 0);
    sigaction(SIGBUS, &sa, 0);
    sigaction(SIGABRT, &sa, 0);
}
