/* LD_PRELOAD helper for a box without a debugger: on SIGSEGV / SIGABRT / SIGBUS every frame of the faulting thread goes to stderr with its
 * module and offset (backtrace_symbols_fd), then the default action.   gcc -O1 -g -shared -fPIC tools/segv_trace.c -o tools/bin/segv_trace.so
 *   LD_PRELOAD=tools/bin/segv_trace.so python -m pytest -p no:faulthandler ...          (pytest's own handler would replace this one) */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>
static void on_fault(int sig, siginfo_t* si, void* uc)
{
    (void)uc;
    void* frames[96];
    const char msg[] = "\nsegv_trace: fatal signal, backtrace of the faulting thread:\n";
    (void)!write(2, msg, sizeof msg - 1);
    const int n = backtrace(frames, 96);
    backtrace_symbols_fd(frames, n, 2);
    (void)si;
    signal(sig, SIG_DFL);
    raise(sig);
}
static char alt_stack[1 << 16];
__attribute__((constructor)) static void install(void)
{
    stack_t ss; ss.ss_sp = alt_stack; ss.ss_size = sizeof alt_stack; ss.ss_flags = 0;
    sigaltstack(&ss, 0);                      /* (the main thread's: a stack overflow must still be able to say so) */
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_fault;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, 0); sigaction(SIGBUS, &sa, 0); sigaction(SIGABRT, &sa, 0);
}
