/* hang_probe.c -- LD_PRELOAD helper for tools/hang/hang_hunt.py (no debugger exists in this image): on SIGUSR2 a thread writes its own call
 * stack (return addresses + what dladdr knows) to the file named by RCGPU_HANG_DUMP.  The hunter sends the signal to every thread of a
 * process that ran into its time-out (tgkill) and resolves the addresses against `nm` afterwards.  TEST INFRASTRUCTURE.
 *   gcc -O1 -fPIC -shared tools/hang/hang_probe.c -o tools/bin/hang_probe.so -ldl */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

static int g_fd = -1;

static void on_usr2(int sig)
{
    (void)sig;
    if (g_fd < 0) return;
    void* pc[64];
    const int n = backtrace(pc, 64);
    char line[512];
    int len = snprintf(line, sizeof line, "thread %ld: %d frames\n", (long)syscall(SYS_gettid), n);
    if (write(g_fd, line, (size_t)len) < 0) return;
    for (int i = 0; i < n; i++) {
        Dl_info di; memset(&di, 0, sizeof di);
        (void)dladdr(pc[i], &di);
        len = snprintf(line, sizeof line, "  #%d %p %s +0x%lx [%s base %p]\n", i, pc[i], di.dli_sname ? di.dli_sname : "?",
                       di.dli_saddr ? (unsigned long)((char*)pc[i] - (char*)di.dli_saddr) : 0ul, di.dli_fname ? di.dli_fname : "?", di.dli_fbase);
        if (write(g_fd, line, (size_t)len) < 0) return;
    }
}

__attribute__((constructor)) static void probe_init(void)
{
    const char* path = getenv("RCGPU_HANG_DUMP");
    if (!path) return;
    g_fd = open(path, O_WRONLY | O_CREAT | O_APPEND, 0644);
    void* warm[4]; (void)backtrace(warm, 4);            /* loads libgcc's unwinder now, not inside the handler */
    struct sigaction sa; memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_usr2; sa.sa_flags = SA_RESTART;
    sigaction(SIGUSR2, &sa, NULL);
}
