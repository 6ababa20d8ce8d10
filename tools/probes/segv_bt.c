/* Native backtrace on SIGSEGV / SIGBUS / SIGABRT (diagnosis tool, not product code).
 *
 *   gcc -O1 -g -shared -fPIC -o tools/probes/libsegv_bt.so tools/probes/segv_bt.c
 *   RF_SEGV_BT=1 python -m pytest ...      (tests/conftest.py loads it after pytest's own faulthandler)
 *
 * Prints the frames as "module(+offset)" plus the executable mappings of the modules involved, so the offsets can be
 * looked up in the (stripped) runtime libraries with llvm-objdump afterwards. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>

static int out_fd = 2;

static void dump_maps(void) {
    int fd = open("/proc/self/maps", O_RDONLY);
    if (fd < 0) return;
    static char buf[1 << 20];
    ssize_t n, tot = 0;
    while ((n = read(fd, buf + tot, sizeof(buf) - 1 - tot)) > 0) tot += n;
    close(fd);
    buf[tot] = 0;
    const char* hdr = "---- executable mappings ----\n";
    write(out_fd, hdr, strlen(hdr));
    char* line = buf;
    while (line && *line) {
        char* nl = strchr(line, '\n');
        if (nl) *nl = 0;
        if (strstr(line, " r-xp ") && (strstr(line, "hip") || strstr(line, "hsa") || strstr(line, "torch") || strstr(line, "c10") ||
                                        strstr(line, "librf") || strstr(line, "libc.so"))) {
            write(out_fd, line, strlen(line));
            write(out_fd, "\n", 1);
        }
        line = nl ? nl + 1 : NULL;
    }
}

static void handler(int sig, siginfo_t* info, void* uctx) {
    (void)uctx;
    char msg[128];
    int n = snprintf(msg, sizeof msg, "\n==== segv_bt: signal %d, fault address %p ====\n", sig, info ? info->si_addr : NULL);
    write(out_fd, msg, n);
    void* frames[96];
    int depth = backtrace(frames, 96);
    backtrace_symbols_fd(frames, depth, out_fd);
    dump_maps();
    signal(sig, SIG_DFL);
    raise(sig);
}

/* fd < 0: stderr.  (pytest captures fd 2 into a temporary file that dies with the process.) */
int rf_segv_bt_install(int fd) {
    if (fd >= 0) out_fd = fd;
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = handler;
    sa.sa_flags = SA_SIGINFO | SA_RESETHAND | SA_ONSTACK;
    static char stack[1 << 16];
    stack_t ss = {.ss_sp = stack, .ss_size = sizeof stack, .ss_flags = 0};
    sigaltstack(&ss, NULL);
    void* warm[4];
    backtrace(warm, 4); /* loads libgcc now, not inside the handler */
    int rc = 0;
    rc |= sigaction(SIGSEGV, &sa, NULL);
    rc |= sigaction(SIGBUS, &sa, NULL);
    return rc;
}
