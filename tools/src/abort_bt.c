/* Native stack on SIGABRT / SIGSEGV / SIGBUS for test runs on the GPU box (no gdb there).  Loaded by tests/conftest.py
 * (ctypes) or with LD_PRELOAD; the previous handler (Python's faulthandler) still runs afterwards.
 *   gcc -shared -fPIC -o /tmp/lgmi_abort_bt.so tools/src/abort_bt.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
static struct sigaction old_abrt, old_segv, old_bus;
static void on_sig(int sig, siginfo_t* si, void* uc) {
    void* bt[64];
    const int n = backtrace(bt, 64);
    const char* m = sig == SIGABRT ? "\n[abort_bt] SIGABRT, native stack:\n" : (sig == SIGSEGV ? "\n[abort_bt] SIGSEGV, native stack:\n" : "\n[abort_bt] SIGBUS, native stack:\n");
    (void)!write(2, m, strlen(m));
    backtrace_symbols_fd(bt, n, 2);
    struct sigaction* old = sig == SIGABRT ? &old_abrt : (sig == SIGSEGV ? &old_segv : &old_bus);
    if ((old->sa_flags & SA_SIGINFO) && old->sa_sigaction) { old->sa_sigaction(sig, si, uc); return; }
    if (!(old->sa_flags & SA_SIGINFO) && old->sa_handler != SIG_DFL && old->sa_handler != SIG_IGN) { old->sa_handler(sig); return; }
    signal(sig, SIG_DFL);
    raise(sig);
}
void lgmi_abort_bt_install(void) {
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_sig;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER;
    sigaction(SIGABRT, &sa, &old_abrt);
    sigaction(SIGSEGV, &sa, &old_segv);
    sigaction(SIGBUS, &sa, &old_bus);
}
__attribute__((constructor)) static void init(void) { if (getenv("LD_PRELOAD")) lgmi_abort_bt_install(); }
