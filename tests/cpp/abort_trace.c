/* abort_trace.c -- test infrastructure (loaded by tests/conftest.py, never by the product): when the process receives
 * SIGABRT -- a runtime library calling abort() under it, which Python's faulthandler can only report as "Fatal Python
 * error: Aborted" with the PYTHON stacks -- write the NATIVE stack of the aborting thread to stderr first, then hand
 * the signal to whoever handled it before (faulthandler, then the default action).  HISTORY.md section 10: two silent
 * aborts in some seventy full GPU runs had no message at all; the next one will at least say where it came from.
 * build: gcc -O1 -g -shared -fPIC -o abort_trace.so abort_trace.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static struct sigaction prev_action;
static int out_fd = 2;  /* a duplicate of stderr taken at install time: pytest redirects fd 2 while a test runs */

static void on_abort(int sig, siginfo_t *info, void *uctx)
{
    (void)sig; (void)info; (void)uctx;
    static const char msg[] = "\n[abort_trace] SIGABRT -- native stack of the aborting thread:\n";
    void *frames[96];
    ssize_t w = write(out_fd, msg, sizeof msg - 1);
    (void)w;
    backtrace_symbols_fd(frames, backtrace(frames, 96), out_fd);
    sigaction(SIGABRT, &prev_action, NULL);
    raise(SIGABRT);
}

int abort_trace_install(void)
{
    void *warm[4];
    struct sigaction sa;
    int d = dup(2);
    if (d >= 0) out_fd = d;
    (void)backtrace(warm, 4);  /* the first call loads libgcc: do it here, not inside the handler */
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_abort;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    return sigaction(SIGABRT, &sa, &prev_action);
}
