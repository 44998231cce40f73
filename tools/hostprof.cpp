// Wall-clock sampling profiler for the host side of an engine run (developer aid; nothing in the product links it):
//   g++ -O2 -shared -fPIC tools/hostprof.cpp -o tools/libhostprof.so -lrt
//   HOSTPROF_OUT=prof.txt LD_PRELOAD=tools/libhostprof.so dmrg.x_amd/dmrgx-square-lattice ...
//   python3 tools/hostprof_report.py prof.txt
// A POSIX timer signals the MAIN thread every 200 us; the handler stores a backtrace (return addresses only).  At exit the samples and
// /proc/self/maps go to HOSTPROF_OUT; the report script turns addresses into symbols (module offset -> nm table).
#include <execinfo.h>
#include <signal.h>
#include <time.h>
#include <unistd.h>
#include <sys/syscall.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {
constexpr int DEPTH = 24;
constexpr size_t CAP = 400000;
void** g_buf = nullptr;
int* g_len = nullptr;
volatile size_t g_n = 0;
timer_t g_timer;
volatile int g_on = 0;

void on_tick(int, siginfo_t*, void*)
{
    if (!g_on || g_n >= CAP) return;
    const size_t i = g_n;
    g_len[i] = backtrace(g_buf + i * DEPTH, DEPTH);
    g_n = i + 1;
}

__attribute__((constructor)) void start()
{
    if (!getenv("HOSTPROF_OUT")) return;
    g_buf = (void**)calloc(CAP * DEPTH, sizeof(void*));
    g_len = (int*)calloc(CAP, sizeof(int));
    void* warm[4];
    backtrace(warm, 4);                                   // (loads libgcc's unwinder outside the handler)
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = on_tick;
    sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigaction(SIGRTMIN + 3, &sa, nullptr);
    struct sigevent ev;
    memset(&ev, 0, sizeof(ev));
    ev.sigev_notify = SIGEV_THREAD_ID;
    ev.sigev_signo = SIGRTMIN + 3;
    ev._sigev_un._tid = (pid_t)syscall(SYS_gettid);
    if (timer_create(CLOCK_MONOTONIC, &ev, &g_timer) != 0) { perror("hostprof: timer_create"); return; }
    const long us = getenv("HOSTPROF_US") ? atol(getenv("HOSTPROF_US")) : 200;
    struct itimerspec its;
    its.it_interval.tv_sec = 0; its.it_interval.tv_nsec = us * 1000;
    // first tick after the runtime has opened the device: a signal during hipInit's ioctls makes it report "no device"
    its.it_value.tv_sec = getenv("HOSTPROF_DELAY_S") ? atol(getenv("HOSTPROF_DELAY_S")) : 5; its.it_value.tv_nsec = 0;
    g_on = 1;
    timer_settime(g_timer, 0, &its, nullptr);
}

__attribute__((destructor)) void stop()
{
    if (!g_on) return;
    if (g_n == 0) { g_on = 0; timer_delete(g_timer); return; }      // (a launcher process -- timeout, env -- that inherited the preload: it must not overwrite the engine's samples)
    g_on = 0;
    timer_delete(g_timer);
    FILE* f = fopen(getenv("HOSTPROF_OUT"), "w");
    if (!f) return;
    FILE* m = fopen("/proc/self/maps", "r");
    char line[1024];
    while (m && fgets(line, sizeof(line), m)) if (strstr(line, " r-xp ") || strstr(line, " r--p ")) fprintf(f, "M %s", line);
    if (m) fclose(m);
    for (size_t i = 0; i < g_n; ++i) {
        fprintf(f, "S");
        for (int k = 0; k < g_len[i]; ++k) fprintf(f, " %p", g_buf[i * DEPTH + k]);
        fprintf(f, "\n");
    }
    fclose(f);
}
}  // namespace
