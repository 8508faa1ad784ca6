// Diagnosis helper (NOT part of the product): dump the C stack of the main thread when asked to.
//   gcc -O1 -g -shared -fPIC -o stallwatch.so stallwatch.c -ldl -lpthread
// sw_install() on the main thread installs a SIGUSR2 handler that writes backtrace_symbols_fd() of whatever the main
// thread is executing (or blocked in) to stderr; sw_poke() from a watchdog thread delivers the signal to it.
#define _GNU_SOURCE
#include <execinfo.h>
#include <pthread.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

static pthread_t g_main;
static volatile int g_n = 0;

static void handler(int sig) {
  (void)sig;
  void* frames[48];
  int n = backtrace(frames, 48);
  char head[96];
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  int len = snprintf(head, sizeof head, "[stallwatch] dump %d at %ld.%06ld: %d frames\n", ++g_n, (long)ts.tv_sec, ts.tv_nsec / 1000, n);
  (void)!write(2, head, len);
  backtrace_symbols_fd(frames, n, 2);
}

int sw_install(void) {
  g_main = pthread_self();
  void* warm[4];
  backtrace(warm, 4);                       // loads libgcc's unwinder outside the handler
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_handler = handler;
  sa.sa_flags = SA_RESTART;
  return sigaction(SIGUSR2, &sa, NULL);
}

int sw_poke(void) { return pthread_kill(g_main, SIGUSR2); }
