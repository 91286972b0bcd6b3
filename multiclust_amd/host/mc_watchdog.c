/*
 * mc_watchdog.c -- opt-in "where does it stand" timer of the command line (MC_WATCHDOG_S=<seconds>).
 *
 * The reference is single-threaded C that waits on nothing; this build waits on the HIP runtime behind every entry point of
 * include/multiclust_hip.h.  The library keeps a record of where each host thread stands and counts an event whenever that
 * changes (mchip_progress_report); this thread polls the count and, when it has not moved for the given number of seconds,
 * writes the record, the kernel's view of every thread of the process (/proc/self/task: name, state, wait channel) and leaves
 * with _exit(3) -- a fresh exit, no atexit handlers, no runtime teardown, nothing launched again.  A run that is merely long
 * keeps counting events (every C-ABI call, every runtime call inside it, every phase the host names) and is left alone.
 */
#include "mc_host.h"

#include <dirent.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static double wd_seconds;

static double wd_now(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* first line of a small /proc file into buf (empty when it cannot be read: wchan and stack need privileges on some hosts) */
static void first_line(const char *path, char *buf, size_t len)
{
	buf[0] = 0;
	const int fd = open(path, O_RDONLY);
	if (fd < 0) return;
	const ssize_t n = read(fd, buf, len - 1);
	close(fd);
	if (n <= 0) { buf[0] = 0; return; }
	buf[n] = 0;
	char *nl = strchr(buf, '\n');
	if (nl) *nl = 0;
}

static void state_of(const char *task, char *buf, size_t len)
{
	char path[320], text[1024];
	buf[0] = 0;
	snprintf(path, sizeof path, "/proc/self/task/%s/status", task);
	const int fd = open(path, O_RDONLY);
	if (fd < 0) return;
	const ssize_t n = read(fd, text, sizeof text - 1);
	close(fd);
	if (n <= 0) return;
	text[n] = 0;
	const char *s = strstr(text, "State:");
	if (!s) return;
	s += 6;
	while (*s == ' ' || *s == '\t') s++;
	size_t x = 0;
	while (s[x] && s[x] != '\n' && x + 1 < len) { buf[x] = s[x]; x++; }
	buf[x] = 0;
}

void mc_watchdog_report(FILE *fp, double quiet_seconds)
{
	char rep[4096];
	unsigned long long events = 0;
	mchip_progress_report(rep, (int)sizeof rep, &events);
	fprintf(fp, "WATCHDOG [mc_watchdog.c]: no progress for %.1f s (MC_WATCHDOG_S=%g) after %llu library events; where every thread stands:\n%s",
		quiet_seconds, wd_seconds, events, rep[0] ? rep : "(no thread has entered the library)\n");
	DIR *d = opendir("/proc/self/task");
	if (d) {
		struct dirent *e;
		while ((e = readdir(d))) {
			char path[320], comm[64], wchan[128], state[64];
			if (e->d_name[0] == '.') continue;
			snprintf(path, sizeof path, "/proc/self/task/%s/comm", e->d_name);
			first_line(path, comm, sizeof comm);
			snprintf(path, sizeof path, "/proc/self/task/%s/wchan", e->d_name);
			first_line(path, wchan, sizeof wchan);
			state_of(e->d_name, state, sizeof state);
			fprintf(fp, "  task %s (%s): state %s, wait channel %s\n", e->d_name, comm, state[0] ? state : "?", wchan[0] ? wchan : "?");
		}
		closedir(d);
	}
	fflush(fp);
}

static void *wd_main(void *arg)
{
	unsigned long long last = ~0ull, now_events = 0;
	double t_last = wd_now();
	(void)arg;
	for (;;) {
		struct timespec nap = { 0, 200 * 1000 * 1000 };
		nanosleep(&nap, NULL);
		mchip_progress_report(NULL, 0, &now_events);
		const double t = wd_now();
		if (now_events != last) { last = now_events; t_last = t; continue; }
		if (t - t_last < wd_seconds) continue;
		fflush(stdout);		/* what the run printed so far (line-buffered anyway when MC_WATCHDOG_S is set) */
		mc_watchdog_report(stderr, t - t_last);
		_exit(3);
	}
	return NULL;
}

int mc_watchdog_start(double seconds)
{
	static int started;
	pthread_t th;
	pthread_attr_t at;
	if (started || !(seconds > 0)) return 0;
	wd_seconds = seconds;
	pthread_attr_init(&at);
	pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
	const int rc = pthread_create(&th, &at, wd_main, NULL);
	pthread_attr_destroy(&at);
	if (rc) { fprintf(stderr, "WARNING [mc_watchdog.c]: cannot start the watchdog thread\n"); return 1; }
	started = 1;
	return 0;
}

int mc_watchdog_from_env(void)
{
	const char *e = getenv("MC_WATCHDOG_S");
	if (!e || !*e) return 0;
	const double s = atof(e);
	if (!(s > 0)) { fprintf(stderr, "WARNING [mc_watchdog.c]: MC_WATCHDOG_S='%s' is not a positive number of seconds; no watchdog\n", e); return 0; }
	return mc_watchdog_start(s);
}
