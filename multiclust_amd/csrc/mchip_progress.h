/*
 * mchip_progress.h -- where every host thread that is inside libmulticlust_hip.so stands (private; the C-ABI view is
 * mchip_progress_report / mchip_progress_note in include/multiclust_hip.h).
 *
 * Every C-ABI entry point names itself on entry (MCHIP_ENTRY) and every HIP runtime call that can block -- stream
 * synchronisation, copies, allocation, graph capture and launch, stream and event teardown -- names itself while it runs
 * (HIPCHK / MCHIP_WAIT).  Each of those moments bumps one process-wide event counter.  A watchdog (host side:
 * mc_watchdog_start, MC_WATCHDOG_S) that sees the counter stand still knows the process makes no progress and can say in which
 * call of which thread: the evidence that was missing when a launch of the command line did not return.
 * Cost per event: one relaxed atomic add, two stores and a vDSO clock read.
 */
#ifndef MCHIP_PROGRESS_H
#define MCHIP_PROGRESS_H

#include <atomic>
#include <time.h>

#define MCHIP_PROGRESS_SLOTS 128

struct mchip_progress_slot {
	std::atomic<const char *> entry;	/* C-ABI function the thread is in (static string), or nullptr */
	std::atomic<const char *> wait;		/* runtime call it is waiting in (static string), or nullptr */
	std::atomic<const char *> note;		/* last phase the host named (mchip_progress_note) */
	std::atomic<long long> since_ns;	/* CLOCK_MONOTONIC of the last change */
	std::atomic<long> tid;
};

extern mchip_progress_slot mchip_progress_slots[MCHIP_PROGRESS_SLOTS];
extern std::atomic<unsigned long long> mchip_progress_events;
mchip_progress_slot *mchip_progress_my_slot(void);

static inline long long mchip_progress_now(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return (long long)t.tv_sec * 1000000000LL + t.tv_nsec;
}

struct mchip_entry_guard {
	mchip_progress_slot *s;
	const char *prev;
	explicit mchip_entry_guard(const char *name) : s(mchip_progress_my_slot())
	{
		prev = s->entry.load(std::memory_order_relaxed);
		s->entry.store(name, std::memory_order_relaxed);
		s->since_ns.store(mchip_progress_now(), std::memory_order_relaxed);
		mchip_progress_events.fetch_add(1, std::memory_order_relaxed);
	}
	~mchip_entry_guard()
	{
		s->entry.store(prev, std::memory_order_relaxed);
		s->since_ns.store(mchip_progress_now(), std::memory_order_relaxed);
		mchip_progress_events.fetch_add(1, std::memory_order_relaxed);
	}
};

struct mchip_wait_guard {
	mchip_progress_slot *s;
	explicit mchip_wait_guard(const char *what) : s(mchip_progress_my_slot())
	{
		s->wait.store(what, std::memory_order_relaxed);
		s->since_ns.store(mchip_progress_now(), std::memory_order_relaxed);
		mchip_progress_events.fetch_add(1, std::memory_order_relaxed);
	}
	~mchip_wait_guard()
	{
		s->wait.store(nullptr, std::memory_order_relaxed);
		s->since_ns.store(mchip_progress_now(), std::memory_order_relaxed);
		mchip_progress_events.fetch_add(1, std::memory_order_relaxed);
	}
};

#define MCHIP_STR2(x) #x
#define MCHIP_STR(x) MCHIP_STR2(x)
#define MCHIP_ENTRY() mchip_entry_guard mchip_entry_guard_(__func__)
/* evaluates a runtime call with its text and source line on record while it runs */
#define MCHIP_WAIT(call) ([&]() { mchip_wait_guard w_(#call " (" __FILE__ ":" MCHIP_STR(__LINE__) ")"); return (call); }())

#endif
