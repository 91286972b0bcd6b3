/*
 * mchip_comm.hip -- the path's one exchange step for a single process that drives several GPUs: an RCCL all-reduce
 * of the per-unit result table.  RCCL is loaded lazily with dlopen, so that
 * libmulticlust_hip.so has no link-time dependency on it (bench.py's processes already carry the RCCL that ships
 * with PyTorch and exchange through torch.distributed instead).
 */
#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "multiclust_hip.h"
#include "mchip_progress.h"

struct mchip_comm {
	int n;
	std::vector<int> devices;
	std::vector<ncclComm_t> comms;
	std::vector<hipStream_t> streams;
	std::vector<double *> dbuf;
	size_t cap;		/* doubles per device buffer */
	void *dl;
	ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *);
	ncclResult_t (*CommDestroy)(ncclComm_t);
	ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
	ncclResult_t (*GroupStart)(void);
	ncclResult_t (*GroupEnd)(void);
	const char *(*GetErrorString)(ncclResult_t);
	ncclResult_t (*GetVersion)(int *);	/* optional */
	unsigned long long n_reductions;	/* all-reduces completed on this communicator */
	char err[512];
};

/* The RCCL that belongs to the HIP runtime this library is bound to.  A process can hold two ROCm stacks: PyTorch ships its own
 * libamdhip64.so, librccl.so (no SONAME), librocm_smi64.so ... next to ROCm's.  When PyTorch is loaded first (bench.py), this
 * library's "libamdhip64.so.7" resolves to PyTorch's runtime, and the RCCL to use is PyTorch's -- ROCm's librccl.so.1 beside it
 * failed in ncclCommInitAll ("unhandled cuda error").  Without PyTorch (the command line) it is ROCm's.  So: the librccl in the
 * directory of the libamdhip64 that hipGetDeviceCount resolves to; MCHIP_RCCL_PATH overrides; the loader's search path last. */
static void *open_rccl(void)
{
	if (const char *path = getenv("MCHIP_RCCL_PATH")) {
		void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
		if (h) return h;
		fprintf(stderr, "mchip_comm_create: MCHIP_RCCL_PATH=%s: %s\n", path, dlerror());
	}
	Dl_info info;
	if (dladdr((void *)&hipGetDeviceCount, &info) && info.dli_fname) {
		std::string dir(info.dli_fname);
		const size_t slash = dir.rfind('/');
		if (slash != std::string::npos) {
			dir.resize(slash + 1);
			for (const char *n : { "librccl.so.1", "librccl.so" }) {
				void *h = dlopen((dir + n).c_str(), RTLD_NOW | RTLD_LOCAL);
				if (h) return h;
			}
		}
	}
	void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
	return h ? h : dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
}

static int cfail(mchip_comm *c, int code, const char *what, const char *detail)
{
	if (c) snprintf(c->err, sizeof c->err, "%s: %s", what, detail ? detail : "");
	return code;
}

extern "C" {

const char *mchip_comm_last_error(const mchip_comm *comm) { return comm ? comm->err : "null communicator"; }

int mchip_comm_create(mchip_comm **out, int n_devices, const int *devices)
{
	MCHIP_ENTRY();
	if (!out || n_devices < 1 || !devices) return MCHIP_ERR_INVALID;
	*out = nullptr;
	int have = 0;
	if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return MCHIP_ERR_NO_DEVICE;
	for (int d = 0; d < n_devices; d++)
		if (devices[d] < 0 || devices[d] >= have) return MCHIP_ERR_INVALID;
	mchip_comm *c = new mchip_comm();
	c->n = n_devices;
	c->devices.assign(devices, devices + n_devices);
	c->cap = 0;
	c->err[0] = 0;
	c->dl = open_rccl();
	if (!c->dl) {
		fprintf(stderr, "mchip_comm_create: cannot load RCCL: %s\n", dlerror());
		delete c;
		return MCHIP_ERR_UNSUPPORTED;
	}
	*(void **)&c->CommInitAll = dlsym(c->dl, "ncclCommInitAll");
	*(void **)&c->CommDestroy = dlsym(c->dl, "ncclCommDestroy");
	*(void **)&c->AllReduce = dlsym(c->dl, "ncclAllReduce");
	*(void **)&c->GroupStart = dlsym(c->dl, "ncclGroupStart");
	*(void **)&c->GroupEnd = dlsym(c->dl, "ncclGroupEnd");
	*(void **)&c->GetErrorString = dlsym(c->dl, "ncclGetErrorString");
	*(void **)&c->GetVersion = dlsym(c->dl, "ncclGetVersion");
	c->n_reductions = 0;
	if (!c->CommInitAll || !c->CommDestroy || !c->AllReduce || !c->GroupStart || !c->GroupEnd || !c->GetErrorString) {
		fprintf(stderr, "mchip_comm_create: RCCL symbols missing\n");
		dlclose(c->dl);
		delete c;
		return MCHIP_ERR_UNSUPPORTED;
	}
	c->comms.resize(n_devices);
	/* RCCL prints a version banner on stdout at first initialisation; stdout belongs to the caller's result lines */
	fflush(stdout);
	const int saved = dup(1), devnull = open("/dev/null", O_WRONLY);
	if (saved >= 0 && devnull >= 0) dup2(devnull, 1);
	ncclResult_t r = MCHIP_WAIT(c->CommInitAll(c->comms.data(), n_devices, devices));
	fflush(stdout);
	if (saved >= 0) { dup2(saved, 1); close(saved); }
	if (devnull >= 0) close(devnull);
	if (r != ncclSuccess) {
		fprintf(stderr, "mchip_comm_create: ncclCommInitAll failed: %s\n", c->GetErrorString(r));
		dlclose(c->dl);
		delete c;
		return MCHIP_ERR_HIP;
	}
	c->dbuf.assign(n_devices, nullptr);
	c->streams.reserve(n_devices);
	for (int d = 0; d < n_devices; d++) {
		hipStream_t st = nullptr;
		if (hipSetDevice(devices[d]) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
			mchip_comm_destroy(c);	/* the communicators are initialised by now: destroy them, the streams made so far, RCCL's handle */
			return MCHIP_ERR_HIP;
		}
		c->streams.push_back(st);
	}
	*out = c;
	return MCHIP_OK;
}

int mchip_comm_all_reduce(mchip_comm *c, double *const *host_bufs, int count, int op)
{
	MCHIP_ENTRY();
	if (!c || !host_bufs || count <= 0 || (op != 0 && op != 1)) return MCHIP_ERR_INVALID;
	if ((size_t)count > c->cap) {
		for (int d = 0; d < c->n; d++) {
			if (hipSetDevice(c->devices[d]) != hipSuccess) return cfail(c, MCHIP_ERR_HIP, "hipSetDevice", "");
			if (c->dbuf[d]) (void)hipFree(c->dbuf[d]);
			if (hipMalloc((void **)&c->dbuf[d], sizeof(double) * (size_t)count) != hipSuccess) return cfail(c, MCHIP_ERR_ALLOC, "hipMalloc", "");
		}
		c->cap = (size_t)count;
	}
	for (int d = 0; d < c->n; d++) {
		if (hipSetDevice(c->devices[d]) != hipSuccess ||
		    hipMemcpyAsync(c->dbuf[d], host_bufs[d], sizeof(double) * (size_t)count, hipMemcpyHostToDevice, c->streams[d]) != hipSuccess)
			return cfail(c, MCHIP_ERR_HIP, "upload", "");
	}
	ncclResult_t r = c->GroupStart();
	for (int d = 0; d < c->n && r == ncclSuccess; d++)
		r = c->AllReduce(c->dbuf[d], c->dbuf[d], (size_t)count, ncclFloat64, op ? ncclMax : ncclSum, c->comms[d], c->streams[d]);
	if (r == ncclSuccess) r = MCHIP_WAIT(c->GroupEnd());
	if (r != ncclSuccess) return cfail(c, MCHIP_ERR_HIP, "ncclAllReduce", c->GetErrorString(r));
	for (int d = 0; d < c->n; d++) {
		if (hipSetDevice(c->devices[d]) != hipSuccess ||
		    hipMemcpyAsync(host_bufs[d], c->dbuf[d], sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, c->streams[d]) != hipSuccess ||
		    MCHIP_WAIT(hipStreamSynchronize(c->streams[d])) != hipSuccess)
			return cfail(c, MCHIP_ERR_HIP, "download", "");
	}
	c->n_reductions++;
	return MCHIP_OK;
}

int mchip_comm_info(const mchip_comm *c, int *n_devices, int *rccl_version, unsigned long long *n_reductions)
{
	if (!c) return MCHIP_ERR_INVALID;
	if (n_devices) *n_devices = c->n;
	if (rccl_version) {
		int v = 0;
		if (!c->GetVersion || c->GetVersion(&v) != ncclSuccess) v = 0;
		*rccl_version = v;
	}
	if (n_reductions) *n_reductions = c->n_reductions;
	return MCHIP_OK;
}

int mchip_comm_destroy(mchip_comm *c)
{
	MCHIP_ENTRY();
	if (!c) return MCHIP_OK;
	for (int d = 0; d < c->n; d++) {
		(void)hipSetDevice(c->devices[d]);
		if (c->dbuf[d]) (void)MCHIP_WAIT(hipFree(c->dbuf[d]));
		if (d < (int)c->streams.size()) (void)MCHIP_WAIT(hipStreamDestroy(c->streams[d]));
		if (d < (int)c->comms.size() && c->comms[d]) (void)MCHIP_WAIT(c->CommDestroy(c->comms[d]));
	}
	if (c->dl) dlclose(c->dl);
	delete c;
	return MCHIP_OK;
}

}  /* extern "C" */
