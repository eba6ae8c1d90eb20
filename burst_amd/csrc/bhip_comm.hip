// burst_amd/csrc/bhip_comm.hip -- the one exchange step of the multi-GPU path (no reference counterpart: the reference is one
// process with shared memory): unique queries are sharded across the GPUs of a node, every device aligns
// its shard independently against its own copy of the database, and the hit records travel to rank 0 in ONE variable-length
// gather over xGMI: ncclAllGather of the record counts + grouped ncclSend / ncclRecv of the 20-byte BhipHit records (RCCL has
// no gatherv).  One host thread per device (ncclCommInitAll: all ranks in this process); every thread calls
// bhip_comm_gather_hits with its own rank.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <vector>
#include "burst_hip.h"

extern "C" const char *bhip_last_error(void);
int bhip_fail_msg(int code, const char *fmt, ...);      // bhip_api.hip: sets the calling thread's error text

struct Comm {
	int n = 0;
	std::vector<int> dev;
	std::vector<ncclComm_t> comm;
	std::vector<hipStream_t> stream;
	std::vector<void *> d_send, d_counts;       // per rank: send buffer (grow-only) and the gathered counts
	std::vector<size_t> send_cap;
	void *d_recv = nullptr; size_t recv_cap = 0; // rank 0
};
#define CCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return bhip_fail_msg(BHIP_E_DEVICE, "%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); } while (0)
#define NCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return bhip_fail_msg(BHIP_E_DEVICE, "%s:%d %s: %s", __FILE__, __LINE__, #x, ncclGetErrorString(r_)); } while (0)

extern "C" int bhip_comm_create(int n_ranks, const int *devices, void **comm_out) {
	if (!comm_out || n_ranks < 1 || !devices) return bhip_fail_msg(BHIP_E_ARG, "bad communicator arguments");
	*comm_out = nullptr;
	Comm *C = new Comm();
	C->n = n_ranks; C->dev.assign(devices, devices + n_ranks); C->comm.resize(n_ranks); C->stream.assign(n_ranks, nullptr);
	C->d_send.assign(n_ranks, nullptr); C->d_counts.assign(n_ranks, nullptr); C->send_cap.assign(n_ranks, 0);
	ncclResult_t r = ncclCommInitAll(C->comm.data(), n_ranks, C->dev.data());
	if (r != ncclSuccess) { delete C; return bhip_fail_msg(BHIP_E_DEVICE, "ncclCommInitAll(%d ranks): %s", n_ranks, ncclGetErrorString(r)); }
	for (int k = 0; k < n_ranks; ++k) {
		CCHK(hipSetDevice(devices[k]));
		CCHK(hipStreamCreateWithFlags(&C->stream[k], hipStreamNonBlocking));
		CCHK(hipMalloc(&C->d_counts[k], sizeof(unsigned long long) * (size_t)(n_ranks + 1)));
	}
	*comm_out = C;
	return BHIP_OK;
}

extern "C" void bhip_comm_destroy(void *comm) {
	Comm *C = (Comm *)comm;
	if (!C) return;
	for (int k = 0; k < C->n; ++k) {
		(void)hipSetDevice(C->dev[k]);
		if (C->stream[k]) { (void)hipStreamSynchronize(C->stream[k]); (void)hipStreamDestroy(C->stream[k]); }
		if (C->d_send[k]) (void)hipFree(C->d_send[k]);
		if (C->d_counts[k]) (void)hipFree(C->d_counts[k]);
		if (k == 0 && C->d_recv) (void)hipFree(C->d_recv);
		(void)ncclCommDestroy(C->comm[k]);
	}
	delete C;
}

// Called by the host thread of every rank (all n_ranks calls must be in flight together).  hits / n: the rank's records in
// host memory.  Rank 0 receives every rank's records, rank order, into out (capacity cap records); counts[n_ranks] gets the
// per-rank numbers on every rank.  BHIP_E_CAPACITY when out is too small (*n_total is what is needed; nothing is copied).
extern "C" int bhip_comm_gather_hits(void *comm, int rank, const BhipHit *hits, uint64_t n, BhipHit *out, uint64_t cap, uint64_t *n_total, uint64_t *counts) {
	Comm *C = (Comm *)comm;
	if (!C || rank < 0 || rank >= C->n || !n_total) return bhip_fail_msg(BHIP_E_ARG, "bad gather arguments");
	CCHK(hipSetDevice(C->dev[rank]));
	hipStream_t st = C->stream[rank];
	const size_t bytes = (size_t)n * sizeof(BhipHit);
	if (bytes > C->send_cap[rank]) {
		if (C->d_send[rank]) CCHK(hipFree(C->d_send[rank]));
		C->send_cap[rank] = bytes + bytes / 8 + 4096;
		CCHK(hipMalloc(&C->d_send[rank], C->send_cap[rank]));
	}
	if (bytes) CCHK(hipMemcpyAsync(C->d_send[rank], hits, bytes, hipMemcpyHostToDevice, st));
	// 1. everybody learns everybody's count
	unsigned long long *dc = (unsigned long long *)C->d_counts[rank];
	unsigned long long mine = n;
	CCHK(hipMemcpyAsync(dc + C->n, &mine, sizeof mine, hipMemcpyHostToDevice, st));
	NCHK(ncclAllGather(dc + C->n, dc, 1, ncclUint64, C->comm[rank], st));
	std::vector<unsigned long long> hc((size_t)C->n);
	CCHK(hipMemcpyAsync(hc.data(), dc, sizeof(unsigned long long) * (size_t)C->n, hipMemcpyDeviceToHost, st));
	CCHK(hipStreamSynchronize(st));
	uint64_t total = 0;
	for (int k = 0; k < C->n; ++k) { if (counts) counts[k] = hc[k]; total += hc[k]; }
	*n_total = total;
	const bool fits = total <= cap || rank != 0;
	// (the capacity decision is rank 0's alone, but every rank must take the same path through the collectives: rank 0 receives
	// into its device buffer in any case and only the copy to the caller's memory is skipped when it does not fit)
	if (rank == 0) {
		const size_t need = (size_t)total * sizeof(BhipHit);
		if (need > C->recv_cap) {
			if (C->d_recv) CCHK(hipFree(C->d_recv));
			C->recv_cap = need + need / 8 + 4096;
			CCHK(hipMalloc(&C->d_recv, C->recv_cap));
		}
	}
	// 2. the records: grouped point-to-point, 7 peers -> 7 different xGMI links into rank 0
	NCHK(ncclGroupStart());
	if (rank == 0) {
		size_t off = 0;
		for (int k = 0; k < C->n; ++k) {
			const size_t b = (size_t)hc[k] * sizeof(BhipHit);
			if (k == 0) { if (b) CCHK(hipMemcpyAsync((char *)C->d_recv + off, C->d_send[0], b, hipMemcpyDeviceToDevice, st)); }
			else if (b) NCHK(ncclRecv((char *)C->d_recv + off, b, ncclUint8, k, C->comm[0], st));
			off += b;
		}
	} else if (bytes) NCHK(ncclSend(C->d_send[rank], bytes, ncclUint8, 0, C->comm[rank], st));
	NCHK(ncclGroupEnd());
	if (rank == 0 && fits && total) {
		if (!out) return bhip_fail_msg(BHIP_E_ARG, "null output");
		CCHK(hipMemcpyAsync(out, C->d_recv, (size_t)total * sizeof(BhipHit), hipMemcpyDeviceToHost, st));
	}
	CCHK(hipStreamSynchronize(st));
	if (!fits) return bhip_fail_msg(BHIP_E_CAPACITY, "record buffer holds %llu records, %llu needed", (unsigned long long)cap, (unsigned long long)total);
	return BHIP_OK;
}
