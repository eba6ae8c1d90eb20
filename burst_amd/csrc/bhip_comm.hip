// burst_amd/csrc/bhip_comm.hip -- the exchange steps of the multi-GPU path (no reference counterpart: the reference is one
// process with shared memory).
//   Query-sharded (the database replicated): every device aligns its range of unique queries independently and the hit records
//   travel to rank 0 in ONE variable-length gather over xGMI: ncclAllGather of the record counts + grouped ncclSend / ncclRecv of
//   the 20-byte BhipHit records (RCCL has no gatherv; 7 peers -> 7 different links into rank 0).
//   Database-sharded (databases beyond one device): every device aligns ALL queries against its range of clumps; the hits of a
//   query are the references at its GLOBAL minimum edit distance (burst.c:4217-4277), so the ranks combine one byte per unique
//   query with ncclAllReduce(MIN) before the same gather.
// The ranks are either the threads of one process (bhip_comm_create: ncclCommInitAll) or one process each (bhip_comm_unique_id
// on one rank, the 128 bytes handed to the others by whatever launched them, bhip_comm_create_rank: ncclCommInitRank); every
// rank's thread calls the collectives with its own rank.
// Failures: everything a rank can fail on locally (allocations, copies) happens BEFORE its first collective of a call and is
// exchanged with the counts, so that all ranks leave together with an error instead of one rank returning and the others
// waiting for it; nothing returns between ncclGroupStart and ncclGroupEnd; a failed collective aborts the communicator.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <pthread.h>
#include <algorithm>
#include <vector>
#include "burst_hip.h"
#include <atomic>
std::atomic<int> &bhip_vmm_peer_access_flag();

int bhip_fail_msg(int code, const char *fmt, ...);      // bhip_init.hip: sets the calling thread's error text

struct Peer {                                      // one rank that lives in this process
	int rank = 0, dev = 0;
	ncclComm_t comm = nullptr;
	hipStream_t stream = nullptr;
	void *d_send = nullptr; size_t send_cap = 0;   // records out / bytes of the reduction
	void *d_counts = nullptr;                      // 2 x (n + 1) x u64: gathered counts, gathered flags
	void *d_recv = nullptr; size_t recv_cap = 0;   // rank 0
	uint64_t recv_total = 0;                       // records of the last gather, resident in d_recv
	uint64_t staged = 0;                           // records put into d_send device to device (bhip_comm_stage_device) since the last gather
	bool stage_failed = false;
	bool broken = false;
};
struct Comm { int n = 0; std::vector<Peer> local; };
static const unsigned long long FAILED = ~0ull;

static Peer *peer_of(Comm *C, int rank) {
	if (!C) return nullptr;
	for (Peer &p : C->local) if (p.rank == rank) return &p;
	return nullptr;
}
static bool grow(void **p, size_t *cap, size_t bytes) {
	if (bytes <= *cap) return true;
	if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
	const size_t want = bytes + bytes / 8 + 4096;
	if (hipMalloc(p, want) != hipSuccess) { (void)hipGetLastError(); *p = nullptr; return false; }
	*cap = want;
	return true;
}
// the same, keeping the first `keep` bytes
static bool grow_keep(void **p, size_t *cap, size_t bytes, size_t keep, hipStream_t st) {
	if (bytes <= *cap) return true;
	const size_t want = bytes + bytes / 2 + 4096;
	void *n = nullptr;
	if (hipMalloc(&n, want) != hipSuccess) { (void)hipGetLastError(); return false; }
	if (keep && *p && (hipMemcpyAsync(n, *p, keep, hipMemcpyDeviceToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) { (void)hipGetLastError(); (void)hipFree(n); return false; }
	if (*p) (void)hipFree(*p);
	*p = n; *cap = want;
	return true;
}
static int peer_init(Peer &p, int n) {
	if (hipSetDevice(p.dev) != hipSuccess || hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking) != hipSuccess ||
	    hipMalloc(&p.d_counts, sizeof(unsigned long long) * 2 * (size_t)(n + 1)) != hipSuccess) { (void)hipGetLastError(); return -1; }
	return 0;
}
static void abort_peer(Peer &p) { if (p.comm && !p.broken) { (void)ncclCommAbort(p.comm); p.comm = nullptr; } p.broken = true; }

extern "C" void bhip_comm_destroy(void *comm) {
	Comm *C = (Comm *)comm;
	if (!C) return;
	for (Peer &p : C->local) {
		(void)hipSetDevice(p.dev);
		if (p.stream) { (void)hipStreamSynchronize(p.stream); (void)hipStreamDestroy(p.stream); }
		if (p.d_send) (void)hipFree(p.d_send);
		if (p.d_counts) (void)hipFree(p.d_counts);
		if (p.d_recv) (void)hipFree(p.d_recv);
		if (p.comm) (void)ncclCommDestroy(p.comm);
	}
	(void)hipGetLastError();
	delete C;
}

extern "C" int bhip_comm_create(int n_ranks, const int *devices, void **comm_out) {
	if (!comm_out || n_ranks < 1 || !devices) return bhip_fail_msg(BHIP_E_ARG, "bad communicator arguments");
	*comm_out = nullptr;
	Comm *C = new Comm();
	C->n = n_ranks; C->local.resize((size_t)n_ranks);
	std::vector<ncclComm_t> cs((size_t)n_ranks, nullptr);
	ncclResult_t r = ncclCommInitAll(cs.data(), n_ranks, devices);
	if (r != ncclSuccess) { delete C; return bhip_fail_msg(BHIP_E_DEVICE, "ncclCommInitAll(%d ranks): %s", n_ranks, ncclGetErrorString(r)); }
	for (int k = 0; k < n_ranks; ++k) { C->local[k].rank = k; C->local[k].dev = devices[k]; C->local[k].comm = cs[k]; }
	for (int k = 0; k < n_ranks; ++k) if (peer_init(C->local[k], n_ranks)) { bhip_comm_destroy(C); return bhip_fail_msg(BHIP_E_DEVICE, "communicator set-up failed on device %d", devices[k]); }
	*comm_out = C;
	return BHIP_OK;
}

extern "C" int bhip_comm_unique_id(void *id128) {
	if (!id128) return bhip_fail_msg(BHIP_E_ARG, "null id");
	static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
	ncclUniqueId id;
	ncclResult_t r = ncclGetUniqueId(&id);
	if (r != ncclSuccess) return bhip_fail_msg(BHIP_E_DEVICE, "ncclGetUniqueId: %s", ncclGetErrorString(r));
	memcpy(id128, &id, sizeof id);
	return BHIP_OK;
}

extern "C" int bhip_comm_create_rank(int n_ranks, int rank, int device, const void *id128, void **comm_out) {
	if (!comm_out || n_ranks < 1 || rank < 0 || rank >= n_ranks || !id128) return bhip_fail_msg(BHIP_E_ARG, "bad communicator arguments");
	*comm_out = nullptr;
	if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return bhip_fail_msg(BHIP_E_DEVICE, "device %d not present", device); }
	ncclUniqueId id;
	memcpy(&id, id128, sizeof id);
	Comm *C = new Comm();
	C->n = n_ranks; C->local.resize(1);
	C->local[0].rank = rank; C->local[0].dev = device;
	ncclResult_t r = ncclCommInitRank(&C->local[0].comm, n_ranks, id, rank);
	if (r != ncclSuccess) { C->local[0].comm = nullptr; bhip_comm_destroy(C); return bhip_fail_msg(BHIP_E_DEVICE, "ncclCommInitRank(rank %d of %d): %s", rank, n_ranks, ncclGetErrorString(r)); }
	if (peer_init(C->local[0], n_ranks)) { bhip_comm_destroy(C); return bhip_fail_msg(BHIP_E_DEVICE, "communicator set-up failed on device %d", device); }
	*comm_out = C;
	return BHIP_OK;
}

// every rank contributes one 64-bit word (FAILED = "I cannot go on"); all[n] is the same on every rank afterwards.
// which = 0 / 1: the two halves of d_counts (counts, flags).
static int exchange_word(Comm *C, Peer *P, int which, unsigned long long mine, std::vector<unsigned long long> &all) {
	unsigned long long *dc = (unsigned long long *)P->d_counts + (size_t)which * (size_t)(C->n + 1);
	all.assign((size_t)C->n, 0);
	if (P->broken) return bhip_fail_msg(BHIP_E_DEVICE, "communicator was aborted by an earlier failure");
	bool ok = hipMemcpyAsync(dc + C->n, &mine, sizeof mine, hipMemcpyHostToDevice, P->stream) == hipSuccess;
	ncclResult_t r = ncclAllGather(dc + C->n, dc, 1, ncclUint64, P->comm, P->stream);
	ok = ok && r == ncclSuccess && hipMemcpyAsync(all.data(), dc, sizeof(unsigned long long) * (size_t)C->n, hipMemcpyDeviceToHost, P->stream) == hipSuccess
	     && hipStreamSynchronize(P->stream) == hipSuccess;
	if (!ok) { (void)hipGetLastError(); abort_peer(*P); return bhip_fail_msg(BHIP_E_DEVICE, "count exchange failed on rank %d%s%s", P->rank, r != ncclSuccess ? ": " : "", r != ncclSuccess ? ncclGetErrorString(r) : ""); }
	return BHIP_OK;
}

// Called by the host thread of every rank (all n_ranks calls must be in flight together).  hits / n: the rank's records in
// host memory.  Rank 0 receives every rank's records, rank order, into out (capacity cap records); counts[n_ranks] gets the
// per-rank numbers on every rank.  BHIP_E_CAPACITY when out is too small (*n_total is what is needed; nothing is copied).
// staged = true: the rank's n records are in its send buffer already (bhip_comm_stage_device put them there device to device, batch
// after batch); false: they are uploaded from `hits`
static int gather_impl(Comm *C, int rank, const BhipHit *hits, bool staged, uint64_t n, BhipHit *out, uint64_t cap, uint64_t *n_total, uint64_t *counts) {
	Peer *P = peer_of(C, rank);
	if (!P || !n_total) return bhip_fail_msg(BHIP_E_ARG, "bad gather arguments");
	*n_total = 0;
	// 0. local preparation: send buffer + copy (a failure is announced with the count)
	bool ok = hipSetDevice(P->dev) == hipSuccess;
	const size_t bytes = (size_t)n * sizeof(BhipHit);
	if (staged) ok = ok && !P->stage_failed && P->staged == n && (!bytes || (P->d_send && P->send_cap >= bytes));
	else {
		ok = ok && (!bytes || hits) && grow(&P->d_send, &P->send_cap, bytes ? bytes : 1);
		if (ok && bytes) ok = hipMemcpyAsync(P->d_send, hits, bytes, hipMemcpyHostToDevice, P->stream) == hipSuccess;
	}
	P->staged = 0; P->stage_failed = false;
	if (!ok) (void)hipGetLastError();
	// 1. everybody learns everybody's count
	std::vector<unsigned long long> hc;
	int rc = exchange_word(C, P, 0, ok ? (unsigned long long)n : FAILED, hc);
	if (rc) return rc;
	for (int k = 0; k < C->n; ++k) if (hc[k] == FAILED) return bhip_fail_msg(BHIP_E_DEVICE, "rank %d could not stage its records for the gather (device memory?)", k);
	uint64_t total = 0;
	for (int k = 0; k < C->n; ++k) { if (counts) counts[k] = hc[k]; total += hc[k]; }
	*n_total = total;
	// 2. rank 0 makes room, and says so (the capacity decision is rank 0's alone, but every rank must take the same path through
	// the collectives: rank 0 receives into its device buffer in any case, only the copy to the caller's memory depends on `cap`)
	const bool fits = total <= cap || rank != 0;
	bool ready = true;
	if (rank == 0) ready = grow(&P->d_recv, &P->recv_cap, total ? (size_t)total * sizeof(BhipHit) : 1) && (out || !total || !fits);
	std::vector<unsigned long long> flags;
	if ((rc = exchange_word(C, P, 1, ready ? 1ull : FAILED, flags))) return rc;
	if (flags[0] == FAILED) return bhip_fail_msg(BHIP_E_DEVICE, "rank 0 has no room for the %llu gathered records", (unsigned long long)total);
	// 3. the records: grouped point-to-point
	ncclResult_t r = ncclGroupStart(), r2 = ncclSuccess;
	hipError_t he = hipSuccess;
	if (r == ncclSuccess) {
		if (rank == 0) {
			size_t off = 0;
			for (int k = 0; k < C->n; ++k) {
				const size_t b = (size_t)hc[k] * sizeof(BhipHit);
				if (k == 0) { if (b && he == hipSuccess) he = hipMemcpyAsync((char *)P->d_recv + off, P->d_send, b, hipMemcpyDeviceToDevice, P->stream); }
				else if (b && r2 == ncclSuccess) r2 = ncclRecv((char *)P->d_recv + off, b, ncclUint8, k, P->comm, P->stream);
				off += b;
			}
		} else if (bytes) r2 = ncclSend(P->d_send, bytes, ncclUint8, 0, P->comm, P->stream);
		r = ncclGroupEnd();
	}
	if (r == ncclSuccess && r2 == ncclSuccess && he == hipSuccess && rank == 0 && fits && total)
		he = hipMemcpyAsync(out, P->d_recv, (size_t)total * sizeof(BhipHit), hipMemcpyDeviceToHost, P->stream);
	if (he == hipSuccess) he = hipStreamSynchronize(P->stream);
	if (r != ncclSuccess || r2 != ncclSuccess || he != hipSuccess) {
		(void)hipGetLastError(); abort_peer(*P);
		return bhip_fail_msg(BHIP_E_DEVICE, "record gather failed on rank %d: %s", rank, r != ncclSuccess ? ncclGetErrorString(r) : r2 != ncclSuccess ? ncclGetErrorString(r2) : hipGetErrorString(he));
	}
	if (rank == 0) P->recv_total = total;
	if (!fits) return bhip_fail_msg(BHIP_E_CAPACITY, "record buffer holds %llu records, %llu needed", (unsigned long long)cap, (unsigned long long)total);
	return BHIP_OK;
}
extern "C" int bhip_comm_gather_hits(void *comm, int rank, const BhipHit *hits, uint64_t n, BhipHit *out, uint64_t cap, uint64_t *n_total, uint64_t *counts) {
	return gather_impl((Comm *)comm, rank, hits, false, n, out, cap, n_total, counts);
}
// The records of the handle's LAST alignment call (still resident on its device) into the rank's send buffer, behind the
// `first_record` records staged before: device to device, no second trip over PCIe.  A failure (memory) is remembered and makes
// bhip_comm_gather_staged report this rank as failed -- the caller then falls back on bhip_comm_gather_hits for everybody.
extern "C" int bhip_comm_stage_device(void *comm, int rank, void *handle, uint64_t first_record, uint64_t *n_records) {
	Comm *C = (Comm *)comm;
	Peer *P = peer_of(C, rank);
	if (!P || !handle) return bhip_fail_msg(BHIP_E_ARG, "bad staging arguments");
	if (first_record != P->staged) { P->stage_failed = true; return bhip_fail_msg(BHIP_E_ARG, "records staged out of order (%llu staged, batch starts at %llu)", (unsigned long long)P->staged, (unsigned long long)first_record); }
	uint64_t n = 0;
	int rc = bhip_copy_hits_device(handle, nullptr, 0, &n);      // (how many)
	if (rc && rc != BHIP_E_CAPACITY) { P->stage_failed = true; return rc; }
	if (hipSetDevice(P->dev) != hipSuccess || !grow_keep(&P->d_send, &P->send_cap, (size_t)(first_record + n) * sizeof(BhipHit) + 1, (size_t)first_record * sizeof(BhipHit), P->stream)) {
		(void)hipGetLastError(); P->stage_failed = true;
		return bhip_fail_msg(BHIP_E_DEVICE, "no device memory for %llu staged records", (unsigned long long)(first_record + n));
	}
	if (n && (rc = bhip_copy_hits_device(handle, (char *)P->d_send + (size_t)first_record * sizeof(BhipHit), n, &n))) { P->stage_failed = true; return rc; }
	P->staged = first_record + n;
	if (n_records) *n_records = n;
	return BHIP_OK;
}
// the gather of bhip_comm_gather_hits with the rank's n records taken from its send buffer (staged batch by batch)
extern "C" int bhip_comm_gather_staged(void *comm, int rank, uint64_t n, BhipHit *out, uint64_t cap, uint64_t *n_total, uint64_t *counts) {
	return gather_impl((Comm *)comm, rank, nullptr, true, n, out, cap, n_total, counts);
}
// forget what has been staged (a search that ends without a gather)
extern "C" void bhip_comm_stage_reset(void *comm, int rank) {
	Peer *P = peer_of((Comm *)comm, rank);
	if (P) { P->staged = 0; P->stage_failed = false; }
}

// rank 0, after a gather that ended with BHIP_E_CAPACITY: the gathered records are still on its device; this copies them into a
// buffer that is large enough.  No collective -- the other ranks are not involved.
extern "C" int bhip_comm_fetch_gathered(void *comm, BhipHit *out, uint64_t cap, uint64_t *n_total) {
	Comm *C = (Comm *)comm;
	Peer *P = peer_of(C, 0);
	if (!P || !n_total) return bhip_fail_msg(BHIP_E_ARG, "rank 0 does not live in this process");
	*n_total = P->recv_total;
	if (P->recv_total > cap) return bhip_fail_msg(BHIP_E_CAPACITY, "record buffer holds %llu records, %llu needed", (unsigned long long)cap, (unsigned long long)P->recv_total);
	if (!P->recv_total) return BHIP_OK;
	if (!out) return bhip_fail_msg(BHIP_E_ARG, "null output");
	if (hipSetDevice(P->dev) != hipSuccess || hipMemcpyAsync(out, P->d_recv, (size_t)P->recv_total * sizeof(BhipHit), hipMemcpyDeviceToHost, P->stream) != hipSuccess ||
	    hipStreamSynchronize(P->stream) != hipSuccess) { (void)hipGetLastError(); return bhip_fail_msg(BHIP_E_DEVICE, "copy of the gathered records failed"); }
	return BHIP_OK;
}

// element-wise minimum over the ranks of n bytes of host memory, in place (database-sharded mode: one byte per unique query =
// the smallest edit distance any reference of the rank's clump range reaches, 255 = none)
extern "C" int bhip_comm_allreduce_min(void *comm, int rank, uint8_t *buf, uint64_t n) {
	Comm *C = (Comm *)comm;
	Peer *P = peer_of(C, rank);
	if (!P || (n && !buf)) return bhip_fail_msg(BHIP_E_ARG, "bad reduction arguments");
	bool ok = hipSetDevice(P->dev) == hipSuccess && grow(&P->d_send, &P->send_cap, n ? n : 1);
	if (ok && n) ok = hipMemcpyAsync(P->d_send, buf, n, hipMemcpyHostToDevice, P->stream) == hipSuccess;
	if (!ok) (void)hipGetLastError();
	std::vector<unsigned long long> all;
	int rc = exchange_word(C, P, 0, ok ? (unsigned long long)n : FAILED, all);
	if (rc) return rc;
	for (int k = 0; k < C->n; ++k) {
		if (all[k] == FAILED) return bhip_fail_msg(BHIP_E_DEVICE, "rank %d could not stage its minima for the reduction", k);
		if (all[k] != n) return bhip_fail_msg(BHIP_E_ARG, "ranks disagree on the number of queries (%llu on rank %d, %llu on rank %d)", (unsigned long long)all[k], k, (unsigned long long)n, rank);
	}
	if (!n) return BHIP_OK;
	ncclResult_t r = ncclAllReduce(P->d_send, P->d_send, n, ncclUint8, ncclMin, P->comm, P->stream);
	hipError_t he = r == ncclSuccess ? hipMemcpyAsync(buf, P->d_send, n, hipMemcpyDeviceToHost, P->stream) : hipSuccess;
	if (he == hipSuccess) he = hipStreamSynchronize(P->stream);
	if (r != ncclSuccess || he != hipSuccess) {
		(void)hipGetLastError(); abort_peer(*P);
		return bhip_fail_msg(BHIP_E_DEVICE, "minimum reduction failed on rank %d: %s", rank, r != ncclSuccess ? ncclGetErrorString(r) : hipGetErrorString(he));
	}
	return BHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// Exchanges of the cooperative accelerator build (bhip_build_accelerator_shared, bhip_acx.hip): region r of an array that every rank
// holds is valid on rank r and must become valid everywhere.
// RCCL: the statuses with the count exchange, then ONE group of n broadcasts, every region in place from its builder (n - 1 regions
// arrive over n - 1 different xGMI links; pieces of at most 1 GiB so that no count exceeds what a transport takes in one call).
extern "C" int bhip_comm_share(void *comm_rank, void *device_base, const uint64_t *byte_off, int part, int n_parts, int status) {
	BhipCommRank *cr = (BhipCommRank *)comm_rank;
	Comm *C = cr ? (Comm *)cr->comm : nullptr;
	Peer *P = cr ? peer_of(C, cr->rank) : nullptr;
	if (!P || !byte_off || C->n != n_parts || cr->rank != part) return bhip_fail_msg(BHIP_E_ARG, "bad arguments of the region exchange");
	if (hipSetDevice(P->dev) != hipSuccess) { (void)hipGetLastError(); status = 1; }
	std::vector<unsigned long long> all;
	if (exchange_word(C, P, 0, status ? FAILED : (unsigned long long)byte_off[n_parts], all)) return -1;
	for (int k = 0; k < n_parts; ++k) if (all[k] == FAILED) return 1;
	for (int k = 0; k < n_parts; ++k) if (all[k] != byte_off[n_parts]) { bhip_fail_msg(BHIP_E_ARG, "ranks disagree on the size of the shared array"); return -1; }
	const size_t kPiece = (size_t)1 << 30;
	ncclResult_t r = ncclGroupStart();
	for (int k = 0; k < n_parts && r == ncclSuccess; ++k)
		for (uint64_t a = byte_off[k]; a < byte_off[k + 1] && r == ncclSuccess; a += kPiece) {
			char *p = (char *)device_base + a;
			r = ncclBroadcast(p, p, (size_t)std::min<uint64_t>(kPiece, byte_off[k + 1] - a), ncclInt8, k, P->comm, P->stream);
		}
	ncclResult_t r2 = ncclGroupEnd();
	if (r == ncclSuccess) r = r2;
	hipError_t he = r == ncclSuccess ? hipStreamSynchronize(P->stream) : hipSuccess;
	if (r != ncclSuccess || he != hipSuccess) {
		(void)hipGetLastError(); abort_peer(*P);
		bhip_fail_msg(BHIP_E_DEVICE, "region exchange failed on rank %d: %s", part, r != ncclSuccess ? ncclGetErrorString(r) : hipGetErrorString(he));
		return -1;
	}
	return 0;
}

// The threads of one process (burst_hip --gpus N): a barrier, every rank's array published, every rank PULLS the other regions device
// to device (hipMemcpyPeerAsync: over xGMI between two devices, an ordinary copy when the ranks share one), a barrier.
struct Team {
	int n = 0;
	pthread_barrier_t bar;
	std::vector<void *> base; std::vector<int> dev, status, failed;
};
extern "C" int bhip_team_create(int n_ranks, void **team) {
	if (!team || n_ranks < 1) return bhip_fail_msg(BHIP_E_ARG, "bad team arguments");
	Team *T = new Team();
	T->n = n_ranks; T->base.assign((size_t)n_ranks, nullptr); T->dev.assign((size_t)n_ranks, 0); T->status.assign((size_t)n_ranks, 0); T->failed.assign((size_t)n_ranks, 0);
	if (pthread_barrier_init(&T->bar, nullptr, (unsigned)n_ranks)) { delete T; return bhip_fail_msg(BHIP_E_INTERNAL, "no barrier for %d ranks", n_ranks); }
	if (n_ranks > 1) bhip_vmm_peer_access_flag().fetch_add(1);      // (record areas reserved while a team of several ranks exists are readable by the peer devices)
	*team = T;
	return BHIP_OK;
}
extern "C" void bhip_team_destroy(void *team) {
	Team *T = (Team *)team;
	if (!T) return;
	if (T->n > 1) bhip_vmm_peer_access_flag().fetch_sub(1);      // (the last team gone: a process with one rank touches no other device again)
	pthread_barrier_destroy(&T->bar);
	delete T;
}
extern "C" int bhip_team_share(void *team, void *device_base, const uint64_t *byte_off, int part, int n_parts, int status) {
	Team *T = (Team *)team;
	if (!T || !byte_off || T->n != n_parts || part < 0 || part >= n_parts) return bhip_fail_msg(BHIP_E_ARG, "bad arguments of the region exchange");
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); status = 1; }
	// (everything this rank has enqueued on its device -- whatever the stream -- has happened before a peer reads its region)
	if (!status && hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); status = 1; }
	T->base[(size_t)part] = device_base; T->dev[(size_t)part] = dev; T->status[(size_t)part] = status; T->failed[(size_t)part] = 0;
	pthread_barrier_wait(&T->bar);
	int any = 0;
	for (int k = 0; k < n_parts; ++k) any |= T->status[(size_t)k];
	if (!any) {
		hipStream_t st = nullptr;
		bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
		const size_t kPiece = (size_t)1 << 30;
		for (int k = 0; k < n_parts && ok; ++k) {
			if (k == part) continue;
			for (uint64_t a = byte_off[k]; a < byte_off[k + 1] && ok; a += kPiece) {
				const size_t n = (size_t)std::min<uint64_t>(kPiece, byte_off[k + 1] - a);
				ok = (T->dev[(size_t)k] == dev ? hipMemcpyAsync((char *)device_base + a, (const char *)T->base[(size_t)k] + a, n, hipMemcpyDeviceToDevice, st)
				                              : hipMemcpyPeerAsync((char *)device_base + a, dev, (const char *)T->base[(size_t)k] + a, T->dev[(size_t)k], n, st)) == hipSuccess;
			}
		}
		ok = ok && hipStreamSynchronize(st) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
		if (st) (void)hipStreamDestroy(st);
		if (!ok) {      // the peers' memory is not reachable device to device here: the same regions through host memory, 256 MiB at a time
			(void)hipGetLastError();
			const size_t kHop = (size_t)256 << 20;
			void *hop = nullptr;
			ok = hipHostMalloc(&hop, kHop, hipHostMallocDefault) == hipSuccess;
			for (int k = 0; k < n_parts && ok; ++k) {
				if (k == part) continue;
				for (uint64_t a = byte_off[k]; a < byte_off[k + 1] && ok; a += kHop) {
					const size_t n = (size_t)std::min<uint64_t>(kHop, byte_off[k + 1] - a);
					ok = hipMemcpy(hop, (const char *)T->base[(size_t)k] + a, n, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy((char *)device_base + a, hop, n, hipMemcpyHostToDevice) == hipSuccess;
				}
			}
			if (hop) (void)hipHostFree(hop);
		}
		if (!ok) { bhip_fail_msg(BHIP_E_DEVICE, "region exchange (peer copies) failed on rank %d: %s", part, hipGetErrorString(hipGetLastError())); T->failed[(size_t)part] = 1; }
	}
	pthread_barrier_wait(&T->bar);      // nobody touches its array while a peer still reads it
	if (any) return 1;
	int bad = 0;
	for (int k = 0; k < n_parts; ++k) bad |= T->failed[(size_t)k];
	pthread_barrier_wait(&T->bar);      // (the flags are read before the next exchange resets them)
	return bad ? -1 : 0;
}
