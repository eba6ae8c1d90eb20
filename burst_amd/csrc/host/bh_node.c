/* bh_node.c -- hand-over of the hit records between the ranks of ONE node that live in different processes (bench.py and
 * burst_amd.run under torch.distributed.run: one process per GPU).  No reference counterpart: the reference is one process
 * (SURVEY.md 8e).
 *
 * The records of a search end in rank 0's HOST memory, where the consolidation runs (bh_report: burst.c:4582-4891).  An RCCL
 * gather takes them there by way of the devices -- host -> device on every rank, xGMI to rank 0's device, device -> host on rank
 * 0 -- i.e. all N shares through rank 0's one PCIe link, after the alignment has ended.  Here every rank's record buffer IS a
 * shared-memory segment: page-locked, so the device copies of a batch's records land in it behind the batch that produced them,
 * over the rank's OWN PCIe link and beside the next batch's kernels, as they do in a single-rank run; rank 0 maps the other ranks'
 * segments.  What is left of the exchange is one word per rank (record count + call number, release / acquire) and the
 * concatenation in rank 0's memory.  The query-sharded path needs no collective at all; the database-sharded one keeps its
 * ncclAllReduce(MIN) of one byte per query (bhip_comm_allreduce_min), which is a real exchange.
 *
 * Segment = one 4 KB header + records.  The first `cap` records are allocated (posix_fallocate: a full /dev/shm says so here, not
 * with a SIGBUS later) and page-locked; the mapping is four times as long so that a search that brings more than expected can
 * still publish (those pages are allocated when -- if ever -- they are written, by a plain copy).
 */
#define _GNU_SOURCE
#include "burst_host.h"
#include <errno.h>
#include <fcntl.h>
#include <omp.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define NODE_MAGIC 0x42484E4F44453031ull      /* "BHNODE01" */
#define NODE_HDR   4096u
#define NODE_VIRT  4u                         /* mapping length / allocated length */

typedef struct {
	uint64_t magic;
	uint64_t cap, virt;                       /* records allocated + page-locked / records the mapping has room for */
	uint64_t n_hits; int64_t status;          /* of call `published` */
	uint64_t published;                       /* call number whose records are complete (release store, acquire load) */
	uint64_t consumed;                        /* rank 0's segment only: call number rank 0 has finished reading */
} NodeHdr;

struct BhNode {
	int rank, n_ranks, locked, fd;             /* fd: this rank's segment (kept for posix_fallocate when a search outgrows it) */
	char job[96];
	NodeHdr *own; size_t own_bytes;
	NodeHdr *peer[BH_MAX_RANKS]; size_t peer_bytes[BH_MAX_RANKS];      /* rank 0: everybody's segment; others: [0] = rank 0's */
	uint64_t seq;                             /* calls begun */
	double timeout;
};

static double now_sec(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static void seg_name(const BhNode *N, int rank, char *out, size_t cap) { snprintf(out, cap, "/burst_hip.%s.%d", N->job, rank); }
static BhipHit *seg_records(NodeHdr *h) { return (BhipHit *)((char *)h + NODE_HDR); }

/* wait until *word >= want (acquire); 0 = reached, 1 = timed out */
static int wait_for(const uint64_t *word, uint64_t want, double timeout) {
	const double t0 = now_sec();
	for (unsigned spin = 0;; ++spin) {
		if (__atomic_load_n(word, __ATOMIC_ACQUIRE) >= want) return 0;
		if (spin < 2000) sched_yield();
		else { struct timespec ts = {0, 20000}; nanosleep(&ts, NULL); if (now_sec() - t0 > timeout) return 1; }
	}
}

static NodeHdr *map_peer(BhNode *N, int rank, size_t *bytes) {
	char nm[160];
	seg_name(N, rank, nm, sizeof nm);
	const double t0 = now_sec();
	for (;;) {
		const int fd = shm_open(nm, O_RDWR, 0600);
		if (fd >= 0) {
			struct stat sb;
			if (!fstat(fd, &sb) && (size_t)sb.st_size >= NODE_HDR) {
				NodeHdr *h = mmap(NULL, (size_t)sb.st_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
				close(fd);
				if (h == MAP_FAILED) return NULL;
				if (!wait_for(&h->magic, NODE_MAGIC, N->timeout) && h->magic == NODE_MAGIC) {
					/* the allocated part into THIS process's page tables now (set-up), not page by page inside the first large
					 * hand-over: 0.4 GB of a peer's records = 100 000 minor faults = more time than copying them.  (Not the
					 * unallocated tail: touching a hole of a tmpfs file allocates it.) */
					const size_t locked = NODE_HDR + (size_t)h->cap * sizeof(BhipHit);
					#ifndef MADV_POPULATE_READ
					#define MADV_POPULATE_READ 22
					#endif
					if (madvise(h, locked, MADV_POPULATE_READ)) {
						volatile const char *c = (volatile const char *)h; char sink = 0;
						for (size_t o = 0; o < locked; o += 4096) sink ^= c[o];
						(void)sink;
					}
					*bytes = (size_t)sb.st_size; return h;
				}
				munmap(h, (size_t)sb.st_size);
				return NULL;
			}
			close(fd);
		}
		if (now_sec() - t0 > N->timeout) return NULL;
		struct timespec ts = {0, 2000000}; nanosleep(&ts, NULL);
	}
}

int bh_node_open(const char *job, int rank, int n_ranks, uint64_t cap_records, BhNode **out) {
	if (!out || !job || !*job || strlen(job) > 80 || rank < 0 || rank >= n_ranks || n_ranks > BH_MAX_RANKS) return bh_set_error(BH_E_USAGE, "bad node exchange arguments");
	*out = NULL;
	BhNode *N = calloc(1, sizeof(*N));
	if (!N) return bh_set_error(BH_E_OOM, "OOM:node");
	N->rank = rank; N->n_ranks = n_ranks; N->timeout = 120.0; N->fd = -1;
	{ const char *e = getenv("BURST_NODE_TIMEOUT"); if (e && atof(e) > 0) N->timeout = atof(e); }
	snprintf(N->job, sizeof N->job, "%s", job);
	if (cap_records < 1024) cap_records = 1024;
	char nm[160];
	seg_name(N, rank, nm, sizeof nm);
	(void)shm_unlink(nm);      /* a segment of this name left by a job that died */
	const int fd = shm_open(nm, O_CREAT | O_EXCL | O_RDWR, 0600);
	if (fd < 0) { free(N); return bh_set_error(BH_E_IO, "shm_open(%s): %s", nm, strerror(errno)); }
	const size_t locked = NODE_HDR + (size_t)cap_records * sizeof(BhipHit), bytes = NODE_HDR + (size_t)cap_records * NODE_VIRT * sizeof(BhipHit);
	int e = ftruncate(fd, (off_t)bytes) ? errno : posix_fallocate(fd, 0, (off_t)locked);
	NodeHdr *h = e ? MAP_FAILED : mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
	if (!e && h == MAP_FAILED) e = errno;
	if (e) { close(fd); (void)shm_unlink(nm); free(N); return bh_set_error(e == ENOSPC ? BH_E_OOM : BH_E_IO, "shared-memory segment of %lu bytes (%s): %s", (unsigned long)locked, nm, strerror(e)); }
	N->own = h; N->own_bytes = bytes; N->fd = fd;
	h->cap = cap_records; h->virt = cap_records * NODE_VIRT; h->n_hits = 0; h->status = 0; h->published = 0; h->consumed = 0;
	/* page-locked: the asynchronous device copies of the records land here (without a device -- tests -- it stays pageable) */
	N->locked = bhip_host_register(seg_records(h), (uint64_t)cap_records * sizeof(BhipHit)) == 0;
	__atomic_store_n(&h->magic, NODE_MAGIC, __ATOMIC_RELEASE);
	if (rank == 0) { N->peer[0] = h; N->peer_bytes[0] = bytes; }
	else if (!(N->peer[0] = map_peer(N, 0, &N->peer_bytes[0]))) { bh_node_close(N); return bh_set_error(BH_E_IO, "rank 0's shared-memory segment did not appear (job %s)", job); }
	*out = N;
	return BH_OK;
}

void bh_node_close(BhNode *N) {
	if (!N) return;
	for (int r = 0; r < N->n_ranks; ++r) if (N->peer[r] && N->peer[r] != N->own) munmap(N->peer[r], N->peer_bytes[r]);
	if (N->own) {
		if (N->locked) (void)bhip_host_unregister(seg_records(N->own));
		munmap(N->own, N->own_bytes);
		if (N->fd >= 0) close(N->fd);
		char nm[160];
		seg_name(N, N->rank, nm, sizeof nm);
		(void)shm_unlink(nm);
	}
	free(N);
}

/* the rank's BhRun takes the segment's allocated part as its record buffer (hitsPinned = 2: memory the run does not own); a run
 * that has outgrown the segment keeps its own, larger buffer (bh_node_publish then copies) */
void bh_node_attach(BhNode *N, BhRun *run) {
	if (run->hits == seg_records(N->own)) return;
	if (run->hits && run->capHits > N->own->cap) return;
	if (run->hits) { BhRun old = *run; memset(run, 0, sizeof *run); bh_run_free(&old); }
	run->hits = seg_records(N->own); run->capHits = N->own->cap; run->hitsPinned = 2; run->nHits = 0;
}

/* before a rank writes records of a new call into its segment: rank 0 must have read the previous call's */
int bh_node_begin(BhNode *N) {
	const uint64_t s = ++N->seq;
	if (s > 1 && wait_for(&N->peer[0]->consumed, s - 1, N->timeout)) return bh_set_error(BH_E_INTERNAL, "rank 0 has not taken the records of call %lu (job %s)", (unsigned long)(s - 1), N->job);
	return BH_OK;
}

/* the call's records are complete (status != 0: the rank failed, it brings nothing) */
int bh_node_publish(BhNode *N, const BhRun *run, int status) {
	NodeHdr *h = N->own;
	uint64_t n = status ? 0 : run->nHits;
	if (n && run->hits != seg_records(h)) {      /* the search outgrew the segment's buffer and went on in a private one */
		if (n > h->virt) { status = BH_E_OOM; n = 0; }
		else {
			/* (allocate first: a full /dev/shm is an error here, not a signal inside memcpy) */
			const int e = posix_fallocate(N->fd, 0, (off_t)(NODE_HDR + n * sizeof(BhipHit)));
			if (e) { status = BH_E_OOM; n = 0; }
			else memcpy(seg_records(h), run->hits, n * sizeof(BhipHit));
		}
	}
	h->n_hits = n; h->status = status;
	__atomic_store_n(&h->published, N->seq, __ATOMIC_RELEASE);
	return status == BH_E_OOM && !n && run->nHits ? bh_set_error(BH_E_OOM, "%lu records do not fit the shared-memory segment (%lu allocated)", (unsigned long)run->nHits, (unsigned long)h->cap) : BH_OK;
}

/* rank 0: every rank's records of the current call, in rank order, into `all` (page-locked or not: no device writes to it);
 * counts[n_ranks] optional.  A rank that failed or does not answer makes the call fail -- nobody waits for ever. */
int bh_node_collect(BhNode *N, BhRun *all, uint64_t *counts) {
	if (N->rank != 0) return bh_set_error(BH_E_USAGE, "only rank 0 collects");
	int rc = BH_OK;
	uint64_t at[BH_MAX_RANKS + 1]; at[0] = 0;
	for (int r = 0; r < N->n_ranks && !rc; ++r) {
		if (!N->peer[r]) {
			if (!(N->peer[r] = map_peer(N, r, &N->peer_bytes[r]))) { rc = bh_set_error(BH_E_IO, "rank %d's shared-memory segment did not appear (job %s)", r, N->job); break; }
			/* mapped: the name has done its work (a job that dies later leaves nothing of this rank in /dev/shm) */
			char nm[160]; seg_name(N, r, nm, sizeof nm); (void)shm_unlink(nm);
		}
		if (wait_for(&N->peer[r]->published, N->seq, N->timeout)) { rc = bh_set_error(BH_E_INTERNAL, "rank %d did not deliver the records of call %lu within %.0f s", r, (unsigned long)N->seq, N->timeout); break; }
		if (N->peer[r]->status) { rc = bh_set_error(BH_E_DEVICE, "rank %d failed in its search (code %ld)", r, (long)N->peer[r]->status); break; }
		if (N->peer[r]->n_hits > N->peer[r]->virt) { rc = bh_set_error(BH_E_INTERNAL, "rank %d announces %lu records in a segment of %lu", r, (unsigned long)N->peer[r]->n_hits, (unsigned long)N->peer[r]->virt); break; }
		at[r + 1] = at[r] + N->peer[r]->n_hits;
		if (counts) counts[r] = N->peer[r]->n_hits;
	}
	if (!rc && bh_run_reserve_plain(all, at[N->n_ranks] + 1)) rc = BH_E_OOM;
	if (!rc) {
		/* pieces of 4 MB over a team: the concatenation of 0.8 GB (40 M reads, BEST) is memory-bound, one thread copies ~10 GB/s */
		const uint64_t piece = (4u << 20) / sizeof(BhipHit);
		uint64_t n_pieces = 0, first[BH_MAX_RANKS + 1]; first[0] = 0;
		for (int r = 0; r < N->n_ranks; ++r) { n_pieces += (N->peer[r]->n_hits + piece - 1) / piece; first[r + 1] = n_pieces; }
		const int nt = omp_get_max_threads() > 16 ? 16 : omp_get_max_threads();      /* memory-bound: more threads do not copy faster */
		#pragma omp parallel for schedule(static) num_threads(nt)
		for (uint64_t p = 0; p < n_pieces; ++p) {
			int r = 0;
			while (p >= first[r + 1]) ++r;
			const uint64_t a = (p - first[r]) * piece, n = N->peer[r]->n_hits, b = a + piece < n ? a + piece : n;
			memcpy(all->hits + at[r] + a, seg_records(N->peer[r]) + a, (b - a) * sizeof(BhipHit));
		}
		all->nHits = at[N->n_ranks];
	}
	/* read or given up: the ranks may overwrite their segments */
	__atomic_store_n(&N->own->consumed, N->seq, __ATOMIC_RELEASE);
	return rc;
}
