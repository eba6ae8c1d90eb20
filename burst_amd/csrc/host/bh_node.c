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
	char *view; size_t stride;                /* rank 0: one address range, slot r = the allocated records of rank r's segment */
	uint8_t in_view[BH_MAX_RANKS]; int lent;  /* lent: the last hand-over gave out a view; it ends when rank 0 begins its next search */
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

/* rank 0: the allocated records of rank r's segment (file offset NODE_HDR) at slot r of the view, read-only, populated now */
#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif
static void view_slot(BhNode *N, int r, int fd, uint64_t cap) {
	if (!N->view || N->in_view[r]) return;
	const size_t len = ((size_t)cap * sizeof(BhipHit) + 4095) & ~(size_t)4095;
	if (len > N->stride) return;                                        /* a rank with a larger segment than rank 0's: it is copied */
	void *p = mmap(N->view + (size_t)r * N->stride, len, PROT_READ, MAP_SHARED | MAP_FIXED, fd, NODE_HDR);
	if (p == MAP_FAILED) return;
	if (madvise(p, len, MADV_POPULATE_READ)) { volatile const char *c = p; char sink = 0; for (size_t o = 0; o < len; o += 4096) sink ^= c[o]; (void)sink; }
	N->in_view[r] = 1;
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
				if (h == MAP_FAILED) { close(fd); return NULL; }
				if (!wait_for(&h->magic, NODE_MAGIC, N->timeout) && h->magic == NODE_MAGIC) {
					/* rank 0: the records' pages into THIS process's page tables now (set-up), not page by page inside the first large
					 * hand-over: 0.4 GB of a peer's records = 100 000 minor faults = more time than copying them.  (Only the
					 * allocated part: touching a hole of a tmpfs file allocates it.) */
					if (N->rank == 0) view_slot(N, rank, fd, h->cap);
					close(fd);
					*bytes = (size_t)sb.st_size; return h;
				}
				close(fd);
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
	if (rank == 0) {
		N->peer[0] = h; N->peer_bytes[0] = bytes;
		/* one address range over everybody's records: slots of a stride that is a multiple of the page and of the record size */
		N->stride = (((size_t)cap_records * sizeof(BhipHit) + 20479) / 20480) * 20480;
		void *v = mmap(NULL, N->stride * (size_t)n_ranks, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
		if (v != MAP_FAILED) { N->view = v; view_slot(N, 0, fd, cap_records); }
	}
	else if (!(N->peer[0] = map_peer(N, 0, &N->peer_bytes[0]))) { bh_node_close(N); return bh_set_error(BH_E_IO, "rank 0's shared-memory segment did not appear (job %s)", job); }
	*out = N;
	return BH_OK;
}

void bh_node_close(BhNode *N) {
	if (!N) return;
	for (int r = 0; r < N->n_ranks; ++r) if (N->peer[r] && N->peer[r] != N->own) munmap(N->peer[r], N->peer_bytes[r]);
	if (N->view) munmap(N->view, N->stride * (size_t)N->n_ranks);
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
	if (N->lent) { N->lent = 0; __atomic_store_n(&N->own->consumed, s - 1, __ATOMIC_RELEASE); }      /* rank 0 is done with the view it was given */
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

/* rank 0: wait until every rank has delivered the current call; at[r] = records of the ranks before r */
static int wait_all(BhNode *N, uint64_t *at, uint64_t *counts) {
	if (N->rank != 0) return bh_set_error(BH_E_USAGE, "only rank 0 collects");
	at[0] = 0;
	for (int r = 0; r < N->n_ranks; ++r) {
		if (!N->peer[r]) {
			if (!(N->peer[r] = map_peer(N, r, &N->peer_bytes[r]))) return bh_set_error(BH_E_IO, "rank %d's shared-memory segment did not appear (job %s)", r, N->job);
			/* mapped: the name has done its work (a job that dies later leaves nothing of this rank in /dev/shm) */
			char nm[160]; seg_name(N, r, nm, sizeof nm); (void)shm_unlink(nm);
		}
		if (wait_for(&N->peer[r]->published, N->seq, N->timeout)) return bh_set_error(BH_E_INTERNAL, "rank %d did not deliver the records of call %lu within %.0f s", r, (unsigned long)N->seq, N->timeout);
		if (N->peer[r]->status) return bh_set_error(BH_E_DEVICE, "rank %d failed in its search (code %ld)", r, (long)N->peer[r]->status);
		if (N->peer[r]->n_hits > N->peer[r]->virt) return bh_set_error(BH_E_INTERNAL, "rank %d announces %lu records in a segment of %lu", r, (unsigned long)N->peer[r]->n_hits, (unsigned long)N->peer[r]->virt);
		at[r + 1] = at[r] + N->peer[r]->n_hits;
		if (counts) counts[r] = N->peer[r]->n_hits;
	}
	return BH_OK;
}

/* rank 0: every rank's records of the current call, in rank order, into `all` (page-locked or not: no device writes to it);
 * counts[n_ranks] optional.  A rank that failed or does not answer makes the call fail -- nobody waits for ever. */
int bh_node_collect(BhNode *N, BhRun *all, uint64_t *counts) {
	uint64_t at[BH_MAX_RANKS + 1];
	int rc = wait_all(N, at, counts);
	if (rc == BH_E_USAGE) return rc;
	if (!rc && bh_run_reserve_plain(all, at[N->n_ranks] + 1)) rc = BH_E_OOM;
	if (!rc) {
		/* one piece per thread where that is >= 4 MB (0.8 GB for 40 M reads: memory-bound; long copies take the C library's
		 * streaming stores and spare the read of the destination) */
		const int nt = omp_get_max_threads() > 16 ? 16 : omp_get_max_threads();
		uint64_t piece = (at[N->n_ranks] + (uint64_t)nt - 1) / (uint64_t)nt;
		if (piece < (4u << 20) / sizeof(BhipHit)) piece = (4u << 20) / sizeof(BhipHit);
		uint64_t n_pieces = 0, first[BH_MAX_RANKS + 1]; first[0] = 0;
		for (int r = 0; r < N->n_ranks; ++r) { n_pieces += (N->peer[r]->n_hits + piece - 1) / piece; first[r + 1] = n_pieces; }
		#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
		for (uint64_t p = 0; p < n_pieces; ++p) {
			int r = 0;
			while (p >= first[r + 1]) ++r;
			const uint64_t a = (p - first[r]) * piece, n = N->peer[r]->n_hits, b = a + piece < n ? a + piece : n;
			/* (from the view's slot where the records lie inside it: that mapping was populated when it was made) */
			const BhipHit *src = N->view && N->in_view[r] && n <= N->peer[r]->cap ? (const BhipHit *)(N->view + (size_t)r * N->stride) : seg_records(N->peer[r]);
			memcpy(all->hits + at[r] + a, src + a, (b - a) * sizeof(BhipHit));
		}
		all->nHits = at[N->n_ranks];
	}
	/* read or given up: the ranks may overwrite their segments */
	N->lent = 0;
	__atomic_store_n(&N->own->consumed, N->seq, __ATOMIC_RELEASE);
	return rc;
}

/* rank 0, without a copy: the records where they lie, as runs of one address range (v->base + v->off[r], v->n[r] records, rank
 * order).  The ranks' segments stay untouched until rank 0 begins its next search (bh_node_begin) or collects.  Returns
 * BH_E_CAPACITY -- nothing consumed, call bh_node_collect -- when a rank's records do not lie inside its slot (a search that
 * outgrew its segment, or a rank with a segment of another size). */
int bh_node_collect_view(BhNode *N, BhRunView *v, uint64_t *counts) {
	uint64_t at[BH_MAX_RANKS + 1];
	int rc = wait_all(N, at, counts);
	if (rc) { if (rc != BH_E_USAGE) { N->lent = 0; __atomic_store_n(&N->own->consumed, N->seq, __ATOMIC_RELEASE); } return rc; }
	for (int r = 0; r < N->n_ranks; ++r) if (!N->view || !N->in_view[r] || N->peer[r]->n_hits > N->peer[r]->cap || N->peer[r]->cap * sizeof(BhipHit) > N->stride) return BH_E_CAPACITY;
	v->base = (const BhipHit *)N->view; v->n_runs = N->n_ranks; v->total = at[N->n_ranks];
	for (int r = 0; r < N->n_ranks; ++r) { v->off[r] = (uint64_t)r * (N->stride / sizeof(BhipHit)); v->n[r] = N->peer[r]->n_hits; }
	N->lent = 1;
	return BH_OK;
}
