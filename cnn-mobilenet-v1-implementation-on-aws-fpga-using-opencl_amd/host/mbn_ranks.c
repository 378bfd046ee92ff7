/*
 * mbn_ranks.c — one host thread per rank (GPU) with a start gate and a cancellable barrier: the thread skeleton of
 * `mobilenet --gpus G` (the reference drives exactly one device from one thread, MobileNet.c:155).
 *
 * Why not bare pthread_create + pthread_barrier: if creating the thread of rank r > 0 fails, the ranks already started
 * would sit in pthread_barrier_wait (count = G) for ever while main returns. Here every rank first waits at a
 * gate that opens only after ALL threads exist; a failed creation closes it with "cancelled", the started ranks return
 * without having run the job, and the caller gets an error code. The barrier a job uses (mbn_rank_barrier) belongs to the
 * same object and is released early when a rank has failed, so a job that bails out cannot strand its peers either.
 * No GPU code: lives in libmbn_host.so too, and the skeleton is tested on the CPU with fake jobs.
 */
#include <pthread.h>
#include <stdlib.h>

#include "mbn.h"

struct mbn_rank_sync {
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int n;                  /* ranks */
    int gate;               /* 0 = closed, 1 = open, -1 = cancelled */
    int waiting;            /* ranks inside the current barrier episode */
    unsigned long episode;
    int failed;             /* a rank left with an error: barriers no longer block */
};

typedef struct {
    struct mbn_rank_sync *s;
    mbn_rank_fn fn;
    void *arg;
    int rank, rc, ran;
} rank_thread;

int mbn_rank_barrier(mbn_rank_sync *s)
{
    if (!s) return MBN_EINVAL;
    pthread_mutex_lock(&s->mu);
    if (s->failed) { pthread_mutex_unlock(&s->mu); return MBN_EDEVICE; }
    const unsigned long ep = s->episode;
    if (++s->waiting == s->n) {
        s->waiting = 0;
        s->episode++;
        pthread_cond_broadcast(&s->cv);
    } else {
        while (s->episode == ep && !s->failed) pthread_cond_wait(&s->cv, &s->mu);
    }
    const int rc = s->failed ? MBN_EDEVICE : MBN_OK;
    pthread_mutex_unlock(&s->mu);
    return rc;
}

int mbn_rank_fail(mbn_rank_sync *s)
{
    if (!s) return MBN_EINVAL;
    pthread_mutex_lock(&s->mu);
    s->failed = 1;
    pthread_cond_broadcast(&s->cv);
    pthread_mutex_unlock(&s->mu);
    return MBN_OK;
}

static void *rank_main(void *p)
{
    rank_thread *t = (rank_thread *)p;
    struct mbn_rank_sync *s = t->s;
    pthread_mutex_lock(&s->mu);
    while (s->gate == 0) pthread_cond_wait(&s->cv, &s->mu);
    const int go = s->gate == 1;
    pthread_mutex_unlock(&s->mu);
    if (!go) return NULL;                               /* a sibling could not be created: nobody runs */
    t->ran = 1;
    t->rc = t->fn(t->rank, t->arg, s);
    if (t->rc != MBN_OK) mbn_rank_fail(s);              /* peers waiting in a barrier are released with an error */
    return NULL;
}

int mbn_run_ranks(int n, mbn_rank_fn fn, void *arg, int fail_create_at, int *rank_rc)
{
    if (n < 1 || n > 64 || !fn) return MBN_EINVAL;
    struct mbn_rank_sync s;
    pthread_mutex_init(&s.mu, NULL);
    pthread_cond_init(&s.cv, NULL);
    s.n = n; s.gate = 0; s.waiting = 0; s.episode = 0; s.failed = 0;
    rank_thread *t = (rank_thread *)calloc((size_t)n, sizeof(*t));
    pthread_t *th = (pthread_t *)calloc((size_t)n, sizeof(*th));
    if (!t || !th) { free(t); free(th); return MBN_ENOMEM; }
    int started = 0, rc = MBN_OK;
    for (int r = 0; r < n; r++) {
        t[r].s = &s; t[r].fn = fn; t[r].arg = arg; t[r].rank = r; t[r].rc = MBN_OK; t[r].ran = 0;
        if (r == fail_create_at || pthread_create(&th[r], NULL, rank_main, &t[r]) != 0) { rc = MBN_ENOMEM; break; }
        started++;
    }
    pthread_mutex_lock(&s.mu);
    s.gate = rc == MBN_OK ? 1 : -1;
    pthread_cond_broadcast(&s.cv);
    pthread_mutex_unlock(&s.mu);
    for (int r = 0; r < started; r++) pthread_join(th[r], NULL);
    for (int r = 0; r < n; r++) {
        if (rank_rc) rank_rc[r] = r < started && t[r].ran ? t[r].rc : MBN_EUNSUPPORTED;      /* EUNSUPPORTED = the job did not run */
        if (rc == MBN_OK && t[r].rc != MBN_OK) rc = t[r].rc;
    }
    free(t); free(th);
    pthread_cond_destroy(&s.cv);
    pthread_mutex_destroy(&s.mu);
    return rc;
}
