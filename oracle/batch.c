/*
 * batch.c -- threaded batch drivers over the CPU restatement (oracle; test
 * infrastructure only, see rsx_oracle.h).  Mirrors src/batch.rs: one task per
 * buffer, per-worker reusable compressor state (rayon map_init, batch.rs:36-38
 * and :82-84), failures isolated per item (batch.rs:52-54, :95-97).
 */
#include "rsx_oracle.h"

#include <pthread.h>
#include <stdlib.h>

typedef struct {
    int compress, format, level;
    const uint8_t *in;
    const uint64_t *in_off;
    size_t n;
    uint8_t *out;
    const uint64_t *out_off;
    uint64_t *out_sizes;
    int32_t *status;
    size_t next; /* shared work counter */
} job;

static void *worker(void *arg)
{
    job *j = arg;
    orc_compressor *c = j->compress ? orc_compressor_new(j->level) : NULL;
    for (;;) {
        size_t i = __atomic_fetch_add(&j->next, 1, __ATOMIC_RELAXED);
        if (i >= j->n) break;
        const uint8_t *src = j->in + j->in_off[i];
        size_t slen = j->in_off[i + 1] - j->in_off[i];
        uint8_t *dst = j->out + j->out_off[i];
        size_t cap = j->out_off[i + 1] - j->out_off[i];
        size_t produced = 0;
        int st;
        if (j->compress) {
            st = c ? orc_compress(c, j->format, src, slen, dst, cap, &produced) : 1;
        } else {
            size_t consumed;
            st = orc_decompress(j->format, src, slen, dst, cap, &consumed, &produced);
        }
        j->status[i] = st;
        j->out_sizes[i] = st == 0 ? produced : 0;
    }
    orc_compressor_free(c);
    return NULL;
}

static void run(job *j, int threads)
{
    if (threads < 1) threads = 1;
    if ((size_t)threads > j->n) threads = j->n ? (int)j->n : 1;
    pthread_t *t = malloc(sizeof(pthread_t) * (size_t)threads);
    int started = 0;
    for (int k = 1; k < threads; k++)
        if (pthread_create(&t[started], NULL, worker, j) == 0) started++;
    worker(j);
    for (int k = 0; k < started; k++) pthread_join(t[k], NULL);
    free(t);
}

void orc_batch_compress(int format, int level, const uint8_t *in, const uint64_t *in_off,
                        size_t n, uint8_t *out, const uint64_t *out_off, uint64_t *out_sizes,
                        int32_t *status, int threads)
{
    job j = {1, format, level, in, in_off, n, out, out_off, out_sizes, status, 0};
    run(&j, threads);
}

void orc_batch_decompress(int format, const uint8_t *in, const uint64_t *in_off, size_t n,
                          uint8_t *out, const uint64_t *out_off, uint64_t *out_sizes,
                          int32_t *status, int threads)
{
    job j = {0, format, 0, in, in_off, n, out, out_off, out_sizes, status, 0};
    run(&j, threads);
}
