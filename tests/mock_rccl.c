/*
 * tests/mock_rccl.c -- TEST INFRASTRUCTURE: a stand-in for librccl.so on a box without GPUs, so that the C exchange
 * (cmusphinx_amd/csrc/s3a_gather.hip: counts, padded headers, padded words, ordering by utterance index) runs with
 * world > 1 in the CPU test suite.  Built by tests/test_gather_mock.py into a temporary directory as "librccl.so" and
 * found by the library's dlopen through LD_LIBRARY_PATH.  Exports s3a_comm_takes_host_pointers: the library then
 * stages in host memory.  The all-gather goes through files in $MOCK_RCCL_DIR (write + rename, poll).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

const int s3a_comm_takes_host_pointers = 1;

typedef struct { char internal[128]; } id_t128;
typedef struct { int rank, world, seq; char tag[40]; } comm_t;

int ncclGetUniqueId(id_t128 *id)
{
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "mock%ld_%ld", (long)getpid(), (long)random());
    return 0;
}

int ncclCommInitRank(void **comm, int world, id_t128 id, int rank)
{
    comm_t *c = calloc(1, sizeof *c);
    c->rank = rank; c->world = world;
    snprintf(c->tag, sizeof c->tag, "%.39s", id.internal);
    *comm = c;
    return 0;
}

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, void *stream)
{
    comm_t *c = comm;
    const char *dir = getenv("MOCK_RCCL_DIR");
    char path[4096], tmp[4200];
    (void)dtype; (void)stream;
    if (!dir) return 1;
    snprintf(path, sizeof path, "%s/%s.%d.%d", dir, c->tag, c->seq, c->rank);
    snprintf(tmp, sizeof tmp, "%s.tmp", path);
    FILE *fp = fopen(tmp, "wb");
    if (!fp || (count && fwrite(send, 1, count, fp) != count) || fclose(fp) != 0 || rename(tmp, path) != 0) return 2;
    for (int r = 0; r < c->world; r++) {
        int tries = 0;
        snprintf(path, sizeof path, "%s/%s.%d.%d", dir, c->tag, c->seq, r);
        for (;; tries++) {
            fp = fopen(path, "rb");
            if (fp) {
                size_t k = count ? fread((char *)recv + (size_t)r * count, 1, count, fp) : 0;
                fclose(fp);
                if (k == count) break;
            }
            if (tries > 3000) return 3;
            usleep(10000);
        }
    }
    c->seq++;
    return 0;
}

int ncclCommDestroy(void *comm) { free(comm); return 0; }
const char *ncclGetErrorString(int r) { return r == 1 ? "MOCK_RCCL_DIR not set" : r == 2 ? "cannot write" : r == 3 ? "peer timed out" : "ok"; }
