/* sweep_model.c -- timing model of the chunked natural-order sweep (design tool, CPU only).
 *
 * Rows [c K, (c+1) K) of the processing order form chunk c; chunks are taken in order by W waves (each takes the next chunk
 * when it is free); a wave solves its chunk's rows one after the other.  A row can start when its predecessor in the chunk is
 * done and every operand is there: operands of the same chunk cost nothing extra, operands of other chunks arrive h after
 * they were produced (the hand-off through memory).  A row of len entries takes t0 + t1 * len.
 *   gcc -O2 -shared -fPIC -o tools/libsweepmodel.so tools/sweep_model.c
 */
#include <stdint.h>
#include <stdlib.h>

typedef struct { double t; int w; } slot;

static void sift_down(slot *h, int n, int i) {
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && h[l].t < h[m].t) m = l;
        if (r < n && h[r].t < h[m].t) m = r;
        if (m == i) return;
        slot s = h[i]; h[i] = h[m]; h[m] = s; i = m;
    }
}

/* rp/col: strictly lower triangular in the PROCESSING order (operands have smaller indices).  chunk_of[r] ascending.
 * out[0] = makespan, out[1] = cross-chunk hops on the critical path, out[2] = in-chunk hops on it, out[3] = sum of row times */
void sweep_model(int64_t n, const int64_t *rp, const int32_t *col, const int32_t *chunk_of, int W, double h, double t0, double t1,
                 double t_chunk, double *out, double *finish) {
    slot *heap = (slot *)malloc(sizeof(slot) * (size_t)W);
    int32_t *crit = (int32_t *)malloc(sizeof(int32_t) * (size_t)n); /* the operand (or predecessor) that determined the start */
    for (int i = 0; i < W; ++i) { heap[i].t = 0.0; heap[i].w = i; }
    double wave_t = 0.0, total = 0.0, makespan = 0.0;
    int64_t last = -1;
    for (int64_t r = 0; r < n; ++r) {
        if (r == 0 || chunk_of[r] != chunk_of[r - 1]) { /* new chunk: give the finished wave back, take the earliest-free one */
            if (r > 0) { heap[0].t = wave_t; sift_down(heap, W, 0); }
            wave_t = heap[0].t + t_chunk;
        }
        double start = wave_t;
        int32_t c = -1; /* -1: the wave itself (predecessor / chunk start) */
        for (int64_t k = rp[r]; k < rp[r + 1]; ++k) {
            const int32_t d = col[k];
            const double a = finish[d] + (chunk_of[d] == chunk_of[r] ? 0.0 : h);
            if (a > start) { start = a; c = d; }
        }
        const double tr = t0 + t1 * (double)(rp[r + 1] - rp[r]);
        finish[r] = start + tr;
        total += tr;
        crit[r] = c;
        wave_t = finish[r];
        if (finish[r] > makespan) { makespan = finish[r]; last = r; }
    }
    /* walk the critical path back */
    double cross = 0, inchunk = 0;
    for (int64_t r = last; r >= 0;) {
        const int32_t c = crit[r];
        if (c < 0) { /* waited for the wave: predecessor row of the chunk, or the chunk's start */
            if (r > 0 && chunk_of[r - 1] == chunk_of[r]) { inchunk += 1; r = r - 1; } else break;
        } else {
            if (chunk_of[c] == chunk_of[r]) inchunk += 1; else cross += 1;
            r = c;
        }
    }
    out[0] = makespan; out[1] = cross; out[2] = inchunk; out[3] = total;
    free(heap); free(crit);
}

/* dependency levels (longest path) */
int64_t sweep_levels(int64_t n, const int64_t *rp, const int32_t *col, int32_t *level) {
    int64_t mx = 0;
    for (int64_t r = 0; r < n; ++r) {
        int32_t l = 0;
        for (int64_t k = rp[r]; k < rp[r + 1]; ++k) if (level[col[k]] + 1 > l) l = level[col[k]] + 1;
        level[r] = l;
        if (l > mx) mx = l;
    }
    return n ? mx + 1 : 0;
}
