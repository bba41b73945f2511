/*
 * walk_stats.c -- analysis build of the CPU oracle (test infrastructure, like the oracle itself): renders one dispatch
 * and records, for every pixel's first sample, how many traversal steps (interior-node visits + triangle tests) each
 * ray of the path took, by bounce index.  Compiled by tests/analysis/walk_stats.py together with oracle/pt_oracle.c
 * (included below, so that its static functions and the PTO_RAY_HOOK point in ray_color are reachable).
 */
#include <stdint.h>
#define WS_LEVELS 32
struct ws_tls { uint64_t last; uint16_t steps[WS_LEVELS]; uint32_t n; };
static __thread struct ws_tls ws;
#define PTO_RAY_HOOK(cx, i) ws_hook(cx, i)
struct ctx_fwd;
static void ws_hook(void *cx, uint32_t level);
#include "../../oracle/pt_oracle.c"

static void ws_hook(void *cxv, uint32_t level)
{
    ctx_t *cx = (ctx_t *)cxv;
    const uint64_t now = cx->st.interior + cx->st.tri_tests;
    const uint64_t d = now - ws.last;
    ws.last = now;
    if (level < WS_LEVELS) { ws.steps[level] = d > 65535u ? 65535u : (uint16_t)d; if (level + 1u > ws.n) ws.n = level + 1u; }
}

typedef struct { const pto_scene *sc; frame_t fr; uint32_t time; int tid, nthreads; uint16_t *out; float *accum; pto_stats st; } ws_job;

static void *ws_worker(void *arg)
{
    ws_job *j = (ws_job *)arg;
    ctx_t cx; memset(&cx, 0, sizeof cx);
    cx.sc = j->sc; cx.t_min = j->sc->settings->t_min; cx.t_max = j->sc->settings->t_max;
    for (uint32_t y = (uint32_t)j->tid; y < j->sc->height; y += (uint32_t)j->nthreads)
        for (uint32_t x = 0; x < j->sc->width; x++) {
            memset(&ws, 0, sizeof ws);
            ws.last = cx.st.interior + cx.st.tri_tests;
            render_pixel(&cx, &j->fr, x, y, j->time, j->accum, 0, 0);
            memcpy(j->out + ((size_t)y * j->sc->width + x) * WS_LEVELS, ws.steps, sizeof ws.steps);
        }
    j->st = cx.st;
    return 0;
}

/* out: height x width x WS_LEVELS uint16 (steps of the ray at each bounce index; 0 where the path had no such ray or
 * the ray ended at the spheres / the root's child boxes... a root visit counts as one step). */
int ws_render(const pto_scene *sc, uint32_t time, uint16_t *out, float *accum, pto_stats *stats, int nthreads)
{
    ws_job jobs[64]; pthread_t th[64];
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    const frame_t fr = camera_frame(sc->camera, sc->width, sc->height);
    for (int t = 0; t < nthreads; t++) {
        jobs[t].sc = sc; jobs[t].fr = fr; jobs[t].time = time; jobs[t].tid = t; jobs[t].nthreads = nthreads; jobs[t].out = out; jobs[t].accum = accum;
    }
    for (int t = 0; t < nthreads; t++) if (pthread_create(&th[t], 0, ws_worker, &jobs[t])) return -1;
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], 0);
    if (stats) for (int t = 0; t < nthreads; t++) stats_add(stats, &jobs[t].st);
    return 0;
}
