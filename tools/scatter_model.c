// CPU replay of the hash-grid scatter's flush logic: counts memory-side 64-byte atomic requests per
// level for several accumulation policies on captured sample positions (tools/scatter_model.py).
// Not product code and not part of the oracle: an analysis tool for DESIGN.md section 4.
//
// A request = one 64-byte line touched by one wave-level atomic instruction.  A table row is F=8
// floats = 32 bytes, so a line holds the rows (2k, 2k+1) of the level (level offsets are multiples
// of 8 rows).
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint32_t size, res, hashed, pow2;
    float scale;
} level_t;

static uint32_t row_index(const level_t* li, uint32_t x, uint32_t y, uint32_t z)
{
    uint32_t idx;
    if (li->hashed) {
        idx = x ^ (y * 2654435761u) ^ (z * 805459861u);
        idx = li->pow2 ? (idx & (li->size - 1u)) : (idx % li->size);
    } else {
        idx = x + y * li->res + z * li->res * li->res;
        if (idx >= li->size) idx %= li->size;
    }
    return idx;
}

void model_layout(int n_levels, int log2_T, int base_res, double per_level_scale, level_t* out)
{
    const float l2 = log2f((float)per_level_scale);
    for (int l = 0; l < n_levels; l++) {
        const float sc = exp2f(l * l2) * base_res - 1.0f;
        const uint32_t res = (uint32_t)ceilf(sc) + 1;
        const uint32_t cap = 1u << log2_T;
        const uint64_t dense = (uint64_t)res * res * res;
        uint32_t p = dense > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)dense;
        p = (p + 7u) / 8u * 8u;
        if (p > cap) p = cap;
        uint64_t stride = 1;
        for (int k = 0; k < 3 && stride <= p; k++) stride *= res;
        out[l].size = p; out[l].res = res; out[l].scale = sc;
        out[l].hashed = p < stride; out[l].pow2 = (p & (p - 1)) == 0;
    }
}

// ---- open-corner bookkeeping ---------------------------------------------------------------
typedef struct { int x, y, z; int dirty; } corner_t;
#define MAXOPEN 64

typedef struct {
    corner_t c[MAXOPEN];
    int n;
} open_t;

static corner_t* find(open_t* o, int x, int y, int z)
{
    for (int i = 0; i < o->n; i++)
        if (o->c[i].x == x && o->c[i].y == y && o->c[i].z == z) return &o->c[i];
    return 0;
}

static void touch(open_t* o, int x, int y, int z, int dirty)
{
    corner_t* c = find(o, x, y, z);
    if (!c) {
        if (o->n >= MAXOPEN) abort();
        c = &o->c[o->n++];
        c->x = x; c->y = y; c->z = z; c->dirty = 0;
    }
    c->dirty |= dirty;
}

// flushes every open corner for which keep() is false; rows flushed "together" (same call) that
// share a 64-byte line count as one request.  `pair_all`: 1 = all rows of this flush call may
// share lines (they leave in the same wave-instruction group); 0 = every dirty row is its own request
static int64_t flush_where(open_t* o, const level_t* li, int (*keep)(const corner_t*, const int*), const int* arg,
                           int pair_all, int64_t* rows_out)
{
    uint32_t lines[MAXOPEN];
    int nl = 0;
    int64_t req = 0;
    int w = 0;
    for (int i = 0; i < o->n; i++) {
        corner_t c = o->c[i];
        if (keep && keep(&c, arg)) { o->c[w++] = c; continue; }
        if (!c.dirty) continue;
        (*rows_out)++;
        const uint32_t line = row_index(li, (uint32_t)c.x, (uint32_t)c.y, (uint32_t)c.z) >> 1;
        if (pair_all) {
            int seen = 0;
            for (int k = 0; k < nl; k++) if (lines[k] == line) { seen = 1; break; }
            if (!seen) { lines[nl++] = line; req++; }
        } else {
            req++;
        }
    }
    o->n = w;
    return req;
}

static int keep_window(const corner_t* c, const int* g)   // exact 2x2x2 window at base g
{
    return c->x >= g[0] && c->x <= g[0] + 1 && c->y >= g[1] && c->y <= g[1] + 1 && c->z >= g[2] && c->z <= g[2] + 1;
}

// policy 0: the shipped sliding-window kernel (grid_bwd_param_slide_kernel): y/z-leaving corners go out
//           as x-pairs (one request when the pair shares a line), x-leaving rows go out alone
// policy 1: line-aligned window: a row stays open while ITS 64-byte line still intersects the 2x2x2
//           window of the current sample; a line is flushed once, whole
// policy 2: perfect merging inside a chunk (distinct dirty lines per chunk): the floor for the chunk size
// counts[l*2+0] += requests, counts[l*2+1] += dirty rows flushed
void model_run(const float* x, const uint8_t* nz /* (n, L) 0/1 */, int64_t n, int n_levels, const level_t* lv,
               int chunk, int policy, int64_t* counts)
{
    for (int l = 0; l < n_levels; l++) {
        const level_t* li = &lv[l];
        int64_t req = 0, rows = 0;
        for (int64_t s0 = 0; s0 < n; s0 += chunk) {
            const int64_t s1 = s0 + chunk < n ? s0 + chunk : n;
            open_t o; o.n = 0;
            int b[3] = {0, 0, 0}, have = 0;
            // policy 2 bookkeeping: hash set of dirty lines of this chunk
            uint32_t* set = 0; int cap = 0, used = 0;
            if (policy == 2) { cap = 1; while (cap < (int)(s1 - s0) * 16) cap <<= 1; set = (uint32_t*)malloc(cap * 4); memset(set, 0xff, cap * 4); }
            for (int64_t s = s0; s < s1; s++) {
                int g[3]; float w[3];
                for (int k = 0; k < 3; k++) {
                    const float p = fmaf(li->scale, x[3 * s + k], 0.5f);
                    const float fl = floorf(p);
                    g[k] = (int)fl; w[k] = p - fl;
                }
                const int d = nz[s * n_levels + l];
                if (policy == 2) {
                    if (d) for (int c = 0; c < 8; c++) {
                        const float wt = ((c & 1) ? w[0] : 1 - w[0]) * ((c & 2) ? w[1] : 1 - w[1]) * ((c & 4) ? w[2] : 1 - w[2]);
                        if (wt == 0.0f) continue;
                        const uint32_t line = row_index(li, g[0] + (c & 1), g[1] + ((c >> 1) & 1), g[2] + ((c >> 2) & 1)) >> 1;
                        uint32_t h = (line * 2654435761u) & (cap - 1);
                        while (set[h] != 0xffffffffu && set[h] != line) h = (h + 1) & (cap - 1);
                        if (set[h] == 0xffffffffu) { set[h] = line; used++; }
                    }
                    continue;
                }
                if (have && (g[0] != b[0] || g[1] != b[1] || g[2] != b[2])) {
                    if (policy == 0) {
                        const int dx = g[0] - b[0], dy = g[1] - b[1], dz = g[2] - b[2];
                        const int nearm = dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1 && dz >= -1 && dz <= 1;
                        if (!nearm) {
                            req += flush_where(&o, li, 0, 0, 1, &rows);
                        } else {
                            // 1) corners leaving in y / z (both x lanes together: pairs may share a line)
                            int gy[3] = { b[0], g[1], g[2] };
                            req += flush_where(&o, li, keep_window, gy, 1, &rows);
                            // 2) the x row that leaves goes out alone
                            if (dx != 0) req += flush_where(&o, li, keep_window, g, 0, &rows);
                        }
                    } else {
                        // keep a corner while its line (x pair per (y,z), in ROW-INDEX space) still has a row inside the new window
                        int w2 = 0;
                        uint32_t lines[MAXOPEN]; int nl = 0;
                        for (int i = 0; i < o.n; i++) {
                            corner_t c = o.c[i];
                            int keepit = 0;
                            if (c.y >= g[1] && c.y <= g[1] + 1 && c.z >= g[2] && c.z <= g[2] + 1) {
                                const uint32_t line = row_index(li, c.x, c.y, c.z) >> 1;
                                for (int xx = g[0]; xx <= g[0] + 1 && !keepit; xx++)
                                    if ((row_index(li, xx, c.y, c.z) >> 1) == line) keepit = 1;
                            }
                            if (keepit) { o.c[w2++] = c; continue; }
                            if (!c.dirty) continue;
                            rows++;
                            const uint32_t line = row_index(li, c.x, c.y, c.z) >> 1;
                            int seen = 0;
                            for (int k = 0; k < nl; k++) if (lines[k] == line) { seen = 1; break; }
                            if (!seen) { lines[nl++] = line; req++; }
                        }
                        o.n = w2;
                    }
                }
                b[0] = g[0]; b[1] = g[1]; b[2] = g[2]; have = 1;
                for (int c = 0; c < 8; c++) {
                    const float wt = ((c & 1) ? w[0] : 1 - w[0]) * ((c & 2) ? w[1] : 1 - w[1]) * ((c & 4) ? w[2] : 1 - w[2]);
                    touch(&o, g[0] + (c & 1), g[1] + ((c >> 1) & 1), g[2] + ((c >> 2) & 1), d && wt != 0.0f);
                }
            }
            if (policy == 2) { req += used; free(set); }
            else req += flush_where(&o, li, 0, 0, 1, &rows);
        }
        counts[2 * l] += req; counts[2 * l + 1] += rows;
    }
}
