/*
 * track2d_oracle.c — TEST INFRASTRUCTURE ONLY (see track2d_oracle.h).
 *
 * Scalar C restatement of the reference gym-track2d environment. Every function names the
 * reference lines it follows (paths relative to /root/reference; G/ = envs/gym-track2d/gym_track2d/).
 * Compile with -ffp-contract=off: reward arithmetic must round exactly like CPython's float64.
 */
#include "track2d_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* numpy legacy RandomState primitives (MT19937). numpy/random/src/mt19937 + legacy-distributions
 * are a third-party dependency of the reference (requirements.txt:2, numpy); the stream is frozen
 * by NEP 19 and is checked against the installed numpy in tests/test_oracle_rng.py.               */
/* ------------------------------------------------------------------------------------------ */
struct orc_mt {
    uint32_t key[624];
    int pos;
};

static void mt_seed(orc_mt *m, uint32_t seed)
{
    for (int i = 0; i < 624; i++) {
        m->key[i] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
    }
    m->pos = 624;
}

static void mt_gen(orc_mt *m)
{
    const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MAT = 0x9908b0dfu;
    uint32_t *k = m->key, y;
    int i;
    for (i = 0; i < 624 - 397; i++) {
        y = (k[i] & UPPER) | (k[i + 1] & LOWER);
        k[i] = k[i + 397] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
    }
    for (; i < 623; i++) {
        y = (k[i] & UPPER) | (k[i + 1] & LOWER);
        k[i] = k[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
    }
    y = (k[623] & UPPER) | (k[0] & LOWER);
    k[623] = k[396] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
    m->pos = 0;
}

static uint32_t mt_u32(orc_mt *m)
{
    if (m->pos == 624) mt_gen(m);
    uint32_t y = m->key[m->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

orc_mt *orc_mt_new(uint32_t seed)
{
    orc_mt *m = (orc_mt *)malloc(sizeof(orc_mt));
    mt_seed(m, seed);
    return m;
}
void orc_mt_free(orc_mt *m) { free(m); }
uint32_t orc_mt_u32(orc_mt *m) { return mt_u32(m); }

/* ------------------------------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al., SC'11) — the device random source.                            */
/* ------------------------------------------------------------------------------------------ */
void orc_philox4x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                    uint32_t out[4])
{
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Keyed permutation of Z80 x Z80 = [0,6400): 8-round Feistel, round function = murmur3 fmix32
 * reduced to [0,80) by multiply-high. Device spec for "first K of a random permutation"
 * (the semantics of np.random.choice(6400, K, replace=False), G/envs/generators.py:166). */
static inline uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
uint32_t orc_perm6400(const uint32_t rk[8], uint32_t i)
{
    uint32_t a = i / 80u, b = i % 80u;
    for (int r = 0; r < 8; r++) {
        uint32_t f = (uint32_t)(((uint64_t)fmix32(b * 0x9E3779B1u + rk[r]) * 80u) >> 32);
        uint32_t t = a + f;
        if (t >= 80u) t -= 80u;
        a = b; b = t;
    }
    return a * 80u + b;
}

/* ------------------------------------------------------------------------------------------ */
/* Environment                                                                                  */
/* ------------------------------------------------------------------------------------------ */
enum { STREAM_MAP = 0, STREAM_SPAWN = 1, STREAM_TARGET = 2, NUM_STREAMS = 3 };
#define PLAN_CAP 8192

struct orc_env {
    int map_type, target_mode, level, max_steps, rng_mode;
    /* random sources */
    orc_mt mt;
    uint32_t k0, k1, env_id, episode;
    uint32_t ctr[NUM_STREAMS];          /* next word index per stream (philox) */
    uint32_t cache[NUM_STREAMS][4];     /* last generated block per stream     */
    uint32_t cache_blk[NUM_STREAMS];
    /* map + state */
    int side;
    uint8_t maze[ORC_MAX_SIDE * ORC_MAX_SIDE];
    int pos[2][2];
    int goal[2][2];
    int nav_goal[2];
    int c_far, t;
    double w_p;
    int64_t d2;
    /* scripted target */
    int plan[PLAN_CAP];
    int plan_len, plan_cur;
    uint8_t dir[ORC_MAX_SIDE * ORC_MAX_SIDE]; /* philox-mode Nav direction field */
    int nav_planb;                            /* philox-mode Nav: following the 10 random actions */
    /* RPF (static goals, generators.py:12-19): the generator's maze has the four patrol cells cleared AFTER the env
     * copied it (track_1v1.py:233-236), so spawning/planning use gmaze while moves and observations use maze. */
    uint8_t gmaze[ORC_MAX_SIDE * ORC_MAX_SIDE];
    int cand[4][2], vector;
    int vpos[2], remaining;                   /* philox-mode RPF: open-loop plan = field descent of a virtual position */
    int obs_full;                             /* obs_type: 0 'Partial' (13x13 crops), 1 'Full' (whole map) */
    int moore;                                /* action_type: 0 'VonNeumann' (4 actions), 1 'Moore' (8) — track_1v1.py:243-249 */
};

static uint32_t next_u32(orc_env *e, int stream)
{
    if (e->rng_mode == ORC_RNG_NP) return mt_u32(&e->mt);
    uint32_t i = e->ctr[stream]++;
    uint32_t blk = i >> 2;
    if (e->cache_blk[stream] != blk) {
        orc_philox4x32(e->k0, e->k1, blk, e->episode, e->env_id, (uint32_t)stream, e->cache[stream]);
        e->cache_blk[stream] = blk;
    }
    return e->cache[stream][i & 3u];
}

/* legacy random_sample: 53-bit double from two 32-bit words. */
static double next_double(orc_env *e, int stream)
{
    uint32_t a = next_u32(e, stream) >> 5, b = next_u32(e, stream) >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
}
double orc_mt_double(orc_mt *m)
{
    uint32_t a = mt_u32(m) >> 5, b = mt_u32(m) >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

static inline uint32_t mask_of(uint32_t max)
{
    uint32_t mask = max;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    return mask;
}

/* uniform integer in [0, max] by masked rejection: legacy random_interval / the masked path of
 * RandomState.randint (one 32-bit word per attempt; max == 0 draws nothing). */
static uint32_t bounded(orc_env *e, int stream, uint32_t max)
{
    if (max == 0) return 0;
    uint32_t mask = mask_of(max), v;
    while ((v = next_u32(e, stream) & mask) > max) {}
    return v;
}
uint32_t orc_mt_interval(orc_mt *m, uint32_t max)
{
    if (max == 0) return 0;
    uint32_t mask = mask_of(max), v;
    while ((v = mt_u32(m) & mask) > max) {}
    return v;
}
void orc_mt_permutation(orc_mt *m, int n, int32_t *out)
{
    for (int i = 0; i < n; i++) out[i] = i;
    for (int i = n - 1; i >= 1; i--) {
        int j = (int)orc_mt_interval(m, (uint32_t)i);
        int32_t tmp = out[i]; out[i] = out[j]; out[j] = tmp;
    }
}

/* np.random.choice(n, size=k, replace=False) == permutation(n)[:k] (legacy RandomState): the full
 * Fisher-Yates shuffle is always executed, also for k == 0 (generators.py:28 via :73). */
static void np_choice_norep(orc_env *e, int n, int k, int32_t *out, int32_t *scratch)
{
    orc_mt_permutation(&e->mt, n, scratch);
    for (int i = 0; i < k; i++) out[i] = scratch[i];
}

/* ---- free-cell helpers: np.where(maze == 0) row-major order (generators.py:42-43) ---- */
/* the generator's maze: identical to the env's except in RPF mode */
static const uint8_t *gen_maze_of(const orc_env *e) { return e->target_mode == ORC_TGT_RPF ? e->gmaze : e->maze; }
static int count_free(const orc_env *e)
{
    int n = 0, S = e->side;
    const uint8_t *m = gen_maze_of(e);
    for (int i = 0; i < S * S; i++) n += (m[i] == 0);
    return n;
}
static void select_free(const orc_env *e, int k, int *rc)
{
    int S = e->side;
    const uint8_t *m = gen_maze_of(e);
    for (int i = 0; i < S * S; i++)
        if (m[i] == 0 && k-- == 0) { rc[0] = i / S; rc[1] = i % S; return; }
    rc[0] = rc[1] = -1;
}

/* MazeGenerator.sample_goal(num) — generators.py:38-51 (non-static branch). */
static void sample_goal(orc_env *e, int num, int out[][2], int32_t *scratch)
{
    int n = count_free(e);
    int32_t idx[2];
    if (e->rng_mode == ORC_RNG_NP) {
        np_choice_norep(e, n, num, idx, scratch);
    } else {
        idx[0] = (int32_t)bounded(e, STREAM_SPAWN, (uint32_t)(n - 1));
        if (num > 1) {
            idx[1] = (int32_t)bounded(e, STREAM_SPAWN, (uint32_t)(n - 2));
            if (idx[1] >= idx[0]) idx[1]++;
        }
    }
    for (int i = 0; i < num; i++) select_free(e, idx[i], out[i]);
}

/* MazeGenerator.get_around(state, 1) — generators.py:82-94: the slice [x0:x1, y0:y1] has an
 * exclusive upper bound, i.e. the 2x2 block {r-1,r} x {c-1,c} (clamped at the map edge). */
static void get_around(orc_env *e, const int *st, int *out, int32_t *scratch)
{
    int S = e->side;
    int x0 = st[0] - 1 < 0 ? 0 : st[0] - 1, x1 = st[0] + 1 > S - 1 ? S - 1 : st[0] + 1;
    int y0 = st[1] - 1 < 0 ? 0 : st[1] - 1, y1 = st[1] + 1 > S - 1 ? S - 1 : st[1] + 1;
    int cand[4][2], m = 0;
    const uint8_t *gm = gen_maze_of(e);
    for (int r = x0; r < x1; r++)
        for (int c = y0; c < y1; c++)
            if (gm[r * S + c] == 0) { cand[m][0] = r; cand[m][1] = c; m++; }
    int j;
    if (e->rng_mode == ORC_RNG_NP) {
        int32_t idx[1];
        np_choice_norep(e, m, 1, idx, scratch);
        j = idx[0];
    } else {
        j = (int)bounded(e, STREAM_SPAWN, (uint32_t)(m - 1));
    }
    out[0] = cand[j][0]; out[1] = cand[j][1];
}

/* MazeGenerator.sample_close_states(2, 1) — generators.py:53-77. Draws two indices, keeps the
 * first as the tracker spawn; target spawn from get_around; sample_state(0) still shuffles. */
static void sample_close_states(orc_env *e, int32_t *scratch)
{
    int n = count_free(e);
    if (e->rng_mode == ORC_RNG_NP) {
        int32_t idx[2];
        np_choice_norep(e, n, 2, idx, scratch);   /* drawn even when static (:61) */
        if (e->target_mode == ORC_TGT_RPF) { e->pos[0][0] = e->cand[0][0]; e->pos[0][1] = e->cand[0][1]; } /* :68 */
        else select_free(e, idx[0], e->pos[0]);
        get_around(e, e->pos[0], e->pos[1], scratch);
        np_choice_norep(e, n, 0, idx, scratch); /* sample_state(num-2 = 0): choice(n, 0) */
    } else {
        if (e->target_mode == ORC_TGT_RPF) { e->pos[0][0] = e->cand[0][0]; e->pos[0][1] = e->cand[0][1]; }
        else select_free(e, (int)bounded(e, STREAM_SPAWN, (uint32_t)(n - 1)), e->pos[0]);
        get_around(e, e->pos[0], e->pos[1], scratch);
    }
}

/* RandomBlockMazeGenerator._generate_maze — generators.py:157-176. */
static void gen_block(orc_env *e, double ratio, int32_t *scratch)
{
    const int M = 80, S = 82;
    e->side = S;
    memset(e->maze, 0, sizeof(e->maze));
    int K = (int)(ratio * (double)(M * M));
    if (e->rng_mode == ORC_RNG_NP) {
        orc_mt_permutation(&e->mt, M * M, scratch);
        for (int i = 0; i < K; i++) {
            int c = scratch[i];
            e->maze[(c / M + 1) * S + (c % M + 1)] = 1;
        }
    } else {
        uint32_t rk[8];
        e->ctr[STREAM_MAP] = 4; /* words 4..11 = blocks 1,2 of the MAP stream */
        for (int i = 0; i < 8; i++) rk[i] = next_u32(e, STREAM_MAP);
        for (int i = 0; i < K; i++) {
            uint32_t c = orc_perm6400(rk, (uint32_t)i);
            e->maze[(c / M + 1) * S + (c % M + 1)] = 1;
        }
    }
    for (int i = 0; i < S; i++) {
        e->maze[i] = e->maze[(S - 1) * S + i] = 1;
        e->maze[i * S] = e->maze[i * S + S - 1] = 1;
    }
}

/* RandomMazeGenerator._generate_maze — generators.py:115-145 (width = height = 80 -> 81x81). */
static void gen_maze(orc_env *e, double ratio)
{
    const int S = 81;
    e->side = S;
    memset(e->maze, 0, sizeof(e->maze));
    int complexity = (int)(ratio * (double)(5 * (S + S)));
    int density = (int)(ratio * (double)((S / 2) * (S / 2)));
    for (int i = 0; i < S; i++) {
        e->maze[i] = e->maze[(S - 1) * S + i] = 1;
        e->maze[i * S] = e->maze[i * S + S - 1] = 1;
    }
    for (int i = 0; i < density; i++) {
        int x = (int)bounded(e, STREAM_MAP, (uint32_t)(S / 2)) * 2;
        int y = (int)bounded(e, STREAM_MAP, (uint32_t)(S / 2)) * 2;
        e->maze[y * S + x] = 1;
        for (int j = 0; j < complexity; j++) {
            int nb[4][2], n = 0;
            if (x > 1)     { nb[n][0] = y;     nb[n][1] = x - 2; n++; }
            if (x < S - 2) { nb[n][0] = y;     nb[n][1] = x + 2; n++; }
            if (y > 1)     { nb[n][0] = y - 2; nb[n][1] = x;     n++; }
            if (y < S - 2) { nb[n][0] = y + 2; nb[n][1] = x;     n++; }
            if (n) {
                int k = (int)bounded(e, STREAM_MAP, (uint32_t)(n - 1));
                int y_ = nb[k][0], x_ = nb[k][1];
                if (e->maze[y_ * S + x_] == 0) {
                    e->maze[y_ * S + x_] = 1;
                    /* Python floor division: (y - y_) // 2 and (x - x_) // 2, values in {-1,0,1} */
                    int dy = (y - y_) / 2, dx = (x - x_) / 2;
                    e->maze[(y_ + dy) * S + (x_ + dx)] = 1;
                    x = x_; y = y_;
                }
            }
        }
    }
}

/* ---- heapq-faithful A* (Astar_solver.py:42-149) ---- */
typedef struct { int r, c, prev, action, g; } anode;
typedef struct { double f; int node; } hitem;

static inline int h_lt(const hitem *a, const hitem *b, const anode *nodes)
{
    /* list comparison [f, node] < [f', node'] with Node.__lt__ on path cost (:30-32,55) */
    if (a->f != b->f) return a->f < b->f;
    if (a->node == b->node) return 0;
    return nodes[a->node].g < nodes[b->node].g;
}
static void h_siftdown(hitem *heap, int startpos, int pos, const anode *nodes)
{
    hitem newitem = heap[pos];
    while (pos > startpos) {
        int parentpos = (pos - 1) >> 1;
        if (h_lt(&newitem, &heap[parentpos], nodes)) { heap[pos] = heap[parentpos]; pos = parentpos; continue; }
        break;
    }
    heap[pos] = newitem;
}
static void h_siftup(hitem *heap, int n, int pos, const anode *nodes)
{
    int startpos = pos, childpos = 2 * pos + 1;
    hitem newitem = heap[pos];
    while (childpos < n) {
        int rightpos = childpos + 1;
        if (rightpos < n && !h_lt(&heap[childpos], &heap[rightpos], nodes)) childpos = rightpos;
        heap[pos] = heap[childpos];
        pos = childpos;
        childpos = 2 * pos + 1;
    }
    heap[pos] = newitem;
    h_siftdown(heap, startpos, pos, nodes);
}

int orc_astar(int S, const uint8_t *maze, const int *start, const int *goal, int *actions, int cap)
{
    static const int DR[4] = {-1, 1, 0, 0}, DC[4] = {0, 0, -1, 1};
    int ncell = S * S, maxnodes = 4 * ncell + 8;
    anode *nodes = (anode *)malloc(sizeof(anode) * (size_t)maxnodes);
    hitem *heap = (hitem *)malloc(sizeof(hitem) * (size_t)(ncell + 8));
    int *in_front = (int *)malloc(sizeof(int) * (size_t)ncell);   /* state_nodes dict (:48) */
    uint8_t *explored = (uint8_t *)calloc((size_t)ncell, 1);
    for (int i = 0; i < ncell; i++) in_front[i] = -1;
    int nn = 0, hn = 0, result = -1;
#define HEUR(r_, c_) sqrt((double)(((r_) - goal[0]) * ((r_) - goal[0]) + ((c_) - goal[1]) * ((c_) - goal[1])))
    nodes[nn] = (anode){start[0], start[1], -1, -1, 0};
    heap[hn].f = 0.0 + HEUR(start[0], start[1]); heap[hn].node = nn; hn++;
    h_siftdown(heap, 0, hn - 1, nodes);
    in_front[start[0] * S + start[1]] = nn; nn++;
    while (hn > 0) {
        /* Frontier.pop (:58-63) = heapq.heappop */
        hitem last = heap[--hn], top = last;
        if (hn > 0) { top = heap[0]; heap[0] = last; h_siftup(heap, hn, 0, nodes); }
        int ni = top.node;
        anode nd = nodes[ni];
        in_front[nd.r * S + nd.c] = -1;
        if (nd.r == goal[0] && nd.c == goal[1]) { /* goal test (:133) */
            int len = 0;
            for (int k = ni; nodes[k].prev >= 0; k = nodes[k].prev) len++;
            result = len;
            if (len <= cap) {
                int w = len;
                for (int k = ni; nodes[k].prev >= 0; k = nodes[k].prev) actions[--w] = nodes[k].action;
            }
            break;
        }
        explored[nd.r * S + nd.c] = 1;
        for (int a = 0; a < 4; a++) {
            int r2 = nd.r + DR[a], c2 = nd.c + DC[a];
            if (maze[r2 * S + c2] == 1) { r2 = nd.r; c2 = nd.c; } /* bump: child state = parent (:170-171) */
            int cell = r2 * S + c2;
            int g2 = nd.g + 1;
            if (!explored[cell] && in_front[cell] < 0) {
                nodes[nn] = (anode){r2, c2, ni, a, g2};
                heap[hn].f = (double)g2 + HEUR(r2, c2); heap[hn].node = nn; hn++;
                h_siftdown(heap, 0, hn - 1, nodes);
                in_front[cell] = nn; nn++;
            } else if (in_front[cell] >= 0 && nodes[in_front[cell]].g < g2) {
                /* inverted replace test (:146-147) + Frontier.replace (:65-72) */
                nodes[nn] = (anode){r2, c2, ni, a, g2};
                for (int i = 0; i < hn; i++) {
                    const anode *o = &nodes[heap[i].node];
                    if (o->r == r2 && o->c == c2) {
                        heap[i].f = (double)g2 + HEUR(r2, c2); heap[i].node = nn;
                        h_siftdown(heap, 0, i, nodes);
                        in_front[cell] = nn;
                    }
                }
                nn++;
            }
        }
    }
#undef HEUR
    free(nodes); free(heap); free(in_front); free(explored);
    return result;
}

/* ---- BFS direction field: device spec for the Nav target ---- */
void orc_bfs_field(int S, const uint8_t *maze, const int *goal, uint8_t *dir, int32_t *dist_out)
{
    static const int DR[4] = {-1, 1, 0, 0}, DC[4] = {0, 0, -1, 1};
    int ncell = S * S;
    int32_t *dist = (int32_t *)malloc(sizeof(int32_t) * (size_t)ncell);
    int *queue = (int *)malloc(sizeof(int) * (size_t)ncell);
    for (int i = 0; i < ncell; i++) { dist[i] = -1; dir[i] = 255; }
    int qh = 0, qt = 0, g = goal[0] * S + goal[1];
    dist[g] = 0; dir[g] = 4; queue[qt++] = g;
    while (qh < qt) {
        int cell = queue[qh++], r = cell / S, c = cell % S;
        for (int a = 0; a < 4; a++) {
            int r2 = r + DR[a], c2 = c + DC[a];
            if (r2 < 0 || r2 >= S || c2 < 0 || c2 >= S) continue;
            int n2 = r2 * S + c2;
            if (maze[n2] == 1 || dist[n2] >= 0) continue;
            dist[n2] = dist[cell] + 1;
            queue[qt++] = n2;
        }
    }
    /* direction = first action (0 up, 1 down, 2 left, 3 right) reaching a cell one step closer */
    for (int cell = 0; cell < ncell; cell++) {
        if (dist[cell] <= 0) continue;
        int r = cell / S, c = cell % S;
        for (int a = 0; a < 4; a++) {
            int r2 = r + DR[a], c2 = c + DC[a];
            if (r2 < 0 || r2 >= S || c2 < 0 || c2 >= S) continue;
            if (dist[r2 * S + c2] == dist[cell] - 1) { dir[cell] = (uint8_t)a; break; }
        }
    }
    if (dist_out) memcpy(dist_out, dist, sizeof(int32_t) * (size_t)ncell);
    free(dist); free(queue);
}

/* ---- scripted targets ---- */

/* RamAgent.reset — navigator.py:90-93: randint(1,10) is evaluated before choice(all_actions, n); all_actions =
 * range(action_space.n) (navigator.py:74-75), i.e. 4 or, with action_type 'Moore', 8 actions. */
static uint32_t ram_amax(const orc_env *e) { return e->moore ? 7u : 3u; }
static void ram_reset(orc_env *e)
{
    int n = 1 + (int)bounded(e, STREAM_TARGET, 8);
    for (int i = 0; i < n; i++) e->plan[i] = (int)bounded(e, STREAM_TARGET, ram_amax(e));
    e->plan_len = n; e->plan_cur = 0;
}
/* RamAgent.step — navigator.py:77-88. */
static int ram_step(orc_env *e)
{
    int action = e->plan[e->plan_cur++];
    if (e->plan_cur >= e->plan_len) {
        if (bounded(e, STREAM_TARGET, 1) == 0) {
            action = (int)bounded(e, STREAM_TARGET, ram_amax(e));
            int n = 1 + (int)bounded(e, STREAM_TARGET, 8);
            for (int i = 0; i < n; i++) e->plan[i] = action;
            e->plan_len = n;
        } else {
            int n = 1 + (int)bounded(e, STREAM_TARGET, 8);
            for (int i = 0; i < n; i++) e->plan[i] = (int)bounded(e, STREAM_TARGET, ram_amax(e));
            e->plan_len = n;
        }
        e->plan_cur = 0;
    }
    return action;
}

/* sample_goal(1)[0] as used by the Navigator (navigator.py:17,28,56). */
static void nav_sample_goal(orc_env *e, int *g, int32_t *scratch)
{
    if (e->target_mode == ORC_TGT_RPF) { /* static: next patrol cell, no random draw (generators.py:48-50) */
        e->vector = (e->vector + 1) % 4;
        g[0] = e->cand[e->vector][0]; g[1] = e->cand[e->vector][1];
        return;
    }
    int n = count_free(e);
    int idx;
    if (e->rng_mode == ORC_RNG_NP) {
        int32_t i1[1];
        np_choice_norep(e, n, 1, i1, scratch);
        idx = i1[0];
    } else {
        idx = (int)bounded(e, STREAM_TARGET, (uint32_t)(n - 1));
    }
    select_free(e, idx, g);
}

/* Shared by Navigator.reset (navigator.py:43-63) and the re-plan branch of Navigator.step (:15-38):
 * plan from `from` to e->nav_goal; on failure/empty plan resample the goal, 6th failure -> plan B. */
static void nav_plan(orc_env *e, const int *from, int32_t *scratch)
{
    int count_res = 0, planb = 0;
    for (;;) {
        int ok;
        if (e->rng_mode == ORC_RNG_NP) {
            int len = orc_astar(e->side, gen_maze_of(e), from, e->nav_goal, e->plan, PLAN_CAP);
            ok = len >= 1;
            if (ok) e->plan_len = len;
        } else {
            int32_t *dist = (int32_t *)malloc(sizeof(int32_t) * ORC_MAX_SIDE * ORC_MAX_SIDE);
            orc_bfs_field(e->side, gen_maze_of(e), e->nav_goal, e->dir, dist);
            uint8_t d = e->dir[from[0] * e->side + from[1]];
            ok = d < 4; /* reachable and not already at the goal */
            if (ok) { e->vpos[0] = from[0]; e->vpos[1] = from[1]; e->remaining = dist[from[0] * e->side + from[1]]; }
            free(dist);
        }
        if (ok) break;
        if (++count_res > 5) { planb = 1; break; }
        nav_sample_goal(e, e->nav_goal, scratch);
    }
    e->nav_planb = 0;
    if (planb) {
        for (int i = 0; i < 10; i++) e->plan[i] = (int)bounded(e, STREAM_TARGET, 3);
        e->plan_len = 10;
        e->nav_planb = 1;
    }
    e->plan_cur = 0;
}

/* Navigator.step — navigator.py:11-41. _goal_test (:65-70) is always falsy for a flat [r, c]
 * goal, so re-planning happens only when the plan is exhausted. */
static int nav_step(orc_env *e, const int *state, int32_t *scratch)
{
    if (e->rng_mode == ORC_RNG_NP) {
        if (e->plan_cur >= e->plan_len) {
            nav_sample_goal(e, e->nav_goal, scratch);
            nav_plan(e, state, scratch);
        }
        return e->plan[e->plan_cur++];
    }
    /* device spec: closed-loop descent of the BFS field; "plan exhausted" == standing on the goal
     * (or the 10 plan-B actions used up). RPF: the plan was made on the generator's map, which differs from the
     * env's at the patrol cells, so the reference's OPEN-LOOP action list is reproduced by descending the field
     * from a virtual position that ignores the env's extra walls, for exactly the planned number of steps. */
    const int rpf = e->target_mode == ORC_TGT_RPF;
    int exhausted = e->nav_planb ? (e->plan_cur >= e->plan_len)
                    : rpf ? (e->remaining <= 0)
                          : (state[0] == e->nav_goal[0] && state[1] == e->nav_goal[1]);
    if (exhausted) {
        nav_sample_goal(e, e->nav_goal, scratch);
        nav_plan(e, state, scratch);
    }
    if (e->nav_planb) return e->plan[e->plan_cur++];
    if (rpf) {
        static const int DR[4] = {-1, 1, 0, 0}, DC[4] = {0, 0, -1, 1};
        int a = e->dir[e->vpos[0] * e->side + e->vpos[1]];
        e->vpos[0] += DR[a]; e->vpos[1] += DC[a];
        e->remaining--;
        return a;
    }
    return e->dir[state[0] * e->side + state[1]];
}

/* ---- observation: _get_obs/_get_full_obs/_get_partial_obs — track_1v1.py:287-326 ---- */
static void write_obs(const orc_env *e, uint8_t *obs)
{
    int S = e->side;
    if (e->obs_full) { /* obs_type 'Full': both agents get _get_full_obs() (:288-290,295-307), shape [2][S][S] */
        for (int id = 0; id < 2; id++) {
            uint8_t *o = obs + id * S * S;
            memcpy(o, e->maze, (size_t)(S * S));
            o[e->pos[0][0] * S + e->pos[0][1]] = 2;
            o[e->pos[1][0] * S + e->pos[1][1]] = 4;
        }
        return;
    }
    for (int id = 0; id < 2; id++) {
        int r0 = e->pos[id][0] - ORC_POB, c0 = e->pos[id][1] - ORC_POB;
        for (int y = 0; y < ORC_WIN; y++)
            for (int x = 0; x < ORC_WIN; x++) {
                int r = r0 + y, c = c0 + x;
                uint8_t v;
                if (r < 0 || r >= S || c < 0 || c >= S) v = 1;           /* np.pad(..., 1) (:321) */
                else {
                    v = e->maze[r * S + c];
                    if (r == e->pos[0][0] && c == e->pos[0][1]) v = 2;     /* tracker (:300-305) */
                    if (r == e->pos[1][0] && c == e->pos[1][1]) v = 4;     /* target painted last */
                    if (r == e->pos[id][0] && c == e->pos[id][1]) v = (uint8_t)(2 + 2 * id); /* :313 */
                }
                obs[id * ORC_OBS_CELLS + y * ORC_WIN + x] = v;
            }
    }
}

/* reward — track_1v1.py:94-104, evaluated in float64 in the reference's operation order. */
void orc_reward(int64_t d2, double w_p, double *r_track, double *r_target)
{
    const double max_distance = 6.0;
    double distance = sqrt((double)d2);
    double rt = 1 - 2 * distance / max_distance;
    rt = rt > -1 ? rt : -1;
    double over = distance - max_distance;
    over = over > 0 ? over : 0;
    double rg = -rt - w_p * over / max_distance;
    rg = rg > -1 ? rg : -1;
    *r_track = rt; *r_target = rg;
}

orc_env *orc_create(int map_type, int target_mode, int level, int max_steps, int rng_mode,
                    uint64_t seed, uint32_t env_id)
{
    orc_env *e = (orc_env *)calloc(1, sizeof(orc_env));
    e->map_type = map_type; e->target_mode = target_mode; e->level = level;
    e->max_steps = max_steps; e->rng_mode = rng_mode;
    e->k0 = (uint32_t)seed; e->k1 = (uint32_t)(seed >> 32); e->env_id = env_id;
    e->episode = 0;
    e->side = map_type == ORC_MAP_MAZE ? 81 : 82;
    mt_seed(&e->mt, (uint32_t)seed);
    for (int s = 0; s < NUM_STREAMS; s++) e->cache_blk[s] = 0xffffffffu;
    return e;
}
void orc_destroy(orc_env *e) { free(e); }
void orc_seed_np(orc_env *e, uint32_t seed) { mt_seed(&e->mt, seed); }
void orc_set_obs_full(orc_env *e, int full) { e->obs_full = full ? 1 : 0; }
void orc_set_action_type(orc_env *e, int moore) { e->moore = moore ? 1 : 0; }
int orc_obs_size(const orc_env *e) { return e->obs_full ? 2 * e->side * e->side : 2 * ORC_OBS_CELLS; }

/* Track1v1Env.init_maze — track_1v1.py:218-240. */
static void init_maze(orc_env *e, int32_t *scratch)
{
    if (e->map_type == ORC_MAP_MAZE) {
        double r = e->level > 0 ? e->level * 0.02 : .03 * next_double(e, STREAM_MAP);
        gen_maze(e, r);
    } else if (e->map_type == ORC_MAP_BLOCK) {
        double r = e->level > 0 ? e->level * 0.05 : 0.15 * next_double(e, STREAM_MAP);
        gen_block(e, r, scratch);
    } else {
        gen_block(e, 0.0, scratch);
    }
    if (e->target_mode == ORC_TGT_RPF) { /* static_goals() after the env copied the maze (track_1v1.py:233-236) */
        int S = e->side;
        int lo = S / 6, hi = S * 5 / 6;
        int c4[4][2] = {{lo, lo}, {hi, lo}, {hi, hi}, {lo, hi}};
        memcpy(e->gmaze, e->maze, (size_t)(S * S));
        for (int i = 0; i < 4; i++) { e->cand[i][0] = c4[i][0]; e->cand[i][1] = c4[i][1]; e->gmaze[c4[i][0] * S + c4[i][1]] = 0; }
        e->vector = 1;                    /* sample_goal(2): vector = (0 + 1) % 4, both agents get that cell */
        for (int i = 0; i < 2; i++) { e->goal[i][0] = e->cand[1][0]; e->goal[i][1] = e->cand[1][1]; }
        sample_close_states(e, scratch);
        return;                           /* goal_test(init_states[0]) is false: spawn = cell 0, goal = cell 1 */
    }
    sample_goal(e, 2, e->goal, scratch);
    sample_close_states(e, scratch);
    while ((e->pos[0][0] == e->goal[0][0] && e->pos[0][1] == e->goal[0][1]) ||
           (e->pos[0][0] == e->goal[1][0] && e->pos[0][1] == e->goal[1][1]))
        sample_goal(e, 2, e->goal, scratch); /* goal_test loop (:239-240) */
}

static void set_wp(orc_env *e)
{
    e->w_p = e->target_mode == ORC_TGT_PZR ? 1.0 : (e->target_mode == ORC_TGT_FAR ? -0.5 : 0.0);
}

/* Track1v1Env.reset — track_1v1.py:134-168 (+ TimeLimit.reset zeroing elapsed steps). */
void orc_reset(orc_env *e, uint8_t *obs)
{
    int32_t *scratch = (int32_t *)malloc(sizeof(int32_t) * ORC_MAX_SIDE * ORC_MAX_SIDE);
    e->episode++;
    for (int s = 0; s < NUM_STREAMS; s++) { e->ctr[s] = 0; e->cache_blk[s] = 0xffffffffu; }
    init_maze(e, scratch);
    if (e->target_mode == ORC_TGT_NAV || e->target_mode == ORC_TGT_RPF) {
        e->nav_goal[0] = e->goal[1][0]; e->nav_goal[1] = e->goal[1][1];
        nav_plan(e, e->pos[1], scratch);
    }
    if (e->target_mode == ORC_TGT_RAM) ram_reset(e);
    set_wp(e);
    e->c_far = 0; e->t = 0;
    int dr = e->pos[1][0] - e->pos[0][0], dc = e->pos[1][1] - e->pos[0][1];
    e->d2 = (int64_t)dr * dr + (int64_t)dc * dc;
    if (obs) write_obs(e, obs);
    free(scratch);
}

/* Track1v1Env._next_state — track_1v1.py:271-285: the VonNeumann transitions are the first four of the Moore table; only
 * the destination cell is tested (a diagonal move may cut a corner). */
static void move_agent(orc_env *e, int id, int action)
{
    static const int DR[8] = {-1, 1, 0, 0, -1, 1, -1, 1}, DC[8] = {0, 0, -1, 1, 1, 1, -1, -1};
    int r = e->pos[id][0] + DR[action], c = e->pos[id][1] + DC[action];
    if (e->maze[r * e->side + c] != 1) { e->pos[id][0] = r; e->pos[id][1] = c; }
}

/* Track1v1Env.step — track_1v1.py:71-127, wrapped by gym TimeLimit.step (done |= elapsed >= max). */
int orc_step(orc_env *e, const int *actions, uint8_t *obs, double *rewards, int *done, int *applied)
{
    int act[2] = {actions[0], actions[1]};
    int32_t *scratch = NULL;
    if (e->target_mode == ORC_TGT_RAM) act[1] = ram_step(e);
    if (e->target_mode == ORC_TGT_NAV || e->target_mode == ORC_TGT_RPF) {
        scratch = (int32_t *)malloc(sizeof(int32_t) * ORC_MAX_SIDE * ORC_MAX_SIDE);
        act[1] = nav_step(e, e->pos[1], scratch); /* old_state[1] (:84) */
        free(scratch);
    }
    const int amax = e->moore ? 7 : 3;
    if (act[0] < 0 || act[0] > amax || act[1] < 0 || act[1] > amax) return -1;
    move_agent(e, 0, act[0]);
    move_agent(e, 1, act[1]);
    int dr = e->pos[1][0] - e->pos[0][0], dc = e->pos[1][1] - e->pos[0][1];
    e->d2 = (int64_t)dr * dr + (int64_t)dc * dc;
    orc_reward(e->d2, e->w_p, &rewards[0], &rewards[1]);
    if (e->d2 <= 36) e->c_far = 0; else e->c_far += 1;   /* distance <= 6 (:106-109) */
    int d = e->c_far > 10;
    e->t += 1;
    if (e->max_steps > 0 && e->t >= e->max_steps) d = 1;  /* TimeLimit._past_limit */
    *done = d;
    if (obs) write_obs(e, obs);
    if (applied) { applied[0] = act[0]; applied[1] = act[1]; }
    return 0;
}

int orc_inject(orc_env *e, int side, const uint8_t *maze, const int *pos, const int *goals)
{
    if (side != 81 && side != 82) return -1;
    e->side = side;
    memset(e->maze, 0, sizeof(e->maze));
    memcpy(e->maze, maze, (size_t)(side * side));
    for (int i = 0; i < 2; i++)
        for (int k = 0; k < 2; k++) {
            e->pos[i][k] = pos[i * 2 + k];
            if (goals) e->goal[i][k] = goals[i * 2 + k];
        }
    set_wp(e);
    e->c_far = 0; e->t = 0;
    /* scripted-target plan cleared: a Nav target re-plans at its next step */
    e->plan_len = 0; e->plan_cur = 0; e->nav_planb = 0;
    e->nav_goal[0] = e->pos[1][0]; e->nav_goal[1] = e->pos[1][1];
    if (e->target_mode == ORC_TGT_RPF) { /* the injected map is the env's; the planning map clears the patrol cells */
        int lo = side / 6, hi = side * 5 / 6;
        int c4[4][2] = {{lo, lo}, {hi, lo}, {hi, hi}, {lo, hi}};
        memcpy(e->gmaze, e->maze, sizeof(e->maze));
        for (int i = 0; i < 4; i++) { e->cand[i][0] = c4[i][0]; e->cand[i][1] = c4[i][1]; e->gmaze[c4[i][0] * side + c4[i][1]] = 0; }
        e->vector = 0; e->remaining = 0;
    }
    int dr = e->pos[1][0] - e->pos[0][0], dc = e->pos[1][1] - e->pos[0][1];
    e->d2 = (int64_t)dr * dr + (int64_t)dc * dc;
    return 0;
}

int orc_inject_plan(orc_env *e, const int *plan, int len, int cursor)
{
    if (len < 1 || len > PLAN_CAP) return -1;
    for (int i = 0; i < len; i++) e->plan[i] = plan[i];
    e->plan_len = len; e->plan_cur = cursor;
    return 0;
}

int orc_side(const orc_env *e) { return e->side; }
void orc_get_maze(const orc_env *e, uint8_t *maze) { memcpy(maze, e->maze, (size_t)(e->side * e->side)); }
void orc_get_state(const orc_env *e, int *pos, int *goals, int *c_far, int *t, int64_t *d2)
{
    for (int i = 0; i < 2; i++)
        for (int k = 0; k < 2; k++) {
            if (pos) pos[i * 2 + k] = e->pos[i][k];
            if (goals) goals[i * 2 + k] = e->goal[i][k];
        }
    if (c_far) *c_far = e->c_far;
    if (t) *t = e->t;
    if (d2) *d2 = e->d2;
}
void orc_get_obs(const orc_env *e, uint8_t *obs) { write_obs(e, obs); }
int orc_get_plan(const orc_env *e, int *plan, int *cursor)
{
    int n = e->plan_len < 1024 ? e->plan_len : 1024;
    for (int i = 0; i < n; i++) plan[i] = e->plan[i];
    if (cursor) *cursor = e->plan_cur;
    return e->plan_len;
}
uint32_t orc_episode(const orc_env *e) { return e->episode; }
/* Nav / RPF target: the Navigator's current goal (re-drawn when the first one was unreachable, navigator.py:50-56), whether it
 * fell back to plan B (:58-59) and the length of its plan (NP mode: the A* action list; PHILOX mode: the BFS distance). */
void orc_get_nav(const orc_env *e, int *nav_goal, int *planb, int *plan_len)
{
    if (nav_goal) { nav_goal[0] = e->nav_goal[0]; nav_goal[1] = e->nav_goal[1]; }
    if (planb) *planb = e->nav_planb;
    if (plan_len) *plan_len = e->nav_planb ? e->plan_len : (e->rng_mode == ORC_RNG_NP ? e->plan_len : e->remaining);
}

/* Lock-step helper for the full-size parity tests: n envs, one step each, and — as the vectorised product does inside its
 * step launch — a finished env is reset at once and reports the FIRST observation of its next episode (the reference worker
 * discards the terminal observation too: train.py:73-74). Plain loop over orc_step / orc_reset. */
int orc_step_batch(orc_env **envs, int n, const int *actions, uint8_t *obs, double *rewards, uint8_t *done, int auto_reset)
{
    for (int i = 0; i < n; i++) {
        orc_env *e = envs[i];
        const int sz = orc_obs_size(e);
        int d = 0;
        if (orc_step(e, actions + 2 * i, obs + (size_t)i * sz, rewards + 2 * i, &d, NULL) != 0) return -1 - i;
        done[i] = (uint8_t)d;
        if (d && auto_reset) orc_reset(e, obs + (size_t)i * sz);
    }
    return 0;
}
