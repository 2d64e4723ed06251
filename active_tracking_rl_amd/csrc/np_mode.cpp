// np_mode.cpp — the reference-exact episode source behind include/track2d_np.h (host only).
//
// The reference env draws every random number from numpy's legacy GLOBAL stream, so "identical episodes on identical
// seeds" needs that stream and the exact order the reference consumes it in — including draws whose values are thrown
// away (a full 6400-element shuffle to pick K block cells, two extra free-cell shuffles per reset, ...). This file
// restates that consumer: generators.py / navigator.py / Astar_solver.py / the random part of track_1v1.reset, over an
// MT19937 with numpy's legacy derived distributions. The device does everything else (track2d_hip.hip).
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/track2d_np.h"

namespace npm {

static thread_local char g_err[256] = "";
static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// ---- numpy legacy stream ---------------------------------------------------------------------------------------
// MT19937 (Matsumoto & Nishimura) seeded the way RandomState.seed(int) does (init_genrand), plus the derived draws
// numpy's legacy RandomState makes from its 32-bit words.
struct Stream {
    uint32_t mt[624];
    int pos;

    void seed(uint32_t s)
    {
        mt[0] = s;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        pos = 624;
    }
    void refill()
    {
        for (int k = 0; k < 624; k++) {
            const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
            mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        pos = 0;
    }
    uint32_t word()
    {
        if (pos >= 624) refill();
        uint32_t y = mt[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    // random_sample(): 53 bits out of two words
    double uniform()
    {
        const uint32_t a = word() >> 5, b = word() >> 6;
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
    // uniform integer in [0, top]: smallest all-ones mask >= top, one word per attempt, rejected while > top; top == 0
    // consumes nothing (random_interval / the masked bounded-integer path of randint)
    uint32_t upto(uint32_t top)
    {
        if (top == 0u) return 0u;
        uint32_t mask = top;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        uint32_t v;
        do { v = word() & mask; } while (v > top);
        return v;
    }
    // randint(low, high): high exclusive
    int randint(int low, int high) { return low + (int)upto((uint32_t)(high - 1 - low)); }
    // permutation(n): arange(n) shuffled from the top (i = n-1 .. 1: swap with a uniform j in [0, i])
    void permutation(int n, std::vector<int> &out)
    {
        out.resize((size_t)n);
        for (int i = 0; i < n; i++) out[(size_t)i] = i;
        for (int i = n - 1; i >= 1; i--) {
            const int j = (int)upto((uint32_t)i);
            const int t = out[(size_t)i]; out[(size_t)i] = out[(size_t)j]; out[(size_t)j] = t;
        }
    }
};

// ---- maps ----------------------------------------------------------------------------------------------------------
constexpr int kMax = 82;
struct Grid {
    int side = 0;
    uint8_t c[kMax * kMax];
    uint8_t &at(int r, int col) { return c[r * kMax + col]; }
    uint8_t at(int r, int col) const { return c[r * kMax + col]; }
    void clear(int s) { side = s; std::memset(c, 0, sizeof(c)); }
};
struct Cell { int r, c; };
static bool same(const Cell &a, const Cell &b) { return a.r == b.r && a.c == b.c; }

// np.where(maze == 0) zipped: free cells in row-major order
static void free_cells(const Grid &g, std::vector<Cell> &out)
{
    out.clear();
    for (int r = 0; r < g.side; r++)
        for (int col = 0; col < g.side; col++)
            if (g.at(r, col) == 0) out.push_back(Cell{r, col});
}

// RandomBlockMazeGenerator._generate_maze (generators.py:157-176): exactly int(ratio * 6400) distinct interior cells =
// the first K entries of permutation(6400) (the whole shuffle is drawn even for K = 0), then the wall border.
static void block_map(Stream &rs, double ratio, Grid &g)
{
    g.clear(82);
    const int K = (int)(ratio * 6400.0);
    std::vector<int> perm;
    rs.permutation(6400, perm);
    for (int i = 0; i < K; i++) g.at(perm[(size_t)i] / 80 + 1, perm[(size_t)i] % 80 + 1) = 1;
    for (int i = 0; i < 82; i++) { g.at(0, i) = g.at(81, i) = 1; g.at(i, 0) = g.at(i, 81) = 1; }
}

// RandomMazeGenerator._generate_maze (generators.py:115-145) for width = height = 80 -> 81 x 81
static void maze_map(Stream &rs, double ratio, Grid &g)
{
    const int S = 81;
    g.clear(S);
    const int complexity = (int)(ratio * (5.0 * (S + S)));
    const int density = (int)(ratio * (double)((S / 2) * (S / 2)));
    for (int i = 0; i < S; i++) { g.at(0, i) = g.at(S - 1, i) = 1; g.at(i, 0) = g.at(i, S - 1) = 1; }
    for (int i = 0; i < density; i++) {
        int x = rs.randint(0, S / 2 + 1) * 2;       // the tuple on generators.py:131 evaluates x's draw first
        int y = rs.randint(0, S / 2 + 1) * 2;
        g.at(y, x) = 1;
        for (int j = 0; j < complexity; j++) {
            Cell nb[4];
            int n = 0;
            if (x > 1) nb[n++] = Cell{y, x - 2};
            if (x < S - 2) nb[n++] = Cell{y, x + 2};
            if (y > 1) nb[n++] = Cell{y - 2, x};
            if (y < S - 2) nb[n++] = Cell{y + 2, x};
            if (n == 0) continue;
            const Cell pick = nb[rs.randint(0, n)];
            if (g.at(pick.r, pick.c) == 0) {
                g.at(pick.r, pick.c) = 1;
                // Z[y_ + (y - y_) // 2, x_ + (x - x_) // 2]: floor division of +-2 or 0 is exact
                g.at(pick.r + (y - pick.r) / 2, pick.c + (x - pick.c) / 2) = 1;
                x = pick.c; y = pick.r;
            }
        }
    }
}

// ---- A* (Astar_solver.py:42-173), with heapq's exact sift order ---------------------------------------------------
struct Astar {
    struct Node { Cell s; int prev; int action; int cost; };
    struct Item { double f; int node; };
    const Grid &g;
    Cell goal;
    std::vector<Node> nodes;
    std::vector<Item> heap;
    std::vector<int> in_frontier;    // state -> node index or -1 (Frontier.state_nodes)
    std::vector<uint8_t> explored;
    int solution = -1;

    Astar(const Grid &grid, Cell start, Cell goal_) : g(grid), goal(goal_)
    {
        in_frontier.assign((size_t)(kMax * kMax), -1);
        explored.assign((size_t)(kMax * kMax), 0);
        nodes.push_back(Node{start, -1, -1, 0});
        push(0);
        search();
    }
    static int key(const Cell &s) { return s.r * kMax + s.c; }
    // f = path_cost + Euclidean distance to the goal (:151-153; norm of an int vector = sqrt of an exact dot product)
    double f_of(int n) const
    {
        const double dr = (double)(nodes[(size_t)n].s.r - goal.r), dc = (double)(nodes[(size_t)n].s.c - goal.c);
        return (double)nodes[(size_t)n].cost + std::sqrt(dr * dr + dc * dc);
    }
    // [f, node] < [f', node'] as Python compares lists: by f unless equal, then Node.__lt__ = smaller path cost (:30-32)
    bool less(const Item &a, const Item &b) const
    {
        if (a.f != b.f) return a.f < b.f;
        return nodes[(size_t)a.node].cost < nodes[(size_t)b.node].cost;
    }
    void sift_down(int start, int p)          // heapq._siftdown: towards the root
    {
        const Item it = heap[(size_t)p];
        while (p > start) {
            const int parent = (p - 1) >> 1;
            if (!less(it, heap[(size_t)parent])) break;
            heap[(size_t)p] = heap[(size_t)parent];
            p = parent;
        }
        heap[(size_t)p] = it;
    }
    void sift_up(int p)                        // heapq._siftup: to a leaf, then back
    {
        const int end = (int)heap.size(), start = p;
        const Item it = heap[(size_t)p];
        int child = 2 * p + 1;
        while (child < end) {
            const int right = child + 1;
            if (right < end && !less(heap[(size_t)child], heap[(size_t)right])) child = right;
            heap[(size_t)p] = heap[(size_t)child];
            p = child;
            child = 2 * p + 1;
        }
        heap[(size_t)p] = it;
        sift_down(start, p);
    }
    void push(int n)                           // Frontier.add
    {
        heap.push_back(Item{f_of(n), n});
        sift_down(0, (int)heap.size() - 1);
        in_frontier[(size_t)key(nodes[(size_t)n].s)] = n;
    }
    int pop()                                  // Frontier.pop
    {
        const Item last = heap.back();
        heap.pop_back();
        Item top = last;
        if (!heap.empty()) {
            top = heap[0];
            heap[0] = last;
            sift_up(0);
        }
        in_frontier[(size_t)key(nodes[(size_t)top.node].s)] = -1;
        return top.node;
    }
    void replace(int n)                        // Frontier.replace: scan in array order, re-sift every match
    {
        for (size_t i = 0; i < heap.size(); i++)
            if (same(nodes[(size_t)heap[i].node].s, nodes[(size_t)n].s)) {
                heap[i] = Item{f_of(n), n};
                sift_down(0, (int)i);
                in_frontier[(size_t)key(nodes[(size_t)n].s)] = n;
            }
    }
    Cell moved(const Cell &s, int action) const   // _next_state: a wall bump returns the same state
    {
        static const int dr[4] = {-1, 1, 0, 0}, dc[4] = {0, 0, -1, 1};
        const Cell t{s.r + dr[action], s.c + dc[action]};
        return g.at(t.r, t.c) == 1 ? s : t;
    }
    void search()
    {
        while (!heap.empty()) {
            const int n = pop();
            const Cell s = nodes[(size_t)n].s;
            if (same(s, goal)) { solution = n; return; }
            explored[(size_t)key(s)] = 1;
            for (int a = 0; a < 4; a++) {
                const Cell cs = moved(s, a);
                const int cost = nodes[(size_t)n].cost + 1;
                const int k = key(cs);
                if (!explored[(size_t)k] && in_frontier[(size_t)k] < 0) {
                    nodes.push_back(Node{cs, n, a, cost});
                    push((int)nodes.size() - 1);
                } else if (in_frontier[(size_t)k] >= 0 && nodes[(size_t)in_frontier[(size_t)k]].cost < cost) {
                    // the reference's (inverted) test, :146-147: the queued node is CHEAPER than the child -> replace it
                    nodes.push_back(Node{cs, n, a, cost});
                    replace((int)nodes.size() - 1);
                }
            }
        }
    }
    bool solvable() const { return solution >= 0; }
    void actions(std::vector<int> &out) const
    {
        out.clear();
        for (int n = solution; n >= 0 && nodes[(size_t)n].prev >= 0; n = nodes[(size_t)n].prev)
            out.push_back(nodes[(size_t)n].action);
        for (size_t i = 0, j = out.size(); i + 1 < j; i++, j--) { const int t = out[i]; out[i] = out[j - 1]; out[j - 1] = t; }
    }
};

enum { MAP_BLOCK = 0, MAP_MAZE = 1, MAP_EMPTY = 2 };
enum { TGT_ADV = 0, TGT_PZR = 1, TGT_FAR = 2, TGT_NAV = 3, TGT_RAM = 4, TGT_RPF = 5 };

} // namespace npm

using namespace npm;

struct t2d_np {
    int map_type, mode, level;
    Stream rs;
    Grid gen;          // maze_generator.maze (RPF: patrol cells cleared)
    Grid env;          // Track1v1Env.maze = a copy taken BEFORE static_goals() (track_1v1.py:233-236)
    bool is_static = false;
    int vector = 0;
    Cell cand[4];
    Cell goals[2], init[2];
    // scripted target
    std::vector<int> plan;
    int a_i = 0;
    Cell nav_goal{0, 0};
    Cell target{0, 0};     // mirror of state[1]
    bool reset_done = false;

    // MazeGenerator.sample_goal(1)[0] / sample_goal(2) (generators.py:38-51)
    void sample_goal(int num, Cell *out)
    {
        if (!is_static) {
            std::vector<Cell> fs;
            free_cells(gen, fs);
            std::vector<int> perm;
            rs.permutation((int)fs.size(), perm);            // choice(len, size=num, replace=False)
            for (int i = 0; i < num; i++) out[i] = fs[(size_t)perm[(size_t)i]];
        } else {
            vector = (vector + 1) % 4;
            for (int i = 0; i < num; i++) out[i] = cand[vector];
        }
    }
    // get_around(state, 1) (generators.py:82-94): the exclusive slice bounds make it the 2 x 2 block up-left of state
    Cell get_around(const Cell &s)
    {
        const int x0 = s.r - 1 > 0 ? s.r - 1 : 0, x1 = s.r + 1 < gen.side - 1 ? s.r + 1 : gen.side - 1;
        const int y0 = s.c - 1 > 0 ? s.c - 1 : 0, y1 = s.c + 1 < gen.side - 1 ? s.c + 1 : gen.side - 1;
        std::vector<Cell> fs;
        for (int r = x0; r < x1; r++)
            for (int c = y0; c < y1; c++)
                if (gen.at(r, c) == 0) fs.push_back(Cell{r, c});
        std::vector<int> perm;
        rs.permutation((int)fs.size(), perm);                // choice(len, size=1, replace=False)
        return fs[(size_t)perm[0]];
    }
    // init_maze (track_1v1.py:218-240)
    void init_maze()
    {
        is_static = false;
        if (map_type == MAP_MAZE) {
            const double r = level > 0 ? (double)level * 0.02 : .03 * rs.uniform();
            maze_map(rs, r, gen);
        } else if (map_type == MAP_BLOCK) {
            const double r = level > 0 ? (double)level * 0.05 : 0.15 * rs.uniform();
            block_map(rs, r, gen);
        } else {
            block_map(rs, 0.0, gen);
        }
        env = gen;
        if (mode == TGT_RPF) {                                // static_goals (generators.py:12-19)
            is_static = true;
            const int s = gen.side;
            cand[0] = Cell{s / 6, s / 6}; cand[1] = Cell{s * 5 / 6, s / 6};
            cand[2] = Cell{s * 5 / 6, s * 5 / 6}; cand[3] = Cell{s / 6, s * 5 / 6};
            for (const Cell &g : cand) gen.at(g.r, g.c) = 0;
            vector = 0;
        }
        sample_goal(2, goals);
        {   // sample_close_states(2, 1) (generators.py:53-77)
            std::vector<Cell> fs;
            free_cells(gen, fs);
            std::vector<int> perm;
            rs.permutation((int)fs.size(), perm);            // drawn in the static case too, then ignored
            init[0] = is_static ? cand[0] : fs[(size_t)perm[0]];
            init[1] = get_around(init[0]);
            free_cells(gen, fs);                             // sample_state(0): choice(len, size=0, replace=False)
            rs.permutation((int)fs.size(), perm);
        }
        while (same(init[0], goals[0]) || same(init[0], goals[1])) sample_goal(2, goals);   // goal_test loop
    }
    // the planning loop shared by Navigator.reset and Navigator.step (navigator.py:22-38, :46-62)
    void plan_to_goal(const Cell &from)
    {
        int count_res = 0;
        bool plan_b = false;
        std::vector<int> acts;
        for (;;) {
            Astar a(gen, from, nav_goal);
            if (a.solvable()) a.actions(acts);
            if (a.solvable() && !acts.empty()) break;
            if (++count_res > 5) { plan_b = true; break; }
            sample_goal(1, &nav_goal);
        }
        if (plan_b) {                                        // np.random.choice(all_actions, 10)
            acts.resize(10);
            for (int i = 0; i < 10; i++) acts[(size_t)i] = rs.randint(0, 4);
        }
        plan = acts;
        a_i = 0;
    }
    void ram_new_plan()                                      // choice(all_actions, randint(1, 10)): the length first
    {
        const int n = rs.randint(1, 10);
        plan.resize((size_t)n);
        for (int i = 0; i < n; i++) plan[(size_t)i] = rs.randint(0, 4);
        a_i = 0;
    }
    Cell moved_on_env(const Cell &s, int action) const
    {
        static const int dr[4] = {-1, 1, 0, 0}, dc[4] = {0, 0, -1, 1};
        const Cell t{s.r + dr[action], s.c + dc[action]};
        return env.at(t.r, t.c) == 1 ? s : t;
    }
};

extern "C" const char *t2d_np_last_error(void) { return g_err; }

extern "C" int t2d_np_create(int map_type, int target_mode, int level, uint32_t seed, t2d_np **out)
{
    if (!out) return fail(-1, "t2d_np_create: null argument");
    if (map_type < 0 || map_type > 2 || target_mode < 0 || target_mode > 5 || level < 0 || level > 15)
        return fail(-1, "t2d_np_create: map_type %d / target_mode %d / level %d out of range", map_type, target_mode, level);
    t2d_np *e = new (std::nothrow) t2d_np();
    if (!e) return fail(-1, "t2d_np_create: out of memory");
    e->map_type = map_type; e->mode = target_mode; e->level = level;
    e->rs.seed(seed);
    *out = e;
    return 0;
}

extern "C" int t2d_np_destroy(t2d_np *e)
{
    delete e;
    return 0;
}

extern "C" int t2d_np_seed(t2d_np *e, uint32_t seed)
{
    if (!e) return fail(-1, "t2d_np_seed: null handle");
    e->rs.seed(seed);
    return 0;
}

extern "C" int t2d_np_reset(t2d_np *e, uint8_t *maze, int32_t *side, int32_t pos[4], int32_t goals[4])
{
    if (!e || !maze || !side || !pos || !goals) return fail(-1, "t2d_np_reset: null argument");
    e->init_maze();
    e->target = e->init[1];
    e->plan.clear();
    e->a_i = 0;
    if (e->mode == TGT_NAV || e->mode == TGT_RPF) {          // Navigator.reset(init_states[1], goal_states[1], generator)
        e->nav_goal = e->goals[1];
        e->plan_to_goal(e->init[1]);
    } else if (e->mode == TGT_RAM) {                          // RamAgent.reset
        e->ram_new_plan();
    }
    std::memcpy(maze, e->env.c, sizeof(e->env.c));
    *side = e->env.side;
    pos[0] = e->init[0].r; pos[1] = e->init[0].c; pos[2] = e->init[1].r; pos[3] = e->init[1].c;
    goals[0] = e->goals[0].r; goals[1] = e->goals[0].c; goals[2] = e->goals[1].r; goals[3] = e->goals[1].c;
    e->reset_done = true;
    return 0;
}

extern "C" int t2d_np_target_action(t2d_np *e, int32_t *action)
{
    if (!e || !action) return fail(-1, "t2d_np_target_action: null argument");
    if (!e->reset_done) return fail(-2, "t2d_np_target_action: reset first");
    int act;
    if (e->mode == TGT_RAM) {                                 // RamAgent.step (navigator.py:77-88)
        act = e->plan[(size_t)e->a_i];
        e->a_i++;
        if (e->a_i >= (int)e->plan.size()) {
            if (e->rs.randint(0, 2) == 0) {                   // np.random.choice([0, 1], 1) == 0
                act = e->rs.randint(0, 4);                    // overrides the action being returned
                const int n = e->rs.randint(1, 10);
                e->plan.assign((size_t)n, act);
            } else {
                e->ram_new_plan();
            }
            e->a_i = 0;
        }
    } else if (e->mode == TGT_NAV || e->mode == TGT_RPF) {    // Navigator.step (navigator.py:11-41)
        // _goal_test is never true (the goal is a flat [r, c]): a new plan only when the old one is used up
        if (e->a_i >= (int)e->plan.size()) {
            e->sample_goal(1, &e->nav_goal);
            e->plan_to_goal(e->target);
        }
        act = e->plan[(size_t)e->a_i];
        e->a_i++;
    } else {
        return fail(-2, "t2d_np_target_action: target_mode %d is driven by the policy", e->mode);
    }
    e->target = e->moved_on_env(e->target, act);
    *action = act;
    return 0;
}

// The MT19937 state RandomState.seed(seed) leaves behind: 624 words + the read position (624 = "regenerate before the next
// word"). What t2d_np_attach (csrc/track2d_hip.hip) uploads per env for the DEVICE-side numpy-exact generators.
extern "C" int t2d_np_mt_state(uint32_t seed, uint32_t out[625])
{
    if (!out) return fail(-1, "t2d_np_mt_state: null argument");
    Stream rs;
    rs.seed(seed);
    std::memcpy(out, rs.mt, sizeof(rs.mt));
    out[624] = (uint32_t)rs.pos;
    return 0;
}

// ---- many envs at once: the per-env calls above spread over host threads (each env owns its stream: no shared state) ----
#include <atomic>
#include <thread>

template <class F> static int for_each_env(int count, int threads, F &&body)
{
    if (count <= 0) return 0;
    int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > count) nt = count;
    std::atomic<int> next(0), err(0);
    char first_err[sizeof(g_err)] = "";
    std::atomic<bool> have_err(false);
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= count) return;
            const int rc = body(i);
            if (rc != 0) {
                err.store(rc);
                bool expected = false;
                if (have_err.compare_exchange_strong(expected, true)) snprintf(first_err, sizeof(first_err), "env %d: %s", i, g_err);
            }
        }
    };
    if (nt == 1) work();
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; t++) pool.emplace_back(work);
        for (auto &t : pool) t.join();
    }
    if (err.load() != 0) { snprintf(g_err, sizeof(g_err), "%s", first_err); return err.load(); }
    return 0;
}

extern "C" int t2d_np_reset_many(t2d_np *const *envs, int count, uint8_t *mazes, int32_t *sides, int32_t *pos, int32_t *goals,
                                 int threads)
{
    if (!envs || !mazes || !sides || !pos || !goals) return fail(-1, "t2d_np_reset_many: null argument");
    return for_each_env(count, threads, [&](int i) {
        return t2d_np_reset(envs[i], mazes + (size_t)i * 82 * 82, sides + i, pos + 4 * i, goals + 4 * i);
    });
}

extern "C" int t2d_np_target_actions(t2d_np *const *envs, int count, int32_t *actions, int threads)
{
    if (!envs || !actions) return fail(-1, "t2d_np_target_actions: null argument");
    return for_each_env(count, threads, [&](int i) { return t2d_np_target_action(envs[i], actions + i); });
}

extern "C" int t2d_np_get_plan(const t2d_np *e, int32_t *plan, int32_t max_len, int32_t *len, int32_t *cursor,
                               int32_t navgoal[2])
{
    if (!e) return fail(-1, "t2d_np_get_plan: null handle");
    if (len) *len = (int32_t)e->plan.size();
    if (cursor) *cursor = e->a_i;
    if (plan)
        for (int i = 0; i < max_len && i < (int)e->plan.size(); i++) plan[i] = e->plan[(size_t)i];
    if (navgoal) { navgoal[0] = e->nav_goal.r; navgoal[1] = e->nav_goal.c; }
    return 0;
}

extern "C" int t2d_np_astar(const uint8_t *maze, int32_t side, const int32_t start[2], const int32_t goal[2],
                            int32_t *actions, int32_t max_len, int32_t *n, int32_t *solvable)
{
    if (!maze || !start || !goal || !n || !solvable) return fail(-1, "t2d_np_astar: null argument");
    if (side < 3 || side > kMax) return fail(-1, "t2d_np_astar: side %d", side);
    for (int k = 0; k < 2; k++)
        if (start[k] < 1 || start[k] > side - 2 || goal[k] < 0 || goal[k] >= side)
            return fail(-1, "t2d_np_astar: start must be an interior cell, the goal inside the map");
    Grid g;
    g.clear(side);
    for (int r = 0; r < side; r++)
        for (int c = 0; c < side; c++) g.at(r, c) = maze[r * side + c] ? 1 : 0;
    Astar a(g, Cell{start[0], start[1]}, Cell{goal[0], goal[1]});
    *solvable = a.solvable() ? 1 : 0;
    std::vector<int> acts;
    if (a.solvable()) a.actions(acts);
    *n = (int32_t)acts.size();
    if (actions)
        for (int i = 0; i < max_len && i < (int)acts.size(); i++) actions[i] = acts[(size_t)i];
    return 0;
}

extern "C" int t2d_np_draw(t2d_np *e, int kind, uint32_t arg, uint32_t count, double *out)
{
    if (!e || !out) return fail(-1, "t2d_np_draw: null argument");
    if (kind == 0) {
        for (uint32_t i = 0; i < count; i++) out[i] = e->rs.uniform();
    } else if (kind == 1) {
        for (uint32_t i = 0; i < count; i++) out[i] = (double)e->rs.randint(0, (int)arg);
    } else if (kind == 2) {
        std::vector<int> perm;
        e->rs.permutation((int)arg, perm);
        for (uint32_t i = 0; i < arg; i++) out[i] = (double)perm[i];
    } else {
        return fail(-1, "t2d_np_draw: kind %d", kind);
    }
    return 0;
}
