// CPU baseline for bench.py: the REFERENCE'S ARCHITECTURE on this box — one OS thread per game doing a
// sequential PUCT search on the host, evaluations batched through a double buffer to an evaluator the
// host loop drives (the GPU tower, or a null evaluator) — written from scratch as a measurement
// target.  It is not part of the product library and nothing in ataxxzero_amd/ loads it.
//
// What it mirrors (file:line under /root/reference, cpp/self_play_client.cpp):
//   Worker::thread_main        :620-645   forever: generate_game
//   generate_game              :508-582   while (root.N < visits) step(); sample ~ visits; play
//   MCTS::step                 :419-473   select by PUCT (:310-366), expand, evaluate, backup
//   Evaluations::populate      :153-272   833-way softmax in double, legal gather, renormalise, root Dirichlet
//   request_evaluation         :648-681   global mutex, slot = fill_levels[cur]++, wait on the worker's condvar
//   get_workload               :708-721   poll (100 us) for a full buffer
//   complete_workload          :723-738   copy 833 logits + value to each worker, notify_one
// Differences, all in the baseline's favour: children live in contiguous arrays instead of
// unordered_map + shared_ptr, re-rooting compacts by copying, and a leaf travels to the evaluator as its
// 16-byte board (the GPU tower builds the planes itself) instead of 196 floats.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "../../ataxxzero_amd/csrc/azh_device.h"  // host side of the rules: make_move, ring masks, policy_index

using namespace azh;

namespace {

constexpr int POLICY = 833;
constexpr double C_PUCT = 1.0, ALPHA = 0.15, NOISE_W = 0.25;  // :31-33
constexpr int MAX_PLIES = 400;                                 // :34
constexpr u64 START_X = (1ULL << 42) | (1ULL << 6), START_O = (1ULL << 48) | (1ULL << 0);
constexpr u64 BLOCKERS = (1ULL << 31) | (1ULL << 23) | (1ULL << 25) | (1ULL << 17);  // :23

struct StopWorking {};

int host_movegen(const Board &b, u16 *out)
{
    const u64 own = b.turn ? b.o : b.x;
    const u64 empty = BOARD_MASK & ~(b.x | b.o | BLOCKERS);
    int n = 0;
    for (u64 p = own; p; p &= p - 1) {
        const int from = __builtin_ctzll(p);
        for (u64 t = double_jump_bb(1ULL << from) & empty; t; t &= t - 1)
            out[n++] = (u16)(from | (__builtin_ctzll(t) << 8));
    }
    for (u64 c = single_jump_bb(own) & empty; c; c &= c - 1) {
        const int to = __builtin_ctzll(c);
        out[n++] = (u16)(to | (to << 8));
    }
    return n;
}

int board_result(const Board &b, int n_moves)  // :109-144
{
    int p1 = __builtin_popcountll(b.x), p2 = __builtin_popcountll(b.o);
    const int bl = __builtin_popcountll(BLOCKERS);
    if (p1 == 0) return 2;
    if (p2 == 0) return 1;
    if (n_moves == 0) {
        const int emp = 49 - p1 - p2 - bl;
        if (b.turn == 0) p2 += emp; else p1 += emp;
    }
    if (p1 + p2 + bl == 49)
        return p1 < p2 ? 2 : 1;
    return 0;
}

struct Edge {
    u16 move;
    int child;
    double prior, visits, total;
};
struct Node {
    Board board;
    int first, count, result;
    double value, all_visits;
};

struct Shared;

struct Worker {
    Shared *S;
    int id;
    std::mutex m;
    std::condition_variable cv;
    bool filled = false;
    float logits[POLICY];
    float value = 0;
    std::thread t;
    std::vector<Node> nodes, nodes2;
    std::vector<Edge> edges, edges2;
    std::mt19937_64 rng;

    void evaluate(int node, bool root);
    int add_node(const Board &b, bool root);
    void step();
    void play(int edge_index);
    void game();
    void main();
};

struct Shared {
    int B = 0, threads = 0, visits = 0;
    std::vector<std::unique_ptr<Worker>> workers;
    std::vector<u64> boards[2];        // [B][2] (mover, opponent)
    std::vector<int> slot_owner[2];
    int fill[2] = {0, 0};
    int cur = 0;
    std::deque<int> filled;
    std::mutex global;
    std::atomic<bool> keep{false};
    std::atomic<long long> steps{0}, evals{0}, plies{0}, games{0};
};

Shared *G = nullptr;

// request_evaluation (:648-681)
void request(Worker &w, const Board &b)
{
    Shared &S = *w.S;
    {
        std::lock_guard<std::mutex> lock(S.global);
        const int buf = S.cur, slot = S.fill[buf]++;
        S.boards[buf][2 * slot] = b.turn ? b.o : b.x;
        S.boards[buf][2 * slot + 1] = b.turn ? b.x : b.o;
        S.slot_owner[buf][slot] = w.id;
        w.filled = false;
        if (S.fill[buf] == S.B) {
            S.filled.push_back(buf);
            S.cur ^= 1;
        }
    }
    std::unique_lock<std::mutex> lk(w.m);
    while (!w.filled) {
        w.cv.wait_for(lk, std::chrono::milliseconds(250));
        if (!S.keep.load())
            throw StopWorking();
    }
    S.evals++;
}

// Evaluations::populate (:153-272) for the node's edges
void Worker::evaluate(int ni, bool root)
{
    Node &n = nodes[ni];
    if (n.result != 0) {
        n.value = (n.result == 1) == (n.board.turn == 0) ? 1.0 : -1.0;
        return;
    }
    request(*this, n.board);
    n.value = value;
    double e[POLICY], total = 0;  // softmax over all 833 in double, no max-subtraction (:208-218)
    for (int i = 0; i < POLICY; i++) {
        e[i] = exp((double)logits[i]);
        total += e[i];
    }
    double legal = 0;
    for (int j = 0; j < n.count; j++) {
        Edge &ed = edges[n.first + j];
        ed.prior = e[policy_index(ed.move)] / total;
        legal += ed.prior;
    }
    if (legal > 0)
        for (int j = 0; j < n.count; j++)
            edges[n.first + j].prior /= legal;
    if (root) {  // :250-271
        std::gamma_distribution<double> gd(ALPHA, 1.0);
        std::vector<double> g(n.count);
        double gs = 0;
        for (auto &x : g) gs += (x = gd(rng));
        for (int j = 0; j < n.count; j++)
            edges[n.first + j].prior = (1.0 - NOISE_W) * edges[n.first + j].prior + NOISE_W * g[j] / gs;
    }
}

int Worker::add_node(const Board &b, bool root)
{
    u16 mv[MAX_MOVES];
    const int cnt = host_movegen(b, mv);
    Node n;
    n.board = b;
    n.result = board_result(b, cnt);
    n.first = (int)edges.size();
    n.count = n.result ? 0 : cnt;
    n.value = 0;
    n.all_visits = 0;
    for (int j = 0; j < n.count; j++)
        edges.push_back(Edge{mv[j], -1, 0.0, 0.0, 0.0});
    nodes.push_back(n);
    const int id = (int)nodes.size() - 1;
    evaluate(id, root);
    return id;
}

// MCTS::step (:419-473)
void Worker::step()
{
    int path[MAX_PLIES + 8], depth = 0;
    int ni = 0;
    for (;;) {
        Node &n = nodes[ni];
        if (n.result != 0 || n.count == 0)
            break;  // select_action -> NO_MOVE
        const double sq = sqrt(1.0 + n.all_visits);
        double best = -1;
        int bj = 0;
        for (int j = 0; j < n.count; j++) {  // total_action_score (:310-324), ties to the last (:354)
            const Edge &ed = edges[n.first + j];
            const double u = C_PUCT * ed.prior * sq / (1.0 + ed.visits);
            const double q = ed.visits > 0 ? ed.total / ed.visits : 0.0;
            if (u + q >= best) {
                best = u + q;
                bj = j;
            }
        }
        const int ei = n.first + bj;
        path[depth++] = ei;
        if (edges[ei].child < 0) {
            const u16 m = edges[ei].move;
            const Board cb = make_move(n.board, m & 0xFF, m >> 8);
            const int child = add_node(cb, false);  // may reallocate: no references held across it
            edges[ei].child = child;
            ni = child;
            break;
        }
        ni = edges[ei].child;
    }
    double s = (nodes[ni].value + 1.0) / 2.0;
    for (int i = depth - 1; i >= 0; i--) {  // :449-459
        s = 1.0 - s;
        Edge &ed = edges[path[i]];
        ed.visits += 1;
        ed.total += s;
    }
    // parent.all_edge_visits along the path
    int at = 0;
    for (int i = 0; i < depth; i++) {
        nodes[at].all_visits += 1;
        at = edges[path[i]].child;
    }
    S->steps++;
}

// MCTS::play (:475-492): keep the chosen subtree (compacting copy), then re-evaluate the new root with noise
void Worker::play(int ei)
{
    const int c = edges[ei].child;
    if (c < 0) {
        const Board nb = make_move(nodes[0].board, edges[ei].move & 0xFF, edges[ei].move >> 8);
        nodes.clear();
        edges.clear();
        add_node(nb, true);
        return;
    }
    nodes2.clear();
    edges2.clear();
    nodes2.push_back(nodes[c]);
    for (size_t q = 0; q < nodes2.size(); q++) {
        const int of = nodes2[q].first, cnt = nodes2[q].count;
        nodes2[q].first = (int)edges2.size();
        for (int j = 0; j < cnt; j++) {
            Edge ed = edges[of + j];
            if (ed.child >= 0) {
                nodes2.push_back(nodes[ed.child]);
                ed.child = (int)nodes2.size() - 1;
            }
            edges2.push_back(ed);
        }
    }
    nodes.swap(nodes2);
    edges.swap(edges2);
    evaluate(0, true);  // :489-490
}

void Worker::game()  // generate_game (:508-582)
{
    nodes.clear();
    edges.clear();
    Board b;
    b.x = START_X;
    b.o = START_O;
    b.turn = 0;
    add_node(b, true);
    for (int ply = 0; ply < MAX_PLIES; ply++) {
        while (nodes[0].all_visits < S->visits)
            step();
        // sample_proportionally_to_visits (:495-506)
        const Node &r = nodes[0];
        double x = std::uniform_real_distribution<double>(0, r.all_visits)(rng);
        int pick = r.first;
        for (int j = 0; j < r.count; j++) {
            pick = r.first + j;
            x -= edges[pick].visits;
            if (x <= 0 && edges[pick].visits > 0)
                break;
        }
        play(pick);
        S->plies++;
        if (nodes[0].result != 0)
            break;
    }
    S->games++;
}

void Worker::main()
{
    try {
        while (S->keep.load())
            game();
    } catch (StopWorking &) {
    }
}

}  // namespace

extern "C" {

// launch_threads (:683-706): 2 * buffer_entries game threads
int cb_launch(int visits, int buffer_entries, int thread_count, uint64_t seed)
{
    if (G || buffer_entries <= 0 || thread_count != 2 * buffer_entries)
        return -1;  // exactly two buffers' worth of game threads (accelerated_generate_games.py:36): a full buffer
                    // can then never be written again before the host has completed it
    G = new Shared();
    G->B = buffer_entries;
    G->threads = thread_count;
    G->visits = visits;
    for (int k = 0; k < 2; k++) {
        G->boards[k].assign((size_t)2 * buffer_entries, 0);
        G->slot_owner[k].assign((size_t)buffer_entries, -1);
    }
    G->keep = true;
    for (int i = 0; i < thread_count; i++) {
        G->workers.emplace_back(new Worker());
        Worker &w = *G->workers.back();
        w.S = G;
        w.id = i;
        w.rng.seed(seed * 1000003ULL + (uint64_t)i);
    }
    for (auto &w : G->workers)
        w->t = std::thread(&Worker::main, w.get());
    return 0;
}

// get_workload (:708-721): blocks (100 us poll) until a buffer is full; boards_out [B][2] (mover, opponent)
int cb_get_workload(uint64_t *boards_out)
{
    for (;;) {
        {
            std::lock_guard<std::mutex> lock(G->global);
            if (!G->filled.empty()) {
                const int buf = G->filled.front();
                G->filled.pop_front();
                memcpy(boards_out, G->boards[buf].data(), (size_t)G->B * 16);
                return buf;
            }
        }
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
}

// complete_workload (:723-738): logits [B][833], values [B]
void cb_complete_workload(int buf, const float *logits, const float *values)
{
    std::vector<int> owners;
    {
        std::lock_guard<std::mutex> lock(G->global);
        owners = G->slot_owner[buf];
        G->fill[buf] = 0;
    }
    for (int i = 0; i < G->B; i++) {
        Worker &w = *G->workers[(size_t)owners[(size_t)i]];
        std::lock_guard<std::mutex> lk(w.m);
        memcpy(w.logits, logits + (size_t)i * POLICY, POLICY * sizeof(float));
        w.value = values[i];
        w.filled = true;
        w.cv.notify_one();
    }
}

void cb_stats(long long *out4)
{
    out4[0] = G->steps.load();
    out4[1] = G->evals.load();
    out4[2] = G->plies.load();
    out4[3] = G->games.load();
}

void cb_shutdown(void)
{
    if (!G)
        return;
    G->keep = false;
    for (auto &w : G->workers)
        if (w->t.joinable())
            w->t.join();
    delete G;
    G = nullptr;
}
}
