// The reference's four-symbol ctypes ABI (link.py:6-32 <-> cpp/self_play_client.cpp:683-749)
// served by the GPU engine instead of worker threads.
//
// Reference behaviour: `thread_count` worker threads each play one game; a worker that
// needs an evaluation copies its (7,7,4) feature row into the current fill buffer
// (request_evaluation :648-681); a full buffer of `buffer_entries` rows is handed to the
// host by get_workload (:708-721); complete_workload (:723-738) copies each row's 833
// logits + value back to its worker.  Finished games are appended to `output_path` as
// JSON lines and flushed (Worker::thread_main :637-642).
//
// Here the `thread_count` games are GPU game slots.  One search iteration yields at most
// one leaf per game; the leaves are dealt, in game order, into the two caller-owned
// buffers (rows past the leaf count are zero).  When the host has completed both
// workloads the iteration is backed up on the GPU, finished games are appended to the
// file, and the next iteration's leaves are selected.  The caller-visible contract —
// buffer ownership, row layout, synchronous copy in complete_workload, append + flush —
// is the reference's; get_workload never has to wait because selection is synchronous.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <deque>
#include <random>
#include <vector>

#define AZH_NO_REFERENCE_ABI
#include "../../include/ataxxzero_hip.h"

namespace {

struct RefState {
    azh_engine *eng = nullptr;
    FILE *out = nullptr;
    float *bufs[2] = {nullptr, nullptr};
    int B = 0, G = 0;
    int n_leaves = 0;
    std::vector<int32_t> leaf_games;
    std::vector<float> feat;
    std::vector<float> logits, values;
    std::deque<int> ready;
    int outstanding = 0;
    std::vector<char> json;
};

RefState R;

void die(const char *what)
{
    fprintf(stderr, "ataxxzero_hip reference ABI: %s: %s\n", what, azh_last_error());
    abort();  // the reference's failure mode is assert/abort (no return codes in this ABI)
}

void write_finished_games()
{
    for (;;) {
        int64_t used = 0;
        int32_t n = 0;
        if (azh_engine_drain_json(R.eng, R.json.data(), (int64_t)R.json.size(), &used, &n)) {
            if (R.json.size() > ((size_t)1 << 30))
                die("drain_json");
            R.json.resize(R.json.size() * 2);  // one game line did not fit
            continue;
        }
        if (n == 0)
            break;
        if (R.out) {
            fwrite(R.json.data(), 1, (size_t)used, R.out);
            fflush(R.out);
        }
    }
}

void next_iteration()
{
    int32_t n = 0;
    if (azh_engine_select(R.eng, &n))
        die("select");
    R.n_leaves = n;
    if (n > 0 && azh_engine_leaf_features(R.eng, R.feat.data(), R.leaf_games.data()))
        die("leaf_features");
    for (int w = 0; w < 2; w++) {
        const int lo = w * R.B;
        const int rows = n - lo < 0 ? 0 : (n - lo > R.B ? R.B : n - lo);
        memset(R.bufs[w], 0, (size_t)R.B * AZH_FEATURE_SIZE * sizeof(float));
        if (rows > 0)
            memcpy(R.bufs[w], R.feat.data() + (size_t)lo * AZH_FEATURE_SIZE, (size_t)rows * AZH_FEATURE_SIZE * sizeof(float));
    }
    R.ready.clear();
    R.ready.push_back(0);
    R.ready.push_back(1);
    R.outstanding = 2;
}

}  // namespace

extern "C" void launch_threads(char *output_path, int visits, float *fill_buffer1, float *fill_buffer2,
                               int buffer_entries, int thread_count)
{
    if (R.eng) {
        fprintf(stderr, "ataxxzero_hip reference ABI: launch_threads called twice without shutdown\n");
        abort();
    }
    if (buffer_entries <= 0 || thread_count <= 0 || thread_count > 2 * buffer_entries) {
        fprintf(stderr, "ataxxzero_hip reference ABI: need 0 < thread_count <= 2 * buffer_entries (got %d, %d)\n",
                thread_count, buffer_entries);
        abort();
    }
    azh_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.games = thread_count;
    cfg.visits = visits;
    cfg.max_plies = 400;              // maximum_game_plies (:34)
    cfg.edges_per_node = 96;
    cfg.c_puct = 1.0f;                // (:31)
    cfg.dirichlet_alpha = 0.15f;      // (:32)
    cfg.dirichlet_weight = 0.25f;     // (:33)
    cfg.start_turn = 0;
    // the reference seeds its generator from std::random_device at every launch (:39-40), so a restarted generator
    // never replays its games; AZH_SEED pins the Philox seed for reproducible runs
    const char *seed = getenv("AZH_SEED");
    if (seed) {
        cfg.seed = strtoull(seed, nullptr, 10);
    } else {
        std::random_device rd;
        cfg.seed = ((uint64_t)rd() << 32) ^ (uint64_t)rd();
    }
    // STARTING_GAME_POSITION "x5o/7/3-3/2-1-2/3-3/7/o5x x" (:23)
    cfg.start_x = (1ULL << 42) | (1ULL << 6);
    cfg.start_o = (1ULL << 48) | (1ULL << 0);
    cfg.blockers = (1ULL << 31) | (1ULL << 23) | (1ULL << 25) | (1ULL << 17);
    if (azh_engine_create(&cfg, &R.eng))
        die("engine_create");
    printf("Launching into %p, %p with %d entries and %d GPU game slots.\n", (void *)fill_buffer1,
           (void *)fill_buffer2, buffer_entries, thread_count);
    printf("Writing to: %s\n", output_path);
    R.out = fopen(output_path, "a");  // std::ios_base::app (:691)
    if (!R.out) {
        fprintf(stderr, "ataxxzero_hip reference ABI: cannot open %s\n", output_path);
        abort();
    }
    R.bufs[0] = fill_buffer1;
    R.bufs[1] = fill_buffer2;
    R.B = buffer_entries;
    R.G = thread_count;
    R.leaf_games.assign((size_t)R.G, 0);
    R.feat.assign((size_t)R.G * AZH_FEATURE_SIZE, 0.0f);
    R.logits.assign((size_t)R.G * AZH_POLICY_SIZE, 0.0f);
    R.values.assign((size_t)R.G, 0.0f);
    R.json.assign(1 << 22, 0);
    next_iteration();
}

extern "C" int get_workload(void)
{
    if (!R.eng || R.ready.empty()) {
        fprintf(stderr, "ataxxzero_hip reference ABI: get_workload with no workload pending "
                        "(complete the outstanding workloads first)\n");
        abort();
    }
    const int w = R.ready.front();
    R.ready.pop_front();
    return w;
}

extern "C" void complete_workload(int workload, float *posteriors, float *values)
{
    if (!R.eng || workload < 0 || workload > 1 || R.outstanding <= 0) {
        fprintf(stderr, "ataxxzero_hip reference ABI: unexpected complete_workload(%d)\n", workload);
        abort();
    }
    const int lo = workload * R.B;
    const int rows = R.n_leaves - lo < 0 ? 0 : (R.n_leaves - lo > R.B ? R.B : R.n_leaves - lo);
    for (int i = 0; i < rows; i++) {
        const int g = R.leaf_games[(size_t)lo + i];
        memcpy(&R.logits[(size_t)g * AZH_POLICY_SIZE], posteriors + (size_t)i * AZH_POLICY_SIZE,
               AZH_POLICY_SIZE * sizeof(float));
        R.values[(size_t)g] = values[i];
    }
    if (--R.outstanding > 0)
        return;
    if (azh_engine_set_evals(R.eng, R.logits.data(), R.values.data()))
        die("set_evals");
    if (azh_engine_backup(R.eng))
        die("backup");
    write_finished_games();
    next_iteration();
}

extern "C" void shutdown(void)
{
    if (R.eng) {
        azh_engine_destroy(R.eng);
        R.eng = nullptr;
    }
    if (R.out) {
        fclose(R.out);
        R.out = nullptr;
    }
    R.ready.clear();
    R.outstanding = 0;
}
