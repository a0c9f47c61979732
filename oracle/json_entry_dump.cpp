// ORACLE — test infrastructure only (tests/test_json_format.py); never linked into or called by the product.
//
// The game line as the reference writes it, produced by the library the reference uses: nlohmann::json (the reference
// includes an un-vendored "json.hpp", cpp/self_play_client.cpp:18; this image carries version 3.1.1 under
// /opt/conda/include — a third-party header, not part of the reference).  The entry is BUILT the way
// generate_game builds it (cpp/self_play_client.cpp:512 `{{"boards", {}}, {"moves", {}}, {"dists", {}}}`, :517
// `entry["random_ply"]`, :565-572 push_back of the 49-int board / the move string / an object filled with
// `entry["dists"].back()[move_string] = edge_visits (double) / all_edge_visits (int)`, :578 `entry["result"]`) and
// WRITTEN the way Worker::thread_main writes it (:639-641 `stream << game << "\n"`), so key order, separators and the
// lay-out of every number are the library's own.
//
// Input (stdin, binary): finished-game records in the engine's ring format, one after the other — 8 u32 header
// {magic "AZHG", slot, uid, plies, result, words, random_ply + 1, kind}, then per ply {x lo, x hi, o lo, o hi,
// move | nd << 16, 0, nd x (move | visits << 16)}; move = from | to << 8, square = file + 7 * rank.  Output: one line per record.
//
// Build (tests do it on demand):  g++ -O1 -std=c++17 -I/opt/conda/include -o _build/json_entry_dump json_entry_dump.cpp
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <string>
#include <vector>

#include "json.hpp"

using json = nlohmann::json;

// cpp/move.cpp:11-21: a clone (from == to) is written as its destination square, a jump as both squares
static std::string square_name(int sq) { return std::string(1, (char)('a' + sq % 7)) + std::string(1, (char)('1' + sq / 7)); }
static std::string move_name(uint32_t mv)
{
    const int from = mv & 0xFF, to = (mv >> 8) & 0xFF;
    return from == to ? square_name(to) : square_name(from) + square_name(to);
}

int main()
{
    std::vector<uint32_t> words;
    uint32_t w;
    while (fread(&w, 4, 1, stdin) == 1)
        words.push_back(w);
    size_t pos = 0;
    while (pos + 8 <= words.size()) {
        const uint32_t *rec = words.data() + pos;
        if (rec[0] != 0x415A4847u || rec[5] < 8 || pos + rec[5] > words.size()) {
            fprintf(stderr, "json_entry_dump: bad record at word %zu\n", pos);
            return 2;
        }
        json entry = {{"boards", {}}, {"moves", {}}, {"dists", {}}};
        if (rec[6])
            entry["random_ply"] = (int)rec[6] - 1;
        size_t p = 8;
        for (uint32_t ply = 0; ply < rec[3]; ply++) {
            const uint64_t x = (uint64_t)rec[p] | ((uint64_t)rec[p + 1] << 32), o = (uint64_t)rec[p + 2] | ((uint64_t)rec[p + 3] << 32);
            const uint32_t nd = rec[p + 4] >> 16;
            std::vector<int> board;   // serialize_board_for_json, cpp/self_play_client.cpp:88-107
            for (int y = 0; y < 7; y++)
                for (int xx = 0; xx < 7; xx++) {
                    const uint64_t mask = 1ull << (xx + 7 * (6 - y));
                    board.push_back((x & mask) ? 1 : ((o & mask) ? 2 : 0));
                }
            entry["boards"].push_back(board);
            entry["moves"].push_back(move_name(rec[p + 4] & 0xFFFFu));
            entry["dists"].push_back(json::object());
            int all_edge_visits = 0;
            for (uint32_t j = 0; j < nd; j++)
                all_edge_visits += (int)(rec[p + 6 + j] >> 16);
            for (uint32_t j = 0; j < nd; j++) {
                const double edge_visits = (double)(rec[p + 6 + j] >> 16);
                const double weight = all_edge_visits ? edge_visits / all_edge_visits : 0.0;
                entry["dists"].back()[move_name(rec[p + 6 + j] & 0xFFFFu)] = weight;
            }
            p += 6 + nd;
        }
        entry["result"] = (int)rec[4];
        std::cout << entry << "\n";
        pos += rec[5];
    }
    return 0;
}
